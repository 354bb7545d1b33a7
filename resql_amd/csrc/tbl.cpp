// tbl.cpp — '.tbl' text ingest straight into columns, with the semantics of the reference's BULK INSERT
// (reference src/execute.h:332-388): one tuple per line, fields split at a one-character terminator the way
// std::getline splits them (a trailing terminator yields no extra field, an empty line has no fields), every field
// parsed by the constant parser of its column's type CATEGORY (reference src/expressions.h:369-515) and stored with
// ValueMoves::toAddress (reference src/values.h:151-198).  Consequences kept on purpose:
//   * DECIMAL fields are "the digits with the first '.' removed" — the column's scale is not consulted;
//   * BIGINT fields pass through an int32_t (expressions.h:369-373);
//   * DATE accepts yyyy-mm-dd and yyyy/mm/dd (the two sscanf formats that can succeed, expressions.h:413-440);
//   * CHAR(n)/VARCHAR(n) keep at most n characters (ValueMoves::writeString).
// The numeric conversions call the same libc routines the reference reaches through std::stoll / std::stoi / sscanf.
//
// This is ingest, not the hot path, but SF10 lineitem is 7.5 GB of text: the file is split at line boundaries and
// parsed by all host threads into preallocated columns.
#include <cerrno>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <thread>

#include "engine.h"

namespace rsq {

namespace {

struct ParseError { int64_t line = -1; std::string what; };

struct Chunk { size_t begin = 0, end = 0; int64_t firstLine = 0, nLines = 0; };

// std::stoll(token): strtoll base 10, at least one digit, no overflow; trailing characters are ignored
bool stollLike(const char* s, long long* out) {
    errno = 0;
    char* endp = nullptr;
    long long v = strtoll(s, &endp, 10);
    if (endp == s || errno == ERANGE) return false;
    *out = v;
    return true;
}

// parse one field into the column cell; returns nullptr or a static error text
const char* parseField(const Type& t, char* tok /* NUL terminated, writable */, size_t len, uint8_t* cell) {
    switch (t.tag) {
        case RSQ_DECIMAL: {
            // parseDecimalConstant: the `sym == "-"` test only matches a lone minus sign (then nothing is left to
            // parse); the first '.' is removed and the rest goes through stoll, which handles a leading '-' itself
            if (len == 1 && tok[0] == '-') return "stoll: no conversion";
            char* dot = (char*)memchr(tok, '.', len);
            if (dot) memmove(dot, dot + 1, len - (size_t)(dot - tok));      // moves the NUL too
            long long v;
            if (!stollLike(tok, &v)) return "stoll: no conversion";
            int64_t x = v; memcpy(cell, &x, 8);
            return nullptr;
        }
        case RSQ_BIGINT: {
            long long v;
            if (!stollLike(tok, &v)) return "stoll: no conversion";
            int64_t x = (int64_t)(int32_t)v; memcpy(cell, &x, 8);           // int32_t v = std::stoll(..)
            return nullptr;
        }
        case RSQ_INT: {
            errno = 0;
            char* endp = nullptr;
            long v = strtol(tok, &endp, 10);
            if (endp == tok || errno == ERANGE || v < INT_MIN || v > INT_MAX) return "stoi: no conversion";
            int32_t x = (int32_t)v; memcpy(cell, &x, 4);
            return nullptr;
        }
        case RSQ_DATE: {
            bool success = false;
            int year = 0, day = 0, month = 0;
            if (sscanf(tok, "%4d-%2d-%2d", &year, &month, &day) == 3) success = true;
            if (sscanf(tok, "%4d/%2d/%2d", &year, &month, &day) == 3) success = true;
            if (!success) return "Unsupported string type or unsupported date format (formats: \"yyyy/mm/dd\", \"mm/dd/yyyy\")";
            uint32_t x = (uint32_t)(year * 10000 + month * 100 + day); memcpy(cell, &x, 4);
            return nullptr;
        }
        case RSQ_BOOL:
            if (strcmp(tok, "true") == 0) cell[0] = 1;
            else if (strcmp(tok, "false") == 0) cell[0] = 0;
            else return "Couldnt parse BOOL constant.";
            return nullptr;
        case RSQ_CHAR: case RSQ_VARCHAR: {
            const size_t w = (size_t)columnWidth(t);           // cells are zero-initialised: NUL padded
            for (size_t i = 0; i < w && i < len && tok[i]; i++) cell[i] = (uint8_t)tok[i];
            return nullptr;
        }
        default: return "parseConstant(..) not implemented for type.";
    }
}

}  // namespace

// -> columns (columnWidth(type) bytes per row) and the row count
void parseTblFile(const std::string& path, const std::vector<Type>& types, char terminator, int nThreads,
                  std::vector<std::vector<uint8_t>>& cols, int64_t& nRows) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f.is_open()) throw Error(RSQ_ERR_INVALID, "Could not open file " + path);
    const size_t size = (size_t)f.tellg();
    std::vector<char> buf(size + 1);
    f.seekg(0);
    if (size && !f.read(buf.data(), (std::streamsize)size)) throw Error(RSQ_ERR_INVALID, "Could not read file " + path);
    buf[size] = '\0';

    // ---- split at line boundaries, count lines per chunk ----
    if (nThreads < 1) nThreads = 1;
    if (size < (1u << 20)) nThreads = 1;
    std::vector<Chunk> chunks((size_t)nThreads);
    size_t pos = 0;
    for (int c = 0; c < nThreads; c++) {
        chunks[(size_t)c].begin = pos;
        size_t target = c == nThreads - 1 ? size : std::max(pos, size / (size_t)nThreads * (size_t)(c + 1));
        if (target < size) {
            const char* nl = (const char*)memchr(buf.data() + target, '\n', size - target);
            target = nl ? (size_t)(nl - buf.data()) + 1 : size;
        }
        chunks[(size_t)c].end = pos = target;
    }
    auto countLines = [&](Chunk& ch) {
        int64_t n = 0;
        const char* p = buf.data() + ch.begin; const char* e = buf.data() + ch.end;
        while (p < e) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
            n++;                                             // std::getline: a last line without '\n' still counts
            if (!nl) break;
            p = nl + 1;
        }
        ch.nLines = n;
    };
    {
        std::vector<std::thread> th;
        for (auto& ch : chunks) th.emplace_back(countLines, std::ref(ch));
        for (auto& t : th) t.join();
    }
    nRows = 0;
    for (auto& ch : chunks) { ch.firstLine = nRows; nRows += ch.nLines; }

    cols.assign(types.size(), {});
    std::vector<int> width(types.size());
    for (size_t i = 0; i < types.size(); i++) {
        width[i] = columnWidth(types[i]);
        cols[i].assign((size_t)nRows * (size_t)width[i], 0);
    }

    // ---- parse ----
    std::mutex errLock;
    ParseError err;
    auto fail = [&](int64_t line, const std::string& what) {
        std::lock_guard<std::mutex> g(errLock);
        if (err.line < 0 || line < err.line) { err.line = line; err.what = what; }
    };
    auto parseChunk = [&](const Chunk& ch) {
        char* p = buf.data() + ch.begin; char* e = buf.data() + ch.end;
        int64_t row = ch.firstLine;
        while (p < e) {
            char* nl = (char*)memchr(p, '\n', (size_t)(e - p));
            char* lineEnd = nl ? nl : e;
            size_t att = 0;
            bool bad = false;
            char* q = p;
            while (q < lineEnd) {                            // std::getline(lineStream, token, terminator)
                char* d = (char*)memchr(q, terminator, (size_t)(lineEnd - q));
                char* tokEnd = d ? d : lineEnd;
                if (att >= types.size()) { fail(row, "contains extra attributes."); bad = true; break; }
                const char saved = *tokEnd;
                *tokEnd = '\0';
                const char* what = parseField(types[att], q, (size_t)(tokEnd - q), &cols[att][(size_t)row * (size_t)width[att]]);
                *tokEnd = saved;
                if (what) { fail(row, std::string("field ") + std::to_string(att + 1) + ": " + what); bad = true; break; }
                att++;
                q = d ? d + 1 : lineEnd;
            }
            if (!bad && att < types.size()) fail(row, "is missing attributes.");
            row++;
            if (!nl) break;
            p = nl + 1;
        }
    };
    {
        std::vector<std::thread> th;
        for (auto& ch : chunks) th.emplace_back(parseChunk, std::cref(ch));
        for (auto& t : th) t.join();
    }
    if (err.line >= 0)
        throw Error(RSQ_ERR_INVALID, "Line " + std::to_string(err.line) + " in " + path + " " + err.what);
}

}  // namespace rsq
