// codegen.cpp — operator tree -> device pipelines (HIP source per pipeline).
//
// Mirrors the reference's produce/consume code generation (reference src/operators/*.h driven from
// src/execute.h:228): produce() walks down to the scans; every scan opens a pipeline (scan.h:227-263)
// and the operators above it consume() into the body of that pipeline's row function until a
// pipeline breaker ends it — hash-join build (hashjoin.h:226-256) or aggregation
// (aggregation.h:240-295).  The hand-written skeleton around the row function (tile loads,
// reductions, hash-table access) comes from kernels/rsq_device.h.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <set>
#include <sstream>

#include "engine_internal.h"

namespace rsq {

namespace {

struct Sym { std::string var; Type type; };

int envInt(const char* name, int def, int lo, int hi) {
    const char* e = getenv(name);
    int v = e ? atoi(e) : def;
    return v < lo ? lo : v > hi ? hi : v;
}

int64_t nextPow2(int64_t v) { int64_t p = 1; while (p < v) p <<= 1; return p; }

// ================================================================================================
// expressions -> device code (emitExpression, reference src/ExpressionsJitFlounder.h:1080-1114)
// ================================================================================================
struct ExprGen {
    std::map<std::string, Sym> symbols;     // JitContextFlounder::symbolTable of the current pipeline
    std::map<std::string, int> strWordVars; // string columns whose first words arrive as row-function parameters <var>_w0, _w1 (loaded with the tile): how many
    bool int16Cast = false;                 // rsq_config.compat_flags & RSQ_COMPAT_JIT_INT16_CAST

    static std::string ctype(const Type& t) {
        switch (t.tag) {
            case RSQ_BIGINT: case RSQ_DECIMAL: return "i64";
            case RSQ_INT: case RSQ_DATE: return "i32";
            case RSQ_BOOL: return "u8";
            case RSQ_CHAR: return t.len == 1 ? "u8" : "rsq::Str";
            case RSQ_VARCHAR: return "rsq::Str";
            default: failType("no device type for " + serializeType(t));
        }
    }
    static std::string lit64(int64_t v) {
        if (v == INT64_MIN) return "((i64)0x8000000000000000ull)";
        return "((i64)" + std::to_string((long long)v) + "ll)";
    }
    static std::string cstring(const std::string& s) {
        std::string o = "\"";
        for (unsigned char c : s) {
            char buf[8];
            if (c == '"' || c == '\\') { o += '\\'; o += (char)c; }
            else if (c < 32 || c > 126) { snprintf(buf, sizeof buf, "\\%03o", c); o += buf; }
            else o += (char)c;
        }
        return o + "\"";
    }
    static int64_t pow10(int n) { int64_t v = 1; while (n-- > 0) v *= 10; return v; }

    std::string constant(const Expr* e) {
        switch (e->type.tag) {
            case RSQ_DECIMAL: case RSQ_BIGINT: return lit64(e->ival);
            case RSQ_INT: case RSQ_DATE: return "((i32)" + std::to_string((long long)(int32_t)e->ival) + ")";
            case RSQ_BOOL: return "((u8)" + std::to_string((int)(uint8_t)e->ival) + ")";
            case RSQ_CHAR:
                if (e->type.len == 1) return "((u8)" + std::to_string((int)(uint8_t)e->ival) + ")";
                [[fallthrough]];
            case RSQ_VARCHAR: return "rsq::str(" + cstring(e->symbol) + ", " + std::to_string(e->type.len) + ")";
            default: failType("Constant code generation not implemented for datatype");
        }
    }

    std::string emit(Expr* e) {
        if (e->type.tag == RSQ_NT) failType("Expression type undefined in emitExpression(..). Have you derived the expression types?");
        auto it = symbols.find(expressionName(e));
        if (it != symbols.end()) return it->second.var;          // value already available under this name
        switch (e->structure) {
            case LITERAL:
                if (e->tag == RSQ_E_ATTRIBUTE) failType("attribute " + e->symbol + " is not available in this pipeline");
                if (e->tag == RSQ_E_CONSTANT) return constant(e);
                if (e->tag == RSQ_E_STAR) return "((i64)0)";
                failType(std::string("emitExpressionLiteral(..) not implemented for expression type") + exprTagNames[e->tag]);
            case UNARY: return emitUnary(e);
            case BINARY: return emitBinary(e);
            case OTHER: return emitCase(e);
            default: failType("emitExpression(..)");
        }
    }

    std::string emitUnary(Expr* e) {
        if (e->tag == RSQ_E_COUNT) {
            // emitCount: every row counts.  The reference still emits the argument first (emitExpressionUnary,
            // ExpressionsJitFlounder.h:925-928), so an argument it cannot compile refuses the whole plan: emit it for
            // its checks and drop the text.
            if (e->child && e->child->tag != RSQ_E_STAR) (void)emit(e->child);
            return "((i64)1)";
        }
        std::string c = emit(e->child);
        const Type from = e->child->type, to = e->type;
        switch (e->tag) {
            case RSQ_E_SUM: case RSQ_E_AVG: case RSQ_E_MIN: case RSQ_E_MAX: case RSQ_E_AS: return c;
            case RSQ_E_TYPECAST:
                if (to.tag == RSQ_DECIMAL) {
                    if (from.tag == RSQ_DECIMAL) {
                        if (to.scale == from.scale) return c;
                        int d = to.scale - from.scale;
                        if (d > 8 || d < -8) failType("typecast beyond the supported scale difference");
                        if (d > 0) return "rsq::mul(" + c + ", " + lit64(pow10(d)) + ")";
                        return "((i64)((" + c + ") / " + lit64(pow10(-d)) + "))";
                    }
                    if (from.tag == RSQ_BIGINT) {
                        if (to.scale > 8) failType("typecast beyond the supported scale");
                        return "rsq::mul(" + c + ", " + lit64(pow10(to.scale)) + ")";
                    }
                    failType("emitTypecastToDECIMAL(..) code generation not implemented for datatype");
                }
                if (to.tag == RSQ_BIGINT) {
                    // INT -> BIGINT is a 32 -> 64 sign extension (ExpressionsJitFlounder.h:818-824 `movsx`).  The reference's
                    // asmjit back end encodes the 16-bit movsx for it, so its JIT extends the low 16
                    // bits; rsq_config.compat_flags & RSQ_COMPAT_JIT_INT16_CAST reproduces exactly that for hosts that need the JIT's answers.
                    if (from.tag == RSQ_INT) return int16Cast ? "((i64)(short)(" + c + "))" : "((i64)(" + c + "))";
                    if (from.tag == RSQ_DECIMAL) {
                        if (from.scale > 8) failType("typecast beyond the supported scale");
                        return "((i64)((" + c + ") / " + lit64(pow10(from.scale)) + "))";
                    }
                    if (from.tag == RSQ_BIGINT) return c;
                    failType("emitTypecastToBIGINT(..) code generation not implemented for datatype");
                }
                failType("emitTypecast(..) code generation not implemented for datatype");
            default:
                failType(std::string("emitExpression(..) not implemented for expression type") + exprTagNames[e->tag]);
        }
    }

    std::string emitBinary(Expr* e) {
        std::string l = emit(e->child), r = emit(e->child->next);
        const Type res = e->type, op = e->child->type;
        auto arithOk = [&]() {
            if (res.tag != RSQ_DECIMAL && res.tag != RSQ_BIGINT)
                failType(std::string(exprTagNames[e->tag]) + " code generation not implemented for datatype");
        };
        auto orderedOk = [&]() {
            if (op.tag != RSQ_DECIMAL && op.tag != RSQ_DATE && op.tag != RSQ_BIGINT)
                failType(std::string(exprTagNames[e->tag]) + " code generation not implemented for datatype");
        };
        // string = constant: word-wise against the constant's bytes (rsq_device.h ld_bytes) instead of the byte loop
        auto equalsConstant = [&](bool charSemantics) -> std::string {
            Expr* lc = e->child; Expr* rc = e->child->next;
            const bool lConst = lc->tag == RSQ_E_CONSTANT, rConst = rc->tag == RSQ_E_CONSTANT;
            if (lConst == rConst || !lc->type.isString() || !rc->type.isString() || 1 == 0) return "";
            const Expr* cst = lConst ? lc : rc;
            const std::string& x = lConst ? r : l;
            const int cap = (lConst ? rc : lc)->type.len;
            std::string text = cst->symbol;
            if (text.find('\0') != std::string::npos) return "";
            if (charSemantics) while (!text.empty() && text.back() == ' ') text.pop_back();
            if ((int)text.size() > cap) return "((u8)0)";            // longer than any value of the column
            std::string cond, condRest;          // (condRest: the words behind the prefetched ones - fetched only if those match)
            const int nPre = strWordVars.count(x) ? strWordVars[x] : 0;
            for (int w = 0; w * 8 < cap; w++) {
                const int rbytes = std::min(8, cap - w * 8);
                uint64_t cw = 0, mask = 0;
                for (int i = 0; i < rbytes; i++) {
                    const size_t k = (size_t)(w * 8 + i);
                    if (k < text.size()) { cw |= (uint64_t)(uint8_t)text[k] << (8 * i); mask |= 0xFFull << (8 * i); }
                    else mask |= (charSemantics ? 0xDFull : 0xFFull) << (8 * i);
                }
                char buf[200];
                if (w < nPre) snprintf(buf, sizeof buf, "((%s_w%d ^ 0x%llxull) & 0x%llxull)", x.c_str(), w, (unsigned long long)cw, (unsigned long long)mask);
                else
                snprintf(buf, sizeof buf, "((rsq::ld_bytes<%d>((%s).p + %d) ^ 0x%llxull) & 0x%llxull)", rbytes, x.c_str(), w * 8,
                         (unsigned long long)cw, (unsigned long long)mask);
                std::string& into = nPre > 0 && w >= nPre ? condRest : cond;
                into += (into.empty() ? "" : " | ") + std::string(buf);
            }
            if (!condRest.empty()) return "((u8)(((" + cond + ") == 0ull) && ((" + condRest + ") == 0ull)))";
            return "((u8)((" + cond + ") == 0ull))";
        };
        auto equals = [&]() -> std::string {
            if (op.tag == RSQ_VARCHAR || (op.tag == RSQ_CHAR && op.len > 1)) {
                const std::string fast = equalsConstant(op.tag == RSQ_CHAR);
                if (!fast.empty()) return fast;
            }
            switch (op.tag) {
                case RSQ_DECIMAL: case RSQ_INT: case RSQ_BIGINT: case RSQ_BOOL: case RSQ_DATE:
                    return "((u8)((" + l + ") == (" + r + ")))";
                case RSQ_CHAR:
                    if (op.len > 1) return "rsq::compare_char(" + l + ", " + r + ")";
                    return "((u8)((" + l + ") == (" + r + ")))";
                case RSQ_VARCHAR: return "rsq::compare_varchar(" + l + ", " + r + ")";
                default: failType("EQUALS code generation not implemented for datatype");
            }
        };
        switch (e->tag) {
            case RSQ_E_ADD: arithOk(); return "rsq::add(" + l + ", " + r + ")";
            case RSQ_E_SUB: arithOk(); return "rsq::sub(" + l + ", " + r + ")";
            case RSQ_E_MUL: arithOk(); return "rsq::mul(" + l + ", " + r + ")";
            case RSQ_E_DIV: arithOk(); return "rsq::div(" + l + ", " + r + ", a.err)";
            case RSQ_E_AND: return "((u8)((" + l + ") & (" + r + ")))";      // no short circuit, as in the reference
            case RSQ_E_OR: return "((u8)((" + l + ") | (" + r + ")))";
            case RSQ_E_LT: orderedOk(); return "((u8)((" + l + ") < (" + r + ")))";
            case RSQ_E_LE: orderedOk(); return "((u8)((" + l + ") <= (" + r + ")))";
            case RSQ_E_GT: orderedOk(); return "((u8)((" + l + ") > (" + r + ")))";
            case RSQ_E_GE: orderedOk(); return "((u8)((" + l + ") >= (" + r + ")))";
            case RSQ_E_EQ: return equals();
            case RSQ_E_NEQ: return "((u8)(1 - " + equals() + "))";
            case RSQ_E_LIKE: {
                // emitLike passes both operands to stringLikeCheck as char* (ExpressionsJitFlounder.h:695-705): a CHAR(1)
                // operand is a byte there, not a pointer — undefined in the reference, refused here
                const Type rt = e->child->next->type;
                if (!op.isString() || !rt.isString()) failType("LIKE on a CHAR(1) operand is undefined in the reference");
                return "rsq::like(" + l + ", " + r + ")";
            }
            default: failType(std::string("emitExpressionBinary(..) not implemented for expression type") + exprTagNames[e->tag]);
        }
    }

    std::string emitCase(Expr* e) {   // ExpressionsJitFlounder.h:720-754
        std::string out, close;
        Expr* c = e->child;
        for (; c && c->tag == RSQ_E_WHENTHEN; c = c->next) {
            out += "((" + emit(c->child) + ") ? (" + emit(c->child->next) + ") : ";
            close += ")";
        }
        if (c) out += "(" + emit(c) + ")";
        else out += (e->type.isString() ? std::string("rsq::str(\"\", 0)") : "((" + ctype(e->type) + ")0)");
        return out + close;
    }
};

// value of a 64-bit table word as a typed device value, and back
std::string fromWord(const std::string& w, const Type& t) {
    switch (t.tag) {
        case RSQ_BIGINT: case RSQ_DECIMAL: return w;
        case RSQ_INT: case RSQ_DATE: return "((i32)(" + w + "))";
        case RSQ_BOOL: return "((u8)(" + w + "))";
        case RSQ_CHAR: if (t.len == 1) return "((u8)(" + w + "))"; [[fallthrough]];
        case RSQ_VARCHAR: return "rsq::str_from_addr(" + w + ", " + std::to_string(t.len) + ")";     // payload strings travel by address
        default: failUnsupported("value type cannot be carried in a hash table word");
    }
}
std::string toWord(const std::string& v, const Type& t) {
    // a string is carried as the device address of its bytes in the (immutable, device-resident) column it comes from
    if (t.isString()) return "rsq::str_addr(" + v + ")";
    return "((i64)(" + v + "))";
}

// ================================================================================================
// the walk
// ================================================================================================
struct Walker {
    Query& q;
    ExprGen eg;

    // state of the pipeline under construction
    Pipeline pipe;
    std::vector<std::string> colTypes;        // device type per scanned (vector-loadable) column
    std::vector<int> colIsString;
    std::string rowParams, rowArgsTail, rowArgsTailGuarded;
    // key-bitmap words fetched for both rows of a lane (and all tiles in flight) before the first row is processed:
    // (table name, scanned column index) — see consumeProbe
    struct BitmapPrefetch { std::string first; int second; bool interleaved; };
    std::vector<BitmapPrefetch> bitmapPrefetch;
    std::string body;                          // row function body
    std::string closers;                       // closing braces of the open scopes
    std::string stateDecl, stateInit, prologue, epilogue, fileScope;
    std::string helperFns;                     // device functions behind Args / State, in front of the row function
    std::vector<std::string> explainSteps;
    int indent = 1;
    int matchSlotTable = -1;                   // innermost single-match probe whose slot variable is in scope
    std::map<int, std::string> slotVar;        // hash table id -> device variable holding the matched slot
    std::map<std::string, int> symbolOrigin;   // symbol -> hash table id it was read from (or -1: scan column)
    std::map<std::string, int> symbolWord;     // symbol -> word index in that table
    bool multiMatchAbove = false;
    // wave-level compaction (see compactThen)
    bool selective = false, compacted = false;
    // the selection directly above the scan: its text over the row's column variables, the columns it reads and the fraction of
    // rows it is expected to pass (column statistics, values taken as uniform) - the late-load form of the tile loop (below)
    std::string leadCond; std::vector<int> leadCols; double leadPass = 1.0;
    std::string stage2Prefix;               // necessary conditions of later joins, tested at the top of stage 2 (consumeProbe: component bitmaps)
    struct CompFilter { std::string table, symbol, stage2Var; int64_t bits; };
    std::vector<CompFilter> compFilters;    // ... as consumeProbe found them while stage 2 was generated
    bool inStage2 = false;                  // the walk is generating the code behind the wave compaction
    bool leadPassComplete = true;       // every part of the predicate was understood (else the estimate is an upper bound only)
    std::string stage2Body;
    std::vector<std::pair<std::string, Sym>> cqLive;     // carried symbols: name -> stage-1 variable and type

    explicit Walker(Query& q_) : q(q_) { eg.int16Cast = jitInt16Cast(q_.ctx); }

    void line(const std::string& s) { body += std::string((size_t)indent * 4, ' ') + s + "\n"; }
    void openScope(const std::string& head) { line(head); indent++; }
    void closeScope() { indent--; line("}"); }
    void addArg(const std::string& name, const std::string& ctype, uint64_t v) {
        for (auto& a : pipe.args) if (a.name == name) return;
        pipe.args.push_back({name, ctype, v});
    }

    // -------------------------------------------------------------------------------------------
    void produce(OpNode* o, std::vector<std::string> request) {
        switch (o->tag) {
            case RSQ_OP_SCAN: produceScan(o, request); break;
            case RSQ_OP_SELECTION: {                    // selection.h:39-49
                o->schema.clear();
                std::vector<std::string> r = request;
                requiredAttributes(o->exprs[0], r);
                requestOf[o] = request;
                produce(o->child[0], r);
                break;
            }
            case RSQ_OP_PROJECTION: {                   // projection.h:40-59
                std::vector<std::string> r;
                for (Expr* e : o->exprs) requiredAttributes(e, r);
                produce(o->child[0], r);
                break;
            }
            case RSQ_OP_HASHJOIN: {                     // hashjoin.h:98-116
                requestOf[o] = request;
                std::vector<std::string> all = request;
                for (Expr* e : o->exprs) requiredAttributes(e, all);
                joinPhase[o] = 1;
                produce(o->child[0], all);              // build pipeline(s)
                joinPhase[o] = 2;
                produce(o->child[1], all);              // probe pipeline(s)
                break;
            }
            case RSQ_OP_AGGREGATION: {                  // aggregation.h:155-164
                if (q.agg) failUnsupported("more than one aggregation in a plan");
                q.agg = o;
                std::vector<std::string> r;
                for (Expr* e : o->exprs) requiredAttributes(e, r);
                for (Expr* e : o->exprs2) requiredAttributes(e, r);
                produce(o->child[0], r);
                break;                                   // everything above runs on the host (tail.cpp)
            }
            case RSQ_OP_MATERIALIZE: case RSQ_OP_ORDERBY:
                produce(o->child[0], request);
                break;
            default: failUnsupported("operator not supported by the GPU engine");
        }
    }

    std::map<OpNode*, std::vector<std::string>> requestOf;
    std::map<OpNode*, int> joinPhase;

    static bool has(const std::vector<std::string>& v, const std::string& s) { return std::find(v.begin(), v.end(), s) != v.end(); }

    Schema prune(const Schema& s, const std::vector<std::string>& req) {
        Schema r;
        for (auto& a : s) if (has(req, a.name)) r.push_back(a);
        return r;
    }

    // -------------------------------------------------------------------------------------------
    void produceScan(OpNode* o, std::vector<std::string> request) {     // scan.h:221-263
        Table* t = o->table;
        if (q.requestAll) request.clear();
        pipe = Pipeline();
        pipe.src = t;
        // tuned on MI355X with TPC-H Q1 SF10 (profiles/): 1 tile in flight per wave + non-temporal loads
        // 0.395 ms; 2 tiles 0.42 ms; 4 tiles 0.47 ms (fewer resident waves); without nt loads 0.45-0.47 ms
        pipe.unroll = 1;
        pipe.blockThreads = 256;
        // workgroups (of 256 threads) per launch; 0 = 2 per CU.  Measured on MI355X (Q1 SF10): 512 workgroups 0.348 ms,
        // 768: 0.367, 1024: 0.374, 2048: 0.395, 4096: 0.448 - a streaming kernel wants exactly 2 resident workgroups per CU
        pipe.maxGrid = (unsigned)0;
        colTypes.clear(); colIsString.clear(); rowParams.clear(); rowArgsTail.clear(); rowArgsTailGuarded.clear(); bitmapPrefetch.clear();
        body.clear(); stateDecl.clear(); stateInit.clear(); prologue.clear(); epilogue.clear(); fileScope.clear(); helperFns.clear();
        explainSteps.clear(); indent = 1; matchSlotTable = -1; slotVar.clear(); symbolOrigin.clear(); symbolWord.clear();
        multiMatchAbove = false;
        selective = false; compacted = false; stage2Body.clear(); cqLive.clear();
        leadCond.clear(); leadCols.clear(); leadPass = 1.0; leadPassComplete = true;
        strPrefetch.clear(); strPrefetchWidth.clear(); strStaged.clear(); strStagedBytes = 0; postTile.clear(); eg.strWordVars.clear();
        eg.symbols.clear();
        o->schema.clear();
        for (size_t ci = 0; ci < t->cols.size(); ci++) {
            const TableColumn& c = t->cols[ci];
            // Values::dematerialize(..., required): an empty request set means all attributes
            if (!request.empty() && !has(request, c.name)) continue;
            if (!c.dptr) {
                if (request.empty()) continue;   // declared without data: cannot be part of `select *`
                failInvalid("column " + c.name + " is needed by the plan but was declared without data");
            }
            int k = (int)pipe.cols.size();
            pipe.cols.push_back((int)ci);
            std::string var = "v_" + std::to_string(k);
            eg.symbols[c.name] = Sym{var, c.type};
            symbolOrigin[c.name] = -1;
            o->schema.push_back({c.name, c.type});
            pipe.bytesPerRow += columnWidth(c.type);
            if (c.type.isString()) {
                addArg("c" + std::to_string(k), "const char*", (uint64_t)(uintptr_t)c.dptr);
                colIsString.push_back(1); colTypes.push_back("");
                line("const rsq::Str " + var + " = rsq::str(a.c" + std::to_string(k) + " + lr * " + std::to_string(c.type.len) + ", " +
                     std::to_string(c.type.len) + ");");
            } else {
                std::string ct = ExprGen::ctype(c.type);
                addArg("c" + std::to_string(k), "const " + ct + "*", (uint64_t)(uintptr_t)c.dptr);
                colIsString.push_back(0); colTypes.push_back(ct);
                rowParams += ", " + ct + " " + var;
                rowArgsTail += ", a.c" + std::to_string(k) + "[r]";
                rowArgsTailGuarded += ", (valid ? a.c" + std::to_string(k) + "[r] : (" + ct + ")0)";
            }
        }
        explainSteps.push_back("scan " + t->name + " [" + std::to_string((long long)t->nRows) + " rows, " +
                               std::to_string((long long)pipe.bytesPerRow) + " B/row]");
        consume(o->parent, o);
        finishPipeline();
    }

    // ---- the selection directly above the scan ---------------------------------------------------------------------------------
    // value range of one side of a comparison: a column of the scanned table (its statistics), a constant, or either under a cast
    bool sideRange(const Expr* e, double& lo, double& hi, int& col) {
        if (e->tag == RSQ_E_TYPECAST && e->child) {
            if (!sideRange(e->child, lo, hi, col)) return false;
            const int ds = (e->type.tag == RSQ_DECIMAL ? e->type.scale : 0) - (e->child->type.tag == RSQ_DECIMAL ? e->child->type.scale : 0);
            const double f = std::pow(10.0, (double)ds);
            lo *= f; hi *= f;
            return true;
        }
        // DATE values are yyyymmdd integers: spread them evenly (12 x 31 days a year) before they are taken as uniform
        auto linear = [&](double v) -> double {
            if (e->type.tag != RSQ_DATE) return v;
            const int64_t d = (int64_t)v;
            return (double)((d / 10000) * 372 + ((d / 100) % 100 - 1) * 31 + (d % 100 - 1));
        };
        if (e->tag == RSQ_E_CONSTANT && !e->type.isString()) { lo = hi = linear((double)e->ival); col = -1; return true; }
        if ((e->tag == RSQ_E_ADD || e->tag == RSQ_E_SUB) && e->child && e->child->next && e->type.tag != RSQ_DATE) {      // constant arithmetic (0.06 - 0.01)
            double alo, ahi, blo, bhi; int ac = -1, bc = -1;
            if (!sideRange(e->child, alo, ahi, ac) || !sideRange(e->child->next, blo, bhi, bc) || ac >= 0 || bc >= 0) return false;
            const int s0 = e->type.tag == RSQ_DECIMAL ? e->type.scale : 0;
            const double fa = std::pow(10.0, (double)(s0 - (e->child->type.tag == RSQ_DECIMAL ? e->child->type.scale : 0)));
            const double fb = std::pow(10.0, (double)(s0 - (e->child->next->type.tag == RSQ_DECIMAL ? e->child->next->type.scale : 0)));
            lo = hi = e->tag == RSQ_E_ADD ? alo * fa + blo * fb : alo * fa - blo * fb; col = -1;
            return true;
        }
        if (e->tag == RSQ_E_ATTRIBUTE) {
            auto so = symbolOrigin.find(e->symbol);
            if (so == symbolOrigin.end() || so->second != -1) return false;
            const int ci = pipe.src->findCol(e->symbol);
            if (ci < 0 || !pipe.src->cols[(size_t)ci].stats.valid || pipe.src->cols[(size_t)ci].type.isString()) return false;
            lo = linear((double)pipe.src->cols[(size_t)ci].stats.min); hi = linear((double)pipe.src->cols[(size_t)ci].stats.max); col = ci;
            return true;
        }
        return false;
    }
    // expected fraction of rows a predicate passes; 1 (no claim) for whatever it does not understand
    double passFraction(const Expr* e) {
        if (e->tag == RSQ_E_AND || e->tag == RSQ_E_OR) {
            double all = 1.0, none = 1.0;
            for (Expr* c : e->children()) { const double f = passFraction(c); all *= f; none *= 1.0 - f; }
            return e->tag == RSQ_E_AND ? all : 1.0 - none;
        }
        if (e->tag < RSQ_E_LT || e->tag > RSQ_E_NEQ || !e->child || !e->child->next) { leadPassComplete = false; return 1.0; }
        double alo, ahi, blo, bhi; int ac = -1, bc = -1;
        if (!sideRange(e->child, alo, ahi, ac) || !sideRange(e->child->next, blo, bhi, bc)) { leadPassComplete = false; return 1.0; }
        if ((ac >= 0) == (bc >= 0)) { leadPassComplete = false; return 1.0; }                      // column against constant only
        int tag = e->tag;
        if (ac < 0) {                                                // constant OP column -> column OP' constant
            std::swap(alo, blo); std::swap(ahi, bhi);
            tag = tag == RSQ_E_LT ? RSQ_E_GT : tag == RSQ_E_LE ? RSQ_E_GE : tag == RSQ_E_GT ? RSQ_E_LT : tag == RSQ_E_GE ? RSQ_E_LE : tag;
        }
        const double width = ahi - alo + 1.0, c = blo;
        double below = (c - alo) / width;                            // fraction of values < c
        below = std::min(1.0, std::max(0.0, below));
        const double at = (c >= alo && c <= ahi) ? 1.0 / width : 0.0;
        switch (tag) {
            case RSQ_E_LT: return below;
            case RSQ_E_LE: return std::min(1.0, below + at);
            case RSQ_E_GT: return std::max(0.0, 1.0 - below - at);
            case RSQ_E_GE: return 1.0 - below;
            case RSQ_E_EQ: return at;
            default: return 1.0 - at;
        }
    }
    void leadColumnsOf(const Expr* e, std::vector<int>& out, bool& ok) {
        if (e->tag == RSQ_E_ATTRIBUTE) {
            auto sy = eg.symbols.find(e->symbol);
            auto so = symbolOrigin.find(e->symbol);
            if (sy == eg.symbols.end() || so == symbolOrigin.end() || so->second != -1 || sy->second.var.compare(0, 2, "v_") != 0) { ok = false; return; }
            const int k = atoi(sy->second.var.c_str() + 2);
            if (k < 0 || k >= (int)colIsString.size() || colIsString[(size_t)k]) { ok = false; return; }
            if (std::find(out.begin(), out.end(), k) == out.end()) out.push_back(k);
            return;
        }
        for (Expr* c : e->children()) leadColumnsOf(c, out, ok);
    }
    // Short string columns the selection right above the scan compares with constants: their bytes are loaded WITH the tile (one or
    // two 8-byte words per row, in flight together with the numeric columns) and reach the row function as parameters, instead of
    // being fetched inside it row by row - eight dependent round trips per lane and iteration (TPC-H Q3's customer pipeline:
    // c_mktsegment = 'BUILDING').  RSQ_STRING_PREFETCH=0: never.
    std::vector<std::pair<int, int>> strPrefetch;         // (scanned column, bytes of it that arrive with the tile), in the order of the row function's parameters
    std::map<int, int> strPrefetchWidth;                   // scanned column -> its width (the row stride)
    // Staged string tiles.  A lane that fetches ITS two rows of a CHAR(25) column asks for 8 bytes at a stride of 50: the wave's one load
    // instruction touches 25 memory lines, the next word's the same 25 again, and the texture unit, not the memory, bounds the kernel (TPC-H
    // Q19 at SF10: 63 B rows at 3.5 TB/s, against 6.8 for Q1's plain columns).  A column of at most 32 bytes is therefore fetched as what it
    // is - 128 rows x W contiguous bytes per tile, 16 bytes per lane and load, every line once - and passed through the wave's own LDS
    // region, from which each lane reads its rows' words (ds_read_b64 takes any address on gfx950).  All words of the value then arrive
    // as row-function parameters.
    std::string postTile;                                  // code behind the two row_fn calls of a tile in the tile loops ($TILE = the tile's number; wave-uniform)
    std::map<int, int> strStaged;                          // scanned column -> byte offset of its tile in the wave's LDS region
    int strStagedBytes = 0;                                // bytes of that region (128 x the staged widths)
    void prefetchComparedStrings(const Expr* e) {
        if (1 == 0 || 1 == 0) return;
        if ((e->tag == RSQ_E_EQ || e->tag == RSQ_E_NEQ) && e->child && e->child->next) {
            const Expr* l = e->child; const Expr* r = e->child->next;
            const Expr* col = l->tag == RSQ_E_ATTRIBUTE && r->tag == RSQ_E_CONSTANT ? l : r->tag == RSQ_E_ATTRIBUTE && l->tag == RSQ_E_CONSTANT ? r : nullptr;
            if (!col || !l->type.isString() || !r->type.isString()) return;
            auto sy = eg.symbols.find(col->symbol);
            auto so = symbolOrigin.find(col->symbol);
            if (sy == eg.symbols.end() || so == symbolOrigin.end() || so->second != -1 || sy->second.var.compare(0, 2, "v_") != 0) return;
            const int k = atoi(sy->second.var.c_str() + 2);
            const int W = sy->second.type.len;
            if (k < 0 || k >= (int)colIsString.size() || !colIsString[(size_t)k] || W < 2) return;
            for (auto& sp : strPrefetch) if (sp.first == k) return;
            // (up to 16 bytes: the whole value; longer: its first word - most values differ there, and the line it sits in is on its
            // way when the row function asks for the rest)
            const bool stage = W <= 32 && strStagedBytes + 128 * W <= 128 * 40;
            if (stage) { strStaged[k] = strStagedBytes; strStagedBytes += 128 * W; }
            const int PW = stage || W <= 16 ? W : 8;
            strPrefetch.push_back({k, PW});
            strPrefetchWidth[k] = W;
            eg.strWordVars[sy->second.var] = (PW + 7) / 8;
            for (int w = 0; w * 8 < PW; w++) {
                const std::string ld = "rsq::ld_bytes<" + std::to_string(std::min(8, PW - w * 8)) + ">(a.c" + std::to_string(k) + " + r * " + std::to_string(W) + " + " + std::to_string(w * 8) + ")";
                rowParams += ", u64 " + sy->second.var + "_w" + std::to_string(w);
                rowArgsTail += ", " + ld;
                rowArgsTailGuarded += ", (valid ? " + ld + " : 0ull)";
            }
            return;
        }
        if (e->tag == RSQ_E_AND || e->tag == RSQ_E_OR) for (Expr* c : e->children()) prefetchComparedStrings(c);
    }
    void noteLeadingSelection(const Expr* e, const std::string& cond) {
        bool ok = true;
        std::vector<int> cols;
        leadColumnsOf(e, cols, ok);
        if (!ok || cols.empty()) return;
        leadCond = cond; leadCols = cols; leadPassComplete = true; leadPass = passFraction(e);
    }

    // -------------------------------------------------------------------------------------------
    void consume(OpNode* o, OpNode* from) {
        if (!o) failInvalid("plan root must be a materializing operator");
        switch (o->tag) {
            case RSQ_OP_SELECTION: {                    // selection.h:52-70
                o->schema = from->schema;
                if (!q.requestAll) o->schema = prune(o->schema, requestOf[o]);
                q.pool.addId(o->exprs[0]);
                const bool wasSelective = selective;
                selective = true;
                {
                    // (RSQ_STRING_PREFETCH=2, measurement: also behind probes, as long as no selection came before)
                    if (from->tag == RSQ_OP_SCAN || (1 == 2 && !wasSelective)) prefetchComparedStrings(o->exprs[0]);
                    const std::string cond = eg.emit(o->exprs[0]);
                    if (from->tag == RSQ_OP_SCAN && leadCond.empty()) noteLeadingSelection(o->exprs[0], cond);
                    openScope("if (" + cond + ") {");
                }
                explainSteps.push_back("selection " + serializeExpr(o->exprs[0]));
                consume(o->parent, o);
                closeScope();
                break;
            }
            case RSQ_OP_PROJECTION: {                   // projection.h:62-72
                Schema s;
                std::vector<std::pair<std::string, Sym>> defs;
                openScope("{");
                int k = 0;
                for (Expr* e : o->exprs) {
                    q.pool.addId(e);
                    std::string var = "p" + std::to_string((int)(size_t)o->exprs.size()) + "_" + std::to_string(k++) + "_" + std::to_string(indent);
                    line("const " + ExprGen::ctype(e->type) + " " + var + " = " + eg.emit(e) + ";");
                    defs.push_back({expressionName(e), Sym{var, e->type}});
                    s.push_back({expressionName(e), e->type});
                }
                for (auto& d : defs) { eg.symbols[d.first] = d.second; symbolOrigin[d.first] = -2; }
                o->schema = s;
                consume(o->parent, o);
                closeScope();
                break;
            }
            case RSQ_OP_HASHJOIN:
                if (joinPhase[o] == 1) consumeBuild(o, from); else consumeProbe(o, from);
                break;
            case RSQ_OP_AGGREGATION: consumeAggregation(o, from); break;
            case RSQ_OP_MATERIALIZE: consumeMaterialize(o, from); break;
            default: failUnsupported("operator not supported by the GPU engine");
        }
    }

    // ---- wave-level selection compaction ---------------------------------------------------------
    // A selective predicate (or a join's key bitmap) leaves few lanes of a wave alive, and everything after it — hash
    // probes, inserts, HBM atomics — is a chain of dependent random accesses whose latency the wave pays for no matter
    // how few lanes take part.  So the row function is cut at that point: stage 1 (scan, predicates, bitmap test) pushes
    // the values the rest needs into a per-wave LDS queue (ballot + prefix popcount, no atomics), and stage 2 runs only
    // when 64 rows are queued, with every lane busy.  TPC-H Q3's orders pipeline (9.7 % of the rows reach the probe) then
    // pays for one probe / insert pass per ~5 tiles instead of two per tile.
    // Not for pipelines that materialise (output positions depend on the scan order of each lane).
    bool downstreamMaterializes(OpNode* o) {
        for (OpNode* p = o; p; p = p->parent) {
            if (p->tag == RSQ_OP_AGGREGATION) return false;
            if (p->tag == RSQ_OP_HASHJOIN && joinPhase[p] == 1) return false;
            if (p->tag == RSQ_OP_MATERIALIZE) return true;
        }
        return true;
    }
    bool compactThen(OpNode* o, const std::function<void()>& downstream) {
        if (compacted || !selective || !envInt("RSQ_COMPACT", 1, 0, 1) || downstreamMaterializes(o)) return false;
        // not inside the match loop of a join probed for all matches: the queue takes ONE entry per row function call, and a row
        // with several matches would keep only its last (found with a constant build key: every build row the same key)
        if (multiMatchAbove) return false;
        compacted = true;
        pipe.compact = true;
        cqLive.assign(eg.symbols.begin(), eg.symbols.end());
        // Late column loads: a scanned (non-string) column that stage 1 never looked at is needed only by the rows that reach
        // stage 2.  The kernel exists in two forms from one source: RSQ_LAZY 0 loads it with the tile and carries it in the
        // queue; RSQ_LAZY 1 leaves it out of the tile loads and stage 2 reads it by row index.  The engine picks the lazy
        // form when the previous execution sent fewer than 1/32 of the rows to stage 2 (TPC-H Q3: 1.6 % of lineitem need
        // l_extendedprice and l_discount, 16 of the 24 bytes per row) — gathers for a few rows beat streaming for all, but
        // only then: a row gathered costs a 64-byte request per column.
        std::vector<int> lazyOf(cqLive.size(), -1);
        if (1) {
            for (size_t k = 0; k < cqLive.size(); k++) {
                const std::string& var = cqLive[k].second.var;
                auto org = symbolOrigin.find(cqLive[k].first);
                if (org == symbolOrigin.end() || org->second != -1 || cqLive[k].second.type.isString()) continue;
                if (var.compare(0, 2, "v_") != 0) continue;
                bool used = false;                      // does the stage-1 text mention the variable?
                for (size_t pos = body.find(var); pos != std::string::npos && !used; pos = body.find(var, pos + 1)) {
                    const size_t end = pos + var.size();
                    const bool left = pos == 0 || !(isalnum((unsigned char)body[pos - 1]) || body[pos - 1] == '_');
                    const bool right = end >= body.size() || !(isalnum((unsigned char)body[end]) || body[end] == '_');
                    if (left && right) used = true;
                }
                if (!used) { lazyOf[k] = atoi(var.c_str() + 2); pipe.lazyCols.push_back(lazyOf[k]); }
            }
        }
        // ---- everything downstream goes into stage 2, which sees the carried values under the names q_<k> ----
        const std::string stage1 = body; const int stage1Indent = indent;
        body.clear(); indent = 1;
        stage2Prefix.clear(); compFilters.clear(); inStage2 = true;
        for (size_t k = 0; k < cqLive.size(); k++) eg.symbols[cqLive[k].first] = Sym{"q_" + std::to_string(k), cqLive[k].second.type};
        explainSteps.push_back("wave compaction");
        downstream();
        while (indent > 1) closeScope();
        inStage2 = false;
        const std::string down = body;
        // Component bitmaps of later joins (consumeProbe): the value is a column of this scan, so the test belongs in STAGE 1 - the column
        // then streams with the tiles instead of being gathered row by row by every survivor, and the rows it rejects never enter the
        // queue (TPC-H Q5: 9.1 M of 60 M lineitem rows find their order, 1.8 M of those a supplier in ASIA).  A column the test reads
        // is not loaded late.  Where the stage-1 name of the value is not at hand the test stands at the top of stage 2.
        std::string stage1Cond;
        for (auto& cf : compFilters) {
            std::string v1;
            for (size_t k = 0; k < cqLive.size(); k++)
                if (cqLive[k].first == cf.symbol && cqLive[k].second.var.compare(0, 2, "v_") == 0) {
                    v1 = cqLive[k].second.var;
                    if (lazyOf[k] >= 0) { pipe.lazyCols.erase(std::remove(pipe.lazyCols.begin(), pipe.lazyCols.end(), lazyOf[k]), pipe.lazyCols.end()); lazyOf[k] = -1; }
                }
            const std::string C = cf.table + "_c";
            std::string test = "rsq::bit_in(a." + C + "_bm, (u64)((i64)(" + (v1.empty() ? cf.stage2Var : v1) + ") - a." + C + "_bmmin), a." + C + "_bmbits)";
            // A small bitmap (the supplier keys of TPC-H Q5: 12 KB) is read for BOTH rows of the lane and every tile in flight with the
            // tile loads, like the first probe's key bitmap: tested inside the row function, the eight rows a lane handles per iteration
            // each wait for their own load (measured: the pipeline 323 us; the loads hit the L1, their latency does not overlap).
            bool already = false;
            for (auto& pf : bitmapPrefetch) already = already || pf.first == C;
            if (!v1.empty() && cf.bits <= (1 << 20) && !already) {
                const int col = atoi(v1.c_str() + 2);
                bitmapPrefetch.push_back({C, col, false});
                const std::string call = "rsq::bm_word(a." + C + "_bm, a." + C + "_bmmin, a." + C + "_bmbits, (i64)";
                rowParams += ", const u32 pf_" + C;
                rowArgsTail += ", " + call + "a.c" + std::to_string(col) + "[r])";
                rowArgsTailGuarded += ", (valid ? " + call + "a.c" + std::to_string(col) + "[r]) : 0u)";
                test = "rsq::bit_of_word(pf_" + C + ", (u64)((i64)(" + v1 + ") - a." + C + "_bmmin), a." + C + "_bmbits)";
            }
            if (!v1.empty()) stage1Cond += (stage1Cond.empty() ? "" : " && ") + test;
            else stage2Prefix += "    if (!" + test + ") return;      // no build row of " + cf.table + " has this key component: the row cannot reach the sink\n";
        }
        // only the values stage 2 really reads travel through the queue (a date that was only filtered on does not); the
        // late-loaded ones take the LAST slots, which exist in the RSQ_LAZY 0 form only: the lazy form's queues are
        // smaller, more workgroups fit a CU, and a latency-bound pipeline (tile load, then the key bitmap's L2 load) gets
        // twice the waves
        auto mentions = [](const std::string& text, const std::string& var) {
            for (size_t pos = text.find(var); pos != std::string::npos; pos = text.find(var, pos + 1)) {
                const size_t end = pos + var.size();
                const bool left = pos == 0 || !(isalnum((unsigned char)text[pos - 1]) || text[pos - 1] == '_');
                const bool right = end >= text.size() || !(isalnum((unsigned char)text[end]) || text[end] == '_');
                if (left && right) return true;
            }
            return false;
        };
        std::vector<int> slot(cqLive.size(), -1);
        int nSlots = 0;
        for (int pass = 0; pass < 2; pass++)
            for (size_t k = 0; k < cqLive.size(); k++)
                if ((lazyOf[k] >= 0) == (pass == 1) && mentions(down, "q_" + std::to_string(k))) slot[k] = nSlots++;
        int nLazySlots = 0;
        for (size_t k = 0; k < cqLive.size(); k++) if (slot[k] >= 0 && lazyOf[k] < 0) nLazySlots++;
        // drop lazy columns nobody reads from the list of late loads (their tile loads can go in both forms... keep it simple:
        // they stay eager in the RSQ_LAZY 0 form and are simply not loaded in the lazy one)
        body.clear(); indent = 1;
        for (size_t k = 0; k < cqLive.size(); k++) {
            if (slot[k] < 0) continue;
            const Type& t = cqLive[k].second.type;
            const std::string v = "q_" + std::to_string(k);
            const std::string carried = "const " + ExprGen::ctype(t) + " " + v + " = " + fromWord("qw_" + std::to_string(slot[k]), t) + ";";
            if (lazyOf[k] < 0) line(carried);
            else {
                body += "#if RSQ_LAZY\n";
                line("const " + ExprGen::ctype(t) + " " + v + " = a.c" + std::to_string(lazyOf[k]) + "[row - a.row0];");
                body += "#else\n";
                line(carried);
                body += "#endif\n";
            }
        }
        stage2Body = body + stage2Prefix + down;
        body = stage1; indent = stage1Indent;
        if (!stage1Cond.empty()) openScope("if (" + stage1Cond + ") {");
        line("cq_pass = true;");
        for (size_t k = 0; k < cqLive.size(); k++) {
            if (slot[k] < 0) continue;
            const std::string push = "cq_" + std::to_string(slot[k]) + " = " + toWord(cqLive[k].second.var, cqLive[k].second.type) + ";";
            if (lazyOf[k] < 0) line(push);
            else { body += "#if !RSQ_LAZY\n"; line(push); body += "#endif\n"; }
        }
        if (!stage1Cond.empty()) closeScope();
        pipe.compactWords = nSlots;
        pipe.compactWordsLazy = pipe.lazyCols.empty() ? nSlots : nLazySlots;
        return true;
    }

    // Entries are counted per thread in a register and added to the table's counter once per wave at the end of
    // the kernel.  (One atomic per inserted entry on a single word serialises: 1.45 M of them cost 4.6 ms on
    // MI355X, more than the rest of TPC-H Q3 together; inside a divergent probe loop neither the compiler nor a
    // ballot folds them, the matching lanes arrive one at a time.)
    void countPerThread(const std::string& T) {
        stateDecl += "    u32 n_" + T + " = 0;\n";
        // wave sum -> LDS -> ONE global atomic per workgroup: atomics on a single word serialise (~11 ns each), and a
        // random-access pipeline launches 8 workgroups per CU
        epilogue += "    {\n        __shared__ u32 s_n_" + T + ";\n        if (threadIdx.x == 0) s_n_" + T + " = 0;\n        __syncthreads();\n";
        epilogue += "        const u64 v = rsq::wave_sum((u64)st.n_" + T + ");\n        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_n_" + T + ", (u32)v);\n";
        epilogue += "        __syncthreads();\n        if (threadIdx.x == 0 && s_n_" + T + ") atomicAdd(a." + T + "_count, s_n_" + T + ");\n    }\n";
    }

    // Key value(s) of one expression as table words: one word for numbers, ceil(len / 8) words of bytes for strings
    // (see rsq_device.h: str_word).  stripChar: CHAR(n) equality ignores trailing spaces (joins); group keys keep the
    // exact bytes and the host merges space-equivalent groups, because the group shows the FIRST row's spelling.
    // stripMode: -1 by the expression's own type (CHAR keys ignore trailing spaces when stripChar), 0 exact bytes, 1 ignore trailing spaces
    std::vector<std::string> keyWords(Expr* e, const std::string& prefix, bool stripChar, std::vector<std::string>* endsWithSpace = nullptr, int stripMode = -1) {
        std::vector<std::string> out;
        const std::string v = eg.emit(e);
        if (!e->type.isString()) {
            line("const i64 " + prefix + " = " + toWord(v, e->type) + ";");
            out.push_back(prefix);
            return out;
        }
        const bool strip = stripMode >= 0 ? stripMode == 1 : (stripChar && e->type.tag == RSQ_CHAR);
        if (e->tag == RSQ_E_CONSTANT) {
            // a string constant as a key: its words are literals (the text, NUL padded to the constant's declared length;
            // without its trailing spaces where the comparison ignores them)
            std::string text = e->symbol.substr(0, (size_t)std::max(0, e->type.len));
            const size_t nul = text.find('\0');
            if (nul != std::string::npos) text.resize(nul);
            if (endsWithSpace && e->type.tag == RSQ_CHAR && !strip) endsWithSpace->push_back(!text.empty() && text.back() == ' ' ? "true" : "false");
            if (strip) while (!text.empty() && text.back() == ' ') text.pop_back();
            for (int w = 0; w < (e->type.len + 7) / 8; w++) {
                uint64_t word = 0;
                for (int b = 0; b < 8; b++) { const size_t i = (size_t)w * 8 + (size_t)b; if (i < text.size()) word |= (uint64_t)(unsigned char)text[i] << (8 * b); }
                std::string kv = prefix + "_" + std::to_string(w);
                line("const i64 " + kv + " = (i64)" + std::to_string((unsigned long long)word) + "ull;");
                out.push_back(kv);
            }
            return out;
        }
        if (!strip) {
            // exact bytes: the column is NUL padded to its width (resql_plan.h), so the key words ARE the stored bytes —
            // one unaligned load per word instead of a byte loop per word (32 key words for TPC-H Q10's group-by)
            if (endsWithSpace && e->type.tag == RSQ_CHAR) endsWithSpace->push_back("rsq::ends_with_space(" + v + ")");
            for (int w = 0; w < (e->type.len + 7) / 8; w++) {
                std::string kv = prefix + "_" + std::to_string(w);
                const int rbytes = std::min(8, e->type.len - w * 8);
                line("const i64 " + kv + " = (i64)rsq::ld_bytes<" + std::to_string(rbytes) + ">((" + v + ").p + " + std::to_string(w * 8) + ");");
                out.push_back(kv);
            }
            return out;
        }
        line("const int " + prefix + "_n = rsq::str_len_char(" + v + ");");
        for (int w = 0; w < (e->type.len + 7) / 8; w++) {
            std::string kv = prefix + "_" + std::to_string(w);
            line("const i64 " + kv + " = rsq::str_word(" + v + ", " + prefix + "_n, " + std::to_string(w) + ");");
            out.push_back(kv);
        }
        return out;
    }

    // String join keys of different declared lengths.  The reference hashes each side with its own type: hashVarchar stops
    // at the NUL, so VARCHAR(a) = VARCHAR(b) matches equal strings — both sides take the word count of the wider one, the
    // narrower side's missing words are zero.  hashChar pads with spaces to the DECLARED length (qlib/hash.h:131-147), so
    // CHAR(a) = CHAR(b), a != b, never has equal hashes and never matches: the two sides get pad words that differ.
    // CHAR against VARCHAR (any lengths): equal hashes need the VARCHAR value to be exactly as long as the CHAR column is wide
    // (hashChar counts the pad spaces, hashVarchar only the characters), and the key comparison is the PROBE side's
    // (checkEquality(probeKeys, entryKeys), hashjoin.h:142/191: compareChar ignores trailing spaces, compareVarchar does not).
    // So both sides form their words the probe side's way (joinKeyStripMode), padded to the wider side's word count, plus one
    // word that holds the hashed length: the declared width of a CHAR key, the actual length of a VARCHAR key.
    static bool mixedStringKinds(const Expr* a, const Expr* b) { return a->type.isString() && b->type.isString() && a->type.tag != b->type.tag; }
    static int joinKeyStripMode(const Expr* side, const Expr* probeSide, const Expr* buildSide) {
        (void)side;
        return mixedStringKinds(probeSide, buildSide) ? (probeSide->type.tag == RSQ_CHAR ? 1 : 0) : -1;
    }
    void padKeyWords(Expr* mine, Expr* other, size_t w0, std::vector<std::string>& keyVars, bool buildSide) {
        if (!mine->type.isString() || !other->type.isString()) return;
        if (mine->type.tag != other->type.tag) {
            const size_t want = (size_t)(std::max(mine->type.len, other->type.len) + 7) / 8;
            while (keyVars.size() - w0 < want) keyVars.push_back("((i64)0)");
            keyVars.push_back(mine->type.tag == RSQ_CHAR ? "((i64)" + std::to_string(mine->type.len) + ")" : "((i64)rsq::str_len_exact(" + eg.emit(mine) + "))");
            return;
        }
        if (mine->type.len == other->type.len) return;
        const size_t want = (size_t)(std::max(mine->type.len, other->type.len) + 7) / 8 + (mine->type.tag == RSQ_CHAR ? 1 : 0);
        const std::string pad = mine->type.tag == RSQ_CHAR && !buildSide ? "((i64)-1)" : "((i64)0)";
        while (keyVars.size() - w0 < want) keyVars.push_back(pad);
    }

    // ---- hash join build (hashjoin.h:226-256) ---------------------------------------------------
    std::string hashOf(const std::vector<std::string>& keyVars) {
        std::string h = "rsq::hash64((u64)" + keyVars[0] + ")";
        for (size_t i = 1; i < keyVars.size(); i++) h = "rsq::hash64(" + h + " ^ ((u64)" + keyVars[i] + " * 0x9E3779B97F4A7C15ull))";
        return h;
    }

    // Word w of the slot in `T_s` of a join table.  Join tables keep a slot's words next to each other (array of
    // structures): the CAS on the key and the payload stores of an insert fall into one cache line, which the memory side
    // then writes back once instead of read-modify-writing three lines; a probe that matches finds the payload in the line
    // it already fetched for the key.  (The generic aggregation's tables stay structure-of-arrays: their key words are
    // compared one array at a time and their accumulators live in separate blocks anyway.)
    static std::string wordAt(const HashTable& ht, const std::string& T, int w) {
        const int nw = std::max<int>(1, (int)(ht.keys.size() + ht.payload.size()));
        if (ht.aos) return "a." + T + "_words[" + T + "_s * " + std::to_string(nw) + " + " + std::to_string(w) + "]";
        return "a." + T + "_words[" + std::to_string(w) + " * a." + T + "_cap + " + T + "_s]";
    }

    // Home slot of a join key.  RSQ_BLOCKED_HASH=1 (off by default — measured and rejected) makes the hash of one integer key
    // of known range BLOCKED: 128 consecutive key values share a hashed base slot and spread, in key order, over the slots
    // behind it, so that tables clustered by the key insert and probe neighbouring slots from neighbouring rows.  On MI355X
    // that is 2-13x SLOWER (Q3 SF10 0.51 -> 1.1 ms, Q14 SF1 0.09 -> 1.2 ms): the 64 CAS of a wave then land in a handful of
    // cache lines, and atomics on one line serialise at the memory side just like atomics on one word.  Scattering the
    // inserts over the table is what keeps them fast.
    std::string slotOf(const HashTable& ht, const std::string& T, const std::vector<std::string>& keyVars) {
        (void)ht;
        return hashOf(keyVars) + " & " + T + "_mask";
    }

    void consumeBuild(OpNode* o, OpNode* from) {
        if (compactThen(o, [&] { consumeBuildBody(o, from); })) return;
        consumeBuildBody(o, from);
    }
    void consumeBuildBody(OpNode* o, OpNode* from) {
        pipe.gridPerCU = 8;
        std::unique_ptr<HashTable> ht(new HashTable());
        ht->id = (int)q.hashTables.size();
        ht->unique = o->singleMatch;
        const std::string T = "ht" + std::to_string(ht->id);
        std::vector<std::string> keyVars;
        std::vector<int> keyFirstWord;           // per key expression: its table word (-1 for multi-word string keys)
        openScope("{");
        int k = 0;
        for (Expr* eq : o->exprs) {
            if (eq->tag != RSQ_E_EQ) failType("The elements of the expression list passed to equalitiesLeftSide(..) need the tag Expr::EQ");
            Expr* l = eq->child;
            q.pool.addId(l);
            size_t w0 = keyVars.size();
            keyFirstWord.push_back(l->type.isString() ? -1 : (int)w0);
            for (auto& kv : keyWords(l, T + "_k" + std::to_string(k++), true, nullptr, joinKeyStripMode(l, eq->child->next, l))) keyVars.push_back(kv);
            padKeyWords(l, eq->child->next, w0, keyVars, true);
            for (size_t w = w0; w < keyVars.size(); w++)
                ht->keys.push_back({w == w0 ? expressionName(l) : expressionName(l) + "#" + std::to_string(w - w0), w == w0 && !l->type.isString() ? l->type : Type(RSQ_BIGINT)});
        }
        // build payload = the attributes of the left child's schema (Values::get(_lChild->_schema)); an attribute that is
        // itself a (one-word) join key is not stored again: a matching probe already holds its value
        for (auto& a : from->schema) {
            auto it = eg.symbols.find(a.name);
            if (it == eg.symbols.end()) failType("hash join build value " + a.name + " has no symbol");
            int alias = -1;
            for (size_t ki = 0; ki < o->exprs.size(); ki++) {
                Expr* l = o->exprs[ki]->child;
                if (l->tag == RSQ_E_ATTRIBUTE && l->symbol == a.name && !l->type.isString() && keyFirstWord[ki] >= 0) alias = keyFirstWord[ki];
            }
            if (alias >= 0) ht->keyAlias.push_back({{a.name, it->second.type}, alias});
            else ht->payload.push_back({a.name, it->second.type});
        }
        // key-domain bitmap (see HashTable): one integer key that is a column of this pipeline's scan with usable statistics
        if (o->exprs.size() == 1 && keyVars.size() == 1 && envInt("RSQ_JOIN_BITMAP", 1, 0, 1)) {
            Expr* l = o->exprs[0]->child;
            auto org = symbolOrigin.find(l->symbol);
            if (l->tag == RSQ_E_ATTRIBUTE && !l->type.isString() && org != symbolOrigin.end() && org->second == -1) {
                int ci = pipe.src->findCol(l->symbol);
                if (ci >= 0 && pipe.src->cols[(size_t)ci].stats.valid && pipe.src->nRows > 0) {
                    const ColumnStats& st = pipe.src->cols[(size_t)ci].stats;
                    unsigned __int128 range = (unsigned __int128)((__int128)st.max - (__int128)st.min) + 1;
                    if (range <= ((unsigned __int128)1 << 28)) { ht->hasBitmap = true; ht->bmMin = st.min; ht->bmBits = (int64_t)range; }
                }
            }
        }
        // ... or, for a table with several key words, a bitmap over ONE integer component (HashTable::hasCompBitmap)
        if (!ht->hasBitmap && keyVars.size() > 1 && envInt("RSQ_JOIN_BITMAP", 1, 0, 1))
            for (size_t ki = 0; ki < o->exprs.size() && !ht->hasCompBitmap; ki++) {
                Expr* l = o->exprs[ki]->child;
                auto org = symbolOrigin.find(l->symbol);
                if (l->tag != RSQ_E_ATTRIBUTE || l->type.isString() || org == symbolOrigin.end() || org->second != -1 || keyFirstWord[ki] < 0) continue;
                const int ci = pipe.src->findCol(l->symbol);
                if (ci < 0 || !pipe.src->cols[(size_t)ci].stats.valid || pipe.src->nRows == 0) continue;
                const ColumnStats& st = pipe.src->cols[(size_t)ci].stats;
                const unsigned __int128 range = (unsigned __int128)((__int128)st.max - (__int128)st.min) + 1;
                if (range > ((unsigned __int128)1 << 26)) continue;
                ht->hasCompBitmap = true; ht->compWord = keyFirstWord[ki]; ht->cbMin = st.min; ht->cbBits = (int64_t)range;
            }
        // capacity: the reference sizes its table lChild.getSize() * 5 / 3 and grows it; ours cannot grow
        // inside a kernel, so it is sized for twice the rows the build pipeline can deliver and re-run
        // at double size if it still overflows (engine.cpp).
        ht->aos = 1 != 0;
        ht->capacity = 0;     // decided by the sizing pass at execute time (engine.cpp)
        // One integer key word whose values can never be INT64_MIN: the key word itself is the slot's state.  A 64-bit CAS
        // from the EMPTY sentinel claims the slot and publishes the key in one memory request (instead of a CAS on a state
        // word plus a key store), a probe step reads one word instead of two dependent ones.  Scattered HBM requests are what
        // a build costs.
        if (keyVars.size() == 1 && o->exprs.size() == 1 && 1) {
            Expr* l = o->exprs[0]->child;
            const int tg = l->type.tag;
            if (tg == RSQ_INT || tg == RSQ_DATE || tg == RSQ_BOOL || (tg == RSQ_CHAR && l->type.len == 1)) ht->keyCas = true;   // widened 32-bit / 8-bit values
            else if ((tg == RSQ_BIGINT || tg == RSQ_DECIMAL) && l->tag == RSQ_E_ATTRIBUTE) {
                auto org = symbolOrigin.find(l->symbol);
                int ci = pipe.src->findCol(l->symbol);
                if (org != symbolOrigin.end() && org->second == -1 && ci >= 0 && pipe.src->cols[(size_t)ci].stats.valid &&
                    pipe.src->cols[(size_t)ci].stats.min > INT64_MIN) ht->keyCas = true;
            }
        }
        // Bitmap-rank dictionary (HashTable::rankCapable, kernels/rsq_device.h rank_of): a table that is probed single-match over
        // one integer key with a key bitmap needs no hashing when its build keys prove unique.  The same kernel carries both
        // forms behind a uniform branch on a.<T>_rank; the host decides once, from the sizing pass.
        // (A join probed for ALL matches qualifies too: with unique build keys every probe has at most one.  If such a table carries
        // nothing but its key - TPC-H Q3's customer side - the bitmap IS the table in the rank form: a KEY SET, no entries at all.)
        ht->rankCapable = (ht->unique || 1) && ht->hasBitmap && ht->keyCas && ht->aos && keyVars.size() == 1 &&
                          envInt("RSQ_JOIN_RANK", 1, 0, 1) != 0;
        ht->setOnly = ht->rankCapable && !ht->unique && ht->payload.empty();
        // IDENTITY: the build pipeline is the bare scan of a table in the order of its (engine-owned, hence immutable) key column.  If the
        // sizing pass then finds the keys unique and every row inserted, entry number rank(key) IS the row's number: the build writes
        // its record straight to words[row] - coalesced, streaming - and the arrival buffer and the placement kernel are not needed
        // (TPC-H Q12 builds on all 15 M orders: 240 MB appended, read again and scattered to entries 16 bytes at a time).
        {
            const int ci = o->exprs[0]->child->tag == RSQ_E_ATTRIBUTE ? pipe.src->findCol(o->exprs[0]->child->symbol) : -1;
            ht->identityCapable = ht->rankCapable && !ht->setOnly && from->tag == RSQ_OP_SCAN && ci >= 0 && pipe.src->cols[(size_t)ci].owned &&
                                  pipe.src->cols[(size_t)ci].stats.valid && pipe.src->cols[(size_t)ci].stats.ascending;
        }
        // (a table that may become a rank dictionary keeps its bitmap in the interleaved layout, rsq_device.h bmi_word)
        ht->bmInterleaved = ht->rankCapable;
        const std::string bmw = ht->bmInterleaved ? "rsq::bmi_word(d)" : "d >> 5";
        // (a key outside the range the statistics promised sets no bit and raises ERR_GROUP_OVERFLOW: the host fails the execution)
        bool checkKey = true, combineBits = false;
        if (ht->hasBitmap && o->exprs[0]->child->tag == RSQ_E_ATTRIBUTE) {
            const int ci = pipe.src->findCol(o->exprs[0]->child->symbol);
            if (ci >= 0 && pipe.src->cols[(size_t)ci].owned && !envInt("RSQ_CHECK_STATS", 0, 0, 1)) checkKey = false;      // (engine-owned columns cannot change)
            // a table scanned in the order of its build key (column statistics): the rows of a wave fall into a few bitmap words, and
            // the lanes that meet in one word set their bits with ONE atomic (rsq_device.h bm_set_combined).  Memory-side atomics
            // run at ~25 G/s chip-wide: a build over all 15 M orders (TPC-H Q12) spent 0.6 of its 0.73 ms on them.
            if (ci >= 0 && pipe.src->cols[(size_t)ci].stats.valid && pipe.src->cols[(size_t)ci].stats.ascending && 1)
                combineBits = true;
        }
        const std::string setBit = combineBits ? "rsq::bm_set_combined(a." + T + "_bm, (u32)(" + bmw + "), 1u << (d & 31));"
                                               : "atomicOr(&a." + T + "_bm[" + bmw + "], 1u << (d & 31));";
        const std::string bitSet = !ht->hasBitmap ? std::string() : !checkKey ? "const u64 d = (u64)(" + keyVars[0] + " - a." + T + "_bmmin); " :
                                   "const u64 d0 = (u64)(" + keyVars[0] + " - a." + T + "_bmmin); if (d0 >= a." + T +
                                   "_bmbits) atomicOr(a.err, (u32)rsq::ERR_GROUP_OVERFLOW); const u64 d = d0 < a." + T + "_bmbits ? d0 : 0; ";
        addArg(T + "_state", "u32*", 0); addArg(T + "_words", "i64*", 0); addArg(T + "_cap", "u64", 0); addArg(T + "_count", "u32*", 0);
        addArg(T + "_countonly", "u64", 0);
        if (ht->hasBitmap) { addArg(T + "_bm", "u32*", 0); addArg(T + "_bmmin", "i64", (uint64_t)ht->bmMin); addArg(T + "_bmbits", "u64", (uint64_t)ht->bmBits); }
        if (ht->hasCompBitmap) {
            // every build row sets its component's bit, in the sizing pass and in the build alike (a value outside the range the statistics promised raises ERR_GROUP_OVERFLOW)
            addArg(T + "_c_bm", "u32*", 0); addArg(T + "_c_bmmin", "i64", (uint64_t)ht->cbMin); addArg(T + "_c_bmbits", "u64", (uint64_t)ht->cbBits);
            line("{ const u64 cd = (u64)(" + keyVars[(size_t)ht->compWord] + " - a." + T + "_c_bmmin); if (cd < a." + T + "_c_bmbits) { const u32 cb = 1u << (cd & 31); if (!(a." + T +
                 "_c_bm[cd >> 5] & cb)) atomicOr(&a." + T + "_c_bm[cd >> 5], cb); } else atomicOr(a.err, (u32)rsq::ERR_GROUP_OVERFLOW); }");
        }
        // sizing pass: the same pipeline run once with countonly = 1 tells the host how many entries to expect — and, for a
        // table that could be a rank dictionary, whether two build rows share a key (a bit that is already set)
        countPerThread(T);
        if (ht->rankCapable) {
            addArg(T + "_rank", "u64", 0); addArg(T + "_temp", "i64*", 0);
            openScope("if (a." + T + "_countonly) {");
            line("st.n_" + T + "++;");
            line("{ " + bitSet + "const u32 b = 1u << (d & 31); if (atomicOr(&a." + T + "_bm[" + bmw + "], b) & b) atomicOr(a.err, (u32)rsq::NOTE_BUILD_KEYS_NOT_UNIQUE); }");
            closeScope();
            if (ht->setOnly) {
                // key set: the bit is everything; a bit that is already set means the build side changed since the sizing pass
                openScope("else if (a." + T + "_rank) {");
                line("{ " + bitSet + "const u32 b = 1u << (d & 31); if (atomicOr(&a." + T + "_bm[" + bmw + "], b) & b) atomicOr(a.err, (u32)rsq::NOTE_BUILD_KEYS_NOT_UNIQUE); }");
                line("st.n_" + T + "++;");
                closeScope();
                openScope("else {");
            } else {
            openScope("else if (a." + T + "_rank) {");
            // The record goes to the arrival-order buffer, into the region of the wave that produced it: a.<T>_treg records
            // per wave (the host sizes the regions at four times the mean from the sizing pass; tiles are dealt to the waves
            // round-robin, so every wave sees an even sample of the table).  No atomics: a returning atomic on ONE counter word
            // serialises at ~11 ns, and even one reservation per wave and 256 records made this pipeline 2x slower.  A wave's
            // fill count lives in LDS, because the lanes of a wave reach this point in diverged groups; it is written to
            // a.<T>_tused[wave] at the end, where the placement kernel finds it.  A wave that overflows its region says so
            // (the host then keeps the hash form).
            line("{ " + bitSet + setBit + " }");
            if (ht->identityCapable) {
                addArg(T + "_ident", "u64", 0);
                // Records at the row's number: the 128 rows of a tile are 128 x NW consecutive words of the table.  A lane storing ITS rows' words
                // writes 8 bytes at a stride of 16 x NW per instruction (TPC-H Q12's orders table: 240 MB of records in 94 us, Q10's customers:
                // seven words per row in 48 us); the records go through the wave's LDS region instead and leave as the wave's 16-byte stores,
                // every line whole (the flush behind the tile's rows, finishPipeline).  The rows behind the last tile store directly.
                const std::string NWI = std::to_string(1 + (int)ht->payload.size());
                stateDecl += "    i64* rec_" + T + ";\n    bool in_tile = true;\n";
                prologue += "    __shared__ __attribute__((aligned(16))) i64 s_rec_" + T + "[(RSQ_BLOCK_THREADS / 64) * 128 * " + NWI + "];\n    st.rec_" + T + " = s_rec_" + T +
                            " + (threadIdx.x >> 6) * 128 * " + NWI + ";\n";
                pipe.extraLdsBytes += (pipe.blockThreads / 64) * 128 * 8 * (1 + (int)ht->payload.size());
                postTile += "            if (a." + T + "_ident && a." + T + "_rank && !a." + T + "_countonly) rsq::flush_tile_records<" + NWI + ">(st.rec_" + T + ", a." + T + "_words + (u64)(($TILE) << 7) * " + NWI +
                            "ull, lane);\n";
                openScope("if (a." + T + "_ident) {");
                line("i64* rec = st.in_tile ? st.rec_" + T + " + (u32)(lr & 127) * " + NWI + "u : a." + T + "_words + (u64)(row - a.row0) * " + NWI + "ull;");
                line("rec[0] = " + keyVars[0] + ";");
                int iw = 1;
                for (auto& p : ht->payload) line("rec[" + std::to_string(iw++) + "] = " + toWord(eg.symbols[p.name].var, p.type) + ";");
                closeScope();
                openScope("else {");
            }
            addArg(T + "_treg", "u64", 0); addArg(T + "_tused", "u32*", 0);
            stateDecl += "    u32* tch_" + T + ";\n";
            prologue += "    __shared__ u32 s_tch_" + T + "[RSQ_BLOCK_THREADS / 64];\n    st.tch_" + T + " = s_tch_" + T + " + (threadIdx.x >> 6);\n" +
                        "    if ((threadIdx.x & 63) == 0) st.tch_" + T + "[0] = 0u;\n";
            const int nw = 1 + (int)ht->payload.size();
            const std::string NW = std::to_string(nw);
            line("u64 " + T + "_t;");
            openScope("{");
            line("const u64 act = __ballot(1);");
            line("const int ln = (int)(threadIdx.x & 63), leader = __ffsll((long long)act) - 1;");
            line("const u32 pos = st.tch_" + T + "[0];");
            line(T + "_t = (u64)pos + (u64)__popcll(act & ((1ull << ln) - 1ull));");
            line("if (ln == leader) st.tch_" + T + "[0] = pos + (u32)__popcll(act);");
            closeScope();
            {
                openScope("if (" + T + "_t < a." + T + "_treg) {");
                line("i64* rec = a." + T + "_temp + (((u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * a." + T + "_treg + " + T + "_t) * " + NW + ";");
                line("rec[0] = " + keyVars[0] + ";");
                int tw = 1;
                for (auto& p : ht->payload)
                    line("rec[" + std::to_string(tw++) + "] = " + toWord(eg.symbols[p.name].var, p.type) + ";");
                closeScope();
                line("else atomicOr(a.err, (u32)rsq::NOTE_BUILD_KEYS_NOT_UNIQUE);      // the region is full: this table is not for the dictionary");
            }
            if (ht->identityCapable) closeScope();
            epilogue += "    if (a." + T + "_rank && !a." + T + "_countonly && (threadIdx.x & 63) == 0) {\n        const u32 used = st.tch_" + T + "[0];\n" +
                        "        a." + T + "_tused[(u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = used < a." + T + "_treg ? used : (u32)a." + T + "_treg;\n    }\n";
            line("st.n_" + T + "++;");
            closeScope();
            openScope("else {");
            }
        } else
        openScope("if (a." + T + "_countonly) { st.n_" + T + "++; } else {");
        line("const u64 " + T + "_mask = a." + T + "_cap - 1;");
        line("u64 " + T + "_s = " + slotOf(*ht, T, keyVars) + ";");
        line("u64 " + T + "_n = 0;");
        // (measurement only, wrong results: 1 no payload stores, 2 a plain store into the home slot instead of the CAS loop,
        // 4 no insert at all — to see what each part of an insert costs)
        const int dbgBuild = 0;
        openScope("for (;; " + T + "_n++) {");
        line("if (" + T + "_n > " + T + "_mask) { atomicOr(a.err, (u32)rsq::ERR_HT_FULL); break; }");
        if (dbgBuild & 4) line("break;");
        else if ((dbgBuild & 2) && ht->keyCas) line(wordAt(*ht, T, 0) + " = " + keyVars[0] + "; break;");
        if (ht->keyCas)
            line("if (atomicCAS(reinterpret_cast<unsigned long long*>(&" + wordAt(*ht, T, 0) + "), 0x8000000000000000ull, (unsigned long long)" +
                 keyVars[0] + ") == 0x8000000000000000ull) break;");
        else line("if (atomicCAS(&a." + T + "_state[" + T + "_s], 0u, 1u) == 0u) break;");
        line(T + "_s = (" + T + "_s + 1) & " + T + "_mask;");
        closeScope();
        openScope("if (" + T + "_n <= " + T + "_mask) {");
        int w = 0;
        for (auto& kv : keyVars) {
            if (!ht->keyCas) line(wordAt(*ht, T, w) + " = " + kv + ";");
            w++;
        }
        for (auto& p : ht->payload) {
            if (dbgBuild & 5) { w++; continue; }
            line(wordAt(*ht, T, w++) + " = " + toWord(eg.symbols[p.name].var, p.type) + ";");
        }
        line("st.n_" + T + "++;");
        if (ht->hasBitmap) {
            line("{ " + bitSet + setBit + " }");
        }
        closeScope();
        closeScope();
        closeScope();
        pipe.sink = SinkKind::BUILD;
        pipe.buildTable = ht->id;
        o->hashTable = ht->id;
        explainSteps.push_back("build hash table " + T + " (" + std::to_string(ht->keys.size()) + " key(s), " +
                               std::to_string(ht->payload.size()) + " payload word(s), sized by a counting pass" +
                               (ht->hasBitmap ? ", key bitmap of " + std::to_string((long long)ht->bmBits) + " bits" : "") +
                               (ht->keyCas ? ", key word is the slot state" : "") +
                               (ht->setOnly ? "; nothing but the bitmap when the build keys prove unique" :
                                ht->rankCapable ? "; a bitmap-rank dictionary instead when the build keys prove unique" : "") + ")");
        q.hashTables.push_back(std::move(ht));
    }

    // ---- hash join probe (hashjoin.h:118-214) ---------------------------------------------------
    // key words of the probe side of join `o` (emits their computation at the current position)
    void probeKeys(OpNode* o, const std::string& T, std::vector<std::string>& keyVars, std::vector<std::string>& probeKeyNames) {
        int k = 0;
        for (Expr* eq : o->exprs) {
            Expr* r = eq->child->next;
            q.pool.addId(r);
            const size_t w0 = keyVars.size();
            for (auto& kv : keyWords(r, T + "_p" + std::to_string(k++), true, nullptr, joinKeyStripMode(r, r, eq->child))) keyVars.push_back(kv);
            padKeyWords(r, eq->child, w0, keyVars, false);
            // (only a one-word key can stand in for the build key of the matched entry, see tryJoinEntry)
            probeKeyNames.push_back(keyVars.size() - w0 == 1 ? expressionName(r) : std::string());
            for (size_t w = w0 + 1; w < keyVars.size(); w++) probeKeyNames.push_back(std::string());
        }
    }

    void consumeProbe(OpNode* o, OpNode* from) {
        pipe.gridPerCU = 8;
        HashTable& ht = *q.hashTables[(size_t)o->hashTable];
        const std::string T = "ht" + std::to_string(ht.id);
        o->schema = o->child[0]->schema;
        for (auto& a : from->schema) o->schema.push_back(a);
        if (!q.requestAll) o->schema = prune(o->schema, requestOf[o]);
        std::vector<std::string> keyVars, probeKeyNames;
        openScope("{");
        probeKeys(o, T, keyVars, probeKeyNames);
        if (keyVars.size() != ht.keys.size()) failUnsupported("string join keys of different declared lengths");
        addArg(T + "_state", "const u32*", 0); addArg(T + "_words", "const i64*", 0); addArg(T + "_cap", "u64", 0);
        bool compScope = false;
        if (ht.hasCompBitmap && !compacted) {
            // the probe in front of which the compaction is cut (or a pipeline without one): the component's bit is tested right here
            int ki = -1, w = 0;
            for (size_t k = 0; k < o->exprs.size(); k++) {
                if (w == ht.compWord) { ki = (int)k; break; }
                Expr* r = o->exprs[k]->child->next;
                w += r->type.isString() ? (r->type.len + 7) / 8 : 1;
            }
            if (ki >= 0 && !o->exprs[(size_t)ki]->child->next->type.isString()) {
                Expr* r = o->exprs[(size_t)ki]->child->next;
                const std::string C = T + "_c";
                addArg(C + "_bm", "const u32*", 0); addArg(C + "_bmmin", "i64", (uint64_t)ht.cbMin); addArg(C + "_bmbits", "u64", (uint64_t)ht.cbBits);
                auto org = r->tag == RSQ_E_ATTRIBUTE ? symbolOrigin.find(r->symbol) : symbolOrigin.end();
                auto sym = r->tag == RSQ_E_ATTRIBUTE ? eg.symbols.find(r->symbol) : eg.symbols.end();
                bool already = false;
                for (auto& pf : bitmapPrefetch) already = already || pf.first == C;
                const std::string d = "(u64)(" + keyVars[(size_t)ht.compWord] + " - a." + C + "_bmmin)";
                if (!already && ht.cbBits <= (1 << 20) && org != symbolOrigin.end() && org->second == -1 && sym != eg.symbols.end() && sym->second.var.compare(0, 2, "v_") == 0) {
                    const int col = atoi(sym->second.var.c_str() + 2);      // (a small bitmap: the word arrives with the tile, for both rows of the lane)
                    bitmapPrefetch.push_back({C, col, false});
                    const std::string call = "rsq::bm_word(a." + C + "_bm, a." + C + "_bmmin, a." + C + "_bmbits, (i64)";
                    rowParams += ", const u32 pf_" + C;
                    rowArgsTail += ", " + call + "a.c" + std::to_string(col) + "[r])";
                    rowArgsTailGuarded += ", (valid ? " + call + "a.c" + std::to_string(col) + "[r]) : 0u)";
                    openScope("if (rsq::bit_of_word(pf_" + C + ", " + d + ", a." + C + "_bmbits)) {");
                } else openScope("if (rsq::bit_in(a." + C + "_bm, " + d + ", a." + C + "_bmbits)) {");
                compScope = true;
                selective = true;
                explainSteps.push_back("component bitmap of " + T + " tested in front of the probe");
            }
        }
        if (ht.hasCompBitmap && compacted && inStage2) {
            // the component's value on the probe side, if it is a column of this pipeline's scan: known at the top of stage 2, where the test goes
            int ki = -1, w = 0;
            for (size_t k = 0; k < o->exprs.size(); k++) {
                if (w == ht.compWord) { ki = (int)k; break; }
                Expr* r = o->exprs[k]->child->next;
                w += r->type.isString() ? (r->type.len + 7) / 8 : 1;
            }
            if (ki >= 0) {
                Expr* r = o->exprs[(size_t)ki]->child->next;
                auto org = r->tag == RSQ_E_ATTRIBUTE ? symbolOrigin.find(r->symbol) : symbolOrigin.end();
                auto sym = r->tag == RSQ_E_ATTRIBUTE ? eg.symbols.find(r->symbol) : eg.symbols.end();
                if (org != symbolOrigin.end() && org->second == -1 && sym != eg.symbols.end() && sym->second.var.compare(0, 2, "q_") == 0 && !r->type.isString()) {
                    addArg(T + "_c_bm", "const u32*", 0); addArg(T + "_c_bmmin", "i64", (uint64_t)ht.cbMin); addArg(T + "_c_bmbits", "u64", (uint64_t)ht.cbBits);
                    compFilters.push_back({T, r->symbol, sym->second.var, ht.cbBits});      // (compactThen places the test: in stage 1 if it can, else at the top of stage 2)
                    explainSteps.push_back("component bitmap of " + T + " tested in front of the compaction");
                }
            }
        }
        if (ht.hasBitmap) {
            // keys outside the build side's [min, max] or with a clear bit cannot match: skip the table altogether
            addArg(T + "_bm", "const u32*", 0); addArg(T + "_bmmin", "i64", (uint64_t)ht.bmMin); addArg(T + "_bmbits", "u64", (uint64_t)ht.bmBits);
            line("const u64 " + T + "_d = (u64)(" + keyVars[0] + " - a." + T + "_bmmin);");
            // When the probe key is a column of this pipeline's scan, its bitmap word is fetched by the scan skeleton for BOTH
            // rows of the lane (and every tile in flight) before the first row is processed, and handed to the row function:
            // the two row functions of a lane otherwise run one after the other, each with its own dependent load — a cache
            // round trip per row that nothing overlaps (TPC-H Q3's lineitem pipeline spent a quarter of its time there).
            int pfCol = -1;
            {
                Expr* r = o->exprs[0]->child->next;
                auto org = r->tag == RSQ_E_ATTRIBUTE ? symbolOrigin.find(r->symbol) : symbolOrigin.end();
                auto sym = r->tag == RSQ_E_ATTRIBUTE ? eg.symbols.find(r->symbol) : eg.symbols.end();
                // ... worth it when the table is clustered by the key (column statistics): the 64 lanes of a wave then read one or
                // two cache lines.  For keys in random order (orders.o_custkey) a wave's load touches 64 lines, and fetching for
                // the rows the filter in front would have dropped made TPC-H Q3's orders pipeline 30 % slower.
                if (!compacted && o->exprs.size() == 1 && org != symbolOrigin.end() && org->second == -1 && sym != eg.symbols.end() &&
                    sym->second.var.compare(0, 2, "v_") == 0 && !r->type.isString()) {
                    const int ci = pipe.src->findCol(r->symbol);
                    const int mode = 1;       // 0 never, 1 clustered keys (or gated, below), 2 always
                    if (ci >= 0 && (mode == 2 || (mode == 1 && pipe.src->cols[(size_t)ci].stats.valid && pipe.src->cols[(size_t)ci].stats.ascending)))
                        pfCol = atoi(sym->second.var.c_str() + 2);
                    // (Keys in random order behind a selection: fetching their bitmap words with the tile for the rows the selection passes
                    // was tried and measured no gain - TPC-H Q3's orders pipeline 0.306-0.308 against 0.305-0.315 ms for the query: what
                    // its probes cost, 34 of its 88 us, is the cache lines they move from the L2 - 7 M probes of a 187 KB bitmap, one
                    // 128-byte line each - not their latency.)
                }
                for (auto& pf : bitmapPrefetch) if (pf.first == T) pfCol = -1;        // (one probe per table and pipeline)
            }
            if (pfCol >= 0) {
                bitmapPrefetch.push_back({T, pfCol, ht.bmInterleaved});
                const std::string call = std::string(ht.bmInterleaved ? "rsq::bmi_load(a." : "rsq::bm_word(a.") + T + "_bm, a." + T + "_bmmin, a." + T + "_bmbits, (i64)";
                rowParams += ", const u32 pf_" + T;
                rowArgsTail += ", " + call + "a.c" + std::to_string(pfCol) + "[r])";
                rowArgsTailGuarded += ", (valid ? " + call + "a.c" + std::to_string(pfCol) + "[r]) : 0u)";
                openScope("if (" + T + "_d < a." + T + "_bmbits && ((pf_" + T + " >> (" + T + "_d & 31)) & 1u)) {");
            } else
            openScope("if (" + T + "_d < a." + T + "_bmbits && ((a." + T + "_bm[" + (ht.bmInterleaved ? "rsq::bmi_word(" + T + "_d)" : T + "_d >> 5") + "] >> (" + T + "_d & 31)) & 1u)) {");
            selective = true;
        }
        // the table walk (dependent random accesses) runs behind the wave compaction when the pipeline is selective
        const bool cut = compactThen(o, [&] {
            std::vector<std::string> kv2, names2;
            openScope("{");
            probeKeys(o, T, kv2, names2);
            probeTable(o, ht, T, kv2, names2);
            closeScope();
        });
        if (!cut) probeTable(o, ht, T, keyVars, probeKeyNames);
        if (ht.hasBitmap) closeScope();
        if (compScope) closeScope();
        closeScope();
    }

    // what a match exposes: the build side's values become symbols (hashjoin.h:146-147 / 204-205), then the parent consumes
    void consumeMatch(OpNode* o, HashTable& ht, const std::string& T, const std::vector<std::string>& keyVars,
                      const std::vector<std::string>& probeKeyNames) {
        int w = (int)ht.keys.size();
        for (auto& p : ht.payload) {
            std::string var = T + "_v" + std::to_string(w);
            line("const " + ExprGen::ctype(p.type) + " " + var + " = " + fromWord(wordAt(ht, T, w), p.type) + ";");
            eg.symbols[p.name] = Sym{var, p.type};
            symbolOrigin[p.name] = ht.id; symbolWord[p.name] = w;
            w++;
        }
        // build-side attributes that are key values: equal to this row's probe key, nothing to load
        for (auto& al : ht.keyAlias) {
            std::string var = T + "_a" + std::to_string(al.second) + "_" + std::to_string(w);
            line("const " + ExprGen::ctype(al.first.type) + " " + var + " = " + fromWord(keyVars[(size_t)al.second], al.first.type) + ";");
            eg.symbols[al.first.name] = Sym{var, al.first.type};
            symbolOrigin[al.first.name] = ht.id; symbolWord[al.first.name] = al.second;
            w++;
        }
        // probe-side key attributes are equal to the build keys of the matched entry
        for (size_t i = 0; i < probeKeyNames.size(); i++)
            if (!probeKeyNames[i].empty() && symbolOrigin.count(probeKeyNames[i]) && symbolOrigin[probeKeyNames[i]] == -1) probeKeyOf[probeKeyNames[i]] = {ht.id, (int)i};
        int prevMatch = matchSlotTable; bool prevMulti = multiMatchAbove;
        slotVar[ht.id] = T + "_s";
        if (o->singleMatch) matchSlotTable = ht.id; else { multiMatchAbove = true; }
        explainSteps.push_back(std::string("probe ") + T + (o->singleMatch ? " (single match)" : " (all matches)"));
        selective = true;                       // whatever follows a join probe sees only the matching rows
        {
            ProbeInScope ps{ht.id, o->singleMatch, ht.rankCapable, {}};
            for (Expr* eq : o->exprs) { Expr* r = eq->child->next; ps.keySymbols.push_back(r->tag == RSQ_E_ATTRIBUTE && !r->type.isString() ? r->symbol : std::string()); }
            probesInScope.push_back(ps);
        }
        consume(o->parent, o);
        probesInScope.pop_back();
        matchSlotTable = prevMatch; multiMatchAbove = prevMulti;
    }

    void probeTable(OpNode* o, HashTable& ht, const std::string& T, const std::vector<std::string>& keyVars,
                    const std::vector<std::string>& probeKeyNames) {
        if (ht.rankCapable && o->singleMatch) {
            // both forms of the table behind a uniform branch: the entry of a key whose bit is set (tested above) is entry
            // number rank(key) of the dictionary — or the first key-equal slot of the hash walk when the host kept the hash form
            addArg(T + "_rank", "u64", 0);
            line("u64 " + T + "_s = 0; bool " + T + "_hit = false;");
            openScope("if (a." + T + "_rank) {");
            {
                const int dbgRank = 0;      // (measurement only, wrong results: 1 the key offset, 2 its hash instead of the rank)
                const std::string dd = "(u64)(" + keyVars[0] + " - a." + T + "_bmmin)";
                if (dbgRank == 1) line(T + "_s = " + dd + " & (a." + T + "_cap - 1);");
                else if (dbgRank == 2) line(T + "_s = rsq::hash64(" + dd + ") & (a." + T + "_cap - 1);");
                else
                line(T + "_s = rsq::rank_of(a." + T + "_bm, (u64)(" + keyVars[0] + " - a." + T + "_bmmin));");
            }
            line(T + "_hit = true;");
            closeScope();
            openScope("else {");
            line("const u64 " + T + "_mask = a." + T + "_cap - 1;");
            line(T + "_s = " + slotOf(ht, T, keyVars) + ";");
            openScope("for (u64 " + T + "_n = 0; " + T + "_n <= " + T + "_mask; " + T + "_n++, " + T + "_s = (" + T + "_s + 1) & " + T + "_mask) {");
            line("const i64 " + T + "_kk = " + wordAt(ht, T, 0) + ";");
            line("if (" + T + "_kk == (i64)0x8000000000000000ull) break;");
            line("if (" + T + "_kk == " + keyVars[0] + ") { " + T + "_hit = true; break; }");
            closeScope();
            closeScope();
            openScope("if (" + T + "_hit) {");
            consumeMatch(o, ht, T, keyVars, probeKeyNames);
            closeScope();
            return;
        }
        if (ht.rankCapable) {
            // all matches of a table that may be a rank dictionary (unique build keys): the walk below in the hash form; in the rank
            // form the one entry of a key whose bit is set (tested above) - one pass through the same loop body
            addArg(T + "_rank", "u64", 0);
            line("const u64 " + T + "_mask = a." + T + "_cap - 1;");
            if (ht.setOnly) line("u64 " + T + "_s = a." + T + "_rank ? 0ull : " + slotOf(ht, T, keyVars) + ";");
            else line("u64 " + T + "_s = a." + T + "_rank ? rsq::rank_of(a." + T + "_bm, (u64)(" + keyVars[0] + " - a." + T + "_bmmin)) : " + slotOf(ht, T, keyVars) + ";");
            openScope("for (u64 " + T + "_n = 0; " + T + "_n <= " + T + "_mask; " + T + "_n++, " + T + "_s = (" + T + "_s + 1) & " + T + "_mask) {");
            line("bool " + T + "_eq = true;");
            openScope("if (!a." + T + "_rank) {");
            line("const i64 " + T + "_kk = " + wordAt(ht, T, 0) + ";");
            line("if (" + T + "_kk == (i64)0x8000000000000000ull) break;");
            line(T + "_eq = " + T + "_kk == " + keyVars[0] + ";");
            closeScope();
            openScope("if (" + T + "_eq) {");
            consumeMatch(o, ht, T, keyVars, probeKeyNames);
            closeScope();
            line("if (a." + T + "_rank) break;");
            closeScope();
            return;
        }
        line("const u64 " + T + "_mask = a." + T + "_cap - 1;");
        line("u64 " + T + "_s = " + slotOf(ht, T, keyVars) + ";");
        openScope("for (u64 " + T + "_n = 0; " + T + "_n <= " + T + "_mask; " + T + "_n++, " + T + "_s = (" + T + "_s + 1) & " + T + "_mask) {");
        std::string cond;
        if (ht.keyCas) {
            line("const i64 " + T + "_kk = " + wordAt(ht, T, 0) + ";");
            line("if (" + T + "_kk == (i64)0x8000000000000000ull) break;");
            cond = T + "_kk == " + keyVars[0];
        } else {
            line("if (a." + T + "_state[" + T + "_s] == 0u) break;");
            for (size_t i = 0; i < keyVars.size(); i++)
                cond += (i ? " && " : "") + wordAt(ht, T, (int)i) + " == " + keyVars[i];
        }
        openScope("if (" + cond + ") {");
        consumeMatch(o, ht, T, keyVars, probeKeyNames);
        if (o->singleMatch) line("break;");
        closeScope();
        closeScope();
    }
    std::map<std::string, std::pair<int, int>> probeKeyOf;   // probe-side key symbol -> (table, key word)
    // the join probes whose match is in scope (innermost last): table, single match?, the probe side's key symbols ("" where a key
    // is not a one-word attribute) — emitHashAggregation's functional dependencies
    struct ProbeInScope { int table; bool single; bool rankCapable; std::vector<std::string> keySymbols; };
    std::vector<ProbeInScope> probesInScope;

    // ---- aggregation (aggregation.h:240-295) ----------------------------------------------------
    void collectAccumulators(OpNode* o) {
        for (Expr* g : o->exprs2) q.pool.addId(g);
        for (Expr* s : o->splitAgg) q.pool.addId(s);
        q.accums.clear(); q.splitToAccum.clear();
        // accumulator 0: first input row of the group (drives the reference's emission order)
        q.accums.push_back(Accum{RSQ_E_MIN, "#firstrow", "row", Type(RSQ_BIGINT), 2});
        for (Expr* s : o->splitAgg) {
            Accum ac; ac.kind = s->tag; ac.type = s->type;
            switch (s->tag) {
                case RSQ_E_COUNT: ac.key = "COUNT"; ac.input = eg.emit(s); ac.merge = 0; ac.inputExpr = nullptr; break;
                case RSQ_E_SUM:
                    if (s->type.tag != RSQ_DECIMAL && s->type.tag != RSQ_BIGINT) failType("ADD code generation not implemented for datatype");
                    ac.key = "SUM" + structuralKey(s->child); ac.input = eg.emit(s); ac.merge = 0; ac.inputExpr = s->child; break;
                case RSQ_E_MIN: case RSQ_E_MAX:
                    if (s->type.tag != RSQ_DECIMAL && s->type.tag != RSQ_BIGINT && s->type.tag != RSQ_DATE)
                        failType("LESS_THAN code generation not implemented for datatype");
                    ac.key = std::string(s->tag == RSQ_E_MIN ? "MIN" : "MAX") + structuralKey(s->child);
                    ac.input = "((i64)(" + eg.emit(s) + "))"; ac.merge = s->tag == RSQ_E_MIN ? 2 : 3; ac.inputExpr = s->child; break;
                default: failType("Aggregation type not implemented in updateAggregates(..).");
            }
            int found = -1;
            for (size_t i = 1; i < q.accums.size(); i++) if (q.accums[i].key == ac.key) found = (int)i;
            if (found < 0) { q.accums.push_back(ac); found = (int)q.accums.size() - 1; }
            q.splitToAccum.push_back(found);
        }
        // word blocks ordered [min | max | sum] so that each segment reduces with ONE collective across GPUs
        q.accumSlot.assign(q.accums.size(), 0);
        int slot = 0;
        q.nMinBlocks = q.nMaxBlocks = 0;
        for (int m : {2, 3, 0}) {
            for (size_t i = 0; i < q.accums.size(); i++) if (q.accums[i].merge == m) q.accumSlot[i] = slot++;
            if (m == 2) q.nMinBlocks = slot; else if (m == 3) q.nMaxBlocks = slot - q.nMinBlocks;
        }
        q.nSumBlocks = (int64_t)q.accums.size() - q.nMinBlocks - q.nMaxBlocks;
    }

    bool tryDenseKeys(OpNode* o) {
        Table* t = pipe.src;
        q.denseKeys.clear();
        int64_t total = 1;
        for (Expr* g : o->exprs2) {
            if (g->tag != RSQ_E_ATTRIBUTE) return false;
            auto org = symbolOrigin.find(g->symbol);
            if (org == symbolOrigin.end() || org->second != -1) return false;      // not a column of this pipeline's scan
            int ci = t->findCol(g->symbol);
            if (ci < 0 || !t->cols[(size_t)ci].dptr) return false;
            const TableColumn& c = t->cols[(size_t)ci];
            if (c.type.isString()) return false;               // string keys: generic hash aggregation (bytes as key words)
            DenseKey k; k.expr = g; k.type = c.type;
            // (an empty SHARD of a table plans with the statistics of the whole table, like every other shard: Table::nRowsTotal)
            if (t->nRows == 0 && !c.stats.valid) { k.card = 1; k.min = 0; }      // empty input: no row reaches the aggregation
            else if (!c.stats.valid) return false;
            else if (!c.stats.distinctBytes.empty()) { k.byteSet = true; k.values = c.stats.distinctBytes; k.card = (int64_t)k.values.size(); }
            else {
                if (c.type.isString()) return false;
                k.min = c.stats.min;
                unsigned __int128 range = (unsigned __int128)((__int128)c.stats.max - (__int128)c.stats.min) + 1;
                if (range > (unsigned __int128)(1u << 24)) return false;
                k.card = (int64_t)range;
            }
            if (total > (int64_t)(1 << 24) / k.card) return false;
            total *= k.card;
            q.denseKeys.push_back(k);
        }
        int64_t stride = 1;
        for (size_t i = q.denseKeys.size(); i-- > 0;) { q.denseKeys[i].stride = stride; stride *= q.denseKeys[i].card; }
        q.denseGroups = total;
        return true;
    }

    bool tryJoinEntry(OpNode* o) {
        // Every group-by value is determined by the entry matched by a single-match probe of this pipeline
        // (a build-side payload value, or the probe key that equals the build key): the group IS the entry,
        // and the aggregates can live beside it.  (TPC-H Q3: l_orderkey = o_orderkey, o_orderdate,
        // o_shippriority all hang off the matched orders entry.)
        if (matchSlotTable < 0 || o->exprs2.empty()) return false;
        q.groupSource.clear();
        for (Expr* g : o->exprs2) {
            if (g->tag != RSQ_E_ATTRIBUTE) return false;
            if (g->type.isString()) return false;      // string keys go through the generic table (bytes as key words)
            auto org = symbolOrigin.find(g->symbol);
            if (org != symbolOrigin.end() && org->second == matchSlotTable) { q.groupSource.push_back(symbolWord[g->symbol]); continue; }
            auto pk = probeKeyOf.find(g->symbol);
            if (pk != probeKeyOf.end() && pk->second.first == matchSlotTable) { q.groupSource.push_back(pk->second.second); continue; }
            return false;
        }
        // a probe key counts only if ALL keys of the table are covered, otherwise two entries could share the group
        HashTable& ht = *q.hashTables[(size_t)matchSlotTable];
        for (size_t kw = 0; kw < ht.keys.size(); kw++)
            if (std::find(q.groupSource.begin(), q.groupSource.end(), (int)kw) == q.groupSource.end()) return false;
        q.aggTable = matchSlotTable;
        return true;
    }

    void consumeAggregation(OpNode* o, OpNode* from) {
        (void)from;
        collectAccumulators(o);
        const int W = (int)q.accums.size();
        std::string mode;
        const int forced = envInt("RSQ_AGG_MODE", 0, 0, 5);     // 5 = generic hash aggregation even where a dense id exists (tests)
        if (!(forced == 5 && !o->exprs2.empty()) && tryDenseKeys(o)) {
            const int64_t D = q.denseGroups, cells = D * W;
            // measured on MI355X (Q1 SF10, 42 cells): registers 0.47 ms, lane-private LDS 0.71 ms
            if ((cells <= 64 && forced == 0) || forced == 1) { q.aggMode = AggMode::DENSE_REG; if (cells > 64) failUnsupported("too many groups for register accumulators"); }
            else if ((cells <= 56 && forced == 0) || forced == 2) { q.aggMode = AggMode::DENSE_LDS_PRIVATE; if (cells > 56) failUnsupported("too many groups for lane-private LDS accumulators"); }
            else if ((cells <= 6144 && forced == 0) || forced == 3) { q.aggMode = AggMode::DENSE_LDS_SHARED; if (cells > 6144) failUnsupported("too many groups for an LDS table"); }
            else q.aggMode = AggMode::DENSE_GLOBAL;
            // The HBM-table forms could sit behind the compaction too; measured (200 M rows, 2^20 groups): 3 % faster at 1 %
            // selectivity, 15-25 % SLOWER at 10 / 50 % (the count / scatter passes pay for the queue without needing it),
            // so it stays off unless asked for.
            if (!(q.aggMode == AggMode::DENSE_GLOBAL && 0 &&
                  compactThen(o, [&] { collectAccumulators(o); emitDenseAggregation(o); })))
                emitDenseAggregation(o);
        } else if (forced != 5 && tryJoinEntry(o)) {
            q.aggMode = AggMode::AT_JOIN_ENTRY;
            emitJoinEntryAggregation(o);
        } else {
            q.aggMode = AggMode::HASH;
            // (behind the compaction the accumulator inputs must be emitted again: they name stage-2 values now)
            if (!compactThen(o, [&] { collectAccumulators(o); emitHashAggregation(o); })) emitHashAggregation(o);
        }
        pipe.sink = SinkKind::AGGREGATE;
    }

    // Generic hash aggregation (computed keys, wide domains): an open-addressing table in HBM keyed by the group
    // values, insert-or-find inside the kernel, aggregates beside the entries.  Slot protocol: state 0 empty ->
    // CAS to 1 (being written) -> keys stored -> fence -> 2 (ready); a lane that loses the CAS or meets state 1
    // looks at the slot again (see the note at the loop about keeping this safe inside one wave).
    void emitHashAggregation(OpNode* o) {
        pipe.gridPerCU = 8;
        std::unique_ptr<HashTable> ht(new HashTable());
        ht->id = (int)q.hashTables.size();
        const std::string T = "ht" + std::to_string(ht->id);
        const int W = (int)q.accums.size();
        ht->nAccBlocks = W;
        std::vector<std::string> keyVars;
        std::vector<std::pair<size_t, size_t>> charKeyWords;      // per CHAR(n) group value: [first, last] key word
        openScope("{");
        int k = 0;
        // ---- group values that are functions of other group values -------------------------------------------------------
        // A single-match probe hands every row with the same probe key the same entry, so the payload values of that entry are
        // functions of the key.  If the group-by list holds the key of such a table (its build-side key attributes, or the probe-
        // side attributes equal to them) — or the table is probed with values that are themselves determined this way — the other
        // group values taken from its entry cannot tell two groups apart: they are CARRIED (stored once, when the group is
        // created) instead of hashed and compared.  A probe for all matches does the same whenever its table is a bitmap-rank
        // dictionary in this execution (a.htN_rank: the build keys proved unique — the planner's list of unique attributes,
        // planner.h:218-241, misses c_custkey): the kernel then takes the short comparison, and the full one when the table fell
        // back to the hash form.  TPC-H Q10 groups by c_custkey and six more values hanging off the customer and nation entries,
        // 32 key words of which 31 are carried.  The group rows, and so the result, are the same.  RSQ_GROUP_FD=0 compares every
        // value as before.
        std::vector<bool> carried(o->exprs2.size(), false);
        std::string fdCond;                      // run-time condition of the dependencies ("" = they always hold)
        if (1) {
            std::set<std::string> groupSyms, determined;
            for (Expr* g : o->exprs2) if (g->tag == RSQ_E_ATTRIBUTE) groupSyms.insert(g->symbol);
            determined = groupSyms;
            std::set<int> detTables, condTables; std::set<std::string> covers;
            for (bool changed = true; changed;) {
                changed = false;
                for (auto& ps : probesInScope) {
                    if ((!ps.single && !ps.rankCapable) || detTables.count(ps.table)) continue;
                    HashTable& bt = *q.hashTables[(size_t)ps.table];
                    bool viaProbeKeys = !ps.keySymbols.empty();
                    for (auto& ks : ps.keySymbols) viaProbeKeys = viaProbeKeys && !ks.empty() && determined.count(ks);
                    bool viaCover = !bt.keys.empty();
                    std::vector<std::string> cv;
                    for (size_t kw = 0; kw < bt.keys.size() && viaCover; kw++) {
                        std::string hit;
                        for (auto& gs : groupSyms) {
                            auto org = symbolOrigin.find(gs);
                            if (org != symbolOrigin.end() && org->second == ps.table && symbolWord[gs] == (int)kw) { hit = gs; break; }
                            auto pk = probeKeyOf.find(gs);
                            if (pk != probeKeyOf.end() && pk->second.first == ps.table && pk->second.second == (int)kw) { hit = gs; break; }
                        }
                        if (hit.empty()) viaCover = false; else cv.push_back(hit);
                    }
                    if (!viaProbeKeys && !viaCover) continue;
                    detTables.insert(ps.table);
                    if (!ps.single) condTables.insert(ps.table);
                    if (!viaProbeKeys) covers.insert(cv.begin(), cv.end());
                    for (auto& so : symbolOrigin) if (so.second == ps.table) determined.insert(so.first);
                    changed = true;
                }
            }
            size_t kept = 0;
            for (size_t i = 0; i < o->exprs2.size(); i++) {
                Expr* g = o->exprs2[i];
                if (g->tag != RSQ_E_ATTRIBUTE) { kept++; continue; }
                auto org = symbolOrigin.find(g->symbol);
                carried[i] = org != symbolOrigin.end() && org->second >= 0 && detTables.count(org->second) && !covers.count(g->symbol);
                if (!carried[i]) kept++;
            }
            if (kept == 0) std::fill(carried.begin(), carried.end(), false);      // (cannot happen: a chain of dependencies ends in a kept value)
            for (int t : condTables) fdCond += (fdCond.empty() ? "" : " && ") + std::string("a.ht") + std::to_string(t) + "_rank != 0";
        }
        bool anyCarried = false;
        for (bool c : carried) anyCarried = anyCarried || c;
        q.groupSource.assign(o->exprs2.size(), 0);
        for (size_t gi = 0; gi < o->exprs2.size(); gi++) {
            if (carried[gi]) continue;
            Expr* g = o->exprs2[gi];
            const size_t w0 = keyVars.size();
            q.groupSource[gi] = (int)w0;                // first table word of this group value
            for (auto& kv : keyWords(g, T + "_g" + std::to_string(k++), false)) keyVars.push_back(kv);
            if (g->type.tag == RSQ_CHAR && g->type.len > 1) charKeyWords.push_back({w0, keyVars.size() - 1});
            for (size_t w = w0; w < keyVars.size(); w++)
                ht->keys.push_back({w == w0 ? expressionName(g) : expressionName(g) + "#" + std::to_string(w - w0), w == w0 && !g->type.isString() ? g->type : Type(RSQ_BIGINT)});
        }
        if (keyVars.empty()) failUnsupported("hash aggregation without group keys");
        const int K = (int)keyVars.size();
        // carried values: their words follow the key words in the table (HashTable::payload), written by the lane that creates the group
        struct Carried { Expr* g; std::string var, ctype; int firstWord, nWords; };
        std::vector<Carried> carriedVals;
        {
            int cw = K;
            for (size_t gi = 0; gi < o->exprs2.size(); gi++) {
                if (!carried[gi]) continue;
                Expr* g = o->exprs2[gi];
                const int nw = g->type.isString() ? (g->type.len + 7) / 8 : 1;
                q.groupSource[gi] = cw;
                carriedVals.push_back({g, eg.symbols[g->symbol].var, ExprGen::ctype(g->type), cw, nw});
                for (int w = 0; w < nw; w++)
                    ht->payload.push_back({w == 0 ? expressionName(g) : expressionName(g) + "#" + std::to_string(w), w == 0 && !g->type.isString() ? g->type : Type(RSQ_BIGINT)});
                cw += nw;
            }
        }
        for (int w = 1; w < W; w++) line("const i64 in" + std::to_string(w) + " = " + q.accums[(size_t)w].input + ";");
        addArg(T + "_state", "u32*", 0); addArg(T + "_words", "i64*", 0); addArg(T + "_cap", "u64", 0); addArg(T + "_count", "u32*", 0);
        addArg(T + "_acc", "u64*", 0);

        // A group's words next to each other (words[slot][w]) when it has several: creating a group with 32 words is then a few
        // cache lines instead of 32 stores a table-length apart (TPC-H Q10 at SF10: 380 K new groups per execution).
        int NWtab = K;
        for (auto& c : carriedVals) NWtab += c.nWords;
        ht->aos = NWtab > 1 && 1 != 0;
        const bool aggAos = ht->aos;
        auto aggWord = [&, NWtab, aggAos](int w) {
            return aggAos ? "a." + T + "_words[" + T + "_s * " + std::to_string(NWtab) + " + " + std::to_string(w) + "]"
                          : "a." + T + "_words[" + std::to_string(w) + " * a." + T + "_cap + " + T + "_s]";
        };

        // ---- LDS front table (per workgroup) ------------------------------------------------------------------------
        // Direct-mapped slots {state, key words, accumulators} in LDS (256 .. 1024, by their size) in front of the HBM table: a row whose group
        // already owns its slot is aggregated with LDS atomics and never leaves the CU; a row that finds the slot taken by
        // another group, or still being written, goes to the HBM table as before (no waiting, so no wave can block
        // itself).  At the end of the kernel every occupied slot is merged into the HBM table by the same upsert.  With
        // few groups (TPC-H Q12: 2, Q5: 5) nearly every row stays in LDS; with many, nearly every row pays one LDS probe.
        const int slotBytes = 8 * (K + W) + 4;
        int LS = 0;
        if (LS == 0) LS = slotBytes * 1024 <= 48 * 1024 ? 1024 : slotBytes * 512 <= 48 * 1024 ? 512 : 256;
        while (LS & (LS - 1)) LS &= LS - 1;           // power of two
        const bool lds = 1 && LS >= 64 && slotBytes * LS <= 48 * 1024 && !anyCarried;     // (a front-table slot holds no carried values to create its group with)
        if (lds) {
            stateDecl += "    u32* lc_state;\n    i64* lc_key;\n    u64* lc_acc;\n";
            prologue += "    __shared__ u32 s_lc_state[" + std::string("RSQ_LC_SLOTS") + "];\n    __shared__ i64 s_lc_key[" + std::to_string(K) + " * RSQ_LC_SLOTS" +
                        "];\n    __shared__ u64 s_lc_acc[" + std::to_string(W) + " * RSQ_LC_SLOTS];\n";
            prologue += "    st.lc_state = s_lc_state; st.lc_key = s_lc_key; st.lc_acc = s_lc_acc;\n";
            prologue += "    for (int i = threadIdx.x; i < RSQ_LC_SLOTS; i += blockDim.x) s_lc_state[i] = 0u;\n    __syncthreads();\n";
            pipe.extraLdsBytes += (8 * (K + W) + 4) * LS;
            pipe.ldsSlots = LS; pipe.ldsSlotBytes = 8 * (K + W) + 4;
            // (the slot count is a macro: the engine compiles the same text with a handful of slots once it knows that the query has a handful of
            // groups - TPC-H Q12: 2, Q5: 5 -, and the table no longer costs the scan its occupancy: engine.cpp launchPipeline)
            fileScope += "#ifndef RSQ_LC_SLOTS\n#define RSQ_LC_SLOTS " + std::to_string(LS) + "\n#endif\n";
            // flush (before the entry counter's flush below: the upserts count new entries)
            std::string f = "    __syncthreads();\n    for (int i = threadIdx.x; i < RSQ_LC_SLOTS; i += blockDim.x) {\n";
            f += "        if (st.lc_state[i] == 2u) " + T + "_upsert(a, st, (i64)st.lc_acc[i]";
            for (int i = 0; i < K; i++) f += ", st.lc_key[" + std::to_string(i) + " * RSQ_LC_SLOTS + i]";
            for (int w = 1; w < W; w++) f += ", (i64)st.lc_acc[" + std::to_string(w) + " * RSQ_LC_SLOTS + i]";
            f += ");\n    }\n";
            epilogue += f;
        }
        countPerThread(T);

        // ---- the HBM table's insert-or-find + update, as a function of (first row, key words, accumulator inputs) ---
        // Slot protocol: state 0 empty -> CAS to 1 (being written) -> keys stored -> fence -> 2 (ready); a lane that
        // loses the CAS or meets state 1 looks at the slot again.
        {
            std::string savedBody = body; const int savedIndent = indent;
            body.clear(); indent = 1;
            std::vector<std::string> kp;
            for (int i = 0; i < K; i++) kp.push_back("k" + std::to_string(i));
            line("const u64 " + T + "_mask = a." + T + "_cap - 1;");
            line("u64 " + T + "_s = " + hashOf(kp) + " & " + T + "_mask;");
            // Insert-or-find, written so that it cannot deadlock inside a wave: the lane that wins the CAS writes the keys and
            // publishes state 2 in a plain if-block that is followed by code every lane runs (the reload), so the publish
            // stays inside the loop body.  (With `if (won) {publish; hit} if (!hit) continue; ...; break;` the compiler threads
            // the winner straight to the loop exit, the structurizer parks it there until the whole wave has left the loop,
            // and the losers of the same wave spin on a slot that is never published.)
            // carried group values: their words are needed by the lane that creates a group — and, while the dependencies are not
            // certain (a table of the chain is in its hash form), by every lane for the full comparison
            int nCarriedWords = 0;
            for (auto& c : carriedVals) nCarriedWords += c.nWords;
            if (nCarriedWords) {
                line("const bool " + T + "_fd = " + (fdCond.empty() ? std::string("true") : fdCond) + ";");

            }
            line("u64 " + T + "_adv = 0; u32 " + T + "_spin = 0; bool " + T + "_found = false;");
            line("if (rsq::ld_agent(a.err) & (u32)rsq::ERR_HT_FULL) return;      // another lane found the table too small: this run is void");
            openScope("for (;;) {");
            line("u32 stt = rsq::ld_agent(&a." + T + "_state[" + T + "_s]);");
            openScope("if (stt == 0u) {");
            openScope("if (atomicCAS(&a." + T + "_state[" + T + "_s], 0u, 1u) == 0u) {");
            for (int i = 0; i < K; i++)
                line("rsq::st_agent(&" + aggWord(i) + ", " + kp[(size_t)i] + ");");
            // the carried group values of the new group (written by the lane that creates it; compared only in the full form)
            if (nCarriedWords) {
                // (with the dependencies certain the words go from their loads straight into the table, value by value: staged in
                // the array first, 31 words of TPC-H Q10's group values were 62 more live VGPRs - the kernel held 163 and ran three
                // waves per SIMD)
                // While the dependencies hold (T_fd) nobody READS the carried words inside this kernel - they are not compared, and the group
                // rows are gathered by the next kernel -, so they are PLAIN stores: the compiler merges neighbouring words into 16-byte
                // stores and nothing waits for a write-through to be acknowledged word by word.  Agent-scope stores only in the full form,
                // where other lanes compare them.  TPC-H Q10 at SF10 (380 K new groups of 31 carried words): the pipeline 559 -> 419 us.
                auto storeCarried = [&](bool plain, const std::string& tag) {
                    for (size_t ci = 0; ci < carriedVals.size(); ci++) {
                        const Carried& c = carriedVals[ci];
                        openScope("{");
                        std::vector<std::string> words = keyWords(c.g, T + "_n" + tag + std::to_string(ci), false);
                        for (int w = 0; w < c.nWords; w++)
                            line(plain ? aggWord(c.firstWord + w) + " = " + words[(size_t)w] + ";" : "rsq::st_agent(&" + aggWord(c.firstWord + w) + ", " + words[(size_t)w] + ");");
                        closeScope();
                    }
                };
                if (fdCond.empty()) storeCarried(true, "");
                else {
                    openScope("if (" + T + "_fd) {");
                    storeCarried(true, "p");
                    closeScope();
                    openScope("else {");
                    storeCarried(false, "");
                    closeScope();
                }
            }
            // The key (and carried) words must be visible before the state says "ready".  They are agent-scope stores (write-through
            // to the level all XCDs see); once the stores have been ACKNOWLEDGED (s_waitcnt vmcnt(0)) a reader that sees state 2
            // with its own agent-scope loads finds them.  A __threadfence() here instead — buffer_wbl2 + buffer_inv, tens of
            // microseconds under load — made every NEW group cost a cache write-back: TPC-H Q10 at SF10 creates 380 K groups and
            // spent 2.5 of its 2.9 ms there (device timestamps; round 3).  RSQ_HASH_FENCE=1 restores the fence.
            if (0) line("__threadfence();");
            else line("asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");");
            line("rsq::st_agent(&a." + T + "_state[" + T + "_s], 2u);");
            line("st.n_" + T + "++;");
            if (!charKeyWords.empty()) {
                // Groups are keyed by the exact bytes; the reference's CHAR equality ignores trailing spaces, so the host merges
                // such groups — which is only ever needed when some group value ends with a space.  The lane that creates a
                // group tells the host (once per group, nearly never): without the flag the host skips the merge and may take
                // the candidate path of ORDER BY ... LIMIT.  The last character is the top non-zero byte of the value's last
                // non-zero key word.
                std::string any;
                for (auto& r : charKeyWords) {
                    std::string last = kp[r.first];
                    for (size_t w = r.first + 1; w <= r.second; w++) last = "(" + kp[w] + " != 0 ? " + kp[w] + " : " + last + ")";
                    any += (any.empty() ? "" : " || ") + std::string("rsq::top_byte_is_space(") + last + ")";
                }
                line("if (" + any + ") atomicOr(a.err, (u32)rsq::NOTE_CHAR_GROUP_ENDS_WITH_SPACE);");
            }
            closeScope();
            line("stt = rsq::ld_agent(&a." + T + "_state[" + T + "_s]);      // our own publish, or whoever won the slot");
            closeScope();
            openScope("if (stt == 2u) {");
            std::string cond;
            for (int i = 0; i < K; i++)
                cond += (i ? " && " : "") + std::string("rsq::ld_agent(&") + aggWord(i) + ") == " + kp[(size_t)i];
            if (nCarriedWords && !fdCond.empty()) {
                // (the full comparison, while a table of the dependency chain is in its hash form: the values' words are made here, where
                // they are compared - kept in an array across the loop they were 62 live VGPRs for TPC-H Q10's 31 words)
                line("bool " + T + "_eq = " + cond + ";");
                openScope("if (" + T + "_eq && !" + T + "_fd) {");
                for (size_t ci = 0; ci < carriedVals.size(); ci++) {
                    const Carried& c = carriedVals[ci];
                    openScope("{");
                    std::vector<std::string> words = keyWords(c.g, T + "_m" + std::to_string(ci), false);
                    for (int w = 0; w < c.nWords; w++)
                        line(T + "_eq = " + T + "_eq && rsq::ld_agent(&" + aggWord(c.firstWord + w) + ") == " + words[(size_t)w] + ";");
                    closeScope();
                }
                closeScope();
                cond = T + "_eq";
            }
            openScope("if (" + cond + ") {");
            line(T + "_found = true;");
            line("break;");
            closeScope();
            line(T + "_s = (" + T + "_s + 1) & " + T + "_mask;");
            // A probe sequence of thousands of slots means the table is (nearly) full: linear probing degrades to a scan of
            // the table per row long before every slot is taken (1 M groups in 1 M slots: 90 ns per row, 2.2 s per 25 M rows).
            // Report "full" early; the host re-runs with a four times larger table and keeps the load below one half.
            line("if (++" + T + "_adv > (" + T + "_mask < 4096 ? " + T + "_mask : 4096)) { atomicOr(a.err, (u32)rsq::ERR_HT_FULL); break; }");
            closeScope();
            line("else if (++" + T + "_spin > (1u << 22)) { atomicOr(a.err, (u32)rsq::ERR_STUCK); break; }   // a slot another wave is writing");
            closeScope();
            // The updates, after the loop (the wave has reconverged): lanes of this wave that found the SAME slot are folded
            // into one update by their first lane while such sets are large (a few groups in the whole input); as soon as the
            // first set is small the group domain is wide and every lane updates its own slot.
            auto updates = [&](const std::string& slot, const std::string& members) {
                for (int w = 0; w < W; w++) {
                    const std::string in = "x" + std::to_string(w);
                    const std::string op = std::to_string(q.accums[(size_t)w].merge);
                    const std::string dst = "a." + T + "_acc + " + std::to_string(q.accumSlot[(size_t)w]) + " * a." + T + "_cap + " + slot;
                    if (members.empty()) line("rsq::global_merge_always<" + op + ">(" + dst + ", (u64)" + in + ");");
                    else line("{ const u64 r = rsq::subset_reduce<" + op + ">((u64)" + in + ", " + members + "); if (wl_lane == wl_leader) rsq::global_merge_always<" +
                              op + ">(" + dst + ", r); }");
                }
            };
            if (1) {
                line("const int wl_lane = (int)(threadIdx.x & 63);");
                line("bool wl_mine = " + T + "_found;");
                line("u64 wl_todo = __ballot(wl_mine);");
                openScope("while (wl_todo) {");
                line("const int wl_leader = __ffsll((long long)wl_todo) - 1;");
                line("const u64 wl_slot = rsq::readlane_u64(" + T + "_s, wl_leader);");
                line("const u64 wl_set = __ballot(wl_mine && " + T + "_s == wl_slot);");
                line("if (__popcll(wl_set) < 4) break;");
                updates("wl_slot", "wl_set");
                line("if (" + T + "_s == wl_slot) wl_mine = false;");
                line("wl_todo &= ~wl_set;");
                closeScope();
                openScope("if (wl_mine) {");
                updates(T + "_s", "");
                closeScope();
            } else {
                openScope("if (" + T + "_found) {");
                updates(T + "_s", "");
                closeScope();
            }
            std::string fn = "static RSQ_DEV void " + T + "_upsert(const Args& a, State& st, const i64 x0";
            for (int i = 0; i < K; i++) fn += ", const i64 k" + std::to_string(i);
            for (int w = 1; w < W; w++) fn += ", const i64 x" + std::to_string(w);
            for (auto& c : carriedVals) fn += ", const " + c.ctype + " " + c.var;          // (named like the row function's symbol: keyWords above refers to it)
            fn += ") {\n" + body + "}\n";
            helperFns += fn;
            body = savedBody; indent = savedIndent;
        }

        // ---- the row: LDS front table first, the HBM table otherwise ---------------------------------------------------
        std::string call = T + "_upsert(a, st, row";
        for (int i = 0; i < K; i++) call += ", " + keyVars[(size_t)i];
        for (int w = 1; w < W; w++) call += ", in" + std::to_string(w);
        for (auto& c : carriedVals) call += ", " + c.var;
        call += ");";
        if (lds) {
            line("bool " + T + "_done = false;");
            openScope("{");
            // up to four consecutive slots: two groups that map to the same slot would otherwise send one of them to the
            // HBM table for good — with few groups that is a handful of HBM words taking every update of a hot group
            // (64 groups, 1024 slots: 5.1 ms per 100 M rows against 2.9 ms for 1024 groups, before the probing)
            line("u32 ls = (u32)(" + hashOf(keyVars) + " >> 44) & (u32)(RSQ_LC_SLOTS - 1);");
            openScope("for (int lt = 0; lt < 4; lt++, ls = (ls + 1u) & (u32)(RSQ_LC_SLOTS - 1)) {");
            line("u32 lst = __hip_atomic_load(&st.lc_state[ls], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);");
            openScope("if (lst == 0u && atomicCAS(&st.lc_state[ls], 0u, 1u) == 0u) {");
            for (int i = 0; i < K; i++) line("st.lc_key[" + std::to_string(i) + " * RSQ_LC_SLOTS + ls] = " + keyVars[(size_t)i] + ";");
            for (int w = 0; w < W; w++) {
                const int m = q.accums[(size_t)w].merge;
                line("st.lc_acc[" + std::to_string(w) + " * RSQ_LC_SLOTS + ls] = " + (m == 0 ? "0ull" : m == 2 ? "0x7fffffffffffffffull" : m == 3 ? "0x8000000000000000ull" : "~0ull") + ";");
            }
            line("__hip_atomic_store(&st.lc_state[ls], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);");
            line("lst = 2u;");
            closeScope();
            line("if (lst != 2u) break;           // another lane is writing this slot: do not wait, take the HBM table");
            std::string eq;
            for (int i = 0; i < K; i++) eq += std::string(i ? " && " : "") + "st.lc_key[" + std::to_string(i) + " * RSQ_LC_SLOTS + ls] == " + keyVars[(size_t)i];
            openScope("if (" + eq + ") {");
            for (int w = 0; w < W; w++)
                line("rsq::lds_merge<" + std::to_string(q.accums[(size_t)w].merge) + ">(&st.lc_acc[" + std::to_string(w) + " * RSQ_LC_SLOTS + ls], (u64)(" +
                     (w == 0 ? std::string("row") : "in" + std::to_string(w)) + "));");
            line(T + "_done = true;");
            line("break;");
            closeScope();
            closeScope();
            closeScope();
            line("if (!" + T + "_done) " + call);
        } else line(call);
        closeScope();
        q.aggTable = ht->id;
        explainSteps.push_back("hash aggregation in " + T + " (" + std::to_string(ht->keys.size()) + " key word(s)" +
                               (anyCarried ? " + " + std::to_string(ht->payload.size()) + " carried word(s) of group values that depend on them" : "") + (lds ? ", LDS front table" : "") +
                               ") accumulators=" + std::to_string(W - 1) + " (of " + std::to_string(o->splitAgg.size()) + " in the reference)");
        q.hashTables.push_back(std::move(ht));
    }

    // Dense group id from the column statistics the table was created with.  The statistics are a promise about the data,
    // not a guarantee (rsq_table_create_device adopts caller-owned memory): every rank is checked, a value outside its
    // column's recorded domain raises ERR_GROUP_OVERFLOW and is counted into group 0 — no access leaves the table, and the
    // host fails the execution.
    std::string groupIdExpr() {
        std::string gid = "0";
        for (size_t ki = 0; ki < q.denseKeys.size(); ki++) {
            DenseKey& k = q.denseKeys[ki];
            std::string v = eg.emit(k.expr), rank;
            const std::string rv = "gk" + std::to_string(ki);
            // (columns the engine owns — uploaded, generated, loaded from '.tbl' — cannot change after their statistics were
            // taken: only adopted columns pay for the checks; TPC-H Q1's kernel is 6 % slower with them)
            bool check = true;
            if (k.expr->tag == RSQ_E_ATTRIBUTE) { const int ci = pipe.src->findCol(k.expr->symbol); if (ci >= 0 && pipe.src->cols[(size_t)ci].owned) check = false; }
            if (envInt("RSQ_CHECK_STATS", 0, 0, 1)) check = true;
            if (k.byteSet && !check) {
                rank = "0";
                for (size_t d = 1; d < k.values.size(); d++) {
                    std::string an = "k" + std::to_string(ki) + "_" + std::to_string(d);
                    addArg(an, "u64", k.values[d]);
                    rank += " + (int)((u8)(" + v + ") >= (u8)a." + an + ")";
                }
                line("const int " + rv + " = " + rank + ";");
            } else if (!k.byteSet && !check) {
                std::string an = "k" + std::to_string(ki) + "_min";
                addArg(an, "i64", (uint64_t)k.min);
                line("const int " + rv + " = (int)((i64)(" + v + ") - a." + an + ");");
            } else
            if (k.byteSet) {
                rank = "0";
                std::string member;
                for (size_t d = 0; d < k.values.size(); d++) {
                    std::string an = "k" + std::to_string(ki) + "_" + std::to_string(d);
                    addArg(an, "u64", k.values[d]);
                    if (d) rank += " + (int)((u8)(" + v + ") >= (u8)a." + an + ")";
                    member += std::string(d ? " | " : "") + "(int)((u8)(" + v + ") == (u8)a." + an + ")";
                }
                line("int " + rv + " = " + rank + ";");
                line("if (!(" + (member.empty() ? std::string("1") : member) + ")) { atomicOr(a.err, (u32)rsq::ERR_GROUP_OVERFLOW); " + rv + " = 0; }");
            } else {
                std::string an = "k" + std::to_string(ki) + "_min";
                addArg(an, "i64", (uint64_t)k.min);
                line("int " + rv + " = (int)((i64)(" + v + ") - a." + an + ");");
                line("if ((u64)((i64)(" + v + ") - a." + an + ") >= " + std::to_string((long long)k.card) + "ull) { atomicOr(a.err, (u32)rsq::ERR_GROUP_OVERFLOW); " + rv + " = 0; }");
            }
            gid += " + " + rv + " * " + std::to_string((long long)k.stride);
        }
        return gid;
    }

    static const char* identityOf(int merge) { return merge == 0 ? "0ull" : merge == 2 ? "0x7fffffffffffffffull" : "0x8000000000000000ull"; }

    std::string blockIdentityExpr(const std::string& blk) {
        return blk + " < " + std::to_string((long long)q.nMinBlocks) + " ? 0x7fffffffffffffffull : " + blk + " < " +
               std::to_string((long long)(q.nMinBlocks + q.nMaxBlocks)) + " ? 0x8000000000000000ull : 0ull";
    }

    // `stride` words between the cells of the table the kernel flushes into (1: the [block][group] table itself)
    void emitGlobalFlush(std::ostringstream& s, const std::string& count, const std::string& srcExpr, int64_t D, int stride = 1) {
        // padded flush: the stride is a macro, so that the same source also gives the unpadded kernel partial executions use
        const std::string at = stride == 1 ? "a.out + i" : "a.out + i * RSQ_OUT_STRIDE";
        s << "    for (int i = threadIdx.x; i < " << count << "; i += blockDim.x) {\n";
        s << "        const int blk = i / " << D << ";\n        const u64 v = " << srcExpr << ";\n";
        s << "        if (blk < " << q.nMinBlocks << ") rsq::global_merge<2>(" << at << ", v);\n";
        s << "        else if (blk < " << (q.nMinBlocks + q.nMaxBlocks) << ") rsq::global_merge<3>(" << at << ", v);\n";
        s << "        else rsq::global_merge<0>(" << at << ", v);\n    }\n";
    }

    // Form 3 of a large dense aggregation (rsq_device.h "staged partitioning"): the passing row becomes a PACKED record — the
    // group's index inside its partition and the accumulator inputs, each in as many bits as its column's statistics need —
    // handed to the workgroup's LDS rings; the first-row tracker is kept beside it (stage_track).  Available when the
    // record fits 128 bits and the partitions fit the rings (<= 256); the wider cases stay with form 2.
    void emitStagedScatter(int64_t D, int W, int gpp, int shift, int P) {
        if (P > 256 || !envInt("RSQ_STAGED", 1, 0, 1) || q.accums[0].merge != 2) return;
        struct Field { int w; int bits; int64_t min; bool check; int word, off; };
        std::vector<Field> fields;
        fields.push_back({-1, shift, 0, false, 0, 0});
        for (int w : pipe.partRecordInputs) {
            Field f{w, 64, 0, false, 0, 0};
            const Expr* e = q.accums[(size_t)w].inputExpr;
            if (e && e->tag == RSQ_E_ATTRIBUTE) {
                const int ci = pipe.src->findCol(e->symbol);
                if (ci >= 0 && pipe.src->cols[(size_t)ci].stats.valid) {
                    const TableColumn& c = pipe.src->cols[(size_t)ci];
                    const uint64_t range = (uint64_t)c.stats.max - (uint64_t)c.stats.min;
                    int bits = 1; while (bits < 64 && (range >> bits) != 0) bits++;
                    if (bits < 64) { f.bits = bits; f.min = c.stats.min; f.check = !c.owned || envInt("RSQ_CHECK_STATS", 0, 0, 1); }
                }
            }
            fields.push_back(f);
        }
        int used[2] = {0, 0};
        for (auto& f : fields) {
            int wd = 0;
            while (wd < 2 && used[wd] + f.bits > 64) wd++;
            if (wd == 2) return;                                  // wider than two words: form 2
            f.word = wd; f.off = used[wd]; used[wd] += f.bits;
        }
        const int RECW = used[1] ? 2 : 1;
        const int ncolsNow = (int)colTypes.size();
        // rows per thread and round: 4 (2 for wide rows: registers); 8 when few rows are expected to pass - the rounds' barriers then
        // weigh more than the records (1.25 B rows, 2^20 groups: 10 % 7.54 -> 7.15 ms; at 50 % 8 rows cost 11.1 instead of 9.6 ms)
        const int RPT = ncolsNow <= 6 ? (!leadCond.empty() && leadPass <= 0.15 ? 8 : 4) : 2;
        pipe.staged = true; pipe.stagedRecWords = RECW; pipe.stagedRows = RPT;
        const std::string Ps = std::to_string(P), Rs = std::to_string(RECW), Ts = std::to_string(RPT);
        const std::string LDS = "rsq::StageLds<" + Rs + ", " + Ps + ">";
        line("#elif RSQ_AGG_VARIANT == 3");
        openScope("{");
        line("const u32 sp_p = (u32)(gid >> " + std::to_string(shift) + ");");
        line("if (a.sp_mode) atomicAdd(&st.sp->tail[sp_p], 1u);          // counting only: exact region sizes after an overflow");
        openScope("else {");
        line("if ((u64)row < st.sp_wm) rsq::stage_track(*st.sp, a.out + " + std::to_string((long long)(q.accumSlot[0] * D)) + " + gid, row);");
        for (int wd = 0; wd < RECW; wd++) {
            std::string ex;
            for (auto& f : fields) {
                if (f.word != wd) continue;
                std::string v;
                if (f.w < 0) v = "(u64)(gid & " + std::to_string(gpp - 1) + ")";
                else {
                    const std::string in = "in" + std::to_string(f.w);
                    if (f.bits == 64) v = "(u64)" + in;
                    else {
                        const std::string an = "sp_min" + std::to_string(f.w);
                        addArg(an, "i64", (uint64_t)f.min);
                        const std::string mask = std::to_string((unsigned long long)((1ull << f.bits) - 1ull)) + "ull";
                        if (f.check) line("if ((u64)(" + in + " - a." + an + ") > " + mask + ") atomicOr(a.err, (u32)rsq::ERR_GROUP_OVERFLOW);");
                        v = "((u64)(" + in + " - a." + an + ") & " + mask + ")";
                    }
                }
                if (f.off) v = "(" + v + " << " + std::to_string(f.off) + ")";
                ex += (ex.empty() ? "" : " | ") + v;
            }
            line("st.sp_rec[SP_SLOT * " + Rs + " + " + std::to_string(wd) + "] = " + (ex.empty() ? std::string("0ull") : ex) + ";");
        }
        line("st.sp_p[SP_SLOT] = sp_p;");
        line("st.sp_k[SP_SLOT] = atomicAdd(&st.sp->tail[sp_p], 1u);");
        line("st.sp_pending |= 1u << SP_SLOT;");
        closeScope();
        closeScope();
        addArg("sp_base", "const u64*", 0); addArg("sp_cap", "const u32*", 0); addArg("sp_ctl", "rsq::StageCtl*", 0);
        addArg("sp_counts", "u32*", 0); addArg("sp_rec", "u64*", 0); addArg("sp_mode", "u32", 0);
        stateDecl += "#if RSQ_AGG_VARIANT == 3\n    " + LDS + "* sp;\n    u64 sp_wm;\n    u64 sp_rec[" + std::to_string(RPT * RECW) + "];\n    u32 sp_k[" + Ts +
                     "];\n    u32 sp_p[" + Ts + "];\n    u32 sp_pending;\n#endif\n";
        prologue += "#if RSQ_AGG_VARIANT == 3\n    __shared__ " + LDS + " s_stage;\n    rsq::stage_init(s_stage, a.sp_base, a.sp_cap, a.sp_ctl);\n";
        prologue += "    st.sp = &s_stage; st.sp_pending = 0u; st.sp_wm = ~0ull;\n#endif\n";
        // ---- aggregation of one partition's records (all workgroups' regions of it) in an LDS table ----
        std::ostringstream k;
        auto& A3 = pipe.argsStagedAgg;
        A3.push_back({"sp_rec", "const u64*", 0});
        A3.push_back({"sp_base", "const u64*", 0});
        A3.push_back({"sp_cap", "const u32*", 0});
        A3.push_back({"sp_counts", "const u32*", 0});
        A3.push_back({"sp_nwg", "u32", 0});
        A3.push_back({"out", "u64*", 0});
        for (auto& f : fields) if (f.w >= 0 && f.bits < 64) A3.push_back({"sp_min" + std::to_string(f.w), "i64", (uint64_t)f.min});
        k << "// generated by resql_amd/csrc/codegen.cpp: aggregation of one partition of packed records in an LDS table\n";
        k << "#include \"rsq_device.h\"\nstruct Args {\n";
        for (auto& a : A3) k << "    " << a.ctype << " " << a.name << ";\n";
        k << "};\n";
        // accumulator blocks 1 .. W-1 of the table (block 0, the first row, is the tracker's)
        k << "static RSQ_DEV void merge_record(const Args& a, u64* tab";
        for (int wd = 0; wd < RECW; wd++) k << ", const u64 w" << wd;
        k << ") {\n";
        for (auto& f : fields) {
            std::string v = "w" + std::to_string(f.word);
            if (f.off) v = "(" + v + " >> " + std::to_string(f.off) + ")";
            if (f.bits < 64) v = "(" + v + " & " + std::to_string((unsigned long long)((1ull << f.bits) - 1ull)) + "ull)";
            if (f.w < 0) k << "    const int g = (int)" << v << ";\n";
            else if (f.bits < 64) k << "    const u64 in" << f.w << " = (u64)((i64)" << v << " + a.sp_min" << f.w << ");\n";
            else k << "    const u64 in" << f.w << " = " << v << ";\n";
        }
        for (int w = 1; w < W; w++) {
            std::string in = "(u64)" + q.accums[(size_t)w].input;             // a constant (COUNT's 1) unless it travels
            for (auto& f : fields) if (f.w == w) in = "in" + std::to_string(w);
            k << "    rsq::lds_merge<" << q.accums[(size_t)w].merge << ">(&tab[" << (int64_t)(w - 1) * gpp << " + g], " << in << ");\n";
        }
        k << "}\n";
        k << "extern \"C\" __global__ void __launch_bounds__(1024) rsq_staged_agg(Args a) {\n";
        k << "    __shared__ u64 s_tab[" << (int64_t)(W - 1) * gpp << "];\n";
        k << "    for (int i = threadIdx.x; i < " << (int64_t)(W - 1) * gpp << "; i += blockDim.x) { const int w = 1 + (i >> " << shift << "); s_tab[i] = ";
        for (int w = 1; w < W; w++) k << (w > 1 ? " : " : "") << (w < W - 1 ? "w == " + std::to_string(w) + " ? " : "") << identityOf(q.accums[(size_t)w].merge);
        k << "; }\n    __syncthreads();\n";
        k << "    const int p = blockIdx.x, lane = threadIdx.x & 63;\n";
        k << "    const u64 base = a.sp_base[p]; const u32 cap = a.sp_cap[p];\n";
        k << "    for (u32 wg = threadIdx.x >> 6; wg < a.sp_nwg; wg += blockDim.x >> 6) {\n";
        k << "        const u64 st = base + (u64)wg * cap;\n        const u32 cnt = min(a.sp_counts[(u64)wg * " << P << " + p], cap);\n";
        const int AU = 4;
        // AU 16-byte loads per lane in flight (one per lane leaves a CU with 16 KB outstanding: 4.8 TB/s; four: see DESIGN §4)
        const int step = RECW == 1 ? 128 : 64;              // records one wave-load covers
        k << "        for (u32 i0 = 0; i0 < cnt; i0 += " << AU * step << ") {\n";
        k << "            rsq::u64x2 v[" << AU << "];\n";
        k << "#pragma unroll\n            for (int u = 0; u < " << AU << "; u++) {\n";
        k << "                const u32 i = i0 + u * " << step << " + lane * " << (RECW == 1 ? 2 : 1) << ";\n";
        k << "                if (i < cnt) v[u] = *reinterpret_cast<const rsq::u64x2*>(a.sp_rec + (st + i) * " << RECW << ");\n            }\n";
        k << "#pragma unroll\n            for (int u = 0; u < " << AU << "; u++) {\n";
        k << "                const u32 i = i0 + u * " << step << " + lane * " << (RECW == 1 ? 2 : 1) << ";\n";
        if (RECW == 1) k << "                if (i < cnt) merge_record(a, s_tab, v[u].x);\n                if (i + 1 < cnt) merge_record(a, s_tab, v[u].y);\n";
        else k << "                if (i < cnt) merge_record(a, s_tab, v[u].x, v[u].y);\n";
        k << "            }\n        }\n";
        k << "    }\n    __syncthreads();\n";
        k << "    for (int i = threadIdx.x; i < " << (int64_t)(W - 1) * gpp << "; i += blockDim.x) {\n";
        k << "        const int w = 1 + (i >> " << shift << ");\n";
        k << "        const i64 g = (i64)p * " << gpp << " + (i & " << (gpp - 1) << ");\n";
        // accumulator w lives in block accumSlot[w] of the [block][group] table
        k << "        const i64 blk = ";
        for (int w = 1; w < W; w++) k << (w > 1 ? " : " : "") << (w < W - 1 ? "w == " + std::to_string(w) + " ? " : "") << "(i64)" << q.accumSlot[(size_t)w];
        k << ";\n        if (g < " << D << ") a.out[blk * " << D << " + g] = s_tab[i];\n    }\n}\n";
        pipe.sourceStagedAgg = k.str();
    }

    void emitDenseAggregation(OpNode* o) {
        const int64_t D = q.denseGroups;
        const int W = (int)q.accums.size();
        line("const int gid = " + groupIdExpr() + ";");
        for (int w = 1; w < W; w++) line("const i64 in" + std::to_string(w) + " = " + q.accums[(size_t)w].input + ";");
        auto inOf = [&](int w) { return w == 0 ? std::string("row") : "in" + std::to_string(w); };
        addArg("out", "u64*", 0);
        std::ostringstream ep;
        if (q.aggMode == AggMode::DENSE_REG) {
            // accumulators in VGPRs, branch-free per-group update.  (An `if (gid == g) acc_g += x` chain gets its
            // common tail sunk by the compiler into one store through a selected pointer, which forces every
            // accumulator into scratch.)
            for (int w = 0; w < W; w++)
                for (int64_t g = 0; g < D; g++)
                    stateDecl += "    i64 acc_" + std::to_string(w) + "_" + std::to_string((long long)g) + " = (i64)" + identityOf(q.accums[(size_t)w].merge) + ";\n";
            const bool branchy = (D > 1 ? 1 : 0) == 1;
            for (int64_t g = 0; g < D; g++) {
                if (branchy) {
                    // EXEC-masked update of one group's accumulators (2 VALU per 64-bit add instead of the
                    // select form's 4).  The distinct asm comment at the end of every block is load-bearing: without
                    // it the compiler sinks the identical tails of the blocks into one store through a selected
                    // pointer, which forces all accumulators into scratch memory.
                    openScope("if (gid == " + std::to_string((long long)g) + ") {");
                    for (int w = 0; w < W; w++) {
                        std::string acc = "st.acc_" + std::to_string(w) + "_" + std::to_string((long long)g), in = inOf(w);
                        int m = q.accums[(size_t)w].merge;
                        if (m == 0) line(acc + " = rsq::add(" + acc + ", " + in + ");");
                        else if (m == 2) line(acc + " = " + in + " < " + acc + " ? " + in + " : " + acc + ";");
                        else line(acc + " = " + in + " > " + acc + " ? " + in + " : " + acc + ";");
                    }
                    line("asm volatile(\"; rsq group " + std::to_string((long long)g) + "\");");
                    closeScope();
                    continue;
                }
                openScope("{");
                line("const bool m = gid == " + std::to_string((long long)g) + ";");
                for (int w = 0; w < W; w++) {
                    std::string acc = "st.acc_" + std::to_string(w) + "_" + std::to_string((long long)g), in = inOf(w);
                    int m = q.accums[(size_t)w].merge;
                    if (m == 0) line(acc + " = rsq::add(" + acc + ", m ? " + in + " : (i64)0);");
                    else if (m == 2) line(acc + " = (m && " + in + " < " + acc + ") ? " + in + " : " + acc + ";");
                    else line(acc + " = (m && " + in + " > " + acc + ") ? " + in + " : " + acc + ";");
                }
                closeScope();
            }
            const bool dbgTail = envInt("RSQ_DEBUG_TAIL", 0, 0, 1) != 0;      // (measurement only: device timestamps of the epilogue's stages)
            auto stamp = [&](int k) { if (dbgTail) ep << "    if (a.dbg && threadIdx.x == 0) a.dbg[(u64)blockIdx.x * 8 + " << k << "] = (u64)wall_clock64();\n"; };
            if (dbgTail) { addArg("dbg", "u64*", 0); prologue += "    if (a.dbg && threadIdx.x == 0) a.dbg[(u64)blockIdx.x * 8 + 0] = (u64)wall_clock64();\n"; }
            stamp(1);
            ep << "    __shared__ u64 s_acc[" << W * D << "];\n";
            if (1) {
                // The workgroup's accumulators meet LANE BY LANE first: every wave merges its cells into s_lane[cell][lane] (LDS
                // atomics, no two lanes on one word), then each wave folds a share of the cells across the 64 lanes (DPP, rsq_device.h
                // wave_reduce_to_lane63).  One cross-lane reduction per cell and workgroup instead of one per cell and WAVE: the
                // reductions of TPC-H Q1's 42 cells in all 8 waves took 14-20 us of every launch as ds_bpermute butterflies and
                // still 9-11 us as DPP (device timestamps, RSQ_DEBUG_TAIL).
                const int64_t cells = W * D;
                ep << "    __shared__ u64 s_lane[" << cells * 64 << "];\n";
                ep << "    for (int i = threadIdx.x; i < " << cells * 64 << "; i += blockDim.x) { const int blk = (i >> 6) / " << D << "; s_lane[i] = " << blockIdentityExpr("blk") << "; }\n";
                ep << "    __syncthreads();\n";
                for (int w = 0; w < W; w++)
                    for (int64_t g = 0; g < D; g++)
                        ep << "    rsq::lds_merge<" << q.accums[(size_t)w].merge << ">(&s_lane[" << (q.accumSlot[(size_t)w] * D + g) * 64 << " + (threadIdx.x & 63)], (u64)st.acc_" << w << "_" << g << ");\n";
                ep << "    __syncthreads();\n";
                ep << "    for (int c = threadIdx.x >> 6; c < " << cells << "; c += blockDim.x >> 6) {\n";
                ep << "        const int blk = c / " << D << ";\n        const u64 v = s_lane[c * 64 + (threadIdx.x & 63)];\n";
                ep << "        const u64 r = blk < " << q.nMinBlocks << " ? rsq::wave_reduce_to_lane63<2>(v) : blk < " << (q.nMinBlocks + q.nMaxBlocks)
                   << " ? rsq::wave_reduce_to_lane63<3>(v) : rsq::wave_reduce_to_lane63<0>(v);\n";
                ep << "        if ((threadIdx.x & 63) == 63) s_acc[c] = r;\n    }\n";
                ep << "    __syncthreads();\n";
            } else {
            ep << "    for (int i = threadIdx.x; i < " << W * D << "; i += blockDim.x) { const int blk = i / " << D << "; s_acc[i] = " << blockIdentityExpr("blk") << "; }\n";
            ep << "    __syncthreads();\n";
            for (int w = 0; w < W; w++)
                for (int64_t g = 0; g < D; g++)
                    ep << "    rsq::wave_to_lds<" << q.accums[(size_t)w].merge << ">(&s_acc[" << (q.accumSlot[(size_t)w] * D + g) << "], (u64)st.acc_" << w << "_" << g << ");\n";
            ep << "    __syncthreads();\n";
            }
            // The workgroups flush into a PADDED copy of the table, one cell per 64-byte line (engine.cpp unpads it):
            // memory-side atomics serialise per line, and the 42 cells of TPC-H Q1 otherwise share six lines.
            q.aggPad = 8;
            stamp(2);
            emitGlobalFlush(ep, std::to_string((long long)(W * D)), "s_acc[i]", D, q.aggPad);
            stamp(3);
            // The step in ONE launch (engine.cpp runFusedStep): the workgroup that flushes last hands the finished table to
            // the host — plain stores into host-mapped pinned memory (a full execution) or into the partial table the
            // group-by merge reads (a multi-GPU step) — together with the device error word, and puts the working table,
            // the error word and the ticket back to their identities for the next execution.  That replaces the D2D
            // copy that readied the table, the error-word memset and the two read-back copies of every step.
            // Order: a thread waits until its flush atomics have been performed (s_waitcnt vmcnt(0): device-scope atomics are
            // coherent across the XCDs once performed) before the workgroup takes its ticket, so the holder of the last ticket
            // finds every cell final; it reads with agent-scope exchanges, which execute where the flush atomics did.  A
            // release fence instead (__threadfence: buffer_wbl2 + buffer_inv in every wave) cost 30 us per launch — more than
            // the copies it was meant to save.
            addArg("fin_out", "u64*", 0);
            addArg("fin_err", "u64*", 0);
            addArg("fin_ticket", "u32*", 0);
            addArg("fin_seq", "u64", 0);
            ep << "    if (a.fin_out) {\n        __shared__ u32 s_last;\n        asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n        __syncthreads();\n";
            ep << "        if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(a.fin_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u ? 1u : 0u;\n        __syncthreads();\n";
            if (dbgTail) ep << "        if (a.dbg && threadIdx.x == 0) a.dbg[(u64)blockIdx.x * 8 + 4] = (u64)wall_clock64();\n";
            ep << "        if (s_last) {\n";
            ep << "            for (int i = threadIdx.x; i < " << W * D << "; i += blockDim.x) {\n                const int blk = i / " << D << ";\n";
            ep << "                const u64 idv = " << blockIdentityExpr("blk") << ";\n";
            ep << "                a.fin_out[i] = __hip_atomic_exchange(a.out + i" << (q.aggPad > 1 ? " * RSQ_OUT_STRIDE" : "") << ", idv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n            }\n";
            ep << "            if (threadIdx.x == 0) {\n                a.fin_err[0] = (u64)atomicExch(a.err, 0u);\n";
            ep << "                __hip_atomic_store(a.fin_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);\n            }\n";
            // a full execution is announced to the polling host by a sequence number behind the table: written after every
            // thread's table stores have been acknowledged, with a system-scope release (one wave, once per launch)
            if (dbgTail) ep << "            if (a.dbg && threadIdx.x == 0) a.dbg[(u64)blockIdx.x * 8 + 5] = (u64)wall_clock64();\n";
            ep << "            if (a.fin_seq) {\n                asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n                __syncthreads();\n";
            if (dbgTail) ep << "                if (a.dbg && threadIdx.x == 0) a.dbg[(u64)blockIdx.x * 8 + 6] = (u64)wall_clock64();\n";
            ep << "                if (threadIdx.x == 0) __hip_atomic_store(a.fin_err + 1, a.fin_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);\n";
            if (dbgTail) ep << "                if (a.dbg && threadIdx.x == 0) a.dbg[(u64)blockIdx.x * 8 + 7] = (u64)wall_clock64();\n";
            ep << "            }\n";
            ep << "        }\n    }\n";
            // One 512-thread workgroup per CU: the same 8 waves per CU as 2 x 256, but half as many workgroups flush.
            // The flush is 42 atomics per workgroup (TPC-H Q1) onto six 64-byte lines, where they serialise: going from
            // 512 to 256 workgroups took 8 us off the 352 us SF10 kernel and 9 off the 67 us SF1 kernel (1024 and 2048
            // workgroups: +25 / +75 us).  A slab-per-workgroup + ticket + last-workgroup reduction was tried instead of
            // the atomics and measured 28 us SLOWER (write-through slab stores, a serial reducer), so it is not here.
            pipe.blockThreads = 512;
        } else if (q.aggMode == AggMode::DENSE_LDS_PRIVATE) {
            // one private copy of the [block][group] table per LANE in LDS, laid out [cell][thread] so that a
            // wave's 64 accesses to one cell are 64 consecutive 8-byte words: conflict-free, no contention,
            // one ds_add_u64 / ds_min_i64 per accumulator and row instead of a select+add per group.
            const int64_t cells = W * D;
            pipe.blockThreads = cells <= 28 ? 256 : 128;
            const int B = pipe.blockThreads;
            prologue += "    __shared__ u64 s_priv[" + std::to_string((long long)(cells * B)) + "];\n";
            prologue += "    for (int i = threadIdx.x; i < " + std::to_string((long long)(cells * B)) + "; i += blockDim.x) { const int blk = (i / " +
                        std::to_string(B) + ") / " + std::to_string((long long)D) + "; s_priv[i] = " + blockIdentityExpr("blk") + "; }\n";
            prologue += "    __syncthreads();\n    st.priv = s_priv + threadIdx.x;\n";
            stateDecl += "    u64* priv;\n";
            for (int w = 0; w < W; w++) {
                std::string cell = "st.priv + (" + std::to_string((long long)(q.accumSlot[(size_t)w] * D)) + " + gid) * " + std::to_string(B);
                int m = q.accums[(size_t)w].merge;
                if (m == 0) line("rsq::lds_merge<0>(" + cell + ", (u64)" + inOf(w) + ");");
                else line("rsq::lds_merge<" + std::to_string(m) + ">(" + cell + ", (u64)" + inOf(w) + ");");
            }
            // flush: every wave folds whole cells: lanes stride over the B private copies, butterfly, one atomic
            ep << "    __syncthreads();\n";
            ep << "    for (int c = (threadIdx.x >> 6); c < " << cells << "; c += (blockDim.x >> 6)) {\n";
            ep << "        const int blk = c / " << D << ";\n        const int lane = threadIdx.x & 63;\n";
            ep << "        u64 v = s_priv[c * " << B << " + lane];\n";
            ep << "        for (int j = lane + 64; j < " << B << "; j += 64) {\n            const u64 o = s_priv[c * " << B << " + j];\n";
            ep << "            if (blk < " << q.nMinBlocks << ") v = (i64)o < (i64)v ? o : v; else if (blk < " << (q.nMinBlocks + q.nMaxBlocks)
               << ") v = (i64)o > (i64)v ? o : v; else v += o;\n        }\n";
            ep << "        if (blk < " << q.nMinBlocks << ") { v = (u64)rsq::wave_min_i64((i64)v); if (lane == 0) rsq::global_merge<2>(a.out + c, v); }\n";
            ep << "        else if (blk < " << (q.nMinBlocks + q.nMaxBlocks) << ") { v = (u64)rsq::wave_max_i64((i64)v); if (lane == 0) rsq::global_merge<3>(a.out + c, v); }\n";
            ep << "        else { v = rsq::wave_sum(v); if (lane == 0) rsq::global_merge<0>(a.out + c, v); }\n    }\n";
        } else if (q.aggMode == AggMode::DENSE_LDS_SHARED) {
            // one [block][group] table per workgroup in LDS, LDS atomics (many groups => little contention),
            // flushed once per workgroup with global atomics
            const int64_t cells = W * D;
            prologue += "    __shared__ u64 s_tab[" + std::to_string((long long)cells) + "];\n";
            prologue += "    for (int i = threadIdx.x; i < " + std::to_string((long long)cells) + "; i += blockDim.x) { const int blk = i / " +
                        std::to_string((long long)D) + "; s_tab[i] = " + blockIdentityExpr("blk") + "; }\n    __syncthreads();\n    st.tab = s_tab;\n";
            stateDecl += "    u64* tab;\n";
            for (int w = 0; w < W; w++)
                line("rsq::lds_merge<" + std::to_string(q.accums[(size_t)w].merge) + ">(st.tab + " + std::to_string((long long)(q.accumSlot[(size_t)w] * D)) +
                     " + gid, (u64)" + inOf(w) + ");");
            ep << "    __syncthreads();\n";
            emitGlobalFlush(ep, std::to_string((long long)cells), "s_tab[i]", D);
        } else {   // DENSE_GLOBAL: the table lives in HBM
            // Three forms of the same pipeline, one source (RSQ_AGG_VARIANT):
            //  0 direct     every passing row merges into the table with HBM atomics.  Those execute at the memory side,
            //               ≈25 G requests/s chip-wide, so this form is atomic-bound once many rows pass the filter.
            //  1 count      per (workgroup, partition) row counts in LDS (partition = group id / groups-per-partition);
            //               with a.tile_step > 1 it samples every n-th tile: the engine's selectivity estimate.
            //  2 scatter    each passing row becomes a record (group-in-partition, row, accumulator inputs) written to
            //               its partition's region at a position taken from a workgroup-local LDS cursor that starts
            //               at the exclusive prefix of the counts: no HBM atomics at all.
            // A fourth kernel (rsq_part_agg, emitted below) aggregates each partition in an LDS table and stores the
            // finished groups with plain stores.  The engine picks direct or partitioned per execution from the counts.
            pipe.gridPerCU = 8;
            int gpp = 1;
            while ((int64_t)gpp * 2 * W * 8 <= 128 * 1024 && gpp * 2 <= (1 << 20)) gpp *= 2;
            const int64_t P = (D + gpp - 1) / gpp;
            int shift = 0; while ((1 << shift) < gpp) shift++;
            const bool part = P >= 2 && P <= 4096 && envInt("RSQ_PARTITION", 1, 0, 2) != 0;
            if (part) {
                pipe.partitioned = true; pipe.partCount = (int)P; pipe.partGroups = gpp;
                line("#if RSQ_AGG_VARIANT == 1");
                line("atomicAdd(&st.part[gid >> " + std::to_string(shift) + "], 1u);");
                line("#elif RSQ_AGG_VARIANT == 2");
                openScope("{");
                // records are arrays of R words, stored whole (array of structures): a workgroup then streams into ONE
                // address range per partition, and with one 1024-thread workgroup per CU the partially written lines of
                // all its partitions stay in the XCD's L2 until they are full.  (Struct of arrays with 8 workgroups per
                // CU measured 4.4 ms for 100 M records — every 8-byte store left L2 as its own partial write.)
                for (int w = 1; w < W; w++)
                    if (q.accums[(size_t)w].input != "((i64)1)") pipe.partRecordInputs.push_back(w);   // COUNT's input is the constant 1
                const std::string R = std::to_string(1 + pipe.partRecordInputs.size());
                line("const u32 pos = atomicAdd(&st.part[gid >> " + std::to_string(shift) + "], 1u);");
                line("u64* rec = a.rec + (u64)pos * " + R + ";");
                line("rec[0] = ((u64)(gid & " + std::to_string(gpp - 1) + ") << 40) | (u64)(row - a.row0);");
                addArg("rec", "u64*", 0);
                for (size_t j = 0; j < pipe.partRecordInputs.size(); j++)
                    line("rec[" + std::to_string(j + 1) + "] = (u64)in" + std::to_string(pipe.partRecordInputs[j]) + ";");
                closeScope();
                emitStagedScatter(D, W, gpp, shift, (int)P);
                line("#else");
            }
            pipe.partAtomicsPerRow = 0;
            for (int w = 0; w < W; w++) {
                if (q.accums[(size_t)w].merge == 0) pipe.partAtomicsPerRow++;
                line("rsq::global_merge<" + std::to_string(q.accums[(size_t)w].merge) + ">(a.out + " + std::to_string((long long)(q.accumSlot[(size_t)w] * D)) +
                     " + gid, (u64)" + inOf(w) + ");");
            }
            if (part) {
                line("#endif");
                addArg("part_counts", "u32*", 0); addArg("part_start", "const u32*", 0); addArg("tile_step", "i64", 1);
                stateDecl += "    u32* part;\n";
                const std::string Ps = std::to_string((long long)P);
                prologue += "#if RSQ_AGG_VARIANT == 1 || RSQ_AGG_VARIANT == 2\n    __shared__ u32 s_part[" + Ps + "];\n";
                prologue += "    for (int i = threadIdx.x; i < " + Ps + "; i += blockDim.x)\n";
                prologue += "        s_part[i] = RSQ_AGG_VARIANT == 2 ? a.part_start[i] + a.part_counts[(u64)blockIdx.x * " + Ps + " + i] : 0u;\n";
                prologue += "    __syncthreads();\n    st.part = s_part;\n#endif\n";
                ep << "#if RSQ_AGG_VARIANT == 1\n    __syncthreads();\n";
                ep << "    for (int i = threadIdx.x; i < " << P << "; i += blockDim.x) a.part_counts[(u64)blockIdx.x * " << P << " + i] = s_part[i];\n#endif\n";
                // ---- the per-partition aggregation kernel ----
                std::ostringstream k;
                auto& A2 = pipe.argsPartAgg;
                A2.push_back({"rec", "const u64*", 0});
                A2.push_back({"part_start", "const u32*", 0});
                A2.push_back({"out", "u64*", 0});
                A2.push_back({"row0", "i64", (uint64_t)pipe.src->row0});
                k << "// generated by resql_amd/csrc/codegen.cpp: aggregation of one partition of records in an LDS table\n";
                k << "#include \"rsq_device.h\"\nstruct Args {\n";
                for (auto& a : A2) k << "    " << a.ctype << " " << a.name << ";\n";
                k << "};\nextern \"C\" __global__ void __launch_bounds__(1024) rsq_part_agg(Args a) {\n";
                k << "    __shared__ u64 s_tab[" << (int64_t)W * gpp << "];\n";
                k << "    for (int i = threadIdx.x; i < " << (int64_t)W * gpp << "; i += blockDim.x) { const int blk = i >> " << shift << "; s_tab[i] = " << blockIdentityExpr("blk") << "; }\n";
                k << "    __syncthreads();\n";
                k << "    const u32 b = a.part_start[blockIdx.x], e = a.part_start[blockIdx.x + 1];\n";
                k << "    for (u32 i = b + threadIdx.x; i < e; i += blockDim.x) {\n";
                const size_t RW = 1 + pipe.partRecordInputs.size();
                k << "        const u64* rec = a.rec + (u64)i * " << RW << ";\n";
                k << "        const u64 key = rec[0];\n        const int g = (int)(key >> 40);\n";
                k << "        const i64 row = a.row0 + (i64)(key & ((1ull << 40) - 1));\n";
                for (int w = 0; w < W; w++) {
                    std::string in = "row";
                    if (w > 0) {
                        in = "(i64)1";
                        for (size_t j = 0; j < pipe.partRecordInputs.size(); j++)
                            if (pipe.partRecordInputs[j] == w) in = "rec[" + std::to_string(j + 1) + "]";
                    }
                    k << "        rsq::lds_merge<" << q.accums[(size_t)w].merge << ">(&s_tab[" << (int64_t)q.accumSlot[(size_t)w] * gpp << " + g], (u64)(" << in << "));\n";
                }
                k << "    }\n    __syncthreads();\n";
                k << "    for (int i = threadIdx.x; i < " << (int64_t)W * gpp << "; i += blockDim.x) {\n";
                k << "        const i64 g = (i64)blockIdx.x * " << gpp << " + (i & " << (gpp - 1) << ");\n";
                k << "        if (g < " << D << ") a.out[(i64)(i >> " << shift << ") * " << D << " + g] = s_tab[i];\n    }\n}\n";
                pipe.sourcePartAgg = k.str();
            }
        }
        epilogue += ep.str();
        static const char* names[] = {"none", "registers", "lane-private LDS", "workgroup LDS table", "HBM table", "join entry", "hash"};
        explainSteps.push_back("aggregation dense groups=" + std::to_string((long long)D) + " accumulators=" + std::to_string(W - 1) +
                               " (of " + std::to_string(o->splitAgg.size()) + " in the reference) in " + names[(int)q.aggMode] +
                               (pipe.partitioned ? " (atomics, or " + std::to_string(pipe.partCount) + " partitions x " + std::to_string(pipe.partGroups) +
                                                   " groups aggregated in LDS when many rows pass)" : ""));
    }

    void emitJoinEntryAggregation(OpNode* o) {
        HashTable& ht = *q.hashTables[(size_t)q.aggTable];
        const std::string T = "ht" + std::to_string(ht.id);
        const int W = (int)q.accums.size();
        ht.nAccBlocks = W;
        addArg(T + "_acc", "u64*", 0);
        // Entries of a rank dictionary are in key order, and rows clustered by the key update neighbouring entries: their
        // atomics would queue on a handful of cache lines.  The accumulators of entry r therefore live at rsq::rank_mix(r), a
        // bijection of [0, capacity) (rsq_device.h; the capacity of a dictionary that carries aggregates is a power of two).
        std::string accIdx = slotVar[ht.id];
        if (ht.rankCapable && 1) {
            line("const u64 " + T + "_ai = a." + T + "_rank ? rsq::rank_mix(" + slotVar[ht.id] + ", a." + T + "_cap) : " + slotVar[ht.id] + ";");
            accIdx = T + "_ai";
        }
        const int dbgAcc = 0;      // (measurement only: 1 no first-row tracker, 2 no aggregates, 3 neither)
        for (int w = 0; w < W; w++) {
            if ((w == 0 && (dbgAcc & 1)) || (w > 0 && (dbgAcc & 2))) continue;
            std::string in = w == 0 ? "row" : q.accums[(size_t)w].input;
            line("rsq::global_merge_always<" + std::to_string(q.accums[(size_t)w].merge) + ">(a." + T + "_acc + " + std::to_string(q.accumSlot[(size_t)w]) +
                 " * a." + T + "_cap + " + accIdx + ", (u64)(" + in + "));");
        }
        explainSteps.push_back("aggregation at the matched entry of " + T + " accumulators=" + std::to_string(W - 1) + " (of " +
                               std::to_string(o->splitAgg.size()) + " in the reference)");
    }

    // ---- materialisation of a pipeline without aggregation (materialize.h:78-220) ----------------
    // The reference appends tuples in scan order.  On the device the same order is kept with two passes of the
    // same pipeline: pass 1 counts the tuples every lane emits per 128-row tile, an exclusive scan turns the counts
    // into output offsets, pass 2 writes each tuple to its final position (struct of arrays; the host packs
    // ReSQL tuples from them).
    void consumeMaterialize(OpNode* o, OpNode* from) {
        if (q.agg) failUnsupported("materialize inside an aggregation input");
        if (q.matOp) failUnsupported("more than one materialisation on the device");
        o->schema = from->schema;
        q.matOp = o;
        q.matSchema = o->schema;
        openScope("{");
        line("#if RSQ_PASS == 1");
        line("st.cnt++;");
        line("#else");
        line("const u64 pos = st.pos++;");
        openScope("if (pos < a.out_limit) {");
        int k = 0;
        for (auto& a : o->schema) {
            auto it = eg.symbols.find(a.name);
            if (it == eg.symbols.end()) failType("materialize: symbol " + a.name + " not found");
            const Type& t = it->second.type;
            std::string on = "o" + std::to_string(k++);
            if (t.isString()) {
                addArg(on, "char*", 0);
                line("for (int i = 0; i < " + std::to_string(t.len) + "; i++) a." + on + "[pos * " + std::to_string(t.len) + " + i] = rsq::str_at(" + it->second.var + ", i);");
            } else {
                addArg(on, ExprGen::ctype(t) + "*", 0);
                line("a." + on + "[pos] = " + it->second.var + ";");
            }
        }
        closeScope();
        line("#endif");
        closeScope();
        // cnt[tile * 64 + lane]: tuples the lane's two rows emit; tcnt[tile]: their sum, by the wave; toffs = exclusive scan of tcnt.  The
        // write pass finds a lane's first position as toffs[tile] + the wave's exclusive prefix over cnt: the scan runs over one count
        // per 128 ROWS, not per lane (TPC-H Q19 at SF10: 30 M lane counts, 0.27 ms of scan kernels behind a 1.13 ms count pass).
        addArg("cnt", "u32*", 0); addArg("tcnt", "u32*", 0); addArg("toffs", "const u64*", 0); addArg("out_limit", "u64", 0);
        stateDecl += "    u32 cnt = 0;\n    u64 pos = 0;\n";
        pipe.sink = SinkKind::MATERIALIZE;
        explainSteps.push_back("materialize " + std::to_string(o->schema.size()) + " column(s) in scan order (count / scan / write)");
    }

    // -------------------------------------------------------------------------------------------
    // staged string tiles (see strStaged): chunk c = 64 * round + lane of the tile's 8 * W 16-byte chunks
    std::string postTileFor(const std::string& tile) {
        std::string out = postTile;
        for (size_t at; (at = out.find("$TILE")) != std::string::npos;) out.replace(at, 5, tile);
        return out;
    }
    int stagedRounds(int col) { return (8 * strPrefetchWidth[col] + 63) / 64; }
    void stagedChunkDecls(std::ostringstream& s, const std::string& ind, const char* pre, int col, int u) {
        for (int r = 0; r < stagedRounds(col); r++) s << ind << "rsq::u32v4 " << pre << col << "_" << u << "_" << r << " = {0u, 0u, 0u, 0u};\n";
    }
    void stagedChunkLoads(std::ostringstream& s, const std::string& ind, const char* pre, int col, int u) {     // (`b` = the lane's first row of the tile)
        const int W = strPrefetchWidth[col];
        for (int r = 0; r < stagedRounds(col); r++)
            s << ind << pre << col << "_" << u << "_" << r << " = rsq::ld_str_chunk<" << W << ", " << r << ">(a.c" << col << " + (b - lane * 2) * " << W << ", lane);\n";
    }
    // ... through the wave's LDS region into the words the row function takes (same-wave LDS operations execute in order)
    void stagedUnstage(std::ostringstream& s, const std::string& ind, int u) {
        for (auto& sp : strPrefetch) {
            auto it = strStaged.find(sp.first);
            if (it == strStaged.end()) continue;
            const int W = strPrefetchWidth[sp.first];
            for (int r = 0; r < stagedRounds(sp.first); r++)
                s << ind << "rsq::st_str_chunk<" << W << ", " << r << ">(strt + " << it->second << ", lane, q" << sp.first << "_" << u << "_" << r << ");\n";
        }
        if (!strStaged.empty()) s << ind << "rsq::wave_lds_order();\n";
        for (auto& sp : strPrefetch) {
            auto it = strStaged.find(sp.first);
            if (it == strStaged.end()) continue;
            const int W = strPrefetchWidth[sp.first];
            for (int j = 0; j < 2; j++)
                for (int w = 0; w * 8 < sp.second; w++)
                    s << ind << "const u64 s" << sp.first << "_" << u << "_" << j << "_" << w << " = rsq::ld_bytes<" << std::min(8, sp.second - w * 8) << ">(strt + " << it->second
                      << " + (lane * 2 + " << j << ") * " << W << " + " << w * 8 << ");\n";
        }
        if (!strStaged.empty()) s << ind << "rsq::wave_lds_order();\n";
    }

    // -------------------------------------------------------------------------------------------
    // second round of loads of tile `tile` (unrolled copy u): the late columns, by the lanes that hold a row the leading selection passes
    void emitLateLoads(std::ostringstream& s, const std::string& tile, int u, const std::vector<char>& lateCol, const std::string& tileEnd = "ntiles") {
        s << "        if (" << tile << " < " << tileEnd << ") {\n";
        for (int j = 0; j < 2; j++) {
            s << "            const bool lp" << j << " = lead_pred(a";
            for (int k : leadCols) s << ", t" << k << "_" << u << "[" << j << "]";
            s << ");\n";
        }
        s << "            if (lp0 | lp1) {\n                const i64 b = ((" << tile << ") << 7) + lane * 2;\n";
        for (size_t k = 0; k < lateCol.size(); k++) if (lateCol[k]) s << "                rsq::ld2(a.c" << k << " + b, t" << k << "_" << u << ");\n";
        s << "            }\n        }\n";
    }

    void finishPipeline() {
        while (indent > 1) closeScope();
        addArg("n_rows", "i64", (uint64_t)pipe.src->nRows);
        addArg("row0", "i64", (uint64_t)pipe.src->row0);
        addArg("err", "u32*", (uint64_t)(uintptr_t)q.ctx.dErr);
        // Bytes in flight: a CU streams fastest with ~40 KB of loads outstanding (8 waves x one 128-row tile of TPC-H Q1's
        // 38 B rows).  Narrower rows keep the same amount in flight with more tiles per wave: Q6 (28 B/row) went
        // 0.293 -> 0.254 ms with two tiles, the 32 B synthetic rows gained ~1.5 %; Q1 itself is slower with two (0.363 vs 0.348).
        if (pipe.gridPerCU == 2 && pipe.bytesPerRow > 0)
            pipe.unroll = (int)std::max<int64_t>(1, std::min<int64_t>(4, (4608 + pipe.bytesPerRow * 128 - 1) / (pipe.bytesPerRow * 128)));
        // A pipeline behind a wave compaction waits twice per tile — for the key columns, then (join probes) for the bitmap words their
        // values address — and few of its rows go further: it wants several tiles in flight per wave.  TPC-H Q3 at SF10, all kernels:
        // 0.415 ms with one tile, 0.381 with two, 0.367 with three, 0.363 with four (RSQ_COMPACT_UNROLL).
        if (pipe.compact) pipe.unroll = 4;
        const bool mat = pipe.sink == SinkKind::MATERIALIZE;
        std::ostringstream s;
        s << "// generated by resql_amd/csrc/codegen.cpp\n//   ";
        // (the header comment names the steps WITHOUT the row counts: the source text is the code-object cache key, and a plan
        // shape must find its kernel whatever the table sizes — the cache filled at build time from SF 0.01 tables serves SF 10)
        for (size_t i = 0; i < explainSteps.size(); i++) {
            std::string step = explainSteps[i];
            const size_t lb = step.find(" [");
            if (step.compare(0, 5, "scan ") == 0 && lb != std::string::npos) step = step.substr(0, lb);
            s << (i ? " -> " : "") << step;
        }
        s << "\n";
        if (1) s << "#define RSQ_NT_LOADS 1\n";
        s << "#include \"rsq_device.h\"\n";
        s << fileScope;
        const bool cq = pipe.compact;
        // (measurement only, RSQ_DEBUG_TAIL=1: device timestamps per workgroup - [0] start, [1] rows done, [2] drains done, [3] end)
        const bool dbgStamps = envInt("RSQ_DEBUG_TAIL", 0, 0, 1) != 0 && !(pipe.sink == SinkKind::AGGREGATE && q.aggMode == AggMode::DENSE_REG);
        if (dbgStamps) addArg("dbg", "u64*", 0);
        // (the staged form's round loop knows neither the compaction queues nor prefetched bitmap words nor string columns)
        if (pipe.staged && (cq || mat || !bitmapPrefetch.empty() || !pipe.lazyCols.empty() ||
                            std::find(colIsString.begin(), colIsString.end(), true) != colIsString.end())) pipe.staged = false;
        // Late loads: the columns the leading selection does not read are loaded only by lanes that hold a passing row.  Memory
        // is fetched in 128-byte lines (16 rows of an 8-byte column): at 1 % selectivity 85 % of those columns' lines are never
        // fetched (measured: 1.25 B rows x 4 int64 at 1 %: 6.0 -> 3.5 ms, TPC-H Q6 SF10 0.259 -> 0.207 ms), at 10 % 19 % (-1..4 %);
        // above that the second, dependent round of loads costs more than it saves.  Decided here from the column statistics,
        // values taken as uniform (RSQ_LATE_LOADS: 0 never, 2 whenever there is a leading selection; RSQ_LATE_LOADS_BELOW percent).
        std::vector<char> lateCol(colTypes.size(), 0);
        bool late = false;
        {
            const int mode = envInt("RSQ_LATE_LOADS", 1, 0, 2);
            const double below = (double)12 / 100.0;
            if (mode && !cq && !mat && !leadCond.empty() && bitmapPrefetch.empty() && pipe.lazyCols.empty() && (mode == 2 || leadPass <= below)) {
                for (size_t k = 0; k < colTypes.size(); k++)
                    if (!colIsString[k] && std::find(leadCols.begin(), leadCols.end(), (int)k) == leadCols.end()) { lateCol[k] = 1; late = true; }
            }
        }
        pipe.lateLoads = late;
        // The late-load form keeps the LEADING columns in flight, so its tiles per wave follow their width, not the row's (1.25 B
        // synthetic rows: 8 of 32 bytes lead -> four tiles).  It is software-pipelined (the main loop below): not for the few
        // shapes whose tile loads carry more than plain columns.
        // Measured (MI355X, 1.25 B synthetic rows, G = 8; kernel ms, plain order with two tiles -> pipelined with 2 / 3 / 4 tiles):
        // 1 %: 3.51 -> 2.48 / 2.40 / 2.24; 10 %: 5.86 -> 5.17 / 5.21 / 5.35; TPC-H Q6 SF10 (2 %, 20 leading bytes): 0.206 -> 0.192 / 0.191
        // / 0.197.  The more rows pass, the more of the late registers are really in use and the fewer tiles pay.
        const bool latePipelined = late;
        // ... and so is the loop of a pipeline behind a wave compaction, where its tiles are narrow (below).  Measured at SF10, whole
        // statements: TPC-H Q5 0.707 -> 0.687 ms, Q14 0.289 -> 0.276, Q3 0.289 -> 0.285, the others within noise: these pipelines are
        // bound by the dependent accesses of stage 2 (Q5's lineitem pipeline as two kernels: scan 81 us, stage 2 363 us), not by the stream.
        const bool cqPipelined = pipe.compact && !late && pipe.sink != SinkKind::MATERIALIZE;
        if (late && pipe.gridPerCU == 2) {
            int64_t leadBytes = 0;
            for (int k : leadCols) leadBytes += colTypes[(size_t)k] == "i64" ? 8 : colTypes[(size_t)k] == "i32" ? 4 : 1;
            if (leadBytes > 0) pipe.unroll = (int)std::max<int64_t>(1, std::min<int64_t>(leadPass <= 0.04 ? 4 : 2, (4608 + leadBytes * 128 - 1) / (leadBytes * 128)));
        }
        pipe.leadPass = leadCond.empty() || !leadPassComplete ? -1.0 : leadPass;      // (for the engine's first layout of staged regions: only a complete estimate)
        const int U = pipe.unroll;
        if (strStagedBytes > 0) {
            pipe.extraLdsBytes += (pipe.blockThreads / 64) * strStagedBytes;
            prologue += "    __shared__ __attribute__((aligned(16))) char s_strt[(RSQ_BLOCK_THREADS / 64) * " + std::to_string(strStagedBytes) + "];\n    char* const strt = s_strt + (threadIdx.x >> 6) * " +
                        std::to_string(strStagedBytes) + ";\n";
        }
        // 63 left over + 128 pushed by one tile, rounded up.  (RSQ_QCAP=128 drains after every row_fn call instead: smaller
        // queues, 7 instead of 4 workgroups of a five-word pipeline per CU — measured slower: Q3's orders pipeline 0.24 ->
        // 0.31 ms, its inserts do not want more waves.)
        const int NV = 1 + pipe.compactWords;       // the row index + the carried values
        // (... unless the queues are what limits the workgroups per CU: with six or more words per row three workgroups fit next to each
        // other at 192 entries; at 128 four or five do, and TPC-H Q5's lineitem pipeline - five probes per surviving row - went from
        // 0.55 to 0.46 ms.  Pipelines whose registers set the limit - Q10, Q3 - lose 2-3 % to the extra drains and keep 192.)
        const bool queuesLimit = (144 * 1024) / std::max(1, (pipe.blockThreads / 64) * NV * 192 * 8 + pipe.extraLdsBytes) < 4;
        const int QCAP = (queuesLimit ? 128 : 192);
        const int NVL = 1 + pipe.compactWordsLazy;  // ... in the RSQ_LAZY 1 form
        const bool twoForms = !pipe.lazyCols.empty();
        if (cq) {
            // the queues take LDS: as many workgroups per CU as fit next to each other, at most the 8 of a random-access pipeline
            const int ldsPerWG = (pipe.blockThreads / 64) * NV * QCAP * 8 + pipe.extraLdsBytes;
            const int ldsPerWGLazy = (pipe.blockThreads / 64) * NVL * QCAP * 8 + pipe.extraLdsBytes;
            // (at most 6: with four tiles in flight these kernels hold 70-80 VGPRs, and a seventh workgroup per CU is not resident
            // whatever the occupancy query says - TPC-H Q3's lineitem pipeline started 256 of 1792 workgroups 67 us late; the
            // engine also clamps every grid to the query's answer, engine.cpp residentWorkgroupsPerCU)
            const int wgCap = pipe.unroll >= 3 ? 6 : 8;
            pipe.gridPerCU = std::max(2, std::min(wgCap, (144 * 1024) / std::max(1, ldsPerWG)));
            pipe.gridPerCULazy = std::max(2, std::min(wgCap, (144 * 1024) / std::max(1, ldsPerWGLazy)));
            stateDecl += "    int cq_n = 0;\n    u32 cq_rows = 0;\n    i64* cq;\n";
            // rows that reached stage 2, for the host's choice between the two forms of the kernel (see compactThen).  Only
            // the first 64 workgroups report (tiles are dealt round-robin, so they are a fair sample; the host scales): every
            // wave of the grid adding to one word cost 90 us — atomics on one address serialise
            addArg("cq_total", "unsigned long long*", 0);
            epilogue += "    if (blockIdx.x < 64) {\n        const u64 v = rsq::wave_sum((u64)st.cq_rows);\n        if ((threadIdx.x & 63) == 0 && v) atomicAdd(a.cq_total, (unsigned long long)v);\n    }\n";
            const std::string nv = twoForms ? "RSQ_CQ_NV" : std::to_string(NV);
            prologue += "    __shared__ i64 s_cq[(RSQ_BLOCK_THREADS / 64) * " + nv + " * " + std::to_string(QCAP) + "];\n";
            prologue += "    st.cq = s_cq + (threadIdx.x >> 6) * " + nv + " * " + std::to_string(QCAP) + ";\n";
        }
        if (twoForms) s << "#if RSQ_LAZY\n#define RSQ_CQ_NV " << NVL << "\n#else\n#define RSQ_CQ_NV " << NV << "\n#endif\n";
        s << "#ifndef RSQ_BLOCK_THREADS\n#define RSQ_BLOCK_THREADS " << pipe.blockThreads << "\n#endif\n";
        s << "struct Args {\n";
        for (auto& a : pipe.args) s << "    " << a.ctype << " " << a.name << ";\n";
        s << "};\nstruct State {\n" << stateDecl << "};\n";
        s << helperFns;
        if (cq) {
            // stage 2: everything behind the compaction point, called with dense lanes
            // (Stage 2 as a real function CALLED from the drains instead of inlined at every drain site compiles three times faster - TPC-H
            // Q5's lineitem kernel: 72 KB of code and 1.9 s of hiprtc against 24 KB and 0.66 s - and runs 40-65 % slower: Q5 0.69 -> 1.14 ms,
            // Q10 0.92 -> 1.29, Q3 0.29 -> 0.40 at SF10 (the State lives in scratch memory across the call).  Inlined.)
            s << "static RSQ_DEV void stage2(const Args& a, State& st, const i64 row";
            for (int k = 0; k < pipe.compactWords; k++) s << ", const i64 qw_" << k;
            s << ") {\n" << stage2Body << "}\n";
            s << "static RSQ_DEV void cq_drain(const Args& a, State& st, const int count) {\n";
            s << "    const int lane = threadIdx.x & 63;\n    const int i = st.cq_n - count + lane;\n    rsq::wave_lds_order();      // (the entries were pushed by other lanes)\n";
            s << "    if (lane < count) {\n        stage2(a, st, st.cq[i]";
            for (int k = 0; k < pipe.compactWordsLazy; k++) s << ", st.cq[" << (k + 1) * QCAP << " + i]";
            if (pipe.compactWords > pipe.compactWordsLazy) {
                s << "\n#if RSQ_LAZY\n            ";
                for (int k = pipe.compactWordsLazy; k < pipe.compactWords; k++) s << ", 0";
                s << "\n#else\n            ";
                for (int k = pipe.compactWordsLazy; k < pipe.compactWords; k++) s << ", st.cq[" << (k + 1) * QCAP << " + i]";
                s << "\n#endif\n        ";
            }
            s << ");\n        st.cq_rows++;\n    }\n    st.cq_n -= count;\n}\n";
        }
        if (late) {
            s << "static RSQ_DEV bool lead_pred(const Args& a";
            for (int k : leadCols) s << ", " << colTypes[(size_t)k] << " v_" << k;
            s << ") { return " << leadCond << "; }\n";
        }
        if (pipe.staged) s << "#if RSQ_AGG_VARIANT == 3\ntemplate <int SP_SLOT>       // the row's place among the rows a thread handles per round\n#endif\n";
        s << "static RSQ_DEV void row_fn(const Args& a, State& st, const i64 lr" << (cq ? ", const bool valid" : "") << rowParams << ") {\n";
        s << "    const i64 row = a.row0 + lr;\n";
        if (cq) {
            s << "    bool cq_pass = false;\n";
            for (int k = 0; k < pipe.compactWords; k++) s << "    i64 cq_" << k << " = 0;\n";
            s << "    if (valid) {\n" << body << "    }\n";
            // push: every lane of the wave is here (the callers keep the control flow wave-uniform)
            s << "    {\n        const int lane = threadIdx.x & 63;\n        const u64 m = __ballot(cq_pass);\n";
            s << "        if (cq_pass) {\n            const int s = st.cq_n + (int)__popcll(m & ((1ull << lane) - 1ull));\n            st.cq[s] = row;\n";
            for (int k = 0; k < pipe.compactWords; k++) {
                if (k == pipe.compactWordsLazy && pipe.compactWords > pipe.compactWordsLazy) s << "#if !RSQ_LAZY\n";
                s << "            st.cq[" << (k + 1) * QCAP << " + s] = cq_" << k << ";\n";
            }
            if (pipe.compactWords > pipe.compactWordsLazy) s << "#endif\n";
            s << "        }\n        st.cq_n += (int)__popcll(m);\n    }\n}\n";
        } else s << body << "}\n";
        // one kernel name per pipeline — rsq_p<index>_<scanned table>_<sink> — so that a kernel trace (rocprofv3
        // --kernel-trace --stats) splits a multi-pipeline query into its phases
        {
            std::string tn;
            for (char c : pipe.src->name) tn += (isalnum((unsigned char)c) ? c : '_');
            const char* sk = pipe.sink == SinkKind::BUILD ? "build" : pipe.sink == SinkKind::MATERIALIZE ? "materialize" : "aggregate";
            pipe.entry = "rsq_p" + std::to_string(q.pipelines.size()) + "_" + tn + "_" + sk;
            if (pipe.sink == SinkKind::BUILD) pipe.entry += "_ht" + std::to_string(pipe.buildTable);
        }
        s << "#ifdef RSQ_MIN_WG\nextern \"C\" __global__ void __launch_bounds__(RSQ_BLOCK_THREADS, RSQ_MIN_WG) " << pipe.entry << "(Args a) {\n#else\n";
        s << "extern \"C\" __global__ void __launch_bounds__(RSQ_BLOCK_THREADS) " << pipe.entry << "(Args a) {\n#endif\n";
        s << "    State st;\n" << prologue;
        if (dbgStamps) s << "    if (a.dbg && threadIdx.x == 0) a.dbg[(u64)blockIdx.x * 8 + 0] = (u64)wall_clock64();\n";
        s << "    const int lane = threadIdx.x & 63;\n";
        s << "    const i64 wave = (i64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);\n";
        s << "    const i64 nwaves = (i64)gridDim.x * (blockDim.x >> 6);\n";
        s << "    const i64 ntiles = a.n_rows >> 7;\n";
        if (pipe.partitioned) s << "#if RSQ_AGG_VARIANT == 1\n    const i64 tstep = a.tile_step;      // > 1: sample every n-th tile\n#else\n    const i64 tstep = 1;\n#endif\n";
        else s << "    const i64 tstep = 1;\n";
        const int ncols = (int)colTypes.size();
        if (pipe.staged) {
            // Form 3 walks the table in ROUNDS of blockDim.x * RPT rows with the workgroup in step (every thread reaches the
            // barriers of stage_commit): wave w of the workgroup takes RPT/2 consecutive 128-row tiles of the round.
            const int H = pipe.stagedRows / 2;
            s << "#if RSQ_AGG_VARIANT == 3\n";
            s << "    const i64 tpr = (i64)(blockDim.x >> 6) * " << H << ";        // tiles per round\n";
            s << "    const i64 nrounds = (ntiles + tpr - 1) / tpr;\n";
            s << "    for (i64 round = blockIdx.x; round < nrounds; round += gridDim.x) {\n";
            s << "        st.sp_pending = 0u; st.sp_wm = s_stage.wm;\n";
            s << "        const i64 t0 = round * tpr + (i64)(threadIdx.x >> 6) * " << H << ";\n";
            for (int u = 0; u < H; u++) {
                for (int k = 0; k < ncols; k++) if (!colIsString[(size_t)k]) s << "        " << colTypes[(size_t)k] << " t" << k << "_" << u << "[2]" << (lateCol[(size_t)k] ? " = {0, 0}" : "") << ";\n";
                s << "        if (t0 + " << u << " < ntiles) {\n            const i64 b = ((t0 + " << u << ") << 7) + lane * 2;\n";
                for (int k = 0; k < ncols; k++) if (!colIsString[(size_t)k] && !lateCol[(size_t)k]) s << "            rsq::ld2(a.c" << k << " + b, t" << k << "_" << u << ");\n";
                s << "        }\n";
            }
            if (late) for (int u = 0; u < H; u++) emitLateLoads(s, "t0 + " + std::to_string(u), u, lateCol);
            for (int u = 0; u < H; u++) {
                s << "        if (t0 + " << u << " < ntiles) {\n";
                for (int j = 0; j < 2; j++) {
                    s << "            row_fn<" << (2 * u + j) << ">(a, st, ((t0 + " << u << ") << 7) + lane * 2 + " << j;
                    for (int k = 0; k < ncols; k++) if (!colIsString[(size_t)k]) s << ", t" << k << "_" << u << "[" << j << "]";
                    s << ");\n";
                }
                s << "        }\n";
            }
            s << "        if (!a.sp_mode) rsq::stage_commit<" << pipe.stagedRecWords << ", " << pipe.partCount << ", " << pipe.stagedRows
              << ">(s_stage, st.sp_rec, st.sp_k, st.sp_p, st.sp_pending, a.sp_rec, a.sp_ctl, " << q.denseGroups << "u);\n";
            s << "    }\n";
            s << "    if (blockIdx.x == 0 && (a.n_rows & 127)) {        // the rows behind the last whole tile: one more round of workgroup 0\n";
            s << "        st.sp_pending = 0u; st.sp_wm = s_stage.wm;\n";
            s << "        const i64 r = (ntiles << 7) + threadIdx.x;\n";
            s << "        if (r < a.n_rows) row_fn<0>(a, st, r" << rowArgsTail << ");\n";
            s << "        if (!a.sp_mode) rsq::stage_commit<" << pipe.stagedRecWords << ", " << pipe.partCount << ", " << pipe.stagedRows
              << ">(s_stage, st.sp_rec, st.sp_k, st.sp_p, st.sp_pending, a.sp_rec, a.sp_ctl, " << q.denseGroups << "u);\n";
            s << "    }\n";
            s << "    rsq::stage_finish(s_stage, a.sp_rec, a.sp_counts, a.sp_ctl, a.sp_mode != 0u);\n";
            s << "#else\n";
        }
        // main loop, textually unrolled: the loads of U tiles are issued before the first row is processed
        // (Tiles handed out DYNAMICALLY - per-pool counters, a wave drawing its next chunk one iteration ahead - were built and measured
        // slower: TPC-H Q1 SF10 370 instead of 338 us, Q3 0.341 instead of 0.302 ms, Q6 0.24 instead of 0.207 ms; the returning atomics
        // cost more than the tail they remove.  Tiles are dealt to the waves round-robin.)
        auto emitPipelinedLoop = [&]() {
            // The software-pipelined main loop.  Per iteration a wave
            //   (1) works on its U tiles whose columns were requested ONE ITERATION AGO: the late-load form decides the leading
            //       selection and requests the other columns for the lanes that hold a passing row; a pipeline behind a wave
            //       compaction requests the key-bitmap words its rows address;
            //   (2) requests the NEXT iteration's tile columns;
            //   (3) runs the row function (and the drains of the compaction queues).
            // Memory operations return in issue order, so (2) must stand behind (1): whoever waits for the dependent loads of (1) waits
            // for everything issued before them, never for what was issued after.  The wave then stalls once per iteration with the
            // next tiles' stream in flight the whole time; the plain order - tile loads, wait, dependent loads, wait, rows - had
            // nothing streaming during the second wait and the drains (1.25 B rows at 1 %: 3.69 ms for 14.5 GB fetched, 0.49 of peak).
            // The loads of (2) are unconditional, their tile index clamped to the last tile: the compiler counts the loads it KNOWS
            // stand behind the dependent ones when it places the wait in front of the row function - a load under a condition
            // would not count and the wait would cover the next tiles too.
            std::vector<int> tileCols;
            if (late) tileCols = leadCols;
            else for (int k = 0; k < ncols; k++) if (!colIsString[(size_t)k]) tileCols.push_back(k);
            auto isLazy = [&](int k) { return std::find(pipe.lazyCols.begin(), pipe.lazyCols.end(), k) != pipe.lazyCols.end(); };
            auto tileLoads = [&](const std::string& ind, const char* pre, int u) {      // pre: "t" / "s" (this iteration's) or "n" / "ns" (the next one's)
                const std::string spre = pre[0] == 'n' ? "ns" : "s";
                for (auto& sp : strPrefetch) {
                    if (strStaged.count(sp.first)) { stagedChunkLoads(s, ind, pre[0] == 'n' ? "nq" : "q", sp.first, u); continue; }
                    for (int j = 0; j < 2; j++)
                        for (int w = 0; w * 8 < sp.second; w++)
                            s << ind << spre << sp.first << "_" << u << "_" << j << "_" << w << " = rsq::ld_bytes<" << std::min(8, sp.second - w * 8) << ">(a.c" << sp.first
                              << " + (b + " << j << ") * " << strPrefetchWidth[sp.first] << " + " << w * 8 << ");\n";
                }
                for (int k : tileCols) {
                    if (isLazy(k)) s << "#if !RSQ_LAZY\n";
                    s << ind << "rsq::ld2(a.c" << k << " + b, " << pre << k << "_" << u << ");\n";
                    if (isLazy(k)) s << "#endif\n";
                }
            };
            s << "    const i64 tend = ntiles;\n";
            for (int u = 0; u < U; u++) {
                for (int k : tileCols) s << "    " << colTypes[(size_t)k] << " t" << k << "_" << u << "[2] = {0, 0};\n";
                for (auto& sp : strPrefetch) {
                    if (strStaged.count(sp.first)) { stagedChunkDecls(s, "    ", "q", sp.first, u); continue; }
                    for (int j = 0; j < 2; j++) for (int w = 0; w * 8 < sp.second; w++) s << "    u64 s" << sp.first << "_" << u << "_" << j << "_" << w << " = 0;\n";
                }
            }
            for (int u = 0; u < U; u++) {
                s << "    {\n        const i64 p = wave * tstep + " << u << " * nwaves * tstep;\n        if (p < tend) {\n            const i64 b = (p << 7) + lane * 2;\n";
                tileLoads("            ", "t", u);
                s << "        }\n    }\n";
            }
            s << "    for (i64 t = wave * tstep; t < ntiles; t += nwaves * tstep * " << U << ") {\n";
            for (int u = 0; u < U; u++) {
                s << "        const i64 tt" << u << " = t + " << u << " * nwaves * tstep;\n";
                for (int k = 0; k < ncols; k++) if (!colIsString[(size_t)k] && lateCol[(size_t)k]) s << "        " << colTypes[(size_t)k] << " t" << k << "_" << u << "[2] = {0, 0};\n";
            }
            if (late) for (int u = 0; u < U; u++) emitLateLoads(s, "tt" + std::to_string(u), u, lateCol, "tend");
            for (int u = 0; u < U; u++)
                for (auto& pf : bitmapPrefetch) {
                    s << "        u32 pf_" << pf.first << "_" << u << "[2] = {0u, 0u};\n";
                    s << "        if (tt" << u << " < tend) {\n";
                    for (int j = 0; j < 2; j++)
                        s << "            pf_" << pf.first << "_" << u << "[" << j << "] = " << (pf.interleaved ? "rsq::bmi_load(a." : "rsq::bm_word(a.") << pf.first << "_bm, a." << pf.first
                          << "_bmmin, a." << pf.first << "_bmbits, (i64)t" << pf.second << "_" << u << "[" << j << "]);\n";
                    s << "        }\n";
                }
            for (int u = 0; u < U; u++) {
                for (int k : tileCols) s << "        " << colTypes[(size_t)k] << " n" << k << "_" << u << "[2] = {0, 0};\n";
                for (auto& sp : strPrefetch) {
                    if (strStaged.count(sp.first)) { stagedChunkDecls(s, "        ", "nq", sp.first, u); continue; }
                    for (int j = 0; j < 2; j++) for (int w = 0; w * 8 < sp.second; w++) s << "        u64 ns" << sp.first << "_" << u << "_" << j << "_" << w << " = 0;\n";
                }
                s << "        {\n            const i64 nt = tt" << u << " + nwaves * tstep * " << U << ";\n            const i64 b = ((nt < tend ? nt : tend - 1) << 7) + lane * 2;\n";
                tileLoads("            ", "n", u);
                s << "        }\n";
            }
            for (int u = 0; u < U; u++) {
                s << "        if (tt" << u << " < tend) {\n";
                stagedUnstage(s, "            ", u);
                for (int j = 0; j < 2; j++) {
                    s << "            row_fn(a, st, (tt" << u << " << 7) + lane * 2 + " << j << (cq ? ", true" : "");
                    for (int k = 0; k < ncols; k++) if (!colIsString[(size_t)k]) s << ", t" << k << "_" << u << "[" << j << "]";
                    for (auto& sp : strPrefetch) for (int w = 0; w * 8 < sp.second; w++) s << ", s" << sp.first << "_" << u << "_" << j << "_" << w;
                    for (auto& pf : bitmapPrefetch) s << ", pf_" << pf.first << "_" << u << "[" << j << "]";
                    s << ");\n";
                    if (cq && (QCAP < 192 || j == 1)) s << "            while (st.cq_n >= 64) cq_drain(a, st, 64);\n";
                }
                s << postTileFor("tt" + std::to_string(u));
                s << "        }\n";
            }
            for (int u = 0; u < U; u++) {
                for (int k : tileCols) s << "        t" << k << "_" << u << "[0] = n" << k << "_" << u << "[0]; t" << k << "_" << u << "[1] = n" << k << "_" << u << "[1];\n";
                for (auto& sp : strPrefetch) {
                    if (strStaged.count(sp.first)) {
                        for (int r = 0; r < stagedRounds(sp.first); r++) s << "        q" << sp.first << "_" << u << "_" << r << " = nq" << sp.first << "_" << u << "_" << r << ";\n";
                        continue;
                    }
                    for (int j = 0; j < 2; j++) for (int w = 0; w * 8 < sp.second; w++)
                        s << "        s" << sp.first << "_" << u << "_" << j << "_" << w << " = ns" << sp.first << "_" << u << "_" << j << "_" << w << ";\n";
                }
            }
            s << "    }\n";
        };
        auto emitPlainLoop = [&]() {
        s << "    const i64 tend = ntiles;\n";
        s << "    for (i64 t = wave * tstep; t < ntiles; t += nwaves * tstep * " << U << ") {\n";
        const bool matSkip = mat && !cq && !late && 1 != 0;
        pipe.matSkip = matSkip;
        auto tileLive = [&](int u) { return matSkip ? "live" + std::to_string(u) : "tt" + std::to_string(u) + " < tend"; };
        for (int u = 0; u < U; u++) {
            s << "        const i64 tt" << u << " = t + " << u << " * nwaves * tstep;\n";
            for (int k = 0; k < ncols; k++) if (!colIsString[(size_t)k]) s << "        " << colTypes[(size_t)k] << " t" << k << "_" << u << "[2]" << (lateCol[(size_t)k] ? " = {0, 0}" : "") << ";\n";
            for (auto& sp : strPrefetch) {
                if (strStaged.count(sp.first)) { stagedChunkDecls(s, "        ", "q", sp.first, u); continue; }
                for (int j = 0; j < 2; j++)
                    for (int w = 0; w * 8 < sp.second; w++) s << "        u64 s" << sp.first << "_" << u << "_" << j << "_" << w << " = 0;\n";
            }
            // The write pass of a materialisation skips every tile whose 64 lane slots counted nothing in the count pass: a selective
            // statement (TPC-H Q19: 1107 rows out of 60 M) then reads its columns once, not twice.
            if (matSkip) s << "#if RSQ_PASS == 2\n        const bool live" << u << " = tt" << u << " < tend && a.tcnt[tt" << u << "] != 0u;\n#else\n"
                           << "        const bool live" << u << " = tt" << u << " < tend;\n#endif\n";
            s << "        if (" << tileLive(u) << ") {\n            const i64 b = (tt" << u << " << 7) + lane * 2;\n";
            for (auto& sp : strPrefetch) {
                if (strStaged.count(sp.first)) { stagedChunkLoads(s, "            ", "q", sp.first, u); continue; }
                for (int j = 0; j < 2; j++)
                    for (int w = 0; w * 8 < sp.second; w++)
                        s << "            s" << sp.first << "_" << u << "_" << j << "_" << w << " = rsq::ld_bytes<" << std::min(8, sp.second - w * 8) << ">(a.c" << sp.first
                          << " + (b + " << j << ") * " << strPrefetchWidth[sp.first] << " + " << w * 8 << ");\n";
            }
            for (int k = 0; k < ncols; k++) if (!colIsString[(size_t)k] && !lateCol[(size_t)k]) {
                const bool lazy = std::find(pipe.lazyCols.begin(), pipe.lazyCols.end(), k) != pipe.lazyCols.end();
                if (lazy) s << "#if !RSQ_LAZY\n";
                s << "            rsq::ld2(a.c" << k << " + b, t" << k << "_" << u << ");\n";
                if (lazy) s << "#else\n            t" << k << "_" << u << "[0] = t" << k << "_" << u << "[1] = 0;\n#endif\n";
            }
            s << "        }\n";
        }
        if (late) for (int u = 0; u < U; u++) emitLateLoads(s, "tt" + std::to_string(u), u, lateCol, "tend");
        for (int u = 0; u < U; u++)
            for (auto& pf : bitmapPrefetch) {
                s << "        u32 pf_" << pf.first << "_" << u << "[2] = {0u, 0u};\n";
                s << "        if (" << tileLive(u) << ") {\n";
                for (int j = 0; j < 2; j++) {
                    s << "            pf_" << pf.first << "_" << u << "[" << j << "] = ";
                    s << (pf.interleaved ? "rsq::bmi_load(a." : "rsq::bm_word(a.") << pf.first << "_bm, a." << pf.first << "_bmmin, a." << pf.first
                      << "_bmbits, (i64)t" << pf.second << "_" << u << "[" << j << "]);\n";
                }
                s << "        }\n";
            }
        for (int u = 0; u < U; u++) {
            s << "        if (" << tileLive(u) << ") {\n";
            stagedUnstage(s, "            ", u);
            if (mat) s << "            const i64 slot = tt" << u << " * 64 + lane;\n#if RSQ_PASS == 2\n            st.pos = a.toffs[tt" << u << "] + (u64)rsq::wave_excl_sum_u32(a.cnt[slot]);\n#endif\n";
            for (int j = 0; j < 2; j++) {
                s << "            row_fn(a, st, (tt" << u << " << 7) + lane * 2 + " << j << (cq ? ", true" : "");
                for (int k = 0; k < ncols; k++) if (!colIsString[(size_t)k]) s << ", t" << k << "_" << u << "[" << j << "]";
                for (auto& sp : strPrefetch) for (int w = 0; w * 8 < sp.second; w++) s << ", s" << sp.first << "_" << u << "_" << j << "_" << w;
                for (auto& pf : bitmapPrefetch) s << ", pf_" << pf.first << "_" << u << "[" << j << "]";
                s << ");\n";
                if (cq && (QCAP < 192 || j == 1)) s << "            while (st.cq_n >= 64) cq_drain(a, st, 64);\n";
            }
            if (mat) s << "#if RSQ_PASS == 1\n            a.cnt[slot] = st.cnt;\n            { const u32 ts = (u32)rsq::wave_sum((u64)st.cnt); if (lane == 0) a.tcnt[tt" << u << "] = ts; }\n            st.cnt = 0;\n#endif\n";
            s << postTileFor("tt" + std::to_string(u));
            s << "        }\n";
        }
        s << "    }\n";
        };
        // Double-buffered tile registers pay where the tiles are narrow: the late-load form (leading columns only), the RSQ_LAZY 1
        // form of a compaction pipeline (the columns stage 1 reads), any compaction pipeline whose tiles take few registers.  A
        // wide eager tile set would cost the occupancy the plan needs (TPC-H Q5 / Q10 lineitem: 133 -> 183 VGPRs).
        {
            int eagerRegs = 0, lazyRegs = 0;      // VGPRs of one tile's columns: all of them / without the lazily loaded ones
            for (int k = 0; k < ncols; k++) if (!colIsString[(size_t)k]) {
                const int r = colTypes[(size_t)k] == "i64" ? 4 : colTypes[(size_t)k] == "i32" ? 2 : 1;
                eagerRegs += r;
                if (std::find(pipe.lazyCols.begin(), pipe.lazyCols.end(), k) == pipe.lazyCols.end()) lazyRegs += r;
            }
            for (auto& sp : strPrefetch) { const int r = strStaged.count(sp.first) ? 4 * stagedRounds(sp.first) : 4 * ((sp.second + 7) / 8); eagerRegs += r; lazyRegs += r; }
            const int budget = 24;      // (64: TPC-H Q5 0.69 -> 0.78 ms, the eager form's 133 -> 183 VGPRs)
            const bool eagerOk = cqPipelined && eagerRegs * U <= budget, lazyOk = cqPipelined && lazyRegs * U <= budget;
            if (latePipelined || (eagerOk && (lazyOk || pipe.lazyCols.empty()))) emitPipelinedLoop();
            else if (lazyOk && !pipe.lazyCols.empty()) { s << "#if RSQ_LAZY\n"; emitPipelinedLoop(); s << "#else\n"; emitPlainLoop(); s << "#endif\n"; }
            else emitPlainLoop();
        }
        if (dbgStamps) s << "    if (a.dbg && threadIdx.x == 0) a.dbg[(u64)blockIdx.x * 8 + 1] = (u64)wall_clock64();\n";
        if (!postTile.empty()) s << "    st.in_tile = false;\n";
        if (cq) {
            // tail rows with a wave-uniform trip count (the push votes across the wave)
            s << "    for (i64 rb = (ntiles << 7) + (i64)blockIdx.x * blockDim.x; rb < a.n_rows; rb += (i64)gridDim.x * blockDim.x) {\n";
            s << "        const i64 r = rb + threadIdx.x;\n        const bool valid = r < a.n_rows;\n";
            s << "        row_fn(a, st, r, valid" << rowArgsTailGuarded << ");\n";
            s << "        while (st.cq_n >= 64) cq_drain(a, st, 64);\n    }\n";
            s << "    while (st.cq_n > 0) cq_drain(a, st, st.cq_n < 64 ? st.cq_n : 64);\n";
        } else {
            if (mat) {
                // the rows behind the last whole tile, 64 to a wave: pseudo-tiles ntiles and ntiles + 1 of the count arrays (wave-uniform trip count)
                s << "    for (i64 rb = (ntiles << 7) + (i64)blockIdx.x * blockDim.x + (i64)(threadIdx.x & ~63u); rb < a.n_rows; rb += (i64)gridDim.x * blockDim.x) {\n";
                s << "        const i64 r = rb + lane;\n        const bool valid = r < a.n_rows;\n";
                s << "        const i64 ttile = ntiles + ((rb - (ntiles << 7)) >> 6);\n        const i64 slot = ttile * 64 + lane;\n";
                s << "#if RSQ_PASS == 2\n        st.pos = a.toffs[ttile] + (u64)rsq::wave_excl_sum_u32(valid ? a.cnt[slot] : 0u);\n#endif\n";
                s << "        if (valid) row_fn(a, st, r" << rowArgsTail << ");\n";
                s << "#if RSQ_PASS == 1\n        a.cnt[slot] = st.cnt;\n        { const u32 ts = (u32)rsq::wave_sum((u64)st.cnt); if (lane == 0) a.tcnt[ttile] = ts; }\n        st.cnt = 0;\n#endif\n";
                s << "    }\n";
            } else {
            s << "    for (i64 r = (ntiles << 7) + (i64)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_rows; r += (i64)gridDim.x * blockDim.x) {\n";
            s << "        row_fn(a, st, r" << rowArgsTail << ");\n";
            s << "    }\n";
            }
        }
        if (pipe.staged) s << "#endif\n";
        if (dbgStamps) s << "    if (a.dbg && threadIdx.x == 0) a.dbg[(u64)blockIdx.x * 8 + 2] = (u64)wall_clock64();\n";
        s << epilogue;
        if (dbgStamps) s << "    __syncthreads();\n    if (a.dbg && threadIdx.x == 0) a.dbg[(u64)blockIdx.x * 8 + 3] = (u64)wall_clock64();\n";
        s << "}\n";
        pipe.source = s.str();
        if (!pipe.lazyCols.empty()) pipe.source = "#define RSQ_LAZY 0\n" + pipe.source;
        if (mat) {   // two code objects from one source
            pipe.sourcePass1 = "#define RSQ_PASS 1\n" + pipe.source;
            pipe.source = "#define RSQ_PASS 2\n" + pipe.source;
        }
        if (q.aggPad > 1 && pipe.sink == SinkKind::AGGREGATE && !pipe.partitioned) {
            // padded flush for full executions, flat flush (straight into the [block][group] table that is merged
            // across GPUs) for partial ones
            pipe.sourceFlat = "#define RSQ_OUT_STRIDE 1\n" + pipe.source;
            pipe.source = "#define RSQ_OUT_STRIDE " + std::to_string(q.aggPad) + "\n" + pipe.source;
        }
        if (pipe.partitioned) {   // three code objects from one source (see emitDenseAggregation)
            // count and scatter run as ONE 1024-thread workgroup per CU (see the note at the record stores)
            pipe.sourcePartCount = "#define RSQ_AGG_VARIANT 1\n#define RSQ_BLOCK_THREADS 1024\n" + pipe.source;
            pipe.sourcePartScatter = "#define RSQ_AGG_VARIANT 2\n#define RSQ_BLOCK_THREADS 1024\n" + pipe.source;
            if (pipe.staged) pipe.sourceStagedScatter = "#define RSQ_AGG_VARIANT 3\n#define RSQ_BLOCK_THREADS 1024\n" + pipe.source;
            pipe.source = "#define RSQ_AGG_VARIANT 0\n" + pipe.source;
        }
        if (!pipe.lazyCols.empty()) {          // the late-load form of the full-execution kernel: same text, RSQ_LAZY 1
            pipe.sourceLazy = pipe.source;
            const std::string off = "#define RSQ_LAZY 0\n";
            const size_t at = pipe.sourceLazy.find(off);
            if (at == std::string::npos) pipe.sourceLazy.clear();
            else {
                // The late-load form runs stage 2 for a few per cent of the rows, but its registers are the kernel's: a hash aggregation
                // with dozens of carried words holds 160 VGPRs and leaves the scan three waves per SIMD.  RSQ_LAZY_MIN_WG = n asks
                // the compiler for n resident workgroups per CU (it spills in stage 2 instead).
                const int minWg = 0;
                pipe.sourceLazy.replace(at, off.size(), "#define RSQ_LAZY 1\n" + (minWg ? "#define RSQ_MIN_WG " + std::to_string(minWg) + "\n" : std::string()));
            }
        }
        std::string ex = "pipeline " + std::to_string(q.pipelines.size()) + ": ";
        for (size_t i = 0; i < explainSteps.size(); i++) ex += (i ? " -> " : "") + explainSteps[i];
        if (pipe.lateLoads) {
            char buf[96];
            snprintf(buf, sizeof buf, " [late loads: ~%.1f %% of the rows expected to pass the leading selection]", leadPass * 100.0);
            ex += buf;
        }
        if (pipe.staged) ex += " [partitioned as packed " + std::to_string(8 * pipe.stagedRecWords) + "-byte records staged through LDS rings]";
        pipe.explain = ex;
        q.pipelines.push_back(pipe);
    }
};

}  // namespace

void buildPipelines(Query& q) {
    Walker w(q);
    w.produce(q.root, {});
    if (!q.agg && !q.matOp) failInvalid("plan has neither an aggregation nor a materialisation");
}

}  // namespace rsq
