// ref_harness — drives the UNMODIFIED reference (Henning1/resql, asmjit back end) on a
// "case file" (tables + plan in the resqlplan text format, see tests/planfmt.py).
//
// TEST INFRASTRUCTURE ONLY.  This file is ours; everything it includes below lives in
// /root/reference and is compiled where it lies (never copied).  It is built by
// oracle/Makefile into oracle/_ref/ (git-ignored, shipped to the GPU box as a binary).
// It plays the role of the reference's own execute.h:213-247 `executeSelectPlan`
// (same call sequence on the reference's own classes) without pulling in the SQL
// parser (flex is not in this image) — plans are built with the reference's operator
// constructors exactly like the reference's test/test_operators.h does.
//
// usage: ref_harness CASEFILE [--threads N] [--blocksize BYTES] [--repeat K] [--out FILE] [--quiet]
//
// stdout (or --out): "#schema name:TYPE|..." then the result serialised by the
// reference's serializeRelation (dbdata.h:688-701), then "#timing compile_ms exec_ms" lines
// on stderr.

#include "operators/JitOperators.h"
#include "expressions.h"
#include "dbdata.h"
#include "schema.h"
#include "JitContextFlounder.h"
// The reference's SQL front end: its Lemon grammar (parser.h / parser.c are generated into oracle/_ref/ by the
// reference's own vendored lemon.c from src/parser/parser.y — see oracle/Makefile) and its planner.  The flex
// tokenizer (src/parser/lexer.y) cannot be generated here (no flex in the image), so `--sql-tokens FILE` feeds the
// reference's Parse() a token stream instead of calling yylex().  parseSql() — the only user of the flex symbols — is
// given internal linkage for this translation unit (`Query static parseSql_unused(...)`): it is never called, so it is
// not emitted and the binary links without any stand-in for the tokenizer.
#define parseSql static parseSql_unused
#include "parser/parseSql.h"
#undef parseSql
#include "planner.h"

#ifdef RSQ_WITH_HIP_BINDING
#include "resql_hip_binding.h"     // integration/: ReSQL operator tree -> include/resql_hip.h
#endif

#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include <map>
#include <cstdio>
#include <cstdlib>

size_t DataBlock::Size = 2 << 20;   // as in src/resql.cpp:23 / test/test.cpp:7

namespace {

struct ColSpec {
    std::string name;
    SqlType type;
    std::string source;   // "bin" | "zero"
    std::string path;
};

struct TableSpec {
    std::string name;
    size_t nrows = 0;
    std::vector<ColSpec> cols;
    std::string tblPath;  // optional: '|' separated text rows for all columns
};

struct ExprSpec {
    std::string tag;
    std::vector<std::string> args;
    std::string rest;     // raw remainder (for CONSTANT text)
};

struct OpSpec {
    std::string tag;
    std::vector<std::string> args;
};

struct CaseSpec {
    std::vector<TableSpec> tables;
    std::map<int, ExprSpec> exprs;
    std::map<int, OpSpec> ops;
    int root = -1;
    bool hasLimit = false;
    size_t limit = 0;
    bool requestAll = false;
    std::string dir;
};

[[noreturn]] void die(const std::string& m) {
    std::cerr << "ref_harness: " << m << std::endl;
    exit(2);
}

SqlType parseType(std::istringstream& in) {
    std::string t; in >> t;
    if (t == "INT") return TypeInit::INT();
    if (t == "BIGINT") return TypeInit::BIGINT();
    if (t == "DATE") return TypeInit::DATE();
    if (t == "BOOL") return TypeInit::BOOL();
    if (t == "DECIMAL") { int p, s; in >> p >> s; return TypeInit::DECIMAL(p, s); }
    if (t == "CHAR") { size_t n; in >> n; return TypeInit::CHAR(n); }
    if (t == "VARCHAR") { size_t n; in >> n; return TypeInit::VARCHAR(n); }
    die("unknown type " + t);
}

SqlType::Tag parseCategory(const std::string& t) {
    if (t == "INT") return SqlType::INT;
    if (t == "BIGINT") return SqlType::BIGINT;
    if (t == "DATE") return SqlType::DATE;
    if (t == "BOOL") return SqlType::BOOL;
    if (t == "DECIMAL") return SqlType::DECIMAL;
    if (t == "CHAR") return SqlType::CHAR;
    if (t == "VARCHAR") return SqlType::VARCHAR;
    die("unknown constant category " + t);
}

CaseSpec parseCase(const std::string& path) {
    CaseSpec c;
    auto slash = path.find_last_of('/');
    c.dir = (slash == std::string::npos) ? "." : path.substr(0, slash);
    std::ifstream f(path);
    if (!f.is_open()) die("cannot open " + path);
    std::string line;
    TableSpec* cur = nullptr;
    while (std::getline(f, line)) {
        std::istringstream in(line);
        std::string kw; in >> kw;
        if (kw.empty() || kw[0] == '#') continue;
        if (kw == "resqlplan") continue;
        if (kw == "table") {
            c.tables.emplace_back();
            cur = &c.tables.back();
            in >> cur->name >> cur->nrows;
        } else if (kw == "col") {
            if (!cur) die("col outside table");
            ColSpec cs; in >> cs.name; cs.type = parseType(in);
            in >> cs.source;
            if (cs.source == "bin") in >> cs.path;
            cur->cols.push_back(cs);
        } else if (kw == "tbl") {
            if (!cur) die("tbl outside table");
            in >> cur->tblPath;
        } else if (kw == "end") {
            cur = nullptr;
        } else if (kw == "expr") {
            int id; ExprSpec e; in >> id >> e.tag;
            if (e.tag == "CONSTANT") {
                std::string cat; in >> cat; e.args.push_back(cat);
                std::getline(in, e.rest);
                if (!e.rest.empty() && e.rest[0] == ' ') e.rest.erase(0, 1);
            } else {
                std::string a; while (in >> a) e.args.push_back(a);
            }
            c.exprs[id] = e;
        } else if (kw == "op") {
            int id; OpSpec o; in >> id >> o.tag;
            std::string a; while (in >> a) o.args.push_back(a);
            c.ops[id] = o;
        } else if (kw == "root") {
            in >> c.root;
            std::string a;
            while (in >> a) {
                if (a == "limit") { in >> c.limit; c.hasLimit = true; }
                else if (a == "requestall") c.requestAll = true;
            }
        } else {
            die("unknown keyword " + kw);
        }
    }
    return c;
}

std::string resolve(const CaseSpec& c, const std::string& p) {
    if (!p.empty() && p[0] == '/') return p;
    return c.dir + "/" + p;
}

std::vector<char> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary | std::ios::ate);
    if (!f.is_open()) die("cannot open " + p);
    std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<char> buf(n);
    if (n > 0 && !f.read(buf.data(), n)) die("read failed " + p);
    return buf;
}

size_t fileWidth(SqlType t) {
    switch (t.tag) {
        case SqlType::INT: case SqlType::DATE: return 4;
        case SqlType::BIGINT: case SqlType::DECIMAL: return 8;
        case SqlType::BOOL: return 1;
        case SqlType::CHAR: return t.charSpec().num;
        case SqlType::VARCHAR: return t.varcharSpec().num;
        default: die("fileWidth: unsupported type");
    }
}

void loadTable(const CaseSpec& c, const TableSpec& ts, Database& db, bool tblOnDevice) {
    std::vector<Attribute> atts;
    for (auto& cs : ts.cols) atts.push_back({cs.name, cs.type});
    Schema schema(atts);
    db.relations.try_emplace(ts.name, schema);
    Relation& rel = db.relations[ts.name];
    auto its = AttributeIterator::getAll(rel._schema);
    Relation::AppendIterator app(&rel);

    if (!ts.tblPath.empty() && tblOnDevice) return;      // --engine hip: BULK INSERT goes straight to device columns
    if (!ts.tblPath.empty()) {
        // text rows, parsed with the reference's own constant parser (as execute.h:332-388 does)
        std::ifstream f(resolve(c, ts.tblPath));
        if (!f.is_open()) die("cannot open " + ts.tblPath);
        std::string line;
        while (std::getline(f, line)) {
            Data* t = app.get();     // an empty line has no fields: "missing attributes", as in the reference
            std::istringstream ls(line);
            std::string tok; size_t a = 0;
            while (std::getline(ls, tok, '|')) {
                if (a >= its.size()) die("extra attributes in " + ts.tblPath);
                SqlType type = its[a].attribute.type;
                Expr* e = ExprGen::constant(tok, type.tag);
                ValueMoves::toAddress(its[a].getPtr(t), e->value, type);
                freeExpr(e);
                a++;
            }
            if (a < its.size()) die("missing attributes in " + ts.tblPath);
        }
        return;
    }

    std::vector<std::vector<char>> data(ts.cols.size());
    for (size_t i = 0; i < ts.cols.size(); i++) {
        if (ts.cols[i].source == "bin") {
            data[i] = slurp(resolve(c, ts.cols[i].path));
            if (data[i].size() != ts.nrows * fileWidth(ts.cols[i].type))
                die("size mismatch for column " + ts.cols[i].name);
        }
    }
    for (size_t r = 0; r < ts.nrows; r++) {
        Data* t = app.get();
        for (size_t i = 0; i < ts.cols.size(); i++) {
            if (data[i].empty()) continue;   // "zero": blocks are value-initialised
            size_t w = fileWidth(ts.cols[i].type);
            const char* src = data[i].data() + r * w;
            Data* dst = its[i].getPtr(t);
            memcpy(dst, src, w);
            SqlType::Tag tg = ts.cols[i].type.tag;
            if (tg == SqlType::CHAR || tg == SqlType::VARCHAR) dst[w] = '\0';
        }
    }
}

struct PlanBuilder {
    const CaseSpec& c;
    Database& db;
    std::map<int, Expr*> exprs;
    std::map<int, RelOperator*> ops;

    Expr* expr(int id) {
        auto it = exprs.find(id);
        if (it != exprs.end()) return it->second;
        auto sp = c.exprs.find(id);
        if (sp == c.exprs.end()) die("undefined expr " + std::to_string(id));
        const ExprSpec& s = sp->second;
        auto A = [&](size_t i) { return expr(std::stoi(s.args.at(i))); };
        Expr* e = nullptr;
        const std::string& t = s.tag;
        if (t == "ATTRIBUTE") e = ExprGen::attr(s.args.at(0));
        else if (t == "CONSTANT") {
            if (s.rest.compare(0, 4, "neg ") == 0) {
                // a negated literal of the SQL grammar (parser.y:149-151): parsed from the unsigned text, then the value is negated
                e = ExprGen::constant(s.rest.substr(4), parseCategory(s.args.at(0)));
                if (e->type.tag == SqlType::BIGINT) e->value.bigintData *= -1;
                else if (e->type.tag == SqlType::DECIMAL) e->value.decimalData *= -1;
                else die("neg constant of unsupported category");
            } else e = ExprGen::constant(s.rest, parseCategory(s.args.at(0)));
        }
        else if (t == "STAR") e = ExprGen::star();
        else if (t == "TYPECAST") {                       // explicit `expr :: type`: "TYPECAST child TYPE..."
            std::string ty;
            for (size_t i = 1; i < s.args.size(); i++) ty += (i > 1 ? " " : "") + s.args[i];
            std::istringstream tin(ty);
            e = ExprGen::typecast(parseType(tin), A(0));
        }
        else if (t == "AS") e = ExprGen::as(s.args.at(0), expr(std::stoi(s.args.at(1))));
        else if (t == "SUM") e = ExprGen::sum(A(0));
        else if (t == "COUNT") e = ExprGen::count(A(0));
        else if (t == "AVG") e = ExprGen::avg(A(0));
        else if (t == "MIN") e = ExprGen::min(A(0));
        else if (t == "MAX") e = ExprGen::max(A(0));
        else if (t == "ASC") e = ExprGen::asc(A(0));
        else if (t == "DESC") e = ExprGen::desc(A(0));
        else if (t == "ADD") e = bin(&ExprGen::add, s);
        else if (t == "SUB") e = bin(&ExprGen::sub, s);
        else if (t == "MUL") e = bin(&ExprGen::mul, s);
        else if (t == "DIV") e = bin(&ExprGen::div, s);
        else if (t == "AND") e = bin(&ExprGen::and_, s);
        else if (t == "OR") e = bin(&ExprGen::or_, s);
        else if (t == "LT") e = bin(&ExprGen::lt, s);
        else if (t == "LE") e = bin(&ExprGen::le, s);
        else if (t == "GT") e = bin(&ExprGen::gt, s);
        else if (t == "GE") e = bin(&ExprGen::ge, s);
        else if (t == "EQ") e = bin(&ExprGen::eq, s);
        else if (t == "NEQ") e = bin(&ExprGen::neq, s);
        else if (t == "LIKE") e = bin(&ExprGen::like, s);
        else if (t == "WHENTHEN") e = bin(&ExprGen::whenThen, s);
        else if (t == "CASE") {
            Expr* first = nullptr; Expr* prev = nullptr;
            for (size_t i = 0; i < s.args.size(); i++) {
                Expr* ch = A(i);
                if (!first) first = ch; else prev->next = ch;
                prev = ch;
            }
            e = ExprGen::case_(first);
        }
        else die("unsupported expr tag " + t);
        exprs[id] = e;
        return e;
    }

    Expr* bin(Expr* (*mk)(Expr*, Expr*), const ExprSpec& s) {
        Expr* l = expr(std::stoi(s.args.at(0)));
        Expr* r = expr(std::stoi(s.args.at(1)));
        if (l->next != nullptr && l->next != r)
            die("expression node shared as left child with different right siblings");
        return mk(l, r);
    }

    ExprVec exprList(const OpSpec& o, size_t& pos) {
        size_t n = std::stoul(o.args.at(pos++));
        ExprVec v;
        for (size_t i = 0; i < n; i++) v.push_back(expr(std::stoi(o.args.at(pos++))));
        return v;
    }

    RelOperator* op(int id) {
        auto it = ops.find(id);
        if (it != ops.end()) return it->second;
        auto sp = c.ops.find(id);
        if (sp == c.ops.end()) die("undefined op " + std::to_string(id));
        const OpSpec& o = sp->second;
        RelOperator* r = nullptr;
        if (o.tag == "SCAN") {
            const std::string& tn = o.args.at(0);
            if (!db.relations.count(tn)) die("no table " + tn);
            r = new ScanOp(&db.relations[tn], tn);
        } else if (o.tag == "SELECTION") {
            r = new SelectionOp(expr(std::stoi(o.args.at(1))), op(std::stoi(o.args.at(0))));
        } else if (o.tag == "PROJECTION") {
            size_t pos = 1; ExprVec v = exprList(o, pos);
            r = new ProjectionOp(v, op(std::stoi(o.args.at(0))));
        } else if (o.tag == "HASHJOIN") {
            size_t pos = 3; ExprVec v = exprList(o, pos);
            HashJoinOp* hj = new HashJoinOp(v, op(std::stoi(o.args.at(0))), op(std::stoi(o.args.at(1))));
            hj->_singleMatch = std::stoi(o.args.at(2)) != 0;
            r = hj;
        } else if (o.tag == "AGGREGATION") {
            size_t pos = 1; ExprVec aggs = exprList(o, pos); ExprVec grps = exprList(o, pos);
            r = new AggregationOp(aggs, grps, op(std::stoi(o.args.at(0))));
        } else if (o.tag == "MATERIALIZE") {
            r = new MaterializeOp(op(std::stoi(o.args.at(0))));
        } else if (o.tag == "ORDERBY") {
            size_t pos = 1; ExprVec v = exprList(o, pos);
            r = new OrderByOp(std::move(v), op(std::stoi(o.args.at(0))));
        } else die("unsupported op " + o.tag);
        ops[id] = r;
        return r;
    }
};

// ---- SQL front end: tokens -> the reference's Lemon parser -> the reference's planner --------------------------
struct TokenName { const char* name; int code; };
#define TKN(x) {#x, x}
const TokenName kTokenNames[] = {
    TKN(OR_TK), TKN(AND_TK), TKN(LT_TK), TKN(GT_TK), TKN(LE_TK), TKN(GE_TK), TKN(EQ_TK), TKN(NEQ_TK), TKN(BETWEEN_TK), TKN(IN_TK),
    TKN(PLUS_TK), TKN(MINUS_TK), TKN(MUL_TK), TKN(DIV_TK), TKN(LIKE_TK), TKN(TYPECAST_TK), TKN(SUM_TK), TKN(CREATE_TABLE_TK),
    TKN(IDENTIFIER), TKN(LPAREN), TKN(RPAREN), TKN(BULK_INSERT_TK), TKN(FROM), TKN(STRING_CONSTANT), TKN(WITH_TK), TKN(COMMA),
    TKN(FIRSTROW_TK), TKN(INTEGER_CONSTANT), TKN(FIELDTERMINATOR_TK), TKN(SELECT_TK), TKN(WHERE), TKN(GROUPBY), TKN(ORDERBY),
    TKN(LIMIT_TK), TKN(AS_TK), TKN(AVG_TK), TKN(MIN_TK), TKN(MAX_TK), TKN(COUNT_TK), TKN(ASC_TK), TKN(DESC_TK), TKN(CASE_TK),
    TKN(END_TK), TKN(ELSE_TK), TKN(WHEN_TK), TKN(THEN_TK), TKN(DECIMAL_CONSTANT), TKN(FLOAT_CONSTANT), TKN(DATE_TK), TKN(INT_TK),
    TKN(BIGINT_TK), TKN(CHAR_TK), TKN(VARCHAR_TK), TKN(DECIMAL_TK),
};
#undef TKN

// parseSql (parseSql.h:130-166) with the tokens read from a file ("NAME text" per line) instead of yylex()
Query parseTokens(const std::string& path) {
    Query query = {Query::UNKNOWN, nullptr, nullptr, nullptr, nullptr, nullptr, "", nullptr, "", ",", 0, false, false, nullptr, {}, {},
                   false, false, 0};
    std::ifstream f(path);
    if (!f.is_open()) die("cannot open " + path);
    void* parser = ParseAlloc(malloc);
    std::string line;
    while (std::getline(f, line)) {
        if (line == "ERROR") { query.parseError = true; break; }        // the tokenizer met a character no rule matches
        size_t sp = line.find(' ');
        std::string name = line.substr(0, sp), text = sp == std::string::npos ? "" : line.substr(sp + 1);
        int code = -1;
        for (auto& t : kTokenNames) if (name == t.name) code = t.code;
        if (code < 0) die("unknown token name " + name);
        Expr* node = initializeIfConstant(code, text);
        Parse(parser, code, node, &query);
    }
    Parse(parser, 0, 0, &query);
    ParseFree(parser, free);
    return query;
}

void dumpExpr(Expr* e, std::ostream& os) {
    os << "(" << exprTagNames[e->tag];
    if (e->tag == Expr::ATTRIBUTE || e->tag == Expr::AS || e->tag == Expr::TABLE) os << " " << e->symbol;
    if (e->tag == Expr::CONSTANT) os << " " << serializeType(e->type) << " [" << serializeSqlValue(e->value, e->type) << "]";
    if (e->tag == Expr::TYPECAST) os << " " << serializeType(e->type);
    // operands as the reference's consumers see them (child, child->next for BINARY), not the whole sibling chain
    const int limit = e->structureTag == Expr::UNARY ? 1 : e->structureTag == Expr::BINARY ? 2 : e->structureTag == Expr::LITERAL ? 0 : 1 << 30;
    int n = 0;
    for (Expr* c = e->child; c && n < limit; c = c->next, n++) { os << " "; dumpExpr(c, os); }
    os << ")";
}
void dumpList(const char* label, Expr* e, std::ostream& os) {
    os << label << ":";
    for (; e; e = e->next) { os << " "; dumpExpr(e, os); }
    os << "\n";
}
void dumpQuery(Query& q, std::ostream& os) {
    if (q.parseError) { os << "SYNTAX ERROR\n"; return; }
    if (q.tag == Query::SELECT) {
        os << "SELECT\n";
        dumpList("select", q.selectExpr, os); dumpList("from", q.fromExpr, os); dumpList("where", q.whereExpr, os);
        dumpList("groupby", q.groupbyExpr, os); dumpList("orderby", q.orderbyExpr, os);
        os << "limit: " << (q.useLimit ? std::to_string(q.limit) : std::string("none")) << "\n";
    } else if (q.tag == Query::CREATE_TABLE) {
        os << "CREATE_TABLE " << q.tableName << "\n";
        for (Expr* e = q.schemaExpr; e; e = e->next) os << "column " << e->symbol << " " << serializeType(e->type) << "\n";
    } else if (q.tag == Query::BULK_INSERT) {
        os << "BULK_INSERT " << q.tableName << "\nfile " << q.fileName << "\nfieldterminator " << q.fieldTerminator << "\nfirstrow "
           << q.firstRow << "\n";
    } else os << "UNKNOWN\n";
}
void dumpVec(ExprVec& v, std::ostream& os) {
    os << " [";
    for (size_t i = 0; i < v.size(); i++) { if (i) os << " "; dumpExpr(v[i], os); }
    os << "]";
}
void dumpOp(RelOperator* o, Database& db, std::ostream& os) {
    // by dynamic type, not by RelOperator::tag: MaterializeOp constructs itself with the SELECTION tag (materialize.h:33)
    if (dynamic_cast<MaterializeOp*>(o)) os << "MATERIALIZE";
    else if (auto* sc = dynamic_cast<ScanOp*>(o)) {
        os << "SCAN";
        for (auto& r : db.relations) if (&r.second == sc->_rel) os << " " << r.first;
    }
    else if (auto* se = dynamic_cast<SelectionOp*>(o)) { os << "SELECTION ["; dumpExpr(se->_condition, os); os << "]"; }
    else if (auto* pr = dynamic_cast<ProjectionOp*>(o)) { os << "PROJECTION"; dumpVec(pr->_expr, os); }
    else if (auto* hj = dynamic_cast<HashJoinOp*>(o)) { os << "HASHJOIN single=" << (hj->_singleMatch ? 1 : 0); dumpVec(hj->_equalities, os); }
    else if (auto* ag = dynamic_cast<AggregationOp*>(o)) { os << "AGGREGATION"; dumpVec(ag->_aggExpr, os); dumpVec(ag->_groupExpr, os); }
    else if (auto* ob = dynamic_cast<OrderByOp*>(o)) { os << "ORDERBY"; dumpVec(ob->_orderExpressions, os); }
    else if (dynamic_cast<NestedLoopsJoinOp*>(o)) os << "NESTEDLOOPSJOIN";
    else os << "UNDEFINED";
    os << " {";
    for (size_t i = 0; i < o->children.size(); i++) { if (i) os << " "; dumpOp(o->children[i], db, os); }
    os << "}";
}

std::map<std::string, SqlType> identTypes(Database& db) {
    std::map<std::string, SqlType> res;
    for (auto const& rel : db.relations)
        for (auto const& att : rel.second._schema._attribs) res[att.name] = att.type;
    return res;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) die("usage: ref_harness CASEFILE [--threads N] [--blocksize B] [--repeat K] [--out FILE] [--quiet]");
    std::string casePath = argv[1];
    int threads = 1, repeat = 1; bool quiet = false; std::string outPath;
    std::string engine = "flounder"; int device = 0;
    std::string sqlTokens; bool dumpParse = false, dumpPlan = false;
    bool grow = false; int compat = -1;
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--threads") threads = atoi(argv[++i]);
        else if (a == "--blocksize") DataBlock::Size = strtoull(argv[++i], nullptr, 10);
        else if (a == "--repeat") repeat = atoi(argv[++i]);
        else if (a == "--out") outPath = argv[++i];
        else if (a == "--quiet") quiet = true;
        else if (a == "--engine") engine = argv[++i];        // "flounder" (the reference's own JIT) | "hip"
        else if (a == "--device") device = atoi(argv[++i]);
        else if (a == "--sql-tokens") sqlTokens = argv[++i];   // plan from the reference's parser + planner instead of the case's op lines
        else if (a == "--dump-parse") dumpParse = true;
        else if (a == "--dump-plan") dumpPlan = true;
        else if (a == "--compat") compat = atoi(argv[++i]);    // rsq_compat bits of the HIP context (default: the binding's)
        else if (a == "--grow") grow = true;                   // run, load every table AGAIN behind its rows (BULK INSERT appends, execute.h:332-388), run again: both results

        else die("unknown option " + a);
    }

    if (dumpParse) {
        if (sqlTokens.empty()) die("--dump-parse needs --sql-tokens");
        Query q = parseTokens(sqlTokens);
        dumpQuery(q, std::cout);
        return 0;
    }
    CaseSpec c = parseCase(casePath);
    Database db;
    Timer tLoad;
    for (auto& t : c.tables) loadTable(c, t, db, engine == "hip");
    std::cerr << "#load_ms " << tLoad.get() << std::endl;

    std::unique_ptr<Relation> result;
#ifdef RSQ_WITH_HIP_BINDING
    std::unique_ptr<resql_hip::JitContextHip> hip;
    if (engine == "hip") {
        JitConfig jc; jc.numThreads = threads;
        try {
            hip = compat < 0 ? std::make_unique<resql_hip::JitContextHip>(jc, device) : std::make_unique<resql_hip::JitContextHip>(jc, device, (uint32_t)compat);
            for (auto& t : c.tables)
                if (!t.tblPath.empty()) hip->bulkInsert(&db.relations[t.name], t.name, resolve(c, t.tblPath), '|');
        }
        catch (ResqlError& err) { std::cerr << "ResqlError: " << err.message(); return 3; }
    }
#else
    if (engine == "hip") die("built without the HIP binding");
#endif
    if (grow) repeat = 2;
    auto emit = [&](Relation& rel, std::ostream& os) {
        os << "#schema ";
        for (auto& a : rel._schema._attribs) os << a.name << ":" << serializeType(a.type) << "|";
        os << "\n";
        serializeRelation(rel, os);
    };
    for (int rep = 0; rep < repeat; rep++) {
        if (grow && rep == 1) {
            if (!quiet) { emit(*result, std::cout); std::cout << "#grown\n"; }
            for (auto& t : c.tables) loadTable(c, t, db, engine == "hip");
#ifdef RSQ_WITH_HIP_BINDING
            if (hip)
                for (auto& t : c.tables)
                    if (!t.tblPath.empty()) hip->bulkInsert(&db.relations[t.name], t.name, resolve(c, t.tblPath), '|');
#endif
        }
        // plans are single use (operators own iterators / hash tables): rebuild per repetition
        PlanBuilder pb{c, db, {}, {}};
        RelOperator* root = nullptr;
        if (!sqlTokens.empty()) {
            // executeStatement / executeSelect (execute.h:508-545, 250-260) minus the tokenizer
            Query query = parseTokens(sqlTokens);
            if (query.parseError) { std::cerr << "ResqlError: Syntax error."; return 3; }
            if (query.tag != Query::SELECT) die("--sql-tokens expects a select statement");
            try { buildQuery(query, db); }
            catch (ResqlError& err) { std::cerr << "ResqlError: " << err.message(); return 3; }
            if (query.plan == nullptr) { std::cerr << "ResqlError: Could not generate query plan."; return 3; }
            root = query.plan;
            c.requestAll = query.requestAll;
            if (dumpPlan) {
                std::cout << "limit " << (query.useLimit ? std::to_string(query.limit) : std::string("none")) << " requestall "
                          << (query.requestAll ? 1 : 0) << "\n";
                dumpOp(root, db, std::cout);
                std::cout << "\n";
                return 0;
            }
        } else {
            root = pb.op(c.root);
            if (c.hasLimit) root->addLimit(c.limit);
        }

#ifdef RSQ_WITH_HIP_BINDING
        if (hip) {
            // the drop-in: same plan objects, the HIP engine instead of produceFlounder/compile/execute
            try { result = hip->run(root, c.requestAll); }
            catch (ResqlError& err) { std::cerr << "ResqlError: " << err.message(); return 3; }
            // NOT root->deletePlan(): HashJoinOp::~HashJoinOp (hashjoin.h:83-85) frees _ht unconditionally and
            // crashes when produceFlounder never allocated it (AggregationOp's destructor has the null check,
            // aggregation.h:66-70).  A ReSQL-side integration adds the same check; the harness leaks the plan.
            // (every repetition is what a ReSQL host issues per SELECT: a fresh rsq_query, compiled, executed ONCE, destroyed - exec_ms IS
            // the first execution; from the second repetition on the context's plan memo and arenas serve it)
            std::cerr << "#timing compile_ms " << hip->report.compilationTime << " exec_ms " << hip->report.executionTime
                      << " instrs " << hip->report.numMachineInstructions << " kernel_ms " << hip->kernelTimeMs
                      << " first_exec_ms " << hip->report.executionTime << " repetition " << rep << std::endl;
            continue;
        }
#endif
        // the reference's executeSelectPlan call sequence (execute.h:213-247)
        ExpressionContext exprCtx;
        root->defineExpressionsForPlan(exprCtx);
        auto ids = identTypes(db);
        exprCtx.deriveExpressionTypes(ids);
        exprCtx.unifyExpressions();
        JitConfig jc;
        jc.numThreads = threads;
        jc.emitMachineCode = true;          // asmjit; nasm is not in this image
        JitContextFlounder ctx(jc);
        ctx.requestAll = c.requestAll;
        try {
            root->produceFlounder(ctx, {});
            ctx.compile();
        } catch (ResqlError& err) {
            std::cerr << "ResqlError: " << err.message();
            return 3;
        }
        ctx.execute();
        result = root->retrieveResult();
        root->deletePlan();
        std::cerr << "#timing compile_ms " << ctx.report.compilationTime
                  << " exec_ms " << ctx.report.executionTime
                  << " instrs " << ctx.report.numMachineInstructions << std::endl;
    }

    if (!quiet) {
        std::ofstream fo;
        std::ostream* os = &std::cout;
        if (!outPath.empty()) { fo.open(outPath); os = &fo; }
        *os << "#schema ";
        for (auto& a : result->_schema._attribs) *os << a.name << ":" << serializeType(a.type) << "|";
        *os << "\n";
        serializeRelation(*result, *os);
    }
    return 0;
}
