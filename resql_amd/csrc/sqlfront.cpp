// sqlfront.cpp — SQL text -> plan description: the caller side of the hot path.
//
// The engine's own implementation of what sits in front of executeSelectPlan in the reference:
//   tokens   reference src/parser/lexer.y (flex: longest match, earlier rule wins a tie)
//   grammar  reference src/parser/parser.y (Lemon: LALR(1) with the %left precedence ladder of lines 11-23;
//            a hand-written precedence-climbing parser reproduces its shift / reduce decisions) and
//            src/parser/parseSql.h:97-127 (constants are typed from their token text)
//   planner  reference src/planner.h:409-497 buildQuery and the helpers above it: unify select / group by,
//            extract aggregations, push selections to the scans, hash joins from equalities (build side = smaller
//            relation, join order by selections then probe size), remaining conditions, aggregation, projection,
//            order by, materialize, limit
// The output is the plain-C plan description of include/resql_plan.h — exactly what a ReSQL host hands to
// rsq_query_compile after its own planner ran — so SQL text and hand-built operator trees take the same path.
// Behaviour that looks odd is the reference's and is kept (each place says so): results have to be identical.
#include <algorithm>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <sstream>

#include "sqlfront.h"

namespace rsq {
namespace sql {

// ================================================================================================
// tokens (lexer.y)
// ================================================================================================
namespace {

struct Keyword { const char* text; const char* name; };
// in the order of lexer.y: on equal length the earlier rule wins, and every keyword precedes {ID}
const Keyword kKeywords[] = {
    {"select", "SELECT_TK"}, {"from", "FROM"}, {"where", "WHERE"}, {"group by", "GROUPBY"}, {"order by", "ORDERBY"},
    {"limit", "LIMIT_TK"}, {"asc", "ASC_TK"}, {"desc", "DESC_TK"}, {"create table", "CREATE_TABLE_TK"},
    {"bulk insert", "BULK_INSERT_TK"}, {"fieldterminator", "FIELDTERMINATOR_TK"}, {"firstrow", "FIRSTROW_TK"},
    {"with", "WITH_TK"}, {"sum", "SUM_TK"}, {"count", "COUNT_TK"}, {"avg", "AVG_TK"}, {"min", "MIN_TK"}, {"max", "MAX_TK"},
    {"between", "BETWEEN_TK"}, {"(", "LPAREN"}, {")", "RPAREN"}, {"+", "PLUS_TK"}, {"-", "MINUS_TK"}, {"*", "MUL_TK"},
    {"/", "DIV_TK"}, {">=", "GE_TK"}, {">", "GT_TK"}, {"<=", "LE_TK"}, {"<", "LT_TK"}, {"=", "EQ_TK"}, {"<>", "NEQ_TK"},
    {",", "COMMA"}, {"::", "TYPECAST_TK"}, {"and", "AND_TK"}, {"in", "IN_TK"}, {"like", "LIKE_TK"}, {"or", "OR_TK"},
    {"as", "AS_TK"}, {"bigint", "BIGINT_TK"}, {"int", "INT_TK"}, {"date", "DATE_TK"}, {"decimal", "DECIMAL_TK"},
    {"char", "CHAR_TK"}, {"varchar", "VARCHAR_TK"}, {"case", "CASE_TK"}, {"when", "WHEN_TK"}, {"then", "THEN_TK"},
    {"else", "ELSE_TK"}, {"end", "END_TK"},
};

bool isDigit(char c) { return c >= '0' && c <= '9'; }
bool isIdStart(char c) { return c >= 'a' && c <= 'z'; }
bool isIdChar(char c) { return isIdStart(c) || isDigit(c) || c == '_'; }

// lengths of the three numeric rules at s[i..): FLOAT, DECIMAL, INTEGER (0 = no match)
void numberMatches(const std::string& s, size_t i, size_t& fl, size_t& dec, size_t& in) {
    fl = dec = in = 0;
    size_t a = i;
    while (a < s.size() && isDigit(s[a])) a++;
    const size_t intDigits = a - i;
    if (intDigits) in = intDigits;
    size_t mantissa = intDigits;                          // {DIGIT}+ form of the float mantissa
    if (a < s.size() && s[a] == '.') {
        size_t b = a + 1;
        while (b < s.size() && isDigit(s[b])) b++;
        const size_t fracDigits = b - (a + 1);
        if (intDigits || fracDigits) { dec = b - i; mantissa = dec; }     // D+.D*  |  D*.D+
    }
    if (mantissa) {
        size_t e = i + mantissa;
        if (e < s.size() && s[e] == 'e') {
            size_t x = e + 1;
            if (x < s.size() && (s[x] == '+' || s[x] == '-')) x++;
            size_t y = x;
            while (y < s.size() && isDigit(s[y])) y++;
            if (y > x) fl = y - i;
        }
    }
}

// (\"([^\\\"]|\\.)*\")|(\'([^\\\']|\\.)*\')   ('.' does not match a newline)
size_t stringMatch(const std::string& s, size_t i) {
    const char q = s[i];
    if (q != '"' && q != '\'') return 0;
    size_t p = i + 1;
    while (p < s.size()) {
        if (s[p] == q) return p + 1 - i;
        if (s[p] == '\\') {
            if (p + 1 >= s.size() || s[p + 1] == '\n') return 0;
            p += 2;
        } else p++;
    }
    return 0;
}

}  // namespace

std::vector<Token> tokenize(const std::string& s, bool& error) {
    std::vector<Token> out;
    error = false;
    size_t i = 0;
    while (i < s.size()) {
        // candidates in rule order; the longest wins, the first on a tie
        size_t best = 0; const char* name = nullptr; int kind = 0;   // kind: 0 token, 1 skip
        size_t fl, dec, in;
        numberMatches(s, i, fl, dec, in);
        if (fl > best) { best = fl; name = "FLOAT_CONSTANT"; kind = 0; }
        if (dec > best) { best = dec; name = "DECIMAL_CONSTANT"; kind = 0; }
        if (in > best) { best = in; name = "INTEGER_CONSTANT"; kind = 0; }
        const size_t str = stringMatch(s, i);
        if (str > best) { best = str; name = "STRING_CONSTANT"; kind = 0; }
        for (const Keyword& k : kKeywords) {
            const size_t n = strlen(k.text);
            if (n > best && s.compare(i, n, k.text) == 0) { best = n; name = k.name; kind = 0; }
        }
        if (isIdStart(s[i])) {
            size_t p = i + 1;
            while (p < s.size() && isIdChar(s[p])) p++;
            if (p - i > best) { best = p - i; name = "IDENTIFIER"; kind = 0; }
        }
        if (s[i] == '-' && i + 1 < s.size() && s[i + 1] == '-') {        // "--"[^\n]*"\n"
            size_t p = i + 2;
            while (p < s.size() && s[p] != '\n') p++;
            if (p < s.size() && p + 1 - i > best) { best = p + 1 - i; name = nullptr; kind = 1; }
        }
        if (s[i] == ' ' || s[i] == '\t' || s[i] == '\n') {
            size_t p = i;
            while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n')) p++;
            if (p - i > best) { best = p - i; name = nullptr; kind = 1; }
        }
        if (best == 0) { error = true; return out; }                      // "Unrecognized character"
        if (kind == 0) out.push_back(Token{name, s.substr(i, best)});
        i += best;
    }
    return out;
}

// ================================================================================================
// grammar (parser.y)
// ================================================================================================
namespace {

struct SyntaxError {};

// precedence ladder of parser.y:11-23, lowest first
enum { P_OR = 1, P_AND, P_CMP, P_EQ, P_BETWEEN, P_IN, P_ADD, P_MUL, P_LIKE, P_CAST };

struct Parser {
    ExprPool& pool;
    const std::vector<Token>& t;
    size_t p = 0;
    bool sawFloat = false;
    Parser(ExprPool& pl, const std::vector<Token>& toks) : pool(pl), t(toks) {}

    bool at(const char* name) const { return p < t.size() && strcmp(t[p].name, name) == 0; }
    bool accept(const char* name) { if (at(name)) { p++; return true; } return false; }
    const Token& expect(const char* name) { if (!at(name)) throw SyntaxError(); return t[p++]; }

    // ---- ExprGen:: (expressions.h:518-705); binaryExpr links left->next = right and leaves right->next alone ----
    Expr* literal(int tag, const std::string& sym) { return pool.make(tag, LITERAL, sym); }
    Expr* unary(int tag, const std::string& sym, Expr* c) { Expr* e = pool.make(tag, UNARY, sym); e->child = c; return e; }
    Expr* binary(int tag, const std::string& sym, Expr* l, Expr* r) {
        l->next = r;
        Expr* e = pool.make(tag, BINARY, sym);
        e->child = l;
        return e;
    }
    Expr* constant(const std::string& sym, int category) { return pool.constant(sym, category); }
    // copyExpr (expressions.h:258-262): a shallow copy — the children are shared with the original
    Expr* shallowCopy(Expr* e) { Expr* c = pool.make(e->tag, e->structure, e->symbol); *c = *e; return c; }
    // ExprGen::copy (expressions.h:688-702, "todo: fix"): only the LAST child survives and siblings are not linked.
    // Literals copy fine; a copied operator keeps one operand — the reference reads through a null pointer when it types
    // such a node, here the plan description refuses it ("binary expression needs two children").  The parse is kept
    // identical to the reference's either way.
    Expr* deepCopy(Expr* e) {
        Expr* c = pool.make(e->tag, e->structure, e->symbol);
        c->type = e->type; c->ival = e->ival; c->negated = e->negated; c->category = e->category;
        for (Expr* k = e->child; k; k = k->next) c->child = deepCopy(k);
        return c;
    }

    // parseSql.h:97-127 initializeIfConstant
    Expr* constantFromToken(const Token& k) {
        if (strcmp(k.name, "DECIMAL_CONSTANT") == 0) return constant(k.text, RSQ_DECIMAL);
        if (strcmp(k.name, "FLOAT_CONSTANT") == 0) {
            // the reference parses these (parseFloatConstant, expressions.h:469-472) but has no code generation for FLOAT;
            // the statement is parsed to its end first (a syntax error is the more specific answer), then refused
            sawFloat = true;
            return constant("0", RSQ_BIGINT);
        }
        if (strcmp(k.name, "INTEGER_CONSTANT") == 0) return constant(k.text, RSQ_BIGINT);
        std::string str = k.text.substr(1, k.text.length() - 2);          // STRING_CONSTANT: quotes removed, escapes kept
        return constant(str, str.length() == 1 ? RSQ_CHAR : RSQ_VARCHAR);
    }
    int64_t integerToken() {
        Expr* c = constantFromToken(expect("INTEGER_CONSTANT"));
        return c->ival;
    }

    // type(A) ::= INT_TK | BIGINT_TK | DATE_TK | CHAR_TK ( n ) | VARCHAR_TK ( n ) | DECIMAL_TK ( p , s )
    Type type() {
        if (accept("INT_TK")) return Type(RSQ_INT);
        if (accept("BIGINT_TK")) return Type(RSQ_BIGINT);
        if (accept("DATE_TK")) return Type(RSQ_DATE);
        if (accept("CHAR_TK")) { expect("LPAREN"); Type x(RSQ_CHAR); x.len = (int)integerToken(); expect("RPAREN"); return x; }
        if (accept("VARCHAR_TK")) { expect("LPAREN"); Type x(RSQ_VARCHAR); x.len = (int)integerToken(); expect("RPAREN"); return x; }
        if (accept("DECIMAL_TK")) {
            expect("LPAREN"); int64_t pr = integerToken(); expect("COMMA"); int64_t sc = integerToken(); expect("RPAREN");
            return Type::decimal((int)(uint8_t)pr, (int)(uint8_t)sc);
        }
        throw SyntaxError();
    }

    bool startsValue() const {
        return at("INTEGER_CONSTANT") || at("DECIMAL_CONSTANT") || at("FLOAT_CONSTANT") || at("STRING_CONSTANT");
    }

    Expr* primary() {
        if (at("SUM_TK") || at("AVG_TK") || at("MIN_TK") || at("MAX_TK")) {
            const std::string k = t[p++].name;
            expect("LPAREN"); Expr* c = expr(0); expect("RPAREN");
            if (k == "SUM_TK") return unary(RSQ_E_SUM, "sum", c);
            if (k == "AVG_TK") return unary(RSQ_E_AVG, "avg", c);
            if (k == "MIN_TK") return unary(RSQ_E_MIN, "min", c);
            return unary(RSQ_E_MAX, "max", c);
        }
        if (accept("COUNT_TK")) {
            expect("LPAREN");
            Expr* c;
            if (at("MUL_TK")) { p++; c = literal(RSQ_E_STAR, "*"); } else c = expr(0);
            expect("RPAREN");
            return unary(RSQ_E_COUNT, "count", c);
        }
        if (accept("LPAREN")) { Expr* e = expr(0); expect("RPAREN"); return e; }
        if (at("IDENTIFIER")) {
            Expr* a = literal(RSQ_E_ATTRIBUTE, t[p++].text);
            // expr ::= IDENTIFIER ASC_TK | IDENTIFIER DESC_TK (shift wins over reducing the bare identifier)
            if (accept("ASC_TK")) return unary(RSQ_E_ASC, "asc", a);
            if (accept("DESC_TK")) return unary(RSQ_E_DESC, "desc", a);
            return a;
        }
        if (accept("MINUS_TK")) {                         // value ::= MINUS_TK constant: the VALUE is negated, the text is not
            if (at("INTEGER_CONSTANT") || at("DECIMAL_CONSTANT") || at("FLOAT_CONSTANT")) {
                Expr* c = constantFromToken(t[p++]);
                c->ival = (int64_t)(0 - (uint64_t)c->ival);
                c->negated = true;
                return c;
            }
            throw SyntaxError();
        }
        if (startsValue()) return constantFromToken(t[p++]);
        if (accept("DATE_TK")) return constant(constantFromToken(expect("STRING_CONSTANT"))->symbol, RSQ_DATE);
        if (accept("CASE_TK")) {
            // whenThenList is linked through ->next, then `B->next = C` on the FIRST element (parser.y:143-146): a CASE with
            // several WHENs keeps only the first one plus the ELSE — the reference's behaviour, kept
            std::vector<Expr*> whens;
            while (accept("WHEN_TK")) {
                Expr* w = expr(0); expect("THEN_TK"); Expr* th = expr(0);
                whens.push_back(binary(RSQ_E_WHENTHEN, "whenThen", w, th));
            }
            if (whens.empty()) throw SyntaxError();
            Expr* els = nullptr;
            if (accept("ELSE_TK")) els = expr(0);
            expect("END_TK");
            for (size_t i = 0; i + 1 < whens.size(); i++) whens[i]->next = whens[i + 1];
            whens[0]->next = els;
            Expr* c = pool.make(RSQ_E_CASE, OTHER, "case");
            c->child = whens[0];
            return c;
        }
        throw SyntaxError();
    }

    static int infixPrec(const Token& k) {
        const std::string n = k.name;
        if (n == "OR_TK") return P_OR;
        if (n == "AND_TK") return P_AND;
        if (n == "LT_TK" || n == "GT_TK" || n == "LE_TK" || n == "GE_TK") return P_CMP;
        if (n == "EQ_TK" || n == "NEQ_TK") return P_EQ;
        if (n == "BETWEEN_TK") return P_BETWEEN;
        if (n == "IN_TK") return P_IN;
        if (n == "PLUS_TK" || n == "MINUS_TK") return P_ADD;
        if (n == "MUL_TK" || n == "DIV_TK") return P_MUL;
        if (n == "LIKE_TK") return P_LIKE;
        if (n == "TYPECAST_TK") return P_CAST;
        return 0;
    }

    // every operator is %left: an operator of the same level to the right reduces first
    Expr* expr(int minPrec) {
        Expr* l = primary();
        for (;;) {
            if (p >= t.size()) return l;
            const int pr = infixPrec(t[p]);
            if (pr == 0 || pr <= minPrec) return l;
            const std::string n = t[p++].name;
            if (n == "BETWEEN_TK") {
                // expr BETWEEN expr AND expr: the lower bound ends at the AND, the upper bound binds tighter than BETWEEN
                // (the reduce of this rule, precedence BETWEEN, beats both the shift of a weaker operator and the
                // reduce of `expr AND expr`)
                Expr* lo = expr(P_AND);
                expect("AND_TK");
                Expr* hi = expr(P_BETWEEN);
                Expr* ge = binary(RSQ_E_GE, ">=", shallowCopy(l), lo);
                Expr* le = binary(RSQ_E_LE, "<=", l, hi);
                l = binary(RSQ_E_AND, "and", ge, le);
            } else if (n == "IN_TK") {
                expect("LPAREN");
                std::vector<Expr*> list;
                do { list.push_back(expr(0)); } while (accept("COMMA"));
                expect("RPAREN");
                // parser.y:118-131: eq(copy(B), copy(first)); every further entry is OR-ed in FRONT of the result so far
                Expr* res = binary(RSQ_E_EQ, ">=", deepCopy(l), deepCopy(list[0]));
                for (size_t i = 1; i < list.size(); i++)
                    res = binary(RSQ_E_OR, "or", binary(RSQ_E_EQ, ">=", deepCopy(l), deepCopy(list[i])), res);
                l = res;
            } else if (n == "LIKE_TK") {
                l = binary(RSQ_E_LIKE, "like", l, constantFromToken(expect("STRING_CONSTANT")));
            } else if (n == "TYPECAST_TK") {
                Expr* c = unary(RSQ_E_TYPECAST, "typecast", l);
                c->type = type();
                c->explicitCast = true;
                l = c;
            } else {
                Expr* r = expr(pr);
                int tag; const char* sym;
                if (n == "OR_TK") { tag = RSQ_E_OR; sym = "or"; }
                else if (n == "AND_TK") { tag = RSQ_E_AND; sym = "and"; }
                else if (n == "LT_TK") { tag = RSQ_E_LT; sym = "<"; }
                else if (n == "GT_TK") { tag = RSQ_E_GT; sym = ">"; }
                else if (n == "LE_TK") { tag = RSQ_E_LE; sym = "<="; }
                else if (n == "GE_TK") { tag = RSQ_E_GE; sym = ">="; }
                else if (n == "EQ_TK") { tag = RSQ_E_EQ; sym = ">="; }       // sic: expressions.h:576-584 label EQ / NEQ ">="
                else if (n == "NEQ_TK") { tag = RSQ_E_NEQ; sym = ">="; }
                else if (n == "PLUS_TK") { tag = RSQ_E_ADD; sym = "+"; }
                else if (n == "MINUS_TK") { tag = RSQ_E_SUB; sym = "-"; }
                else if (n == "MUL_TK") { tag = RSQ_E_MUL; sym = "*"; }
                else { tag = RSQ_E_DIV; sym = "/"; }
                l = binary(tag, sym, l, r);
            }
        }
    }

    // X ::= item COMMA X | item, linked through ->next (right recursive: the links are set back to front)
    Expr* exprList(bool namable) {
        std::vector<Expr*> v;
        do {
            Expr* e = expr(0);
            if (namable && accept("AS_TK")) e = unary(RSQ_E_AS, expect("IDENTIFIER").text, e);
            v.push_back(e);
        } while (accept("COMMA"));
        for (size_t i = v.size(); i-- > 0;) v[i]->next = i + 1 < v.size() ? v[i + 1] : nullptr;
        return v[0];
    }

    void statement(Statement& st) {
        if (accept("CREATE_TABLE_TK")) {
            st.kind = Statement::CREATE_TABLE;
            st.tableName = expect("IDENTIFIER").text;
            expect("LPAREN");
            do {
                const std::string col = expect("IDENTIFIER").text;
                st.schema.emplace_back(col, type());
            } while (accept("COMMA"));
            expect("RPAREN");
        } else if (accept("BULK_INSERT_TK")) {
            st.kind = Statement::BULK_INSERT;
            st.tableName = expect("IDENTIFIER").text;
            expect("FROM");
            st.fileName = constantFromToken(expect("STRING_CONSTANT"))->symbol;
            if (accept("WITH_TK")) {
                expect("LPAREN");
                do {
                    if (accept("FIRSTROW_TK")) { expect("EQ_TK"); st.firstRow = (uint64_t)integerToken(); }
                    else if (accept("FIELDTERMINATOR_TK")) { expect("EQ_TK"); st.fieldTerminator = constantFromToken(expect("STRING_CONSTANT"))->symbol; }
                    else throw SyntaxError();
                } while (accept("COMMA"));
                expect("RPAREN");
            }
        } else {
            // entry ::= select from where groupby orderby limit
            st.kind = Statement::SELECT;
            expect("SELECT_TK");
            // select ::= SELECT_TK MUL_TK: a lone '*' (a select list cannot start with one otherwise)
            if (at("MUL_TK")) { p++; st.selectExpr = literal(RSQ_E_STAR, "*"); }
            else st.selectExpr = exprList(true);
            if (accept("FROM")) {
                std::vector<Expr*> v;
                do { v.push_back(literal(RSQ_E_TABLE, expect("IDENTIFIER").text)); } while (accept("COMMA"));
                for (size_t i = 0; i + 1 < v.size(); i++) v[i]->next = v[i + 1];
                st.fromExpr = v[0];
            }
            if (accept("WHERE")) st.whereExpr = expr(0);
            if (accept("GROUPBY")) st.groupbyExpr = exprList(false);
            if (accept("ORDERBY")) st.orderbyExpr = exprList(false);
            if (accept("LIMIT_TK")) { st.useLimit = true; st.limit = integerToken(); }
        }
        if (p != t.size()) throw SyntaxError();
    }
};

}  // namespace

void parse(const std::string& text, ExprPool& pool, Statement& st) {
    bool lexError = false;
    std::vector<Token> toks = tokenize(text, lexError);
    if (lexError) throw Error(RSQ_ERR_INVALID, "Syntax error.");
    Parser ps(pool, toks);
    try { ps.statement(st); }
    catch (const SyntaxError&) { throw Error(RSQ_ERR_INVALID, "Syntax error."); }
    if (ps.sawFloat) failType("FLOAT constants are not supported (the reference parses them but cannot generate code for FLOAT)");
}

// ---- canonical text of a parse (tests compare it with the reference's Lemon parser driven by the same tokens) ----
static void dumpExpr(const Expr* e, std::string& out) {
    out += "("; out += exprTagNames[e->tag];
    if (e->tag == RSQ_E_ATTRIBUTE || e->tag == RSQ_E_AS || e->tag == RSQ_E_TABLE) out += " " + e->symbol;
    if (e->tag == RSQ_E_CONSTANT) {
        Val v; v.i = e->ival;
        if (e->type.isString()) v.s = e->symbol.c_str();
        out += " " + serializeType(e->type) + " [" + serializeSqlValue(v, e->type) + "]";
    }
    if (e->tag == RSQ_E_TYPECAST) out += " " + serializeType(e->type);
    // operands as the consumers see them: a unary node uses its child, a binary node its child and that child's next,
    // whatever else the sibling chain holds (planner rewrites leave stale links behind, e.g. BETWEEN's two halves)
    const int limit = e->structure == UNARY ? 1 : e->structure == BINARY ? 2 : e->structure == LITERAL ? 0 : 1 << 30;
    int n = 0;
    for (const Expr* c = e->child; c && n < limit; c = c->next, n++) { out += " "; dumpExpr(c, out); }
    out += ")";
}
static void dumpList(const char* label, const Expr* e, std::string& out) {
    out += label; out += ":";
    for (; e; e = e->next) { out += " "; dumpExpr(e, out); }
    out += "\n";
}
std::string dumpStatement(const Statement& st) {
    std::string out;
    if (st.kind == Statement::SELECT) {
        out += "SELECT\n";
        dumpList("select", st.selectExpr, out);
        dumpList("from", st.fromExpr, out);
        dumpList("where", st.whereExpr, out);
        dumpList("groupby", st.groupbyExpr, out);
        dumpList("orderby", st.orderbyExpr, out);
        out += "limit: " + (st.useLimit ? std::to_string((long long)st.limit) : std::string("none")) + "\n";
    } else if (st.kind == Statement::CREATE_TABLE) {
        out += "CREATE_TABLE " + st.tableName + "\n";
        for (auto& c : st.schema) out += "column " + c.first + " " + serializeType(c.second) + "\n";
    } else if (st.kind == Statement::BULK_INSERT) {
        out += "BULK_INSERT " + st.tableName + "\nfile " + st.fileName + "\nfieldterminator " + st.fieldTerminator + "\nfirstrow " +
               std::to_string((unsigned long long)st.firstRow) + "\n";
    }
    return out;
}

// ================================================================================================
// planner (planner.h)
// ================================================================================================
namespace {

typedef std::vector<Expr*> ExprVec;

ExprVec listToVector(Expr* e) { ExprVec v; for (; e; e = e->next) v.push_back(e); return v; }      // expressions.h:970-977

// expressions.h:133-173: same tag, attributes also by name — constants match whatever their value
bool exprEquals(const Expr* a, const Expr* b) {
    if (a->tag != b->tag) return false;
    if (a->tag == RSQ_E_ATTRIBUTE && a->symbol != b->symbol) return false;
    return true;
}
bool traceMatch(const Expr* hay, const Expr* needle) {
    if (!hay) return needle == nullptr;
    if (!needle) return false;
    if (!exprEquals(hay, needle)) return false;
    bool ok = true;
    const Expr *h = hay->child, *n = needle->child;
    while (h && n) { ok &= traceMatch(h, n); h = h->next; n = n->next; }
    ok &= (h == n);
    return ok;
}
// planner.h:13-33: a select subtree that matches a group-by expression IS that expression from now on
Expr* matchAndUnify(Expr* select, Expr* group) {
    if (traceMatch(select, group)) { group->next = select->next; return group; }
    if (!select->child) return select;
    Expr* fc = select->child;
    select->child = matchAndUnify(fc, group);
    Expr* prev = fc;
    Expr* c = fc->next;
    while (c) { prev->next = matchAndUnify(c, group); prev = c; c = c->next; }
    return select;
}

bool isAggregation(const Expr* e) {
    return e->tag == RSQ_E_SUM || e->tag == RSQ_E_MIN || e->tag == RSQ_E_MAX || e->tag == RSQ_E_AVG || e->tag == RSQ_E_COUNT;
}
void filterAggregations(Expr* e, ExprVec& out) {          // planner.h:36-51 filterExpr with isAggregationExpr
    if (!e) return;
    if (isAggregation(e)) out.push_back(e);
    for (Expr* c = e->child; c; c = c->next) filterAggregations(c, out);
}
void collectConjunctionsInner(Expr* e, ExprVec& out) {     // planner.h:54-73
    if (!e || e->tag != RSQ_E_AND) return;
    for (Expr* c = e->child; c; c = c->next) {
        if (c->tag != RSQ_E_AND) out.push_back(c);
        else collectConjunctionsInner(c, out);
    }
}
void collectAttributes(const Expr* e, std::vector<std::string>& out) {     // planner.h:136-149 (duplicates kept)
    if (e->tag == RSQ_E_ATTRIBUTE) out.push_back(e->symbol);
    for (const Expr* c = e->child; c; c = c->next) collectAttributes(c, out);
}

// ---- plan under construction ----
struct POp {
    int tag = RSQ_OP_UNDEFINED;
    POp* child[2] = {nullptr, nullptr};
    int table = -1;
    ExprVec exprs, exprs2;
    bool singleMatch = false;
};
struct PlanTable { POp* op; const Table* table; int index; };

struct Planner {
    ExprPool& pool;
    const std::vector<Table*>& db;
    std::vector<std::unique_ptr<POp>> ops;
    std::map<std::string, PlanTable> planTables;       // std::map: iterated in name order, like the reference's
    std::vector<POp*> pieces;                          // query.planPieces (see combinePieces)

    Planner(ExprPool& p, const std::vector<Table*>& d) : pool(p), db(d) {}

    POp* op(int tag, POp* c0 = nullptr, POp* c1 = nullptr) {
        ops.emplace_back(new POp());
        POp* o = ops.back().get();
        o->tag = tag; o->child[0] = c0; o->child[1] = c1;
        return o;
    }
    Expr* and_(Expr* l, Expr* r) { l->next = r; Expr* e = pool.make(RSQ_E_AND, BINARY, "and"); e->child = l; return e; }
    Expr* eq_(Expr* l, Expr* r) { l->next = r; Expr* e = pool.make(RSQ_E_EQ, BINARY, ">="); e->child = l; return e; }
    void erasePiece(POp* o) { pieces.erase(std::remove(pieces.begin(), pieces.end(), o), pieces.end()); }

    // planner.h:153-175: the first table (in name order) whose schema has every symbol
    bool matchingTable(const std::vector<std::string>& symbols, std::string& result) {
        for (auto& kv : planTables) {
            bool all = true;
            for (auto& s : symbols) if (kv.second.table->findCol(s) < 0) { all = false; break; }
            if (all) { result = kv.first; return true; }
        }
        return false;
    }

    // planner.h:181-218
    ExprVec pushDownSelection(const ExprVec& where) {
        ExprVec remaining;
        for (Expr* e : where) {
            std::vector<std::string> symbols;
            collectAttributes(e, symbols);
            if (symbols.empty()) break;        // sic: a condition without attributes ends the loop — it and every later condition are dropped
            std::string name;
            if (matchingTable(symbols, name)) {
                PlanTable& pt = planTables[name];
                if (pt.op->tag == RSQ_OP_SELECTION) pt.op->exprs[0] = and_(pt.op->exprs[0], e);
                else {
                    erasePiece(pt.op);
                    POp* s = op(RSQ_OP_SELECTION, pt.op);
                    s->exprs.push_back(e);
                    pt.op = s;
                    pieces.push_back(s);
                }
            } else remaining.push_back(e);
        }
        return remaining;
    }

    static bool isUniqueAttribute(const Expr* e) {        // planner.h:227-249: hard-coded TPC-H primary keys
        if (e->tag != RSQ_E_ATTRIBUTE) return false;
        return e->symbol == "o_orderkey" || e->symbol == "p_partkey" || e->symbol == "s_suppkey" || e->symbol == "n_nationkey" ||
               e->symbol == "r_regionkey";
    }

    typedef std::pair<std::string, std::string> RelationPair;
    typedef std::pair<Expr*, Expr*> ExprPair;
    typedef std::pair<RelationPair, std::vector<ExprPair>> JoinEntry;

    int64_t tupleNum(const std::string& name) { return planTables[name].table->totalRows(); }      // (a shard plans as the table it is a range of)

    // planner.h:268-392
    ExprVec addEqualityHashJoins(const ExprVec& where) {
        ExprVec equalities, remaining;
        for (Expr* e : where) { if (e->tag != RSQ_E_EQ) remaining.push_back(e); else equalities.push_back(e); }
        std::map<RelationPair, std::vector<ExprPair>> joinMap;
        for (Expr* e : equalities) {
            std::vector<std::string> la, ra;
            collectAttributes(e->child, la);
            collectAttributes(e->child->next, ra);
            std::string nameA, nameB;
            const bool ma = matchingTable(la, nameA), mb = matchingTable(ra, nameB);
            if (!ma || !mb) { remaining.push_back(e); continue; }
            Expr* a = e->child; Expr* b = e->child->next;
            if (tupleNum(nameA) >= tupleNum(nameB)) { std::swap(nameA, nameB); std::swap(a, b); }   // smaller relation builds
            joinMap[std::make_pair(nameA, nameB)].push_back(std::make_pair(a, b));
        }
        std::vector<JoinEntry> joinList(joinMap.begin(), joinMap.end());
        auto before = [&](const JoinEntry& x, const JoinEntry& y) {
            int selX = 0, selY = 0;
            if (planTables[x.first.first].op->tag == RSQ_OP_SELECTION) selX++;
            if (planTables[x.first.second].op->tag == RSQ_OP_SELECTION) selX++;
            if (planTables[y.first.first].op->tag == RSQ_OP_SELECTION) selY++;
            if (planTables[y.first.second].op->tag == RSQ_OP_SELECTION) selY++;
            if ((selX > 0 || selY > 0) && selX != selY) return selX > selY;
            return tupleNum(x.first.second) < tupleNum(y.first.second);
        };
        // std::sort of libstdc++ on at most 16 elements is this insertion sort (bits/stl_algo.h __insertion_sort); the
        // comparator is not a strict weak order in general, so the algorithm itself is reproduced
        if (joinList.size() > 16) failUnsupported("more than 16 joins in one query");
        for (size_t i = 1; i < joinList.size(); i++) {
            if (before(joinList[i], joinList[0])) {
                JoinEntry v = joinList[i];
                for (size_t j = i; j > 0; j--) joinList[j] = joinList[j - 1];
                joinList[0] = v;
            } else {
                JoinEntry v = joinList[i];
                size_t j = i;
                while (before(v, joinList[j - 1])) { joinList[j] = joinList[j - 1]; j--; }
                joinList[j] = v;
            }
        }
        for (auto& join : joinList) {
            const std::string& nameA = join.first.first;
            const std::string& nameB = join.first.second;
            bool singleMatch = false;
            ExprVec cond;
            for (auto& pr : join.second) {
                if (isUniqueAttribute(pr.first)) singleMatch = true;
                cond.push_back(eq_(pr.first, pr.second));
                pr.second->next = nullptr;
            }
            POp* a = planTables[nameA].op;
            POp* b = planTables[nameB].op;
            if (a == b) {                          // both sides already in one piece: a further condition of that join
                if (a->tag != RSQ_OP_HASHJOIN) failInvalid("Planner expects operator to be hash join");
                a->exprs.insert(a->exprs.begin(), cond.begin(), cond.end());
                continue;
            }
            POp* hj = op(RSQ_OP_HASHJOIN, a, b);
            hj->exprs = cond;
            hj->singleMatch = singleMatch;
            pieces.push_back(hj);
            erasePiece(a); erasePiece(b);
            for (auto& pt : planTables) if (pt.second.op == a || pt.second.op == b) pt.second.op = hj;
        }
        return remaining;
    }
};

}  // namespace

struct PlanBuilder {
    std::vector<rsq_expr> exprs;
    std::vector<rsq_op> ops;
    std::map<const Expr*, int> exprIndex;
    std::map<const POp*, int> opIndex;

    int addExpr(const Expr* e) {
        auto it = exprIndex.find(e);
        if (it != exprIndex.end()) return it->second;
        std::vector<int> kids;
        const size_t limit = e->structure == UNARY ? 1 : e->structure == BINARY ? 2 : e->structure == LITERAL ? 0 : (size_t)-1;
        for (const Expr* c = e->child; c && kids.size() < limit; c = c->next) kids.push_back(addExpr(c));
        if (kids.size() > RSQ_MAX_CHILDREN) failUnsupported("expression with more than " + std::to_string(RSQ_MAX_CHILDREN) + " operands");
        rsq_expr d;
        memset(&d, 0, sizeof d);
        d.tag = e->tag;
        d.n_children = (int32_t)kids.size();
        for (size_t i = 0; i < kids.size(); i++) d.child[i] = kids[i];
        d.const_category = e->tag == RSQ_E_CONSTANT ? e->category : RSQ_NT;
        std::string sym = e->symbol;
        if (e->tag == RSQ_E_CONSTANT && e->negated) sym = "neg " + sym;
        if (e->tag == RSQ_E_TYPECAST) {          // the target type in the plan text form (include/resql_plan.h)
            const Type& t = e->type;
            switch (t.tag) {
                case RSQ_DECIMAL: sym = "DECIMAL " + std::to_string(t.precision) + " " + std::to_string(t.scale); break;
                case RSQ_CHAR: sym = "CHAR " + std::to_string(t.len); break;
                case RSQ_VARCHAR: sym = "VARCHAR " + std::to_string(t.len); break;
                case RSQ_INT: sym = "INT"; break;
                case RSQ_BIGINT: sym = "BIGINT"; break;
                case RSQ_DATE: sym = "DATE"; break;
                default: failType("typecast to an unknown type");
            }
        }
        if (sym.size() >= RSQ_SYMBOL_MAX) failUnsupported("symbol or constant longer than " + std::to_string(RSQ_SYMBOL_MAX - 1) + " characters");
        memcpy(d.symbol, sym.c_str(), sym.size());
        exprs.push_back(d);
        exprIndex[e] = (int)exprs.size() - 1;
        return (int)exprs.size() - 1;
    }
    int addOp(const POp* o) {
        auto it = opIndex.find(o);
        if (it != opIndex.end()) return it->second;
        int c0 = o->child[0] ? addOp(o->child[0]) : -1;
        int c1 = o->child[1] ? addOp(o->child[1]) : -1;
        rsq_op d;
        memset(&d, 0, sizeof d);
        d.tag = o->tag; d.child[0] = c0; d.child[1] = c1; d.table = o->table;
        if (o->exprs.size() > RSQ_MAX_OP_EXPRS || o->exprs2.size() > RSQ_MAX_OP_EXPRS)
            failUnsupported("operator with more than " + std::to_string(RSQ_MAX_OP_EXPRS) + " expressions");
        d.n_exprs = (int32_t)o->exprs.size();
        for (size_t i = 0; i < o->exprs.size(); i++) d.exprs[i] = addExpr(o->exprs[i]);
        d.n_exprs2 = (int32_t)o->exprs2.size();
        for (size_t i = 0; i < o->exprs2.size(); i++) d.exprs2[i] = addExpr(o->exprs2[i]);
        d.single_match = o->singleMatch ? 1 : 0;
        ops.push_back(d);
        opIndex[o] = (int)ops.size() - 1;
        return (int)ops.size() - 1;
    }
};

// planner.h:409-497 buildQuery
void planSelect(Statement& st, ExprPool& pool, const std::vector<Table*>& db, PlanDesc& out) {
    if (st.kind != Statement::SELECT) failInvalid("not a select statement");
    ExprVec select = listToVector(st.selectExpr);
    ExprVec from = listToVector(st.fromExpr);
    ExprVec where;
    if (st.whereExpr) { if (st.whereExpr->tag != RSQ_E_AND) where.push_back(st.whereExpr); else collectConjunctionsInner(st.whereExpr, where); }
    ExprVec groupby = listToVector(st.groupbyExpr);
    ExprVec orderby = listToVector(st.orderbyExpr);
    bool requestAll = false;
    if (st.selectExpr->tag == RSQ_E_STAR) {
        requestAll = true;
        if (from.empty()) failInvalid("Need from-clause for 'select *'");
    }
    for (Expr* grp : groupby) for (size_t i = 0; i < select.size(); i++) select[i] = matchAndUnify(select[i], grp);
    // matchAndUnify rewrites group->next at every match.  When a group-by expression is a select item of its own AND occurs
    // inside another one (select k, max(k + 1) ... group by k), the node inside the arithmetic ends up linked to the next
    // SELECT ITEM, which contains that arithmetic: a cycle.  The reference then recurses until its stack is gone
    // (filterExpr, planner.h:36-51); here the statement is refused.
    {
        std::set<const Expr*> open, done;
        std::function<bool(const Expr*)> cyclic = [&](const Expr* e) -> bool {
            if (!e || done.count(e)) return false;
            if (!open.insert(e).second) return true;
            for (const Expr* c = e->child; c; c = c->next) {
                if (open.count(c) && !done.count(c)) return true;
                if (cyclic(c)) return true;
            }
            open.erase(e); done.insert(e);
            return false;
        };
        for (Expr* sel : select)
            if (cyclic(sel)) failUnsupported("a group-by expression is both a select item and part of another one: the reference's planner "
                                             "links its expression tree into a cycle here (planner.h:13-33) and crashes");
    }
    ExprVec aggregations;
    for (Expr* s : select) filterAggregations(s, aggregations);

    Planner pl(pool, db);
    for (Expr* f : from) {
        int idx = -1;
        for (size_t i = 0; i < db.size(); i++) if (db[i]->name == f->symbol) { idx = (int)i; break; }
        if (idx < 0) failInvalid("Table " + f->symbol + " does not exist.");
        POp* scan = pl.op(RSQ_OP_SCAN);
        scan->table = idx;
        // a table named twice replaces its first scan in the name map but both scans stay plan pieces (planner.h:436-441)
        pl.planTables[f->symbol] = PlanTable{scan, db[(size_t)idx], idx};
        pl.pieces.push_back(scan);
    }
    where = pl.pushDownSelection(where);
    where = pl.addEqualityHashJoins(where);
    // planner.h:458-469 folds the remaining pieces into nested-loops joins in the order of a std::set of operator
    // ADDRESSES; the engine has no nested-loops join, so more than one piece is refused either way
    POp* plan = nullptr;
    if (pl.pieces.size() > 1) failUnsupported("the plan needs a nested-loops join (tables without an equality join condition)");
    if (pl.pieces.size() == 1) plan = pl.pieces[0];
    if (!plan) failUnsupported("select without a from-clause");
    if (!where.empty()) {
        Expr* c = where[0];
        for (size_t i = 1; i < where.size(); i++) c = pl.and_(c, where[i]);
        POp* s = pl.op(RSQ_OP_SELECTION, plan);
        s->exprs.push_back(c);
        plan = s;
    }
    if (!groupby.empty() || !aggregations.empty()) {
        POp* a = pl.op(RSQ_OP_AGGREGATION, plan);
        a->exprs = aggregations; a->exprs2 = groupby;
        plan = a;
    }
    if (st.selectExpr->tag != RSQ_E_STAR) {
        POp* p = pl.op(RSQ_OP_PROJECTION, plan);
        p->exprs = select;
        plan = p;
    }
    if (!orderby.empty()) { POp* o = pl.op(RSQ_OP_ORDERBY, plan); o->exprs = orderby; plan = o; }
    else plan = pl.op(RSQ_OP_MATERIALIZE, plan);

    PlanBuilder b;
    const int root = b.addOp(plan);
    out.exprs = std::move(b.exprs);
    out.ops = std::move(b.ops);
    memset(&out.desc, 0, sizeof out.desc);
    out.desc.exprs = out.exprs.data(); out.desc.n_exprs = (int32_t)out.exprs.size();
    out.desc.ops = out.ops.data(); out.desc.n_ops = (int32_t)out.ops.size();
    out.desc.root = root;
    out.desc.request_all = requestAll ? 1 : 0;
    out.desc.has_limit = st.useLimit ? 1 : 0;
    out.desc.limit = st.limit;
}

// canonical text of a plan (tests compare it with the reference's planner run on the same parsed statement)
static void dumpVec(const std::vector<Expr*>& ex, const int32_t* idx, int n, std::string& out) {
    out += " [";
    for (int i = 0; i < n; i++) { if (i) out += " "; dumpExpr(ex[(size_t)idx[i]], out); }
    out += "]";
}
static void dumpOp(const rsq_plan_desc& d, const std::vector<Expr*>& ex, const std::vector<Table*>& db, int i, std::string& out) {
    const rsq_op& o = d.ops[i];
    int nChildren = 1;
    switch (o.tag) {
        case RSQ_OP_SCAN: out += "SCAN " + db[(size_t)o.table]->name; nChildren = 0; break;
        case RSQ_OP_SELECTION: out += "SELECTION"; dumpVec(ex, o.exprs, o.n_exprs, out); break;
        case RSQ_OP_PROJECTION: out += "PROJECTION"; dumpVec(ex, o.exprs, o.n_exprs, out); break;
        case RSQ_OP_HASHJOIN: out += std::string("HASHJOIN single=") + (o.single_match ? "1" : "0"); dumpVec(ex, o.exprs, o.n_exprs, out); nChildren = 2; break;
        case RSQ_OP_AGGREGATION: out += "AGGREGATION"; dumpVec(ex, o.exprs, o.n_exprs, out); dumpVec(ex, o.exprs2, o.n_exprs2, out); break;
        case RSQ_OP_MATERIALIZE: out += "MATERIALIZE"; break;
        case RSQ_OP_ORDERBY: out += "ORDERBY"; dumpVec(ex, o.exprs, o.n_exprs, out); break;
        default: out += "UNDEFINED"; nChildren = 0; break;
    }
    out += " {";
    // OrderByOp wraps its child into a MaterializeOp (orderby.h:32-38); the plan description leaves that to the engine
    if (o.tag == RSQ_OP_ORDERBY) { out += "MATERIALIZE {"; dumpOp(d, ex, db, o.child[0], out); out += "}"; }
    else for (int k = 0; k < nChildren; k++) { if (k) out += " "; dumpOp(d, ex, db, o.child[k], out); }
    out += "}";
}
std::string dumpPlan(const rsq_plan_desc& d, const std::vector<Table*>& db) {
    ExprPool pool;
    std::vector<Expr*> ex = pool.build(d);
    std::string out = "limit " + (d.has_limit ? std::to_string((long long)d.limit) : std::string("none")) + " requestall " +
                      (d.request_all ? "1" : "0") + "\n";
    dumpOp(d, ex, db, d.root, out);
    out += "\n";
    return out;
}

}  // namespace sql
}  // namespace rsq
