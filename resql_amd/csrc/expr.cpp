// expr.cpp — constants, type derivation and naming of scalar expressions (see expr.h).
#include "expr.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

namespace rsq {

void failType(const std::string& m) { throw Error(RSQ_ERR_TYPE, m); }
void failUnsupported(const std::string& m) { throw Error(RSQ_ERR_UNSUPPORTED, m); }
void failInvalid(const std::string& m) { throw Error(RSQ_ERR_INVALID, m); }
void failRuntime(const std::string& m) { throw Error(RSQ_ERR_RUNTIME, m); }

static const char* const typeTagNames[] = {"VARCHAR", "CHAR", "BOOL", "INT", "BIGINT", "DECIMAL", "FLOAT", "DATE", ""};

const char* const exprTagNames[] = {
    "ADD", "SUB", "MUL", "DIV", "AND", "OR", "LT", "LE", "GT", "GE", "EQ", "NEQ", "LIKE",
    "SUM", "COUNT", "AVG", "MIN", "MAX", "ASC", "DESC", "CASE", "WHENTHEN",
    "ATTRIBUTE", "TYPECAST", "CONSTANT", "AS", "TYPE", "TABLE", "STAR", "UNDEFINED"};

bool equalTypes(const Type& a, const Type& b) {
    if (a.tag != b.tag) return false;
    if (a.tag == RSQ_DECIMAL) return a.precision == b.precision && a.scale == b.scale;
    if (a.tag == RSQ_CHAR || a.tag == RSQ_VARCHAR) return a.len == b.len;
    return true;
}

std::string serializeType(const Type& t) {
    std::ostringstream ss;
    ss << typeTagNames[t.tag];
    if (t.tag == RSQ_DECIMAL) ss << "(" << t.precision << "," << t.scale << ")";
    else if (t.tag == RSQ_CHAR || t.tag == RSQ_VARCHAR) ss << "(" << t.len << ")";
    return ss.str();
}

int sizeInTuple(const Type& t, bool stringsByVal) {
    switch (t.tag) {
        case RSQ_BOOL: return 1;
        case RSQ_DATE: case RSQ_INT: return 4;
        case RSQ_DECIMAL: case RSQ_BIGINT: case RSQ_FLOAT: return 8;
        case RSQ_CHAR: if (t.len == 1) return 2; return stringsByVal ? t.len + 1 : 8;
        case RSQ_VARCHAR: return stringsByVal ? t.len + 1 : 8;
        default: failType("getSizeInTuple(..) for undefined type.");
    }
}

int columnWidth(const Type& t) {
    switch (t.tag) {
        case RSQ_BOOL: return 1;
        case RSQ_DATE: case RSQ_INT: return 4;
        case RSQ_DECIMAL: case RSQ_BIGINT: return 8;
        case RSQ_CHAR: case RSQ_VARCHAR: return t.len;
        default: failInvalid("unsupported column type");
    }
}

static int structureOf(int tag) {
    switch (tag) {
        case RSQ_E_ATTRIBUTE: case RSQ_E_CONSTANT: case RSQ_E_STAR: case RSQ_E_TYPE: case RSQ_E_TABLE:
        case RSQ_E_UNDEFINED: return LITERAL;
        case RSQ_E_SUM: case RSQ_E_COUNT: case RSQ_E_AVG: case RSQ_E_MIN: case RSQ_E_MAX:
        case RSQ_E_ASC: case RSQ_E_DESC: case RSQ_E_TYPECAST: case RSQ_E_AS: return UNARY;
        case RSQ_E_CASE: return OTHER;
        default: return BINARY;
    }
}

static void parseConstant(Expr* e, int category);

Expr* ExprPool::make(int tag, int structure, const std::string& symbol) {
    nodes.emplace_back(new Expr());
    Expr* e = nodes.back().get();
    e->tag = tag; e->structure = structure; e->symbol = symbol;
    return e;
}

Expr* ExprPool::constant(const std::string& symbol, int category) {
    Expr* e = make(RSQ_E_CONSTANT, LITERAL, symbol);
    parseConstant(e, category);
    return e;
}

Expr* ExprPool::unary(int tag, const std::string& symbol, Expr* child) {
    Expr* e = make(tag, UNARY, symbol);
    e->child = child;
    return e;
}

// ---- constants: the value and exact type follow from the literal text --------------------------
static long long stoll_strict(const std::string& s) {
    char* end = nullptr;
    long long v = strtoll(s.c_str(), &end, 10);
    if (end == s.c_str()) failType("invalid numeric constant '" + s + "'");
    return v;
}

static void parseConstant(Expr* e, int category) {
    e->category = category;
    if (e->symbol.compare(0, 4, "neg ") == 0 && (category == RSQ_DECIMAL || category == RSQ_BIGINT)) {
        // a negated literal of the SQL grammar (value ::= MINUS_TK constant, parser.y:149-151): typed from the unsigned text
        e->symbol = e->symbol.substr(4);
        parseConstant(e, category);
        e->ival = (int64_t)(0 - (uint64_t)e->ival);
        e->negated = true;
        return;
    }
    const std::string& sym = e->symbol;
    switch (category) {
        case RSQ_DECIMAL: {
            // digits with the point removed; scale = digits after the point; precision = remaining length
            std::string digits; int scale = 0;
            size_t pos = sym.find('.');
            if (pos != std::string::npos) scale = (int)(sym.length() - (pos + 1));
            for (size_t i = 0; i < sym.size(); i++) if (i != pos) digits.push_back(sym[i]);
            e->ival = stoll_strict(digits);
            e->type = Type::decimal((int)(uint8_t)digits.length(), (int)(uint8_t)scale);
            break;
        }
        case RSQ_DATE: {
            int y, m, d; bool ok = false;
            if (sscanf(sym.c_str(), "%4d-%2d-%2d", &y, &m, &d) == 3) ok = true;
            if (sscanf(sym.c_str(), "%4d/%2d/%2d", &y, &m, &d) == 3) ok = true;
            if (!ok) failType("Unsupported string type or unsupported date format (formats: \"yyyy/mm/dd\", \"mm/dd/yyyy\")");
            e->ival = (int64_t)(uint32_t)(y * 10000 + m * 100 + d);
            e->type = Type(RSQ_DATE);
            break;
        }
        case RSQ_INT: {
            long long v = stoll_strict(sym);
            if (v > INT_MAX || v < INT_MIN) failType("integer constant out of range");
            e->ival = (int32_t)v; e->type = Type(RSQ_INT);
            break;
        }
        case RSQ_BIGINT:   // the reference parses BIGINT literals through an int32_t
            e->ival = (int32_t)stoll_strict(sym); e->type = Type(RSQ_BIGINT);
            break;
        case RSQ_BOOL:
            if (sym == "true") e->ival = 1; else if (sym == "false") e->ival = 0;
            else failType("Couldnt parse BOOL constant.");
            e->type = Type(RSQ_BOOL);
            break;
        case RSQ_CHAR:
            e->type = Type(RSQ_CHAR); e->type.len = (int)sym.length();
            e->ival = sym.empty() ? 0 : (uint8_t)sym[0];
            break;
        case RSQ_VARCHAR:
            e->type = Type(RSQ_VARCHAR); e->type.len = (int)sym.length();
            break;
        default: failType("parseConstant(..) not implemented for type.");
    }
}

std::vector<Expr*> ExprPool::build(const rsq_plan_desc& p) {
    std::vector<Expr*> v((size_t)p.n_exprs);
    for (int i = 0; i < p.n_exprs; i++) {
        const rsq_expr& d = p.exprs[i];
        if (d.tag < 0 || d.tag > RSQ_E_UNDEFINED) failInvalid("bad expression tag");
        std::string sym(d.symbol, strnlen(d.symbol, RSQ_SYMBOL_MAX));
        v[i] = make(d.tag, structureOf(d.tag), sym);
        if (d.tag == RSQ_E_CONSTANT) parseConstant(v[i], d.const_category);
        if (d.tag == RSQ_E_TYPECAST) {
            // an explicit `expr :: type` of the query (ExprGen::typecast, expressions.h:656-660): the symbol is the target
            // type in the plan text form ("BIGINT", "DECIMAL 12 2", "CHAR 3", ...); type derivation leaves it alone
            char name[16] = {0}; int a = 0, b = 0;
            const int n = sscanf(sym.c_str(), "%15s %d %d", name, &a, &b);
            Type t;
            const std::string nm = name;
            if (nm == "INT" && n == 1) t = Type(RSQ_INT);
            else if (nm == "BIGINT" && n == 1) t = Type(RSQ_BIGINT);
            else if (nm == "DATE" && n == 1) t = Type(RSQ_DATE);
            else if (nm == "BOOL" && n == 1) t = Type(RSQ_BOOL);
            else if (nm == "DECIMAL" && n == 3) t = Type::decimal(a, b);
            else if (nm == "CHAR" && n == 2) { t = Type(RSQ_CHAR); t.len = a; }
            else if (nm == "VARCHAR" && n == 2) { t = Type(RSQ_VARCHAR); t.len = a; }
            else failInvalid("TYPECAST needs its target type as symbol, got '" + sym + "'");
            v[i]->type = t; v[i]->explicitCast = true; v[i]->symbol = "typecast";
        }
    }
    for (int i = 0; i < p.n_exprs; i++) {
        const rsq_expr& d = p.exprs[i];
        if (d.n_children < 0 || d.n_children > RSQ_MAX_CHILDREN) failInvalid("bad child count");
        Expr* prev = nullptr;
        for (int k = 0; k < d.n_children; k++) {
            int ci = d.child[k];
            if (ci < 0 || ci >= p.n_exprs || ci == i) failInvalid("bad child index in expression " + std::to_string(i));
            Expr* ch = v[ci];
            // children are chained through their `next` pointers (expressions.h:83-96): one node twice below a parent would chain
            // it to itself and every walk over the children would never end (the reference's ExprGen cannot build such a tree
            // either: its binary constructors link two distinct nodes)
            for (int j = 0; j < k; j++) if (d.child[j] == ci) failInvalid("expression " + std::to_string(i) + " has the same child twice: copy the node (ExprGen::copy)");
            if (k == 0) v[i]->child = ch;
            else {
                if (prev->next && prev->next != ch) failInvalid("expression node shared as child with different right siblings");
                prev->next = ch;
            }
            prev = ch;
        }
        if (v[i]->structure == UNARY && d.n_children != 1) failInvalid("unary expression needs one child");
        if (v[i]->structure == BINARY && d.n_children != 2) failInvalid("binary expression needs two children");
    }
    // nodes shared between parents may still close a ring of siblings (a below x as [a, b], below y as [b, a]): no chain may be
    // longer than the plan has expressions
    for (int i = 0; i < p.n_exprs; i++) {
        int steps = 0;
        for (Expr* e = v[i]; e; e = e->next) if (++steps > p.n_exprs) failInvalid("expression nodes shared between parents form a ring of siblings");
    }
    return v;
}

std::string expressionName(const Expr* e) {
    if (e->tag == RSQ_E_ATTRIBUTE || e->tag == RSQ_E_AS) return e->symbol;
    return "expr" + std::to_string(e->id);
}

// ---- value text (what the reference prints for a value of each type, values.h:30-127), one small writer per type ----------------
namespace {

// CHAR(n): the value's characters, then blanks up to n.  n > 1 is a NUL-terminated string, n <= 1 a single byte (a NUL prints nothing)
void writePaddedChars(std::string& out, const Val& v, int n) {
    size_t have;
    if (n > 1) { have = strlen(v.s); out.append(v.s, have); }
    else { const char c = (char)v.i; have = c ? 1 : 0; if (c) out.push_back(c); }
    if (have < (size_t)std::max(n, 0)) out.append((size_t)n - have, ' ');
}

// DATE: the integer yyyymmdd as y/mm/dd
void writeDate(std::string& out, uint32_t yyyymmdd) {
    const unsigned day = yyyymmdd % 100, month = yyyymmdd / 100 % 100;
    out += std::to_string(yyyymmdd / 10000);
    const char tail[6] = {'/', (char)('0' + month / 10), (char)('0' + month % 10), '/', (char)('0' + day / 10), (char)('0' + day % 10)};
    out.append(tail, 6);
}

// DECIMAL(p, s): sign, then the magnitude's digits with the point s places from the right (zeros in front where the digits run out)
void writeDecimal(std::string& out, int64_t raw, int scale) {
    if (raw < 0) out.push_back('-');
    // (the magnitude is formed the reference's way, as a signed negation: the smallest value keeps its sign in the digit string)
    std::string digits = std::to_string((long long)(raw < 0 ? (int64_t)(0 - (uint64_t)raw) : raw));
    const size_t places = (size_t)std::max(scale, 0);
    if (digits.size() <= places) digits.insert((size_t)0, places + 1 - digits.size(), '0');
    if (places > 0) digits.insert(digits.size() - places, 1, '.');
    out += digits;
}

}  // namespace

std::string serializeSqlValue(Val v, const Type& t) {
    std::string out;
    if (t.tag == RSQ_VARCHAR) out = v.s;
    else if (t.tag == RSQ_CHAR) writePaddedChars(out, v, t.len);
    else if (t.tag == RSQ_DATE) writeDate(out, (uint32_t)v.i);
    else if (t.tag == RSQ_DECIMAL) writeDecimal(out, v.i, t.scale);
    else if (t.tag == RSQ_INT) out = std::to_string((int32_t)v.i);
    else if (t.tag == RSQ_BIGINT) out = std::to_string((long long)v.i);
    else if (t.tag == RSQ_BOOL) out = (unsigned char)v.i ? "true" : "false";
    else failType("serializeSqlValue(..) not implemented for datatype.");
    return out;
}

std::string serializeExpr(const Expr* e) {
    std::string s = "{";
    s += exprTagNames[e->tag]; s += ","; s += serializeType(e->type);
    if (e->tag == RSQ_E_CONSTANT) {
        Val v; v.i = e->ival;
        if (e->type.isString()) v.s = e->symbol.c_str();
        s += "," + serializeSqlValue(v, e->type);
    }
    for (const Expr* c = e->child; c; c = c->next) s += "," + serializeExpr(c);
    return s + "}";
}

std::string structuralKey(const Expr* e) {
    std::string s = "(";
    s += exprTagNames[e->tag]; s += ":"; s += serializeType(e->type);
    if (e->tag == RSQ_E_CONSTANT || e->tag == RSQ_E_ATTRIBUTE) s += ":" + e->symbol;
    for (const Expr* c = e->child; c; c = c->next) s += structuralKey(c);
    return s + ")";
}

void requiredAttributes(const Expr* e, std::vector<std::string>& out) {
    if (!e) return;
    if (e->tag == RSQ_E_ATTRIBUTE && std::find(out.begin(), out.end(), e->symbol) == out.end()) out.push_back(e->symbol);
    for (const Expr* c = e->child; c; c = c->next) requiredAttributes(c, out);
}

// ---- type derivation -------------------------------------------------------------------------
namespace {

// `insert` takes `child`'s place among `parent`'s children and becomes its only parent: the link that points at `child` - the
// parent's child pointer or a sibling's next pointer - is redirected
void insertBetween(Expr* parent, Expr* child, Expr* insert) {
    Expr** link = &parent->child;
    while (*link != nullptr && *link != child) link = &(*link)->next;
    if (*link == nullptr) failType("Child in insertUnaryBetweenParentAndChild(..) not found.");
    *link = insert;
    insert->next = child->next;
    insert->child = child;
    child->next = nullptr;
}

struct Deriver {
    ExprPool& pool;

    Expr* typecastNode(const Type& t) { Expr* e = pool.make(RSQ_E_TYPECAST, UNARY, "typecast"); e->type = t; return e; }

    // A cast towards a string type is never materialised; a cast towards DECIMAL starts as
    // DECIMAL(19,0) and is re-scaled by sameScale() below.
    void insertTypecast(Expr* e, Expr* child, Type to) {
        if (to.tag == RSQ_CHAR || to.tag == RSQ_VARCHAR) return;
        if (to.tag == RSQ_DECIMAL) { to.scale = 0; to.precision = 19; }
        insertBetween(e, child, typecastNode(to));
    }

    // type category precedence = enum order (higher wins)
    void applyPrecedence(Expr* e, Expr* left, Expr* right) {
        if (left->type.tag == right->type.tag) return;
        if (left->type.tag > right->type.tag) insertTypecast(e, right, left->type);
        else insertTypecast(e, left, right->type);
    }

    static Type scaleToOther(Type spec, const Type& other) {
        int diff = other.scale - spec.scale;
        spec.scale = (uint8_t)(spec.scale + diff);
        spec.precision = (uint8_t)(spec.precision + diff);
        if (spec.precision > 19) spec.precision = 19;
        return spec;
    }

    void sameScale(Expr* e, Expr* left, Expr* right) {
        const Type ls = left->type, rs = right->type;
        if (ls.scale < rs.scale) {
            Type t = scaleToOther(ls, rs);
            if (left->tag == RSQ_E_TYPECAST) left->type = t; else insertBetween(e, left, typecastNode(t));
        } else if (ls.scale > rs.scale) {
            Type t = scaleToOther(rs, ls);
            if (right->tag == RSQ_E_TYPECAST) right->type = t; else insertBetween(e, right, typecastNode(t));
        }
    }

    void configurableInputs(Expr* e) {
        Expr* left = e->child; Expr* right = e->child->next;
        if (left->type.tag != RSQ_DECIMAL) return;
        switch (e->tag) {
            case RSQ_E_LT: case RSQ_E_GT: case RSQ_E_LE: case RSQ_E_GE: case RSQ_E_EQ: case RSQ_E_NEQ:
            case RSQ_E_ADD: case RSQ_E_SUB: sameScale(e, left, right); break;
            case RSQ_E_MUL: break;
            case RSQ_E_DIV: failType("Decimal division not yet implemented");
            default: failType("Invalid expression type or type not implemented in typecastDecimalInputs(..)");
        }
    }

    static void arithmeticResult(Expr* e) {
        if (e->type.tag != RSQ_DECIMAL) return;
        const Type l = e->child->type, r = e->child->next->type;
        int p = 0, s = 0;
        switch (e->tag) {
            case RSQ_E_ADD: case RSQ_E_SUB: p = (uint8_t)(std::max(l.precision, r.precision) + 1); s = l.scale; break;
            case RSQ_E_MUL: p = (uint8_t)(l.precision + r.precision); s = (uint8_t)(l.scale + r.scale); break;
            default: failType("Decimal division not yet implemented");
        }
        if (p > 19) p = 19;
        e->type.precision = p; e->type.scale = s;
    }

    static void aggregationResult(Expr* e) {
        if (e->child->type.tag != RSQ_DECIMAL) return;
        const Type cs = e->child->type;
        if (e->tag == RSQ_E_SUM) { e->type.scale = cs.scale; e->type.precision = 19; }
        if (e->tag == RSQ_E_AVG) { e->type.scale = (uint8_t)(cs.scale + 2); e->type.precision = std::min(cs.precision + 2, 19); }
    }

    static void needNumeric(const Expr* op, const Expr* e) {
        int t = e->type.tag;
        if (t != RSQ_DECIMAL && t != RSQ_BIGINT && t != RSQ_INT && t != RSQ_FLOAT)
            failType(std::string("Incompatible types: ") + exprTagNames[op->tag] + " expression requires a numeric operand at " + serializeExpr(e));
    }
    static void needOrdered(const Expr* op, const Expr* e) {
        int t = e->type.tag;
        if (t != RSQ_DECIMAL && t != RSQ_BIGINT && t != RSQ_INT && t != RSQ_FLOAT && t != RSQ_DATE)
            failType(std::string("Incompatible types: ") + exprTagNames[op->tag] + " expression requires an ordered operand type at " + serializeExpr(e));
    }
    static void needBool(const Expr* op, const Expr* e) {
        if (e->type.tag != RSQ_BOOL)
            failType(std::string("Incompatible types: ") + exprTagNames[op->tag] + " expression requires bool operand at " + serializeExpr(e));
    }
    static void needString(const Expr* op, const Expr* e) {
        if (e->type.tag != RSQ_CHAR && e->type.tag != RSQ_VARCHAR)
            failType(std::string("Incompatible types: ") + exprTagNames[op->tag] + " expression requires a char or varchar operand at " + serializeExpr(e));
    }

    static Type superType(const Type& a, const Type& b) {
        if (a.tag == b.tag) {
            if (a.tag == RSQ_DECIMAL) return Type::decimal(std::max(a.precision, b.precision), std::max(a.scale, b.scale));
            if (a.tag == RSQ_VARCHAR || a.tag == RSQ_CHAR) { Type t = a; t.len = std::max(a.len, b.len); return t; }
            return a;
        }
        if ((a.tag == RSQ_BIGINT || a.tag == RSQ_INT) && b.tag == RSQ_DECIMAL) return b;
        if ((b.tag == RSQ_BIGINT || b.tag == RSQ_INT) && a.tag == RSQ_DECIMAL) return b;   // sic: the reference returns b here too
        failType("Incompatible or unimplemented type combination in getCommonSuperType(..):" + serializeType(a) + " and " + serializeType(b));
    }

    void deriveCase(Expr* e) {
        Expr* child = e->child;
        derive(child);
        Type thenType = child->type;
        child = child->next;
        while (child && child->tag == RSQ_E_WHENTHEN) {
            Expr* when = child->child; Expr* then = when->next;
            derive(when); derive(then);
            thenType = superType(thenType, then->type);
            child = child->next;
        }
        if (child) { derive(child); thenType = superType(thenType, child->type); }
        child = e->child;
        while (child && child->tag == RSQ_E_WHENTHEN) {
            Expr* when = child->child; Expr* then = when->next;
            if (!equalTypes(then->type, thenType)) insertTypecast(child, then, thenType);
            derive(child);
            child = child->next;
        }
        if (child && !equalTypes(child->type, thenType)) insertTypecast(e, child, thenType);
        e->type = thenType;
    }

    void derive(Expr* e) {
        switch (e->structure) {
            case LITERAL:
                if (e->type.tag != RSQ_NT) {
                    if (e->tag == RSQ_E_ATTRIBUTE) pool.identTypes[e->symbol] = e->type;
                    return;
                }
                if (e->tag == RSQ_E_ATTRIBUTE) {
                    auto it = pool.identTypes.find(e->symbol);
                    if (it == pool.identTypes.end()) failType("Attribute " + e->symbol + " not found.");
                    e->type = it->second;
                } else if (e->tag == RSQ_E_STAR) e->type = Type(RSQ_BIGINT);
                else if (e->tag != RSQ_E_CONSTANT) failType("deriveExpressionTypesLiteral(..) not implemented for " + serializeExpr(e));
                break;
            case UNARY: {
                Expr* child = e->child;
                // (a node that lost an operand: the SQL planner shares ONE node between identical expressions - as the reference's does, which
                // dies on such statements, e.g. two equal aggregates over the group column - and linking it below a second parent cut its chain)
                if (!child) failType(std::string("malformed expression: ") + exprTagNames[e->tag] + " without its operand (one expression node used in two places)");
                derive(child);
                switch (e->tag) {
                    case RSQ_E_TYPECAST: break;
                    case RSQ_E_AS: pool.identTypes[e->symbol] = child->type; e->type = child->type; break;
                    case RSQ_E_COUNT: e->type = Type(RSQ_BIGINT); break;
                    case RSQ_E_SUM: needNumeric(e, child); e->type = child->type; aggregationResult(e); break;
                    case RSQ_E_AVG: needNumeric(e, child); e->type = Type::decimal(19, 2); aggregationResult(e); break;
                    case RSQ_E_MAX: case RSQ_E_MIN: needOrdered(e, child); e->type = child->type; break;
                    case RSQ_E_DESC: case RSQ_E_ASC: e->type = child->type; break;
                    default: failType("deriveExpressionTypesUnary(..) not implemented for " + serializeExpr(e));
                }
                break;
            }
            case BINARY: {
                if (!e->child || !e->child->next) failType(std::string("malformed expression: ") + exprTagNames[e->tag] + " without both operands (one expression node used in two places)");
                Expr* left = e->child; Expr* right = e->child->next;
                derive(left); derive(right);
                switch (e->tag) {
                    case RSQ_E_ADD: case RSQ_E_SUB: case RSQ_E_MUL: case RSQ_E_DIV:
                        needNumeric(e, left); needNumeric(e, right);
                        applyPrecedence(e, left, right);
                        e->type = e->child->type;
                        configurableInputs(e);
                        arithmeticResult(e);
                        break;
                    case RSQ_E_LT: case RSQ_E_LE: case RSQ_E_GT: case RSQ_E_GE:
                        needOrdered(e, left); needOrdered(e, right);
                        [[fallthrough]];
                    case RSQ_E_EQ: case RSQ_E_NEQ:
                        applyPrecedence(e, left, right);
                        configurableInputs(e);
                        e->type = Type(RSQ_BOOL);
                        break;
                    case RSQ_E_OR: case RSQ_E_AND:
                        needBool(e, left); needBool(e, right); e->type = Type(RSQ_BOOL); break;
                    case RSQ_E_LIKE:
                        needString(e, left); needString(e, right); e->type = Type(RSQ_BOOL); break;
                    case RSQ_E_WHENTHEN:
                        needBool(e, left); e->type = right->type; break;
                    default: failType("deriveExpressionTypesBinary(..) not implemented for " + serializeExpr(e));
                }
                break;
            }
            case OTHER:
                if (e->tag == RSQ_E_CASE) deriveCase(e);
                else failType("deriveExpressionTypesOther(..) not implemented for " + serializeExpr(e));
                break;
            default: failType("deriveExpressionTypes(..)");
        }
    }
};

}  // namespace

void ExprPool::derive(Expr* e) { Deriver d{*this}; d.derive(e); }

}  // namespace rsq
