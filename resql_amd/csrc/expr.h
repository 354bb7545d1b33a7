// expr.h — typed scalar expression DAG of the engine's host side.
//
// Product code (independent of oracle/): the engine's own implementation of the reference's
// typing rules, which decide the bits of every result (SURVEY.md §8 row a-T):
//   constants        reference src/expressions.h:369-515
//   type derivation  reference src/expressions.h:742-951, 1204-1392
//   expression names reference src/expressions.h:954-966
#pragma once

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include <map>

#include "resql_hip.h"

namespace rsq {

// Error carrying an rsq_status; caught at the C ABI (the reference throws ResqlError,
// src/util/ResqlError.h, or calls exit() in query_error, src/qlib/error.h:29-68).
struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string& m) : std::runtime_error(m), status(st) {}
};
[[noreturn]] void failType(const std::string& m);         // RSQ_ERR_TYPE
[[noreturn]] void failUnsupported(const std::string& m);  // RSQ_ERR_UNSUPPORTED
[[noreturn]] void failInvalid(const std::string& m);      // RSQ_ERR_INVALID
[[noreturn]] void failRuntime(const std::string& m);      // RSQ_ERR_RUNTIME

struct Type {
    int tag = RSQ_NT;
    int precision = 0, scale = 0, len = 0;
    Type() = default;
    Type(int t) : tag(t) {}
    static Type decimal(int p, int s) { Type t(RSQ_DECIMAL); t.precision = p; t.scale = s; return t; }
    static Type fromC(const rsq_type& c) { Type t(c.tag); t.precision = c.precision; t.scale = c.scale; t.len = c.len; return t; }
    rsq_type toC() const { return rsq_type{tag, precision, scale, len}; }
    bool isString() const { return tag == RSQ_VARCHAR || (tag == RSQ_CHAR && len > 1); }
    bool isInt64() const { return tag == RSQ_BIGINT || tag == RSQ_DECIMAL; }
};
bool equalTypes(const Type& a, const Type& b);           // types.h:153-173
std::string serializeType(const Type& t);                // types.h:121-150
int sizeInTuple(const Type& t, bool stringsByVal);       // types.h:213-261
int columnWidth(const Type& t);                          // bytes per row of a device column

enum Structure { LITERAL, UNARY, BINARY, TERNARY, OTHER };

struct Expr {
    int tag = RSQ_E_UNDEFINED;
    int structure = LITERAL;
    std::string symbol;
    Expr* next = nullptr;      // sibling (the reference's linked representation; typecast insertion relies on it)
    Expr* child = nullptr;
    Type type;
    int64_t ival = 0;          // numeric / date / bool / char(1) constant value
    size_t id = 0;             // expression id, assigned when first evaluated (expressions.h:1354-1358)
    int category = RSQ_NT;     // constants: the type category the literal was parsed as
    bool negated = false;      // constants from SQL text: `- literal` negates the VALUE, the type still follows from the
                               // unsigned text (parser.y:149-151); travels in the plan description as "neg <text>"
    bool explicitCast = false; // TYPECAST written in the query (expr :: type)
    std::vector<Expr*> children() const { std::vector<Expr*> v; for (Expr* c = child; c; c = c->next) v.push_back(c); return v; }
};

extern const char* const exprTagNames[];

// owns every node of one query
struct ExprPool {
    std::vector<std::unique_ptr<Expr>> nodes;
    std::map<std::string, Type> identTypes;   // planner.h:395-404 mapIdentifierTypes + AS aliases
    int exprIdGen = 1;                        // RelationalContext.h:16
    Expr* make(int tag, int structure, const std::string& symbol);
    Expr* unary(int tag, const std::string& symbol, Expr* child);
    Expr* constant(const std::string& symbol, int category);      // ExprGen::constant (expressions.h:520-524)
    // build from the C description; returns node per index
    std::vector<Expr*> build(const rsq_plan_desc& plan);
    void derive(Expr* e);                     // deriveExpressionTypes, expressions.h:1367-1392
    void addId(Expr* e) { if (e->id == 0) e->id = (size_t)(++exprIdGen); }
};

std::string expressionName(const Expr* e);               // expressions.h:954-966
std::string serializeExpr(const Expr* e);                // expressions.h:177-204
void requiredAttributes(const Expr* e, std::vector<std::string>& out);   // expressions.h:1416-1431
// structural key of a typed expression (for common-subexpression detection between aggregates)
std::string structuralKey(const Expr* e);

// value formatting (values.h:30-127)
union Val { int64_t i; const char* s; };
std::string serializeSqlValue(Val v, const Type& t);

}  // namespace rsq
