// sqlfront.h — SQL text -> plan description (sqlfront.cpp): tokens, grammar and planner of the reference's front end
// (reference src/parser/lexer.y, src/parser/parser.y, src/parser/parseSql.h, src/planner.h).
#pragma once

#include <string>
#include <utility>
#include <vector>

#include "engine.h"

namespace rsq {
namespace sql {

struct Token { const char* name; std::string text; };        // name = the token name of lexer.y / parser.y

// lexer.y; `error` is set at the first character no rule matches (the tokens before it are returned)
std::vector<Token> tokenize(const std::string& text, bool& error);

// parseSql.h:36-88 struct Query, the parsed part
struct Statement {
    enum Kind { UNKNOWN, SELECT, CREATE_TABLE, BULK_INSERT };
    int kind = UNKNOWN;
    Expr* selectExpr = nullptr;          // lists are linked through Expr::next, as the grammar actions link them
    Expr* fromExpr = nullptr;
    Expr* whereExpr = nullptr;
    Expr* groupbyExpr = nullptr;
    Expr* orderbyExpr = nullptr;
    bool useLimit = false;
    int64_t limit = 0;
    std::string tableName;
    std::vector<std::pair<std::string, Type>> schema;
    std::string fileName;
    std::string fieldTerminator = ",";
    uint64_t firstRow = 0;
};

// throws Error(RSQ_ERR_INVALID, "Syntax error.") like executeStatement (execute.h:520-523)
void parse(const std::string& text, ExprPool& pool, Statement& out);
std::string dumpStatement(const Statement& st);

// plan description with its storage
struct PlanDesc {
    std::vector<rsq_expr> exprs;
    std::vector<rsq_op> ops;
    rsq_plan_desc desc;
};

// buildQuery (planner.h:409-497) over the tables of `db` (scan operators index into it)
void planSelect(Statement& st, ExprPool& pool, const std::vector<Table*>& db, PlanDesc& out);

std::string dumpPlan(const rsq_plan_desc& d, const std::vector<Table*>& db);

}  // namespace sql
}  // namespace rsq
