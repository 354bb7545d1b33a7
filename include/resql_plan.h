/*
 * resql_plan.h — plain-C description of a ReSQL query plan (operator tree + scalar
 * expression DAG) and of columnar input tables.
 *
 * This is the data contract of the drop-in boundary: what ReSQL's host side (parser,
 * planner, `RelOperator` tree — reference src/operators/RelOperator.h:160-189) hands to the
 * engine INSTEAD of calling `produceFlounder()` on its operators
 * (reference src/execute.h:228-229).  It is plain old data: no pointers into C++ objects,
 * no torch types.  Both the HIP engine (include/resql_hip.h) and the CPU oracle
 * (oracle/resql_oracle.h) consume exactly this struct, so a test can feed one plan to both.
 *
 * Tag values mirror the reference enums one-to-one so a binding can cast:
 *   rsq_type_tag  == SqlType::Tag      (reference src/types.h:66-76; the order IS the type
 *                                       precedence used by applyPrecedence, expressions.h:781-796)
 *   rsq_expr_tag  == Expr::Tag         (reference src/expressions.h:25-63)
 *   rsq_op_tag    == RelOperator::OperatorTag (reference src/operators/RelOperator.h:25-35)
 */
#ifndef RESQL_PLAN_H
#define RESQL_PLAN_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- SQL types (reference src/types.h:60-105) ------------------------------------------ */
enum rsq_type_tag {
    RSQ_VARCHAR = 0, /* lowest precedence */
    RSQ_CHAR    = 1,
    RSQ_BOOL    = 2,
    RSQ_INT     = 3,
    RSQ_BIGINT  = 4,
    RSQ_DECIMAL = 5,
    RSQ_FLOAT   = 6, /* declared by the reference, no arithmetic implemented there either */
    RSQ_DATE    = 7, /* uint32 yyyymmdd */
    RSQ_NT      = 8  /* "no type" (not derived yet) */
};

typedef struct rsq_type {
    int32_t tag;        /* rsq_type_tag */
    int32_t precision;  /* DECIMAL */
    int32_t scale;      /* DECIMAL */
    int32_t len;        /* CHAR(n) / VARCHAR(n) */
} rsq_type;

/* ---- scalar expressions (reference src/expressions.h:23-90) ---------------------------- */
enum rsq_expr_tag {
    RSQ_E_ADD = 0, RSQ_E_SUB, RSQ_E_MUL, RSQ_E_DIV,
    RSQ_E_AND, RSQ_E_OR,
    RSQ_E_LT, RSQ_E_LE, RSQ_E_GT, RSQ_E_GE, RSQ_E_EQ, RSQ_E_NEQ, RSQ_E_LIKE,
    RSQ_E_SUM, RSQ_E_COUNT, RSQ_E_AVG, RSQ_E_MIN, RSQ_E_MAX,
    RSQ_E_ASC, RSQ_E_DESC,
    RSQ_E_CASE, RSQ_E_WHENTHEN,
    RSQ_E_ATTRIBUTE, RSQ_E_TYPECAST, RSQ_E_CONSTANT, RSQ_E_AS, RSQ_E_TYPE, RSQ_E_TABLE,
    RSQ_E_STAR, RSQ_E_UNDEFINED
};

#define RSQ_SYMBOL_MAX 64
#define RSQ_MAX_CHILDREN 8

/*
 * One expression node.  Nodes are referenced by index; a node referenced from two places
 * is ONE value (the reference shares `Expr*` between the select list, the group-by list
 * and the aggregate list and re-uses already computed values by expression name,
 * ExpressionsJitFlounder.h:1088-1094).
 *
 *  ATTRIBUTE : symbol = column name
 *  CONSTANT  : symbol = literal text exactly as the SQL text has it, const_category = the
 *              type category the parser assigns (ExprGen::constant(symbol, category),
 *              expressions.h:520-524); value and exact type are derived from the text
 *              (parseConstant, expressions.h:369-515)
 *              A literal the SQL grammar negates (`- 5`, `- 0.25`; parser.y:149-151) is typed from its unsigned
 *              text and only its value is negated; it travels as symbol = "neg " + unsigned text.
 *  TYPECAST  : an explicit `expr :: type` of the query (ExprGen::typecast, expressions.h:656-660): child[0], symbol =
 *              the target type in text form ("INT", "BIGINT", "DATE", "DECIMAL p s", "CHAR n", "VARCHAR n").  Casts
 *              that type derivation inserts are never part of a description.
 *  AS        : symbol = alias, child[0]
 *  unary / binary / CASE: child[0..n_children)
 */
typedef struct rsq_expr {
    int32_t tag;                         /* rsq_expr_tag */
    int32_t n_children;
    int32_t child[RSQ_MAX_CHILDREN];     /* expression indices */
    int32_t const_category;              /* rsq_type_tag, CONSTANT only */
    char    symbol[RSQ_SYMBOL_MAX];
} rsq_expr;

/* ---- relational operators (reference src/operators/) -------------------------------- */
enum rsq_op_tag {
    RSQ_OP_UNDEFINED = 0,
    RSQ_OP_SCAN,            /* operators/scan.h:178-273      */
    RSQ_OP_PROJECTION,      /* operators/projection.h:5-74   */
    RSQ_OP_SELECTION,       /* operators/selection.h:6-72    */
    RSQ_OP_MATERIALIZE,     /* operators/materialize.h:9-262 */
    RSQ_OP_NESTEDLOOPSJOIN, /* out of scope      */
    RSQ_OP_HASHJOIN,        /* operators/hashjoin.h:23-256   */
    RSQ_OP_AGGREGATION,     /* operators/aggregation.h:17-345*/
    RSQ_OP_ORDERBY          /* operators/orderby.h:14-141    */
};

#define RSQ_MAX_OP_EXPRS 32

typedef struct rsq_op {
    int32_t tag;                     /* rsq_op_tag */
    int32_t child[2];                /* operator indices; -1 if unused.  HASHJOIN: [0]=build (left), [1]=probe (right) */
    int32_t table;                   /* SCAN: index into the table array handed to compile/execute */
    int32_t n_exprs;                 /* SELECTION: 1 condition; PROJECTION: select list; HASHJOIN: EQ list;
                                        AGGREGATION: aggregate list; ORDERBY: order list (ATTRIBUTE, ASC(..) or DESC(..)) */
    int32_t exprs[RSQ_MAX_OP_EXPRS];
    int32_t n_exprs2;                /* AGGREGATION: group-by list */
    int32_t exprs2[RSQ_MAX_OP_EXPRS];
    int32_t single_match;            /* HASHJOIN: HashJoinOp::_singleMatch (hashjoin.h:40, planner.h:353-358) */
} rsq_op;

typedef struct rsq_plan_desc {
    const rsq_expr* exprs;
    int32_t         n_exprs;
    const rsq_op*   ops;
    int32_t         n_ops;
    int32_t         root;            /* operator index; must be MATERIALIZE or ORDERBY (planner.h:488-491) */
    int32_t         request_all;     /* JitContextFlounder::requestAll (select *) */
    int32_t         has_limit;       /* root->addLimit(limit) (planner.h:494-496) */
    int64_t         limit;
} rsq_plan_desc;

/* ---- tables ---------------------------------------------------------------------------- */
/*
 * A table is handed over column-major.  Element widths follow the reference's in-tuple
 * widths (types.h:213-261) minus its string terminators:
 *   INT, DATE 4 B; BIGINT, DECIMAL 8 B; BOOL 1 B; CHAR(n) / VARCHAR(n): n bytes per row,
 *   NUL padded (CHAR(1): 1 byte).
 * `data` may be NULL: the column exists in the schema (it contributes to the reference's
 * tuple stride, e.g. lineitem's 146 B) but no plan may touch it.
 */
typedef struct rsq_column {
    char        name[RSQ_SYMBOL_MAX];
    rsq_type    type;
    const void* data;      /* host pointer (oracle, rsq_table_create) or device pointer (rsq_table_create_device) */
} rsq_column;

typedef struct rsq_table_desc {
    char              name[RSQ_SYMBOL_MAX];
    int64_t           n_rows;
    int32_t           n_cols;
    const rsq_column* cols;
} rsq_table_desc;

/* ---- results --------------------------------------------------------------------------- */
/*
 * Result relation in the reference's own packed-tuple layout (schema.h:76-106 offsets,
 * strings by value and NUL terminated, values.h:136-148), one contiguous buffer of
 * n_rows * tuple_size bytes, so `serializeRelation` (dbdata.h:688-701) style printing on the
 * ReSQL side is unchanged.
 */
typedef struct rsq_result_view {
    int32_t         n_cols;
    const char    (*names)[RSQ_SYMBOL_MAX];
    const rsq_type* types;
    const int32_t*  offsets;     /* byte offset of each attribute in a tuple */
    int32_t         tuple_size;
    int64_t         n_rows;
    const uint8_t*  tuples;
} rsq_result_view;

#ifdef __cplusplus
}
#endif
#endif /* RESQL_PLAN_H */
