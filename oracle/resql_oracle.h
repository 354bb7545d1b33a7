/*
 * resql_oracle.h — CPU ORACLE for the ReSQL operator pipelines.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded, tuple-at-a-time restatement of what the reference's
 * Flounder/asmjit JIT path computes for scan -> selection -> hash join -> hash aggregation
 * -> projection -> materialize -> order by (reference src/operators/ *.h,
 * src/ExpressionsJitFlounder.h, src/ValuesJitFlounder.h, src/expressions.h, src/qlib/ *.h).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (resql_amd/, include/resql_hip.h) never links, imports or calls it.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py) against
 *   (1) the literal input/expected tables of the reference's own tests
 *       (test/test_operators.h, test/test_expressions.h, test/test_datatypes.h), and
 *   (2) outputs of the unmodified reference itself, compiled from /root/reference by
 *       oracle/Makefile (`make ref`) and run on seeded inputs; the outputs are committed
 *       under tests/golden/ together with the generating script.
 */
#ifndef RESQL_ORACLE_H
#define RESQL_ORACLE_H

#include "resql_plan.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_result orc_result;

/* Run `plan` over `tables` (host columnar data).  Returns 0 and *out on success, non-zero and a
 * message in err otherwise (the reference throws ResqlError / exits in the same situations). */
int orc_execute(const rsq_plan_desc* plan, const rsq_table_desc* tables, int n_tables,
                orc_result** out, char* err, size_t errlen);

/* Packed-tuple view of the result (reference layout, schema.h:76-106). */
const rsq_result_view* orc_result_view(const orc_result* r);

/* Result serialised like the reference's serializeRelation (dbdata.h:688-701): one line per
 * tuple, every value followed by '|'.  First line "#schema name:TYPE|...".  malloc'ed; free
 * with orc_free_string. */
char* orc_result_serialize(const orc_result* r);

/* Slot order / sizes of the aggregation hash table of the (last) AGGREGATION operator, for
 * tests that pin the hash-table restatement: number of slots and number of grow events. */
int64_t orc_result_agg_slots(const orc_result* r);
int64_t orc_result_agg_grows(const orc_result* r);
/* number of probes for which the reference itself reads one byte past its hash table (qlib/hash.h:441-451, a continued
 * probe after the last slot): when > 0 the reference's result for this input depends on heap contents */
int64_t orc_result_ref_oob_probes(const orc_result* r);
/* INT -> BIGINT casts of values outside the int16 range: the reference's asmjit back end sign-extends only the low 16
 * bits there (see resql_oracle.c g_narrowCasts), so its own answer is not comparable when this is non-zero */
int64_t orc_result_ref_narrow_casts(const orc_result* r);

void orc_result_free(orc_result* r);
void orc_free_string(char* s);

/* Type derivation only: serialise expression `expr` of `plan` like the reference's
 * serializeExpr (expressions.h:177-204), before (derive=0) or after (derive=1)
 * deriveExpressionTypes (expressions.h:1367-1392).  Column types come from `tables`
 * (may be NULL/0 for constant-only expressions).  malloc'ed string or NULL + err. */
char* orc_serialize_expr(const rsq_plan_desc* plan, int expr, int derive,
                         const rsq_table_desc* tables, int n_tables, char* err, size_t errlen);

/* Evaluate a constant-only scalar expression like the reference's
 * compileAndEvaluateScalarExpression (test/test_common.h:65-82) and serialise the value with
 * serializeSqlValue (values.h:30-127).  malloc'ed string or NULL + err. */
char* orc_eval_scalar(const rsq_plan_desc* plan, int expr, char* err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif
