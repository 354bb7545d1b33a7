/* sql_host.c — a plain-C host of the engine's statement loop (rsq_db_*), the shape of the reference's executeStatement
 * (src/execute.h:508-545): every argument is one statement; SELECT results are printed with serializeRelation's format.
 *
 *   gcc -std=c11 -Iinclude integration/examples/sql_host.c -Lresql_amd -lresql_hip -Wl,-rpath,$PWD/resql_amd -o sql_host
 *   ./sql_host 'create table nation ( n_nationkey int, n_name char(25), n_regionkey int, n_comment varchar(152) )' \
 *              'bulk insert nation from "nation.tbl" with ( fieldterminator="|" )' \
 *              'select n_name from nation where n_regionkey = 2 order by n_name'
 *
 * Control statements of the reference's processControl (execute.h:454-474) work as arguments too: 'showperf=true',
 * 'showplan=true', 'tofile=true', 'threads=4', 'tables', ...
 *
 * RSQ_DEVICE=-1 in the environment selects a compile-only context (no GPU: statements are parsed, planned and compiled for
 * gfx950, executing a SELECT then fails with RSQ_ERR_DEVICE) — which is how the CPU test suite links and runs this file. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "resql_hip.h"

int main(int argc, char** argv) {
    rsq_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.device = getenv("RSQ_DEVICE") ? atoi(getenv("RSQ_DEVICE")) : 0;
    rsq_ctx* ctx = NULL;
    if (rsq_ctx_create(&cfg, &ctx) != RSQ_OK) { fprintf(stderr, "context: %s\n", rsq_last_error(NULL)); return 2; }
    rsq_db* db = NULL;
    if (rsq_db_create(ctx, &db) != RSQ_OK) { fprintf(stderr, "database: %s\n", rsq_last_error(ctx)); return 2; }
    int failed = 0;
    for (int i = 1; i < argc; i++) {
        int32_t kind = 0;
        rsq_result_view view;
        int st = rsq_db_execute(db, argv[i], &kind, &view);
        if (st != RSQ_OK) { printf("error %d: %s\n", st, rsq_last_error(ctx)); failed++; continue; }
        if (kind == 1) {
            /* printQueryResult (execute.h:178-183): plan line and report first (empty unless showplan / showperf / ... are set) */
            const char* msg = rsq_db_message(db);
            if (msg[0] && strcmp(msg, "\n") != 0) printf("%s", msg);
            char* text = rsq_result_serialize(&view);
            printf("%lld row(s)\n%s", (long long)view.n_rows, text ? text : "");
            rsq_free(text);
        } else if (kind == 4) printf("%s", rsq_db_message(db));            /* control statement: its answer, if any */
        else printf("%s ok\n", kind == 2 ? "create table" : "bulk insert");
    }
    rsq_db_destroy(db);
    rsq_ctx_destroy(ctx);
    return failed ? 1 : 0;
}
