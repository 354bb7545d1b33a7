// resql_hip_binding.h — the ReSQL-side binding of the MI355X engine (include/resql_hip.h).
//
// This is the file a ReSQL maintainer adds to ReSQL's src/ to run SELECT plans on the GPU: it walks ReSQL's OWN
// operator tree (src/operators/*.h) and expression trees (src/expressions.h), describes them as the plain-C
// rsq_plan_desc of include/resql_plan.h, hands ReSQL's row-store Relations (src/dbdata.h) to the engine, and turns
// the result back into a ReSQL Relation — so that executeSelectPlanHip() is a drop-in for executeSelectPlan()
// (src/execute.h:213-247).  It contains no engine logic: everything it calls is the C ABI.
//
// It is compiled against the reference's headers where they live (it must be included AFTER
// "operators/JitOperators.h"); oracle/Makefile builds it into oracle/_ref/ref_harness (`--engine hip`), which is how
// the drop-in is exercised with the reference's real classes on the GPU box.
#pragma once

#include <map>
#include <memory>
#include <string>
#include <vector>
#include <cstring>

#include "resql_hip.h"

namespace resql_hip {

struct HipError : public ResqlError {
    int status;
    HipError(int st, const std::string& m) : ResqlError(m), status(st) {}
};

inline rsq_type toRsqType(SqlType t) {
    rsq_type r{(int32_t)t.tag, 0, 0, 0};     // SqlType::Tag == rsq_type_tag by construction
    if (t.tag == SqlType::DECIMAL) { r.precision = t.decimalSpec().precision; r.scale = t.decimalSpec().scale; }
    if (t.tag == SqlType::CHAR) r.len = (int32_t)t.charSpec().num;
    if (t.tag == SqlType::VARCHAR) r.len = (int32_t)t.varcharSpec().num;
    return r;
}

inline SqlType fromRsqType(const rsq_type& t) {
    switch (t.tag) {
        case RSQ_DECIMAL: return TypeInit::DECIMAL((uint8_t)t.precision, (uint8_t)t.scale);
        case RSQ_CHAR: return TypeInit::CHAR((size_t)t.len);
        case RSQ_VARCHAR: return TypeInit::VARCHAR((size_t)t.len);
        case RSQ_INT: return TypeInit::INT();
        case RSQ_BIGINT: return TypeInit::BIGINT();
        case RSQ_DATE: return TypeInit::DATE();
        case RSQ_BOOL: return TypeInit::BOOL();
        default: throw ResqlError("result type not representable");
    }
}

// Describes a ReSQL plan (RelOperator tree + Expr trees) as rsq_plan_desc.
class PlanDescriber {
public:
    std::vector<rsq_expr> exprs;
    std::vector<rsq_op> ops;
    std::vector<Relation*> tables;
    std::vector<std::string> tableNames;

    int expr(Expr* e) {
        // The binding runs INSTEAD of ReSQL's deriveExpressionTypes (the engine derives types itself, with the same
        // rules), so a TYPECAST met here was written in the query (expr :: type, parser.y:96): it travels with its
        // target type as symbol (resql_plan.h).  Call run() on a plan whose types have not been derived.
        auto it = exprIds_.find(e);
        if (it != exprIds_.end()) return it->second;           // shared Expr* => shared node (planner.h:90-99, :430)
        rsq_expr d{};
        d.tag = (int32_t)e->tag;                                 // Expr::Tag == rsq_expr_tag by construction
        // operands as ReSQL's own code generation reads them: child for UNARY, child and child->next for BINARY — the
        // sibling chain may be longer (planner.h rewrites leave stale ->next links, e.g. below a flattened BETWEEN)
        const size_t limit = e->structureTag == Expr::UNARY ? 1 : e->structureTag == Expr::BINARY ? 2
                           : e->structureTag == Expr::LITERAL ? 0 : (size_t)-1;
        std::vector<int> kids;
        for (Expr* c = e->child; c != nullptr && kids.size() < limit; c = c->next) kids.push_back(expr(c));
        if (kids.size() > RSQ_MAX_CHILDREN) throw ResqlError("expression with too many children");
        d.n_children = (int32_t)kids.size();
        for (size_t i = 0; i < kids.size(); i++) d.child[i] = kids[i];
        d.const_category = (e->tag == Expr::CONSTANT) ? (int32_t)e->type.tag : RSQ_NT;
        std::string sym = e->symbol;
        if (e->tag == Expr::TYPECAST) {
            SqlType t = e->type;
            switch (t.tag) {
                case SqlType::DECIMAL: sym = "DECIMAL " + std::to_string((int)t.decimalSpec().precision) + " " + std::to_string((int)t.decimalSpec().scale); break;
                case SqlType::CHAR: sym = "CHAR " + std::to_string(t.charSpec().num); break;
                case SqlType::VARCHAR: sym = "VARCHAR " + std::to_string(t.varcharSpec().num); break;
                case SqlType::INT: sym = "INT"; break;
                case SqlType::BIGINT: sym = "BIGINT"; break;
                case SqlType::DATE: sym = "DATE"; break;
                case SqlType::BOOL: sym = "BOOL"; break;
                default: throw ResqlError("typecast to an unsupported type");
            }
        }
        if (e->tag == Expr::CONSTANT && (e->type.tag == SqlType::BIGINT || e->type.tag == SqlType::DECIMAL)) {
            // parser.y:149-151 negates the VALUE of `- literal` and leaves the text alone: tell the engine (resql_plan.h)
            Expr* again = ExprGen::constant(e->symbol, e->type.tag);
            const long long parsed = e->type.tag == SqlType::BIGINT ? (long long)again->value.bigintData : (long long)again->value.decimalData;
            const long long held = e->type.tag == SqlType::BIGINT ? (long long)e->value.bigintData : (long long)e->value.decimalData;
            freeExpr(again);
            if (parsed != 0 && held == -parsed) sym = "neg " + sym;
        }
        if (sym.size() >= RSQ_SYMBOL_MAX) throw ResqlError("symbol too long: " + sym);
        std::strncpy(d.symbol, sym.c_str(), RSQ_SYMBOL_MAX - 1);
        exprs.push_back(d);
        exprIds_[e] = (int)exprs.size() - 1;
        return (int)exprs.size() - 1;
    }

    int table(Relation* rel, const std::string& name) {
        for (size_t i = 0; i < tables.size(); i++) if (tables[i] == rel) return (int)i;
        tables.push_back(rel);
        tableNames.push_back(name.empty() ? "rel" + std::to_string(tables.size()) : name);
        return (int)tables.size() - 1;
    }

    int op(RelOperator* o, bool* hasLimit = nullptr, int64_t* limit = nullptr) {
        rsq_op d{};
        d.child[0] = d.child[1] = -1; d.table = -1;
        auto list = [&](std::vector<Expr*>& v, int32_t& n, int32_t* dst) {
            if (v.size() > RSQ_MAX_OP_EXPRS) throw ResqlError("too many expressions on one operator");
            n = (int32_t)v.size();
            for (size_t i = 0; i < v.size(); i++) dst[i] = expr(v[i]);
        };
        if (auto* s = dynamic_cast<ScanOp*>(o)) {
            d.tag = RSQ_OP_SCAN; d.table = table(s->_rel, s->relationName);
        } else if (auto* s = dynamic_cast<SelectionOp*>(o)) {
            d.tag = RSQ_OP_SELECTION; d.child[0] = op(s->_child);
            d.n_exprs = 1; d.exprs[0] = expr(s->_condition);
        } else if (auto* p = dynamic_cast<ProjectionOp*>(o)) {
            if (p->_child == nullptr) throw ResqlError("leaf projections are not supported by the HIP engine");
            d.tag = RSQ_OP_PROJECTION; d.child[0] = op(p->_child); list(p->_expr, d.n_exprs, d.exprs);
        } else if (auto* h = dynamic_cast<HashJoinOp*>(o)) {
            d.tag = RSQ_OP_HASHJOIN; d.child[0] = op(h->_lChild); d.child[1] = op(h->_rChild);
            list(h->_equalities, d.n_exprs, d.exprs); d.single_match = h->_singleMatch ? 1 : 0;
        } else if (auto* a = dynamic_cast<AggregationOp*>(o)) {
            d.tag = RSQ_OP_AGGREGATION; d.child[0] = op(a->_child);
            list(a->_aggExpr, d.n_exprs, d.exprs); list(a->_groupExpr, d.n_exprs2, d.exprs2);
        } else if (auto* ob = dynamic_cast<OrderByOp*>(o)) {
            // OrderByOp owns a MaterializeOp child (orderby.h:32-38); the engine re-creates it
            d.tag = RSQ_OP_ORDERBY; d.child[0] = op(ob->_child->_child);
            list(ob->_orderExpressions, d.n_exprs, d.exprs);
            if (hasLimit && ob->_hasLimitClause) { *hasLimit = true; *limit = (int64_t)ob->_limit; }
        } else if (auto* m = dynamic_cast<MaterializeOp*>(o)) {
            d.tag = RSQ_OP_MATERIALIZE; d.child[0] = op(m->_child);
            if (hasLimit && m->_hasLimitClause) { *hasLimit = true; *limit = (int64_t)m->_limit; }
        } else {
            throw ResqlError("operator " + o->name() + " is not supported by the HIP engine");
        }
        ops.push_back(d);
        return (int)ops.size() - 1;
    }

private:
    std::map<Expr*, int> exprIds_;
};

// One engine context + the device copies of the Relations it has seen.
class JitContextHip {
public:
    // compat: rsq_compat bits.  The drop-in's default is RSQ_COMPAT_JIT_INT16_CAST: the answers ReSQL's asmjit JIT gives TODAY, bit for
    // bit; 0 computes what the reference's source specifies (what ReSQL answers once its movsx is fixed)
    explicit JitContextHip(const JitConfig& cfg, int device = 0, uint32_t compat = RSQ_COMPAT_JIT_INT16_CAST) {
        rsq_config c{};
        c.struct_size = sizeof c;
        c.compat_flags = compat;
        c.print_assembly = cfg.printAssembly; c.print_flounder = cfg.printFlounder; c.print_performance = cfg.printPerformance;
        c.num_threads = cfg.numThreads; c.emit_machine_code = cfg.emitMachineCode; c.optimize = cfg.optimizeFlounder;
        c.device = device;
        int rc = rsq_ctx_create(&c, &ctx_);
        if (rc != RSQ_OK) throw HipError(rc, rsq_last_error(nullptr));
        report.config = cfg;
    }
    ~JitContextHip() {
        for (auto& kv : tables_) rsq_table_destroy(kv.second.t);
        if (ctx_) rsq_ctx_destroy(ctx_);
    }

    // Relation (row store of 2 MiB DataBlocks, dbdata.h:23-102) -> device columns.  A Relation grows (executeBulkInsert appends,
    // execute.h:332-388; AppendIterator, dbdata.h:246-330) and never shrinks or rewrites: what was transposed stays valid, and a
    // Relation that holds more tuples than its device copy gets only the NEW tuples transposed and appended (rsq_table_append) -
    // the tail of the block that was last in use, then the blocks added since.
    rsq_table* deviceTable(Relation* rel, const std::string& name) {
        auto it = tables_.find(rel);
        if (it == tables_.end()) {
            Resident r;
            r.t = fromBlocks(rel, name, 0, 0);
            noteSynced(rel, r);
            tables_[rel] = r;
            return r.t;
        }
        Resident& r = it->second;
        if (r.deviceOnly || rel->tupleNum() == r.tuples) return r.t;
        if (rel->tupleNum() < r.tuples) throw ResqlError("Relation " + name + " holds fewer tuples than its device copy.");
        rsq_table* more = fromBlocks(rel, name, r.blocks ? r.blocks - 1 : 0, r.blocks ? r.lastBlockBytes : 0);
        int rc = rsq_table_append(r.t, more);
        if (rc != RSQ_OK) { rsq_table_destroy(more); check(rc); }
        noteSynced(rel, r);
        return r.t;
    }

    // BULK INSERT (execute.h:332-388) straight into device columns: the Relation object stays the table's identity
    // (plans scan it), its rows live on the GPU only.  Same field rules as the reference's loop (csrc/tbl.cpp).  A second
    // BULK INSERT into the same table appends, as the reference's does.
    int64_t bulkInsert(Relation* rel, const std::string& name, const std::string& fileName, char fieldTerminator) {
        std::vector<rsq_column> cols = columnsOf(rel);
        rsq_table_desc d{};
        std::strncpy(d.name, name.c_str(), RSQ_SYMBOL_MAX - 1);
        d.n_cols = (int32_t)cols.size(); d.cols = cols.data();
        rsq_table* t = nullptr;
        check(rsq_table_load_tbl(ctx_, &d, fileName.c_str(), fieldTerminator, 0, &t));
        const int64_t inserted = rsq_table_rows(t);
        auto it = tables_.find(rel);
        if (it == tables_.end()) {
            if (rel->tupleNum() != 0) { rsq_table_destroy(t); throw ResqlError("Table " + name + " holds host tuples: BULK INSERT onto the device needs an empty or device-resident table."); }
            Resident r; r.t = t; r.deviceOnly = true;
            tables_[rel] = r;
        } else {
            int rc = rsq_table_append(it->second.t, t);
            if (rc != RSQ_OK) { rsq_table_destroy(t); check(rc); }
        }
        return inserted;
    }

    // describe + compile + execute + retrieve: the body of executeSelectPlan (execute.h:213-247)
    std::unique_ptr<Relation> run(RelOperator* root, bool requestAll) {
        PlanDescriber pd;
        bool hasLimit = false; int64_t limit = 0;
        int rootIdx = pd.op(root, &hasLimit, &limit);
        rsq_plan_desc plan{};
        plan.exprs = pd.exprs.data(); plan.n_exprs = (int32_t)pd.exprs.size();
        plan.ops = pd.ops.data(); plan.n_ops = (int32_t)pd.ops.size();
        plan.root = rootIdx; plan.request_all = requestAll ? 1 : 0;
        plan.has_limit = hasLimit ? 1 : 0; plan.limit = limit;
        std::vector<rsq_table*> tabs;
        for (size_t i = 0; i < pd.tables.size(); i++) tabs.push_back(deviceTable(pd.tables[i], pd.tableNames[i]));
        rsq_query* q = nullptr;
        check(rsq_query_compile(ctx_, &plan, tabs.data(), (int32_t)tabs.size(), &q));
        struct Guard { rsq_query* q; ~Guard() { rsq_query_destroy(q); } } guard{q};
        check(rsq_query_execute(q));
        rsq_report r{};
        rsq_query_report(q, &r);
        report.compilationTime = r.compilation_time_ms;
        report.executionTime = r.execution_time_ms;
        report.numMachineInstructions = r.num_kernels;
        kernelTimeMs = r.kernel_time_ms; hbmGBps = r.hbm_gbps;
        rsq_result_view v{};
        check(rsq_query_result(q, &v));
        // packed tuples have ReSQL's own layout (schema.h:76-106): copy them into a Relation block by block
        std::vector<Attribute> atts;
        for (int i = 0; i < v.n_cols; i++) atts.push_back({std::string(v.names[i]), fromRsqType(v.types[i])});
        auto rel = std::make_unique<Relation>(Schema(atts));
        if ((size_t)v.tuple_size != rel->_schema._tupSize) throw ResqlError("result tuple layout mismatch");
        Relation::AppendIterator app(rel.get());
        for (int64_t t = 0; t < v.n_rows; t++)
            std::memcpy(app.get(), v.tuples + (size_t)t * (size_t)v.tuple_size, (size_t)v.tuple_size);
        return rel;
    }

    JitExecutionReport report;
    double kernelTimeMs = 0, hbmGBps = 0;

private:
    void check(int rc) { if (rc != RSQ_OK) throw HipError(rc, rsq_last_error(ctx_)); }
    // the device copy of a Relation and how much of the Relation it holds
    struct Resident { rsq_table* t = nullptr; size_t tuples = 0, blocks = 0, lastBlockBytes = 0; bool deviceOnly = false; };
    std::vector<rsq_column> columnsOf(Relation* rel) {
        std::vector<rsq_column> cols;
        for (auto& a : rel->_schema._attribs) {
            rsq_column c{};
            std::strncpy(c.name, a.name.c_str(), RSQ_SYMBOL_MAX - 1);
            c.type = toRsqType(a.type);
            cols.push_back(c);
        }
        return cols;
    }
    // the tuples from byte `firstOffset` of block `firstBlock` on, transposed into a new device table
    rsq_table* fromBlocks(Relation* rel, const std::string& name, size_t firstBlock, size_t firstOffset) {
        std::vector<rsq_column> cols = columnsOf(rel);
        rsq_table_desc d{};
        std::strncpy(d.name, name.c_str(), RSQ_SYMBOL_MAX - 1);
        d.n_cols = (int32_t)cols.size(); d.cols = cols.data();
        std::vector<const uint8_t*> blocks; std::vector<size_t> sizes;
        for (size_t b = firstBlock; b < rel->_dataBlocks.size(); b++) {
            const size_t off = b == firstBlock ? firstOffset : 0;
            blocks.push_back((const uint8_t*)rel->_dataBlocks[b]->begin() + off);
            sizes.push_back(rel->_dataBlocks[b]->_contentSize - off);
        }
        rsq_table* t = nullptr;
        check(rsq_table_from_rowstore(ctx_, &d, blocks.data(), sizes.data(), (int32_t)blocks.size(), &t));
        return t;
    }
    void noteSynced(Relation* rel, Resident& r) {
        r.tuples = rel->tupleNum(); r.blocks = rel->_dataBlocks.size();
        r.lastBlockBytes = r.blocks ? rel->_dataBlocks.back()->_contentSize : 0;
    }
    rsq_ctx* ctx_ = nullptr;
    std::map<Relation*, Resident> tables_;
};

}  // namespace resql_hip
