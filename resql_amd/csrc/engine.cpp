// engine.cpp — describe -> compile -> execute -> retrieve for one query.
//
// The walk mirrors the reference's produce/consume code generation (reference
// src/operators/*.h, driven from src/execute.h:213-247): produce() recurses to the scans, every scan
// opens a pipeline (scan.h:227-263), and the operators above it consume() into that pipeline
// until a pipeline breaker (hash-join build hashjoin.h:226-256, aggregation aggregation.h:240-295)
// or the materialisation ends it.  Instead of Flounder IR the operators emit the body of a HIP
// row function; the hand-written kernel skeleton around it (tile loads, reductions, hash tables)
// is kernels/rsq_device.h.
//
// Everything above the last pipeline breaker — AVG finalisation, projection, materialisation,
// ORDER BY, LIMIT over the group rows (aggregation.h:298-343, projection.h:62-72,
// materialize.h:78-220, orderby.h:87-136) — is evaluated on the host: it touches #groups rows.
#include "engine.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <set>
#include <sstream>

#include "hostref.h"

namespace rsq {

namespace {

double nowMs() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

struct Attr { std::string name; Type type; };
typedef std::vector<Attr> Schema;

int schemaTupleSize(const Schema& s) { int n = 0; for (auto& a : s) n += sizeInTuple(a.type, true); return n; }
int schemaOffset(const Schema& s, const std::string& name) {
    int off = 0;
    for (auto& a : s) { if (a.name == name) return off; off += sizeInTuple(a.type, true); }
    failType("The attribute " + name + " was not found in the schema");
}

struct OpNode {
    int tag = RSQ_OP_UNDEFINED;
    OpNode* parent = nullptr;
    OpNode* child[2] = {nullptr, nullptr};
    int nChildren = 0;
    std::vector<Expr*> exprs, exprs2;
    bool singleMatch = false;
    bool hasLimit = false;
    int64_t limit = 0;
    Table* table = nullptr;
    std::vector<Expr*> splitAgg;     // aggregation.h:167-179
    Schema schema;                   // RelOperator::_schema
};

// device variable bound to a symbol during code generation
struct Sym { std::string var; Type type; };

// one accumulator the aggregation kernel keeps per group
struct Accum {
    int kind;            // RSQ_E_SUM / RSQ_E_MIN / RSQ_E_MAX / RSQ_E_COUNT
    std::string key;     // structural key of (kind, typed input) for common-subexpression detection
    std::string input;   // device expression of the input value
    Type type;
    int merge;           // rsq::Merge on the device
};

struct DenseKey {
    Expr* expr = nullptr;
    Type type;
    bool byteSet = false;
    std::vector<uint8_t> values;     // byteSet: sorted distinct values
    int64_t min = 0;
    int64_t card = 1;
    int64_t stride = 1;
};

struct ArgSlot { std::string name; std::string ctype; uint64_t value; };

struct Pipeline {
    Table* src = nullptr;
    std::vector<int> cols;           // scanned columns (indices into src->cols)
    std::string source;              // generated HIP source
    std::string entry = "rsq_pipeline";
    std::vector<ArgSlot> args;
    Kernel* kernel = nullptr;
    int64_t bytesPerRow = 0;
    std::string explain;
};

}  // namespace

struct Query {
    Context& ctx;
    ExprPool pool;
    std::vector<Expr*> exprs;
    std::vector<std::unique_ptr<OpNode>> ops;
    OpNode* root = nullptr;
    bool requestAll = false;
    std::vector<Table*> tables;

    // device side
    std::vector<Pipeline> pipelines;
    OpNode* agg = nullptr;                 // the aggregation whose input pipeline runs on the device
    std::vector<DenseKey> denseKeys;
    int64_t denseGroups = 1;
    int unroll = 4;
    std::vector<Accum> accums;             // [0] is the first-row tracker
    std::vector<int> splitToAccum;         // splitAgg index -> accums index
    bool dAggOwned = true;
    uint64_t* dAgg = nullptr;              // [min words | max words | sum words], each nWords(op) * denseGroups
    std::vector<int> accumSlot;            // accums index -> word-block index in dAgg
    int64_t nMinBlocks = 0, nMaxBlocks = 0, nSumBlocks = 0;
    std::vector<uint64_t> hAgg;
    uint64_t* dAggInit = nullptr;          // identity image, copied over dAgg at the start of every execute
    uint64_t* hPinned = nullptr;           // pinned read-back buffer: aggregate words + error word

    // result
    Schema resultSchema;
    std::vector<uint8_t> resultTuples;
    int64_t resultRows = 0;
    std::vector<char> rvNames;
    std::vector<rsq_type> rvTypes;
    std::vector<int32_t> rvOffsets;

    rsq_report report{};
    std::string allSource, explainText;

    explicit Query(Context& c) : ctx(c) {}
    ~Query() {
        if (dAgg && dAggOwned) ctx.free(dAgg);
        if (dAggInit) ctx.free(dAggInit);
        if (hPinned) (void)hipHostFree(hPinned);
    }
};

namespace {

// ================================================================================================
// expression -> device code
// ================================================================================================
struct DeviceCodegen {
    Query& q;
    std::map<std::string, Sym> symbols;     // JitContextFlounder::symbolTable
    bool usesErr = false;

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

    // emitExpression (reference src/ExpressionsJitFlounder.h:1080-1114)
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
        if (e->tag == RSQ_E_COUNT) return "((i64)1)";           // emitCount: every row counts
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
                    if (from.tag == RSQ_INT) return "((i64)(" + c + "))";
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

    static int64_t pow10(int n) { int64_t v = 1; while (n-- > 0) v *= 10; return v; }

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
        auto equals = [&]() -> std::string {
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
            case RSQ_E_DIV: arithOk(); usesErr = true; return "rsq::div(" + l + ", " + r + ", a.err)";
            case RSQ_E_AND: return "((u8)((" + l + ") & (" + r + ")))";      // no short circuit, as in the reference
            case RSQ_E_OR: return "((u8)((" + l + ") | (" + r + ")))";
            case RSQ_E_LT: orderedOk(); return "((u8)((" + l + ") < (" + r + ")))";
            case RSQ_E_LE: orderedOk(); return "((u8)((" + l + ") <= (" + r + ")))";
            case RSQ_E_GT: orderedOk(); return "((u8)((" + l + ") > (" + r + ")))";
            case RSQ_E_GE: orderedOk(); return "((u8)((" + l + ") >= (" + r + ")))";
            case RSQ_E_EQ: return equals();
            case RSQ_E_NEQ: return "((u8)(1 - " + equals() + "))";
            case RSQ_E_LIKE: failUnsupported("LIKE is not implemented by the GPU engine (outside the hot path, SURVEY §2)");
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

// ================================================================================================
// host evaluation over group rows (same semantics, same typing)
// ================================================================================================
struct HostEval {
    std::map<std::string, std::pair<Val, Type>> symbols;

    static int cmp(Val a, Val b, const Type& t) {
        if (t.tag == RSQ_DATE || t.tag == RSQ_INT) {
            int32_t x = (int32_t)(uint32_t)a.i, y = (int32_t)(uint32_t)b.i;
            return x < y ? -1 : x > y;
        }
        return a.i < b.i ? -1 : a.i > b.i;
    }
    static bool strEqChar(const char* a, const char* b) {
        while (*a && *b) { if (*a != *b) return false; a++; b++; }
        while (*a) { if (*a != ' ') return false; a++; }
        while (*b) { if (*b != ' ') return false; b++; }
        return true;
    }
    static bool strEqVarchar(const char* a, const char* b) {
        while (*a && *b) { if (*a != *b) return false; a++; b++; }
        return *a == *b;
    }
    static bool equals(Val a, Val b, const Type& t) {
        switch (t.tag) {
            case RSQ_DECIMAL: case RSQ_BIGINT: return a.i == b.i;
            case RSQ_INT: case RSQ_DATE: return (uint32_t)a.i == (uint32_t)b.i;
            case RSQ_BOOL: return (uint8_t)a.i == (uint8_t)b.i;
            case RSQ_CHAR: return t.len > 1 ? strEqChar(a.s, b.s) : (uint8_t)a.i == (uint8_t)b.i;
            case RSQ_VARCHAR: return strEqVarchar(a.s, b.s);
            default: failType("EQUALS code generation not implemented for datatype");
        }
    }
    static int64_t sdiv(int64_t a, int64_t b) {
        if (b == 0) failRuntime("Division by zero");
        if (a == INT64_MIN && b == -1) failRuntime("Division overflow");
        return a / b;
    }
    static int64_t pow10(int n) { int64_t v = 1; while (n-- > 0) v *= 10; return v; }

    Val eval(Expr* e) {
        auto it = symbols.find(expressionName(e));
        if (it != symbols.end()) return it->second.first;
        Val r; r.i = 0;
        switch (e->structure) {
            case LITERAL:
                if (e->tag == RSQ_E_CONSTANT) { if (e->type.isString()) r.s = e->symbol.c_str(); else r.i = e->ival; return r; }
                if (e->tag == RSQ_E_STAR) return r;
                failType("attribute " + e->symbol + " is not available after the aggregation");
            case UNARY: {
                if (e->tag == RSQ_E_COUNT) { r.i = 1; return r; }
                Val c = eval(e->child);
                const Type from = e->child->type, to = e->type;
                if (e->tag != RSQ_E_TYPECAST) return c;
                if (to.tag == RSQ_DECIMAL) {
                    if (from.tag == RSQ_DECIMAL) {
                        if (to.scale == from.scale) return c;
                        if (to.scale > from.scale) r.i = (int64_t)((uint64_t)c.i * (uint64_t)pow10(to.scale - from.scale));
                        else r.i = sdiv(c.i, pow10(from.scale - to.scale));
                    } else r.i = (int64_t)((uint64_t)c.i * (uint64_t)pow10(to.scale));
                } else {
                    if (from.tag == RSQ_INT) r.i = (int64_t)(int32_t)c.i;
                    else if (from.tag == RSQ_DECIMAL) r.i = sdiv(c.i, pow10(from.scale));
                    else r = c;
                }
                return r;
            }
            case BINARY: {
                Val a = eval(e->child), b = eval(e->child->next);
                const Type op = e->child->type;
                switch (e->tag) {
                    case RSQ_E_ADD: r.i = (int64_t)((uint64_t)a.i + (uint64_t)b.i); break;
                    case RSQ_E_SUB: r.i = (int64_t)((uint64_t)a.i - (uint64_t)b.i); break;
                    case RSQ_E_MUL: r.i = (int64_t)((uint64_t)a.i * (uint64_t)b.i); break;
                    case RSQ_E_DIV: r.i = sdiv(a.i, b.i); break;
                    case RSQ_E_AND: r.i = (uint8_t)a.i & (uint8_t)b.i; break;
                    case RSQ_E_OR: r.i = (uint8_t)a.i | (uint8_t)b.i; break;
                    case RSQ_E_LT: r.i = cmp(a, b, op) < 0; break;
                    case RSQ_E_LE: r.i = cmp(a, b, op) <= 0; break;
                    case RSQ_E_GT: r.i = cmp(a, b, op) > 0; break;
                    case RSQ_E_GE: r.i = cmp(a, b, op) >= 0; break;
                    case RSQ_E_EQ: r.i = equals(a, b, op); break;
                    case RSQ_E_NEQ: r.i = 1 - (int)equals(a, b, op); break;
                    default: failUnsupported(std::string("host evaluation of ") + exprTagNames[e->tag]);
                }
                return r;
            }
            case OTHER: {
                Expr* c = e->child;
                for (; c && c->tag == RSQ_E_WHENTHEN; c = c->next)
                    if ((uint8_t)eval(c->child).i) return eval(c->child->next);
                if (c) return eval(c);
                return r;
            }
            default: failType("emitExpression(..)");
        }
    }
};

// ================================================================================================
// plan construction + typing
// ================================================================================================
void buildOps(Query& q, const rsq_plan_desc& p) {
    int extra = 0;
    for (int i = 0; i < p.n_ops; i++) if (p.ops[i].tag == RSQ_OP_ORDERBY) extra++;
    for (int i = 0; i < p.n_ops + extra; i++) q.ops.emplace_back(new OpNode());
    int nextExtra = p.n_ops;
    for (int i = 0; i < p.n_ops; i++) {
        const rsq_op& d = p.ops[i];
        OpNode* o = q.ops[i].get();
        o->tag = d.tag;
        if (d.n_exprs < 0 || d.n_exprs > RSQ_MAX_OP_EXPRS || d.n_exprs2 < 0 || d.n_exprs2 > RSQ_MAX_OP_EXPRS) failInvalid("bad expression count");
        for (int k = 0; k < d.n_exprs; k++) { if (d.exprs[k] < 0 || d.exprs[k] >= p.n_exprs) failInvalid("bad expression index"); o->exprs.push_back(q.exprs[d.exprs[k]]); }
        for (int k = 0; k < d.n_exprs2; k++) { if (d.exprs2[k] < 0 || d.exprs2[k] >= p.n_exprs) failInvalid("bad expression index"); o->exprs2.push_back(q.exprs[d.exprs2[k]]); }
        o->singleMatch = d.single_match != 0;
        switch (d.tag) {
            case RSQ_OP_SCAN:
                if (d.table < 0 || d.table >= (int)q.tables.size()) failInvalid("bad table index");
                o->table = q.tables[d.table]; o->nChildren = 0; break;
            case RSQ_OP_HASHJOIN: o->nChildren = 2; break;
            case RSQ_OP_SELECTION: if (d.n_exprs != 1) failInvalid("selection needs one condition"); o->nChildren = 1; break;
            case RSQ_OP_PROJECTION: case RSQ_OP_AGGREGATION: case RSQ_OP_MATERIALIZE: case RSQ_OP_ORDERBY: o->nChildren = 1; break;
            case RSQ_OP_NESTEDLOOPSJOIN: failUnsupported("NestedLoopsJoin is outside the hot path (SURVEY §2)");
            default: failInvalid("unknown operator tag");
        }
        for (int k = 0; k < o->nChildren; k++) {
            int ci = d.child[k];
            if (ci < 0 || ci >= p.n_ops || ci == i) failInvalid("bad child operator index");
            o->child[k] = q.ops[ci].get();
        }
        if (d.tag == RSQ_OP_ORDERBY) {   // OrderByOp wraps its child into a MaterializeOp (orderby.h:32-38)
            OpNode* m = q.ops[nextExtra++].get();
            m->tag = RSQ_OP_MATERIALIZE; m->nChildren = 1; m->child[0] = o->child[0];
            o->child[0] = m;
        }
    }
    for (auto& o : q.ops)
        for (int k = 0; k < o->nChildren; k++) {
            if (o->child[k]->parent && o->child[k]->parent != o.get()) failInvalid("operator has two parents");
            o->child[k]->parent = o.get();
        }
    if (p.root < 0 || p.root >= p.n_ops) failInvalid("bad root");
    q.root = q.ops[p.root].get();
    if (q.root->tag != RSQ_OP_MATERIALIZE && q.root->tag != RSQ_OP_ORDERBY)
        failInvalid("Calling retrieveResult on non-materialized operator");
    if (p.has_limit) { q.root->hasLimit = true; q.root->limit = p.limit; }
}

// defineExpressionsForPlan (RelOperator.h:203-208) + deriveExpressionTypes in definition order
void defineAndDerive(Query& q, OpNode* o) {
    for (int i = 0; i < o->nChildren; i++) defineAndDerive(q, o->child[i]);
    auto all = [&](std::vector<Expr*>& v) { for (Expr* e : v) q.pool.derive(e); };
    switch (o->tag) {
        case RSQ_OP_SCAN: case RSQ_OP_MATERIALIZE: break;
        case RSQ_OP_SELECTION: case RSQ_OP_PROJECTION: case RSQ_OP_HASHJOIN: all(o->exprs); break;
        case RSQ_OP_AGGREGATION:
            for (Expr* e : o->exprs) {
                if (e->tag == RSQ_E_AVG) {
                    o->splitAgg.push_back(q.pool.unary(RSQ_E_SUM, "sum", e->child));
                    o->splitAgg.push_back(q.pool.unary(RSQ_E_COUNT, "count", e->child));
                } else o->splitAgg.push_back(e);
            }
            all(o->exprs2); all(o->splitAgg); all(o->exprs);
            break;
        case RSQ_OP_ORDERBY:
            for (auto& e : o->exprs) if (e->tag != RSQ_E_ASC && e->tag != RSQ_E_DESC) e = q.pool.unary(RSQ_E_ASC, "asc", e);
            for (Expr* e : o->exprs)
                if (e->child->tag != RSQ_E_ATTRIBUTE) failType("Order by only supports attribute expressions currently." + serializeExpr(e->child));
            all(o->exprs);
            break;
        default: failInvalid("unsupported operator in plan");
    }
}

uint64_t opSize(OpNode* o) {   // getSize() estimates (operators/*.h)
    switch (o->tag) {
        case RSQ_OP_SCAN: return (uint64_t)o->table->nRows;
        case RSQ_OP_SELECTION: return opSize(o->child[0]) / 2;
        case RSQ_OP_PROJECTION: case RSQ_OP_ORDERBY: return opSize(o->child[0]);
        case RSQ_OP_HASHJOIN: return opSize(o->child[0]) + opSize(o->child[1]) / 2;
        case RSQ_OP_AGGREGATION: {
            if (o->exprs2.empty()) return 1;
            int red = 512;
            for (size_t i = 1; i < o->exprs2.size() && red > 2; i++) red /= 2;
            return opSize(o->child[0]) / (uint64_t)red;
        }
        case RSQ_OP_MATERIALIZE: { uint64_t s = opSize(o->child[0]); if (o->hasLimit && (uint64_t)o->limit < s) s = (uint64_t)o->limit; return s; }
        default: return 0;
    }
}

// ================================================================================================
// device pipeline: scan -> selection* -> aggregation with a dense group id
// ================================================================================================
const int kRegisterGroupsMax = 8;       // groups whose accumulators live in VGPRs
int unrollFactor() {                    // tiles in flight per wave (RSQ_UNROLL overrides, for tuning)
    const char* e = getenv("RSQ_UNROLL");
    int u = e ? atoi(e) : 2;
    return u < 1 ? 1 : u > 8 ? 8 : u;
}

struct ScanAggBuilder {
    Query& q;
    OpNode* agg;
    OpNode* scan = nullptr;
    std::vector<OpNode*> selections;    // bottom-up
    DeviceCodegen cg;
    Pipeline pipe;

    ScanAggBuilder(Query& q_, OpNode* a) : q(q_), agg(a), cg{q_} {}

    void findChain() {
        OpNode* o = agg->child[0];
        std::vector<OpNode*> sels;
        while (o->tag == RSQ_OP_SELECTION) { sels.push_back(o); o = o->child[0]; }
        if (o->tag != RSQ_OP_SCAN)
            failUnsupported("aggregation input must be selection(s) over a scan in this engine version");
        scan = o;
        selections.assign(sels.rbegin(), sels.rend());
    }

    void addArg(const std::string& name, const std::string& ctype, uint64_t v) { pipe.args.push_back({name, ctype, v}); }

    void planDenseKeys() {
        Table* t = scan->table;
        int64_t total = 1;
        for (Expr* g : agg->exprs2) {
            if (g->tag != RSQ_E_ATTRIBUTE) failUnsupported("group-by on a computed expression needs the hash aggregation path");
            int ci = t->findCol(g->symbol);
            if (ci < 0 || !t->cols[ci].dptr) failType("Attribute " + g->symbol + " not found.");
            const TableColumn& c = t->cols[ci];
            DenseKey k; k.expr = g; k.type = c.type;
            if (t->nRows == 0) { k.card = 1; k.min = 0; }      // empty input: no row reaches the aggregation, no group exists
            else if (!c.stats.valid) failUnsupported("no statistics for group-by column " + g->symbol);
            else if (!c.stats.distinctBytes.empty()) { k.byteSet = true; k.values = c.stats.distinctBytes; k.card = (int64_t)k.values.size(); }
            else {
                if (c.type.isString()) failUnsupported("string group-by keys need the hash aggregation path");
                k.min = c.stats.min;
                unsigned __int128 range = (unsigned __int128)((__int128)c.stats.max - (__int128)c.stats.min) + 1;
                if (range > (unsigned __int128)(1u << 30)) failUnsupported("group-by domain too large for dense aggregation");
                k.card = (int64_t)range;
            }
            if (total > (int64_t)(1 << 30) / k.card) failUnsupported("group-by domain too large for dense aggregation");
            total *= k.card;
            q.denseKeys.push_back(k);
        }
        int64_t stride = 1;
        for (size_t i = q.denseKeys.size(); i-- > 0;) { q.denseKeys[i].stride = stride; stride *= q.denseKeys[i].card; }
        q.denseGroups = total;
        if (total > kRegisterGroupsMax)
            failUnsupported("dense aggregation with " + std::to_string((long long)total) + " groups: only the register variant (<= 8) is built yet");
    }

    void build() {
        findChain();
        Table* t = scan->table;

        // ---- requested attributes (produce: request sets flow down, scan.h:221-263) ----
        std::vector<std::string> req;
        for (Expr* e : agg->exprs) requiredAttributes(e, req);
        for (Expr* e : agg->exprs2) requiredAttributes(e, req);
        for (OpNode* s : selections) requiredAttributes(s->exprs[0], req);
        std::string rowParams, rowArgsTail;
        std::vector<std::string> colTypes;
        for (size_t ci = 0; ci < t->cols.size(); ci++) {
            const TableColumn& c = t->cols[ci];
            if (std::find(req.begin(), req.end(), c.name) == req.end()) continue;
            if (!c.dptr) failInvalid("column " + c.name + " is needed by the plan but was declared without data");
            if (c.type.isString()) failUnsupported("string columns in a scan+aggregate pipeline are not built yet");
            int k = (int)pipe.cols.size();
            pipe.cols.push_back((int)ci);
            std::string ct = DeviceCodegen::ctype(c.type);
            std::string var = "v_" + std::to_string(k);
            cg.symbols[c.name] = Sym{var, c.type};
            addArg("c" + std::to_string(k), "const " + ct + "*", (uint64_t)(uintptr_t)c.dptr);
            colTypes.push_back(ct);
            rowParams += ", " + ct + " " + var;
            rowArgsTail += ", a.c" + std::to_string(k) + "[r]";
            pipe.bytesPerRow += columnWidth(c.type);
        }
        // scan output schema = requested attributes in table order
        for (int ci : pipe.cols) scan->schema.push_back({t->cols[ci].name, t->cols[ci].type});

        // ---- selections: cmp res, 0; je nextTuple (selection.h:52-70) ----
        std::string body;
        for (OpNode* s : selections) {
            q.pool.addId(s->exprs[0]);
            body += "        if (!(" + cg.emit(s->exprs[0]) + ")) return;\n";
        }

        // ---- aggregation (aggregation.h:240-295): group id, accumulator inputs ----
        planDenseKeys();
        for (Expr* g : agg->exprs2) q.pool.addId(g);
        for (Expr* s : agg->splitAgg) q.pool.addId(s);

        // accumulator 0: first input row of the group (drives the reference's emission order)
        q.accums.push_back(Accum{RSQ_E_MIN, "#firstrow", "row", Type(RSQ_BIGINT), 2 /*M_MIN_I64*/});
        for (Expr* s : agg->splitAgg) {
            Accum ac; ac.kind = s->tag; ac.type = s->type;
            switch (s->tag) {
                case RSQ_E_COUNT: ac.key = "COUNT"; ac.input = "((i64)1)"; ac.merge = 0; break;
                case RSQ_E_SUM:
                    if (s->type.tag != RSQ_DECIMAL && s->type.tag != RSQ_BIGINT) failType("ADD code generation not implemented for datatype");
                    ac.key = "SUM" + structuralKey(s->child); ac.input = cg.emit(s); ac.merge = 0; break;
                case RSQ_E_MIN: case RSQ_E_MAX:
                    if (s->type.tag != RSQ_DECIMAL && s->type.tag != RSQ_BIGINT && s->type.tag != RSQ_DATE)
                        failType("LESS_THAN code generation not implemented for datatype");
                    ac.key = std::string(s->tag == RSQ_E_MIN ? "MIN" : "MAX") + structuralKey(s->child);
                    ac.input = "((i64)(" + cg.emit(s) + "))"; ac.merge = s->tag == RSQ_E_MIN ? 2 : 3; break;
                default: failType("Aggregation type not implemented in updateAggregates(..).");
            }
            int found = -1;
            for (size_t i = 1; i < q.accums.size(); i++) if (q.accums[i].key == ac.key) found = (int)i;
            if (found < 0) { q.accums.push_back(ac); found = (int)q.accums.size() - 1; }
            q.splitToAccum.push_back(found);
        }
        // output blocks ordered [min | max | sum] so each segment reduces with one collective
        q.accumSlot.assign(q.accums.size(), 0);
        int slot = 0;
        for (int m : {2, 3, 0}) {
            for (size_t i = 0; i < q.accums.size(); i++) if (q.accums[i].merge == m) q.accumSlot[i] = slot++;
            if (m == 2) q.nMinBlocks = slot; else if (m == 3) q.nMaxBlocks = slot - q.nMinBlocks;
        }
        q.nSumBlocks = (int64_t)q.accums.size() - q.nMinBlocks - q.nMaxBlocks;

        const int D = (int)q.denseGroups, W = (int)q.accums.size();

        // group id
        std::string gid = "0";
        for (size_t ki = 0; ki < q.denseKeys.size(); ki++) {
            DenseKey& k = q.denseKeys[ki];
            std::string v = cg.emit(k.expr), rank;
            if (k.byteSet) {
                rank = "0";
                for (size_t d = 1; d < k.values.size(); d++) {
                    std::string an = "k" + std::to_string(ki) + "_" + std::to_string(d);
                    addArg(an, "u64", k.values[d]);
                    rank += " + (int)((u8)(" + v + ") >= (u8)a." + an + ")";
                }
            } else {
                std::string an = "k" + std::to_string(ki) + "_min";
                addArg(an, "i64", (uint64_t)k.min);
                rank = "(int)((i64)(" + v + ") - a." + an + ")";
            }
            gid += " + (" + rank + ") * " + std::to_string((long long)k.stride);
        }
        body += "        const int gid = " + gid + ";\n";
        for (int w = 1; w < W; w++) body += "        const i64 in" + std::to_string(w) + " = " + q.accums[w].input + ";\n";
        // Branch-free per-group update.  (An `if (gid == g) acc_g += x` chain gets its common tail sunk by the
        // compiler into one store through a selected pointer, which forces every accumulator into scratch.)
        for (int g = 0; g < D; g++) {
            body += "    { const bool m = gid == " + std::to_string(g) + ";\n";
            for (int w = 0; w < W; w++) {
                std::string acc = "st.acc_" + std::to_string(w) + "_" + std::to_string(g);
                std::string in = w == 0 ? "row" : "in" + std::to_string(w);
                if (q.accums[w].merge == 0) body += "      " + acc + " = rsq::add(" + acc + ", m ? " + in + " : (i64)0);\n";
                else if (q.accums[w].merge == 2) body += "      " + acc + " = (m && " + in + " < " + acc + ") ? " + in + " : " + acc + ";\n";
                else body += "      " + acc + " = (m && " + in + " > " + acc + ") ? " + in + " : " + acc + ";\n";
            }
            body += "    }\n";
        }

        addArg("n_rows", "i64", (uint64_t)t->nRows);
        addArg("row0", "i64", (uint64_t)t->row0);
        addArg("out", "u64*", 0);       // patched at execute
        addArg("err", "u32*", (uint64_t)(uintptr_t)q.ctx.dErr);

        // ---- assemble the kernel ----
        std::ostringstream s;
        s << "// generated by resql_amd/csrc/engine.cpp: scan -> selection -> aggregation (dense group id, register accumulators)\n";
        s << "#include \"rsq_device.h\"\n";
        const int unroll = unrollFactor();
        q.unroll = unroll;
        s << "#define U " << unroll << "\n";
        s << "struct Args {\n";
        for (auto& a : pipe.args) s << "    " << a.ctype << " " << a.name << ";\n";
        s << "};\n";
        s << "struct State {\n";
        for (int w = 0; w < W; w++)
            for (int g = 0; g < D; g++) {
                const char* init = q.accums[w].merge == 0 ? "0" : q.accums[w].merge == 2 ? "0x7fffffffffffffffll" : "(i64)0x8000000000000000ull";
                s << "    i64 acc_" << w << "_" << g << " = " << init << ";\n";
            }
        s << "};\n";
        s << "static RSQ_DEV void row_fn(const Args& a, State& st, i64 row" << rowParams << ") {\n" << body << "}\n";
        s << "extern \"C\" __global__ void __launch_bounds__(256) rsq_pipeline(Args a) {\n";
        s << "    State st;\n";
        s << "    const int lane = threadIdx.x & 63;\n";
        s << "    const i64 wave = (i64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);\n";
        s << "    const i64 nwaves = (i64)gridDim.x * (blockDim.x >> 6);\n";
        s << "    const i64 ntiles = a.n_rows >> 7;\n";
        // main loop, textually unrolled: all loads of U tiles are issued before the first row is processed
        s << "    for (i64 t = wave; t < ntiles; t += nwaves * U) {\n";
        const int ncols = (int)colTypes.size();
        for (int u = 0; u < unroll; u++) {
            s << "        const i64 tt" << u << " = t + " << u << " * nwaves;\n";
            for (int k = 0; k < ncols; k++) s << "        " << colTypes[k] << " t" << k << "_" << u << "[2];\n";
            s << "        if (tt" << u << " < ntiles) {\n            const i64 b = (tt" << u << " << 7) + lane * 2;\n";
            for (int k = 0; k < ncols; k++) s << "            rsq::ld2(a.c" << k << " + b, t" << k << "_" << u << ");\n";
            s << "        }\n";
        }
        for (int u = 0; u < unroll; u++) {
            s << "        if (tt" << u << " < ntiles) {\n";
            for (int j = 0; j < 2; j++) {
                s << "            row_fn(a, st, a.row0 + (tt" << u << " << 7) + lane * 2 + " << j;
                for (int k = 0; k < ncols; k++) s << ", t" << k << "_" << u << "[" << j << "]";
                s << ");\n";
            }
            s << "        }\n";
        }
        s << "    }\n";
        s << "    for (i64 r = (ntiles << 7) + (i64)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_rows; r += (i64)gridDim.x * blockDim.x)\n";
        s << "        row_fn(a, st, a.row0 + r" << rowArgsTail << ");\n";
        // epilogue: registers -> wave -> LDS -> global
        s << "    __shared__ u64 s_acc[" << W * D << "];\n";
        s << "    for (int i = threadIdx.x; i < " << W * D << "; i += blockDim.x) {\n        const int blk = i / " << D << ";\n";
        s << "        s_acc[i] = blk < " << q.nMinBlocks << " ? 0x7fffffffffffffffull : blk < " << (q.nMinBlocks + q.nMaxBlocks)
          << " ? 0x8000000000000000ull : 0ull;\n    }\n";
        s << "    __syncthreads();\n";
        for (int w = 0; w < W; w++)
            for (int g = 0; g < D; g++)
                s << "    rsq::wave_to_lds<" << q.accums[w].merge << ">(&s_acc[" << (q.accumSlot[w] * D + g) << "], (u64)st.acc_" << w << "_" << g << ");\n";
        s << "    __syncthreads();\n";
        s << "    for (int i = threadIdx.x; i < " << W * D << "; i += blockDim.x) {\n";
        s << "        const int blk = i / " << D << ";\n";
        s << "        if (blk < " << q.nMinBlocks << ") rsq::global_merge<2>(a.out + i, s_acc[i]);\n";
        s << "        else if (blk < " << (q.nMinBlocks + q.nMaxBlocks) << ") rsq::global_merge<3>(a.out + i, s_acc[i]);\n";
        s << "        else rsq::global_merge<0>(a.out + i, s_acc[i]);\n";
        s << "    }\n}\n";
        pipe.source = s.str();
        pipe.src = t;

        std::ostringstream ex;
        ex << "pipeline 0: scan " << t->name << " [" << t->nRows << " rows, " << pipe.bytesPerRow << " B/row]";
        for (OpNode* sel : selections) ex << " -> selection " << serializeExpr(sel->exprs[0]);
        ex << " -> aggregation dense groups=" << D << " accumulators=" << (W - 1) << " (of " << agg->splitAgg.size() << " in the reference)";
        pipe.explain = ex.str();
    }
};

// ================================================================================================
// host tail: from the aggregate table to the result relation
// ================================================================================================
struct Tail {
    Query& q;
    OpNode* agg;
    HostEval ev;

    std::vector<OpNode*> chainAbove() {   // operators between the aggregation and the root, bottom-up
        std::vector<OpNode*> v;
        for (OpNode* o = agg->parent; o; o = o->parent) v.push_back(o);
        return v;
    }

    void run() {
        const int64_t D = q.denseGroups;
        const int W = (int)q.accums.size();
        auto word = [&](int w, int64_t g) { return q.hAgg[(size_t)(q.accumSlot[w] * D + g)]; };

        // groups that exist, ordered by the first input row that produced them
        std::vector<int64_t> present;
        for (int64_t g = 0; g < D; g++) if ((int64_t)word(0, g) != INT64_MAX) present.push_back(g);
        std::sort(present.begin(), present.end(), [&](int64_t a, int64_t b) { return (int64_t)word(0, a) < (int64_t)word(0, b); });

        // decode group keys, hash them the reference's way, replay its hash table
        std::vector<std::vector<Val>> keys(present.size());
        std::vector<uint64_t> hashes(present.size());
        for (size_t i = 0; i < present.size(); i++) {
            uint64_t h = 0;
            for (auto& k : q.denseKeys) {
                int64_t rank = (present[i] / k.stride) % k.card;
                Val v; v.i = k.byteSet ? (int64_t)k.values[(size_t)rank] : k.min + rank;
                keys[i].push_back(v);
                h = refHashValue(h, v, k.type);
            }
            hashes[i] = h;
        }
        std::vector<size_t> order = refEmissionOrder(hashes, opSize(agg));

        // schemas above the aggregation (consume order)
        std::vector<OpNode*> above = chainAbove();
        Schema aggSchema;
        for (Expr* g : agg->exprs2) aggSchema.push_back({expressionName(g), g->type});
        {
            size_t si = 0;
            for (Expr* a : agg->exprs) {
                if (a->tag == RSQ_E_AVG) { q.pool.addId(a); aggSchema.push_back({expressionName(a), a->type}); si += 2; }
                else { aggSchema.push_back({expressionName(agg->splitAgg[si]), agg->splitAgg[si]->type}); si += 1; }
            }
        }
        agg->schema = aggSchema;

        OpNode* mat = nullptr; OpNode* orderBy = nullptr;
        std::vector<OpNode*> projections;
        for (OpNode* o : above) {
            if (o->tag == RSQ_OP_PROJECTION) { if (mat) failUnsupported("projection above materialize"); projections.push_back(o); }
            else if (o->tag == RSQ_OP_MATERIALIZE) { if (mat) failUnsupported("two materializations"); mat = o; }
            else if (o->tag == RSQ_OP_ORDERBY) orderBy = o;
            else failUnsupported("operator above an aggregation other than projection / materialize / order by");
        }
        if (!mat) failInvalid("plan has no materialization");

        // result schema: evaluate the consume chain once for names/types
        Schema cur = aggSchema;
        for (OpNode* p : projections) {
            Schema s;
            for (Expr* e : p->exprs) { q.pool.addId(e); s.push_back({expressionName(e), e->type}); }
            p->schema = s; cur = s;
        }
        mat->schema = cur;
        q.resultSchema = cur;
        const size_t ts = (size_t)schemaTupleSize(cur);
        std::vector<int> offs; for (auto& a : cur) offs.push_back(schemaOffset(cur, a.name));

        q.resultTuples.clear(); q.resultRows = 0;
        for (size_t oi : order) {
            int64_t g = present[oi];
            ev.symbols.clear();
            // dematerialize the entry: group keys, then aggregates with AVG = (sum * 100) / count
            for (size_t k = 0; k < agg->exprs2.size(); k++) ev.symbols[expressionName(agg->exprs2[k])] = {keys[oi][k], agg->exprs2[k]->type};
            size_t si = 0;
            for (Expr* a : agg->exprs) {
                if (a->tag == RSQ_E_AVG) {
                    Type st = agg->splitAgg[si]->type;
                    if (st.tag != RSQ_BIGINT && st.tag != RSQ_DECIMAL) failType("getAvgFromSumAndCount(..) not supported for datatype");
                    int64_t sum = (int64_t)word(q.splitToAccum[si], g), cnt = (int64_t)word(q.splitToAccum[si + 1], g);
                    Val v; v.i = HostEval::sdiv((int64_t)((uint64_t)sum * 100ull), cnt);
                    ev.symbols[expressionName(a)] = {v, a->type};
                    si += 2;
                } else {
                    Val v; v.i = (int64_t)word(q.splitToAccum[si], g);
                    ev.symbols[expressionName(agg->splitAgg[si])] = {v, agg->splitAgg[si]->type};
                    si += 1;
                }
            }
            for (OpNode* p : projections) {
                std::vector<std::pair<std::string, std::pair<Val, Type>>> vals;
                for (Expr* e : p->exprs) vals.push_back({expressionName(e), {ev.eval(e), e->type}});
                for (auto& kv : vals) ev.symbols[kv.first] = kv.second;
            }
            size_t base = q.resultTuples.size();
            q.resultTuples.resize(base + ts, 0);
            for (size_t c = 0; c < cur.size(); c++) {
                auto it = ev.symbols.find(cur[c].name);
                if (it == ev.symbols.end()) failType("materialize: symbol " + cur[c].name + " not found");
                storeValue(&q.resultTuples[base + (size_t)offs[c]], it->second.first, cur[c].type);
            }
            q.resultRows++;
            if (mat->hasLimit && q.resultRows >= mat->limit) break;      // materialize.h:197-206
        }

        if (orderBy) {
            std::vector<OrderRequest> reqs;
            for (Expr* e : orderBy->exprs) {
                const std::string& n = e->child->symbol;
                bool found = false;
                for (auto& a : cur) if (a.name == n) { reqs.push_back({schemaOffset(cur, n), a.type, e->tag != RSQ_E_DESC}); found = true; break; }
                if (!found) failType("Order By attribute not found.");
            }
            refQuicksort(q.resultTuples.data(), q.resultRows, ts, reqs);
            if (orderBy->hasLimit && q.resultRows > orderBy->limit) {      // applyLimit after the sort (orderby.h:87-93)
                q.resultRows = orderBy->limit;
                q.resultTuples.resize((size_t)q.resultRows * ts);
            }
        }
        (void)W;
    }
};

OpNode* findAggregation(OpNode* root) {
    OpNode* o = root;
    while (o && o->tag != RSQ_OP_AGGREGATION) {
        if (o->nChildren != 1) failUnsupported("plan shape: expected a single chain above the aggregation");
        o = o->child[0];
    }
    return o;
}

}  // namespace

// ================================================================================================
// public (engine.h)
// ================================================================================================
static void prepareAggBuffers(Query& q);

Query* compileQuery(Context& ctx, const rsq_plan_desc& plan, rsq_table* const* tables, int nTables) {
    double t0 = nowMs();
    std::unique_ptr<Query> q(new Query(ctx));
    int hits0 = ctx.jitCacheHits, comp0 = ctx.jitCompiles;
    for (int i = 0; i < nTables; i++) {
        Table* t = reinterpret_cast<Table*>(tables[i]);
        if (!t) failInvalid("null table");
        q->tables.push_back(t);
        for (auto& c : t->cols) q->pool.identTypes[c.name] = c.type;
    }
    q->requestAll = plan.request_all != 0;
    q->exprs = q->pool.build(plan);
    buildOps(*q, plan);
    defineAndDerive(*q, q->root);

    q->agg = findAggregation(q->root);
    if (!q->agg) failUnsupported("plans without an aggregation are not built yet in this engine version");
    ScanAggBuilder b(*q, q->agg);
    b.build();
    q->pipelines.push_back(b.pipe);
    for (auto& p : q->pipelines) {
        p.kernel = &ctx.getKernel(p.source, p.entry);
        q->allSource += p.source + "\n";
        q->explainText += p.explain + "\n";
    }
    if (ctx.device >= 0) {
        size_t words = (size_t)q->accums.size() * (size_t)q->denseGroups;
        q->dAgg = (uint64_t*)ctx.alloc(words * 8);
        q->hAgg.assign(words, 0);
        prepareAggBuffers(*q);
    }
    q->report.compilation_time_ms = nowMs() - t0;
    q->report.jit_cache_hits = ctx.jitCacheHits - hits0;
    q->report.jit_compiles = ctx.jitCompiles - comp0;
    if (ctx.cfg.print_assembly) fprintf(stderr, "%s\n", q->allSource.c_str());
    if (ctx.cfg.print_flounder) fprintf(stderr, "%s\n", q->explainText.c_str());
    return q.release();
}

// Device-resident identity image of the aggregate table (0 for sums, +/-inf for min/max) and a pinned
// host buffer for the read-back: one execute is then {D2D init, kernel, D2H} on one stream with a single
// host synchronisation at the end.
static void prepareAggBuffers(Query& q) {
    const int64_t D = q.denseGroups;
    const size_t words = q.accums.size() * (size_t)D;
    std::vector<uint64_t> init(words);
    for (size_t w = 0; w < q.accums.size(); w++) {
        uint64_t idv = q.accums[w].merge == 0 ? 0ull : q.accums[w].merge == 2 ? 0x7fffffffffffffffull : 0x8000000000000000ull;
        for (int64_t g = 0; g < D; g++) init[(size_t)(q.accumSlot[w] * D + g)] = idv;
    }
    q.dAggInit = (uint64_t*)q.ctx.alloc(words * 8);
    RSQ_HIP(hipMemcpy(q.dAggInit, init.data(), words * 8, hipMemcpyHostToDevice));
    RSQ_HIP(hipHostMalloc((void**)&q.hPinned, words * 8 + 8, hipHostMallocDefault));
}

void executeQuery(Query& q, bool partialOnly) {
    Context& ctx = q.ctx;
    if (ctx.device < 0) throw Error(RSQ_ERR_DEVICE, "this context has no device (compile-only)");
    RSQ_HIP(hipSetDevice(ctx.device));
    double t0 = nowMs();
    const size_t words = q.hAgg.size();
    RSQ_HIP(hipMemcpyAsync(q.dAgg, q.dAggInit, words * 8, hipMemcpyDeviceToDevice, ctx.stream));
    RSQ_HIP(hipMemsetAsync(ctx.dErr, 0, 4, ctx.stream));
    q.report.num_kernels = 0; q.report.bytes_read = 0;
    RSQ_HIP(hipEventRecord(ctx.ev0, ctx.stream));
    for (auto& p : q.pipelines) {
        std::vector<uint64_t> args;
        for (auto& a : p.args) args.push_back(a.name == "out" ? (uint64_t)(uintptr_t)q.dAgg : a.value);
        int64_t tiles = p.src->nRows >> 7;
        unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(256 * 8, (tiles + 4 * q.unroll - 1) / (4 * q.unroll)));
        launch(ctx, *p.kernel, grid, 256, args);
        q.report.num_kernels++;
        q.report.bytes_read += (uint64_t)(p.bytesPerRow * p.src->nRows);
    }
    RSQ_HIP(hipEventRecord(ctx.ev1, ctx.stream));
    RSQ_HIP(hipMemcpyAsync(q.hPinned + words, ctx.dErr, 4, hipMemcpyDeviceToHost, ctx.stream));
    if (!partialOnly) RSQ_HIP(hipMemcpyAsync(q.hPinned, q.dAgg, words * 8, hipMemcpyDeviceToHost, ctx.stream));
    RSQ_HIP(hipStreamSynchronize(ctx.stream));
    float ms = 0; RSQ_HIP(hipEventElapsedTime(&ms, ctx.ev0, ctx.ev1));
    q.report.kernel_time_ms = ms;
    q.report.hbm_gbps = ms > 0 ? (double)q.report.bytes_read / (ms * 1e-3) / 1e9 : 0;
    uint32_t err = (uint32_t)q.hPinned[words];
    if (err & 1) failRuntime("Division by zero");
    if (err) failRuntime("device error word " + std::to_string(err));
    if (!partialOnly) {
        double t1 = nowMs();
        memcpy(q.hAgg.data(), q.hPinned, words * 8);
        Tail tail{q, q.agg, {}};
        tail.run();
        q.report.finalize_time_ms = nowMs() - t1;
    }
    q.report.execution_time_ms = nowMs() - t0;
}

void finalizeQuery(Query& q) {
    Context& ctx = q.ctx;
    RSQ_HIP(hipSetDevice(ctx.device));
    double t1 = nowMs();
    RSQ_HIP(hipMemcpyAsync(q.hAgg.data(), q.dAgg, q.hAgg.size() * 8, hipMemcpyDeviceToHost, ctx.stream));
    RSQ_HIP(hipStreamSynchronize(ctx.stream));
    Tail tail{q, q.agg, {}};
    tail.run();
    q.report.finalize_time_ms = nowMs() - t1;
}

void bindPartial(Query& q, void* dptr, size_t bytes) {
    size_t need = q.accums.size() * (size_t)q.denseGroups * 8;
    if (!dptr || bytes < need) failInvalid("partial buffer too small: need " + std::to_string(need) + " bytes");
    if (q.dAgg && q.dAggOwned) q.ctx.free(q.dAgg);
    q.dAgg = (uint64_t*)dptr;
    q.dAggOwned = false;
}

void partialBuffer(Query& q, void** dptr, int64_t* nMin, int64_t* nMax, int64_t* nSum) {
    *dptr = q.dAgg;
    *nMin = q.nMinBlocks * q.denseGroups;
    *nMax = q.nMaxBlocks * q.denseGroups;
    *nSum = q.nSumBlocks * q.denseGroups;
}

void queryResult(Query& q, rsq_result_view* out) {
    size_t n = q.resultSchema.size();
    q.rvNames.assign(n * RSQ_SYMBOL_MAX + 1, 0);
    q.rvTypes.resize(n); q.rvOffsets.resize(n);
    int off = 0;
    for (size_t i = 0; i < n; i++) {
        snprintf(&q.rvNames[i * RSQ_SYMBOL_MAX], RSQ_SYMBOL_MAX, "%s", q.resultSchema[i].name.c_str());
        q.rvTypes[i] = q.resultSchema[i].type.toC();
        q.rvOffsets[i] = off;
        off += sizeInTuple(q.resultSchema[i].type, true);
    }
    out->n_cols = (int32_t)n;
    out->names = reinterpret_cast<const char(*)[RSQ_SYMBOL_MAX]>(q.rvNames.data());
    out->types = q.rvTypes.data();
    out->offsets = q.rvOffsets.data();
    out->tuple_size = off;
    out->n_rows = q.resultRows;
    out->tuples = q.resultTuples.data();
}

void queryReport(const Query& q, rsq_report* out) { *out = q.report; }
const char* querySource(const Query& q) { return q.allSource.c_str(); }
const char* queryExplain(const Query& q) { return q.explainText.c_str(); }
void destroyQuery(Query* q) { delete q; }

std::string serializeResultView(const rsq_result_view& v) {
    std::string out;
    for (int64_t r = 0; r < v.n_rows; r++) {
        const uint8_t* t = v.tuples + (size_t)r * (size_t)v.tuple_size;
        for (int c = 0; c < v.n_cols; c++) {
            Type ty = Type::fromC(v.types[c]);
            out += serializeSqlValue(loadValue(t + v.offsets[c], ty), ty) + "|";
        }
        out += "\n";
    }
    return out;
}

}  // namespace rsq
