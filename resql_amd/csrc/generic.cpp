// generic.cpp — the host side of the pre-compiled generic pipeline: a plan's scan -> selection -> dense aggregation becomes a
// register program for the AOT interpreter kernel (aot_kernels.hip k_generic_aggregate).
//
// Why it exists: ReSQL is "a query compilation-based database system with low compilation times" — its Flounder back end
// turns a plan into machine code in 0.6-3 ms (reference README:1-6, src/JitContextFlounder.h:410-456).  hiprtc needs
// 0.4-1.3 s for a pipeline shape it has not seen.  A plan shape that is not in the code-object cache therefore starts on this
// interpreter at once, while hiprtc builds the specialised kernel on a host thread; the next execution after that switches.
// The program reproduces the typed semantics codegen.cpp's ExprGen emits as C++ (64-bit wrap-around arithmetic, truncating
// division with the /0 flag, byte-wise AND / OR, CASE as nested selects, TYPECAST as scale multiplications / divisions), the
// aggregate table has the specialised kernels' layout ([block][group] words), so the host tail — and the answer, byte for
// byte — is the same.
//
// Eligible: one pipeline that scans numeric / date / CHAR(1) / BOOL columns into a dense aggregation.  Joins, strings,
// materialisation and hash aggregation wait for their specialised kernels (the compile blocks, as before).
#include <cstring>

#include "engine_internal.h"

namespace rsq {

namespace {

struct Builder {
    Query& q;
    GenericProgram& p;
    Pipeline& pipe;
    std::map<std::string, int> colReg;       // column name -> register (loaded once, kept)
    std::vector<bool> used = std::vector<bool>(G_REGS, false);
    std::string why;

    Builder(Query& q_, GenericProgram& p_) : q(q_), p(p_), pipe(q_.pipelines[0]) {}

    [[noreturn]] void no(const std::string& m) { throw Error(RSQ_ERR_UNSUPPORTED, m); }
    int alloc() {
        for (int r = 0; r < G_REGS; r++) if (!used[(size_t)r]) { used[(size_t)r] = true; return r; }
        no("expression needs more than " + std::to_string(G_REGS) + " registers");
    }
    void release(int r, bool temp) { if (temp && r >= 0) used[(size_t)r] = false; }
    void emit(uint8_t op, int dst, int a = 0, int b = 0, uint32_t c = 0, int64_t imm = 0) {
        if (p.code.size() >= G_MAX_INSTR) no("program too long");
        p.code.push_back(GenericInstr{op, (uint8_t)dst, (uint8_t)a, (uint8_t)b, c, imm});
    }
    static int64_t pow10(int n) { int64_t v = 1; while (n-- > 0) v *= 10; return v; }

    // returns (register, is a temporary the caller may release)
    std::pair<int, bool> gen(Expr* e) {
        if (e->type.tag == RSQ_NT) failType("Expression type undefined in emitExpression(..). Have you derived the expression types?");
        if (e->type.isString()) no("string values");
        switch (e->structure) {
            case LITERAL: {
                if (e->tag == RSQ_E_ATTRIBUTE) {
                    auto it = colReg.find(e->symbol);
                    if (it != colReg.end()) return {it->second, false};
                    int ci = pipe.src->findCol(e->symbol);
                    if (ci < 0 || !pipe.src->cols[(size_t)ci].dptr) no("attribute " + e->symbol + " is not a scanned column");
                    const TableColumn& c = pipe.src->cols[(size_t)ci];
                    int kind = 0;
                    switch (c.type.tag) {
                        case RSQ_BIGINT: case RSQ_DECIMAL: kind = 3; break;
                        case RSQ_INT: case RSQ_DATE: kind = 2; break;
                        case RSQ_BOOL: kind = 1; break;
                        case RSQ_CHAR: if (c.type.len == 1) { kind = 1; break; } [[fallthrough]];
                        default: no("column type");
                    }
                    if (p.cols.size() >= G_MAX_COLS) no("too many columns");
                    const int r = alloc();
                    emit(G_COL, r, (int)p.cols.size());
                    p.cols.push_back({c.dptr, kind});
                    colReg[e->symbol] = r;
                    return {r, false};
                }
                if (e->tag == RSQ_E_CONSTANT) {
                    int64_t v = e->ival;
                    if (e->type.tag == RSQ_INT || e->type.tag == RSQ_DATE) v = (int64_t)(int32_t)v;
                    else if (e->type.tag == RSQ_BOOL || e->type.tag == RSQ_CHAR) v = (int64_t)(uint8_t)v;
                    const int r = alloc();
                    emit(G_CONST, r, 0, 0, 0, v);
                    return {r, true};
                }
                if (e->tag == RSQ_E_STAR) { const int r = alloc(); emit(G_CONST, r, 0, 0, 0, 0); return {r, true}; }
                no("literal kind");
            }
            case UNARY: {
                if (e->tag == RSQ_E_COUNT) { const int r = alloc(); emit(G_CONST, r, 0, 0, 0, 1); return {r, true}; }
                auto c = gen(e->child);
                const Type from = e->child->type, to = e->type;
                switch (e->tag) {
                    case RSQ_E_SUM: case RSQ_E_AVG: case RSQ_E_MIN: case RSQ_E_MAX: case RSQ_E_AS: return c;
                    case RSQ_E_TYPECAST: {
                        int64_t mul = 0, div = 0;
                        if (to.tag == RSQ_DECIMAL) {
                            if (from.tag == RSQ_DECIMAL) {
                                const int d = to.scale - from.scale;
                                if (d == 0) return c;
                                if (d > 8 || d < -8) failType("typecast beyond the supported scale difference");
                                if (d > 0) mul = pow10(d); else div = pow10(-d);
                            } else if (from.tag == RSQ_BIGINT) {
                                if (to.scale > 8) failType("typecast beyond the supported scale");
                                mul = pow10(to.scale);
                            } else failType("emitTypecastToDECIMAL(..) code generation not implemented for datatype");
                        } else if (to.tag == RSQ_BIGINT) {
                            if (from.tag == RSQ_INT) {
                                if (jitInt16Cast(q.ctx)) no("the 16-bit cast switch");
                                return c;                      // values travel sign-extended to 64 bits already
                            }
                            if (from.tag == RSQ_BIGINT) return c;
                            if (from.tag == RSQ_DECIMAL) { if (from.scale > 8) failType("typecast beyond the supported scale"); div = pow10(from.scale); }
                            else failType("emitTypecastToBIGINT(..) code generation not implemented for datatype");
                        } else failType("emitTypecast(..) code generation not implemented for datatype");
                        const int r = alloc();
                        emit(mul ? G_MULI : G_DIVI, r, c.first, 0, 0, mul ? mul : div);
                        release(c.first, c.second);
                        return {r, true};
                    }
                    default: no("unary operator");
                }
            }
            case BINARY: {
                const Type res = e->type, op = e->child->type;
                uint8_t code = 0;
                bool negate = false;
                auto arith = [&] { if (res.tag != RSQ_DECIMAL && res.tag != RSQ_BIGINT) failType(std::string(exprTagNames[e->tag]) + " code generation not implemented for datatype"); };
                auto ordered = [&] { if (op.tag != RSQ_DECIMAL && op.tag != RSQ_DATE && op.tag != RSQ_BIGINT) failType(std::string(exprTagNames[e->tag]) + " code generation not implemented for datatype"); };
                switch (e->tag) {
                    case RSQ_E_ADD: arith(); code = G_ADD; break;
                    case RSQ_E_SUB: arith(); code = G_SUB; break;
                    case RSQ_E_MUL: arith(); code = G_MUL; break;
                    case RSQ_E_DIV: arith(); code = G_DIV; break;
                    case RSQ_E_AND: code = G_AND; break;
                    case RSQ_E_OR: code = G_OR; break;
                    case RSQ_E_LT: ordered(); code = G_LT; break;
                    case RSQ_E_LE: ordered(); code = G_LE; break;
                    case RSQ_E_GT: ordered(); code = G_GT; break;
                    case RSQ_E_GE: ordered(); code = G_GE; break;
                    case RSQ_E_EQ: case RSQ_E_NEQ:
                        if (op.isString() || e->child->next->type.isString()) no("string comparison");
                        code = e->tag == RSQ_E_EQ ? G_EQ : G_NE; break;
                    default: no("binary operator");
                }
                (void)negate;
                auto l = gen(e->child);
                auto r = gen(e->child->next);
                const int d = alloc();
                emit(code, d, l.first, r.first);
                release(l.first, l.second); release(r.first, r.second);
                return {d, true};
            }
            case OTHER: {       // CASE: nested selects, innermost first; no ELSE: 0 (ExpressionsJitFlounder.h:720-754)
                std::vector<std::pair<Expr*, Expr*>> whens;
                Expr* c = e->child;
                for (; c && c->tag == RSQ_E_WHENTHEN; c = c->next) whens.push_back({c->child, c->child->next});
                std::pair<int, bool> acc;
                if (c) acc = gen(c); else { const int r = alloc(); emit(G_CONST, r, 0, 0, 0, 0); acc = {r, true}; }
                for (size_t i = whens.size(); i-- > 0;) {
                    auto cond = gen(whens[i].first);
                    auto val = gen(whens[i].second);
                    const int d = alloc();
                    emit(G_SELECT, d, cond.first, val.first, (uint32_t)acc.first);
                    release(cond.first, cond.second); release(val.first, val.second); release(acc.first, acc.second);
                    acc = {d, true};
                }
                return acc;
            }
            default: no("expression structure");
        }
    }

    // a value that must stay until the end of the row (group keys, accumulator inputs): never released
    int pinned(Expr* e) { return gen(e).first; }
};

}  // namespace

bool buildGenericProgram(Query& q, GenericProgram& out, std::string& why) {
    out = GenericProgram();
    // (the two structural tests without an exception: the first C++ exception of a process walks the unwind tables of every loaded
    // library — 80 ms with the ROCm stack loaded, which a cold TPC-H Q3 then paid on its way to the whole-pipeline interpreter)
    if (q.pipelines.size() != 1 || q.pipelines[0].sink != SinkKind::AGGREGATE || !q.agg) { why = "not a single scan -> aggregation pipeline"; return false; }
    if (!(q.aggMode == AggMode::DENSE_REG || q.aggMode == AggMode::DENSE_LDS_PRIVATE || q.aggMode == AggMode::DENSE_LDS_SHARED || q.aggMode == AggMode::DENSE_GLOBAL)) {
        why = "not a dense aggregation"; return false;
    }
    try {
        if (q.denseGroups * (int64_t)q.accums.size() > ((int64_t)1 << 27)) throw Error(RSQ_ERR_UNSUPPORTED, "aggregate table too large");
        Builder b(q, out);
        // the operators between the scan and the aggregation: selections only
        std::vector<OpNode*> chain;
        for (OpNode* o = q.agg->child[0]; o; o = o->nChildren ? o->child[0] : nullptr) {
            if (o->tag == RSQ_OP_SELECTION) chain.push_back(o);
            else if (o->tag == RSQ_OP_SCAN) break;
            else throw Error(RSQ_ERR_UNSUPPORTED, "an operator other than a selection between scan and aggregation");
        }
        for (size_t i = chain.size(); i-- > 0;) {           // scan order: the selection next to the scan first
            auto c = b.gen(chain[i]->exprs[0]);
            b.emit(G_FILTER, 0, c.first);
            b.release(c.first, c.second);
        }
        if (q.denseKeys.size() > G_MAX_KEYS) throw Error(RSQ_ERR_UNSUPPORTED, "too many group keys");
        for (auto& k : q.denseKeys) {
            GenericProgram::Key gk{};
            gk.reg = b.pinned(k.expr);
            gk.byteSet = k.byteSet ? 1 : 0; gk.min = k.min; gk.card = k.card; gk.stride = k.stride;
            if (k.byteSet) {
                if (k.values.size() > G_MAX_SET) throw Error(RSQ_ERR_UNSUPPORTED, "byte set too large");
                gk.nValues = (int)k.values.size();
                for (size_t i = 0; i < k.values.size(); i++) gk.values[i] = k.values[i];
            }
            out.keys.push_back(gk);
        }
        if (q.accums.size() > G_MAX_ACCS) throw Error(RSQ_ERR_UNSUPPORTED, "too many accumulators");
        for (size_t w = 0; w < q.accums.size(); w++) {
            GenericProgram::Acc a{};
            a.merge = q.accums[w].merge;
            a.block = q.accumSlot[w];
            if (w == 0) a.reg = -1;
            else if (q.accums[w].kind == RSQ_E_COUNT) a.reg = -2;
            else { if (!q.accums[w].inputExpr) throw Error(RSQ_ERR_UNSUPPORTED, "accumulator without an input expression"); a.reg = b.pinned(q.accums[w].inputExpr); }
            out.accs.push_back(a);
        }
        return true;
    } catch (const Error& e) {
        if (e.status != RSQ_ERR_UNSUPPORTED) throw;        // a typing error is the plan's, whoever compiles it
        why = e.what();
        return false;
    }
}

}  // namespace rsq
