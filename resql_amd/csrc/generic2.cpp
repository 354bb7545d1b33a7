// generic2.cpp — register programs for the interpreter of whole pipelines (generic_kernels.hip k_generic_pipeline).
//
// buildPipelines (codegen.cpp) has already walked the operator tree the way the reference's produce / consume does
// (reference src/operators/RelOperator.h:160-189): it left one Pipeline per scan, the join tables' layouts (key words, payload
// attributes, key aliases), the aggregation strategy, the materialisation schema.  This file walks the same tree once more and
// turns every pipeline into a program: column loads, the typed expressions of selections and projections, probes of the join
// tables built by earlier pipelines, and the sink — with the same types, word layouts and group-row formats, so that everything
// behind the pipelines (entry compaction, candidate selection, the host tail) runs unchanged and the answer is the same bytes.
//
// Not interpreted (the specialised kernels are compiled before the first execution, as before): string join keys, string-valued
// CASE / TYPECAST results, more than G2_REGS live values, more than G2_MAX_DEPTH probes for all matches in one pipeline.
#include <cstring>
#include <functional>

#include "engine_internal.h"

namespace rsq {

namespace {

struct Val2 { int reg; bool temp; };

struct Builder2 {
    Query& q;
    GenericProgram2& p;
    Pipeline& pipe;
    std::map<std::string, std::pair<int, Type>> sym;     // value available under this name -> (register, type)
    std::vector<bool> used = std::vector<bool>(G2_REGS, false);
    int multiProbes = 0;

    Builder2(Query& q_, GenericProgram2& p_, Pipeline& pipe_) : q(q_), p(p_), pipe(pipe_) {}

    [[noreturn]] void no(const std::string& m) { throw Error(RSQ_ERR_UNSUPPORTED, m); }
    int alloc() {
        for (int r = 0; r < G2_REGS; r++) if (!used[(size_t)r]) { used[(size_t)r] = true; p.nRegs = std::max(p.nRegs, r + 1); return r; }
        no("more than " + std::to_string((int)G2_REGS) + " live values");
    }
    void release(const Val2& v) { if (v.temp && v.reg >= 0) used[(size_t)v.reg] = false; }
    void emit(uint8_t op, int dst, int a = 0, int b = 0, uint32_t c = 0, int64_t imm = 0) {
        if (p.code.size() >= 250) no("program too long");
        p.code.push_back(GenericInstr{op, (uint8_t)dst, (uint8_t)a, (uint8_t)b, c, imm});
    }
    static int64_t pow10(int n) { int64_t v = 1; while (n-- > 0) v *= 10; return v; }

    int column(const std::string& name, Type& typeOut) {
        const int ci = pipe.src->findCol(name);
        if (ci < 0) failType("attribute " + name + " is not available in this pipeline");
        const TableColumn& c = pipe.src->cols[(size_t)ci];
        if (!c.dptr) failInvalid("column " + name + " is needed by the plan but was declared without data");
        if (p.cols.size() >= G2_MAX_COLS) no("too many columns");
        const int r = alloc();
        emit(c.type.isString() ? (uint8_t)G_COLADDR : (uint8_t)G_COL, r, (int)p.cols.size());
        p.cols.push_back({c.dptr, columnWidth(c.type)});
        sym[name] = {r, c.type};
        typeOut = c.type;
        return r;
    }

    Val2 gen(Expr* e) {
        if (e->type.tag == RSQ_NT) failType("Expression type undefined in emitExpression(..). Have you derived the expression types?");
        auto it = sym.find(expressionName(e));
        if (it != sym.end()) return {it->second.first, false};
        switch (e->structure) {
            case LITERAL: {
                if (e->tag == RSQ_E_ATTRIBUTE) { Type t; return {column(e->symbol, t), false}; }
                if (e->tag == RSQ_E_CONSTANT) {
                    const int r = alloc();
                    if (e->type.isString()) {
                        // the constant's text, NUL padded to its declared length (+ a terminator)
                        const size_t off = p.constPool.size();
                        std::string text = e->symbol.substr(0, (size_t)std::max(0, e->type.len));
                        p.constPool.insert(p.constPool.end(), text.begin(), text.end());
                        p.constPool.insert(p.constPool.end(), (size_t)e->type.len + 1 - text.size(), '\0');
                        emit(G_CONSTADDR, r, 0, 0, 0, (int64_t)off);
                        return {r, true};
                    }
                    int64_t v = e->ival;
                    if (e->type.tag == RSQ_INT || e->type.tag == RSQ_DATE) v = (int64_t)(int32_t)v;
                    else if (e->type.tag == RSQ_BOOL || e->type.tag == RSQ_CHAR) v = (int64_t)(uint8_t)v;
                    emit(G_CONST, r, 0, 0, 0, v);
                    return {r, true};
                }
                if (e->tag == RSQ_E_STAR) { const int r = alloc(); emit(G_CONST, r, 0, 0, 0, 0); return {r, true}; }
                failType(std::string("emitExpressionLiteral(..) not implemented for expression type") + exprTagNames[e->tag]);
            }
            case UNARY: {
                if (e->tag == RSQ_E_COUNT) {
                    if (e->child && e->child->tag != RSQ_E_STAR) release(gen(e->child));       // (the reference emits the argument for its checks)
                    const int r = alloc(); emit(G_CONST, r, 0, 0, 0, 1); return {r, true};
                }
                Val2 c = gen(e->child);
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
                                if (!jitInt16Cast(q.ctx)) return c;      // values travel sign-extended already
                                const int r = alloc(); emit(G_CAST16, r, c.reg); release(c); return {r, true};
                            }
                            if (from.tag == RSQ_BIGINT) return c;
                            if (from.tag == RSQ_DECIMAL) { if (from.scale > 8) failType("typecast beyond the supported scale"); div = pow10(from.scale); }
                            else failType("emitTypecastToBIGINT(..) code generation not implemented for datatype");
                        } else failType("emitTypecast(..) code generation not implemented for datatype");
                        const int r = alloc();
                        emit(mul ? G_MULI : G_DIVI, r, c.reg, 0, 0, mul ? mul : div);
                        release(c);
                        return {r, true};
                    }
                    default: failType(std::string("emitExpression(..) not implemented for expression type") + exprTagNames[e->tag]);
                }
            }
            case BINARY: {
                const Type res = e->type, op = e->child->type, rt = e->child->next->type;
                auto arith = [&] { if (res.tag != RSQ_DECIMAL && res.tag != RSQ_BIGINT) failType(std::string(exprTagNames[e->tag]) + " code generation not implemented for datatype"); };
                auto ordered = [&] { if (op.tag != RSQ_DECIMAL && op.tag != RSQ_DATE && op.tag != RSQ_BIGINT) failType(std::string(exprTagNames[e->tag]) + " code generation not implemented for datatype"); };
                uint8_t code = 0; int64_t imm = 0;
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
                        if (op.isString()) {
                            // the LEFT operand's type picks the comparison (ExprGen::emitBinary): CHAR(n) ignores trailing spaces
                            if (!rt.isString()) failType("EQUALS code generation not implemented for datatype");
                            code = G_STREQ;
                            imm = (int64_t)(op.len & 0xffff) | ((int64_t)(rt.len & 0xffff) << 16) | ((int64_t)(op.tag == RSQ_CHAR ? 1 : 0) << 32) |
                                  ((int64_t)(e->tag == RSQ_E_NEQ ? 1 : 0) << 33);
                        } else {
                            switch (op.tag) {
                                case RSQ_DECIMAL: case RSQ_INT: case RSQ_BIGINT: case RSQ_BOOL: case RSQ_DATE: case RSQ_CHAR: break;
                                default: failType("EQUALS code generation not implemented for datatype");
                            }
                            if (rt.isString()) failType("EQUALS code generation not implemented for datatype");
                            code = e->tag == RSQ_E_EQ ? G_EQ : G_NE;
                        }
                        break;
                    case RSQ_E_LIKE:
                        if (!op.isString() || !rt.isString()) failType("LIKE on a CHAR(1) operand is undefined in the reference");
                        code = G_LIKE; imm = (int64_t)(op.len & 0xffff) | ((int64_t)(rt.len & 0xffff) << 16);
                        break;
                    default: failType(std::string("emitExpressionBinary(..) not implemented for expression type") + exprTagNames[e->tag]);
                }
                Val2 l = gen(e->child), r = gen(e->child->next);
                const int d = alloc();
                emit(code, d, l.reg, r.reg, 0, imm);
                release(l); release(r);
                return {d, true};
            }
            case OTHER: {       // CASE: nested selects, innermost first; no ELSE: 0 (ExpressionsJitFlounder.h:720-754)
                if (e->type.isString()) no("string-valued CASE");
                std::vector<std::pair<Expr*, Expr*>> whens;
                Expr* c = e->child;
                for (; c && c->tag == RSQ_E_WHENTHEN; c = c->next) whens.push_back({c->child, c->child->next});
                Val2 acc;
                if (c) acc = gen(c); else { const int r = alloc(); emit(G_CONST, r, 0, 0, 0, 0); acc = {r, true}; }
                for (size_t i = whens.size(); i-- > 0;) {
                    Val2 cond = gen(whens[i].first), val = gen(whens[i].second);
                    const int d = alloc();
                    emit(G_SELECT, d, cond.reg, val.reg, (uint32_t)acc.reg);
                    release(cond); release(val); release(acc);
                    acc = {d, true};
                }
                return acc;
            }
            default: failType("emitExpression(..)");
        }
    }

    // a value that must survive to the end of the row (and every backtrack): its register is never handed out again
    int pinned(Expr* e) { return gen(e).reg; }
    // ... in a register of its own (a value that is another symbol's register would be shared: fine, registers are written once per row)
    void define(const std::string& name, int reg, const Type& t) { sym[name] = {reg, t}; }

    // the words of a string value (its bytes, NUL padded — the columns' contract), for group keys
    void stringWords(int addrReg, const Type& t, std::vector<int>& out) {
        for (int w = 0; w * 8 < t.len; w++) {
            const int n = std::min(8, t.len - w * 8);
            const int r = alloc();
            emit(G_STRWORD, r, addrReg, 0, 0, (int64_t)(w * 8) | ((int64_t)n << 16));
            out.push_back(r);
        }
    }
};

void collectScans(OpNode* o, std::vector<OpNode*>& out) {
    if (o->tag == RSQ_OP_SCAN) { out.push_back(o); return; }
    for (int i = 0; i < o->nChildren; i++) collectScans(o->child[i], out);        // a join's build side first, like produce()
}

void buildOne(Query& q, size_t pi, OpNode* scan, GenericProgram2& prog) {
    Pipeline& pipe = q.pipelines[pi];
    Builder2 b(q, prog, pipe);
    GenericSinkDesc& S = prog.sink;
    memset(&S, 0, sizeof S);
    S.slotReg = -1;
    std::map<int, int> slotRegOf;            // join table -> register holding the matched slot
    OpNode* from = scan;
    for (OpNode* o = scan->parent; ; from = o, o = o->parent) {
        if (!o) throw Error(RSQ_ERR_UNSUPPORTED, "pipeline without a sink");
        if (o->tag == RSQ_OP_SELECTION) {
            Val2 c = b.gen(o->exprs[0]);
            b.emit(G_FILTER, 0, c.reg);
            b.release(c);
            continue;
        }
        if (o->tag == RSQ_OP_PROJECTION) {
            std::vector<std::pair<std::string, std::pair<int, Type>>> defs;
            for (Expr* e : o->exprs) { q.pool.addId(e); defs.push_back({expressionName(e), {b.pinned(e), e->type}}); }
            for (auto& d : defs) b.sym[d.first] = d.second;
            continue;
        }
        if (o->tag == RSQ_OP_HASHJOIN) {
            if (o->hashTable < 0 || o->hashTable >= (int)q.hashTables.size() || o->hashTable >= G2_MAX_TABLES) b.no("join table index");
            HashTable& ht = *q.hashTables[(size_t)o->hashTable];
            const bool build = from == o->child[0];
            // key words: one per equality, numeric only (string keys have word forms of their own, codegen.cpp padKeyWords)
            std::vector<int> keyRegs;
            for (Expr* eq : o->exprs) {
                if (eq->tag != RSQ_E_EQ) failType("The elements of the expression list passed to equalitiesLeftSide(..) need the tag Expr::EQ");
                Expr* side = build ? eq->child : eq->child->next;
                q.pool.addId(side);
                if (side->type.isString() || eq->child->type.isString() || eq->child->next->type.isString()) b.no("string join keys");
                keyRegs.push_back(b.pinned(side));
            }
            if (keyRegs.size() != ht.keys.size() || keyRegs.size() > 8) b.no("join key words");
            if (build) {
                S.kind = G2_SINK_BUILD; S.table = ht.id; S.nKeys = (int)keyRegs.size();
                for (size_t i = 0; i < keyRegs.size(); i++) S.keyReg[i] = (uint8_t)keyRegs[i];
                if (ht.payload.size() > G2_MAX_PAYLOAD) b.no("too many payload words");
                S.nPayload = (int)ht.payload.size();
                for (size_t i = 0; i < ht.payload.size(); i++) {
                    auto it = b.sym.find(ht.payload[i].name);
                    int r;
                    if (it != b.sym.end()) r = it->second.first;
                    else { Type t; r = b.column(ht.payload[i].name, t); }
                    S.payloadReg[i] = (uint8_t)r;
                }
                return;
            }
            if (b.p.probes.size() >= 16) b.no("too many probes");
            GenericProbeDesc P;
            memset(&P, 0, sizeof P);
            P.table = ht.id; P.nKeys = (int)keyRegs.size(); P.single = o->singleMatch ? 1 : 0; P.slotReg = -1;
            for (size_t i = 0; i < keyRegs.size(); i++) P.keyReg[i] = (uint8_t)keyRegs[i];
            if (!o->singleMatch && ++b.multiProbes > G2_MAX_DEPTH) b.no("more probes for all matches than the interpreter's stack holds");
            if (ht.payload.size() > G2_MAX_PAYLOAD) b.no("too many payload words");
            P.nPayload = (int)ht.payload.size();
            for (size_t i = 0; i < ht.payload.size(); i++) {
                const int r = b.alloc();
                P.payloadReg[i] = (uint8_t)r;
                b.define(ht.payload[i].name, r, ht.payload[i].type);
            }
            for (auto& al : ht.keyAlias) b.define(al.first.name, keyRegs[(size_t)al.second], al.first.type);      // equal to this row's probe key
            if (q.aggMode == AggMode::AT_JOIN_ENTRY && q.aggTable == ht.id) { P.slotReg = b.alloc(); slotRegOf[ht.id] = P.slotReg; }
            b.emit(G_PROBE, 0, 0, 0, (uint32_t)b.p.probes.size());
            b.p.probes.push_back(P);
            continue;
        }
        if (o->tag == RSQ_OP_AGGREGATION) {
            if (q.accums.size() > G2_MAX_ACCS) b.no("too many accumulators");
            S.nAccs = (int)q.accums.size();
            auto accumulators = [&] {
                for (size_t w = 0; w < q.accums.size(); w++) {
                    S.accMerge[w] = q.accums[w].merge; S.accBlock[w] = q.accumSlot[w];
                    if (w == 0) S.accReg[w] = -1;
                    else if (q.accums[w].kind == RSQ_E_COUNT) S.accReg[w] = -2;
                    else { if (!q.accums[w].inputExpr) b.no("accumulator without an input expression"); S.accReg[w] = b.pinned(q.accums[w].inputExpr); }
                }
            };
            switch (q.aggMode) {
                case AggMode::DENSE_REG: case AggMode::DENSE_LDS_PRIVATE: case AggMode::DENSE_LDS_SHARED: case AggMode::DENSE_GLOBAL: {
                    if (q.denseKeys.size() > G2_MAX_KEYS) b.no("too many group keys");
                    S.kind = G2_SINK_DENSE; S.nKeys = (int)q.denseKeys.size();
                    for (size_t k = 0; k < q.denseKeys.size(); k++) {
                        const DenseKey& dk = q.denseKeys[k];
                        S.keyReg[k] = (uint8_t)b.pinned(dk.expr);
                        S.keyByteSet[k] = dk.byteSet ? 1 : 0; S.keyMin[k] = dk.min; S.keyCard[k] = dk.card; S.keyStride[k] = dk.stride;
                        if (dk.byteSet) {
                            if (dk.values.size() > G2_MAX_SET) b.no("byte set too large");
                            S.keyNValues[k] = (int)dk.values.size();
                            for (size_t i = 0; i < dk.values.size(); i++) S.keyValues[k][i] = dk.values[i];
                        }
                    }
                    accumulators();
                    return;
                }
                case AggMode::AT_JOIN_ENTRY: {
                    auto it = slotRegOf.find(q.aggTable);
                    if (it == slotRegOf.end()) b.no("the aggregation's join entry is not probed in this pipeline");
                    S.kind = G2_SINK_ENTRY; S.table = q.aggTable; S.slotReg = it->second;
                    accumulators();
                    return;
                }
                case AggMode::HASH: {
                    if (q.aggTable < 0 || q.aggTable >= G2_MAX_TABLES) b.no("aggregation table index");
                    HashTable& ht = *q.hashTables[(size_t)q.aggTable];
                    const size_t NW = ht.keys.size() + ht.payload.size();
                    if (NW > G2_MAX_KEYW) b.no("too many group key words");
                    S.kind = G2_SINK_HASH; S.table = ht.id; S.nKeys = (int)ht.keys.size();
                    std::vector<int> wordReg(NW, -1);
                    for (size_t gi = 0; gi < o->exprs2.size(); gi++) {
                        Expr* g = o->exprs2[gi];
                        const int first = q.groupSource[gi];
                        if (g->type.isString()) {
                            // the words are taken from the string when the row reaches the sink (one register for the address, not one per word:
                            // TPC-H Q10's seven group values are 32 words)
                            if (g->type.len > 255) b.no("group value wider than 255 bytes");
                            const int a = b.pinned(g);
                            const int nw = (g->type.len + 7) / 8;
                            for (int w = 0; w < nw; w++) {
                                if ((size_t)(first + w) >= NW) b.no("group row layout");
                                wordReg[(size_t)(first + w)] = a;
                                S.wordStr[first + w] = 1; S.wordOff[first + w] = (uint8_t)(w * 8); S.wordN[first + w] = (uint8_t)std::min(8, g->type.len - w * 8);
                            }
                            if (g->type.tag == RSQ_CHAR && g->type.len > 1) {
                                if (S.nCharKeys >= G2_MAX_CHARKEYS) b.no("too many CHAR group values");
                                S.charFirst[S.nCharKeys] = (uint8_t)first; S.charLast[S.nCharKeys] = (uint8_t)(first + nw - 1); S.nCharKeys++;
                            }
                        } else {
                            if ((size_t)first >= NW) b.no("group row layout");
                            wordReg[(size_t)first] = b.pinned(g);
                        }
                    }
                    for (size_t w = 0; w < NW; w++) { if (wordReg[w] < 0) b.no("group row layout"); S.keyReg[w] = (uint8_t)wordReg[w]; }
                    accumulators();
                    return;
                }
                default: b.no("aggregation strategy");
            }
        }
        if (o->tag == RSQ_OP_MATERIALIZE) {
            if (q.matSchema.size() > G2_MAX_OUT) b.no("too many output columns");
            S.kind = G2_SINK_MATERIALIZE; S.nOut = (int)q.matSchema.size();
            for (size_t c = 0; c < q.matSchema.size(); c++) {
                const Attr& a = q.matSchema[c];
                auto it = b.sym.find(a.name);
                int r; Type t;
                if (it != b.sym.end()) { r = it->second.first; t = it->second.second; }
                else r = b.column(a.name, t);
                S.outReg[c] = (uint8_t)r;
                S.outString[c] = t.isString() ? 1 : 0;
                S.outWidth[c] = columnWidth(a.type);
                S.outSrcCap[c] = t.isString() ? t.len : 0;
            }
            return;
        }
        b.no("operator in a pipeline");
    }
}

}  // namespace

bool buildGenericPlan(Query& q, std::vector<GenericProgram2>& out, std::string& why) {
    out.clear();
    try {
        if (q.hashTables.size() > G2_MAX_TABLES) throw Error(RSQ_ERR_UNSUPPORTED, "too many hash tables");
        std::vector<OpNode*> scans;
        collectScans(q.root, scans);
        if (scans.size() != q.pipelines.size()) throw Error(RSQ_ERR_UNSUPPORTED, "pipelines and scans do not correspond");
        out.resize(q.pipelines.size());
        for (size_t i = 0; i < q.pipelines.size(); i++) {
            if (scans[i]->table != q.pipelines[i].src) throw Error(RSQ_ERR_UNSUPPORTED, "pipelines and scans do not correspond");
            if (q.pipelines[i].src->nRows >= ((int64_t)1 << 31)) throw Error(RSQ_ERR_UNSUPPORTED, "table too large for the interpreter's tables");
            buildOne(q, i, scans[i], out[i]);
        }
        return true;
    } catch (const Error& e) {
        if (e.status != RSQ_ERR_UNSUPPORTED) throw;        // a typing error is the plan's, whoever compiles it
        why = e.what();
        out.clear();
        return false;
    }
}

}  // namespace rsq
