// IR generation on the GPU.  Restates generate_ir.rs (IrGenerator, :75-1199) as an explicit-stack machine:
// one WAVE lowers one function (functions only depend on the global tables of sem.h, SURVEY §7 hard part 3); all 64
// lanes run the same steps (wave-uniform control flow), the machine's registers are static LDS variables, read-only
// tables come in by scalar loads, outputs are staged in LDS and leave as whole lines.
//
// Emission order is the reference's: variables are numbered in creation order, basic blocks in creation order,
// and every instruction is tagged (bb, rank-in-bb) at push time — deferred terminators (`bbs[branch].push(..)`,
// generate_ir.rs:310-312,357-359) simply take the next rank of their block — so that the compaction kernel can
// scatter instructions into the final bb-major layout in parallel.
//
//   k_lower_functions : AST + tables -> per-function sparse regions (variables, instructions, bbs, local types,
//                       call-argument pool) + counts
//   scans             : dense bases
//   k_ir_bb_offsets   : per function, exclusive scan of its bb sizes
//   k_ir_compact      : one wave per function: sparse -> dense, ids rebased
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "sem.h"

namespace opusfe {

struct LFrame { uint8_t st; uint8_t aux; uint16_t cnt; uint32_t node; uint32_t a, b, c, d, e, f; };
static_assert(sizeof(LFrame) == 32, "frame");

// capacity multipliers of the per-function sparse regions (n = nodes of the definition); every push is checked
enum { CAP_VARS = 5, CAP_INSTRS = 6, CAP_BBS = 3, CAP_TYPES = 3, CAP_CARGS = 1, CAP_LOCALS = 1, CAP_FRAMES = 2, CAP_ERRS = 1, CAP_CONST = 16 };
__host__ __device__ __forceinline__ uint64_t lower_base(uint32_t mult, uint32_t def_node, uint32_t fidx) { return (uint64_t)mult * def_node + (uint64_t)CAP_CONST * fidx; }
__host__ __device__ __forceinline__ uint64_t lower_cap(uint32_t mult, uint32_t n_nodes) { return (uint64_t)mult * n_nodes + CAP_CONST; }

// per-function result header (sparse counts; dense bases are filled after the scans)
struct FuncIr { uint32_t def_node, func, n_vars, n_instrs, n_bbs, n_types, n_cargs, n_errs, var_base, instr_base, bb_base, type_base, carg_base, err_base, status, pad; };
struct IrCnt { uint32_t vars, instrs, bbs, types, cargs, errs; };
struct FbIdx { uint32_t nodes, funcs; };   // exclusive (nodes, functions) of the functions lowered by the sequential machine
struct OpAddFbIdx { __host__ __device__ FbIdx operator()(FbIdx a, FbIdx b) const { return FbIdx{a.nodes + b.nodes, a.funcs + b.funcs}; } };
struct OpAddIrCnt { __host__ __device__ IrCnt operator()(IrCnt a, IrCnt b) const { return IrCnt{a.vars + b.vars, a.instrs + b.instrs, a.bbs + b.bbs, a.types + b.types, a.cargs + b.cargs, a.errs + b.errs}; } };

struct LowerBuffers {
    opusfe_variable* vars; opusfe_instr* instrs; uint32_t* ibb; uint32_t* irank; uint32_t* bbcnt; uint32_t* bblast;
    uint32_t* cargs; uint32_t* lname; uint32_t* lvar; uint32_t* lstack; LFrame* frames; SemErr* errs;
    uint32_t type_sparse_base;   // id of the first local type slot (== n_gtypes)
};

// Per-unit working set kept in LDS (one wave == one function): the first entries of every table live here and only
// larger functions spill to the sparse global regions.  All lanes execute the same instruction stream on the same
// addresses, so an LDS read is a broadcast and a global access touches one cache line.
#ifndef OPUSFE_LDS_VARS
#define OPUSFE_LDS_VARS 304
#endif
#ifndef OPUSFE_LDS_BBS
#define OPUSFE_LDS_BBS 96
#endif
#ifndef OPUSFE_LDS_TYPES
#define OPUSFE_LDS_TYPES 64
#endif
#ifndef OPUSFE_ISTAGE
#define OPUSFE_ISTAGE 8
#endif
enum { ISTAGE = OPUSFE_ISTAGE };
static_assert(sizeof(opusfe_instr) == 24, "instruction staging copies records as 6 words");
#ifndef OPUSFE_LDS_FRAMES
#define OPUSFE_LDS_FRAMES 32
#endif
#ifndef OPUSFE_LDS_LOCALS
#define OPUSFE_LDS_LOCALS 64
#endif
enum { LDS_FRAMES = OPUSFE_LDS_FRAMES, LDS_VARS = OPUSFE_LDS_VARS, LDS_BBS = OPUSFE_LDS_BBS, LDS_LOCALS = OPUSFE_LDS_LOCALS, LDS_TYPES = OPUSFE_LDS_TYPES };
struct LowerLds {
    LFrame frames[LDS_FRAMES];
    uint32_t vtype[LDS_VARS];
    uint32_t vnamed[(LDS_VARS + 31) / 32];     // bit v: variable v is not a temporary
    uint32_t bbcnt[LDS_BBS], bblast[LDS_BBS];
    uint32_t lname[LDS_LOCALS], lvar[LDS_LOCALS], lhash[LDS_LOCALS], lstack[LDS_LOCALS], lshash[LDS_LOCALS];
    TypeRec types[LDS_TYPES];
    uint32_t match_stack[96];   // work list of TypeOps::types_match
    // the newest instructions, staged as {6 instruction words, bb, rank}: they reach the sparse global arrays ISTAGE at a
    // time as whole lines (one instruction at a time was three partial-line stores and a store wait in every push)
    uint32_t istage[ISTAGE][8];
};

using ::uni;
__device__ __forceinline__ uint8_t uni(uint8_t v) { return (uint8_t)uni((uint32_t)v); }
__device__ __forceinline__ uint16_t uni(uint16_t v) { return (uint16_t)uni((uint32_t)v); }
__device__ __forceinline__ bool uni(bool v) { return uni((uint32_t)v) != 0; }
__device__ __forceinline__ NameKey uni(NameKey k) { return NameKey{uni(k.tok), uni(k.sym), uni(k.mix)}; }
// Arguments of a helper that is not inlined arrive in VGPRs and the compiler has to assume that they differ between
// lanes (every `if` becomes a compare + exec-mask sequence).  They never do here: say so once at the top of the helper.
__device__ __forceinline__ LFrame uni(const LFrame& f) {
    union { LFrame f; uint32_t w[8]; } a, b;
    a.f = f;
#pragma unroll
    for (int i = 0; i < 8; i++) b.w[i] = uni(a.w[i]);
    return b.f;
}
__device__ __forceinline__ opusfe_node uni(const opusfe_node& n) {
    union { opusfe_node n; uint32_t w[6]; } a, b;
    a.n = n;
#pragma unroll
    for (int i = 0; i < 6; i++) b.w[i] = uni(a.w[i]);
    return b.n;
}

// optional cycle accounting of the lowering machine (build with -DOPUSFE_LOWER_PROF; bench prints the table)
#ifdef OPUSFE_LOWER_PROF
#define LPROF(id) ProfScope _ps_##id{prof, id}
struct ProfScope { uint64_t* p; int id; uint64_t t0; __device__ ProfScope(uint64_t* p_, int i) : p(p_), id(i), t0(__builtin_readcyclecounter()) {} __device__ ~ProfScope() { p[id] += __builtin_readcyclecounter() - t0; } };
#else
#define LPROF(id) do { } while (0)
#endif
enum { LP_TOTAL = 0, LP_NODE, LP_LOCALFIND, LP_CALLVARS, LP_AUTOCAST, LP_TRYCAST, LP_LOOKUPNAME, LP_FIELD, LP_PUSH, LP_NEWVAR, LP_FRAMES, LP_LOOKUPFN, LP_NAMED,
       LP_STATE_CYCLES = LP_NAMED, LP_STATE_STEPS = LP_STATE_CYCLES + 48, LP_COUNT = LP_STATE_STEPS + 48 };   // + per machine state: cycles and steps

enum LState : uint8_t {
    LS_BLOCK, LS_BLOCK_LOOP, LS_BLOCK_ITEM,
    LS_STMT, LS_ST_EXPR_1, LS_ST_VDEF_1, LS_ST_ASG_1, LS_ST_ASG_2, LS_ST_RET_1, LS_ST_IF_1, LS_ST_IF_2, LS_ST_IF_3, LS_ST_WH_1, LS_ST_FOR_1,
    LS_FL_0, LS_FL_1, LS_FL_2,
    LS_E, LS_OP_IS_1, LS_OP_IS_2, LS_OP_AND_1, LS_OP_AND_2, LS_OP_NOT_1, LS_OP_BIN_1, LS_OP_BIN_2, LS_CAST_1, LS_NEG_1,
    LS_P, LS_P_DEFAULT_1, LS_P_DEREF_1, LS_P_FIELD_1, LS_P_NC_LOOP, LS_P_NC_ITEM, LS_P_COI_1, LS_P_IDX_0, LS_P_IDX_LOOP, LS_P_IDX_ITEM,
    LS_P_REC_LOOP, LS_P_REC_ITEM,
    LS_DONE
};
enum { PF_DEREF = 4, PF_MUSTLV = 8 };                 // flags in LFrame.aux of a P frame (low 2 bits: PtrKind)
enum { IDX_PTR = 0, IDX_ARR = 1, IDX_FN = 2 };

// The machine's registers are static LDS variables (one machine per workgroup == one wave): every helper reaches
// them with ds_read/ds_write at constant addresses.  As fields of an object they were reached through a generic
// `this` pointer (flat loads, or scratch when the object was private) in every helper the compiler keeps out of line.
struct Lowerer {
    // The tables: on the GPU `S.nodes[i]` etc. are scalar loads (defs.h STable); the host emulation keeps plain loads.
#ifdef OPUSFE_EMU
    MACHINE_STATE SemView S;
#else
    struct View {
        STable<uint8_t> src; STable<opusfe_token> tok; STable<uint64_t> val64; STable<opusfe_node> nodes; STable<uint32_t> tid_of_node;
        STable<NameSlot> slots; STable<NameEntry> entries; STable<FuncRec> funcs; STable<GlobalRec> globals;
        TypeRec* types; uint32_t slot_mask, n_gtypes;
        __device__ void operator=(const SemView& v) {
            src.p = v.src.p; tok.p = v.tok.p; val64.p = v.val64.p; nodes.p = v.nodes.p; tid_of_node.p = v.tid_of_node.p;
            slots.p = v.slots.p; entries.p = v.entries.p; funcs.p = v.funcs.p; globals.p = v.globals.p;
            types = v.types; slot_mask = v.slot_mask; n_gtypes = v.n_gtypes;
        }
    };
    MACHINE_STATE View S;
#endif
    MACHINE_STATE LowerLds lds;
    MACHINE_STATE uint32_t fidx; MACHINE_STATE FuncRec F;
    // sparse regions of this function
    MACHINE_STATE opusfe_variable* vars; MACHINE_STATE uint32_t nvars, vcap;
    MACHINE_STATE opusfe_instr* instrs; MACHINE_STATE uint32_t* ibb; MACHINE_STATE uint32_t* irank; MACHINE_STATE uint32_t ninstrs, icap;
    MACHINE_STATE uint32_t iflushed;   // instructions [iflushed, ninstrs) are still in lds.istage
    MACHINE_STATE uint32_t* bbcnt; MACHINE_STATE uint32_t* bblast; MACHINE_STATE uint32_t nbbs, bcap;
    MACHINE_STATE uint32_t tbase, ntypes, tcap;
    MACHINE_STATE uint32_t* cargs; MACHINE_STATE uint32_t ncargs, ccap;
    MACHINE_STATE uint32_t* lname; MACHINE_STATE uint32_t* lvar; MACHINE_STATE uint32_t nlocals, lcap;
    MACHINE_STATE uint32_t* lstack; MACHINE_STATE uint32_t nstack;
    MACHINE_STATE LFrame* frames; MACHINE_STATE uint32_t sp, fcap;
    MACHINE_STATE SemErr* errs; MACHINE_STATE uint32_t nerrs, ecap, ncont /* continuation records among nerrs */;
    // generate_ir.rs:75-97 state
    MACHINE_STATE uint32_t bb_index, break_head, loop_begin;
    MACHINE_STATE uint32_t dev_status;
    MACHINE_STATE LFrame cur;
    MACHINE_STATE uint32_t rv; MACHINE_STATE bool rl; MACHINE_STATE bool rdiv;
    MACHINE_STATE uint32_t sym_mix;  // the file mix of this function's symbol keys (sem.h NameKey; 0 outside group mode)
    MACHINE_STATE bool sym_locals;   // some local is named by an operator symbol (`operator +` as a variable name): only then can a symbol key hit a local
#ifdef OPUSFE_LOWER_PROF
    MACHINE_STATE uint64_t prof[LP_COUNT];
#endif

    __device__ __forceinline__ void internal() { if (dev_status < DS_INTERNAL_CAPACITY) dev_status = DS_INTERNAL_CAPACITY; }
    __device__ __forceinline__ void unsupported() { if (dev_status < DS_UNSUPPORTED_SEMANTIC) dev_status = DS_UNSUPPORTED_SEMANTIC; }
    // (the machine state is shared by the wave, so the lane id cannot be part of it)
    __device__ __forceinline__ static uint32_t lane_of() { return threadIdx.x % UNIT_LANES; }
    // node / token i (wave-uniform index): scalar loads on the GPU
    __device__ __forceinline__ opusfe_node N(uint32_t i) const { LPROF(LP_NODE); return S.nodes[i]; }
    __device__ __forceinline__ opusfe_token TK(uint32_t i) const { return S.tok[i]; }
    struct TokView { __device__ __forceinline__ opusfe_token operator[](uint32_t i) const { return Lowerer{}.TK(i); } };   // what the name helpers index
    __device__ __forceinline__ static TypeOps<Lowerer> T() { return TypeOps<Lowerer>{}; }
    __device__ __forceinline__ TypeRec trec(uint32_t t) { return T().rec(t); }

    // ---- names (wave-cooperative byte compare: one memory round trip instead of one per character) ----
    __device__ bool neq(NameKey a, NameKey b) const {
        uint8_t sa = name_sym(TokView{}, a), sb = name_sym(TokView{}, b);
        if (sa != sb) return false;
        if (sa != T_Identifier) return name_mix(TokView{}, a) == name_mix(TokView{}, b);
        const opusfe_token x = TK(a.tok), y = TK(b.tok);
        if (x.len != y.len || x.payload != y.payload) return false;
        if (x.low == y.low) return true;
#ifdef OPUSFE_EMU
        for (uint32_t i = 0; i < x.len; i++) if (gld(S.src.ptr() + x.low + i) != gld(S.src.ptr() + y.low + i)) return false;
        return true;
#else
        const uint32_t len = uni(x.len), xl = uni(x.low), yl = uni(y.low);
        for (uint32_t base = 0; base < len; base += 64) {
            uint32_t i = base + lane_of();
            bool ne = i < len && gld(S.src.ptr() + xl + i) != gld(S.src.ptr() + yl + i);
            if (__ballot(ne)) return false;
        }
        return true;
#endif
    }
    __device__ uint32_t find_slot(NameKey k) const {
        k = uni(k);
        uint32_t h = name_hash(TokView{}, k) & S.slot_mask;
        for (uint32_t probe = 0; probe <= S.slot_mask; probe++) {
            uint32_t rep = uni(S.slots[h].rep);
            if (rep == OPUSFE_NONE) return OPUSFE_NONE;
            if (neq(k, key_tok(rep))) return h;
            h = (h + 1) & S.slot_mask;
        }
        return OPUSFE_NONE;
    }
    __device__ uint32_t last_entry(NameKey k, uint32_t kind) const {
        k = uni(k); kind = uni(kind);
        uint32_t s = find_slot(k);
        if (s == OPUSFE_NONE) return OPUSFE_NONE;
        const NameSlot sl = S.slots[s];
        for (uint32_t j = uni(sl.count); j > 0; j--) {
            NameEntry e = S.entries[uni(sl.offset) + j - 1];
            if ((uni(e.kind_index) >> 30) == kind) return uni(sl.offset) + j - 1;
        }
        return OPUSFE_NONE;
    }
    MACHINE_STATE uint64_t user_syms;   // bit per operator-symbol token kind that has user-defined overloads in this unit

    // ---- spans / errors -------------------------------------------------------------------------
    // what the message of the NEXT report() prints besides its span (SemErr in sem.h): set right before the report, consumed by it
    MACHINE_STATE uint32_t ei_name, ei_nargs, ei_t0, ei_t1, ei_t2, ei_t3;
    __device__ void einfo(uint32_t name, uint32_t nargs, uint32_t t0, uint32_t t1 = OPUSFE_NONE, uint32_t t2 = OPUSFE_NONE, uint32_t t3 = OPUSFE_NONE) {
        ei_name = uni(name); ei_nargs = uni(nargs); ei_t0 = uni(t0); ei_t1 = uni(t1); ei_t2 = uni(t2); ei_t3 = uni(t3);
    }
    __device__ void einfo_clear() { ei_name = OPUSFE_NONE; ei_nargs = 0; ei_t0 = ei_t1 = ei_t2 = ei_t3 = OPUSFE_NONE; }
    __device__ static uint32_t einfo_key(NameKey k) { return k.tok != OPUSFE_NONE ? k.tok : (0x80000000u | (uint32_t)k.sym); }
    __device__ void report(uint32_t kind, uint32_t low, uint32_t high) {
        kind = uni(kind); low = uni(low); high = uni(high);
        if (nerrs < ecap) gst(errs + nerrs, sem_err(kind, low, high - low, ei_name, ei_nargs, ei_t0, ei_t1, ei_t2, ei_t3));
        else internal();   // more diagnostics than the region holds (one per node + 16): an internal-capacity status, never a clipped list
        nerrs++;
        einfo_clear();
    }
    __device__ void report_node(uint32_t kind, uint32_t node) { kind = uni(kind); node = uni(node); report(kind, N(node).low, N(node).high); }
    __device__ void report_tok(uint32_t kind, uint32_t t) { kind = uni(kind); t = uni(t); const opusfe_token x = TK(t); report(kind, x.low, x.low + x.len); }

    // ---- variables / instructions / blocks (generate_ir.rs:118-131) -------------------------------
    __device__ __forceinline__ uint32_t new_var(uint32_t type_id, bool is_temporary) {
        LPROF(LP_NEWVAR);
        if (nvars >= vcap) { internal(); return 0; }
        // the first LDS_VARS variables exist only in LDS (type id; one bit for the few that are not temporaries) until flush_vars() writes them out as
        // one contiguous run when the function is done; later ones go straight to the sparse global array
        if (nvars < LDS_VARS) { lds.vtype[nvars] = type_id; if (!is_temporary) lds.vnamed[nvars >> 5] |= 1u << (nvars & 31); }
        else gst(vars + nvars, opusfe_variable{type_id, is_temporary ? 1u : 0u});
        return nvars++;
    }
    __device__ __forceinline__ uint32_t vtype(uint32_t v) const { return v < nvars ? uni(v < LDS_VARS ? lds.vtype[v] : gld(&vars[v].type_id)) : (uint32_t)TP_Error; }
    __device__ __forceinline__ void set_vtype(uint32_t v, uint32_t t) { if (v < nvars) { if (v < LDS_VARS) lds.vtype[v] = t; else gst(&vars[v].type_id, t); } }
    __device__ void flush_types() {    // all lanes
        const uint32_t n = ntypes < LDS_TYPES ? ntypes : (uint32_t)LDS_TYPES;
        for (uint32_t j = lane_of(); j < n; j += UNIT_LANES) gst(S.types + tbase + j, (TypeRec)lds.types[j]);
    }
    __device__ void flush_vars() {     // all lanes
        const uint32_t n = nvars < LDS_VARS ? nvars : (uint32_t)LDS_VARS;
        for (uint32_t j = lane_of(); j < n; j += UNIT_LANES) gst(vars + j, opusfe_variable{lds.vtype[j], ((lds.vnamed[j >> 5] >> (j & 31)) & 1u) ^ 1u});
    }
    // staged instructions -> sparse global arrays (all lanes; the instruction words of the stage are one contiguous run)
    __device__ __noinline__ void flush_instrs() {
        const uint32_t n = ninstrs - iflushed, lane = lane_of();
        uint32_t* dst = (uint32_t*)(instrs + iflushed);
        for (uint32_t w = lane; w < n * 6; w += UNIT_LANES) gst(dst + w, lds.istage[w / 6][w % 6]);
        for (uint32_t j = lane; j < n; j += UNIT_LANES) { gst(ibb + iflushed + j, lds.istage[j][6]); gst(irank + iflushed + j, lds.istage[j][7]); }
        iflushed = ninstrs;
    }
    __device__ opusfe_instr instr_get(uint32_t t) const {
        t = uni(t);
        if (t < iflushed) return gld(instrs + t);
        union { opusfe_instr i; uint32_t w[6]; } u;
#pragma unroll
        for (int k = 0; k < 6; k++) u.w[k] = uni(lds.istage[t - iflushed][k]);
        return u.i;
    }
    __device__ void instr_set(uint32_t t, opusfe_instr in) {    // the block and rank of instruction t stay as they are
        t = uni(t);
        if (t < iflushed) { gst(instrs + t, in); return; }
        union { opusfe_instr i; uint32_t w[6]; } u; u.i = in;
#pragma unroll
        for (int k = 0; k < 6; k++) lds.istage[t - iflushed][k] = u.w[k];
    }
    __device__ uint32_t push_to(uint32_t bb, opusfe_instr in) {   // by value: a reference would go through scratch memory
        bb = uni(bb);
        LPROF(LP_PUSH);
        if (ninstrs >= icap || bb >= nbbs) { internal(); return 0; }
        if (ninstrs - iflushed >= ISTAGE) flush_instrs();
        uint32_t t = ninstrs++;
        uint32_t rank;
        if (bb < LDS_BBS) { rank = lds.bbcnt[bb]; lds.bbcnt[bb] = rank + 1; lds.bblast[bb] = t; }
        else { rank = gld(bbcnt + bb); gst(bbcnt + bb, rank + 1); gst(bblast + bb, t); }
        union { opusfe_instr i; uint32_t w[6]; } u; u.i = in;
        uint32_t* e = lds.istage[t - iflushed];
#pragma unroll
        for (int k = 0; k < 6; k++) e[k] = u.w[k];
        e[6] = bb; e[7] = rank;
        return t;
    }
    __device__ __forceinline__ uint32_t push(const opusfe_instr& in) { return push_to(bb_index, in); }
    __device__ static opusfe_instr ins(uint16_t op, uint32_t dst = 0, uint32_t a = 0, uint32_t b = 0, uint32_t c = 0, uint32_t extra = 0, uint16_t sub = 0) {
        opusfe_instr i; i.opcode = op; i.sub = sub; i.dst = dst; i.a = a; i.b = b; i.c = c; i.extra = extra; return i;
    }
    __device__ uint32_t new_bb() {
        if (nbbs >= bcap) { internal(); return 0; }
        if (nbbs < LDS_BBS) { lds.bbcnt[nbbs] = 0; lds.bblast[nbbs] = OPUSFE_NONE; } else { gst(bbcnt + nbbs, 0u); gst(bblast + nbbs, (uint32_t)OPUSFE_NONE); }
        return nbbs++;
    }
    __device__ __forceinline__ uint32_t bb_last(uint32_t bb) const { return uni(bb < LDS_BBS ? lds.bblast[bb] : gld(bblast + bb)); }
    __device__ uint32_t generate_error() { uint32_t v = new_var(TP_Error, true); push(ins(I_Error, v)); return v; }
    __device__ uint32_t new_type(const TypeRec& r) {
        if (ntypes >= tcap) { internal(); return TP_Error; }
        // the first LDS_TYPES records exist only in LDS until flush_types(); TypeOps::rec reads them there
        const uint32_t id = tbase + ntypes;
        if (ntypes < LDS_TYPES) lds.types[ntypes] = r; else gst(S.types + id, r);
        ntypes++;
        return id;
    }
    __device__ uint32_t type_pointer(uint8_t pt, uint32_t sub) { pt = uni(pt); sub = uni(sub); TypeRec r{}; r.kind = TY_Pointer; r.ptr = pt; r.sub = sub; r.alias = OPUSFE_NONE; return new_type(r); }
    __device__ uint32_t type_mut(uint32_t sub) { sub = uni(sub); return type_pointer(PT_Mutable, sub); }
    __device__ uint32_t reserve_cargs(uint32_t n) { n = uni(n); if (ncargs + n > ccap) { internal(); return 0; } uint32_t o = ncargs; ncargs += n; return o; }

    // ---- functions as values --------------------------------------------------------------------
    __device__ __forceinline__ bool fn_is_builtin(uint32_t f) const { return (f & FN_BUILTIN) != 0; }
    __device__ uint32_t fn_nargs(uint32_t f) const { f = uni(f); return fn_is_builtin(f) ? builtin_nargs((f >> 8) & 0xFF) : S.funcs[f].nargs; }
    __device__ uint32_t fn_ret(uint32_t f) const { f = uni(f); return fn_is_builtin(f) ? builtin_ret_type((f >> 8) & 0xFF, f & 0xFF) : S.funcs[f].ret; }
    __device__ uint32_t fn_arg_cursor(uint32_t f) const { f = uni(f); return fn_is_builtin(f) ? 0u : S.funcs[f].first_arg; }
    __device__ uint32_t fn_arg_type(uint32_t f, uint32_t& cursor) const {   // returns the type and advances
        f = uni(f);
        if (fn_is_builtin(f)) return builtin_arg_type((f >> 8) & 0xFF, f & 0xFF);
        uint32_t t = S.tid_of_node[N(cursor).first_child];
        cursor = N(cursor).next;
        return t;
    }
    __device__ uint32_t type_function(uint32_t f) { f = uni(f); TypeRec r{}; r.kind = TY_Function; r.src = TS_FUNC; r.ref = f; r.sub = S.funcs[f].ret; r.alias = OPUSFE_NONE; return new_type(r); }

    // resolve_type of a parse-tree type node (compile.rs:919-966): precomputed id; UndefinedType errors are raised
    // here, when the reference would resolve it, in source order (the subtree is a contiguous node range)
    __device__ uint32_t resolve_type(uint32_t y, const uint8_t* undefined_flag) {
        y = uni(y);
        uint32_t last = y;
        while (N(last).nchild > 0) { uint32_t c = N(last).first_child; while (N(c).next != OPUSFE_NONE) c = N(c).next; last = c; }
        for (uint32_t k = y; k <= last; k++) if (gld(undefined_flag + k)) report_node(SE_UndefinedType, k);
        return S.tid_of_node[y];
    }
    MACHINE_STATE const uint8_t* undefined_flag;

    // ---- scopes / locals (generate_ir.rs:176-185) -------------------------------------------------
    __device__ __forceinline__ uint32_t l_name(uint32_t j) const { return uni(j < LDS_LOCALS ? lds.lname[j] : gld(lname + j)); }
    __device__ __forceinline__ uint32_t l_var(uint32_t j) const { return uni(j < LDS_LOCALS ? lds.lvar[j] : gld(lvar + j)); }
    __device__ __forceinline__ void l_set_var(uint32_t j, uint32_t v) { if (j < LDS_LOCALS) lds.lvar[j] = v; else gst(lvar + j, v); }
    __device__ __forceinline__ uint32_t l_stack(uint32_t j) const { return uni(j < LDS_LOCALS ? lds.lstack[j] : gld(lstack + j)); }
    // The LDS-resident part of the table is searched by all lanes at once (lane j looks at entry base + j; the newest
    // match wins); a hash match is confirmed on the bytes.  Entries beyond LDS_LOCALS live in global memory.
    __device__ uint32_t local_find(NameKey k) const {
        k = uni(k);
        LPROF(LP_LOCALFIND);
        const uint32_t h = name_hash(TokView{}, k);
        for (uint32_t j = nlocals; j > LDS_LOCALS; j--) {          // spilled entries (rare), newest first
            const uint32_t e = j - 1;
            if (l_var(e) != OPUSFE_NONE && neq(k, key_tok(l_name(e)))) return e;
        }
        const uint32_t n = nlocals < LDS_LOCALS ? nlocals : (uint32_t)LDS_LOCALS;
#ifdef OPUSFE_EMU
        for (uint32_t j = n; j > 0; j--) {
            const uint32_t e = j - 1;
            if (lds.lvar[e] != OPUSFE_NONE && lds.lhash[e] == h && neq(k, key_tok(lds.lname[e]))) return e;
        }
        return OPUSFE_NONE;
#else
        for (uint32_t top = n; top > 0;) {
            const uint32_t base = top > 64 ? top - 64 : 0;
            const uint32_t e = base + lane_of();
            const bool m = e < top && lds.lvar[e] != OPUSFE_NONE && lds.lhash[e] == h;
            uint64_t mask = __ballot(m);
            while (mask) {
                const uint32_t hit = base + 63u - (uint32_t)__builtin_clzll(mask);
                if (neq(k, key_tok(uni(lds.lname[hit])))) return hit;
                mask &= ~(1ull << (hit - base));
            }
            top = base;
        }
        return OPUSFE_NONE;
#endif
    }
    __device__ void local_set(uint32_t name_tok, uint32_t v) {
        name_tok = uni(name_tok); v = uni(v);
        if (name_sym(TokView{}, key_tok(name_tok)) != T_Identifier) sym_locals = true;
        uint32_t j = local_find(key_tok(name_tok));
        if (j != OPUSFE_NONE) { l_set_var(j, v); return; }
        if (nlocals >= lcap) { internal(); return; }
        if (nlocals < LDS_LOCALS) { lds.lname[nlocals] = name_tok; lds.lvar[nlocals] = v; lds.lhash[nlocals] = name_hash(TokView{}, key_tok(name_tok)); }
        else { gst(lname + nlocals, name_tok); gst(lvar + nlocals, v); }
        nlocals++;
    }
    __device__ bool contains_local(uint32_t name_tok) const {
        name_tok = uni(name_tok);
        const NameKey k = key_tok(name_tok);
        const uint32_t h = name_hash(TokView{}, k);
        for (uint32_t j = LDS_LOCALS; j < nstack; j++) if (neq(k, key_tok(l_stack(j)))) return true;
        const uint32_t n = nstack < LDS_LOCALS ? nstack : (uint32_t)LDS_LOCALS;
#ifdef OPUSFE_EMU
        for (uint32_t j = 0; j < n; j++) if (lds.lshash[j] == h && neq(k, key_tok(lds.lstack[j]))) return true;
        return false;
#else
        for (uint32_t base = 0; base < n; base += 64) {
            const uint32_t e = base + lane_of();
            uint64_t mask = __ballot(e < n && lds.lshash[e] == h);
            while (mask) {
                const uint32_t hit = base + (uint32_t)__builtin_ctzll(mask);
                if (neq(k, key_tok(uni(lds.lstack[hit])))) return true;
                mask &= mask - 1;
            }
        }
        return false;
#endif
    }
    __device__ void stack_push(uint32_t name_tok) {
        name_tok = uni(name_tok);
        if (nstack >= lcap) { internal(); return; }
        if (nstack < LDS_LOCALS) { lds.lstack[nstack] = name_tok; lds.lshash[nstack] = name_hash(TokView{}, key_tok(name_tok)); } else gst(lstack + nstack, name_tok);
        nstack++;
    }
    __device__ void end_scope(uint32_t n) {
        n = uni(n);
        while (nstack > n) { uint32_t name = l_stack(--nstack); uint32_t j = local_find(key_tok(name)); if (j != OPUSFE_NONE) l_set_var(j, OPUSFE_NONE); }
    }

    // ---- generate_ir.rs:207-231 -------------------------------------------------------------------
    // The common cases are decided in the caller; the general algorithm stays out of line (a helper of that size pays
    // for its register saves on every call, even when it returns at once).
    __device__ __forceinline__ uint32_t autocast(uint32_t variable, uint32_t expected, uint32_t span_node) {
        const uint32_t found = vtype(variable);
        if (found < TP_COUNT && expected < TP_COUNT && found != TP_Error && expected != TP_Error) {      // both primitive: types_match is id equality
            if (found == expected) return variable;
            if ((found == TP_GenericInteger && kind_is_integral(prim_kind(expected))) || found == TP_Generic) { set_vtype(variable, expected); return variable; }
        }
        return autocast_general(variable, expected, span_node);
    }
    __device__ uint32_t autocast_general(uint32_t variable, uint32_t expected, uint32_t span_node) {
        variable = uni(variable); expected = uni(expected); span_node = uni(span_node);
        LPROF(LP_AUTOCAST);
        uint32_t found = vtype(variable);
        if (T().types_match(found, expected)) return variable;
        int ct = T().can_autocast(found, expected);
        if (ct >= 0) {
            uint8_t fk = trec(found).kind;
            if (fk == TY_GenericInteger || fk == TY_Generic) { set_vtype(variable, expected); return variable; }
            uint32_t r = new_var(expected, true);
            push(ins(I_Cast, r, variable, 0, 0, 0, (uint16_t)ct));
            return r;
        }
        einfo(OPUSFE_NONE, 0, expected, found);
        report_node(SE_UnexpectedType, span_node);
        return generate_error();
    }
    // :235-242
    __device__ uint32_t get_lvalue_type(uint32_t v) {
        v = uni(v);
        const TypeRec r = trec(vtype(v));
        if (r.kind == TY_Pointer) return r.sub;
        if (r.kind != TY_Error) unsupported();   // reference: unreachable!()
        return TP_Error;
    }
    // :495-503
    __device__ uint32_t dereference_pointer(uint32_t p) { p = uni(p); uint32_t v = new_var(get_lvalue_type(p), true); push(ins(I_Load, v, p)); return v; }

    // :808-855; `sp_low/sp_high` is the span for errors
    __device__ __forceinline__ uint32_t try_cast_pointer(uint32_t v, uint8_t pt, uint32_t sp_low, uint32_t sp_high) {
        const TypeRec r = trec(vtype(v));
        if (r.kind == TY_Pointer && r.ptr == pt) return v;     // the common case, decided in the caller
        return try_cast_pointer_general(v, pt, sp_low, sp_high);
    }
    __device__ uint32_t try_cast_pointer_general(uint32_t v, uint8_t pt, uint32_t sp_low, uint32_t sp_high) {
        v = uni(v); pt = uni(pt); sp_low = uni(sp_low); sp_high = uni(sp_high);
        LPROF(LP_TRYCAST);
        const TypeRec r = trec(vtype(v));
        if (r.kind == TY_Pointer) {
            uint8_t pt2 = r.ptr;
            if (pt2 == pt) return v;
            if ((pt2 == PT_Mutable && pt == PT_Reference) || (pt2 == PT_ArrayMutable && pt == PT_Array) || (pt2 == PT_Array && pt == PT_Reference)
                || (pt2 == PT_ArrayMutable && pt == PT_Mutable) || (pt2 == PT_ArrayMutable && pt == PT_Reference)) {
                uint32_t nv = new_var(type_pointer(pt, r.sub), true);
                push(ins(I_Cast, nv, v, 0, 0, 0, CT_PointerType));
                return nv;
            }
            if ((pt2 == PT_Reference && pt == PT_Mutable) || (pt2 == PT_Array && pt == PT_ArrayMutable)) { report(SE_ImmutableLvalue, sp_low, sp_high); return generate_error(); }
            report(SE_ArrayPointerToNonArray, sp_low, sp_high);
            return generate_error();
        }
        if (r.kind == TY_Error) return generate_error();
        einfo(OPUSFE_NONE, 0, vtype(v));
        report(SE_InvalidOperandType, sp_low, sp_high);
        return generate_error();
    }

    // Compiler::lookup_function (compile.rs:1083-1139) over builtins followed by the user overloads in definition order
    struct NameFns { uint32_t slot; bool has; };   // (results travel in registers: an out-parameter of a helper that is not inlined lives in scratch memory)
    __device__ NameFns name_has_functions(NameKey name) const {
        name = uni(name);
        uint8_t sym = name_sym(TokView{}, name);
        uint32_t s = find_slot(name);
        if (sym != T_Identifier && builtin_count(sym) > 0) return NameFns{s, true};
        if (s == OPUSFE_NONE) return NameFns{s, false};
        const NameSlot sl = S.slots[s];
        for (uint32_t j = 0; j < sl.count; j++) if ((S.entries[sl.offset + j].kind_index >> 30) == NK_FUNCTION) return NameFns{s, true};
        return NameFns{s, false};
    }
    __device__ uint32_t candidate(NameKey name, uint32_t slot, uint32_t c, uint32_t* total) const {
        name = uni(name); slot = uni(slot); c = uni(c);
        uint8_t sym = name_sym(TokView{}, name);
        uint32_t nb = sym != T_Identifier ? (uint32_t)builtin_count(sym) : 0u;
        uint32_t nu = slot != OPUSFE_NONE ? S.slots[slot].count : 0u;
        *total = nb + nu;
        if (c < nb) return FN_BUILTIN | ((uint32_t)sym << 8) | c;
        NameEntry e = S.entries[S.slots[slot].offset + (c - nb)];
        if ((e.kind_index >> 30) != NK_FUNCTION) return FN_NONE;
        return e.kind_index & 0x3FFFFFFFu;
    }
    // call arguments: variable ids in the call-argument pool (`mem`, global memory) or, for operator calls, the two
    // operands by value: a stack array handed to a helper that is not inlined would live in scratch memory
    struct Callee { uint32_t fv, out, a0, a1; bool builtin, ok; };   // a0 / a1: the operands after their autocasts (operator calls)
    struct Args {
        uint32_t* mem; uint32_t a0, a1;
        __device__ __forceinline__ uint32_t get(uint32_t i) const { return mem ? uni(gld(mem + i)) : (i == 0 ? a0 : a1); }
        __device__ __forceinline__ void set(uint32_t i, uint32_t v) { if (mem) gst(mem + i, v); else if (i == 0) a0 = v; else a1 = v; }
    };
    __device__ bool candidate_ok(uint32_t f, Args argv, const uint32_t* types_in, uint32_t n) {
        f = uni(f); n = uni(n);
        if (fn_nargs(f) != n) return false;
        uint32_t cur = fn_arg_cursor(f);
        for (uint32_t i = 0; i < n; i++) if (T().can_autocast(types_in ? uni(gld(types_in + i)) : vtype(argv.get(i)), fn_arg_type(f, cur)) < 0) return false;
        return true;
    }
    // arg types are read through `argv` (variable ids) unless `types` is given (Proc)
    __device__ uint32_t compiler_lookup_function(uint32_t sp_low, uint32_t sp_high, NameKey name, Args argv, const uint32_t* types_in, uint32_t n) {
        sp_low = uni(sp_low); sp_high = uni(sp_high); name = uni(name); n = uni(n);
        LPROF(LP_LOOKUPFN);
        {   // builtin operator on primitive operands with no user overloads of that symbol: the general algorithm below
            // reduces to arithmetic.  The candidates have pairwise different parameter types, so with ok_i(t) = "operand i
            // autocasts to t" (a_i == t, GenericInteger -> integral t, Generic -> any t) the set of viable overloads is
            //   both concrete: {a0} if a0 == a1;   concrete + GenericInteger: {a0} if a0 is integral;   concrete + Generic: {a0}
            //   (one viable overload is returned as it is, :cnt == 1)
            //   GenericInteger twice: the ten integral overloads, of which Integer32 matches with default generics
            //   anything else with Generic: several viable, none matches with default generics -> general path (Ambiguous)
            const uint8_t sym = name_sym(TokView{}, name);
            if (sym != T_Identifier && !types_in && builtin_count(sym) > 0 && !((user_syms >> sym) & 1) && n == builtin_nargs(sym)) {
                const uint32_t a0 = vtype(argv.get(0)), a1 = n > 1 ? vtype(argv.get(1)) : a0;
                if (a0 < TP_COUNT && a1 < TP_COUNT && a0 != TP_Error && a1 != TP_Error) {
                    const int idx = builtin_pick(sym, a0, a1);
                    if (idx >= 0) return FN_BUILTIN | ((uint32_t)sym << 8) | (uint32_t)idx;
                }
            }
        }
        const NameFns nf = name_has_functions(name);
        const uint32_t slot = nf.slot;
#define ARG_T(i) (types_in ? uni(gld(types_in + (i))) : vtype(argv.get(i)))
#define FN_IDENT() einfo(einfo_key(name), n, n > 0 ? ARG_T(0) : OPUSFE_NONE, n > 1 ? ARG_T(1) : OPUSFE_NONE, n > 2 ? ARG_T(2) : OPUSFE_NONE, n > 3 ? ARG_T(3) : OPUSFE_NONE)
        // a function identifier prints ALL its argument types (compile.rs:115-122, 300-320): the record holds four, the rest follow in
        // continuation records (kind SE_Continuation: eight type ids in the words behind `kind`), which are no diagnostics of their own
#define FN_REPORT(kind_) do { FN_IDENT(); report(kind_, sp_low, sp_high); \
            for (uint32_t q_ = 4; q_ < n; q_ += 8) { SemErr ce_; ce_.kind = SE_Continuation; uint32_t* w_ = &ce_.low; \
                for (uint32_t j_ = 0; j_ < 8; j_++) w_[j_] = q_ + j_ < n ? ARG_T(q_ + j_) : OPUSFE_NONE; \
                if (nerrs < ecap) gst(errs + nerrs, ce_); else internal(); nerrs++; ncont++; } } while (0)
        if (!nf.has) { FN_REPORT(SE_UndefinedFunction); return FN_NONE; }
        if (n > 64) { unsupported(); return FN_NONE; }
        uint32_t total = 0; candidate(name, slot, 0, &total);
        uint32_t first = FN_NONE; uint64_t ignorable = ~0ull;
        for (uint32_t c = 0; c < total; c++) {
            uint32_t f = candidate(name, slot, c, &total);
            if (f == FN_NONE || !candidate_ok(f, argv, types_in, n)) continue;
            if (first == FN_NONE) first = f;
            uint32_t c1 = fn_arg_cursor(first), c2 = fn_arg_cursor(f);
            for (uint32_t i = 0; i < n; i++) {
                uint32_t t1 = fn_arg_type(first, c1), t2 = fn_arg_type(f, c2);
                if ((ignorable >> i) & 1) if (!T().types_match_allow_generic(t2, t1)) ignorable &= ~(1ull << i);
            }
        }
        if (first == FN_NONE) { FN_REPORT(SE_NoOverloadForFunction); return FN_NONE; }
        for (uint32_t c = 0; c < total; c++) {
            uint32_t f = candidate(name, slot, c, &total);
            if (f == FN_NONE || !candidate_ok(f, argv, types_in, n)) continue;
            bool all = true;
            uint32_t cf = fn_arg_cursor(f);
            for (uint32_t i = 0; i < n; i++) {
                uint32_t ft = fn_arg_type(f, cf);
                if ((ignorable >> i) & 1) continue;
                if (!T().types_match_default_generic(ft, ARG_T(i))) { all = false; break; }
            }
            if (all) return f;
        }
        FN_REPORT(SE_AmbiguousOverload);
        return FN_NONE;
#undef FN_REPORT
#undef FN_IDENT
#undef ARG_T
    }

    // An operator applied to primitive operands when no user overload of that symbol exists: call_variables below reduces
    // to this (no local can shadow a symbol unless sym_locals; overload choice as in compiler_lookup_function's first
    // block; the autocasts of primitives create no variables).  Everything else goes through the general helper.
    __device__ __forceinline__ Callee call_op(uint32_t sp_low, uint32_t sp_high, uint8_t sym, uint32_t a0, uint32_t a1, uint32_t n, uint32_t span_node, bool same_span) {
        if (!sym_locals && !((user_syms >> sym) & 1) && builtin_count(sym) > 0 && n == builtin_nargs(sym)) {
            const uint32_t t0 = vtype(a0), t1 = n > 1 ? vtype(a1) : t0;
            if (t0 < TP_COUNT && t1 < TP_COUNT && t0 != TP_Error && t1 != TP_Error) {
                const int idx = builtin_pick(sym, t0, t1);
                if (idx >= 0) {
                    const uint32_t want = builtin_arg_type(sym, idx);
                    Callee res{};
                    res.a0 = autocast(a0, want, span_node);
                    res.a1 = a1;
                    if (n > 1) {
                        uint32_t sn = span_node;
                        if (!same_span && sn != OPUSFE_NONE && N(sn).next != OPUSFE_NONE) sn = N(sn).next;
                        res.a1 = autocast(a1, want, sn);
                    }
                    res.fv = OPUSFE_NONE; res.builtin = true; res.ok = true;
                    res.out = new_var(builtin_ret_type(sym, idx), true);
                    return res;
                }
            }
        }
        return call_variables(sp_low, sp_high, key_sym(sym, sym_mix), Args{nullptr, a0, a1}, n, span_node, same_span);
    }
    // IrGenerator::lookup_function + get_function_call_variables (generate_ir.rs:1151-1198).
    // argv[0..n) are updated in place by the autocasts; span of argument i = node `span_node` (same for all when
    // `same_span`) else the i-th node of the sibling chain starting there.  Returns false when there is no callee.
    __device__ Callee call_variables(uint32_t sp_low, uint32_t sp_high, NameKey name, Args argv, uint32_t n, uint32_t span_node, bool same_span) {
        Callee res{}; res.ok = false;
        sp_low = uni(sp_low); sp_high = uni(sp_high); name = uni(name); n = uni(n); span_node = uni(span_node); same_span = uni(same_span);
        LPROF(LP_CALLVARS);
        for (uint32_t i = 0; i < n; i++) if (trec(vtype(argv.get(i))).kind == TY_Error) return res;
        // lookup_function: a local of that name shadows the global functions
        uint32_t fv, ret; bool builtin = false;
        uint32_t arg_src = 0;        // 0: function id `fid`; 1: members of function type `ftype`
        uint32_t fid = FN_NONE; TypeRec ftype{};
        uint32_t lj = (name.tok == OPUSFE_NONE && !sym_locals) ? OPUSFE_NONE : local_find(name);
        if (lj != OPUSFE_NONE) {
            uint32_t variable = l_var(lj);
            uint32_t typ = get_lvalue_type(variable);
            uint32_t nv = new_var(typ, true);
            push(ins(I_Load, nv, variable));
            ftype = trec(typ);
            if (ftype.kind != TY_Function) { einfo(OPUSFE_NONE, 0, typ); report(SE_CallOfNonFunction, sp_low, sp_high); return res; }
            fv = nv; ret = ftype.sub; arg_src = 1;
        } else {
            fid = compiler_lookup_function(sp_low, sp_high, name, argv, nullptr, n);
            if (fid == FN_NONE) return res;
            ret = fn_ret(fid);
            if (fn_is_builtin(fid)) { fv = OPUSFE_NONE; builtin = true; }
            else {
                uint32_t typ = type_function(fid);
                fv = new_var(typ, true);
                push(ins(I_Function, fv, 0, 0, fid));
            }
        }
        // zip(arguments, types): the shorter one ends the loop (debug_assert only in debug builds, :1189)
        uint32_t m = n, cursor = 0;
        if (arg_src == 1) { uint32_t k = T().member_count(ftype); if (k < m) m = k; cursor = T().member_first(ftype); }
        else cursor = fn_arg_cursor(fid);
        uint32_t sn = span_node;
        for (uint32_t i = 0; i < m; i++) {
            uint32_t want;
            if (arg_src == 1) { want = T().member_type(cursor); cursor = T().member_next(cursor); }
            else want = fn_arg_type(fid, cursor);
            argv.set(i, autocast(argv.get(i), want, sn));
            if (!same_span && sn != OPUSFE_NONE && N(sn).next != OPUSFE_NONE) sn = N(sn).next;
        }
        res.fv = fv; res.out = new_var(ret, true); res.builtin = builtin; res.a0 = argv.a0; res.a1 = argv.a1; res.ok = true;
        return res;
    }

    // get_field_type (:1090-1101) -> type id or NONE
    __device__ uint32_t get_field_type(uint32_t rec, uint32_t field_tok) {
        rec = uni(rec); field_tok = uni(field_tok);
        const TypeRec r = trec(get_lvalue_type(rec));
        if (r.kind != TY_Record) return OPUSFE_NONE;
        uint32_t n = T().member_count(r), c = T().member_first(r);
        for (uint32_t i = 0; i < n; i++) { if (neq(key_tok(field_tok), key_tok(T().member_name(c)))) return T().member_type(c); c = T().member_next(c); }
        return OPUSFE_NONE;
    }
    // generate_ir_for_field_access (:1103-1149)
    __device__ uint32_t field_access(uint32_t rec, uint32_t field_tok, uint8_t pt, uint32_t sp_low, uint32_t sp_high) {
        rec = uni(rec); field_tok = uni(field_tok); pt = uni(pt); sp_low = uni(sp_low); sp_high = uni(sp_high);
        LPROF(LP_FIELD);
        if (pt == PT_Array || pt == PT_ArrayMutable) { report(SE_ArrayPointerToNonArray, sp_low, sp_high); return generate_error(); }
        const TypeRec r = trec(get_lvalue_type(rec));
        if (r.kind == TY_Record) {
            uint32_t n = T().member_count(r), c = T().member_first(r);
            for (uint32_t i = 0; i < n; i++) {
                if (neq(key_tok(field_tok), key_tok(T().member_name(c)))) {
                    uint32_t rvv = new_var(type_pointer(pt, T().member_type(c)), true);
                    push(ins(I_Field, rvv, rec, 0, field_tok));
                    return rvv;
                }
                c = T().member_next(c);
            }
            einfo(field_tok, 0, get_lvalue_type(rec));
            report(SE_FieldDoesNotExist, sp_low, sp_high);
            return generate_error();
        }
        if (r.kind == TY_String) {
            uint32_t idx, ft;
            if (name_is(S.src, TokView{}, field_tok, "data", 4)) { idx = 0; ft = type_pointer(PT_Array, TP_Natural8); }
            else if (name_is(S.src, TokView{}, field_tok, "length", 6)) { idx = 1; ft = TP_Size; }
            else { einfo(field_tok, 0, get_lvalue_type(rec)); report(SE_FieldDoesNotExist, sp_low, sp_high); return generate_error(); }
            uint32_t rvv = new_var(type_pointer(pt, ft), true);
            push(ins(I_BuiltinField, rvv, rec, 0, idx));
            return rvv;
        }
        einfo(OPUSFE_NONE, 0, get_lvalue_type(rec));
        report(SE_FieldAccessOnNonRecord, sp_low, sp_high);
        return generate_error();
    }

    // lookup_name (:857-893).  Returns false for "None" (the name is a constant global)
    struct Looked { uint32_t var; bool found, lv; };
    __device__ Looked lookup_name(uint32_t name_tok, uint32_t sp_low, uint32_t sp_high, uint8_t pt) {
        name_tok = uni(name_tok); sp_low = uni(sp_low); sp_high = uni(sp_high); pt = uni(pt);
        LPROF(LP_LOOKUPNAME);
        NameKey key = key_tok(name_tok);
        uint32_t lj = local_find(key);
        if (lj != OPUSFE_NONE) return Looked{try_cast_pointer(l_var(lj), pt, sp_low, sp_high), true, true};
        uint32_t ge = last_entry(key, NK_GLOBAL);
        if (ge != OPUSFE_NONE) {
            uint32_t gid = S.entries[ge].kind_index & 0x3FFFFFFFu;
            if (S.globals[gid].is_const) return Looked{0, false, true};
            uint32_t v = new_var(type_mut(S.globals[gid].type_id), true);
            push(ins(I_GlobalVariable, v, 0, 0, gid));
            return Looked{try_cast_pointer(v, pt, sp_low, sp_high), true, true};
        }
        const NameFns nf = name_has_functions(key);
        const uint32_t slot = nf.slot;
        if (nf.has) {
            uint32_t total = 0, count = 0, only = FN_NONE;
            candidate(key, slot, 0, &total);
            for (uint32_t c = 0; c < total; c++) { uint32_t f = candidate(key, slot, c, &total); if (f != FN_NONE) { count++; only = f; } }
            if (count > 1) { report_tok(SE_AmbiguousFunctionReference, name_tok); return Looked{generate_error(), true, true}; }
            if (fn_is_builtin(only)) { unsupported(); return Looked{generate_error(), true, true}; }   // single builtin (`not`): the reference builds a function value of a builtin
            uint32_t v = new_var(type_function(only), true);
            push(ins(I_Function, v, 0, 0, only));
            return Looked{v, true, false};
        }
        report_tok(SE_UndefinedVariable, name_tok);
        return Looked{generate_error(), true, true};
    }

    // ---- machine ----------------------------------------------------------------------------------
    __device__ __forceinline__ bool call(uint8_t entry, uint8_t resume, uint32_t node, uint8_t aux, uint32_t a, uint32_t f = 0) {
        LPROF(LP_FRAMES);
        cur.st = resume;
        if (sp + 1 >= fcap) { internal(); return false; }
        if (sp < LDS_FRAMES) lds.frames[sp] = cur; else gst(frames + sp, (LFrame)cur);
        sp++;
        cur.st = entry; cur.aux = aux; cur.cnt = 0; cur.node = node; cur.a = a; cur.b = cur.c = cur.d = cur.e = 0; cur.f = f;
        return true;
    }
    __device__ __forceinline__ void retn() { LPROF(LP_FRAMES); if (sp == 0) { cur.st = LS_DONE; return; } --sp; cur = uni(sp < LDS_FRAMES ? lds.frames[sp] : gld(frames + sp)); }
    __device__ __forceinline__ void push_raw(const LFrame& fr) { if (sp + 1 >= fcap) { internal(); return; } if (sp < LDS_FRAMES) lds.frames[sp] = fr; else gst(frames + sp, fr); sp++; }
    __device__ __forceinline__ LFrame pop_raw() { if (sp == 0) { internal(); return LFrame{}; } --sp; return uni(sp < LDS_FRAMES ? lds.frames[sp] : gld(frames + sp)); }

    // result of a P frame with flags `aux` (f = node for InvalidLvalue) -> rv / rl
    __device__ void p_result(uint32_t var, bool lv, uint8_t aux, uint32_t f) {
        var = uni(var); lv = uni(lv); aux = uni(aux); f = uni(f);
        if ((aux & PF_MUSTLV) && !lv) { report_node(SE_InvalidLvalue, f); var = generate_error(); }   // generate_pointer_from_expression :970-979
        if (aux & PF_DEREF) var = lv ? dereference_pointer(var) : var;                             // :577-585
        rv = var; rl = lv;
    }
    __device__ __forceinline__ void p_finish(uint32_t var, bool lv) { p_result(var, lv, cur.aux, cur.f); retn(); }

    // Leaves of the expression tree are lowered where their parent asks for them: no frame push / pop, no machine step.
    // (Same code as the E_Variable case of LS_P and the E_Integer case of LS_E, which call these too.)
    __device__ void p_variable(uint32_t node, uint8_t aux, uint32_t f) {
        node = uni(node); aux = uni(aux); f = uni(f);
        const opusfe_node vn = N(node);
        const Looked lk = lookup_name(vn.payload, vn.low, vn.high, aux & 3);
        if (lk.found) { p_result(lk.var, lk.lv, aux, f); return; }
        // constant global (:993-998)
        uint32_t ge = last_entry(key_tok(vn.payload), NK_GLOBAL);
        uint32_t gid = S.entries[ge].kind_index & 0x3FFFFFFFu;
        uint32_t v = new_var(S.globals[gid].type_id, true);
        push(ins(I_GlobalConstant, v, 0, 0, gid));
        p_result(v, false, aux, f);
    }
    __device__ uint32_t e_integer(uint32_t node, uint32_t expected) {
        node = uni(node); expected = uni(expected);
        const opusfe_token t = TK(N(node).payload);
        uint32_t typ;
        if (t.aux & 0x8000) {
            uint32_t size = t.aux & 0xFF; bool sg = (t.aux & 0x100) != 0;
            if (!sg) typ = size == 8 ? TP_Natural8 : size == 16 ? TP_Natural16 : size == 32 ? TP_Natural32 : size == 64 ? TP_Natural64 : TP_Size;
            else typ = size == 8 ? TP_Integer8 : size == 16 ? TP_Integer16 : size == 32 ? TP_Integer32 : size == 64 ? TP_Integer64 : TP_Offset;
        } else if (expected != OPUSFE_NONE && (kind_is_integral(trec(expected).kind) || trec(expected).kind == TY_GenericInteger)) typ = expected;
        else typ = TP_Integer32;
        uint64_t val = (t.flags & OPUSFE_TF_INLINE) ? (uint64_t)t.payload : S.val64[t.payload];
        uint32_t v = new_var(typ, true);
        push(ins(I_Integer, v, 0, 0, (uint32_t)val, (uint32_t)(val >> 32)));
        return v;
    }

    // generate_ir_from_function (:133-174)
    __device__ void run(uint32_t body) {
#define CALL_E(node, exp, resume) do { const uint32_t n_ = uni(node); \
        if (n_ != OPUSFE_NONE) { const uint8_t k_ = N(n_).kind; \
            if (k_ == E_Variable) { p_variable(n_, PT_Reference | PF_DEREF, n_); cur.st = resume; goto next; } \
            if (k_ == E_Integer) { rv = e_integer(n_, exp); cur.st = resume; goto next; } } \
        if (!call(LS_E, resume, n_, 0, exp)) return; goto next; } while (0)
#define CALL_P(node, ptflags, resume) do { const uint32_t n_ = uni(node); \
        if (n_ != OPUSFE_NONE && N(n_).kind == E_Variable) { p_variable(n_, ptflags, n_); cur.st = resume; goto next; } \
        if (!call(LS_P, resume, n_, ptflags, 0, n_)) return; goto next; } while (0)
#define CALL_BLOCK(node, resume) do { if (!call(LS_BLOCK, resume, node, 0, 0)) return; goto next; } while (0)
#define CALL_STMT(node, resume) do { if (!call(LS_STMT, resume, node, 0, 0)) return; goto next; } while (0)
        const uint32_t def = F.def_node;
        nbbs = 0; new_bb(); bb_index = 0;
        {   // prologue
            uint32_t a = F.first_arg;
            for (uint32_t i = 0; i < F.nargs; i++) { new_var(resolve_type(N(a).first_child, undefined_flag), true); a = N(a).next; }
            a = F.first_arg;
            for (uint32_t i = 0; i < F.nargs; i++) {
                uint32_t v = new_var(type_mut(vtype(i)), false);
                push(ins(I_Allocate, v));
                push(ins(I_Store, 0, v, i));
                local_set(N(a).payload, v);
                a = N(a).next;
            }
        }
        const bool returns_null = trec(F.ret).kind == TY_Null;
        cur.st = LS_BLOCK; cur.aux = 0; cur.cnt = 0; cur.node = body; cur.a = cur.b = cur.c = cur.d = cur.e = cur.f = 0;
        sp = 0;
#ifdef OPUSFE_LOWER_PROF
        uint64_t step_t0 = __builtin_readcyclecounter(); uint32_t step_state = 47;
#endif
        for (;;) {
#ifdef OPUSFE_LOWER_PROF
            { const uint64_t t = __builtin_readcyclecounter(); prof[LP_STATE_CYCLES + step_state] += t - step_t0; prof[LP_STATE_STEPS + step_state]++; step_t0 = t; step_state = cur.st < 47 ? cur.st : 47; }
#endif
            if (dev_status >= DS_INTERNAL_CAPACITY) return;
            // the node of the current frame, loaded where a state uses it (several states never look at it; the loads are
            // scalar and invariant, so repeated uses inside one state are merged by the compiler)
            const uint32_t cnode = uni(cur.node);
#define nd (cnode != OPUSFE_NONE ? N(cnode) : opusfe_node{})
            switch (cur.st) {
            case LS_DONE: goto finished;
            // ---- generate_ir_from_block (:189-203) ----
            case LS_BLOCK: cur.a = nstack; cur.b = cur.node != OPUSFE_NONE ? nd.first_child : OPUSFE_NONE; cur.aux = 0; cur.st = LS_BLOCK_LOOP; goto next;
            case LS_BLOCK_LOOP:
                if (cur.b == OPUSFE_NONE) { end_scope(cur.a); rdiv = cur.aux != 0; retn(); goto next; }
                CALL_STMT(cur.b, LS_BLOCK_ITEM);
            case LS_BLOCK_ITEM: if (rdiv) cur.aux = 1; cur.b = N(cur.b).next; cur.st = LS_BLOCK_LOOP; goto next;

            // ---- generate_ir_from_statement (:246-493) ----
            case LS_STMT:
                switch (nd.kind) {
                    case S_Expression: CALL_E(nd.first_child, OPUSFE_NONE, LS_ST_EXPR_1);
                    case S_VariableDefinition: CALL_E(nd.first_child, OPUSFE_NONE, LS_ST_VDEF_1);
                    case S_Assignment: CALL_P(nd.first_child, PT_Mutable | PF_MUSTLV, LS_ST_ASG_1);
                    case S_Return: CALL_E(nd.first_child, F.ret, LS_ST_RET_1);
                    case S_If: CALL_E(nd.first_child, TP_Bool, LS_ST_IF_1);
                    case S_While: {
                        uint32_t loop_start = nbbs;
                        push(ins(I_Jump, 0, 0, 0, loop_start));
                        bb_index = loop_start; new_bb();
                        cur.b = loop_start;
                        CALL_E(nd.first_child, TP_Bool, LS_ST_WH_1);
                    }
                    case S_For: {
                        uint32_t c = nd.first_child;
                        if (nd.sub) c = N(c).next;               // skip the optional type
                        CALL_E(c, OPUSFE_NONE, LS_ST_FOR_1);
                    }
                    case S_Break:
                        if (loop_begin != OPUSFE_NONE) {
                            uint32_t t = push(ins(I_BreakPlaceholder, 0, break_head));   // chain of this loop's breaks (patched in LS_FL_2)
                            break_head = t;
                            bb_index = new_bb();
                        } else report_node(SE_BreakOutsideLoop, cur.node);
                        rdiv = false; retn(); goto next;
                    case S_Continue:
                        if (loop_begin != OPUSFE_NONE) { push(ins(I_Jump, 0, 0, 0, loop_begin)); bb_index = new_bb(); }
                        else report_node(SE_ContinueOutsideLoop, cur.node);
                        rdiv = false; retn(); goto next;
                    default: internal(); return;
                }
            case LS_ST_EXPR_1: rdiv = false; retn(); goto next;
            case LS_ST_VDEF_1: {
                uint32_t v = rv, typ = vtype(v);
                if (contains_local(nd.payload)) { einfo(nd.payload, 0, OPUSFE_NONE); report_node(SE_ShadowedName, cur.node); }
                else {
                    uint32_t nv = new_var(type_mut(typ), false);
                    push(ins(I_Allocate, nv));
                    push(ins(I_Store, 0, nv, v));
                    stack_push(nd.payload);
                    local_set(nd.payload, nv);
                }
                rdiv = false; retn(); goto next;
            }
            case LS_ST_ASG_1: {
                cur.a = rv; cur.b = get_lvalue_type(rv);
                CALL_E(N(nd.first_child).next, cur.b, LS_ST_ASG_2);
            }
            case LS_ST_ASG_2: {
                uint32_t fin = autocast(rv, cur.b, N(nd.first_child).next);
                push(ins(I_Store, 0, cur.a, fin));
                rdiv = false; retn(); goto next;
            }
            case LS_ST_RET_1: {
                uint32_t r = autocast(rv, F.ret, nd.first_child);
                push(ins(I_Return, 0, r));
                rdiv = true; retn(); goto next;       // no new bb after `return` (Q12)
            }
            case LS_ST_IF_1: {
                cur.a = autocast(rv, TP_Bool, nd.first_child);
                cur.b = bb_index;
                cur.c = new_bb(); bb_index = cur.c;
                CALL_BLOCK(N(nd.first_child).next, LS_ST_IF_2);
            }
            case LS_ST_IF_2: {
                cur.aux = rdiv ? 1 : 0;
                cur.d = bb_index;
                cur.e = new_bb(); bb_index = cur.e;
                uint32_t else_block = N(N(nd.first_child).next).next;
                CALL_BLOCK(else_block, LS_ST_IF_3);
            }
            case LS_ST_IF_3: {
                bool div = cur.aux && rdiv;
                uint32_t else_end = bb_index;
                uint32_t merge = new_bb(); bb_index = merge;
                push_to(cur.b, ins(I_Branch, 0, cur.a, cur.c, cur.e));
                push_to(cur.d, ins(I_Jump, 0, 0, 0, merge));
                push_to(else_end, ins(I_Jump, 0, 0, 0, merge));
                rdiv = div; retn(); goto next;
            }
            case LS_ST_WH_1: {
                const opusfe_node cn = N(nd.first_child);
                cur.a = autocast(rv, TP_Bool, nd.first_child);
                cur.aux = (cn.kind == E_Bool && cn.sub) ? 1 : 0;      // is_infinite; bit1 = is_for
                cur.e = 0;                                            // scope (unused for while)
                cur.d = N(nd.first_child).next;                       // then-block node
                cur.st = LS_FL_0;
                goto next;
            }
            case LS_ST_FOR_1: {   // :363-431
                uint32_t tnode = nd.sub ? nd.first_child : OPUSFE_NONE;
                uint32_t inode = nd.sub ? N(nd.first_child).next : nd.first_child;
                const uint32_t elow = N(inode).low, ehigh = N(inode).high;
                uint32_t off = reserve_cargs(1);
                gst(cargs + off, rv);
                uint32_t for_fn, ret_var;
                Callee cl = call_variables(elow, ehigh, key_sym(T_For, sym_mix), Args{cargs + off, 0, 0}, 1, inode, true);
                if (cl.ok) { for_fn = cl.fv; ret_var = cl.out; }
                else { for_fn = generate_error(); ret_var = generate_error(); }
                uint32_t iterator_variable = new_var(type_mut(vtype(ret_var)), true);
                push(ins(I_Call, ret_var, for_fn, 0, off, 0, 1));
                push(ins(I_Allocate, iterator_variable));
                push(ins(I_Store, 0, iterator_variable, ret_var));
                uint32_t loop_start = nbbs;
                push(ins(I_Jump, 0, 0, 0, loop_start));
                bb_index = loop_start; new_bb();
                uint32_t scope = nstack;
                uint32_t pointer_type = tnode != OPUSFE_NONE ? type_mut(resolve_type(tnode, undefined_flag)) : (uint32_t)TP_Generic;
                uint32_t iteration_variable;
                if (contains_local(nd.payload)) { einfo(nd.payload, 0, OPUSFE_NONE); report_node(SE_ShadowedName, cur.node); iteration_variable = generate_error(); }
                else {
                    iteration_variable = new_var(pointer_type, false);
                    push(ins(I_Allocate, iteration_variable));
                    stack_push(nd.payload);
                    local_set(nd.payload, iteration_variable);
                }
                uint32_t off2 = reserve_cargs(2);
                gst(cargs + off2, iterator_variable); gst(cargs + off2 + 1, iteration_variable);
                uint32_t cont_fn, cond;
                cl = call_variables(elow, ehigh, key_sym(T_Continue, sym_mix), Args{cargs + off2, 0, 0}, 2, inode, true);
                if (cl.ok) { cont_fn = cl.fv; cond = cl.out; }
                else { cont_fn = generate_error(); cond = generate_error(); }
                push(ins(I_Call, cond, cont_fn, 0, off2, 0, 2));
                cond = autocast(cond, TP_Bool, inode);
                cur.a = cond; cur.b = loop_start; cur.aux = 2; cur.e = scope;
                cur.d = N(inode).next;                                // then-block node
                cur.st = LS_FL_0;
                goto next;
            }
            // ---- shared tail of While (:327-361) and For (:432-468) ----
            // frame: a cond, b loop_start, c branch, d then-block node -> then bb, e scope, f then_end; aux bit0 infinite, bit1 for
            case LS_FL_0: {
                cur.c = bb_index;
                LFrame save{}; save.a = break_head; save.b = loop_begin;
                break_head = OPUSFE_NONE; loop_begin = cur.b;
                uint32_t then_block = cur.d;
                cur.d = new_bb(); bb_index = cur.d;
                cur.f = then_block;                                   // remembered to find the else block
                push_raw(save);
                CALL_BLOCK(then_block, LS_FL_1);
            }
            case LS_FL_1: {
                LFrame save = pop_raw();
                uint32_t then_end = bb_index;
                uint32_t new_head = break_head;
                break_head = save.a; loop_begin = save.b;
                if (cur.aux & 2) end_scope(cur.e);
                uint32_t else_block = N(cur.f).next;
                uint32_t else_branch = new_bb(); bb_index = else_branch;
                LFrame keep{}; keep.a = then_end; keep.b = new_head; keep.c = else_branch;
                push_raw(keep);
                CALL_BLOCK(else_block, LS_FL_2);
            }
            case LS_FL_2: {
                LFrame keep = pop_raw();
                bool div = rdiv;
                if (!(cur.aux & 2)) div = div || ((cur.aux & 1) && keep.b == OPUSFE_NONE);
                uint32_t else_end = bb_index;
                uint32_t merge = new_bb(); bb_index = merge;
                push_to(cur.c, ins(I_Branch, 0, cur.a, cur.d, keep.c));
                push_to(keep.a, ins(I_Jump, 0, 0, 0, cur.b));
                push_to(else_end, ins(I_Jump, 0, 0, 0, merge));
                for (uint32_t t = keep.b; t != OPUSFE_NONE;) {        // patch this loop's break placeholders (:357-359)
                    if (t >= ninstrs) { internal(); break; }
                    uint32_t nx = instr_get(t).a;
                    instr_set(t, ins(I_Jump, 0, 0, 0, merge));
                    t = nx;
                }
                rdiv = div; retn(); goto next;
            }

            // ---- generate_ir_from_expression (:510-806); frame.a = expected type or NONE ----
            case LS_E:
                switch (nd.kind) {
                    case E_Integer: rv = e_integer(cnode, cur.a); retn(); goto next;
                    case E_Float: {
                        const opusfe_token t = TK(nd.payload);
                        uint64_t bits = S.val64[t.payload];
                        uint32_t v = new_var(t.aux == 32 ? TP_Float32 : TP_Float64, true);
                        push(ins(I_Float, v, 0, 0, (uint32_t)bits, (uint32_t)(bits >> 32)));
                        rv = v; retn(); goto next;
                    }
                    case E_Null: { uint32_t v = new_var(TP_Null, true); push(ins(I_Null, v)); rv = v; retn(); goto next; }
                    case E_Bool: { uint32_t v = new_var(TP_Bool, true); push(ins(I_Bool, v, 0, 0, 0, 0, nd.sub)); rv = v; retn(); goto next; }
                    case E_String: { uint32_t v = new_var(TP_String, true); push(ins(I_String, v, 0, 0, nd.payload)); rv = v; retn(); goto next; }
                    case E_Variable: case E_Dereference: case E_VariableDefinition: case E_Field: case E_NamedCall: case E_CallOrIndex: case E_Record:
                        cur.st = LS_P; cur.aux = PT_Reference | PF_DEREF; cur.f = cur.node; goto next;
                    case E_Proc: {   // :586-602
                        if (nd.nchild > 64) { unsupported(); rv = generate_error(); retn(); goto next; }
                        const uint32_t saved_cargs = ncargs;
                        uint32_t* types = cargs + reserve_cargs(nd.nchild);    // scratch use of the pool, released below
                        uint32_t c = nd.first_child;
                        for (uint32_t i = 0; i < nd.nchild; i++) { types[i] = resolve_type(c, undefined_flag); c = N(c).next; }
                        uint32_t f = compiler_lookup_function(nd.low, nd.high, key_tok(nd.payload), Args{nullptr, 0, 0}, types, nd.nchild);
                        ncargs = saved_cargs;
                        if (f == FN_NONE) { rv = generate_error(); retn(); goto next; }
                        if (fn_is_builtin(f)) { report_node(SE_ReferenceToBuiltinFunction, cur.node); rv = generate_error(); retn(); goto next; }
                        uint32_t v = new_var(type_function(f), true);
                        push(ins(I_Function, v, 0, 0, f));
                        rv = v; retn(); goto next;
                    }
                    case E_Operator: {
                        uint32_t left = nd.first_child, right = N(left).next;
                        switch (nd.sub) {
                            case OP_Is: CALL_E(left, OPUSFE_NONE, LS_OP_IS_1);
                            case OP_And: case OP_Or: CALL_E(left, TP_Bool, LS_OP_AND_1);
                            case OP_Not: CALL_E(right, TP_GenericInteger, LS_OP_NOT_1);      // the left operand is never evaluated (Q9)
                            default: CALL_E(left, TP_GenericInteger, LS_OP_BIN_1);
                        }
                    }
                    case E_Cast: CALL_E(nd.first_child, OPUSFE_NONE, LS_CAST_1);
                    case E_Reference: case E_MutableReference: case E_ArrayReference: case E_MutableArrayReference: {
                        uint8_t pt = nd.kind == E_Reference ? PT_Reference : nd.kind == E_MutableReference ? PT_Mutable : nd.kind == E_ArrayReference ? PT_Array : PT_ArrayMutable;
                        cur.node = nd.first_child; cur.f = cur.node; cur.aux = pt | PF_MUSTLV; cur.st = LS_P;
                        goto next;
                    }
                    case E_Negate: CALL_E(nd.first_child, cur.a, LS_NEG_1);
                    case E_Sizeof: case E_Alignof: {
                        uint32_t v = new_var(TP_Size, true);
                        uint32_t typ = resolve_type(nd.first_child, undefined_flag);
                        push(ins(nd.kind == E_Sizeof ? I_Sizeof : I_Alignof, v, 0, 0, typ));
                        rv = v; retn(); goto next;
                    }
                    case E_Parentheses: cur.node = nd.first_child; goto next;
                    default: internal(); return;
                }
            case LS_OP_IS_1: cur.b = rv; CALL_E(N(nd.first_child).next, OPUSFE_NONE, LS_OP_IS_2);
            case LS_OP_IS_2: {
                uint32_t l = cur.b, r = rv;
                const TypeRec lt = trec(vtype(l)), rt = trec(vtype(r));
                if (lt.kind == TY_Pointer && rt.kind == TY_Pointer && lt.ptr == rt.ptr && T().types_match(lt.sub, rt.sub)) {
                    uint32_t v = new_var(TP_Bool, true);
                    push(ins(I_Equals, v, l, r));
                    rv = v;
                } else { einfo(0x80000000u | (uint32_t)T_Is, 0, vtype(l), vtype(r)); report_node(SE_InvalidOperandTypes, cur.node); rv = generate_error(); }
                retn(); goto next;
            }
            case LS_OP_AND_1: {
                cur.b = autocast(rv, TP_Bool, nd.first_child);
                cur.c = bb_index;
                cur.d = new_bb(); bb_index = cur.d;
                CALL_E(N(nd.first_child).next, TP_Bool, LS_OP_AND_2);
            }
            case LS_OP_AND_2: {
                uint32_t then_var = rv;
                if (!T().types_match(vtype(then_var), TP_Bool)) {       // Q15: returns with bb_index left in the RHS block, no Branch
                    einfo(OPUSFE_NONE, 0, vtype(then_var));
                    report_node(SE_InvalidOperandType, N(nd.first_child).next);
                    rv = generate_error(); retn(); goto next;
                }
                uint32_t then_end = bb_index;
                uint32_t merge = new_bb();
                push(ins(I_Jump, 0, 0, 0, merge));
                if (nd.sub == OP_And) push_to(cur.c, ins(I_Branch, 0, cur.b, cur.d, merge));
                else push_to(cur.c, ins(I_Branch, 0, cur.b, merge, cur.d));
                bb_index = merge;
                uint32_t out = new_var(TP_Bool, true);
                push(ins(I_Phi, out, cur.b, then_var, cur.c, then_end));
                rv = out; retn(); goto next;
            }
            case LS_OP_NOT_1: {
                uint32_t right = N(nd.first_child).next;
                const Callee cl = call_op(nd.low, nd.high, T_Not, rv, 0, 1, right, true);
                if (cl.ok) {
                    if (cl.builtin) push(ins(I_Not, cl.out, cl.a0));
                    else { uint32_t off = reserve_cargs(1); gst(cargs + off, cl.a0); push(ins(I_Call, cl.out, cl.fv, 0, off, 0, 1)); }
                    rv = cl.out;
                } else rv = generate_error();
                retn(); goto next;
            }
            case LS_OP_BIN_1: cur.b = rv; CALL_E(N(nd.first_child).next, TP_GenericInteger, LS_OP_BIN_2);
            case LS_OP_BIN_2: {
                const Callee cl = call_op(nd.low, nd.high, sym_of_op(nd.sub), cur.b, rv, 2, nd.first_child, false);
                if (cl.ok) {
                    if (cl.builtin) {
                        uint16_t op;
                        switch (nd.sub) {
                            case OP_Plus: op = I_Add; break; case OP_Minus: op = I_Subtract; break; case OP_Times: op = I_Multiply; break;
                            case OP_Divide: op = I_Divide; break; case OP_Modulo: op = I_Modulo; break; case OP_Equals: op = I_Equals; break;
                            case OP_LessThan: op = I_LessThan; break; case OP_GreaterThan: op = I_GreaterThan; break;
                            case OP_LessThanEquals: op = I_LessThanEquals; break; default: op = I_GreaterThanEquals; break;
                        }
                        push(ins(op, cl.out, cl.a0, cl.a1));
                    } else { uint32_t off = reserve_cargs(2); gst(cargs + off, cl.a0); gst(cargs + off + 1, cl.a1); push(ins(I_Call, cl.out, cl.fv, 0, off, 0, 2)); }
                    rv = cl.out;
                } else rv = generate_error();
                retn(); goto next;
            }
            case LS_CAST_1: {
                uint32_t v = rv, old_type = vtype(v);
                uint32_t typ = resolve_type(N(nd.first_child).next, undefined_flag);
                int ct = T().can_cast(old_type, typ);
                if (ct >= 0) { uint32_t nv = new_var(typ, true); push(ins(I_Cast, nv, v, 0, 0, 0, (uint16_t)ct)); rv = nv; }
                else { einfo(OPUSFE_NONE, 0, old_type, typ); report_node(SE_InvalidCast, cur.node); rv = generate_error(); }
                retn(); goto next;
            }
            case LS_NEG_1: {   // :749-789
                uint32_t sub = rv, st = vtype(sub);
                uint8_t k = trec(st).kind;
                if (k == TY_Integer8 || k == TY_Integer16 || k == TY_Integer32 || k == TY_Integer64 || k == TY_Offset) {
                    uint32_t out = new_var(st, true);
                    push(ins(I_Negate, out, sub));
                    rv = out;
                } else if (k == TY_GenericInteger) {
                    uint32_t t = bb_index < nbbs ? bb_last(bb_index) : OPUSFE_NONE;
                    opusfe_instr li{};
                    if (t != OPUSFE_NONE) li = instr_get(t);
                    if (t != OPUSFE_NONE && li.opcode == I_Integer) {
                        uint64_t val = (uint64_t)li.c | ((uint64_t)li.extra << 32);
                        val = 0 - val;
                        li.c = (uint32_t)val; li.extra = (uint32_t)(val >> 32);
                        instr_set(t, li);
                        rv = li.dst;
                    } else { unsupported(); rv = generate_error(); }   // reference: unreachable!()
                } else if (k == TY_Error) rv = generate_error();
                else { einfo(OPUSFE_NONE, 0, st); report_node(SE_InvalidOperandType, cur.node); rv = generate_error(); }
                retn(); goto next;
            }

            // ---- try_generate_pointer_from_expression (:985-1088); aux = PtrKind | flags, f = node for InvalidLvalue ----
            case LS_P: {
                const uint8_t pt = cur.aux & 3;
                switch (nd.kind) {
                    case E_Variable: p_variable(cnode, cur.aux, cur.f); retn(); goto next;
                    case E_VariableDefinition: {
                        uint32_t var;
                        if (contains_local(nd.payload)) { einfo(nd.payload, 0, OPUSFE_NONE); report_node(SE_ShadowedName, cur.node); var = generate_error(); }
                        else {
                            uint32_t typ = resolve_type(nd.first_child, undefined_flag);
                            uint32_t v = new_var(type_mut(typ), false);
                            push(ins(I_Allocate, v));
                            stack_push(nd.payload);
                            local_set(nd.payload, v);
                            var = try_cast_pointer(v, pt, nd.low, nd.high);
                        }
                        p_finish(var, true); goto next;
                    }
                    case E_Dereference: CALL_E(nd.first_child, OPUSFE_NONE, LS_P_DEREF_1);
                    case E_Field: CALL_P(nd.first_child, pt | PF_MUSTLV, LS_P_FIELD_1);
                    case E_NamedCall: {
                        if (name_has_functions(key_tok(nd.payload)).has) {
                            cur.a = reserve_cargs(nd.nchild);          // arguments are evaluated straight into the call-argument pool
                            cur.b = nd.first_child; cur.c = 0;
                            cur.st = LS_P_NC_LOOP;
                            goto next;
                        }
                        const Looked lk = lookup_name(nd.payload, nd.low, nd.high, pt);
                        if (!lk.found) {
                            report_node(SE_AttemptToCallOrIndexConstant, cur.node);
                            p_finish(generate_error(), true); goto next;
                        }
                        cur.a = lk.var; cur.b = lk.lv ? 1 : 0; cur.c = nd.first_child;
                        cur.st = LS_P_IDX_0;
                        goto next;
                    }
                    case E_CallOrIndex: {
                        uint8_t pt2 = (pt == PT_Array || pt == PT_Reference) ? PT_Reference : PT_Mutable;
                        CALL_P(nd.first_child, pt2, LS_P_COI_1);
                    }
                    case E_Record: {
                        uint32_t typ = type_mut(resolve_type(nd.first_child, undefined_flag));
                        uint32_t out = new_var(typ, true);
                        push(ins(I_Allocate, out));
                        cur.a = out; cur.b = N(nd.first_child).next;
                        cur.st = LS_P_REC_LOOP;
                        goto next;
                    }
                    case E_Parentheses: cur.node = nd.first_child; goto next;
                    default: CALL_E(cur.node, OPUSFE_NONE, LS_P_DEFAULT_1);
                }
            }
            case LS_P_DEFAULT_1: p_finish(rv, false); goto next;
            case LS_P_DEREF_1: {
                const opusfe_node c = N(nd.first_child);
                uint32_t var = try_cast_pointer(rv, cur.aux & 3, c.low, c.high);
                p_finish(var, true); goto next;
            }
            case LS_P_FIELD_1: { uint32_t var = field_access(rv, nd.payload, cur.aux & 3, nd.low, nd.high); p_finish(var, true); goto next; }
            case LS_P_NC_LOOP:
                if (cur.b != OPUSFE_NONE) CALL_E(cur.b, TP_GenericInteger, LS_P_NC_ITEM);
                {
                    uint32_t var;
                    const Callee cl = call_variables(nd.low, nd.high, key_tok(nd.payload), Args{cargs + cur.a, 0, 0}, nd.nchild, nd.first_child, false);
                    if (cl.ok) {
                        push(ins(I_Call, cl.out, cl.fv, 0, cur.a, 0, nd.nchild));
                        var = cl.out;
                    } else var = generate_error();
                    p_finish(var, false); goto next;
                }
            case LS_P_NC_ITEM: gst(cargs + cur.a + cur.c, rv); cur.c++; cur.b = N(cur.b).next; cur.st = LS_P_NC_LOOP; goto next;
            case LS_P_COI_1: cur.a = rv; cur.b = rl ? 1 : 0; cur.c = N(nd.first_child).next; cur.st = LS_P_IDX_0; goto next;
            // ---- generate_ir_for_index (:895-967): a indexee, b is_lvalue | mode << 1, c argument cursor, d last value / pool offset, e count ----
            case LS_P_IDX_0: {
                bool il = cur.b & 1;
                uint32_t it = vtype(cur.a);
                uint32_t indexee_type = it;
                if (il) { const TypeRec r = trec(it); if (r.kind == TY_Pointer) indexee_type = r.sub; else { if (r.kind != TY_Error) unsupported(); indexee_type = TP_Error; } }
                const TypeRec r = trec(indexee_type);
                uint32_t isl, ish;
                if (nd.kind == E_NamedCall) { const opusfe_token ct = TK(nd.payload); isl = ct.low; ish = isl + ct.len; } else { isl = N(nd.first_child).low; ish = N(nd.first_child).high; }
                if (r.kind == TY_Pointer) { if (il) cur.a = dereference_pointer(cur.a); cur.b = (IDX_PTR << 1); }
                else if (r.kind == TY_Array) {
                    if (!il) { report_node(SE_InvalidLvalue, cur.node); p_finish(generate_error(), true); goto next; }
                    cur.b = (IDX_ARR << 1);
                } else if (r.kind == TY_Function) {
                    if (il) cur.a = dereference_pointer(cur.a);
                    uint32_t n = 0; for (uint32_t c = cur.c; c != OPUSFE_NONE; c = N(c).next) n++;
                    cur.b = (IDX_FN << 1);
                    cur.d = reserve_cargs(n);
                } else { einfo(OPUSFE_NONE, 0, indexee_type); report(SE_InvalidOperandType, isl, ish); p_finish(generate_error(), true); goto next; }
                cur.e = 0;
                cur.st = LS_P_IDX_LOOP;
                goto next;
            }
            case LS_P_IDX_LOOP: {
                uint32_t mode = cur.b >> 1;
                if (cur.c != OPUSFE_NONE) CALL_E(cur.c, mode == IDX_FN ? OPUSFE_NONE : (uint32_t)TP_Size, LS_P_IDX_ITEM);
                const uint8_t pt = cur.aux & 3;
                uint32_t isl, ish;
                if (nd.kind == E_NamedCall) { const opusfe_token ct = TK(nd.payload); isl = ct.low; ish = isl + ct.len; } else { isl = N(nd.first_child).low; ish = N(nd.first_child).high; }
                if (mode == IDX_FN) {
                    const TypeRec r = trec(vtype(cur.a));
                    uint32_t out = new_var(r.sub, true);
                    push(ins(I_Call, out, cur.a, 0, cur.d, 0, (uint16_t)cur.e));
                    if (cur.e > 0xFFFF) unsupported();
                    p_finish(out, false); goto next;
                }
                if (cur.e != 1) report_node(SE_InvalidNumberOfArgs, cur.node);
                if (cur.e == 0) { unsupported(); p_finish(generate_error(), true); goto next; }   // reference: pop().unwrap() panics
                const TypeRec r = trec(vtype(cur.a));
                if (mode == IDX_PTR) {
                    uint32_t result_type;
                    if (r.ptr == PT_Array) result_type = type_pointer(PT_Array, r.sub);
                    else if (r.ptr == PT_ArrayMutable) result_type = type_pointer(PT_ArrayMutable, r.sub);
                    else { einfo(OPUSFE_NONE, 0, vtype(cur.a)); report(SE_InvalidOperandType, isl, ish); p_finish(generate_error(), true); goto next; }
                    uint32_t v = new_var(result_type, true);
                    push(ins(I_IndexPointer, v, cur.a, cur.d));
                    p_finish(try_cast_pointer(v, pt, isl, ish), true); goto next;
                }
                // array: the indexee is an lvalue pointer to the array
                if (r.kind != TY_Pointer) { unsupported(); p_finish(generate_error(), true); goto next; }
                uint32_t result_type = type_pointer(r.ptr, trec(r.sub).sub);
                uint32_t v = new_var(result_type, true);
                push(ins(I_IndexArray, v, cur.a, cur.d));
                p_finish(try_cast_pointer(v, pt, isl, ish), true); goto next;
            }
            case LS_P_IDX_ITEM:
                if ((cur.b >> 1) == IDX_FN) gst(cargs + cur.d + cur.e, rv); else cur.d = rv;
                cur.e++; cur.c = N(cur.c).next; cur.st = LS_P_IDX_LOOP; goto next;
            // ---- record literal (:1064-1078): a out, b field-init cursor, c value ----
            case LS_P_REC_LOOP: {
                if (cur.b == OPUSFE_NONE) { p_finish(cur.a, true); goto next; }
                uint32_t ft = get_field_type(cur.a, N(cur.b).payload);
                CALL_E(N(cur.b).first_child, ft, LS_P_REC_ITEM);
            }
            case LS_P_REC_ITEM: {
                const opusfe_node val = N(N(cur.b).first_child);
                uint32_t fv = field_access(cur.a, N(cur.b).payload, PT_Mutable, val.low, val.high);
                push(ins(I_Store, 0, fv, rv));
                cur.b = N(cur.b).next; cur.st = LS_P_REC_LOOP; goto next;
            }
            default: internal(); return;
            }
        next:;
        }
    finished:
        // epilogue (:151-171)
        if (rdiv || returns_null) {
            uint32_t t = bb_last(bb_index);
            bool term = false;
            if (t != OPUSFE_NONE) { uint16_t op = instr_get(t).opcode; term = op == I_Return || op == I_Jump || op == I_Branch || op == I_Unreachable; }
            if (!term) {
                if (returns_null) { uint32_t v = new_var(TP_Null, true); push(ins(I_Null, v)); push(ins(I_Return, 0, v)); }
                else push(ins(I_Unreachable));
            }
        } else report_node(SE_FunctionMightNotReturn, def);
#undef CALL_E
#undef nd
#undef CALL_P
#undef CALL_BLOCK
#undef CALL_STMT
    }
};

// which operator symbols have user-defined overloads (one lane per symbol)
__global__ void k_sem_user_syms(SemView S, uint64_t* __restrict__ out) {
    const uint8_t SYMS[11] = {T_Plus, T_Minus, T_Asterisk, T_Slash, T_Percent, T_EqualsEquals, T_LessThan, T_GreaterThan, T_LessThanEquals, T_GreaterThanEquals, T_Not};
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint64_t m = 0;
        for (int i = 0; i < 11; i++) {
            uint32_t s = name_find(S, key_sym(SYMS[i], 0));
            if (s == OPUSFE_NONE) continue;
            const NameSlot sl = S.slots[s];
            for (uint32_t j = 0; j < sl.count; j++) if ((S.entries[sl.offset + j].kind_index >> 30) == NK_FUNCTION) { m |= 1ull << SYMS[i]; break; }
        }
        *out = m;
    }
}

// one lane per function definition
__global__ MACHINE_ATTR(OPUSFE_LOWER_WAVES) __launch_bounds__(64) void k_lower_functions(SemView S, LowerBuffers B, const uint8_t* __restrict__ undefined_flag, const uint32_t* __restrict__ def_nodes,
                                                         uint32_t n_defs, uint32_t n_nodes, uint32_t n_tokens, const DefCnt* __restrict__ def_excl, FuncIr* __restrict__ out,
                                                         uint32_t* __restrict__ status, uint64_t* __restrict__ prof_out, const uint64_t* __restrict__ user_syms, const uint32_t* __restrict__ order,
                                                         const uint32_t* __restrict__ fb_flag, const FbIdx* __restrict__ fbx) {
    const uint32_t slot = blockIdx.x * (blockDim.x / UNIT_LANES) + threadIdx.x / UNIT_LANES;
    const uint32_t lane = threadIdx.x % UNIT_LANES;
    if (slot >= n_defs) return;
    const uint32_t d = order ? order[slot] : slot;
    const uint32_t dn = def_nodes[d];
    if (S.nodes[dn].kind != D_Function) return;
    const uint32_t fidx = def_excl[d].funcs;
    // only the functions that the node-parallel path (lower_par.h) declined come here; their sparse regions are laid out by
    // (nodes, functions) of the declined functions before them
    if (!fb_flag[fidx]) return;
    const uint32_t sb_n = fbx[fidx].nodes, sb_f = fbx[fidx].funcs;
    const uint32_t dn_end = d + 1 < n_defs ? def_nodes[d + 1] : n_nodes;
    const uint32_t nn = dn_end - dn;
    Lowerer L;      // no fields of its own: the state is the static LDS block above
    L.user_syms = *user_syms;
    L.S = S; L.fidx = fidx; L.F = S.funcs[fidx]; L.undefined_flag = undefined_flag;
    L.vars = B.vars + lower_base(CAP_VARS, sb_n, sb_f); L.nvars = 0; L.vcap = (uint32_t)lower_cap(CAP_VARS, nn);
    uint64_t ib = lower_base(CAP_INSTRS, sb_n, sb_f);
    L.instrs = B.instrs + ib; L.ibb = B.ibb + ib; L.irank = B.irank + ib; L.ninstrs = 0; L.iflushed = 0; L.icap = (uint32_t)lower_cap(CAP_INSTRS, nn);
    uint64_t bbb = lower_base(CAP_BBS, sb_n, sb_f);
    L.bbcnt = B.bbcnt + bbb; L.bblast = B.bblast + bbb; L.nbbs = 0; L.bcap = (uint32_t)lower_cap(CAP_BBS, nn);
    L.tbase = B.type_sparse_base + (uint32_t)lower_base(CAP_TYPES, sb_n, sb_f); L.ntypes = 0; L.tcap = (uint32_t)lower_cap(CAP_TYPES, nn);
    L.cargs = B.cargs + lower_base(CAP_CARGS, sb_n, sb_f); L.ncargs = 0; L.ccap = (uint32_t)lower_cap(CAP_CARGS, nn);
    uint64_t lb = lower_base(CAP_LOCALS, sb_n, sb_f);
    L.lname = B.lname + lb; L.lvar = B.lvar + lb; L.nlocals = 0; L.lcap = (uint32_t)lower_cap(CAP_LOCALS, nn);
    L.lstack = B.lstack + lb; L.nstack = 0;
    L.frames = B.frames + lower_base(CAP_FRAMES, sb_n, sb_f); L.sp = 0; L.fcap = (uint32_t)lower_cap(CAP_FRAMES, nn);
    L.errs = B.errs + lower_base(CAP_ERRS, sb_n, sb_f); L.nerrs = 0; L.ncont = 0; L.ecap = (uint32_t)lower_cap(CAP_ERRS, nn); L.einfo_clear();
    L.bb_index = 0; L.break_head = OPUSFE_NONE; L.loop_begin = OPUSFE_NONE; L.dev_status = DS_OK;
    L.rv = 0; L.rl = false; L.rdiv = false; L.sym_locals = false; L.sym_mix = S.mix_of_def(dn);
    for (uint32_t j = lane; j < (LDS_VARS + 31) / 32; j += UNIT_LANES) L.lds.vnamed[j] = 0;
    // body = last child of the definition
    uint32_t c = S.nodes[dn].first_child;
    while (S.nodes[c].next != OPUSFE_NONE) c = S.nodes[c].next;
#ifdef OPUSFE_LOWER_PROF
    for (int i = 0; i < LP_COUNT; i++) L.prof[i] = 0;
    { uint64_t t0 = __builtin_readcyclecounter(); L.run(c); L.prof[LP_TOTAL] = __builtin_readcyclecounter() - t0; }
    if (lane == 0) for (int i = 0; i < LP_COUNT; i++) atomicAdd((unsigned long long*)prof_out + i, (unsigned long long)L.prof[i]);
#else
    L.run(c);
#endif
    L.flush_instrs();
    L.flush_vars();
    L.flush_types();
    // block sizes of the LDS-resident blocks go to the global table read by k_ir_bb_offsets
    for (uint32_t b = lane; b < L.nbbs && b < LDS_BBS; b += UNIT_LANES) L.bbcnt[b] = L.lds.bbcnt[b];
    if (lane != 0) return;
    FuncIr r{};
    r.def_node = dn; r.func = fidx; r.n_vars = L.nvars; r.n_instrs = L.ninstrs; r.n_bbs = L.nbbs; r.n_types = L.ntypes; r.n_cargs = L.ncargs;
    r.n_errs = L.nerrs < L.ecap ? L.nerrs : L.ecap; r.status = L.dev_status;
    out[fidx] = r;
    if (L.dev_status != DS_OK) atomicMax(status, L.dev_status);      // status = {max lane status, number of semantic errors}
    if (L.nerrs - L.ncont) atomicAdd(status + 1, L.nerrs - L.ncont);
}

// Place of a sequentially lowered function in the dense arrays (same bump allocation as the node-parallel path,
// lower_par.h bump_allocate) and the global instruction offsets of its blocks.  One lane per function.
}  // namespace opusfe
#include "lower_par.h"
namespace opusfe {
__global__ void k_fb_bases(FuncIr* __restrict__ f, uint32_t n, LowerBuffers B, par::ParArgs A, const uint32_t* __restrict__ fb_flag, const FbIdx* __restrict__ fbx) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !fb_flag[i]) return;
    FuncIr r = f[i];
    if (r.n_bbs == 0) return;
    const par::Bases b = par::bump_allocate(A, r.n_vars, r.n_instrs, r.n_bbs, r.n_types, r.n_cargs);
    r.var_base = b.var_base; r.instr_base = b.instr_base; r.bb_base = b.bb_base; r.type_base = b.type_base; r.carg_base = b.carg_base;
    r.err_base = r.n_errs ? atomicAdd(A.err_counter, r.n_errs) : 0u;
    bool fits = b.fits;
    if ((uint64_t)r.err_base + r.n_errs > A.cap_errs) { fits = false; atomicMax(A.overflow, 1u); }
    if (!fits) r.status |= 0x80000000u;              // compaction skips it; the host runs again
    f[i] = r;
    if (!fits) return;
    const uint32_t* cnt = B.bbcnt + lower_base(CAP_BBS, fbx[i].nodes, fbx[i].funcs);
    uint32_t run = r.instr_base;
    for (uint32_t k = 0; k < r.n_bbs; k++) { A.bb_offsets[r.bb_base + k] = run; run += cnt[k]; }
}

struct IrOut { opusfe_variable* vars; opusfe_instr* instrs; uint32_t* cargs; TypeRec* types; SemErr* errs; const uint32_t* bb_offsets; uint32_t n_gtypes; };

// one wave per function (grid-stride)
__global__ __launch_bounds__(256) void k_ir_compact(const FuncIr* __restrict__ f, uint32_t n, LowerBuffers B, const TypeRec* __restrict__ sparse_types, IrOut O,
                                                    const uint32_t* __restrict__ fb_flag, const FbIdx* __restrict__ fbx) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t i = wave; i < n; i += nwaves) {
        const FuncIr r = f[i];
        if (r.n_bbs == 0 || !fb_flag[i] || (r.status & 0x80000000u)) continue;
        const uint32_t sb_n = fbx[i].nodes, sb_f = fbx[i].funcs;
        const uint32_t tsb = B.type_sparse_base + (uint32_t)lower_base(CAP_TYPES, sb_n, sb_f);   // first sparse local type id
        const uint32_t tdb = O.n_gtypes + r.type_base;                                                   // first dense local type id
        #define REMAP_TYPE(x) ((x) < O.n_gtypes ? (x) : (x) - tsb + tdb)
        const opusfe_variable* sv = B.vars + lower_base(CAP_VARS, sb_n, sb_f);
        for (uint32_t j = lane; j < r.n_vars; j += 64) { opusfe_variable v = sv[j]; v.type_id = REMAP_TYPE(v.type_id); O.vars[r.var_base + j] = v; }
        for (uint32_t j = lane; j < r.n_types; j += 64) {
            TypeRec t = sparse_types[tsb + j];
            if (t.kind == TY_Pointer || t.kind == TY_Array || t.kind == TY_Function) t.sub = REMAP_TYPE(t.sub);
            O.types[tdb + j] = t;
        }
        const uint32_t* sc = B.cargs + lower_base(CAP_CARGS, sb_n, sb_f);
        for (uint32_t j = lane; j < r.n_cargs; j += 64) O.cargs[r.carg_base + j] = sc[j];
        const SemErr* se = B.errs + lower_base(CAP_ERRS, sb_n, sb_f);
        for (uint32_t j = lane; j < r.n_errs; j += 64) {
            SemErr e = se[j];
            uint32_t* w = e.kind == SE_Continuation ? &e.low : e.t;      // a continuation record is eight type ids
            const int nw = e.kind == SE_Continuation ? 8 : 4;
            for (int q = 0; q < nw; q++) if (w[q] != OPUSFE_NONE) w[q] = REMAP_TYPE(w[q]);
            O.errs[r.err_base + j] = e;
        }
        const uint64_t ib = lower_base(CAP_INSTRS, sb_n, sb_f);
        for (uint32_t j = lane; j < r.n_instrs; j += 64) {
            opusfe_instr in = B.instrs[ib + j];
            if (in.opcode == I_Sizeof || in.opcode == I_Alignof) in.c = REMAP_TYPE(in.c);
            else if (in.opcode == I_Call) in.c += r.carg_base;
            O.instrs[O.bb_offsets[r.bb_base + B.ibb[ib + j]] + B.irank[ib + j]] = in;
        }
        #undef REMAP_TYPE
    }
}


// which functions the node-parallel path declined, and how much sparse space the sequential machine needs for them
__global__ void k_fb_mark(const FuncIr* __restrict__ f, uint32_t n_funcs, const uint32_t* __restrict__ def_nodes, uint32_t n_defs, uint32_t n_nodes, const DefCnt* __restrict__ def_excl,
                          const opusfe_node* __restrict__ nodes, uint32_t* __restrict__ fb_flag, FbIdx* __restrict__ fbx) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_defs) return;
    const uint32_t dn = def_nodes[d];
    const uint8_t k = nodes[dn].kind;
    if (k != D_Function && k != D_Extern) return;
    const uint32_t fidx = def_excl[d].funcs;
    const bool fb = k == D_Function && f[fidx].status == par::DS_FALLBACK;
    const uint32_t nn = (d + 1 < n_defs ? def_nodes[d + 1] : n_nodes) - dn;
    fb_flag[fidx] = fb ? 1u : 0u;
    fbx[fidx] = FbIdx{fb ? nn : 0u, fb ? 1u : 0u};
}

// The diagnostics of IR generation in the reference's report order (functions in definition order): the passes place each
// function's errors by bump allocation; when there are any (rare) they are copied into definition order.
__global__ void k_err_counts(const FuncIr* __restrict__ f, uint32_t n, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = f[i].n_errs;
}
__global__ void k_err_order(FuncIr* __restrict__ f, uint32_t n, const uint32_t* __restrict__ excl, const SemErr* __restrict__ in, SemErr* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    FuncIr r = f[i];
    for (uint32_t k = 0; k < r.n_errs; k++) out[excl[i] + k] = in[r.err_base + k];
    r.err_base = excl[i];
    f[i] = r;
}

// functions too large for the node-parallel path (known before it runs): the machine starts on them at once, on a second stream
__global__ void k_fb_mark_size(const uint32_t* __restrict__ def_nodes, uint32_t n_defs, uint32_t n_nodes, const DefCnt* __restrict__ def_excl,
                               const opusfe_node* __restrict__ nodes, uint32_t limit, uint32_t* __restrict__ fb_flag, FbIdx* __restrict__ fbx, const uint8_t* __restrict__ skip) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_defs) return;
    const uint32_t dn = def_nodes[d];
    const uint8_t k = nodes[dn].kind;
    if (k != D_Function && k != D_Extern) return;
    const uint32_t fidx = def_excl[d].funcs;
    const uint32_t nn = (d + 1 < n_defs ? def_nodes[d + 1] : n_nodes) - dn;
    const bool big = k == D_Function && nn >= limit && !(skip && skip[d]);
    fb_flag[fidx] = big ? 1u : 0u;
    fbx[fidx] = FbIdx{big ? nn : 0u, big ? 1u : 0u};
}

// ---- results ------------------------------------------------------------------------------------
// Dense IR of one unit (device resident; host mirrors are filled lazily for the text dump).
struct IrResult {
    // device
    FuncIr* d_funcs = nullptr; opusfe_variable* d_vars = nullptr; opusfe_instr* d_instrs = nullptr; uint32_t* d_bb_offsets = nullptr;
    uint32_t* d_cargs = nullptr; TypeRec* d_types = nullptr; SemErr* d_errs = nullptr; FuncRec* d_funcrecs = nullptr; GlobalRec* d_globals = nullptr;
    uint32_t* d_tid_of_node = nullptr;
    uint64_t n_funcs = 0, n_globals = 0, n_vars = 0, n_instrs = 0, n_bbs = 0, n_types = 0, n_gtypes = 0, n_cargs = 0, n_errs = 0, n_sem_errors = 0;
    uint32_t n_rounds = 0;                   // rounds of the count pass of the flat lowering (longest dependency chain of a function)
    uint64_t n_fallback = 0, n_bodies = 0;   // functions that the sequential machine lowered, of n_bodies functions with bodies
    // host mirrors
    bool have_host = false;
    std::vector<FuncIr> funcs; std::vector<opusfe_variable> vars; std::vector<opusfe_instr> instrs; std::vector<uint32_t> bb_offsets, cargs, tid_of_node;
    std::vector<TypeRec> types; std::vector<SemErr> errs; std::vector<FuncRec> funcrecs; std::vector<GlobalRec> globals;
};

#ifndef OPUSFE_EMU
inline void free_ir(opusfe_ctx* ctx, opusfe_unit* u) {
    IrResult* r = (IrResult*)u->ir;
    if (!r) return;
    dev_free(ctx, r->d_funcs); dev_free(ctx, r->d_vars); dev_free(ctx, r->d_instrs); dev_free(ctx, r->d_bb_offsets); dev_free(ctx, r->d_cargs);
    dev_free(ctx, r->d_types); dev_free(ctx, r->d_errs); dev_free(ctx, r->d_funcrecs); dev_free(ctx, r->d_globals); dev_free(ctx, r->d_tid_of_node);
    delete r;
    u->ir = nullptr;
}

enum { WL_STATUS = 32, WL_ERRS = 33, WL_FBTOT = 34 /* FbIdx: 2 */, WL_BODIES = 52, WL_DEFTOT = 36 /* DefCnt: 4 */, WL_GT = 40, WL_FINERR = 41, WL_IRTOT = 44 /* IrCnt: 6 */, WL_SLOTTOT = 50 };

// group mode: first function / global of every file (NONE for a file without definitions)
__global__ void k_group_file_ranges(const uint32_t* __restrict__ first_def, const DefCnt* __restrict__ def_excl, uint32_t n_files, uint32_t* __restrict__ first_func,
                                    uint32_t* __restrict__ first_global) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_files) return;
    const uint32_t d = first_def[f];
    first_func[f] = d == OPUSFE_NONE ? OPUSFE_NONE : def_excl[d].funcs;
    first_global[f] = d == OPUSFE_NONE ? OPUSFE_NONE : def_excl[d].globals;
}

inline uint32_t next_pow2(uint64_t v) { uint32_t p = 16; while (p < v) p <<= 1; return p; }

// frontend.rs:140-187 (IR part): finalize every definition, then one IrGenerator per Definition::Function
inline void run_lower(opusfe_ctx* ctx, opusfe_unit* u) {
    hipStream_t st = ctx->stream;
    TempList tmp(ctx);
    free_ir(ctx, u);
    IrResult* R = new IrResult();
    u->ir = R;
    u->lowered = true;
    u->n_funcs = u->n_instrs = u->n_vars = u->n_bbs = u->n_types = 0;
    const uint32_t n_defs = (uint32_t)u->n_defs, n_nodes = (uint32_t)u->n_nodes;
    if (n_defs == 0) { u->status = OPUSFE_OK; R->have_host = true; return; }
    HIP_TRY(hipMemsetAsync(ctx->d_status + 32, 0, 32 * sizeof(uint32_t), st));
    uint32_t* scan_tmp = tmp.get<uint32_t>(8 * ((std::max<uint64_t>(n_nodes, n_defs) + SCAN_TILE - 1) / SCAN_TILE + 2));

    // A. classify definitions, number functions / globals / type definitions
    DefCnt* def_excl = tmp.get<DefCnt>(n_defs);
    LAUNCH(ctx, "sem_classify", k_sem_classify, grid_for(n_defs, 256), 256, u->d_nodes, u->d_def_nodes, n_defs, def_excl);
    prof_begin(ctx, "sem_scan_defs");
    device_scan<true>(def_excl, def_excl, n_defs, OpAddDefCnt(), DefCnt{0, 0, 0, 0}, (DefCnt*)scan_tmp, (DefCnt*)(ctx->d_status + WL_DEFTOT), st);
    prof_end(ctx);
    // C1. type slots
    uint32_t* slot_of_node = tmp.get<uint32_t>(n_nodes);
    LAUNCH(ctx, "sem_type_flags", k_sem_type_flags, grid_for(n_nodes, 256), 256, u->d_nodes, n_nodes, slot_of_node);
    prof_begin(ctx, "sem_scan_types");
    device_scan_chained<true>(slot_of_node, slot_of_node, n_nodes, OpAddU32(), (uint32_t)0, (uint64_t*)scan_tmp, ctx->d_status + WL_GT, ctx->d_status + WL_STATUS, st);      // (scan_tmp: 32 B per 2048 elements, the single-pass scan needs 8 per 4096)
    prof_end(ctx);
    HIP_TRY(hipMemcpyAsync(ctx->h_status, ctx->d_status, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    prof_collect(ctx);
    if (ctx->h_status[WL_STATUS] >= DS_INTERNAL_CAPACITY) { u->status = OPUSFE_ERR_INTERNAL; ctx->msg = "type-slot scan: a tile never saw its predecessors (scan.h)"; return; }
    const uint32_t n_funcs = ctx->h_status[WL_DEFTOT], n_globals = ctx->h_status[WL_DEFTOT + 1], n_named = ctx->h_status[WL_DEFTOT + 3];
    // global type ids: primitives, one per structured parse-tree type / `type` definition, then the pointer-to-primitive table
    const uint32_t n_gt0 = TP_COUNT + ctx->h_status[WL_GT];
    const uint32_t pp_base = n_gt0, n_gtypes = n_gt0 + par::PP_COUNT;

    // B. name table
    const uint32_t n_slots = next_pow2(2ull * n_named + 16);
    NameSlot* slots = tmp.get<NameSlot>(n_slots);
    NameEntry* entries = tmp.get<NameEntry>(n_named + 1);
    uint32_t* def_slot = tmp.get<uint32_t>(n_defs);
    uint32_t* slot_off = tmp.get<uint32_t>(n_slots);
    LAUNCH(ctx, "sem_slots_init", k_sem_slots_init, grid_for(n_slots, 256), 256, slots, n_slots);
    // group mode: which file every definition belongs to
    GroupInfo* const grp = u->group;
    uint32_t* def_file = nullptr; uint8_t* def_skip = nullptr;
    if (grp) {
        def_file = tmp.get<uint32_t>(n_defs); def_skip = tmp.get<uint8_t>(n_defs);
        HIP_TRY(hipMemsetAsync(grp->d_first_def, 0xFF, (size_t)grp->n_files * 4, st));
        LAUNCH(ctx, "group_first_defs", k_group_first_defs, grid_for(n_defs, 256), 256, u->d_nodes, u->d_def_nodes, n_defs, (const uint32_t*)grp->d_start, grp->n_files, grp->d_first_def, def_file);
        LAUNCH(ctx, "group_file_ranges", k_group_file_ranges, grid_for(grp->n_files, 256), 256, (const uint32_t*)grp->d_first_def, (const DefCnt*)def_excl, grp->n_files, grp->d_first_func, grp->d_first_global);
    }
    LAUNCH(ctx, "sem_table_insert", k_sem_table_insert, grid_for(n_defs, 256), 256, u->d_src, u->d_tokens, u->d_nodes, u->d_def_nodes, n_defs, slots, n_slots - 1, def_slot);
    LAUNCH(ctx, "sem_slot_counts", k_sem_slot_counts, grid_for(n_slots, 256), 256, slots, n_slots, slot_off);
    uint32_t* scan_tmp2 = tmp.get<uint32_t>((n_slots + SCAN_TILE - 1) / SCAN_TILE + 2);
    prof_begin(ctx, "sem_scan_slots");
    device_scan<true>(slot_off, slot_off, n_slots, OpAddU32(), (uint32_t)0, scan_tmp2, ctx->d_status + WL_SLOTTOT, st);
    prof_end(ctx);
    LAUNCH(ctx, "sem_table_fill", k_sem_table_fill, grid_for(n_defs, 256), 256, u->d_nodes, u->d_def_nodes, n_defs, def_excl, def_slot, slot_off, slots, entries);
    LAUNCH(ctx, "sem_table_sort", k_sem_table_sort, grid_for(n_slots, 256), 256, slots, n_slots, slot_off, entries);

    // C2. resolve every parse-tree type node; D. finalize.  The global type records go straight into the result array when the
    // unit lowers without sequential-machine functions; the machine's sparse type regions need one contiguous id space with them
    TypeRec* gtypes = tmp.get<TypeRec>(n_gtypes);
    R->d_tid_of_node = dev_alloc_t<uint32_t>(ctx, n_nodes);
    uint8_t* undefined_flag = tmp.get<uint8_t>(n_nodes);
    uint32_t* ph_def = tmp.get<uint32_t>(n_gt0);
    uint32_t* ph_prev = tmp.get<uint32_t>(n_gt0);     // `type` definition -> the previous one of the same name (duplicates)
    R->d_funcrecs = dev_alloc_t<FuncRec>(ctx, n_funcs);
    R->d_globals = dev_alloc_t<GlobalRec>(ctx, n_globals);
    LAUNCH(ctx, "sem_resolve_types", k_sem_resolve_types, grid_for(std::max<uint32_t>(n_nodes, TP_COUNT), 256), 256, u->d_src, u->d_tokens, u->d_nodes, n_nodes,
           slot_of_node, slots, n_slots - 1, entries, R->d_tid_of_node, gtypes, undefined_flag, ph_def, ph_prev);
    LAUNCH(ctx, "sem_pp_types", k_sem_pp_types, 1, 128, gtypes + pp_base);
    LAUNCH(ctx, "sem_finalize", k_sem_finalize, grid_for(n_defs, 256), 256, u->d_nodes, u->d_def_nodes, n_defs, def_excl, R->d_tid_of_node, undefined_flag,
           ph_def, ph_prev, n_nodes, gtypes, R->d_funcrecs, R->d_globals, ctx->d_status + WL_FINERR);
    // size classes of the definitions (largest first): which workgroup shape lowers a function, and the dispatch order of the machine
    uint32_t* order = tmp.get<uint32_t>(n_defs);
    uint32_t* order_hist = tmp.get<uint32_t>(ORDER_CLASSES);
    HIP_TRY(hipMemsetAsync(order_hist, 0, ORDER_CLASSES * sizeof(uint32_t), st));
    LAUNCH(ctx, "order_hist", k_order_hist, grid_for(n_defs, 256), 256, u->d_def_nodes, n_defs, n_nodes, order_hist);
    LAUNCH(ctx, "order_scan", k_order_scan, 1, 64, order_hist);
    LAUNCH(ctx, "order_scatter", k_order_scatter, grid_for(n_defs, 256), 256, u->d_def_nodes, n_defs, n_nodes, order_hist, order);
    // functions of 2048 nodes and more never fit the node-parallel path: flagged now, so that the machine can start on them at once
    uint32_t* big_flag = tmp.get<uint32_t>(n_funcs + 1);
    FbIdx* big_idx = tmp.get<FbIdx>(n_funcs + 1);
    HIP_TRY(hipMemsetAsync(ctx->d_status + WL_FBTOT, 0, 2 * sizeof(uint32_t), st));
    if (n_funcs) {
        LAUNCH(ctx, "fb_mark_size", k_fb_mark_size, grid_for(n_defs, 256), 256, u->d_def_nodes, n_defs, n_nodes, def_excl, u->d_nodes, (uint32_t)2048, big_flag, big_idx, (const uint8_t*)nullptr);
        prof_begin(ctx, "fb_scan");
        device_scan<true>(big_idx, big_idx, n_funcs, OpAddFbIdx(), FbIdx{0, 0}, (FbIdx*)scan_tmp, (FbIdx*)(ctx->d_status + WL_FBTOT), st);
        prof_end(ctx);
    }
    uint32_t class_end[ORDER_CLASSES];
    HIP_TRY(hipMemcpyAsync(class_end, order_hist, sizeof class_end, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(ctx->h_status, ctx->d_status, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    prof_collect(ctx);
    R->n_funcs = n_funcs; R->n_globals = n_globals; R->n_gtypes = n_gtypes;
    // The diagnostic switch exists only in a -DOPUSFE_DIAG build (tools/build_diag.sh): the shipped library has one behaviour.
#ifdef OPUSFE_DIAG
    static const bool seq_only = getenv("OPUSFE_LOWER_SEQ") != nullptr;     // everything through the sequential machine
#else
    constexpr bool seq_only = false;
#endif
    const uint32_t big_nodes = seq_only ? 0u : ctx->h_status[WL_FBTOT], big_funcs = seq_only ? 0u : ctx->h_status[WL_FBTOT + 1];
    if (grp) {
        // finalize-time errors concern their FILE (it takes the single-file path, which answers OPUSFE_SEMANTIC for it); the rest goes on.
        // Functions of files whose packed result is discarded anyway are not lowered.
        if (ctx->h_status[WL_FINERR] != 0) {
            uint32_t* fin_cnt = tmp.get<uint32_t>(n_defs);
            LAUNCH(ctx, "sem_fin_count", k_sem_fin_count, grid_for(n_defs, 256), 256, u->d_nodes, u->d_def_nodes, n_defs, R->d_tid_of_node, undefined_flag, ph_def, ph_prev, n_nodes, gtypes, fin_cnt);
            LAUNCH(ctx, "group_finerr", k_group_finerr, grid_for(n_defs, 256), 256, (const uint32_t*)fin_cnt, (const uint32_t*)def_file, n_defs, grp->d_flags);
        }
        LAUNCH(ctx, "group_def_mask", k_group_def_mask, grid_for(n_defs, 256), 256, (const uint32_t*)def_file, (const uint32_t*)grp->d_flags, n_defs, (uint32_t)(FF_SINGLE | FF_SYNTAX), def_skip);
    } else
    if (ctx->h_status[WL_FINERR] != 0) {   // frontend.rs:145: semantic errors while finalizing -> no IR
        R->n_sem_errors = ctx->h_status[WL_FINERR];
        // the diagnostics themselves, in the reference's report order (definitions in file order)
        uint32_t* fin_cnt = tmp.get<uint32_t>(n_defs);
        LAUNCH(ctx, "sem_fin_count", k_sem_fin_count, grid_for(n_defs, 256), 256, u->d_nodes, u->d_def_nodes, n_defs, R->d_tid_of_node, undefined_flag, ph_def, ph_prev, n_nodes, gtypes, fin_cnt);
        prof_begin(ctx, "sem_fin_scan");
        device_scan<true>(fin_cnt, fin_cnt, n_defs, OpAddU32(), (uint32_t)0, scan_tmp, (uint32_t*)nullptr, st);
        prof_end(ctx);
        R->n_errs = R->n_sem_errors;
        R->d_errs = dev_alloc_t<SemErr>(ctx, R->n_errs);
        LAUNCH(ctx, "sem_fin_emit", k_sem_fin_emit, grid_for(n_defs, 256), 256, u->d_nodes, u->d_def_nodes, n_defs, R->d_tid_of_node, undefined_flag, ph_def, ph_prev, n_nodes, gtypes, fin_cnt, R->d_errs);
        // the types the diagnostics name (InvalidType prints one, compile.rs:844-847) stay readable: OPUSFE_A_TYPES = the global records
        R->d_types = dev_alloc_t<TypeRec>(ctx, n_gtypes);
        HIP_TRY(hipMemcpyAsync(R->d_types, gtypes, (size_t)n_gtypes * sizeof(TypeRec), hipMemcpyDeviceToDevice, st));
        R->n_types = n_gtypes; R->n_gtypes = n_gtypes;
        HIP_TRY(hipStreamSynchronize(st));
        prof_collect(ctx);
        u->status = OPUSFE_SEMANTIC;
        ctx->msg = "finalize_definition reported semantic errors; IR generation skipped (frontend.rs:145)";
        return;
    }
    if (n_funcs == 0) { u->status = OPUSFE_OK; R->have_host = true; return; }

    // E. IR generation.  The node-parallel path (lower_par.h) lowers every function it can in ONE pass straight into the dense
    // result arrays (sizes by counting, places by bump allocation); the functions it declines go through the sequential
    // machine, into sparse regions sized for them alone, and are compacted into the same arrays.  The arrays are sized from
    // the node count (ratios of the previous unit on this context, with head-room); if one is too small the pass raises
    // `overflow`, still counts exactly, and everything runs once more with the exact sizes.
    SemView S{u->d_src, u->d_tokens, u->d_val64, u->d_nodes, R->d_tid_of_node, gtypes, slots, n_slots - 1, entries, R->d_funcrecs, R->d_globals, n_gtypes, grp ? (const uint32_t*)grp->d_start : (const uint32_t*)nullptr, grp ? grp->n_files : 0u};
    R->d_funcs = dev_alloc_t<FuncIr>(ctx, n_funcs);
    uint64_t* user_syms = tmp.get<uint64_t>(1);
    LAUNCH(ctx, "sem_user_syms", k_sem_user_syms, 1, 64, S, user_syms);
    if (grp) LAUNCH(ctx, "group_user_syms", k_group_user_syms, grid_for(n_defs, 256), 256, u->d_nodes, u->d_tokens, u->d_def_nodes, n_defs, (unsigned long long*)user_syms);
    // order[] holds the definitions by size class, largest class first; class c ends at class_end[c] and starts where class c + 1 ends
    unsigned long long* counters = tmp.get<unsigned long long>(par::CTR_COUNT + 2);     // + overflow flag, + error counter
    uint32_t* fb_flag = tmp.get<uint32_t>(n_funcs);
    FbIdx* fbx = tmp.get<FbIdx>(n_funcs);
    uint64_t cap[6];     // vars, instrs, bbs, types, cargs, diagnostics
    for (int k = 0; k < 5; k++) cap[k] = (uint64_t)(ctx->ir_ratio[k] * (double)n_nodes) + (1u << 16);
    cap[5] = (1u << 16) + n_nodes / 64;
    for (int attempt = 0; attempt < 2; attempt++) {
        dev_free(ctx, R->d_vars); dev_free(ctx, R->d_instrs); dev_free(ctx, R->d_bb_offsets); dev_free(ctx, R->d_cargs); dev_free(ctx, R->d_types); dev_free(ctx, R->d_errs);
        R->d_vars = nullptr; R->d_instrs = nullptr; R->d_bb_offsets = nullptr; R->d_cargs = nullptr; R->d_types = nullptr; R->d_errs = nullptr;
        if (cap[1] >= (1ull << par::CTR_BB_SHIFT) || (uint64_t)n_gtypes + cap[3] >= 0xFFFFFFF0ull || cap[0] >= 0xFFFFFFF0ull) { u->status = OPUSFE_ERR_ARG; ctx->msg = "unit too large for 32-bit IR indices: shard it"; return; }
        R->d_vars = dev_alloc_t<opusfe_variable>(ctx, cap[0]);
        R->d_instrs = dev_alloc_t<opusfe_instr>(ctx, cap[1]);
        R->d_bb_offsets = dev_alloc_t<uint32_t>(ctx, cap[2] + 1);
        R->d_types = dev_alloc_t<TypeRec>(ctx, (uint64_t)n_gtypes + cap[3]);
        R->d_cargs = dev_alloc_t<uint32_t>(ctx, cap[4]);
        R->d_errs = dev_alloc_t<SemErr>(ctx, cap[5]);
        HIP_TRY(hipMemsetAsync(counters, 0, (par::CTR_COUNT + 2) * sizeof(unsigned long long), st));
        HIP_TRY(hipMemsetAsync(ctx->d_status + WL_STATUS, 0, 2 * sizeof(uint32_t), st));
        HIP_TRY(hipMemsetAsync(ctx->d_status + WL_BODIES, 0, sizeof(uint32_t), st));
        HIP_TRY(hipMemcpyAsync(R->d_types, gtypes, (size_t)n_gtypes * sizeof(TypeRec), hipMemcpyDeviceToDevice, st));
        TempList tmp2(ctx);       // the machine's sparse regions and the per-node state of the flat passes live for one attempt
        // per-node state of the flat passes (lower_par.h FlatArgs): 32 + 16 + 2 B per node, the list of the schedule (4 B per listed node)
        par::FlatArgs X{};
        X.ctl = tmp2.get<uint32_t>(par::CTL_WORDS);
        X.fninfo = tmp2.get<par::FnInfo>(n_funcs); X.fb = tmp2.get<uint32_t>(n_funcs);
        X.rec = tmp2.get<par::NodeRec>(n_nodes); X.cnt = tmp2.get<uint64_t>(2ull * n_nodes); X.tail = tmp2.get<uint8_t>(2ull * n_nodes);
        X.cg = (uint16_t*)slot_of_node; X.key1 = X.cg + n_nodes;      // (the type-slot numbers are not read any more: their block serves these two)
        HIP_TRY(hipMemsetAsync(X.key1, 0xFF, (size_t)n_nodes * sizeof(uint16_t), st));      // NONE16: no stage 1
        HIP_TRY(hipMemsetAsync(X.ctl, 0, par::CTL_WORDS * sizeof(uint32_t), st));
        HIP_TRY(hipMemsetAsync(X.rec, 0xFF, (size_t)n_nodes * sizeof(par::NodeRec), st));      // result type none, function NONE
        HIP_TRY(hipMemsetAsync(X.cnt, 0, (size_t)n_nodes * 2 * sizeof(uint64_t), st));
        HIP_TRY(hipMemsetAsync(X.tail, 0xFF, (size_t)n_nodes * 2, st));                        // (until the schedule is sorted: every node's bin, NONE16 = not listed)
        LAUNCH(ctx, "par_init", par::k_par_init, grid_for(n_defs, 256), 256, u->d_nodes, u->d_def_nodes, n_defs, def_excl, R->d_funcs, ctx->d_status + WL_BODIES, (const uint8_t*)def_skip, X.fninfo, X.fb);
        par::ParArgs PA{};
        PA.def_nodes = u->d_def_nodes; PA.n_defs = n_defs; PA.n_nodes = n_nodes; PA.def_excl = def_excl; PA.undefined_flag = undefined_flag; PA.user_syms = user_syms;
        PA.pp_base = pp_base; PA.out = R->d_funcs; PA.status = ctx->d_status + WL_STATUS; PA.skip = def_skip;
        PA.counters = counters; PA.overflow = (uint32_t*)(counters + par::CTR_COUNT); PA.err_counter = (uint32_t*)(counters + par::CTR_COUNT + 1);
        PA.cap_vars = cap[0]; PA.cap_instrs = cap[1]; PA.cap_bbs = cap[2]; PA.cap_types = cap[3]; PA.cap_cargs = cap[4]; PA.cap_errs = cap[5];
        PA.vars = R->d_vars; PA.instrs = R->d_instrs; PA.bb_offsets = R->d_bb_offsets; PA.cargs = R->d_cargs; PA.types_out = R->d_types;
        PA.dbg = X.ctl + par::CTL_DBG;
#ifdef OPUSFE_DIAG
        { static const char* sp = getenv("OPUSFE_PRE_STOP"); PA.stop_phase = sp ? (uint32_t)atoi(sp) : 0u; }
#endif
        S.types = gtypes;
        // The sequential machine on the functions flagged in `flag` (sparse regions laid out by `idx`), then their places in the
        // dense arrays and the compaction into them; all on stream `ms`.
        // Declared behind the temporaries, so destroyed before them: if anything below throws while the second stream still runs the
        // machine on tmp2's blocks (and writes R's arrays), both streams are drained before a block returns to the allocator.
        struct ForkGuard {
            opusfe_ctx* c; bool armed;
            ~ForkGuard() { if (armed) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamSynchronize(c->stream); } }
        } fork_guard{ctx, false};
        auto machine_pass = [&](hipStream_t ms, const uint32_t* flag, const FbIdx* idx, uint32_t nodes_total, uint32_t funcs_total, bool timed) -> bool {
            LowerBuffers B{};
            B.type_sparse_base = n_gtypes;
            const uint64_t type_cap = (uint64_t)n_gtypes + lower_base(CAP_TYPES, nodes_total, funcs_total) + CAP_CONST;
            if (type_cap >= 0xFFFFFFF0ull) { u->status = OPUSFE_ERR_ARG; ctx->msg = "unit too large for 32-bit type ids: shard it"; return false; }
            TypeRec* types = tmp2.get<TypeRec>(type_cap);     // global records followed by the machine's sparse per-function regions
            HIP_TRY(hipMemcpyAsync(types, gtypes, (size_t)n_gtypes * sizeof(TypeRec), hipMemcpyDeviceToDevice, ms));
            SemView S2 = S; S2.types = types;
            B.vars = tmp2.get<opusfe_variable>(lower_base(CAP_VARS, nodes_total, funcs_total) + CAP_CONST);
            const uint64_t icap = lower_base(CAP_INSTRS, nodes_total, funcs_total) + CAP_CONST;
            B.instrs = tmp2.get<opusfe_instr>(icap); B.ibb = tmp2.get<uint32_t>(icap); B.irank = tmp2.get<uint32_t>(icap);
            const uint64_t bcap = lower_base(CAP_BBS, nodes_total, funcs_total) + CAP_CONST;
            B.bbcnt = tmp2.get<uint32_t>(bcap); B.bblast = tmp2.get<uint32_t>(bcap);
            B.cargs = tmp2.get<uint32_t>(lower_base(CAP_CARGS, nodes_total, funcs_total) + CAP_CONST);
            const uint64_t lcap = lower_base(CAP_LOCALS, nodes_total, funcs_total) + CAP_CONST;
            B.lname = tmp2.get<uint32_t>(lcap); B.lvar = tmp2.get<uint32_t>(lcap); B.lstack = tmp2.get<uint32_t>(lcap);
            B.frames = tmp2.get<LFrame>(lower_base(CAP_FRAMES, nodes_total, funcs_total) + CAP_CONST);
            const uint64_t ecap = lower_base(CAP_ERRS, nodes_total, funcs_total) + CAP_CONST;
            B.errs = tmp2.get<SemErr>(ecap);
            uint64_t* prof_buf = tmp2.get<uint64_t>(LP_COUNT);
            HIP_TRY(hipMemsetAsync(prof_buf, 0, LP_COUNT * 8, ms));
            par::ParArgs PB = PA;
            if (timed) prof_begin(ctx, "lower_functions");
            hipLaunchKernelGGL(k_lower_functions, dim3(grid_for((uint64_t)n_defs * UNIT_LANES, 64)), dim3(64), 0, ms, S2, B, (const uint8_t*)undefined_flag, (const uint32_t*)u->d_def_nodes, n_defs, n_nodes,
                               (uint32_t)u->n_tokens, (const DefCnt*)def_excl, R->d_funcs, ctx->d_status + WL_STATUS, prof_buf, (const uint64_t*)user_syms, (const uint32_t*)order, flag, idx);
            HIP_TRY(hipGetLastError());
            if (timed) prof_end(ctx);
            hipLaunchKernelGGL(k_fb_bases, dim3(grid_for(n_funcs, 64)), dim3(64), 0, ms, R->d_funcs, n_funcs, B, PB, flag, idx);
            HIP_TRY(hipGetLastError());
            IrOut O{R->d_vars, R->d_instrs, R->d_cargs, R->d_types, R->d_errs, R->d_bb_offsets, n_gtypes};
            unsigned grid = (unsigned)((n_funcs + 3) / 4);
            if (grid > 8192) grid = 8192;
            hipLaunchKernelGGL(k_ir_compact, dim3(grid), dim3(256), 0, ms, (const FuncIr*)R->d_funcs, n_funcs, B, (const TypeRec*)types, O, flag, idx);
            HIP_TRY(hipGetLastError());
            return true;
        };
        // the functions that are too large for the node-parallel path start at once, next to it
        HIP_TRY(hipEventRecord(ctx->ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
        fork_guard.armed = true;
        if (big_funcs) { if (!machine_pass(ctx->stream2, big_flag, big_idx, big_nodes, big_funcs, false)) { HIP_TRY(hipStreamSynchronize(ctx->stream2)); return; } }
        if (!seq_only) {
            // k_flat_pre: one launch per half-octave size class (scan.h order_class_of_size): the LDS a workgroup reserves is sized for the
            // largest function of its class, so at most a third of it is unused.  Functions of 2048 nodes and more, and the ones the
            // flat path declines, go through the sequential machine.
            static const char* const NAMES[12] = {"lower_pre_64", "lower_pre_96", "lower_pre_128", "lower_pre_192", "lower_pre_256", "lower_pre_384", "lower_pre_512",
                                                  "lower_pre_768", "lower_pre_1024", "lower_pre_1536", "lower_pre_2048", "lower_pre"};
            const uint32_t c_top = order_class_of_size(2047), c_small = order_class_of_size(63);     // classes above c_top: sequential machine
            int name_i = 10;
            // (a small unit — BASELINE configuration 2 — is one launch of the largest shape: ten launches of a few functions each are ten
            // times the latency of one function)
            const bool one_launch = n_funcs <= 1024;
            for (uint32_t c = c_top; c >= c_small; c--, name_i--) {
                // the smallest launch takes every class up to c_small
                const uint32_t lo = class_end[c + 1], hi = (c == c_small || one_launch) ? n_defs : class_end[c];
                if (one_launch && c != c_top) break;
                if (hi <= lo) continue;
                par::ParArgs a = PA;
                a.list = order + lo; a.n_list = hi - lo;
                a.nmax = (uint32_t)order_class_lo(c + 1) - 1u; if (a.nmax < 63) a.nmax = 63;
                a.maxdef = a.nmax / 8 < 32 ? 32 : a.nmax / 8;
                // any number of waves per workgroup: the one that keeps the most functions resident among those that fill the
                // CU's wave slots to within one wave (4 SIMDs x OPUSFE_PAR_WAVES = 32; LDS is allocated in 1280-byte granules)
                uint32_t best_waves = 0;
                uint32_t resident[17] = {0};
                for (uint32_t w = 1; w <= 16; w++) {
                    const size_t need = (par::lds_bytes(a.nmax, a.maxdef, 64 * w) + 256 + 1279) / 1280 * 1280;
                    const uint32_t blocks = (uint32_t)std::min<size_t>((160u << 10) / need, (4 * OPUSFE_PAR_WAVES) / w);
                    resident[w] = blocks * w;
                    best_waves = std::max(best_waves, resident[w]);
                }
                uint32_t w = 1;
                while (w < 16 && resident[w] + 1u < best_waves) w++;
                const uint32_t threads = 64 * w;
                const size_t lds = par::lds_bytes(a.nmax, a.maxdef, threads);
                prof_begin(ctx, NAMES[name_i < 0 ? 11 : name_i]);
                HIP_TRY(hipFuncSetAttribute((const void*)par::k_flat_pre, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(par::k_flat_pre, dim3(a.n_list), dim3(threads), lds, st, S, a, X);
                HIP_TRY(hipGetLastError());
                prof_end(ctx);
            }
            // The schedule: every node's (class, round) bin was written by k_flat_pre; a counting sort makes the list that the count pass
            // (one launch per class and round, in round order; a round with few nodes is one launch of the kernel that holds every
            // class) and the emit pass (one launch per class) go through.  The host reads the table of offsets once and knows every launch.
            const uint32_t per_block = 1u << 16;
            const unsigned sgrid = grid_for(n_nodes, per_block);
            const size_t slds = (size_t)(par::SCHED_BINS + par::KC_ACTIVE) * sizeof(uint32_t);
            HIP_TRY(hipFuncSetAttribute((const void*)par::k_sched_pass<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds));
            HIP_TRY(hipFuncSetAttribute((const void*)par::k_sched_pass<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds));
            prof_begin(ctx, "sched_hist");
            hipLaunchKernelGGL(par::k_sched_pass<false>, dim3(sgrid), dim3(1024), slds, st, X, n_nodes, per_block, 0u);
            HIP_TRY(hipGetLastError());
            prof_end(ctx);
            LAUNCH(ctx, "sched_scan", par::k_sched_scan, 1, 1024, X.ctl);
            ctx->sched_off.resize(par::SCHED_BINS + 1);
            HIP_TRY(hipMemcpyAsync(ctx->sched_off.data(), X.ctl + par::CTL_BINS, (par::SCHED_BINS + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            prof_collect(ctx);
            const uint32_t* off = ctx->sched_off.data();
            const uint32_t n_sched = off[par::SCHED_BINS];
            X.sched = tmp2.get<uint32_t>((size_t)n_sched + 64); X.order = tmp2.get<uint32_t>((size_t)n_sched + 64);
            prof_begin(ctx, "sched_scatter");
            hipLaunchKernelGGL(par::k_sched_pass<true>, dim3(sgrid), dim3(1024), slds, st, X, n_nodes, per_block, n_sched);
            HIP_TRY(hipGetLastError());
            prof_end(ctx);
            HIP_TRY(hipMemsetAsync(X.tail, 0, (size_t)n_nodes * 2, st));      // (the bins travelled in the tail bytes)
#ifdef OPUSFE_DIAG
            auto count_class = [&](auto cls, uint32_t r) {
                constexpr int c = decltype(cls)::value;
                const uint32_t begin = off[c * par::SCHED_R + r], n = off[c * par::SCHED_R + r + 1] - begin;
                if (n) hipLaunchKernelGGL(par::k_flat_count<c>, dim3(std::min<unsigned>(grid_for(n, 256), 4096u)), dim3(256), 0, st, S, PA, X, begin, n);
            };
#endif
            auto round_nodes = [&](uint32_t r) { uint64_t n = 0; for (uint32_t c = 0; c < par::KC_ACTIVE; c++) n += off[c * par::SCHED_R + r + 1] - off[c * par::SCHED_R + r]; return n; };
            // (every function's rounds start at 0: the first empty round ends them all); the tail = the rounds from which on none has more than 2048 nodes
            uint32_t rounds_run = 0, tail_from = 0;
            while (rounds_run < par::FLAT_MAX_ROUNDS && round_nodes(rounds_run)) { if (round_nodes(rounds_run) > 2048u) tail_from = rounds_run + 1; rounds_run++; }
            prof_begin(ctx, "lower_count");
            for (uint32_t r = 0; r < rounds_run; r++) {
                const uint64_t n_round = round_nodes(r);
                if (r >= tail_from && rounds_run - r >= 2) { hipLaunchKernelGGL(par::k_flat_count_tail, dim3(1), dim3(512), 0, st, S, PA, X, r, rounds_run); break; }
                // every round is ONE launch of the visitor for all kinds over the round's bins taken as one list (count_round_nodes).  One
                // launch per kind class and round — specialised kernels, fewer registers — was the plan of the first flat version and
                // is slower since the bins are one list: 33.7 ms with the rounds above 4 M nodes per class, 32.3 with none.
#ifdef OPUSFE_DIAG
                static const uint64_t round_nodes_max = getenv("OPUSFE_FLAT_ROUND_NODES") ? (uint64_t)atoll(getenv("OPUSFE_FLAT_ROUND_NODES")) : ~0ull;
                if (n_round > round_nodes_max) {
                    count_class(std::integral_constant<int, 0>(), r); count_class(std::integral_constant<int, 1>(), r); count_class(std::integral_constant<int, 2>(), r);
                    count_class(std::integral_constant<int, 3>(), r); count_class(std::integral_constant<int, 4>(), r); count_class(std::integral_constant<int, 5>(), r);
                    count_class(std::integral_constant<int, 6>(), r); count_class(std::integral_constant<int, 7>(), r);
                    continue;
                }
#endif
                hipLaunchKernelGGL(par::k_flat_count_round, dim3(std::min<unsigned>(grid_for(n_round, 256), 4096u)), dim3(256), 0, st, S, PA, X, r);
            }
            HIP_TRY(hipGetLastError());
            prof_end(ctx);
            R->n_rounds = rounds_run;
            IrCnt* fn_tot = tmp2.get<IrCnt>(n_funcs + 2);      // per function: its totals, then (exclusive scan) its place; [n_funcs] = the sum, [n_funcs + 1] = where the flat path's ranges start
            HIP_TRY(hipMemsetAsync(fn_tot, 0, ((size_t)n_funcs + 2) * sizeof(IrCnt), st));
            LAUNCH(ctx, "lower_scan", par::k_flat_scan, grid_for((uint64_t)n_defs * 64, 256), 256, S, PA, X, fn_tot);
            prof_begin(ctx, "lower_scan_funcs");
            device_scan<true>(fn_tot, fn_tot, n_funcs, OpAddIrCnt(), IrCnt{0, 0, 0, 0, 0, 0}, (IrCnt*)scan_tmp, fn_tot + n_funcs, st);
            prof_end(ctx);
            LAUNCH(ctx, "lower_reserve", par::k_flat_reserve, 1, 64, PA, (const IrCnt*)(fn_tot + n_funcs), fn_tot + n_funcs + 1);
            LAUNCH(ctx, "lower_bases", par::k_flat_bases, grid_for(n_funcs, 256), 256, PA, X, (const IrCnt*)fn_tot, (const IrCnt*)(fn_tot + n_funcs + 1), n_funcs);
            LAUNCH(ctx, "lower_rv", par::k_flat_rv, std::min<unsigned>(grid_for(n_nodes, 256), 4096u), 256, PA, X);
            auto launch_classes = [&](auto kernel_of_class) {
                kernel_of_class(std::integral_constant<int, 0>()); kernel_of_class(std::integral_constant<int, 1>()); kernel_of_class(std::integral_constant<int, 2>());
                kernel_of_class(std::integral_constant<int, 3>()); kernel_of_class(std::integral_constant<int, 4>()); kernel_of_class(std::integral_constant<int, 5>());
                kernel_of_class(std::integral_constant<int, 6>()); kernel_of_class(std::integral_constant<int, 7>());
            };
            const unsigned rgrid = std::min<unsigned>(grid_for(n_nodes, 256), 2048u);
            auto emit_class = [&](auto cls) {
                hipLaunchKernelGGL(par::k_flat_emit<decltype(cls)::value>, dim3(rgrid), dim3(256), 0, st, S, PA, X);
            };
            prof_begin(ctx, "lower_emit");
            launch_classes(emit_class);
            HIP_TRY(hipGetLastError());
            prof_end(ctx);
            LAUNCH(ctx, "flat_check", par::k_flat_check, 1, 64, (const uint32_t*)X.ctl, ctx->d_status + WL_STATUS);
        }
        HIP_TRY(hipEventRecord(ctx->ev_join, ctx->stream2));
        HIP_TRY(hipStreamWaitEvent(st, ctx->ev_join, 0));
        fork_guard.armed = false;      // from here on everything on `st` is ordered behind the second stream
        // the functions it declined
        LAUNCH(ctx, "fb_mark", k_fb_mark, grid_for(n_defs, 256), 256, R->d_funcs, n_funcs, u->d_def_nodes, n_defs, n_nodes, def_excl, u->d_nodes, fb_flag, fbx);
        prof_begin(ctx, "fb_scan");
        device_scan<true>(fbx, fbx, n_funcs, OpAddFbIdx(), FbIdx{0, 0}, (FbIdx*)scan_tmp, (FbIdx*)(ctx->d_status + WL_FBTOT), st);
        prof_end(ctx);
        HIP_TRY(hipMemcpyAsync(ctx->h_status, ctx->d_status, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        prof_collect(ctx);
        const uint32_t fb_nodes = ctx->h_status[WL_FBTOT], fb_funcs = ctx->h_status[WL_FBTOT + 1];
        R->n_fallback = fb_funcs + big_funcs; R->n_bodies = ctx->h_status[WL_BODIES];
        if (fb_funcs) { if (!machine_pass(st, fb_flag, fbx, fb_nodes, fb_funcs, true)) return; }
        unsigned long long hc[par::CTR_COUNT + 2];
        HIP_TRY(hipMemcpyAsync(hc, counters, sizeof hc, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(ctx->h_status, ctx->d_status, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        prof_collect(ctx);
        const uint32_t lst = ctx->h_status[WL_STATUS];
        if (lst >= DS_INTERNAL_CAPACITY) {
            uint32_t line = 0;
            (void)hipMemcpy(&line, X.ctl + par::CTL_DBG, 4, hipMemcpyDeviceToHost);
            u->status = OPUSFE_ERR_INTERNAL; ctx->msg = "lowering: internal capacity / consistency check failed (lower_par.h:" + std::to_string(line) + ")"; return;
        }
        if (lst != DS_OK) { u->status = OPUSFE_UNSUPPORTED; u->dev_status = lst; ctx->msg = "construct on which the reference panics (unreachable!/unwrap) or overflows its stack"; return; }
        const uint64_t tot[5] = {hc[par::CTR_VARS], hc[par::CTR_INSTR_BB] & ((1ull << par::CTR_BB_SHIFT) - 1), hc[par::CTR_INSTR_BB] >> par::CTR_BB_SHIFT, hc[par::CTR_TYPES], hc[par::CTR_CARGS]};
        const bool overflow = (uint32_t)hc[par::CTR_COUNT] != 0;
        if (overflow) {
            if (attempt == 1) { u->status = OPUSFE_ERR_INTERNAL; ctx->msg = "lowering: the exact-size pass overflowed"; return; }
            for (int k = 0; k < 5; k++) cap[k] = tot[k] + 16;      // the counters are exact whether or not everything fitted
            cap[5] = (uint32_t)hc[par::CTR_COUNT + 1] + 16;
            continue;
        }
        R->n_sem_errors = ctx->h_status[WL_ERRS];
        R->n_vars = tot[0]; R->n_instrs = tot[1]; R->n_bbs = tot[2]; R->n_types = (uint64_t)n_gtypes + tot[3]; R->n_cargs = tot[4]; R->n_errs = (uint32_t)hc[par::CTR_COUNT + 1];
        for (int k = 0; k < 5; k++) ctx->ir_ratio[k] = std::max(0.01, 1.04 * (double)tot[k] / (double)n_nodes);      // sizes of the next unit: this one's, plus head-room
        if (R->n_errs) {       // report order = definition order
            uint32_t* ecnt = tmp.get<uint32_t>(n_funcs);
            SemErr* ordered = dev_alloc_t<SemErr>(ctx, R->n_errs);
            LAUNCH(ctx, "err_counts", k_err_counts, grid_for(n_funcs, 256), 256, R->d_funcs, n_funcs, ecnt);
            device_scan<true>(ecnt, ecnt, n_funcs, OpAddU32(), (uint32_t)0, scan_tmp, (uint32_t*)nullptr, st);
            LAUNCH(ctx, "err_order", k_err_order, grid_for(n_funcs, 256), 256, R->d_funcs, n_funcs, ecnt, R->d_errs, ordered);
            HIP_TRY(hipStreamSynchronize(st));
            dev_free(ctx, R->d_errs);
            R->d_errs = ordered;
        }
        const uint32_t n_instrs32 = (uint32_t)R->n_instrs;
        HIP_TRY(hipMemcpyAsync(R->d_bb_offsets + R->n_bbs, &n_instrs32, 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        break;
    }
    u->n_funcs = n_funcs; u->n_instrs = R->n_instrs; u->n_vars = R->n_vars; u->n_bbs = R->n_bbs; u->n_types = R->n_types;
    u->status = OPUSFE_OK;
}

inline void ensure_host_ir(opusfe_unit* u) {
    IrResult* R = (IrResult*)u->ir;
    if (!R || R->have_host) return;
    hipStream_t st = u->ctx->stream;
    auto pull = [&](auto& vec, const auto* dptr, uint64_t n) {
        vec.resize(n);
        if (n && dptr) HIP_TRY(hipMemcpyAsync(vec.data(), dptr, n * sizeof(vec[0]), hipMemcpyDeviceToHost, st));
    };
    pull(R->funcs, R->d_funcs, R->n_funcs); pull(R->vars, R->d_vars, R->n_vars); pull(R->instrs, R->d_instrs, R->n_instrs);
    pull(R->bb_offsets, R->d_bb_offsets, R->d_bb_offsets ? R->n_bbs + 1 : 0); pull(R->cargs, R->d_cargs, R->n_cargs); pull(R->types, R->d_types, R->d_types ? R->n_types : 0);
    pull(R->errs, R->d_errs, R->n_errs); pull(R->funcrecs, R->d_funcrecs, R->n_funcs); pull(R->globals, R->d_globals, R->n_globals);
    pull(R->tid_of_node, R->d_tid_of_node, u->n_nodes);
    HIP_TRY(hipStreamSynchronize(st));
    R->have_host = true;
}
#endif  // OPUSFE_EMU

}  // namespace opusfe
