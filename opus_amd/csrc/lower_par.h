// Node-parallel IR generation (generate_ir.rs:100-1199 as count -> scan -> emit instead of a sequential walk), FLAT over the
// whole unit: one lane per AST node of ANY function, nodes of one kind class in neighbouring lanes, per-node state in HBM.
//
// Round 2 ran all of this inside one workgroup per function with the state in LDS; a function has a handful of ready nodes
// of one kind per dependency level, so 11 of 64 lanes worked and the kernel was bound by instructions issued (1 % of the HBM
// roofline).  Here only the part that is local to a function by nature stays in a workgroup:
//
//   k_flat_pre    (one workgroup per function, LDS) structure: parent / subtree size / depth / pre-order rank of every node
//                 (nodes arrive in the parser's creation order, not pre-order: none of this is index arithmetic); evaluation
//                 contexts (generate_ir_from_expression with which expected type :510, try_generate_pointer_from_expression with
//                 which pointer kind :985); local definitions and name resolution by (definition point <= use < end of the
//                 defining block) — no scope stack is walked (generate_ir.rs:176-185, 857-893); the kind class of every node and
//                 what it WAITS for: its operands, the `var` / `for` definition it reads, the left side of its assignment.
//                 Everything a later pass reads is written to the unit-wide per-node arrays (FlatArgs).
//   k_flat_count  (one lane per READY node, one launch per round of the schedule that k_flat_pre computed) lowering a node =
//                 running its slice of the reference's emission schema against an emitter that only COUNTS
//                 variables, types, instructions and basic blocks per EVENT (the points of the Euler tour of the tree: ENTER(node),
//                 EXIT(node)).  The schedule is static: a counting sort of (class, round) bins, no atomic and no queue between
//                 rounds; a round's bins are gone through as one list.  Rounds = the longest dependency chain of any function.
//   k_flat_scan   (one wave per function) exclusive scan of the counts in Euler order: variable ids, block ids, instruction
//                 slots.  Two facts make this the final layout: (1) the current block is always the newest block, so emission
//                 order IS block-major order; (2) every deferred terminator (`bbs[branch].push(..)`, :310-312, 357-359) goes to a
//                 block at the moment that block stops being current.  A device-wide scan over the functions then gives every
//                 function its place in the dense result arrays, in definition order.
//   k_flat_rv     (one lane per node) every node learns the id of the variable that holds its result.
//   k_flat_emit   (one lane per node, one launch per kind class) the same visitor again, writing variables / instructions /
//                 types at the scanned positions, straight into the dense result arrays.
//
// The fast path declines (the function is lowered by the sequential machine of lower.h instead, bit-identical by
// construction) whenever it meets something it does not model: any semantic error, shadowing, structural type
// comparisons of records / function types with different ids, E_Proc / CallOrIndex, functions larger than the
// LDS tables of k_flat_pre.  How many functions decline is reported (FuncIr.status == DS_FALLBACK, IrResult::n_fallback).
#pragma once
// (included from lower.h, behind the sequential machine: uses FuncIr, DefCnt, the tables of sem.h)


namespace opusfe {
namespace par {

#ifdef OPUSFE_EMU
#define PAR_SYNC() do { } while (0)
inline bool par_sync_or(bool v) { return v; }
inline void par_fence() {}
inline uint32_t* g_emu_lds = nullptr;          // tests/emu: heap block of exactly the kernel's dynamic LDS size (ASan checks it)
#define PAR_LDS_BASE() (::opusfe::par::g_emu_lds)
inline uint64_t g_emu_stat[8] = {0};          // tests/emu: rounds of the three fixpoints, weighted by nodes (EMU_VERBOSE prints them)
#else
#define PAR_SYNC() __syncthreads()
__device__ __forceinline__ bool par_sync_or(bool v) { return __syncthreads_or(v ? 1 : 0) != 0; }
__device__ __forceinline__ void par_fence() { __threadfence_block(); }
extern __shared__ __align__(16) uint32_t par_dyn_lds[];
#define PAR_LDS_BASE() (::opusfe::par::par_dyn_lds)
#endif

enum : uint32_t { DS_FALLBACK = 0x100 };      // FuncIr.status of a function the fast path declined (never a unit status)
enum { PP_COUNT = 4 * TP_COUNT };             // pointer-to-primitive type ids: pp_base + PtrKind * TP_COUNT + primitive

// ---- type descriptors ---------------------------------------------------------------------------------------------
// A type as the fast path sees it: an id that exists before lowering (primitive, parse-tree type, pointer-to-primitive)
// or `function type of function f`, wrapped in up to 12 pointer levels created during lowering.  Only the structure is
// observable (TypePrinter, compile.rs:200-274); the emit pass materialises the wrappers as fresh type records.
typedef uint64_t Desc;
constexpr Desc DESC_NONE = ~0ull;
enum { DESC_MAXW = 12 };
__host__ __device__ __forceinline__ uint32_t d_base(Desc d) { return (uint32_t)d; }
__host__ __device__ __forceinline__ uint32_t d_nw(Desc d) { return (uint32_t)(d >> 32) & 15u; }
__host__ __device__ __forceinline__ bool d_fn(Desc d) { return ((d >> 60) & 1u) != 0; }
__host__ __device__ __forceinline__ uint32_t d_wrap(Desc d, uint32_t k) { return (uint32_t)(d >> (36 + 2 * k)) & 3u; }
__host__ __device__ __forceinline__ Desc d_id(uint32_t id) { return (Desc)id; }
__host__ __device__ __forceinline__ Desc d_func(uint32_t f) { return (Desc)f | (1ull << 60); }
__host__ __device__ __forceinline__ bool d_plain(Desc d) { return (d >> 32) == 0; }
__host__ __device__ __forceinline__ Desc d_inner(Desc d) {      // nw > 0: the pointee
    const uint32_t nw = d_nw(d);
    d &= ~(3ull << (36 + 2 * (nw - 1)));
    d = (d & ~(15ull << 32)) | ((uint64_t)(nw - 1) << 32);
    return d;
}

struct ParArgs {
    const uint32_t* def_nodes; uint32_t n_defs, n_nodes; const DefCnt* def_excl; const uint8_t* undefined_flag; const uint64_t* user_syms;
    const uint32_t* list; uint32_t n_list; uint32_t nmax, maxdef; uint32_t pp_base;
    const uint8_t* skip;      // group mode: definitions of files whose packed result is discarded (syntax error / single-file path): not lowered
    FuncIr* out; uint32_t* status;
    // The dense result arrays, filled in ONE pass: a function that knows its sizes takes its place in each array with an
    // atomic add on `counters` (bump allocation: the layout is dense, its order is the order in which functions finish).
    // counters[0] = instructions | blocks << 36 (one atomic, so that block order and instruction order agree: the end of a
    // function's last block is the start of the next function's first), [1] variables, [2] local types, [3] call arguments.
    // When an array is too small nothing is written, `overflow` is raised and the host runs again with the exact totals.
    unsigned long long* counters; uint32_t* overflow; uint32_t* err_counter;   // (semantic errors: sequential machine only)
    uint64_t cap_vars, cap_instrs, cap_bbs, cap_types, cap_cargs, cap_errs;
    opusfe_variable* vars; opusfe_instr* instrs; uint32_t* bb_offsets; uint32_t* cargs; TypeRec* types_out;
    uint32_t* dbg;            // source line of the consistency check that failed (diagnostics of DS_INTERNAL_CAPACITY)
    uint32_t stop_phase;      // -DOPUSFE_DIAG builds (tools/pre_phases.sh): k_flat_pre leaves after phase n and declines the function; 0 = run everything
};

// Kind class of a node: neighbouring lanes lower the same kind of node, and every class has a kernel of its own that contains
// only its cases of the visitor (registers, code size).  KC_NONE: nothing to lower (type nodes, carriers, dead code).
enum { KC_VARIABLE = 0, KC_INTEGER = 1, KC_OPERATOR = 2, KC_SIMPLE = 3, KC_STATEMENT = 4, KC_CONTROL = 5, KC_CALL = 6, KC_OTHER = 7, KC_ACTIVE = 8, KC_NONE = 15, KC_ANY = 99 };

// The unit-wide per-node state of the flat passes (temporaries of one lowering; index = node index in the unit).
struct FnInfo { uint32_t dn, nn, nargs, ret, sym_mix, pad0, pad1, pad2; };     // what a lane needs of its function (nn == 0: not on the flat path)
// ctl[]: control words of one lowering.  The schedule of the count pass is a table of SCHED_BINS list lengths / offsets, one per
// (kind class, round): bin = class * SCHED_R + round.  A class's bins follow each other, so the nodes of a class — what the emit
// pass goes through — are one range of the list, and (class, round) — one launch of the count pass — is a range inside it.
// (The last round of a class is no round: it holds the CHAINED nodes, which the count pass reaches through their operand.)
enum : uint32_t { SCHED_R = 2048, FLAT_MAX_ROUNDS = SCHED_R - 1, SCHED_BINS = KC_ACTIVE * SCHED_R,
                  CTL_OCUR = 16 /* KC_ACTIVE cursors: the end of every class's part of `order` */, CTL_DBG = 30, CTL_LISTOVF = 31 /* a list was too small (never) */,
                  CTL_BINS = 64 /* SCHED_BINS + 1: lengths, then exclusive offsets, of the schedule's bins */,
                  CTL_CURSOR = CTL_BINS + SCHED_BINS + 64 /* SCHED_BINS: where the next block's nodes of a bin go */, CTL_WORDS = CTL_CURSOR + SCHED_BINS + 64 };
enum : uint32_t { FB_NOFIT = 0xFFFFFFFEu };   // fb[]: the function counted, but its place in a result array is beyond the array (the host runs again)
// What the visits read and write of a node, in ONE 32-byte record (one memory sector per node and per operand instead of one per field):
// rt = type of the node's result (def slots: of the variable), st = state word, ref = name resolution / literal suffix / type id,
// rvi = who created the result variable (then its id), pe = ENTER position in the Euler tour, sz = subtree size, pa = parent,
// fn = function index (NONE: not a node of a function on the flat path).  The host fills the array with 0xFF: rt = none, fn = NONE.
struct NodeRec { uint64_t rt; uint32_t st, ref, rvi; uint16_t pe, sz, pa, loc /* index in its function: node - loc = the function's first node */; uint32_t fn; };
static_assert(sizeof(NodeRec) == 32, "NodeRec is one 32-byte sector");
struct FlatArgs {
    NodeRec* rec; uint64_t* cnt; uint8_t* tail; uint16_t* cg;
    uint16_t* key1;           // the schedule bin of stage 1 of a `for` node (NONE16: none).  The bin of the node itself travels in its two
                              // `tail` bytes (NONE16: not listed), which are cleared again before the count pass writes them.
    FnInfo* fninfo; uint32_t* fb;      // per function: fb != 0 = declined (source line of the reason)
    uint32_t* sched;          // every node that is lowered, by (class, round), and stage 1 of every `for` (entry | SCHED_STAGE1): the count pass
    uint32_t* order;          // every node that is lowered, by class and within a class in node order (blocks of 64 K nodes): the emit pass, which
                              // then meets a function's nodes together.  Class c starts where it starts in `sched` and ends at ctl[CTL_OCUR + c].
    uint32_t* ctl;
};

enum { CTR_INSTR_BB = 0, CTR_VARS = 1, CTR_TYPES = 2, CTR_CARGS = 3, CTR_COUNT = 4, CTR_BB_SHIFT = 36 };
struct Bases { uint32_t var_base, instr_base, bb_base, type_base, carg_base; bool fits; };
// a function's place in the dense arrays (one lane)
__device__ __forceinline__ Bases bump_allocate(const ParArgs& A, uint32_t n_vars, uint32_t n_instrs, uint32_t n_bbs, uint32_t n_types, uint32_t n_cargs) {
    Bases b;
    const unsigned long long ib = atomicAdd(A.counters + CTR_INSTR_BB, (unsigned long long)n_instrs | ((unsigned long long)n_bbs << CTR_BB_SHIFT));
    const uint64_t i0 = ib & ((1ull << CTR_BB_SHIFT) - 1), b0 = ib >> CTR_BB_SHIFT;
    const uint64_t v0 = atomicAdd(A.counters + CTR_VARS, (unsigned long long)n_vars);
    const uint64_t t0 = n_types ? atomicAdd(A.counters + CTR_TYPES, (unsigned long long)n_types) : 0ull;
    const uint64_t c0 = n_cargs ? atomicAdd(A.counters + CTR_CARGS, (unsigned long long)n_cargs) : 0ull;
    b.fits = i0 + n_instrs <= A.cap_instrs && b0 + n_bbs <= A.cap_bbs && v0 + n_vars <= A.cap_vars && t0 + n_types <= A.cap_types && c0 + n_cargs <= A.cap_cargs;
    b.instr_base = (uint32_t)i0; b.bb_base = (uint32_t)b0; b.var_base = (uint32_t)v0; b.type_base = (uint32_t)t0; b.carg_base = (uint32_t)c0;
    if (!b.fits) atomicMax(A.overflow, 1u);
    return b;
}
struct DefEnt { uint32_t hash, low, len, node, dpoint, send, var; uint32_t pad; Desc vt; };   // a local definition, visible to nodes of pre-order rank [dpoint, send); vt = type of its variable
struct ArgEnt { uint32_t hash, low, len, pad; Desc vt; };
enum { MAX_ARGS = 16, MAX_CALL_ARGS = 8 };

// per-node state bits (st[])
enum : uint32_t { ST_DONE = 1u, ST_STAGE1 = 2u /* for: the loop variable has its type */, ST_DIV = 4u, ST_DEAD = 8u, ST_CTX = 16u /* rank, depth and context assigned */, ST_HASBREAK = 32u, ST_CLAIM = 64u /* a lane is lowering it */,
                  ST_CHAINED = 1u << 15 /* lowered by the lane that lowers its one operand, right behind it */,
                  ST_MODE_E = 1u << 8, ST_MODE_P = 2u << 8, ST_MODE_MASK = 3u << 8, ST_PT_SHIFT = 10, ST_DEREF = 1u << 12, ST_MUSTLV = 1u << 13,
                  ST_EXP_SHIFT = 16, ST_EXP_MASK = 15u << 16,
                  ST_PEND_SHIFT = 20, ST_PEND_ONE = 1u << 20, ST_PEND_MASK = 0xFFFu << 20 /* operands that are not lowered yet (type fixpoint) */ };
enum : uint32_t { EXP_NONE = 0, EXP_GENINT, EXP_BOOL, EXP_RET, EXP_SIZE, EXP_ASSIGN, EXP_FIELD, EXP_ARG };
// name resolution results (ref[]): tag in the top 3 bits
enum : uint32_t { RF_NONE = 0u, RF_LOCAL = 1u << 29, RF_ARG = 2u << 29, RF_GLOBAL = 3u << 29, RF_CONST = 4u << 29, RF_FUNC = 5u << 29, RF_CALL = 6u << 29, RF_CALL1 = 7u << 29 /* exactly one function of that name: its index */, RF_TAG = 7u << 29, RF_VAL = ~(7u << 29) };
// result-variable info (rvi[]): who created the variable that holds a node's result
enum : uint32_t { RV_OWN = 0u, RV_PASS = 1u << 29, RV_LOCAL = 2u << 29, RV_ARG = 3u << 29, RV_RESOLVED = 7u << 29, RV_TAG = 7u << 29, RV_VAL = ~(7u << 29) };
enum : uint8_t { TAIL_NONE = 0, TAIL_INSTR = 1, TAIL_TERM = 2, TAIL_NEWBB = 3 };

struct Cnt4 { uint32_t v, t, i, b; };
__device__ __forceinline__ uint64_t pack4(const Cnt4& c) { return (uint64_t)c.v | ((uint64_t)c.t << 16) | ((uint64_t)c.i << 32) | ((uint64_t)c.b << 48); }
__device__ __forceinline__ Cnt4 unpack4(uint64_t w) { return Cnt4{(uint32_t)(w & 0xFFFF), (uint32_t)((w >> 16) & 0xFFFF), (uint32_t)((w >> 32) & 0xFFFF), (uint32_t)(w >> 48)}; }
#ifdef OPUSFE_EMU
#define LDS_AS
#else
#define LDS_AS __attribute__((address_space(3)))
#endif
typedef LDS_AS uint64_t lds_u64; typedef LDS_AS uint32_t lds_u32; typedef LDS_AS uint16_t lds_u16; typedef LDS_AS uint8_t lds_u8;
#ifdef OPUSFE_EMU
inline uint32_t lds_add(lds_u32* p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
inline uint32_t lds_sub(lds_u32* p, uint32_t v) { const uint32_t o = *p; *p = o - v; return o; }
inline uint32_t lds_or(lds_u32* p, uint32_t v) { const uint32_t o = *p; *p = o | v; return o; }
inline uint32_t lds_and(lds_u32* p, uint32_t v) { const uint32_t o = *p; *p = o & v; return o; }
inline uint32_t lds_max(lds_u32* p, uint32_t v) { const uint32_t o = *p; if (v > o) *p = v; return o; }
#else
__device__ __forceinline__ uint32_t lds_add(lds_u32* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t lds_sub(lds_u32* p, uint32_t v) { return __hip_atomic_fetch_add(p, 0u - v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t lds_or(lds_u32* p, uint32_t v) { return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t lds_and(lds_u32* p, uint32_t v) { return __hip_atomic_fetch_and(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t lds_max(lds_u32* p, uint32_t v) { return __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif

// ---- the function a lane works in (flat passes) ----------------------------------------------------------------------
// Per-node arrays are unit-wide arrays in HBM seen through the function's first node (`base`): f.st[i] is the state word of the
// function's i-th node, exactly as the visitor addressed its LDS arrays in round 2.  Links (first child, next sibling), kinds
// and payloads are read from the parser's node records directly.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(OPUSFE_EMU)
#define G_AS OPUSFE_GLOBAL
#else
#define G_AS
#endif
template <class T> struct GArr {
    G_AS T* p; uint32_t base;
    __device__ __forceinline__ G_AS T& operator[](uint32_t i) const { return p[base + i]; }
};
// a field of the nodes' records (NodeRec)
template <class T, int OFF> struct RecArr {
    G_AS uint8_t* p; uint32_t base;
    __device__ __forceinline__ G_AS T& operator[](uint32_t i) const { return *(G_AS T*)(p + (((size_t)base + i) << 5) + OFF); }
};
#define NONE16 0xFFFFu
struct NodeKn {      // kind | sub << 8 | nchild << 16 (the first word of the node record)
    const G_AS opusfe_node* p; uint32_t base;
    __device__ __forceinline__ uint32_t operator[](uint32_t i) const { const G_AS opusfe_node* q = p + base + i; return (uint32_t)q->kind | ((uint32_t)q->sub << 8) | ((uint32_t)q->nchild << 16); }
};
struct NodeFc { const G_AS opusfe_node* p; uint32_t base; __device__ __forceinline__ uint32_t operator[](uint32_t i) const { const uint32_t v = p[base + i].first_child; return v == OPUSFE_NONE ? NONE16 : v - base; } };
struct NodeNx { const G_AS opusfe_node* p; uint32_t base; __device__ __forceinline__ uint32_t operator[](uint32_t i) const { const uint32_t v = p[base + i].next; return v == OPUSFE_NONE ? NONE16 : v - base; } };
struct NodePay { const G_AS opusfe_node* p; uint32_t base; __device__ __forceinline__ uint32_t operator[](uint32_t i) const { return p[base + i].payload; } };
struct Fn {
    SemView S; ParArgs A;
    uint32_t dn, nn, fidx, nargs, ret; uint64_t user_syms; uint32_t sym_mix /* file mix of symbol keys (sem.h NameKey) */;
    bool fb, emitting; G_AS uint32_t* fbp;     // fb: this lane declined the function (fbp[fidx] tells the other lanes)
    RecArr<uint64_t, 0> rt; RecArr<uint32_t, 8> st; RecArr<uint32_t, 12> ref; RecArr<uint32_t, 16> rvi; RecArr<uint16_t, 20> pe; RecArr<uint16_t, 22> sz; RecArr<uint16_t, 24> pa;
    GArr<uint64_t> cnt; GArr<uint16_t> cg; GArr<uint8_t> tail;
    NodeKn kn; NodeFc fc; NodeNx nx; NodePay pay;
    // emit bases
    uint32_t var_base, instr_base, bb_base, type_base, carg_base, n_vars_tot, n_instrs_tot, n_bbs_tot, n_types_tot, n_cargs_tot;
};
#ifdef OPUSFE_EMU
inline uint32_t at_or(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o | v; return o; }
inline uint32_t at_sub(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o - v; return o; }
inline uint32_t at_add(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
#else
__device__ __forceinline__ uint32_t at_or(G_AS uint32_t* p, uint32_t v) { return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t at_sub(G_AS uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, 0u - v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t at_add(G_AS uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
// the count and emit visits must agree, and nothing may be written outside a function's place: an internal error of the unit
__device__ __forceinline__ void bad_at(const Fn& f, uint32_t line) { atomicMax(f.A.status, (uint32_t)DS_INTERNAL_CAPACITY); if (f.A.dbg) atomicMax(f.A.dbg, line); }
#define inconsistent(f) bad_at(f, __LINE__)
// declines the function; a reason (source line of the check) is kept for the fallback statistics
__device__ __forceinline__ void fallback_at(Fn& f, uint32_t line) { f.fb = true; if (f.emitting) inconsistent(f); else f.fbp[f.fidx] = line; }
#define fallback(f) fallback_at(f, __LINE__)
__device__ __forceinline__ bool failed(const Fn& f) { return f.fb; }

template <class F> __device__ __forceinline__ uint8_t KIND(const F& f, uint32_t i) { return (uint8_t)f.kn[i]; }
template <class F> __device__ __forceinline__ uint8_t SUB(const F& f, uint32_t i) { return (uint8_t)(f.kn[i] >> 8); }
template <class F> __device__ __forceinline__ uint32_t NCH(const F& f, uint32_t i) { return f.kn[i] >> 16; }
template <class F> __device__ __forceinline__ uint32_t POS_ENTER(const F& f, uint32_t i) { return f.pe[i]; }
template <class F> __device__ __forceinline__ uint32_t POS_EXIT(const F& f, uint32_t i) { return (uint32_t)f.pe[i] + 2u * (uint32_t)f.sz[i] - 1u; }
// pre-order rank of a node, and of the first node behind its subtree (nodes are stored in creation order, not pre-order)
template <class F> __device__ __forceinline__ uint32_t RANK(const F& f, uint32_t i) { return f.rk[i]; }
template <class F> __device__ __forceinline__ uint32_t RANK_END(const F& f, uint32_t i) { return (uint32_t)f.rk[i] + f.sz[i]; }
template <class F> __device__ __forceinline__ uint32_t child_at(const F& f, uint32_t i, uint32_t k) { uint32_t c = f.fc[i]; while (k-- && c != NONE16) c = f.nx[c]; return c; }
template <class F> __device__ __forceinline__ uint32_t last_child(const F& f, uint32_t i) { uint32_t c = f.fc[i], l = NONE16; while (c != NONE16) { l = c; c = f.nx[c]; } return l; }
// global type id of a parse-tree type node of this function
template <class F> __device__ __forceinline__ uint32_t TID(const F& f, uint32_t local_node) { return f.ref[local_node]; }      // (type nodes keep their resolved id in ref[], loaded with the node)

// ---- views of a type ------------------------------------------------------------------------------------------------
struct TView { uint8_t kind, ptr; bool plain; Desc sub; TypeRec rec; };
__device__ __forceinline__ TView view(const Fn& f, Desc d) {
    TView v{}; v.plain = false; v.sub = DESC_NONE;
    if (d == DESC_NONE) { v.kind = TY_Error; return v; }      // no type (an input that is not ready / a declined function)
    const uint32_t nw = d_nw(d);
    if (nw) { v.kind = TY_Pointer; v.ptr = (uint8_t)d_wrap(d, nw - 1); v.sub = d_inner(d); return v; }
    if (d_fn(d)) { v.kind = TY_Function; v.rec.kind = TY_Function; v.rec.src = TS_FUNC; v.rec.ref = d_base(d); v.sub = d_id(f.S.funcs[d_base(d)].ret); return v; }
    const uint32_t id = d_base(d);
    v.plain = true;
    if (id < TP_COUNT) { v.kind = prim_kind(id); return v; }
    if (id >= f.A.pp_base && id < f.A.pp_base + PP_COUNT) { v.kind = TY_Pointer; v.ptr = (uint8_t)((id - f.A.pp_base) / TP_COUNT); v.sub = d_id((id - f.A.pp_base) % TP_COUNT); return v; }
    v.rec = gld(f.S.types + id);
    v.kind = v.rec.kind; v.ptr = v.rec.ptr;
    if (v.kind == TY_Pointer || v.kind == TY_Array || v.kind == TY_Function) v.sub = d_id(v.rec.sub);
    return v;
}
// pointer to `sub` (type_pointer, compile.rs:563-589); pointers to primitives are pre-made ids
template <class F> __device__ __forceinline__ Desc d_ptr(F& f, uint32_t pt, Desc sub) {
    if (sub == DESC_NONE) return DESC_NONE;
    if (d_plain(sub) && d_base(sub) < TP_COUNT) return d_id(f.A.pp_base + pt * TP_COUNT + d_base(sub));
    const uint32_t nw = d_nw(sub);
    if (nw >= DESC_MAXW) { fallback(f); return DESC_NONE; }
    return (sub & ~(15ull << 32)) | ((uint64_t)(nw + 1) << 32) | ((uint64_t)pt << (36 + 2 * nw));
}
__device__ __forceinline__ uint32_t type_records(Desc d) { return d_nw(d) + (d_fn(d) ? 1u : 0u); }   // records the emit pass writes for a variable of this type

enum { M_NO = 0, M_YES = 1, M_UNSURE = 2 };
__device__ __forceinline__ uint64_t array_len(const Fn& f, const TypeRec& r) {
    const opusfe_token t = f.S.tok[r.ref];
    return (t.flags & OPUSFE_TF_INLINE) ? (uint64_t)t.payload : f.S.val64[t.payload];
}
// compile.rs:698-751 on descriptors; records and function types are only compared by id (M_UNSURE otherwise)
__device__ __forceinline__ int match3(const Fn& f, Desc a, Desc b) {
    for (uint32_t guard = 0; guard < 64; guard++) {
        if (a == DESC_NONE || b == DESC_NONE) return M_UNSURE;
        if (a == b) return M_YES;
        const TView va = view(f, a), vb = view(f, b);
        if (va.kind != vb.kind) return M_NO;
        switch (va.kind) {
            case TY_Pointer: if (va.ptr != vb.ptr) return M_NO; a = va.sub; b = vb.sub; break;
            case TY_Array: if (array_len(f, va.rec) != array_len(f, vb.rec)) return M_NO; a = va.sub; b = vb.sub; break;
            case TY_Record: case TY_Function: return M_UNSURE;
            default: return M_YES;
        }
    }
    return M_UNSURE;
}
__device__ __forceinline__ int match_allow_generic3(const Fn& f, Desc a, Desc b) {      // compile.rs:753-763
    const uint8_t ka = view(f, a).kind, kb = view(f, b).kind;
    if ((ka == TY_GenericInteger && kind_is_integral(kb)) || (kb == TY_GenericInteger && kind_is_integral(ka))) return M_YES;
    if (ka == TY_Generic || kb == TY_Generic) return M_YES;
    return match3(f, a, b);
}
__device__ __forceinline__ int match_default_generic3(const Fn& f, Desc a, Desc b) {    // compile.rs:765-775
    const uint8_t ka = view(f, a).kind, kb = view(f, b).kind;
    if ((ka == TY_GenericInteger && kb == TY_Integer32) || (ka == TY_Integer32 && kb == TY_GenericInteger)) return M_YES;
    if (ka == TY_Generic || kb == TY_Generic) return M_NO;
    return match3(f, a, b);
}
enum { CA_NO = -1, CA_UNSURE = -2 };
// compile.rs:624-670; CastKind, CA_NO or CA_UNSURE
__device__ __forceinline__ int can_autocast3(const Fn& f, Desc from, Desc to) {
    int m = match3(f, from, to);
    if (m == M_UNSURE) return CA_UNSURE;
    if (m == M_YES) return CT_None;
    const TView t1 = view(f, from), t2 = view(f, to);
    if (t1.kind == TY_Error || t2.kind == TY_Error) return CT_Error;
    if (t1.kind == TY_GenericInteger && kind_is_integral(t2.kind)) return CT_Error;
    if (t1.kind == TY_Generic) return CT_Error;
    if (t1.kind == TY_Pointer && t2.kind == TY_Pointer) {
        const uint8_t p1 = t1.ptr, p2 = t2.ptr;
        if (p1 == p2 || (p1 == PT_Mutable && p2 == PT_Reference) || (p1 == PT_ArrayMutable && p2 == PT_Array)) {
            m = match3(f, t1.sub, t2.sub);
            if (m == M_UNSURE) return CA_UNSURE;
            if (m == M_YES) return CT_PointerType;
        }
        if ((p1 == PT_Mutable && p2 == PT_ArrayMutable) || (p1 == PT_Mutable && p2 == PT_Array) || (p1 == PT_Reference && p2 == PT_Array)) {
            m = match3(f, t2.sub, d_id(TP_Natural8));
            if (m == M_UNSURE) return CA_UNSURE;
            if (m == M_YES) return CT_PointerType;
        }
        if ((p1 == PT_ArrayMutable && p2 == PT_Mutable) || (p1 == PT_ArrayMutable && p2 == PT_Reference) || (p1 == PT_Array && p2 == PT_Reference)) {
            m = match3(f, t1.sub, d_id(TP_Natural8));
            if (m == M_UNSURE) return CA_UNSURE;
            if (m == M_YES) return CT_PointerType;
        }
        const TView s1 = view(f, t1.sub);
        if (s1.kind == TY_Array) {
            m = match3(f, s1.sub, t2.sub);
            if (m == M_UNSURE) return CA_UNSURE;
            if (m == M_YES) {
                if ((p1 == PT_Reference && p2 == PT_Array) || (p1 == PT_Mutable && p2 == PT_ArrayMutable) || (p1 == PT_Mutable && p2 == PT_Array)) return CT_PointerToArray;
                return CA_NO;
            }
        }
        return CA_NO;
    }
    return CA_NO;
}
// compile.rs:682-696
__device__ __forceinline__ int can_cast3(const Fn& f, Desc from, Desc to) {
    const int c = can_autocast3(f, from, to);
    if (c != CA_NO) return c;
    const TView t1 = view(f, from), t2 = view(f, to);
    if (kind_is_integral(t1.kind) && kind_is_integral(t2.kind)) return CT_Integer;
    if (t1.kind == TY_Pointer && t2.kind == TY_Pointer && t1.ptr == t2.ptr) return CT_Pointer;
    if ((t1.kind == TY_Pointer || t1.kind == TY_Function) && kind_is_integral(t2.kind)) return CT_PointerToInteger;
    if ((t2.kind == TY_Pointer || t2.kind == TY_Function) && kind_is_integral(t1.kind)) return CT_IntegerToPointer;
    return CA_NO;
}

// ---- the emitter ------------------------------------------------------------------------------------------------------
// COUNT: v / t / i / b count what an event creates.  EMIT: they are the next function-relative variable id, local type
// index, instruction index and block id (from the scan) and everything is written to the dense arrays.
template <bool EMIT> struct Em {
    uint32_t v, t, i, b; uint8_t tail; uint32_t pos;
    __device__ __forceinline__ uint32_t emit_type(Fn& f, const TypeRec& r) {
        const uint32_t id = f.S.n_gtypes + f.type_base + t;
        if (EMIT) { if (t < f.n_types_tot) gst(f.A.types_out + id, r); else inconsistent(f); }
        t++;
        return id;
    }
    __device__ __forceinline__ uint32_t type_id(Fn& f, Desc d) {
        if (d_plain(d)) return d_base(d);
        uint32_t cur = d_base(d);
        if (d_fn(d)) { TypeRec r{}; r.kind = TY_Function; r.src = TS_FUNC; r.ref = d_base(d); r.sub = f.S.funcs[d_base(d)].ret; r.alias = OPUSFE_NONE; cur = emit_type(f, r); }
        const uint32_t nw = d_nw(d);
        for (uint32_t k = 0; k < nw; k++) { TypeRec r{}; r.kind = TY_Pointer; r.ptr = (uint8_t)d_wrap(d, k); r.sub = cur; r.alias = OPUSFE_NONE; cur = emit_type(f, r); }
        return cur;
    }
    __device__ __forceinline__ uint32_t newvar(Fn& f, Desc ty, bool temp) {
        if (ty == DESC_NONE) { fallback(f); ty = d_id(TP_Error); }
        const uint32_t tid = type_id(f, ty);
        if (EMIT) { if (v < f.n_vars_tot) gst(f.A.vars + f.var_base + v, opusfe_variable{tid, temp ? 1u : 0u}); else inconsistent(f); }
        return v++;
    }
    __device__ __forceinline__ void ins(Fn& f, uint16_t op, uint32_t dst = 0, uint32_t a = 0, uint32_t b_ = 0, uint32_t c = 0, uint32_t extra = 0, uint16_t sub = 0) {
        if (EMIT) {
            opusfe_instr in; in.opcode = op; in.sub = sub; in.dst = dst; in.a = a; in.b = b_; in.c = c; in.extra = extra;
            if (i < f.n_instrs_tot) gst(f.A.instrs + f.instr_base + i, in); else inconsistent(f);
        }
        i++;
        tail = (op == I_Return || op == I_Jump || op == I_Branch || op == I_Unreachable) ? TAIL_TERM : TAIL_INSTR;
    }
    __device__ __forceinline__ uint32_t newbb(Fn& f) {
        if (EMIT) { if (b < f.n_bbs_tot) gst(f.A.bb_offsets + f.bb_base + b, f.instr_base + i); else inconsistent(f); }
        tail = TAIL_NEWBB;
        return b++;
    }
};
template <bool EMIT> __device__ __forceinline__ Em<EMIT> open_slot(Fn& f, uint32_t pos) {
    Em<EMIT> e; e.pos = pos; e.tail = TAIL_NONE;
    if (EMIT) { const Cnt4 c = unpack4(f.cnt[pos]); e.v = c.v; e.t = c.t; e.i = c.i; e.b = c.b; }
    else { e.v = e.t = e.i = e.b = 0; }
    return e;
}
template <bool EMIT> __device__ __forceinline__ void close_slot(Fn& f, const Em<EMIT>& e) {
    if (!EMIT) {
        if (e.v > 0xFFF || e.t > 0xFFF || e.i > 0xFFF || e.b > 0xFFF) fallback(f);
        f.cnt[e.pos] = pack4(Cnt4{e.v, e.t, e.i, e.b});
        f.tail[e.pos] = e.tail;
    }
}
// first block created by the event at `pos` (emit pass: the scanned prefix)
__device__ __forceinline__ uint32_t bb_at(const Fn& f, uint32_t pos) { return unpack4(f.cnt[pos]).b; }

// ---- names --------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool bytes_eq(const SemView& S, uint32_t lo1, uint32_t lo2, uint32_t len) {
    if (lo1 == lo2) return true;
    for (uint32_t k = 0; k < len; k++) if (S.src[lo1 + k] != S.src[lo2 + k]) return false;
    return true;
}
__device__ __forceinline__ bool slot_has_functions(const SemView& S, uint32_t slot) {
    if (slot == OPUSFE_NONE) return false;
    const NameSlot sl = S.slots[slot];
    for (uint32_t j = 0; j < sl.count; j++) if ((S.entries[sl.offset + j].kind_index >> 30) == NK_FUNCTION) return true;
    return false;
}

// ---- function values ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool fn_builtin(uint32_t fn) { return (fn & FN_BUILTIN) != 0; }
__device__ __forceinline__ uint32_t fn_nargs(const Fn& f, uint32_t fn) { return fn_builtin(fn) ? builtin_nargs((fn >> 8) & 0xFF) : f.S.funcs[fn].nargs; }
__device__ __forceinline__ uint32_t fn_ret(const Fn& f, uint32_t fn) { return fn_builtin(fn) ? builtin_ret_type((fn >> 8) & 0xFF, fn & 0xFF) : f.S.funcs[fn].ret; }
__device__ __forceinline__ uint32_t fn_arg_type(const Fn& f, uint32_t fn, uint32_t k) {
    if (fn_builtin(fn)) return builtin_arg_type((fn >> 8) & 0xFF, fn & 0xFF);
    uint32_t c = f.S.funcs[fn].first_arg;
    while (k--) c = f.S.nodes[c].next;
    return f.S.tid_of_node[f.S.nodes[c].first_child];
}
// Compiler::lookup_function (compile.rs:1083-1139): builtins of the symbol, then the user overloads in definition order.
// FN_NONE: an error the reference reports (the function falls back); the caller checks failed().
__device__ __forceinline__ uint32_t lookup_function(Fn& f, NameKey name, const Desc* argt, uint32_t n) {
    const uint8_t sym = name_sym(f.S.tok, name);
    if (sym != T_Identifier && builtin_count(sym) > 0 && !((f.user_syms >> sym) & 1) && n == builtin_nargs(sym)) {
        const Desc a0 = argt[0], a1 = n > 1 ? argt[1] : argt[0];
        if (d_plain(a0) && d_plain(a1) && d_base(a0) < TP_COUNT && d_base(a1) < TP_COUNT && d_base(a0) != TP_Error && d_base(a1) != TP_Error) {
            const int idx = builtin_pick(sym, d_base(a0), d_base(a1));
            if (idx >= 0) return FN_BUILTIN | ((uint32_t)sym << 8) | (uint32_t)idx;
        }
    }
    if (n > MAX_CALL_ARGS) { fallback(f); return FN_NONE; }
    const uint32_t slot = name_find(f.S, name);
    const uint32_t nb = sym != T_Identifier ? (uint32_t)builtin_count(sym) : 0u;
    const bool has_user = slot_has_functions(f.S, slot);
    if (nb == 0 && !has_user) { fallback(f); return FN_NONE; }                      // UndefinedFunction
    const NameSlot sl = slot != OPUSFE_NONE ? f.S.slots[slot] : NameSlot{OPUSFE_NONE, 0, 0, 0};
    const uint32_t total = nb + sl.count;
    auto cand = [&](uint32_t c) -> uint32_t {
        if (c < nb) return FN_BUILTIN | ((uint32_t)sym << 8) | c;
        const NameEntry e = f.S.entries[sl.offset + (c - nb)];
        return (e.kind_index >> 30) == NK_FUNCTION ? (e.kind_index & 0x3FFFFFFFu) : FN_NONE;
    };
    auto viable = [&](uint32_t fn) -> bool {
        if (fn_nargs(f, fn) != n) return false;
        for (uint32_t k = 0; k < n; k++) {
            const int c = can_autocast3(f, argt[k], d_id(fn_arg_type(f, fn, k)));
            if (c == CA_UNSURE) { fallback(f); return false; }
            if (c < 0) return false;
        }
        return true;
    };
    uint32_t first = FN_NONE; uint32_t ignorable = ~0u;
    for (uint32_t c = 0; c < total && !failed(f); c++) {
        const uint32_t fn = cand(c);
        if (fn == FN_NONE || !viable(fn)) continue;
        if (first == FN_NONE) first = fn;
        for (uint32_t k = 0; k < n; k++) {
            if (!((ignorable >> k) & 1)) continue;
            const int m = match_allow_generic3(f, d_id(fn_arg_type(f, fn, k)), d_id(fn_arg_type(f, first, k)));
            if (m == M_UNSURE) { fallback(f); return FN_NONE; }
            if (m == M_NO) ignorable &= ~(1u << k);
        }
    }
    if (failed(f)) return FN_NONE;
    if (first == FN_NONE) { fallback(f); return FN_NONE; }                           // NoOverloadForFunction
    for (uint32_t c = 0; c < total; c++) {
        const uint32_t fn = cand(c);
        if (fn == FN_NONE || !viable(fn)) continue;
        if (failed(f)) return FN_NONE;
        bool all = true;
        for (uint32_t k = 0; k < n; k++) {
            if ((ignorable >> k) & 1) continue;
            const int m = match_default_generic3(f, d_id(fn_arg_type(f, fn, k)), argt[k]);
            if (m == M_UNSURE) { fallback(f); return FN_NONE; }
            if (m == M_NO) { all = false; break; }
        }
        if (all) return fn;
    }
    fallback(f);                                                                      // AmbiguousOverload
    return FN_NONE;
}

// ---- the visitor --------------------------------------------------------------------------------------------------------
// Result variable of node c as an operand (emit pass: the id computed by the rv pass; count pass: unused)
template <bool EMIT> __device__ __forceinline__ uint32_t RV(const Fn& f, uint32_t c) { return EMIT ? (f.rvi[c] & RV_VAL) : 0u; }
__device__ __forceinline__ uint32_t own_rv(uint32_t pos, uint32_t off) { return RV_OWN | (pos << 12) | (off & 0xFFF); }

// the node that created the variable holding the result of node c (follows pass-through results)
__device__ __forceinline__ uint32_t creator_of(const Fn& f, uint32_t c) {
    for (uint32_t guard = 0; guard < 4096; guard++) {
        const uint32_t r = f.rvi[c];
        if ((r & RV_TAG) != RV_PASS) return c;
        c = r & 0xFFFF;
    }
    return c;
}

template <bool EMIT> struct Val { uint32_t var; Desc t; uint32_t info; };   // info: rvi encoding of `var` (count pass)

// autocast_variable_to_type (generate_ir.rs:207-231) of the result of node `src` (type t, variable in val) to `want`
template <bool EMIT> __device__ __forceinline__ Val<EMIT> autocast(Fn& f, Em<EMIT>& e, Val<EMIT> val, uint32_t src_node, Desc want) {
    const int m = match3(f, val.t, want);
    if (m == M_UNSURE) { fallback(f); return val; }
    if (m == M_YES) return val;
    const int ct = can_autocast3(f, val.t, want);
    if (ct < 0) { fallback(f); return val; }                                           // UnexpectedType (or unsure)
    const uint8_t fk = view(f, val.t).kind;
    if (fk == TY_Error || view(f, want).kind == TY_Error) { fallback(f); return val; }
    if (fk == TY_GenericInteger || fk == TY_Generic) {
        // retype in place (:214-218): the creator of the variable writes it with its final type
        if (src_node == NONE16) { fallback(f); return val; }
        if (!EMIT) {
            const uint32_t x = creator_of(f, src_node);
            if (KIND(f, x) != E_Integer || !d_plain(want)) { fallback(f); return val; }
            if (d_base(want) >= TP_COUNT) { fallback(f); return val; }
            f.ref[x] = (f.ref[x] & 0xFFFFFFu) | ((d_base(want) + 1u) << 24);
        }
        val.t = want;
        return val;
    }
    Val<EMIT> r; r.t = want;
    r.info = own_rv(e.pos, e.v);
    r.var = e.newvar(f, want, true);
    e.ins(f, I_Cast, r.var, val.var, 0, 0, 0, (uint16_t)ct);
    return r;
}
// try_cast_pointer (:808-855)
template <bool EMIT> __device__ __forceinline__ Val<EMIT> try_cast_pointer(Fn& f, Em<EMIT>& e, Val<EMIT> val, uint32_t pt) {
    const TView r = view(f, val.t);
    if (r.kind != TY_Pointer) { fallback(f); return val; }
    if (r.ptr == pt) return val;
    const uint32_t p2 = r.ptr;
    if ((p2 == PT_Mutable && pt == PT_Reference) || (p2 == PT_ArrayMutable && pt == PT_Array) || (p2 == PT_Array && pt == PT_Reference)
        || (p2 == PT_ArrayMutable && pt == PT_Mutable) || (p2 == PT_ArrayMutable && pt == PT_Reference)) {
        Val<EMIT> o; o.t = d_ptr(f, pt, r.sub);
        o.info = own_rv(e.pos, e.v);
        o.var = e.newvar(f, o.t, true);
        e.ins(f, I_Cast, o.var, val.var, 0, 0, 0, CT_PointerType);
        return o;
    }
    fallback(f);                                                                       // ImmutableLvalue / ArrayPointerToNonArray
    return val;
}
// pointee of a pointer-typed variable (get_lvalue_type :235-242)
__device__ __forceinline__ Desc pointee(Fn& f, Desc t) {
    const TView r = view(f, t);
    if (r.kind != TY_Pointer) { fallback(f); return DESC_NONE; }
    return r.sub;
}
// p_result of lower.h: generate_pointer_from_expression's lvalue check (:970-979) and the rvalue load (:577-585)
template <bool EMIT> __device__ __forceinline__ Val<EMIT> p_result(Fn& f, Em<EMIT>& e, Val<EMIT> val, bool lv, uint32_t st) {
    if ((st & ST_MUSTLV) && !lv) { fallback(f); return val; }                          // InvalidLvalue
    if ((st & ST_DEREF) && lv) {
        Val<EMIT> o; o.t = pointee(f, val.t);
        o.info = own_rv(e.pos, e.v);
        o.var = e.newvar(f, o.t, true);
        e.ins(f, I_Load, o.var, val.var);
        return o;
    }
    return val;
}
template <bool EMIT> __device__ __forceinline__ void set_result(Fn& f, uint32_t i, const Val<EMIT>& v) {
    if (!EMIT) { f.rt[i] = v.t; f.rvi[i] = v.info; }
}
template <bool EMIT> __device__ __forceinline__ Val<EMIT> child_val(const Fn& f, uint32_t c) {
    Val<EMIT> v; v.var = RV<EMIT>(f, c); v.t = f.rt[c]; v.info = RV_PASS | c;
    return v;
}

// field of a record type (get_field_type :1090-1101): type id of the field or NONE
__device__ __forceinline__ uint32_t record_field(const Fn& f, const TypeRec& rec, const opusfe_token& name) {
    if (rec.src != TS_NODE) return OPUSFE_NONE;
    uint32_t c = f.S.nodes[rec.ref].first_child;
    const uint32_t n = f.S.nodes[rec.ref].nchild;
    for (uint32_t k = 0; k < n && c != OPUSFE_NONE; k++) {
        const opusfe_node m = f.S.nodes[c];
        const opusfe_token t = f.S.tok[m.payload];
        if (t.kind == name.kind && (t.kind != T_Identifier || (t.payload == name.payload && t.len == name.len && bytes_eq(f.S, t.low, name.low, t.len)))) return f.S.tid_of_node[m.first_child];
        c = m.next;
    }
    return OPUSFE_NONE;
}
// generate_ir_for_field_access (:1103-1149) on the record pointer `rec`
template <bool EMIT> __device__ __forceinline__ Val<EMIT> field_access(Fn& f, Em<EMIT>& e, Val<EMIT> rec, uint32_t field_tok, uint32_t pt) {
    Val<EMIT> o = rec;
    if (pt == PT_Array || pt == PT_ArrayMutable) { fallback(f); return o; }
    const Desc pt_t = pointee(f, rec.t);
    if (failed(f)) return o;
    const TView r = view(f, pt_t);
    const opusfe_token name = f.S.tok[field_tok];
    if (r.kind == TY_Record && r.plain) {
        const uint32_t ft = record_field(f, r.rec, name);
        if (ft == OPUSFE_NONE) { fallback(f); return o; }
        o.t = d_ptr(f, pt, d_id(ft));
        o.info = own_rv(e.pos, e.v);
        o.var = e.newvar(f, o.t, true);
        e.ins(f, I_Field, o.var, rec.var, 0, field_tok);
        return o;
    }
    if (r.kind == TY_String) {
        uint32_t idx; Desc ft;
        if (name_is(f.S.src, f.S.tok, field_tok, "data", 4)) { idx = 0; ft = d_ptr(f, PT_Array, d_id(TP_Natural8)); }
        else if (name_is(f.S.src, f.S.tok, field_tok, "length", 6)) { idx = 1; ft = d_id(TP_Size); }
        else { fallback(f); return o; }
        o.t = d_ptr(f, pt, ft);
        o.info = own_rv(e.pos, e.v);
        o.var = e.newvar(f, o.t, true);
        e.ins(f, I_BuiltinField, o.var, rec.var, 0, idx);
        return o;
    }
    fallback(f);
    return o;
}

// type of an unsuffixed integer literal from the expected type of its position (generate_ir.rs:514-549)
__device__ __forceinline__ Desc expected_of(Fn& f, uint32_t i, bool* ready) {
    *ready = true;
    const uint32_t exp = (f.st[i] & ST_EXP_MASK) >> ST_EXP_SHIFT;
    switch (exp) {
        case EXP_GENINT: return d_id(TP_GenericInteger);
        case EXP_BOOL: return d_id(TP_Bool);
        case EXP_RET: return d_id(f.ret);
        case EXP_SIZE: return d_id(TP_Size);
        case EXP_ASSIGN: case EXP_FIELD: case EXP_ARG: {
            uint32_t a = f.pa[i];
            while (a != NONE16 && (KIND(f, a) == E_Parentheses || KIND(f, a) == E_Negate)) a = f.pa[a];
            if (a == NONE16) { fallback(f); return DESC_NONE; }
            if (exp == EXP_ARG) return d_id(((f.ref[a] & RF_TAG) == RF_CALL || (f.ref[a] & RF_TAG) == RF_CALL1) ? (uint32_t)TP_GenericInteger : (uint32_t)TP_Size);   // call argument / index
            if (exp == EXP_ASSIGN) {
                const uint32_t lhs = f.fc[a];
                if (!(f.st[lhs] & ST_DONE)) { *ready = false; return DESC_NONE; }
                return pointee(f, f.rt[lhs]);
            }
            // X_FieldInit under E_Record: the type of the named field, or none when the record has no such field
            const uint32_t recnode = f.pa[a];
            const TView r = view(f, d_id(TID(f, f.fc[recnode])));
            if (r.kind != TY_Record) return DESC_NONE;
            const uint32_t ft = record_field(f, r.rec, f.S.tok[f.pay[a]]);
            return ft == OPUSFE_NONE ? DESC_NONE : d_id(ft);
        }
        default: return DESC_NONE;
    }
}

// the loop a `break` / `continue` at node i belongs to (generate_ir.rs:327-331, 432-436: the loop target is set for the then-block only)
template <class F> __device__ __forceinline__ uint32_t enclosing_loop(const F& f, uint32_t i) {
    uint32_t c = i, a = f.pa[i];
    while (a != NONE16) {
        const uint8_t k = KIND(f, a);
        if (k == S_While) { if (f.nx[f.fc[a]] == c) return a; }
        else if (k == S_For) { const uint32_t iter = SUB(f, a) ? f.nx[f.fc[a]] : f.fc[a]; if (f.nx[iter] == c) return a; }
        c = a; a = f.pa[a];
    }
    return NONE16;
}
// then-block and else-block (or NONE16) of an if / while / for node
template <class F> __device__ __forceinline__ void blocks_of(const F& f, uint32_t s, uint32_t* cond, uint32_t* then_b, uint32_t* else_b) {
    uint32_t c = f.fc[s];
    if (KIND(f, s) == S_For && SUB(f, s)) c = f.nx[c];
    *cond = c; *then_b = f.nx[c]; *else_b = f.nx[*then_b];
}
// Who waits for node i in the schedule of the count pass: its parent — except that an `if` / `while` / `for` does not wait for its
// blocks (all it took from them was whether they diverge, which k_flat_pre knows from the structure); they report to the nearest
// enclosing block or the function instead, so that the function is still lowered behind everything in it (its epilogue looks at
// the last thing emitted).  The rounds of the count pass are then as deep as the expressions, not as the nesting of the statements.
template <class F> __device__ __forceinline__ uint32_t wait_target(const F& f, uint32_t i) {
    uint32_t p = f.pa[i];
    if (p == NONE16) return NONE16;
    const uint8_t pk = KIND(f, p);
    if (pk == S_If || pk == S_While || pk == S_For) {
        uint32_t cond, tb, eb; blocks_of(f, p, &cond, &tb, &eb);
        if (i != cond && (i == tb || i == eb)) {
            uint32_t guard = 0;
            do { p = f.pa[p]; guard++; } while (p != NONE16 && KIND(f, p) != X_Block && KIND(f, p) != D_Function && guard < 4096);
        }
    }
    return p;
}
// block ids of a loop (emit pass)
__device__ __forceinline__ uint32_t loop_start_bb(const Fn& f, uint32_t loop) {
    if (KIND(f, loop) == S_While) return bb_at(f, POS_ENTER(f, loop));
    uint32_t cond, tb, eb; blocks_of(f, loop, &cond, &tb, &eb);
    return bb_at(f, POS_EXIT(f, cond));
}
__device__ __forceinline__ uint32_t merge_bb(const Fn& f, uint32_t s) {
    uint32_t cond, tb, eb; blocks_of(f, s, &cond, &tb, &eb);
    return eb != NONE16 ? bb_at(f, POS_EXIT(f, eb)) : bb_at(f, POS_EXIT(f, tb)) + 1;
}

// Type of the variable of `for x [T] in e` (generate_ir.rs:363-431): `@mut T`, or without a type the parameter type of the
// `continue` overload that the (iterator, <unknown>) call resolves to (Q13).  Needs the type of e; the body waits for this.
__device__ __forceinline__ Desc for_variable_type(Fn& f, uint32_t i) {
    uint32_t cond, tb, eb; blocks_of(f, i, &cond, &tb, &eb);
    const uint32_t tnode = SUB(f, i) ? f.fc[i] : NONE16;
    if (tnode != NONE16) return d_ptr(f, PT_Mutable, d_id(TID(f, tnode)));
    Desc at[2]; at[0] = f.rt[cond];
    if (at[0] == DESC_NONE || view(f, at[0]).kind == TY_Error) { fallback(f); return DESC_NONE; }
    const uint32_t ff = lookup_function(f, key_sym(T_For, f.sym_mix), at, 1);
    if (failed(f) || ff == FN_NONE || fn_builtin(ff)) { fallback(f); return DESC_NONE; }
    at[0] = d_ptr(f, PT_Mutable, d_id(fn_ret(f, ff))); at[1] = d_id(TP_Generic);
    const uint32_t cf = lookup_function(f, key_sym(T_Continue, f.sym_mix), at, 2);
    if (failed(f) || cf == FN_NONE || fn_builtin(cf)) { fallback(f); return DESC_NONE; }
    return d_id(fn_arg_type(f, cf, 1));
}

// Lowers node i: all events it owns.  Returns false when an input is not ready yet (count pass only).
template <int C> __device__ __forceinline__ constexpr bool has_class(int c) { return C == KC_ANY || C == c; }
template <bool EMIT, int C> __device__ __forceinline__ bool visit(Fn& f, uint32_t i) {
    const uint32_t st = f.st[i];
    const uint8_t kind = KIND(f, i);
    const uint32_t mode = st & ST_MODE_MASK;
    const uint32_t pt = (st >> ST_PT_SHIFT) & 3u;
    // ---- statements, blocks, the function itself ----
    if (mode == 0) {
        switch (kind) {
            case D_Function: { if (!has_class<C>(KC_CONTROL)) return true;
                // prologue (generate_ir.rs:133-149) at ENTER; epilogue (:151-171) after the body
                const uint32_t body = last_child(f, i);
                Em<EMIT> e = open_slot<EMIT>(f, POS_ENTER(f, i));
                e.newbb(f);
                { uint32_t a = f.fc[i]; for (uint32_t k = 0; k < f.nargs && a != NONE16; k++, a = f.nx[a]) e.newvar(f, pointee(f, f.rt[a]), true); }
                { uint32_t a = f.fc[i]; for (uint32_t k = 0; k < f.nargs && a != NONE16; k++, a = f.nx[a]) { const uint32_t v = e.newvar(f, f.rt[a], false); e.ins(f, I_Allocate, v); e.ins(f, I_Store, 0, v, k); } }
                close_slot(f, e);
                const bool returns_null = view(f, d_id(f.ret)).kind == TY_Null;
                const bool rdiv = (f.st[body] & ST_DIV) != 0;
                if (!(rdiv || returns_null)) { fallback(f); return true; }             // FunctionMightNotReturn
                // last thing emitted before this point: an instruction (which?) or a new block
                Em<EMIT> x = open_slot<EMIT>(f, POS_EXIT(f, body));
                uint32_t p = POS_EXIT(f, body);
                uint8_t tl = TAIL_NONE;
                while (p > 0 && tl == TAIL_NONE) { p--; tl = f.tail[p]; }
                if (tl != TAIL_TERM) {
                    if (returns_null) { const uint32_t v = x.newvar(f, d_id(TP_Null), true); x.ins(f, I_Null, v); x.ins(f, I_Return, 0, v); }
                    else x.ins(f, I_Unreachable);
                }
                close_slot(f, x);
                return true;
            }
            case X_Block: { if (!has_class<C>(KC_CONTROL)) return true;
                return true;       // (whether it diverges is structural: k_flat_pre left ST_DIV in the state word)
            }
            case S_Expression: return true;
            case S_VariableDefinition: { if (!has_class<C>(KC_STATEMENT)) return true;   // :252-264
                const uint32_t c = f.fc[i];
                Em<EMIT> e = open_slot<EMIT>(f, POS_EXIT(f, c));
                const Desc t = f.rt[c];
                if (view(f, t).kind == TY_Error) { fallback(f); return true; }
                const Desc vt = d_ptr(f, PT_Mutable, t);
                const uint32_t info = own_rv(e.pos, e.v);
                const uint32_t nv = e.newvar(f, vt, false);
                e.ins(f, I_Allocate, nv);
                e.ins(f, I_Store, 0, nv, RV<EMIT>(f, c));
                close_slot(f, e);
                if (!EMIT) { f.rt[i] = vt; f.rvi[i] = info; }      // (the statement node is the slot of its definition)
                return true;
            }
            case S_Assignment: { if (!has_class<C>(KC_STATEMENT)) return true;           // :265-274
                const uint32_t l = f.fc[i], r = f.nx[l];
                Em<EMIT> e = open_slot<EMIT>(f, POS_EXIT(f, r));
                const Desc T = pointee(f, f.rt[l]);
                if (failed(f)) return true;
                Val<EMIT> fin = autocast<EMIT>(f, e, child_val<EMIT>(f, r), r, T);
                e.ins(f, I_Store, 0, RV<EMIT>(f, l), fin.var);
                close_slot(f, e);
                return true;
            }
            case S_Return: { if (!has_class<C>(KC_STATEMENT)) return true;               // :275-283
                const uint32_t c = f.fc[i];
                Em<EMIT> e = open_slot<EMIT>(f, POS_EXIT(f, c));
                Val<EMIT> fin = autocast<EMIT>(f, e, child_val<EMIT>(f, c), c, d_id(f.ret));
                e.ins(f, I_Return, 0, fin.var);
                close_slot(f, e);
                return true;
            }
            case S_If: case S_While: case S_For: { if (!has_class<C>(KC_CONTROL)) return true;
                uint32_t cond, tb, eb; blocks_of(f, i, &cond, &tb, &eb);
                if (tb == NONE16) { fallback(f); return true; }
                uint32_t loop_bb = 0;
                Val<EMIT> cv = child_val<EMIT>(f, cond);
                if (kind == S_While) {     // :315-326
                    Em<EMIT> e0 = open_slot<EMIT>(f, POS_ENTER(f, i));
                    loop_bb = e0.b;
                    e0.ins(f, I_Jump, 0, 0, 0, loop_bb);
                    e0.newbb(f);
                    close_slot(f, e0);
                }
                Em<EMIT> e = open_slot<EMIT>(f, POS_EXIT(f, cond));
                uint32_t cg = EMIT ? f.carg_base + f.cg[i] : 0u;
                if (kind == S_For) {       // :363-431
                    const uint32_t tnode = SUB(f, i) ? f.fc[i] : NONE16;
                    Desc at[2]; at[0] = cv.t;
                    if (view(f, cv.t).kind == TY_Error) { fallback(f); return true; }
                    const uint32_t ff = lookup_function(f, key_sym(T_For, f.sym_mix), at, 1);
                    if (failed(f) || ff == FN_NONE || fn_builtin(ff)) { fallback(f); return true; }
                    const uint32_t for_fn = e.newvar(f, d_func(ff), true);
                    e.ins(f, I_Function, for_fn, 0, 0, ff);
                    Val<EMIT> it = autocast<EMIT>(f, e, cv, cond, d_id(fn_arg_type(f, ff, 0)));
                    const Desc ret_t = d_id(fn_ret(f, ff));
                    const uint32_t ret_var = e.newvar(f, ret_t, true);
                    const Desc iter_t = d_ptr(f, PT_Mutable, ret_t);
                    const uint32_t iterator_variable = e.newvar(f, iter_t, true);
                    if (EMIT) gst(f.A.cargs + cg, it.var);
                    e.ins(f, I_Call, ret_var, for_fn, 0, cg, 0, 1);
                    e.ins(f, I_Allocate, iterator_variable);
                    e.ins(f, I_Store, 0, iterator_variable, ret_var);
                    loop_bb = e.b;
                    e.ins(f, I_Jump, 0, 0, 0, loop_bb);
                    e.newbb(f);
                    Desc pointer_type = tnode != NONE16 ? d_ptr(f, PT_Mutable, d_id(TID(f, tnode))) : d_id(TP_Generic);
                    const uint32_t info = own_rv(e.pos, e.v);
                    // the variable is created with its FINAL type: `<unknown>` is retyped in place by the autocast below (Q13)
                    at[0] = iter_t; at[1] = pointer_type;
                    const uint32_t cf = lookup_function(f, key_sym(T_Continue, f.sym_mix), at, 2);
                    if (failed(f) || cf == FN_NONE || fn_builtin(cf)) { fallback(f); return true; }
                    if (tnode == NONE16) {
                        pointer_type = d_id(fn_arg_type(f, cf, 1));
                        if (view(f, pointer_type).kind == TY_Error) { fallback(f); return true; }
                    }
                    const uint32_t iteration_variable = e.newvar(f, pointer_type, false);
                    e.ins(f, I_Allocate, iteration_variable);
                    if (!EMIT) { f.rt[i] = pointer_type; f.rvi[i] = info; }
                    const uint32_t cont_fn = e.newvar(f, d_func(cf), true);
                    e.ins(f, I_Function, cont_fn, 0, 0, cf);
                    Val<EMIT> a0; a0.var = iterator_variable; a0.t = iter_t; a0.info = 0;
                    a0 = autocast<EMIT>(f, e, a0, NONE16, d_id(fn_arg_type(f, cf, 0)));
                    Val<EMIT> a1; a1.var = iteration_variable; a1.t = pointer_type; a1.info = 0;
                    a1 = autocast<EMIT>(f, e, a1, NONE16, d_id(fn_arg_type(f, cf, 1)));
                    if (failed(f)) return true;
                    cv.t = d_id(fn_ret(f, cf));
                    cv.var = e.newvar(f, cv.t, true);
                    if (EMIT) { gst(f.A.cargs + cg + 1, a0.var); gst(f.A.cargs + cg + 2, a1.var); }
                    e.ins(f, I_Call, cv.var, cont_fn, 0, cg + 1, 0, 2);
                    cond = NONE16;         // the condition is no node's result: a <integer> cannot reach the autocast below
                }
                cv = autocast<EMIT>(f, e, cv, cond, d_id(TP_Bool));
                if (failed(f)) return true;
                const uint32_t then_bb = e.b;
                uint32_t else_bb = 0, merge = 0;
                if (EMIT) { else_bb = bb_at(f, POS_EXIT(f, tb)); merge = merge_bb(f, i); }
                e.ins(f, I_Branch, 0, cv.var, then_bb, else_bb);
                e.newbb(f);
                close_slot(f, e);
                Em<EMIT> e2 = open_slot<EMIT>(f, POS_EXIT(f, tb));
                e2.ins(f, I_Jump, 0, 0, 0, kind == S_If ? merge : loop_bb);
                e2.newbb(f);
                if (eb == NONE16) { e2.ins(f, I_Jump, 0, 0, 0, merge); e2.newbb(f); }
                close_slot(f, e2);
                if (eb != NONE16) {
                    Em<EMIT> e3 = open_slot<EMIT>(f, POS_EXIT(f, eb));
                    e3.ins(f, I_Jump, 0, 0, 0, merge);
                    e3.newbb(f);
                    close_slot(f, e3);
                }
                return true;
            }
            case S_Break: case S_Continue: { if (!has_class<C>(KC_CONTROL)) return true;   // :470-489
                const uint32_t loop = f.ref[i];
                if (loop == NONE16) { fallback(f); return true; }
                Em<EMIT> e = open_slot<EMIT>(f, POS_ENTER(f, i));
                uint32_t target = 0;
                if (EMIT) target = kind == S_Break ? merge_bb(f, loop) : loop_start_bb(f, loop);
                e.ins(f, I_Jump, 0, 0, 0, target);
                e.newbb(f);
                close_slot(f, e);
                return true;
            }
            default: return true;          // type nodes, field-init / named-type carriers: nothing of their own
        }
    }
    // ---- expressions ----
    if (mode == ST_MODE_E) {
        switch (kind) {
            case E_Integer: { if (!has_class<C>(KC_INTEGER)) return true;
                // ref[i] = the token's aux word (suffix) | the final type + 1 of the variable << 24 once an autocast retyped it
                const uint32_t aux = f.ref[i] & 0xFFFF;
                Desc typ;
                if (aux & 0x8000) {
                    const uint32_t size = aux & 0xFF; const bool sg = (aux & 0x100) != 0;
                    uint32_t p;
                    if (!sg) p = size == 8 ? TP_Natural8 : size == 16 ? TP_Natural16 : size == 32 ? TP_Natural32 : size == 64 ? TP_Natural64 : TP_Size;
                    else p = size == 8 ? TP_Integer8 : size == 16 ? TP_Integer16 : size == 32 ? TP_Integer32 : size == 64 ? TP_Integer64 : TP_Offset;
                    typ = d_id(p);
                } else {
                    bool ready;
                    const Desc exp = expected_of(f, i, &ready);
                    if (!ready) return false;
                    if (failed(f)) return true;
                    uint8_t k = TY_Error;
                    if (exp != DESC_NONE) k = view(f, exp).kind;
                    typ = (exp != DESC_NONE && (kind_is_integral(k) || k == TY_GenericInteger)) ? exp : d_id(TP_Integer32);
                }
                uint64_t val = 0;
                if (EMIT) {
                    const opusfe_token t = f.S.tok[f.pay[i]];
                    val = (t.flags & OPUSFE_TF_INLINE) ? (uint64_t)t.payload : f.S.val64[t.payload];
                    // a <integer> literal under unary minus is negated in place (generate_ir.rs:766-780): one sign flip per Negate above it
                    if (view(f, typ).kind == TY_GenericInteger) {
                        uint32_t a = f.pa[i]; bool neg = false;
                        while (a != NONE16 && (KIND(f, a) == E_Parentheses || KIND(f, a) == E_Negate)) { if (KIND(f, a) == E_Negate) neg = !neg; a = f.pa[a]; }
                        if (neg) val = 0 - val;
                    }
                }
                Em<EMIT> e = open_slot<EMIT>(f, POS_ENTER(f, i));
                Val<EMIT> o; o.t = typ; o.info = own_rv(e.pos, e.v);
                const uint32_t fin = f.ref[i] >> 24;
                const Desc final_t = (EMIT && fin) ? d_id(fin - 1) : typ;
                o.var = e.newvar(f, final_t, true);
                e.ins(f, I_Integer, o.var, 0, 0, (uint32_t)val, (uint32_t)(val >> 32));
                close_slot(f, e);
                set_result<EMIT>(f, i, o);
                return true;
            }
            case E_Float: case E_Null: case E_Bool: case E_String: { if (!has_class<C>(KC_SIMPLE)) return true;
                Em<EMIT> e = open_slot<EMIT>(f, POS_ENTER(f, i));
                Val<EMIT> o; o.info = own_rv(e.pos, e.v);
                if (kind == E_Float) {
                    uint64_t bits = 0;
                    if (EMIT) bits = f.S.val64[f.S.tok[f.pay[i]].payload];
                    o.t = d_id((f.ref[i] & 0xFFFF) == 32 ? TP_Float32 : TP_Float64);      // (the token's aux word, fetched with the names)
                    o.var = e.newvar(f, o.t, true);
                    e.ins(f, I_Float, o.var, 0, 0, (uint32_t)bits, (uint32_t)(bits >> 32));
                } else if (kind == E_Null) { o.t = d_id(TP_Null); o.var = e.newvar(f, o.t, true); e.ins(f, I_Null, o.var); }
                else if (kind == E_Bool) { o.t = d_id(TP_Bool); o.var = e.newvar(f, o.t, true); e.ins(f, I_Bool, o.var, 0, 0, 0, 0, SUB(f, i)); }
                else { o.t = d_id(TP_String); o.var = e.newvar(f, o.t, true); e.ins(f, I_String, o.var, 0, 0, f.pay[i]); }
                close_slot(f, e);
                set_result<EMIT>(f, i, o);
                return true;
            }
            case E_Operator: { if (!has_class<C>(KC_OPERATOR)) return true;
                const uint32_t l = f.fc[i], r = f.nx[l];
                const uint8_t op = SUB(f, i);
                if (op == OP_Is) {       // :616-634: pointer identity; anything else is InvalidOperandTypes (sequential machine)
                    const TView vl = view(f, f.rt[l]), vr = view(f, f.rt[r]);
                    if (vl.kind != TY_Pointer || vr.kind != TY_Pointer || vl.ptr != vr.ptr || match3(f, vl.sub, vr.sub) != M_YES) { fallback(f); return true; }
                    Em<EMIT> e = open_slot<EMIT>(f, POS_EXIT(f, r));
                    Val<EMIT> o; o.t = d_id(TP_Bool); o.info = own_rv(e.pos, e.v);
                    o.var = e.newvar(f, o.t, true);
                    e.ins(f, I_Equals, o.var, child_val<EMIT>(f, l).var, child_val<EMIT>(f, r).var);
                    close_slot(f, e);
                    set_result<EMIT>(f, i, o);
                    return true;
                }
                if (op == OP_And || op == OP_Or) {   // :635-669
                    Em<EMIT> e = open_slot<EMIT>(f, POS_EXIT(f, l));
                    Val<EMIT> c = autocast<EMIT>(f, e, child_val<EMIT>(f, l), l, d_id(TP_Bool));
                    if (failed(f)) return true;
                    const uint32_t branch_bb = e.b - 1, then_bb = e.b;
                    uint32_t merge = 0;
                    if (EMIT) merge = bb_at(f, POS_EXIT(f, r));
                    if (op == OP_And) e.ins(f, I_Branch, 0, c.var, then_bb, merge); else e.ins(f, I_Branch, 0, c.var, merge, then_bb);
                    e.newbb(f);
                    close_slot(f, e);
                    if (match3(f, f.rt[r], d_id(TP_Bool)) != M_YES) { fallback(f); return true; }   // Q15
                    Em<EMIT> e2 = open_slot<EMIT>(f, POS_EXIT(f, r));
                    const uint32_t then_end = e2.b - 1;
                    e2.ins(f, I_Jump, 0, 0, 0, e2.b);
                    e2.newbb(f);
                    Val<EMIT> o; o.t = d_id(TP_Bool); o.info = own_rv(e2.pos, e2.v);
                    o.var = e2.newvar(f, o.t, true);
                    e2.ins(f, I_Phi, o.var, c.var, RV<EMIT>(f, r), branch_bb, then_end);
                    close_slot(f, e2);
                    set_result<EMIT>(f, i, o);
                    return true;
                }
                // `not` (the left operand is never evaluated, Q9) and the binary builtin / overloaded operators (:603-718)
                const bool is_not = op == OP_Not;
                const uint8_t sym = sym_of_op(op);
                Em<EMIT> e = open_slot<EMIT>(f, POS_EXIT(f, r));
                Desc at[2]; uint32_t n;
                if (is_not) { at[0] = f.rt[r]; n = 1; } else { at[0] = f.rt[l]; at[1] = f.rt[r]; n = 2; }
                for (uint32_t k = 0; k < n; k++) if (at[k] == DESC_NONE || view(f, at[k]).kind == TY_Error) { fallback(f); return true; }
                const uint32_t fn = lookup_function(f, key_sym(sym, f.sym_mix), at, n);
                if (failed(f) || fn == FN_NONE) { fallback(f); return true; }
                uint32_t fv = 0;
                if (!fn_builtin(fn)) { fv = e.newvar(f, d_func(fn), true); e.ins(f, I_Function, fv, 0, 0, fn); }
                Val<EMIT> a0, a1{};
                if (is_not) a0 = autocast<EMIT>(f, e, child_val<EMIT>(f, r), r, d_id(fn_arg_type(f, fn, 0)));
                else {
                    a0 = autocast<EMIT>(f, e, child_val<EMIT>(f, l), l, d_id(fn_arg_type(f, fn, 0)));
                    a1 = autocast<EMIT>(f, e, child_val<EMIT>(f, r), r, d_id(fn_arg_type(f, fn, 1)));
                }
                if (failed(f)) return true;
                Val<EMIT> o; o.t = d_id(fn_ret(f, fn)); o.info = own_rv(e.pos, e.v);
                o.var = e.newvar(f, o.t, true);
                if (fn_builtin(fn)) {
                    uint16_t opc;
                    switch (op) {
                        case OP_Plus: opc = I_Add; break; case OP_Minus: opc = I_Subtract; break; case OP_Times: opc = I_Multiply; break;
                        case OP_Divide: opc = I_Divide; break; case OP_Modulo: opc = I_Modulo; break; case OP_Equals: opc = I_Equals; break;
                        case OP_LessThan: opc = I_LessThan; break; case OP_GreaterThan: opc = I_GreaterThan; break;
                        case OP_LessThanEquals: opc = I_LessThanEquals; break; case OP_Not: opc = I_Not; break; default: opc = I_GreaterThanEquals; break;
                    }
                    if (is_not) e.ins(f, I_Not, o.var, a0.var); else e.ins(f, opc, o.var, a0.var, a1.var);
                } else {
                    const uint32_t cg = EMIT ? f.carg_base + f.cg[i] : 0u;
                    if (EMIT) { gst(f.A.cargs + cg, a0.var); if (n > 1) gst(f.A.cargs + cg + 1, a1.var); }
                    e.ins(f, I_Call, o.var, fv, 0, cg, 0, (uint16_t)n);
                }
                close_slot(f, e);
                set_result<EMIT>(f, i, o);
                return true;
            }
            case E_Cast: { if (!has_class<C>(KC_OTHER)) return true;                   // :720-732
                const uint32_t c = f.fc[i], ty = f.nx[c];
                Em<EMIT> e = open_slot<EMIT>(f, POS_EXIT(f, ty));
                const Desc typ = d_id(TID(f, ty));
                const int ct = can_cast3(f, f.rt[c], typ);
                if (ct < 0 || view(f, f.rt[c]).kind == TY_Error || view(f, typ).kind == TY_Error) { fallback(f); return true; }
                Val<EMIT> o; o.t = typ; o.info = own_rv(e.pos, e.v);
                o.var = e.newvar(f, typ, true);
                e.ins(f, I_Cast, o.var, RV<EMIT>(f, c), 0, 0, 0, (uint16_t)ct);
                close_slot(f, e);
                set_result<EMIT>(f, i, o);
                return true;
            }
            case E_Reference: case E_MutableReference: case E_ArrayReference: case E_MutableArrayReference: { if (!has_class<C>(KC_OTHER)) return true;   // :733-748: the child's pointer itself
                Val<EMIT> o = child_val<EMIT>(f, f.fc[i]);
                set_result<EMIT>(f, i, o);
                return true;
            }
            case E_Negate: { if (!has_class<C>(KC_SIMPLE)) return true;                 // :749-789
                const uint32_t c = f.fc[i];
                const Desc stp = f.rt[c];
                const uint8_t k = view(f, stp).kind;
                if (k == TY_Integer8 || k == TY_Integer16 || k == TY_Integer32 || k == TY_Integer64 || k == TY_Offset) {
                    Em<EMIT> e = open_slot<EMIT>(f, POS_EXIT(f, c));
                    Val<EMIT> o; o.t = stp; o.info = own_rv(e.pos, e.v);
                    o.var = e.newvar(f, stp, true);
                    e.ins(f, I_Negate, o.var, RV<EMIT>(f, c));
                    close_slot(f, e);
                    set_result<EMIT>(f, i, o);
                } else if (k == TY_GenericInteger) {
                    // folded into the literal (its lane flips the sign); the literal must be what this chain ends in
                    if (KIND(f, creator_of(f, c)) != E_Integer && !EMIT) { fallback(f); return true; }
                    set_result<EMIT>(f, i, child_val<EMIT>(f, c));
                } else fallback(f);
                return true;
            }
            case E_Sizeof: case E_Alignof: { if (!has_class<C>(KC_OTHER)) return true;  // :790-801
                Em<EMIT> e = open_slot<EMIT>(f, POS_ENTER(f, i));
                Val<EMIT> o; o.t = d_id(TP_Size); o.info = own_rv(e.pos, e.v);
                o.var = e.newvar(f, o.t, true);
                e.ins(f, kind == E_Sizeof ? I_Sizeof : I_Alignof, o.var, 0, 0, TID(f, f.fc[i]));
                close_slot(f, e);
                set_result<EMIT>(f, i, o);
                return true;
            }
            case E_Parentheses: set_result<EMIT>(f, i, child_val<EMIT>(f, f.fc[i])); return true;
            default: fallback(f); return true;   // E_Proc and anything unexpected
        }
    }
    // ---- pointers / lvalues (try_generate_pointer_from_expression :985-1088) ----
    switch (kind) {
        case E_Variable: { if (!has_class<C>(KC_VARIABLE)) return true;
            const uint32_t r = f.ref[i], tag = r & RF_TAG, idx = r & RF_VAL;
            Em<EMIT> e = open_slot<EMIT>(f, POS_ENTER(f, i));
            Val<EMIT> v{}; bool lv = true;
            if (tag == RF_LOCAL) {
                const Desc dvt = f.rt[idx];                      // idx = the slot node of the definition
                if (!EMIT && dvt == DESC_NONE) return false;    // its type is not known yet (the scheduler never lets this happen)
                v.t = dvt; v.info = RV_LOCAL | idx; v.var = EMIT ? (f.rvi[idx] & RV_VAL) : 0u;
                v = try_cast_pointer<EMIT>(f, e, v, pt);
            } else if (tag == RF_ARG) {
                v.t = f.rt[idx >> 4]; v.info = RV_ARG | (idx & 15u); v.var = f.nargs + (idx & 15u);      // idx = argument number | its X_NamedType node << 4
                v = try_cast_pointer<EMIT>(f, e, v, pt);
            } else if (tag == RF_GLOBAL) {
                v.t = d_ptr(f, PT_Mutable, EMIT ? d_id(f.S.globals[idx].type_id) : f.rt[i]);
                v.info = own_rv(e.pos, e.v);
                v.var = e.newvar(f, v.t, true);
                e.ins(f, I_GlobalVariable, v.var, 0, 0, idx);
                v = try_cast_pointer<EMIT>(f, e, v, pt);
            } else if (tag == RF_CONST) {
                v.t = EMIT ? d_id(f.S.globals[idx].type_id) : f.rt[i];
                v.info = own_rv(e.pos, e.v);
                v.var = e.newvar(f, v.t, true);
                e.ins(f, I_GlobalConstant, v.var, 0, 0, idx);
                lv = false;
            } else if (tag == RF_FUNC) {
                v.t = d_func(idx);
                v.info = own_rv(e.pos, e.v);
                v.var = e.newvar(f, v.t, true);
                e.ins(f, I_Function, v.var, 0, 0, idx);
                lv = false;
            } else { fallback(f); return true; }
            if (failed(f)) return true;
            v = p_result<EMIT>(f, e, v, lv, st);
            close_slot(f, e);
            set_result<EMIT>(f, i, v);
            return true;
        }
        case E_VariableDefinition: { if (!has_class<C>(KC_OTHER)) return true;         // :1008-1021
            Em<EMIT> e = open_slot<EMIT>(f, POS_EXIT(f, f.fc[i]));
            const uint32_t slot = f.fc[i];            // (the type node is the slot of this definition)
            Val<EMIT> v; v.t = f.rt[slot]; v.info = own_rv(e.pos, e.v);
            v.var = e.newvar(f, v.t, false);
            if (!EMIT) f.rvi[slot] = v.info;
            e.ins(f, I_Allocate, v.var);
            v = try_cast_pointer<EMIT>(f, e, v, pt);
            if (failed(f)) return true;
            v = p_result<EMIT>(f, e, v, true, st);
            close_slot(f, e);
            set_result<EMIT>(f, i, v);
            return true;
        }
        case E_Dereference: { if (!has_class<C>(KC_OTHER)) return true;                // :1022-1026
            const uint32_t c = f.fc[i];
            Em<EMIT> e = open_slot<EMIT>(f, POS_EXIT(f, c));
            Val<EMIT> v = try_cast_pointer<EMIT>(f, e, child_val<EMIT>(f, c), pt);
            if (failed(f)) return true;
            v = p_result<EMIT>(f, e, v, true, st);
            close_slot(f, e);
            set_result<EMIT>(f, i, v);
            return true;
        }
        case E_Field: { if (!has_class<C>(KC_OTHER)) return true;                      // :1027-1030
            const uint32_t c = f.fc[i];
            Em<EMIT> e = open_slot<EMIT>(f, POS_EXIT(f, c));
            Val<EMIT> v = field_access<EMIT>(f, e, child_val<EMIT>(f, c), f.pay[i], pt);
            if (failed(f)) return true;
            v = p_result<EMIT>(f, e, v, true, st);
            close_slot(f, e);
            set_result<EMIT>(f, i, v);
            return true;
        }
        case E_NamedCall: { if (!has_class<C>(KC_CALL)) return true;
            const uint32_t r = f.ref[i], tag = r & RF_TAG, idx = r & RF_VAL;
            const uint32_t n = NCH(f, i);
            const uint32_t lastc = last_child(f, i);
            if (tag == RF_CALL || tag == RF_CALL1) {   // :1031-1045 + get_function_call_variables :1181-1198
                if (n > MAX_CALL_ARGS) { fallback(f); return true; }
                Em<EMIT> e = open_slot<EMIT>(f, n ? POS_EXIT(f, lastc) : POS_ENTER(f, i));
                Desc at[MAX_CALL_ARGS]; uint32_t an[MAX_CALL_ARGS], want[MAX_CALL_ARGS];
                uint32_t k = 0;
                for (uint32_t c = f.fc[i]; c != NONE16 && k < MAX_CALL_ARGS; c = f.nx[c], k++) { at[k] = f.rt[c]; an[k] = c; if (at[k] == DESC_NONE || view(f, at[k]).kind == TY_Error) { fallback(f); return true; } }
                uint32_t fn; Desc ret_t;
                if (tag == RF_CALL1) {
                    // the only candidate: it is the callee iff every argument autocasts to its parameter (compile.rs:1085-1139 with one function)
                    fn = idx;
                    if (EMIT) { for (k = 0; k < n; k++) want[k] = fn_arg_type(f, fn, k); ret_t = d_id(fn_ret(f, fn)); }
                    else {
                        for (k = 0; k < n; k++) {
                            want[k] = (uint32_t)f.cnt[POS_EXIT(f, an[k])];
                            f.cnt[POS_EXIT(f, an[k])] = 0;                     // back to an empty event
                            const int c = can_autocast3(f, at[k], d_id(want[k]));
                            if (c < 0) { fallback(f); return true; }
                        }
                        ret_t = f.rt[i];
                    }
                } else {
                    fn = lookup_function(f, key_tok(f.pay[i]), at, n);
                    if (failed(f) || fn == FN_NONE || fn_builtin(fn)) { fallback(f); return true; }
                    for (k = 0; k < n; k++) want[k] = fn_arg_type(f, fn, k);
                    ret_t = d_id(fn_ret(f, fn));
                }
                const uint32_t fv = e.newvar(f, d_func(fn), true);
                e.ins(f, I_Function, fv, 0, 0, fn);
                const uint32_t cg = EMIT ? f.carg_base + f.cg[i] : 0u;
                for (k = 0; k < n; k++) {
                    Val<EMIT> a = autocast<EMIT>(f, e, child_val<EMIT>(f, an[k]), an[k], d_id(want[k]));
                    if (EMIT) gst(f.A.cargs + cg + k, a.var);
                }
                if (failed(f)) return true;
                Val<EMIT> o; o.t = ret_t; o.info = own_rv(e.pos, e.v);
                o.var = e.newvar(f, o.t, true);
                e.ins(f, I_Call, o.var, fv, 0, cg, 0, (uint16_t)n);
                o = p_result<EMIT>(f, e, o, false, st);
                if (failed(f)) return true;
                close_slot(f, e);
                set_result<EMIT>(f, i, o);
                return true;
            }
            // index through a variable (:1046-1052 + generate_ir_for_index :895-967); the indexee is an lvalue
            if (n != 1) { fallback(f); return true; }
            Em<EMIT> e = open_slot<EMIT>(f, POS_ENTER(f, i));
            Val<EMIT> v{};
            if (tag == RF_LOCAL) {
                const Desc dvt = f.rt[idx];
                if (!EMIT && dvt == DESC_NONE) return false;
                v.t = dvt; v.var = EMIT ? (f.rvi[idx] & RV_VAL) : 0u;
            } else if (tag == RF_ARG) { v.t = f.rt[idx >> 4]; v.var = f.nargs + (idx & 15u); }
            else if (tag == RF_GLOBAL) { v.t = d_ptr(f, PT_Mutable, EMIT ? d_id(f.S.globals[idx].type_id) : f.rt[i]); v.var = e.newvar(f, v.t, true); e.ins(f, I_GlobalVariable, v.var, 0, 0, idx); }
            else { fallback(f); return true; }
            v = try_cast_pointer<EMIT>(f, e, v, pt);
            if (failed(f)) return true;
            const Desc indexee_t = pointee(f, v.t);
            const TView it = view(f, indexee_t);
            bool is_ptr;
            if (it.kind == TY_Pointer) {       // a pointer variable: load it, it must be an array pointer
                Val<EMIT> ld; ld.t = indexee_t; ld.var = e.newvar(f, indexee_t, true); ld.info = 0;
                e.ins(f, I_Load, ld.var, v.var);
                v = ld; is_ptr = true;
                if (it.ptr != PT_Array && it.ptr != PT_ArrayMutable) { fallback(f); return true; }
            } else if (it.kind == TY_Array) is_ptr = false;
            else { fallback(f); return true; } // functions through variables, errors
            close_slot(f, e);
            Em<EMIT> e2 = open_slot<EMIT>(f, POS_EXIT(f, lastc));
            const TView vt = view(f, v.t);
            Val<EMIT> o;
            o.t = is_ptr ? d_ptr(f, vt.ptr, vt.sub) : d_ptr(f, vt.ptr, it.sub);
            o.info = own_rv(e2.pos, e2.v);
            o.var = e2.newvar(f, o.t, true);
            e2.ins(f, is_ptr ? I_IndexPointer : I_IndexArray, o.var, v.var, RV<EMIT>(f, lastc));
            o = try_cast_pointer<EMIT>(f, e2, o, pt);
            if (failed(f)) return true;
            o = p_result<EMIT>(f, e2, o, true, st);
            if (failed(f)) return true;
            close_slot(f, e2);
            set_result<EMIT>(f, i, o);
            return true;
        }
        case E_Record: { if (!has_class<C>(KC_OTHER)) return true;                     // :1064-1078
            const uint32_t tyn = f.fc[i];
            Em<EMIT> e = open_slot<EMIT>(f, POS_EXIT(f, tyn));
            const Desc rec_t = d_id(TID(f, tyn));
            if (view(f, rec_t).kind != TY_Record) { fallback(f); return true; }
            Val<EMIT> out; out.t = d_ptr(f, PT_Mutable, rec_t); out.info = own_rv(e.pos, e.v);
            out.var = e.newvar(f, out.t, true);
            e.ins(f, I_Allocate, out.var);
            Val<EMIT> res = out;
            uint32_t fi = f.nx[tyn];
            if (fi == NONE16) { res = p_result<EMIT>(f, e, out, true, st); }
            close_slot(f, e);
            for (; fi != NONE16; fi = f.nx[fi]) {
                Em<EMIT> e2 = open_slot<EMIT>(f, POS_EXIT(f, fi));
                const uint32_t val = f.fc[fi];
                Val<EMIT> fv = field_access<EMIT>(f, e2, out, f.pay[fi], PT_Mutable);
                if (failed(f)) return true;
                e2.ins(f, I_Store, 0, fv.var, RV<EMIT>(f, val));
                if (f.nx[fi] == NONE16) res = p_result<EMIT>(f, e2, out, true, st);
                close_slot(f, e2);
            }
            if (failed(f)) return true;
            set_result<EMIT>(f, i, res);
            return true;
        }
        case E_Parentheses: set_result<EMIT>(f, i, child_val<EMIT>(f, f.fc[i])); return true;
        default: fallback(f); return true;     // E_CallOrIndex
    }
}

// ---- kind classes ---------------------------------------------------------------------------------------------------------
// class of a node from its state word and kind (the visitor is one big switch; every class kernel holds only its cases)
__host__ __device__ __forceinline__ uint32_t kind_class_of(uint32_t s, uint8_t k) {
    if (s & ST_DEAD) return KC_NONE;
    if ((s & ST_MODE_MASK) == 0) {
        if (k == S_Expression || k == S_Assignment || k == S_VariableDefinition || k == S_Return) return KC_STATEMENT;
        if (k == S_If || k == S_While || k == S_For || k == S_Break || k == S_Continue || k == X_Block || k == D_Function || k == X_FieldInit) return KC_CONTROL;   // (a field initialiser waits for its value)
        return KC_NONE;
    }
    if (k == E_Variable) return KC_VARIABLE;
    if (k == E_Integer) return KC_INTEGER;
    if (k == E_Operator) return KC_OPERATOR;
    if (k == E_Parentheses || k == E_Negate || k == E_Float || k == E_Bool || k == E_Null || k == E_String) return KC_SIMPLE;
    if (k == E_NamedCall) return KC_CALL;
    return KC_OTHER;
}
// ---- the lane's function ----------------------------------------------------------------------------------------------------
// false: the function is off the flat path (declined by an earlier visit)
__device__ __forceinline__ bool make_fn(Fn& f, const SemView& S, const ParArgs& A, const FlatArgs& X, uint32_t g, uint32_t* local, bool emitting) {
    // (only nodes of functions that k_flat_pre handed over are ever listed; the record itself says where the function starts, so
    // the loads of the operands' records do not wait for the function's)
    const uint64_t w3 = gld((const uint64_t*)(X.rec + g) + 3);
    const uint32_t fidx = (uint32_t)(w3 >> 32), loc = (uint32_t)(w3 >> 16) & 0xFFFFu;
    const FnInfo I = gld(X.fninfo + fidx);
    f.S = S; f.A = A; f.dn = g - loc; f.nn = I.nn; f.fidx = fidx; f.nargs = I.nargs; f.ret = I.ret; f.sym_mix = I.sym_mix; f.user_syms = *A.user_syms;
    f.fb = false; f.emitting = emitting; f.fbp = (G_AS uint32_t*)X.fb;
    const uint32_t dn = g - loc;
    G_AS uint8_t* const rp = (G_AS uint8_t*)X.rec;
    f.rt = RecArr<uint64_t, 0>{rp, dn}; f.st = RecArr<uint32_t, 8>{rp, dn}; f.ref = RecArr<uint32_t, 12>{rp, dn}; f.rvi = RecArr<uint32_t, 16>{rp, dn};
    f.pe = RecArr<uint16_t, 20>{rp, dn}; f.sz = RecArr<uint16_t, 22>{rp, dn}; f.pa = RecArr<uint16_t, 24>{rp, dn};
    f.cnt = GArr<uint64_t>{(G_AS uint64_t*)X.cnt, 2u * dn}; f.cg = GArr<uint16_t>{(G_AS uint16_t*)X.cg, dn}; f.tail = GArr<uint8_t>{(G_AS uint8_t*)X.tail, 2u * dn};
    const G_AS opusfe_node* nd = (const G_AS opusfe_node*)S.nodes.ptr();
    f.kn = NodeKn{nd, dn}; f.fc = NodeFc{nd, dn}; f.nx = NodeNx{nd, dn}; f.pay = NodePay{nd, dn};
    f.var_base = f.instr_base = f.bb_base = f.type_base = f.carg_base = 0;
    f.n_vars_tot = f.n_instrs_tot = f.n_bbs_tot = f.n_types_tot = f.n_cargs_tot = 0;
    *local = g - dn;
    return gld(X.fb + fidx) == 0;
}

// ---- the schedule of the count pass -------------------------------------------------------------------------------------
// A node can be lowered once its operands, the definition of the local it reads and (an integer literal on the right of an
// assignment) the left side are lowered.  k_flat_pre plays that through per function with counters in LDS — no visit, only "who
// releases whom" — and gives every node the ROUND in which it is lowered; the count pass is then one launch per (round, class) over a
// list that was sorted before it started: no atomic, no queue, no host look at a counter between rounds.
// Sorting = counting sort of (bin, node) pairs with bin = round * KC_ACTIVE + class: k_sched_hist, k_sched_scan, k_sched_scatter.
enum : uint32_t { SCHED_STAGE1 = 1u << 31 };      // list entry: stage 1 of a `for` node (its loop variable gets its type), not the node itself
__device__ __forceinline__ bool single_operand(uint8_t k) {      // a node of this kind waits for ONE operand
    return k == E_Parentheses || k == E_Negate || k == S_Expression || k == S_VariableDefinition || k == S_Return || k == E_Reference || k == E_MutableReference
        || k == E_ArrayReference || k == E_MutableArrayReference || k == E_Dereference || k == E_Field || k == E_Cast || k == X_FieldInit;
}
// is node c an unsuffixed integer literal that takes its type from the left side of its assignment (generate_ir.rs:265-274, 514-549)?
template <class F> __device__ __forceinline__ bool literal_waits_for_lhs(const F& f, uint32_t c) {
    const uint32_t s = f.st[c];
    return KIND(f, c) == E_Integer && !(s & ST_DEAD) && (s & ST_MODE_MASK) == ST_MODE_E && ((s & ST_EXP_MASK) >> ST_EXP_SHIFT) == EXP_ASSIGN && !(f.ref[c] & 0x8000u);
}
// Both passes over the (bin, node) pairs: a block histograms its share in LDS (64 KB: one word per bin); HIST adds the non-empty bins
// to the unit's table, SCATTER takes a range of every non-empty bin from the bin's cursor and deals it out to the block's nodes.
template <bool SCATTER> __global__ __launch_bounds__(1024) void k_sched_pass(FlatArgs X, uint32_t n_nodes, uint32_t per_block, uint32_t cap) {
    const uint16_t* const key = (const uint16_t*)X.tail;
#ifdef OPUSFE_EMU
    // (one emulated lane: plain counting)
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (uint32_t g = 0; g < n_nodes; g++) for (int w = 0; w < 2; w++) {
        const uint32_t k = w ? X.key1[g] : key[g];
        if (k == NONE16) continue;
        if (k >= SCHED_BINS) { X.ctl[CTL_LISTOVF] = 1; continue; }
        if (!SCATTER) X.ctl[CTL_BINS + k]++;
        else {
            const uint32_t at = X.ctl[CTL_CURSOR + k]++;
            if (at < cap) X.sched[at] = g | (w ? (uint32_t)SCHED_STAGE1 : 0u); else X.ctl[CTL_LISTOVF] = 1;
            if (!w) { const uint32_t ao = X.ctl[CTL_OCUR + k / SCHED_R]++; if (ao < cap) X.order[ao] = g; else X.ctl[CTL_LISTOVF] = 1; }
        }
    }
#else
    extern __shared__ __align__(16) uint32_t sched_lds[];      // SCHED_BINS words, then KC_ACTIVE words (the class list)
    uint32_t* const cls = sched_lds + SCHED_BINS;
    for (uint32_t b = threadIdx.x; b < SCHED_BINS + KC_ACTIVE; b += blockDim.x) sched_lds[b] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * per_block, hi = lo + per_block < n_nodes ? lo + per_block : n_nodes;
    for (uint64_t g = lo + threadIdx.x; g < hi; g += blockDim.x) {
        const uint32_t k = key[g], k1 = X.key1[g];
        if (k != NONE16) { if (k < SCHED_BINS) { atomicAdd(&sched_lds[k], 1u); if (SCATTER) atomicAdd(&cls[k / SCHED_R], 1u); } else atomicMax(X.ctl + CTL_LISTOVF, 1u); }
        if (k1 != NONE16) { if (k1 < SCHED_BINS) atomicAdd(&sched_lds[k1], 1u); else atomicMax(X.ctl + CTL_LISTOVF, 1u); }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < SCHED_BINS; b += blockDim.x) {
        const uint32_t c = sched_lds[b];
        if (!c) continue;
        if (!SCATTER) atomicAdd(X.ctl + CTL_BINS + b, c);
        else sched_lds[b] = atomicAdd(X.ctl + CTL_CURSOR + b, c);      // the block's first slot of the bin
    }
    if (!SCATTER) return;
    if (threadIdx.x < KC_ACTIVE) { const uint32_t c = cls[threadIdx.x]; cls[threadIdx.x] = c ? atomicAdd(X.ctl + CTL_OCUR + threadIdx.x, c) : 0u; }
    __syncthreads();
    for (uint64_t g = lo + threadIdx.x; g < hi; g += blockDim.x) {
        const uint32_t k = key[g], k1 = X.key1[g];
        if (k != NONE16 && k < SCHED_BINS) {
            const uint32_t at = atomicAdd(&sched_lds[k], 1u), ao = atomicAdd(&cls[k / SCHED_R], 1u);
            if (at < cap) X.sched[at] = (uint32_t)g; else atomicMax(X.ctl + CTL_LISTOVF, 1u);
            if (ao < cap) X.order[ao] = (uint32_t)g; else atomicMax(X.ctl + CTL_LISTOVF, 1u);
        }
        if (k1 != NONE16 && k1 < SCHED_BINS) { const uint32_t at = atomicAdd(&sched_lds[k1], 1u); if (at < cap) X.sched[at] = (uint32_t)g | SCHED_STAGE1; else atomicMax(X.ctl + CTL_LISTOVF, 1u); }
    }
#endif
}
// lengths -> exclusive offsets (CTL_BINS, SCHED_BINS + 1 words) and the cursors of the scatter pass.  One block.
__global__ __launch_bounds__(1024) void k_sched_scan(uint32_t* __restrict__ ctl) {
#ifdef OPUSFE_EMU
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t run = 0;
    for (uint32_t b = 0; b < SCHED_BINS; b++) { const uint32_t c = ctl[CTL_BINS + b]; ctl[CTL_BINS + b] = run; ctl[CTL_CURSOR + b] = run; run += c; }
    ctl[CTL_BINS + SCHED_BINS] = run;
    for (uint32_t c = 0; c < KC_ACTIVE; c++) ctl[CTL_OCUR + c] = ctl[CTL_BINS + c * SCHED_R];
#else
    __shared__ uint32_t part[1024];
    constexpr uint32_t PER = SCHED_BINS / 1024;
    const uint32_t t = threadIdx.x;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < PER; k++) sum += ctl[CTL_BINS + t * PER + k];
    part[t] = sum;
    __syncthreads();
    if (t == 0) { uint32_t run = 0; for (uint32_t k = 0; k < 1024; k++) { const uint32_t v = part[k]; part[k] = run; run += v; } ctl[CTL_BINS + SCHED_BINS] = run; }
    __syncthreads();
    uint32_t run = part[t];
    for (uint32_t k = 0; k < PER; k++) { const uint32_t c = ctl[CTL_BINS + t * PER + k]; ctl[CTL_BINS + t * PER + k] = run; ctl[CTL_CURSOR + t * PER + k] = run; if ((t * PER + k) % SCHED_R == 0) ctl[CTL_OCUR + (t * PER + k) / SCHED_R] = run; run += c; }
#endif
}

// One scheduled node: lower it (count pass).  The lane goes on with the parent when the schedule chained it to this node.
template <int C> __device__ __forceinline__ void count_one(const SemView& S, const ParArgs& A, const FlatArgs& X, uint32_t entry) {
    Fn f; uint32_t i;
    if (!make_fn(f, S, A, X, entry & ~(uint32_t)SCHED_STAGE1, &i, false)) return;
    if (C == KC_CONTROL || C == KC_ANY) {
        if (entry & SCHED_STAGE1) {
            // stage 1 of a `for`, scheduled behind the iteree: the loop variable gets its type (Q13), the body may go on
            const Desc vt = for_variable_type(f, i);
            if (failed(f)) return;
            f.rt[i] = vt;
            f.st[i] |= ST_STAGE1;
            return;
        }
    }
    for (uint32_t guard = 0; guard <= f.nn; guard++) {
        const uint32_t s = f.st[i];
        if (s & (ST_DONE | ST_DEAD)) { inconsistent(f); return; }         // (a node is listed once)
        if (!visit<false, C>(f, i)) { fallback(f); return; }              // (an input that is not ready: the schedule never lets this happen)
        if (failed(f)) return;
        f.st[i] |= ST_DONE;
        const uint32_t p = f.pa[i];
        if (p == NONE16 || !(f.st[p] & ST_CHAINED)) return;
        i = p;
    }
}
// One launch per (class, round) of the schedule: entries [begin, begin + n) of `sched`.
#ifndef OPUSFE_FLAT_WAVES
#define OPUSFE_FLAT_WAVES 0
#endif
#if OPUSFE_FLAT_WAVES
#define FLAT_ATTR __attribute__((amdgpu_waves_per_eu(OPUSFE_FLAT_WAVES)))
#else
#define FLAT_ATTR
#endif
template <int C> __global__ FLAT_ATTR __launch_bounds__(256) void k_flat_count(SemView S, ParArgs A, FlatArgs X, uint32_t begin, uint32_t n) {
    const uint32_t* in = X.sched + begin;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) count_one<C>(S, A, X, in[k]);
}
// One round of the schedule, every class: the round's eight bins seen as ONE list, so that a lane lowers one node of the round whatever
// its class (bin after bin, every lane paid the latency of eight visits one after the other: 130-180 us per round however few its nodes).
__device__ __forceinline__ void count_round_nodes(const SemView& S, const ParArgs& A, const FlatArgs& X, uint32_t round, uint64_t first, uint64_t stride) {
    uint32_t begin[KC_ACTIVE], upto[KC_ACTIVE + 1];
    upto[0] = 0;
#pragma unroll
    for (uint32_t c = 0; c < KC_ACTIVE; c++) {
        begin[c] = X.ctl[CTL_BINS + c * SCHED_R + round];
        upto[c + 1] = upto[c] + (X.ctl[CTL_BINS + c * SCHED_R + round + 1] - begin[c]);
    }
    for (uint64_t j = first; j < upto[KC_ACTIVE]; j += stride) {
        uint32_t at = 0;
        bool found = false;
#pragma unroll
        for (uint32_t c = 0; c < KC_ACTIVE; c++) if (!found && j < upto[c + 1]) { at = begin[c] + ((uint32_t)j - upto[c]); found = true; }
        count_one<KC_ANY>(S, A, X, X.sched[at]);
    }
}
// The tail of the schedule — rounds [r0, r1) with few nodes each — in ONE launch of ONE workgroup: a barrier between two rounds instead
// of a kernel boundary (what a round writes leaves through this CU's L1; the workgroup-scope barrier orders it for the CU's other waves).
__global__ __launch_bounds__(512) void k_flat_count_tail(SemView S, ParArgs A, FlatArgs X, uint32_t r0, uint32_t r1) {
    for (uint32_t round = r0; round < r1; round++) {
        count_round_nodes(S, A, X, round, threadIdx.x, blockDim.x);
        PAR_SYNC();
    }
}
// a whole round with few nodes in one launch
__global__ __launch_bounds__(256) void k_flat_count_round(SemView S, ParArgs A, FlatArgs X, uint32_t round) {
    count_round_nodes(S, A, X, round, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
}

// ---- scan: one wave per function ------------------------------------------------------------------------------------------
// Exclusive scan of the event counters in Euler order, call-argument slots in node order; the function's totals go to `tot`.
// A device-wide scan over the functions (definition order) then gives every function its place in the dense result arrays
// (k_flat_bases): the layout is dense, in definition order, and the same from run to run.
struct Cnt2x { uint64_t vt, ib; };      // v | t << 32, i | b << 32: the four counters of an event with 32 bits each
struct OpAddCnt2x { __host__ __device__ Cnt2x operator()(Cnt2x a, Cnt2x b) const { return Cnt2x{a.vt + b.vt, a.ib + b.ib}; } };
__global__ __launch_bounds__(256) void k_flat_scan(SemView S, ParArgs A, FlatArgs X, IrCnt* __restrict__ tot) {
#ifdef OPUSFE_EMU
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; const uint32_t lane = 0; constexpr uint32_t W = 1;
#else
    const uint32_t d = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; const uint32_t lane = threadIdx.x & 63u; constexpr uint32_t W = 64;
#endif
    if (d >= A.n_defs) return;
    const uint32_t dn = A.def_nodes[d];
    if (S.nodes[dn].kind != D_Function) return;
    const uint32_t fidx = A.def_excl[d].funcs;
    const FnInfo I = gld(X.fninfo + fidx);
    if (I.nn == 0 || I.dn != dn) return;                     // not on the flat path: stays DS_FALLBACK (k_par_init)
    uint32_t why = gld(X.fb + fidx);
    if (why == 0 && !(gld(&X.rec[dn].st) & ST_DONE)) why = __LINE__;      // (every scheduled node of a function that was not declined is lowered)
    const uint32_t nn = I.nn, ne = 2u * nn;
    uint64_t* cnt = X.cnt + 2ull * dn;
    uint16_t* cg = X.cg + dn;
    // 64 consecutive events per step, one per lane (whole lines in, whole lines out), scanned across the wave; the carry goes on
    Cnt2x carry{0, 0}; uint32_t carry_cg = 0;
    if (why == 0) {
        for (uint32_t base = 0; base < ne; base += W) {
            const uint32_t p = base + lane;
            const Cnt4 c = p < ne ? unpack4(cnt[p]) : Cnt4{0, 0, 0, 0};
            const Cnt2x mine{(uint64_t)c.v | ((uint64_t)c.t << 32), (uint64_t)c.i | ((uint64_t)c.b << 32)};
#ifdef OPUSFE_EMU
            const Cnt2x excl = carry; carry = OpAddCnt2x()(carry, mine);
#else
            const Cnt2x incl = wave_inclusive_scan(mine, OpAddCnt2x());
            const Cnt2x excl{carry.vt + incl.vt - mine.vt, carry.ib + incl.ib - mine.ib};
            carry.vt += (uint64_t)__shfl((long long)incl.vt, 63); carry.ib += (uint64_t)__shfl((long long)incl.ib, 63);
#endif
            // (a prefix beyond 16 bits is cut here and caught by the totals below: the function is declined, nothing reads these words)
            if (p < ne) cnt[p] = pack4(Cnt4{(uint32_t)excl.vt & 0xFFFFu, (uint32_t)(excl.vt >> 32) & 0xFFFFu, (uint32_t)excl.ib & 0xFFFFu, (uint32_t)(excl.ib >> 32) & 0xFFFFu});
        }
        for (uint32_t base = 0; base < nn; base += W) {
            const uint32_t p = base + lane;
            const uint32_t mine = p < nn ? (uint32_t)cg[p] : 0u;
#ifdef OPUSFE_EMU
            const uint32_t excl = carry_cg; carry_cg += mine;
#else
            const uint32_t incl = wave_inclusive_scan(mine, OpAddU32());
            const uint32_t excl = carry_cg + incl - mine;
            carry_cg += (uint32_t)__shfl((int)incl, 63);
#endif
            if (p < nn) cg[p] = (uint16_t)excl;
        }
    }
    const Cnt2x total = carry; const uint32_t total_cg = carry_cg;
    const uint32_t tv = (uint32_t)total.vt, tt = (uint32_t)(total.vt >> 32), ti = (uint32_t)total.ib, tb = (uint32_t)(total.ib >> 32);
    // the packed 16-bit fields of an event must hold every prefix
    if (why == 0 && (tv > 0xFFF0u || tt > 0xFFF0u || ti > 0xFFF0u || tb > 0xFFF0u || total_cg > 0xFFF0u)) why = __LINE__;
    if (lane != 0) return;
    FuncIr r{};
    r.def_node = dn; r.func = fidx;
    if (why) { r.status = DS_FALLBACK; r.pad = why; X.fb[fidx] = why; A.out[fidx] = r; return; }
    r.n_vars = tv; r.n_instrs = ti; r.n_bbs = tb; r.n_types = tt; r.n_cargs = total_cg; r.n_errs = 0; r.status = DS_OK;
    A.out[fidx] = r;                                           // (its bases: k_flat_bases)
    tot[fidx] = IrCnt{tv, ti, tb, tt, total_cg, 0};
}
// The functions' places in the dense arrays: the flat path takes ONE range of every array (its totals, with one atomic per array on the
// bump counters that the functions of the sequential machine use too, k_fb_bases — they may run beside this on the second stream),
// and every function its place inside that range from the exclusive scan of the functions' totals.
__global__ void k_flat_reserve(ParArgs A, const IrCnt* __restrict__ total, IrCnt* __restrict__ origin) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const IrCnt t = *total;
    const Bases b = bump_allocate(A, t.vars, t.instrs, t.bbs, t.types, t.cargs);      // (raises `overflow` when the range does not fit)
    *origin = IrCnt{b.var_base, b.instr_base, b.bb_base, b.type_base, b.carg_base, 0};
}
__global__ void k_flat_bases(ParArgs A, FlatArgs X, const IrCnt* __restrict__ excl, const IrCnt* __restrict__ origin, uint32_t n_funcs) {
    const uint32_t fidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (fidx >= n_funcs) return;
    if (gld(X.fninfo + fidx).nn == 0 || gld(X.fb + fidx) != 0) return;
    FuncIr r = A.out[fidx];
    if (r.status != DS_OK) return;
    const IrCnt e = excl[fidx], o = *origin;
    const uint64_t vb = (uint64_t)o.vars + e.vars, ib = (uint64_t)o.instrs + e.instrs, bb = (uint64_t)o.bbs + e.bbs, tb = (uint64_t)o.types + e.types, cb = (uint64_t)o.cargs + e.cargs;
    r.var_base = (uint32_t)vb; r.instr_base = (uint32_t)ib; r.bb_base = (uint32_t)bb; r.type_base = (uint32_t)tb; r.carg_base = (uint32_t)cb;
    A.out[fidx] = r;
    const bool fits = ib + r.n_instrs <= A.cap_instrs && bb + r.n_bbs <= A.cap_bbs && vb + r.n_vars <= A.cap_vars && tb + r.n_types <= A.cap_types && cb + r.n_cargs <= A.cap_cargs;
    if (!fits) X.fb[fidx] = FB_NOFIT;                        // nothing is written beyond an array; the host runs again with arrays of the exact size
}

// ---- result variables: every node learns the id of the variable that holds its result -----------------------------------
// rvi[] after the count pass says who created it: the node's own event (event, offset), a local or an argument, or "the
// result of node c" (references, parentheses, casts that did nothing).  One lane per node follows that to an id.
__device__ __forceinline__ void resolve_rv(const ParArgs& A, const FlatArgs& X, const FnInfo& I, uint32_t g) {
    const uint32_t dn = I.dn;
    uint32_t r = gld(&X.rec[g].rvi);
    uint32_t res = OPUSFE_NONE;
    for (uint32_t guard = 0; guard <= I.nn + 2u; guard++) {
        const uint32_t tag = r & RV_TAG;
        if (tag == RV_RESOLVED) { res = r; break; }
        if (tag == RV_OWN) { res = RV_RESOLVED | (unpack4(gld(X.cnt + 2ull * dn + ((r >> 12) & 0x1FFFFu))).v + (r & 0xFFFu)); break; }
        if (tag == RV_ARG) { res = RV_RESOLVED | (I.nargs + (r & RV_VAL)); break; }
        if (tag == RV_LOCAL) { r = gld(&X.rec[dn + (r & 0xFFFFu)].rvi); if ((r & RV_TAG) != RV_OWN && (r & RV_TAG) != RV_RESOLVED) break; continue; }      // the slot of the definition holds its variable
        if (tag == RV_PASS) { r = gld(&X.rec[dn + (r & 0xFFFFu)].rvi); continue; }
        break;
    }
    if (res == OPUSFE_NONE) { atomicMax(A.status, (uint32_t)DS_INTERNAL_CAPACITY); if (A.dbg) atomicMax(A.dbg, (uint32_t)__LINE__); return; }
    gst(&X.rec[g].rvi, res);
}
// one lane per node of the unit (the records of nodes that are not on the flat path say fn = NONE)
__global__ __launch_bounds__(256) void k_flat_rv(ParArgs A, FlatArgs X) {
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < A.n_nodes; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t fidx = gld(&X.rec[g].fn);
        if (fidx == OPUSFE_NONE || gld(X.fb + fidx) != 0) continue;
        resolve_rv(A, X, gld(X.fninfo + fidx), (uint32_t)g);
    }
}

// ---- emit -----------------------------------------------------------------------------------------------------------------
template <int C> __device__ __forceinline__ void emit_one(const SemView& S, const ParArgs& A, const FlatArgs& X, uint32_t entry) {
    Fn f; uint32_t i;
    if (!make_fn(f, S, A, X, entry, &i, true)) return;
    const FuncIr r = gld(A.out + f.fidx);
    f.var_base = r.var_base; f.instr_base = r.instr_base; f.bb_base = r.bb_base; f.type_base = r.type_base; f.carg_base = r.carg_base;
    f.n_vars_tot = r.n_vars; f.n_instrs_tot = r.n_instrs; f.n_bbs_tot = r.n_bbs; f.n_types_tot = r.n_types; f.n_cargs_tot = r.n_cargs;
    visit<true, C>(f, i);
}
// entries [lo, lo + step), [lo + stride, ...) ... of class C's part of `order`
template <int C> __device__ __forceinline__ void emit_class(const SemView& S, const ParArgs& A, const FlatArgs& X, uint64_t first, uint64_t stride) {
    const uint32_t begin = X.ctl[CTL_BINS + (uint32_t)C * SCHED_R], n = X.ctl[CTL_OCUR + (uint32_t)C] - begin;      // every node of the class, the chained ones too
    const uint32_t* in = X.order + begin;
    for (uint64_t k = first; k < n; k += stride) emit_one<C>(S, A, X, in[k]);
}
template <int C> __global__ FLAT_ATTR __launch_bounds__(256) void k_flat_emit(SemView S, ParArgs A, FlatArgs X) {
    emit_class<C>(S, A, X, (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
}
// (All classes in ONE launch, the blocks dealt to the classes in proportion to their node counts so that they walk the node-ordered lists at
// the same relative pace and share what they fetch, was measured: 35.3 ms against 26.0 for the eight launches — the kernel then has the
// registers and the code of its largest class.  One launch over ALL nodes in node order, one lane per node with the whole visitor —
// neighbouring lanes on neighbouring records, no list — was measured too: 63 ms; the class-sorted lists as one list under one
// visitor for all kinds (what the count pass does per round): 27.7 ms.  The pass is bound by divergence, not by what it fetches.)

// ======================================================================================================================
// k_flat_pre: what is local to a function by nature (one workgroup per function, one lane per node, state in LDS)
// ======================================================================================================================
struct Pre {
    SemView S; ParArgs A;
    uint32_t dn, nn, fidx, nargs; uint64_t user_syms;
    lds_u32* kn; lds_u32* st; lds_u32* ref; lds_u32* rvi; lds_u32* pay;
    lds_u16* fc; lds_u16* nx; lds_u16* pa; lds_u16* rk; lds_u16* sz; lds_u16* pe; lds_u16* cg; lds_u16* wnext; lds_u16* s1;
    DefEnt* defs; ArgEnt* args; lds_u32* misc; lds_u32* whead;
    uint32_t tid, nthreads;
};
enum { MI_FB = 0, MI_NDEFS = 1, MI_WHY = 7, MI_KHIST = 16 /* 2 x KC_ACTIVE words */, MI_KBASE = 32 /* KC_ACTIVE words: the function's place in the class list */, MI_QCNT = 40 /* length of the schedule's queue */,
       MI_BUCKET = 48 /* 64 words: heads of the definitions' hash chains (index + 1, 0 = empty; DefEnt.pad links on) */, MI_WORDS = 112 };

__host__ __device__ inline size_t lds_bytes(uint32_t nmax, uint32_t maxdef, uint32_t nthreads) {
    (void)nthreads;
    size_t b = 0;
    b += 4ull * 5 * nmax;        // kn st ref rvi pay
    b += 2ull * 9 * nmax;        // fc nx pa rk sz pe cg wnext s1
    b = (b + 7) & ~(size_t)7;
    b += sizeof(DefEnt) * maxdef + sizeof(ArgEnt) * MAX_ARGS + 4 * MI_WORDS + 4ull * maxdef;
    return (b + 15) & ~(size_t)15;
}
__device__ __forceinline__ void fallback_at(Pre& f, uint32_t line) { if (!f.misc[MI_FB]) f.misc[MI_WHY] = line; f.misc[MI_FB] = 1; }
__device__ __forceinline__ bool failed(const Pre& f) { return f.misc[MI_FB] != 0; }
__device__ __forceinline__ opusfe_node ND(const Pre& f, uint32_t i) { return f.S.nodes[f.dn + i]; }
__device__ __forceinline__ uint32_t def_bucket(uint32_t hash) { return (hash * 0x9E3779B1u) >> 26; }
// definition of identifier token `t` visible at node `use`: RF_LOCAL | index, RF_ARG | index, or RF_NONE
__device__ __forceinline__ uint32_t find_local(const Pre& f, const opusfe_token& t, uint32_t use_node) {
    // (the definitions hang in 64 hash chains: a name looks at the few that share its bucket, not at every definition of the function;
    // at most one definition of a name is in scope at a use — shadowing is declined — so the order inside a chain does not matter)
    const uint32_t nd = f.misc[MI_NDEFS], use = RANK(f, use_node);
    for (uint32_t k1 = f.misc[MI_BUCKET + def_bucket(t.payload)], guard = 0; k1 != 0 && guard <= nd; guard++) {
        const DefEnt& d = f.defs[k1 - 1];
        if (d.hash == t.payload && d.len == t.len && d.dpoint <= use && use < d.send && bytes_eq(f.S, d.low, t.low, t.len)) return RF_LOCAL | (k1 - 1);
        k1 = d.pad;
    }
    for (uint32_t k = 0; k < f.nargs; k++) {
        const ArgEnt& a = f.args[k];
        if (a.hash == t.payload && a.len == t.len && bytes_eq(f.S, a.low, t.low, t.len)) return RF_ARG | k;
    }
    return RF_NONE;
}
// the node whose rt / rvi words hold the type and the variable of the local definition made by node i: the statement itself
// (`var x = e`, `for x in e`: statements have no result of their own), the type node of a `var x T` expression
__device__ __forceinline__ uint32_t def_slot(const Pre& f, uint32_t i) { return KIND(f, i) == E_VariableDefinition ? (uint32_t)f.fc[i] : i; }

__device__ __forceinline__ void pre_function(Pre& f, const FlatArgs& X) {
    const uint32_t nn = f.nn, tid = f.tid, T = f.nthreads, dn = f.dn;
    for (uint32_t k = tid; k < MI_WORDS; k += T) f.misc[k] = 0;
    PAR_SYNC();
    // ---- structure ----
    for (uint32_t i = tid; i < nn; i += T) {
        const opusfe_node n = ND(f, i);
        f.kn[i] = (uint32_t)n.kind | ((uint32_t)n.sub << 8) | ((uint32_t)n.nchild << 16);
        const uint32_t fc = n.first_child == OPUSFE_NONE ? NONE16 : n.first_child - f.dn, nx = n.next == OPUSFE_NONE ? NONE16 : n.next - f.dn;
        if ((fc != NONE16 && fc >= nn) || (nx != NONE16 && nx >= nn)) fallback(f);      // links must stay inside the function
        f.fc[i] = (uint16_t)fc; f.nx[i] = (uint16_t)nx;
        f.pay[i] = n.payload;
        f.pa[i] = NONE16; f.st[i] = 0; f.ref[i] = (n.kind >= Y_Name && n.kind <= Y_Array) ? f.S.tid_of_node[f.dn + i] : 0u; f.rvi[i] = 0; f.cg[i] = 0; f.rk[i] = 0; f.sz[i] = 0; f.pe[i] = 0; f.wnext[i] = NONE16;
        // (the event words that start empty — cnt, tail — were cleared by the host, the records filled with 0xFF: result type "none")
        if (f.A.undefined_flag[f.dn + i]) fallback(f);                                // UndefinedType somewhere in the function
    }
    if (par_sync_or(failed(f))) return;      // (uniform exit: every wave sees the same answer)
    for (uint32_t i = tid; i < nn; i += T) { uint32_t guard = 0; for (uint32_t c = f.fc[i]; c != NONE16 && guard <= nn; c = f.nx[c], guard++) f.pa[c] = (uint16_t)i; if (guard > nn) fallback(f); }
    if (par_sync_or(failed(f))) return;
#ifdef OPUSFE_DIAG
    if (f.A.stop_phase == 1) { fallback(f); return; }
#endif
    // Subtree sizes, depths and "dead" marks by one walk up the ancestor chain per node (the nodes arrive in creation order, not
    // pre-order, so none of this is index arithmetic).  Level-synchronous fixpoints did the same in height-of-the-tree rounds of
    // barrier + sweep each (3.7 + 5.8 ms per 256 MB); a walk is depth steps for the one node, with no barrier at all.
    // rvi[] (zero since the load) takes the u32 accumulators; pe[] holds the depth until the ranks are known.
    for (uint32_t i = tid; i < nn; i += T) {
        uint32_t a = f.pa[i], child = i, depth = 0;
        bool dead = false;
        while (a != NONE16 && depth <= nn) {
            lds_add(&f.rvi[a], 1u);
            if (KIND(f, a) == E_Operator && SUB(f, a) == OP_Not && f.fc[a] == child) dead = true;     // Q9: the left operand of `not` is never evaluated
            child = a; a = f.pa[a]; depth++;
        }
        if (depth > nn || child != 0) { fallback(f); continue; }                      // every node must hang under the root (no cycles, no strays)
        f.pe[i] = (uint16_t)depth;
        if (dead) f.st[i] = ST_DEAD;
    }
    if (par_sync_or(failed(f))) return;
#ifdef OPUSFE_DIAG
    if (f.A.stop_phase == 2) { fallback(f); return; }
#endif
    // every parent hands its children their offset from its own rank (1 + the sizes of the children before them)
    for (uint32_t p = tid; p < nn; p += T) {
        f.sz[p] = (uint16_t)(1u + f.rvi[p]);
    }
    PAR_SYNC();
    for (uint32_t p = tid; p < nn; p += T) {
        uint32_t off = 1;
        for (uint32_t c = f.fc[p]; c != NONE16; c = f.nx[c]) { f.rk[c] = (uint16_t)off; off += f.sz[c]; }
        if (p == 0) f.rk[0] = 0;
        f.rvi[p] = 0;
    }
    PAR_SYNC();
    // pre-order rank = the sum of those offsets up the chain
    for (uint32_t i = tid; i < nn; i += T) {
        uint32_t r = 0;
        for (uint32_t a = i, guard = 0; a != NONE16 && guard <= nn; a = f.pa[a], guard++) r += f.rk[a];
        f.rvi[i] = r;
    }
    PAR_SYNC();
    for (uint32_t i = tid; i < nn; i += T) { const uint32_t r = f.rvi[i]; f.rvi[i] = 0; f.rk[i] = (uint16_t)r; f.pe[i] = (uint16_t)(2u * r - f.pe[i]); }
    if (tid == 0) f.st[0] |= ST_CTX;
    PAR_SYNC();
#ifdef OPUSFE_DIAG
    if (f.A.stop_phase == 3) { fallback(f); return; }
#endif
    // Evaluation context (mode, pointer kind, expected type) of every node: a function of its parent's kind and of its place among
    // the children, except under `-e`, `(e)` and `e.f`, which pass (part of) their own context on: those wait until their parent has it (one more
    // round per nesting level of such nodes).
    for (uint32_t round = 0; round <= nn; round++) {
        bool progress = false;
        for (uint32_t i = tid; i < nn; i += T) {
            const uint32_t own = f.st[i];
            if (own & ST_CTX) continue;
            const uint32_t p = f.pa[i];
            const uint8_t pk = KIND(f, p);
            const bool passes = pk == E_Negate || pk == E_Parentheses || pk == E_Field;
            const uint32_t ps = passes ? f.st[p] : 0u;
            if (passes && !(ps & ST_CTX)) continue;
            progress = true;
            const uint32_t pmode = ps & ST_MODE_MASK, ppt = (ps >> ST_PT_SHIFT) & 3u, pexp = (ps & ST_EXP_MASK) >> ST_EXP_SHIFT;
            const bool first = f.fc[p] == i;
            const uint8_t k = KIND(f, i);
            uint32_t m = 0;      // 0: not an evaluated expression
            auto E = [&](uint32_t exp) { return (uint32_t)ST_MODE_E | (exp << ST_EXP_SHIFT); };
            auto P = [&](uint32_t pt_, uint32_t flags) { return (uint32_t)ST_MODE_P | (pt_ << ST_PT_SHIFT) | flags; };
            const bool is_expr = k >= E_Integer && k <= E_Parentheses;
            const bool dead = (own & ST_DEAD) != 0;
            if (is_expr && !dead) {
                switch (pk) {
                    case S_Expression: case S_VariableDefinition: m = E(EXP_NONE); break;
                    case S_Assignment: m = first ? P(PT_Mutable, ST_MUSTLV) : E(EXP_ASSIGN); break;
                    case S_Return: m = E(EXP_RET); break;
                    case S_If: case S_While: m = E(EXP_BOOL); break;
                    case S_For: m = E(EXP_NONE); break;
                    case E_Operator: { const uint8_t op = SUB(f, p); m = op == OP_Is ? E(EXP_NONE) : (op == OP_And || op == OP_Or) ? E(EXP_BOOL) : E(EXP_GENINT); break; }
                    case E_Cast: case E_Dereference: m = E(EXP_NONE); break;
                    case E_Reference: m = P(PT_Reference, ST_MUSTLV); break;
                    case E_MutableReference: m = P(PT_Mutable, ST_MUSTLV); break;
                    case E_ArrayReference: m = P(PT_Array, ST_MUSTLV); break;
                    case E_MutableArrayReference: m = P(PT_ArrayMutable, ST_MUSTLV); break;
                    case E_Negate: m = E(pmode == ST_MODE_E ? pexp : (uint32_t)EXP_NONE); break;
                    case E_Parentheses: m = pmode == ST_MODE_E ? E(pexp) : P(ppt, ps & (ST_DEREF | ST_MUSTLV)); break;
                    case E_Field: m = P(ppt, ST_MUSTLV); break;                         // (a field node is always in pointer mode, see below)
                    case E_NamedCall: m = E(EXP_ARG); break;                            // <integer> for a call, size for an index: known once the name is resolved
                    case X_FieldInit: m = E(EXP_FIELD); break;
                    default: fallback(f); break;                                         // E_CallOrIndex
                }
                // expressions that are lvalue-shaped are lowered through the pointer path and loaded (generate_ir.rs:577-585);
                // everything else asked for as a pointer is an rvalue (:1082-1087)
                const bool pkind = k == E_Variable || k == E_Dereference || k == E_VariableDefinition || k == E_Field || k == E_NamedCall || k == E_CallOrIndex || k == E_Record;
                if ((m & ST_MODE_MASK) == ST_MODE_E && pkind) m = P(PT_Reference, ST_DEREF);
                else if ((m & ST_MODE_MASK) == ST_MODE_P && !pkind && k != E_Parentheses) {
                    if (m & ST_MUSTLV) fallback(f);                                      // InvalidLvalue
                    m = E(EXP_NONE);
                }
            }
            par_fence();
            f.st[i] = (own & ST_DEAD) | m | ST_CTX;
        }
        if (!par_sync_or(progress)) break;
    }
    if (failed(f)) return;
#ifdef OPUSFE_DIAG
    if (f.A.stop_phase == 4) { fallback(f); return; }
#endif
    // ---- arguments and local definitions ----
    if (tid == 0) {
        uint32_t a = f.fc[0], k = 0;
        while (a != NONE16 && KIND(f, a) == X_NamedType) {
            if (k >= MAX_ARGS) { fallback(f); break; }
            const opusfe_token t = f.S.tok[f.pay[a]];
            if (t.kind != T_Identifier) { fallback(f); break; }                         // a symbol as a variable name (sym_locals)
            ArgEnt e; e.hash = t.payload; e.low = t.low; e.len = t.len; e.pad = a; e.vt = d_ptr(f, PT_Mutable, d_id(TID(f, f.fc[a])));
            f.args[k++] = e;
            gst(&X.rec[dn + a].rt, (uint64_t)e.vt);                                        // (the argument's node holds its type for the visits)
            a = f.nx[a];
        }
        if (k != f.nargs) fallback(f);
    }
    for (uint32_t i = tid; i < nn; i += T) {
        const uint8_t k = KIND(f, i);
        if (!(f.st[i] & ST_CTX)) { fallback(f); continue; }
        if (k == S_VariableDefinition || k == E_VariableDefinition || k == S_For) {
            if (f.st[i] & ST_DEAD) { fallback(f); continue; }                           // a definition in code that is never evaluated
            const uint32_t slot = lds_add(&f.misc[MI_NDEFS], 1u);
            if (slot >= f.A.maxdef) { fallback(f); continue; }
            const opusfe_token t = f.S.tok[f.pay[i]];
            if (t.kind != T_Identifier) { fallback(f); continue; }
            DefEnt d; d.hash = t.payload; d.low = t.low; d.len = t.len; d.node = i; d.var = 0; d.pad = 0; d.vt = DESC_NONE;
            if (k == S_For) { uint32_t cond, tb, eb; blocks_of(f, i, &cond, &tb, &eb); if (tb == NONE16) { fallback(f); continue; } d.dpoint = RANK(f, tb); d.send = RANK_END(f, tb); }
            else {
                // visible from behind the statement (`var x = e`: e is lowered first) / from the expression on (`var x T`), to the
                // end of the enclosing block (generate_ir.rs:176-203)
                uint32_t blk = f.pa[i];
                while (blk != NONE16 && KIND(f, blk) != X_Block) blk = f.pa[blk];
                if (blk == NONE16) { fallback(f); continue; }
                d.dpoint = k == S_VariableDefinition ? RANK_END(f, i) : RANK(f, i) + 1;
                d.send = RANK_END(f, blk);
                if (k == E_VariableDefinition) { if (f.fc[i] == NONE16) { fallback(f); continue; } d.vt = d_ptr(f, PT_Mutable, d_id(TID(f, f.fc[i]))); }
            }
            {   // into its hash chain
                lds_u32* const head = &f.misc[MI_BUCKET + def_bucket(d.hash)];
#ifdef OPUSFE_EMU
                d.pad = *head; *head = slot + 1;
#else
                d.pad = __hip_atomic_exchange(head, slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
            }
            f.defs[slot] = d;
            f.whead[slot] = NONE16;
            f.ref[i] = slot;
        }
        if (k == S_Break || k == S_Continue) {
            const uint32_t loop = enclosing_loop(f, i);
            f.ref[i] = loop;
            if (loop != NONE16 && k == S_Break) lds_or(&f.st[loop], (uint32_t)ST_HASBREAK);
        }
    }
    if (par_sync_or(failed(f))) return;
    {   // shadowing (generate_ir.rs:255,410,1009) and redefinition of an argument (Q17): the sequential machine handles those
        const uint32_t nd = f.misc[MI_NDEFS];
        for (uint32_t a = tid; a < nd; a += T) {
            const DefEnt da = f.defs[a];
            for (uint32_t b = 0; b < nd; b++) {
                if (b == a) continue;
                const DefEnt& db = f.defs[b];
                if (db.hash != da.hash || db.len != da.len || !bytes_eq(f.S, da.low, db.low, da.len)) continue;
                // is a in scope when b's shadow check runs?  (a `var` statement checks after its initialiser, a `for` after its
                // iteree, a `var x T` expression when it is reached)
                const uint8_t bk = KIND(f, db.node);
                const uint32_t at = (bk == S_VariableDefinition || bk == S_For) ? db.dpoint - 1 : RANK(f, db.node);
                if (at >= da.dpoint && at < da.send) fallback(f);
            }
            for (uint32_t k = 0; k < f.nargs; k++) { const ArgEnt& g = f.args[k]; if (g.hash == da.hash && g.len == da.len && bytes_eq(f.S, g.low, da.low, da.len)) fallback(f); }
        }
        for (uint32_t a = tid; a < f.nargs; a += T) for (uint32_t b = 0; b < a; b++) { const ArgEnt& x = f.args[a]; const ArgEnt& y = f.args[b]; if (x.hash == y.hash && x.len == y.len && bytes_eq(f.S, x.low, y.low, x.len)) fallback(f); }
    }
    PAR_SYNC();
#ifdef OPUSFE_DIAG
    if (f.A.stop_phase == 5) { fallback(f); return; }
#endif
    // ---- names ----
    // A use of a local whose type is not known yet (`var x = e`, `for x in e`) WAITS for the definition: one more pending operand,
    // and a place in the definition's waiter chain (whead / wnext), which the visit of the definition walks.
    for (uint32_t i = tid; i < nn; i += T) {
        const uint8_t k = KIND(f, i);
        if (f.st[i] & ST_DEAD) continue;
        if (k == E_Integer || k == E_Float) { f.ref[i] = f.S.tok[f.pay[i]].aux; continue; }     // the literal's suffix decides its type
        if (k != E_Variable && k != E_NamedCall) continue;
        const uint32_t name_tok = f.pay[i];
        const opusfe_token t = f.S.tok[name_tok];
        if (t.kind != T_Identifier) { fallback(f); continue; }
        const uint32_t local = find_local(f, t, i);
        auto use_local = [&](uint32_t l) {
            // the visits address a local by the node that holds its type and variable (def_slot), an argument by its number and node
            if ((l & RF_TAG) == RF_ARG) { f.ref[i] = RF_ARG | (l & RF_VAL) | (f.args[l & RF_VAL].pad << 4); return; }
            const uint32_t di = l & RF_VAL;
            const DefEnt& d = f.defs[di];
            f.ref[i] = RF_LOCAL | def_slot(f, d.node);
            if (d.vt == DESC_NONE && (f.st[i] & ST_MODE_MASK) != 0) {
#ifdef OPUSFE_EMU
                const uint32_t oldh = f.whead[di]; f.whead[di] = i;
#else
                const uint32_t oldh = __hip_atomic_exchange(&f.whead[di], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
                f.wnext[i] = (uint16_t)oldh;
                lds_add(&f.st[i], (uint32_t)ST_PEND_ONE);
            }
        };
        if (k == E_Variable && local != RF_NONE) { use_local(local); continue; }          // (most names: no table lookup at all)
        const uint32_t slot = name_find(f.S, key_tok(name_tok));
        if (k == E_NamedCall && slot_has_functions(f.S, slot)) {
            if (local != RF_NONE) { fallback(f); continue; }      // a local shadows the functions of that name (:1152-1158)
            f.ref[i] = RF_CALL | (slot & RF_VAL);
            f.cg[i] = (uint16_t)NCH(f, i);
            // One function of that name (the common case): everything the visit needs from the tables is fetched HERE, where all
            // call nodes do it at once, instead of inside the type fixpoint: the return type goes to rt[i], the parameter types to
            // the EXIT slots of the arguments (this node owns them; they are read before it counts into the last one).
            const NameSlot sl = f.S.slots[slot];
            uint32_t nfn = 0, only = 0;
            for (uint32_t j = 0; j < sl.count; j++) { const NameEntry e = f.S.entries[sl.offset + j]; if ((e.kind_index >> 30) == NK_FUNCTION) { nfn++; only = e.kind_index & 0x3FFFFFFFu; } }
            if (nfn == 1 && NCH(f, i) <= MAX_CALL_ARGS) {
                const FuncRec fr = f.S.funcs[only];
                if (fr.nargs != NCH(f, i)) { fallback(f); continue; }                      // NoOverloadForFunction
                uint32_t c = f.fc[i], a = fr.first_arg;
                for (uint32_t q = 0; q < fr.nargs && c != NONE16; q++) { gst(X.cnt + 2ull * dn + POS_EXIT(f, c), (uint64_t)f.S.tid_of_node[f.S.nodes[a].first_child]); a = f.S.nodes[a].next; c = f.nx[c]; }
                gst(&X.rec[dn + i].rt, (uint64_t)d_id(fr.ret));
                f.ref[i] = RF_CALL1 | only;
            }
            continue;
        }
        if (local != RF_NONE) { use_local(local); continue; }
        uint32_t r = RF_NONE;
        if (slot != OPUSFE_NONE) {
            const NameSlot sl = f.S.slots[slot];
            uint32_t nfn = 0, only = 0;
            for (uint32_t j = sl.count; j > 0; j--) {
                const NameEntry e = f.S.entries[sl.offset + j - 1];
                const uint32_t ek = e.kind_index >> 30;
                if (ek == NK_GLOBAL && r == RF_NONE) { const uint32_t gid = e.kind_index & 0x3FFFFFFFu; r = (f.S.globals[gid].is_const ? RF_CONST : RF_GLOBAL) | gid; }
                if (ek == NK_FUNCTION) { nfn++; only = e.kind_index & 0x3FFFFFFFu; }
            }
            if (r == RF_NONE && nfn == 1) r = RF_FUNC | only;
        }
        if (r == RF_NONE || (k == E_NamedCall && ((r & RF_TAG) == RF_CONST || (r & RF_TAG) == RF_FUNC))) { fallback(f); continue; }
        if ((r & RF_TAG) == RF_GLOBAL || (r & RF_TAG) == RF_CONST) gst(&X.rec[dn + i].rt, (uint64_t)d_id(f.S.globals[r & RF_VAL].type_id));    // fetched here, read by the visit
        f.ref[i] = r;
    }
    // operator calls and `for` reserve their call arguments too (user overloads only use them; sized for the worst case)
    for (uint32_t i = tid; i < nn; i += T) {
        const uint8_t k = KIND(f, i);
        if (k == E_Operator && !(f.st[i] & ST_DEAD)) { const uint8_t sym = sym_of_op(SUB(f, i)); if ((f.user_syms >> sym) & 1) f.cg[i] = 2; }
        if (k == S_For) f.cg[i] = 3;
    }
    if (par_sync_or(failed(f))) return;
#ifdef OPUSFE_DIAG
    if (f.A.stop_phase == 6) { fallback(f); return; }
#endif
    // ---- what every node waits for ----
    for (uint32_t i = tid; i < nn; i += T) {
        const uint32_t s = f.st[i];
        if (kind_class_of(s, KIND(f, i)) >= KC_ACTIVE && !(s & ST_DEAD)) f.st[i] = s | ST_DONE;      // nothing to lower: its parent need not wait
    }
    PAR_SYNC();
    // every node counts the operands it has to wait for (in its own state word): each node that will be lowered adds one to its
    // parent; an integer literal that takes its type from the left side of its assignment waits for that
    for (uint32_t i = tid; i < nn; i += T) {
        if (!(f.st[i] & (ST_DONE | ST_DEAD))) { const uint32_t w = wait_target(f, i); if (w != NONE16) lds_add(&f.st[w], (uint32_t)ST_PEND_ONE); }
        if (literal_waits_for_lhs(f, i)) lds_add(&f.st[i], (uint32_t)ST_PEND_ONE);
    }
    // Which statements diverge (every path through them returns, generate_ir.rs:275-283, 284-462): structural.  Every source — a
    // `return`, a `while true` without `break` — climbs: a block diverges with any of its statements, an `if` when both branches do,
    // a `for` / `while` when its else branch does.  (A climber stops at a node that already diverges: whoever marked it went on.)
    for (uint32_t i = tid; i < nn; i += T) {
        const uint8_t k = KIND(f, i);
        bool source = k == S_Return;
        if (k == S_While) { const uint32_t c = f.fc[i]; source = c != NONE16 && KIND(f, c) == E_Bool && SUB(f, c) && !(f.st[i] & ST_HASBREAK); }
        if (!source) continue;
        uint32_t c = i;
        for (uint32_t guard = 0; guard <= nn; guard++) {
            if (lds_or(&f.st[c], (uint32_t)ST_DIV) & ST_DIV) break;
            const uint32_t p = f.pa[c];
            if (p == NONE16) break;
            const uint8_t pk = KIND(f, p);
            if (pk == X_Block) { c = p; continue; }
            if (pk != S_If && pk != S_While && pk != S_For) break;
            uint32_t cond, tb, eb; blocks_of(f, p, &cond, &tb, &eb);
            bool up = false;
            if (pk == S_If) {
                const uint32_t other = c == tb ? eb : c == eb ? tb : (uint32_t)NONE16;
                par_fence();                                  // (this branch is marked before the other one is looked at: of two climbers that meet here, one sees both)
                up = other != NONE16 && (lds_or(&f.st[other], 0u) & ST_DIV);
            } else up = c == eb && c != NONE16;
            if (!up) break;
            c = p;
        }
    }
    PAR_SYNC();
#ifdef OPUSFE_DIAG
    if (f.A.stop_phase == 7) { fallback(f); return; }
#endif
    // ---- the schedule (see "the schedule of the count pass"): the dependency counters played through ----
    // round(node) = 1 + the latest round of what it waits for (operands, the awaited definition, the left side; a CHAINED parent —
    // waits for this one operand only, same class — takes its operand's round: same lane, not listed).  No barrier per round: who
    // lowers a node hands `its round + 1` to everything that waits for the node (a running maximum, acc[] = pay[], free since the names
    // phase) and counts itself off; the lane whose count-off is the LAST one knows the maximum is final and goes on with that node at
    // once — up the tree, across kind classes.  Only where ONE node releases SEVERAL (the uses of a variable) the others go to a queue
    // (rvi[], free too) that the workgroup drains in phases: a barrier per fan-out generation, not per round.
    {
        lds_u32* const acc = f.pay;
        lds_u16* const q = (lds_u16*)f.rvi;
        const uint32_t qcap = 2u * f.A.nmax;
        for (uint32_t i = tid; i < nn; i += T) { f.rk[i] = NONE16; f.s1[i] = NONE16; acc[i] = 0; }
        if (tid == 0) f.misc[MI_QCNT] = 0;
        PAR_SYNC();
        auto push = [&](uint32_t entry) { const uint32_t at = lds_add(&f.misc[MI_QCNT], 1u); if (at < qcap) q[at] = (uint16_t)entry; else fallback(f); };
        // `node` need not wait for something lowered in round r - 1 any longer; true: that was the last thing it waited for
        auto arrive = [&](uint32_t node, uint32_t r) -> bool { lds_max(&acc[node], r); return (lds_sub(&f.st[node], (uint32_t)ST_PEND_ONE) & ST_PEND_MASK) == ST_PEND_ONE; };
        auto release_waiters = [&](uint32_t def_index, uint32_t r) {
            for (uint32_t w = f.whead[def_index], guard = 0; w != NONE16 && guard <= nn; guard++) { const uint32_t after = f.wnext[w]; if (arrive(w, r)) push(w); w = after; }
        };
        for (uint32_t i = tid; i < nn; i += T) {
            const uint32_t s = f.st[i];
            if (!(s & (ST_DONE | ST_DEAD | ST_PEND_MASK))) push(i);
        }
        uint32_t done = 0;
        for (uint32_t phase = 0; phase <= nn + 1; phase++) {
            PAR_SYNC();
            const uint32_t n = f.misc[MI_QCNT] < qcap ? f.misc[MI_QCNT] : qcap;
            if (n == done) break;
            PAR_SYNC();                                          // (every lane has read the length before the next pushes move it)
            for (uint32_t j = done + tid; j < n; j += T) {
                uint32_t i = q[j];
                uint32_t r = acc[i];
                uint32_t my_class = kind_class_of(f.st[i], KIND(f, i));
                for (uint32_t guard = 0; guard <= nn; guard++) {
                    if (r >= FLAT_MAX_ROUNDS) { fallback(f); break; }      // a dependency chain longer than the schedule's table: the sequential machine
                    f.rk[i] = (uint16_t)r;
                    lds_or(&f.st[i], (uint32_t)ST_DONE);
                    if (KIND(f, i) == S_VariableDefinition) release_waiters(f.ref[i], r + 1);
                    const uint32_t p = wait_target(f, i);
                    if (p == NONE16) break;
                    const uint8_t pk = KIND(f, p);
                    if (pk == S_Assignment && f.fc[p] == i) {            // the left side: an integer literal on the right takes its type from it
                        uint32_t c = f.nx[i];
                        while (c != NONE16 && (KIND(f, c) == E_Parentheses || KIND(f, c) == E_Negate)) c = f.fc[c];
                        if (c != NONE16 && literal_waits_for_lhs(f, c) && arrive(c, r + 1)) push(c);
                    }
                    bool iteree = false;
                    if (pk == S_For) { uint32_t cond, tb, eb; blocks_of(f, p, &cond, &tb, &eb); iteree = cond == i; }
                    bool last;
                    if (iteree) {
                        // stage 1 of the loop is the round behind its iteree — the loop variable gets its type, the body may go on; the
                        // loop itself is lowered behind stage 1 and everything below it
                        const uint32_t r1 = r + 1;
                        if (r1 >= FLAT_MAX_ROUNDS) { fallback(f); break; }
                        f.s1[p] = (uint16_t)r1;
                        release_waiters(f.ref[p], r1 + 1);
                        last = arrive(p, r1 + 1);
                    } else last = arrive(p, r + 1);
                    if (!last) break;                                    // the parent waits for others: the lane that counts off last goes on with it
                    const uint32_t pc = kind_class_of(f.st[p], pk);
                    if (pk != S_For && single_operand(pk) && pc == my_class) lds_or(&f.st[p], (uint32_t)ST_CHAINED);      // same round, same lane
                    else { r = acc[p]; my_class = pc; }
                    i = p;
                }
            }
            done = n;
        }
        PAR_SYNC();
        for (uint32_t i = tid; i < nn; i += T) f.rvi[i] = 0;             // (the queue's words)
    }
    if (par_sync_or(failed(f))) return;
#ifdef OPUSFE_DIAG
    if (f.A.stop_phase == 8) { fallback(f); return; }
#endif
    // every node that is lowered must have its round
    for (uint32_t i = tid; i < nn; i += T) {
        const uint32_t c = kind_class_of(f.st[i], KIND(f, i));
        if (c < KC_ACTIVE && (f.rk[i] == NONE16 || !(f.st[i] & ST_DONE) || (KIND(f, i) == S_For && f.s1[i] == NONE16))) fallback(f);
    }
    if (par_sync_or(failed(f))) return;
    // ---- hand-over (nothing below declines the function: only nodes of functions that are handed over are ever listed) ----
    // the per-node words the flat passes read: bytes 8..31 of the record (its result type was written above), the schedule bins
    for (uint32_t i = tid; i < nn; i += T) {
        const uint32_t g = dn + i;
        uint32_t s = f.st[i];
        const uint32_t c = kind_class_of(s, KIND(f, i));
        const bool listed = c < KC_ACTIVE;
        // (the schedule's counters go; ST_DONE stays with the nodes that have nothing to lower, ST_CHAINED with the chained ones)
        s &= ~(uint32_t)(ST_CLAIM | ST_PEND_MASK);
        if (listed) s &= ~(uint32_t)ST_DONE;
        uint64_t* const w = (uint64_t*)(X.rec + g);
        gst(w + 1, (uint64_t)s | ((uint64_t)(uint32_t)f.ref[i] << 32));
        gst(w + 2, (uint64_t)0 /* rvi */ | ((uint64_t)(uint16_t)f.pe[i] << 32) | ((uint64_t)(uint16_t)f.sz[i] << 48));
        gst(w + 3, (uint64_t)(uint16_t)f.pa[i] | ((uint64_t)i << 16) | ((uint64_t)f.fidx << 32));
        gst(X.cg + g, (uint16_t)f.cg[i]);
        gst((uint16_t*)(X.tail + 2ull * g), (uint16_t)(listed ? c * SCHED_R + ((s & ST_CHAINED) ? (uint32_t)FLAT_MAX_ROUNDS : (uint32_t)f.rk[i]) : (uint32_t)NONE16));
        gst(X.key1 + g, (uint16_t)((listed && KIND(f, i) == S_For) ? (uint32_t)KC_CONTROL * SCHED_R + f.s1[i] : (uint32_t)NONE16));
    }
    {
        const uint32_t nd = f.misc[MI_NDEFS];
        for (uint32_t k = tid; k < nd; k += T) {
            const DefEnt& d = f.defs[k];
            if (d.vt != DESC_NONE) gst(&X.rec[dn + def_slot(f, d.node)].rt, (uint64_t)d.vt);
        }
    }
    if (tid == 0) { FnInfo I{}; I.dn = dn; I.nn = nn; I.nargs = f.nargs; I.ret = f.S.funcs[f.fidx].ret; I.sym_mix = f.S.mix_of_def(dn); gst(X.fninfo + f.fidx, I); }
}

// One workgroup per function of `list` (definition indices).
#ifndef OPUSFE_PAR_WAVES
#define OPUSFE_PAR_WAVES 8      // (53 VGPRs: eight waves per SIMD fit; measured 59 -> 48 ms per GB against four)
#endif
__global__ MACHINE_ATTR(OPUSFE_PAR_WAVES) __launch_bounds__(1024) void k_flat_pre(SemView S, ParArgs A, FlatArgs X) {
    const uint32_t slot = blockIdx.x;
    if (slot >= A.n_list) return;
    const uint32_t d = A.list ? A.list[slot] : slot;
    const uint32_t dn = A.def_nodes[d];
    if (S.nodes[dn].kind != D_Function) return;
    if (A.skip && A.skip[d]) return;
    const uint32_t fidx = A.def_excl[d].funcs;
    const uint32_t dn_end = d + 1 < A.n_defs ? A.def_nodes[d + 1] : A.n_nodes;
    const uint32_t nn = dn_end - dn;
    if (nn > A.nmax || nn < 2) return;                       // stays DS_FALLBACK (k_par_init)
    Pre f;
    f.S = S; f.A = A; f.dn = dn; f.nn = nn; f.fidx = fidx; f.nargs = S.funcs[fidx].nargs; f.user_syms = *A.user_syms;
    f.tid = threadIdx.x; f.nthreads = blockDim.x;
    // carve the dynamic LDS block (same order as lds_bytes)
    LDS_AS uint8_t* base = (LDS_AS uint8_t*)PAR_LDS_BASE();
    const uint32_t nmax = A.nmax;
    f.kn = (lds_u32*)base; base += 4u * nmax;
    f.st = (lds_u32*)base; base += 4u * nmax;
    f.ref = (lds_u32*)base; base += 4u * nmax;
    f.rvi = (lds_u32*)base; base += 4u * nmax;
    f.pay = (lds_u32*)base; base += 4u * nmax;
    f.fc = (lds_u16*)base; base += 2u * nmax;
    f.nx = (lds_u16*)base; base += 2u * nmax;
    f.pa = (lds_u16*)base; base += 2u * nmax;
    f.rk = (lds_u16*)base; base += 2u * nmax;
    f.sz = (lds_u16*)base; base += 2u * nmax;
    f.pe = (lds_u16*)base; base += 2u * nmax;
    f.cg = (lds_u16*)base; base += 2u * nmax;
    f.wnext = (lds_u16*)base; base += 2u * nmax;
    f.s1 = (lds_u16*)base; base += 2u * nmax;
    base += (8u - (uint32_t)(uintptr_t)base % 8u) % 8u;
    f.defs = (DefEnt*)base; base += sizeof(DefEnt) * A.maxdef;
    f.args = (ArgEnt*)base; base += sizeof(ArgEnt) * MAX_ARGS;
    f.misc = (lds_u32*)base; base += 4 * MI_WORDS;
    f.whead = (lds_u32*)base;
    pre_function(f, X);
    PAR_SYNC();
    if (f.tid == 0) {
        if (f.misc[MI_FB]) {       // declined (before anything of it was listed)
            FuncIr r{}; r.def_node = dn; r.func = fidx; r.status = DS_FALLBACK; r.pad = f.misc[MI_WHY]; A.out[fidx] = r;
            FnInfo I{}; I.dn = dn; I.nn = 0; gst(X.fninfo + fidx, I);
            gst(X.fb + fidx, (uint32_t)f.misc[MI_WHY]);
        }
    }
}

// a ready list that was too small for its class (the regions hold every node of the class: never) is an internal error of the unit
__global__ void k_flat_check(const uint32_t* __restrict__ ctl, uint32_t* __restrict__ status) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && ctl[CTL_LISTOVF]) atomicMax(status, (uint32_t)DS_INTERNAL_CAPACITY);
}

// every function starts as "declined": the flat passes overwrite the ones they lower
__global__ void k_par_init(const opusfe_node* __restrict__ nodes, const uint32_t* __restrict__ def_nodes, uint32_t n_defs, const DefCnt* __restrict__ def_excl, FuncIr* __restrict__ out,
                           uint32_t* __restrict__ n_bodies, const uint8_t* __restrict__ skip, FnInfo* __restrict__ fninfo, uint32_t* __restrict__ fb) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_defs) return;
    const uint32_t dn = def_nodes[d];
    const uint8_t k = nodes[dn].kind;
    if (k != D_Function && k != D_Extern) return;
    const bool body = k == D_Function && !(skip && skip[d]);      // (a skipped function is left like an extern: no blocks)
    FuncIr r{}; r.def_node = dn; r.func = def_excl[d].funcs; r.status = body ? (uint32_t)DS_FALLBACK : (uint32_t)DS_OK;
    out[r.func] = r;
    if (fninfo) { fninfo[r.func] = FnInfo{}; fb[r.func] = 0; }
    if (body) atomicAdd(n_bodies, 1u);
}

}  // namespace par
}  // namespace opusfe
