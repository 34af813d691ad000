// Semantic tables on the GPU (restates the parts of compile.rs that IR generation leans on):
//   names       : open-addressing table  name -> definitions of that name in definition order
//                 (name_resolution_map / global_resolution_map / type_resolution_map, compile.rs:12-16,30-31)
//   types       : one 16-byte record per type.  resolve_type (compile.rs:919-966) is a pure function of
//                 the parse-tree type node, so it is evaluated for EVERY Y_* node of the AST in one
//                 node-parallel kernel; `type X` placeholders (preload_definition, compile.rs:968-981)
//                 are filled by finalize (compile.rs:1055-1062) with the Q14 "copy the current contents" rule.
//   functions   : user functions / externs in definition order; the 131 builtin overloads
//                 (compile.rs:404-500) are encoded arithmetically in the function id.
//   type algebra: types_match*, can_autocast, can_cast (compile.rs:599-775) as device functions.
#pragma once
#include <hip/hip_runtime.h>
#include "unit.h"
#include "scan.h"

namespace opusfe {

// PrimitiveType order of compile.rs:324-344 == type ids 0..17
enum : uint32_t { TP_Integer8 = 0, TP_Integer16, TP_Integer32, TP_Integer64, TP_Natural8, TP_Natural16, TP_Natural32, TP_Natural64,
                  TP_Bool, TP_Null, TP_Error, TP_GenericInteger, TP_Generic, TP_String, TP_Size, TP_Offset, TP_Float32, TP_Float64,
                  TP_COUNT };
enum TypeSrc : uint8_t { TS_NONE = 0, TS_NODE = 1, TS_FUNC = 2 };

struct TypeRec {
    uint8_t kind;      // TyKind
    uint8_t ptr;       // PtrKind (Pointer)
    uint8_t src;       // Record / Function: where the member list lives
    uint8_t pad;
    uint32_t sub;      // Pointer / Array: element type; Function: return type
    uint32_t ref;      // Record: Y_Record node; Function: Y_Proc node (TS_NODE) or function index (TS_FUNC); Array: Integer token
    uint32_t alias;    // name token of the `type` definition this id is the placeholder of, else NONE
};
static_assert(sizeof(TypeRec) == 16, "type record");

struct FuncRec { uint32_t def_node; uint32_t name_tok; uint32_t nargs; uint32_t ret; uint32_t first_arg; uint32_t flags; };   // flags bit0 = extern
struct GlobalRec { uint32_t def_node; uint32_t type_id; uint32_t is_const; uint32_t name_tok; };
struct NameSlot { uint32_t rep; uint32_t count; uint32_t offset; uint32_t fill; };   // rep = representative name token (NONE = empty)
struct NameEntry { uint32_t def_node; uint32_t kind_index; };                          // kind in the top 2 bits: 0 function, 1 global, 2 type
enum { NK_FUNCTION = 0u, NK_GLOBAL = 1u, NK_TYPE = 2u };

// builtin function ids: 0x80000000 | sym-token-kind << 8 | index into the overload list of that symbol
#define FN_BUILTIN 0x80000000u
#define FN_NONE 0xFFFFFFFFu

// Everything device code needs to look things up; passed by value to kernels.
struct SemView {
    GTable<uint8_t> src;
    GTable<opusfe_token> tok;
    GTable<uint64_t> val64;
    GTable<opusfe_node> nodes;
    GTable<uint32_t> tid_of_node;    // Y_* nodes: resolved type id; D_Type: placeholder id
    TypeRec* types;                  // global records [0, n_gtypes), then per-function local regions (gld / gst)
    GTable<NameSlot> slots; uint32_t slot_mask;
    GTable<NameEntry> entries;
    GTable<FuncRec> funcs;
    GTable<GlobalRec> globals;
    uint32_t n_gtypes;
    const uint32_t* file_start; uint32_t n_files;      // group mode: where every file of the packed unit starts (else null, 0)
    // the mix of the symbol keys a definition's code looks up (NameKey)
    __device__ __forceinline__ uint32_t mix_of_def(uint32_t def_node) const { return n_files ? pos_file(file_start, n_files, nodes[def_node].low) * FILE_MIX : 0u; }
};

// ---- names ------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t fnv1a(const char* s, int n) { uint32_t h = 2166136261u; for (int i = 0; i < n; i++) h = (h ^ (uint8_t)s[i]) * 16777619u; return h; }
// symbol text of the token that may follow `operator` (parse.rs:224-243, 773-801)
__host__ __device__ __forceinline__ const char* sym_text(uint8_t k, int* n) {
    switch (k) {
        case T_Plus: *n = 1; return "+"; case T_Minus: *n = 1; return "-"; case T_Asterisk: *n = 1; return "*"; case T_Slash: *n = 1; return "/";
        case T_Percent: *n = 1; return "%"; case T_EqualsEquals: *n = 1; return "="; case T_LessThan: *n = 1; return "<";
        case T_GreaterThan: *n = 1; return ">"; case T_LessThanEquals: *n = 2; return "<="; case T_GreaterThanEquals: *n = 2; return ">=";
        case T_Is: *n = 2; return "is"; case T_Not: *n = 3; return "not"; case T_For: *n = 3; return "for"; case T_Continue: *n = 8; return "continue";
        default: *n = 0; return "";
    }
}
__host__ __device__ __forceinline__ uint8_t sym_of_op(uint8_t op) {
    switch (op) {
        case OP_Plus: return T_Plus; case OP_Minus: return T_Minus; case OP_Times: return T_Asterisk; case OP_Divide: return T_Slash;
        case OP_Modulo: return T_Percent; case OP_Equals: return T_EqualsEquals; case OP_LessThan: return T_LessThan;
        case OP_GreaterThan: return T_GreaterThan; case OP_LessThanEquals: return T_LessThanEquals;
        case OP_GreaterThanEquals: return T_GreaterThanEquals; case OP_Is: return T_Is; case OP_Not: return T_Not;
        default: return T_EndOfFile;
    }
}
// A name is either an identifier token or the symbol of an operator-ish token kind.
// Group mode (files packed into one unit): the payload of every name token carries file * FILE_MIX (lex.h k_lex_emit) — an identifier's
// hash is shifted by it, a symbol token's payload IS it — and a symbol key carries the `mix` of the function that looks it up, so that
// one name table scopes every name to its file.  Outside group mode every mix is 0.
struct NameKey { uint32_t tok; uint8_t sym; uint32_t mix; };   // tok != NONE: identifier/operator token in the source; else symbol kind (+ file mix)
// (TokP / SrcP: a raw pointer or a GTable)
template <class TokP> __host__ __device__ __forceinline__ uint8_t name_sym(TokP tok, NameKey k) { return k.tok != OPUSFE_NONE ? (uint8_t)tok[k.tok].kind : k.sym; }
__host__ __device__ __forceinline__ uint32_t sym_hash(uint8_t s);
template <class TokP> __host__ __device__ __forceinline__ uint32_t name_mix(TokP tok, NameKey k) { return k.tok != OPUSFE_NONE ? (uint32_t)tok[k.tok].payload : k.mix; }   // symbol keys only
template <class TokP> __host__ __device__ __forceinline__ uint32_t name_hash(TokP tok, NameKey k) {
    uint8_t s = name_sym(tok, k);
    if (s == T_Identifier) return tok[k.tok].payload;
    return sym_hash(s) + name_mix(tok, k);
}
__host__ __device__ __forceinline__ uint32_t sym_hash(uint8_t s) {
    switch (s) {   // FNV-1a of the symbol text, folded at compile time
        case T_Plus: return fnv1a("+", 1); case T_Minus: return fnv1a("-", 1); case T_Asterisk: return fnv1a("*", 1); case T_Slash: return fnv1a("/", 1);
        case T_Percent: return fnv1a("%", 1); case T_EqualsEquals: return fnv1a("=", 1); case T_LessThan: return fnv1a("<", 1);
        case T_GreaterThan: return fnv1a(">", 1); case T_LessThanEquals: return fnv1a("<=", 2); case T_GreaterThanEquals: return fnv1a(">=", 2);
        case T_Is: return fnv1a("is", 2); case T_Not: return fnv1a("not", 3); case T_For: return fnv1a("for", 3); case T_Continue: return fnv1a("continue", 8);
        default: return fnv1a("", 0);
    }
}
template <class SrcP, class TokP> __host__ __device__ __forceinline__ bool name_eq(SrcP src, TokP tok, NameKey a, NameKey b) {
    uint8_t sa = name_sym(tok, a), sb = name_sym(tok, b);
    if (sa != sb) return false;
    if (sa != T_Identifier) return name_mix(tok, a) == name_mix(tok, b);
    const opusfe_token x = tok[a.tok], y = tok[b.tok];
    if (x.len != y.len || x.payload != y.payload) return false;
    if (x.low == y.low) return true;
    for (uint32_t i = 0; i < x.len; i++) if (src[x.low + i] != src[y.low + i]) return false;
    return true;
}
template <class SrcP, class TokP> __host__ __device__ __forceinline__ bool name_is(SrcP src, TokP tok, uint32_t t, const char* lit, uint32_t n) {
    const opusfe_token x = tok[t];
    if (x.kind != T_Identifier || x.len != n) return false;
    for (uint32_t i = 0; i < n; i++) if (src[x.low + i] != (uint8_t)lit[i]) return false;
    return true;
}
__host__ __device__ __forceinline__ NameKey key_tok(uint32_t t) { return NameKey{t, 0, 0}; }
__host__ __device__ __forceinline__ NameKey key_sym(uint8_t s, uint32_t mix) { return NameKey{OPUSFE_NONE, s, mix}; }

// slot of a name, or NONE
__device__ __forceinline__ uint32_t name_find(const SemView& S, NameKey k) {
    uint32_t h = name_hash(S.tok, k) & S.slot_mask;
    for (uint32_t probe = 0; probe <= S.slot_mask; probe++) {
        uint32_t rep = S.slots[h].rep;
        if (rep == OPUSFE_NONE) return OPUSFE_NONE;
        if (name_eq(S.src, S.tok, k, key_tok(rep))) return h;
        h = (h + 1) & S.slot_mask;
    }
    return OPUSFE_NONE;
}
// last entry of the given kind for a name (HashMap::insert overwrites: the last definition wins), or NONE
__device__ __forceinline__ uint32_t name_last(const SemView& S, NameKey k, uint32_t kind) {
    uint32_t s = name_find(S, k);
    if (s == OPUSFE_NONE) return OPUSFE_NONE;
    const NameSlot sl = S.slots[s];
    for (uint32_t j = sl.count; j > 0; j--) {
        NameEntry e = S.entries[sl.offset + j - 1];
        if ((e.kind_index >> 30) == kind) return sl.offset + j - 1;
    }
    return OPUSFE_NONE;
}

// ---- builtin overloads (compile.rs:404-500) -----------------------------------------------------
__host__ __device__ __forceinline__ int builtin_count(uint8_t sym) {
    switch (sym) {
        case T_Plus: case T_Minus: case T_Asterisk: case T_Slash: case T_Percent: return 12;
        case T_EqualsEquals: case T_LessThan: case T_GreaterThan: case T_LessThanEquals: case T_GreaterThanEquals: return 14;
        case T_Not: return 1;
        default: return 0;
    }
}
// overload idx of a symbol -> its (single) parameter type.  Arithmetic, not a table: a constant array in device
// code is a load from memory.  Overloads 0..7 are Integer8..Natural64 == primitive ids 0..7; then
//   arithmetic (12): Size Offset Float32 Float64;   comparison (14): Size Offset Bool Null Float32 Float64
__host__ __device__ __forceinline__ uint32_t builtin_arg_type(uint8_t sym, int idx) {
    static_assert(TP_Integer8 == 0 && TP_Natural64 == 7 && TP_Bool == 8 && TP_Null == 9 && TP_Size == 14 && TP_Offset == 15 && TP_Float32 == 16 && TP_Float64 == 17, "primitive ids");
    const int n = builtin_count(sym);
    const uint32_t i = (uint32_t)idx;
    if (n == 12) return i < 8 ? i : i + 6;
    if (n == 14) return i < 8 ? i : i < 10 ? i + 6 : i < 12 ? i - 2 : i + 4;
    return TP_Bool;
}
// inverse of builtin_arg_type: index of the overload of `sym` whose parameter type is primitive t, or -1
__host__ __device__ __forceinline__ int builtin_index_of(uint8_t sym, uint32_t t) {
    const int n = builtin_count(sym);
    if (n == 1) return t == TP_Bool ? 0 : -1;
    if (n != 12 && n != 14) return -1;
    if (t < 8) return (int)t;
    if (t == TP_Size || t == TP_Offset) return (int)t - 6;
    if (t == TP_Float32 || t == TP_Float64) return n == 12 ? (int)t - 6 : (int)t - 4;
    if (n == 14 && (t == TP_Bool || t == TP_Null)) return (int)t + 2;
    return -1;
}
// overload of `sym` chosen for primitive, non-Error operand types a0 / a1 when only the builtins are candidates
// (derivation: lower.h compiler_lookup_function), or -1 when the general algorithm has to decide
__host__ __device__ __forceinline__ int builtin_pick(uint8_t sym, uint32_t a0, uint32_t a1);
__host__ __device__ __forceinline__ uint32_t builtin_ret_type(uint8_t sym, int idx) { return builtin_count(sym) == 12 ? builtin_arg_type(sym, idx) : (uint32_t)TP_Bool; }
__host__ __device__ __forceinline__ uint32_t builtin_nargs(uint8_t sym) { return sym == T_Not ? 1u : 2u; }

// ---- type algebra -------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint8_t prim_kind(uint32_t p) {   // a byte of a packed constant (no table in memory)
    constexpr uint64_t K0 = (uint64_t)TY_Integer8 | (uint64_t)TY_Integer16 << 8 | (uint64_t)TY_Integer32 << 16 | (uint64_t)TY_Integer64 << 24 |
                            (uint64_t)TY_Natural8 << 32 | (uint64_t)TY_Natural16 << 40 | (uint64_t)TY_Natural32 << 48 | (uint64_t)TY_Natural64 << 56;
    constexpr uint64_t K1 = (uint64_t)TY_Bool | (uint64_t)TY_Null << 8 | (uint64_t)TY_Error << 16 | (uint64_t)TY_GenericInteger << 24 |
                            (uint64_t)TY_Generic << 32 | (uint64_t)TY_String << 40 | (uint64_t)TY_Size << 48 | (uint64_t)TY_Offset << 56;
    constexpr uint64_t K2 = (uint64_t)TY_Float32 | (uint64_t)TY_Float64 << 8;
    static_assert(TP_Bool == 8 && TP_Error == 10 && TP_GenericInteger == 11 && TP_Generic == 12 && TP_String == 13 && TP_COUNT == 18, "primitive ids");
    const uint64_t w = p < 8 ? K0 : p < 16 ? K1 : K2;
    return (uint8_t)(w >> ((p & 7) * 8));
}
__host__ __device__ __forceinline__ bool kind_is_integral(uint8_t k) {   // compile.rs:172-180
    return k <= TY_Natural64 || k == TY_Size || k == TY_Offset;
}

__host__ __device__ __forceinline__ int builtin_pick(uint8_t sym, uint32_t a0, uint32_t a1) {
    const bool g0 = a0 == TP_GenericInteger || a0 == TP_Generic, g1 = a1 == TP_GenericInteger || a1 == TP_Generic;
    if (!g0 && !g1) return a0 == a1 ? builtin_index_of(sym, a0) : -1;
    if (!g0) return (a1 == TP_Generic || kind_is_integral(prim_kind(a0))) ? builtin_index_of(sym, a0) : -1;
    if (!g1) return (a0 == TP_Generic || kind_is_integral(prim_kind(a1))) ? builtin_index_of(sym, a1) : -1;
    if (a0 == TP_GenericInteger && a1 == TP_GenericInteger) return builtin_index_of(sym, TP_Integer32);
    return -1;
}

// M: the machine whose static LDS state the operations work on (lower.h Lowerer): M::S (tables), M::dev_status
// (DS_UNSUPPORTED_SEMANTIC when the reference would overflow its stack), M::tbase / M::ntypes / M::lds.types (records
// of the types the running function created itself, kept in LDS), M::lds.match_stack (2 x 48 words of work list).
template <class M> struct TypeOps {
    // primitives need no memory access; the function's own types come from LDS; everything else from the table
    __device__ __forceinline__ static TypeRec rec(uint32_t t) {
        if (t < TP_COUNT) { TypeRec r{}; r.kind = prim_kind(t); r.alias = OPUSFE_NONE; return r; }
        constexpr uint32_t CACHE = sizeof(M::lds.types) / sizeof(TypeRec);
        const uint32_t k = t - M::tbase, cached = M::ntypes < CACHE ? M::ntypes : CACHE;
        if (t >= M::tbase && k < cached) return M::lds.types[k];
        return gld(M::S.types + t);
    }
    __device__ __forceinline__ static uint8_t kind(uint32_t t) { return t < TP_COUNT ? prim_kind(t) : rec(t).kind; }
    __device__ __forceinline__ static bool is_integral(uint32_t t) { return kind_is_integral(kind(t)); }

    // member lists of Record / Function types
    __device__ static uint32_t member_count(const TypeRec& r) {
        if (r.kind == TY_Record) return M::S.nodes[r.ref].nchild;
        if (r.kind == TY_Function) return r.src == TS_FUNC ? M::S.funcs[r.ref].nargs : (uint32_t)M::S.nodes[r.ref].nchild - 1u;
        return 0;
    }
    // iterator over members: cursor is an AST node (X_NamedType for records and functions from definitions; Y_* for Proc types)
    __device__ static uint32_t member_first(const TypeRec& r) {
        if (r.kind == TY_Function && r.src == TS_FUNC) return M::S.funcs[r.ref].first_arg;
        return M::S.nodes[r.ref].first_child;
    }
    __device__ static uint32_t member_next(uint32_t cur) { return M::S.nodes[cur].next; }
    __device__ static uint32_t member_type(uint32_t cur) {
        const opusfe_node& n = M::S.nodes[cur];
        return n.kind == X_NamedType ? M::S.tid_of_node[n.first_child] : M::S.tid_of_node[cur];
    }
    __device__ static uint32_t member_name(uint32_t cur) { return M::S.nodes[cur].payload; }
    __device__ static uint64_t array_size(const TypeRec& r) {
        const opusfe_token& t = M::S.tok[r.ref];
        return (t.flags & OPUSFE_TF_INLINE) ? (uint64_t)t.payload : M::S.val64[t.payload];
    }

    // compile.rs:698-751, recursion replaced by a bounded work list (the reference recurses without a
    // cycle check: on a self-referential record it overflows its stack; we report UNSUPPORTED instead)
    __device__ static bool types_match(uint32_t a0, uint32_t b0) {
        a0 = uni(a0); b0 = uni(b0);
        if (a0 < TP_COUNT && b0 < TP_COUNT) return a0 == b0;    // primitives: one id per kind
        uint32_t* stack_a = M::lds.match_stack; uint32_t* stack_b = M::lds.match_stack + 48;
        int sp = 0;
        uint32_t a = a0, b = b0;
        uint32_t budget = 1u << 20;
        for (;;) {
            if (--budget == 0) { M::dev_status = DS_UNSUPPORTED_SEMANTIC; return false; }
            bool descend = false;
            {
                const TypeRec ra = rec(a), rb = rec(b);
                if (ra.kind != rb.kind) return false;
                switch (ra.kind) {
                    case TY_Pointer: if (ra.ptr != rb.ptr) return false; a = ra.sub; b = rb.sub; descend = true; break;
                    case TY_Array: if (array_size(ra) != array_size(rb)) return false; a = ra.sub; b = rb.sub; descend = true; break;
                    case TY_Record: {
                        uint32_t n = member_count(ra);
                        if (n != member_count(rb)) return false;
                        uint32_t ca = member_first(ra), cb = member_first(rb);
                        for (uint32_t i = 0; i < n; i++) {
                            if (!name_eq(M::S.src, M::S.tok, key_tok(member_name(ca)), key_tok(member_name(cb)))) return false;
                            if (sp >= 48) { M::dev_status = DS_UNSUPPORTED_SEMANTIC; return false; }
                            stack_a[sp] = member_type(ca); stack_b[sp] = member_type(cb); sp++;
                            ca = member_next(ca); cb = member_next(cb);
                        }
                        break;
                    }
                    case TY_Function: {
                        uint32_t n = member_count(ra);
                        if (n != member_count(rb)) return false;
                        if (sp >= 48) { M::dev_status = DS_UNSUPPORTED_SEMANTIC; return false; }
                        stack_a[sp] = ra.sub; stack_b[sp] = rb.sub; sp++;
                        uint32_t ca = member_first(ra), cb = member_first(rb);
                        for (uint32_t i = 0; i < n; i++) {
                            if (sp >= 48) { M::dev_status = DS_UNSUPPORTED_SEMANTIC; return false; }
                            stack_a[sp] = member_type(ca); stack_b[sp] = member_type(cb); sp++;
                            ca = member_next(ca); cb = member_next(cb);
                        }
                        break;
                    }
                    default: break;   // payload-free kinds, including (Error, Error)
                }
            }
            if (descend) continue;
            if (sp == 0) return true;
            sp--; a = stack_a[sp]; b = stack_b[sp];
        }
    }
    // compile.rs:753-763
    __device__ static bool types_match_allow_generic(uint32_t a, uint32_t b) {
        a = uni(a); b = uni(b);
        uint8_t ka = kind(a), kb = kind(b);
        if ((ka == TY_GenericInteger && kind_is_integral(kb)) || (kb == TY_GenericInteger && kind_is_integral(ka))) return true;
        if (ka == TY_Generic || kb == TY_Generic) return true;
        return types_match(a, b);
    }
    // compile.rs:765-775
    __device__ static bool types_match_default_generic(uint32_t a, uint32_t b) {
        a = uni(a); b = uni(b);
        uint8_t ka = kind(a), kb = kind(b);
        if ((ka == TY_GenericInteger && kb == TY_Integer32) || (ka == TY_Integer32 && kb == TY_GenericInteger)) return true;
        if (ka == TY_Generic || kb == TY_Generic) return false;
        return types_match(a, b);
    }
    // compile.rs:672-680
    __device__ static bool can_autocast_pointer_type(uint8_t from, uint8_t to) {
        return from == to || (from == PT_Mutable && to == PT_Reference) || (from == PT_ArrayMutable && to == PT_Array);
    }
    // compile.rs:624-670; returns CastKind or -1
    __device__ static int can_autocast(uint32_t from, uint32_t to) {
        from = uni(from); to = uni(to);
        if (from < TP_COUNT && to < TP_COUNT) {   // both primitive: no table access
            if (from == to) return CT_None;
            if (from == TP_Error || to == TP_Error) return CT_Error;
            if (from == TP_GenericInteger && kind_is_integral(prim_kind(to))) return CT_Error;
            if (from == TP_Generic) return CT_Error;
            return -1;
        }
        if (types_match(from, to)) return CT_None;
        const TypeRec t1 = rec(from), t2 = rec(to);
        if (t1.kind == TY_Error || t2.kind == TY_Error) return CT_Error;
        if (t1.kind == TY_GenericInteger && kind_is_integral(t2.kind)) return CT_Error;
        if (t1.kind == TY_Generic) return CT_Error;
        if (t1.kind == TY_Pointer && t2.kind == TY_Pointer) {
            uint8_t p1 = t1.ptr, p2 = t2.ptr;
            if (can_autocast_pointer_type(p1, p2) && types_match(t1.sub, t2.sub)) return CT_PointerType;
            if (((p1 == PT_Mutable && p2 == PT_ArrayMutable) || (p1 == PT_Mutable && p2 == PT_Array) || (p1 == PT_Reference && p2 == PT_Array))
                && types_match(t2.sub, TP_Natural8)) return CT_PointerType;
            if (((p1 == PT_ArrayMutable && p2 == PT_Mutable) || (p1 == PT_ArrayMutable && p2 == PT_Reference) || (p1 == PT_Array && p2 == PT_Reference))
                && types_match(t1.sub, TP_Natural8)) return CT_PointerType;
            const TypeRec s1 = rec(t1.sub);
            if (s1.kind == TY_Array && types_match(s1.sub, t2.sub)) {
                if ((p1 == PT_Reference && p2 == PT_Array) || (p1 == PT_Mutable && p2 == PT_ArrayMutable) || (p1 == PT_Mutable && p2 == PT_Array)) return CT_PointerToArray;
                return -1;
            }
            return -1;
        }
        return -1;
    }
    // compile.rs:682-696
    __device__ static int can_cast(uint32_t from, uint32_t to) {
        from = uni(from); to = uni(to);
        int c = can_autocast(from, to);
        if (c >= 0) return c;
        const TypeRec t1 = rec(from), t2 = rec(to);
        if (kind_is_integral(t1.kind) && kind_is_integral(t2.kind)) return CT_Integer;
        if (t1.kind == TY_Pointer && t2.kind == TY_Pointer && t1.ptr == t2.ptr) return CT_Pointer;
        if ((t1.kind == TY_Pointer || t1.kind == TY_Function) && kind_is_integral(t2.kind)) return CT_PointerToInteger;
        if ((t2.kind == TY_Pointer || t2.kind == TY_Function) && kind_is_integral(t1.kind)) return CT_IntegerToPointer;
        return -1;
    }
};

// ---- table construction kernels -----------------------------------------------------------------
struct DefCnt { uint32_t funcs, globals, types, named; };
struct OpAddDefCnt { __host__ __device__ DefCnt operator()(DefCnt a, DefCnt b) const { return DefCnt{a.funcs + b.funcs, a.globals + b.globals, a.types + b.types, a.named + b.named}; } };

__global__ void k_sem_classify(const opusfe_node* __restrict__ nodes, const uint32_t* __restrict__ def_nodes, uint32_t n_defs, DefCnt* __restrict__ out) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_defs) return;
    uint8_t k = nodes[def_nodes[d]].kind;
    DefCnt c{0, 0, 0, 0};
    if (k == D_Function || k == D_Extern) c.funcs = 1;
    else if (k == D_Variable || k == D_Constant) c.globals = 1;
    else if (k == D_Type) c.types = 1;
    c.named = c.funcs + c.globals + c.types;
    out[d] = c;
}

__device__ __forceinline__ uint32_t def_entry_kind(uint8_t k) { return (k == D_Function || k == D_Extern) ? NK_FUNCTION : (k == D_Type ? NK_TYPE : NK_GLOBAL); }

// claims / finds the slot of the definition's name and counts it
__global__ void k_sem_table_insert(const uint8_t* __restrict__ src, const opusfe_token* __restrict__ tok, const opusfe_node* __restrict__ nodes,
                                   const uint32_t* __restrict__ def_nodes, uint32_t n_defs, NameSlot* __restrict__ slots, uint32_t slot_mask,
                                   uint32_t* __restrict__ def_slot) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_defs) return;
    const opusfe_node& n = nodes[def_nodes[d]];
    if (n.kind == D_Import || n.kind == D_Library) { def_slot[d] = OPUSFE_NONE; return; }
    NameKey key = key_tok(n.payload);
    uint32_t h = name_hash(tok, key) & slot_mask;
    for (uint32_t probe = 0; probe <= slot_mask; probe++) {
        uint32_t rep = atomicCAS(&slots[h].rep, OPUSFE_NONE, n.payload);
        if (rep == OPUSFE_NONE || name_eq(src, tok, key, key_tok(rep))) { atomicAdd(&slots[h].count, 1u); def_slot[d] = h; return; }
        h = (h + 1) & slot_mask;
    }
    def_slot[d] = OPUSFE_NONE;   // cannot happen: the table has more slots than definitions
}

__global__ void k_sem_slot_counts(const NameSlot* __restrict__ slots, uint32_t n_slots, uint32_t* __restrict__ counts) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_slots) counts[s] = slots[s].count;
}

__global__ void k_sem_table_fill(const opusfe_node* __restrict__ nodes, const uint32_t* __restrict__ def_nodes, uint32_t n_defs,
                                 const DefCnt* __restrict__ def_excl, const uint32_t* __restrict__ def_slot, const uint32_t* __restrict__ slot_off,
                                 NameSlot* __restrict__ slots, NameEntry* __restrict__ entries) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_defs) return;
    uint32_t s = def_slot[d];
    if (s == OPUSFE_NONE) return;
    uint8_t k = nodes[def_nodes[d]].kind;
    uint32_t kind = def_entry_kind(k);
    uint32_t index = kind == NK_FUNCTION ? def_excl[d].funcs : kind == NK_GLOBAL ? def_excl[d].globals : def_excl[d].types;
    uint32_t pos = slot_off[s] + atomicAdd(&slots[s].fill, 1u);
    entries[pos] = NameEntry{def_nodes[d], (kind << 30) | index};
}

// one lane per slot: writes the offset and orders the slot's entries by definition.  Groups are tiny (insertion sort up to
// 16 entries); a name with very many definitions (1e5 overloads of one operator, generated code that reuses one name) is
// ordered by an in-place heapsort, O(k log k) on the one lane instead of O(k^2).
__global__ void k_sem_table_sort(NameSlot* __restrict__ slots, uint32_t n_slots, const uint32_t* __restrict__ slot_off, NameEntry* __restrict__ entries) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    uint32_t off = slot_off[s], n = slots[s].count;
    slots[s].offset = off;
    NameEntry* e = entries + off;
    if (n <= 16) {
        for (uint32_t i = 1; i < n; i++) {
            NameEntry x = e[i];
            uint32_t j = i;
            while (j > 0 && e[j - 1].def_node > x.def_node) { e[j] = e[j - 1]; j--; }
            e[j] = x;
        }
        return;
    }
    auto sift = [&](uint32_t root, uint32_t end) {      // max-heap on def_node over e[0, end)
        for (;;) {
            uint32_t c = 2 * root + 1;
            if (c >= end) return;
            if (c + 1 < end && e[c + 1].def_node > e[c].def_node) c++;
            if (e[root].def_node >= e[c].def_node) return;
            NameEntry t = e[root]; e[root] = e[c]; e[c] = t;
            root = c;
        }
    };
    for (uint32_t i = n / 2; i > 0; i--) sift(i - 1, n);
    for (uint32_t end = n; end > 1; end--) { NameEntry t = e[0]; e[0] = e[end - 1]; e[end - 1] = t; sift(0, end - 1); }
}

// ---- group mode (a corpus of files packed into one unit): every name is scoped to its file by the mix in its key (NameKey) ----
// Which operator symbols have user-defined overloads is asked per unit (k_sem_user_syms: a gate in front of the builtin shortcut, where
// "yes" only means "take the general path"): in a group, a symbol that ANY file defines answers yes.
__global__ void k_group_user_syms(const opusfe_node* __restrict__ nodes, const opusfe_token* __restrict__ tok, const uint32_t* __restrict__ def_nodes, uint32_t n_defs,
                                  unsigned long long* __restrict__ user_syms) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_defs) return;
    const opusfe_node n = nodes[def_nodes[d]];
    if (n.kind != D_Function && n.kind != D_Extern) return;
    const uint8_t k = tok[n.payload].kind;
    if (k != T_Identifier) atomicOr(user_syms, 1ull << (k & 63));
}
// first definition of every file (definitions are in file order); files without definitions keep NONE
__global__ void k_group_first_defs(const opusfe_node* __restrict__ nodes, const uint32_t* __restrict__ def_nodes, uint32_t n_defs, const uint32_t* __restrict__ file_start,
                                   uint32_t n_files, uint32_t* __restrict__ first_def, uint32_t* __restrict__ def_file) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_defs) return;
    const uint32_t f = pos_file(file_start, n_files, nodes[def_nodes[d]].low);
    def_file[d] = f;
    if (d == 0 || pos_file(file_start, n_files, nodes[def_nodes[d - 1]].low) != f) first_def[f] = d;
}

__global__ void k_group_def_mask(const uint32_t* __restrict__ def_file, const uint32_t* __restrict__ flags, uint32_t n_defs, uint32_t mask, uint8_t* __restrict__ out) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < n_defs) out[d] = (flags[def_file[d]] & mask) ? 1 : 0;
}
__global__ void k_group_finerr(const uint32_t* __restrict__ fin_cnt, const uint32_t* __restrict__ def_file, uint32_t n_defs, uint32_t* __restrict__ flags) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < n_defs && fin_cnt[d]) atomicOr(&flags[def_file[d]], (uint32_t)FF_FINERR);
}

__global__ void k_sem_slots_init(NameSlot* __restrict__ slots, uint32_t n_slots) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_slots) slots[s] = NameSlot{OPUSFE_NONE, 0, 0, 0};
}

// nodes that own a fresh type record: structured parse-tree types and `type` definitions (their placeholder)
__global__ void k_sem_type_flags(const opusfe_node* __restrict__ nodes, uint32_t n_nodes, uint32_t* __restrict__ flag) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    uint8_t k = nodes[i].kind;
    flag[i] = (k == D_Type || (k >= Y_Reference && k <= Y_Array)) ? 1u : 0u;
}

// resolve_type for every parse-tree type node (compile.rs:919-966).  The id of a child is a pure function of the
// child node, so no ordering between lanes is needed.
__device__ __forceinline__ uint32_t sem_type_id_of(const uint8_t* src, const opusfe_token* tok, const opusfe_node* nodes, const uint32_t* slot_of_node,
                                                   const NameSlot* slots, uint32_t slot_mask, const NameEntry* entries, uint32_t y, bool* undefined) {
    const opusfe_node& n = nodes[y];
    if (n.kind == Y_Null) return TP_Null;
    if (n.kind != Y_Name) return TP_COUNT + slot_of_node[y];
    uint32_t t = n.payload;
    if (tok[t].kind == T_Identifier) {
        struct { const char* s; uint32_t n; uint32_t id; } const B[14] = {
            {"int8", 4, TP_Integer8}, {"int16", 5, TP_Integer16}, {"int32", 5, TP_Integer32}, {"int64", 5, TP_Integer64},
            {"nat8", 4, TP_Natural8}, {"nat16", 5, TP_Natural16}, {"nat32", 5, TP_Natural32}, {"nat64", 5, TP_Natural64},
            {"size", 4, TP_Size}, {"offset", 6, TP_Offset}, {"float32", 7, TP_Float32}, {"float64", 7, TP_Float64},
            {"bool", 4, TP_Bool}, {"string", 6, TP_String}};
        uint32_t len = tok[t].len;
        if (len >= 4 && len <= 7)
            for (int i = 0; i < 14; i++) if (name_is(src, tok, t, B[i].s, B[i].n)) return B[i].id;
    }
    // type_resolution_map: the last `type` definition of that name
    NameKey key = key_tok(t);
    uint32_t h = name_hash(tok, key) & slot_mask;
    for (uint32_t probe = 0; probe <= slot_mask; probe++) {
        uint32_t rep = slots[h].rep;
        if (rep == OPUSFE_NONE) break;
        if (name_eq(src, tok, key, key_tok(rep))) {
            const NameSlot sl = slots[h];
            for (uint32_t j = sl.count; j > 0; j--) {
                NameEntry e = entries[sl.offset + j - 1];
                if ((e.kind_index >> 30) == NK_TYPE) return TP_COUNT + slot_of_node[e.def_node];
            }
            break;
        }
        h = (h + 1) & slot_mask;
    }
    *undefined = true;
    return TP_Error;
}

// The `type` definition of the same name that precedes definition node `def` (duplicate names: every one of them writes
// the placeholder that type_resolution_map holds for the name, compile.rs:1055-1062, in definition order), else NONE.
__device__ __forceinline__ uint32_t sem_prev_type_def(const uint8_t* src, const opusfe_token* tok, uint32_t name_tok, uint32_t def,
                                                      const NameSlot* slots, uint32_t slot_mask, const NameEntry* entries) {
    NameKey key = key_tok(name_tok);
    uint32_t h = name_hash(tok, key) & slot_mask;
    for (uint32_t probe = 0; probe <= slot_mask; probe++) {
        uint32_t rep = slots[h].rep;
        if (rep == OPUSFE_NONE) break;
        if (name_eq(src, tok, key, key_tok(rep))) {
            const NameSlot sl = slots[h];
            uint32_t best = OPUSFE_NONE;      // the largest type-definition node below `def` (entries are few per name)
            for (uint32_t j = 0; j < sl.count; j++) {
                NameEntry e = entries[sl.offset + j];
                if ((e.kind_index >> 30) == NK_TYPE && e.def_node < def && (best == OPUSFE_NONE || e.def_node > best)) best = e.def_node;
            }
            return best;
        }
        h = (h + 1) & slot_mask;
    }
    return OPUSFE_NONE;
}

__global__ void k_sem_resolve_types(const uint8_t* __restrict__ src, const opusfe_token* __restrict__ tok, const opusfe_node* __restrict__ nodes, uint32_t n_nodes,
                                    const uint32_t* __restrict__ slot_of_node, const NameSlot* __restrict__ slots, uint32_t slot_mask,
                                    const NameEntry* __restrict__ entries, uint32_t* __restrict__ tid_of_node, TypeRec* __restrict__ types,
                                    uint8_t* __restrict__ undefined_flag, uint32_t* __restrict__ ph_def, uint32_t* __restrict__ ph_prev) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < TP_COUNT) { TypeRec p{}; p.kind = prim_kind(i); p.alias = OPUSFE_NONE; types[i] = p; }   // launched with >= TP_COUNT lanes
    if (i >= n_nodes) return;
    const opusfe_node n = nodes[i];
    undefined_flag[i] = 0;
    if (n.kind == D_Type) {
        uint32_t id = TP_COUNT + slot_of_node[i];
        TypeRec r{}; r.kind = TY_Error; r.alias = n.payload;      // preload_definition: placeholder (compile.rs:968-981)
        types[id] = r;
        tid_of_node[i] = id;
        ph_def[slot_of_node[i]] = i;
        ph_prev[slot_of_node[i]] = sem_prev_type_def(src, tok, n.payload, i, slots, slot_mask, entries);
        return;
    }
    if (n.kind < Y_Name || n.kind > Y_Array) { tid_of_node[i] = OPUSFE_NONE; return; }
    bool undef = false;
    uint32_t id = sem_type_id_of(src, tok, nodes, slot_of_node, slots, slot_mask, entries, i, &undef);
    tid_of_node[i] = id;
    undefined_flag[i] = undef ? 1 : 0;
    if (n.kind == Y_Name || n.kind == Y_Null) return;
    ph_def[slot_of_node[i]] = OPUSFE_NONE;
    TypeRec r{}; r.alias = OPUSFE_NONE;
    bool u2 = false;
    switch (n.kind) {
        case Y_Reference: r.kind = TY_Pointer; r.ptr = PT_Reference; break;
        case Y_MutableReference: r.kind = TY_Pointer; r.ptr = PT_Mutable; break;
        case Y_ArrayReference: r.kind = TY_Pointer; r.ptr = PT_Array; break;
        case Y_MutableArrayReference: r.kind = TY_Pointer; r.ptr = PT_ArrayMutable; break;
        case Y_Array: r.kind = TY_Array; r.ref = n.payload; break;
        case Y_Record: r.kind = TY_Record; r.src = TS_NODE; r.ref = i; break;
        default: r.kind = TY_Function; r.src = TS_NODE; r.ref = i; break;   // Y_Proc
    }
    if (n.kind == Y_Proc) {
        uint32_t c = n.first_child;
        for (uint32_t k = 0; k + 1 < n.nchild; k++) c = nodes[c].next;
        r.sub = sem_type_id_of(src, tok, nodes, slot_of_node, slots, slot_mask, entries, c, &u2);   // return type = last child
    } else if (n.kind != Y_Record) {
        r.sub = sem_type_id_of(src, tok, nodes, slot_of_node, slots, slot_mask, entries, n.first_child, &u2);
    }
    types[id] = r;
}

// the pointer-to-primitive table behind the parse-tree types: id = base + PtrKind * TP_COUNT + primitive.  The node-parallel
// lowering (lower_par.h) names `@mut int32`, `@int32`, ... by these ids instead of creating a record per use.
__global__ void k_sem_pp_types(TypeRec* __restrict__ pp) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * TP_COUNT) return;
    TypeRec r{}; r.kind = TY_Pointer; r.ptr = (uint8_t)(i / TP_COUNT); r.sub = i % TP_COUNT; r.alias = OPUSFE_NONE;
    pp[i] = r;
}

// kind, span, and what the message prints besides the span (compile.rs:816-915): `name` = a token index (variable / field /
// function name; an operator token for `operator +` definitions) or 0x80000000 | token kind of a builtin symbol, or NONE;
// `nargs` argument types of a function identifier (t[0..3]: the first four); t[0], t[1] = the one or two types of the typed kinds
// (expected/found, from/to, left/right); type ids index OPUSFE_A_TYPES
struct SemErr { uint32_t kind, low, len, name, nargs, t[4]; };
__host__ __device__ inline SemErr sem_err(uint32_t kind, uint32_t low, uint32_t len, uint32_t name = OPUSFE_NONE, uint32_t nargs = 0, uint32_t t0 = OPUSFE_NONE,
                                          uint32_t t1 = OPUSFE_NONE, uint32_t t2 = OPUSFE_NONE, uint32_t t3 = OPUSFE_NONE) {
    SemErr e; e.kind = kind; e.low = low; e.len = len; e.name = name; e.nargs = nargs; e.t[0] = t0; e.t[1] = t1; e.t[2] = t2; e.t[3] = t3; return e;
}
// compile.rs:124-152 order
enum SErrKind : uint32_t { SE_ParseError, SE_UndefinedVariable, SE_ShadowedName, SE_UndefinedFunction, SE_NoOverloadForFunction, SE_AmbiguousOverload,
    SE_AmbiguousFunctionReference, SE_UnexpectedType, SE_InvalidType, SE_InvalidOperandType, SE_InvalidOperandTypes, SE_FunctionMightNotReturn,
    SE_BreakOutsideLoop, SE_ContinueOutsideLoop, SE_InvalidLvalue, SE_ImmutableLvalue, SE_ArrayPointerToNonArray, SE_InvalidCast, SE_UndefinedType,
    SE_FieldAccessOnNonRecord, SE_FieldDoesNotExist, SE_CallOfNonFunction, SE_ReferenceToBuiltinFunction, SE_AttemptToCallOrIndexConstant,
    SE_InvalidNumberOfArgs, SE_InvalidConstantExpression,
    SE_Continuation = 0x100 /* not a diagnostic: argument types 5.. of the function identifier of the record before it */ };

// Contents of a type id as finalize_definition of the definition at node `at_def` sees them (compile.rs:1055-1062,
// Q14): a `type` placeholder holds Type::Error until its own definition has been finalized, and an alias of an
// alias captures whatever the target held at that moment.  Decided from the AST alone (placeholders are written
// concurrently by other lanes of k_sem_finalize), returns the record and whether it is final.
__device__ __forceinline__ TypeRec sem_contents_at(const opusfe_node* nodes, const uint32_t* tid_of_node, const TypeRec* types,
                                                   const uint32_t* ph_def, const uint32_t* ph_prev, uint32_t tid, uint32_t at_def, uint32_t n_defs) {
    TypeRec content{}; content.kind = TY_Error; content.alias = OPUSFE_NONE;
    for (uint32_t guard = 0; guard < n_defs + 2; guard++) {
        if (tid < TP_COUNT) { content.kind = prim_kind(tid); return content; }
        uint32_t def = ph_def[tid - TP_COUNT];
        if (def == OPUSFE_NONE) return types[tid];               // structured type: its record never changes
        // the placeholder of a name is written by every `type` definition of that name in turn: what it holds now is what the
        // latest of them before `at_def` wrote (ph_prev links a definition to the previous one of the same name)
        for (uint32_t g2 = 0; def != OPUSFE_NONE && def >= at_def && g2 <= n_defs; g2++) def = ph_prev[tid_of_node[def] - TP_COUNT];
        if (def == OPUSFE_NONE || def >= at_def) { content.kind = TY_Error; return content; }   // not finalized yet (or itself)
        at_def = def;
        tid = tid_of_node[nodes[def].first_child];               // what that definition copied from
    }
    return content;
}

// The semantic errors finalize_definition (compile.rs:1002-1065) reports for definition d, in report order: UndefinedType
// for every unknown name among the type nodes it resolves (resolve_type, compile.rs:919-966, left to right = node order:
// arguments then return type; fields; proc arguments then result), then at most one error for a global's value
// (make_value compile.rs:983-1000 -> InvalidConstantExpression, else type_is_valid compile.rs:154-170 -> InvalidType).
// out == nullptr: count only.  Decided from the AST and the immutable structured type records, so the result is the same
// before and after k_sem_finalize wrote the placeholders.
__device__ __forceinline__ uint32_t sem_finalize_errors(const opusfe_node* __restrict__ nodes, const uint32_t* __restrict__ def_nodes, uint32_t n_defs, uint32_t d,
                                                        const uint32_t* __restrict__ tid_of_node, const uint8_t* __restrict__ undefined_flag,
                                                        const uint32_t* __restrict__ ph_def, const uint32_t* __restrict__ ph_prev, uint32_t n_nodes, const TypeRec* __restrict__ types, SemErr* out) {
    const uint32_t dn = def_nodes[d];
    const uint32_t dn_end = d + 1 < n_defs ? def_nodes[d + 1] : n_nodes;
    const opusfe_node n = nodes[dn];
    uint32_t nerr = 0;
    auto undefined_in = [&](uint32_t lo, uint32_t hi) {
        for (uint32_t y = lo; y < hi; y++)
            if (undefined_flag[y]) { if (out) out[nerr] = sem_err(SE_UndefinedType, nodes[y].low, nodes[y].high - nodes[y].low); nerr++; }
    };
    switch (n.kind) {
        case D_Function: case D_Extern: {
            uint32_t c = n.first_child;
            while (nodes[c].kind == X_NamedType) c = nodes[c].next;
            // the signature's nodes are the contiguous range up to the body
            undefined_in(dn + 1, nodes[c].next != OPUSFE_NONE ? nodes[c].next : dn_end);
            break;
        }
        case D_Variable: case D_Constant: {
            const uint32_t ty = n.first_child, val = nodes[ty].next;
            undefined_in(ty, val != OPUSFE_NONE ? val : dn_end);
            if (val != OPUSFE_NONE) {
                const opusfe_node v = nodes[val];
                const bool is_int = v.kind == E_Integer || (v.kind == E_Negate && nodes[v.first_child].kind == E_Integer);
                uint32_t kind = OPUSFE_NONE;
                if (!is_int) kind = SE_InvalidConstantExpression;
                else if (!kind_is_integral(sem_contents_at(nodes, tid_of_node, types, ph_def, ph_prev, tid_of_node[ty], dn, n_defs).kind)) kind = SE_InvalidType;
                if (kind != OPUSFE_NONE) { if (out) out[nerr] = sem_err(kind, v.low, v.high - v.low, OPUSFE_NONE, 0, kind == SE_InvalidType ? tid_of_node[ty] : OPUSFE_NONE); nerr++; }
            }
            break;
        }
        case D_Type: undefined_in(n.first_child, dn_end); break;
        default: break;
    }
    return nerr;
}

// finalize_definition (compile.rs:1002-1065), one lane per definition.  err[0] counts semantic errors
// (has_errors): with any, frontend.rs:145 skips IR generation.
__global__ void k_sem_finalize(const opusfe_node* __restrict__ nodes, const uint32_t* __restrict__ def_nodes, uint32_t n_defs,
                               const DefCnt* __restrict__ def_excl, const uint32_t* __restrict__ tid_of_node, const uint8_t* __restrict__ undefined_flag,
                               const uint32_t* __restrict__ ph_def, const uint32_t* __restrict__ ph_prev, uint32_t n_nodes, TypeRec* __restrict__ types, FuncRec* __restrict__ funcs,
                               GlobalRec* __restrict__ globals, uint32_t* __restrict__ err) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_defs) return;
    const uint32_t dn = def_nodes[d];
    const opusfe_node n = nodes[dn];
    const uint32_t nerr = sem_finalize_errors(nodes, def_nodes, n_defs, d, tid_of_node, undefined_flag, ph_def, ph_prev, n_nodes, types, nullptr);
    switch (n.kind) {
        case D_Function: case D_Extern: {
            FuncRec f{}; f.def_node = dn; f.name_tok = n.payload; f.flags = n.kind == D_Extern ? 1u : 0u;
            uint32_t c = n.first_child;
            f.first_arg = OPUSFE_NONE;
            while (nodes[c].kind == X_NamedType) { if (f.nargs == 0) f.first_arg = c; f.nargs++; c = nodes[c].next; }
            f.ret = tid_of_node[c];
            funcs[def_excl[d].funcs] = f;
            break;
        }
        case D_Variable: case D_Constant: {
            GlobalRec g{dn, tid_of_node[n.first_child], n.kind == D_Constant ? 1u : 0u, n.payload};
            globals[def_excl[d].globals] = g;
            break;
        }
        case D_Type: {
            TypeRec content = sem_contents_at(nodes, tid_of_node, types, ph_def, ph_prev, tid_of_node[n.first_child], dn, n_defs);
            content.alias = n.payload;
            types[tid_of_node[dn]] = content;
            break;
        }
        default: break;
    }
    if (nerr) atomicAdd(err, nerr);
}

// The ordered diagnostics of a unit whose finalize reported errors (the rare path: two more launches and a scan).
__global__ void k_sem_fin_count(const opusfe_node* __restrict__ nodes, const uint32_t* __restrict__ def_nodes, uint32_t n_defs, const uint32_t* __restrict__ tid_of_node,
                                const uint8_t* __restrict__ undefined_flag, const uint32_t* __restrict__ ph_def, const uint32_t* __restrict__ ph_prev, uint32_t n_nodes, const TypeRec* __restrict__ types,
                                uint32_t* __restrict__ cnt) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < n_defs) cnt[d] = sem_finalize_errors(nodes, def_nodes, n_defs, d, tid_of_node, undefined_flag, ph_def, ph_prev, n_nodes, types, nullptr);
}
__global__ void k_sem_fin_emit(const opusfe_node* __restrict__ nodes, const uint32_t* __restrict__ def_nodes, uint32_t n_defs, const uint32_t* __restrict__ tid_of_node,
                               const uint8_t* __restrict__ undefined_flag, const uint32_t* __restrict__ ph_def, const uint32_t* __restrict__ ph_prev, uint32_t n_nodes, const TypeRec* __restrict__ types,
                               const uint32_t* __restrict__ excl, SemErr* __restrict__ out) {
    uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < n_defs) sem_finalize_errors(nodes, def_nodes, n_defs, d, tid_of_node, undefined_flag, ph_def, ph_prev, n_nodes, types, out + excl[d]);
}

}  // namespace opusfe
