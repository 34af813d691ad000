// Shared host/device definitions of the opusfe library (MI355X / gfx950 only).
#pragma once
#include <stdint.h>
#include "../../include/opusfe.h"

#define OPUSFE_NONE 0xFFFFFFFFu

// Lanes that work on one unit (segment / function) in the sequential-machine kernels.  On the GPU a whole wave runs
// ONE unit with wave-uniform control flow: no divergence, and every memory instruction touches one cache line
// instead of 64.  The host emulation (tests/emu) runs one lane per unit.
// occupancy hint of the two sequential-machine kernels (waves per SIMD the register allocator must leave room for)
#ifndef OPUSFE_PARSE_WAVES
#define OPUSFE_PARSE_WAVES 7
#endif
#ifndef OPUSFE_LOWER_WAVES
#define OPUSFE_LOWER_WAVES 6
#endif
#ifdef OPUSFE_EMU
#define MACHINE_ATTR(w)
#else
#ifdef OPUSFE_FLATTEN
#define MACHINE_ATTR(w) __attribute__((flatten, amdgpu_waves_per_eu(w, w)))
#else
#define MACHINE_ATTR(w) __attribute__((amdgpu_waves_per_eu(w, w)))
#endif
#endif
// MACHINE_STATE: a register of a per-workgroup machine (parser, lowering): a static LDS variable on the GPU (one wave
// per workgroup runs one machine); the host emulation runs the lanes one after the other, so a plain static does there.
#ifdef OPUSFE_EMU
#define UNIT_LANES 1u
#define MACHINE_STATE inline static
#else
#define UNIT_LANES 64u
#define MACHINE_STATE inline static __shared__
#endif

#ifdef OPUSFE_EMU
__device__ __forceinline__ uint32_t uni(uint32_t v) { return v; }
#else
// value known to be identical in every lane -> SGPR (scalar branches, fewer VGPRs)
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
#endif

// ---- explicit global-memory accesses ------------------------------------------------------------
// A pointer that device code reads back from LDS (machine state) or from a by-value struct is a generic pointer, and
// the compiler then emits FLAT instructions for it (they tie up the LDS counter as well as the memory counter).  These
// helpers state that the pointee is in global memory: global_load / global_store, vectorised by the backend.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(OPUSFE_EMU)
#define OPUSFE_GLOBAL __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ T gld(const T* p) {
    if constexpr (sizeof(T) % 4 == 0 && sizeof(T) > 8) {
        union U { T v; uint32_t w[sizeof(T) / 4]; __device__ U() {} } u;
        const OPUSFE_GLOBAL uint32_t* g = (const OPUSFE_GLOBAL uint32_t*)p;
#pragma unroll
        for (unsigned i = 0; i < sizeof(T) / 4; i++) u.w[i] = g[i];
        return u.v;
    } else if constexpr (sizeof(T) == 8) {
        union U { T v; uint64_t w; __device__ U() {} } u; u.w = *(const OPUSFE_GLOBAL uint64_t*)p; return u.v;
    } else if constexpr (sizeof(T) == 4) {
        union U { T v; uint32_t w; __device__ U() {} } u; u.w = *(const OPUSFE_GLOBAL uint32_t*)p; return u.v;
    } else if constexpr (sizeof(T) == 2) {
        union U { T v; uint16_t w; __device__ U() {} } u; u.w = *(const OPUSFE_GLOBAL uint16_t*)p; return u.v;
    } else {
        static_assert(sizeof(T) == 1, "gld: unsupported size");
        union U { T v; uint8_t w; __device__ U() {} } u; u.w = *(const OPUSFE_GLOBAL uint8_t*)p; return u.v;
    }
}
template <class T> __device__ __forceinline__ void gst(T* p, const T& v) {
    if constexpr (sizeof(T) % 4 == 0 && sizeof(T) > 8) {
        union U { T v; uint32_t w[sizeof(T) / 4]; __device__ U() {} } u; u.v = v;
        OPUSFE_GLOBAL uint32_t* g = (OPUSFE_GLOBAL uint32_t*)p;
#pragma unroll
        for (unsigned i = 0; i < sizeof(T) / 4; i++) g[i] = u.w[i];
    } else if constexpr (sizeof(T) == 8) {
        union U { T v; uint64_t w; __device__ U() {} } u; u.v = v; *(OPUSFE_GLOBAL uint64_t*)p = u.w;
    } else if constexpr (sizeof(T) == 4) {
        union U { T v; uint32_t w; __device__ U() {} } u; u.v = v; *(OPUSFE_GLOBAL uint32_t*)p = u.w;
    } else if constexpr (sizeof(T) == 2) {
        union U { T v; uint16_t w; __device__ U() {} } u; u.v = v; *(OPUSFE_GLOBAL uint16_t*)p = u.w;
    } else {
        static_assert(sizeof(T) == 1, "gst: unsupported size");
        union U { T v; uint8_t w; __device__ U() {} } u; u.v = v; *(OPUSFE_GLOBAL uint8_t*)p = u.w;
    }
}
#elif defined(OPUSFE_EMU_COUNT)   // tests/emu only: how many global accesses of each kind does a run make
#include <map>
#include <string>
inline std::map<std::string, uint64_t>& emu_counts() { static std::map<std::string, uint64_t> m; return m; }
template <class T> T gld(const T* p) { emu_counts()[__PRETTY_FUNCTION__]++; return *p; }
template <class T> void gst(T* p, const T& v) { emu_counts()[__PRETTY_FUNCTION__]++; *p = v; }
#else
template <class T> __host__ __device__ __forceinline__ T gld(const T* p) { return *p; }
template <class T> __host__ __device__ __forceinline__ void gst(T* p, const T& v) { *p = v; }
#endif
#ifndef OPUSFE_GLOBAL
#define OPUSFE_GLOBAL
#endif
// Scalar loads.  The machines are wave-uniform, so what they read from the read-only tables (nodes, tokens, name table,
// records written by EARLIER kernels) belongs in SGPRs: a load through a constant-address-space pointer with a
// provably uniform address compiles to s_load_dwordxN (one instruction for a whole record, no VGPRs, no
// v_readfirstlane afterwards).  Never use it for memory that the running kernel itself writes: the scalar cache is
// not coherent with vector stores inside one kernel (it is invalidated between kernels).  Loads only: no scalar stores.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(OPUSFE_EMU)
#define OPUSFE_CONST __attribute__((address_space(4)))
template <class T> __device__ __forceinline__ T cld(const OPUSFE_CONST T* p) {
    if constexpr (sizeof(T) % 4 == 0) {
        union U { T v; uint32_t w[sizeof(T) / 4]; __device__ U() {} } u;
        const OPUSFE_CONST uint32_t* g = (const OPUSFE_CONST uint32_t*)p;
#pragma unroll
        for (unsigned i = 0; i < sizeof(T) / 4; i++) u.w[i] = g[i];
        return u.v;
    } else {
        static_assert(sizeof(T) == 1, "cld: unsupported size");
        union U { T v; uint8_t w; __device__ U() {} } u; u.w = *(const OPUSFE_CONST uint8_t*)p; return u.v;
    }
}
// read-only table read with scalar loads; the object (its base pointer) may live in LDS: pointer and index are made
// provably uniform with v_readfirstlane.  (The kernel-argument segment pointer is no alternative for helpers that are
// not inlined: llvm.amdgcn.kernarg.segment.ptr is null outside the kernel function itself.)
template <class T> struct STable {
    const T* p;
    __device__ __forceinline__ const OPUSFE_CONST T* base() const {
        const uint64_t a = (uint64_t)p;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
        return (const OPUSFE_CONST T*)(((uint64_t)hi << 32) | lo);
    }
    __device__ __forceinline__ T operator[](uint32_t i) const { return cld(base() + (uint32_t)__builtin_amdgcn_readfirstlane((int)i)); }
    __device__ __forceinline__ const T* ptr() const { return p; }     // generic pointer (vector accesses with a per-lane offset)
};
#elif !defined(OPUSFE_EMU)   // host pass of hipcc: declarations only have to parse
template <class T> struct STable { const T* p; __device__ T operator[](uint32_t) const { return T{}; } __device__ const T* ptr() const { return p; } };
#endif

// read-only table in global memory: t[i] is a global load that returns the record by value
template <class T> struct GTable {
    const T* p;
    GTable() = default;
    __host__ __device__ GTable(const T* q) : p(q) {}
    __host__ __device__ __forceinline__ T operator[](uint64_t i) const { return gld(p + i); }
    __host__ __device__ __forceinline__ const T* ptr() const { return p; }
};

// parse.rs:5-67, same order as the reference's enum Token.
enum TokKind : uint8_t {
    T_LeftParenthesis, T_RightParenthesis, T_LeftSquareBracket, T_RightSquareBracket, T_LeftBrace, T_RightBrace,
    T_Colon, T_Equals, T_EqualsEquals, T_LessThan, T_GreaterThan, T_LessThanEquals, T_GreaterThanEquals,
    T_Plus, T_Minus, T_Asterisk, T_Slash, T_Percent, T_At, T_Tilde, T_Dot, T_Comma, T_QuestionMark,
    T_Integer, T_Float, T_String, T_Identifier, T_Indent, T_Dedent,
    T_Var, T_Const, T_Return, T_If, T_Else, T_While, T_For, T_Break, T_Continue, T_Extern, T_Ref, T_Mut, T_Refs, T_Muts,
    T_Null, T_False, T_True, T_Type, T_Record, T_Is, T_In, T_And, T_Or, T_Not, T_Sizeof, T_Alignof, T_Import, T_Library,
    T_Proc, T_Operator, T_EndOfFile,
    T_Error = OPUSFE_TOK_ERROR
};

// parse.rs:352-360
enum PErrKind : uint8_t { PE_UnexpectedCharacter, PE_UnexpectedEndOfFile, PE_InvalidEscapeSequence, PE_UnexpectedToken,
                          PE_ExpectedFoundToken, PE_ExpectedIdentifier, PE_InvalidNumericSize };

// parse.rs:160-176
enum OpKind : uint8_t { OP_Plus, OP_Minus, OP_Times, OP_Divide, OP_Modulo, OP_Equals, OP_LessThan, OP_GreaterThan,
                        OP_LessThanEquals, OP_GreaterThanEquals, OP_Is, OP_And, OP_Or, OP_Not, OP_None = 0xFF };

// AST node kinds.  D_* = parse.rs:308-317 Definition, S_* = :281-291 Statement, E_* = :254-278
// Expression, Y_* = :294-304 Type; X_* are list/pair carriers that have no reference counterpart.
enum NodeKind : uint8_t {
    D_Function, D_Extern, D_Type, D_Variable, D_Constant, D_Import, D_Library,
    S_Expression, S_VariableDefinition, S_Assignment, S_Return, S_If, S_While, S_For, S_Break, S_Continue,
    E_Integer, E_Float, E_Null, E_Bool, E_String, E_Variable, E_VariableDefinition, E_Proc, E_CallOrIndex, E_NamedCall,
    E_Record, E_Operator, E_Reference, E_MutableReference, E_ArrayReference, E_MutableArrayReference, E_Dereference,
    E_Field, E_Cast, E_Negate, E_Sizeof, E_Alignof, E_Parentheses,
    Y_Name, Y_Null, Y_Reference, Y_MutableReference, Y_ArrayReference, Y_MutableArrayReference, Y_Proc, Y_Record, Y_Array,
    X_Block,       // [Box<Statement>]
    X_FieldInit,   // (name, Box<Expression>) of a record literal; payload = name token
    X_NamedType,   // (name, Box<Type>) of a record type / signature argument; payload = name token
    N_KindCount
};

// generate_ir.rs:8-56, same order.
enum IrOp : uint16_t { I_Integer, I_Float, I_Null, I_Bool, I_String, I_Sizeof, I_Alignof, I_Call, I_Allocate, I_Load, I_Store,
    I_Field, I_BuiltinField, I_IndexPointer, I_IndexArray, I_Add, I_Subtract, I_Multiply, I_Divide, I_Modulo, I_Equals,
    I_LessThan, I_GreaterThan, I_LessThanEquals, I_GreaterThanEquals, I_Function, I_GlobalVariable, I_GlobalConstant,
    I_Not, I_Negate, I_Cast, I_Return, I_Jump, I_Branch, I_Unreachable, I_Phi, I_BreakPlaceholder, I_Error };

// compile.rs:45-75 Type
enum TyKind : uint8_t { TY_Integer8, TY_Integer16, TY_Integer32, TY_Integer64, TY_Natural8, TY_Natural16, TY_Natural32,
    TY_Natural64, TY_Float32, TY_Float64, TY_Offset, TY_Size, TY_GenericInteger, TY_Generic, TY_Bool, TY_Null, TY_Pointer,
    TY_String, TY_Record, TY_Function, TY_Array, TY_Error };
enum PtrKind : uint8_t { PT_Reference, PT_Mutable, PT_Array, PT_ArrayMutable };
enum CastKind : uint8_t { CT_None, CT_Pointer, CT_PointerType, CT_Integer, CT_PointerToInteger, CT_IntegerToPointer,
                          CT_PointerToArray, CT_Error };

// status word written by kernels (max wins)
enum DevStatus : uint32_t { DS_OK = 0, DS_UNSUPPORTED_NONASCII = 1, DS_UNSUPPORTED_FLOAT = 2, DS_UNSUPPORTED_HANG = 3,
                            DS_UNSUPPORTED_RESYNC = 4, DS_UNSUPPORTED_SEMANTIC = 5, DS_INTERNAL_CAPACITY = 16,
                            DS_INTERNAL_SEGMENT = 17 };

// ---- group mode: a corpus of independent files packed into one unit (BASELINE config 5; api.hip opusfe_parse_group) ----
namespace opusfe {
// index of the last listed position <= p: the file that byte p belongs to (pos[0] == 0)
__host__ __device__ inline uint32_t pos_file(const uint32_t* pos, uint32_t n, uint32_t p) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (pos[mid] <= p) lo = mid + 1; else hi = mid; }
    return lo ? lo - 1 : 0;
}
// identifier hashes of file f are shifted by f * FILE_MIX (odd, so one name never has the same hash in two files of a group): the
// name tables of the semantic passes then scope every name to its file without knowing about files
constexpr uint32_t FILE_MIX = 0x9E3779B1u;
// per-file flags: FF_SYNTAX = the file has a first syntax error (a result, not a problem); FF_SINGLE = reasons why the packed pass
// cannot answer for the file exactly — it is then run alone through the single-source pipeline
enum : uint32_t { FF_EMPTY = 1u, FF_INDENTED = 2u, FF_UNSUP = 16u, FF_RESYNC = 32u, FF_INTERNAL = 64u, FF_FINERR = 256u, FF_SYNTAX = 512u,
                  FF_SINGLE = FF_EMPTY | FF_INDENTED | FF_UNSUP | FF_RESYNC | FF_INTERNAL | FF_FINERR };
}  // namespace opusfe
