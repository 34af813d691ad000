// GPU lexer for gfx950.  Restates the lexer embedded in the reference's Parser
// (parse.rs:415-721: advance/peek, skip_whitespace, parse_token) as data-parallel passes:
//
//   1. k_ctx_chunk   : every 64-byte chunk -> composed transition function of the 6-state context
//                      automaton {N, S, E, X1, X2, C}; tile aggregates.
//   2. k_ctx_tiles   : single-block scan of the tile aggregates (function composition).
//   3. k_lex_count   : per chunk: start state from a block scan of the chunk functions; lex the chunk
//                      sequentially (tokens that START in it), count tokens / val64 / string bytes.
//   4. scans         : tile totals -> offsets.
//   5. k_lex_emit    : same walk, writes raw (dent-free) tokens + literal pools.
//   6. k_tok_delta   : bracket-depth delta per raw token (derivation of parser-controlled
//                      `ignore_dents`, parse.rs:874,879,890,928,939,987,990,1054,1069).
//   7. k_gap         : per inter-token gap: does it cross a line in dent-sensitive mode and how many
//                      TABs follow the last '\n' (parse.rs:427-472).
//   8. k_final       : Indent/Dedent synthesis (parse.rs:495-501) + top-level segment boundaries.
#pragma once
#include <hip/hip_runtime.h>
#include "defs.h"
#include "scan.h"

namespace opusfe {

constexpr int LEX_CHUNK = 64;                 // bytes lexed sequentially by one lane
constexpr int LEX_THREADS = 256;
constexpr int LEX_TILE = LEX_CHUNK * LEX_THREADS;  // 16 KiB per workgroup

// ---- context automaton ---------------------------------------------------------------------
enum CtxState : uint32_t { CS_N = 0, CS_S = 1, CS_E = 2, CS_X1 = 3, CS_X2 = 4, CS_C = 5 };
#define CTXFN(n, s, e, x1, x2, c) ((uint32_t)(n) | ((uint32_t)(s) << 3) | ((uint32_t)(e) << 6) | ((uint32_t)(x1) << 9) | ((uint32_t)(x2) << 12) | ((uint32_t)(c) << 15))
constexpr uint32_t CTX_IDENTITY = CTXFN(0, 1, 2, 3, 4, 5);
constexpr uint32_t CTX_OTHER = CTXFN(CS_N, CS_S, CS_S, CS_X2, CS_S, CS_C);
constexpr uint32_t CTX_QUOTE = CTXFN(CS_S, CS_N, CS_S, CS_X2, CS_S, CS_C);
constexpr uint32_t CTX_BSL = CTXFN(CS_N, CS_E, CS_S, CS_X2, CS_S, CS_C);
constexpr uint32_t CTX_SEMI = CTXFN(CS_C, CS_S, CS_S, CS_X2, CS_S, CS_C);
constexpr uint32_t CTX_NL = CTXFN(CS_N, CS_S, CS_S, CS_X2, CS_S, CS_N);
constexpr uint32_t CTX_X = CTXFN(CS_N, CS_S, CS_X1, CS_X2, CS_S, CS_C);

__host__ __device__ __forceinline__ uint32_t ctx_of_byte(uint32_t b) {
    return b == '"' ? CTX_QUOTE : b == '\\' ? CTX_BSL : b == ';' ? CTX_SEMI : b == '\n' ? CTX_NL : b == 'x' ? CTX_X : CTX_OTHER;
}
__host__ __device__ __forceinline__ uint32_t ctx_apply(uint32_t f, uint32_t s) { return (f >> (3 * s)) & 7u; }
// first f, then g
__host__ __device__ __forceinline__ uint32_t ctx_compose(uint32_t f, uint32_t g) {
    uint32_t r = 0;
#pragma unroll
    for (int s = 0; s < 6; s++) r |= ctx_apply(g, ctx_apply(f, s)) << (3 * s);
    return r;
}
// true when some start state currently maps to E / X1 / X2 (010, 011, 100): only then does an
// ordinary byte change the function.
__host__ __device__ __forceinline__ bool ctx_has_transient(uint32_t f) {
    const uint32_t m0 = 0x9249u;   // bit 0 of each 3-bit lane (6 lanes)
    uint32_t b0 = f & m0, b1 = (f >> 1) & m0, b2 = (f >> 2) & m0;
    return (b1 | (b2 & ~b0)) != 0;
}
struct OpCtx { __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return ctx_compose(a, b); } };

// Resync points (sorted byte positions, normally none): after some lexical errors the reference resumes lexing in the
// middle of what the context automaton sees as a string (parse.rs:683-717: a bad escape returns Err with the position just
// behind it; parse.rs:563-581: an invalid float suffix swallows two characters, whatever they are).  If the parser then
// drops that Err token (Q6/Q7) the bytes behind it are code.  The host finds such a position R from a first pass
// (api.hip parse_with_resync) and lexes again with R in this list: at R the context is N and a token starts.
// Group mode (a corpus of independent files packed into one buffer, BASELINE config 5; api.hip opusfe_parse_group): the same list
// holds the START of every file (pos[0] == 0, every start a multiple of LEX_CHUNK, '\n' between the files) — a file start is exactly
// "context N and a token boundary" — with `files` set; `fstatus` then receives the unsupported-input records per file (2 words each).
// Every file is lexed with ITS OWN end as the end of the source (`flen`), so end-of-file behaviour (parse.rs:461-464, 574-580, 701, 712) is the file's.
struct Resync { const uint32_t* pos; uint32_t n; uint32_t files; uint32_t* fstatus; const uint32_t* flen; };

constexpr uint32_t CTX_RESET = 0;   // every state -> N
// smallest listed position >= p, else 0xFFFFFFFF
__host__ __device__ __forceinline__ uint32_t rs_next(const Resync& rs, uint32_t p) {
    uint32_t lo = 0, hi = rs.n;
    while (lo < hi) { uint32_t mid = (lo + hi) / 2; if (rs.pos[mid] < p) lo = mid + 1; else hi = mid; }
    return lo < rs.n ? rs.pos[lo] : 0xFFFFFFFFu;
}
__host__ __device__ __forceinline__ bool rs_has(const Resync& rs, uint32_t p) { return rs.n != 0 && rs_next(rs, p) == p; }


__global__ __launch_bounds__(LEX_THREADS) void k_ctx_chunk(const uint8_t* __restrict__ src, uint32_t n, uint32_t* __restrict__ chunk_fn, uint32_t* __restrict__ tile_fn, Resync rs) {
    __shared__ uint32_t lds[4];
    uint32_t chunk = blockIdx.x * LEX_THREADS + threadIdx.x;
    uint64_t base = (uint64_t)chunk * LEX_CHUNK;
    uint32_t f = CTX_IDENTITY;
    if (base < n) {
        uint32_t end = (uint32_t)((base + LEX_CHUNK < n) ? base + LEX_CHUNK : n);
        uint32_t nr = rs.n ? rs_next(rs, (uint32_t)base) : 0xFFFFFFFFu;
        if (nr < end) {   // a resync point in this chunk: everything before it is forgotten
            for (uint32_t i = (uint32_t)base; i < end; i++) {
                if (i == nr) { f = CTX_RESET; nr = rs_next(rs, i + 1); }
                f = ctx_compose(f, ctx_of_byte(src[i]));
            }
        } else if (end - (uint32_t)base == LEX_CHUNK) {
            const uint4* p = reinterpret_cast<const uint4*>(src + base);
#pragma unroll
            for (int q = 0; q < LEX_CHUNK / 16; q++) {
                uint4 v = p[q];
                uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        uint32_t g = ctx_of_byte((w[k] >> (8 * j)) & 0xFF);
                        if (g != CTX_OTHER || ctx_has_transient(f)) f = ctx_compose(f, g);
                    }
                }
            }
        } else {
            for (uint32_t i = (uint32_t)base; i < end; i++) f = ctx_compose(f, ctx_of_byte(src[i]));
        }
        if (chunk_fn) chunk_fn[chunk] = f;   // (null: only the tile aggregates are wanted, opusfe_shard_ctx_static)
    }
    uint32_t total;
    block_inclusive_scan(f, OpCtx(), lds, &total);
    if (threadIdx.x == 0) tile_fn[blockIdx.x] = total;
}

// tile_fn[i] (composed function of tile i)  ->  tile_state[i] = context state at the start of tile i
__global__ __launch_bounds__(SCAN_THREADS) void k_ctx_tiles(const uint32_t* __restrict__ tile_fn, uint8_t* __restrict__ tile_state, uint32_t ntiles,
                                                            uint32_t start_state, uint32_t* __restrict__ total_fn) {
    __shared__ uint32_t lds[4];
    __shared__ uint32_t wave_last[4];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = CTX_IDENTITY;
    __syncthreads();
    for (uint32_t base = 0; base < ntiles; base += SCAN_TILE) {
        uint32_t b = base + threadIdx.x * SCAN_ITEMS;
        uint32_t v[SCAN_ITEMS];
        uint32_t acc = CTX_IDENTITY;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) { v[i] = (b + i < ntiles) ? tile_fn[b + i] : CTX_IDENTITY; acc = ctx_compose(acc, v[i]); }
        uint32_t total;
        uint32_t incl = block_inclusive_scan(acc, OpCtx(), lds, &total);
        uint32_t prev = shfl_up_t(incl, 1);
        if ((threadIdx.x & 63) == 63) wave_last[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t excl;
        if (threadIdx.x == 0) excl = carry_s;
        else if ((threadIdx.x & 63) == 0) excl = ctx_compose(carry_s, wave_last[(threadIdx.x >> 6) - 1]);
        else excl = ctx_compose(carry_s, prev);
        uint32_t run = excl;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) { if (tile_state && b + i < ntiles) tile_state[b + i] = (uint8_t)ctx_apply(run, start_state); run = ctx_compose(run, v[i]); }
        __syncthreads();
        if (threadIdx.x == 0) carry_s = ctx_compose(carry_s, total);
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_fn) *total_fn = carry_s;   // composed function of the whole byte range (the shard carry)
}

// ---- one token ----------------------------------------------------------------------------------
struct LexTok {
    uint8_t kind, flags; uint16_t aux;
    uint32_t low, end;        // [low, end): end = where lexing resumes
    uint32_t nval;            // val64 entries (0..2)
    uint64_t v0, v1;
    uint32_t payload;         // inline integer / identifier hash
    uint32_t strbytes;        // pool bytes incl. the 4-byte length prefix, padded to 4
};

// records an input the reference cannot be followed on (code: max wins) and where it starts (min wins):
// the host reports OPUSFE_UNSUPPORTED only if no syntax error is found before that position.
__device__ __forceinline__ void lx_unsupported(uint32_t* status, uint32_t code, uint32_t pos) { atomicMax(status, code); atomicMin(status + 1, pos); }

__device__ __forceinline__ int lx_at(const uint8_t* __restrict__ s, uint32_t n, uint32_t i) { return i < n ? (int)s[i] : -1; }
__device__ __forceinline__ bool lx_is_ws(int c) { return c == ' ' || (c >= 9 && c <= 13); }
__device__ __forceinline__ bool lx_is_digit(int c) { return c >= '0' && c <= '9'; }
__device__ __forceinline__ bool lx_is_alpha(int c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }
__device__ __forceinline__ bool lx_is_identc(int c) { return lx_is_alpha(c) || lx_is_digit(c) || c == '\\'; }
// bytes over which the lookback of a chunk must continue to find a certain token boundary
__device__ __forceinline__ bool lx_is_run(int c) { return lx_is_identc(c) || c == '.' || c == '=' || c == '<' || c == '>'; }
__device__ __forceinline__ int lx_hex(int c) { return lx_is_digit(c) ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : (c >= 'A' && c <= 'F') ? c - 'A' + 10 : -1; }
__device__ __forceinline__ uint32_t lx_utf8_len(int b0) { return b0 < 0x80 ? 1 : b0 >= 0xF0 ? 4 : b0 >= 0xE0 ? 3 : b0 >= 0xC0 ? 2 : 1; }

// One UTF-8 scalar at p (the source is valid UTF-8: frontend.rs:85 read_to_string; a truncated tail is clamped to n)
__device__ __forceinline__ uint32_t lx_decode(const uint8_t* __restrict__ s, uint32_t n, uint32_t p, uint32_t* len) {
    uint32_t b0 = s[p];
    uint32_t L = lx_utf8_len((int)b0);
    if (p + L > n) L = n - p;
    uint32_t c = L == 4 ? (b0 & 0x07) : L == 3 ? (b0 & 0x0F) : L == 2 ? (b0 & 0x1F) : b0;
    for (uint32_t i = 1; i < L; i++) c = (c << 6) | (s[p + i] & 0x3Fu);
    *len = L;
    return c;
}
// char::is_whitespace = Unicode White_Space: 25 code points, unchanged since Unicode 6.3, hence the same in every Rust release
// that builds the reference (unlike is_alphanumeric, whose tables are not pinned: identifiers stay ASCII here)
__device__ __forceinline__ bool lx_unicode_ws(uint32_t c) {
    return c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) || c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}

__device__ __forceinline__ uint64_t kw_pack(const char* s) {
    uint64_t v = 0;
    for (int i = 0; s[i] && i < 8; i++) v |= (uint64_t)(uint8_t)s[i] << (8 * i);
    return v;
}
__device__ __forceinline__ uint8_t lx_keyword(uint64_t w, uint32_t len) {
    // parse.rs:640-671
    switch (len) {
        case 2:
            if (w == kw_pack("if")) return T_If; if (w == kw_pack("is")) return T_Is; if (w == kw_pack("in")) return T_In;
            if (w == kw_pack("or")) return T_Or; break;
        case 3:
            if (w == kw_pack("var")) return T_Var; if (w == kw_pack("for")) return T_For; if (w == kw_pack("ref")) return T_Ref;
            if (w == kw_pack("mut")) return T_Mut; if (w == kw_pack("and")) return T_And; if (w == kw_pack("not")) return T_Not; break;
        case 4:
            if (w == kw_pack("else")) return T_Else; if (w == kw_pack("refs")) return T_Refs; if (w == kw_pack("muts")) return T_Muts;
            if (w == kw_pack("null")) return T_Null; if (w == kw_pack("true")) return T_True; if (w == kw_pack("type")) return T_Type;
            if (w == kw_pack("proc")) return T_Proc; break;
        case 5:
            if (w == kw_pack("const")) return T_Const; if (w == kw_pack("while")) return T_While; if (w == kw_pack("break")) return T_Break;
            if (w == kw_pack("false")) return T_False; break;
        case 6:
            if (w == kw_pack("return")) return T_Return; if (w == kw_pack("extern")) return T_Extern; if (w == kw_pack("record")) return T_Record;
            if (w == kw_pack("sizeof")) return T_Sizeof; if (w == kw_pack("import")) return T_Import; break;
        case 7:
            if (w == kw_pack("alignof")) return T_Alignof; if (w == kw_pack("library")) return T_Library; break;
        case 8:
            if (w == kw_pack("continue")) return T_Continue; if (w == kw_pack("operator")) return T_Operator; break;
        default: break;
    }
    return T_Identifier;
}

__device__ __forceinline__ void lx_error(LexTok& o, uint8_t perr, uint32_t span_low, uint32_t span_len, uint64_t payload, uint32_t end) {
    o.kind = T_Error; o.aux = perr; o.end = end; o.nval = 2;
    o.v0 = (uint64_t)span_low | ((uint64_t)span_len << 32);
    o.v1 = payload;
}

// From a position inside a string body (context S at p): where the context automaton returns to N.
// Used for error tokens so that the token stream stays consistent with the context scan; the
// reference resumes elsewhere (inside the string body), hence OPUSFE_TF_RESYNC on such tokens.
__device__ __forceinline__ uint32_t lx_string_end(const uint8_t* __restrict__ s, uint32_t n, uint32_t p) {
    uint32_t st = CS_S;
    while (p < n && st != CS_N) { st = ctx_apply(ctx_of_byte(s[p]), st); p++; }
    return p;
}

// Lexes the token starting at `pos` (s[pos] is not whitespace, not ';', pos < n, context N).
// parse.rs:503-720.  `strdst` (may be null) receives [u32 n][bytes] of a String token.
// ---- correctly rounded decimal -> f64 for literals outside the fast path ---------------------------
// parse.rs:562 uses str::parse::<f64> (correctly rounded, ties to even).  A literal here is DIGITS '.' DIGITS, i.e. the
// rational W / 10^frac.  Up to BIGDEC_DIGITS significant digits are converted exactly with fixed-size big integers:
// Q = floor(W * 2^E / 10^frac) with E chosen so that Q has 55..56 bits, then 53 bits are kept and the rest plus the
// division remainder decide the rounding.  (Rare path: the arrays live in scratch memory.)
// 300 digits: W, D < 10^300 < 2^997, the scaled operands stay below 2^1053 (34 limbs), and every value is in f64's normal
// range (10^-300 .. 10^300), so no subnormal / overflow rounding is needed
enum { BIGDEC_LIMBS = 34, BIGDEC_DIGITS = 300 };
struct BigDec { uint32_t w[BIGDEC_LIMBS]; };
__device__ __forceinline__ void bd_zero(BigDec& a) {
    _Pragma("unroll 1") for (int i = 0; i < BIGDEC_LIMBS; i++) a.w[i] = 0;
}
__device__ __forceinline__ void bd_mul10_add(BigDec& a, uint32_t d) {
    uint64_t c = d;
    _Pragma("unroll 1") for (int i = 0; i < BIGDEC_LIMBS; i++) { c += (uint64_t)a.w[i] * 10u; a.w[i] = (uint32_t)c; c >>= 32; }
}
__device__ __forceinline__ int bd_bits(const BigDec& a) {
    _Pragma("unroll 1") for (int i = BIGDEC_LIMBS - 1; i >= 0; i--) if (a.w[i]) return i * 32 + 32 - __builtin_clz(a.w[i]);
    return 0;
}
__device__ __forceinline__ void bd_shl(BigDec& a, int k) {
    const int ws = k >> 5, bs = k & 31;
    _Pragma("unroll 1") for (int i = BIGDEC_LIMBS - 1; i >= 0; i--) {
        uint32_t v = 0;
        if (i - ws >= 0) v = a.w[i - ws] << bs;
        if (bs && i - ws - 1 >= 0) v |= a.w[i - ws - 1] >> (32 - bs);
        a.w[i] = v;
    }
}
__device__ __forceinline__ void bd_shr1(BigDec& a) {
    _Pragma("unroll 1") for (int i = 0; i < BIGDEC_LIMBS; i++) a.w[i] = (a.w[i] >> 1) | (i + 1 < BIGDEC_LIMBS ? a.w[i + 1] << 31 : 0u);
}
__device__ __forceinline__ bool bd_ge(const BigDec& a, const BigDec& b) {
    _Pragma("unroll 1") for (int i = BIGDEC_LIMBS - 1; i >= 0; i--) if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
    return true;
}
__device__ __forceinline__ void bd_sub(BigDec& a, const BigDec& b) {
    uint64_t br = 0;
    _Pragma("unroll 1") for (int i = 0; i < BIGDEC_LIMBS; i++) { uint64_t d = (uint64_t)a.w[i] - b.w[i] - br; a.w[i] = (uint32_t)d; br = (d >> 63) & 1; }
}
__device__ __forceinline__ bool bd_is_zero(const BigDec& a) {
    _Pragma("unroll 1") for (int i = 0; i < BIGDEC_LIMBS; i++) if (a.w[i]) return false;
    return true;
}
// digits = the literal's digits without the '.', frac of them behind the point; false when there are too many
__device__ __noinline__ bool lx_decimal_to_f64(const uint8_t* __restrict__ s, uint32_t int_lo, uint32_t int_hi, uint32_t frac_lo, uint32_t frac_hi, double* out) {
    while (frac_hi > frac_lo && s[frac_hi - 1] == '0') frac_hi--;          // trailing zeros of the fraction
    while (int_lo < int_hi && s[int_lo] == '0') int_lo++;                  // leading zeros
    uint32_t lead = frac_lo;                                                // 0.000ddd: zeros that are not significant
    if (int_lo == int_hi) while (lead < frac_hi && s[lead] == '0') lead++;
    const uint32_t sig = (int_hi - int_lo) + (frac_hi - lead), frac = frac_hi - frac_lo;
    if (sig == 0) { *out = 0.0; return true; }
    if (sig > BIGDEC_DIGITS || frac > BIGDEC_DIGITS) return false;
    BigDec W, D; bd_zero(W); bd_zero(D);
    for (uint32_t q = int_lo; q < int_hi; q++) bd_mul10_add(W, (uint32_t)(s[q] - '0'));
    for (uint32_t q = frac_lo; q < frac_hi; q++) bd_mul10_add(W, (uint32_t)(s[q] - '0'));
    D.w[0] = 1;
    for (uint32_t k = 0; k < frac; k++) bd_mul10_add(D, 0);
    // scale so that the quotient has 55 or 56 bits
    const int E = 55 - (bd_bits(W) - bd_bits(D));
    if (E >= 0) bd_shl(W, E); else bd_shl(D, -E);
    uint64_t Q = 0;
    bd_shl(D, 56);
    for (int bit = 56; bit >= 0; bit--) {
        if (bd_ge(W, D)) { bd_sub(W, D); Q |= 1ull << bit; }
        bd_shr1(D);
    }
    const bool rem = !bd_is_zero(W);
    const int qb = 64 - __builtin_clzll(Q), sh = qb - 53;                   // qb is 55 or 56
    uint64_t m = Q >> sh;
    const bool round = (Q >> (sh - 1)) & 1, sticky = (Q & ((1ull << (sh - 1)) - 1)) != 0 || rem;
    int e2 = sh - E;                                                          // value = m * 2^e2
    if (round && (sticky || (m & 1))) { m++; if (m == (1ull << 53)) { m >>= 1; e2++; } }
    const uint64_t bits = ((uint64_t)(e2 + 52 + 1023) << 52) | (m & ((1ull << 52) - 1));   // normal range by the digit limits
    *out = __longlong_as_double((long long)bits);
    return true;
}

__device__ void lex_one(const uint8_t* __restrict__ s, uint32_t n, uint32_t pos, LexTok& o, uint8_t* strdst, uint32_t* status, const Resync& rs) {
    o.kind = T_EndOfFile; o.flags = 0; o.aux = 0; o.low = pos; o.end = pos + 1; o.nval = 0; o.v0 = o.v1 = 0; o.payload = 0; o.strbytes = 0;
    int c = s[pos];
    switch (c) {
        case '(': o.kind = T_LeftParenthesis; return;
        case ')': o.kind = T_RightParenthesis; return;
        case '[': o.kind = T_LeftSquareBracket; return;
        case ']': o.kind = T_RightSquareBracket; return;
        case '{': o.kind = T_LeftBrace; return;
        case '}': o.kind = T_RightBrace; return;
        case ':': o.kind = T_Colon; return;
        case '=': if (lx_at(s, n, pos + 1) == '=') { o.kind = T_EqualsEquals; o.end = pos + 2; } else o.kind = T_Equals; return;
        case '<': if (lx_at(s, n, pos + 1) == '=') { o.kind = T_LessThanEquals; o.end = pos + 2; } else o.kind = T_LessThan; return;
        case '>': if (lx_at(s, n, pos + 1) == '=') { o.kind = T_GreaterThanEquals; o.end = pos + 2; } else o.kind = T_GreaterThan; return;
        case '+': o.kind = T_Plus; return;
        case '-': o.kind = T_Minus; return;
        case '*': o.kind = T_Asterisk; return;
        case '/': o.kind = T_Slash; return;
        case '%': o.kind = T_Percent; return;
        case '@': o.kind = T_At; return;
        case '~': o.kind = T_Tilde; return;
        case '.': o.kind = T_Dot; return;
        case ',': o.kind = T_Comma; return;
        case '?': o.kind = T_QuestionMark; return;
        default: break;
    }
    if (lx_is_digit(c)) {
        // parse.rs:549-628
        uint64_t v = (uint64_t)(c - '0');
        uint32_t p = pos + 1;
        for (;;) { int d = lx_at(s, n, p); if (!lx_is_digit(d)) break; v = v * 10 + (uint64_t)(d - '0'); p++; }
        if (lx_at(s, n, p) == '.') {
            p++;
            // mantissa as an exact integer (Clinger fast path: w < 2^53, 10^k exact for k <= 22)
            uint64_t w = 0; bool exact = true; uint32_t frac = 0;
            for (uint32_t q = pos; q < p - 1; q++) { uint64_t d = (uint64_t)(s[q] - '0'); if (w > (0x1FFFFFFFFFFFFFull - d) / 10) exact = false; else w = w * 10 + d; }
            for (;;) {
                int d = lx_at(s, n, p); if (!lx_is_digit(d)) break;
                uint64_t dd = (uint64_t)(d - '0');
                if (w > (0x1FFFFFFFFFFFFFull - dd) / 10) exact = false; else w = w * 10 + dd;
                frac++; p++;
            }
            if (frac > 22) exact = false;
            double f = 0.0;
            if (exact) {
                const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
                f = (double)w / P10[frac];
            } else if (!lx_decimal_to_f64(s, pos, p - 1 - frac, p - frac, p, &f)) {
                lx_unsupported(status, DS_UNSUPPORTED_FLOAT, pos);      // more than 300 significant digits
            }
            o.kind = T_Float; o.nval = 1; o.v0 = (uint64_t)__double_as_longlong(f);
            if (lx_at(s, n, p) == 'f') {
                // parse.rs:563-581: after 'f' two advance()s regardless, over characters; the size arithmetic is on code points
                uint32_t la = 0, lb = 0;
                int a = -1, b = -1;
                if (p + 1 < n) a = (int)lx_decode(s, n, p + 1, &la);
                if (a >= 0 && p + 1 + la < n) b = (int)lx_decode(s, n, p + 1 + la, &lb);
                const uint32_t e2 = p + 1 + la, e3 = e2 + lb;      // behind one / two swallowed characters
                if (a == '3' && b == '2') { o.aux = 32; p += 3; }
                else if (a == '6' && b == '4') { o.aux = 64; p += 3; }
                else if (a >= 0 && b < 0) { lx_error(o, PE_InvalidNumericSize, pos, e2 - pos, (uint32_t)a - (uint32_t)'0', e2); if (!lx_is_run(a) && !rs_has(rs, e2)) o.flags |= OPUSFE_TF_RESYNC; return; }
                else if (a >= 0 && b >= 0) { lx_error(o, PE_InvalidNumericSize, pos, e3 - pos, ((uint32_t)a - (uint32_t)'0') * 10u + ((uint32_t)b - (uint32_t)'0'), e3); if ((!lx_is_run(a) || !lx_is_run(b)) && !rs_has(rs, e3)) o.flags |= OPUSFE_TF_RESYNC; return; }
                else { lx_unsupported(status, DS_UNSUPPORTED_HANG, pos); p += 1; }   // `<float>f<EOF>`: the reference loops forever (Q4)
            }
            o.end = p;
            return;
        }
        o.kind = T_Integer; o.end = p;
        int sfx = lx_at(s, n, p);
        if (sfx == 's') o.aux = 0x8000;                       // (false, 0), not consumed (Q1)
        else if (sfx == 'o') o.aux = 0x8000 | 0x100;          // (true, 0)
        else if (sfx == 'i' || sfx == 'n') {
            uint16_t sg = sfx == 'i' ? 0x100 : 0;
            int a = lx_at(s, n, p + 1), b = a < 0 ? -1 : lx_at(s, n, p + 2);
            if (a == '8') { o.aux = 0x8000 | sg | 8; o.end = p + 2; }
            else if (a == '1' && b == '6') { o.aux = 0x8000 | sg | 16; o.end = p + 3; }
            else if (a == '3' && b == '2') { o.aux = 0x8000 | sg | 32; o.end = p + 3; }
            else if (a == '6' && b == '4') { o.aux = 0x8000 | sg | 64; o.end = p + 3; }
            else if (lx_is_digit(a) && lx_is_digit(b)) { lx_error(o, PE_InvalidNumericSize, p + 1, 2, (uint32_t)((a - '0') * 10 + b - '0'), p + 3); return; }
            else if (lx_is_digit(a)) { lx_error(o, PE_InvalidNumericSize, p + 1, 1, (uint32_t)(a - '0'), p + 2); return; }
            else { o.aux = 0x8000 | sg | 32; o.end = p + 1; }   // rewind to just after the letter (Q2)
        }
        if (v <= 0xFFFFFFFFull) { o.flags |= OPUSFE_TF_INLINE; o.payload = (uint32_t)v; }
        else { o.nval = 1; o.v0 = v; }
        return;
    }
    if (lx_is_alpha(c)) {
        // parse.rs:629-673
        uint32_t p = pos + 1;
        uint64_t w = (uint64_t)c;
        uint32_t h = (2166136261u ^ (uint32_t)c) * 16777619u;
        for (;;) {
            int d = lx_at(s, n, p);
            if (d >= 0x80) {   // White_Space is never alphanumeric: the identifier ends; anything else needs the unpinned tables
                uint32_t L;
                if (!lx_unicode_ws(lx_decode(s, n, p, &L))) lx_unsupported(status, DS_UNSUPPORTED_NONASCII, pos);   // from the token's first byte: which token this is depends on the tables
                break;
            }
            if (!lx_is_identc(d)) break;
            if (p - pos < 8) w |= (uint64_t)d << (8 * (p - pos));
            h = (h ^ (uint32_t)d) * 16777619u;
            p++;
        }
        o.end = p;
        o.kind = lx_keyword(w, p - pos);
        if (o.kind == T_Identifier) o.payload = h;
        return;
    }
    if (c == '"') {
        // parse.rs:674-717.  A string that ends in a lexical error still reserves the pool bytes it wrote
        // (strbytes) so that the emit pass, which decodes straight into the pool, stays inside its slice.
        uint32_t p = pos + 1;
        uint32_t nout = 0;
        uint8_t* dst = strdst ? strdst + 4 : nullptr;
        for (;;) {
            int ch = lx_at(s, n, p);
            if (ch < 0) { lx_error(o, PE_UnexpectedEndOfFile, n, 0, 0, n); o.strbytes = 4 + ((nout + 3) & ~3u); return; }
            if (ch == '"') { p++; break; }
            if (ch == '\\') {
                uint32_t escape_position = p + 1;
                int e = lx_at(s, n, p + 1);
                int val = -1;
                switch (e) {
                    case '\\': val = '\\'; break; case '"': val = '"'; break; case 'n': val = '\n'; break;
                    case 'r': val = '\r'; break; case 't': val = '\t'; break; case '0': val = 0; break;
                    default: break;
                }
                if (val >= 0) { if (dst) dst[nout] = (uint8_t)val; nout++; p += 2; continue; }
                if (e == 'x') {
                    // two advance()s over *chars* (parse.rs:691-692)
                    int hc = lx_at(s, n, p + 2);
                    uint32_t q = p + 2 + (hc < 0 ? 0 : lx_utf8_len(hc));
                    int lc = hc < 0 ? -1 : lx_at(s, n, q);
                    uint32_t q2 = q + (lc < 0 ? 0 : lx_utf8_len(lc));
                    int hv = hc < 0 ? -1 : lx_hex(hc), lv = lc < 0 ? -1 : lx_hex(lc);
                    if (hv >= 0 && lv >= 0) { if (dst) dst[nout] = (uint8_t)(hv * 16 + lv); nout++; p = q2; continue; }
                    if (q2 > n) q2 = n;
                    if (rs_has(rs, q2)) lx_error(o, PE_InvalidEscapeSequence, escape_position, q2 - escape_position, 0, q2);   // second pass: lexing resumes at q2
                    else { lx_error(o, PE_InvalidEscapeSequence, escape_position, q2 - escape_position, 0, lx_string_end(s, n, p)); o.flags |= OPUSFE_TF_RESYNC; }
                    o.strbytes = 4 + ((nout + 3) & ~3u);
                    return;
                }
                if (e < 0) { lx_error(o, PE_UnexpectedEndOfFile, n, 0, 0, n); o.strbytes = 4 + ((nout + 3) & ~3u); return; }
                uint32_t el = lx_utf8_len(e);
                if (escape_position + el > n) el = n - escape_position;
                if (rs_has(rs, escape_position + el)) lx_error(o, PE_InvalidEscapeSequence, escape_position, el, 0, escape_position + el);
                else { lx_error(o, PE_InvalidEscapeSequence, escape_position, el, 0, lx_string_end(s, n, p)); o.flags |= OPUSFE_TF_RESYNC; }
                o.strbytes = 4 + ((nout + 3) & ~3u);
                return;
            }
            if (dst) dst[nout] = (uint8_t)ch;
            nout++; p++;
        }
        o.kind = T_String; o.end = p;
        o.strbytes = 4 + ((nout + 3) & ~3u);
        if (strdst) { *reinterpret_cast<uint32_t*>(strdst) = nout; for (uint32_t k = nout; k < ((nout + 3) & ~3u); k++) dst[k] = 0; }
        return;
    }
    // parse.rs:718: any other character, ASCII or not, is an UnexpectedCharacter error that spans the character
    uint32_t L = 1, cp = (uint32_t)c;
    if (c >= 0x80) cp = lx_decode(s, n, pos, &L);
    lx_error(o, PE_UnexpectedCharacter, pos, L, (uint64_t)cp, pos + L);
}

// Is `cbeg` inside the two characters swallowed by an invalid float suffix whose error token ends at the resync point `nr`?
// (`<digits>.<digits>f a b`: the `f` sits at nr - 1 - len(a) - len(b), or at nr - 1 - len(a) when the file ends behind a.)
__device__ __forceinline__ bool lx_inside_suffix_error(const uint8_t* __restrict__ s, uint32_t n, uint32_t cbeg, uint32_t nr) {
    for (uint32_t fp = nr >= 9 ? nr - 9 : 0; fp + 2 <= nr && fp < cbeg; fp++) {
        if (s[fp] != 'f' || fp == 0 || !(lx_is_digit(s[fp - 1]) || s[fp - 1] == '.')) continue;
        uint32_t la = 0, lb = 0;
        lx_decode(s, n, fp + 1, &la);
        if (fp + 1 + la == nr && nr == n) return true;
        if (fp + 1 + la < n) lx_decode(s, n, fp + 1 + la, &lb);
        if (lb && fp + 1 + la + lb == nr) return true;
    }
    return false;
}

// Walks one chunk [cbeg, cend) given the context state at cbeg; calls f(tok) for every token that
// starts inside the chunk.  Returns nothing; `f` accumulates.
template <class F>
__device__ __forceinline__ void lex_chunk(const uint8_t* __restrict__ s, uint32_t n, uint32_t cbeg, uint32_t cend, uint32_t state, uint32_t* status, const Resync& rs, F&& f) {
    uint32_t pos = cbeg;
    bool forced = false;      // a resync point: context N and a token boundary, whatever precedes it
    if (rs.n) {
        uint32_t nr = rs_next(rs, cbeg);
        if (nr == cbeg) { state = CS_N; forced = true; }
        // the invalid-float-suffix error swallows two characters (parse.rs:563-581): a chunk that starts inside them, in code
        // context, starts inside that Err token, which ends at the resync point
        else if (!rs.files && state == CS_N && nr != 0xFFFFFFFFu && nr - cbeg <= 8 && lx_inside_suffix_error(s, n, cbeg, nr)) { pos = nr < cend ? nr : cend; forced = true; }
        // (a) as below, but a resync point ends the string: context N and a token boundary at pos (> cbeg: no look-back)
        while (state != CS_N && pos < cend) {
            if (pos == nr) { state = CS_N; break; }
            state = ctx_apply(ctx_of_byte(s[pos]), state); pos++;
        }
    } else {
        // (a) finish a string / comment that began in an earlier chunk
        while (state != CS_N && pos < cend) { state = ctx_apply(ctx_of_byte(s[pos]), state); pos++; }
    }
    if (state != CS_N) return;
    // (b) we are in context N at pos (== cbeg unless (a) moved us, in which case pos is a boundary).
    if (pos == cbeg && cbeg > 0 && !forced && lx_is_run(s[cbeg - 1])) {
        // The chunk may begin inside a token: find the closest certain boundary to the left
        // (a byte that is not a run character in context N always ends a token; so does a resync point) and re-lex from it.
        uint32_t p = cbeg;
        while (p > 0 && lx_is_run(s[p - 1]) && !rs_has(rs, p)) p--;
        while (p < cbeg) {
            LexTok t;
            // only run characters here: never whitespace, ';' or '"'
            lex_one(s, n, p, t, nullptr, status, rs);
            p = t.end;
        }
        pos = p;   // >= cbeg: first boundary at or after the chunk start
    }
    while (pos < cend) {
        int c = s[pos];
        if (lx_is_ws(c)) { pos++; continue; }
        if (c >= 0x80) {
            if (c < 0xC0) { pos++; continue; }   // continuation byte at the chunk start: the character belongs to the lane of its first byte
            uint32_t L;
            if (lx_unicode_ws(lx_decode(s, n, pos, &L))) { pos += L; continue; }   // parse.rs:432,474 char::is_whitespace
        }
        if (c == ';') {   // comment: to the next '\n' (parse.rs:450-457); the '\n' itself is whitespace
            pos++;
            while (pos < n && s[pos] != '\n') pos++;
            continue;
        }
        pos = f(pos);
    }
}

// per-chunk packed counts: tok (8 bits) | val (8 bits); string-pool bytes separately
#ifndef OPUSFE_LEX_WAVES
#define OPUSFE_LEX_WAVES 8
#endif
#if OPUSFE_LEX_WAVES
#define LEX_ATTR __attribute__((amdgpu_waves_per_eu(OPUSFE_LEX_WAVES, OPUSFE_LEX_WAVES)))
#else
#define LEX_ATTR
#endif
__global__ LEX_ATTR __launch_bounds__(LEX_THREADS) void k_lex_count(const uint8_t* __restrict__ src, uint32_t n, const uint32_t* __restrict__ chunk_fn,
                                                         const uint8_t* __restrict__ tile_state, uint8_t* __restrict__ chunk_state,
                                                         uint32_t* __restrict__ chunk_cnt, uint32_t* __restrict__ chunk_str,
                                                         Cnt3* __restrict__ tile_cnt, uint32_t* status, Resync rs, uint64_t* __restrict__ chunk_mask) {
    __shared__ uint32_t lds[4];
    __shared__ Cnt3 lds3[4];
    uint32_t chunk = blockIdx.x * LEX_THREADS + threadIdx.x;
    uint64_t base = (uint64_t)chunk * LEX_CHUNK;
    bool live = base < n;
    uint32_t f = live ? chunk_fn[chunk] : CTX_IDENTITY;
    uint32_t incl = block_inclusive_scan(f, OpCtx(), lds, (uint32_t*)nullptr);
    // exclusive prefix within the tile
    uint32_t prev = shfl_up_t(incl, 1);
    __shared__ uint32_t wave_last[4];
    if ((threadIdx.x & 63) == 63) wave_last[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t excl = threadIdx.x == 0 ? CTX_IDENTITY : ((threadIdx.x & 63) == 0 ? wave_last[(threadIdx.x >> 6) - 1] : prev);
    uint32_t state = ctx_apply(excl, tile_state[blockIdx.x]);
    Cnt3 c{0, 0, 0, 0};
    if (live) {
        uint32_t cbeg = (uint32_t)base, cend = (uint32_t)((base + LEX_CHUNK < n) ? base + LEX_CHUNK : n);
        if (rs.files) {      // group mode: the source ends where the chunk's file ends; unsupported inputs are recorded per file
            const uint32_t f = pos_file(rs.pos, rs.n, cbeg);
            status = rs.fstatus + 2 * f;
            n = rs.pos[f] + rs.flen[f];
            if (cend > n) cend = n;
        }
        uint64_t mask = 0;      // bit b: a token starts at byte cbeg + b (what the token-parallel emit pass goes by)
        if (cbeg < cend) lex_chunk(src, n, cbeg, cend, state, status, rs, [&](uint32_t pos) {
            LexTok t;
            lex_one(src, n, pos, t, nullptr, status, rs);
            c.tok++; c.val += t.nval; c.str += t.strbytes;
            mask |= 1ull << (pos - cbeg);
            return t.end;
        });
        if (chunk_mask) chunk_mask[chunk] = mask;
        chunk_state[chunk] = (uint8_t)state;
        chunk_cnt[chunk] = c.tok | (c.val << 8);
        chunk_str[chunk] = c.str;
    }
    Cnt3 total;
    block_inclusive_scan(c, OpAddCnt3(), lds3, &total);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

__global__ LEX_ATTR __launch_bounds__(LEX_THREADS) void k_lex_emit(const uint8_t* __restrict__ src, uint32_t n, const uint8_t* __restrict__ chunk_state,
                                                        const uint32_t* __restrict__ chunk_cnt, const uint32_t* __restrict__ chunk_str,
                                                        const Cnt3* __restrict__ tile_off, opusfe_token* __restrict__ raw, uint64_t* __restrict__ val64,
                                                        uint8_t* __restrict__ strpool, uint32_t* status, Resync rs) {
    __shared__ Cnt3 lds3[4];
    __shared__ Cnt3 wave_last[4];
    uint32_t chunk = blockIdx.x * LEX_THREADS + threadIdx.x;
    uint64_t base = (uint64_t)chunk * LEX_CHUNK;
    bool live = base < n;
    Cnt3 c{0, 0, 0, 0};
    if (live) { uint32_t pc = chunk_cnt[chunk]; c.tok = pc & 0xFF; c.val = pc >> 8; c.str = chunk_str[chunk]; }
    Cnt3 incl = block_inclusive_scan(c, OpAddCnt3(), lds3, (Cnt3*)nullptr);
    Cnt3 off = tile_off[blockIdx.x];
    off.tok += incl.tok - c.tok; off.val += incl.val - c.val; off.str += incl.str - c.str;
    (void)wave_last;
    if (!live || c.tok == 0) return;
    uint32_t cbeg = (uint32_t)base, cend = (uint32_t)((base + LEX_CHUNK < n) ? base + LEX_CHUNK : n);
    uint32_t state = chunk_state[chunk];
    uint32_t ti = off.tok, vi = off.val, si = off.str;
    uint32_t mix = 0;
    if (rs.files) {
        const uint32_t f = pos_file(rs.pos, rs.n, cbeg);
        status = rs.fstatus + 2 * f; mix = f * FILE_MIX;
        n = rs.pos[f] + rs.flen[f];
        if (cend > n) cend = n;
    }
    if (cbeg < cend) lex_chunk(src, n, cbeg, cend, state, status, rs, [&](uint32_t pos) {
        LexTok t;
        lex_one(src, n, pos, t, strpool + si, status, rs);
        opusfe_token o;
        o.kind = t.kind; o.flags = t.flags; o.aux = t.aux; o.low = t.low; o.len = t.end - t.low;
        if (t.kind == T_Error) { o.low = pos; o.payload = vi; }
        else if (t.kind == T_String) o.payload = si;
        else if (t.nval) o.payload = vi;
        else o.payload = t.kind == T_Integer ? t.payload : t.payload + mix;      // group mode: names (identifiers, operator symbols) carry their file
        if (t.nval >= 1) val64[vi] = t.v0;
        if (t.nval >= 2) val64[vi + 1] = t.v1;
        vi += t.nval; si += t.strbytes;
        raw[ti++] = o;
        return t.end;
    });
}

// Token-parallel emit: ONE THREAD PER TOKEN.  The count pass left, per 64-byte chunk, the number of tokens that start in it and the
// bitmask of their start bytes; a workgroup takes a tile (256 chunks), finds for its t-th token the chunk (binary search over the
// tile's prefix counts in LDS) and the start byte (t-th set bit of the chunk's mask), lexes exactly that token, and writes
// raw[tile base + t]: neighbouring lanes write neighbouring 16-byte records (whole lines leave the wave) and read neighbouring
// source bytes.  Literal-pool slots come from a block scan of (values, string bytes) over the 256 tokens of an iteration.
// (The chunk-per-lane version wrote 64 scattered token streams per wave: 11.7 GB fetched and 6.7 GB written per GB of source.)
__global__ LEX_ATTR __launch_bounds__(LEX_THREADS) void k_lex_emit_tok(const uint8_t* __restrict__ src, uint32_t n, const uint32_t* __restrict__ chunk_cnt,
                                                            const uint64_t* __restrict__ chunk_mask, const Cnt3* __restrict__ tile_off, opusfe_token* __restrict__ raw,
                                                            uint64_t* __restrict__ val64, uint8_t* __restrict__ strpool, uint32_t* status, Resync rs) {
    __shared__ uint32_t lds1[4];
    __shared__ Cnt3 lds3[4];
    __shared__ uint32_t off[LEX_THREADS + 1];
    __shared__ uint64_t masks[LEX_THREADS];
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = blockIdx.x * LEX_THREADS + tid;
    const uint64_t cbase = (uint64_t)chunk * LEX_CHUNK;
    const bool live = cbase < n;
    const uint32_t my_cnt = live ? (chunk_cnt[chunk] & 0xFFu) : 0u;
    masks[tid] = live ? chunk_mask[chunk] : 0ull;
    __shared__ uint32_t cfile[LEX_THREADS];      // group mode: the file of every chunk of the tile (a chunk never holds two files)
    if (rs.files) cfile[tid] = live ? pos_file(rs.pos, rs.n, (uint32_t)cbase) : 0u;
    uint32_t total = 0;
    const uint32_t incl = block_inclusive_scan(my_cnt, OpAddU32(), lds1, &total);
    off[tid] = incl - my_cnt;
    if (tid == 0) off[LEX_THREADS] = total;
    __syncthreads();
    const Cnt3 base = tile_off[blockIdx.x];
    uint32_t run_val = base.val, run_str = base.str;
    const uint32_t tile_byte0 = blockIdx.x * LEX_THREADS * LEX_CHUNK;
    const uint32_t n_all = n;
    for (uint32_t t0 = 0; t0 < total; t0 += LEX_THREADS) {      // (uniform trip count: the block scans inside are collective)
        const uint32_t t = t0 + tid;
        const bool active = t < total;
        LexTok tk; tk.nval = 0; tk.strbytes = 0; tk.kind = T_EndOfFile;
        uint32_t pos = 0, mix = 0, n_f = n_all;
        uint32_t* st = status;
        if (active) {
            uint32_t lo = 0, hi = LEX_THREADS;                  // last k with off[k] <= t
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= t) lo = mid; else hi = mid; }
            uint64_t m = masks[lo];
            for (uint32_t r = t - off[lo]; r > 0; r--) m &= m - 1;
            pos = tile_byte0 + lo * LEX_CHUNK + (uint32_t)__builtin_ctzll(m);
            if (rs.files) {
                const uint32_t f = cfile[lo];
                st = rs.fstatus + 2 * f; mix = f * FILE_MIX; n_f = rs.pos[f] + rs.flen[f];
            }
            lex_one(src, n_f, pos, tk, nullptr, st, rs);
        }
        Cnt3 mine{0, active ? tk.nval : 0u, active ? tk.strbytes : 0u, 0};
        Cnt3 tot3;
        const Cnt3 inc3 = block_inclusive_scan(mine, OpAddCnt3(), lds3, &tot3);
        if (active) {
            const uint32_t vi = run_val + inc3.val - mine.val, si = run_str + inc3.str - mine.str;
            opusfe_token o;
            o.kind = tk.kind; o.flags = tk.flags; o.aux = tk.aux; o.low = tk.low; o.len = tk.end - tk.low;
            if (tk.kind == T_Error) { o.low = pos; o.payload = vi; }
            else if (tk.kind == T_String) o.payload = si;
            else if (tk.nval) o.payload = vi;
            else o.payload = tk.kind == T_Integer ? tk.payload : tk.payload + mix;      // group mode: names (identifiers, operator symbols) carry their file
            if (tk.nval >= 1) val64[vi] = tk.v0;
            if (tk.nval >= 2) val64[vi + 1] = tk.v1;
            raw[base.tok + t] = o;
            if (tk.strbytes) { LexTok again; lex_one(src, n_f, pos, again, strpool + si, st, rs); }      // strings only: decode into the pool
        }
        run_val += tot3.val; run_str += tot3.str;
    }
}

// ---- token-level passes ----------------------------------------------------------------------
// kind of the token `back` places before raw token k; tokens before raw[0] come from the shard carry `prev_kinds`
// (byte j = kind of the (j+1)-th token before the shard, T_EndOfFile = none; 0xFF..FF for a whole source)
#define TK_NO_PREV 0x3B3B3B3B3B3B3B3Bull   /* T_EndOfFile in every byte */
__device__ __forceinline__ uint8_t tk_kind_back(const opusfe_token* __restrict__ raw, int64_t k, int back, uint64_t prev_kinds) {
    int64_t i = k - back;
    if (i >= 0) return raw[i].kind;
    int64_t j = -i - 1;
    return j < 8 ? (uint8_t)(prev_kinds >> (8 * j)) : (uint8_t)T_EndOfFile;
}
__device__ __forceinline__ bool tk_record_open(const opusfe_token* __restrict__ raw, int64_t k, uint64_t prev_kinds) {
    // `{` that raises ignore_dents: `record {` (parse.rs:925-928) or `record <name> {` (parse.rs:1047-1054),
    // where <name> may be the two-token form `operator X` (parse.rs:776).
    if (tk_kind_back(raw, k, 1, prev_kinds) == T_Record) return true;
    if (tk_kind_back(raw, k, 1, prev_kinds) == T_Identifier && tk_kind_back(raw, k, 2, prev_kinds) == T_Record) return true;
    if (tk_kind_back(raw, k, 2, prev_kinds) == T_Operator && tk_kind_back(raw, k, 3, prev_kinds) == T_Record) return true;
    return false;
}

// Group mode tables of the token-level passes: file of every raw token, first raw token of every file (NONE: the file has none),
// bracket depth in front of every file (the running sum is taken over the whole group; a file counts from its own start).
struct FileTok { const uint32_t* start; const uint32_t* len; uint32_t n; const uint32_t* fid; const uint32_t* first_tok; const int32_t* base; };

__global__ void k_tok_file(const opusfe_token* __restrict__ raw, uint32_t T, const uint32_t* __restrict__ start, uint32_t n_files, uint32_t* __restrict__ fid,
                           uint32_t* __restrict__ first_tok, uint32_t* __restrict__ flags) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= T) return;
    const opusfe_token t = raw[k];
    const uint32_t f = pos_file(start, n_files, t.low);
    fid[k] = f;
    if (k == 0 || pos_file(start, n_files, raw[k - 1].low) != f) first_tok[f] = k;
    if (t.kind == T_Error && (t.flags & OPUSFE_TF_RESYNC)) atomicOr(&flags[f], (uint32_t)FF_RESYNC);
}
__global__ void k_file_base(const int32_t* __restrict__ depth_incl, const uint32_t* __restrict__ first_tok, uint32_t n_files, int32_t* __restrict__ base) {
    uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_files) return;
    const uint32_t k = first_tok[f];
    base[f] = (k == OPUSFE_NONE || k == 0) ? 0 : depth_incl[k - 1];
}

__global__ void k_tok_delta(const opusfe_token* __restrict__ raw, uint32_t T, int32_t* __restrict__ delta, int force_dents, uint64_t prev_kinds, FileTok ft) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= T) return;
    int32_t d = 0;
    if (ft.n) {      // the look-back never leaves the token's file
        const uint32_t lo = ft.first_tok[ft.fid[k]];
        raw += lo; k -= lo; prev_kinds = TK_NO_PREV;
        delta += lo;
    }
    if (!force_dents) {
        uint8_t kd = raw[k].kind;
        if (kd == T_LeftParenthesis) d = 1;
        else if (kd == T_RightParenthesis) d = -1;
        else if (kd == T_LeftBrace) d = tk_record_open(raw, k, prev_kinds) ? 1 : 0;
        else if (kd == T_RightBrace) {
            // `{}` / `{N}` array forms never touch ignore_dents (parse.rs:943-958)
            bool array_close = (tk_kind_back(raw, k, 1, prev_kinds) == T_LeftBrace && !tk_record_open(raw, (int64_t)k - 1, prev_kinds))
                            || (tk_kind_back(raw, k, 1, prev_kinds) == T_Integer && tk_kind_back(raw, k, 2, prev_kinds) == T_LeftBrace
                                && !tk_record_open(raw, (int64_t)k - 2, prev_kinds));
            d = array_close ? 0 : -1;
        }
    }
    delta[k] = d;
}

// For gap k (before raw token k; k == T is the gap before end of file): value | 1<<31 when the gap
// is dent-sensitive and crosses a line (or is the very first gap / the EOF gap).
__global__ void k_gap(const uint8_t* __restrict__ src, uint32_t n, const opusfe_token* __restrict__ raw, uint32_t T,
                      const int32_t* __restrict__ depth_incl, uint32_t* __restrict__ gapval, int32_t depth_in, FileTok ft) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > T) return;
    // group mode: a gap belongs to the file of the token behind it (the gap behind the last token to the last file); the first gap
    // of a file starts at the file's first byte and is measured like the first line of a source (parse.rs:429)
    int32_t before = k == 0 ? 0 : depth_incl[k - 1];
    bool first = false;
    uint32_t fstart = 0;
    if (ft.n && T > 0) {
        const uint32_t f = ft.fid[k < T ? k : T - 1];
        before -= ft.base[f];
        first = k < T && ft.first_tok[f] == k;
        fstart = ft.start[f];
    }
    bool sens = before + depth_in == 0;
    uint32_t out = 0;
    if (sens) {
        if (k == T) out = 0x80000000u;                       // EOF: -(indent) (parse.rs:461-464)
        else {
            uint32_t gbeg = k == 0 ? 0 : raw[k - 1].low + raw[k - 1].len;
            if (first) gbeg = fstart;
            uint32_t gend = raw[k].low;
            if (raw[k].kind == T_Error) gend = raw[k].low;
            uint32_t p = gend;
            bool nl = false;
            while (p > gbeg) { if (src[p - 1] == '\n') { nl = true; break; } p--; }
            if (nl || gbeg == 0 || first) {
                // p = position just after the last '\n' (or the start of the source): count leading TABs (parse.rs:438-444)
                uint32_t q = nl ? p : gbeg, tabs = 0;
                while (q < gend && src[q] == '\t') { tabs++; q++; }
                out = 0x80000000u | tabs;
            }
        }
    }
    gapval[k] = out;
}

// count[k] = |delta| + 1
// Group mode: the gap in front of the first token of file B (and the gap behind the last token) also ENDS file A = the file of the
// token before it: A's dedents back to column 0 (only in dent-sensitive mode, Q8) and A's EndOfFile token, both at A's last byte;
// then B's first line (indents from column 0).
struct FileClose { bool closes; uint32_t nclose, tabs, end_a; };
__device__ __forceinline__ FileClose file_close(const FileTok& ft, uint32_t k, uint32_t T, const uint32_t* __restrict__ gapval, const uint32_t* __restrict__ prev_excl,
                                                const int32_t* __restrict__ depth_incl) {
    FileClose c{false, 0, 0, 0};
    if (!ft.n || k == 0 || T == 0) return c;
    if (k < T && ft.first_tok[ft.fid[k]] != k) return c;
    const uint32_t a = ft.fid[k - 1];
    c.closes = true;
    const bool a_sens = depth_incl[k - 1] - ft.base[a] == 0;
    c.nclose = a_sens ? (prev_excl[k] & 0x7FFFFFFFu) : 0u;
    c.tabs = k < T ? (gapval[k] & 0x7FFFFFFFu) : 0u;
    c.end_a = ft.start[a] + ft.len[a];
    return c;
}

__global__ void k_dent_count(const uint32_t* __restrict__ gapval, const uint32_t* __restrict__ prev_excl, uint32_t T, uint32_t* __restrict__ cnt, FileTok ft,
                             const int32_t* __restrict__ depth_incl) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > T) return;
    const FileClose fc = file_close(ft, k, T, gapval, prev_excl, depth_incl);
    if (fc.closes) { cnt[k] = fc.nclose + 1u + (k < T ? fc.tabs + 1u : 0u); return; }
    uint32_t g = gapval[k];
    int32_t d = 0;
    if (g >> 31) d = (int32_t)(g & 0x7FFFFFFFu) - (int32_t)(prev_excl[k] & 0x7FFFFFFFu);
    cnt[k] = (uint32_t)(d < 0 ? -d : d) + 1;
}

__global__ void k_final(const opusfe_token* __restrict__ raw, uint32_t T, uint32_t n, const uint32_t* __restrict__ gapval,
                        const uint32_t* __restrict__ prev_excl, const uint32_t* __restrict__ out_off, opusfe_token* __restrict__ tokens,
                        uint32_t* __restrict__ seg_flag, FileTok ft, const int32_t* __restrict__ depth_incl, uint32_t* __restrict__ file_tok_begin) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > T) return;
    const FileClose fc = file_close(ft, k, T, gapval, prev_excl, depth_incl);
    if (fc.closes) {
        uint32_t o = out_off[k];
        opusfe_token t; t.flags = 0; t.aux = 0; t.len = 0; t.payload = 0; t.low = fc.end_a;
        t.kind = T_Dedent;
        for (uint32_t i = 0; i < fc.nclose; i++) tokens[o++] = t;
        t.kind = T_EndOfFile; tokens[o++] = t;
        if (k < T) {
            file_tok_begin[ft.fid[k]] = o;                  // the file's slice of the final stream starts behind its predecessor's EndOfFile
            t.kind = T_Indent; t.low = raw[k].low;
            for (uint32_t i = 0; i < fc.tabs; i++) tokens[o++] = t;
            tokens[o] = raw[k];
            seg_flag[k] = 1;                                // a file starts a segment, whatever its first token is
        } else seg_flag[k] = 0;
        return;
    }
    if (ft.n && k == 0 && T > 0) file_tok_begin[ft.fid[0]] = 0;
    uint32_t g = gapval[k];
    int32_t d = 0;
    if (g >> 31) d = (int32_t)(g & 0x7FFFFFFFu) - (int32_t)(prev_excl[k] & 0x7FFFFFFFu);
    uint32_t nd = (uint32_t)(d < 0 ? -d : d);
    uint32_t o = out_off[k];
    opusfe_token dent;
    dent.kind = d > 0 ? T_Indent : T_Dedent; dent.flags = 0; dent.aux = 0; dent.len = 0; dent.payload = 0;
    dent.low = k < T ? raw[k].low : n;
    for (uint32_t i = 0; i < nd; i++) tokens[o + i] = dent;
    if (k < T) tokens[o + nd] = raw[k];
    else { opusfe_token e; e.kind = T_EndOfFile; e.flags = 0; e.aux = 0; e.low = n; e.len = 0; e.payload = 0; tokens[o + nd] = e; }
    // top-level segment starts: token 0, and the first token after a dedent back to level 0 unless it
    // is `else`, a lexical-error token or EOF (see DESIGN.md "segments").
    uint32_t flag = 0;
    if (k < T) {
        uint8_t kd = raw[k].kind;
        if (k == 0) flag = 1;
        else if (d < 0 && (g & 0x7FFFFFFFu) == 0 && kd != T_Else && kd != T_Error) flag = 1;
        // `extern`, `type`, `const`, `import`, `library` occur nowhere in the grammar except as the first token of a
        // definition (parse.rs:1262-1322): wherever one stands, a parser that is not between definitions fails ON it, so it
        // may always open a segment (splits long runs of unindented declarations, e.g. the header blocks of shard.py)
        else if (kd == T_Extern || kd == T_Type || kd == T_Const || kd == T_Import || kd == T_Library) flag = 1;
    }
    seg_flag[k] = flag;
}

// Shard cut search (multi-GPU): byte position of the first raw token that starts a top-level segment by
// evidence found INSIDE the shard — its gap crosses a line in dent-sensitive mode with no leading TAB, and the previous
// dent-sensitive line crossing of this shard had some (the dedent to column 0 that closes every open block).
__global__ void k_shard_first_segment(const opusfe_token* __restrict__ raw, uint32_t T, const uint32_t* __restrict__ gapval,
                                      const uint32_t* __restrict__ prev_excl, uint32_t* __restrict__ out_pos) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= T) return;
    uint32_t g = gapval[k], p = prev_excl[k];
    if (!(g >> 31) || (g & 0x7FFFFFFFu) != 0) return;
    if (!(p >> 31) || (p & 0x7FFFFFFFu) == 0) return;
    uint8_t kd = raw[k].kind;
    if (kd == T_Else || kd == T_Error) return;
    atomicMin(out_pos, raw[k].low);
}

// Group mode, one lane per file: what only the lexer can see of a file whose packed lexing may differ from lexing it alone (the
// file then takes the single-file path): no token at all (its EndOfFile token has no gap to live in), an indented first line (its
// Indent tokens would fall into the previous file's last segment), an input the reference cannot be followed on.
__global__ void k_file_lex_flags(const uint32_t* __restrict__ first_tok, const uint32_t* __restrict__ gapval, uint32_t n_files,
                                 const uint32_t* __restrict__ fstatus, uint32_t* __restrict__ flags) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_files) return;
    uint32_t fl = 0;
    if (first_tok[f] == OPUSFE_NONE) fl |= FF_EMPTY;
    else if (gapval[first_tok[f]] & 0x7FFFFFFFu) fl |= FF_INDENTED;
    if (fstatus[2 * f]) fl |= FF_UNSUP;
    if (fl) atomicOr(&flags[f], fl);
}

// seg_start[id] = index in the final token stream where segment id starts
__global__ void k_seg_starts(const uint32_t* __restrict__ seg_flag, const uint32_t* __restrict__ seg_excl, const uint32_t* __restrict__ out_off,
                             const uint32_t* __restrict__ cnt, uint32_t T, uint32_t* __restrict__ seg_start) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= T) return;
    if (seg_flag[k]) seg_start[seg_excl[k]] = (k == 0) ? 0 : out_off[k] + cnt[k] - 1;
}

}  // namespace opusfe
