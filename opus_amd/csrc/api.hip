// libopusfe: C ABI entry points (include/opusfe.h) and host drivers of the HIP pipeline.
#include <hip/hip_runtime.h>
#include <atomic>
#include <thread>
#include "defs.h"
#include "runtime.h"
#include "unit.h"
#include "scan.h"
// Code layout: the lowering machine is the largest and hottest code (Lowerer::run is 192 KB against a 64 KB instruction
// cache) and its speed moves by ~5 % with where it lands in the code object (measured: 45.1 .. 47.6 ms per 256 MB over six
// paddings, tools/variants.sh).  It is therefore emitted first, behind a fixed pad, so that edits to the lexer, the parser or
// this file no longer shift it; OPUSFE_CODE_PAD (s_nop count) is the knob that was swept.
#ifndef OPUSFE_CODE_PAD
#define OPUSFE_CODE_PAD 256
#endif
#define OPUSFE_STR2(x) #x
#define OPUSFE_STR(x) OPUSFE_STR2(x)
namespace opusfe { __global__ void k_code_pad() { asm volatile(".rept " OPUSFE_STR(OPUSFE_CODE_PAD) "\n s_nop 0\n .endr"); } }
#include "lower.h"
#include "lex.h"
#include "parse.h"
#include "join.h"
#include "dump.h"
#include "dump_ast.h"
#include "dump_ir.h"

using namespace opusfe;

namespace {

// status / totals block layout (uint32 words) inside ctx->d_status
enum { W_STATUS = 0, W_UNSUP_POS = 1, W_LEXTOT = 4 /* Cnt3: 4 words */, W_FINTOK = 8, W_NSEG = 9, W_ENDFN = 10, W_ENDDEPTH = 11, W_PARSE = 16, W_LOWER = 32 };

int fail(opusfe_ctx* ctx, int code, const std::string& msg) { if (ctx) ctx->msg = msg; return code; }

void sync_stream(opusfe_ctx* ctx) { HIP_TRY(hipStreamSynchronize(ctx->stream)); prof_collect(ctx); }

void fetch_status(opusfe_ctx* ctx) {
    HIP_TRY(hipMemcpyAsync(ctx->h_status, ctx->d_status, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    sync_stream(ctx);
}

int status_to_code(uint32_t ds) {
    if (ds == DS_OK) return OPUSFE_OK;
    if (ds >= DS_INTERNAL_CAPACITY) return OPUSFE_ERR_INTERNAL;
    return OPUSFE_UNSUPPORTED;
}
const char* status_text(uint32_t ds) {
    switch (ds) {
        case DS_UNSUPPORTED_NONASCII: return "non-ASCII character right behind an identifier (char::is_alphanumeric tables unpinned)";
        case DS_UNSUPPORTED_FLOAT: return "float literal with more than 300 significant digits";
        case DS_UNSUPPORTED_HANG: return "float literal followed by 'f' at end of file (the reference never terminates)";
        case DS_UNSUPPORTED_RESYNC: return "swallowed lexical error after which the reference re-lexes a string body as code";
        case DS_UNSUPPORTED_SEMANTIC: return "construct on which the reference panics (unreachable!/unwrap)";
        case DS_INTERNAL_CAPACITY: return "internal capacity exceeded";
        case DS_INTERNAL_SEGMENT: return "internal segment consistency check failed";
        default: return "";
    }
}

// ---- lexer driver ---------------------------------------------------------------------------
void run_lexer(opusfe_ctx* ctx, opusfe_unit* u, int force_dents) {
    const uint8_t* src = u->d_src;
    const uint32_t n = (uint32_t)u->len;
    hipStream_t st = ctx->stream;
    TempList tmp(ctx);
    HIP_TRY(hipMemsetAsync(ctx->d_status, 0, 64 * sizeof(uint32_t), st));
    HIP_TRY(hipMemsetAsync(ctx->d_status + W_UNSUP_POS, 0xFF, sizeof(uint32_t), st));
    uint32_t* status = ctx->d_status + W_STATUS;

    const uint64_t nchunks = ((uint64_t)n + LEX_CHUNK - 1) / LEX_CHUNK;
    const uint64_t ntiles = (nchunks + LEX_THREADS - 1) / LEX_THREADS;
    uint32_t T = 0, nval = 0, nstr = 0;
    opusfe_token* raw = nullptr;
    Resync rs{nullptr, 0};
    GroupInfo* const grp = u->group;      // a corpus of files packed into this unit (opusfe_parse_group), else null
    if (grp) rs = Resync{grp->d_start, grp->n_files, 1u, grp->d_fstatus, grp->d_len};
    if (!u->resync.empty()) {   // a later pass of parse_with_resync
        uint32_t* d = tmp.get<uint32_t>(u->resync.size());
        HIP_TRY(hipMemcpyAsync(d, u->resync.data(), u->resync.size() * 4, hipMemcpyHostToDevice, st));
        rs = Resync{d, (uint32_t)u->resync.size()};
    }
    if (ntiles > 0) {
        uint32_t* chunk_fn = tmp.get<uint32_t>(nchunks);
        uint32_t* tile_fn = tmp.get<uint32_t>(ntiles);
        uint8_t* tile_state = tmp.get<uint8_t>(ntiles);
        uint8_t* chunk_state = tmp.get<uint8_t>(nchunks);
        uint32_t* chunk_cnt = tmp.get<uint32_t>(nchunks);
        uint32_t* chunk_str = tmp.get<uint32_t>(nchunks);
        Cnt3* tile_cnt = tmp.get<Cnt3>(ntiles);
        Cnt3* scan_tmp = tmp.get<Cnt3>((ntiles + SCAN_TILE - 1) / SCAN_TILE + 1);
        LAUNCH(ctx, "lex_ctx_chunk", k_ctx_chunk, ntiles, LEX_THREADS, src, n, chunk_fn, tile_fn, rs);
        LAUNCH(ctx, "lex_ctx_tiles", k_ctx_tiles, 1, SCAN_THREADS, tile_fn, tile_state, (uint32_t)ntiles, (uint32_t)CS_N, ctx->d_status + W_ENDFN);
        uint64_t* chunk_mask = tmp.get<uint64_t>(nchunks);
        LAUNCH(ctx, "lex_count", k_lex_count, ntiles, LEX_THREADS, src, n, chunk_fn, tile_state, chunk_state, chunk_cnt, chunk_str, tile_cnt, status, rs, chunk_mask);
        prof_begin(ctx, "scan_cnt3");
        device_scan<true>(tile_cnt, tile_cnt, ntiles, OpAddCnt3(), Cnt3{0, 0, 0, 0}, scan_tmp, (Cnt3*)(ctx->d_status + W_LEXTOT), st);
        prof_end(ctx);
        fetch_status(ctx);
        T = ctx->h_status[W_LEXTOT]; nval = ctx->h_status[W_LEXTOT + 1]; nstr = ctx->h_status[W_LEXTOT + 2];
        raw = tmp.get<opusfe_token>(T);
        u->d_val64 = dev_alloc_t<uint64_t>(ctx, nval);
        u->d_strpool = dev_alloc_t<uint8_t>(ctx, nstr);
        u->n_val64 = nval; u->n_strpool = nstr;
#ifdef OPUSFE_DIAG
        static const bool old_emit = getenv("OPUSFE_LEX_EMIT_OLD") != nullptr;      // A/B against the chunk-per-lane emit
        if (old_emit) LAUNCH(ctx, "lex_emit", k_lex_emit, ntiles, LEX_THREADS, src, n, chunk_state, chunk_cnt, chunk_str, tile_cnt, raw, u->d_val64, u->d_strpool, status, rs);
        else
#endif
        LAUNCH(ctx, "lex_emit", k_lex_emit_tok, ntiles, LEX_THREADS, src, n, (const uint32_t*)chunk_cnt, (const uint64_t*)chunk_mask, (const Cnt3*)tile_cnt, raw, u->d_val64, u->d_strpool, status, rs);
    }
    // token-level passes over T raw tokens and T + 1 gaps
    const uint64_t G = (uint64_t)T + 1;
    int32_t* depth = tmp.get<int32_t>(G);
    uint32_t* gapval = tmp.get<uint32_t>(G);
    uint32_t* prev = tmp.get<uint32_t>(G);
    uint32_t* cnt = tmp.get<uint32_t>(G);
    uint32_t* out_off = tmp.get<uint32_t>(G);
    uint32_t* seg_flag = tmp.get<uint32_t>(G);
    uint32_t* seg_excl = prev;      // (prev is dead behind k_final, which writes seg_flag; the segment scan then writes seg_excl)
    uint64_t* scan_scr = tmp.get<uint64_t>(chained_scan_words(G));      // (the single-pass scans' tile words, scan.h)
    FileTok ft{};
    if (grp && T > 0) {
        // which file every token belongs to, the first token of every file, and (behind the depth scan) the depth in front of every file
        uint32_t* fid = tmp.get<uint32_t>(T);
        uint32_t* first_tok = tmp.get<uint32_t>(grp->n_files);
        int32_t* base = tmp.get<int32_t>(grp->n_files);
        HIP_TRY(hipMemsetAsync(first_tok, 0xFF, (size_t)grp->n_files * 4, st));
        LAUNCH(ctx, "tok_file", k_tok_file, grid_for(T, 256), 256, raw, T, grp->d_start, grp->n_files, fid, first_tok, grp->d_flags);
        ft = FileTok{grp->d_start, grp->d_len, grp->n_files, fid, first_tok, base};
    }
    if (T > 0) {
        LAUNCH(ctx, "tok_delta", k_tok_delta, grid_for(T, 256), 256, raw, T, depth, force_dents, (uint64_t)TK_NO_PREV, ft);
        prof_begin(ctx, "scan_depth");
        device_scan_chained<false>(depth, depth, T, OpAddI32(), (int32_t)0, scan_scr, (int32_t*)nullptr, status, st);
        prof_end(ctx);
        HIP_TRY(hipMemcpyAsync(ctx->d_status + W_ENDDEPTH, depth + (T - 1), 4, hipMemcpyDeviceToDevice, st));      // (opusfe_unit_end_state)
        if (ft.n) LAUNCH(ctx, "file_base", k_file_base, grid_for(ft.n, 256), 256, (const int32_t*)depth, ft.first_tok, ft.n, (int32_t*)ft.base);
    }
    LAUNCH(ctx, "tok_gap", k_gap, grid_for(G, 256), 256, src, n, raw, T, depth, gapval, (int32_t)0, ft);
    prof_begin(ctx, "scan_indent");
    device_scan_chained<true>(gapval, prev, G, OpLast(), (uint32_t)0, scan_scr, (uint32_t*)nullptr, status, st);
    prof_end(ctx);
    LAUNCH(ctx, "tok_dent_count", k_dent_count, grid_for(G, 256), 256, gapval, prev, T, cnt, ft, (const int32_t*)depth);
    prof_begin(ctx, "scan_tokoff");
    device_scan_chained<true>(cnt, out_off, G, OpAddU32(), (uint32_t)0, scan_scr, ctx->d_status + W_FINTOK, status, st);
    prof_end(ctx);
    fetch_status(ctx);
    const uint32_t total = ctx->h_status[W_FINTOK];
    u->d_tokens = dev_alloc_t<opusfe_token>(ctx, total);
    u->n_tokens = total;
    LAUNCH(ctx, "tok_final", k_final, grid_for(G, 256), 256, raw, T, n, gapval, prev, out_off, u->d_tokens, seg_flag, ft, (const int32_t*)depth, grp ? grp->d_tok_begin : (uint32_t*)nullptr);
    if (grp && ft.n)      // what only the lexer can see of a file that needs the single-file path
        LAUNCH(ctx, "file_lex_flags", k_file_lex_flags, grid_for(grp->n_files, 256), 256, ft.first_tok, (const uint32_t*)gapval, grp->n_files, (const uint32_t*)grp->d_fstatus, grp->d_flags);
    if (T > 0) {
        prof_begin(ctx, "scan_seg");
        device_scan_chained<true>(seg_flag, seg_excl, T, OpAddU32(), (uint32_t)0, scan_scr, ctx->d_status + W_NSEG, status, st);
        prof_end(ctx);
    }
    fetch_status(ctx);
    u->n_segs = T > 0 ? ctx->h_status[W_NSEG] : 0;
    if (T > 0) {      // (sized from the count: one entry per segment, not per token)
        u->d_seg_start = dev_alloc_t<uint32_t>(ctx, (uint64_t)u->n_segs + 1);
        LAUNCH(ctx, "tok_seg_starts", k_seg_starts, grid_for(T, 256), 256, seg_flag, seg_excl, out_off, cnt, T, u->d_seg_start);
    }
    u->end_ctx_state = ntiles > 0 ? ctx_apply(ctx->h_status[W_ENDFN], CS_N) : (uint32_t)CS_N;
    u->end_depth = T > 0 ? (int32_t)ctx->h_status[W_ENDDEPTH] : 0;
    u->dev_status = ctx->h_status[W_STATUS];
    u->unsup_pos = ctx->h_status[W_UNSUP_POS];
    u->status = status_to_code(u->dev_status);
    if (u->status != OPUSFE_OK) ctx->msg = status_text(u->dev_status);
}

void ensure_host_lex(opusfe_unit* u) {
    if (u->have_h_lex) return;
    opusfe_ctx* ctx = u->ctx;
    u->h_tokens.resize(u->n_tokens); u->h_val64.resize(u->n_val64); u->h_strpool.resize(u->n_strpool);
    if (u->n_tokens) HIP_TRY(hipMemcpyAsync(u->h_tokens.data(), u->d_tokens, u->n_tokens * sizeof(opusfe_token), hipMemcpyDeviceToHost, ctx->stream));
    if (u->n_val64) HIP_TRY(hipMemcpyAsync(u->h_val64.data(), u->d_val64, u->n_val64 * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (u->n_strpool) HIP_TRY(hipMemcpyAsync(u->h_strpool.data(), u->d_strpool, u->n_strpool, hipMemcpyDeviceToHost, ctx->stream));
    if (!u->have_h_src) {
        u->h_src.resize(u->len);
        if (u->len) HIP_TRY(hipMemcpyAsync(u->h_src.data(), u->d_src, u->len, hipMemcpyDeviceToHost, ctx->stream));
        u->have_h_src = true;
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    u->have_h_lex = true;
}

opusfe_unit* new_unit(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t len, bool own) {
    opusfe_unit* u = new opusfe_unit();
    u->ctx = ctx; u->d_src = d_src; u->len = len; u->own_src = own;
    return u;
}

template <class F>
int guarded(opusfe_ctx* ctx, F&& f) {
    if (!ctx) return OPUSFE_ERR_ARG;
    try {
        HIP_TRY(hipSetDevice(ctx->device));
        ctx->epoch++;
        return f();
    } catch (HipError& e) {
        ctx->pending.clear();
        return fail(ctx, e.e == hipErrorOutOfMemory ? OPUSFE_ERR_OOM : OPUSFE_ERR_HIP, std::string("HIP error: ") + hipGetErrorString(e.e) + " in " + e.what);
    } catch (ScanLaunchError&) {
        ctx->pending.clear();
        return fail(ctx, OPUSFE_ERR_HIP, "HIP error: a scan kernel failed to launch");
    } catch (std::bad_alloc&) {
        return fail(ctx, OPUSFE_ERR_OOM, "host allocation failed");
    }
}

constexpr uint64_t MAX_LEN = 0xFFFF0000ull;   // positions are u32 per unit (SURVEY §7 hard part 6: larger inputs are sharded)

}  // namespace


// ---- multi-GPU shard carries ---------------------------------------------------------
// A shard is a byte range of one huge source that starts right after a '\n'.  The carries form a chain:
//   context-DFA function of the bytes  ->  start state  ->  raw tokens  ->  kinds of the last tokens  ->  bracket depth
// Each link is a few bytes per rank exchanged with all_gather by the host (opus_amd/shard.py); these entry points
// compute the rank-local part on the GPU.
struct opusfe_shard {
    opusfe_ctx* ctx; const uint8_t* d_src; uint32_t len;
    opusfe_token* raw; uint32_t T;
    int32_t* depth; uint32_t* gapval; uint32_t* prev; uint32_t* scan_tmp;
    uint64_t* val64; uint8_t* strpool;
};

namespace {
void shard_release(opusfe_shard* s) {
    opusfe_ctx* ctx = s->ctx;
    dev_free(ctx, s->raw); dev_free(ctx, s->depth); dev_free(ctx, s->gapval); dev_free(ctx, s->prev); dev_free(ctx, s->scan_tmp);
    dev_free(ctx, s->val64); dev_free(ctx, s->strpool);
    delete s;
}
}  // namespace

extern "C" {

// composed context-automaton transition function (18 bits, lex.h CTXFN) of d_src[0, len)
int opusfe_shard_ctx(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t len, uint32_t* fn) {
    if (!fn || (len && !d_src) || len > MAX_LEN) return OPUSFE_ERR_ARG;
    return guarded(ctx, [&] {
        *fn = CTX_IDENTITY;
        if (len == 0) return (int)OPUSFE_OK;
        TempList tmp(ctx);
        const uint32_t n = (uint32_t)len;
        const uint64_t nchunks = ((uint64_t)n + LEX_CHUNK - 1) / LEX_CHUNK, ntiles = (nchunks + LEX_THREADS - 1) / LEX_THREADS;
        uint32_t* chunk_fn = tmp.get<uint32_t>(nchunks);
        uint32_t* tile_fn = tmp.get<uint32_t>(ntiles);
        uint8_t* tile_state = tmp.get<uint8_t>(ntiles);
        LAUNCH(ctx, "lex_ctx_chunk", k_ctx_chunk, ntiles, LEX_THREADS, d_src, n, chunk_fn, tile_fn, Resync{nullptr, 0});
        LAUNCH(ctx, "lex_ctx_tiles", k_ctx_tiles, 1, SCAN_THREADS, tile_fn, tile_state, (uint32_t)ntiles, (uint32_t)CS_N, ctx->d_status + 60);
        HIP_TRY(hipMemcpyAsync(ctx->h_status + 60, ctx->d_status + 60, 4, hipMemcpyDeviceToHost, ctx->stream));
        sync_stream(ctx);
        *fn = ctx->h_status[60];
        return (int)OPUSFE_OK;
    });
}

// ---- the one entry point that allocates nothing (no counterpart in the reference) -----------------------------------
// opusfe_shard_ctx for callers that cannot call hipMalloc (HIP graph capture, tight serving loops, harnesses that forbid
// it): same kernels, but the tile aggregates (one composed context function per 16 KiB tile, the only scratch
// opusfe_shard_ctx needs when the per-chunk functions are not kept) live in a file-scope device array instead of the
// caching allocator, and the result goes to a caller-provided device word instead of through a host read-back.
// `scratch` is a host-side stand-in for the symbol, NOT a device address: opusfe_create resolves it once per device
// with hipGetSymbolAddress and every use goes through that pointer (ctx->d_static_scratch).
constexpr uint64_t STATIC_MAX_LEN = 1ull << 30;                                 // largest byte range the entry point accepts
constexpr uint32_t STATIC_MAX_TILES = (uint32_t)(STATIC_MAX_LEN / LEX_TILE);    // 65536 tiles of 16 KiB -> 256 KiB of scratch
static __device__ __align__(256) uint32_t scratch[STATIC_MAX_TILES];

int opusfe_shard_ctx_static(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t len, uint32_t* d_fn) {
    if (!ctx || !d_fn || (len && !d_src)) return OPUSFE_ERR_ARG;
    if (len > STATIC_MAX_LEN) { ctx->msg = "opusfe_shard_ctx_static: byte range larger than 1 GiB (the static scratch holds 65536 tile aggregates): split it or use opusfe_shard_ctx"; return OPUSFE_ERR_ARG; }
    if (!ctx->d_static_scratch) { ctx->msg = "opusfe_shard_ctx_static: the static scratch of this device was not resolved at opusfe_create"; return OPUSFE_ERR_HIP; }
    return guarded(ctx, [&] {
        const uint32_t n = (uint32_t)len;
        const uint64_t nchunks = ((uint64_t)n + LEX_CHUNK - 1) / LEX_CHUNK, ntiles = (nchunks + LEX_THREADS - 1) / LEX_THREADS;
        // both kernels write every word they read later (tile_fn[0, ntiles) are all stored before k_ctx_tiles runs), so the
        // array needs no clearing between calls; ntiles == 0 leaves the identity function in *d_fn
        if (ntiles > 0) {
            hipLaunchKernelGGL(k_ctx_chunk, dim3((unsigned)ntiles), dim3(LEX_THREADS), 0, ctx->stream, d_src, n, (uint32_t*)nullptr, ctx->d_static_scratch, Resync{nullptr, 0});
            HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(k_ctx_tiles, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, (const uint32_t*)ctx->d_static_scratch, (uint8_t*)nullptr, (uint32_t)ntiles, (uint32_t)CS_N, d_fn);
        HIP_TRY(hipGetLastError());
        return (int)OPUSFE_OK;
    });
}

// raw (dent-free) tokens of a shard whose first byte is in context `start_state`; *n_tokens and the kinds of its last
// (up to 8) tokens, most recent in the low byte, T_EndOfFile where there is none.  fn_out / head_kinds (may be null): the composed
// context function of the bytes, and the kinds of the FIRST (up to 8) tokens, first token in the low byte.
static int shard_lex_impl(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t len, uint32_t start_state, opusfe_shard** out, uint64_t* n_tokens, uint64_t* tail_kinds,
                          uint32_t* fn_out, uint64_t* head_kinds);
int opusfe_shard_lex(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t len, uint32_t start_state, opusfe_shard** out, uint64_t* n_tokens, uint64_t* tail_kinds) {
    return shard_lex_impl(ctx, d_src, len, start_state, out, n_tokens, tail_kinds, nullptr, nullptr);
}
static int shard_lex_impl(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t len, uint32_t start_state, opusfe_shard** out, uint64_t* n_tokens, uint64_t* tail_kinds,
                          uint32_t* fn_out, uint64_t* head_kinds) {
    if (!out || !n_tokens || !tail_kinds || (len && !d_src) || len > MAX_LEN || start_state > CS_C) return OPUSFE_ERR_ARG;
    *out = nullptr;
    if (fn_out) *fn_out = CTX_IDENTITY;
    if (head_kinds) *head_kinds = TK_NO_PREV;
    return guarded(ctx, [&] {
        opusfe_shard* s = new opusfe_shard();
        s->ctx = ctx; s->d_src = d_src; s->len = (uint32_t)len; s->raw = nullptr; s->T = 0;
        s->depth = nullptr; s->gapval = nullptr; s->prev = nullptr; s->scan_tmp = nullptr; s->val64 = nullptr; s->strpool = nullptr;
        *out = s;
        *n_tokens = 0; *tail_kinds = TK_NO_PREV;
        const uint32_t n = (uint32_t)len;
        hipStream_t st = ctx->stream;
        HIP_TRY(hipMemsetAsync(ctx->d_status, 0, 64 * sizeof(uint32_t), st));
        HIP_TRY(hipMemsetAsync(ctx->d_status + W_UNSUP_POS, 0xFF, sizeof(uint32_t), st));
        if (n == 0) return (int)OPUSFE_OK;
        TempList tmp(ctx);
        const uint64_t nchunks = ((uint64_t)n + LEX_CHUNK - 1) / LEX_CHUNK, ntiles = (nchunks + LEX_THREADS - 1) / LEX_THREADS;
        uint32_t* chunk_fn = tmp.get<uint32_t>(nchunks);
        uint32_t* tile_fn = tmp.get<uint32_t>(ntiles);
        uint8_t* tile_state = tmp.get<uint8_t>(ntiles);
        uint8_t* chunk_state = tmp.get<uint8_t>(nchunks);
        uint32_t* chunk_cnt = tmp.get<uint32_t>(nchunks);
        uint32_t* chunk_str = tmp.get<uint32_t>(nchunks);
        Cnt3* tile_cnt = tmp.get<Cnt3>(ntiles);
        Cnt3* scan_tmp = tmp.get<Cnt3>((ntiles + SCAN_TILE - 1) / SCAN_TILE + 1);
        uint32_t* status = ctx->d_status + W_STATUS;
        LAUNCH(ctx, "lex_ctx_chunk", k_ctx_chunk, ntiles, LEX_THREADS, d_src, n, chunk_fn, tile_fn, Resync{nullptr, 0});
        LAUNCH(ctx, "lex_ctx_tiles", k_ctx_tiles, 1, SCAN_THREADS, tile_fn, tile_state, (uint32_t)ntiles, start_state, ctx->d_status + 60);
        uint64_t* chunk_mask = tmp.get<uint64_t>(nchunks);
        LAUNCH(ctx, "lex_count", k_lex_count, ntiles, LEX_THREADS, d_src, n, chunk_fn, tile_state, chunk_state, chunk_cnt, chunk_str, tile_cnt, status, Resync{nullptr, 0}, chunk_mask);
        device_scan<true>(tile_cnt, tile_cnt, ntiles, OpAddCnt3(), Cnt3{0, 0, 0, 0}, scan_tmp, (Cnt3*)(ctx->d_status + W_LEXTOT), st);
        fetch_status(ctx);
        if (fn_out) *fn_out = ctx->h_status[60];
        const uint32_t T = ctx->h_status[W_LEXTOT], nval = ctx->h_status[W_LEXTOT + 1], nstr = ctx->h_status[W_LEXTOT + 2];
        s->T = T;
        s->raw = dev_alloc_t<opusfe_token>(ctx, T);
        s->val64 = dev_alloc_t<uint64_t>(ctx, nval);
        s->strpool = dev_alloc_t<uint8_t>(ctx, nstr);
        LAUNCH(ctx, "lex_emit", k_lex_emit_tok, ntiles, LEX_THREADS, d_src, n, (const uint32_t*)chunk_cnt, (const uint64_t*)chunk_mask, (const Cnt3*)tile_cnt, s->raw, s->val64, s->strpool, status, Resync{nullptr, 0});
        const uint64_t G = (uint64_t)T + 1;
        s->depth = dev_alloc_t<int32_t>(ctx, G); s->gapval = dev_alloc_t<uint32_t>(ctx, G); s->prev = dev_alloc_t<uint32_t>(ctx, G);
        s->scan_tmp = dev_alloc_t<uint32_t>(ctx, (G + SCAN_TILE - 1) / SCAN_TILE + 2);
        // kinds of the last tokens
        opusfe_token tail[8], head[8];
        const uint32_t nt = T < 8 ? T : 8;
        if (nt) HIP_TRY(hipMemcpyAsync(tail, s->raw + (T - nt), nt * sizeof(opusfe_token), hipMemcpyDeviceToHost, st));
        if (nt && head_kinds) HIP_TRY(hipMemcpyAsync(head, s->raw, nt * sizeof(opusfe_token), hipMemcpyDeviceToHost, st));
        sync_stream(ctx);
        if (head_kinds) {
            uint64_t hk = TK_NO_PREV;
            for (uint32_t j = 0; j < nt; j++) hk = (hk & ~(0xFFull << (8 * j))) | ((uint64_t)head[j].kind << (8 * j));
            *head_kinds = hk;
        }
        uint64_t kinds = TK_NO_PREV;
        for (uint32_t j = 0; j < nt; j++) {
            uint8_t k = tail[nt - 1 - j].kind;
            if (k == T_Error) k = T_String;   // an "unterminated" string at the end of a shard continues in the next one
            kinds = (kinds & ~(0xFFull << (8 * j))) | ((uint64_t)k << (8 * j));
        }
        *n_tokens = T; *tail_kinds = kinds;
        return (int)OPUSFE_OK;
    });
}

// net change of the bracket counter (`ignore_dents`) over the shard, given the kinds of the tokens before it
int opusfe_shard_net_depth(opusfe_shard* s, uint64_t prev_kinds, int64_t* net) {
    if (!s || !net) return OPUSFE_ERR_ARG;
    opusfe_ctx* ctx = s->ctx;
    return guarded(ctx, [&] {
        *net = 0;
        if (s->T == 0) return (int)OPUSFE_OK;
        hipStream_t st = ctx->stream;
        LAUNCH(ctx, "tok_delta", k_tok_delta, grid_for(s->T, 256), 256, s->raw, s->T, s->depth, 0, prev_kinds, FileTok{});
        device_scan<false>(s->depth, s->depth, s->T, OpAddI32(), (int32_t)0, (int32_t*)s->scan_tmp, (int32_t*)(ctx->d_status + 61), st);
        HIP_TRY(hipMemcpyAsync(ctx->h_status + 61, ctx->d_status + 61, 4, hipMemcpyDeviceToHost, st));
        sync_stream(ctx);
        *net = (int32_t)ctx->h_status[61];
        return (int)OPUSFE_OK;
    });
}

// byte position of the first top-level segment start that is provable inside the shard (after opusfe_shard_net_depth
// with the same prev_kinds), or UINT64_MAX when there is none
int opusfe_shard_boundary(opusfe_shard* s, int64_t depth_in, uint64_t* pos) {
    if (!s || !pos) return OPUSFE_ERR_ARG;
    opusfe_ctx* ctx = s->ctx;
    return guarded(ctx, [&] {
        *pos = ~0ull;
        if (s->T == 0) return (int)OPUSFE_OK;
        hipStream_t st = ctx->stream;
        const uint64_t G = (uint64_t)s->T + 1;
        LAUNCH(ctx, "tok_gap", k_gap, grid_for(G, 256), 256, s->d_src, s->len, s->raw, s->T, s->depth, s->gapval, (int32_t)depth_in, FileTok{});
        device_scan<true>(s->gapval, s->prev, G, OpLast(), (uint32_t)0, s->scan_tmp, (uint32_t*)nullptr, st);
        HIP_TRY(hipMemsetAsync(ctx->d_status + 62, 0xFF, 4, st));
        LAUNCH(ctx, "shard_first_segment", k_shard_first_segment, grid_for(s->T, 256), 256, s->raw, s->T, s->gapval, s->prev, ctx->d_status + 62);
        HIP_TRY(hipMemcpyAsync(ctx->h_status + 62, ctx->d_status + 62, 4, hipMemcpyDeviceToHost, st));
        sync_stream(ctx);
        if (ctx->h_status[62] != 0xFFFFFFFFu) *pos = ctx->h_status[62];
        return (int)OPUSFE_OK;
    });
}

// Everything a shard contributes to the ONE carry all-gather of a huge source, computed SPECULATIVELY for the
// common entry of a shard that starts behind a '\n': lexer context N, bracket depth 0, no `record {` look-back across the cut.
//   carry[0] composed context function of the bytes        [1] number of raw tokens      [2] kinds of the last 8 tokens
//   carry[3] kinds of the first 8 tokens                   [4] net bracket depth          [5] first provable top-level segment start
//   carry[6] len                                           [7] bit 0: a brace among the first tokens (its depth delta looks back over the cut)
// Once every rank holds every carry, each can check the speculation of ALL ranks (entry state of rank q = fold of carry[0] of the
// ranks before it; entry depth = sum of carry[4]; no look-back flag) and they all reach the same verdict without another
// exchange; if it fails the host falls back to the dependent chain (opusfe_shard_lex with the true state, _net_depth, _boundary).
// The shard handle stays valid for that chain.  No counterpart in the reference.
int opusfe_shard_carry(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t len, uint64_t* carry, opusfe_shard** out) {
    if (!ctx || !carry || !out) return OPUSFE_ERR_ARG;
    uint32_t fn = CTX_IDENTITY; uint64_t ntok = 0, tail = TK_NO_PREV, head = TK_NO_PREV, cut = ~0ull; int64_t net = 0;
    int rc = shard_lex_impl(ctx, d_src, len, CS_N, out, &ntok, &tail, &fn, &head);
    if (rc == OPUSFE_OK) rc = opusfe_shard_net_depth(*out, TK_NO_PREV, &net);
    if (rc == OPUSFE_OK) rc = opusfe_shard_boundary(*out, 0, &cut);
    if (rc != OPUSFE_OK) return rc;
    uint64_t sensitive = 0;
    for (uint32_t j = 0; j < 8 && j < ntok; j++) { const uint8_t k = (uint8_t)(head >> (8 * j)); if (k == T_LeftBrace || k == T_RightBrace) sensitive = 1; }
    carry[0] = fn; carry[1] = ntok; carry[2] = tail; carry[3] = head; carry[4] = (uint64_t)net; carry[5] = cut; carry[6] = len; carry[7] = sensitive;
    return OPUSFE_OK;
}

void opusfe_shard_free(opusfe_shard* s) {
    if (!s) return;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    shard_release(s);
}

}  // extern "C"

extern "C" {

int opusfe_create(opusfe_ctx** out, int device) {
    if (!out) return OPUSFE_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return OPUSFE_ERR_HIP;   // no GPU: fail loudly, no CPU path
    if (device < 0 || device >= count) return OPUSFE_ERR_ARG;
    opusfe_ctx* ctx = new opusfe_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess
        || hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess
        || hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess
        || hipMalloc((void**)&ctx->d_status, 64 * sizeof(uint32_t)) != hipSuccess
        || hipHostMalloc((void**)&ctx->h_status, 64 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
        delete ctx;
        return OPUSFE_ERR_HIP;
    }
    // the static scratch of opusfe_shard_ctx_static: every device holds its own copy of the array at its own address
    void* sp = nullptr;
    if (hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)) != hipSuccess || !sp) { opusfe_destroy(ctx); return OPUSFE_ERR_HIP; }
    ctx->d_static_scratch = (uint32_t*)sp;
    *out = ctx;
    return OPUSFE_OK;
}

void opusfe_destroy(opusfe_ctx* ctx) {
    if (!ctx) return;
    for (opusfe_ctx* w : ctx->workers) opusfe_destroy(w);
    ctx->workers.clear();
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->free_blocks) (void)hipFree(kv.second);
    for (auto& kv : ctx->live_blocks) (void)hipFree(kv.first);
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    (void)hipFree(ctx->d_status);
    (void)hipFree(ctx->d_pp_table);
    (void)hipHostFree(ctx->h_status);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->arena) (void)hipFree(ctx->arena);
    (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    delete ctx;
}

const char* opusfe_last_message(opusfe_ctx* ctx) { return ctx ? ctx->msg.c_str() : ""; }
void* opusfe_stream(opusfe_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int opusfe_lex_device(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t len, int force_dents, opusfe_unit** out) {
    if (!out || (len && !d_src)) return OPUSFE_ERR_ARG;
    *out = nullptr;
    if (len > MAX_LEN) return fail(ctx, OPUSFE_ERR_ARG, "unit larger than 4 GiB: shard it (opusfe positions are 32-bit per unit)");
    return guarded(ctx, [&] {
        opusfe_unit* u = new_unit(ctx, d_src, len, false);
        *out = u;
        run_lexer(ctx, u, force_dents);
        return u->status;
    });
}

static int with_host_source(opusfe_ctx* ctx, const uint8_t* src, uint64_t len, opusfe_unit** out, int (*body)(opusfe_ctx*, opusfe_unit*, int), int arg) {
    if (!out || (len && !src)) return OPUSFE_ERR_ARG;
    *out = nullptr;
    if (len > MAX_LEN) return fail(ctx, OPUSFE_ERR_ARG, "unit larger than 4 GiB: shard it (opusfe positions are 32-bit per unit)");
    return guarded(ctx, [&] {
        uint8_t* d = dev_alloc_t<uint8_t>(ctx, len + 16);
        if (len) HIP_TRY(hipMemcpyAsync(d, src, len, hipMemcpyHostToDevice, ctx->stream));
        opusfe_unit* u = new_unit(ctx, d, len, true);
        u->h_src.assign(src, src + len); u->have_h_src = true;
        *out = u;
        return body(ctx, u, arg);
    });
}

static int body_lex(opusfe_ctx* ctx, opusfe_unit* u, int force_dents) { run_lexer(ctx, u, force_dents); return u->status; }
// index of the Err token flagged OPUSFE_TF_RESYNC that starts at byte `pos` (one match at most)
__global__ void k_find_resync_token(const opusfe_token* __restrict__ tok, uint32_t n, uint32_t pos, uint32_t* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && tok[i].low == pos && tok[i].kind == T_Error && (tok[i].flags & OPUSFE_TF_RESYNC)) atomicMin(out, i);
}

// After a pass that ended OPUSFE_UNSUPPORTED because the parser dropped an Err token behind which the reference resumes
// lexing inside a string body (Q6/Q7): the byte position where the reference resumes, else NONE.
static uint32_t swallowed_resync_point(opusfe_ctx* ctx, opusfe_unit* u) {
    if (u->status != OPUSFE_UNSUPPORTED || u->dev_status != DS_UNSUPPORTED_RESYNC || u->n_tokens == 0) return OPUSFE_NONE;
    const uint32_t pos = ctx->h_status[W_UNSUP_POS];
    uint32_t* d_idx = ctx->d_status + 63;
    HIP_TRY(hipMemsetAsync(d_idx, 0xFF, 4, ctx->stream));
    hipLaunchKernelGGL(k_find_resync_token, dim3(grid_for(u->n_tokens, 256)), dim3(256), 0, ctx->stream, u->d_tokens, (uint32_t)u->n_tokens, pos, d_idx);
    HIP_TRY(hipGetLastError());
    uint32_t idx = OPUSFE_NONE;
    HIP_TRY(hipMemcpyAsync(&idx, d_idx, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (idx == OPUSFE_NONE) return OPUSFE_NONE;     // the unsupported byte is something else (non-ASCII, ...)
    opusfe_token t{}; uint64_t span = 0;
    HIP_TRY(hipMemcpy(&t, u->d_tokens + idx, sizeof t, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&span, u->d_val64 + t.payload, 8, hipMemcpyDeviceToHost));
    return (uint32_t)span + (uint32_t)(span >> 32);   // the error span ends where the reference resumes (lex.h lx_error callers)
}

static void drop_lex_and_parse(opusfe_ctx* ctx, opusfe_unit* u) {
    dev_free(ctx, u->d_tokens); dev_free(ctx, u->d_val64); dev_free(ctx, u->d_strpool); dev_free(ctx, u->d_seg_start);
    dev_free(ctx, u->d_nodes); dev_free(ctx, u->d_seg_info); dev_free(ctx, u->d_def_nodes); dev_free(ctx, u->d_parents); u->d_parents = nullptr;
    u->d_tokens = nullptr; u->d_val64 = nullptr; u->d_strpool = nullptr; u->d_seg_start = nullptr; u->d_nodes = nullptr; u->d_seg_info = nullptr; u->d_def_nodes = nullptr;
    u->n_tokens = u->n_val64 = u->n_strpool = u->n_segs = u->n_nodes = u->n_defs = 0;
    u->parsed = false; u->has_error = false; u->have_h_lex = u->have_h_ast = false;
}

static int body_parse(opusfe_ctx* ctx, opusfe_unit* u, int) {
    // Normally one pass.  A pass that meets a swallowed Err token behind which the reference lexes a string body as code
    // (malformed input only) yields that resync point; the unit is lexed and parsed again with it (lex.h Resync), up to
    // MAX_RESYNC such points per unit, after which OPUSFE_UNSUPPORTED stands.
    enum { MAX_RESYNC = 16 };
    for (;;) {
        run_lexer(ctx, u, 0);
        if (u->status != OPUSFE_OK && u->status != OPUSFE_UNSUPPORTED) return u->status;
        // an input the lexer cannot follow the reference on is still parsed: a syntax error that the
        // reference reports BEFORE reaching that byte wins (run_parser decides with u->unsup_pos)
        run_parser(ctx, u);
        if (u->resync.size() >= MAX_RESYNC) break;
        const uint32_t r = swallowed_resync_point(ctx, u);
        if (r == OPUSFE_NONE || r > u->len || (!u->resync.empty() && r <= u->resync.back())) break;
        u->resync.push_back(r);
        drop_lex_and_parse(ctx, u);
    }
    if (u->status != OPUSFE_OK) ctx->msg = u->status == OPUSFE_UNSUPPORTED ? status_text(u->dev_status) : u->status == OPUSFE_SYNTAX ? "syntax error" : ctx->msg;
    return u->status;
}

int opusfe_lex(opusfe_ctx* ctx, const uint8_t* src, uint64_t len, int force_dents, opusfe_unit** out) {
    return with_host_source(ctx, src, len, out, body_lex, force_dents);
}
int opusfe_parse(opusfe_ctx* ctx, const uint8_t* src, uint64_t len, opusfe_unit** out) {
    return with_host_source(ctx, src, len, out, body_parse, 0);
}
int opusfe_parse_device(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t len, opusfe_unit** out) {
    if (!out || (len && !d_src)) return OPUSFE_ERR_ARG;
    *out = nullptr;
    if (len > MAX_LEN) return fail(ctx, OPUSFE_ERR_ARG, "unit larger than 4 GiB: shard it (opusfe positions are 32-bit per unit)");
    return guarded(ctx, [&] {
        opusfe_unit* u = new_unit(ctx, d_src, len, false);
        *out = u;
        return body_parse(ctx, u, 0);
    });
}

int opusfe_lower(opusfe_ctx* ctx, opusfe_unit* unit) {
    if (!unit || !ctx) return OPUSFE_ERR_ARG;
    opusfe_ctx* owner = unit->ctx;       // a unit made by opusfe_parse_batch belongs to one of ctx's worker contexts
    if (owner != ctx) {
        bool mine = false;
        for (opusfe_ctx* w : ctx->workers) mine = mine || w == owner;
        if (!mine) return OPUSFE_ERR_ARG;
    }
    if (!unit->parsed || unit->status != OPUSFE_OK) return fail(ctx, OPUSFE_ERR_ARG, "opusfe_lower needs a successfully parsed unit");
    int rc = guarded(owner, [&] { run_lower(owner, unit); return unit->status; });
    if (owner != ctx) ctx->msg = owner->msg;
    return rc;
}

// Corpus of independent files (BASELINE config 5): every file is its own compilation universe like one run of the
// reference (frontend.rs:52-55).  Round-1 implementation: every file goes through the single-source pipeline, several
// files at a time on worker contexts (a segmented multi-file pipeline with one launch per stage is DESIGN.md §8 work).
int opusfe_parse_batch(opusfe_ctx* ctx, uint32_t n, const uint8_t* const* srcs, const uint64_t* lens, int lower, opusfe_unit** units, int* statuses) {
    if (!ctx || (n && (!srcs || !lens || !units || !statuses))) return OPUSFE_ERR_ARG;
    // A small file keeps only a few waves busy and its per-function / per-segment machines are sequential, so one file
    // after the other is latency-bound (milliseconds per file whatever its size).  Files are independent: W host threads,
    // each with a context (stream, allocator) of its own, run the single-source pipeline on different files at once.
    uint32_t want = 8;
    if (const char* e = getenv("OPUSFE_BATCH_WORKERS")) { int v = atoi(e); if (v >= 1 && v <= 64) want = (uint32_t)v; }
    const uint32_t W = n < want ? (n ? n : 1) : want;
    while (ctx->workers.size() + 1 < W) {
        opusfe_ctx* w = nullptr;
        int rc = opusfe_create(&w, ctx->device);
        if (rc != OPUSFE_OK) { ctx->msg = "opusfe_parse_batch: cannot create a worker context"; return rc; }
        w->parser_mode = ctx->parser_mode;
        ctx->workers.push_back(w);
    }
    for (uint32_t i = 0; i < n; i++) { units[i] = nullptr; statuses[i] = OPUSFE_ERR_INTERNAL; }
    std::atomic<uint32_t> next{0};
    auto work = [&](opusfe_ctx* c) {
        (void)hipSetDevice(c->device);
        for (;;) {
            const uint32_t i = next.fetch_add(1);
            if (i >= n) break;
            int rc = opusfe_parse(c, srcs[i], lens[i], &units[i]);
            if (rc == OPUSFE_OK && lower) rc = opusfe_lower(c, units[i]);
            statuses[i] = rc;
        }
    };
    std::vector<std::thread> threads;
    for (uint32_t w = 1; w < W; w++) threads.emplace_back(work, ctx->workers[w - 1]);
    work(ctx);
    for (auto& t : threads) t.join();
    int worst = OPUSFE_OK;
    for (uint32_t i = 0; i < n; i++) if (statuses[i] >= OPUSFE_ERR_HIP && statuses[i] != OPUSFE_SEMANTIC) worst = statuses[i];
    if (worst != OPUSFE_OK) ctx->msg = "opusfe_parse_batch: a file failed with a device / argument error (see statuses)";
    return worst;
}

// own ⊕ hdr (join.h): one parsed unit whose definitions are hdr's definitions that start before byte `split_pos` of hdr's
// source, then own's, then the rest of hdr's.  Neither input is modified; the result owns all of its arrays.
int opusfe_unit_join(opusfe_ctx* ctx, opusfe_unit* own, opusfe_unit* hdr, uint64_t split_pos, opusfe_unit** out) {
    if (!ctx || !own || !hdr || !out) return OPUSFE_ERR_ARG;
    *out = nullptr;
    if (own->ctx != ctx || hdr->ctx != ctx) return OPUSFE_ERR_ARG;
    if (!own->parsed || own->status != OPUSFE_OK || !hdr->parsed || hdr->status != OPUSFE_OK) return fail(ctx, OPUSFE_ERR_ARG, "opusfe_unit_join needs two successfully parsed units");
    return guarded(ctx, [&] {
        const uint64_t a_len = (own->len + 63) & ~(uint64_t)63;          // H's source starts on a fresh 64-byte line
        const uint64_t len = a_len + hdr->len, n_tok = own->n_tokens + hdr->n_tokens, n_nodes = own->n_nodes + hdr->n_nodes;
        const uint64_t n_defs = own->n_defs + hdr->n_defs, n_val = own->n_val64 + hdr->n_val64, n_str = own->n_strpool + hdr->n_strpool;
        if (len >= 0xFFFF0000ull || n_tok >= 0xFFFFFFF0ull || n_nodes >= 0xFFFFFFF0ull) return fail(ctx, OPUSFE_ERR_ARG, "joined unit too large for 32-bit positions");
        hipStream_t st = ctx->stream;
        opusfe_unit* u = new opusfe_unit();
        u->ctx = ctx; u->status = OPUSFE_OK; u->len = len; u->own_src = true;
        u->n_tokens = n_tok; u->n_nodes = n_nodes; u->n_defs = n_defs; u->n_val64 = n_val; u->n_strpool = n_str;
        uint8_t* src = dev_alloc_t<uint8_t>(ctx, len + 64);
        u->d_src = src;
        u->d_tokens = dev_alloc_t<opusfe_token>(ctx, n_tok + 1);
        u->d_val64 = dev_alloc_t<uint64_t>(ctx, n_val + 1);
        u->d_strpool = dev_alloc_t<uint8_t>(ctx, n_str + 16);
        u->d_nodes = dev_alloc_t<opusfe_node>(ctx, n_nodes + 1);
        u->d_def_nodes = dev_alloc_t<uint32_t>(ctx, n_defs + 1);
        try {
            HIP_TRY(hipMemsetAsync(src, '\n', len + 64, st));
            HIP_TRY(hipMemcpyAsync(src, own->d_src, own->len, hipMemcpyDeviceToDevice, st));
            if (hdr->len) HIP_TRY(hipMemcpyAsync(src + a_len, hdr->d_src, hdr->len, hipMemcpyDeviceToDevice, st));
            if (own->n_tokens) HIP_TRY(hipMemcpyAsync(u->d_tokens, own->d_tokens, own->n_tokens * sizeof(opusfe_token), hipMemcpyDeviceToDevice, st));
            if (own->n_val64) HIP_TRY(hipMemcpyAsync(u->d_val64, own->d_val64, own->n_val64 * 8, hipMemcpyDeviceToDevice, st));
            if (hdr->n_val64) HIP_TRY(hipMemcpyAsync(u->d_val64 + own->n_val64, hdr->d_val64, hdr->n_val64 * 8, hipMemcpyDeviceToDevice, st));
            if (own->n_strpool) HIP_TRY(hipMemcpyAsync(u->d_strpool, own->d_strpool, own->n_strpool, hipMemcpyDeviceToDevice, st));
            if (hdr->n_strpool) HIP_TRY(hipMemcpyAsync(u->d_strpool + own->n_strpool, hdr->d_strpool, hdr->n_strpool, hipMemcpyDeviceToDevice, st));
            TempList tmp(ctx);
            JoinSplit* split = tmp.get<JoinSplit>(1);
            LAUNCH(ctx, "join_split", k_join_split, 1, 64, hdr->d_nodes, hdr->d_def_nodes, (uint32_t)hdr->n_defs, (uint32_t)hdr->n_nodes,
                   (uint32_t)(split_pos > hdr->len ? hdr->len : split_pos), split);
            if (hdr->n_tokens)
                LAUNCH(ctx, "join_tokens", k_join_tokens, grid_for(hdr->n_tokens, 256), 256, hdr->d_tokens, (uint32_t)hdr->n_tokens, u->d_tokens + own->n_tokens,
                       (uint32_t)a_len, (uint32_t)own->n_val64, (uint32_t)own->n_strpool);
            if (n_nodes)
                LAUNCH(ctx, "join_nodes", k_join_nodes, grid_for(n_nodes, 256), 256, own->d_nodes, (uint32_t)own->n_nodes, hdr->d_nodes, (uint32_t)hdr->n_nodes,
                       split, (uint32_t)own->n_tokens, (uint32_t)a_len, u->d_nodes);
            if (n_defs)
                LAUNCH(ctx, "join_defs", k_join_defs, grid_for(n_defs, 256), 256, own->d_def_nodes, (uint32_t)own->n_defs, (uint32_t)own->n_nodes,
                       hdr->d_def_nodes, (uint32_t)hdr->n_defs, split, u->d_def_nodes);
            sync_stream(ctx);
        } catch (...) { opusfe_free_unit(u); throw; }
        u->parsed = true;
        *out = u;
        return (int)OPUSFE_OK;
    });
}

int opusfe_unit_from_arrays(opusfe_ctx* ctx, const uint8_t* d_src, uint64_t src_len, const opusfe_token* d_tokens, uint64_t n_tokens,
                            const uint64_t* d_val64, uint64_t n_val64, const uint8_t* d_strings, uint64_t n_strings,
                            const opusfe_node* d_nodes, uint64_t n_nodes, const uint32_t* d_defs, uint64_t n_defs, opusfe_unit** out) {
    if (!ctx || !out) return OPUSFE_ERR_ARG;
    *out = nullptr;
    if ((src_len && !d_src) || (n_tokens && !d_tokens) || (n_val64 && !d_val64) || (n_strings && !d_strings) || (n_nodes && !d_nodes) || (n_defs && !d_defs)) return OPUSFE_ERR_ARG;
    if (src_len > MAX_LEN || n_tokens >= 0xFFFFFFF0ull || n_nodes >= 0xFFFFFFF0ull || n_defs > n_nodes) return fail(ctx, OPUSFE_ERR_ARG, "opusfe_unit_from_arrays: sizes out of range");
    return guarded(ctx, [&] {
        hipStream_t st = ctx->stream;
        opusfe_unit* u = new opusfe_unit();
        u->ctx = ctx; u->status = OPUSFE_OK; u->len = src_len; u->own_src = true;
        u->n_tokens = n_tokens; u->n_val64 = n_val64; u->n_strpool = n_strings; u->n_nodes = n_nodes; u->n_defs = n_defs;
        uint8_t* src = dev_alloc_t<uint8_t>(ctx, src_len + 64);
        u->d_src = src;
        u->d_tokens = dev_alloc_t<opusfe_token>(ctx, n_tokens + 1);
        u->d_val64 = dev_alloc_t<uint64_t>(ctx, n_val64 + 1);
        u->d_strpool = dev_alloc_t<uint8_t>(ctx, n_strings + 16);
        u->d_nodes = dev_alloc_t<opusfe_node>(ctx, n_nodes + 1);
        u->d_def_nodes = dev_alloc_t<uint32_t>(ctx, n_defs + 1);
        try {
            HIP_TRY(hipMemsetAsync(src + src_len, '\n', 64, st));
            if (src_len) HIP_TRY(hipMemcpyAsync(src, d_src, src_len, hipMemcpyDeviceToDevice, st));
            if (n_tokens) HIP_TRY(hipMemcpyAsync(u->d_tokens, d_tokens, n_tokens * sizeof(opusfe_token), hipMemcpyDeviceToDevice, st));
            if (n_val64) HIP_TRY(hipMemcpyAsync(u->d_val64, d_val64, n_val64 * 8, hipMemcpyDeviceToDevice, st));
            if (n_strings) HIP_TRY(hipMemcpyAsync(u->d_strpool, d_strings, n_strings, hipMemcpyDeviceToDevice, st));
            if (n_nodes) HIP_TRY(hipMemcpyAsync(u->d_nodes, d_nodes, n_nodes * sizeof(opusfe_node), hipMemcpyDeviceToDevice, st));
            if (n_defs) HIP_TRY(hipMemcpyAsync(u->d_def_nodes, d_defs, n_defs * 4, hipMemcpyDeviceToDevice, st));
            sync_stream(ctx);
        } catch (...) { opusfe_free_unit(u); throw; }
        u->parsed = true;
        *out = u;
        return (int)OPUSFE_OK;
    });
}

// Text another shard needs to know about this unit's definitions (multi-GPU IR generation): device buffer owned by
// the unit.  Functions appear as `extern <signature>`, everything else verbatim, one definition per line, file order.
int opusfe_unit_headers(opusfe_unit* u, const uint8_t** d_out, uint64_t* len) {
    if (!u || !d_out || !len) return OPUSFE_ERR_ARG;
    if (!u->parsed || u->status != OPUSFE_OK) return OPUSFE_ERR_ARG;
    opusfe_ctx* ctx = u->ctx;
    return guarded(ctx, [&] {
        if (!u->d_headers) {
            const uint32_t n_defs = (uint32_t)u->n_defs;
            u->n_headers = 0;
            if (n_defs) {
                TempList tmp(ctx);
                uint32_t* off = tmp.get<uint32_t>(n_defs);
                uint32_t* scan_tmp = tmp.get<uint32_t>((n_defs + SCAN_TILE - 1) / SCAN_TILE + 2);
                LAUNCH(ctx, "hdr_len", k_hdr_len, grid_for(n_defs, 256), 256, u->d_nodes, u->d_def_nodes, n_defs, off);
                device_scan<true>(off, off, n_defs, OpAddU32(), (uint32_t)0, scan_tmp, ctx->d_status + 63, ctx->stream);
                HIP_TRY(hipMemcpyAsync(ctx->h_status + 63, ctx->d_status + 63, 4, hipMemcpyDeviceToHost, ctx->stream));
                sync_stream(ctx);
                u->n_headers = ctx->h_status[63];
                u->d_headers = dev_alloc_t<uint8_t>(ctx, u->n_headers + 16);
                unsigned grid = (unsigned)((n_defs + 3) / 4); if (grid > 4096) grid = 4096;
                LAUNCH(ctx, "hdr_copy", k_hdr_copy, grid, 256, u->d_src, u->d_nodes, u->d_def_nodes, n_defs, off, u->d_headers);
                sync_stream(ctx);
            } else u->d_headers = dev_alloc_t<uint8_t>(ctx, 16);
        }
        *d_out = u->d_headers; *len = u->n_headers;
        return (int)OPUSFE_OK;
    });
}

int opusfe_semantic_error_count(const opusfe_unit* u, uint64_t* n) {
    if (!u || !n) return OPUSFE_ERR_ARG;
    *n = u->ir ? ((const IrResult*)u->ir)->n_sem_errors : 0;
    return OPUSFE_OK;
}

int opusfe_lowering_stats(const opusfe_unit* u, uint64_t* n_parallel, uint64_t* n_sequential) {
    if (!u || !n_parallel || !n_sequential) return OPUSFE_ERR_ARG;
    *n_parallel = *n_sequential = 0;
    const IrResult* R = (const IrResult*)u->ir;
    if (!R || !u->lowered || u->status != OPUSFE_OK) return OPUSFE_OK;
    *n_sequential = R->n_fallback;
    *n_parallel = R->n_bodies - R->n_fallback;
    return OPUSFE_OK;
}

int opusfe_parsing_stats(const opusfe_unit* u, uint64_t* n_parallel, uint64_t* n_sequential) {
    if (!u || !n_parallel || !n_sequential) return OPUSFE_ERR_ARG;
    *n_parallel = *n_sequential = 0;
    if (!u->parsed) return OPUSFE_OK;
    *n_sequential = u->n_seg_sequential;
    *n_parallel = u->n_segs - u->n_seg_sequential;
    return OPUSFE_OK;
}

int opusfe_set_parser_mode(opusfe_ctx* ctx, int mode) {
    if (!ctx || (mode != 0 && mode != 1)) return OPUSFE_ERR_ARG;
    ctx->parser_mode = mode;
    for (opusfe_ctx* w : ctx->workers) w->parser_mode = mode;
    return OPUSFE_OK;
}

int opusfe_memory_held(opusfe_ctx* ctx, uint64_t* bytes) {
    if (!ctx || !bytes) return OPUSFE_ERR_ARG;
    *bytes = ctx->bytes_allocated;
    return OPUSFE_OK;
}

int opusfe_memory_stats(opusfe_ctx* ctx, uint64_t* stats) {
    if (!ctx || !stats) return OPUSFE_ERR_ARG;
    uint64_t live = 0, cached = 0;
    for (auto& kv : ctx->live_blocks) live += kv.second;
    for (auto& kv : ctx->free_blocks) cached += kv.first;
    stats[0] = live; stats[1] = cached; stats[2] = ctx->arena_size;
    stats[3] = ctx->bytes_peak > ctx->bytes_allocated ? ctx->bytes_peak : ctx->bytes_allocated;
    return OPUSFE_OK;
}

int opusfe_unit_resync_passes(const opusfe_unit* u, uint32_t* n) {
    if (!u || !n) return OPUSFE_ERR_ARG;
    *n = (uint32_t)u->resync.size();
    return OPUSFE_OK;
}

int opusfe_unit_end_state(const opusfe_unit* u, uint32_t* ctx_state, int32_t* depth) {
    if (!u || !ctx_state || !depth) return OPUSFE_ERR_ARG;
    *ctx_state = u->end_ctx_state; *depth = u->end_depth;
    return OPUSFE_OK;
}

int opusfe_unit_status(const opusfe_unit* unit) { return unit ? unit->status : OPUSFE_ERR_ARG; }

int opusfe_get_counts(const opusfe_unit* u, opusfe_counts* out) {
    if (!u || !out) return OPUSFE_ERR_ARG;
    out->source_bytes = u->len; out->tokens = u->n_tokens; out->nodes = u->n_nodes; out->definitions = u->n_defs;
    out->functions = u->n_funcs; out->instrs = u->n_instrs; out->vars = u->n_vars; out->bbs = u->n_bbs; out->types = u->n_types;
    return OPUSFE_OK;
}

int opusfe_get_error(const opusfe_unit* u, opusfe_error* out) {
    if (!u || !out) return OPUSFE_ERR_ARG;
    if (!u->has_error) return OPUSFE_ERR_ARG;
    *out = u->error;
    return OPUSFE_OK;
}

int opusfe_copy_tokens(opusfe_unit* u, opusfe_token* dst, uint64_t cap, uint64_t* n) {
    if (!u || !n) return OPUSFE_ERR_ARG;
    *n = u->n_tokens;
    if (!dst) return OPUSFE_OK;
    if (cap < u->n_tokens) return OPUSFE_ERR_ARG;
    return guarded(u->ctx, [&] {
        if (u->n_tokens) HIP_TRY(hipMemcpy(dst, u->d_tokens, u->n_tokens * sizeof(opusfe_token), hipMemcpyDeviceToHost));
        return (int)OPUSFE_OK;
    });
}

static_assert(sizeof(opusfe_function_ir) == sizeof(FuncIr) && sizeof(opusfe_type) == sizeof(TypeRec) && sizeof(opusfe_signature) == sizeof(FuncRec)
              && sizeof(opusfe_global) == sizeof(GlobalRec) && sizeof(opusfe_semantic_error) == sizeof(SemErr), "public records mirror the device records");

int opusfe_array(opusfe_unit* u, int which, const void** d_ptr, uint64_t* count, uint32_t* elem_size) {
    if (!u || which < 0 || which >= OPUSFE_A_COUNT) return OPUSFE_ERR_ARG;
    // the IR-side arrays exist only after a successful opusfe_lower (finalize errors leave no IR, frontend.rs:145)
    const IrResult* R = u->lowered && u->status == OPUSFE_OK ? (const IrResult*)u->ir : nullptr;
    const IrResult* RS = u->lowered && u->ir && (u->status == OPUSFE_OK || u->status == OPUSFE_SEMANTIC) ? (const IrResult*)u->ir : nullptr;
    const void* p = nullptr; uint64_t n = 0; uint32_t es = 0;
    switch (which) {
        case OPUSFE_A_SOURCE: p = u->d_src; n = u->len; es = 1; break;
        case OPUSFE_A_TOKENS: p = u->d_tokens; n = u->n_tokens; es = sizeof(opusfe_token); break;
        case OPUSFE_A_VAL64: p = u->d_val64; n = u->n_val64; es = 8; break;
        case OPUSFE_A_STRINGS: p = u->d_strpool; n = u->n_strpool; es = 1; break;
        case OPUSFE_A_NODES: p = u->d_nodes; n = u->n_nodes; es = sizeof(opusfe_node); break;
        case OPUSFE_A_DEFINITIONS: p = u->d_def_nodes; n = u->n_defs; es = 4; break;
        case OPUSFE_A_SIGNATURES: es = sizeof(FuncRec); if (R) { p = R->d_funcrecs; n = R->n_funcs; } break;
        case OPUSFE_A_GLOBALS: es = sizeof(GlobalRec); if (R) { p = R->d_globals; n = R->n_globals; } break;
        // (types and the type of every type node also after OPUSFE_SEMANTIC: the finalize-time diagnostics name them)
        case OPUSFE_A_TYPES: es = sizeof(TypeRec); if (RS) { p = RS->d_types; n = RS->n_types; } break;
        case OPUSFE_A_NODE_TYPES: es = 4; if (RS) { p = RS->d_tid_of_node; n = u->n_nodes; } break;
        case OPUSFE_A_FUNCTIONS: es = sizeof(FuncIr); if (R) { p = R->d_funcs; n = R->n_funcs; } break;
        case OPUSFE_A_VARIABLES: es = sizeof(opusfe_variable); if (R) { p = R->d_vars; n = R->n_vars; } break;
        case OPUSFE_A_BB_OFFSETS: es = 4; if (R) { p = R->d_bb_offsets; n = R->n_bbs + 1; } break;
        case OPUSFE_A_INSTRS: es = sizeof(opusfe_instr); if (R) { p = R->d_instrs; n = R->n_instrs; } break;
        case OPUSFE_A_CALL_ARGS: es = 4; if (R) { p = R->d_cargs; n = R->n_cargs; } break;
        case OPUSFE_A_SEM_ERRORS:   // also after OPUSFE_SEMANTIC: the finalize-time diagnostics
            es = sizeof(SemErr);
            if (u->lowered && u->ir && (u->status == OPUSFE_OK || u->status == OPUSFE_SEMANTIC)) { p = ((const IrResult*)u->ir)->d_errs; n = ((const IrResult*)u->ir)->n_errs; }
            break;
        case OPUSFE_A_PARENTS: {
            es = 4;
            if (!u->parsed || !u->d_nodes || !u->n_nodes || (u->status != OPUSFE_OK && u->status != OPUSFE_SEMANTIC)) break;
            if (!u->d_parents) {
                int rc = guarded(u->ctx, [&] {
                    opusfe_ctx* ctx = u->ctx;
                    uint32_t* d = dev_alloc_t<uint32_t>(ctx, u->n_nodes);
                    HIP_TRY(hipMemsetAsync(d, 0xFF, u->n_nodes * 4, ctx->stream));
                    LAUNCH(ctx, "ast_parents", k_ast_parents, grid_for(u->n_nodes, 256), 256, (const opusfe_node*)u->d_nodes, (uint32_t)u->n_nodes, d);
                    HIP_TRY(hipStreamSynchronize(ctx->stream));
                    u->d_parents = d;
                    return (int)OPUSFE_OK;
                });
                if (rc != OPUSFE_OK) return rc;
            }
            p = u->d_parents; n = u->n_nodes;
            break;
        }
    }
    if (!p) n = 0;   // a stage that did not run (or stopped early) has no array
    if (d_ptr) *d_ptr = n ? p : nullptr;
    if (count) *count = n;
    if (elem_size) *elem_size = es;
    return OPUSFE_OK;
}

int opusfe_copy_array(opusfe_unit* u, int which, void* dst, uint64_t cap, uint64_t* n) {
    const void* p = nullptr; uint64_t cnt = 0; uint32_t es = 0;
    int rc = opusfe_array(u, which, &p, &cnt, &es);
    if (rc != OPUSFE_OK || !n) return OPUSFE_ERR_ARG;
    *n = cnt;
    if (!dst) return OPUSFE_OK;
    if (cap < cnt) return OPUSFE_ERR_ARG;
    return guarded(u->ctx, [&] {
        HIP_TRY(hipStreamSynchronize(u->ctx->stream));
        if (cnt) HIP_TRY(hipMemcpy(dst, p, cnt * es, hipMemcpyDeviceToHost));
        return (int)OPUSFE_OK;
    });
}

int opusfe_dump(opusfe_unit* u, int what, char** out, uint64_t* out_len) {
    if (!u || !out) return OPUSFE_ERR_ARG;
    *out = nullptr;
    return guarded(u->ctx, [&] {
        std::string s;
        ensure_host_lex(u);
        switch (what) {
            case 0: {
                // after a syntax error only the tokens the reference had lexed are meaningful
                uint64_t limit = u->has_error ? (uint64_t)u->error.tok_index + 1 : ~0ull;
                s = dump_tokens(u, limit);
                break;
            }
            case 1: if (!u->parsed || (u->status != OPUSFE_OK && u->status != OPUSFE_SEMANTIC) || (u->n_nodes && !u->d_nodes)) { u->ctx->msg = "no AST: the unit did not parse"; return (int)OPUSFE_ERR_ARG; }
                    ensure_host_ast(u); s = dump_ast(u); break;
            case 2: ensure_host_ast(u); ensure_host_ir(u); s = dump_ir(u); break;
            case 4: ensure_host_ir(u); s = dump_semantic_errors(u); break;
            case 3: s = dump_error_line(u); break;
            default: return (int)OPUSFE_ERR_ARG;
        }
        char* p = (char*)malloc(s.size() + 1);
        if (!p) return (int)OPUSFE_ERR_OOM;
        memcpy(p, s.data(), s.size()); p[s.size()] = 0;
        *out = p;
        if (out_len) *out_len = s.size();
        return (int)OPUSFE_OK;
    });
}

void opusfe_free_text(char* p) { free(p); }

void opusfe_free_unit(opusfe_unit* u) {
    if (!u) return;
    opusfe_ctx* ctx = u->ctx;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (u->own_src) dev_free(ctx, (void*)u->d_src);
    dev_free(ctx, u->d_tokens); dev_free(ctx, u->d_val64); dev_free(ctx, u->d_strpool); dev_free(ctx, u->d_seg_start);
    dev_free(ctx, u->d_nodes); dev_free(ctx, u->d_seg_info); dev_free(ctx, u->d_def_nodes); dev_free(ctx, u->d_headers); dev_free(ctx, u->d_parents);
    free_ir(ctx, u);
    delete u;
}

}  // extern "C"

// ---- a corpus of independent files as ONE packed unit (BASELINE config 5) ---------------------------------------
// Every file is its own compilation universe, like one run of the reference (frontend.rs:52-55).  opusfe_parse_batch gives every
// file a pipeline of its own: ~40 launches and half a dozen host synchronisations per file, latency-bound whatever the GPU.  Here
// the files are packed into one buffer (every file starts on a 64-byte boundary, '\n' in between) and go through the single-source
// kernels TOGETHER: one launch per stage for the whole group.  What keeps the files apart:
//   lexer     a file start resets the context automaton and is a token boundary (lex.h Resync in group mode); every file is lexed
//             with its own end as the end of the source; Indent / Dedent synthesis, bracket depth and the EndOfFile token are per file
//   parser    a file starts a top-level segment; the lowest failing segment of a file holds the file's first syntax error
//   names     the payload of every name token (identifier hash, operator symbol) is shifted by file * FILE_MIX (defs.h) and a symbol
//             key carries the mix of the function that looks it up (sem.h NameKey), so the name tables of sem.h scope every name to
//             its file without knowing about files
//   the rest  a file the packed pass cannot answer for exactly (FF_SINGLE: empty, indented first line, unsupported input, swallowed
//             lexical errors, finalize-time semantic errors) is compiled alone by the single-source pipeline
// The result is the packed unit plus a table per file (status, first error, token / definition / function ranges); the files that
// took the single-file path carry a unit of their own.
struct opusfe_group {
    opusfe_ctx* ctx = nullptr;
    opusfe_unit* unit = nullptr;
    GroupInfo info;
    bool lowered = false;
    std::vector<uint32_t> start, len, flags, tok_begin, first_def, first_func, first_global, err_seg;
    std::vector<int> status;
    std::vector<opusfe_error> error;
    std::vector<opusfe_unit*> single;
    uint64_t n_single = 0;
};

namespace {

__global__ void k_group_init(uint32_t* __restrict__ fstatus, uint32_t n_files) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n_files) { fstatus[2 * f] = 0; fstatus[2 * f + 1] = 0xFFFFFFFFu; }
}
// line number of the first error of every faulty file: 1 + the '\n' before it, counted from the file's first byte (frontend.rs:213-225)
__global__ void k_group_err_lines(const uint8_t* __restrict__ src, const uint32_t* __restrict__ lo, const uint32_t* __restrict__ hi, uint32_t n, uint32_t* __restrict__ lines) {
    const uint32_t i = blockIdx.x;
    if (i >= n) return;
    uint32_t c = 0;
    for (uint32_t p = lo[i] + threadIdx.x; p < hi[i]; p += blockDim.x) c += src[p] == '\n';
    if (c) atomicAdd(&lines[i], c);
}

void group_release(opusfe_group* g) {
    opusfe_ctx* ctx = g->ctx;
    for (opusfe_unit* u : g->single) if (u) opusfe_free_unit(u);
    if (g->unit) { g->unit->group = nullptr; opusfe_free_unit(g->unit); }
    GroupInfo& I = g->info;
    dev_free(ctx, I.d_start); dev_free(ctx, I.d_len); dev_free(ctx, I.d_flags); dev_free(ctx, I.d_fstatus); dev_free(ctx, I.d_tok_begin);
    dev_free(ctx, I.d_err_seg); dev_free(ctx, I.d_first_def); dev_free(ctx, I.d_first_func); dev_free(ctx, I.d_first_global);
    delete g;
}

// NONE entries of a "first x of file f" table: the file has none, so its range is empty at the next file's start
void fill_forward(std::vector<uint32_t>& v, uint32_t end) {
    uint32_t next = end;
    for (size_t i = v.size(); i-- > 0;) { if (v[i] == OPUSFE_NONE) v[i] = next; else next = v[i]; }
    v.push_back(end);
}

// the packed pass; returns false when the group as a whole has to take the single-file path (a status the packed unit cannot attribute to a file)
bool group_packed_pass(opusfe_ctx* ctx, opusfe_group* g, int lower) {
    opusfe_unit* u = g->unit;
    GroupInfo& I = g->info;
    const uint32_t F = I.n_files;
    hipStream_t st = ctx->stream;
    auto declined = [&](const char* stage) {
        ctx->msg = std::string("group: the packed pass stopped in the ") + stage + " (status " + std::to_string(u->status) + ", device status " + std::to_string(u->dev_status) + "): every file takes the single-file path. " + ctx->msg;
        return false;
    };
    run_lexer(ctx, u, 0);
    if (u->status != OPUSFE_OK || u->n_tokens == 0) return declined("lexer");
    run_parser(ctx, u);
    if (u->status != OPUSFE_OK) return declined("parser");
    if (lower) {
        run_lower(ctx, u);
        if (u->status != OPUSFE_OK) return declined("lowering");
        g->lowered = true;
    }
    auto pull = [&](std::vector<uint32_t>& v, const uint32_t* d) { v.resize(F); HIP_TRY(hipMemcpyAsync(v.data(), d, (size_t)F * 4, hipMemcpyDeviceToHost, st)); };
    pull(g->flags, I.d_flags); pull(g->tok_begin, I.d_tok_begin); pull(g->err_seg, I.d_err_seg);
    if (lower) { pull(g->first_def, I.d_first_def); pull(g->first_func, I.d_first_func); pull(g->first_global, I.d_first_global); }
    HIP_TRY(hipStreamSynchronize(st));
    fill_forward(g->tok_begin, (uint32_t)u->n_tokens);
    if (lower) {
        const IrResult* R = (const IrResult*)u->ir;
        fill_forward(g->first_def, (uint32_t)u->n_defs); fill_forward(g->first_func, R ? (uint32_t)R->n_funcs : 0u); fill_forward(g->first_global, R ? (uint32_t)R->n_globals : 0u);
    }
    // first syntax error of every faulty file that stays on the packed path (parse.rs:352-360 as data, positions relative to the file)
    std::vector<uint32_t> bad;
    for (uint32_t f = 0; f < F; f++) if ((g->flags[f] & FF_SYNTAX) && !(g->flags[f] & FF_SINGLE) && g->err_seg[f] != OPUSFE_NONE) bad.push_back(f);
    if (!bad.empty()) {
        std::vector<uint32_t> lo(bad.size()), hi(bad.size()), lines(bad.size(), 0);
        for (size_t i = 0; i < bad.size(); i++) {
            const uint32_t f = bad[i];
            uint32_t si[SI_WORDS];
            HIP_TRY(hipMemcpy(si, u->d_seg_info + (uint64_t)g->err_seg[f] * SI_WORDS, sizeof si, hipMemcpyDeviceToHost));
            opusfe_token t{};
            HIP_TRY(hipMemcpy(&t, u->d_tokens + si[SI_ERR_TOK], sizeof t, hipMemcpyDeviceToHost));
            opusfe_error e{};
            e.tok_index = si[SI_ERR_TOK];
            if (si[SI_ERR_KIND] & 0x100u) {      // a lexical error token the parser consumed: its own span and payload
                uint64_t v[2];
                HIP_TRY(hipMemcpy(v, u->d_val64 + t.payload, sizeof v, hipMemcpyDeviceToHost));
                e.kind = si[SI_ERR_KIND] & 0xFFu; e.low = (uint32_t)v[0]; e.len = (uint32_t)(v[0] >> 32); e.payload = v[1]; e.tok_kind = T_Error;
            } else { e.kind = si[SI_ERR_KIND]; e.low = si[SI_ERR_LOW]; e.len = si[SI_ERR_LEN]; e.expected = si[SI_ERR_EXPECTED]; e.tok_kind = t.kind; }
            lo[i] = g->start[f];
            hi[i] = (uint32_t)std::min<uint64_t>(e.low, (uint64_t)g->start[f] + g->len[f]);
            if (hi[i] < lo[i]) hi[i] = lo[i];
            e.low -= g->start[f];
            g->error[f] = e;
        }
        TempList tmp(ctx);
        uint32_t* d_lo = tmp.get<uint32_t>(bad.size()); uint32_t* d_hi = tmp.get<uint32_t>(bad.size()); uint32_t* d_lines = tmp.get<uint32_t>(bad.size());
        HIP_TRY(hipMemcpyAsync(d_lo, lo.data(), bad.size() * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_hi, hi.data(), bad.size() * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(d_lines, 0, bad.size() * 4, st));
        LAUNCH(ctx, "group_err_lines", k_group_err_lines, bad.size(), 256, u->d_src, (const uint32_t*)d_lo, (const uint32_t*)d_hi, (uint32_t)bad.size(), d_lines);
        HIP_TRY(hipMemcpyAsync(lines.data(), d_lines, bad.size() * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        prof_collect(ctx);
        for (size_t i = 0; i < bad.size(); i++) g->error[bad[i]].line = 1 + lines[i];
    }
    for (uint32_t f = 0; f < F; f++) g->status[f] = (g->flags[f] & FF_SINGLE) ? OPUSFE_ERR_INTERNAL : (g->flags[f] & FF_SYNTAX) ? OPUSFE_SYNTAX : OPUSFE_OK;
    return true;
}

int group_run(opusfe_ctx* ctx, opusfe_group* g, const uint8_t* d_packed, uint64_t packed_len, bool own_src, const uint8_t* const* host_srcs, int lower) {
    GroupInfo& I = g->info;
    const uint32_t F = I.n_files;
    hipStream_t st = ctx->stream;
    g->unit = new_unit(ctx, d_packed, packed_len, own_src);      // first: from here on the group owns the packed buffer
    g->unit->group = &I;
    I.d_start = dev_alloc_t<uint32_t>(ctx, F); I.d_len = dev_alloc_t<uint32_t>(ctx, F); I.d_flags = dev_alloc_t<uint32_t>(ctx, F);
    I.d_fstatus = dev_alloc_t<uint32_t>(ctx, 2ull * F); I.d_tok_begin = dev_alloc_t<uint32_t>(ctx, F); I.d_err_seg = dev_alloc_t<uint32_t>(ctx, F);
    I.d_first_def = dev_alloc_t<uint32_t>(ctx, F); I.d_first_func = dev_alloc_t<uint32_t>(ctx, F); I.d_first_global = dev_alloc_t<uint32_t>(ctx, F);
    HIP_TRY(hipMemcpyAsync(I.d_start, g->start.data(), (size_t)F * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(I.d_len, g->len.data(), (size_t)F * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(I.d_flags, 0, (size_t)F * 4, st));
    for (uint32_t* p : {I.d_tok_begin, I.d_err_seg, I.d_first_def, I.d_first_func, I.d_first_global}) HIP_TRY(hipMemsetAsync(p, 0xFF, (size_t)F * 4, st));
    hipLaunchKernelGGL(k_group_init, dim3(grid_for(F, 256)), dim3(256), 0, st, I.d_fstatus, F);
    HIP_TRY(hipGetLastError());
    g->status.assign(F, OPUSFE_ERR_INTERNAL); g->error.assign(F, opusfe_error{}); g->single.assign(F, nullptr);
    const bool packed_ok = group_packed_pass(ctx, g, lower);
    if (!packed_ok) { g->flags.assign(F, FF_INTERNAL); g->lowered = false; }
    // the files the packed pass cannot answer for: alone through the single-source pipeline, several at a time (opusfe_parse_batch)
    std::vector<uint32_t> todo;
    for (uint32_t f = 0; f < F; f++) if (g->flags[f] & FF_SINGLE) todo.push_back(f);
    g->n_single = todo.size();
    if (!todo.empty()) {
        std::vector<std::vector<uint8_t>> copies;
        std::vector<const uint8_t*> srcs(todo.size()); std::vector<uint64_t> lens(todo.size());
        if (!host_srcs) copies.resize(todo.size());
        for (size_t i = 0; i < todo.size(); i++) {
            const uint32_t f = todo[i];
            lens[i] = g->len[f];
            if (host_srcs) srcs[i] = host_srcs[f];
            else {
                copies[i].resize(g->len[f] + 1);
                if (g->len[f]) HIP_TRY(hipMemcpyAsync(copies[i].data(), d_packed + g->start[f], g->len[f], hipMemcpyDeviceToHost, st));
                srcs[i] = copies[i].data();
            }
        }
        HIP_TRY(hipStreamSynchronize(st));
        std::vector<opusfe_unit*> units(todo.size(), nullptr); std::vector<int> rcs(todo.size(), OPUSFE_ERR_INTERNAL);
        const int rc = opusfe_parse_batch(ctx, (uint32_t)todo.size(), srcs.data(), lens.data(), lower, units.data(), rcs.data());
        for (size_t i = 0; i < todo.size(); i++) { g->single[todo[i]] = units[i]; g->status[todo[i]] = rcs[i]; }
        if (rc != OPUSFE_OK) return rc;
    }
    return OPUSFE_OK;
}

bool group_layout_ok(const std::vector<uint32_t>& start, const std::vector<uint32_t>& len, uint64_t packed_len) {
    if (start.empty() || start[0] != 0) return false;
    for (size_t f = 0; f < start.size(); f++) {
        if (start[f] % LEX_CHUNK) return false;
        const uint64_t end = (uint64_t)start[f] + len[f];
        const uint64_t next = f + 1 < start.size() ? start[f + 1] : packed_len;
        if (f + 1 < start.size() ? end >= next : end > next) return false;      // at least one byte of padding between two files
    }
    return true;
}

}  // namespace

extern "C" {

int opusfe_parse_group_device(opusfe_ctx* ctx, const uint8_t* d_packed, uint64_t packed_len, uint32_t n_files, const uint64_t* starts, const uint64_t* lens, int lower, opusfe_group** out) {
    if (!ctx || !out || !n_files || !starts || !lens || !d_packed) return OPUSFE_ERR_ARG;
    *out = nullptr;
    if (packed_len > MAX_LEN) return fail(ctx, OPUSFE_ERR_ARG, "group larger than 4 GiB: split it (opusfe positions are 32-bit per unit)");
    opusfe_group* g = new opusfe_group();
    g->ctx = ctx; g->info.n_files = n_files;
    g->start.resize(n_files); g->len.resize(n_files);
    for (uint32_t f = 0; f < n_files; f++) { g->start[f] = (uint32_t)starts[f]; g->len[f] = (uint32_t)lens[f]; }
    if (!group_layout_ok(g->start, g->len, packed_len)) { delete g; return fail(ctx, OPUSFE_ERR_ARG, "group layout: starts must be multiples of 64, ascending, with at least one byte ('\\n') between files, starts[0] == 0"); }
    *out = g;
    int rc = guarded(ctx, [&] { return group_run(ctx, g, d_packed, packed_len, false, nullptr, lower); });
    if (rc != OPUSFE_OK && rc != OPUSFE_SYNTAX) { (void)hipStreamSynchronize(ctx->stream); group_release(g); *out = nullptr; }
    return rc;
}

int opusfe_parse_group(opusfe_ctx* ctx, uint32_t n_files, const uint8_t* const* srcs, const uint64_t* lens, int lower, opusfe_group** out) {
    if (!ctx || !out || !n_files || !srcs || !lens) return OPUSFE_ERR_ARG;
    *out = nullptr;
    std::vector<uint64_t> starts(n_files);
    uint64_t total = 0;
    for (uint32_t f = 0; f < n_files; f++) { starts[f] = total; total = (total + lens[f] + 1 + (LEX_CHUNK - 1)) / LEX_CHUNK * LEX_CHUNK; }
    if (total > MAX_LEN) return fail(ctx, OPUSFE_ERR_ARG, "group larger than 4 GiB: split it (opusfe positions are 32-bit per unit)");
    opusfe_group* g = new opusfe_group();
    g->ctx = ctx; g->info.n_files = n_files;
    g->start.resize(n_files); g->len.resize(n_files);
    for (uint32_t f = 0; f < n_files; f++) { g->start[f] = (uint32_t)starts[f]; g->len[f] = (uint32_t)lens[f]; }
    *out = g;
    int rc = guarded(ctx, [&] {
        // pack on the host into the context's pinned staging buffer (a few threads: the copy is memory-bound), one copy to the device
        if (ctx->h_stage_bytes < total + 64) {
            if (ctx->h_stage) { (void)hipHostFree(ctx->h_stage); ctx->h_stage = nullptr; ctx->h_stage_bytes = 0; }
            const size_t want = (size_t)((total + 64) * 5 / 4 + (1 << 20));
            HIP_TRY(hipHostMalloc((void**)&ctx->h_stage, want, hipHostMallocDefault));
            ctx->h_stage_bytes = want;
        }
        uint8_t* h = ctx->h_stage;
        auto pack = [&](uint32_t f0, uint32_t f1) {
            for (uint32_t f = f0; f < f1; f++) {
                if (lens[f]) memcpy(h + starts[f], srcs[f], lens[f]);
                const uint64_t end = starts[f] + lens[f], next = f + 1 < n_files ? starts[f + 1] : total + 64;
                memset(h + end, '\n', next - end);
            }
        };
        const uint32_t n_thr = total > (8u << 20) ? 8u : 1u;
        if (n_thr == 1) pack(0, n_files);
        else {
            std::vector<std::thread> thr;
            uint32_t f0 = 0;
            for (uint32_t t = 0; t < n_thr; t++) {      // equal byte shares
                uint32_t f1 = f0;
                const uint64_t upto = total / n_thr * (t + 1);
                while (f1 < n_files && (t + 1 == n_thr || starts[f1] < upto)) f1++;
                thr.emplace_back(pack, f0, f1);
                f0 = f1;
            }
            for (auto& t : thr) t.join();
        }
        uint8_t* d = dev_alloc_t<uint8_t>(ctx, total + 64);
        try {
            HIP_TRY(hipMemcpyAsync(d, h, total + 64, hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));      // the staging buffer is reused by the next call
        } catch (...) { dev_free(ctx, d); throw; }
        return group_run(ctx, g, d, total, true, srcs, lower);
    });
    if (rc != OPUSFE_OK && rc != OPUSFE_SYNTAX) { (void)hipStreamSynchronize(ctx->stream); group_release(g); *out = nullptr; }
    return rc;
}

void opusfe_free_group(opusfe_group* g) {
    if (!g) return;
    (void)hipSetDevice(g->ctx->device);
    (void)hipStreamSynchronize(g->ctx->stream);
    group_release(g);
}

int opusfe_group_file_status(const opusfe_group* g, uint32_t file) { return g && file < g->info.n_files ? g->status[file] : OPUSFE_ERR_ARG; }

int opusfe_group_file_error(const opusfe_group* g, uint32_t file, opusfe_error* out) {
    if (!g || !out || file >= g->info.n_files) return OPUSFE_ERR_ARG;
    if (g->single[file]) return opusfe_get_error(g->single[file], out);
    if (g->status[file] != OPUSFE_SYNTAX) return OPUSFE_ERR_ARG;
    *out = g->error[file];
    return OPUSFE_OK;
}

/* the packed unit (every flat array of opusfe_array, positions relative to the packed buffer) and where a file's results lie in it:
   ranges[0..1] tokens, [2..3] definitions, [4..5] functions (signatures / function IR), [6..7] globals, [8] byte offset of the file,
   [9] the packed pass's flags for the file (defs.h FF_*: why it took the single-file path; FF_SYNTAX).
   A file that took the single-file path has an empty range here and a unit of its own (opusfe_group_file_unit). */
opusfe_unit* opusfe_group_unit(opusfe_group* g) { return g ? g->unit : nullptr; }
opusfe_unit* opusfe_group_file_unit(opusfe_group* g, uint32_t file) { return g && file < g->info.n_files ? g->single[file] : nullptr; }
int opusfe_group_file_ranges(const opusfe_group* g, uint32_t file, uint64_t* ranges) {
    if (!g || !ranges || file >= g->info.n_files) return OPUSFE_ERR_ARG;
    for (int k = 0; k < 10; k++) ranges[k] = 0;
    ranges[8] = g->start[file]; ranges[9] = g->flags[file];
    if (g->single[file] || g->tok_begin.size() <= file + 1) return OPUSFE_OK;
    ranges[0] = g->tok_begin[file]; ranges[1] = g->tok_begin[file + 1];
    if (g->lowered && g->status[file] == OPUSFE_OK) {
        ranges[2] = g->first_def[file]; ranges[3] = g->first_def[file + 1]; ranges[4] = g->first_func[file]; ranges[5] = g->first_func[file + 1];
        ranges[6] = g->first_global[file]; ranges[7] = g->first_global[file + 1];
    }
    return OPUSFE_OK;
}

int opusfe_group_stats(const opusfe_group* g, uint64_t* n_packed, uint64_t* n_single, uint64_t* n_syntax) {
    if (!g || !n_packed || !n_single || !n_syntax) return OPUSFE_ERR_ARG;
    *n_single = g->n_single; *n_packed = g->info.n_files - g->n_single; *n_syntax = 0;
    for (int st : g->status) *n_syntax += st == OPUSFE_SYNTAX;
    return OPUSFE_OK;
}

/* canonical text of ONE file of the group, identical to opusfe_dump of the file compiled alone (what: 0 tokens, 1 AST, 2 IR, 3 first-error line, 4 semantic errors) */
int opusfe_group_dump(opusfe_group* g, uint32_t file, int what, char** out, uint64_t* out_len) {
    if (!g || !out || file >= g->info.n_files) return OPUSFE_ERR_ARG;
    *out = nullptr;
    if (g->single[file]) return opusfe_dump(g->single[file], what, out, out_len);
    if (g->tok_begin.size() <= (size_t)file + 1) return fail(g->ctx, OPUSFE_ERR_ARG, "the file has no result (its single-file run failed)");
    opusfe_unit* u = g->unit;
    return guarded(g->ctx, [&] {
        std::string s;
        ensure_host_lex(u);
        DumpView w;
        w.pos_off = g->start[file];
        w.t0 = g->tok_begin[file]; w.t1 = g->tok_begin[file + 1];
        const int fst = g->status[file];
        if (g->lowered && fst == OPUSFE_OK) { w.d0 = g->first_def[file]; w.d1 = g->first_def[file + 1]; w.f0 = g->first_func[file]; w.f1 = g->first_func[file + 1]; w.glob_off = g->first_global[file]; }
        else { w.d0 = w.d1 = 0; w.f0 = w.f1 = 0; }
        switch (what) {
            case 0: s = dump_tokens(u, fst == OPUSFE_SYNTAX ? (uint64_t)g->error[file].tok_index + 1 : ~0ull, w); break;
            case 1: if (fst != OPUSFE_OK || !g->lowered) { g->ctx->msg = "no AST: the file did not parse (or the group was not lowered: the definition ranges come with the tables)"; return (int)OPUSFE_ERR_ARG; }
                    ensure_host_ast(u); s = dump_ast(u, w); break;
            case 2: if (fst == OPUSFE_OK && g->lowered) { ensure_host_ast(u); ensure_host_ir(u); s = dump_ir(u, w); } break;
            case 4: if (fst == OPUSFE_OK && g->lowered) { ensure_host_ir(u); s = dump_semantic_errors(u, &w); } break;
            case 3: {
                if (fst != OPUSFE_SYNTAX) { s = "Ok\n"; break; }
                HostView v{u, w.pos_off};
                const opusfe_error& e = g->error[file];
                const opusfe_token* found = e.tok_index < u->h_tokens.size() ? &u->h_tokens[e.tok_index] : nullptr;
                fmt_error_body(s, e.kind, e.low, e.len, e.payload, v, found, (int)e.expected);
                s += " line "; s += std::to_string(e.line); s += '\n';
                break;
            }
            default: return (int)OPUSFE_ERR_ARG;
        }
        char* p = (char*)malloc(s.size() + 1);
        if (!p) return (int)OPUSFE_ERR_OOM;
        memcpy(p, s.data(), s.size()); p[s.size()] = 0;
        *out = p;
        if (out_len) *out_len = s.size();
        return (int)OPUSFE_OK;
    });
}

int opusfe_set_profiling(opusfe_ctx* ctx, int enabled) {
    if (!ctx) return OPUSFE_ERR_ARG;
    ctx->profiling = enabled != 0;
    ctx->kstats.clear();
    return OPUSFE_OK;
}
int opusfe_kernel_count(opusfe_ctx* ctx) { return ctx ? (int)ctx->kstats.size() : 0; }
int opusfe_kernel_info(opusfe_ctx* ctx, int index, const char** name, double* ms, uint64_t* launches) {
    if (!ctx || index < 0 || index >= (int)ctx->kstats.size()) return OPUSFE_ERR_ARG;
    if (name) *name = ctx->kstats[index].name;
    if (ms) *ms = ctx->kstats[index].ms;
    if (launches) *launches = ctx->kstats[index].launches;
    return OPUSFE_OK;
}

}  // extern "C"
