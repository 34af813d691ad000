// Host-side runtime of libopusfe: context, caching device allocator, HIP-event kernel timing.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "defs.h"

#ifdef OPUSFE_EMU
// tests/emu: the device code is compiled for the host by g++ (sanitizers, gdb); no HIP runtime exists there
struct opusfe_ctx { int device = 0; };
#define HIP_TRY(expr) do { } while (0)
#else

struct opusfe_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // second stream: the sequential lowering machine works through the few huge functions while the node-parallel kernels run
    hipStream_t stream2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::string msg;
    std::vector<uint32_t> sched_off;      // lower.h: the offsets of the count pass's schedule (host copy, kept for its capacity)
    // caching allocator: freed blocks are kept and reused so that steady-state calls never hipMalloc
    std::multimap<size_t, void*> free_blocks;
    std::map<void*, size_t> live_blocks;
    // a cached block remembers the API call (epoch) that released it: when a request misses the cache, large blocks left over from
    // EARLIER calls are returned to the driver first (a block that no longer matches any request would otherwise be held for ever,
    // e.g. the first unit's IR arrays, sized from default ratios); a steady state has no misses and frees nothing
    std::map<void*, uint64_t> cached_epoch; uint64_t epoch = 0;
    size_t bytes_allocated = 0, bytes_peak = 0, bytes_overflow = 0;
    // profiling
    bool profiling = false;
    struct Pending { const char* name; hipEvent_t a, b; };
    std::vector<Pending> pending;
    struct KStat { const char* name; double ms; uint64_t launches; };
    std::vector<KStat> kstats;
    std::vector<hipEvent_t> event_pool;
    uint32_t* d_status = nullptr;   // DevStatus word + scratch totals (64 words)
    uint32_t* h_status = nullptr;   // pinned mirror
    // IR result sizes per AST node (variables, instructions, blocks, local types, call arguments) that lower.h run_lower sizes
    // its arrays with: generous defaults, then what the previous unit on this context needed (+4 %)
    double ir_ratio[5] = {1.35, 1.65, 0.40, 0.30, 0.25};
    // parse_par.h: the automaton's table on this device (uploaded on first use); parser_mode 1 = SegParser alone (opusfe_set_parser_mode)
    uint32_t* d_pp_table = nullptr; uint32_t pp_states = 0, pp_start = 0; int parser_mode = 0;
    uint32_t* d_static_scratch = nullptr;   // this device's address of api.hip `scratch` (hipGetSymbolAddress at opusfe_create)
    // opusfe_parse_batch: worker contexts (own stream and allocator each) owned by this context; units they created
    // are released before the context is destroyed
    std::vector<opusfe_ctx*> workers;
    // opusfe_parse_group: pinned staging buffer the files are packed into (grow-only, kept: pinning memory costs more than the copy)
    uint8_t* h_stage = nullptr; size_t h_stage_bytes = 0;
    // Temporaries (TempList): ONE grow-only arena per context, used as a stack by the stages of a call (lexer, parser, lowering run
    // one after the other, each releasing what it took), so the temporaries of a context cost the LARGEST stage, not the sum of
    // every block size any stage ever asked for.  `want` is the high-water mark the arena is resized to when nothing is in use.
    uint8_t* arena = nullptr; size_t arena_size = 0, arena_off = 0, arena_voff = 0, arena_want = 0;
};

namespace opusfe {

struct HipError { hipError_t e; const char* what; };

#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) throw opusfe::HipError{_e, #expr}; } while (0)

inline void* dev_alloc(opusfe_ctx* ctx, size_t bytes) {
    if (bytes == 0) bytes = 256;
    bytes = (bytes + 255) & ~(size_t)255;
    auto it = ctx->free_blocks.lower_bound(bytes);
    if (it != ctx->free_blocks.end() && it->first <= bytes + bytes / 4 + (1 << 20)) {
        void* p = it->second; size_t cap = it->first;
        ctx->free_blocks.erase(it);
        ctx->cached_epoch.erase(p);
        ctx->live_blocks[p] = cap;
        return p;
    }
    for (auto f = ctx->free_blocks.begin(); f != ctx->free_blocks.end();) {      // miss: stale large blocks go back to the driver
        auto e = ctx->cached_epoch.find(f->second);
        if (f->first >= ((size_t)16 << 20) && e != ctx->cached_epoch.end() && e->second < ctx->epoch) {
            (void)hipFree(f->second); ctx->bytes_allocated -= f->first; ctx->cached_epoch.erase(e); f = ctx->free_blocks.erase(f);
        } else ++f;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        // drop the cache and retry once
        for (auto& kv : ctx->free_blocks) { (void)hipFree(kv.second); ctx->bytes_allocated -= kv.first; }
        ctx->free_blocks.clear(); ctx->cached_epoch.clear();
        HIP_TRY(hipMalloc(&p, bytes));
    }
    ctx->bytes_allocated += bytes;
    if (ctx->bytes_allocated + ctx->bytes_overflow > ctx->bytes_peak) ctx->bytes_peak = ctx->bytes_allocated + ctx->bytes_overflow;
    ctx->live_blocks[p] = bytes;
    return p;
}
inline void dev_free(opusfe_ctx* ctx, void* p) {
    if (!p) return;
    auto it = ctx->live_blocks.find(p);
    if (it == ctx->live_blocks.end()) return;
    ctx->free_blocks.insert({it->second, p});
    ctx->cached_epoch[p] = ctx->epoch;
    ctx->live_blocks.erase(it);
}
template <class T> inline T* dev_alloc_t(opusfe_ctx* ctx, size_t count) { return (T*)dev_alloc(ctx, count * sizeof(T)); }

// RAII list of temporaries: a frame of the context's arena (released at scope exit; scopes nest like the stages that own them).
// What does not fit the arena is a plain allocation of its own, returned to the DRIVER (not to the block cache) at scope exit;
// the arena is then enlarged to the observed high-water mark the next time it is empty, so a steady state never allocates.
struct TempList {
    opusfe_ctx* ctx; size_t mark, vmark; std::vector<void*> overflow; size_t overflow_bytes = 0;
    explicit TempList(opusfe_ctx* c) : ctx(c), mark(c->arena_off), vmark(c->arena_voff) {
        if (ctx->arena_off == 0 && ctx->arena_want > ctx->arena_size) {
            // (hipFree waits for the device: nothing queued can still be using the old arena)
            if (ctx->arena) { (void)hipFree(ctx->arena); ctx->bytes_allocated -= ctx->arena_size; ctx->arena = nullptr; ctx->arena_size = 0; }
            const size_t want = ctx->arena_want + ctx->arena_want / 32 + (1 << 20);
            void* p = nullptr;
            if (hipMalloc(&p, want) != hipSuccess) {
                (void)hipGetLastError();
                for (auto& kv : ctx->free_blocks) { (void)hipFree(kv.second); ctx->bytes_allocated -= kv.first; }
                ctx->free_blocks.clear(); ctx->cached_epoch.clear();
                if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }      // stay on the overflow path
            }
            if (p) { ctx->arena = (uint8_t*)p; ctx->arena_size = want; ctx->bytes_allocated += want; }
            if (ctx->bytes_allocated + ctx->bytes_overflow > ctx->bytes_peak) ctx->bytes_peak = ctx->bytes_allocated + ctx->bytes_overflow;
        }
    }
    ~TempList() {
        if (!overflow.empty()) {
            (void)hipStreamSynchronize(ctx->stream);
            if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
            for (void* p : overflow) (void)hipFree(p);
            ctx->bytes_overflow -= overflow_bytes;
        }
        ctx->arena_off = mark; ctx->arena_voff = vmark;
    }
    TempList(const TempList&) = delete;
    TempList& operator=(const TempList&) = delete;
    template <class T> T* get(size_t count) {
        size_t bytes = count * sizeof(T);
        if (bytes == 0) bytes = 256;
        bytes = (bytes + 255) & ~(size_t)255;
        ctx->arena_voff += bytes;
        if (ctx->arena_voff > ctx->arena_want) ctx->arena_want = ctx->arena_voff;
        if (ctx->arena_off + bytes <= ctx->arena_size) { T* p = (T*)(ctx->arena + ctx->arena_off); ctx->arena_off += bytes; return p; }
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {      // drop the block cache and retry once
            (void)hipGetLastError();
            for (auto& kv : ctx->free_blocks) { (void)hipFree(kv.second); ctx->bytes_allocated -= kv.first; }
            ctx->free_blocks.clear(); ctx->cached_epoch.clear();
            HIP_TRY(hipMalloc(&p, bytes));
        }
        overflow.push_back(p); overflow_bytes += bytes; ctx->bytes_overflow += bytes;
        if (ctx->bytes_allocated + ctx->bytes_overflow > ctx->bytes_peak) ctx->bytes_peak = ctx->bytes_allocated + ctx->bytes_overflow;
        return (T*)p;
    }
};

inline hipEvent_t get_event(opusfe_ctx* ctx) {
    if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
    hipEvent_t e; HIP_TRY(hipEventCreate(&e)); return e;
}
inline void prof_begin(opusfe_ctx* ctx, const char* name) {
    if (!ctx->profiling) return;
    opusfe_ctx::Pending p{name, get_event(ctx), get_event(ctx)};
    HIP_TRY(hipEventRecord(p.a, ctx->stream));
    ctx->pending.push_back(p);
}
inline void prof_end(opusfe_ctx* ctx) {
    if (!ctx->profiling) return;
    HIP_TRY(hipEventRecord(ctx->pending.back().b, ctx->stream));
}
// call after the stream has been synchronised
inline void prof_collect(opusfe_ctx* ctx) {
    for (auto& p : ctx->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            bool found = false;
            for (auto& k : ctx->kstats) if (!strcmp(k.name, p.name)) { k.ms += ms; k.launches++; found = true; break; }
            if (!found) ctx->kstats.push_back({p.name, (double)ms, 1});
        }
        ctx->event_pool.push_back(p.a); ctx->event_pool.push_back(p.b);
    }
    ctx->pending.clear();
}

#define LAUNCH(ctx, name, kernel, grid, block, ...) do { \
        opusfe::prof_begin(ctx, name); \
        hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3((unsigned)(block)), 0, (ctx)->stream, __VA_ARGS__); \
        HIP_TRY(hipGetLastError());   /* a launch that fails (LDS / register limits of a variant build) must not pass as OK */ \
        opusfe::prof_end(ctx); \
    } while (0)

inline unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

}  // namespace opusfe
#endif  // OPUSFE_EMU
