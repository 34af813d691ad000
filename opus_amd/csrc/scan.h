// Device-wide order-preserving scans for gfx950 (wave64).  Three-kernel formulation
// (tile reduce -> scan of tile aggregates -> tile scan with carry-in); the operator only has to be
// associative (function composition and "last defined value" are not commutative).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace opusfe {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// shuffle of an arbitrary trivially-copyable T, one dword at a time (ds_bpermute underneath)
template <class T>
__device__ __forceinline__ T shfl_up_t(T v, int d) {
    static_assert(sizeof(T) % 4 == 0, "T must be a multiple of 4 bytes");
    union { T t; uint32_t w[sizeof(T) / 4]; } a, b;
    a.t = v;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) b.w[i] = (uint32_t)__shfl_up((int)a.w[i], d, 64);
    return b.t;
}

// wave64 inclusive scan; order preserving.
template <class T, class Op>
__device__ __forceinline__ T wave_inclusive_scan(T v, Op op) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = shfl_up_t(v, d);
        if (lane >= d) v = op(o, v);
    }
    return v;
}

// Block inclusive scan (blockDim.x == SCAN_THREADS, 4 waves).  `lds` must hold 4 T's.
template <class T, class Op>
__device__ __forceinline__ T block_inclusive_scan(T v, Op op, T* lds, T* block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T incl = wave_inclusive_scan(v, op);
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    T carry{};
    bool have = false;
    for (int w = 0; w < wave; w++) { carry = have ? op(carry, lds[w]) : lds[w]; have = true; }
    if (have) incl = op(carry, incl);
    if (block_total) {
        T tot = lds[0];
        for (int w = 1; w < SCAN_THREADS / 64; w++) tot = op(tot, lds[w]);
        *block_total = tot;
    }
    __syncthreads();
    return incl;
}

template <class T, class Op>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const T* __restrict__ in, T* __restrict__ tile_agg, uint64_t n, Op op, T identity) {
    __shared__ T lds[4];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    T acc = identity;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) if (base + i < n) acc = op(acc, in[base + i]);
    T total;
    block_inclusive_scan(acc, op, lds, &total);
    if (threadIdx.x == 0) tile_agg[blockIdx.x] = total;
}

// single-block exclusive scan of up to any number of aggregates (loops).
template <class T, class Op>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_aggregates(T* __restrict__ agg, uint64_t n, Op op, T identity, T* __restrict__ total_out) {
    __shared__ T lds[4];
    __shared__ T wave_last[4];
    __shared__ T carry_s;
    if (threadIdx.x == 0) carry_s = identity;
    __syncthreads();
    for (uint64_t base = 0; base < n; base += SCAN_TILE) {
        uint64_t b = base + (uint64_t)threadIdx.x * SCAN_ITEMS;
        T v[SCAN_ITEMS];
        T acc = identity;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) { v[i] = (b + i < n) ? agg[b + i] : identity; acc = op(acc, v[i]); }
        T total;
        T incl = block_inclusive_scan(acc, op, lds, &total);
        // exclusive prefix of this thread = carry ∘ (inclusive of previous thread)
        T prev = shfl_up_t(incl, 1);
        if ((threadIdx.x & 63) == 63) wave_last[threadIdx.x >> 6] = incl;
        __syncthreads();
        T excl;
        if (threadIdx.x == 0) excl = carry_s;
        else if ((threadIdx.x & 63) == 0) excl = op(carry_s, wave_last[(threadIdx.x >> 6) - 1]);
        else excl = op(carry_s, prev);
        T run = excl;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) { if (b + i < n) agg[b + i] = run; run = op(run, v[i]); }
        __syncthreads();
        if (threadIdx.x == 0) carry_s = op(carry_s, total);
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry_s;
}

// exclusive (EXCL) or inclusive scan of a tile with carry-in tile_agg[blockIdx.x] (already exclusive).
template <bool EXCL, class T, class Op>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const T* __restrict__ in, T* __restrict__ out, const T* __restrict__ tile_excl, uint64_t n, Op op, T identity) {
    __shared__ T lds[4];
    __shared__ T wave_last[4];
    uint64_t b = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS];
    T acc = identity;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) { v[i] = (b + i < n) ? in[b + i] : identity; acc = op(acc, v[i]); }
    T incl = block_inclusive_scan(acc, op, lds, (T*)nullptr);
    T prev = shfl_up_t(incl, 1);
    if ((threadIdx.x & 63) == 63) wave_last[threadIdx.x >> 6] = incl;
    __syncthreads();
    T carry = tile_excl[blockIdx.x];
    T excl;
    if (threadIdx.x == 0) excl = carry;
    else if ((threadIdx.x & 63) == 0) excl = op(carry, wave_last[(threadIdx.x >> 6) - 1]);
    else excl = op(carry, prev);
    T run = excl;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (b + i < n) {
            if (EXCL) { out[b + i] = run; run = op(run, v[i]); }
            else { run = op(run, v[i]); out[b + i] = run; }
        }
    }
}

#ifndef OPUSFE_EMU
struct ScanLaunchError {};
// Scans n elements of `in` into `out` (may alias).  `tmp` must hold ceil(n / SCAN_TILE) T's.
// `total` (device pointer, may be null) receives the reduction of all elements.
template <bool EXCL, class T, class Op>
void device_scan(const T* in, T* out, uint64_t n, Op op, T identity, T* tmp, T* total, hipStream_t stream) {
    if (n == 0) return;   // caller presets *total
    uint64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL((k_scan_reduce<T, Op>), dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, stream, in, tmp, n, op, identity);
    hipLaunchKernelGGL((k_scan_aggregates<T, Op>), dim3(1), dim3(SCAN_THREADS), 0, stream, tmp, tiles, op, identity, total);
    hipLaunchKernelGGL((k_scan_apply<EXCL, T, Op>), dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, stream, in, out, tmp, n, op, identity);
    if (hipGetLastError() != hipSuccess) throw ScanLaunchError{};
}

#endif  // OPUSFE_EMU

#ifndef OPUSFE_EMU
// ---- single-pass scan of 4-byte elements: decoupled look-back, DPP inside the wave ------------------------------------------
// One kernel, every element read once and written once (the three-kernel form reads everything twice).  A tile is 4096 elements: each
// of its four waves owns 1024 consecutive ones and loads them as four fully coalesced rows of 64 x 16 bytes; a lane folds its four
// elements, the wave scans each row across its lanes with DPP row shifts / row broadcasts (no LDS traffic), rows and waves are chained
// in registers and four LDS words.  Tiles take their number from a ticket counter, so every tile a workgroup waits for belongs to a
// workgroup that is already running; a tile publishes (value | flag << 32) in ONE 64-bit word — flag 1: its own aggregate, flag 2:
// the inclusive prefix up to its end — and wave 0 looks back over 64 predecessors at a time until it meets a prefix.  The operator
// only has to be associative, with `identity` a true identity.  The wait is bounded: a tile that never sees its predecessors reports
// DS_INTERNAL_SEGMENT in *status and publishes anyway, so the grid always drains.
constexpr int CH_ROWS = 4;
constexpr int CH_TILE = SCAN_THREADS * 4 * CH_ROWS;      // 4096 elements
constexpr uint32_t CH_SPIN_MAX = 1u << 20;
constexpr uint32_t CH_FAILED = 17;                       // = DS_INTERNAL_SEGMENT (defs.h)

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_keep(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false); }
template <class T> __device__ __forceinline__ uint32_t bits_of(T v) { union { T t; uint32_t w; } u; u.t = v; return u.w; }
template <class T> __device__ __forceinline__ T of_bits(uint32_t w) { union { T t; uint32_t w; } u; u.w = w; return u.t; }

// wave64 inclusive scan of a 4-byte value in lane order: row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then row_bcast:15 (lane 15 of
// rows 0 and 2 into rows 1 and 3) and row_bcast:31 (lane 31 into rows 2 and 3)
template <class T, class Op>
__device__ __forceinline__ T wave_inclusive_scan_dpp(T v, Op op) {
    static_assert(sizeof(T) == 4, "4-byte elements");
    const uint32_t lane = threadIdx.x & 63, l16 = lane & 15;
    uint32_t x = bits_of(v), o;
    o = dpp_keep<0x111, 0xF>(x); if (l16 >= 1) x = bits_of(op(of_bits<T>(o), of_bits<T>(x)));
    o = dpp_keep<0x112, 0xF>(x); if (l16 >= 2) x = bits_of(op(of_bits<T>(o), of_bits<T>(x)));
    o = dpp_keep<0x114, 0xF>(x); if (l16 >= 4) x = bits_of(op(of_bits<T>(o), of_bits<T>(x)));
    o = dpp_keep<0x118, 0xF>(x); if (l16 >= 8) x = bits_of(op(of_bits<T>(o), of_bits<T>(x)));
    o = dpp_keep<0x142, 0xA>(x); if (lane & 16) x = bits_of(op(of_bits<T>(o), of_bits<T>(x)));
    o = dpp_keep<0x143, 0xC>(x); if (lane & 32) x = bits_of(op(of_bits<T>(o), of_bits<T>(x)));
    return of_bits<T>(x);
}

template <bool EXCL, class T, class Op>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_chained(const T* in, T* out, uint64_t n, Op op, T identity,
                                                                unsigned long long* __restrict__ state, uint32_t* __restrict__ ticket, T* __restrict__ total_out,
                                                                uint32_t* __restrict__ status) {
    static_assert(sizeof(T) == 4, "4-byte elements");
    __shared__ uint32_t s_tile;
    __shared__ T s_wave[SCAN_THREADS / 64];
    __shared__ T s_carry;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_tile, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t wbase = (uint64_t)tile * CH_TILE + (uint64_t)wave * (64 * 4 * CH_ROWS);
    T v[CH_ROWS][4], inc[CH_ROWS], ex[CH_ROWS], rp[CH_ROWS];
#pragma unroll
    for (int r = 0; r < CH_ROWS; r++) {
        const uint64_t at = wbase + (uint64_t)r * 256 + lane * 4;
        if (at + 4 <= n) { const uint4 q = *(const uint4*)(in + at); v[r][0] = of_bits<T>(q.x); v[r][1] = of_bits<T>(q.y); v[r][2] = of_bits<T>(q.z); v[r][3] = of_bits<T>(q.w); }
        else { for (int e = 0; e < 4; e++) v[r][e] = at + e < n ? in[at + e] : identity; }
    }
    T run = identity;      // the rows before row r of this wave
#pragma unroll
    for (int r = 0; r < CH_ROWS; r++) {
        const T mine = op(op(op(v[r][0], v[r][1]), v[r][2]), v[r][3]);
        inc[r] = wave_inclusive_scan_dpp(mine, op);
        const T before = of_bits<T>((uint32_t)__shfl_up((int)bits_of(inc[r]), 1, 64));
        ex[r] = lane == 0 ? identity : before;
        rp[r] = run;
        run = op(run, of_bits<T>((uint32_t)__builtin_amdgcn_readlane((int)bits_of(inc[r]), 63)));
    }
    if (lane == 0) s_wave[wave] = run;
    __syncthreads();
    T wp = identity, total = identity;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; w++) { if (w == (int)wave) wp = total; total = op(total, s_wave[w]); }
    if (wave == 0) {
        T carry = identity;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(state, (unsigned long long)bits_of(total) | (2ull << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(state + tile, (unsigned long long)bits_of(total) | (1ull << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t hi = (int64_t)tile;      // the window is tiles [hi - 64, hi), lane l looks at hi - 64 + l; tiles below 0: an empty prefix
            uint32_t spins = 0;
            bool gave_up = false;
            for (;;) {
                const int64_t k = hi - 64 + (int64_t)lane;
                unsigned long long w;
                int J;
                for (;;) {
                    w = k >= 0 ? __hip_atomic_load(state + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)bits_of(identity) | (2ull << 32));
                    const unsigned long long has_prefix = __ballot((w >> 32) == 2), has_any = __ballot((w >> 32) != 0);
                    J = has_prefix ? 63 - __builtin_clzll(has_prefix) : -1;             // the closest prefix in the window
                    const unsigned long long need = J >= 0 ? ~0ull << J : ~0ull;         // everything from there on must have published
                    if ((has_any & need) == need) break;
                    if (++spins > CH_SPIN_MAX) { gave_up = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (gave_up) break;
                const T x = (int)lane >= J ? of_bits<T>((uint32_t)w) : identity;
                const T red = wave_inclusive_scan_dpp(x, op);
                carry = op(of_bits<T>((uint32_t)__builtin_amdgcn_readlane((int)bits_of(red), 63)), carry);
                if (J >= 0) break;
                hi -= 64;
            }
            if (lane == 0) {
                if (gave_up) atomicMax(status, CH_FAILED);
                __hip_atomic_store(state + tile, (unsigned long long)bits_of(op(carry, total)) | (2ull << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (lane == 0) {
            s_carry = carry;
            if (total_out && (uint64_t)(tile + 1) * CH_TILE >= n) *total_out = op(carry, total);
        }
    }
    __syncthreads();
    const T base = op(s_carry, wp);
#pragma unroll
    for (int r = 0; r < CH_ROWS; r++) {
        const uint64_t at = wbase + (uint64_t)r * 256 + lane * 4;
        T acc = op(base, op(rp[r], ex[r]));
        T o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) { if (EXCL) { o[e] = acc; acc = op(acc, v[r][e]); } else { acc = op(acc, v[r][e]); o[e] = acc; } }
        if (at + 4 <= n) *(uint4*)(out + at) = make_uint4(bits_of(o[0]), bits_of(o[1]), bits_of(o[2]), bits_of(o[3]));
        else { for (int e = 0; e < 4; e++) if (at + e < n) out[at + e] = o[e]; }
    }
}

// words of scratch (8 bytes each) that device_scan_chained needs for n elements
inline uint64_t chained_scan_words(uint64_t n) { return (n + CH_TILE - 1) / CH_TILE + 2; }
// Scans n 4-byte elements of `in` into `out` (may be the same array; both 16-byte aligned).  `scratch`: chained_scan_words(n) words.
template <bool EXCL, class T, class Op>
void device_scan_chained(const T* in, T* out, uint64_t n, Op op, T identity, uint64_t* scratch, T* total, uint32_t* status, hipStream_t stream) {
    if (n == 0) return;   // caller presets *total
    if ((((uintptr_t)in | (uintptr_t)out) & 15) != 0) throw ScanLaunchError{};
    const uint64_t tiles = (n + CH_TILE - 1) / CH_TILE;
    if (hipMemsetAsync(scratch, 0, (tiles + 1) * 8, stream) != hipSuccess) throw ScanLaunchError{};
    hipLaunchKernelGGL((k_scan_chained<EXCL, T, Op>), dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, stream, in, out, n, op, identity,
                       (unsigned long long*)scratch, (uint32_t*)(scratch + tiles), total, status);
    if (hipGetLastError() != hipSuccess) throw ScanLaunchError{};
}
#endif  // OPUSFE_EMU

struct OpAddU32 { __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return a + b; } };
struct OpAddI32 { __host__ __device__ int32_t operator()(int32_t a, int32_t b) const { return a + b; } };
struct OpAddU64 { __host__ __device__ uint64_t operator()(uint64_t a, uint64_t b) const { return a + b; } };

// (ntok, nval, nstr) triple sum
struct Cnt3 { uint32_t tok, val, str; uint32_t pad; };
struct OpAddCnt3 { __host__ __device__ Cnt3 operator()(Cnt3 a, Cnt3 b) const { return Cnt3{a.tok + b.tok, a.val + b.val, a.str + b.str, 0}; } };

// "last defined value": element = value | (1 << 31) when defined
struct OpLast { __host__ __device__ uint32_t operator()(uint32_t a, uint32_t b) const { return (b >> 31) ? b : a; } };

#ifndef OPUSFE_EMU
// ---- largest units first ---------------------------------------------------------------------------
// One wave works through one unit (parser: segment, lowering: function) from start to end, so a long unit that is
// dispatched late finishes long after everything else.  Units are therefore dispatched by size class (bit length of
// their token / node count, largest class first): a counting sort over 66 half-octave classes.  The order only decides WHEN a function runs, never what it produces.
// (class counters are kept in LDS per workgroup: one global atomic per class and workgroup instead of one per unit)
// class = 2 * bit length + the bit below the leading one: half-octave size classes ([512,768), [768,1024), ...), so that a
// consumer that sizes its workgroups by class (lower.h) wastes at most a third of what it reserves
enum { ORDER_CLASSES = 66 };
__host__ __device__ __forceinline__ uint32_t order_class_of_size(uint32_t size) {
    if (size == 0) return 0;
    const uint32_t bl = 32u - (uint32_t)__builtin_clz(size);
    return 2u * bl + (bl >= 2 ? (size >> (bl - 2)) & 1u : 0u);
}
// sizes of class c are [order_class_lo(c), order_class_lo(c + 1))
__host__ __device__ __forceinline__ uint64_t order_class_lo(uint32_t c) {
    const uint32_t bl = c / 2, half = c & 1u;
    if (bl == 0) return 0;
    if (bl == 1) return half ? 2 : 1;      // (class 3 is empty: size 1 has no bit below the leading one)
    return (uint64_t)(2 + half) << (bl - 2);
}
__device__ __forceinline__ uint32_t order_class(const uint32_t* __restrict__ start, uint32_t n, uint32_t end_all, uint32_t i) {
    return order_class_of_size((i + 1 < n ? start[i + 1] : end_all) - start[i]);
}
__global__ __launch_bounds__(256) void k_order_hist(const uint32_t* __restrict__ start, uint32_t n, uint32_t end_all, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[ORDER_CLASSES];
    if (threadIdx.x < ORDER_CLASSES) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&h[order_class(start, n, end_all, i)], 1u);
    __syncthreads();
    if (threadIdx.x < ORDER_CLASSES && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void k_order_scan(uint32_t* __restrict__ hist) {          // in place: count -> first output slot, largest class first
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint32_t acc = 0;
    for (int c = ORDER_CLASSES - 1; c >= 0; c--) { uint32_t v = hist[c]; hist[c] = acc; acc += v; }
}
__global__ __launch_bounds__(256) void k_order_scatter(const uint32_t* __restrict__ start, uint32_t n, uint32_t end_all, uint32_t* __restrict__ hist, uint32_t* __restrict__ order) {
    __shared__ uint32_t h[ORDER_CLASSES], base[ORDER_CLASSES];
    if (threadIdx.x < ORDER_CLASSES) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t c = 0, mine = 0;
    if (i < n) { c = order_class(start, n, end_all, i); mine = atomicAdd(&h[c], 1u); }
    __syncthreads();
    if (threadIdx.x < ORDER_CLASSES && h[threadIdx.x]) base[threadIdx.x] = atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
    __syncthreads();
    if (i < n) order[base[c] + mine] = i;
}

#endif

}  // namespace opusfe
