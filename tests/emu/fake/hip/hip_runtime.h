// TEST INFRASTRUCTURE (tests/emu): stand-in for <hip/hip_runtime.h> so that the device code of
// opus_amd/csrc compiles for the HOST with g++ (-fsanitize=address,undefined, gdb).  Kernels that
// use no cross-lane communication are then executed lane by lane by emu_launch(); kernels built on
// wave shuffles / LDS scans only have to compile here (the harness replaces them by sequential
// scans).  Nothing in the product includes this file.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>

#define __device__
#define __host__
#define __global__
#ifdef EMU_NOINLINE
#define __forceinline__ inline
#else
#define __forceinline__ inline __attribute__((always_inline))
#endif
#define __shared__ static
#define __noinline__
#define __launch_bounds__(...)

struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
struct EmuDim3 { unsigned x = 0, y = 0, z = 0; };
inline thread_local EmuDim3 threadIdx, blockIdx, blockDim, gridDim;

template <class T, class U> inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U, class V> inline T atomicCAS(T* p, U cmp, V val) { T o = *p; if (o == (T)cmp) *p = (T)val; return o; }
template <class T, class U> inline T atomicAdd(T* p, U v) { T o = *p; *p = o + (T)v; return o; }
template <class T, class U> inline T atomicOr(T* p, U v) { T o = *p; *p = o | (T)v; return o; }
template <class T, class U> inline T atomicSub(T* p, U v) { T o = *p; *p = o - (T)v; return o; }
template <class T, class U> inline T atomicAnd(T* p, U v) { T o = *p; *p = o & (T)v; return o; }
[[noreturn]] inline void emu_collective() { fprintf(stderr, "emu: cross-lane primitive executed\n"); abort(); }
inline int __shfl_up(int, int, int) { emu_collective(); }
inline int __shfl_down(int, int, int) { emu_collective(); }
inline unsigned __shfl_down(unsigned, int, int) { emu_collective(); }
inline int __shfl(int, int, int) { emu_collective(); }
inline void __syncthreads() {}
inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }

template <class K, class... A>
inline void emu_launch(K kernel, unsigned grid, unsigned block, A... args) {
    gridDim.x = grid; blockDim.x = block;
    for (unsigned b = 0; b < grid; b++)
        for (unsigned t = 0; t < block; t++) { blockIdx.x = b; threadIdx.x = t; kernel(args...); }
}
