// Checks device_scan_chained (scan.h) against device_scan (three kernels) and a host loop, and times both.
// Build: hipcc --offload-arch=gfx950 -O3 -I opus_amd/csrc tools/scan_check.hip -o tools/_build/scan_check ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "scan.h"
using namespace opusfe;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

template <bool EXCL, class T, class Op>
static int one(uint64_t n, Op op, T identity, uint32_t seed, bool in_place, const char* name) {
    std::vector<T> h(n), want(n), got(n), got3(n);
    uint32_t x = seed * 2654435761u + 12345u;
    for (uint64_t i = 0; i < n; i++) { x = x * 1664525u + 1013904223u; uint32_t r = x >> 8; h[i] = (T)((r % 7 == 0) ? (0x80000000u | (r & 0xFFFF)) : (r & 0x3FF)); if (sizeof(T) == 4 && name[0] == 'i') h[i] = (T)((int)(r % 5) - 2); }
    T acc = identity;
    for (uint64_t i = 0; i < n; i++) { if (EXCL) { want[i] = acc; acc = op(acc, h[i]); } else { acc = op(acc, h[i]); want[i] = acc; } }
    T *d_in, *d_out, *d_out3, *d_tmp, *d_tot; uint64_t* d_scr; uint32_t* d_status;
    CK(hipMalloc(&d_in, n * 4 + 16)); CK(hipMalloc(&d_out, n * 4 + 16)); CK(hipMalloc(&d_out3, n * 4 + 16)); CK(hipMalloc(&d_tmp, (n / SCAN_TILE + 2) * 4));
    CK(hipMalloc(&d_scr, chained_scan_words(n) * 8)); CK(hipMalloc(&d_tot, 8)); CK(hipMalloc(&d_status, 4));
    CK(hipMemset(d_status, 0, 4)); CK(hipMemset(d_tot, 0, 8));
    CK(hipMemcpy(d_in, h.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    device_scan<EXCL>(d_in, d_out3, n, op, identity, d_tmp, d_tot + 1, 0);      // warm
    if (in_place) CK(hipMemcpy(d_out, d_in, n * 4, hipMemcpyDeviceToDevice));
    CK(hipEventRecord(e0, 0));
    device_scan_chained<EXCL>(in_place ? d_out : d_in, d_out, n, op, identity, d_scr, d_tot, d_status, 0);
    CK(hipEventRecord(e1, 0));
    device_scan<EXCL>(d_in, d_out3, n, op, identity, d_tmp, d_tot + 1, 0);
    CK(hipEventRecord(e2, 0));
    CK(hipDeviceSynchronize());
    float ms_c, ms_3; CK(hipEventElapsedTime(&ms_c, e0, e1)); CK(hipEventElapsedTime(&ms_3, e1, e2));
    CK(hipMemcpy(got.data(), d_out, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(got3.data(), d_out3, n * 4, hipMemcpyDeviceToHost));
    T tot[2]; uint32_t st; CK(hipMemcpy(tot, d_tot, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&st, d_status, 4, hipMemcpyDeviceToHost));
    uint64_t bad = 0, bad3 = 0, first = n;
    for (uint64_t i = 0; i < n; i++) { if (got[i] != want[i]) { if (!bad) first = i; bad++; } if (got3[i] != want[i]) bad3++; }
    const bool tot_ok = tot[0] == acc && tot[1] == acc;
    printf("%-6s excl=%d n=%llu in_place=%d  chained %.3f ms  three-kernel %.3f ms  mismatches %llu (first %llu) / %llu  total %s  status %u\n", name, (int)EXCL, (unsigned long long)n, (int)in_place, ms_c, ms_3,
           (unsigned long long)bad, (unsigned long long)first, (unsigned long long)bad3, tot_ok ? "ok" : "WRONG", st);
    CK(hipFree(d_in)); CK(hipFree(d_out)); CK(hipFree(d_out3)); CK(hipFree(d_tmp)); CK(hipFree(d_scr)); CK(hipFree(d_tot)); CK(hipFree(d_status));
    return (bad || bad3 || !tot_ok || st) ? 1 : 0;
}
int main() {
    int fails = 0;
    const uint64_t sizes[] = {1, 3, 4, 5, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 8192, 262144 + 1, 1000003, 16 * 1048576 + 7, 216000000};
    for (uint64_t n : sizes) {
        fails += one<true, uint32_t>(n, OpAddU32(), 0u, (uint32_t)n, false, "addu");
        fails += one<false, int32_t>(n, OpAddI32(), 0, (uint32_t)n + 1, true, "i32");
        fails += one<true, uint32_t>(n, OpLast(), 0u, (uint32_t)n + 2, true, "last");
        fails += one<false, uint32_t>(n, OpLast(), 0u, (uint32_t)n + 3, false, "last");
    }
    printf(fails ? "FAILED %d\n" : "all ok\n", fails);
    return fails ? 1 : 0;
}
