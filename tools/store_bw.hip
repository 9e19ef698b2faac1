// Micro-benchmark (round 6): how fast can every CU push 16-byte-per-lane stores into its XCD's L2 / into HBM, by address pattern?
// One 512-thread workgroup per CU, each writing `bytes_per_wg` per repetition into its own region:
//   pattern 0: a wave instruction = 1 KB contiguous (8 full 128-B lines)
//   pattern 1: the plane epilogue's: a wave instruction = 16 rows x 64 B at a row stride of `ld` bytes (half lines), the other half
//              of the same lines by the NEXT instruction (hi plane, then lo plane)
//   pattern 2: 8 rows x 128 B at the same row stride (full lines, what a permlane32 swap of hi / lo would give)
// flags bit0: non-temporal stores.  Build: hipcc --offload-arch=gfx950 -O3 tools/store_bw.hip -o tools/store_bw.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, bool NT>
__global__ __launch_bounds__(512) void store_kernel(char* base, size_t region, int ld, int reps, int rows_per_wg) {
    char* my = base + (size_t)blockIdx.x * region;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4 v = {threadIdx.x, blockIdx.x, 1u, 2u};
    for (int r = 0; r < reps; ++r) {
        // a wave owns rows [wave * rows_per_wg / 8, ...): walks its rows x (ld bytes) area
        const int rows_w = rows_per_wg / 8;
        char* wbase = my + (size_t)wave * rows_w * ld;
        if (PAT == 0) {
            const int n = rows_w * ld / 1024;
            for (int i = 0; i < n; ++i) {
                u32x4* p = (u32x4*)(wbase + (size_t)i * 1024 + lane * 16);
                if (NT) __builtin_nontemporal_store(v, p); else *p = v;
            }
        } else if (PAT == 1) {
            // blocks of 16 rows x 128 B (one 32-column plane block): hi store then lo store
            for (int rb = 0; rb < rows_w; rb += 16)
                for (int cb = 0; cb < ld; cb += 128) {
                    char* q = wbase + (size_t)(rb + (lane >> 2)) * ld + cb + (lane & 3) * 16;
                    if (NT) { __builtin_nontemporal_store(v, (u32x4*)q); __builtin_nontemporal_store(v, (u32x4*)(q + 64)); }
                    else { *(u32x4*)q = v; *(u32x4*)(q + 64) = v; }
                }
        } else {
            for (int rb = 0; rb < rows_w; rb += 16)
                for (int cb = 0; cb < ld; cb += 128) {
                    char* q = wbase + (size_t)(rb + (lane >> 3)) * ld + cb + (lane & 7) * 16;
                    if (NT) { __builtin_nontemporal_store(v, (u32x4*)q); __builtin_nontemporal_store(v, (u32x4*)(q + 8 * (size_t)ld)); }
                    else { *(u32x4*)q = v; *(u32x4*)(q + 8 * (size_t)ld) = v; }
                }
        }
        v[2] += 1;
    }
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const size_t region_max = 4u << 20;  // per workgroup
    char* d; hipMalloc(&d, region_max * cus);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { const char* name; int rows; int ld; int reps; };
    // 256 rows x 1 KB = 256 KB per workgroup = one tile's output (67 MB over the chip, L2-resident per XCD: 8 MB > 4 MB -> spills);
    // 64 rows x 1 KB = 64 KB per workgroup (2 MB per XCD: stays in L2)
    const Cfg cfgs[] = {{"256 KB / WG (tile output, 1 KB rows)", 256, 1024, 64}, {"256 KB / WG, 2 KB rows", 128, 2048, 64},
                        {"64 KB / WG (L2-resident), 1 KB rows", 64, 1024, 256}};
    for (const Cfg& c : cfgs)
        for (int nt = 0; nt < 2; ++nt)
            for (int pat = 0; pat < 3; ++pat) {
                const size_t region = (size_t)c.rows * c.ld;
                auto launch = [&](int reps) {
#define L(P, N) hipLaunchKernelGGL((store_kernel<P, N>), dim3(cus), dim3(512), 0, 0, d, region, c.ld, reps, c.rows)
                    if (nt) { if (pat == 0) L(0, true); else if (pat == 1) L(1, true); else L(2, true); }
                    else { if (pat == 0) L(0, false); else if (pat == 1) L(1, false); else L(2, false); }
                };
                launch(4); hipDeviceSynchronize();
                hipEventRecord(e0); launch(c.reps); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double bytes = (double)region * cus * c.reps;
                printf("%-40s nt=%d pattern=%d: %7.1f us per %5.1f MB sweep  = %6.2f TB/s  (%5.1f GB/s per CU)\n", c.name, nt, pat,
                       1e3 * ms / c.reps, region * cus / 1e6, bytes / ms / 1e9, bytes / ms / 1e6 / cus);
            }
    return 0;
}
