// Matrix-core issue-rate probe for gfx950 (compile and run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak).
// Register-resident operands, NACC independent accumulator tiles per wave, W waves per SIMD: what fraction of the quoted
// dense peaks (157 TFLOP/s fp32, 2.5 PFLOP/s bf16) a pure MFMA stream sustains - the ceiling for gemm.hip / gemm_x3.hip.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int NACC, bool BF16>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 xa, xb;
    for (int e = 0; e < 8; ++e) { xa[e] = (__bf16)(seed + threadIdx.x * 1e-3f + e); xb[e] = (__bf16)(seed - e); }
    float fa = seed + threadIdx.x, fb = seed * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            if (BF16) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, xb, acc[a], 0, 0, 0);
            else acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[a], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// One k-step of the split-operand GEMM inner loop: READS ds_read_b128 fragments (conflict-free 80-byte row stride) feeding
// 24 bf16 MFMAs on 4 accumulator tiles; READS = 12 is gemm_x3.hip today (both operands from LDS), 6 = one operand from
// registers / straight from L2.
template <int READS>
__global__ __launch_bounds__(256, 2) void probe_lds(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    for (int i = threadIdx.x; i < 6 * 128 * 40; i += 256) lds[i] = (unsigned short)(0x3f80 + (i & 7));
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned short* base = lds + ((wave >> 1) * 64 + (lane & 31)) * 40 + (lane >> 5) * 8;
    bf16x8 f[12];
    for (int q = 0; q < 12; ++q) f[q] = *reinterpret_cast<const bf16x8*>(base + (q % 6) * 128 * 40 + (q / 6) * 32 * 40);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < READS; ++q)
            f[q] = *reinterpret_cast<const bf16x8*>(base + (q % 6) * 128 * 40 + (q / 6) * 32 * 40 + (it & 1) * 16);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[(q + a) % 6], f[6 + (q + 2 * a) % 6], acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int READS>
void run_lds(float* d) {
    const int iters = 20000, cus = 256, wg_per_cu = 2;
    const size_t lds = 6 * 128 * 40 * 2;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe_lds<READS><<<cus * wg_per_cu, 256, lds>>>(d, 100);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe_lds<READS><<<cus * wg_per_cu, 256, lds>>>(d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double flop = 32768.0 * 24 * iters * 4.0 * cus * wg_per_cu;
    printf("bf16 k-step, %2d ds_read_b128 per 24 MFMAs, 2 waves/SIMD : %8.1f TFLOP/s = %.1f fp32-equivalent (6 products)\n", READS,
           flop / ms / 1e9, flop / ms / 1e9 / 6);
}

template <int NACC, bool BF16>
void run(int wg_per_cu, float* d) {
    const int iters = 20000, cus = 256;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe<NACC, BF16><<<cus * wg_per_cu, 256>>>(d, 100, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<NACC, BF16><<<cus * wg_per_cu, 256>>>(d, iters, 1.f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double flop = (BF16 ? 32768.0 : 4096.0) * NACC * iters * 4.0 * cus * wg_per_cu;
    printf("%s  acc tiles/wave %d  waves/SIMD %d : %8.1f TFLOP/s (%.2f ms)\n", BF16 ? "bf16 32x32x16" : "fp32 32x32x2 ", NACC, wg_per_cu,
           flop / ms / 1e9, ms);
}

int main() {
    float* d;
    hipMalloc(&d, sizeof(float) * 256 * 256 * 8);
    for (int w = 1; w <= 4; w *= 2) { run<1, false>(w, d); run<4, false>(w, d); }
    for (int w = 1; w <= 4; w *= 2) { run<1, true>(w, d); run<2, true>(w, d); run<4, true>(w, d); }
    run_lds<0>(d); run_lds<6>(d); run_lds<12>(d);
    return 0;
}
