// Feasibility probe for the successor of gemm_p2c: the K loop of a WAVE-PRIVATE row-block GEMM.
//
// gemm_p2 / gemm_p2c: 8 waves (two per SIMD) share a 256 x 256 tile, a K step takes ~3750 - 4000 cycles for 3072 of matrix pipe
// whether or not the waves meet at a barrier - the two instruction streams of a SIMD do not interleave without loss.
// The wave-private form: 4 waves per workgroup, ONE per SIMD (512 registers), a wave owns 32 rows and ALL 512 output columns of
// MLP0 (16 accumulator blocks of 32 x 32 = 256 accumulation registers), its activation fragments come straight from global memory
// (nobody else needs them), the weight slices of a K step (512 rows x 128 B = 64 KB) go through a two-slot LDS ring filled by
// LDS-direct loads that the matrix waves issue THEMSELVES, one per 6 MFMA slots.  What nobody knows without measuring: does that
// stream keep the matrix pipe busy?  This file is that measurement and nothing else: the f16x2 arithmetic of gemm_p2 (x = hi +
// 2^-11 lo', 2^s W = w_hi + w_lo, three products per block), M x 512 x 512, no epilogue but a checksum (full fp32 output with
// --check, compared with a host evaluation of the same planes).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wp_gemm.hip -o tools/wp_gemm.bin && ./tools/wp_gemm.bin [--check]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int N_OUT = 512, K_IN = 512, ROWS_WG = 128, BK = 32;
constexpr int SLICEB = N_OUT * 128;  // one K step of the weights: 512 rows x 128 B = 64 KB
constexpr int LDSB = 2 * SLICEB;

__device__ __forceinline__ void mfma_a(f32x16& c, f16x8 a, f16x8 b) {  // accumulator in the accumulation registers
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_a0(f32x16& c, f16x8 a, f16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ unsigned pk_mul(unsigned x, unsigned k) {
    unsigned d;
    asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(d) : "v"(x), "v"(k));
    return d;
}
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* dst, unsigned voffset, unsigned soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voffset, soffset, 0, 0);
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// MODE 0: checksum only; 1: full fp32 output C [M][512]
// ABL (timing only, wrong results): 1 no LDS-direct loads in the loop, 2 no multiplies, 4 no fragment reads in the loop, 8 no activation loads,
// 16 no end-of-step wait / barrier
template <int MODE, int ABL = 0>
__global__ __launch_bounds__(256, 1) void wp_mlp0_kernel(const uint16_t* A, const uint16_t* W, float* C, float* sums, int M, int blocks_per_wg, int k_reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned lda_b = K_IN * 4, ldw_b = K_IN * 4;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(W), 0, N_OUT * K_IN * 4, 0x00020000);
    // loader: a wave moves 128 weight rows of a K step: 16 LDS-direct loads of 8 rows x 128 B; lane -> (row lane >> 3, position lane & 7),
    // source chunk = position ^ ((row >> 1) & 7) (the swizzle of p2.h, applied on the source side)
    unsigned w_vo[2];  // rows 0-7 / 8-15 of a 16-row pair differ in the swizzle's bit 2 only: two offsets, the rest in the scalar offset
    {
        const unsigned r = lane >> 3, pos = lane & 7;
        w_vo[0] = r * ldw_b + ((pos ^ ((r >> 1) & 7)) * 16u);
        w_vo[1] = r * ldw_b + ((pos ^ (((r + 8) >> 1) & 7)) * 16u);
    }
    auto load_piece = [&](int slot, int kt, int i) __attribute__((always_inline)) {  // piece i = 0..15 of this wave's 128 rows
        const unsigned row0 = 128u * (unsigned)wave + 8u * (unsigned)i;
        glds16(rsW, smem + slot * SLICEB + row0 * 128, w_vo[i & 1], row0 * ldw_b + (unsigned)kt * 128u);
    };
    // fragment addresses: W row block j (32 rows), lane row l31, chunk c = 4 plane + 2 ks + lh at position c ^ ((l31 >> 1) & 7)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned swz = (l31 >> 1) & 7;
    unsigned fa[2][2];  // [plane][ks]
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[pl][ks] = lds0 + (unsigned)(l31 * 128) + (((unsigned)(4 * pl + 2 * ks + lh) ^ swz) << 4);
    typedef __attribute__((address_space(3))) const f16x8* lds_frag_t;
    auto wfrag = [&](int slot, int j, int pl, int ks) __attribute__((always_inline)) {
        return *reinterpret_cast<lds_frag_t>((uintptr_t)(fa[pl][ks] + (unsigned)(slot * SLICEB + j * 32 * 128)));
    };
    unsigned k2048 = 0x10001000u;
    asm volatile("" : "+v"(k2048));

    for (int blk = 0; blk < blocks_per_wg; ++blk) {
        const int row_base = (blockIdx.x * blocks_per_wg + blk) * ROWS_WG + wave * 32;
        if (row_base >= M) break;
        const char* xrow = reinterpret_cast<const char*>(A) + (size_t)(row_base + l31) * lda_b + lh * 16;
        f32x16 acc[16];
        f16x8 xf[2][2][2];  // [buffer][plane][ks]: the activation fragments of a K step, straight from global memory
        auto load_x = [&](int b, int kt) __attribute__((always_inline)) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) xf[b][pl][ks] = *reinterpret_cast<const f16x8*>(xrow + kt * 128 + pl * 64 + ks * 32);
        };
        constexpr int NK = K_IN / BK;
        // prologue: slice 0 into slot 0, the fragments of step 0
#pragma unroll
        for (int i = 0; i < 16; ++i) load_piece(0, 0, i);
        load_x(0, 0);
        if (ABL & 8) load_x(1, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        auto step = [&](auto FIRST, auto SLOT, int kt) __attribute__((always_inline)) {
            constexpr bool first = decltype(FIRST)::value;
            constexpr int slot = decltype(SLOT)::value;  // (compile-time: the fragment buffers are registers)
            // one stream: 96 MFMAs (2 k-halves x 16 weight blocks x 3 products); behind them, slot by slot: the next block's two
            // fragment reads + the four multiplies that make 2^-11 w_hi, one LDS-direct load of the NEXT slice every 6th slot, the
            // next step's activation fragments in the first slots
            f16x8 wh[3], wl[3];  // the fragments of blocks g, g + 1, g + 2: read TWO blocks (6 MFMA slots) ahead of their use
            wl[0] = wfrag(slot, 0, 1, 0);
            wh[0] = wfrag(slot, 0, 0, 0);
            wl[1] = wfrag(slot, 1, 1, 0);
            wh[1] = wfrag(slot, 1, 0, 0);
            if (ABL & 4) { wl[2] = wl[0]; wh[2] = wh[1]; }
            const bool more = kt + 1 < NK;
            __builtin_amdgcn_sched_barrier(0);
            // blocks in PAIRS (j, j + 1), their products alternating: no MFMA reads the accumulator the MFMA right before it wrote
            static_for<0, 16>([&](auto GG) __attribute__((always_inline)) {
                constexpr int gg = decltype(GG)::value, g0 = 2 * gg, ks = g0 >> 4, j = g0 & 15, p0 = g0 % 3, p1 = (g0 + 1) % 3;
                const u32x4 whu0 = __builtin_bit_cast(u32x4, wh[p0]), whu1 = __builtin_bit_cast(u32x4, wh[p1]);
                u32x4 w2u0, w2u1;
                // slots 0, 1: x_hi w_lo of both blocks; the multiplies
                if constexpr (first && ks == 0) mfma_a0(acc[j], wl[p0], xf[slot][0][ks]); else mfma_a(acc[j], wl[p0], xf[slot][0][ks]);
#pragma unroll
                for (int e = 0; e < 4; ++e) w2u0[e] = (ABL & 2) ? whu0[e] : pk_mul(whu0[e], k2048);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (first && ks == 0) mfma_a0(acc[j + 1], wl[p1], xf[slot][0][ks]); else mfma_a(acc[j + 1], wl[p1], xf[slot][0][ks]);
#pragma unroll
                for (int e = 0; e < 4; ++e) w2u1[e] = (ABL & 2) ? whu1[e] : pk_mul(whu1[e], k2048);
                __builtin_amdgcn_sched_barrier(0);
                // slots 2, 3: x_lo' (2^-11 w_hi); a piece of the next slice behind each (16 pieces over the first 8 pairs), the next step's
                // activation fragments behind pair 0
                mfma_a(acc[j], __builtin_bit_cast(f16x8, w2u0), xf[slot][1][ks]);
                if constexpr (gg < 8 && !(ABL & 1)) { if (more) load_piece(slot ^ 1, kt + 1, 2 * gg); }
                if constexpr (gg == 0 && !(ABL & 8)) { if (more) load_x(slot ^ 1, kt + 1); }
                __builtin_amdgcn_sched_barrier(0);
                mfma_a(acc[j + 1], __builtin_bit_cast(f16x8, w2u1), xf[slot][1][ks]);
                if constexpr (gg < 8 && !(ABL & 1)) { if (more) load_piece(slot ^ 1, kt + 1, 2 * gg + 1); }
                __builtin_amdgcn_sched_barrier(0);
                // slots 4, 5: x_hi w_hi; then the fragments of the pair after next go into the buffers this pair's high planes leave... (two
                // blocks ahead: buffers (g0 + 2) % 3 = the one block g0 - 1 used, and - behind slot 5 - (g0 + 3) % 3 = p0)
                mfma_a(acc[j], wh[p0], xf[slot][0][ks]);
                if constexpr (g0 + 2 < 32 && !(ABL & 4)) {
                    wl[(g0 + 2) % 3] = wfrag(slot, (g0 + 2) & 15, 1, (g0 + 2) >> 4);
                    wh[(g0 + 2) % 3] = wfrag(slot, (g0 + 2) & 15, 0, (g0 + 2) >> 4);
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma_a(acc[j + 1], wh[p1], xf[slot][0][ks]);
                if constexpr (g0 + 3 < 32 && !(ABL & 4)) {
                    wl[p0] = wfrag(slot, (g0 + 3) & 15, 1, (g0 + 3) >> 4);
                    wh[p0] = wfrag(slot, (g0 + 3) & 15, 0, (g0 + 3) >> 4);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            // the next slice has landed (everybody's pieces) and everybody is through this one
            if constexpr (!(ABL & 16)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        };
        step(std::true_type{}, std::integral_constant<int, 0>{}, 0);
        step(std::false_type{}, std::integral_constant<int, 1>{}, 1);
        for (int kt = 2; kt < NK; kt += 2) {
            step(std::false_type{}, std::integral_constant<int, 0>{}, kt);
            step(std::false_type{}, std::integral_constant<int, 1>{}, kt + 1);
        }
        // (timing only: the same 16 steps again, k_reps - 1 times - the per-workgroup overhead drops out of the difference)
        for (int rep = 1; rep < k_reps; ++rep)
            for (int kt = 0; kt < NK; kt += 2) {
                step(std::false_type{}, std::integral_constant<int, 0>{}, kt);
                step(std::false_type{}, std::integral_constant<int, 1>{}, kt + 1 < NK - 1 ? kt + 1 : NK - 2);
            }
        asm volatile("s_nop 15" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), "+a"(acc[7]),
                     "+a"(acc[8]), "+a"(acc[9]), "+a"(acc[10]), "+a"(acc[11]), "+a"(acc[12]), "+a"(acc[13]), "+a"(acc[14]), "+a"(acc[15]));
        if (MODE == 0) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[j][r];
            sums[(size_t)(blockIdx.x * blocks_per_wg + blk) * 256 + tid] = s;
        } else {
            // lane (x row l31, half lh) holds output columns 32 j + 8 (r >> 2) + 4 lh + (r & 3)
            float* crow = C + (size_t)(row_base + l31) * N_OUT;
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4*>(crow + 32 * j + 8 * q + 4 * lh) = f32x4{acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
        }
    }
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

int main(int argc, char** argv) {
    const bool check = argc > 1 && !strcmp(argv[1], "--check");
    const int M = check ? 256 : 65536;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    // planes: A scaled (hi, lo' = 2^11 (x - hi)), W plain (hi, lo) of values of magnitude ~2^13 (the library's 2^s W)
    std::vector<uint16_t> hA((size_t)M * K_IN * 2), hW((size_t)N_OUT * K_IN * 2);
    std::vector<float> xa((size_t)M * K_IN), xw((size_t)N_OUT * K_IN);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((float)(s >> 8) / 16777216.f - 0.5f) * 2.f; };
    auto idx = [&](size_t m, int k) { return m * 2 * K_IN + (size_t)(k >> 5) * 64 + (k & 31); };
    for (size_t m = 0; m < (size_t)M; ++m)
        for (int k = 0; k < K_IN; ++k) {
            const float v = rnd();
            const uint16_t hi = f2h(v), lo = f2h(2048.f * (v - h2f(hi)));
            hA[idx(m, k)] = hi; hA[idx(m, k) + 32] = lo;
            xa[m * K_IN + k] = h2f(hi) + h2f(lo) / 2048.f;
        }
    for (size_t n = 0; n < (size_t)N_OUT; ++n)
        for (int k = 0; k < K_IN; ++k) {
            const float v = rnd() * 8192.f;
            const uint16_t hi = f2h(v), lo = f2h(v - h2f(hi));
            hW[idx(n, k)] = hi; hW[idx(n, k) + 32] = lo;
            xw[n * K_IN + k] = h2f(hi) + h2f(lo);
        }
    uint16_t *dA, *dW;
    float *dC, *dS;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dW, hW.size() * 2);
    hipMalloc(&dC, (size_t)M * N_OUT * 4); hipMalloc(&dS, (size_t)(M / ROWS_WG) * 256 * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)wp_mlp0_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    hipFuncSetAttribute((const void*)wp_mlp0_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    if (check) {
        hipLaunchKernelGGL(wp_mlp0_kernel<1>, dim3(M / ROWS_WG), dim3(256), LDSB, 0, dA, dW, dC, dS, M, 1, 1);
        std::vector<float> hC((size_t)M * N_OUT);
        hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0.0;
        for (int m = 0; m < M; m += 7)
            for (int n = 0; n < N_OUT; n += 5) {
                double ref = 0.0, mag = 0.0;
                for (int k = 0; k < K_IN; ++k) { ref += (double)xa[(size_t)m * K_IN + k] * xw[(size_t)n * K_IN + k]; mag += fabs((double)xa[(size_t)m * K_IN + k] * xw[(size_t)n * K_IN + k]); }
                worst = fmax(worst, fabs(hC[(size_t)m * N_OUT + n] - ref) / mag);
            }
        printf("check: max |C - ref| / sum |a||w| = %.3g over a sample of outputs (3-product f16x2: the dropped lo x lo term is 2^-22)\n", worst);
        return worst < 2e-6 ? 0 : 1;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct Var { const char* name; const void* fn; };
    const Var vars[] = {{"full stream", (const void*)wp_mlp0_kernel<0, 0>}, {"no LDS-direct loads", (const void*)wp_mlp0_kernel<0, 1>}, {"no multiplies", (const void*)wp_mlp0_kernel<0, 2>},
                        {"no fragment reads", (const void*)wp_mlp0_kernel<0, 4>}, {"no activation loads", (const void*)wp_mlp0_kernel<0, 8>},
                        {"no end-of-step wait / barrier", (const void*)wp_mlp0_kernel<0, 16>}, {"MFMAs + multiplies only", (const void*)wp_mlp0_kernel<0, 29>},
                        {"MFMAs only", (const void*)wp_mlp0_kernel<0, 31>}, {"full stream (again)", (const void*)wp_mlp0_kernel<0, 0>},
                        {"MFMAs only (again)", (const void*)wp_mlp0_kernel<0, 31>}};
    for (const Var& v : vars)
    for (int bpw : {1}) {
        const int grid = M / ROWS_WG / bpw;
        hipFuncSetAttribute(v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
        int Mv = M, bv = bpw;
        double us_r[2];
        for (int ri = 0; ri < 2; ++ri) {
            int kr = ri ? 5 : 1;
            void* args[] = {&dA, &dW, &dC, &dS, &Mv, &bv, &kr};
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernel(v.fn, dim3(grid), dim3(256), args, LDSB, 0);
            hipDeviceSynchronize();
            const int n = 10;
            hipEventRecord(e0);
            for (int rep = 0; rep < n; ++rep) hipLaunchKernel(v.fn, dim3(grid), dim3(256), args, LDSB, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms_; hipEventElapsedTime(&ms_, e0, e1);
            us_r[ri] = 1e3 * ms_ / n;
        }
        const double per_step = (us_r[1] - us_r[0]) / (4.0 * 16.0 * ((M / ROWS_WG + cus - 1) / cus));  // us per K step and CU, overhead-free
        printf("%-32s %6.1f us per launch; steady state %5.0f ns per K step = %4.0f TF fp32-equivalent (of 833 nominal)\n", v.name, us_r[0], 1e3 * per_step,
               2.0 * 128 * 512 * 32 * cus / per_step / 1e6);
    }
    return 0;
}
