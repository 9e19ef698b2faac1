// fp32-in / fp32-out "NT" GEMM of the "f16x2" arithmetic mode:   C[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]) (+ R[m][n])
//
// Arithmetic.  An fp32 operand is carried as two fp16 planes,
//   x = x_hi + 2^-11 x_lo',   x_hi = fp16(x),   x_lo' = fp16(2^11 (x - x_hi))        |x - x_hi - 2^-11 x_lo'| <= 2^-22 |x|
// the low plane kept scaled by 2^11 so that it has the exponent range of the high plane (no fp16 underflow for
// 6e-5 <= |x| <= 65504; below that the representation degrades gracefully to an absolute 2^-36).  The weights are static:
// their planes are made once at commit time (ctx.hip, add_split_h2) from 2^s W (s per matrix: max |2^s w| in [2^13, 2^14))
// as w_hi, w_lo (unscaled low plane - always a normal number at that scale).  The three products that matter,
//   x_hi w_hi + x_hi w_lo + x_lo' (2^-11 w_hi)                          (dropped: lo x lo <= 2^-22 |x w|)
// all land in ONE fp32 accumulator with their true weight; 2^-11 w_hi is made from the w_hi fragment in registers (4
// v_pk_mul_f16, exact).  Every product of two 11-bit significands is exact in fp32: 24 v_mfma_f32_32x32x16_f16 per wave
// and K tile where bf16x3 needs 48.  Representation error of a K=512 contraction: 1.0e-7 rms of mean |C| for
// activations of typical magnitude 1e-4 ... 65504 (fp32 sequential accumulation: 3.6e-7) - tests/test_gpu_kernels.py.
//
// Shape.  The 128 x 128 kernel of gemm_x3.hip run with these planes is bound by the L2 -> CU operand stream, not by the
// matrix pipe: taking the MFMAs out changes 148 us to 129 us at (65536 x 512 x 512), and the stream is then 1.34 GB per
// launch = 10.4 TB/s, the L2 ceiling measured in round 1 (A is re-read by each of the N/128 column tiles, the weight
// planes by every one of the M/128 row tiles).  So this kernel is shaped for bytes per MAC:
//   * 256 x 128 output tile, 8 waves (4 x 2) of 64 x 64: the weight planes are re-read half as often, and two planes
//     instead of three cross L2 at all: 4/128 + 4/256 = 0.047 B per MAC (was 4/128 + 6/128 = 0.078);
//   * LDS double-buffered (2 x 60 KB), ONE barrier per K tile: the fp32 A tile of step k+1 is split into planes and
//     stored while the MFMAs of step k run; global prefetch distance of two K tiles in registers;
//   * one workgroup per CU (2 waves per SIMD), persistent over an XCD-contiguous range of tiles ordered N-fastest, so
//     the column tiles that share an A row block run back to back on one XCD's L2.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.h"

namespace e2emv {

typedef __attribute__((ext_vector_type(16))) float h2_f32x16;
typedef __attribute__((ext_vector_type(4))) float h2_f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 h2_f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 h2_f16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int h2_u32x4;

constexpr int H2_BM = 256, H2_BK = 32, H2_LD = 40;  // LDS rows: 32 halves + 8 pad = 80 B (conflict-free b128)
constexpr int H2_APLANE = H2_BM * H2_LD;
// NJ = 32-column blocks per wave: 2 -> 256 x 128 tile, LDS double-buffered (2 x 60 KB), one barrier per K step;
//                                 4 -> 256 x 256 tile, one 80 KB buffer, two barriers per K step, 128 accumulator registers

struct GemmH2Params {
    const float* A;
    const float* A2;
    const uint16_t* WH;  // fp16 planes [N][{hi, lo}][ldw] of 2^s W
    const float* bias;
    const float* R;
    float* C;
    int64_t ldr, ldc;
    unsigned lda, lda2, ldw;
    int M, N, K, K1;
    int tiles_n, total;
    int relu;
    float out_scale;  // 2^-s
    long long* dbg;   // E2EMV_X3_DEBUG=8: phase timestamps of two workgroups
};

template <int DBG, int NJ>
__global__ __launch_bounds__(512, 1) void gemm_h2_kernel(GemmH2Params p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem_h2[];
    constexpr int H2_BN = 64 * NJ, H2_WPLANE = H2_BN * H2_LD, H2_BUF = 2 * H2_APLANE + 2 * H2_WPLANE;
    constexpr bool DB = NJ == 2;   // LDS double-buffered
    constexpr int WR = H2_BN / 128;  // weight rows per thread and plane

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (p.total + 7) / 8;
    const int t_begin = xcd * per_xcd;
    const int t_end = min(t_begin + per_xcd, p.total);
    int tile = t_begin + slot;
    if (tile >= t_end) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    // activation tile: 256 rows x 32 fp32 = 8 chunks of 16 B per row -> rows a_row + 64 i, i < 4
    const int a_row = tid >> 3, a_c4 = (tid & 7) * 4;
    // weight tile: 128 rows x 2 planes x 4 chunks of 16 B -> row tid >> 2, chunk tid & 3, both planes
    const int w_row = tid >> 2, w_k8 = (tid & 3) * 8;
    const int nk = p.K / H2_BK;

    // operand addresses as 32-bit BYTE offsets from the uniform base pointers (SGPR base + VGPR offset loads, no 64-bit
    // address arithmetic in the K loop; the launcher checks the spans); they belong to the load position (ld_tile, ld_kt),
    // which runs two K steps ahead of the compute position across tiles
    unsigned a_off[4], a2_off[4], w_off[WR];
    auto setup = [&](int t) {
        const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned ra = (unsigned)min(tm * H2_BM + a_row + 64 * i, p.M - 1);
            a_off[i] = (ra * p.lda + a_c4) * 4u;
            a2_off[i] = (ra * p.lda2 + a_c4) * 4u;
        }
#pragma unroll
        for (int r = 0; r < WR; ++r) w_off[r] = ((unsigned)min(tn * H2_BN + w_row + 128 * r, p.N - 1) * 2u * p.ldw + w_k8) * 2u;
    };

    h2_f32x4 ra[4];
    h2_u32x4 rw[WR][2];
    int ld_tile = tile, ld_kt = 0;
    // straight-line: issue the loads of the current load position (no control flow - this sits in the MFMA block)
    auto gload = [&]() {
        const int k = ld_kt * H2_BK;
        const bool first = k < p.K1;
        const char* base = reinterpret_cast<const char*>(first ? p.A : p.A2);
        const unsigned kk = (unsigned)(first ? k : k - p.K1) * 4u;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const h2_f32x4*>(base + ((first ? a_off[i] : a2_off[i]) + kk));
        const char* wb = reinterpret_cast<const char*>(p.WH);
#pragma unroll
        for (int r = 0; r < WR; ++r) {
            rw[r][0] = *reinterpret_cast<const h2_u32x4*>(wb + (w_off[r] + 2u * k));
            rw[r][1] = *reinterpret_cast<const h2_u32x4*>(wb + (w_off[r] + 2u * (p.ldw + k)));
        }
    };
    // move the load position one K step on; past the last step of the last tile it stays put (the loads then re-fetch
    // that step, harmlessly, and nothing stores them)
    auto advance = [&]() {
        if ((DBG & 2)) return;  // (profiling: keep re-loading the first K tile - L2 hits only)
        if (ld_kt + 1 < nk) { ++ld_kt; return; }
        if (ld_tile + slots < t_end) {
            asm volatile("" ::: "memory");  // keeps this a (uniform) branch: if-converted, setup's ~30 VALU ran in every K step
            ld_tile += slots;
            ld_kt = 0;
            setup(ld_tile);
        }
    };
    auto lstore = [&](int buf) {
        uint16_t* As = smem_h2 + buf * H2_BUF;
        uint16_t* Ws = As + 2 * H2_APLANE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // hi = fp16(v): one v_cvt_pk_f16_f32 per pair; lo' = 2^11 (v - hi): v_fma_mix{lo,hi}_f16(hi as fp16, -2048, 2048 v),
            // the exact fp32 residual rounded once to fp16 - 2.5 VALU per element
            typedef __attribute__((ext_vector_type(2))) float f32x2;
            typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            u32x2 h0, h1;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float v0 = ra[i][2 * e], v1 = ra[i][2 * e + 1];
                const f32x2 vv = {v0, v1};
                const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(vv, f16x2));
                const float s0 = v0 * 2048.f, s1 = v1 * 2048.f;
                unsigned lo;
                asm("v_fma_mixlo_f16 %0, %1, %4, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %4, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                    : "=&v"(lo) : "v"(hi), "v"(s0), "v"(s1), "s"(-2048.f));
                h0[e] = hi; h1[e] = lo;
            }
            uint16_t* dst = &As[(a_row + 64 * i) * H2_LD + a_c4];
            *reinterpret_cast<u32x2*>(dst) = h0;
            *reinterpret_cast<u32x2*>(dst + H2_APLANE) = h1;
        }
#pragma unroll
        for (int r = 0; r < WR; ++r) {
            *reinterpret_cast<h2_u32x4*>(&Ws[(w_row + 128 * r) * H2_LD + w_k8]) = rw[r][0];
            *reinterpret_cast<h2_u32x4*>(&Ws[H2_WPLANE + (w_row + 128 * r) * H2_LD + w_k8]) = rw[r][1];
        }
    };

    h2_f32x16 acc[NJ][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    };
    // FIRST: first K step of an output tile - the accumulators start from the MFMA's zero C operand instead of being
    // cleared by 128 v_mov per tile (a fifth of the per-tile VALU work at K = 256)
    auto compute = [&](int buf, auto FIRST) {
        constexpr bool first_step = decltype(FIRST)::value;
        const uint16_t* as = smem_h2 + buf * H2_BUF + (wr * 64 + l31) * H2_LD + lh * 8;
        const uint16_t* bs = smem_h2 + buf * H2_BUF + 2 * H2_APLANE + (wc * 32 * NJ + l31) * H2_LD + lh * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h2_f16x8 x[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) x[t][pl] = *reinterpret_cast<const h2_f16x8*>(as + pl * H2_APLANE + t * 32 * H2_LD + ks * 16);
            // weights are the MFMA A operand (rows -> registers), activations B (rows -> lanes); per 32-column block the
            // two accumulators alternate, smallest terms first
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                h2_f16x8 w[3];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) w[pl] = *reinterpret_cast<const h2_f16x8*>(bs + pl * H2_WPLANE + j * 32 * H2_LD + ks * 16);
                w[2] = w[0] * (_Float16)(1.f / 2048.f);  // 2^-11 w_hi: exact (w_hi is >= 2^-3 wherever it matters)
                if (DBG & 1) {  // profiling: operand pipeline only
                    acc[j][0][ks] += (float)x[0][0][0] + (float)x[0][1][0] + (float)x[1][0][0] + (float)x[1][1][0] + (float)w[0][0] + (float)w[1][0];
                    continue;
                }
                constexpr int PW[3] = {1, 2, 0}, PX[3] = {0, 1, 0};  // x_hi w_lo, x_lo' (2^-11 w_hi), x_hi w_hi
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (first_step && ks == 0 && q == 0) {
                            const h2_f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[PW[q]], x[i][PX[q]], zero, 0, 0, 0);
                        } else {
                            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[PW[q]], x[i][PX[q]], acc[j][i], 0, 0, 0);
                        }
                    }
            }
        }
    };
    // Epilogue through LDS.  In the accumulator layout a lane owns a ROW (4 consecutive columns per register group), so a
    // direct store instruction touches 64 different cache lines with 16 bytes each: 32 such instructions per lane and tile
    // cost ~14 k cycles per 256 x 256 tile (measured; line processing, not bandwidth) - 23 % of a K = 256 GEMM.  Each wave
    // therefore transposes its tile through a private LDS slab of 32 rows x (32 JC + 4) floats: 4 JC ds_write_b128 per pass,
    // read back row-contiguous (a store instruction then covers whole 128 / 256-byte row segments: 8 lines instead of 64;
    // the residual read R becomes coalesced the same way).  Slab region: behind the tile buffer (NJ = 4), or the tile
    // buffer that is idle during the epilogue (NJ = 2: the other one already holds the next tile's first K step).
    constexpr int JC = NJ == 4 ? 2 : 1;            // 32-column blocks per pass
    constexpr int SLD = 32 * JC + 4;               // slab row stride in floats (68 / 36: conflict-free b128 writes)
    constexpr int LPR = 8 * JC;                    // lanes per slab row on the way out (float4 each)
    constexpr int RPI = 64 / LPR;                  // rows per store instruction
    auto epilogue = [&](int t, int free_buf) {
        float* slab = reinterpret_cast<float*>(smem_h2 + (DB ? free_buf * H2_BUF : H2_BUF)) + wave * 32 * SLD;
        const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
        const int o_row = lane / LPR, o_col = (lane % LPR) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jc = 0; jc < NJ / JC; ++jc) {
#pragma unroll
                for (int jj = 0; jj < JC; ++jj)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        h2_f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[jc * JC + jj][i][4 * g + e] * p.out_scale;
                        *reinterpret_cast<h2_f32x4*>(&slab[l31 * SLD + jj * 32 + 8 * g + 4 * lh]) = v;
                    }
                const int n = tn * H2_BN + wc * 32 * NJ + jc * 32 * JC + o_col;
                h2_f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
                if (p.bias && n < p.N) bias4 = *reinterpret_cast<const h2_f32x4*>(p.bias + n);
#pragma unroll
                for (int k = 0; k < 32 / RPI; ++k) {
                    const int r = o_row + RPI * k;
                    h2_f32x4 v = *reinterpret_cast<const h2_f32x4*>(&slab[r * SLD + o_col]);
                    const int m = tm * H2_BM + wr * 64 + i * 32 + r;
                    if (m >= p.M || n >= p.N) continue;
                    v += bias4;
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = relu_nan(v[e]);
                    }
                    if (p.R) v += *reinterpret_cast<const h2_f32x4*>(p.R + (int64_t)m * p.ldr + n);
                    *reinterpret_cast<h2_f32x4*>(p.C + (int64_t)m * p.ldc + n) = v;
                }
            }
    };

    // pipeline: step g = (tile, kt) in execution order; LDS buffer g & 1 holds step g while registers hold step g + 1.
    // One barrier per step.  Two alternatives were measured and dropped (s_memtime stamps, E2EMV_X3_DEBUG=8):
    //   * the two waves of a SIMD in opposite phase (one splits/stores/loads while the other issues MFMAs, barrier per half
    //     step): 166 us against 142 us at (65536 x 512 x 512) - the MFMA phase of a wave stretches from 1000 to 1500 cycles
    //     and the split phase from 650 to 900-1800 when they run side by side, i.e. the SIMD does not overlap them;
    //   * sched_group_barrier interleaving of the split into the MFMA shadows of the same wave: not honoured by hipcc 7.2
    //     for this block.
    // A step costs ~3600 cycles against 2 x 768 of matrix-pipe time per SIMD: issuing the 6 global loads alone stalls
    // 200-1000 cycles (the L2 -> CU path is saturated at ~8 TB/s, 14 B/clk/CU), the split + LDS stores take 450-850.
    if (DBG & 1) zero_acc();
    setup(tile);
    gload();      // step 0
    advance();
    lstore(0);
    gload();      // step 1
    advance();
    __syncthreads();
    int buf = 0, dbg_n = 0;
    // one K step; FIRST = first step of an output tile (accumulators start from the MFMA's zero C operand).  The first step
    // is peeled out of the K loop so that each copy of the body keeps the register footprint of a single one.
    auto step = [&](auto FIRST) {
        long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if (DBG & 8) t0 = clock64();
        if constexpr (DB) {
            lstore(buf ^ 1);  // step g + 1 into the other buffer (its readers passed the barrier of step g - 1)
            if (DBG & 8) t1 = clock64();
            gload();          // step g + 2
            if (DBG & 8) t2 = clock64();
            compute(buf, FIRST);
            if (DBG & 8) t3 = clock64();
            advance();
            __syncthreads();
        } else {
            compute(0, FIRST);
            if (DBG & 8) t1 = clock64();
            __syncthreads();  // every wave is done reading step g
            lstore(0);        // step g + 1 (after the last step: stale registers nobody reads)
            if (DBG & 8) t2 = clock64();
            gload();          // step g + 2
            if (DBG & 8) t3 = clock64();
            advance();
            __syncthreads();
        }
        if (DBG & 8) {
            const long long t4 = clock64();
            if (p.dbg && lane == 0 && dbg_n < 48 && (blockIdx.x == 0 || blockIdx.x == 101)) {
                long long* o = p.dbg + ((blockIdx.x ? 1 : 0) * 8 + wave) * 48 * 5 + dbg_n * 5;
                o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4;
                ++dbg_n;
            }
        }
        if constexpr (DB) buf ^= 1;
    };
    for (;;) {
        step(std::true_type{});
        for (int kt = 1; kt < nk; ++kt) step(std::false_type{});
        epilogue(tile, buf ^ 1);  // (DB: `buf` now names the buffer of the NEXT step; the other one was just computed from)
        tile += slots;
        if (tile >= t_end) break;
        if constexpr (DB) __syncthreads();  // the slabs live in the buffer the next step's lstore fills
        if (DBG & 1) zero_acc();
    }
}

int launch_gemm_h2(e2emv_ctx* ctx, const GemmArgs& a, const uint16_t* WH, int64_t ldw, float out_scale, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.batch != 1) return set_err(ctx, E2EMV_ESHAPE, "gemm_h2: empty problem or batch != 1");
    const int K1 = a.A2 ? a.K1 : a.K;
    if (a.K % 32 || K1 % 32 || K1 > a.K || (K1 < a.K && !a.A2)) return set_err(ctx, E2EMV_ESHAPE, "gemm_h2: K=%d K1=%d must be multiples of 32", a.K, K1);
    if (!WH || !a.C || a.N % 4 || a.ldc % 4 || (uintptr_t)a.C % 16 || (a.bias && (uintptr_t)a.bias % 16) ||
        (a.R && (a.ldr % 4 || (uintptr_t)a.R % 16)) || a.lda % 4 || (a.A2 && a.lda2 % 4) || ldw % 8 || a.scale != 1.f)
        return set_err(ctx, E2EMV_ESHAPE, "gemm_h2: needs 16-byte aligned rows (N %% 4, ld %% 4, ldw %% 8) and scale 1");
    if ((int64_t)a.M * a.lda >= (int64_t)1 << 30 || (a.A2 && (int64_t)a.M * a.lda2 >= (int64_t)1 << 30) || (int64_t)a.N * 2 * ldw >= (int64_t)1 << 31)
        return set_err(ctx, E2EMV_ESHAPE, "gemm_h2: operand larger than 4 GB (M=%d lda=%lld): 32-bit byte offsets", a.M, (long long)a.lda);
    GemmH2Params p;
    p.A = a.A; p.A2 = a.A2 ? a.A2 : a.A; p.WH = WH; p.bias = a.bias; p.R = a.R; p.C = a.C;
    p.lda = (unsigned)a.lda; p.lda2 = (unsigned)(a.A2 ? a.lda2 : a.lda); p.ldw = (unsigned)ldw; p.ldr = a.ldr; p.ldc = a.ldc;
    p.M = a.M; p.N = a.N; p.K = a.K; p.K1 = K1;
    // tile shape: 256 x 256 when N fills it (per MAC 4/256 + 4/256 B cross L2 instead of 4/128 + 4/256), else 256 x 128
    static int nj_env = -1;  // E2EMV_H2_NJ=2|4 forces a shape
    static int dbg = -1;     // profiling knob E2EMV_X3_DEBUG: 1 no MFMA, 2 L2-resident operands only, 8 phase timestamps
    if (nj_env < 0) nj_env = dbg_knob("E2EMV_H2_NJ", 0);
    if (dbg < 0) dbg = dbg_knob("E2EMV_X3_DEBUG", 0);
    const int nj = nj_env == 2 || nj_env == 4 ? nj_env : (a.N % 256 == 0 ? 4 : 2);
    const int bn = 64 * nj;
    const int tiles_m = (a.M + H2_BM - 1) / H2_BM;
    p.tiles_n = (a.N + bn - 1) / bn;
    p.total = tiles_m * p.tiles_n;
    p.relu = a.relu ? 1 : 0;
    p.out_scale = out_scale;
    p.dbg = nullptr;
    const int per_xcd = (p.total + 7) / 8;
    const int sl = std::min(per_xcd, std::max(1, ctx->num_cus / 8));
    // NJ = 2: two tile buffers (the idle one carries the epilogue slabs); NJ = 4: one tile buffer + 8 slabs of 32 x 68 floats
    const size_t lds = nj == 2 ? sizeof(uint16_t) * 2 * (2 * H2_APLANE + 2 * bn * H2_LD)
                               : sizeof(uint16_t) * (2 * H2_APLANE + 2 * bn * H2_LD) + sizeof(float) * 8 * 32 * 68;
    const void* fn = nullptr;
    switch (dbg * 10 + nj) {
        case 12: fn = reinterpret_cast<const void*>(gemm_h2_kernel<1, 2>); break;
        case 14: fn = reinterpret_cast<const void*>(gemm_h2_kernel<1, 4>); break;
        case 22: fn = reinterpret_cast<const void*>(gemm_h2_kernel<2, 2>); break;
        case 24: fn = reinterpret_cast<const void*>(gemm_h2_kernel<2, 4>); break;
        case 82: fn = reinterpret_cast<const void*>(gemm_h2_kernel<8, 2>); break;
        case 84: fn = reinterpret_cast<const void*>(gemm_h2_kernel<8, 4>); break;
        default: fn = nj == 2 ? reinterpret_cast<const void*>(gemm_h2_kernel<0, 2>) : reinterpret_cast<const void*>(gemm_h2_kernel<0, 4>);
    }
    if (int rc = ensure_dynamic_lds(ctx, fn, lds)) return rc;
    long long* d_dbg = nullptr;
    const size_t nb = sizeof(long long) * 2 * 8 * 48 * 5;
    if (dbg == 8) {  // phase timestamps of workgroups 0 and 101, printed after the launch (host-synchronising; profiling only)
        static long long* d_buf = nullptr;
        if (!d_buf) E2EMV_HIP(ctx, hipMalloc((void**)&d_buf, nb));
        d_dbg = d_buf;
        E2EMV_HIP(ctx, hipMemsetAsync(d_dbg, 0, nb, s));
        p.dbg = d_dbg;
    }
    void* args[] = {&p};
    E2EMV_HIP(ctx, hipLaunchKernel(fn, dim3(8 * sl), dim3(512), args, lds, s));
    E2EMV_CHECK_LAUNCH(ctx, "gemm_h2_kernel");
    if (dbg == 8) {
        E2EMV_HIP(ctx, hipStreamSynchronize(s));
        std::vector<long long> h(2 * 8 * 48 * 5);
        E2EMV_HIP(ctx, hipMemcpy(h.data(), d_dbg, nb, hipMemcpyDeviceToHost));
        static int printed = 0;
        if (printed++ < 2)
            for (int wg = 0; wg < 2; ++wg)
                for (int w = 0; w < 8; w += 3) {
                    const long long* o = &h[((size_t)wg * 8 + w) * 48 * 5];
                    fprintf(stderr, "gemm_h2 M=%d N=%d K=%d tile 256x%d wg %d wave %d: cycles between the stamps of a step | total\n", p.M, p.N, p.K, bn, wg ? 101 : 0, w);
                    for (int i = 0; i < 40; ++i) {
                        const long long* t = o + i * 5;
                        if (!t[0]) break;
                        fprintf(stderr, "  %2d: %5lld %5lld %5lld %5lld | %5lld   (to next step %lld)\n", i, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3],
                                t[4] - t[0], i + 1 < 48 && t[5] ? t[5] - t[4] : 0);
                    }
                }
    }
    return E2EMV_OK;
}

}  // namespace e2emv
