// Plane-in / plane-out "NT" GEMM of the f16x2 arithmetic mode:  C[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]) (+ R[m][n])
//
// Same arithmetic as gemm_h2.hip (x = x_hi + 2^-11 x_lo', weights 2^s W = w_hi + w_lo, three fp16 MFMA products per
// block into one fp32 accumulator), but the activation arrives ALREADY as the two planes (p2.h): whoever produced it
// split it once, in its epilogue.  What that buys in the K loop, per 256 x 256 x 32 step and workgroup:
//   * no fp32 -> plane split (40 VALU per thread) and no ds_write at all: the 64 KB of operand planes go from global
//     memory straight into LDS, 64 buffer_load_dwordx4 ... lds per step (8 per wave), each a full 128-byte line per row;
//   * LDS double-buffered (2 x 64 KB), ONE barrier per K step: the loads of step g + 1 are issued right after the barrier
//     that opens step g and have that whole step (48 MFMAs per wave) to land;
//   * bank-conflict-free fragment reads without padding (XOR swizzle applied on the source address, p2.h).
// The epilogue goes through per-wave LDS slabs (32 rows x 36 floats) so that bias / ReLU / residual / the plane split
// work on 8 consecutive columns of one row per lane and every store covers 64-byte row segments; the residual is read
// back from its planes (22 bits) in the same layout.  Output kinds: fp32, scaled planes (the next GEMM's operand), or
// the attention operands q | k (plain planes, q pre-scaled) + V^T (transposed plain planes, keys in the order of the
// transposed-score registers) - attention_p2.hip then never touches an fp32 q|k|v matrix.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "p2.h"

namespace e2emv {

typedef __attribute__((ext_vector_type(16))) float p2_f32x16;

constexpr int P2_BM = 256, P2_BN = 256, P2_BK = 32;
constexpr int P2_ROWB = 128;                    // bytes of one tile row per K step: 32 hi halves | 32 lo halves
constexpr int P2_TILEB = P2_BM * P2_ROWB;       // 32 KB per operand tile
constexpr int P2_BUFB = 2 * P2_TILEB;           // A tile | W tile
constexpr int P2_SLABB = 32 * 32 * 4;            // one epilogue slab per wave (32 rows x 32 floats), behind the tile buffers
constexpr int P2_LDSB = 2 * P2_BUFB + 8 * P2_SLABB;  // 160 KB: the whole LDS of a CU

struct GemmP2Params {
    const uint16_t* A;
    const uint16_t* A2;
    const uint16_t* W;
    unsigned a_bytes, a2_bytes, w_bytes;  // extents for the buffer descriptors
    unsigned lda_b, lda2_b, ldw_b;        // row strides in bytes
    const float* bias;
    const uint16_t* Rp;
    float* C32;
    uint16_t* Cp;
    uint16_t* Vt;
    int64_t ldc, ldr;
    int M, N, K, K1;
    int tiles_n, total, relu;
    float out_scale;
    float col_scale[3];
    int n_rows, heads;
    const int* EA;   // tile exponents (p2.h); null = all zero
    const int* EA2;
    const int* ER;
    int* EC;
    int* EVt;
    const float* AR;  // max |value| of the residual's 64 x 64 blocks (true units) - the bound that picks the output exponent
    float* AC;        // the same of the output (written when the output is a later residual: x)
    int eld_a, eld_a2, eld_r, eld_c;  // entries per 64-row block
    float bias_amax;
    unsigned* stats;  // [0]: number of output blocks that needed a non-zero exponent
    char* dummy;     // 4 KB: target of the stores of rows / columns beyond the matrix (a wave always issues all its stores)
    long long* dbg;  // E2EMV_STAMPS builds only: phase timestamps of two workgroups
};

// MFMAs of the pipelined K step as asm statements (order = source order; see compute_p below)
__device__ __forceinline__ void gp_mfma(p2_f32x16& c, p2_f16x8 a, p2_f16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void gp_mfma0(p2_f32x16& c, p2_f16x8 a, p2_f16x8 b) {  // zero C operand: the first product of a tile
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ unsigned gp_pk_mul(unsigned x, unsigned k) {  // two fp16 products (2^-11 w_hi)
    unsigned d;
    asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(d) : "v"(x), "v"(k));
    return d;
}

// ---- epilogue arithmetic, one instruction where hipcc needs several (the epilogue of a tile is ~1000 VALU instructions per
// wave with the matrix pipe idle: its instruction count is its time)
// max(a, |x|, |y|): fmaxf(fabsf()) compiles to a canonicalising v_max_f32 |x|, |x| per value in front of the maximum
__device__ __forceinline__ float gp_amax3(float a, float x, float y) {
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(a) : "v"(x), "v"(y));
    return a;
}
// ReLU of 8 values as x <- x + |x| = 2 max(x, 0), in place, skipped when `on` is 0 - ONE statement with the branch inside: a
// source-level `if` around in-place updates makes hipcc copy all 8 registers on both arms (12 v_mov per arm), a select
// costs 2 instructions per value.  NaN stays NaN (ReLU must not hide one: common.h relu_nan; -inf, which has no finite
// origin, turns NaN instead of 0); the factor 1/2 goes into the scale that follows (a power of two: exact)
__device__ __forceinline__ void gp_relu2x8(p2_f32x4& v0, p2_f32x4& v1, int on) {
    float a = v0[0], b = v0[1], c = v0[2], d = v0[3], e = v1[0], f = v1[1], g = v1[2], h = v1[3];
    asm("s_cmp_eq_u32 %8, 0\n\t"
        "s_cbranch_scc1 1f\n\t"
        "v_add_f32_e64 %0, %0, |%0|\n\t"
        "v_add_f32_e64 %1, %1, |%1|\n\t"
        "v_add_f32_e64 %2, %2, |%2|\n\t"
        "v_add_f32_e64 %3, %3, |%3|\n\t"
        "v_add_f32_e64 %4, %4, |%4|\n\t"
        "v_add_f32_e64 %5, %5, |%5|\n\t"
        "v_add_f32_e64 %6, %6, |%6|\n\t"
        "v_add_f32_e64 %7, %7, |%7|\n"
        "1:"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(on) : "scc");
    v0 = p2_f32x4{a, b, c, d};
    v1 = p2_f32x4{e, f, g, h};
}
// fp32 values of a pair of scaled-plane elements, hi + 2^-11 lo' (exact), straight from the packed halves
__device__ __forceinline__ p2_f32x2 gp_join_scaled(unsigned hi, unsigned lo) {
    float x0, x1;
    asm("v_fma_mix_f32 %0, %2, %4, %3 op_sel_hi:[1,0,1]\n\tv_fma_mix_f32 %1, %2, %4, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]"
        : "=&v"(x0), "=&v"(x1) : "v"(lo), "v"(hi), "s"(1.f / 2048.f));
    return {x0, x1};
}
// 8 values -> 4 + 4 packed plane words.  a = the values, b = 2048 a (scaled planes; both products of ONE fp32 value with
// powers of two) or b = a (plain planes, K = -1): hi = fp16(a), lo = fp16(b - K' hi) with K' = 2048 or 1
template <bool SCALED>
__device__ __forceinline__ void gp_split8(const p2_f32x4& a0, const p2_f32x4& a1, const p2_f32x4& b0, const p2_f32x4& b1, p2_u32x4& hi, p2_u32x4& lo) {
    const p2_f32x2 q0 = {a0[0], a0[1]}, q1 = {a0[2], a0[3]}, q2 = {a1[0], a1[1]}, q3 = {a1[2], a1[3]};
    const unsigned h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(q0, p2_f16x2)), h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(q1, p2_f16x2));
    const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(q2, p2_f16x2)), h3 = __builtin_bit_cast(unsigned, __builtin_convertvector(q3, p2_f16x2));
    unsigned l0, l1, l2, l3;
    // (the four low words first, their high halves behind them: no v_fma_mixhi reads the word the instruction before it wrote)
    asm("v_fma_mixlo_f16 %0, %4, %16, %8 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %5, %16, %10 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %2, %6, %16, %12 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %7, %16, %14 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %4, %16, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %5, %16, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %2, %6, %16, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %7, %16, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "s_nop 1"
        : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "v"(h0), "v"(h1), "v"(h2), "v"(h3), "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3]), "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3]),
          "s"(SCALED ? -2048.f : -1.f));
    hi = p2_u32x4{h0, h1, h2, h3};
    lo = p2_u32x4{l0, l1, l2, l3};
}

// DBG (instantiated only in -DE2EMV_STAMPS builds, tools/p2_stamps.py): 1 no MFMA, 2 no operand loads after the first two K
// steps, 4 no epilogue, 8 s_memtime stamps per K step, 16 loads of step g + 1 issued one per MFMA group, 32 every wave issues
// its loads before it computes (no opposite orders on a SIMD), 256 vmcnt(0) at every step (no store overlap)
template <int OUT, bool HAS_R = false, int DBG = 0>
__global__ __launch_bounds__(512, 1) void gemm_p2_kernel(GemmP2Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem_p2[];

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (p.total + 7) / 8;
    const int t_begin = xcd * per_xcd;
    const int t_end = min(t_begin + per_xcd, p.total);
    int tile = t_begin + slot;
    if (tile >= t_end) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int nk = p.K / P2_BK, nk1 = p.K1 / P2_BK;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.A2), 0, (int)p.a2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.W), 0, (int)p.w_bytes, 0x00020000);

    // ---- loader: wave w moves rows 32 w .. 32 w + 31 of the activation tile and of the weight tile, 4 + 4 LDS-direct loads
    // per K step; one load = 8 rows x 128 B, lane -> (row lane >> 3, LDS position lane & 7), source chunk = position ^ swizzle
    const int ld_r = lane >> 3, ld_p = lane & 7;
    unsigned a_vo[4], a2_vo[4], w_vo[4];
    auto setup = [&](int t) {
        const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 32 * wave + 8 * i + ld_r;
            const unsigned c = (unsigned)(ld_p ^ ((row >> 1) & 7));
            const unsigned gm = (unsigned)min(tm * P2_BM + row, p.M - 1);
            a_vo[i] = gm * p.lda_b + c * 16u;
            a2_vo[i] = gm * p.lda2_b + c * 16u;
            const unsigned gn = (unsigned)min(tn * P2_BN + row, p.N - 1);
            w_vo[i] = gn * p.ldw_b + c * 16u;
        }
    };
    // piece i of a K step: 0..3 activation rows, 4..7 weight rows
    auto issue_piece = [&](int buf, int kt, int i, unsigned dep) {
        char* dst = smem_p2 + buf * P2_BUFB + 32 * wave * P2_ROWB;
        if (i < 4) {
            if (kt < nk1) p2_glds16(rsA, dst + i * 1024, a_vo[i] + dep, (unsigned)kt * 128u);
            else p2_glds16(rsA2, dst + i * 1024, a2_vo[i] + dep, (unsigned)(kt - nk1) * 128u);
        } else {
            p2_glds16(rsW, dst + P2_TILEB + (i - 4) * 1024, w_vo[i - 4] + dep, (unsigned)kt * 128u);
        }
    };
    auto issue = [&](int buf, int kt, unsigned dep) {
        char* dst = smem_p2 + buf * P2_BUFB + 32 * wave * P2_ROWB;
        if (kt < nk1) {
            const unsigned so = (unsigned)kt * 128u;
#pragma unroll
            for (int i = 0; i < 4; ++i) p2_glds16(rsA, dst + i * 1024, a_vo[i] + dep, so);
        } else {
            const unsigned so = (unsigned)(kt - nk1) * 128u;
#pragma unroll
            for (int i = 0; i < 4; ++i) p2_glds16(rsA2, dst + i * 1024, a2_vo[i] + dep, so);
        }
        const unsigned sw = (unsigned)kt * 128u;
#pragma unroll
        for (int i = 0; i < 4; ++i) p2_glds16(rsW, dst + P2_TILEB + i * 1024, w_vo[i] + dep, sw);
    };

    // ---- fragments: lane (row l31, k half lh); chunk index c = 4 plane + 2 ks + lh, stored at position c ^ ((l31 >> 1) & 7)
    const int swz = (l31 >> 1) & 7;
    p2_f32x16 acc[4][2];
    // (`late`, DBG & 16 in measurement builds only: the loads of the next step go out one per MFMA group, each behind a
    // scheduling-only dependency on that group's accumulator - an empty asm, no instruction)
    auto compute = [&](int buf, auto FIRST, bool late, int ld_buf, int ld_k) {
        constexpr bool first_step = decltype(FIRST)::value;
        const char* xs = smem_p2 + buf * P2_BUFB + (wr * 64 + l31) * P2_ROWB;
        const char* ws = smem_p2 + buf * P2_BUFB + P2_TILEB + (wc * 128 + l31) * P2_ROWB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            p2_f16x8 x[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    x[t][pl] = *reinterpret_cast<const p2_f16x8*>(xs + t * 32 * P2_ROWB + (((4 * pl + 2 * ks + lh) ^ swz) << 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p2_f16x8 w[3];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    w[pl] = *reinterpret_cast<const p2_f16x8*>(ws + j * 32 * P2_ROWB + (((4 * pl + 2 * ks + lh) ^ swz) << 4));
                w[2] = w[0] * (_Float16)(1.f / 2048.f);  // 2^-11 w_hi (exact wherever it matters: gemm_h2.hip)
                constexpr int PW[3] = {1, 2, 0}, PX[3] = {0, 1, 0};  // x_hi w_lo, x_lo' (2^-11 w_hi), x_hi w_hi: smallest first
                if (DBG & 1) {  // operand pipeline only
                    asm volatile("" :: "v"(x[0][0]), "v"(x[0][1]), "v"(x[1][0]), "v"(x[1][1]), "v"(w[0]), "v"(w[1]), "v"(w[2]));
                    if ((DBG & 16) && late) issue_piece(ld_buf, ld_k, 4 * ks + j, 0u);
                    continue;
                }
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (first_step && ks == 0 && q == 0) {
                            const p2_f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[PW[q]], x[i][PX[q]], zero, 0, 0, 0);
                        } else {
                            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[PW[q]], x[i][PX[q]], acc[j][i], 0, 0, 0);
                        }
                    }
                if ((DBG & 16) && late) {
                    unsigned dep = 0;
                    asm("" : "+v"(dep) : "v"(acc[j][0]));
                    issue_piece(ld_buf, ld_k, 4 * ks + j, dep);
                }
            }
        }
    };

    // ---- the same K step, software-pipelined inside the wave (the default; `compute` above is kept for the measurement
    // build's ablations).  hipcc's schedule of `compute` reads a group's weight fragments right in front of its MFMAs and
    // waits for them at once (lgkmcnt(0) behind the ds_reads): every group of 6 MFMAs opened with an exposed LDS round trip
    // that only the SIMD's other wave could fill.  Here the 8 groups of a step (2 k-halves x 4 weight row blocks) run as one
    // stream: the fragments of group g + 1 (and the activation fragments of the next k-half) are read under the MFMAs of
    // group g, the four v_pk_mul_f16 that make 2^-11 w_hi sit behind the group's first MFMA; MFMAs and multiplies are asm
    // statements (source order = machine order, fenced per slot), so the distances the hardware needs are kept by
    // placement: multiplies -> the MFMA that reads them: one MFMA and two ds_reads apart.
    auto compute_p = [&](int buf, auto FIRST) {
        constexpr bool first_step = decltype(FIRST)::value;
        const char* xs = smem_p2 + buf * P2_BUFB + (wr * 64 + l31) * P2_ROWB;
        const char* ws = smem_p2 + buf * P2_BUFB + P2_TILEB + (wc * 128 + l31) * P2_ROWB;
        auto rd_x = [&](int ks, int t, int pl) { return *reinterpret_cast<const p2_f16x8*>(xs + t * 32 * P2_ROWB + (((4 * pl + 2 * ks + lh) ^ swz) << 4)); };
        auto rd_w = [&](int ks, int j, int pl) { return *reinterpret_cast<const p2_f16x8*>(ws + j * 32 * P2_ROWB + (((4 * pl + 2 * ks + lh) ^ swz) << 4)); };
        p2_f16x8 xb[2][2][2];  // [k-half parity][row block][plane]
        p2_f16x8 wb[2][2];     // [group parity][plane]
        unsigned k2048 = 0x10001000u;  // two fp16 2^-11
        asm volatile("" : "+v"(k2048));
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) xb[0][t][pl] = rd_x(0, t, pl);
        wb[0][1] = rd_w(0, 0, 1);
        wb[0][0] = rd_w(0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int ks = g >> 2, j = g & 3, gp = g & 1;
            const p2_u32x4 wh = __builtin_bit_cast(p2_u32x4, wb[gp][0]);
            p2_u32x4 w2u;
            const bool z = first_step && ks == 0;
            // slot 0: x_hi w_lo of row block 0; then 2^-11 w_hi
            if (z) gp_mfma0(acc[j][0], wb[gp][1], xb[ks][0][0]); else gp_mfma(acc[j][0], wb[gp][1], xb[ks][0][0]);
#pragma unroll
            for (int e = 0; e < 4; ++e) w2u[e] = gp_pk_mul(wh[e], k2048);
            __builtin_amdgcn_sched_barrier(0);
            // slot 1: x_hi w_lo of row block 1; the next group's weight fragments
            if (z) gp_mfma0(acc[j][1], wb[gp][1], xb[ks][1][0]); else gp_mfma(acc[j][1], wb[gp][1], xb[ks][1][0]);
            if (g < 7) {
                wb[gp ^ 1][1] = rd_w((g + 1) >> 2, (g + 1) & 3, 1);
                wb[gp ^ 1][0] = rd_w((g + 1) >> 2, (g + 1) & 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            const p2_f16x8 w2 = __builtin_bit_cast(p2_f16x8, w2u);
            // slots 2, 3: x_lo' (2^-11 w_hi); the next k-half's activation fragments behind them (groups 2 and 3)
            gp_mfma(acc[j][0], w2, xb[ks][0][1]);
            if (ks == 0 && j == 2) { xb[1][0][0] = rd_x(1, 0, 0); xb[1][1][0] = rd_x(1, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            gp_mfma(acc[j][1], w2, xb[ks][1][1]);
            if (ks == 0 && j == 3) { xb[1][0][1] = rd_x(1, 0, 1); xb[1][1][1] = rd_x(1, 1, 1); }
            __builtin_amdgcn_sched_barrier(0);
            // slots 4, 5: x_hi w_hi
            gp_mfma(acc[j][0], wb[gp][0], xb[ks][0][0]);
            __builtin_amdgcn_sched_barrier(0);
            gp_mfma(acc[j][1], wb[gp][0], xb[ks][1][0]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- epilogue (see the header).  Each wave owns ONE slab of 32 rows x 32 floats behind the two tile buffers (16-byte
    // chunk c of row r at position c ^ (r & 7): conflict-free b128 writes, 2-way reads).  The 8 blocks (32 rows x 32
    // columns) of a wave are software-pipelined through registers: block b + 1 goes through the slab and its residual
    // loads are issued while block b is finished (bias / ReLU / residual / split) and stored.  Two rules keep the store
    // stream asynchronous (gfx950 retires loads AND stores in issue order on one counter, vmcnt):
    //   * no load is issued behind a store whose completion we do not want to wait for: the bias is fetched once, up
    //     front, and the residual of block b + 1 before the stores of block b;
    //   * every wave issues EXACTLY 32 store instructions per tile (rows / columns beyond the matrix go to a dummy line
    //     instead of being skipped), so the K loop of the next tile can wait with a COUNTED vmcnt for its operand loads,
    //     which were issued before these stores, and leave the stores in flight (see the pipeline below).
    auto epilogue = [&](int t, int e_run, int ev) {
        // (the last step's asm MFMAs -> their first VALU readers below: the accumulators pass through the statement, nothing that
        // reads them moves above it)
        asm volatile("s_nop 15" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
        char* sl = smem_p2 + 2 * P2_BUFB + wave * P2_SLABB;
        const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
        const int o_r = lane >> 2, o_c = (lane & 3) * 8;  // row-contiguous view: 16 rows per pass, 8 columns per lane
        const int o_z = o_r & 7;
        const float cs = OUT == P2_OUT_QKV ? p.col_scale[min(tn, 2)] : 1.f;
        char* dummy = p.dummy + lane * 64;
        const float os = p.out_scale * p2_exp2i(e_run);  // the accumulators carry the exponent of the last K block
        // ---- tile exponents of the output: one per 64 columns of this wave's 64 rows, from an upper bound of the values
        const int erow = tm * 4 + wr;
        float rsc[2] = {1.f, 1.f};   // 2^e of the residual blocks
        float osc[2] = {1.f, 1.f};   // 2^-e of the output blocks
        float iosc[2] = {1.f, 1.f};  // 2^e
        float amx[2][2] = {{0.f, 0.f}, {0.f, 0.f}};  // max |final value| of the output blocks (this lane's share, two chains)
        if ((OUT != P2_OUT_F32 && (p.EC || p.EVt)) || (HAS_R && p.ER)) {
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int cb = tn * 4 + wc * 2 + ch;  // 64-column block of the output
                // (the residual block's exponent and max |x| came with the tile's exponent fetch, lanes 32.. - no memory access here)
                int er = 0;
                float ar = 0.f;
                if (HAS_R && p.ER) {
                    er = __builtin_amdgcn_readlane(ev, 32 + ch);
                    ar = p.AR ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(ev, 34 + ch)) : 65536.f * p2_exp2i(er);
                }
                rsc[ch] = p2_exp2i(er);
                int* E = (OUT == P2_OUT_QKV && tn == 2) ? p.EVt : p.EC;
                if (OUT == P2_OUT_F32 || !E) continue;
                float am = 0.f, am1 = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int r = 0; r < 16; r += 4) {
                            am = gp_amax3(am, acc[2 * ch + jj][i][r], acc[2 * ch + jj][i][r + 1]);
                            am1 = gp_amax3(am1, acc[2 * ch + jj][i][r + 2], acc[2 * ch + jj][i][r + 3]);
                        }
                am = p2_wave_max(fmaxf(am, am1));
                const float bound = am * os + p.bias_amax + ar;
                const int e = p2_pick_exponent(bound * cs);
                osc[ch] = p2_exp2i(-e);
                iosc[ch] = p2_exp2i(e);
                const int ecb = (OUT == P2_OUT_QKV && tn == 2) ? wc * 2 + ch : cb;
                const int eld = (OUT == P2_OUT_QKV && tn == 2) ? 4 : p.eld_c;
                if (lane == 0 && erow * 64 < p.M && ecb < eld) {
                    E[erow * eld + ecb] = e;
                    if (e != 0 && p.stats) atomicAdd(p.stats, 1u);
                }
            }
        }
        auto slab_write = [&](int i, int j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                p2_f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[j][i][4 * g + e];
                *reinterpret_cast<p2_f32x4*>(sl + l31 * 128 + (((2 * g + lh) ^ (l31 & 7)) << 4)) = v;
            }
        };
        if (OUT == P2_OUT_QKV && tn == 2) {
            // V^T: lane -> (dim d, 16-byte chunk q of the 32-key block) = 8 keys in accumulator order
            const int dl0 = lane >> 2, q = lane & 3;
            const int rb = 16 * (q >> 1) + 4 * (q & 1);
            // the bias of this lane's 8 dims (4 column blocks x 2 passes) BEFORE the first store: gfx950 retires loads and stores
            // in issue order, a load behind a store waits for that store (one bias load per pass drained the stores 16 times
            // per tile)
            float vb[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int n = tn * P2_BN + wc * 128 + j * 32 + dl0 + 16 * pass;
                    vb[j][pass] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
                }
            const unsigned vt_rd0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sl
                                    + (unsigned)(rb * 128 + ((((dl0 >> 2) ^ (rb & 7)) << 4) | ((dl0 & 3) << 2)));
            const int64_t row2 = 2 * (int64_t)p.n_rows;  // halves per dim row of V^T
            uint16_t* const vt_lane = p.Vt + dl0 * row2 + q * 8;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m0 = tm * P2_BM + wr * 64 + i * 32;
                const int img = m0 / p.n_rows, key0 = m0 - img * p.n_rows;
                const int64_t off_i = (int64_t)img * p.heads * 64 * row2 + (key0 >> 5) * 64;  // (wave-uniform)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    slab_write(i, j);
                    // (v os + b) f with f = 16 x 2^-e: the power of two goes into both operands of ONE fma - the same value
                    const float f = cs * osc[j >> 1], of = os * f;
#pragma unroll
                    for (int pass = 0; pass < 2; ++pass) {
                        const int dl = dl0 + 16 * pass;
                        // slab[key row][dim dl]: row rb + t (t = 0..3; + 8: immediate offset) holds it in chunk (dl >> 2) ^ (row & 7),
                        // i.e. address_t = (address_0 ^ 16 t) + 128 t, pass 1 = pass 0 ^ 64.  Recomputed from ONE register per pass
                        // (the empty asm keeps hipcc from hoisting 8 addresses out of the loops: they were spilled, and a
                        // scratch reload behind a store waits for that store)
                        unsigned a0 = vt_rd0 ^ (pass ? 64u : 0u);
                        asm volatile("" : "+v"(a0));
                        float v[8];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            typedef __attribute__((address_space(3))) const float* lds_f32_t;
                            const unsigned at = (a0 ^ (16u * t)) + 128u * t;
                            v[t] = *reinterpret_cast<lds_f32_t>((uintptr_t)at);
                            v[4 + t] = *reinterpret_cast<lds_f32_t>((uintptr_t)(at + 1024u));
                        }
                        const float bf = vb[j][pass] * f;
                        p2_u32x4 hi, lo;
                        p2_f32x4 w0, w1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { w0[e] = __builtin_fmaf(v[e], of, bf); w1[e] = __builtin_fmaf(v[4 + e], of, bf); }
                        gp_split8<false>(w0, w1, w0, w1, hi, lo);
                        const int dim = (wc * 2 + (j >> 1)) * 64 + (j & 1) * 32 + 16 * pass;  // (+ dl0: in vt_lane)
                        uint16_t* dst = vt_lane + off_i + dim * row2;
                        if (m0 >= p.M || tn * P2_BN + wc * 128 + j * 32 + dl >= p.N) dst = reinterpret_cast<uint16_t*>(dummy);
                        *reinterpret_cast<p2_u32x4*>(dst) = hi;
                        *reinterpret_cast<p2_u32x4*>(dst + 32) = lo;
                    }
                }
            }
            return;
        }
        // bias: without a residual the epilogue issues NO load behind its first store (all four column blocks up front, 32
        // registers); with one, the bias of block b + 1 travels with its residual loads (the registers go to the residual)
        constexpr int NB = HAS_R ? 2 : 4;
        p2_f32x4 bias8[NB][2];
        auto load_bias = [&](int slot, int j) {
            const int n = tn * P2_BN + wc * 128 + j * 32 + o_c;
            bias8[slot][0] = bias8[slot][1] = p2_f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias && n < p.N) bias8[slot][0] = *reinterpret_cast<const p2_f32x4*>(p.bias + n);
            if (p.bias && n + 4 < p.N) bias8[slot][1] = *reinterpret_cast<const p2_f32x4*>(p.bias + n + 4);
        };
        if (!HAS_R) {
#pragma unroll
            for (int j = 0; j < 4; ++j) load_bias(j, j);
        }
        p2_f32x4 rv[2][2][2];              // [parity][pass][half]: the block in the row-contiguous view
        p2_u32x4 rr[HAS_R ? 2 : 1][2][2];  // [parity][pass][plane]: its residual
        // this lane's first row / column of the tile in the row-contiguous view; block (i, j), pass: row0 + 32 i + 16 pass,
        // columns col0 + 32 j .. + 7.  Addresses = one per-tile base + wave-uniform steps (per-store index arithmetic in 64 bits
        // was a tenth of the epilogue's instructions)
        const int row0 = tm * P2_BM + wr * 64 + o_r, col0 = tn * P2_BN + wc * 128 + o_c;
        const int rows_left = p.M - row0;  // row 32 i + 16 pass of the lane exists iff it is < rows_left
        const int64_t rstep = (OUT == P2_OUT_F32 ? 16 : 32) * p.ldc;  // 16 rows further, in elements of the output
        float* const c32_t = OUT == P2_OUT_F32 ? p.C32 + (int64_t)row0 * p.ldc + col0 : nullptr;
        uint16_t* const cp_t = OUT == P2_OUT_F32 ? nullptr : p.Cp + p2_index(row0, col0, p.ldc);
        // ReLU as x + |x| (gp_relu2): the factor 1/2 is folded into the plane scale where one follows directly
        const bool relu = OUT != P2_OUT_QKV && p.relu;
        const bool fold = relu && OUT == P2_OUT_PLANES && !HAS_R;
        const float unfold = relu && !fold ? 0.5f : 1.f;
        auto stage = [&](auto BB) {  // block b -> slab -> registers; residual loads issued
            constexpr int b = decltype(BB)::value;
            constexpr int i = b >> 2, j = b & 3;
            slab_write(i, j);
            if (HAS_R) load_bias(b & 1, j);
            const uint16_t* rcol = nullptr;
            if constexpr (HAS_R) {
                const int nc = min(col0 + j * 32, p.N - 8);
                rcol = p.Rp + ((nc >> 5) * 64 + (nc & 31));
            }
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int r = o_r + 16 * pass;
                const int c0 = 2 * (lane & 3);
                rv[b & 1][pass][0] = *reinterpret_cast<const p2_f32x4*>(sl + r * 128 + ((c0 ^ o_z) << 4));
                rv[b & 1][pass][1] = *reinterpret_cast<const p2_f32x4*>(sl + r * 128 + (((c0 + 1) ^ o_z) << 4));
                if constexpr (HAS_R) {
                    const int m = min(row0 + i * 32 + 16 * pass, p.M - 1);
                    const uint16_t* rp = rcol + (int64_t)m * (2 * p.ldr);
                    rr[b & 1][pass][0] = *reinterpret_cast<const p2_u32x4*>(rp);
                    rr[b & 1][pass][1] = *reinterpret_cast<const p2_u32x4*>(rp + 32);
                }
            }
        };
        auto finish = [&](auto BB) {
            constexpr int b = decltype(BB)::value;
            constexpr int i = b >> 2, j = b & 3;
            const bool col_ok = col0 + j * 32 < p.N;
            // plane scale of the block (wave-uniform): 2^-e of its tile exponent (x the column scale of q / k), x 1/2 behind x + |x|
            const float fa = OUT == P2_OUT_F32 ? 1.f : cs * osc[j >> 1] * (fold ? 0.5f : 1.f);
            // scaled planes: fa is a power of two and goes into the operands of the first fma (the same values; ReLU and the
            // residual sum commute with it) - the bias once per use of its registers, the accumulator scale as a uniform
            float osf = os, rsf = 1.f;
            if constexpr (HAS_R) rsf = rsc[j >> 1];
            if constexpr (OUT == P2_OUT_PLANES) {
                osf = os * fa;
                rsf *= fa;
                if (HAS_R || i == 0) { bias8[HAS_R ? (b & 1) : j][0] *= fa; bias8[HAS_R ? (b & 1) : j][1] *= fa; }
            }
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const bool ok = col_ok && i * 32 + 16 * pass < rows_left;
                p2_f32x4 v0 = rv[b & 1][pass][0] * osf + bias8[HAS_R ? (b & 1) : j][0];
                p2_f32x4 v1 = rv[b & 1][pass][1] * osf + bias8[HAS_R ? (b & 1) : j][1];
                if constexpr (OUT != P2_OUT_QKV) {
                    gp_relu2x8(v0, v1, p.relu);
                    if constexpr (OUT == P2_OUT_F32 || HAS_R) { v0 *= unfold; v1 *= unfold; }  // (no plane scale to fold the 1/2 into)
                }
                if constexpr (HAS_R) {
                    const p2_u32x4 rh = rr[b & 1][pass][0], rl = rr[b & 1][pass][1];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const p2_f32x2 a = gp_join_scaled(rh[e], rl[e]), c = gp_join_scaled(rh[2 + e], rl[2 + e]);
                        v0[2 * e] += a[0] * rsf; v0[2 * e + 1] += a[1] * rsf;
                        v1[2 * e] += c[0] * rsf; v1[2 * e + 1] += c[1] * rsf;
                    }
                }
                if (OUT == P2_OUT_PLANES && p.AC && ok) {  // (of the values in the block's units, 2^-e: back to true units once, below)
                    amx[j >> 1][0] = gp_amax3(amx[j >> 1][0], v0[0], v0[1]); amx[j >> 1][1] = gp_amax3(amx[j >> 1][1], v0[2], v0[3]);
                    amx[j >> 1][0] = gp_amax3(amx[j >> 1][0], v1[0], v1[1]); amx[j >> 1][1] = gp_amax3(amx[j >> 1][1], v1[2], v1[3]);
                }
                if (OUT == P2_OUT_F32) {
                    float* cp = c32_t + (2 * i + pass) * rstep + j * 32;
                    float* cq = cp + 4;
                    if (!ok) cp = reinterpret_cast<float*>(dummy);
                    if (!(ok && col0 + j * 32 + 4 < p.N)) cq = reinterpret_cast<float*>(dummy + 16);
                    *reinterpret_cast<p2_f32x4*>(cp) = v0;
                    *reinterpret_cast<p2_f32x4*>(cq) = v1;
                } else {
                    p2_u32x4 hi, lo;
                    if (OUT == P2_OUT_QKV) {
                        const p2_f32x4 a0 = v0 * fa, a1 = v1 * fa;  // (q: not a power of two - its own product)
                        gp_split8<false>(a0, a1, a0, a1, hi, lo);
                    } else {
                        const p2_f32x4 b0 = v0 * 2048.f, b1 = v1 * 2048.f;
                        gp_split8<true>(v0, v1, b0, b1, hi, lo);
                    }
                    uint16_t* cp = cp_t + (2 * i + pass) * rstep + j * 64;
                    if (!ok) cp = reinterpret_cast<uint16_t*>(dummy);
                    if (DBG & 64) { asm volatile("" :: "v"(hi), "v"(lo), "v"(cp)); continue; }          // measurement: no stores
                    if (DBG & 128) cp = p.Cp + ((cp - p.Cp) & ((1 << 19) - 1) & ~63ll);                   // measurement: 1 MB target
                    *reinterpret_cast<p2_u32x4*>(cp) = hi;
                    *reinterpret_cast<p2_u32x4*>(cp + 32) = lo;
                }
            }
        };
#define P2_BLK(b) std::integral_constant<int, b>{}
        stage(P2_BLK(0));
        stage(P2_BLK(1)); finish(P2_BLK(0));
        stage(P2_BLK(2)); finish(P2_BLK(1));
        stage(P2_BLK(3)); finish(P2_BLK(2));
        stage(P2_BLK(4)); finish(P2_BLK(3));
        stage(P2_BLK(5)); finish(P2_BLK(4));
        stage(P2_BLK(6)); finish(P2_BLK(5));
        stage(P2_BLK(7)); finish(P2_BLK(6));
        finish(P2_BLK(7));
#undef P2_BLK
        if (OUT == P2_OUT_PLANES && p.AC) {
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const float a = p2_wave_max(fmaxf(amx[ch][0], amx[ch][1])) * iosc[ch];
                const int cb = tn * 4 + wc * 2 + ch;
                if (lane == 0 && erow * 64 < p.M && cb < p.eld_c) p.AC[erow * p.eld_c + cb] = a;
            }
        }
    };

    // ---- pipeline.  Step g = (tile, kt) in execution order lives in LDS buffer g & 1.  A step opens with "my loads of this
    // step have landed" + ONE barrier (everybody's have; everybody is done reading the other buffer); the loads of step
    // g + 1 then go into the other buffer.  The two waves that share a SIMD (w and w + 4) take OPPOSITE orders: waves 4-7
    // issue their 8 loads first and compute after, waves 0-3 compute first and issue after (s_memtime stamps: with both
    // issuing first the matrix pipe idled ~700-1000 cycles per step).
    // Across an epilogue the load position runs TWO steps ahead: the loads of step L + 1 went out during the tile's last
    // step L as always, those of step L + 2 go out right after it (its buffer is free: the slabs live behind the tile
    // buffers) - both BEFORE the 32 stores of the epilogue.  vmcnt retires in issue order, so step L + 1 waits with
    // vmcnt(40) (8 loads of L + 2 and 32 stores may stay in flight) and step L + 2 with vmcnt(32): the store burst of the
    // epilogue - every CU of the chip reaches it at the same time - drains under two K steps of the next tile instead of
    // in front of them.  Barriers are raw s_barrier: __syncthreads() would add vmcnt(0) while LDS-direct loads are in flight.
    int ld_tile = tile, ld_kt = 0;
    bool ld_valid = true;
    auto advance = [&]() {
        if (ld_kt + 1 < nk) { ++ld_kt; return; }
        if (ld_tile + slots < t_end) {
            ld_tile += slots;
            ld_kt = 0;
            setup(ld_tile);
        } else {
            ld_valid = false;
        }
    };
    setup(tile);
    issue(0, 0, 0u);
    advance();
    const bool issue_first = (DBG & 32) ? true : wave >= 4;
    const bool overlap = nk >= 3 && !(DBG & (256 | 64 | 4 | 2));
    // tile exponents of the A operand (p2.h): the accumulators live at the exponent of the CURRENT K block; when it changes
    // they are rescaled (exact: a power of two) - a wave-uniform branch that an ordinary network never takes
    // The exponents of a tile's K blocks are fetched ONCE, by one vector load (lane i: block i of this wave's 64 rows; the
    // next tile's before the epilogue of this one), and read with v_readlane in the loop: no memory operation there.
    // Lanes 32, 33 / 34, 35 of the same load bring the exponents / max |x| of the two residual blocks this wave adds in the
    // tile's epilogue (a dependent load there would cost an L2 / HBM round trip per tile with the matrix pipe idle).
    const bool has_e = p.EA != nullptr || (HAS_R && p.ER != nullptr);
    int e_run = 0, cur_kt = 0;
    int ev = 0, ev_next = 0;
    auto fetch_e = [&](int t) {
        int v = 0;
        if (has_e) {
            const int tm_ = t / p.tiles_n, tn_ = t - tm_ * p.tiles_n;
            const int erow = tm_ * 4 + wr;
            const int nb1 = nk1 >> 1, nb = (nk + 1) >> 1;
            if (erow * 64 < p.M) {
                if (lane < nb) {
                    if (p.EA) v = lane < nb1 ? p.EA[erow * p.eld_a + lane] : (p.EA2 ? p.EA2[erow * p.eld_a2 + lane - nb1] : 0);
                } else if (HAS_R && p.ER && lane >= 32 && lane < 36) {
                    const int cb = tn_ * 4 + wc * 2 + (lane & 1);
                    if (cb < p.eld_r) {
                        if (lane < 34) v = p.ER[erow * p.eld_r + cb];
                        else if (p.AR) v = __builtin_bit_cast(int, p.AR[erow * p.eld_r + cb]);
                    }
                }
            }
        }
        return v;
    };
    ev = fetch_e(tile);
    int since = 8;       // K steps since the last epilogue
    bool ahead = false;  // the loads of the step after next were issued before that epilogue
    int buf = 0, dbg_n = 0, dbg_steps = 0;
    if (DBG & 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    }
    auto step = [&](auto FIRST) {
        long long t0 = 0, t1 = 0, t2 = 0;
        if (DBG & 8) t0 = clock64();
        if (since == 0 && ahead) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
        else if (since <= 1 && overlap) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");  // (since == 0 without `ahead`: 8 loads, then the stores)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (DBG & 8) t1 = clock64();
        if (has_e) {
            const int e_step = p.EA ? __builtin_amdgcn_readlane(ev, cur_kt >> 1) : 0;
            if (!decltype(FIRST)::value && e_step != e_run) {
                asm volatile("s_nop 15");  // (the previous step's asm MFMAs -> the VALU below: hipcc does not pad an asm's results)
                const int d = e_run - e_step;
                const float f = d < -126 ? 0.f : p2_exp2i(d);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[j][i][r] *= f;
            }
            e_run = e_step;
            // the NEXT tile's exponents: loaded beside the operand loads of K step 2, picked up at the head of step 3, right
            // behind that step's vmcnt(0) - the wait hipcc attaches to the use is then free.  (Fetched at the tile boundary the
            // wait would sit behind the epilogue's 32 stores and drain them.)
            if (nk >= 4 && tile + slots < t_end) {
                if (cur_kt == 2) ev_next = fetch_e(tile + slots);
                if (cur_kt == 3) asm volatile("" : "+v"(ev_next));
            }
        }
        ++cur_kt;
        const bool ldv = ld_valid && !(since == 0 && ahead) && !((DBG & 2) && dbg_steps >= 1);
        ++dbg_steps;
        if (!(DBG & 16) && issue_first && ldv) issue(buf ^ 1, ld_kt, 0u);
        if (DBG & 8) t2 = clock64();
        if constexpr ((DBG & ~8) == 0) compute_p(buf, FIRST);  // (8 = stamps around the real stream)
        else compute(buf, FIRST, (DBG & 16) && ldv, buf ^ 1, ld_kt);  // ONE call site: two would double the accumulator live ranges
        if (!(DBG & 16) && !issue_first && ldv) {
            unsigned dep = 0;
            if (!(DBG & 1)) asm("" : "+v"(dep) : "v"(acc[3][1]));  // scheduling-only: keeps the loads behind the MFMAs
            issue(buf ^ 1, ld_kt, dep);
        }
        if (ldv) advance();
        if (since == 0) ahead = false;
        ++since;
        if (DBG & 8) {
            const long long t3 = clock64();
            if (p.dbg && lane == 0 && dbg_n < 48 && (blockIdx.x == 0 || blockIdx.x == 101)) {
                long long* o = p.dbg + ((blockIdx.x ? 1 : 0) * 8 + wave) * 50 * 4 + dbg_n * 4;
                o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
                ++dbg_n;
            }
        }
        buf ^= 1;
    };
    for (;;) {
        step(std::true_type{});
        for (int kt = 1; kt < nk; ++kt) step(std::false_type{});
        long long e0 = 0;
        if (DBG & 8) e0 = clock64();
        if (overlap && ld_valid && !(DBG & 2)) {
            // the buffer of the step just computed is free once every wave is through it: the loads of the step after next
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            unsigned dep = 0;
            if (!(DBG & 1)) asm("" : "+v"(dep) : "v"(acc[3][1]));
            issue(buf ^ 1, ld_kt, dep);
            advance();
            ahead = true;
        }
        cur_kt = 0;
        if (has_e && nk < 4 && tile + slots < t_end) ev_next = fetch_e(tile + slots);
        if (!(DBG & 4)) epilogue(tile, e_run, ev);
        else asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]), "v"(acc[0][1]), "v"(acc[1][1]), "v"(acc[2][1]), "v"(acc[3][1]));
        ev = ev_next;
        since = overlap ? 0 : 8;
        if ((DBG & 8) && p.dbg && lane == 0 && dbg_n < 48 && (blockIdx.x == 0 || blockIdx.x == 101)) {
            long long* o = p.dbg + ((blockIdx.x ? 1 : 0) * 8 + wave) * 50 * 4 + dbg_n * 4;
            o[0] = -1; o[1] = e0; o[2] = clock64(); o[3] = 0;
            ++dbg_n;
        }
        tile += slots;
        if (tile >= t_end) break;
    }
}

int launch_gemm_p2(e2emv_ctx* ctx, const GemmP2Args& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: empty problem");
    const int K1 = a.A2 ? a.K1 : a.K;
    if (a.K % 32 || K1 % 32 || K1 <= 0 || K1 > a.K || (K1 < a.K && !a.A2))
        return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: K=%d K1=%d must be multiples of 32", a.K, K1);
    if (!a.A || !a.W || a.lda % 32 || a.lda < K1 || (a.A2 && (a.lda2 % 32 || a.lda2 < a.K - K1)))
        return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: operand planes need whole 32-column blocks (lda=%lld lda2=%lld)", (long long)a.lda, (long long)a.lda2);
    if ((uintptr_t)a.A % 16 || (a.A2 && (uintptr_t)a.A2 % 16) || (uintptr_t)a.W % 16 || (a.bias && (uintptr_t)a.bias % 16) || (a.Rp && (uintptr_t)a.Rp % 16))
        return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: operands must be 16-byte aligned");
    if (a.Rp && (a.ldr % 32 || a.ldr < a.N)) return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: residual planes need ldr %% 32 == 0");
    const int64_t a_bytes = (int64_t)a.M * a.lda * 4, a2_bytes = a.A2 ? (int64_t)a.M * a.lda2 * 4 : 16, w_bytes = (int64_t)a.N * a.K * 4;
    if (a_bytes >= ((int64_t)1 << 31) || a2_bytes >= ((int64_t)1 << 31) || w_bytes >= ((int64_t)1 << 31))
        return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: operand larger than 2 GB (M=%d lda=%lld): 32-bit byte offsets", a.M, (long long)a.lda);
    GemmP2Params p{};
    p.A = a.A; p.A2 = a.A2 ? a.A2 : a.A; p.W = a.W;
    p.a_bytes = (unsigned)a_bytes; p.a2_bytes = (unsigned)(a.A2 ? a2_bytes : a_bytes); p.w_bytes = (unsigned)w_bytes;
    p.lda_b = (unsigned)(a.lda * 4); p.lda2_b = (unsigned)((a.A2 ? a.lda2 : a.lda) * 4); p.ldw_b = (unsigned)(a.K * 4);
    p.bias = a.bias; p.Rp = a.Rp; p.ldr = a.ldr;
    p.C32 = a.C32; p.Cp = a.Cp; p.Vt = a.Vt; p.ldc = a.ldc;
    p.M = a.M; p.N = a.N; p.K = a.K; p.K1 = K1;
    p.tiles_n = (a.N + P2_BN - 1) / P2_BN;
    p.total = ((a.M + P2_BM - 1) / P2_BM) * p.tiles_n;
    p.relu = a.relu ? 1 : 0;
    p.out_scale = a.out_scale;
    p.col_scale[0] = p.col_scale[1] = p.col_scale[2] = 1.f;
    p.n_rows = a.n_rows; p.heads = a.heads;
    p.EA = a.EA; p.EA2 = a.EA2; p.ER = a.ER; p.EC = a.EC; p.EVt = a.EVt; p.AR = a.AR; p.AC = a.AC;
    p.eld_a = (int)(a.lda / 64); p.eld_a2 = (int)(a.lda2 / 64); p.eld_r = (int)(a.ldr / 64); p.eld_c = (int)(a.ldc / 64);
    p.bias_amax = a.bias ? a.bias_amax : 0.f;
    if (a.EA && (a.lda % 64 || K1 % 64 || (a.A2 && (a.lda2 % 64 || (a.K - K1) % 64)) || a.M % 64))
        return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: tile exponents need 64-column / 64-row blocks (lda=%lld K1=%d M=%d)", (long long)a.lda, K1, a.M);
    if ((a.EC || a.EVt) && (a.M % 64 || (a.out == P2_OUT_PLANES && a.ldc % 64))) return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: output exponents need 64 x 64 blocks");
    if (a.ER && (a.ldr % 64 || a.M % 64)) return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: residual exponents need 64 x 64 blocks");
    if (int rc = ensure_flags(ctx)) return rc;
    p.stats = ctx->d_flags + 2;
    const void* fn = nullptr;
    switch (a.out) {
        case P2_OUT_F32:
            if (!a.C32 || a.N % 4 || a.ldc % 4 || (uintptr_t)a.C32 % 16) return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: fp32 output needs N %% 4 == 0, ldc %% 4 == 0");
            fn = a.Rp ? reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_F32, true>) : reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_F32, false>);
            break;
        case P2_OUT_PLANES:
            if (!a.Cp || a.N % 8 || a.ldc % 32 || a.ldc < a.N || (uintptr_t)a.Cp % 16) return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: plane output needs N %% 8 == 0, ldc %% 32 == 0");
            fn = a.Rp ? reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, true>) : reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false>);
            break;
        case P2_OUT_QKV:
            if (!a.Cp || !a.Vt || a.N != 3 * P2_BN || a.heads != 4 || a.n_rows <= 0 || a.n_rows % 32 || a.M % a.n_rows || a.relu || a.Rp ||
                (uintptr_t)a.Cp % 16 || (uintptr_t)a.Vt % 16)
                return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: q|k|v output needs N = 768 (4 heads of 64), n_rows %% 32 == 0, M %% n_rows == 0");
            p.ldc = 2 * P2_BN;
            p.eld_c = 8;
            p.col_scale[0] = 0.125f * 1.4426950408889634f * P2_QS;  // log2(e) / sqrt(64), then the plane pre-scale
            p.col_scale[2] = P2_VS;
            fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_QKV, false>);
            break;
        default: return set_err(ctx, E2EMV_EINVAL, "gemm_p2: unknown output kind %d", a.out);
    }
    const int per_xcd = (p.total + 7) / 8;
    const int sl = std::min(per_xcd, std::max(1, ctx->num_cus / 8));
    const size_t lds = P2_LDSB;
    p.dbg = nullptr;
    if (!ctx->d_dummy) E2EMV_HIP(ctx, hipMalloc((void**)&ctx->d_dummy, 4096));
    p.dummy = ctx->d_dummy;
#ifdef E2EMV_STAMPS
    // measurement build only (tools/p2_stamps.py): E2EMV_P2_DBG selects an ablation / the stamped variant of the planes kernel
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("E2EMV_P2_DBG"); dbg = e ? atoi(e) : 0; }
    static long long* d_buf = nullptr;
    const size_t nb = sizeof(long long) * 2 * 8 * 50 * 4;
    if (dbg && a.out == P2_OUT_PLANES && !a.Rp) {
        switch (dbg) {
            case 1: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 1>); break;
            case 2: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 2>); break;
            case 3: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 3>); break;
            case 4: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 4>); break;
            case 6: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 6>); break;
            case 8: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 8>); break;
            case 16: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 16>); break;
            case 17: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 17>); break;
            case 24: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 24>); break;
            case 32: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 32>); break;
            case 64: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 64>); break;
            case 128: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 128>); break;
            case 256: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 256>); break;
            case 36: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 36>); break;
            case 40: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 40>); break;
            default: break;
        }
        if (dbg & 8) {
            if (!d_buf) E2EMV_HIP(ctx, hipMalloc((void**)&d_buf, nb));
            E2EMV_HIP(ctx, hipMemsetAsync(d_buf, 0, nb, s));
            p.dbg = d_buf;
        }
    }
#endif
    if (int rc = ensure_dynamic_lds(ctx, fn, lds)) return rc;
    void* args[] = {&p};
    E2EMV_HIP(ctx, hipLaunchKernel(fn, dim3(8 * sl), dim3(512), args, lds, s));
    E2EMV_CHECK_LAUNCH(ctx, "gemm_p2_kernel");
#ifdef E2EMV_STAMPS
    if (p.dbg) {
        E2EMV_HIP(ctx, hipStreamSynchronize(s));
        std::vector<long long> h(2 * 8 * 50 * 4);
        E2EMV_HIP(ctx, hipMemcpy(h.data(), d_buf, nb, hipMemcpyDeviceToHost));
        static int printed = 0;
        if (printed++ < 2)
            for (int wg = 0; wg < 2; ++wg)
                for (int w = 0; w < 8; w += 5) {
                    const long long* o = &h[((size_t)wg * 8 + w) * 50 * 4];
                    fprintf(stderr, "gemm_p2 M=%d N=%d K=%d wg %d wave %d: per K step wait+barrier | issue | compute | total   (epilogue rows: -1)\n", p.M, p.N, p.K, wg ? 101 : 0, w);
                    for (int i = 0; i < 44; ++i) {
                        const long long* t = o + i * 4;
                        if (!t[0]) break;
                        if (t[0] == -1) { fprintf(stderr, "  %2d: epilogue %lld\n", i, t[2] - t[1]); continue; }
                        fprintf(stderr, "  %2d: %5lld %5lld %5lld | %5lld\n", i, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[3] - t[0]);
                    }
                }
    }
#endif
    return E2EMV_OK;
}

}  // namespace e2emv
