// fp32 "NT" GEMM on the gfx950 matrix cores:  C[m][n] = act(scale * sum_k A[m][k] W[n][k] + bias[n]) (+ R[m][n])
//
// Every dense descriptor contraction of the matcher that is not attention goes through
// this kernel: the 1x1-Conv projections (q|k|v, merge), the propagation MLP, the keypoint
// encoder, final_proj, the conf head and the per-pair score matrix mdesc_i^T mdesc_j / sqrt(D)
// (upstream SuperGlue superglue.py: MLP / MultiHeadedAttention.proj / final_proj / einsum
// 'bdn,bdm->bnm'; the reference's matcher source is an absent submodule).
// Both operands are K-contiguous (activations [rows][channels], Conv1d weights [out][in]).
//
// Design (MI355X / CDNA4):
//  * v_mfma_f32_32x32x2_f32 - exact fp32 (bitwise an fmaf chain), 157 TFLOP/s peak; parity
//    with the fp32 reference forbids bf16/fp16 here and gfx950 has no tf32/xf32.
//  * 128x128x32 block tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles (64 acc VGPRs).
//  * LDS rows padded to 36 floats: the ds_read_b128 operand fetch (lane = row, 16 B of K) is
//    bank-conflict free (16 rows x 16 B cover the 64 banks: 36*i mod 64 distinct for 16 i).
//  * K-slot permutation: one ds_read_b128 feeds FOUR MFMAs - lanes 0-31 hold k = 8c+e,
//    lanes 32-63 hold k = 8c+4+e for MFMA e; both operands use the same map, the sum over k is
//    unchanged.  4 x b128 reads per 16 MFMAs -> LDS is idle, the kernel is MFMA-issue bound.
//  * transposed accumulators: weights are the MFMA A operand, activations the B operand, so a
//    lane owns ONE output row and its registers hold runs of 4 consecutive output channels:
//    bias / residual / store are 16-byte accesses (4x fewer epilogue instructions).
//  * persistent: 3 workgroups per CU (single 36.9 KB LDS buffer, register prefetch, two
//    barriers per K tile, 3 waves per SIMD); a workgroup walks a strided list of output tiles
//    and the K tiles of consecutive output tiles form ONE pipelined stream (the next tile's
//    first loads are issued under the last MFMAs, the epilogue stores drain under the next
//    tile) - removes the lockstep load/store bursts of one-tile-per-workgroup launches.
//  * XCD-aware tile ranges: the tiles in flight on one XCD are neighbours (shared A panels
//    stay in that XCD's L2).
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace e2emv {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

constexpr int BM = 128, BN = 128;  // BK (K tile, 32 or 64) is a template parameter; LDS rows are BK + 4 floats

struct GemmParams {
    const float* A;
    const float* A2;
    const float* W;
    const float* bias;
    const float* R;
    float* C;
    uint16_t* C3;   // optional output as bf16x3 planes, S3 [M][3][ldc3] (batch 1 only); C may then be null
    int64_t ldc3;
    uint16_t* Vt;   // optional: columns n >= vt_n0 are written TRANSPOSED as bf16x3 planes [img][3][N - vt_n0][n_rows]
    int vt_n0, n_rows;
    int q_cols;     // columns n < q_cols are multiplied by q_scale (attention query pre-scale)
    float q_scale;
    int64_t lda, lda2, ldw, ldr, ldc;
    int64_t sA, sA2, sW, sR, sC;
    int M, N, K, K1;
    int tiles_m, tiles_n, total;  // total = batch * tiles_m * tiles_n
    float scale;
    int relu;
    int vec_store;  // 1: N % 4 == 0 and C/R rows 16-byte aligned -> dwordx4 epilogue
    // CONV (implicit GEMM of a 3x3 / pad 1 / stride 1 convolution over NHWC activations): row m = pixel (img, y, x),
    // K = 9 * conv_c ordered (ky, kx, cin); A = the NHWC input, taps outside the image read as zero
    int conv_h, conv_w, conv_c;
    // conv_pool: rows are enumerated quad-major (4 consecutive rows = one 2x2 block of pixels) and the epilogue writes
    // their maximum: the 2x2 / stride-2 max-pool that follows conv1b / conv2b / conv3b never touches memory
    int conv_pool;
};

// Persistent kernel: gridDim.x = 8 * slots workgroups (2 per CU); workgroup (xcd = id & 7,
// slot = id >> 3) walks tiles xcd*per_xcd + slot, + slots, ... of its XCD's contiguous tile range,
// so the tiles in flight on one XCD are neighbours (shared A panels stay in that L2).  The K
// tiles of consecutive output tiles form ONE software-pipelined stream: the first K tile of
// the next output tile is prefetched under the last MFMAs of the current one and the
// epilogue's stores drain under the next tile's MFMAs - no lockstep load/store bursts.
// EXT = true adds the bf16x3-plane outputs (C3, V^T with swapped operand roles, q pre-scale) used by the
// q|k|v GEMM of the split-operand attention path; it gets a 256-VGPR budget (2 workgroups/CU) so that the
// plain kernel (EXT = false, every other GEMM) keeps its spill-free 168-VGPR / 3-workgroups-per-CU build.
// TN = 2: 128 x 128 output tile, waves 2 x 2 (the default).  TN = 1: 256 x 64 tile, waves 4 x 1 - for outputs only 64
// channels wide (SuperPoint's conv1b / conv2a / conv2b), where half of a 128-wide tile would multiply padding.
template <bool EXT, int BK, int DBG = 0, bool CONV = false, int TN = 2>
// TN = 0: 64 x 64 tile, waves 2 x 2 of ONE 32 x 32 MFMA tile each - the latency shape for small problems (batch 1-4 of the
// reference's eval loop): a tile's K loop is 4x shorter and a 2048-row GEMM fills 256 workgroups instead of 64.
__global__ __launch_bounds__(256, (EXT || BK == 64 || TN == 1) ? 2 : 3) void gemm_nt_kernel(GemmParams p) {
    constexpr int BM = TN == 2 ? 128 : (TN == 1 ? 256 : 64), BN = TN == 2 ? 128 : 64;  // shadow the namespace-scope defaults
    constexpr int WT = TN == 0 ? 1 : 2;   // 32 x 32 MFMA tiles per wave and dimension
    constexpr int WS = 32 * WT;           // rows / columns of the output tile one wave owns
    constexpr int LDK = BK + 4;           // 36: 36*i mod 64, 68: 4*i mod 64 - both give 16 distinct 16-byte slots
    constexpr int CPR = BK / 4;           // 16-byte chunks per tile row
    constexpr int NCH = BM * CPR / 256;   // chunks per thread of the activation tile (4 or 8)
    constexpr int NCHB = BN * CPR / 256;  // chunks per thread of the weight tile
    constexpr int RSTEP = 256 / CPR;      // rows covered by one pass of the 256 threads (32 or 16)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [BM][BK + 4]  activations
    float* Bs = smem + BM * (BK + 4);       // [BN][BK + 4]  weights

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (p.total + 7) / 8;
    const int t_begin = xcd * per_xcd;
    const int t_end = min(t_begin + per_xcd, p.total);
    int tile = t_begin + slot;
    if (tile >= t_end) return;
    const int tiles_mn = p.tiles_m * p.tiles_n;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = TN == 1 ? wave : wave >> 1, wc = TN == 1 ? 0 : (wave & 1);
    const int l31 = lane & 31, lh = lane >> 5;
    const int ld_row = tid / CPR;           // (+RSTEP*i)
    const int ld_c4 = (tid % CPR) * 4;      // float offset inside the K tile
    const int nk = p.K / BK;

    // operand pointers of the tile whose K tiles are being PREFETCHED
    const float* a_ptr[NCH];
    const float* a2_ptr[NCH];
    const float* w_ptr[NCHB];
    int cyx[CONV ? NCH : 1];  // CONV: pixel coordinates (y << 16 | x) of this thread's A rows
    auto setup = [&](int t) {
        const int z = t / tiles_mn, r = t - z * tiles_mn;
        const int tm = r / p.tiles_n, tn = r - tm * p.tiles_n;
        const float* A = p.A + z * p.sA;
        const float* A2 = p.A2 ? p.A2 + z * p.sA2 : nullptr;
        const float* W = p.W + z * p.sW;
#pragma unroll
        for (int i = 0; i < NCHB; ++i) {
            const int rw = min(tn * BN + ld_row + RSTEP * i, p.N - 1);
            w_ptr[i] = W + (int64_t)rw * p.ldw + ld_c4;
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int ra = min(tm * BM + ld_row + RSTEP * i, p.M - 1);
            a_ptr[i] = A + (int64_t)ra * p.lda + ld_c4;
            if (CONV) {
                if (p.conv_pool) {
                    const int q = ra >> 2, sub = ra & 3, wq = p.conv_w >> 1, quads = (p.conv_h >> 1) * wq;
                    const int img = q / quads, rem = q - img * quads, qy = rem / wq;
                    const int yy = 2 * qy + (sub >> 1), xx = 2 * (rem - qy * wq) + (sub & 1);
                    cyx[i] = (yy << 16) | xx;
                    a_ptr[i] = A + ((int64_t)(img * p.conv_h + yy) * p.conv_w + xx) * p.lda + ld_c4;
                } else {
                    const int rem = ra % (p.conv_h * p.conv_w);
                    const int yy = rem / p.conv_w;
                    cyx[i] = (yy << 16) | (rem - yy * p.conv_w);
                }
            }
            a2_ptr[i] = A2 ? A2 + (int64_t)ra * p.lda2 + ld_c4 : nullptr;
        }
    };

    f32x4 ra[NCH], rb[NCHB];
    auto gload = [&](int kt) {
        const int k = kt * BK;
        if (CONV) {
            const int tap = k / p.conv_c, ci = k - tap * p.conv_c;  // a K tile never straddles two taps (conv_c % BK == 0)
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const int off = (dy * p.conv_w + dx) * p.conv_c + ci;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const bool ok = (unsigned)((cyx[i] >> 16) + dy) < (unsigned)p.conv_h && (unsigned)((cyx[i] & 0xFFFF) + dx) < (unsigned)p.conv_w;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = *reinterpret_cast<const f32x4*>(a_ptr[i] + off);
                ra[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const float* src = (k < p.K1) ? a_ptr[i] + k : a2_ptr[i] + (k - p.K1);
                ra[i] = *reinterpret_cast<const f32x4*>(src);
            }
        }
#pragma unroll
        for (int i = 0; i < NCHB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(w_ptr[i] + k);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) *reinterpret_cast<f32x4*>(&As[(buf * BM + ld_row + RSTEP * i) * LDK + ld_c4]) = ra[i];
#pragma unroll
        for (int i = 0; i < NCHB; ++i) *reinterpret_cast<f32x4*>(&Bs[(buf * BN + ld_row + RSTEP * i) * LDK + ld_c4]) = rb[i];
    };

    // acc[j][i]: weights tile j (MFMA A operand, rows -> registers) x activation tile i (B operand,
    // rows -> lanes).  C/D layout: lane = activation row m, registers = 4-runs of output channels n.
    f32x16 acc[WT][WT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < WT; ++j)
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    };
    // SWAP = false: weights are the MFMA A operand (rows -> registers), activations B (rows -> lanes);
    // SWAP = true : roles exchanged (lane = output channel, registers = runs of 4 rows) for V^T tiles
    auto compute = [&](auto swap_tag) {
        constexpr bool SWAP = decltype(swap_tag)::value;
        const float* as = &As[(wr * WS + l31) * LDK + lh * 4];
        const float* bs = &Bs[(wc * WS + l31) * LDK + lh * 4];
        if (DBG & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(as + c * 8);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(bs + c * 8);
            if constexpr (WT == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[0][0] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(x0[e], w0[e], acc[0][0], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], x0[e], acc[0][0], 0, 0, 0);
            } else {
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(as + 32 * LDK + c * 8);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(bs + 32 * LDK + c * 8);
            if (DBG & 2) {  // profiling: operand pipeline only (global -> LDS -> registers), no matrix-core work
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0][0][e] += x0[e] + x1[e] + w0[e] + w1[e];
                continue;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (SWAP) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[e], w0[e], acc[0][0], 0, 0, 0);
                    acc[0][WT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[e], w0[e], acc[0][WT - 1], 0, 0, 0);
                    acc[WT - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[e], w1[e], acc[WT - 1][0], 0, 0, 0);
                    acc[WT - 1][WT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[e], w1[e], acc[WT - 1][WT - 1], 0, 0, 0);
                } else {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], x0[e], acc[0][0], 0, 0, 0);
                    acc[0][WT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[e], x1[e], acc[0][WT - 1], 0, 0, 0);
                    acc[WT - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[e], x0[e], acc[WT - 1][0], 0, 0, 0);
                    acc[WT - 1][WT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[e], x1[e], acc[WT - 1][WT - 1], 0, 0, 0);
                }
            }
            }
        }
        if (DBG & 1) __builtin_amdgcn_s_setprio(0);
    };
    auto split4 = [&](const f32x4& v, bf16x4& h0, bf16x4& h1, bf16x4& h2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const __bf16 a = (__bf16)v[e];
            const float r1 = v[e] - (float)a;
            const __bf16 b2 = (__bf16)r1;
            h0[e] = a; h1[e] = b2; h2[e] = (__bf16)(r1 - (float)b2);
        }
    };
    // V^T tiles (SWAP accumulators): lane = channel n, registers = 4 consecutive rows m -> 8-byte stores along the keys
    auto epilogue_vt = [&](int t) {
        const int r = t % tiles_mn;
        const int tm = r / p.tiles_n, tn = r - tm * p.tiles_n;
        const int nv = p.N - p.vt_n0;
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            const int n = tn * BN + wc * WS + j * 32 + l31;
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m = tm * BM + wr * WS + i * 32 + 8 * g + 4 * lh;
                    if (m >= p.M) continue;
                    const int img = m / p.n_rows, ml = m - img * p.n_rows;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[j][i][4 * g + e] * p.scale + bv;
                    bf16x4 h0, h1, h2;
                    split4(v, h0, h1, h2);
                    uint16_t* dst = p.Vt + ((int64_t)(img * 3) * nv + (n - p.vt_n0)) * p.n_rows + ml;
                    *reinterpret_cast<bf16x4*>(dst) = h0;
                    *reinterpret_cast<bf16x4*>(dst + (int64_t)nv * p.n_rows) = h1;
                    *reinterpret_cast<bf16x4*>(dst + 2 * (int64_t)nv * p.n_rows) = h2;
                }
        }
    };
    auto epilogue = [&](int t) {
        const int z = t / tiles_mn, r = t - z * tiles_mn;
        const int tm = r / p.tiles_n, tn = r - tm * p.tiles_n;
        float* C = p.C ? p.C + z * p.sC : nullptr;
        const float* R = p.R ? p.R + z * p.sR : nullptr;
#pragma unroll
        for (int i = 0; i < WT; ++i) {
            const int m = tm * BM + wr * WS + i * 32 + l31;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < WT; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = tn * BN + wc * WS + j * 32 + 8 * g + 4 * lh;
                    if (n >= p.N) continue;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[j][i][4 * g + e] * p.scale;
                    if (p.vec_store) {
                        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                        if (p.relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = relu_nan(v[e]);
                        }
                        if (R) v += *reinterpret_cast<const f32x4*>(R + (int64_t)m * p.ldr + n);
                        if (EXT && n < p.q_cols) v *= p.q_scale;
                        if (CONV && p.conv_pool) {  // lanes 4q..4q+3 hold the 2x2 block of pooled pixel m/4
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[e] = fmaxf(v[e], __shfl_xor(v[e], 1));
                                v[e] = fmaxf(v[e], __shfl_xor(v[e], 2));
                            }
                            if ((lane & 3) == 0) *reinterpret_cast<f32x4*>(C + (int64_t)(m >> 2) * p.ldc + n) = v;
                            continue;
                        }
                        if (!EXT || p.C) *reinterpret_cast<f32x4*>(C + (int64_t)m * p.ldc + n) = v;
                        if (EXT && p.C3) {  // bf16x3 planes for the split-operand attention (attention3.hip)
                            bf16x4 h0, h1, h2;
                            split4(v, h0, h1, h2);
                            uint16_t* d3 = p.C3 + (int64_t)m * 3 * p.ldc3 + n;
                            *reinterpret_cast<bf16x4*>(d3) = h0;
                            *reinterpret_cast<bf16x4*>(d3 + p.ldc3) = h1;
                            *reinterpret_cast<bf16x4*>(d3 + 2 * p.ldc3) = h2;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (n + e < p.N) {
                                float x = v[e] + (p.bias ? p.bias[n + e] : 0.f);
                                if (p.relu) x = relu_nan(x);
                                if (R) x += R[(int64_t)m * p.ldr + n + e];
                                C[(int64_t)m * p.ldc + n + e] = x;
                            }
                        }
                    }
                }
            }
        }
    };

    // Single LDS buffer, register prefetch, two barriers per K tile; 36.9 KB of LDS lets 3 workgroups
    // share a CU (3 waves per SIMD): while one wave sits at a barrier or in its epilogue the other two
    // keep the SIMD's MFMA pipe busy (2 waves/SIMD measured 81 % MFMA-busy in the main loop).
    zero_acc();
    setup(tile);
    gload(0);
    for (;;) {
        const bool vt = EXT && ((tile % tiles_mn) % p.tiles_n) * BN >= p.vt_n0;  // workgroup-uniform
        for (int kt = 0; kt + 1 < nk; ++kt) {
            __syncthreads();  // previous K tile fully consumed
            lstore(0);
            __syncthreads();
            gload(kt + 1);
            // NB: hipcc sinks these global loads below the MFMA block (it re-uses their registers for
            // the LDS fragments).  Pinning them above with sched_barrier(0) was measured 8-13 % SLOWER
            // (K=2048: 563 -> 645 us): with 3 waves/SIMD the latency is covered by the other waves and
            // the early loads only lengthen the live ranges.  Left to the compiler on purpose.
            if (EXT && vt) compute(std::true_type{}); else compute(std::false_type{});
        }
        // last K tile of this output tile: prefetch the next output tile's first K tile under it
        const int next = tile + slots;
        const bool more = next < t_end;
        __syncthreads();
        lstore(0);
        __syncthreads();
        if (more) {
            setup(next);
            gload(0);
        }
        if (EXT && vt) compute(std::true_type{}); else compute(std::false_type{});
        if (EXT && vt) epilogue_vt(tile); else epilogue(tile);
        if (!more) break;
        zero_acc();
        tile = next;
    }
}

int launch_gemm_nt(e2emv_ctx* ctx, const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.batch <= 0) return set_err(ctx, E2EMV_ESHAPE, "gemm: empty problem");
    const int K1 = a.A2 ? a.K1 : a.K;
    if (a.K % 32 || K1 % 32 || K1 > a.K || (K1 < a.K && !a.A2))
        return set_err(ctx, E2EMV_ESHAPE, "gemm: K=%d K1=%d must be multiples of 32", a.K, K1);
    GemmParams p;
    p.A = a.A; p.A2 = a.A2; p.W = a.W; p.bias = a.bias; p.R = a.R; p.C = a.C;
    p.C3 = a.C3; p.ldc3 = a.ldc3;
    p.Vt = a.Vt; p.vt_n0 = a.Vt ? a.vt_n0 : (1 << 30); p.n_rows = a.n_rows > 0 ? a.n_rows : a.M;
    p.q_cols = a.q_cols; p.q_scale = a.q_scale;
    if (!a.C && !a.C3) return set_err(ctx, E2EMV_EINVAL, "gemm: no output");
    if (a.Vt && (a.vt_n0 % BN || p.n_rows % 128 || a.M % p.n_rows || a.batch != 1))
        return set_err(ctx, E2EMV_ESHAPE, "gemm: V^T output needs vt_n0 %% 128 == 0, whole images and batch 1");
    p.lda = a.lda; p.lda2 = a.lda2; p.ldw = a.ldw; p.ldr = a.ldr; p.ldc = a.ldc;
    p.sA = a.sA; p.sA2 = a.sA2; p.sW = a.sW; p.sR = a.sR; p.sC = a.sC;
    p.M = a.M; p.N = a.N; p.K = a.K; p.K1 = K1;
    p.conv_h = a.conv_h; p.conv_w = a.conv_w; p.conv_c = a.conv_c; p.conv_pool = a.conv_pool ? 1 : 0;
    const bool conv = a.conv_c > 0;
    if (conv && (a.conv_h > 32767 || a.conv_w > 32767 || a.conv_c % 32 || a.K != 9 * a.conv_c || a.A2 || a.batch != 1 || a.conv_h <= 0 || a.conv_w <= 0 ||
                 a.M % (a.conv_h * a.conv_w) || a.lda != a.conv_c || a.C3 || a.Vt || a.q_cols > 0))
        return set_err(ctx, E2EMV_ESHAPE, "gemm: conv mode needs NHWC input with C %% 32 == 0, K = 9 C, whole images, batch 1");
    if (a.conv_pool && (!conv || a.conv_h % 2 || a.conv_w % 2 || a.N % 4 || a.R))
        return set_err(ctx, E2EMV_ESHAPE, "gemm: fused 2x2 max-pool needs conv mode, even image sides and N %% 4 == 0");
    // 64-channel-wide outputs: the 256 x 64 tile variant (conv mode only - the matcher never has N <= 64 at scale)
    const bool narrow = a.conv_c > 0 && a.N <= 64;
    // latency shape: when 128 x 128 tiles would leave most CUs idle (batch 1-4 of the reference's eval loop), 64 x 64 tiles
    // make 4x more, 4x shorter work items (E2EMV_GEMM_SMALL=0 disables)
    static int small_env = -1;
    if (small_env < 0) small_env = dbg_knob("E2EMV_GEMM_SMALL", 1);
    const bool plain = a.conv_c == 0 && !a.C3 && !a.Vt && a.q_cols == 0;
    const int64_t tiles128 = (int64_t)a.batch * ((a.M + 127) / 128) * ((a.N + 127) / 128);
    const bool small = small_env && plain && tiles128 * 2 <= ctx->num_cus;
    const int bm = narrow ? 256 : (small ? 64 : BM), bn = (narrow || small) ? 64 : BN;
    p.tiles_m = (a.M + bm - 1) / bm;
    p.tiles_n = (a.N + bn - 1) / bn;
    p.total = p.tiles_m * p.tiles_n * a.batch;
    p.scale = a.scale;
    p.relu = a.relu ? 1 : 0;
    p.vec_store = (a.N % 4 == 0) && (!a.C || ((a.ldc % 4 == 0) && ((uintptr_t)a.C % 16 == 0) && (a.sC % 4 == 0))) &&
                  (!a.bias || (uintptr_t)a.bias % 16 == 0) &&
                  (!a.R || ((a.ldr % 4 == 0) && ((uintptr_t)a.R % 16 == 0) && (a.sR % 4 == 0)));
    if ((p.C3 || a.Vt) && (!p.vec_store || a.batch != 1 || (p.C3 && a.ldc3 % 4)))
        return set_err(ctx, E2EMV_ESHAPE, "gemm: the bf16x3 side output needs the vector epilogue and batch 1");
    const int per_xcd = (p.total + 7) / 8;
    static int dbg = -1, wg = -1, bk_env = -1;  // profiling knobs: E2EMV_GEMM_DEBUG (bit0: s_setprio), _WG_PER_CU, _BK
    if (dbg < 0) dbg = dbg_knob("E2EMV_GEMM_DEBUG", 0);
    if (wg < 0) wg = dbg_knob("E2EMV_GEMM_WG_PER_CU", 0);
    if (bk_env < 0) bk_env = dbg_knob("E2EMV_GEMM_BK", 0);
    const bool ext = p.C3 || a.Vt || a.q_cols > 0;
    // K tile: 32.  A 64-deep tile (half the barrier / staging episodes per MFMA, 2 workgroups per CU) was measured
    // 1-5 % SLOWER at every shape; so were 1 or 2 workgroups per CU and s_setprio around the MFMA block: the main
    // loop sits at ~125 TFLOP/s (MFMA pipe ~82 % busy at ~2.3 GHz) whatever the schedule.  E2EMV_GEMM_BK=64 keeps
    // the variant reachable for profiling.
    int bk = 32;
    if (bk_env == 64 && a.K % 64 == 0 && K1 % 64 == 0 && !ext && !conv) bk = 64;
    const int per_cu = wg > 0 ? wg : ((ext || bk == 64 || narrow) ? 2 : 3);
    const int sl = std::min(per_xcd, std::max(1, ctx->num_cus * per_cu / 8));
    const size_t lds = sizeof(float) * (bm + bn) * (bk + 4);
    if (small) {
        hipLaunchKernelGGL((gemm_nt_kernel<false, 32, 0, false, 0>), dim3(8 * sl), dim3(256), lds, s, p);
    } else if (conv && narrow) {
        hipLaunchKernelGGL((gemm_nt_kernel<false, 32, 0, true, 1>), dim3(8 * sl), dim3(256), lds, s, p);
    } else if (conv) {
        hipLaunchKernelGGL((gemm_nt_kernel<false, 32, 0, true>), dim3(8 * sl), dim3(256), lds, s, p);
    } else if (bk == 64) {
        if (int rc = ensure_dynamic_lds(ctx, (const void*)gemm_nt_kernel<false, 64, 0>, lds)) return rc;
        hipLaunchKernelGGL((gemm_nt_kernel<false, 64, 0>), dim3(8 * sl), dim3(256), lds, s, p);
    } else if (ext) {
        hipLaunchKernelGGL((gemm_nt_kernel<true, 32, 0>), dim3(8 * sl), dim3(256), lds, s, p);
    } else if (dbg & 2) {
        hipLaunchKernelGGL((gemm_nt_kernel<false, 32, 2>), dim3(8 * sl), dim3(256), lds, s, p);
    } else if (dbg & 1) {
        hipLaunchKernelGGL((gemm_nt_kernel<false, 32, 1>), dim3(8 * sl), dim3(256), lds, s, p);
    } else {
        hipLaunchKernelGGL((gemm_nt_kernel<false, 32, 0>), dim3(8 * sl), dim3(256), lds, s, p);
    }
    E2EMV_CHECK_LAUNCH(ctx, "gemm_nt_kernel");
    return E2EMV_OK;
}

}  // namespace e2emv

extern "C" int e2emv_gemm_nt(e2emv_ctx* ctx, int batch, int M, int Nout, int K, int K1, const float* d_A, int64_t lda,
                             int64_t strideA, const float* d_A2, int64_t lda2, int64_t strideA2, const float* d_W,
                             int64_t ldw, int64_t strideW, const float* d_bias, const float* d_R, int64_t ldr,
                             int64_t strideR, float* d_C, int64_t ldc, int64_t strideC, float scale, int flags,
                             void* stream) {
    if (!ctx || !d_A || !d_W || !d_C) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    e2emv::GemmArgs a;
    a.batch = batch; a.M = M; a.N = Nout; a.K = K; a.K1 = K1;
    a.A = d_A; a.lda = lda; a.sA = strideA;
    a.A2 = d_A2; a.lda2 = lda2; a.sA2 = strideA2;
    a.W = d_W; a.ldw = ldw; a.sW = strideW;
    a.bias = d_bias; a.R = d_R; a.ldr = ldr; a.sR = strideR;
    a.C = d_C; a.ldc = ldc; a.sC = strideC;
    a.scale = scale; a.relu = (flags & 1) != 0;
    e2emv::prof_begin(ctx, e2emv::PS_GEMM, (hipStream_t)stream);
    int rc = e2emv::launch_gemm_nt(ctx, a, (hipStream_t)stream);
    e2emv::prof_end(ctx, (hipStream_t)stream);
    return rc;
}
