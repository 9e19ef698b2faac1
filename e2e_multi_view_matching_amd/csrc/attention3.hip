// Fused multi-head attention on the bf16 matrix pipe with fp32-class accuracy (operand splitting,
// see gemm3.hip): every contraction is the 6-term product of 3-way bf16 splits accumulated in fp32.
//
// Same transposed, lane-owns-a-query structure as attention.hip:
//   S^T[key][q]  = sum over 6 (Ki, Qj) plane pairs of mfma_32x32x16_bf16(A = K rows, B = Q rows)
//   O^T[d][q]   += sum over 6 (Vi, Pj) plane pairs of mfma(A = V^T rows, B = P^T)
// Q|K arrive as S3 planes [row][3][2D] (q pre-scaled by log2(e)/sqrt(d) in the producing GEMM),
// V arrives TRANSPOSED, [img][3][D][n_rows] (gemm3's V^T epilogue), because the MFMA operand of
// the P.V product needs 8 consecutive KEYS per lane.  The softmax probabilities are split into
// three bf16 planes in registers; with the S^T accumulator layout (row = (r&3)+8(r>>2)+4(lane>>5))
// registers 8u..8u+7 of a lane are exactly the 8 K-slots that lane must supply for the u-th
// 16-key MFMA, provided the V^T tile is stored in LDS with the matching key order inside each
// 16-key group (pos = (k&3) + 4*((k>>3)&1) + 8*((k>>2)&1)) - so P never leaves its lane.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace e2emv {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int A3_Q = 128, A3_KV = 64, A3_HD = 64;
constexpr int A3_LD = 72;                 // LDS row: 64 bf16 + 8 pad = 144 B (36 dwords: conflict-free b128 reads)
constexpr int A3_PLANE = 64 * A3_LD;      // bf16 elements per LDS plane tile

struct Attn3Params {
    const uint16_t* qk;   // S3 [n_img*n_rows][3][2D]
    const uint16_t* vt;   // [n_img][3][D][n_rows]
    uint16_t* out;        // S3 [n_img*n_rows][3][D] or null
    float* out32;         // fp32 [n_img*n_rows][D] or null
    int B, T, n_rows, D, H, cross;
    int nv[E2EMV_MAX_TUPLE];  // valid keypoints (queries and keys) of image t of a tuple
    int nq, groups, gper;
};

__device__ __forceinline__ void split3f(float v, __bf16& a, __bf16& b, __bf16& c) {
    a = (__bf16)v;
    const float r1 = v - (float)a;
    b = (__bf16)r1;
    const float r2 = r1 - (float)b;
    c = (__bf16)r2;
}

__global__ __launch_bounds__(256, 2) void attention3_kernel(Attn3Params p) {
    __shared__ __attribute__((aligned(16))) uint16_t Ks[3 * A3_PLANE];
    __shared__ __attribute__((aligned(16))) uint16_t Vs[3 * A3_PLANE];

    const int lin = blockIdx.x;
    const int xcd = lin & 7, idx = lin >> 3;
    const int g = xcd * p.gper + idx / p.nq;
    if (g >= p.groups) return;
    const int qt = idx % p.nq;
    const int img = g / p.H, head = g % p.H;
    const int b = img / p.T, t = img % p.T;
    if (qt * A3_Q >= p.nv[t]) return;  // shorter image of a ragged tuple: no queries in this tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int64_t qk_ld = 2 * (int64_t)p.D;        // plane width of the q|k matrix
    const int64_t qk_row = 3 * qk_ld;              // bf16 per row

    // ---- Q fragments (B operand): lane (q, lh) holds Q_pl[q][16 s + 8 lh .. +7]
    const int q_row = qt * A3_Q + wave * 32 + l31;
    bf16x8 Qf[3][4];
    {
        const uint16_t* qp = p.qk + ((int64_t)img * p.n_rows + q_row) * qk_row + head * A3_HD + lh * 8;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int s = 0; s < 4; ++s) Qf[pl][s] = *reinterpret_cast<const bf16x8*>(qp + pl * qk_ld + s * 16);
    }

    f32x16 O0, O1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { O0[r] = 0.f; O1[r] = 0.f; }
    float m_run = -1e30f, l_run = 0.f;

    const int n_src = p.cross ? p.T - 1 : 1;
    auto src_t = [&](int si) { return !p.cross ? t : (si < t ? si : si + 1); };
    int n_tiles = 0;
    for (int si = 0; si < n_src; ++si) n_tiles += (p.nv[src_t(si)] + A3_KV - 1) / A3_KV;
    // linear key-tile index -> (source image of the tuple, tile inside it); sources may differ in length
    auto locate = [&](int tile, int& tt, int& kt) {
        int si = 0;
        for (;; ++si) {
            const int n = (p.nv[src_t(si)] + A3_KV - 1) / A3_KV;
            if (tile < n || si + 1 == n_src) break;
            tile -= n;
        }
        tt = src_t(si);
        kt = tile;
    };

    // staging: 6 x 16-byte chunks of K and of V^T per thread: chunk i -> plane i>>1, row tid/8 + 32*(i&1), chunk tid&7
    const int st_row = tid >> 3, st_ch = tid & 7;
    u32x4 rk[6], rv[6];
    auto gload = [&](int tile) {
        int tt, kt;
        locate(tile, tt, kt);
        const int simg = b * p.T + tt;
        const uint16_t* kbase = p.qk + ((int64_t)simg * p.n_rows + kt * A3_KV) * qk_row + p.D + head * A3_HD + st_ch * 8;
        const uint16_t* vbase = p.vt + ((int64_t)simg * 3 * p.D + head * A3_HD) * p.n_rows + kt * A3_KV + st_ch * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int pl = i >> 1, row = st_row + 32 * (i & 1);
            rk[i] = *reinterpret_cast<const u32x4*>(kbase + (int64_t)row * qk_row + pl * qk_ld);
            rv[i] = *reinterpret_cast<const u32x4*>(vbase + ((int64_t)pl * p.D + row) * p.n_rows);
        }
    };
    // V^T LDS position of this thread's two 4-key halves (key order permuted inside 16-key groups)
    const int v_grp = st_ch >> 1, v_pos = 4 * (st_ch & 1);
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int pl = i >> 1, row = st_row + 32 * (i & 1);
            *reinterpret_cast<u32x4*>(&Ks[pl * A3_PLANE + row * A3_LD + st_ch * 8]) = rk[i];
            uint16_t* vd = &Vs[pl * A3_PLANE + row * A3_LD + 16 * v_grp + v_pos];
            *reinterpret_cast<u32x2*>(vd) = u32x2{rv[i][0], rv[i][1]};
            *reinterpret_cast<u32x2*>(vd + 8) = u32x2{rv[i][2], rv[i][3]};
        }
    };

    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};  // plane of the A operand (K or V^T), smallest terms first
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};  // plane of the B operand (Q or P)

    gload(0);
    for (int tile = 0; tile < n_tiles; ++tile) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (tile + 1 < n_tiles) gload(tile + 1);

        int tt_cur, kt;
        locate(tile, tt_cur, kt);
        const int valid_in_tile = p.nv[tt_cur] - kt * A3_KV;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (sub * 32 >= valid_in_tile) break;
            f32x16 S;
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] = 0.f;
            const uint16_t* kp = &Ks[(sub * 32 + l31) * A3_LD + lh * 8];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 kf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) kf[pl] = *reinterpret_cast<const bf16x8*>(kp + pl * A3_PLANE + s * 16);
#pragma unroll
                for (int q = 0; q < 6; ++q) S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[PA[q]], Qf[PB[q]][s], S, 0, 0, 0);
            }
            if (valid_in_tile < sub * 32 + 32) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= valid_in_tile) S[r] = -INFINITY;
                }
            }
            float mx = S[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float ps = 0.f;
            bf16x8 Pf[3][2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(S[r] - m_new);
                ps += pv;
                __bf16 a, bb, c;
                split3f(pv, a, bb, c);
                Pf[0][r >> 3][r & 7] = a;
                Pf[1][r >> 3][r & 7] = bb;
                Pf[2][r >> 3][r & 7] = c;
            }
            l_run = l_run * alpha + ps;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) { O0[r] *= alpha; O1[r] *= alpha; }
            // O^T[d][q] += V^T[d][keys] P^T[keys][q]; 16-key group index inside the tile = 2*sub + u
            const uint16_t* vp = &Vs[l31 * A3_LD + lh * 8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bf16x8 v0[3], v1[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    v0[pl] = *reinterpret_cast<const bf16x8*>(vp + pl * A3_PLANE + 16 * (2 * sub + u));
                    v1[pl] = *reinterpret_cast<const bf16x8*>(vp + pl * A3_PLANE + 32 * A3_LD + 16 * (2 * sub + u));
                }
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    O0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0[PA[q]], Pf[PB[q]][u], O0, 0, 0, 0);
                    O1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1[PA[q]], Pf[PB[q]][u], O1, 0, 0, 0);
                }
            }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / l_tot;
    if (p.out32) {
        float* op = p.out32 + ((int64_t)img * p.n_rows + q_row) * p.D + head * A3_HD + 4 * lh;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            f32x4 a, c;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = O0[gq * 4 + e] * inv; c[e] = O1[gq * 4 + e] * inv; }
            *reinterpret_cast<f32x4*>(op + 8 * gq) = a;
            *reinterpret_cast<f32x4*>(op + 32 + 8 * gq) = c;
        }
    }
    if (p.out) {
        uint16_t* op = p.out + ((int64_t)img * p.n_rows + q_row) * 3 * p.D + head * A3_HD + 4 * lh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                bf16x4 h0, h1, h2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = (dt ? O1[gq * 4 + e] : O0[gq * 4 + e]) * inv;
                    __bf16 a, bb, c;
                    split3f(v, a, bb, c);
                    h0[e] = a; h1[e] = bb; h2[e] = c;
                }
                uint16_t* dst = op + 32 * dt + 8 * gq;
                *reinterpret_cast<bf16x4*>(dst) = h0;
                *reinterpret_cast<bf16x4*>(dst + p.D) = h1;
                *reinterpret_cast<bf16x4*>(dst + 2 * p.D) = h2;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Same attention, fed with the fp32 q|k|v matrix of the projection GEMM ([n_img*n_rows][3D], the fp32 kernel's layout):
// the three bf16 planes of Q (registers), K and V^T (LDS) are produced HERE, on the way in, instead of by the producing
// GEMM.  The q|k|v GEMM then is an ordinary split-operand GEMM with a 4-byte-per-element output (201 MB per launch at
// config 2) where the plane-emitting epilogue wrote 644 MB and ran on the fp32 matrix pipe; the price is that the
// query tiles of one (image, head) each split the same K / V tiles again (VALU work next to a matrix-bound loop).
// Numerics are those of attention3_kernel: q is scaled by log2(e)/sqrt(d) in fp32 before it is split.
struct Attn3fParams {
    const float* qkv;     // [n_img*n_rows][3D]  q | k | v, head-major channels
    float* out32;         // [n_img*n_rows][D]
    int B, T, n_rows, D, H, cross;
    int nv[E2EMV_MAX_TUPLE];
    int nq, groups, gper;
    float q_scale;
};

__global__ __launch_bounds__(256, 2) void attention3f_kernel(Attn3fParams p) {
    __shared__ __attribute__((aligned(16))) uint16_t Ks[3 * A3_PLANE];
    __shared__ __attribute__((aligned(16))) uint16_t Vs[3 * A3_PLANE];

    const int lin = blockIdx.x;
    const int xcd = lin & 7, idx = lin >> 3;
    const int g = xcd * p.gper + idx / p.nq;
    if (g >= p.groups) return;
    const int qt = idx % p.nq;
    const int img = g / p.H, head = g % p.H;
    const int b = img / p.T, t = img % p.T;
    if (qt * A3_Q >= p.nv[t]) return;  // shorter image of a ragged tuple: no queries in this tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int64_t ld = 3 * (int64_t)p.D;  // floats per q|k|v row

    // ---- Q fragments (B operand): lane (q, lh) holds Q_pl[q][16 s + 8 lh .. +7]
    const int q_row = qt * A3_Q + wave * 32 + l31;
    bf16x8 Qf[3][4];
    {
        const float* qp = p.qkv + ((int64_t)img * p.n_rows + q_row) * ld + head * A3_HD + lh * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(qp + s * 16), hi = *reinterpret_cast<const f32x4*>(qp + s * 16 + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                __bf16 a, bb, c;
                split3f((e < 4 ? lo[e] : hi[e - 4]) * p.q_scale, a, bb, c);
                Qf[0][s][e] = a; Qf[1][s][e] = bb; Qf[2][s][e] = c;
            }
        }
    }

    f32x16 O0, O1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { O0[r] = 0.f; O1[r] = 0.f; }
    float m_run = -1e30f, l_run = 0.f;

    const int n_src = p.cross ? p.T - 1 : 1;
    auto src_t = [&](int si) { return !p.cross ? t : (si < t ? si : si + 1); };
    int n_tiles = 0;
    for (int si = 0; si < n_src; ++si) n_tiles += (p.nv[src_t(si)] + A3_KV - 1) / A3_KV;
    auto locate = [&](int tile, int& tt, int& kt) {
        int si = 0;
        for (;; ++si) {
            const int n = (p.nv[src_t(si)] + A3_KV - 1) / A3_KV;
            if (tile < n || si + 1 == n_src) break;
            tile -= n;
        }
        tt = src_t(si);
        kt = tile;
    };

    // staging.  K tile (64 keys x 64 dims fp32): thread -> key row tid/4, 16 dims (tid&3)*16: 4 x 16 B, 64 B contiguous.
    // V tile: thread -> a 4 keys x 4 dims block: key block kb = (lane>>2) (16 blocks), dims 16*wave + 4*(lane&3): 4 x 16 B
    // from 4 consecutive key rows; after the split it owns, per dim, 4 consecutive keys = one 8-byte V^T write per plane.
    const int k_row = tid >> 2, k_c16 = (tid & 3) * 16;
    const int v_kb = lane >> 2, v_d0 = 16 * wave + 4 * (lane & 3);
    // V^T LDS position of keys 4 kb .. 4 kb + 3 inside their 16-key group (permuted order, see the header)
    const int v_lds = 16 * (v_kb >> 2) + 4 * ((v_kb & 3) >> 1) + 8 * (v_kb & 1);
    f32x4 rk[4], rv[4];
    auto gload = [&](int tile) {
        int tt, kt;
        locate(tile, tt, kt);
        const float* base = p.qkv + ((int64_t)(b * p.T + tt) * p.n_rows + kt * A3_KV) * ld + head * A3_HD;
        const float* kp = base + p.D + (int64_t)k_row * ld + k_c16;
        const float* vp = base + 2 * p.D + (int64_t)(4 * v_kb) * ld + v_d0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rk[i] = *reinterpret_cast<const f32x4*>(kp + 4 * i);
            rv[i] = *reinterpret_cast<const f32x4*>(vp + (int64_t)i * ld);
        }
    };
    auto lstore = [&]() {
        // K: 16 values -> 3 planes x 32 B
        bf16x8 kh[3][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __bf16 a, bb, c;
                split3f(rk[i][e], a, bb, c);
                kh[0][i >> 1][(i & 1) * 4 + e] = a; kh[1][i >> 1][(i & 1) * 4 + e] = bb; kh[2][i >> 1][(i & 1) * 4 + e] = c;
            }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            uint16_t* kd = &Ks[pl * A3_PLANE + k_row * A3_LD + k_c16];
            *reinterpret_cast<bf16x8*>(kd) = kh[pl][0];
            *reinterpret_cast<bf16x8*>(kd + 8) = kh[pl][1];
        }
        // V^T: rv[i][e] = V[key 4 kb + i][dim d0 + e]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16x4 vh[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __bf16 a, bb, c;
                split3f(rv[i][e], a, bb, c);
                vh[0][i] = a; vh[1][i] = bb; vh[2][i] = c;
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                *reinterpret_cast<bf16x4*>(&Vs[pl * A3_PLANE + (v_d0 + e) * A3_LD + v_lds]) = vh[pl];
        }
    };

    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};  // plane of the A operand (K or V^T), smallest terms first
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};  // plane of the B operand (Q or P)

    gload(0);
    for (int tile = 0; tile < n_tiles; ++tile) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (tile + 1 < n_tiles) gload(tile + 1);

        int tt_cur, kt;
        locate(tile, tt_cur, kt);
        const int valid_in_tile = p.nv[tt_cur] - kt * A3_KV;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (sub * 32 >= valid_in_tile) break;
            f32x16 S;
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] = 0.f;
            const uint16_t* kp = &Ks[(sub * 32 + l31) * A3_LD + lh * 8];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 kf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) kf[pl] = *reinterpret_cast<const bf16x8*>(kp + pl * A3_PLANE + s * 16);
#pragma unroll
                for (int q = 0; q < 6; ++q) S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[PA[q]], Qf[PB[q]][s], S, 0, 0, 0);
            }
            if (valid_in_tile < sub * 32 + 32) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= valid_in_tile) S[r] = -INFINITY;
                }
            }
            float mx = S[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float ps = 0.f;
            bf16x8 Pf[3][2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(S[r] - m_new);
                ps += pv;
                __bf16 a, bb, c;
                split3f(pv, a, bb, c);
                Pf[0][r >> 3][r & 7] = a;
                Pf[1][r >> 3][r & 7] = bb;
                Pf[2][r >> 3][r & 7] = c;
            }
            l_run = l_run * alpha + ps;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) { O0[r] *= alpha; O1[r] *= alpha; }
            const uint16_t* vp = &Vs[l31 * A3_LD + lh * 8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bf16x8 v0[3], v1[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    v0[pl] = *reinterpret_cast<const bf16x8*>(vp + pl * A3_PLANE + 16 * (2 * sub + u));
                    v1[pl] = *reinterpret_cast<const bf16x8*>(vp + pl * A3_PLANE + 32 * A3_LD + 16 * (2 * sub + u));
                }
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    O0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0[PA[q]], Pf[PB[q]][u], O0, 0, 0, 0);
                    O1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1[PA[q]], Pf[PB[q]][u], O1, 0, 0, 0);
                }
            }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / l_tot;
    float* op = p.out32 + ((int64_t)img * p.n_rows + q_row) * p.D + head * A3_HD + 4 * lh;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        f32x4 a, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = O0[gq * 4 + e] * inv; c[e] = O1[gq * 4 + e] * inv; }
        *reinterpret_cast<f32x4*>(op + 8 * gq) = a;
        *reinterpret_cast<f32x4*>(op + 32 + 8 * gq) = c;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same kernel in the "f16x2" arithmetic mode (see gemm_x3.hip): every operand is carried as two fp16 planes,
// x = x_hi + x_lo, x_hi = fp16(x), x_lo = fp16(x - x_hi) - 22 significant bits - and a contraction is the 3 products
// hi hi + hi lo + lo hi (v_mfma_f32_32x32x16_f16, fp32 accumulate): half the MFMAs, two thirds of the LDS planes and of
// the split VALU work of the bf16x3 form.  Unlike the GEMM, where the weight planes are static, every operand here is
// made on the fly, so the low planes are kept in fp16's normal range by power-of-two pre-scales that cancel exactly:
//   q x 2^6 (on top of log2(e)/sqrt(d)), k as it is -> logits in units of 2^-6; the exponent is formed as fma(S, 2^-6, c)
//   p x 2^10 (p in (0, 1] -> (0, 1024]), v x 2^4  -> O = (sum p v) / (l 2^4), l summed from the same scaled p
// Range: |q| < 5.6e3, |k| < 6.5e4, |v| < 4.0e3 (beyond: +-inf -> non-finite scores -> the sticky error of e2emv_sync); the
// absolute floor of a low plane (half an fp16 subnormal step, 3e-8) sits at 5e-10 / 3e-8 / 2e-9 / 3e-11 of the unscaled
// q / k / v / p (k needs no more: its floor enters the logit as 3e-8 |q| 0.18 sqrt(64) = 4e-8 |q| in base-2 units).
constexpr float H2_QS = 64.f, H2_VS = 16.f, H2_SINV = 1.f / 64.f, H2_PLOG = 10.f, H2_LAZY = 5.f;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
// (x0 c, x1 c) -> packed fp16 pairs hi, lo with hi + lo = x c to 2^-22.  hi = one v_cvt_pk_f16_f32; lo = v - hi by
// v_fma_mix{lo,hi}_f16, which read hi as fp16 and round the exact fp32 residual once (1.5 VALU per element instead of 3).
// The products are made opaque before they are converted: left to itself hipcc 7.2 selects the high plane twice -
// v_cvt_pk_f16_f32 of the rounded fp32 product for the stored plane, v_fma_mixlo_f16(x, c, 0) for the copy the residual is
// taken against - and on gfx950 the two differ by an fp16 ulp when x c lies within an fp32 rounding of an fp16 tie (the mix
// instruction rounds the exact product once): hi + lo was then off by 2^-11 of that element.  Found as one query row in
// 512 with 1.4e-4 error; pinned by test_attention_split_kernels_near_fp16_ties.
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
struct H2Pair { unsigned hi, lo; };
__device__ __forceinline__ H2Pair split2h_pair(float x0, float x1, float c) {
    unsigned hi, lo;
    float v0 = x0 * c, v1 = x1 * c;
    asm("" : "+v"(v0), "+v"(v1));
    const f32x2 vv = {v0, v1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(vv, f16x2));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 1"
        : "=&v"(lo) : "v"(hi), "v"(v0), "v"(v1));
    return {hi, lo};
}

// NW = waves per workgroup = 32-query blocks per workgroup: 4 (128 queries, two workgroups per CU) or 8 (256 queries, one
// per CU - the K / V tile is fetched, split and stored half as often per query; waves 0-3 stage K, waves 4-7 stage V)
template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attention_h2f_kernel(Attn3fParams p) {
    constexpr int QT = 32 * NW;
    __shared__ __attribute__((aligned(16))) uint16_t Ks[2 * A3_PLANE];
    __shared__ __attribute__((aligned(16))) uint16_t Vs[2 * A3_PLANE];

    const int lin = blockIdx.x;
    const int xcd = lin & 7, idx = lin >> 3;
    const int g = xcd * p.gper + idx / p.nq;
    if (g >= p.groups) return;
    const int qt = idx % p.nq;
    const int img = g / p.H, head = g % p.H;
    const int b = img / p.T, t = img % p.T;
    if (qt * QT >= p.nv[t]) return;  // shorter image of a ragged tuple: no queries in this tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int64_t ld = 3 * (int64_t)p.D;  // floats per q|k|v row

    // ---- Q fragments (B operand): lane (q, lh) holds Q_pl[q][16 s + 8 lh .. +7]
    const int q_row = qt * QT + wave * 32 + l31;
    const bool q_ok = q_row < p.n_rows;  // (n_rows is a multiple of 128: the last 256-query tile may be half empty)
    u32x4 Qf[2][4];  // 8 fp16 each
    {
        const float* qp = p.qkv + ((int64_t)img * p.n_rows + (q_ok ? q_row : p.n_rows - 1)) * ld + head * A3_HD + lh * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(qp + s * 16), hi = *reinterpret_cast<const f32x4*>(qp + s * 16 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
            {
                const H2Pair pr = split2h_pair(e < 2 ? lo[2 * e] : hi[2 * e - 4], e < 2 ? lo[2 * e + 1] : hi[2 * e - 3], p.q_scale * H2_QS);
                Qf[0][s][e] = pr.hi; Qf[1][s][e] = pr.lo;
            }
        }
    }

    f32x16 O0, O1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { O0[r] = 0.f; O1[r] = 0.f; }
    float m_run = -1e30f, l_run = 0.f;

    const int n_src = p.cross ? p.T - 1 : 1;
    auto src_t = [&](int si) { return !p.cross ? t : (si < t ? si : si + 1); };
    int n_tiles = 0;
    for (int si = 0; si < n_src; ++si) n_tiles += (p.nv[src_t(si)] + A3_KV - 1) / A3_KV;
    // walk over the key tiles of the source images: (source index, tile inside it), advanced one tile at a time
    struct TilePos { int si, kt; };
    auto advance_pos = [&](TilePos& tp) {
        if (++tp.kt * A3_KV >= p.nv[src_t(tp.si)] && tp.si + 1 < n_src) { ++tp.si; tp.kt = 0; }
    };
    // staging.  K tile (64 keys x 64 dims fp32): thread -> key row tid/4, 16 dims (tid&3)*16: 4 x 16 B, 64 B contiguous.
    // V tile: thread -> a 4 keys x 4 dims block: key block kb = (lane>>2) (16 blocks), dims 16*wave + 4*(lane&3): 4 x 16 B
    // from 4 consecutive key rows; after the split it owns, per dim, 4 consecutive keys = one 8-byte V^T write per plane.
    const bool stage_k = NW == 4 || wave < 4, stage_v = NW == 4 || wave >= 4;
    const int stid = tid & 255, swave = wave & 3;
    const int k_row = stid >> 2, k_c16 = (stid & 3) * 16;
    const int v_kb = lane >> 2, v_d0 = 16 * swave + 4 * (lane & 3);
    // V^T LDS position of keys 4 kb .. 4 kb + 3 inside their 16-key group (permuted order, see the header)
    const int v_lds = 16 * (v_kb >> 2) + 4 * ((v_kb & 3) >> 1) + 8 * (v_kb & 1);
    f32x4 rk[4], rv[4];
    auto gload = [&](const TilePos& tp) {
        const int tt = src_t(tp.si), kt = tp.kt;
        const float* base = p.qkv + ((int64_t)(b * p.T + tt) * p.n_rows + kt * A3_KV) * ld + head * A3_HD;
        const float* kp = base + p.D + (int64_t)k_row * ld + k_c16;
        const float* vp = base + 2 * p.D + (int64_t)(4 * v_kb) * ld + v_d0;
        if (stage_k) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rk[i] = *reinterpret_cast<const f32x4*>(kp + 4 * i);
        }
        if (stage_v) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rv[i] = *reinterpret_cast<const f32x4*>(vp + (int64_t)i * ld);
        }
    };
    auto lstore = [&]() {
        // K: 16 values -> 2 planes x 32 B
        if (stage_k) {
        u32x4 kh[2][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 2; ++e)
            {
                const H2Pair pr = split2h_pair(rk[i][2 * e], rk[i][2 * e + 1], 1.f);
                kh[0][i >> 1][(i & 1) * 2 + e] = pr.hi; kh[1][i >> 1][(i & 1) * 2 + e] = pr.lo;
            }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            uint16_t* kd = &Ks[pl * A3_PLANE + k_row * A3_LD + k_c16];
            *reinterpret_cast<u32x4*>(kd) = kh[pl][0];
            *reinterpret_cast<u32x4*>(kd + 8) = kh[pl][1];
        }
        }
        // V^T: rv[i][e] = V[key 4 kb + i][dim d0 + e]
        if (stage_v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u32x2 vh[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const H2Pair pr = split2h_pair(rv[2 * i][e], rv[2 * i + 1][e], H2_VS);
                vh[0][i] = pr.hi; vh[1][i] = pr.lo;
            }
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                *reinterpret_cast<u32x2*>(&Vs[pl * A3_PLANE + (v_d0 + e) * A3_LD + v_lds]) = vh[pl];
        }
    };

    constexpr int PA[3] = {1, 0, 0};  // plane of the A operand (K or V^T), smallest terms first
    constexpr int PB[3] = {0, 1, 0};  // plane of the B operand (Q or P)

    TilePos cur{0, 0}, nxt{0, 0};
    gload(nxt);
    for (int tile = 0; tile < n_tiles; ++tile) {
        __syncthreads();
        lstore();
        __syncthreads();
        cur = nxt;
        if (tile + 1 < n_tiles) {
            advance_pos(nxt);
            gload(nxt);
        }
        const int valid_in_tile = p.nv[src_t(cur.si)] - cur.kt * A3_KV;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (sub * 32 >= valid_in_tile) break;
            f32x16 S;
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] = 0.f;
            const uint16_t* kp = &Ks[(sub * 32 + l31) * A3_LD + lh * 8];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f16x8 kf[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) kf[pl] = *reinterpret_cast<const f16x8*>(kp + pl * A3_PLANE + s * 16);
#pragma unroll
                for (int q = 0; q < 3; ++q) S = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[PA[q]], __builtin_bit_cast(f16x8, Qf[PB[q]][s]), S, 0, 0, 0);
            }
            if (valid_in_tile < sub * 32 + 32) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= valid_in_tile) S[r] = -INFINITY;
                }
            }
            float mx = S[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            // S (and m) are in units of 1 / H2_QS of a base-2 logit; P carries the factor 2^H2_PLOG (cancels in O / l).
            // Lazy running maximum: the reference maximum m_run moves (and O, l are rescaled) only when some row's new
            // maximum exceeds it by more than 2^H2_LAZY - until then P just grows up to 2^(H2_PLOG + H2_LAZY) = 32768, inside
            // fp16's range, and the tile costs no rescale of the 32 O registers (after the first tiles that is every tile)
            float m_new = m_run, alpha = 1.f;
            const bool grow = (mx - m_run) * H2_SINV > H2_LAZY;
            if (__builtin_amdgcn_ballot_w64(grow) != 0) {  // wave-uniform
                m_new = fmaxf(m_run, mx);
                alpha = __builtin_amdgcn_exp2f((m_run - m_new) * H2_SINV);
#pragma unroll
                for (int r = 0; r < 16; ++r) { O0[r] *= alpha; O1[r] *= alpha; }
            }
            const float e0 = H2_PLOG - m_new * H2_SINV;
            float ps = 0.f;
            u32x4 Pf[2][2];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], H2_SINV, e0));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r + 1], H2_SINV, e0));
                ps += p0 + p1;
                const H2Pair pr = split2h_pair(p0, p1, 1.f);
                Pf[0][r >> 3][(r & 7) >> 1] = pr.hi; Pf[1][r >> 3][(r & 7) >> 1] = pr.lo;
            }
            l_run = l_run * alpha + ps;
            m_run = m_new;
            const uint16_t* vp = &Vs[l31 * A3_LD + lh * 8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f16x8 v0[2], v1[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    v0[pl] = *reinterpret_cast<const f16x8*>(vp + pl * A3_PLANE + 16 * (2 * sub + u));
                    v1[pl] = *reinterpret_cast<const f16x8*>(vp + pl * A3_PLANE + 32 * A3_LD + 16 * (2 * sub + u));
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    O0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0[PA[q]], __builtin_bit_cast(f16x8, Pf[PB[q]][u]), O0, 0, 0, 0);
                    O1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1[PA[q]], __builtin_bit_cast(f16x8, Pf[PB[q]][u]), O1, 0, 0, 0);
                }
            }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / (l_tot * H2_VS);
    if (!q_ok) return;
    float* op = p.out32 + ((int64_t)img * p.n_rows + q_row) * p.D + head * A3_HD + 4 * lh;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        f32x4 a, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = O0[gq * 4 + e] * inv; c[e] = O1[gq * 4 + e] * inv; }
        *reinterpret_cast<f32x4*>(op + 8 * gq) = a;
        *reinterpret_cast<f32x4*>(op + 32 + 8 * gq) = c;
    }
}

int launch_attention3f(e2emv_ctx* ctx, int B, int T, int n_rows, const int* nv, int D, int H, const float* qkv, int cross,
                       float* out32, hipStream_t s, bool h2) {
    int n_valid = 0;
    for (int t = 0; t < T; ++t) {
        if (nv[t] <= 0 || nv[t] > n_rows) return set_err(ctx, E2EMV_ESHAPE, "attention3f: image %d has %d keypoints (n_rows %d)", t, nv[t], n_rows);
        n_valid = std::max(n_valid, nv[t]);
    }
    if (D != H * A3_HD) return set_err(ctx, E2EMV_ESHAPE, "attention3f: head dim must be 64 (D=%d H=%d)", D, H);
    if (n_rows % A3_Q || n_valid <= 0 || n_valid > n_rows)
        return set_err(ctx, E2EMV_ESHAPE, "attention3f: n_rows=%d must be a multiple of %d and >= n_valid=%d", n_rows, A3_Q, n_valid);
    if (cross && T < 2) return set_err(ctx, E2EMV_ESHAPE, "attention3f: cross layer needs T >= 2");
    if (!qkv || !out32 || (uintptr_t)qkv % 16 || (uintptr_t)out32 % 16) return set_err(ctx, E2EMV_EINVAL, "attention3f: null / unaligned buffer");
    Attn3fParams p;
    p.qkv = qkv; p.out32 = out32; p.B = B; p.T = T; p.n_rows = n_rows; p.D = D; p.H = H;
    for (int t = 0; t < E2EMV_MAX_TUPLE; ++t) p.nv[t] = t < T ? nv[t] : 0;
    p.cross = cross;
    p.nq = (n_valid + A3_Q - 1) / A3_Q;
    p.groups = B * T * H;
    p.gper = (p.groups + 7) / 8;
    p.q_scale = 0.125f * 1.4426950408889634f;  // log2(e) / sqrt(64)
    if (h2) {
        static int nw_env = -1;  // E2EMV_A3_NW=4|8 forces the workgroup size
        if (nw_env < 0) nw_env = dbg_knob("E2EMV_A3_NW", 0);
        const int nw = nw_env == 4 || nw_env == 8 ? nw_env : (n_valid > 1024 ? 8 : 4);  // measured: equal at 1024 keys, 4-5 % at 2048
        if (nw == 8) {
            p.nq = (n_valid + 255) / 256;
            hipLaunchKernelGGL(attention_h2f_kernel<8>, dim3(8 * p.gper * p.nq), dim3(512), 0, s, p);
        } else {
            hipLaunchKernelGGL(attention_h2f_kernel<4>, dim3(8 * p.gper * p.nq), dim3(256), 0, s, p);
        }
    }
    else hipLaunchKernelGGL(attention3f_kernel, dim3(8 * p.gper * p.nq), dim3(256), 0, s, p);
    E2EMV_CHECK_LAUNCH(ctx, "attention3f_kernel");
    return E2EMV_OK;
}

int launch_attention3(e2emv_ctx* ctx, int B, int T, int n_rows, const int* nv, int D, int H, const uint16_t* qk,
                      const uint16_t* vt, int cross, uint16_t* out3, float* out32, hipStream_t s) {
    int n_valid = 0;
    for (int t = 0; t < T; ++t) {
        if (nv[t] <= 0 || nv[t] > n_rows) return set_err(ctx, E2EMV_ESHAPE, "attention3: image %d has %d keypoints (n_rows %d)", t, nv[t], n_rows);
        n_valid = std::max(n_valid, nv[t]);
    }
    if (D != H * A3_HD) return set_err(ctx, E2EMV_ESHAPE, "attention3: head dim must be 64 (D=%d H=%d)", D, H);
    if (n_rows % A3_Q || n_valid <= 0 || n_valid > n_rows)
        return set_err(ctx, E2EMV_ESHAPE, "attention3: n_rows=%d must be a multiple of %d and >= n_valid=%d", n_rows, A3_Q, n_valid);
    if (cross && T < 2) return set_err(ctx, E2EMV_ESHAPE, "attention3: cross layer needs T >= 2");
    Attn3Params p;
    p.qk = qk; p.vt = vt; p.out = out3; p.out32 = out32; p.B = B; p.T = T; p.n_rows = n_rows; p.D = D; p.H = H;
    for (int t = 0; t < E2EMV_MAX_TUPLE; ++t) p.nv[t] = t < T ? nv[t] : 0;
    p.cross = cross;
    p.nq = (n_valid + A3_Q - 1) / A3_Q;
    p.groups = B * T * H;
    p.gper = (p.groups + 7) / 8;
    hipLaunchKernelGGL(attention3_kernel, dim3(8 * p.gper * p.nq), dim3(256), 0, s, p);
    E2EMV_CHECK_LAUNCH(ctx, "attention3_kernel");
    return E2EMV_OK;
}

}  // namespace e2emv
