// Fused multi-head attention softmax(q k^T / sqrt(d)) v for the attentional GNN, fp32 on the
// gfx950 matrix cores, flash-style (the H x N x N_src probability tensor the reference
// materialises per layer - upstream superglue.py `attention()`, einsum 'bdhn,bdhm->bhnm' -
// never exists).  Self layers attend inside an image, cross layers to every other image of
// the tuple (N_src = (T-1) N; T = 2 is upstream SuperGlue's swap).
//
// Layout: qkv [n_img][n_rows][3D], channels head-major (c = h*64 + dd; the upstream order
// c = dd*H + h is undone on the weights at load time, ctx.hip), image g = b*T + t.
//
// Per workgroup: one (image, head, 128-query tile); 4 waves x 32 queries.  Everything is
// computed TRANSPOSED so that each lane owns ONE query (q = lane & 31) for its whole life:
//   S^T[key][q] = mfma(A = K tile rows, B = Q)          -> a lane holds 16 keys of its query
//   O^T[d][q]  += mfma(A = V^T,           B = P^T)      -> the B operand of step t is exactly
// register t of the exponentiated S^T accumulator (MFMA 32x32 C layout: row = (r&3)+8(r>>2)
// +4(lane>>5)), so P never moves between lanes or through LDS, the running max/sum are
// per-lane scalars (+ one lane^32 exchange), and rescaling O^T is lane-local.
// K tiles sit in LDS padded to 68 floats (conflict-free ds_read_b128, one read feeds four
// MFMAs through the same K-slot permutation as gemm.hip); V tiles unpadded (ds_read_b32 of
// 32 consecutive floats).  128 MFMAs (8192 cycles) per wave per 64-key tile against ~1.3k
// VALU cycles of softmax: the kernel is MFMA-issue bound by construction.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace e2emv {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int ATT_Q = 128;   // queries per workgroup
constexpr int ATT_KV = 64;   // keys per LDS tile
constexpr int HD = 64;       // head dim
constexpr int KLD = 68;      // padded K row (floats)

struct AttnParams {
    const float* qkv;
    float* out;
    int B, T, n_rows, D, H, cross;
    int nv[E2EMV_MAX_TUPLE];  // valid keypoints (queries and keys) of image t of a tuple
    int nq, groups, gper;
    // key-split mode (small problems): split sp of ksplit handles a slice of the key tiles and writes UNNORMALISED partial
    // outputs + (running max, sum) per query; attention_merge_kernel folds them
    int ksplit;
    float* part_o;   // [n_img][H][n_rows][ksplit][64]
    float* part_ml;  // [n_img][H][n_rows][ksplit][2]
};

template <int DBG, bool SPLIT = false>
__global__ __launch_bounds__(256, 3) void attention_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) float Ks[ATT_KV * KLD];
    __shared__ __attribute__((aligned(16))) float Vs[ATT_KV * HD];

    // XCD-aware mapping: all query tiles of one (image, head) share an XCD (K/V stay in its L2)
    const int lin = blockIdx.x;
    const int xcd = lin & 7;
    const int sp = SPLIT ? (lin >> 3) % p.ksplit : 0;
    const int idx = SPLIT ? (lin >> 3) / p.ksplit : lin >> 3;
    const int g = xcd * p.gper + idx / p.nq;
    if (g >= p.groups) return;
    const int qt = idx % p.nq;
    const int img = g / p.H, head = g % p.H;
    const int b = img / p.T, t = img % p.T;
    if (qt * ATT_Q >= p.nv[t]) return;  // shorter image of a ragged tuple: no queries in this tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int64_t row_stride = 3 * (int64_t)p.D;
    const int64_t img_stride = (int64_t)p.n_rows * row_stride;

    // ---- Q fragment: lane (q, lh) holds Q[q][8s + 4lh .. +3], pre-scaled by log2(e)/sqrt(d)
    const int q_row = qt * ATT_Q + wave * 32 + l31;
    const float qscale = 0.125f * 1.4426950408889634f;
    f32x4 Qr[8];
    {
        const float* qp = p.qkv + img * img_stride + (int64_t)q_row * row_stride + head * HD + lh * 4;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            f32x4 v = *reinterpret_cast<const f32x4*>(qp + s * 8);
            Qr[s] = v * qscale;
        }
    }

    f32x16 O0, O1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { O0[r] = 0.f; O1[r] = 0.f; }
    float m_run = -1e30f, l_run = 0.f;

    const int n_src = p.cross ? p.T - 1 : 1;
    auto src_t = [&](int si) { return !p.cross ? t : (si < t ? si : si + 1); };
    int n_tiles = 0;
    for (int si = 0; si < n_src; ++si) n_tiles += (p.nv[src_t(si)] + ATT_KV - 1) / ATT_KV;
    // linear key-tile index -> (source image of the tuple, tile inside it); sources may differ in length
    auto locate = [&](int tile, int& tt, int& kt) {
        int si = 0;
        for (;; ++si) {
            const int n = (p.nv[src_t(si)] + ATT_KV - 1) / ATT_KV;
            if (tile < n || si + 1 == n_src) break;
            tile -= n;
        }
        tt = src_t(si);
        kt = tile;
    };

    // staging map: 4 float4 of K and of V per thread
    const int st_row = tid >> 4;         // 0..15 (+16*i)
    const int st_c4 = (tid & 15) * 4;
    f32x4 rk[4], rv[4];
    auto gload = [&](int tile) {
        int tt, kt;
        locate(tile, tt, kt);
        const float* base = p.qkv + (b * p.T + tt) * img_stride + (int64_t)(kt * ATT_KV) * row_stride + head * HD + st_c4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* rp = base + (int64_t)(st_row + 16 * i) * row_stride;
            rk[i] = *reinterpret_cast<const f32x4*>(rp + p.D);
            rv[i] = *reinterpret_cast<const f32x4*>(rp + 2 * p.D);
        }
    };

    const int tile_lo = SPLIT ? (int)((int64_t)n_tiles * sp / p.ksplit) : 0;
    const int tile_hi = SPLIT ? (int)((int64_t)n_tiles * (sp + 1) / p.ksplit) : n_tiles;
    if (tile_lo < tile_hi) gload(tile_lo);
    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        if (!(DBG & 4) || tile == tile_lo) {
        __syncthreads();  // everyone is done reading the previous tile
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(&Ks[(st_row + 16 * i) * KLD + st_c4]) = rk[i];
            *reinterpret_cast<f32x4*>(&Vs[(st_row + 16 * i) * HD + st_c4]) = rv[i];
        }
        __syncthreads();
        if (tile + 1 < tile_hi) gload(tile + 1);
        }

        int tt_cur, kt;
        locate(tile, tt_cur, kt);
        const int valid_in_tile = p.nv[tt_cur] - kt * ATT_KV;  // >= 1
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (sub * 32 >= valid_in_tile) break;  // wave-uniform
            f32x16 S;
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] = 0.f;
            const float* kp = &Ks[(sub * 32 + l31) * KLD + lh * 4];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                f32x4 kf = *reinterpret_cast<const f32x4*>(kp + s * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) S = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], Qr[s][e], S, 0, 0, 0);
            }
            if (valid_in_tile < sub * 32 + 32) {  // ragged tail: mask keys >= n_valid
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
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (!(DBG & 1)) S[r] = __builtin_amdgcn_exp2f(S[r] - m_new);
                ps += S[r];
            }
            l_run = l_run * alpha + ps;
            m_run = m_new;
            if (!(DBG & 2)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { O0[r] *= alpha; O1[r] *= alpha; }
            }
            // O^T[d][q] += V^T[d][key] P^T[key][q]
            const float* vp = &Vs[(sub * 32 + 4 * lh) * HD + l31];
#pragma unroll
            for (int tt = 0; tt < 16; ++tt) {
                const int krow = (tt & 3) + 8 * (tt >> 2);
                const float v0 = vp[krow * HD];
                const float v1 = vp[krow * HD + 32];
                O0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, S[tt], O0, 0, 0, 0);
                O1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, S[tt], O1, 0, 0, 0);
            }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    if (SPLIT) {
        const int64_t slot = (((int64_t)img * p.H + head) * p.n_rows + q_row) * p.ksplit + sp;
        float* po = p.part_o + slot * HD + 4 * lh;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            f32x4 a, c;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = O0[gq * 4 + e]; c[e] = O1[gq * 4 + e]; }
            *reinterpret_cast<f32x4*>(po + 8 * gq) = a;
            *reinterpret_cast<f32x4*>(po + 32 + 8 * gq) = c;
        }
        if (lh == 0) { p.part_ml[slot * 2] = m_run; p.part_ml[slot * 2 + 1] = l_tot; }
        return;
    }
    const float inv = 1.f / l_tot;
    float* op = p.out + ((int64_t)img * p.n_rows + q_row) * p.D + head * HD + 4 * lh;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        f32x4 a, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = O0[gq * 4 + e] * inv; c[e] = O1[gq * 4 + e] * inv; }
        *reinterpret_cast<f32x4*>(op + 8 * gq) = a;
        *reinterpret_cast<f32x4*>(op + 32 + 8 * gq) = c;
    }
}

// folds the key-split partials: out[q][:] = sum_s O_s 2^(m_s - M) / sum_s l_s 2^(m_s - M)
__global__ __launch_bounds__(256) void attention_merge_kernel(AttnParams p) {
    // one thread per (image, head, query, 4 channels)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c4 = (int)(i & 15);
    const int64_t rowi = i >> 4;  // (img * H + head) * n_rows + q
    if (rowi >= (int64_t)p.groups * p.n_rows) return;
    const int q = (int)(rowi % p.n_rows);
    const int gh = (int)(rowi / p.n_rows), img = gh / p.H, head = gh % p.H;
    if (q >= p.nv[img % p.T]) return;
    float M = -1e30f;
    for (int s2 = 0; s2 < p.ksplit; ++s2) M = fmaxf(M, p.part_ml[(rowi * p.ksplit + s2) * 2]);
    float L = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s2 = 0; s2 < p.ksplit; ++s2) {
        const float w = __builtin_amdgcn_exp2f(p.part_ml[(rowi * p.ksplit + s2) * 2] - M);
        L += p.part_ml[(rowi * p.ksplit + s2) * 2 + 1] * w;
        acc += *reinterpret_cast<const f32x4*>(p.part_o + (rowi * p.ksplit + s2) * HD + 4 * c4) * w;
    }
    *reinterpret_cast<f32x4*>(p.out + ((int64_t)img * p.n_rows + q) * p.D + head * HD + 4 * c4) = acc * (1.f / L);
}

int launch_attention(e2emv_ctx* ctx, int B, int T, int n_rows, const int* nv, int D, int H, const float* qkv,
                     int cross, float* out, hipStream_t s) {
    int n_valid = 0;
    for (int t = 0; t < T; ++t) {
        if (nv[t] <= 0 || nv[t] > n_rows) return set_err(ctx, E2EMV_ESHAPE, "attention: image %d has %d keypoints (n_rows %d)", t, nv[t], n_rows);
        n_valid = std::max(n_valid, nv[t]);
    }
    if (D != H * HD) return set_err(ctx, E2EMV_ESHAPE, "attention: head dim must be 64 (D=%d H=%d)", D, H);
    if (n_rows % ATT_Q || n_valid <= 0 || n_valid > n_rows)
        return set_err(ctx, E2EMV_ESHAPE, "attention: n_rows=%d must be a multiple of %d and >= n_valid=%d", n_rows, ATT_Q, n_valid);
    if (cross && T < 2) return set_err(ctx, E2EMV_ESHAPE, "attention: cross layer needs T >= 2");
    AttnParams p;
    p.qkv = qkv; p.out = out; p.B = B; p.T = T; p.n_rows = n_rows; p.D = D; p.H = H;
    for (int t = 0; t < E2EMV_MAX_TUPLE; ++t) p.nv[t] = t < T ? nv[t] : 0;
    p.cross = cross;
    p.nq = (n_valid + ATT_Q - 1) / ATT_Q;
    p.groups = B * T * H;
    p.gper = (p.groups + 7) / 8;
    dim3 grid(8 * p.gper * p.nq);
    static int dbg = -1;  // E2EMV_ATTN_DEBUG: ablation variants for profiling only (results are wrong when != 0)
    if (dbg < 0) dbg = dbg_knob("E2EMV_ATTN_DEBUG", 0);
    // Small problems (batch 1-2 of the reference's eval loop): one workgroup per (image, head, 128 queries) leaves most CUs
    // idle while each workgroup walks all key tiles serially.  Split the key tiles over up to 8 workgroups + a merge pass.
    static int split_env = -1;  // E2EMV_ATTN_SPLIT: 0 off, n > 1 forces n
    if (split_env < 0) split_env = dbg_knob("E2EMV_ATTN_SPLIT", -1);
    p.ksplit = 1; p.part_o = nullptr; p.part_ml = nullptr;
    {
        int min_tiles = 1 << 30;  // key tiles a query walks (smallest over the images)
        for (int t = 0; t < T; ++t) {
            int n = 0;
            for (int u = 0; u < T; ++u)
                if (cross ? u != t : u == t) n += (nv[u] + ATT_KV - 1) / ATT_KV;
            min_tiles = std::min(min_tiles, n);
        }
        const int blocks = p.groups * p.nq;
        int ks = split_env > 1 ? split_env : (split_env == 0 ? 1 : (blocks * 2 <= ctx->num_cus ? ctx->num_cus / blocks : 1));
        ks = std::max(1, std::min(std::min(ks, 8), min_tiles));
        if (ks > 1 && dbg == 0) {
            const size_t rows = (size_t)p.groups * n_rows * ks;
            const size_t need = rows * (HD + 2) * sizeof(float);
            if (ctx->attn_part_bytes < need) {
                E2EMV_HIP(ctx, hipStreamSynchronize(s));
                if (ctx->d_attn_part) E2EMV_HIP(ctx, hipFree(ctx->d_attn_part));
                ctx->d_attn_part = nullptr; ctx->attn_part_bytes = 0;
                E2EMV_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_attn_part), need));
                ctx->attn_part_bytes = need;
            }
            p.ksplit = ks;
            p.part_o = ctx->d_attn_part;
            p.part_ml = ctx->d_attn_part + rows * HD;
            hipLaunchKernelGGL((attention_kernel<0, true>), dim3(8 * p.gper * p.nq * ks), dim3(256), 0, s, p);
            const int64_t threads = (int64_t)p.groups * n_rows * 16;
            hipLaunchKernelGGL(attention_merge_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, p);
            E2EMV_CHECK_LAUNCH(ctx, "attention_kernel (key-split)");
            return E2EMV_OK;
        }
    }
    switch (dbg) {
        case 1: hipLaunchKernelGGL(attention_kernel<1>, grid, dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL(attention_kernel<2>, grid, dim3(256), 0, s, p); break;
        case 3: hipLaunchKernelGGL(attention_kernel<3>, grid, dim3(256), 0, s, p); break;
        case 4: hipLaunchKernelGGL(attention_kernel<4>, grid, dim3(256), 0, s, p); break;
        case 7: hipLaunchKernelGGL(attention_kernel<7>, grid, dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL(attention_kernel<0>, grid, dim3(256), 0, s, p); break;
    }
    E2EMV_CHECK_LAUNCH(ctx, "attention_kernel");
    return E2EMV_OK;
}

}  // namespace e2emv

extern "C" int e2emv_attention(e2emv_ctx* ctx, int B, int T, int n_rows, int n_valid, int D, int H, const float* d_qkv,
                               int cross, float* d_out, void* stream) {
    if (!ctx || !d_qkv || !d_out) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    e2emv::prof_begin(ctx, e2emv::PS_ATTN, (hipStream_t)stream);
    if (T < 1 || T > E2EMV_MAX_TUPLE) return E2EMV_EINVAL;
    int nv[E2EMV_MAX_TUPLE];
    for (int t = 0; t < E2EMV_MAX_TUPLE; ++t) nv[t] = n_valid;
    int rc = e2emv::launch_attention(ctx, B, T, n_rows, nv, D, H, d_qkv, cross, d_out, (hipStream_t)stream);
    e2emv::prof_end(ctx, (hipStream_t)stream);
    return rc;
}
