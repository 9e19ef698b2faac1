// e2emv_matcher_forward: the MultiViewMatcher.forward replacement (reference call sites
// helpers.py:246, eval_pairs.py:212, eval_multi_view.py:160; algorithm = upstream SuperGlue
// superglue.py SuperGlue.forward, see oracle/matcher.py for the restated spec).
//
//   ingest (descriptor transpose [B,D,N] -> [img][row][D], keypoint normalisation + first
//   keypoint-encoder layer)  ->  kenc MLP (MFMA GEMMs, last one adds the descriptors)
//   -> L x { q|k|v GEMM, fused attention, merge GEMM, MLP0 GEMM (+BN folded, ReLU) over the
//   un-materialised concat [x | message], MLP1 GEMM + residual }  -> final_proj GEMM
//   -> per pair: score GEMM (1/sqrt(D) fused) -> one-sweep Sinkhorn -> match block -> conf head.
//
// HBM layout: image g = b*T + t owns n_rows = round_up(N,128) rows of D contiguous channels
// in every activation buffer, so every GEMM sees one [B*T*n_rows] x D matrix, keys/queries of
// one image are contiguous, and a tuple's images are adjacent (cross-attention sources).
// Rows >= N are zeroed at ingest and stay finite; attention masks them as keys.
#include <hip/hip_fp16.h>

#include <algorithm>

#include "common.h"
#include "ingest.h"
#include "p2.h"

namespace e2emv {

typedef __attribute__((ext_vector_type(4))) float f32x4;

// [B][D][N] (N contiguous) -> [img][n_rows][D] (D contiguous); rows >= N := 0
__global__ __launch_bounds__(256) void ingest_transpose(IngestParams p) {
    __shared__ float tile[64][65];
    const int n0 = blockIdx.x * 64, d0 = blockIdx.y * 64, img = blockIdx.z;
    const int b = img / p.T, t = img % p.T;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = n0 + tx;
    for (int dd = ty; dd < 64; dd += 4) {
        float v = 0.f;
        if (n < p.Nimg[t]) {
            const int64_t o = ((int64_t)b * p.D + d0 + dd) * p.Nimg[t] + n;
            v = p.f16 ? __half2float(reinterpret_cast<const __half*>(p.desc[t])[o])
                      : reinterpret_cast<const float*>(p.desc[t])[o];
        }
        tile[dd][tx] = v;
    }
    __syncthreads();
    for (int nn = ty; nn < 64; nn += 4)
        p.x0[((int64_t)img * p.n_rows + n0 + nn) * p.D + d0 + tx] = tile[tx][nn];
}

// keypoint normalisation + kenc layer 0 (3 -> c0, BN folded, ReLU); rows >= N := 0
__global__ __launch_bounds__(256) void ingest_kenc0(IngestParams p) {
    const int row = blockIdx.x * 256 + threadIdx.x, img = blockIdx.y;
    if (row >= p.n_rows) return;
    const int b = img / p.T, t = img % p.T;
    float* out = p.h0 + ((int64_t)img * p.n_rows + row) * p.c0;
    const int Nn = p.Nimg[t];
    if (row >= Nn) {
        for (int c = 0; c < p.c0; c += 4) *reinterpret_cast<f32x4*>(out + c) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const float W = p.img_w[t], H = p.img_h[t];
    const float sc = fmaxf(W, H) * 0.7f;
    const float kx = (p.kpts[t][((int64_t)b * Nn + row) * 2] - W / 2) / sc;
    const float ky = (p.kpts[t][((int64_t)b * Nn + row) * 2 + 1] - H / 2) / sc;
    const float ks = p.ksc[t][(int64_t)b * Nn + row];
    if (p.inp) *reinterpret_cast<f32x4*>(p.inp + ((int64_t)img * p.n_rows + row) * 4) = f32x4{kx, ky, ks, 0.f};
    for (int c = 0; c < p.c0; c += 4) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* w = p.w0 + (c + e) * 3;
            o[e] = relu_nan(w[0] * kx + w[1] * ky + w[2] * ks + p.b0[c + e]);
        }
        *reinterpret_cast<f32x4*>(out + c) = o;
    }
}

// conf head, step 1: feat2[b][n][:] = mdesc_j[b][max(match,0)][:]
__global__ __launch_bounds__(256) void conf_gather_kernel(int N, int n_rows, int D, const float* mdesc_j, int64_t tuple_stride,
                                                          const int64_t* matches, float* out) {
    const int b = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    int64_t j = row < N ? matches[(int64_t)b * N + row] : 0;
    if (j < 0) j = 0;
    const float* src = mdesc_j + b * tuple_stride + j * D;
    float* dst = out + ((int64_t)b * n_rows + row) * D;
    for (int c = lane * 4; c < D; c += 256) *reinterpret_cast<f32x4*>(dst + c) = *reinterpret_cast<const f32x4*>(src + c);
}

// f16x2 mode: both halves of the conf MLP's input in one matrix, feat[b][n][:] = [mdesc_i[b][n][:] | mdesc_j[b][max(match,0)][:]]
// (the fp16 x 2 GEMM takes one un-batched row-major operand)
__global__ __launch_bounds__(256) void conf_gather2_kernel(int N, int n_rows, int D, const float* mdesc_i, const float* mdesc_j, int64_t tuple_stride,
                                                           const int64_t* matches, float* out) {
    const int b = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    int64_t j = row < N ? matches[(int64_t)b * N + row] : 0;
    if (j < 0) j = 0;
    const float* si = mdesc_i + b * tuple_stride + (int64_t)row * D;
    const float* sj = mdesc_j + b * tuple_stride + j * D;
    float* dst = out + ((int64_t)b * n_rows + row) * 2 * D;
    for (int c = lane * 4; c < D; c += 256) {
        *reinterpret_cast<f32x4*>(dst + c) = *reinterpret_cast<const f32x4*>(si + c);
        *reinterpret_cast<f32x4*>(dst + D + c) = *reinterpret_cast<const f32x4*>(sj + c);
    }
}

// conf head, last step: sigmoid(<hidden, w> + b) for matched keypoints, 0 otherwise;
// without conf_mlp the confidence is the match score (reference quirk E13)
__global__ __launch_bounds__(256) void conf_final_kernel(int N, int n_rows, int D, const float* hidden, const float* w, float bias,
                                                         const int64_t* matches, const float* mscores, int use_mlp, float* conf) {
    const int b = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const bool valid = matches[(int64_t)b * N + row] >= 0;
    float r = 0.f;
    if (use_mlp) {
        const float* h = hidden + ((int64_t)b * n_rows + row) * D;
        float acc = 0.f;
        for (int c = lane * 4; c < D; c += 256) {
            f32x4 hv = *reinterpret_cast<const f32x4*>(h + c), wv = *reinterpret_cast<const f32x4*>(w + c);
            acc += hv[0] * wv[0] + hv[1] * wv[1] + hv[2] * wv[2] + hv[3] * wv[3];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        r = 1.f / (1.f + __expf(-(acc + bias)));
    } else {
        r = mscores[(int64_t)b * N + row];
    }
    if (lane == 0) conf[(int64_t)b * N + row] = valid ? r : 0.f;
}

static int round_up(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace e2emv

using namespace e2emv;

static int forward_joint(e2emv_ctx* ctx, const e2emv_forward_desc* fd, const float* const* d_kpts,
                         const float* const* d_kscores, const void* const* d_desc, float* const* d_logZ,
                         int64_t* const* d_m0, int64_t* const* d_m1, float* const* d_ms0, float* const* d_ms1,
                         float* const* d_conf, hipStream_t s) {
    const int B = fd->batch, T = fd->tuple_size;
    const int D = ctx->model.desc_dim, H = ctx->model.num_heads;
    // per-image keypoint counts (eval_pairs.py feeds images with different numbers of keypoints)
    int Nt[E2EMV_MAX_TUPLE] = {0};
    int N = 0;
    bool uniform = true;
    for (int t = 0; t < T; ++t) {
        Nt[t] = fd->n_kpts_img[t] > 0 ? fd->n_kpts_img[t] : fd->n_kpts;
        N = std::max(N, Nt[t]);
        uniform = uniform && Nt[t] == Nt[0];
    }
    const int n_rows = round_up(N, 128);
    const int n_img = B * T;
    const int64_t Mtot = (int64_t)n_img * n_rows;
    const int P = T * (T - 1) / 2;
    const int ldS = round_up(N, 4);  // allocation stride; a pair (i, j) uses ld = round_up(N_j, 4)
    const bool full = (fd->flags & E2EMV_FLAG_FULL_OUTPUT) != 0;

    // ---- workspace ----
    auto al = [](size_t bytes) { return (bytes + 255) & ~size_t(255); };
    const size_t sz_x = al((size_t)Mtot * D * 4), sz_qkv = al((size_t)Mtot * 3 * D * 4), sz_hid = al((size_t)Mtot * 2 * D * 4);
    const size_t sz_S = al((size_t)P * B * N * ldS * 4);
    const size_t sz_sk = sinkhorn_ws_bytes(P * B, N, N);
    const size_t sz_match = full ? (size_t)P * (al((size_t)B * N * 8) + al((size_t)B * N * 4)) : 0;
    // bf16x3 (split operands on the bf16 pipe) pays once the 128-row GEMM tiles fill the chip; calls below half a tile per
    // CU (a pair or two - the eval_pairs.py loop) run the fp32-MFMA kernels, whose 64 x 64 tile shape and key-split
    // attention are the latency-tuned forms.  Both arithmetic modes meet the same parity bar.
    const int64_t split_min = ctx->split_min_rows >= 0 ? ctx->split_min_rows : (int64_t)128 * (ctx->num_cus / 2);
    const bool b3 = ctx->precision != E2EMV_PRECISION_F32 && ctx->fuse_merge && Mtot >= split_min;
    const bool h2 = b3 && ctx->precision == E2EMV_PRECISION_F16X2;  // fp16 x 2 planes instead of bf16 x 3 (gemm_x3.hip)
    // f16x2 on PLANE activations (p2.h): every producer epilogue emits the two fp16 planes, consumers load them straight
    // into LDS (gemm_p2.hip, attention_p2.hip); h2_legacy keeps the round-2 kernels (fp32 activations, split in the consumer)
    // (their operands are addressed with 32-bit byte offsets: the widest plane matrices, q | k and the hidden layer, are
    // Mtot x 2D x 4 bytes - beyond 2 GB, 2^20 rows at D = 256, the call runs on the round-2 kernels)
    const bool p2 = h2 && !ctx->h2_legacy && D == 256 && H == 4 && !ctx->layers.empty() && Mtot * 8 * D < ((int64_t)1 << 31);
    // bf16x3 attention path: x as S3 planes (6D bytes/row) and V^T planes (6D bytes/row); the q|k planes
    // (S3, 2D wide = 12D bytes/row) live in the fp32 q|k|v buffer, which has exactly that size
    const size_t sz_x3 = b3 ? al((size_t)Mtot * 3 * D * 2) : 0;
    // tile exponents of the plane tensors (p2.h): one int per 64 rows x 64 columns of x (4), attention output (4), q|k (8),
    // V^T (4), hidden (8)
    const size_t sz_e = p2 ? al((size_t)(Mtot / 64) * 32 * sizeof(int)) : 0;  // + max |x| per block (4 floats)
    const size_t need = sz_x * 3 + sz_qkv + sz_hid + sz_S + sz_sk + sz_match + sz_x3 + sz_e + 4096;
    int rc = ws_reserve(ctx, need);
    if (rc) return rc;
    char* w = ctx->d_ws;
    float* x = (float*)w; w += sz_x;
    float* att = (float*)w; w += sz_x;     // attention output, later mdesc
    float* msg = (float*)w; w += sz_x;     // merge output, later conf gather / hidden
    float* qkv = (float*)w; w += sz_qkv;
    float* hid = (float*)w; w += sz_hid;
    float* S = (float*)w; w += sz_S;
    char* skws = w; w += sz_sk;
    uint16_t* qk3 = nullptr; uint16_t* vt3 = nullptr;
    if (b3) {
        vt3 = (uint16_t*)w; w += sz_x3;
        qk3 = (uint16_t*)qkv;
    }
    int* e_x = (int*)w; w += sz_e;
    int* e_att = e_x + (Mtot / 64) * 4;
    int* e_qk = e_att + (Mtot / 64) * 4;
    int* e_vt = e_qk + (Mtot / 64) * 8;
    int* e_hid = e_vt + (Mtot / 64) * 4;
    float* a_x = (float*)(e_hid + (Mtot / 64) * 8);  // max |x| of the blocks: the residual's share of the bound that picks x's next exponent
    std::vector<int64_t*> tmp_m0(P, nullptr);
    std::vector<float*> tmp_ms0(P, nullptr);
    if (full)
        for (int q = 0; q < P; ++q) {
            tmp_m0[q] = (int64_t*)w; w += al((size_t)B * N * 8);
            tmp_ms0[q] = (float*)w; w += al((size_t)B * N * 4);
        }

    // ---- ingest ----
    const std::vector<int>& kd = ctx->kenc_dims;  // [3, c0, ..., D]
    const int c0 = kd[1];
    {
        int64_t tot = 0;
        for (size_t i = 1; i + 1 < kd.size(); ++i) tot += kd[i];
        if (tot > 2 * D) return set_err(ctx, E2EMV_ESHAPE, "keypoint_encoder too wide for the workspace plan");
    }
    IngestParams ip{};
    for (int t = 0; t < T; ++t) {
        ip.kpts[t] = d_kpts[t]; ip.ksc[t] = d_kscores[t]; ip.desc[t] = d_desc[t];
        ip.img_w[t] = fd->img_w[t]; ip.img_h[t] = fd->img_h[t];
    }
    for (int t = 0; t < T; ++t) ip.Nimg[t] = Nt[t];
    ip.B = B; ip.T = T; ip.n_rows = n_rows; ip.D = D; ip.c0 = c0; ip.f16 = fd->desc_dtype == E2EMV_DESC_F16;
    ip.w0 = ctx->kenc_w0; ip.b0 = ctx->kenc_b0; ip.x0 = x; ip.h0 = hid;
    prof_begin(ctx, PS_INGEST, s);
    hipLaunchKernelGGL(ingest_transpose, dim3(n_rows / 64, D / 64, n_img), dim3(256), 0, s, ip);
    hipLaunchKernelGGL(ingest_kenc0, dim3((n_rows + 255) / 256, n_img), dim3(256), 0, s, ip);
    prof_end(ctx, s);
    E2EMV_CHECK_LAUNCH(ctx, "ingest kernels");

    // ---- keypoint encoder layers 1..n through the GEMM; the last adds the descriptors ----
    {
        float* cur = hid;  // [Mtot][c0]
        float* nxt = hid + Mtot * c0;
        const int nl = (int)ctx->kenc_w.size();
        for (int i = 0; i < nl; ++i) {
            const int cin = kd[i + 1], cout = kd[i + 2];
            const bool last = i == nl - 1;
            GemmArgs g;
            g.M = (int)Mtot; g.N = cout; g.K = cin; g.K1 = cin;
            g.A = cur; g.lda = cin;
            g.W = ctx->kenc_w[i]; g.ldw = cin; g.bias = ctx->kenc_b[i];
            g.relu = !last;
            if (last) { g.R = x; g.ldr = D; g.C = x; g.ldc = D; }
            else { g.C = nxt; g.ldc = cout; }
            prof_begin(ctx, PS_GEMM, s);
            // f16x2 mode: the wide layers (fan-in >= 128) on the fp16 x 2 kernel as well - same parity bar, 3x less matrix-core time
            rc = (h2 && ctx->kenc_wh[i]) ? launch_gemm_x3(ctx, g, ctx->kenc_wh[i], cin, s, ctx->kenc_hs[i]) : launch_gemm_nt(ctx, g, s);
            prof_end(ctx, s);
            if (rc) return rc;
            cur = nxt;
            nxt = nxt + Mtot * cout;
        }
    }

    // ---- attentional GNN ----
    uint16_t* xp = (uint16_t*)msg;            // x as scaled planes (msg is free until the conf head)
    uint16_t* attp = (uint16_t*)att;          // attention output as scaled planes
    uint16_t* qkp = (uint16_t*)qkv;           // q | k plain planes [Mtot][2D]
    uint16_t* vtp = qkp + Mtot * 4 * D;       // V^T plain planes [n_img][H][64][n_rows]
    uint16_t* hidp = (uint16_t*)hid;          // hidden as scaled planes [Mtot][2D]
    if (p2) {
        prof_begin(ctx, PS_INGEST, s);
        // (the attention skips query tiles beyond the keypoints of an image: their exponent entries must not be stale)
        E2EMV_HIP(ctx, hipMemsetAsync(e_x, 0, (size_t)(Mtot / 64) * 28 * sizeof(int), s));
        rc = launch_to_planes(ctx, x, Mtot, D, D, xp, s, e_x, a_x);
        prof_end(ctx, s);
        if (rc) return rc;
    }
    // f16x2 kernel generation 5: the row-local GEMMs between two attentions - MLP0, MLP1 of layer l and q | k | v of layer l + 1
    // (final_proj behind the last layer) - chained per 256-row block in ONE launch (gemm_p2c.hip).  A workgroup then walks 6 tiles
    // where the three launches walk ceil(2 rb / CUs) + ceil(rb / CUs) + ceil(3 rb / CUs) (rb = row blocks): chained when that is
    // not more (configs[1]: 256 row blocks on 256 CUs, 6 = 6, and 36 launches fewer per forward; T = 5 shapes with 160 / 320 row
    // blocks keep the three launches).  e2emv_set_f16x2_kernels: 4 = never, 105 = whenever the shapes allow.
    bool chain = false;
    if (p2 && ctx->gemm_chain && Mtot % 256 == 0) {
        const int64_t rb = Mtot / 256, cu = std::max(1, ctx->num_cus);
        auto rounds = [&](int64_t tiles) { return (tiles + cu - 1) / cu; };
        chain = ctx->gemm_chain == 2 || 6 * rounds(rb) <= rounds(2 * rb) + rounds(rb) + rounds(3 * rb);
    }
    bool final_done = false;
    auto qkv_args = [&](const LayerWeights& L) {
        GemmP2Args q;
        q.M = (int)Mtot; q.N = 3 * D; q.K = D; q.K1 = D; q.A = xp; q.lda = D; q.W = L.wp_qkv; q.out_scale = L.hs_qkv; q.bias = L.b_qkv;
        q.out = P2_OUT_QKV; q.Cp = qkp; q.Vt = vtp; q.n_rows = n_rows; q.heads = H;
        q.EA = e_x; q.EC = e_qk; q.EVt = e_vt; q.bias_amax = L.ba_qkv;
        return q;
    };
    for (size_t l = 0; l < ctx->layers.size(); ++l) {
        const LayerWeights& L = ctx->layers[l];
        GemmArgs g;
        if (p2) {
            const bool last = l + 1 == ctx->layers.size();
            if (!chain || l == 0) {  // (chained: the chain of layer l - 1 made this layer's q | k | v)
                const GemmP2Args q = qkv_args(L);
                prof_begin(ctx, PS_GEMM_QKV, s); rc = launch_gemm_p2(ctx, q, s); prof_end(ctx, s);
                if (rc) return rc;
            }
            prof_begin(ctx, PS_ATTN, s);
            rc = launch_attention_p2(ctx, B, T, n_rows, Nt, D, H, qkp, vtp, L.type, attp, s, e_qk, e_vt, e_att);
            prof_end(ctx, s);
            if (rc) return rc;
            // hidden = relu(W0 [x | attention] + b0)   (merge folded into W0, BN folded)
            GemmP2Args m0;
            m0.M = (int)Mtot; m0.N = 2 * D; m0.K = 2 * D; m0.K1 = D; m0.A = xp; m0.lda = D; m0.A2 = attp; m0.lda2 = D;
            m0.W = L.wp_mlp0; m0.out_scale = L.hs_mlp0; m0.bias = L.b_mlp0; m0.relu = true;
            m0.out = P2_OUT_PLANES; m0.Cp = hidp; m0.ldc = 2 * D;
            m0.EA = e_x; m0.EA2 = e_att; m0.EC = e_hid; m0.bias_amax = L.ba_mlp0;
            // x += W1 hidden + b1; the last layer hands x to final_proj as fp32
            GemmP2Args m1;
            m1.M = (int)Mtot; m1.N = D; m1.K = 2 * D; m1.K1 = 2 * D; m1.A = hidp; m1.lda = 2 * D;
            m1.W = L.wp_mlp1; m1.out_scale = L.hs_mlp1; m1.bias = L.b_mlp1; m1.Rp = xp; m1.ldr = D;
            m1.EA = e_hid; m1.ER = e_x; m1.AR = a_x; m1.bias_amax = L.ba_mlp1;
            if (last && !ctx->wp_final) { m1.out = P2_OUT_F32; m1.C32 = x; m1.ldc = D; }  // (final_proj then runs on the fp32-input kernel)
            else { m1.out = P2_OUT_PLANES; m1.Cp = xp; m1.ldc = D; m1.EC = e_x; m1.AC = a_x; }
            if (chain && (!last || ctx->wp_final)) {
                GemmP2Args st[3] = {m0, m1, GemmP2Args()};
                if (!last) {
                    st[2] = qkv_args(ctx->layers[l + 1]);
                } else {
                    GemmP2Args& q = st[2];
                    q.M = (int)Mtot; q.N = D; q.K = D; q.K1 = D; q.A = xp; q.lda = D; q.W = ctx->wp_final; q.out_scale = ctx->hs_final; q.bias = ctx->b_final;
                    q.out = P2_OUT_F32; q.C32 = att; q.ldc = D; q.EA = e_x; q.bias_amax = ctx->ba_final;  // (att = mdesc below)
                }
                // MLP1's K steps 8 .. 15 read the hidden columns MLP0's SECOND tile stores right in front of it; q | k | v and
                // final_proj read x_new from their first K step on
                const int dep[3] = {P2_CHAIN_INDEP, (2 * D - 256) / 32, 0};
                prof_begin(ctx, PS_GEMM_CHAIN, s); rc = launch_gemm_p2_chain(ctx, st, dep, 3, s); prof_end(ctx, s);
                if (rc == E2EMV_OK) {
                    if (last) final_done = true;
                    continue;
                }
                // the chain launcher validates before it launches: a shape it does not take (ESHAPE / EINVAL) leaves nothing enqueued -
                // this layer and the rest run a launch per GEMM (the next layer then makes its own q | k | v: `!chain` above)
                if (rc != E2EMV_ESHAPE && rc != E2EMV_EINVAL) return rc;
                chain = false;
                ctx->err.clear();
            }
            prof_begin(ctx, PS_GEMM_MLP0, s); rc = launch_gemm_p2(ctx, m0, s); prof_end(ctx, s);
            if (rc) return rc;
            prof_begin(ctx, PS_GEMM_MLP1, s); rc = launch_gemm_p2(ctx, m1, s); prof_end(ctx, s);
            if (rc) return rc;
            continue;
        }
        if (b3) {
            // q|k|v on the split-operand GEMM with a plain fp32 output; the attention kernel splits Q / K / V^T into
            // bf16 planes on the way in (E2EMV_B3_PLANES=1 selects the first-generation hand-over: fp32-pipe GEMM whose
            // epilogue emits the planes, 3.2x the bytes)
            if (h2 || !ctx->b3_planes) {
                g = GemmArgs();
                g.M = (int)Mtot; g.N = 3 * D; g.K = D; g.K1 = D; g.A = x; g.lda = D; g.bias = L.b_qkv; g.C = qkv; g.ldc = 3 * D;
                prof_begin(ctx, PS_GEMM, s); rc = h2 ? launch_gemm_x3(ctx, g, L.wh_qkv, D, s, L.hs_qkv) : launch_gemm_x3(ctx, g, L.w3_qkv, D, s); prof_end(ctx, s);
                if (rc) return rc;
                prof_begin(ctx, PS_ATTN, s);
                rc = launch_attention3f(ctx, B, T, n_rows, Nt, D, H, qkv, L.type, att, s, h2);
                prof_end(ctx, s);
                if (rc) return rc;
            } else {
            g = GemmArgs();
            g.M = (int)Mtot; g.N = 3 * D; g.K = D; g.K1 = D; g.A = x; g.lda = D; g.W = L.w_qkv; g.ldw = D; g.bias = L.b_qkv;
            g.C3 = qk3; g.ldc3 = 2 * D; g.Vt = vt3; g.vt_n0 = 2 * D; g.n_rows = n_rows;
            g.q_cols = D; g.q_scale = 0.125f * 1.4426950408889634f;
            prof_begin(ctx, PS_GEMM, s); rc = launch_gemm_nt(ctx, g, s); prof_end(ctx, s);
            if (rc) return rc;
            prof_begin(ctx, PS_ATTN, s);
            rc = launch_attention3(ctx, B, T, n_rows, Nt, D, H, qk3, vt3, L.type, nullptr, att, s);
            prof_end(ctx, s);
            if (rc) return rc;
            }
        } else {
        // q|k|v = x Wqkv^T + b
        g = GemmArgs();
        g.M = (int)Mtot; g.N = 3 * D; g.K = D; g.K1 = D; g.A = x; g.lda = D; g.W = L.w_qkv; g.ldw = D; g.bias = L.b_qkv;
        g.C = qkv; g.ldc = 3 * D;
        prof_begin(ctx, PS_GEMM, s); rc = launch_gemm_nt(ctx, g, s); prof_end(ctx, s);
        if (rc) return rc;
        prof_begin(ctx, PS_ATTN, s);
        rc = launch_attention(ctx, B, T, n_rows, Nt, D, H, qkv, L.type, att, s);
        prof_end(ctx, s);
        if (rc) return rc;
        }
        const float* second = att;  // MLP0's second K segment: attention output (merge folded into W0)
        if (!ctx->fuse_merge) {
            // message = merge(attention)
            g = GemmArgs();
            g.M = (int)Mtot; g.N = D; g.K = D; g.K1 = D; g.A = att; g.lda = D; g.W = L.w_merge; g.ldw = D; g.bias = L.b_merge;
            g.C = msg; g.ldc = D;
            prof_begin(ctx, PS_GEMM, s); rc = launch_gemm_nt(ctx, g, s); prof_end(ctx, s);
            if (rc) return rc;
            second = msg;
        }
        // hidden = relu(BN(W0 [x | message] + b0))   (concat never materialised: two K segments)
        g = GemmArgs();
        g.M = (int)Mtot; g.N = 2 * D; g.K = 2 * D; g.K1 = D; g.A = x; g.lda = D; g.A2 = second; g.lda2 = D;
        g.W = L.w_mlp0; g.ldw = 2 * D; g.bias = L.b_mlp0; g.relu = true; g.C = hid; g.ldc = 2 * D;
        // bf16x3 mode: the two MLP GEMMs (2/3 of the layer's GEMM flops) run on the bf16 pipe with split operands
        prof_begin(ctx, PS_GEMM, s);
        rc = h2 ? launch_gemm_x3(ctx, g, L.wh_mlp0, 2 * D, s, L.hs_mlp0) : b3 ? launch_gemm_x3(ctx, g, L.w3_mlp0, 2 * D, s) : launch_gemm_nt(ctx, g, s);
        prof_end(ctx, s);
        if (rc) return rc;
        // x += W1 hidden + b1
        g = GemmArgs();
        g.M = (int)Mtot; g.N = D; g.K = 2 * D; g.K1 = 2 * D; g.A = hid; g.lda = 2 * D; g.W = L.w_mlp1; g.ldw = 2 * D;
        g.bias = L.b_mlp1; g.R = x; g.ldr = D; g.C = x; g.ldc = D;
        prof_begin(ctx, PS_GEMM, s);
        rc = h2 ? launch_gemm_x3(ctx, g, L.wh_mlp1, 2 * D, s, L.hs_mlp1) : b3 ? launch_gemm_x3(ctx, g, L.w3_mlp1, 2 * D, s) : launch_gemm_nt(ctx, g, s);
        prof_end(ctx, s);
        if (rc) return rc;
    }

    // ---- final projection ----
    float* mdesc = att;
    if (final_done) {
        // (the last layer's chain wrote mdesc)
    } else if (p2 && ctx->wp_final) {  // x arrives as planes with their tile exponents: any magnitude fp32 holds is fine
        GemmP2Args q;
        q.M = (int)Mtot; q.N = D; q.K = D; q.K1 = D; q.A = xp; q.lda = D; q.W = ctx->wp_final; q.out_scale = ctx->hs_final; q.bias = ctx->b_final;
        q.out = P2_OUT_F32; q.C32 = mdesc; q.ldc = D; q.EA = e_x; q.bias_amax = ctx->ba_final;
        prof_begin(ctx, PS_GEMM, s); rc = launch_gemm_p2(ctx, q, s); prof_end(ctx, s);
        if (rc) return rc;
    } else {
        GemmArgs g;
        g.M = (int)Mtot; g.N = D; g.K = D; g.K1 = D; g.A = x; g.lda = D; g.W = ctx->w_final; g.ldw = D; g.bias = ctx->b_final;
        g.C = mdesc; g.ldc = D;
        prof_begin(ctx, PS_GEMM, s); rc = h2 ? launch_gemm_x3(ctx, g, ctx->wh_final, D, s, ctx->hs_final) : launch_gemm_nt(ctx, g, s); prof_end(ctx, s);
        if (rc) return rc;
    }

    ctx->last_mdesc = mdesc; ctx->md_imgs = B * T; ctx->md_rows = n_rows; ctx->md_n = N; ctx->md_dim = D;  // (e2emv_get_descriptors)

    // ---- all pairs: scores -> Sinkhorn -> matches; then the conf head per pair.  With equal keypoint counts all
    // P*B problems go through ONE Sinkhorn batch; a ragged tuple runs one batch per pair (M = N_i, N = N_j).
    const int64_t tuple_stride = (int64_t)T * n_rows * D;
    const int64_t pair_stride = (int64_t)B * N * ldS;
    SinkhornOut so;
    so.n_groups = P;
    so.group_batch = B;
    std::vector<bool> want_conf(P, false);
    std::vector<int64_t*> pm0(P, nullptr);
    std::vector<float*> pms0(P, nullptr);
    int pidx = 0;
    for (int j = 0; j < T; ++j)
        for (int i = 0; i < j; ++i, ++pidx) {
            const int Ni = Nt[i], Nj = Nt[j], ldj = round_up(Nj, 4);
            GemmArgs g;
            g.batch = B; g.M = Ni; g.N = Nj; g.K = D; g.K1 = D;
            g.A = mdesc + (int64_t)i * n_rows * D; g.lda = D; g.sA = tuple_stride;
            g.W = mdesc + (int64_t)j * n_rows * D; g.ldw = D; g.sW = tuple_stride;
            g.C = S + pidx * pair_stride; g.ldc = ldj; g.sC = (int64_t)Ni * ldj;
            g.scale = 1.0f / sqrtf((float)D);
            prof_begin(ctx, PS_SCORE, s); rc = launch_gemm_nt(ctx, g, s); prof_end(ctx, s);
            if (rc) return rc;
            want_conf[pidx] = full && d_conf && d_conf[pidx];
            SinkhornOut one;  // outputs of this pair
            one.logZ[0] = d_logZ ? d_logZ[pidx] : nullptr;
            if (full) {
                one.m0[0] = (d_m0 && d_m0[pidx]) ? d_m0[pidx] : (want_conf[pidx] ? tmp_m0[pidx] : nullptr);
                one.m1[0] = d_m1 ? d_m1[pidx] : nullptr;
                one.ms0[0] = (d_ms0 && d_ms0[pidx]) ? d_ms0[pidx] : (want_conf[pidx] ? tmp_ms0[pidx] : nullptr);
                one.ms1[0] = d_ms1 ? d_ms1[pidx] : nullptr;
            }
            pm0[pidx] = one.m0[0];
            pms0[pidx] = one.ms0[0];
            so.logZ[pidx] = one.logZ[0]; so.m0[pidx] = one.m0[0]; so.m1[pidx] = one.m1[0];
            so.ms0[pidx] = one.ms0[0]; so.ms1[pidx] = one.ms1[0];
            if (!uniform) {
                one.n_groups = 1;
                one.group_batch = B;
                prof_begin(ctx, PS_SINKHORN, s);
                rc = launch_sinkhorn(ctx, B, Ni, Nj, S + pidx * pair_stride, ldj, ctx->bin_score, fd->sinkhorn_iters,
                                     fd->match_threshold, one, skws, s);
                prof_end(ctx, s);
                if (rc) return rc;
            }
        }
    if (uniform) {
        prof_begin(ctx, PS_SINKHORN, s);
        rc = launch_sinkhorn(ctx, P * B, N, N, S, ldS, ctx->bin_score, fd->sinkhorn_iters, fd->match_threshold, so, skws, s);
        prof_end(ctx, s);
        if (rc) return rc;
    }
    pidx = 0;
    for (int j = 0; j < T; ++j)
        for (int i = 0; i < j; ++i, ++pidx) {
            if (!want_conf[pidx]) continue;
            const int Ni = Nt[i];
            const bool use_mlp = ctx->model.conf_mlp != 0;
            float* gathered = msg;          // [B][n_rows][D] ([B][n_rows][2D] in the f16x2 modes)
            float* chid = hid;              // [B][n_rows][D]
            if (use_mlp && p2 && ctx->wp_conf0) {
                // plane kernels: [mdesc_i | mdesc_j(match)] -> planes with tile exponents -> conf_mlp.0 (+ BN, ReLU) on gemm_p2
                // (one pass: the gather is resolved in the source address of the plane conversion - p2_tools.hip)
                prof_begin(ctx, PS_CONF, s);
                rc = launch_conf_gather_planes(ctx, mdesc + (int64_t)i * n_rows * D, mdesc + (int64_t)j * n_rows * D, tuple_stride, pm0[pidx], Ni, n_rows, B, D, qkp,
                                               e_hid, s);
                prof_end(ctx, s);
                if (rc) return rc;
                GemmP2Args q;
                q.M = B * n_rows; q.N = D; q.K = 2 * D; q.K1 = 2 * D; q.A = qkp; q.lda = 2 * D; q.W = ctx->wp_conf0; q.out_scale = ctx->hs_conf0;
                q.bias = ctx->b_conf0; q.relu = true; q.out = P2_OUT_F32; q.C32 = chid; q.ldc = D; q.EA = e_hid; q.bias_amax = ctx->ba_conf0;
                prof_begin(ctx, PS_GEMM, s); rc = launch_gemm_p2(ctx, q, s); prof_end(ctx, s);
                if (rc) return rc;
            } else if (use_mlp && h2 && ctx->wh_conf0) {
                prof_begin(ctx, PS_CONF, s);
                hipLaunchKernelGGL(conf_gather2_kernel, dim3((n_rows + 3) / 4, B), dim3(256), 0, s, Ni, n_rows, D, mdesc + (int64_t)i * n_rows * D,
                                   mdesc + (int64_t)j * n_rows * D, tuple_stride, pm0[pidx], gathered);
                prof_end(ctx, s);
                GemmArgs c;
                c.M = B * n_rows; c.N = D; c.K = 2 * D; c.K1 = 2 * D; c.A = gathered; c.lda = 2 * D;
                c.bias = ctx->b_conf0; c.relu = true; c.C = chid; c.ldc = D;
                prof_begin(ctx, PS_GEMM, s); rc = launch_gemm_x3(ctx, c, ctx->wh_conf0, 2 * D, s, ctx->hs_conf0); prof_end(ctx, s);
                if (rc) return rc;
            } else if (use_mlp) {
                prof_begin(ctx, PS_CONF, s);
                hipLaunchKernelGGL(conf_gather_kernel, dim3((n_rows + 3) / 4, B), dim3(256), 0, s, Ni, n_rows, D,
                                   mdesc + (int64_t)j * n_rows * D, tuple_stride, pm0[pidx], gathered);
                prof_end(ctx, s);
                GemmArgs c;
                c.batch = B; c.M = n_rows; c.N = D; c.K = 2 * D; c.K1 = D;
                c.A = mdesc + (int64_t)i * n_rows * D; c.lda = D; c.sA = tuple_stride;
                c.A2 = gathered; c.lda2 = D; c.sA2 = (int64_t)n_rows * D;
                c.W = ctx->w_conf0; c.ldw = 2 * D; c.bias = ctx->b_conf0; c.relu = true;
                c.C = chid; c.ldc = D; c.sC = (int64_t)n_rows * D;
                prof_begin(ctx, PS_GEMM, s); rc = launch_gemm_nt(ctx, c, s); prof_end(ctx, s);
                if (rc) return rc;
            }
            prof_begin(ctx, PS_CONF, s);
            hipLaunchKernelGGL(conf_final_kernel, dim3((Ni + 3) / 4, B), dim3(256), 0, s, Ni, n_rows, D, chid, ctx->w_conf1,
                               ctx->b_conf1, pm0[pidx], pms0[pidx], use_mlp ? 1 : 0, d_conf[pidx]);
            prof_end(ctx, s);
            E2EMV_CHECK_LAUNCH(ctx, "conf kernels");
        }
    return E2EMV_OK;
}

extern "C" int e2emv_matcher_forward(e2emv_ctx* ctx, const e2emv_forward_desc* fd, const float* const* d_kpts,
                                     const float* const* d_kscores, const void* const* d_desc, float* const* d_logZ,
                                     int64_t* const* d_matches0, int64_t* const* d_matches1, float* const* d_mscores0,
                                     float* const* d_mscores1, float* const* d_conf, void* stream) {
    if (!ctx || !fd || !d_kpts || !d_kscores || !d_desc) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (!ctx->committed) return set_err(ctx, E2EMV_ESTATE, "matcher_forward: weights not committed");
    const int B = fd->batch, T = fd->tuple_size;
    if (B <= 0 || T < 2 || T > E2EMV_MAX_TUPLE) return set_err(ctx, E2EMV_ESHAPE, "matcher_forward: batch=%d tuple_size=%d", B, T);
    for (int t = 0; t < T; ++t) {
        const int n = fd->n_kpts_img[t] > 0 ? fd->n_kpts_img[t] : fd->n_kpts;
        if (n <= 0 || n > 2048) return set_err(ctx, E2EMV_ESHAPE, "matcher_forward: image %d has n_kpts=%d, not in [1, 2048]", t, n);
    }
    if (fd->desc_dtype != E2EMV_DESC_F32 && fd->desc_dtype != E2EMV_DESC_F16) return set_err(ctx, E2EMV_EINVAL, "bad desc_dtype");
    if (fd->sinkhorn_iters < 0) return set_err(ctx, E2EMV_EINVAL, "negative sinkhorn_iters");
    for (int t = 0; t < T; ++t)
        if (!d_kpts[t] || !d_kscores[t] || !d_desc[t]) return set_err(ctx, E2EMV_EINVAL, "matcher_forward: null input for image %d", t);
    (void)hipSetDevice(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    if (T == 2 || (fd->flags & E2EMV_FLAG_MULTI_FRAME))
        return forward_joint(ctx, fd, d_kpts, d_kscores, d_desc, d_logZ, d_matches0, d_matches1, d_mscores0, d_mscores1, d_conf, s);
    // pairwise mode on a tuple: every pair goes through the 2-view network on its own
    int pidx = 0;
    for (int j = 0; j < T; ++j)
        for (int i = 0; i < j; ++i, ++pidx) {
            e2emv_forward_desc f2 = *fd;
            f2.tuple_size = 2;
            f2.img_w[0] = fd->img_w[i]; f2.img_h[0] = fd->img_h[i];
            f2.img_w[1] = fd->img_w[j]; f2.img_h[1] = fd->img_h[j];
            f2.n_kpts_img[0] = fd->n_kpts_img[i]; f2.n_kpts_img[1] = fd->n_kpts_img[j];
            for (int t = 2; t < E2EMV_MAX_TUPLE; ++t) f2.n_kpts_img[t] = 0;
            const float* kp[2] = {d_kpts[i], d_kpts[j]};
            const float* ks[2] = {d_kscores[i], d_kscores[j]};
            const void* de[2] = {d_desc[i], d_desc[j]};
            float* lz[1] = {d_logZ ? d_logZ[pidx] : nullptr};
            int64_t* m0[1] = {d_matches0 ? d_matches0[pidx] : nullptr};
            int64_t* m1[1] = {d_matches1 ? d_matches1[pidx] : nullptr};
            float* s0[1] = {d_mscores0 ? d_mscores0[pidx] : nullptr};
            float* s1[1] = {d_mscores1 ? d_mscores1[pidx] : nullptr};
            float* cf[1] = {d_conf ? d_conf[pidx] : nullptr};
            int rc = forward_joint(ctx, &f2, kp, ks, de, lz, m0, m1, s0, s1, cf, s);
            if (rc) return rc;
        }
    return E2EMV_OK;
}

// The matched descriptors (final_proj output, upstream's mdesc0 / mdesc1) of the last e2emv_matcher_forward call: what the
// score matrix was built from.  An audit output - parity tests compare it with the oracle's descriptors, the quantity the GNN
// arithmetic modes differ in (logZ is dominated by the fp32 Sinkhorn).
extern "C" int e2emv_get_descriptors(e2emv_ctx* ctx, float* d_out, int64_t capacity, int* n_img, int* n_kpts, int* dim, void* stream) {
    if (!ctx) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    // (ws_reserve clears the pointer: it is valid only until the next call that uses the workspace)
    if (!ctx->last_mdesc) return set_err(ctx, E2EMV_ESTATE, "get_descriptors: no matcher_forward on this context since the workspace was last reused");
    if (n_img) *n_img = ctx->md_imgs;
    if (n_kpts) *n_kpts = ctx->md_n;
    if (dim) *dim = ctx->md_dim;
    if (!d_out) return E2EMV_OK;  // (size query)
    const int64_t need = (int64_t)ctx->md_imgs * ctx->md_n * ctx->md_dim;
    if (capacity < need) return set_err(ctx, E2EMV_ESHAPE, "get_descriptors: buffer of %lld floats, %lld needed", (long long)capacity, (long long)need);
    const size_t row = (size_t)ctx->md_dim * sizeof(float);
    E2EMV_HIP(ctx, hipMemcpy2DAsync(d_out, (size_t)ctx->md_n * row, ctx->last_mdesc, (size_t)ctx->md_rows * row, (size_t)ctx->md_n * row,
                                    (size_t)ctx->md_imgs, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return E2EMV_OK;
}
