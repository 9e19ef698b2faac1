// Ground-truth match targets from depth + pose, and the match loss (SURVEY.md 8(f) "next" row 2): the other
// consumer of the matcher's log-assignment `scores_i_j` (helpers.run_matcher, helpers.py:243-253) and the
// quantity the reference's only explicit collective carries (validation loss, train.py:102-106).
//
// Restates helpers.py: transform_kpts :114-118, compute_gt_matches_of_image_pair :121-203, set_weight :205-213,
// compute_match_loss :228-241.  The reference materialises the B x N x N reprojection-error tensor (twice, plus
// expand temporaries) and runs ~40 torch ops on it; here the error of a keypoint pair is recomputed on the fly
// (two sqrt's) inside wave-per-row / wave-per-column arg-min kernels, nothing N x N ever touches HBM.
// Reference quirks kept: keypoints are truncated to integer pixels and those integers enter the errors; first index
// wins arg-min ties (NaN counts as minimal, as in torch); the dustbin slot N keeps index -1 and gets the un-match
// weight; index -1 in the loss addresses the last (dustbin) column.
#include "common.h"

namespace e2emv {

struct GtParams {
    int B, N, H, W;
    const float* k0;      // [B][N][2]
    const float* k1;
    const float* K0;      // [B][4][4]
    const float* K1;
    const float* T01;     // [B][4][4]
    const float* depth0;  // [B][H][W]
    const float* depth1;
    float max_matched, min_unmatched;
    // workspace (per direction s = 0: image 0 -> 1, s = 1: image 1 -> 0), all [2][B][N]
    float* ki;     // [2][B][N][2] truncated integer keypoints (as float)
    float* d;      // depth at the keypoint
    float* kto;    // [2][B][N][2] keypoint reprojected into the other image
    float* dto;    // its depth there
    float* emin;   // [2][B][N] min error along the row (s=0) / column (s=1)
    int* amin;     // [2][B][N] arg-min
    int64_t* idx;  // out [B][2][N+1]
    float* w;      // out [B][2][N+1]
};

__device__ bool inv4(const double* m, double* inv) {  // Gauss-Jordan with partial pivoting
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = m[i * 4 + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r)
            if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
        if (!(fabs(a[p][c]) > 0.0)) return false;
        if (p != c)
            for (int k = 0; k < 8; ++k) { const double t = a[c][k]; a[c][k] = a[p][k]; a[p][k] = t; }
        const double d = 1.0 / a[c][c];
        for (int k = 0; k < 8; ++k) a[c][k] *= d;
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const double f = a[r][c];
                for (int k = 0; k < 8; ++k) a[r][k] -= f * a[c][k];
            }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) inv[i * 4 + j] = a[i][4 + j];
    return true;
}

__device__ void mul4(const double* a, const double* b, double* c) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
            c[i * 4 + j] = s;
        }
}

// transform_kpts for both directions: grid (ceil(N/256), B, 2)
__global__ __launch_bounds__(256) void gt_transform(GtParams p) {
    __shared__ double M[16];
    const int b = blockIdx.y, s = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x == 0) {
        double Ka[16], Kb[16], T[16], Ti[16], Kai[16], tmp[16];
        const float* Ks = (s == 0 ? p.K0 : p.K1) + (int64_t)b * 16;
        const float* Kt = (s == 0 ? p.K1 : p.K0) + (int64_t)b * 16;
        for (int k = 0; k < 16; ++k) { Ka[k] = Ks[k]; Kb[k] = Kt[k]; T[k] = p.T01[(int64_t)b * 16 + k]; }
        if (s == 1) { inv4(T, Ti); for (int k = 0; k < 16; ++k) T[k] = Ti[k]; }
        inv4(Ka, Kai);
        mul4(Kb, T, tmp);
        mul4(tmp, Kai, M);  // K_target T K_source^-1
    }
    __syncthreads();
    if (i >= p.N) return;
    const float* k = (s == 0 ? p.k0 : p.k1) + ((int64_t)b * p.N + i) * 2;
    const float* depth = (s == 0 ? p.depth0 : p.depth1) + (int64_t)b * p.H * p.W;
    const int xi = (int)k[0], yi = (int)k[1];  // .long(): truncation toward zero
    const int xc = min(max(xi, 0), p.W - 1), yc = min(max(yi, 0), p.H - 1);
    const float d = depth[(int64_t)yc * p.W + xc];
    const double x = (double)xi * d, y = (double)yi * d;
    const double px = M[0] * x + M[1] * y + M[2] * d + M[3];
    const double py = M[4] * x + M[5] * y + M[6] * d + M[7];
    const double pz = M[8] * x + M[9] * y + M[10] * d + M[11];
    const int64_t o = ((int64_t)s * p.B + b) * p.N + i;
    p.ki[o * 2] = (float)xi; p.ki[o * 2 + 1] = (float)yi;
    p.d[o] = d;
    p.kto[o * 2] = (float)(px / pz); p.kto[o * 2 + 1] = (float)(py / pz);
    p.dto[o] = (float)pz;
}

// mean bidirectional reprojection error of (kpt i of image 0, kpt j of image 1)  (:136-138)
__device__ __forceinline__ float pair_err(float k0x, float k0y, float t01x, float t01y, float k1x, float k1y, float t10x, float t10y) {
    const float ax = t10x - k0x, ay = t10y - k0y;   // kpts1to0[j] - kpts0[i]
    const float bx = t01x - k1x, by = t01y - k1y;   // kpts0to1[i] - kpts1[j]
    return (sqrtf(ax * ax + ay * ay) + sqrtf(bx * bx + by * by)) / 2.0f;
}

// arg-min along rows (s = 0: for each kpt0 the closest kpt1) and columns (s = 1); one wave per row/column
__global__ __launch_bounds__(256) void gt_argmin(GtParams p) {
    const int b = blockIdx.y, s = blockIdx.z;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= p.N) return;
    const int64_t o0 = (int64_t)b * p.N, o1 = ((int64_t)p.B + b) * p.N;
    float best = INFINITY;
    int bj = 0x7fffffff;
    bool bnan = false;
    if (s == 0) {
        const float k0x = p.ki[(o0 + r) * 2], k0y = p.ki[(o0 + r) * 2 + 1], tx = p.kto[(o0 + r) * 2], ty = p.kto[(o0 + r) * 2 + 1];
        for (int j = lane; j < p.N; j += 64) {
            const float e = pair_err(k0x, k0y, tx, ty, p.ki[(o1 + j) * 2], p.ki[(o1 + j) * 2 + 1], p.kto[(o1 + j) * 2], p.kto[(o1 + j) * 2 + 1]);
            const bool en = e != e;
            if (!bnan && (en || e < best)) { best = e; bj = j; bnan = en; }
        }
    } else {
        const float k1x = p.ki[(o1 + r) * 2], k1y = p.ki[(o1 + r) * 2 + 1], tx = p.kto[(o1 + r) * 2], ty = p.kto[(o1 + r) * 2 + 1];
        for (int i = lane; i < p.N; i += 64) {
            const float e = pair_err(p.ki[(o0 + i) * 2], p.ki[(o0 + i) * 2 + 1], p.kto[(o0 + i) * 2], p.kto[(o0 + i) * 2 + 1], k1x, k1y, tx, ty);
            const bool en = e != e;
            if (!bnan && (en || e < best)) { best = e; bj = i; bnan = en; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oj = __shfl_xor(bj, off);
        const bool on = __shfl_xor((int)bnan, off) != 0;
        bool take;
        if (on != bnan) take = on;                     // NaN is "minimal" (torch.argmin semantics)
        else if (on) take = oj < bj;                   // both NaN: first index
        else take = ob < best || (ob == best && oj < bj);
        if (take) { best = ob; bj = oj; bnan = on; }
    }
    if (lane == 0) {
        const int64_t o = ((int64_t)s * p.B + b) * p.N + r;
        p.emin[o] = best;
        p.amin[o] = bj == 0x7fffffff ? 0 : bj;
    }
}

// match / drop logic and class-balancing weights (:147-203); one workgroup per pair
__global__ __launch_bounds__(256) void gt_finalize(GtParams p) {
    __shared__ int scount[2];
    __shared__ float sw[2];
    const int b = blockIdx.x, tid = threadIdx.x, N = p.N;
    const int64_t o0 = (int64_t)b * N, o1 = ((int64_t)p.B + b) * N;
    int64_t* idx0 = p.idx + (int64_t)b * 2 * (N + 1);
    int64_t* idx1 = idx0 + (N + 1);
    float* w0 = p.w + (int64_t)b * 2 * (N + 1);
    float* w1 = w0 + (N + 1);
    if (tid < 2) scount[tid] = 0;
    for (int j = tid; j <= N; j += 256) { idx0[j] = -1; idx1[j] = -1; w0[j] = 0.f; w1[j] = 0.f; }
    __syncthreads();
    int mc = 0, dc = 0;
    for (int i = tid; i < N; i += 256) {
        const int i1 = p.amin[o0 + i];
        const float e = p.emin[o0 + i];
        const float d0 = p.d[o0 + i], md1 = p.d[o1 + i1];
        const bool vd0 = d0 > 1e-6f, vd1 = md1 > 1e-6f;
        bool match = (p.amin[o1 + i1] == i) && (e <= p.max_matched) && vd0 && vd1;
        if (match) {
            const float r01 = fabsf(p.dto[o0 + i] - md1) / md1;
            const float r10 = fabsf(p.dto[o1 + i1] - d0) / d0;
            match = (r01 < 0.1f) && (r10 < 0.1f);
        }
        if (match) {
            idx0[i] = i1;
            idx1[i1] = i;  // unique: mutual nearest neighbours
            ++mc;
        } else if (!vd0 || !vd1 || e <= p.min_unmatched) {
            w0[i] = -1.f;
            ++dc;
        }
    }
    __syncthreads();  // idx1 complete
    for (int j = tid; j < N; j += 256) {
        if (idx1[j] != -1) continue;
        const int i0 = p.amin[o1 + j];
        const bool invalid = !(p.d[o0 + i0] > 1e-6f) || !(p.d[o1 + j] > 1e-6f);
        if (invalid || p.emin[o1 + j] <= p.min_unmatched) { w1[j] = -1.f; ++dc; }
    }
    atomicAdd(&scount[0], mc);
    atomicAdd(&scount[1], dc);
    __syncthreads();
    if (tid == 0) {
        float mw = 2.f * (float)scount[0] / (2.f * (float)N - (float)scount[1]);
        float uw = 0.5f / (1.f - mw);
        mw = 0.5f / mw;
        if (!(isfinite(mw) && isfinite(uw))) { mw = 0.f; uw = 0.f; }
        sw[0] = mw; sw[1] = uw;
    }
    __syncthreads();
    const float mw = sw[0], uw = sw[1];
    for (int j = tid; j <= N; j += 256) {  // set_weight, including the dustbin slot (index -1 -> un-match weight)
        w0[j] = (w0[j] == -1.f) ? 0.f : (idx0[j] == -1 ? uw : mw);
        w1[j] = (w1[j] == -1.f) ? 0.f : (idx1[j] == -1 ? uw : mw);
    }
}

// compute_match_loss: one workgroup per batch element -> partial sums, then one thread folds them
__global__ __launch_bounds__(256) void match_loss_kernel(int B, int N, const float* logp, const int64_t* idx, const float* w,
                                                         double* partial) {
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x, ft = N + 1;
    const float* lp = logp + (int64_t)b * ft * ft;
    const int64_t* i0 = idx + (int64_t)b * 2 * ft;
    const int64_t* i1 = i0 + ft;
    const float* w0 = w + (int64_t)b * 2 * ft;
    const float* w1 = w0 + ft;
    double acc = 0.0;
    for (int r = tid; r < ft; r += 256) {
        int64_t c0 = i0[r], c1 = i1[r];
        if (c0 < 0) c0 += ft;  // python negative index: -1 -> dustbin column
        if (c1 < 0) c1 += ft;
        acc -= (double)lp[(int64_t)r * ft + c0] * (double)w0[r];   // l0 = -log_p[b, r, idx0[r]]
        acc -= (double)lp[(int64_t)c1 * ft + r] * (double)w1[r];   // l1 = -log_p^T[b, r, idx1[r]] = -log_p[b, idx1[r], r]
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partial[b] = red[0] + red[1] + red[2] + red[3];
}
__global__ void match_loss_fold(int B, const double* partial, float* loss) {
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += partial[b];
    loss[0] = (float)(s / B);
}

}  // namespace e2emv

using namespace e2emv;

extern "C" int e2emv_gt_matches(e2emv_ctx* ctx, int B, int N, const float* d_kpts0, const float* d_kpts1, const float* d_K0,
                                const float* d_K1, const float* d_T0to1, const float* d_depth0, const float* d_depth1, int H,
                                int W, float max_matched_reproj_err, float min_unmatched_reproj_err, int64_t* d_indices,
                                float* d_weights, void* stream) {
    if (!ctx || !d_kpts0 || !d_kpts1 || !d_K0 || !d_K1 || !d_T0to1 || !d_depth0 || !d_depth1 || !d_indices || !d_weights)
        return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (B <= 0 || N <= 0 || H <= 0 || W <= 0) return set_err(ctx, E2EMV_ESHAPE, "gt_matches: B=%d N=%d H=%d W=%d", B, N, H, W);
    hipStream_t s = (hipStream_t)stream;
    auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
    const size_t n2 = (size_t)2 * B * N;
    const size_t need = 2 * al(n2 * 2 * 4) + 4 * al(n2 * 4);
    int rc = ws_reserve(ctx, need);
    if (rc) return rc;
    GtParams p{};
    p.B = B; p.N = N; p.H = H; p.W = W;
    p.k0 = d_kpts0; p.k1 = d_kpts1; p.K0 = d_K0; p.K1 = d_K1; p.T01 = d_T0to1; p.depth0 = d_depth0; p.depth1 = d_depth1;
    p.max_matched = max_matched_reproj_err; p.min_unmatched = min_unmatched_reproj_err;
    char* w = ctx->d_ws;
    p.ki = (float*)w; w += al(n2 * 2 * 4);
    p.kto = (float*)w; w += al(n2 * 2 * 4);
    p.d = (float*)w; w += al(n2 * 4);
    p.dto = (float*)w; w += al(n2 * 4);
    p.emin = (float*)w; w += al(n2 * 4);
    p.amin = (int*)w;
    p.idx = d_indices; p.w = d_weights;
    prof_begin(ctx, PS_MISC, s);
    hipLaunchKernelGGL(gt_transform, dim3((N + 255) / 256, B, 2), dim3(256), 0, s, p);
    hipLaunchKernelGGL(gt_argmin, dim3((N + 3) / 4, B, 2), dim3(256), 0, s, p);
    hipLaunchKernelGGL(gt_finalize, dim3(B), dim3(256), 0, s, p);
    prof_end(ctx, s);
    E2EMV_CHECK_LAUNCH(ctx, "gt_matches kernels");
    return E2EMV_OK;
}

extern "C" int e2emv_match_loss(e2emv_ctx* ctx, int B, int N, const float* d_logZ, const int64_t* d_indices,
                                const float* d_weights, float* d_loss, void* stream) {
    if (!ctx || !d_logZ || !d_indices || !d_weights || !d_loss) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (B <= 0 || N <= 0) return set_err(ctx, E2EMV_ESHAPE, "match_loss: B=%d N=%d", B, N);
    hipStream_t s = (hipStream_t)stream;
    int rc = ws_reserve(ctx, (size_t)B * 8 + 256);
    if (rc) return rc;
    double* partial = (double*)ctx->d_ws;
    prof_begin(ctx, PS_MISC, s);
    hipLaunchKernelGGL(match_loss_kernel, dim3(B), dim3(256), 0, s, B, N, d_logZ, d_indices, d_weights, partial);
    hipLaunchKernelGGL(match_loss_fold, dim3(1), dim3(1), 0, s, B, partial, d_loss);
    prof_end(ctx, s);
    E2EMV_CHECK_LAUNCH(ctx, "match_loss kernels");
    return E2EMV_OK;
}
