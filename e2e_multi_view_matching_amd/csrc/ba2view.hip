// Two-view Levenberg-Marquardt bundle adjustment (SURVEY.md 8(f) "next" row 1): the refinement the
// reference runs right after the weighted 8-point solve in its default eval mode (`w8pt_ba`,
// eval_pairs.py:250-255; bundle_adjust_io.py:18-22).
//
// Restates pose_optimization/two_view/bundle_adjust_gauss_newton_2_view.py (Observations :10-48,
// fill_J :50-67, compute_A_b :69-99, BundleAdjustGaussNewton2View.run :127-201) and
// run_bundle_adjust_2_view (estimate_relative_pose.py:138-144).  Camera 0 is fixed at the identity;
// unknowns = 6 pose parameters of camera 1 + 3 per matched point.
//
// The reference builds the dense (6+3M)^2 normal matrix per sample in Python and LU-factorises it
// (M <= 2048 -> 6150^2 fp32 per iteration).  The structure is block-arrow: the point blocks are
// independent 3x3's.  Here ONE workgroup per pair eliminates them analytically (Schur complement):
//   (Hcc' - sum_p Hcp Hpp'^-1 Hcp^T) dc = gc - sum_p Hcp Hpp'^-1 gp,   dp = Hpp'^-1 (gp - Hcp^T dc)
// with H' = H + lambda * diag(H) - algebraically the reference's Jacobi-preconditioned damped system
// (D^-1 J^T J + lambda I) d = D^-1 b.  All accumulation in fp64, reductions by wavefront shuffles,
// the 6x6 solve by one lane.  Same LM control flow as the reference: update ALWAYS applied, best-residual
// pose kept, lambda /= 3.5 on improvement else *= 1.5, n_iterations + 1 residual evaluations.
#include "common.h"
#include "small_linalg.h"

namespace e2emv {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block-wide sum of NV doubles (256 threads); result valid in every thread
template <int NV>
__device__ __forceinline__ void block_sum_n(double (&v)[NV], double* red /* [4][NV] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double s = wsum(v[i]);
        if (lane == 0) red[wave * NV + i] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = red[i] + red[NV + i] + red[2 * NV + i] + red[3 * NV + i];
}

// symmetric 3x3 (a00 a01 a02 a11 a12 a22) inverse; returns false when singular
__device__ __forceinline__ bool inv3_sym(const double* a, double* inv) {
    const double c00 = a[3] * a[5] - a[4] * a[4], c01 = a[2] * a[4] - a[1] * a[5], c02 = a[1] * a[4] - a[2] * a[3];
    const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    if (!(fabs(det) > 0.0)) return false;
    const double id = 1.0 / det;
    inv[0] = c00 * id; inv[1] = c01 * id; inv[2] = c02 * id;
    inv[3] = (a[0] * a[5] - a[2] * a[2]) * id; inv[4] = (a[1] * a[2] - a[0] * a[4]) * id;
    inv[5] = (a[0] * a[3] - a[1] * a[1]) * id;
    return true;
}

struct BaParams {
    int B, N, n_it;
    const float* k0;    // [B][N][2] normalised keypoints image 0
    const float* k1;
    const float* conf;  // [B][N]
    const float* Tin;   // [B][4][4]
    float* Tout;        // [B][4][4]
    uint8_t* valid;     // [B]
    double* X;          // workspace [B][N][3]
    float lm_inc, lm_dec;
};

// per-point quantities for the current pose: residuals and Jacobian blocks (already confidence weighted)
struct PointTerms {
    double r0[2], r1[2];
    double Jp0[2][3], Jp1[2][3], Jc[2][6];
};
__device__ __forceinline__ void point_terms(const double* Rt, const double* X, double x0, double y0, double x1, double y1,
                                            double c, PointTerms& q) {
    // camera 0: identity
    const double iz0 = 1.0 / X[2];
    q.r0[0] = c * (X[0] * iz0 - x0);
    q.r0[1] = c * (X[1] * iz0 - y0);
    q.Jp0[0][0] = c * iz0; q.Jp0[0][1] = 0.0; q.Jp0[0][2] = -c * X[0] * iz0 * iz0;
    q.Jp0[1][0] = 0.0; q.Jp0[1][1] = c * iz0; q.Jp0[1][2] = -c * X[1] * iz0 * iz0;
    // camera 1: Ap = R X + t
    const double a0 = Rt[0] * X[0] + Rt[1] * X[1] + Rt[2] * X[2] + Rt[9];
    const double a1 = Rt[3] * X[0] + Rt[4] * X[1] + Rt[5] * X[2] + Rt[10];
    const double a2 = Rt[6] * X[0] + Rt[7] * X[1] + Rt[8] * X[2] + Rt[11];
    const double iz = 1.0 / a2;
    q.r1[0] = c * (a0 * iz - x1);
    q.r1[1] = c * (a1 * iz - y1);
    const double j00 = c * iz, j02 = -c * a0 * iz * iz, j11 = c * iz, j12 = -c * a1 * iz * iz;  // c * J_proj (2x3)
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // J_proj R
        q.Jp1[0][k] = j00 * Rt[k] + j02 * Rt[6 + k];
        q.Jp1[1][k] = j11 * Rt[3 + k] + j12 * Rt[6 + k];
    }
    // J_proj [I | -hat(Ap)],  hat(a) = [[0,-a2,a1],[a2,0,-a0],[-a1,a0,0]]
    q.Jc[0][0] = j00; q.Jc[0][1] = 0.0; q.Jc[0][2] = j02;
    q.Jc[1][0] = 0.0; q.Jc[1][1] = j11; q.Jc[1][2] = j12;
    q.Jc[0][3] = -(j02 * (-a1));          // -(J row . hat column 0) ; hat col0 = (0, a2, -a1)
    q.Jc[0][4] = -(j00 * (-a2) + j02 * a0);   // hat col1 = (-a2, 0, a0)
    q.Jc[0][5] = -(j00 * a1);                 // hat col2 = (a1, -a0, 0)
    q.Jc[1][3] = -(j11 * a2 + j12 * (-a1));
    q.Jc[1][4] = -(j12 * a0);
    q.Jc[1][5] = -(j11 * (-a0));
}

__global__ __launch_bounds__(256) void ba2view_kernel(BaParams p) {
    __shared__ double red[4 * 32];
    __shared__ double sRt[12], sBest[12], sDelta[6];
    __shared__ double sLam, sBestR;
    __shared__ int sFlags;  // bit0: skip update this iteration
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = p.N;
    const float* k0 = p.k0 + (int64_t)b * N * 2;
    const float* k1 = p.k1 + (int64_t)b * N * 2;
    const float* cf = p.conf + (int64_t)b * N;
    const float* Ti = p.Tin + (int64_t)b * 16;
    double* X = p.X + (int64_t)b * N * 3;

    // confidence normalisation (:45-48: each match = two observations) and validity (:132-136)
    double st[2] = {0.0, 0.0};
    for (int i = tid; i < N; i += 256)
        if (cf[i] > 0.f) { st[0] += (double)cf[i]; st[1] += 1.0; }
    block_sum_n<2>(st, red);
    const bool valid = st[1] > 6.5;
    if (tid < 16) p.Tout[(int64_t)b * 16 + tid] = Ti[tid];
    if (tid == 0) p.valid[b] = valid ? 1 : 0;
    if (!valid) return;
    const double cden = 0.5 * fmax(2.0 * st[0], 1e-6);

    if (tid < 12) {
        const int r = tid < 9 ? tid / 3 : tid - 9, c = tid < 9 ? tid % 3 : 3;
        sRt[tid] = (double)Ti[r * 4 + c];   // R row-major (0..8), t (9..11)
        sBest[tid] = sRt[tid];
    }
    if (tid == 0) { sLam = 0.1; sBestR = 0.0; sFlags = 0; }
    __syncthreads();
    for (int i = tid; i < N; i += 256)
        if (cf[i] > 0.f) triangulate_xyz(k0[2 * i], k0[2 * i + 1], k1[2 * i], k1[2 * i + 1], sRt, X + 3 * i);
    __syncthreads();

    for (int it = 0; it <= p.n_it; ++it) {
        double Rt[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) Rt[k] = sRt[k];
        // ---- pass 1: residual norm, camera block Hcc (21), gc (6), positivity of the diagonal
        double a[29];
#pragma unroll
        for (int k = 0; k < 29; ++k) a[k] = 0.0;
        for (int i = tid; i < N; i += 256) {
            if (!(cf[i] > 0.f)) continue;
            PointTerms q;
            point_terms(Rt, X + 3 * i, k0[2 * i], k0[2 * i + 1], k1[2 * i], k1[2 * i + 1], (double)cf[i] / cden, q);
            a[27] += q.r0[0] * q.r0[0] + q.r0[1] * q.r0[1] + q.r1[0] * q.r1[0] + q.r1[1] * q.r1[1];
            int idx = 0;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
#pragma unroll
                for (int v = u; v < 6; ++v) a[idx++] += q.Jc[0][u] * q.Jc[0][v] + q.Jc[1][u] * q.Jc[1][v];
                a[21 + u] -= q.Jc[0][u] * q.r1[0] + q.Jc[1][u] * q.r1[1];
            }
            // point diagonal must be > 0 for the Jacobi preconditioner (:171-173)
            bool pos = true;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                pos = pos && (q.Jp0[0][k] * q.Jp0[0][k] + q.Jp0[1][k] * q.Jp0[1][k] + q.Jp1[0][k] * q.Jp1[0][k] + q.Jp1[1][k] * q.Jp1[1][k]) > 0.0;
            if (!pos) a[28] += 1.0;
        }
        block_sum_n<29>(a, red);
        // ---- LM bookkeeping (:150-161), identical in every thread
        const double rn = a[27];
        if (tid == 0) {
            if (it == 0) {
                sBestR = rn;
            } else if (rn < sBestR) {
                sBestR = rn;
#pragma unroll
                for (int k = 0; k < 12; ++k) sBest[k] = Rt[k];
                sLam = sLam / (double)p.lm_dec;
            } else {
                sLam = sLam * (double)p.lm_inc;
            }
        }
        __syncthreads();
        if (it == p.n_it) break;
        const double lam = sLam;
        // camera diagonal positions in the packed upper triangle: 0, 6, 11, 15, 18, 20
        const int dpos[6] = {0, 6, 11, 15, 18, 20};
        bool precond = a[28] < 0.5;
#pragma unroll
        for (int u = 0; u < 6; ++u) precond = precond && a[dpos[u]] > 0.0;

        // ---- pass 2: Schur complement of the point blocks
        double s[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) s[k] = 0.0;
        for (int i = tid; i < N; i += 256) {
            if (!(cf[i] > 0.f)) continue;
            PointTerms q;
            point_terms(Rt, X + 3 * i, k0[2 * i], k0[2 * i + 1], k1[2 * i], k1[2 * i + 1], (double)cf[i] / cden, q);
            double Hpp[6], gp[3], Hcp[6][3], inv[6];
            int idx = 0;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
#pragma unroll
                for (int v = u; v < 3; ++v)
                    Hpp[idx++] = q.Jp0[0][u] * q.Jp0[0][v] + q.Jp0[1][u] * q.Jp0[1][v] + q.Jp1[0][u] * q.Jp1[0][v] + q.Jp1[1][u] * q.Jp1[1][v];
                gp[u] = -(q.Jp0[0][u] * q.r0[0] + q.Jp0[1][u] * q.r0[1] + q.Jp1[0][u] * q.r1[0] + q.Jp1[1][u] * q.r1[1]);
            }
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int v = 0; v < 3; ++v) Hcp[u][v] = q.Jc[0][u] * q.Jp1[0][v] + q.Jc[1][u] * q.Jp1[1][v];
            Hpp[0] += lam * (precond ? fmax(Hpp[0], 1e-12) : 1.0);
            Hpp[3] += lam * (precond ? fmax(Hpp[3], 1e-12) : 1.0);
            Hpp[5] += lam * (precond ? fmax(Hpp[5], 1e-12) : 1.0);
            if (!inv3_sym(Hpp, inv)) continue;
            // W = Hcp Hpp'^-1 (6x3)
            double W[6][3];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                W[u][0] = Hcp[u][0] * inv[0] + Hcp[u][1] * inv[1] + Hcp[u][2] * inv[2];
                W[u][1] = Hcp[u][0] * inv[1] + Hcp[u][1] * inv[3] + Hcp[u][2] * inv[4];
                W[u][2] = Hcp[u][0] * inv[2] + Hcp[u][1] * inv[4] + Hcp[u][2] * inv[5];
            }
            idx = 0;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
#pragma unroll
                for (int v = u; v < 6; ++v) s[idx++] += W[u][0] * Hcp[v][0] + W[u][1] * Hcp[v][1] + W[u][2] * Hcp[v][2];
                s[21 + u] += W[u][0] * gp[0] + W[u][1] * gp[1] + W[u][2] * gp[2];
            }
        }
        block_sum_n<27>(s, red);
        if (tid == 0) {
            // reduced 6x6 system, Gaussian elimination with partial pivoting (fp64)
            double M6[6][7];
            int idx = 0;
            for (int u = 0; u < 6; ++u)
                for (int v = u; v < 6; ++v) {
                    const double h = a[idx] - s[idx];
                    M6[u][v] = h;
                    M6[v][u] = h;
                    ++idx;
                }
            for (int u = 0; u < 6; ++u) {
                M6[u][u] += lam * (precond ? fmax(a[dpos[u]], 1e-12) : 1.0);
                M6[u][6] = a[21 + u] - s[21 + u];
            }
            bool ok = true;
            for (int c = 0; c < 6 && ok; ++c) {
                int piv = c;
                for (int r = c + 1; r < 6; ++r)
                    if (fabs(M6[r][c]) > fabs(M6[piv][c])) piv = r;
                if (!(fabs(M6[piv][c]) > 0.0)) { ok = false; break; }
                if (piv != c)
                    for (int k = 0; k < 7; ++k) { const double t = M6[c][k]; M6[c][k] = M6[piv][k]; M6[piv][k] = t; }
                for (int r = c + 1; r < 6; ++r) {
                    const double f = M6[r][c] / M6[c][c];
                    for (int k = c; k < 7; ++k) M6[r][k] -= f * M6[c][k];
                }
            }
            if (ok) {
                for (int c = 5; c >= 0; --c) {
                    double v = M6[c][6];
                    for (int k = c + 1; k < 6; ++k) v -= M6[c][k] * sDelta[k];
                    sDelta[c] = v / M6[c][c];
                    ok = ok && isfinite(sDelta[c]);
                }
            }
            sFlags = ok ? 0 : 1;
        }
        __syncthreads();
        if (sFlags & 1) continue;  // singular system: the reference skips the update when LU reports info != 0
        double dc[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) dc[k] = sDelta[k];
        // ---- pass 3: back-substitute the points with the OLD pose, then move the pose
        for (int i = tid; i < N; i += 256) {
            if (!(cf[i] > 0.f)) continue;
            PointTerms q;
            point_terms(Rt, X + 3 * i, k0[2 * i], k0[2 * i + 1], k1[2 * i], k1[2 * i + 1], (double)cf[i] / cden, q);
            double Hpp[6], rhs[3], inv[6];
            int idx = 0;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
#pragma unroll
                for (int v = u; v < 3; ++v)
                    Hpp[idx++] = q.Jp0[0][u] * q.Jp0[0][v] + q.Jp0[1][u] * q.Jp0[1][v] + q.Jp1[0][u] * q.Jp1[0][v] + q.Jp1[1][u] * q.Jp1[1][v];
                double g = -(q.Jp0[0][u] * q.r0[0] + q.Jp0[1][u] * q.r0[1] + q.Jp1[0][u] * q.r1[0] + q.Jp1[1][u] * q.r1[1]);
#pragma unroll
                for (int w = 0; w < 6; ++w) g -= (q.Jc[0][w] * q.Jp1[0][u] + q.Jc[1][w] * q.Jp1[1][u]) * dc[w];
                rhs[u] = g;
            }
            Hpp[0] += lam * (precond ? fmax(Hpp[0], 1e-12) : 1.0);
            Hpp[3] += lam * (precond ? fmax(Hpp[3], 1e-12) : 1.0);
            Hpp[5] += lam * (precond ? fmax(Hpp[5], 1e-12) : 1.0);
            if (!inv3_sym(Hpp, inv)) continue;
            X[3 * i] += inv[0] * rhs[0] + inv[1] * rhs[1] + inv[2] * rhs[2];
            X[3 * i + 1] += inv[1] * rhs[0] + inv[3] * rhs[1] + inv[4] * rhs[2];
            X[3 * i + 2] += inv[2] * rhs[0] + inv[4] * rhs[1] + inv[5] * rhs[2];
        }
        __syncthreads();
        if (tid == 0) {
            // extr1 <- exp(dc) extr1, dc = (v, w): pytorch3d se3_exp_map with its 1e-4 clamp of |w|^2 (:193-195)
            const double wx = dc[3], wy = dc[4], wz = dc[5];
            const double th2 = fmax(wx * wx + wy * wy + wz * wz, 1e-4), th = sqrt(th2);
            const double f1 = sin(th) / th, f2 = (1.0 - cos(th)) / th2, f3 = (th - sin(th)) / (th2 * th);
            const double Kx[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
            double K2[9], Rd[9], Vm[9];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) K2[i * 3 + j] = Kx[i * 3] * Kx[j] + Kx[i * 3 + 1] * Kx[3 + j] + Kx[i * 3 + 2] * Kx[6 + j];
            for (int i = 0; i < 9; ++i) {
                const double e = (i % 4 == 0) ? 1.0 : 0.0;
                Rd[i] = e + f1 * Kx[i] + f2 * K2[i];
                Vm[i] = e + f2 * Kx[i] + f3 * K2[i];
            }
            double td[3], Rn[9], tn[3];
            for (int i = 0; i < 3; ++i) td[i] = Vm[i * 3] * dc[0] + Vm[i * 3 + 1] * dc[1] + Vm[i * 3 + 2] * dc[2];
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = Rd[i * 3] * Rt[j] + Rd[i * 3 + 1] * Rt[3 + j] + Rd[i * 3 + 2] * Rt[6 + j];
                tn[i] = Rd[i * 3] * Rt[9] + Rd[i * 3 + 1] * Rt[10] + Rd[i * 3 + 2] * Rt[11] + td[i];
            }
            for (int i = 0; i < 9; ++i) sRt[i] = Rn[i];
            for (int i = 0; i < 3; ++i) sRt[9 + i] = tn[i];
        }
        __syncthreads();
    }
    if (tid < 12) {
        const int r = tid < 9 ? tid / 3 : tid - 9, c = tid < 9 ? tid % 3 : 3;
        p.Tout[(int64_t)b * 16 + r * 4 + c] = (float)sBest[tid];
    }
}

}  // namespace e2emv

using namespace e2emv;

extern "C" int e2emv_ba_2view(e2emv_ctx* ctx, int B, int N, const float* d_kpts0n, const float* d_kpts1n, const float* d_conf,
                              const float* d_T_init, int n_iterations, float* d_T_out, uint8_t* d_valid, void* stream) {
    if (!ctx || !d_kpts0n || !d_kpts1n || !d_conf || !d_T_init || !d_T_out || !d_valid) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (B <= 0 || N <= 0 || n_iterations < 0) return set_err(ctx, E2EMV_ESHAPE, "ba_2view: B=%d N=%d iterations=%d", B, N, n_iterations);
    hipStream_t s = (hipStream_t)stream;
    int rc = ws_reserve(ctx, (size_t)B * N * 3 * sizeof(double) + 256);
    if (rc) return rc;
    BaParams p{};
    p.B = B; p.N = N; p.n_it = n_iterations;
    p.k0 = d_kpts0n; p.k1 = d_kpts1n; p.conf = d_conf; p.Tin = d_T_init; p.Tout = d_T_out; p.valid = d_valid;
    p.X = (double*)ctx->d_ws;
    p.lm_inc = 1.5f; p.lm_dec = 3.5f;
    prof_begin(ctx, PS_W8PT, s);
    hipLaunchKernelGGL(ba2view_kernel, dim3(B), dim3(256), 0, s, p);
    prof_end(ctx, s);
    E2EMV_CHECK_LAUNCH(ctx, "ba2view_kernel");
    return E2EMV_OK;
}
