// Confidence-weighted eight-point relative pose on the GPU (two views), batched over pairs.
//
// Restates pose_optimization/two_view/estimate_relative_pose.py:
//   normalize :9-14, get_kpts :16-31, find_fundamental :34-82, estimate_relative_pose_w8pt
//   :84-128, and compute_pose_error.py:3-22; the kornia 0.7.0 calls the reference makes there
//   (normalize_points, normalize_transformation, motion_from_essential[_choose_solution],
//   triangulate_points, depth_from_point, symmetrical_epipolar_distance) are folded in.
//
// What the reference does with cuSOLVER/LAPACK SVDs is done here without any SVD library:
//   * the N x 9 weighted design matrix is never materialised (the reference even builds an
//     N x N diag, :68-69): its 9 x 9 Gram matrix is accumulated in fp64 with wavefront
//     shuffles; the null vector = eigenvector of the smallest eigenvalue (cyclic Jacobi, fp64).
//     fp64 Gram + Jacobi is MORE accurate than the reference's fp32 N x 9 SVD.
//   * rank-2 projection: F - (F v3) v3^T with v3 from the 3 x 3 Jacobi of F^T F.
//   * essential decomposition: V from Jacobi of E^T E, u_k = E v_k / |E v_k|; the reference's
//     det-sign fixes make U = [u1 u2 u1xu2], V = [v1 v2 v1xv2], so no sign bookkeeping.
//   * 5N four-by-four DLT triangulations per pair (86 % of the reference's time, all in
//     batched 4 x 4 SVDs): one thread per point, 4 x 4 Jacobi of A^T A in registers (fp64).
// Candidate ORDER is SVD-sign dependent in the reference too; only the chosen pose is
// compared.  Cheirality winner = per-sample arg-max of the positive-depth counts (first
// maximum), see oracle/kornia_fns.py for the B > 1 caveat of kornia 0.7.0.
#include "common.h"
#include "small_linalg.h"

namespace e2emv {

// dynamic-index variant for the 9 x 9 problem, operating on LDS arrays from ONE thread
__device__ void jacobi_dynamic(double* A, double* V, int n) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = 0; i < n; ++i) {
            dg += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-60 || off <= 1e-34 * dg) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                A[p * n + p] -= t * apq;
                A[q * n + q] += t * apq;
                A[p * n + q] = 0.0;
                A[q * n + p] = 0.0;
                for (int k = 0; k < n; ++k) {
                    if (k != p && k != q) {
                        const double akp = A[k * n + p], akq = A[k * n + q];
                        const double np_ = c * akp - s * akq, nq_ = s * akp + c * akq;
                        A[k * n + p] = np_;
                        A[p * n + k] = np_;
                        A[k * n + q] = nq_;
                        A[q * n + k] = nq_;
                    }
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
}

// 9 x 9 symmetric eigen-problem on a whole workgroup (>= 81 threads; every thread of the block must call): A (overwritten,
// eigenvalues on its diagonal), B scratch, V0 / V1 eigenvector ping-pong; returns the buffer that holds the eigenvectors
// (columns).  Same rotation formulas, thresholds and sweep limit as jacobi_dynamic; round-robin pair order.
__device__ double* jacobi9_parallel(double* A, double* B, double* V0, double* V1, double* rot, int* part, int tid) {
    const int i = tid / 9, j = tid - 9 * i;
    const bool own = tid < 81;
    if (own) V0[tid] = (i == j) ? 1.0 : 0.0;
    double* Vc = V0;
    double* Vn = V1;
    __syncthreads();
    for (int sweep = 0; sweep < 40; ++sweep) {
        if (tid == 0) {
            double off = 0.0, dg = 0.0;
            for (int a = 0; a < 9; ++a) {
                dg += A[a * 9 + a] * A[a * 9 + a];
                for (int c = a + 1; c < 9; ++c) off += A[a * 9 + c] * A[a * 9 + c];
            }
            part[9] = (off <= 1e-60 || off <= 1e-34 * dg) ? 1 : 0;
        }
        __syncthreads();
        if (part[9]) break;
        for (int r = 0; r < 9; ++r) {
            if (tid < 4) {
                const int k = tid + 1;
                int p = (r + k) % 9, q = (r + 9 - k) % 9;
                if (p > q) { const int t_ = p; p = q; q = t_; }
                const double apq = A[p * 9 + q];
                double c = 1.0, sn = 0.0;
                if (!(fabs(apq) < 1e-300)) {
                    const double theta = (A[q * 9 + q] - A[p * 9 + p]) / (2.0 * apq);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    c = 1.0 / sqrt(t * t + 1.0);
                    sn = t * c;
                }
                rot[2 * p] = c; rot[2 * p + 1] = -sn;   // index p:  new_p = c x_p - s x_q
                rot[2 * q] = c; rot[2 * q + 1] = sn;    // index q:  new_q = c x_q + s x_p
                part[p] = q;
                part[q] = p;
            } else if (tid == 4) {
                part[r] = -1;
            }
            __syncthreads();
            if (own) {  // columns: B = A J, Vn = Vc J
                const int pj = part[j];
                if (pj >= 0) {
                    const double c = rot[2 * j], sg = rot[2 * j + 1];
                    B[tid] = c * A[tid] + sg * A[i * 9 + pj];
                    Vn[tid] = c * Vc[tid] + sg * Vc[i * 9 + pj];
                } else {
                    B[tid] = A[tid];
                    Vn[tid] = Vc[tid];
                }
            }
            __syncthreads();
            if (own) {  // rows: A = J^T B; the rotated pair's off-diagonal entry is zero by construction
                const int pi = part[i];
                double v = B[tid];
                if (pi >= 0) v = rot[2 * i] * v + rot[2 * i + 1] * B[pi * 9 + j];
                if (pi == j) v = 0.0;
                A[tid] = v;
            }
            double* t_ = Vc; Vc = Vn; Vn = t_;
            __syncthreads();
        }
    }
    return Vc;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct W8Params {
    int B, N, kdim, intr_batch;
    const int* n_per;  // optional [B]: sample b uses its first n_per[b] <= N correspondences (rows stay N apart)
    const float* k0;
    const float* k1;
    const float* K0;
    const float* K1;
    const float* conf;
    int choose_closest;
    const float* Tgt;
    int determine_inliers;
    float* T;
    float* k0n;
    float* k1n;
    float* conf_n;
    uint8_t* inliers;
    uint8_t* posdepth;
    float* F;
    int32_t* status;
    // workspace
    double* E;       // [B][9]
    double* cands;   // [B][4][12]  R row-major, t
    int* sel;        // [B] chosen candidate for choose_closest, else -1
    int* counts;     // [B][4]
    uint8_t* pd_c;   // [B][4][N]
};

__device__ __forceinline__ void mat3_mul(const double* a, const double* b, double* c) {  // c = a b
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

__device__ __forceinline__ double angle_err(const double* R, const double* t, const float* Tg) {
    // compute_rotation_error + compute_translation_error_as_angle(reduce=False)
    double tr = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) tr += R[i * 3 + j] * (double)Tg[i * 4 + j];  // trace(R^T Rg)
    double ca = fmin(fmax((tr - 1.0) * 0.5, -1.0), 1.0);
    double er = fabs(acos(ca));
    double n0 = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    double g0 = (double)Tg[3], g1 = (double)Tg[7], g2 = (double)Tg[11];
    double n1 = sqrt(g0 * g0 + g1 * g1 + g2 * g2);
    double et = 0.0;
    if (n0 * n1 > 1e-6) {
        double cd = (t[0] * g0 + t[1] * g1 + t[2] * g2) / (n0 * n1);
        et = fabs(acos(fmin(fmax(cd, -1.0), 1.0)));
    }
    return er + et;
}

// Everything behind the eigenvector of the weighted 8-point solve (estimate_relative_pose.py:74-82 + kornia's
// decompose_essential_matrix): rank-2 projection of F (row-major 3x3, Hartley-normalised frame), de-normalisation, / E22,
// E -> (R1, R2, t).  One thread, fp64.  Shared by the forward kernel and by the backward (which differentiates it
// numerically: it is a smooth function of nine numbers).  Returns status bit 4 (degenerate E: completed deterministically).
__device__ int w8pt_tail(double* Fm, double s0, double mx0, double my0, double s1, double mx1, double my1, double* E, double* R1, double* R2,
                         double* tv) {
    int st = 0;
    // rank 2: remove the smallest singular direction
    {
        double A3[3][3], V3[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) A3[i][j] = Fm[i] * Fm[j] + Fm[3 + i] * Fm[3 + j] + Fm[6 + i] * Fm[6 + j];  // F^T F
        jacobi_static<3>(A3, V3);
        int m3 = 0;
        if (A3[1][1] < A3[m3][m3]) m3 = 1;
        if (A3[2][2] < A3[m3][m3]) m3 = 2;
        const double v[3] = {V3[0][m3], V3[1][m3], V3[2][m3]};
        for (int i = 0; i < 3; ++i) {
            const double fv = Fm[i * 3] * v[0] + Fm[i * 3 + 1] * v[1] + Fm[i * 3 + 2] * v[2];
            for (int j = 0; j < 3; ++j) Fm[i * 3 + j] -= fv * v[j];
        }
    }
    // de-normalise: T2^T F T1 with T = [[s,0,-s mx],[0,s,-s my],[0,0,1]]
    {
        const double T1[9] = {s0, 0, -s0 * mx0, 0, s0, -s0 * my0, 0, 0, 1};
        const double T2t[9] = {s1, 0, 0, 0, s1, 0, -s1 * mx1, -s1 * my1, 1};
        double tmp[9];
        mat3_mul(Fm, T1, tmp);
        mat3_mul(T2t, tmp, E);
        if (fabs(E[8]) > 1e-8) {  // normalize_transformation
            const double d = E[8] + 1e-8;
            for (int i = 0; i < 9; ++i) E[i] /= d;
        }
    }
    // ---- essential decomposition ----
    {
        double A3[3][3], V3[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) A3[i][j] = E[i] * E[j] + E[3 + i] * E[3 + j] + E[6 + i] * E[6 + j];
        jacobi_static<3>(A3, V3);
        // order eigenvalues descending: i0 >= i1 >= i2
        int i0 = 0, i2 = 0;
        for (int i = 1; i < 3; ++i) {
            if (A3[i][i] > A3[i0][i0]) i0 = i;
            if (A3[i][i] < A3[i2][i2]) i2 = i;
        }
        if (i0 == i2) { i0 = 0; i2 = 2; }
        const int i1 = 3 - i0 - i2;
        double v1[3] = {V3[0][i0], V3[1][i0], V3[2][i0]};
        double v2[3] = {V3[0][i1], V3[1][i1], V3[2][i1]};
        double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
        double u1[3], u2[3];
        for (int i = 0; i < 3; ++i) {
            u1[i] = E[i * 3] * v1[0] + E[i * 3 + 1] * v1[1] + E[i * 3 + 2] * v1[2];
            u2[i] = E[i * 3] * v2[0] + E[i * 3 + 1] * v2[1] + E[i * 3 + 2] * v2[2];
        }
        // Degenerate inputs (no usable correspondence: all weights zero, or every gathered point identical) give E of rank
        // 1 or 0.  A library SVD still returns an orthonormal U there (arbitrary in the null space); U = E V / sigma would
        // divide by zero, so the missing columns are completed deterministically and the case is flagged (status bit 2).
        double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
        if (n1 > 1e-150) {
            for (int i = 0; i < 3; ++i) u1[i] /= n1;
        } else {
            u1[0] = 1.0; u1[1] = 0.0; u1[2] = 0.0;
            st |= 4;
        }
        // Gram-Schmidt keeps U orthonormal when sigma1 ~ sigma2 makes E v2 slightly oblique
        double dp = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
        for (int i = 0; i < 3; ++i) u2[i] -= dp * u1[i];
        double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
        if (!(n2 > 1e-7 * n1) || !(n2 > 1e-150)) {  // sigma2 ~ 0: any unit vector orthogonal to u1
            int a = 0;
            if (fabs(u1[1]) < fabs(u1[a])) a = 1;
            if (fabs(u1[2]) < fabs(u1[a])) a = 2;
            for (int i = 0; i < 3; ++i) u2[i] = (i == a ? 1.0 : 0.0) - u1[a] * u1[i];
            n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
            st |= 4;
        }
        for (int i = 0; i < 3; ++i) u2[i] /= n2;
        double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
        // R1 = U W V^T, R2 = U W^T V^T,  W = [[0,-1,0],[1,0,0],[0,0,1]]
        // U W   = [u2, -u1, u3] (columns);  U W^T = [-u2, u1, u3]
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                R1[i * 3 + j] = u2[i] * v1[j] - u1[i] * v2[j] + u3[i] * v3[j];
                R2[i * 3 + j] = -u2[i] * v1[j] + u1[i] * v2[j] + u3[i] * v3[j];
            }
        for (int i = 0; i < 3; ++i) tv[i] = u3[i];
    }
    return st;
}

// Kernel 1: one workgroup per pair -> normalised points, weights, essential matrix, candidates.
__global__ __launch_bounds__(256) void w8pt_fundamental(W8Params p) {
    __shared__ double sG[81];
    __shared__ double sB[81];
    __shared__ double sV1[81];
    __shared__ double sV2[81];
    __shared__ double srot[18];
    __shared__ int spart[12];
    __shared__ double sred[4 * 48];
    __shared__ double sstat[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = p.n_per ? p.n_per[b] : p.N;  // correspondences of this sample
    const int ldN = p.N;                        // row stride of every per-correspondence buffer
    const float* K0 = p.K0 + (p.intr_batch == 1 ? 0 : (int64_t)b * p.kdim * p.kdim);
    const float* K1 = p.K1 + (p.intr_batch == 1 ? 0 : (int64_t)b * p.kdim * p.kdim);
    const float fx0 = K0[0], fy0 = K0[p.kdim + 1], cx0 = K0[2], cy0 = K0[p.kdim + 2];
    const float fx1 = K1[0], fy1 = K1[p.kdim + 1], cx1 = K1[2], cy1 = K1[p.kdim + 2];
    const float* k0 = p.k0 + (int64_t)b * ldN * 2;
    const float* k1 = p.k1 + (int64_t)b * ldN * 2;
    const float* cf = p.conf + (int64_t)b * ldN;
    float* k0n = p.k0n + (int64_t)b * ldN * 2;
    float* k1n = p.k1n + (int64_t)b * ldN * 2;
    for (int i = N + tid; i < ldN; i += 256) {  // rows beyond this sample's count: defined, weightless
        k0n[2 * i] = 0.f; k0n[2 * i + 1] = 0.f; k1n[2 * i] = 0.f; k1n[2 * i + 1] = 0.f;
        p.conf_n[(int64_t)b * ldN + i] = 0.f;
    }

    // pass 1: intrinsics normalisation (fp32 like the reference), sums for Hartley + weights
    double acc[5] = {0, 0, 0, 0, 0};  // sum x0,y0,x1,y1,conf
    for (int i = tid; i < N; i += 256) {
        const float x0 = (k0[2 * i] - cx0) / fx0, y0 = (k0[2 * i + 1] - cy0) / fy0;
        const float x1 = (k1[2 * i] - cx1) / fx1, y1 = (k1[2 * i + 1] - cy1) / fy1;
        k0n[2 * i] = x0; k0n[2 * i + 1] = y0;
        k1n[2 * i] = x1; k1n[2 * i + 1] = y1;
        acc[0] += x0; acc[1] += y0; acc[2] += x1; acc[3] += y1; acc[4] += cf[i];
    }
    for (int j = 0; j < 5; ++j) {
        double v = wave_sum_d(acc[j]);
        if (lane == 0) sred[wave * 48 + j] = v;
    }
    __syncthreads();
    if (tid < 5) sstat[tid] = sred[tid] + sred[48 + tid] + sred[96 + tid] + sred[144 + tid];
    __syncthreads();
    const double mx0 = sstat[0] / N, my0 = sstat[1] / N, mx1 = sstat[2] / N, my1 = sstat[3] / N;
    const float wsum = (float)sstat[4] + 1e-6f;  // confidence / (sum + 1e-6)   (:87-88)
    // pass 2: mean distance to the centroid (Hartley scale, UNWEIGHTED over all N rows - E2)
    double d0 = 0, d1 = 0;
    for (int i = tid; i < N; i += 256) {
        const double ax = (double)k0n[2 * i] - mx0, ay = (double)k0n[2 * i + 1] - my0;
        const double bx = (double)k1n[2 * i] - mx1, by = (double)k1n[2 * i + 1] - my1;
        d0 += sqrt(ax * ax + ay * ay);
        d1 += sqrt(bx * bx + by * by);
    }
    d0 = wave_sum_d(d0);
    d1 = wave_sum_d(d1);
    __syncthreads();
    if (lane == 0) { sred[wave * 48] = d0; sred[wave * 48 + 1] = d1; }
    __syncthreads();
    if (tid < 2) sstat[8 + tid] = sred[tid] + sred[48 + tid] + sred[96 + tid] + sred[144 + tid];
    __syncthreads();
    const double s0 = sqrt(2.0) / (sstat[8] / N + 1e-8), s1 = sqrt(2.0) / (sstat[9] / N + 1e-8);

    // pass 3: weighted Gram matrix of the design rows (45 unique entries, fp64)
    double g[45];
#pragma unroll
    for (int j = 0; j < 45; ++j) g[j] = 0.0;
    for (int i = tid; i < N; i += 256) {
        const float wn = cf[i] / wsum;
        p.conf_n[(int64_t)b * ldN + i] = wn;
        const double w = wn;
        const double x1 = s0 * ((double)k0n[2 * i] - mx0), y1 = s0 * ((double)k0n[2 * i + 1] - my0);
        const double x2 = s1 * ((double)k1n[2 * i] - mx1), y2 = s1 * ((double)k1n[2 * i + 1] - my1);
        // row order (:65): [x2x1, x2y1, x2, y2x1, y2y1, y2, x1, y1, 1], weights multiply ROWS (E1)
        const double r[9] = {w * x2 * x1, w * x2 * y1, w * x2, w * y2 * x1, w * y2 * y1, w * y2, w * x1, w * y1, w};
        int j = 0;
#pragma unroll
        for (int a = 0; a < 9; ++a)
#pragma unroll
            for (int c = a; c < 9; ++c) g[j++] += r[a] * r[c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 45; ++j) {
        double v = wave_sum_d(g[j]);
        if (lane == 0) sred[wave * 48 + j] = v;
    }
    __syncthreads();
    if (tid < 45) {
        const double v = sred[tid] + sred[48 + tid] + sred[96 + tid] + sred[144 + tid];
        int j = 0, a = 0, c = 0;
        for (a = 0; a < 9; ++a) {
            if (tid < j + 9 - a) { c = a + (tid - j); break; }
            j += 9 - a;
        }
        sG[a * 9 + c] = v;
        sG[c * 9 + a] = v;
    }
    __syncthreads();
    // ---- eigen-decomposition of the 9 x 9 Gram matrix: cyclic Jacobi with PARALLEL rotations.  A round rotates the four
    // disjoint index pairs {(r + k) mod 9, (r - k) mod 9}, k = 1..4 (index r rests; 9 rounds = all 36 pairs = one sweep);
    // thread (i, j) owns element (i, j) of A and V: columns first (A J, V J), then rows (J^T A), ping-pong buffers, three
    // barriers per round - the serial form on one thread was 0.1 ms of dependent fp64 LDS traffic per call.
    double* const V = jacobi9_parallel(sG, sB, sV1, sV2, srot, spart, tid);
    if (tid != 0) return;

    // ---- single-thread tail: tiny dense algebra in fp64 ----
    double* const sV = V;
    // The reference takes V[..., -1] of torch.svd(X) with X [N,9] (:72-73).  For N >= 9 that is
    // the right singular vector of the smallest singular value; for N == 8 the thin SVD only has
    // 8 columns, so it is the vector of the smallest of the 8 NON-null singular values, not the
    // null vector.  Replicated: skip (9 - N) eigenvalues from the bottom.
    int mi = 0;
    {
        const int skip = N >= 9 ? 0 : 9 - N;
        bool used[9];
        for (int i = 0; i < 9; ++i) used[i] = false;
        for (int k = 0; k <= skip; ++k) {
            mi = -1;
            for (int i = 0; i < 9; ++i)
                if (!used[i] && (mi < 0 || sG[i * 9 + i] < sG[mi * 9 + mi])) mi = i;
            used[mi] = true;
        }
    }
    double Fm[9];
    for (int i = 0; i < 9; ++i) Fm[i] = sV[i * 9 + mi];  // row-major 3x3 (E6)
    double E[9], R1[9], R2[9], tv[3];
    const int st_tail = w8pt_tail(Fm, s0, mx0, my0, s1, mx1, my1, E, R1, R2, tv);
    int st = st_tail;
    if (!(sstat[4] > 1e-6)) st |= 1;
    for (int i = 0; i < 9; ++i) {
        p.E[b * 9 + i] = E[i];
        if (p.F) p.F[b * 9 + i] = (float)E[i];
        if (!isfinite(E[i])) st |= 2;
    }
    double* cd = p.cands + (int64_t)b * 48;
    for (int c = 0; c < 4; ++c) {
        const double* R = (c < 2) ? R1 : R2;
        const double sg = (c & 1) ? -1.0 : 1.0;
        for (int i = 0; i < 9; ++i) cd[c * 12 + i] = R[i];
        for (int i = 0; i < 3; ++i) cd[c * 12 + 9 + i] = sg * tv[i];
    }
    int sel = -1;
    if (p.choose_closest) {  // :95-107, strict < against 1e6 in candidate order
        const float* Tg = p.Tgt + (int64_t)b * 16;
        double best = 1e6;
        sel = -2;  // none accepted -> identity, like the reference's initial eye(4)
        for (int c = 0; c < 4; ++c) {
            const double e = angle_err(cd + c * 12, cd + c * 12 + 9, Tg);
            if (e < best) { best = e; sel = c; }
        }
    }
    p.sel[b] = sel;
    for (int c = 0; c < 4; ++c) p.counts[b * 4 + c] = 0;
    if (p.status) p.status[b] = st;
}

// ---- backward of the weighted 8-point solve with respect to the confidences (training, second slice: the pose loss) ----------
// T = tail(f), f = eigenvector (smallest eigenvalue) of A = sum_n w_n^2 x_n x_n^T, w_n = c_n / (sum c + 1e-6) (weights multiply
// the design ROWS, E1; the Hartley statistics are unweighted, E2: they do not depend on c).  Given g_T = dL/dT:
//   g_f  = J^T g_T, J = dT/df by central differences of the tail in fp64 (a smooth function of nine numbers; every perturbed
//          evaluation picks the candidate closest to the forward's T, so eigenvector sign / branch flips cannot leak in),
//   y    = (A - lambda I)^+ g_f = sum_{k != min} v_k (v_k . g_f) / (lambda_k - lambda)        (df = -(A - lambda I)^+ dA f),
//   dL/dw_n = -2 w_n (y . x_n)(f . x_n),   dL/dc_m = dL/dw_m / S - (sum_n dL/dw_n c_n) / S^2,  S = sum c + 1e-6.
// One workgroup per sample; the eigen-problem is recomputed (cheaper than a tape: 40 us per call of 32 samples).
struct W8BwdParams {
    int B, N;
    const float* k0n;   // [B][N][2] normalised camera coordinates (info["kpts0_norm"])
    const float* k1n;
    const float* conf;  // [B][N] the confidences handed to the forward (unnormalised)
    const float* T;     // [B][16] the forward's pose
    const float* gT;    // [B][16] dL/dT
    float* gconf;       // [B][N] out
};

__global__ __launch_bounds__(256) void w8pt_backward_kernel(W8BwdParams p) {
    __shared__ double sG[81];
    __shared__ double sB[81];
    __shared__ double sV1[81];
    __shared__ double sV2[81];
    __shared__ double srot[18];
    __shared__ int spart[12];
    __shared__ double sred[4 * 48];
    __shared__ double sstat[16];
    __shared__ double sT[18 * 12];
    __shared__ double sf[9], sgf[9], sy[9];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = p.N;
    const float* k0n = p.k0n + (int64_t)b * N * 2;
    const float* k1n = p.k1n + (int64_t)b * N * 2;
    const float* cf = p.conf + (int64_t)b * N;
    float* gc = p.gconf + (int64_t)b * N;
    double acc[5] = {0, 0, 0, 0, 0};
    for (int i = tid; i < N; i += 256) {
        acc[0] += k0n[2 * i]; acc[1] += k0n[2 * i + 1]; acc[2] += k1n[2 * i]; acc[3] += k1n[2 * i + 1]; acc[4] += cf[i];
    }
    for (int j = 0; j < 5; ++j) {
        double v = wave_sum_d(acc[j]);
        if (lane == 0) sred[wave * 48 + j] = v;
    }
    __syncthreads();
    if (tid < 5) sstat[tid] = sred[tid] + sred[48 + tid] + sred[96 + tid] + sred[144 + tid];
    __syncthreads();
    const double mx0 = sstat[0] / N, my0 = sstat[1] / N, mx1 = sstat[2] / N, my1 = sstat[3] / N;
    const float wsum = (float)sstat[4] + 1e-6f;
    double d0 = 0, d1 = 0;
    for (int i = tid; i < N; i += 256) {
        const double ax = (double)k0n[2 * i] - mx0, ay = (double)k0n[2 * i + 1] - my0;
        const double bx = (double)k1n[2 * i] - mx1, by = (double)k1n[2 * i + 1] - my1;
        d0 += sqrt(ax * ax + ay * ay);
        d1 += sqrt(bx * bx + by * by);
    }
    d0 = wave_sum_d(d0);
    d1 = wave_sum_d(d1);
    __syncthreads();
    if (lane == 0) { sred[wave * 48] = d0; sred[wave * 48 + 1] = d1; }
    __syncthreads();
    if (tid < 2) sstat[8 + tid] = sred[tid] + sred[48 + tid] + sred[96 + tid] + sred[144 + tid];
    __syncthreads();
    const double s0 = sqrt(2.0) / (sstat[8] / N + 1e-8), s1 = sqrt(2.0) / (sstat[9] / N + 1e-8);
    auto design = [&](int i, double* r) {  // the UNWEIGHTED design row of correspondence i (:65)
        const double x1 = s0 * ((double)k0n[2 * i] - mx0), y1 = s0 * ((double)k0n[2 * i + 1] - my0);
        const double x2 = s1 * ((double)k1n[2 * i] - mx1), y2 = s1 * ((double)k1n[2 * i + 1] - my1);
        r[0] = x2 * x1; r[1] = x2 * y1; r[2] = x2; r[3] = y2 * x1; r[4] = y2 * y1; r[5] = y2; r[6] = x1; r[7] = y1; r[8] = 1.0;
    };
    double g[45];
#pragma unroll
    for (int j = 0; j < 45; ++j) g[j] = 0.0;
    for (int i = tid; i < N; i += 256) {
        const double w = (double)(cf[i] / wsum);
        double r[9];
        design(i, r);
        int j = 0;
#pragma unroll
        for (int a = 0; a < 9; ++a)
#pragma unroll
            for (int c = a; c < 9; ++c) g[j++] += (w * r[a]) * (w * r[c]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 45; ++j) {
        double v = wave_sum_d(g[j]);
        if (lane == 0) sred[wave * 48 + j] = v;
    }
    __syncthreads();
    if (tid < 45) {
        const double v = sred[tid] + sred[48 + tid] + sred[96 + tid] + sred[144 + tid];
        int j = 0, a = 0, c = 0;
        for (a = 0; a < 9; ++a) {
            if (tid < j + 9 - a) { c = a + (tid - j); break; }
            j += 9 - a;
        }
        sG[a * 9 + c] = v;
        sG[c * 9 + a] = v;
    }
    __syncthreads();
    double* const V = jacobi9_parallel(sG, sB, sV1, sV2, srot, spart, tid);
    __syncthreads();
    if (tid == 0) {
        int mi = 0;
        const int skip = N >= 9 ? 0 : 9 - N;
        bool used[9];
        for (int i = 0; i < 9; ++i) used[i] = false;
        for (int k = 0; k <= skip; ++k) {
            mi = -1;
            for (int i = 0; i < 9; ++i)
                if (!used[i] && (mi < 0 || sG[i * 9 + i] < sG[mi * 9 + mi])) mi = i;
            used[mi] = true;
        }
        spart[10] = mi;
        for (int i = 0; i < 9; ++i) sf[i] = V[i * 9 + mi];
    }
    __syncthreads();
    const int mi = spart[10];
    const float* To = p.T + (int64_t)b * 16;
    const float* gT = p.gT + (int64_t)b * 16;
    const double h = 1e-6;
    if (tid < 18) {
        double Fm[9], E[9], R1[9], R2[9], tv[3];
        for (int i = 0; i < 9; ++i) Fm[i] = sf[i];
        Fm[tid >> 1] += (tid & 1) ? -h : h;
        (void)w8pt_tail(Fm, s0, mx0, my0, s1, mx1, my1, E, R1, R2, tv);
        int best = 0;
        double bd = 1e300;
        for (int c = 0; c < 4; ++c) {
            const double* R = c < 2 ? R1 : R2;
            const double sg = (c & 1) ? -1.0 : 1.0;
            double d = 0.0;
            for (int r = 0; r < 3; ++r) {
                for (int q = 0; q < 3; ++q) { const double e = R[r * 3 + q] - (double)To[r * 4 + q]; d += e * e; }
                const double e = sg * tv[r] - (double)To[r * 4 + 3];
                d += e * e;
            }
            if (d < bd) { bd = d; best = c; }
        }
        const double* R = best < 2 ? R1 : R2;
        const double sg = (best & 1) ? -1.0 : 1.0;
        for (int i = 0; i < 9; ++i) sT[tid * 12 + i] = R[i];
        for (int i = 0; i < 3; ++i) sT[tid * 12 + 9 + i] = sg * tv[i];
    }
    __syncthreads();
    if (tid < 9) {
        const double* Tp = sT + (2 * tid) * 12;
        const double* Tm = sT + (2 * tid + 1) * 12;
        double gsum = 0.0;
        for (int r = 0; r < 3; ++r) {
            for (int q = 0; q < 3; ++q) gsum += (double)gT[r * 4 + q] * (Tp[r * 3 + q] - Tm[r * 3 + q]);
            gsum += (double)gT[r * 4 + 3] * (Tp[9 + r] - Tm[9 + r]);
        }
        sgf[tid] = gsum / (2.0 * h);
    }
    __syncthreads();
    if (tid == 0) {
        // the forward returned the identity (no candidate accepted / degenerate input): the pose does not depend on c
        const double tn = (double)To[3] * To[3] + (double)To[7] * To[7] + (double)To[11] * To[11];
        double dot = 0.0;
        for (int i = 0; i < 9; ++i) dot += sgf[i] * sf[i];
        for (int i = 0; i < 9; ++i) sgf[i] -= dot * sf[i];
        const double lam = sG[mi * 9 + mi];
        for (int i = 0; i < 9; ++i) sy[i] = 0.0;
        if (tn > 0.25)
            for (int k = 0; k < 9; ++k) {
                if (k == mi) continue;
                const double den = sG[k * 9 + k] - lam;
                if (!(fabs(den) > 1e-300)) continue;
                double pr = 0.0;
                for (int i = 0; i < 9; ++i) pr += V[i * 9 + k] * sgf[i];
                for (int i = 0; i < 9; ++i) sy[i] += V[i * 9 + k] * pr / den;
            }
    }
    __syncthreads();
    double S1 = 0.0;
    for (int i = tid; i < N; i += 256) {
        const double w = (double)(cf[i] / wsum);
        double r[9];
        design(i, r);
        double yx = 0.0, fx = 0.0;
#pragma unroll
        for (int a = 0; a < 9; ++a) { yx += sy[a] * r[a]; fx += sf[a] * r[a]; }
        const double dw = -2.0 * w * yx * fx;
        gc[i] = (float)dw;
        S1 += dw * (double)cf[i];
    }
    S1 = wave_sum_d(S1);
    __syncthreads();
    if (lane == 0) sred[wave] = S1;
    __syncthreads();
    const double Sall = sred[0] + sred[1] + sred[2] + sred[3];
    const double ws = (double)wsum;
    for (int i = tid; i < N; i += 256) gc[i] = (float)((double)gc[i] / ws - Sall / (ws * ws));
}

// DLT triangulation of one correspondence with P1 = [I|0], P2 = [R|t]; returns the two depths.
__device__ __forceinline__ void triangulate(double x1, double y1, double x2, double y2, const double* Rt,
                                            double& depth0, double& depth1) {
    double X[3];
    triangulate_xyz(x1, y1, x2, y2, Rt, X);
    depth0 = X[2];
    depth1 = Rt[6] * X[0] + Rt[7] * X[1] + Rt[8] * X[2] + Rt[11];
}

// Kernel 2: positive-depth test of every point under each of the 4 candidates.
__global__ __launch_bounds__(256) void w8pt_triangulate(W8Params p) {
    const int b = blockIdx.y, c = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    __shared__ double sRt[12];
    if (threadIdx.x < 12) sRt[threadIdx.x] = p.cands[(int64_t)b * 48 + c * 12 + threadIdx.x];
    __syncthreads();
    bool ok = false;
    if (i < (p.n_per ? p.n_per[b] : p.N)) {
        const float* a = p.k0n + ((int64_t)b * p.N + i) * 2;
        const float* q = p.k1n + ((int64_t)b * p.N + i) * 2;
        double d0, d1;
        triangulate(a[0], a[1], q[0], q[1], sRt, d0, d1);
        ok = d0 > 0.0 && d1 > 0.0;
        p.pd_c[((int64_t)b * 4 + c) * p.N + i] = ok ? 1 : 0;
    }
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&p.counts[b * 4 + c], __popcll(m));
}

// Kernel 3: pick the pose, emit T, masks.
__global__ __launch_bounds__(256) void w8pt_select(W8Params p) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    int sel = p.sel[b];
    if (sel == -1) {  // cheirality vote: first maximum
        sel = 0;
        int best = p.counts[b * 4];
        for (int c = 1; c < 4; ++c)
            if (p.counts[b * 4 + c] > best) { best = p.counts[b * 4 + c]; sel = c; }
    }
    const double* cd = p.cands + (int64_t)b * 48 + (sel >= 0 ? sel : 0) * 12;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float* T = p.T + (int64_t)b * 16;
        bool fin = true;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) T[r * 4 + c] = sel >= 0 ? (float)cd[r * 3 + c] : (r == c ? 1.f : 0.f);
            T[r * 4 + 3] = sel >= 0 ? (float)cd[9 + r] : 0.f;
            for (int c = 0; c < 4; ++c) fin = fin && isfinite(T[r * 4 + c]);
        }
        T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
        if (!fin && p.status) p.status[b] |= 2;
    }
    if (i >= p.N) return;
    if (p.n_per && i >= p.n_per[b]) {  // padding row of a ragged batch
        p.posdepth[(int64_t)b * p.N + i] = 0;
        if (p.determine_inliers) p.inliers[(int64_t)b * p.N + i] = 0;
        return;
    }
    bool pos;
    if (sel >= 0) {
        pos = p.pd_c[((int64_t)b * 4 + sel) * p.N + i] != 0;
    } else {  // identity pose (choose_closest accepted nothing): triangulate against [I|0]
        const double Rt[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
        const float* a = p.k0n + ((int64_t)b * p.N + i) * 2;
        const float* q = p.k1n + ((int64_t)b * p.N + i) * 2;
        double d0, d1;
        triangulate(a[0], a[1], q[0], q[1], Rt, d0, d1);
        pos = d0 > 0.0 && d1 > 0.0;
    }
    p.posdepth[(int64_t)b * p.N + i] = pos ? 1 : 0;
    if (p.determine_inliers) {
        // symmetrical_epipolar_distance (squared) -> sqrt, threshold 3 px / mean focal (:121-125)
        const double* E = p.E + (int64_t)b * 9;
        const float* K0 = p.K0 + (p.intr_batch == 1 ? 0 : (int64_t)b * p.kdim * p.kdim);
        const float* K1 = p.K1 + (p.intr_batch == 1 ? 0 : (int64_t)b * p.kdim * p.kdim);
        const float thr = 3.f / ((K0[0] + K0[p.kdim + 1] + K1[0] + K1[p.kdim + 1]) / 4.f);
        const double x1 = p.k0n[((int64_t)b * p.N + i) * 2], y1 = p.k0n[((int64_t)b * p.N + i) * 2 + 1];
        const double x2 = p.k1n[((int64_t)b * p.N + i) * 2], y2 = p.k1n[((int64_t)b * p.N + i) * 2 + 1];
        const double l0 = E[0] * x1 + E[1] * y1 + E[2], l1 = E[3] * x1 + E[4] * y1 + E[5], l2 = E[6] * x1 + E[7] * y1 + E[8];
        const double m0 = E[0] * x2 + E[3] * y2 + E[6], m1 = E[1] * x2 + E[4] * y2 + E[7];
        const double num = (x2 * l0 + y2 * l1 + l2);
        const double sed = num * num * (1.0 / (l0 * l0 + l1 * l1) + 1.0 / (m0 * m0 + m1 * m1));
        p.inliers[(int64_t)b * p.N + i] = (pos && (float)sqrt(sed) <= thr) ? 1 : 0;
    }
}

__global__ void gather_matched_kernel(int N0, int N1, const float* k1, const int64_t* m, const float* conf, float* k1g,
                                      float* conf_out) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N0) return;
    int64_t j = m[(int64_t)b * N0 + i];
    const bool valid = j >= 0;
    if (j < 0) j += N1;  // python negative index: -1 -> last keypoint (its weight is 0)
    j = j < 0 ? 0 : (j >= N1 ? N1 - 1 : j);
    k1g[((int64_t)b * N0 + i) * 2] = k1[((int64_t)b * N1 + j) * 2];
    k1g[((int64_t)b * N0 + i) * 2 + 1] = k1[((int64_t)b * N1 + j) * 2 + 1];
    conf_out[(int64_t)b * N0 + i] = valid ? conf[(int64_t)b * N0 + i] : 0.f;
}

__global__ void pose_errors_kernel(int B, const float* T, const float* Tg, float* rot, float* tr) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* A = T + (int64_t)b * 16;
    const float* G = Tg + (int64_t)b * 16;
    float trc = 0.f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) trc += A[i * 4 + j] * G[i * 4 + j];
    const float ca = fminf(fmaxf((trc - 1.f) * 0.5f, -1.f), 1.f);
    rot[b] = fabsf(acosf(ca));
    const float n0 = sqrtf(A[3] * A[3] + A[7] * A[7] + A[11] * A[11]);
    const float n1 = sqrtf(G[3] * G[3] + G[7] * G[7] + G[11] * G[11]);
    float e = 0.f;
    if (n0 * n1 > 1e-6f) {
        const float cd = (A[3] * G[3] + A[7] * G[7] + A[11] * G[11]) / (n0 * n1);
        e = fabsf(acosf(fminf(fmaxf(cd, -1.f), 1.f)));
    }
    tr[b] = e;
}


// backward of pose_errors_kernel with respect to the FIRST pose: gT = g_rot d rot/dT + g_tr d tr/dT (zero where an angle is
// clamped or the translation norms fail the 1e-6 test - exactly the entries the forward leaves out of its means)
__global__ void pose_errors_backward_kernel(int B, const float* T, const float* Tg, const float* g_rot, const float* g_tr, float* gT) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* A = T + (int64_t)b * 16;
    const float* G = Tg + (int64_t)b * 16;
    float* o = gT + (int64_t)b * 16;
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    double trc = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) trc += (double)A[i * 4 + j] * G[i * 4 + j];
    const double ca = (trc - 1.0) * 0.5;
    if (ca > -1.0 && ca < 1.0) {
        const double f = -(double)g_rot[b] * 0.5 / sqrt(1.0 - ca * ca);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) o[i * 4 + j] = (float)(f * G[i * 4 + j]);
    }
    const double t0[3] = {A[3], A[7], A[11]}, t1[3] = {G[3], G[7], G[11]};
    const double n0 = sqrt(t0[0] * t0[0] + t0[1] * t0[1] + t0[2] * t0[2]), n1 = sqrt(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]);
    if ((float)n0 * (float)n1 > 1e-6f) {
        const double cd = (t0[0] * t1[0] + t0[1] * t1[1] + t0[2] * t1[2]) / (n0 * n1);
        if (cd > -1.0 && cd < 1.0) {
            const double f = -(double)g_tr[b] / sqrt(1.0 - cd * cd);
            for (int i = 0; i < 3; ++i) o[i * 4 + 3] = (float)(f * (t1[i] / (n0 * n1) - cd * t0[i] / (n0 * n0)));
        }
    }
}


// ---- all pairs of a tuple in ONE solve: get_kpts of every pair (i < j) into pair-major batches ----
struct TupleGatherParams {
    int B, T, N, kdim, intr_batch;
    const float* kpts[E2EMV_MAX_TUPLE];      // [B][N][2]
    const float* intr[E2EMV_MAX_TUPLE];      // [intr_batch][kdim][kdim]
    const int64_t* matches[kMaxGroups];      // pair (i, j): matches of image i in image j, [B][N]
    const float* conf[kMaxGroups];           // [B][N]
    const float* Tgt[kMaxGroups];            // [B][4][4] or null
    float* k0;                               // [P*B][N][2]
    float* k1g;                              // [P*B][N][2]
    float* cf;                               // [P*B][N]
    float* K0;                               // [P*B][kdim][kdim]
    float* K1;
    float* Tg;                               // [P*B][16] (choose_closest) or null
};

// grid (ceil(N/256), B, P); pair index q enumerates (i, j) with j outer, i inner (the reference's loop order)
__global__ __launch_bounds__(256) void tuple_gather_kernel(TupleGatherParams p) {
    const int b = blockIdx.y, q = blockIdx.z, n = blockIdx.x * 256 + threadIdx.x;
    int i = 0, j = 1;
    for (int c = 0; c < q; ++c) {
        if (++i == j) { i = 0; ++j; }
    }
    const int64_t ob = (int64_t)q * p.B + b;
    if (blockIdx.x == 0) {
        const int kk = p.kdim * p.kdim;
        const int64_t ib = p.intr_batch == 1 ? 0 : (int64_t)b * kk;
        if ((int)threadIdx.x < kk) {
            p.K0[ob * kk + threadIdx.x] = p.intr[i][ib + threadIdx.x];
            p.K1[ob * kk + threadIdx.x] = p.intr[j][ib + threadIdx.x];
        }
        if (p.Tg && threadIdx.x < 16) p.Tg[ob * 16 + threadIdx.x] = p.Tgt[q][(int64_t)b * 16 + threadIdx.x];
    }
    if (n >= p.N) return;
    const int64_t src = (int64_t)b * p.N + n, dst = ob * p.N + n;
    int64_t m = p.matches[q][src];
    const bool valid = m >= 0;
    if (m < 0) m += p.N;  // python negative index: -1 -> last keypoint (its weight is 0)
    m = m < 0 ? 0 : (m >= p.N ? p.N - 1 : m);
    p.k0[dst * 2] = p.kpts[i][src * 2];
    p.k0[dst * 2 + 1] = p.kpts[i][src * 2 + 1];
    p.k1g[dst * 2] = p.kpts[j][((int64_t)b * p.N + m) * 2];
    p.k1g[dst * 2 + 1] = p.kpts[j][((int64_t)b * p.N + m) * 2 + 1];
    p.cf[dst] = valid ? p.conf[q][src] : 0.f;
}

// normalize (estimate_relative_pose.py:9-14): pixel -> camera coordinates, fp32 like the reference
__global__ void normalize_kpts_kernel(int N, int kdim, int intr_batch, const float* kpts, const float* intr, float* out) {
    const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float* K = intr + (intr_batch == 1 ? 0 : (int64_t)b * kdim * kdim);
    const int64_t o = ((int64_t)b * N + n) * 2;
    out[o] = (kpts[o] - K[2]) / K[0];
    out[o + 1] = (kpts[o + 1] - K[kdim + 2]) / K[kdim + 1];
}

// means of the two angle errors: rotation over all B entries, translation over the entries whose norm product
// exceeds 1e-6 (compute_pose_error.py:12,22; an empty selection gives NaN like torch's mean of nothing)
__global__ __launch_bounds__(64) void pose_error_means_kernel(int B, const float* rot, const float* tr, const uint8_t* valid,
                                                              float* out2) {
    float sr = 0.f, st = 0.f, nv = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) {
        sr += rot[b];
        if (valid[b]) { st += tr[b]; nv += 1.f; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sr += __shfl_xor(sr, o);
        st += __shfl_xor(st, o);
        nv += __shfl_xor(nv, o);
    }
    if (threadIdx.x == 0) {
        out2[0] = sr / (float)B;
        out2[1] = st / nv;  // 0 / 0 = NaN
    }
}

// T_a_to_b = inv(pose_b) @ pose_a for general 4x4 matrices (helpers.py:219, 254): Gauss-Jordan with partial pivoting in
// fp64, rounded once to fp32
__global__ void relative_pose_kernel(int B, const float* pose_a, const float* pose_b, float* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double M[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            M[r][c] = pose_b[(int64_t)b * 16 + r * 4 + c];
            M[r][4 + c] = pose_a[(int64_t)b * 16 + r * 4 + c];
        }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r)
            if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
        if (piv != c)
            for (int k = 0; k < 8; ++k) { const double t = M[c][k]; M[c][k] = M[piv][k]; M[piv][k] = t; }
        const double d = 1.0 / M[c][c];
        for (int k = 0; k < 8; ++k) M[c][k] *= d;
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const double f = M[r][c];
                for (int k = 0; k < 8; ++k) M[r][k] -= f * M[c][k];
            }
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[(int64_t)b * 16 + r * 4 + c] = (float)M[r][4 + c];
}

__global__ void apply_mask_kernel(int64_t n, const float* x, const uint8_t* mask, float* out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = mask[i] ? x[i] : 0.f;
}

__global__ void pose_error_valid_kernel(int B, const float* T, const float* Tg, uint8_t* valid) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* A = T + (int64_t)b * 16;
    const float* G = Tg + (int64_t)b * 16;
    const float n0 = sqrtf(A[3] * A[3] + A[7] * A[7] + A[11] * A[11]);
    const float n1 = sqrtf(G[3] * G[3] + G[7] * G[7] + G[11] * G[11]);
    valid[b] = n0 * n1 > 1e-6f ? 1 : 0;
}

}  // namespace e2emv

using namespace e2emv;

extern "C" int e2emv_gather_matched(e2emv_ctx* ctx, int B, int N0, int N1, const float* d_kpts1, const int64_t* d_matches,
                                    const float* d_conf, float* d_kpts1_g, float* d_conf_out, void* stream) {
    if (!ctx || !d_kpts1 || !d_matches || !d_conf || !d_kpts1_g || !d_conf_out) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (B <= 0 || N0 <= 0 || N1 <= 0) return set_err(ctx, E2EMV_ESHAPE, "gather_matched: empty problem");
    hipStream_t s = (hipStream_t)stream;
    prof_begin(ctx, PS_W8PT, s);
    hipLaunchKernelGGL(gather_matched_kernel, dim3((N0 + 255) / 256, B), dim3(256), 0, s, N0, N1, d_kpts1, d_matches, d_conf,
                       d_kpts1_g, d_conf_out);
    prof_end(ctx, s);
    E2EMV_CHECK_LAUNCH(ctx, "gather_matched_kernel");
    return E2EMV_OK;
}

static int run_w8pt(e2emv_ctx* ctx, char* ws, int B, int N, const float* d_kpts0, const float* d_kpts1, const float* d_intr0,
                    const float* d_intr1, int kdim, int intr_batch, const float* d_conf, int choose_closest, const float* d_T_gt,
                    int determine_inliers, float* d_T, float* d_kpts0n, float* d_kpts1n, float* d_conf_n, uint8_t* d_inliers,
                    uint8_t* d_posdepth, float* d_F, int32_t* d_status, hipStream_t s, const int32_t* d_n_per = nullptr) {
    auto al = [](size_t n) { return (n + 255) & ~size_t(255); };
    W8Params p{};
    p.B = B; p.N = N; p.kdim = kdim; p.intr_batch = intr_batch; p.n_per = d_n_per;
    p.k0 = d_kpts0; p.k1 = d_kpts1; p.K0 = d_intr0; p.K1 = d_intr1; p.conf = d_conf;
    p.choose_closest = choose_closest; p.Tgt = d_T_gt; p.determine_inliers = determine_inliers;
    p.T = d_T; p.k0n = d_kpts0n; p.k1n = d_kpts1n; p.conf_n = d_conf_n; p.inliers = d_inliers; p.posdepth = d_posdepth;
    p.F = d_F; p.status = d_status;
    char* w = ws;
    p.E = (double*)w; w += al((size_t)B * 9 * 8);
    p.cands = (double*)w; w += al((size_t)B * 48 * 8);
    p.sel = (int*)w; w += al((size_t)B * 4 * 4);
    p.counts = (int*)w; w += al((size_t)B * 4 * 4);
    p.pd_c = (uint8_t*)w;
    prof_begin(ctx, PS_W8PT, s);
    hipLaunchKernelGGL(w8pt_fundamental, dim3(B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(w8pt_triangulate, dim3((N + 255) / 256, B, 4), dim3(256), 0, s, p);
    hipLaunchKernelGGL(w8pt_select, dim3((N + 255) / 256, B), dim3(256), 0, s, p);
    prof_end(ctx, s);
    E2EMV_CHECK_LAUNCH(ctx, "w8pt kernels");
    return E2EMV_OK;
}

static size_t w8pt_ws_bytes(int B, int N) {
    auto al = [](size_t n) { return (n + 255) & ~size_t(255); };
    return al((size_t)B * 9 * 8) + al((size_t)B * 48 * 8) + 2 * al((size_t)B * 4 * 4) + al((size_t)B * 4 * N);
}

extern "C" int e2emv_w8pt(e2emv_ctx* ctx, int B, int N, const float* d_kpts0, const float* d_kpts1, const float* d_intr0,
                          const float* d_intr1, int kdim, int intr_batch, const float* d_conf, int choose_closest,
                          const float* d_T_gt, int determine_inliers, float* d_T, float* d_kpts0n, float* d_kpts1n,
                          float* d_conf_n, uint8_t* d_inliers, uint8_t* d_posdepth, float* d_F, int32_t* d_status,
                          void* stream) {
    if (!ctx || !d_kpts0 || !d_kpts1 || !d_intr0 || !d_intr1 || !d_conf || !d_T || !d_kpts0n || !d_kpts1n || !d_conf_n ||
        !d_posdepth)
        return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (choose_closest && !d_T_gt) return set_err(ctx, E2EMV_EINVAL, "w8pt: choose_closest needs T_021");
    if (determine_inliers && !d_inliers) return set_err(ctx, E2EMV_EINVAL, "w8pt: determine_inliers needs an output buffer");
    if (B <= 0) return set_err(ctx, E2EMV_ESHAPE, "w8pt: empty batch");
    if (N < 8) return set_err(ctx, E2EMV_ESHAPE, "w8pt: fewer than 8 correspondences (N=%d)", N);
    if (kdim != 3 && kdim != 4) return set_err(ctx, E2EMV_ESHAPE, "w8pt: intrinsics must be 3x3 or 4x4");
    hipStream_t s = (hipStream_t)stream;
    int rc = ws_reserve(ctx, w8pt_ws_bytes(B, N));
    if (rc) return rc;
    return run_w8pt(ctx, ctx->d_ws, B, N, d_kpts0, d_kpts1, d_intr0, d_intr1, kdim, intr_batch, d_conf, choose_closest, d_T_gt,
                    determine_inliers, d_T, d_kpts0n, d_kpts1n, d_conf_n, d_inliers, d_posdepth, d_F, d_status, s);
}

extern "C" int e2emv_w8pt_backward(e2emv_ctx* ctx, int B, int N, const float* d_kpts0n, const float* d_kpts1n, const float* d_conf, const float* d_T,
                                   const float* d_gT, float* d_gconf, void* stream) {
    if (!ctx || !d_kpts0n || !d_kpts1n || !d_conf || !d_T || !d_gT || !d_gconf) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (B <= 0 || N < 8) return set_err(ctx, E2EMV_ESHAPE, "w8pt_backward: B=%d N=%d", B, N);
    (void)hipSetDevice(ctx->device);
    W8BwdParams p{};
    p.B = B; p.N = N; p.k0n = d_kpts0n; p.k1n = d_kpts1n; p.conf = d_conf; p.T = d_T; p.gT = d_gT; p.gconf = d_gconf;
    hipLaunchKernelGGL(w8pt_backward_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, p);
    E2EMV_CHECK_LAUNCH(ctx, "w8pt_backward_kernel");
    return E2EMV_OK;
}

extern "C" int e2emv_w8pt_ragged(e2emv_ctx* ctx, int B, int N, const int32_t* d_n_per, const float* d_kpts0, const float* d_kpts1,
                                 const float* d_intr0, const float* d_intr1, int kdim, int intr_batch, const float* d_conf,
                                 int choose_closest, const float* d_T_gt, int determine_inliers, float* d_T, float* d_kpts0n,
                                 float* d_kpts1n, float* d_conf_n, uint8_t* d_inliers, uint8_t* d_posdepth, float* d_F,
                                 int32_t* d_status, void* stream) {
    if (!ctx || !d_n_per || !d_kpts0 || !d_kpts1 || !d_intr0 || !d_intr1 || !d_conf || !d_T || !d_kpts0n || !d_kpts1n ||
        !d_conf_n || !d_posdepth)
        return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (choose_closest && !d_T_gt) return set_err(ctx, E2EMV_EINVAL, "w8pt_ragged: choose_closest needs T_021");
    if (determine_inliers && !d_inliers) return set_err(ctx, E2EMV_EINVAL, "w8pt_ragged: determine_inliers needs an output buffer");
    if (B <= 0) return set_err(ctx, E2EMV_ESHAPE, "w8pt_ragged: empty batch");
    if (N < 8) return set_err(ctx, E2EMV_ESHAPE, "w8pt_ragged: fewer than 8 correspondences (N=%d)", N);
    if (kdim != 3 && kdim != 4) return set_err(ctx, E2EMV_ESHAPE, "w8pt_ragged: intrinsics must be 3x3 or 4x4");
    hipStream_t s = (hipStream_t)stream;
    int rc = ws_reserve(ctx, w8pt_ws_bytes(B, N));
    if (rc) return rc;
    return run_w8pt(ctx, ctx->d_ws, B, N, d_kpts0, d_kpts1, d_intr0, d_intr1, kdim, intr_batch, d_conf, choose_closest, d_T_gt,
                    determine_inliers, d_T, d_kpts0n, d_kpts1n, d_conf_n, d_inliers, d_posdepth, d_F, d_status, s, d_n_per);
}

extern "C" int e2emv_w8pt_tuple(e2emv_ctx* ctx, int B, int T, int N, const float* const* d_kpts, const float* const* d_intr,
                                int kdim, int intr_batch, const int64_t* const* d_matches, const float* const* d_conf,
                                int choose_closest, const float* const* d_T_gt, int determine_inliers, float* d_T,
                                float* d_kpts0n, float* d_kpts1n, float* d_conf_n, uint8_t* d_inliers, uint8_t* d_posdepth,
                                float* d_F, int32_t* d_status, void* stream) {
    if (!ctx || !d_kpts || !d_intr || !d_matches || !d_conf || !d_T || !d_kpts0n || !d_kpts1n || !d_conf_n || !d_posdepth)
        return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (B <= 0 || T < 2 || T > E2EMV_MAX_TUPLE) return set_err(ctx, E2EMV_ESHAPE, "w8pt_tuple: batch=%d tuple_size=%d", B, T);
    if (N < 8) return set_err(ctx, E2EMV_ESHAPE, "w8pt_tuple: fewer than 8 correspondences (N=%d)", N);
    if (kdim != 3 && kdim != 4) return set_err(ctx, E2EMV_ESHAPE, "w8pt_tuple: intrinsics must be 3x3 or 4x4");
    if (intr_batch != 1 && intr_batch != B) return set_err(ctx, E2EMV_ESHAPE, "w8pt_tuple: intr_batch must be 1 or B");
    if (choose_closest && !d_T_gt) return set_err(ctx, E2EMV_EINVAL, "w8pt_tuple: choose_closest needs T_021 per pair");
    if (determine_inliers && !d_inliers) return set_err(ctx, E2EMV_EINVAL, "w8pt_tuple: determine_inliers needs an output buffer");
    const int P = T * (T - 1) / 2;
    const int PB = P * B;
    TupleGatherParams g{};
    g.B = B; g.T = T; g.N = N; g.kdim = kdim; g.intr_batch = intr_batch;
    for (int t = 0; t < T; ++t) {
        if (!d_kpts[t] || !d_intr[t]) return set_err(ctx, E2EMV_EINVAL, "w8pt_tuple: null input for image %d", t);
        g.kpts[t] = d_kpts[t];
        g.intr[t] = d_intr[t];
    }
    for (int q = 0; q < P; ++q) {
        if (!d_matches[q] || !d_conf[q] || (choose_closest && !d_T_gt[q]))
            return set_err(ctx, E2EMV_EINVAL, "w8pt_tuple: null input for pair %d", q);
        g.matches[q] = d_matches[q];
        g.conf[q] = d_conf[q];
        g.Tgt[q] = choose_closest ? d_T_gt[q] : nullptr;
    }
    hipStream_t s = (hipStream_t)stream;
    auto al = [](size_t n) { return (n + 255) & ~size_t(255); };
    const size_t sz_k = al((size_t)PB * N * 2 * 4), sz_c = al((size_t)PB * N * 4), sz_K = al((size_t)PB * kdim * kdim * 4),
                 sz_T = al((size_t)PB * 16 * 4);
    int rc = ws_reserve(ctx, 2 * sz_k + sz_c + 2 * sz_K + sz_T + w8pt_ws_bytes(PB, N));
    if (rc) return rc;
    char* w = ctx->d_ws;
    g.k0 = (float*)w; w += sz_k;
    g.k1g = (float*)w; w += sz_k;
    g.cf = (float*)w; w += sz_c;
    g.K0 = (float*)w; w += sz_K;
    g.K1 = (float*)w; w += sz_K;
    g.Tg = choose_closest ? (float*)w : nullptr; w += sz_T;
    prof_begin(ctx, PS_W8PT, s);
    hipLaunchKernelGGL(tuple_gather_kernel, dim3((N + 255) / 256, B, P), dim3(256), 0, s, g);
    prof_end(ctx, s);
    E2EMV_CHECK_LAUNCH(ctx, "tuple_gather_kernel");
    return run_w8pt(ctx, w, PB, N, g.k0, g.k1g, g.K0, g.K1, kdim, PB, g.cf, choose_closest, g.Tg, determine_inliers, d_T,
                    d_kpts0n, d_kpts1n, d_conf_n, d_inliers, d_posdepth, d_F, d_status, s);
}

extern "C" int e2emv_normalize_kpts(e2emv_ctx* ctx, int B, int N, const float* d_kpts, const float* d_intr, int kdim,
                                    int intr_batch, float* d_out, void* stream) {
    if (!ctx || !d_kpts || !d_intr || !d_out) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (B <= 0 || N <= 0) return set_err(ctx, E2EMV_ESHAPE, "normalize_kpts: empty problem");
    if ((kdim != 3 && kdim != 4) || (intr_batch != 1 && intr_batch != B))
        return set_err(ctx, E2EMV_ESHAPE, "normalize_kpts: intrinsics must be [1|B] x 3x3 or 4x4");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(normalize_kpts_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, N, kdim, intr_batch, d_kpts, d_intr, d_out);
    E2EMV_CHECK_LAUNCH(ctx, "normalize_kpts_kernel");
    return E2EMV_OK;
}

extern "C" int e2emv_apply_mask(e2emv_ctx* ctx, int64_t n, const float* d_x, const uint8_t* d_mask, float* d_out, void* stream) {
    if (!ctx || !d_x || !d_mask || !d_out) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (n <= 0) return set_err(ctx, E2EMV_ESHAPE, "apply_mask: empty input");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(apply_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, d_x, d_mask, d_out);
    E2EMV_CHECK_LAUNCH(ctx, "apply_mask_kernel");
    return E2EMV_OK;
}

extern "C" int e2emv_relative_pose(e2emv_ctx* ctx, int B, const float* d_pose_a, const float* d_pose_b, float* d_T_a2b,
                                   void* stream) {
    if (!ctx || !d_pose_a || !d_pose_b || !d_T_a2b) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (B <= 0) return set_err(ctx, E2EMV_ESHAPE, "relative_pose: empty batch");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(relative_pose_kernel, dim3((B + 63) / 64), dim3(64), 0, s, B, d_pose_a, d_pose_b, d_T_a2b);
    E2EMV_CHECK_LAUNCH(ctx, "relative_pose_kernel");
    return E2EMV_OK;
}

extern "C" int e2emv_pose_error_means(e2emv_ctx* ctx, int B, const float* d_T, const float* d_T_gt, float* d_rot_err,
                                      float* d_transl_err, uint8_t* d_transl_valid, float* d_means2, void* stream) {
    if (!ctx || !d_T || !d_T_gt || !d_rot_err || !d_transl_err || !d_transl_valid) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (B <= 0) return set_err(ctx, E2EMV_ESHAPE, "pose_error_means: empty batch");
    hipStream_t s = (hipStream_t)stream;
    prof_begin(ctx, PS_W8PT, s);
    hipLaunchKernelGGL(pose_errors_kernel, dim3((B + 63) / 64), dim3(64), 0, s, B, d_T, d_T_gt, d_rot_err, d_transl_err);
    hipLaunchKernelGGL(pose_error_valid_kernel, dim3((B + 63) / 64), dim3(64), 0, s, B, d_T, d_T_gt, d_transl_valid);
    if (d_means2)
        hipLaunchKernelGGL(pose_error_means_kernel, dim3(1), dim3(64), 0, s, B, d_rot_err, d_transl_err, d_transl_valid, d_means2);
    prof_end(ctx, s);
    E2EMV_CHECK_LAUNCH(ctx, "pose error kernels");
    return E2EMV_OK;
}

extern "C" int e2emv_pose_errors_backward(e2emv_ctx* ctx, int B, const float* d_T, const float* d_T_gt, const float* d_g_rot, const float* d_g_transl,
                                          float* d_gT, void* stream) {
    if (!ctx || !d_T || !d_T_gt || !d_g_rot || !d_g_transl || !d_gT) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (B <= 0) return set_err(ctx, E2EMV_ESHAPE, "pose_errors_backward: empty batch");
    hipLaunchKernelGGL(pose_errors_backward_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, d_T, d_T_gt, d_g_rot, d_g_transl, d_gT);
    E2EMV_CHECK_LAUNCH(ctx, "pose_errors_backward_kernel");
    return E2EMV_OK;
}

extern "C" int e2emv_pose_errors(e2emv_ctx* ctx, int B, const float* d_T, const float* d_T_gt, float* d_rot_err,
                                 float* d_transl_err, void* stream) {
    if (!ctx || !d_T || !d_T_gt || !d_rot_err || !d_transl_err) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (B <= 0) return set_err(ctx, E2EMV_ESHAPE, "pose_errors: empty batch");
    hipStream_t s = (hipStream_t)stream;
    prof_begin(ctx, PS_W8PT, s);
    hipLaunchKernelGGL(pose_errors_kernel, dim3((B + 63) / 64), dim3(64), 0, s, B, d_T, d_T_gt, d_rot_err, d_transl_err);
    prof_end(ctx, s);
    E2EMV_CHECK_LAUNCH(ctx, "pose_errors_kernel");
    return E2EMV_OK;
}
