// Global pose initialisation of the multi-view back-end (SURVEY.md 8(f) "next" row 3) - HOST code.
//
// Replaces the reference's `ba_initializer` executable (pose_optimization/multi_view/bundle_adjustment/
// ba_init/src/ba_init.cpp:10-90, ba_initializer.cpp:7-23), which parses `ba_init_in.csv`, calls
// theia::RobustRotationEstimator::EstimateRotations and theia::LeastUnsquaredDeviationPositionEstimator::
// EstimatePositions (Theia 0.7 - un-vendored, absent here) and writes `ba_init_out.csv`.  The two estimators are
// restated from the papers the reference cites (eval_multi_view.py:24-27):
//   * Chatterjee & Govindu, "Efficient and Robust Large-Scale Rotation Averaging", ICCV 2013: tangent-space updates,
//     a few L1 steps (ADMM) followed by IRLS with the sigma = 5 deg robust weight sigma/(e^2+sigma^2)^2;
//   * Ozyesil & Singer, "Robust Camera Location Estimation by Convex Programming", CVPR 2015 (LUD):
//     min sum |c_j - c_i - s_ij R_i^T p_ij|_1  s.t. s_ij >= 1, first position fixed at the origin, solved by ADMM.
// Problems are tiny (<= 8 views, <= 28 pairs; the reference runs this on the CPU too), so everything is dense fp64 on
// the host.  The output is expressed in the frame of view 0 (identity rotation, origin).
// Wire format (ba_init.cpp:13-51, 58-75): input rows of 10 fields `id, R col-major` and 14 fields
// `id0, id1, R_021 col-major, position of camera 1 in camera 0`; output rows `R col-major (9), t = -R*position`,
// 12 significant digits.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <sstream>

#include "common.h"
#include "mv_host.h"

namespace e2emv {
namespace mv {

// ---- CSV tokenizer with the semantics of io/src/file_utils.cpp:3-25 (empty fields dropped) -----------------------
std::vector<std::string> split_by_char(const std::string& s, char c) {
    std::vector<std::string> out;
    size_t start = 0, end = s.find(c);
    while (end != std::string::npos) {
        if (start != end) out.push_back(s.substr(start, end - start));
        start = end + 1;
        end = s.find(c, start);
    }
    if (start != s.size()) out.push_back(s.substr(start));
    return out;
}

// ---- rotations (column-major 3x3 <-> angle-axis, the conventions of ceres/rotation.h) ----------------------------
void aa_to_R(const double* aa, double* R /* col-major */) {
    const double t2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
    if (t2 > 2.220446049250313e-16) {
        const double th = std::sqrt(t2), wx = aa[0] / th, wy = aa[1] / th, wz = aa[2] / th;
        const double c = std::cos(th), s = std::sin(th), k = 1.0 - c;
        R[0] = c + wx * wx * k;
        R[1] = wz * s + wx * wy * k;
        R[2] = -wy * s + wx * wz * k;
        R[3] = wx * wy * k - wz * s;
        R[4] = c + wy * wy * k;
        R[5] = wx * s + wy * wz * k;
        R[6] = wy * s + wx * wz * k;
        R[7] = -wx * s + wy * wz * k;
        R[8] = c + wz * wz * k;
    } else {  // first-order
        R[0] = 1; R[1] = aa[2]; R[2] = -aa[1];
        R[3] = -aa[2]; R[4] = 1; R[5] = aa[0];
        R[6] = aa[1]; R[7] = -aa[0]; R[8] = 1;
    }
}

void R_to_aa(const double* R /* col-major */, double* aa) {
    // via the unit quaternion (numerically safe near pi)
    const double m00 = R[0], m10 = R[1], m20 = R[2], m01 = R[3], m11 = R[4], m21 = R[5], m02 = R[6], m12 = R[7], m22 = R[8];
    double q[4];
    const double tr = m00 + m11 + m22;
    if (tr >= 0.0) {
        double t = std::sqrt(tr + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (m21 - m12) * t;
        q[2] = (m02 - m20) * t;
        q[3] = (m10 - m01) * t;
    } else {
        int i = 0;
        if (m11 > m00) i = 1;
        if (m22 > (i == 0 ? m00 : m11)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        auto M = [&](int r, int c) { return R[c * 3 + r]; };
        double t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
        q[i + 1] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (M(k, j) - M(j, k)) * t;
        q[j + 1] = (M(j, i) + M(i, j)) * t;
        q[k + 1] = (M(k, i) + M(i, k)) * t;
    }
    const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (s2 > 0.0) {
        const double s = std::sqrt(s2);
        const double two_theta = 2.0 * (q[0] < 0.0 ? std::atan2(-s, -q[0]) : std::atan2(s, q[0]));
        const double k = two_theta / s;
        aa[0] = q[1] * k; aa[1] = q[2] * k; aa[2] = q[3] * k;
    } else {
        aa[0] = 2.0 * q[1]; aa[1] = 2.0 * q[2]; aa[2] = 2.0 * q[3];
    }
}

static void mat3_mul(const double* A, const double* B, double* C) {  // col-major C = A B
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) C[c * 3 + r] = A[r] * B[c * 3] + A[3 + r] * B[c * 3 + 1] + A[6 + r] * B[c * 3 + 2];
}
static void mat3_T(const double* A, double* B) {
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) B[c * 3 + r] = A[r * 3 + c];
}

// ---- tiny dense helpers ----------------------------------------------------------------------------------------------
struct Dense {
    int rows = 0, cols = 0;
    std::vector<double> a;  // row-major
    Dense(int r, int c) : rows(r), cols(c), a(size_t(r) * c, 0.0) {}
    double& operator()(int r, int c) { return a[size_t(r) * cols + c]; }
    double operator()(int r, int c) const { return a[size_t(r) * cols + c]; }
};

static std::vector<double> mul(const Dense& A, const std::vector<double>& x) {
    std::vector<double> y(A.rows, 0.0);
    for (int r = 0; r < A.rows; ++r) {
        double s = 0;
        for (int c = 0; c < A.cols; ++c) s += A(r, c) * x[c];
        y[r] = s;
    }
    return y;
}
static std::vector<double> mul_T(const Dense& A, const std::vector<double>& y) {
    std::vector<double> x(A.cols, 0.0);
    for (int r = 0; r < A.rows; ++r)
        for (int c = 0; c < A.cols; ++c) x[c] += A(r, c) * y[r];
    return x;
}
static double norm(const std::vector<double>& v) {
    double s = 0;
    for (double x : v) s += x * x;
    return std::sqrt(s);
}

// Cholesky factor of the symmetric positive definite n x n matrix M (row-major, lower triangle returned in place)
static bool cholesky(std::vector<double>& M, int n) {
    for (int j = 0; j < n; ++j) {
        double d = M[size_t(j) * n + j];
        for (int k = 0; k < j; ++k) d -= M[size_t(j) * n + k] * M[size_t(j) * n + k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        M[size_t(j) * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = M[size_t(i) * n + j];
            for (int k = 0; k < j; ++k) s -= M[size_t(i) * n + k] * M[size_t(j) * n + k];
            M[size_t(i) * n + j] = s / d;
        }
    }
    return true;
}
static void chol_solve(const std::vector<double>& L, int n, std::vector<double>& b) {
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[size_t(i) * n + k] * b[k];
        b[i] = s / L[size_t(i) * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= L[size_t(k) * n + i] * b[k];
        b[i] = s / L[size_t(i) * n + i];
    }
}
// normal matrix A^T diag(w) A (+ tiny ridge only if needed for definiteness)
static std::vector<double> normal_matrix(const Dense& A, const double* w) {
    const int n = A.cols;
    std::vector<double> M(size_t(n) * n, 0.0);
    for (int r = 0; r < A.rows; ++r) {
        const double wr = w ? w[r] : 1.0;
        for (int i = 0; i < n; ++i) {
            const double ai = A(r, i) * wr;
            if (ai == 0.0) continue;
            for (int j = 0; j <= i; ++j) M[size_t(i) * n + j] += ai * A(r, j);
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) M[size_t(i) * n + j] = M[size_t(j) * n + i];
    return M;
}

// ADMM for  min |A x - b|_1  subject to  x[k] >= lb[k] for the `n_geq` trailing... (general: G x >= d with G rows
// selecting single variables).  Splitting  z = [A; G] x - [b; d]:  soft-threshold on the first block, projection onto
// z >= 0 on the second (Boyd et al. 2011, sec. 6.1 / 5.2).
struct AdmmOptions {
    int max_iterations = 1000;
    double rho = 1.0, alpha = 1.0, abs_tol = 1e-4, rel_tol = 1e-2;
};
static bool admm_l1(const Dense& A, const std::vector<double>& b, const std::vector<int>& geq_var,
                    const std::vector<double>& geq_val, const AdmmOptions& opt, std::vector<double>& x) {
    const int m1 = A.rows, m2 = int(geq_var.size()), m = m1 + m2, n = A.cols;
    Dense S(m, n);
    for (int r = 0; r < m1; ++r)
        for (int c = 0; c < n; ++c) S(r, c) = A(r, c);
    for (int k = 0; k < m2; ++k) S(m1 + k, geq_var[k]) = 1.0;
    std::vector<double> bs(m, 0.0);
    for (int r = 0; r < m1; ++r) bs[r] = b[r];
    for (int k = 0; k < m2; ++k) bs[m1 + k] = geq_val[k];
    std::vector<double> L = normal_matrix(S, nullptr);
    if (!cholesky(L, n)) return false;
    std::vector<double> z(m, 0.0), u(m, 0.0), zold(m), rhs(m), sx, axh(m);
    const double kappa = 1.0 / opt.rho;
    for (int it = 0; it < opt.max_iterations; ++it) {
        for (int r = 0; r < m; ++r) rhs[r] = bs[r] + z[r] - u[r];
        x = mul_T(S, rhs);
        chol_solve(L, n, x);
        sx = mul(S, x);
        zold = z;
        for (int r = 0; r < m; ++r) axh[r] = opt.alpha * sx[r] + (1.0 - opt.alpha) * (zold[r] + bs[r]);
        for (int r = 0; r < m; ++r) {
            const double v = axh[r] - bs[r] + u[r];
            if (r < m1)
                z[r] = v > kappa ? v - kappa : (v < -kappa ? v + kappa : 0.0);
            else
                z[r] = v > 0.0 ? v : 0.0;
            u[r] += axh[r] - z[r] - bs[r];
        }
        // primal / dual residuals
        double rn = 0, zn = 0, sxn = 0;
        std::vector<double> dz(m), ru(m);
        for (int r = 0; r < m; ++r) {
            const double pr = sx[r] - z[r] - bs[r];
            rn += pr * pr;
            zn += z[r] * z[r];
            sxn += sx[r] * sx[r];
            dz[r] = -opt.rho * (z[r] - zold[r]);
            ru[r] = opt.rho * u[r];
        }
        const double s_norm = norm(mul_T(S, dz)), r_norm = std::sqrt(rn);
        const double eps_pri = std::sqrt(double(m)) * opt.abs_tol + opt.rel_tol * std::max({std::sqrt(sxn), std::sqrt(zn), norm(bs)});
        const double eps_dual = std::sqrt(double(n)) * opt.abs_tol + opt.rel_tol * norm(mul_T(S, ru));
        if (r_norm < eps_pri && s_norm < eps_dual) break;
    }
    return true;
}

// Connected components of the view graph.  A view that shares no pair with the rest (an image without matches) is its own
// component: it cannot be estimated and keeps its initial rotation / the origin; each component gets its own gauge view.
static std::vector<int> components(int n_views, const std::vector<Pair>& pairs) {
    std::vector<int> root(n_views);
    for (int v = 0; v < n_views; ++v) root[v] = v;
    auto find = [&](int v) { while (root[v] != v) v = root[v] = root[root[v]]; return v; };
    for (const Pair& e : pairs) root[find(e.i)] = find(e.j);
    for (int v = 0; v < n_views; ++v) root[v] = find(v);
    return root;
}
// column index of every view in the linear systems (-1 = gauge of its component); highest / lowest id of a component is
// the gauge when `highest` is set / cleared
static std::vector<int> gauge_columns(int n_views, const std::vector<Pair>& pairs, bool highest, int& n_free) {
    const std::vector<int> comp = components(n_views, pairs);
    std::vector<int> gauge(n_views, -1), col(n_views, -1);
    for (int v = 0; v < n_views; ++v) {
        int& g = gauge[comp[v]];
        if (g < 0 || (highest ? v > g : v < g)) g = v;
    }
    n_free = 0;
    for (int v = 0; v < n_views; ++v)
        if (gauge[comp[v]] != v) col[v] = n_free++;
    return col;
}

// ---- robust rotation averaging -----------------------------------------------------------------------------------
struct RotAvgOptions {
    int max_l1_steps = 5, max_irls_steps = 100;
    double l1_step_tol = 1e-3, irls_step_tol = 1e-3, sigma = 5.0 * M_PI / 180.0;
};

// rotations: angle-axis per view (in/out, index = view id 0..n-1); pairs: (i, j, angle-axis of R_ij with R_j = R_ij R_i)
bool estimate_rotations(int n_views, const std::vector<Pair>& pairs, std::vector<double>& rot) {
    RotAvgOptions opt;
    // Gauge: the rotation of the LAST view (of every connected component) is held at its initial value.  Theia holds "the
    // first view of its hash map"; the reference's gtests (test_ba_init.cpp:95-180, absolute comparisons under noise) pin
    // that to the last id.
    int n_free = 0;
    const std::vector<int> colv = gauge_columns(n_views, pairs, true, n_free);
    auto col = [&](int v) { return colv[v]; };
    const int E = int(pairs.size()), nu = 3 * n_free;
    if (n_views < 2 || E == 0 || n_free == 0) return n_views >= 1;
    Dense A(3 * E, nu);
    for (int e = 0; e < E; ++e)
        for (int d = 0; d < 3; ++d) {
            if (col(pairs[e].i) >= 0) A(3 * e + d, 3 * col(pairs[e].i) + d) = -1.0;
            if (col(pairs[e].j) >= 0) A(3 * e + d, 3 * col(pairs[e].j) + d) = 1.0;
        }
    std::vector<double> Rm(size_t(9) * n_views), res(3 * E), step(nu, 0.0);
    auto refresh = [&]() {
        for (int v = 0; v < n_views; ++v) aa_to_R(&rot[3 * v], &Rm[9 * v]);
    };
    auto residuals = [&]() {  // log(R_j^T R_ij R_i)
        for (int e = 0; e < E; ++e) {
            double Rij[9], T1[9], RjT[9], loop[9];
            aa_to_R(pairs[e].rot, Rij);
            mat3_mul(Rij, &Rm[9 * pairs[e].i], T1);
            mat3_T(&Rm[9 * pairs[e].j], RjT);
            mat3_mul(RjT, T1, loop);
            R_to_aa(loop, &res[3 * e]);
        }
    };
    auto apply = [&]() -> double {  // R_v <- R_v exp(step_v); returns the mean step angle
        double avg = 0;
        for (int v = 0; v < n_views; ++v) {
            if (col(v) < 0) continue;
            const double* sv = &step[3 * col(v)];
            double dR[9], Rn[9];
            aa_to_R(sv, dR);
            mat3_mul(&Rm[9 * v], dR, Rn);
            R_to_aa(Rn, &rot[3 * v]);
            avg += std::sqrt(sv[0] * sv[0] + sv[1] * sv[1] + sv[2] * sv[2]);
        }
        refresh();
        return avg / n_free;
    };
    refresh();
    residuals();
    // stage 1: L1 steps (robust to outlier pairs even far from the optimum)
    AdmmOptions ao;
    ao.max_iterations = 5;
    for (int it = 0; it < opt.max_l1_steps; ++it) {
        std::fill(step.begin(), step.end(), 0.0);
        if (!admm_l1(A, res, {}, {}, ao, step)) return false;
        const double avg = apply();
        residuals();
        if (avg <= opt.l1_step_tol) break;
        ao.max_iterations *= 2;
    }
    // stage 2: IRLS
    std::vector<double> w(3 * E);
    for (int it = 0; it < opt.max_irls_steps; ++it) {
        for (int e = 0; e < E; ++e) {
            const double e2 = res[3 * e] * res[3 * e] + res[3 * e + 1] * res[3 * e + 1] + res[3 * e + 2] * res[3 * e + 2];
            const double t = e2 + opt.sigma * opt.sigma;
            w[3 * e] = w[3 * e + 1] = w[3 * e + 2] = opt.sigma / (t * t);
        }
        std::vector<double> M = normal_matrix(A, w.data());
        if (!cholesky(M, nu)) return false;
        std::vector<double> wr(3 * E);
        for (int r = 0; r < 3 * E; ++r) wr[r] = w[r] * res[r];
        step = mul_T(A, wr);
        chol_solve(M, nu, step);
        const double avg = apply();
        residuals();
        if (avg <= opt.irls_step_tol) break;
    }
    return true;
}

// ---- least-unsquared-deviation positions --------------------------------------------------------------------------
// positions: 3 per view (out); view 0 at the origin.  Unknowns: positions of views 1.., one scale per pair (>= 1).
bool estimate_positions(int n_views, const std::vector<Pair>& pairs, const std::vector<double>& rot, std::vector<double>& pos) {
    // position gauge: the lowest view of every connected component sits at the origin (view 0 for a connected graph,
    // pinned by test_ba_init.cpp:183-266)
    int n_free = 0;
    const std::vector<int> colv = gauge_columns(n_views, pairs, false, n_free);
    auto col = [&](int v) { return colv[v]; };
    const int E = int(pairs.size()), np = 3 * n_free, nu = np + E;
    pos.assign(size_t(3) * n_views, 0.0);
    if (n_views < 2 || E == 0 || n_free == 0) return n_views >= 1;
    Dense A(3 * E, nu);
    for (int e = 0; e < E; ++e) {
        double Ri[9];
        aa_to_R(&rot[3 * pairs[e].i], Ri);
        // world-frame direction of the baseline: R_i^T p_ij  (R_i maps world -> camera i)
        const double* p = pairs[e].pos;
        const double dir[3] = {Ri[0] * p[0] + Ri[1] * p[1] + Ri[2] * p[2], Ri[3] * p[0] + Ri[4] * p[1] + Ri[5] * p[2],
                               Ri[6] * p[0] + Ri[7] * p[1] + Ri[8] * p[2]};
        for (int d = 0; d < 3; ++d) {
            if (col(pairs[e].i) >= 0) A(3 * e + d, 3 * col(pairs[e].i) + d) = -1.0;
            if (col(pairs[e].j) >= 0) A(3 * e + d, 3 * col(pairs[e].j) + d) = 1.0;
            A(3 * e + d, np + e) = -dir[d];
        }
    }
    std::vector<int> gv(E);
    std::vector<double> gd(E, 1.0), b(3 * E, 0.0), x(nu, 0.0);
    for (int e = 0; e < E; ++e) gv[e] = np + e;
    AdmmOptions ao;
    // iterate to the optimum of the convex programme (Theia stops ADMM at loose tolerances after <= 400 iterations; the
    // gtests' outlier case, test_ba_init.cpp:196-211, needs the exact-recovery property of the L1 optimum to 1e-4)
    ao.max_iterations = 4000;
    ao.abs_tol = 1e-10;
    ao.rel_tol = 1e-10;
    if (!admm_l1(A, b, gv, gd, ao, x)) return false;
    for (int v = 0; v < n_views; ++v)
        if (col(v) >= 0)
            for (int d = 0; d < 3; ++d) pos[3 * v + d] = x[3 * col(v) + d];
    return true;
}

// ---- the BaInit object of the reference (parse / Run / WriteResult) -----------------------------------------------
int run_init(int n_views, const double* init_R, int n_pairs, const int* pair_ids, const double* pair_R, const double* pair_pos,
             double* out_R, double* out_t) {
    std::vector<double> rot(size_t(3) * n_views);
    for (int v = 0; v < n_views; ++v) R_to_aa(init_R + 9 * v, &rot[3 * v]);
    std::vector<Pair> pairs(n_pairs);
    for (int e = 0; e < n_pairs; ++e) {
        pairs[e].i = pair_ids[2 * e];
        pairs[e].j = pair_ids[2 * e + 1];
        if (pairs[e].i < 0 || pairs[e].j < 0 || pairs[e].i >= n_views || pairs[e].j >= n_views || pairs[e].i == pairs[e].j) return 1;
        R_to_aa(pair_R + 9 * e, pairs[e].rot);
        std::memcpy(pairs[e].pos, pair_pos + 3 * e, 3 * sizeof(double));
    }
    int status = 0;
    if (!estimate_rotations(n_views, pairs, rot)) status |= 2;
    std::vector<double> pos;
    if (!estimate_positions(n_views, pairs, rot, pos)) status |= 4;
    // t = -R * position (ba_init.cpp:65), then the world frame is re-based onto camera 0 (R_v <- R_v R_0^T, positions
    // rotated by R_0, so t is unchanged): a pure gauge change that leaves every relative pose untouched and hands the
    // bundle adjuster - which treats camera 0 as the identity (ba_problem.cpp:129-137) - a consistent start.
    double R0[9], R0T[9];
    aa_to_R(&rot[0], R0);
    mat3_T(R0, R0T);
    for (int v = 0; v < n_views; ++v) {
        double R[9];
        aa_to_R(&rot[3 * v], R);
        for (int r = 0; r < 3; ++r) out_t[3 * v + r] = -(R[r] * pos[3 * v] + R[3 + r] * pos[3 * v + 1] + R[6 + r] * pos[3 * v + 2]);
        mat3_mul(R, R0T, out_R + 9 * v);
    }
    return status;
}

}  // namespace mv
}  // namespace e2emv

using namespace e2emv;

extern "C" int e2emv_mv_init(int n_views, const double* init_R, int n_pairs, const int32_t* pair_ids, const double* pair_R,
                             const double* pair_pos, double* out_R, double* out_t, int32_t* status) {
    if (n_views < 1 || n_views > 64 || n_pairs < 0 || !init_R || !out_R || !out_t || (n_pairs && (!pair_ids || !pair_R || !pair_pos)))
        return E2EMV_EINVAL;
    const int st = mv::run_init(n_views, init_R, n_pairs, pair_ids, pair_R, pair_pos, out_R, out_t);
    if (status) *status = st;
    return st == 1 ? E2EMV_EINVAL : E2EMV_OK;
}

extern "C" int e2emv_mv_estimate_rotations(int n_views, int n_pairs, const int32_t* pair_ids, const double* pair_rot_aa,
                                           double* rot_aa) {
    if (n_views < 1 || n_views > 64 || n_pairs < 0 || !rot_aa || (n_pairs && (!pair_ids || !pair_rot_aa))) return E2EMV_EINVAL;
    std::vector<mv::Pair> pairs(n_pairs);
    for (int e = 0; e < n_pairs; ++e) {
        pairs[e].i = pair_ids[2 * e];
        pairs[e].j = pair_ids[2 * e + 1];
        if (pairs[e].i < 0 || pairs[e].j < 0 || pairs[e].i >= n_views || pairs[e].j >= n_views || pairs[e].i == pairs[e].j) return E2EMV_EINVAL;
        std::memcpy(pairs[e].rot, pair_rot_aa + 3 * e, 3 * sizeof(double));
    }
    std::vector<double> rot(rot_aa, rot_aa + 3 * n_views);
    const bool ok = mv::estimate_rotations(n_views, pairs, rot);
    std::copy(rot.begin(), rot.end(), rot_aa);
    return ok ? E2EMV_OK : E2EMV_ESTATE;
}

extern "C" int e2emv_mv_estimate_positions(int n_views, int n_pairs, const int32_t* pair_ids, const double* pair_pos,
                                           const double* rot_aa, double* out_pos) {
    if (n_views < 1 || n_views > 64 || n_pairs < 0 || !rot_aa || !out_pos || (n_pairs && (!pair_ids || !pair_pos))) return E2EMV_EINVAL;
    std::vector<mv::Pair> pairs(n_pairs);
    for (int e = 0; e < n_pairs; ++e) {
        pairs[e].i = pair_ids[2 * e];
        pairs[e].j = pair_ids[2 * e + 1];
        if (pairs[e].i < 0 || pairs[e].j < 0 || pairs[e].i >= n_views || pairs[e].j >= n_views || pairs[e].i == pairs[e].j) return E2EMV_EINVAL;
        std::memcpy(pairs[e].pos, pair_pos + 3 * e, 3 * sizeof(double));
    }
    std::vector<double> rot(rot_aa, rot_aa + 3 * n_views), pos;
    const bool ok = mv::estimate_positions(n_views, pairs, rot, pos);
    std::copy(pos.begin(), pos.end(), out_pos);
    return ok ? E2EMV_OK : E2EMV_ESTATE;
}

extern "C" int e2emv_mv_init_files(const char* in_csv, const char* out_csv) {
    if (!in_csv || !out_csv) return E2EMV_EINVAL;
    std::ifstream file(in_csv);
    if (!file) return E2EMV_EINVAL;
    std::map<int, std::vector<double>> views;
    std::vector<int> ids;
    std::vector<double> pR, pp;
    std::string line;
    try {
        while (std::getline(file, line)) {
            const auto el = mv::split_by_char(line, ',');
            if (el.size() == 10) {
                std::vector<double> R(9);
                for (int i = 0; i < 9; ++i) R[i] = std::stod(el[i + 1]);
                views[std::stoi(el[0])] = R;
            } else if (el.size() == 14) {
                ids.push_back(std::stoi(el[0]));
                ids.push_back(std::stoi(el[1]));
                for (int i = 0; i < 9; ++i) pR.push_back(std::stod(el[i + 2]));
                for (int i = 0; i < 3; ++i) pp.push_back(std::stod(el[i + 11]));
            }
        }
    } catch (...) {
        return E2EMV_EINVAL;
    }
    const int n = int(views.size());
    std::vector<double> iR(size_t(9) * n), oR(size_t(9) * n), ot(size_t(3) * n);
    for (int v = 0; v < n; ++v) {
        auto it = views.find(v);
        if (it == views.end()) return E2EMV_EINVAL;  // WriteResult (ba_init.cpp:61-64) needs ids 0..n-1
        std::copy(it->second.begin(), it->second.end(), iR.begin() + 9 * v);
    }
    int32_t st = 0;
    const int rc = e2emv_mv_init(n, iR.data(), int(ids.size() / 2), ids.data(), pR.data(), pp.data(), oR.data(), ot.data(), &st);
    if (rc != E2EMV_OK) return rc;
    if (st & 2) printf("EstimateRotations failed.\n");
    if (st & 4) printf("EstimatePositions failed.\n");
    std::ofstream out(out_csv);
    if (!out) return E2EMV_EINVAL;
    for (int v = 0; v < n; ++v) {
        for (int i = 0; i < 9; ++i) out << std::setprecision(12) << oR[9 * v + i] << ",";
        out << std::setprecision(12) << ot[3 * v] << "," << ot[3 * v + 1] << "," << ot[3 * v + 2] << "\n";
    }
    return E2EMV_OK;
}
