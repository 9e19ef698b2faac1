// Small fp64 dense helpers shared by pose.hip and ba2view.hip (device-only, header-inline).
#pragma once
#include <hip/hip_runtime.h>

namespace e2emv {

template <int N>
__device__ __forceinline__ void jacobi_static(double (&A)[N][N], double (&V)[N][N]) {
    // cyclic Jacobi, everything statically indexed -> registers.  A symmetric (full storage).
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, dg = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            dg += A[i][i] * A[i][i];
#pragma unroll
            for (int j = i + 1; j < N; ++j) off += A[i][j] * A[i][j];
        }
        if (off <= 1e-60 || off <= 1e-34 * dg) break;
#pragma unroll
        for (int p = 0; p < N - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                const double apq = A[p][q];
                if (fabs(apq) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                A[p][p] -= t * apq;
                A[q][q] += t * apq;
                A[p][q] = 0.0;
                A[q][p] = 0.0;
#pragma unroll
                for (int k = 0; k < N; ++k) {
                    if (k != p && k != q) {
                        const double akp = A[k][p], akq = A[k][q];
                        A[k][p] = A[p][k] = c * akp - s * akq;
                        A[k][q] = A[q][k] = s * akp + c * akq;
                    }
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
}


// DLT triangulation of one correspondence with P1 = [I|0], P2 = [R|t] (Rt = R row-major, t at 9..11):
// smallest right singular vector of the 4x4 system through the Jacobi eigen-decomposition of A^T A
// (kornia triangulate_points semantics incl. the 1e-8 de-homogenisation guard).
__device__ __forceinline__ void triangulate_xyz(double x1, double y1, double x2, double y2, const double* Rt, double* Xo) {
    double Ar[4][4];
    Ar[0][0] = -1.0; Ar[0][1] = 0.0; Ar[0][2] = x1; Ar[0][3] = 0.0;
    Ar[1][0] = 0.0; Ar[1][1] = -1.0; Ar[1][2] = y1; Ar[1][3] = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        Ar[2][j] = x2 * Rt[6 + j] - Rt[j];
        Ar[3][j] = y2 * Rt[6 + j] - Rt[3 + j];
    }
    Ar[2][3] = x2 * Rt[11] - Rt[9];
    Ar[3][3] = y2 * Rt[11] - Rt[10];
    double G[4][4], V[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) {
            const double v = Ar[0][i] * Ar[0][j] + Ar[1][i] * Ar[1][j] + Ar[2][i] * Ar[2][j] + Ar[3][i] * Ar[3][j];
            G[i][j] = v;
            G[j][i] = v;
        }
    jacobi_static<4>(G, V);
    int m = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (G[i][i] < G[m][m]) m = i;
    double X[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) X[i] = (m == 0) ? V[i][0] : (m == 1) ? V[i][1] : (m == 2) ? V[i][2] : V[i][3];
    const double sc = fabs(X[3]) > 1e-8 ? 1.0 / (X[3] + 1e-8) : 1.0;  // convert_points_from_homogeneous
    Xo[0] = X[0] * sc; Xo[1] = X[1] * sc; Xo[2] = X[2] * sc;
}

}  // namespace e2emv
