// Device kernels of the training path (train.hip): a general strided fp32 GEMM on the fp32 matrix cores, the log-domain
// Sinkhorn with its reverse sweep, the row-wise pieces of the attention backward, and the small reductions.  First slice of
// SURVEY 8(f) / VERDICT r2 row (g): correctness first - every kernel here is a plain, readable form; none is tuned.
#pragma once
#include "common.h"

namespace e2emv {

typedef __attribute__((ext_vector_type(16))) float tr_f32x16;

// ---- C[z][m][n] (+)= alpha sum_k A(z; m, k) B(z; k, n): arbitrary element strides, two-level batch (z = z0 * inner + z1) ----
struct GG {
    int batch = 1, inner = 1;
    int M = 0, N = 0, K = 0;
    const float* A = nullptr;
    int64_t am = 0, ak = 0, az0 = 0, az1 = 0;
    const float* B = nullptr;
    int64_t bk = 0, bn = 0, bz0 = 0, bz1 = 0;
    float* C = nullptr;
    int64_t ldc = 0, cz0 = 0, cz1 = 0;
    float alpha = 1.f;
    int mode = 0;    // 0: C = ..., 1: C += ... (one writer per element), 2: atomicAdd (several K splits)
    int splits = 1;  // K is cut into `splits` ranges (mode 2)
};

constexpr int GG_T = 64, GG_BK = 32, GG_LD = 65;

// 64 x 64 tile, 4 waves of one 32 x 32 fp32-MFMA block each, 32-deep K steps.  The operands of step s + 1 are fetched into
// registers while step s is computed from LDS: with load -> LDS -> MFMA between two barriers and nothing in flight, one global
// round trip per step was exposed (backward of a 4-pair step 55 -> 33 ms).  Rows / columns beyond the matrix read as zero.
// (128-wide tiles - 2 x 2 MFMA blocks per wave, every LDS read used twice - run at 2 - 3 waves per SIMD and were SLOWER,
// 78 ms: this plain loop lives on its 8 waves per SIMD.)
__global__ __launch_bounds__(256) void gg_kernel(GG g) {
    __shared__ float As[GG_BK * GG_LD];
    __shared__ float Bs[GG_BK * GG_LD];
    const int zs = blockIdx.z;
    const int z = zs / g.splits, sp = zs - z * g.splits;
    const int z0 = z / g.inner, z1 = z - z0 * g.inner;
    const float* A = g.A + z0 * g.az0 + z1 * g.az1;
    const float* B = g.B + z0 * g.bz0 + z1 * g.bz1;
    float* C = g.C + z0 * g.cz0 + z1 * g.cz1;
    const int m0 = blockIdx.y * GG_T, n0 = blockIdx.x * GG_T;
    const int kper = ((g.K + g.splits - 1) / g.splits + GG_BK - 1) / GG_BK * GG_BK;
    const int k_begin = sp * kper, k_end = min(g.K, k_begin + kper);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
    tr_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // element (m, k) of pass i: threads run along whichever index is contiguous in memory
    const bool a_kc = g.ak == 1, b_kc = g.bk == 1;
    constexpr int NP = GG_T * GG_BK / 256;  // 8 values per thread and operand
    float ra[NP], rb[NP];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int m, k;
            if (a_kc) { k = t & 31; m = (t >> 5) + 8 * i; } else { m = t & 63; k = (t >> 6) + 4 * i; }
            ra[i] = (m0 + m < g.M && k0 + k < k_end) ? A[(int64_t)(m0 + m) * g.am + (int64_t)(k0 + k) * g.ak] : 0.f;
            int n, kb;
            if (b_kc) { kb = t & 31; n = (t >> 5) + 8 * i; } else { n = t & 63; kb = (t >> 6) + 4 * i; }
            rb[i] = (n0 + n < g.N && k0 + kb < k_end) ? B[(int64_t)(k0 + kb) * g.bk + (int64_t)(n0 + n) * g.bn] : 0.f;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int m, k;
            if (a_kc) { k = t & 31; m = (t >> 5) + 8 * i; } else { m = t & 63; k = (t >> 6) + 4 * i; }
            As[k * GG_LD + m] = ra[i];
            int n, kb;
            if (b_kc) { kb = t & 31; n = (t >> 5) + 8 * i; } else { n = t & 63; kb = (t >> 6) + 4 * i; }
            Bs[kb * GG_LD + n] = rb[i];
        }
    };
    if (k_begin < k_end) fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += GG_BK) {
        stash();
        __syncthreads();
        if (k0 + GG_BK < k_end) fetch(k0 + GG_BK);
#pragma unroll
        for (int kk = 0; kk < GG_BK / 2; ++kk) {
            const float a = As[(2 * kk + lh) * GG_LD + wm * 32 + l31];
            const float b = Bs[(2 * kk + lh) * GG_LD + wn * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int n = n0 + wn * 32 + l31;
    if (n >= g.N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= g.M) continue;
        float* c = C + (int64_t)m * g.ldc + n;
        const float v = g.alpha * acc[r];
        if (g.mode == 0) *c = v;
        else if (g.mode == 1) *c += v;
        else atomicAdd(c, v);
    }
}

// ---- small reductions / element-wise ----------------------------------------------------------------------------------
// out[n] (+)= sum_m X[m][n]  (grid: (ceil(N / 256), row chunks); out zeroed by the caller, atomicAdd over the chunks)
__global__ __launch_bounds__(256) void colsum_kernel(const float* X, int64_t M, int N, int64_t ld, float* out, int64_t rows_per) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int64_t m_begin = blockIdx.y * rows_per, m_end = min(M, m_begin + rows_per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four independent chains: the loads of a group are in flight together
    int64_t m = m_begin;
    for (; m + 4 <= m_end; m += 4) {
        s0 += X[m * ld + n]; s1 += X[(m + 1) * ld + n]; s2 += X[(m + 2) * ld + n]; s3 += X[(m + 3) * ld + n];
    }
    for (; m < m_end; ++m) s0 += X[m * ld + n];
    atomicAdd(out + n, (s0 + s1) + (s2 + s3));
}
// d[i] = h[i] > 0 ? d[i] : 0
__global__ __launch_bounds__(256) void relu_bwd_kernel(float* d, const float* h, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        if (!(h[i] > 0.f)) d[i] = 0.f;
}
// dst[i] += src[i]
__global__ __launch_bounds__(256) void add_kernel(float* dst, const float* src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] += src[i];
}
// dst [rows][ld_dst] += src [rows][ld_src], `cols` columns
__global__ __launch_bounds__(256) void add2d_kernel(float* dst, int64_t ld_dst, const float* src, int64_t ld_src, int64_t rows, int cols) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        dst[r * ld_dst + c] += src[r * ld_src + c];
    }
}

__device__ __forceinline__ float tr_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float tr_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- attention backward, row-wise pieces: P [z][nq][ld] (z = image * heads + head) --------------------------------------
// in place: P[row][j] = softmax_j(S[row][j]) over j < n_keys (S already scaled), 0 for j >= n_keys; one wave per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* P, int nq, int n_keys, int64_t ld, int64_t zstride) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nq) return;
    float* p = P + blockIdx.y * zstride + (int64_t)row * ld;
    float mx = -INFINITY;
    for (int j = lane; j < n_keys; j += 64) mx = fmaxf(mx, p[j]);
    mx = tr_wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < n_keys; j += 64) s += __expf(p[j] - mx);
    s = tr_wave_sum(s);
    const float inv = 1.f / s;
    for (int j = lane; j < ld; j += 64) p[j] = j < n_keys ? __expf(p[j] - mx) * inv : 0.f;
}
// in place on dP: dS = P (dP - sum_j P dP) * scale
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* P, float* dP, int nq, int n_keys, int64_t ld, int64_t zstride, float scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= nq) return;
    const float* p = P + blockIdx.y * zstride + (int64_t)row * ld;
    float* d = dP + blockIdx.y * zstride + (int64_t)row * ld;
    float s = 0.f;
    for (int j = lane; j < n_keys; j += 64) s += p[j] * d[j];
    s = tr_wave_sum(s);
    for (int j = lane; j < ld; j += 64) d[j] = j < n_keys ? p[j] * (d[j] - s) * scale : 0.f;
}

// ---- Sinkhorn in the log domain (upstream log_optimal_transport, one kernel per half iteration) and its reverse sweep ---
// couplings C [M+1][N+1]: C[i][j] = S[i][j] (i < M, j < N), alpha otherwise; log_mu_i = norm (i < M), log N + norm (i = M);
// log_nu_j = norm (j < N), log M + norm (j = N); norm = -log(M + N)
struct SkT {
    const float* S;  // [B][M][ldS]
    int64_t ldS;
    int M, N;
    float alpha, norm, logM, logN;
};
__device__ __forceinline__ float skt_c(const SkT& p, const float* Sb, int i, int j) { return (i < p.M && j < p.N) ? Sb[(int64_t)i * p.ldS + j] : p.alpha; }

// u[i] = log_mu_i - LSE_j(C[i][j] + v[j]); one wave per row i in [0, M]
__global__ __launch_bounds__(256) void skt_row_kernel(SkT p, const float* v, float* u, int64_t vstride, int64_t ustride) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, b = blockIdx.y;
    if (i > p.M) return;
    const float* Sb = p.S + (int64_t)b * p.M * p.ldS;
    const float* vb = v + b * vstride;
    float mx = -INFINITY;
    for (int j = lane; j <= p.N; j += 64) mx = fmaxf(mx, skt_c(p, Sb, i, j) + vb[j]);
    mx = tr_wave_max(mx);
    float s = 0.f;
    for (int j = lane; j <= p.N; j += 64) s += __expf(skt_c(p, Sb, i, j) + vb[j] - mx);
    s = tr_wave_sum(s);
    if (lane == 0) u[b * ustride + i] = (i < p.M ? p.norm : p.logN + p.norm) - (mx + __logf(s));
}
// v[j] = log_nu_j - LSE_i(C[i][j] + u[i]); a workgroup of 16 waves owns 64 columns: wave w walks the rows i = w, w + 16, ...
// (65 dependent steps instead of M + 1 - a thread per column was latency-bound: 0.5 ms per launch), lane = column; the 16
// partial (max, sum) pairs of a column are merged through LDS in a fixed order
template <int CW>  // columns per workgroup: 64 (lane = column, 16 row phases) or 32 (lane & 31 = column, 32 row phases: twice the
                    // workgroups - a training batch of 4 problems filled 68 of 256 CUs)
__global__ __launch_bounds__(1024) void skt_col_kernel(SkT p, const float* u, float* v, int64_t ustride, int64_t vstride) {
    constexpr int NPH = 16 * 64 / CW;
    __shared__ float smx[NPH][CW], ssm[NPH][CW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, b = blockIdx.y;
    const int cl = lane & (CW - 1), ph = wv * (64 / CW) + lane / CW;
    const int j = blockIdx.x * CW + cl;
    const float* Sb = p.S + (int64_t)b * p.M * p.ldS;
    const float* ub = u + b * ustride;
    float mx = -INFINITY, s = 0.f;
    if (j <= p.N)
        for (int i = ph; i <= p.M; i += NPH) {
            const float x = skt_c(p, Sb, i, j) + ub[i];
            if (x > mx) { s = s * __expf(mx - x) + 1.f; mx = x; } else { s += __expf(x - mx); }
        }
    smx[ph][cl] = mx;
    ssm[ph][cl] = s;
    __syncthreads();
    if (threadIdx.x < CW && j <= p.N) {
        float M = smx[0][cl], S = ssm[0][cl];
        for (int k = 1; k < NPH; ++k) {
            const float m2 = smx[k][cl], s2 = ssm[k][cl];
            const float nm = fmaxf(M, m2);
            S = (M == -INFINITY ? 0.f : S * __expf(M - nm)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - nm));
            M = nm;
        }
        v[b * vstride + j] = (j < p.N ? p.norm : p.logM + p.norm) - (M + __logf(S));
    }
}
// Z[i][j] = C[i][j] + u[i] + v[j] - norm, dense [B][M+1][N+1]
__global__ __launch_bounds__(256) void skt_out_kernel(SkT p, const float* u, const float* v, int64_t ustride, int64_t vstride, float* Z) {
    const int b = blockIdx.y;
    const int64_t per = (int64_t)(p.M + 1) * (p.N + 1);
    const float* Sb = p.S + (int64_t)b * p.M * p.ldS;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < per; e += (int64_t)gridDim.x * 256) {
        const int i = (int)(e / (p.N + 1)), j = (int)(e - (int64_t)i * (p.N + 1));
        Z[b * per + e] = skt_c(p, Sb, i, j) + u[b * ustride + i] + v[b * vstride + j] - p.norm;
    }
}
// reverse sweep, start: dC = G; du[i] = sum_j G[i][j]; dv[j] = sum_i G[i][j]   (grid: (M + 1 rows / 4, B); dv zeroed before, atomics)
__global__ __launch_bounds__(256) void skb_init_kernel(int M, int N, const float* G, float* dC, float* du, float* dv, int64_t ustride, int64_t vstride) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, b = blockIdx.y;
    if (i > M) return;
    const int64_t per = (int64_t)(M + 1) * (N + 1);
    const float* g = G + b * per + (int64_t)i * (N + 1);
    float* d = dC + b * per + (int64_t)i * (N + 1);
    float s = 0.f;
    for (int j = lane; j <= N; j += 64) {
        const float x = g[j];
        d[j] = x;
        s += x;
        if (x != 0.f) atomicAdd(dv + b * vstride + j, x);
    }
    s = tr_wave_sum(s);
    if (lane == 0) du[b * ustride + i] = s;
}
// v_t = log_nu - LSE_i(C + u_t):  P = exp(C + u_t[i] + v_t[j] - log_nu_j);  dC -= dv[j] P;  du[i] -= sum_j dv[j] P   (wave per row)
__global__ __launch_bounds__(256) void skb_vhalf_kernel(SkT p, const float* u, const float* v, const float* dv, float* du, float* dC,
                                                        int64_t ustride, int64_t vstride, int64_t dstride) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, b = blockIdx.y;
    if (i > p.M) return;
    const float* Sb = p.S + (int64_t)b * p.M * p.ldS;
    const float* vb = v + b * vstride;
    const float* dvb = dv + b * dstride;
    float* d = dC + (int64_t)b * (p.M + 1) * (p.N + 1) + (int64_t)i * (p.N + 1);
    const float ui = u[b * ustride + i];
    float s = 0.f;
    for (int j = lane; j <= p.N; j += 64) {
        const float g = dvb[j];
        if (g == 0.f) continue;
        const float P = __expf(skt_c(p, Sb, i, j) + ui + vb[j] - (j < p.N ? p.norm : p.logM + p.norm));
        d[j] -= g * P;
        s += g * P;
    }
    s = tr_wave_sum(s);
    if (lane == 0) du[b * dstride + i] -= s;
}
// u_t = log_mu - LSE_j(C + v_{t-1}):  P = exp(C + v_{t-1}[j] + u_t[i] - log_mu_i);  dC -= du[i] P;  dv_prev[j] = -sum_i du[i] P
// (16 waves x 64 columns like skt_col_kernel: wave w takes the rows i = w, w + 16, ...; column sums merged through LDS)
template <int CW>  // (as skt_col_kernel)
__global__ __launch_bounds__(1024) void skb_uhalf_kernel(SkT p, const float* u, const float* v_prev, const float* du, float* dv_prev, float* dC,
                                                         int64_t ustride, int64_t vstride, int64_t dstride) {
    constexpr int NPH = 16 * 64 / CW;
    __shared__ float part[NPH][CW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, b = blockIdx.y;
    const int cl = lane & (CW - 1), ph = wv * (64 / CW) + lane / CW;
    const int j = blockIdx.x * CW + cl;
    const float* Sb = p.S + (int64_t)b * p.M * p.ldS;
    const float* ub = u + b * ustride;
    const float* dub = du + b * dstride;
    float s = 0.f;
    if (j <= p.N) {
        float* d = dC + (int64_t)b * (p.M + 1) * (p.N + 1) + j;
        const float vj = v_prev[b * vstride + j];
        for (int i = ph; i <= p.M; i += NPH) {
            const float g = dub[i];
            const float P = __expf(skt_c(p, Sb, i, j) + vj + ub[i] - (i < p.M ? p.norm : p.logN + p.norm));
            d[(int64_t)i * (p.N + 1)] -= g * P;
            s += g * P;
        }
    }
    part[ph][cl] = s;
    __syncthreads();
    if (threadIdx.x < CW && j <= p.N) {
        float t = part[0][cl];
        for (int k = 1; k < NPH; ++k) t += part[k][cl];
        dv_prev[b * dstride + j] = -t;
    }
}
// d alpha += sum over the dustbin row and column of dC (all problems); one workgroup per problem
__global__ __launch_bounds__(256) void skb_alpha_kernel(int M, int N, const float* dC, float* dalpha) {
    __shared__ float red[4];
    const float* d = dC + (int64_t)blockIdx.x * (M + 1) * (N + 1);
    float s = 0.f;
    for (int i = threadIdx.x; i < M; i += 256) s += d[(int64_t)i * (N + 1) + N];
    for (int j = threadIdx.x; j <= N; j += 256) s += d[(int64_t)M * (N + 1) + j];
    s = tr_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dalpha, red[0] + red[1] + red[2] + red[3]);
}

// ---- confidence head (conf_mlp) of the training path --------------------------------------------------------------------
// feat[b][n][:] = [mdesc_i[b][n][:] | mdesc_j[b][max(match, 0)][:]]   (dense [B][N][2D]; mdesc rows n_rows apart per image)
__global__ __launch_bounds__(256) void conf_feat_kernel(int N, int D, const float* mdesc_i, const float* mdesc_j, int64_t tuple_stride,
                                                        const int64_t* matches, float* feat) {
    const int b = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    int64_t j = matches[(int64_t)b * N + row];
    if (j < 0) j = 0;
    const float* si = mdesc_i + b * tuple_stride + (int64_t)row * D;
    const float* sj = mdesc_j + b * tuple_stride + j * D;
    float* dst = feat + ((int64_t)b * N + row) * 2 * D;
    for (int c = lane; c < D; c += 64) { dst[c] = si[c]; dst[D + c] = sj[c]; }
}
// conf[r] = match[r] >= 0 ? sigmoid(<h[r], w1> + b1) : 0   (wave per row)
__global__ __launch_bounds__(256) void conf_fwd_kernel(int64_t rows, int D, const float* hid, const float* w1, const float* b1, const int64_t* matches,
                                                       float* conf) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* h = hid + r * D;
    float acc = 0.f;
    for (int c = lane; c < D; c += 64) acc += h[c] * w1[c];
    acc = tr_wave_sum(acc);
    if (lane == 0) conf[r] = matches[r] >= 0 ? 1.f / (1.f + __expf(-(acc + b1[0]))) : 0.f;
}
// conf = sigmoid(z), z = <h, w1> + b1:  dz[r] = valid ? g[r] sigma (1 - sigma) : 0;  dh[r][:] = dz[r] w1[:] (h > 0)   (wave per row)
__global__ __launch_bounds__(256) void conf_bwd_kernel(int64_t rows, int D, const float* hid, const float* w1, const float* b1, const int64_t* matches,
                                                       const float* gconf, float* dz, float* dh) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float* h = hid + r * D;
    float acc = 0.f;
    for (int c = lane; c < D; c += 64) acc += h[c] * w1[c];
    acc = tr_wave_sum(acc);
    const float sg = 1.f / (1.f + __expf(-(acc + b1[0])));
    const float d = matches[r] >= 0 ? gconf[r] * sg * (1.f - sg) : 0.f;
    if (lane == 0) dz[r] = d;
    for (int c = lane; c < D; c += 64) dh[r * D + c] = h[c] > 0.f ? d * w1[c] : 0.f;
}
// dmdesc_i[b][n][:] += dfeat[b][n][:D];  dmdesc_j[b][match][:] += dfeat[b][n][D:]  (matched rows only; mutual matches are unique,
// atomics keep it safe for any index list)
__global__ __launch_bounds__(256) void conf_scatter_kernel(int N, int D, const float* dfeat, const int64_t* matches, float* dmd_i, float* dmd_j,
                                                           int64_t tuple_stride) {
    const int b = blockIdx.y;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const int64_t j = matches[(int64_t)b * N + row];
    if (j < 0) return;
    const float* src = dfeat + ((int64_t)b * N + row) * 2 * D;
    float* di = dmd_i + b * tuple_stride + (int64_t)row * D;
    float* dj = dmd_j + b * tuple_stride + j * D;
    for (int c = lane; c < D; c += 64) { atomicAdd(di + c, src[c]); atomicAdd(dj + c, src[D + c]); }
}

// ---- folded gradients -> the gradients of the upstream parameters --------------------------------------------------------
// A convolution W [rows][cols] (+ bias) that was committed as  Wf[r'][c'] = s_r W[r][c],  bf[r'] = (b[r] - mean[r]) s_r + beta[r]
// with s = gamma / sqrt(var + eps) (eval-mode BatchNorm folded; s = 1 without one), r' = rmap[r], c' = cmap[c] (head-major
// re-ordering; identity when null).  One workgroup per row r:
//   dW[r][c] = s_r dWf[r'][c'],  db[r] = s_r dbf[r'],  dgamma[r] = (sum_c dWf[r'][c'] W[r][c] + dbf[r'] (b[r] - mean[r])) / sigma_r,  dbeta[r] = dbf[r']
struct UnfoldArgs {
    const float* dWf; const float* dbf;   // folded gradients
    int64_t ldwf;                         // row stride of dWf
    int col0;                             // first folded column of this convolution inside dWf's rows
    const float* W; const float* b;       // upstream parameters
    const float* gamma; const float* beta; const float* mean; const float* var;  // BatchNorm (all null: none)
    const int* rmap; const int* cmap;
    int rows, cols;
    float* dW; float* db; float* dgamma; float* dbeta;
};
__global__ __launch_bounds__(256) void unfold_kernel(UnfoldArgs a) {
    __shared__ float red[4];
    const int r = blockIdx.x, rf = a.rmap ? a.rmap[r] : r;
    float s = 1.f, sigma = 1.f;
    if (a.gamma) { sigma = sqrtf(a.var[r] + 1e-5f); s = a.gamma[r] / sigma; }
    const float* g = a.dWf + (int64_t)rf * a.ldwf + a.col0;
    float dot = 0.f;
    for (int c = threadIdx.x; c < a.cols; c += 256) {
        const float gv = g[a.cmap ? a.cmap[c] : c];
        a.dW[(int64_t)r * a.cols + c] = s * gv;
        if (a.gamma) dot += gv * a.W[(int64_t)r * a.cols + c];
    }
    if (!a.dbf) return;
    dot = tr_wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float gb = a.dbf[rf];
        if (a.db) a.db[r] = s * gb;
        if (a.gamma) {
            a.dgamma[r] = (red[0] + red[1] + red[2] + red[3] + gb * (a.b[r] - a.mean[r])) / sigma;
            a.dbeta[r] = gb;
        }
    }
}

// ---- upstream parameters -> folded training weights, on the device (e2emv_train_update): the inverse walk of unfold_kernel ----
// one workgroup per row r of conv weight W [rows][cols] (+ BatchNorm behind it): Wf[rmap(r)][col0 + cmap(c)] = W[r][c] g / sqrt(var + eps),
// bf[rmap(r)] = (b[r] - mean) g / sqrt(var + eps) + beta - the expressions and the fp64 of the host fold (train_build)
struct FoldArgs {
    const float* W; const float* b;
    const float* gamma; const float* beta; const float* mean; const float* var;
    float* Wf; float* bf;
    int64_t ldwf;
    int col0, rows, cols;
    const int* rmap; const int* cmap;
};
__global__ __launch_bounds__(256) void fold_kernel(FoldArgs a) {
    const int r = blockIdx.x, rf = a.rmap ? a.rmap[r] : r;
    double sc = 1.0;
    if (a.gamma) sc = (double)a.gamma[r] / sqrt((double)a.var[r] + 1e-5);
    float* o = a.Wf + (int64_t)rf * a.ldwf + a.col0;
    for (int c = threadIdx.x; c < a.cols; c += 256) {
        const float w = a.W[(int64_t)r * a.cols + c];
        o[a.cmap ? a.cmap[c] : c] = a.gamma ? (float)((double)w * sc) : w;
    }
    if (threadIdx.x == 0 && a.bf) a.bf[rf] = a.gamma ? (float)(((double)a.b[r] - (double)a.mean[r]) * sc + (double)a.beta[r]) : a.b[r];
}

}  // namespace e2emv
