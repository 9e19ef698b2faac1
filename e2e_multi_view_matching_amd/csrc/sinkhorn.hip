// Sinkhorn log-space optimal transport with dustbins + mutual-arg-max matching.
//
// Restates upstream SuperGlue `log_optimal_transport` / `log_sinkhorn_iterations` and the
// match block of `SuperGlue.forward` (superglue.py; the reference runs them inside its
// absent MultiViewMatcher.forward - call sites helpers.py:246, eval_pairs.py:212).
//
// HBM plan.  The reference (torch) runs, per iteration, two `Z + v` adds and two
// logsumexp's over the (N+1)^2 couplings: ~10 sweeps.  SURVEY.md 8(d)'s byte model charges
// 2 sweeps / iteration.  Here ONE sweep per iteration:
//   * the couplings matrix is never built: the dustbin row/column are the constant alpha, so
//     only the aligned core S [M][ldS] is streamed and the dustbin terms are added
//     analytically;
//   * a workgroup owns 16 full rows (4 waves x 4 rows, 64 floats per lane in registers):
//     it computes u for its rows (row LSE = wave shuffles only) and, FROM THE SAME REGISTERS,
//     the per-column partial (max, sum-exp) of S + u over its 16 rows (cross-wave through
//     LDS).  `sinkhorn_combine` (64 columns x 4 chunk ranges per workgroup) folds the M/16 partials into
//     v and the two dustbin scalars.
// Row loads are 16 B per lane, 1 KiB contiguous per wave instruction.
// The final sweep writes logZ = couplings + u + v + log(M+N) densely ([M+1][N+1], the API
// layout) and fuses the row/column arg-max needed by the match block, so Z is never re-read.
//
// That launch chain (round 1) is the fallback today.  The default path is a RESIDENT kernel: all iterations in one launch, in the
// exponential domain, K = exp(S - rowmax) kept on chip for the whole call, the workgroups of a problem exchanging column sums
// through tagged granules:
//   sinkhorn_resident<KT, ...>   8 waves, 32 / 64 rows per workgroup, K in 64 / 128 compiler-allocated registers per lane
//   sinkhorn_resident128         4 waves, 128 rows x <= 1024 columns: 24 of a wave's 32 rows in registers the kernel addresses
//                                by number (v64 - v255, a64 - a255; the compiler confined to 56), 8 in LDS - 32 problems resident
//   sinkhorn_resident2k          the same for <= 2048 columns, 64 rows per workgroup - 8 problems resident instead of 4
// followed by sinkhorn_rescue (problems that left fp32's range or gave up on a wait: re-solved in the log domain) and the final sweep.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "common.h"

namespace e2emv {

typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int SK_ROWS = 16;  // rows per workgroup (4 waves x 4 rows)

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block-wide (256 threads) max / sum through LDS scratch (>= 8 floats)
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fmaxf(r, red[i]);
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r += red[i];
    return r;
}

struct SkParams {
    const float* S;     // [B][M][ldS]
    int64_t ldS;
    int M, N;
    int chunks;         // ceil(M / SK_ROWS)
    float alpha;        // bin score
    float norm;         // -log(M+N)
    float* u;           // [B][M+1]
    float* v;           // [B][ldV]  (ldV = ldS + 4, v[N] = dustbin column)
    int64_t ldV;
    float* pm;          // [B][chunks][ldS] partial column max
    float* ps;          // [B][chunks][ldS] partial column sum-exp
    float* v_next;      // [B][ldV]  written by sinkhorn_combine (ping-pong with v)
    float* upm;         // [B][chunks] partial max of u over a chunk's rows
    float* ups;         // [B][chunks] partial sum-exp of u over a chunk's rows
    // final sweep
    float* logZ[kMaxGroups];  // per output group: [group_batch][M+1][N+1] or null
    int group_batch;    // batch elements per output group
    float* max0;        // [B][M] row max of the core (value of logZ)
    int* idx0;          // [B][M]
    float* pv;          // [B][chunks][ldS] partial column max value (final)
    int* pi;            // [B][chunks][ldS] partial column arg-max row (final)
};

// One sweep of S: u for 16 rows + column partials of S + u.  KT = ceil(ldS / 256).
// FULL = (N == ldS == KT*256): every lane owns valid columns only, so the per-element column guards (which
// hipcc turns into ~90 exec-mask branches) disappear - the case of the 256/512/1024/2048-keypoint configs.
template <int KT, bool FINAL, bool FULL>
__global__ __launch_bounds__(256) void sinkhorn_sweep(SkParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [4 waves][2][KT*256]
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = chunk * SK_ROWS + wave * 4;
    const float* Sb = p.S + (int64_t)b * p.M * p.ldS;
    const float* vb = p.v + (int64_t)b * p.ldV;

    if (FINAL && chunk == p.chunks) {
        // dustbin row of logZ: (alpha + u_M) + v_j + log(M+N)
        float* const zbase = p.logZ[b / p.group_batch];
        if (zbase) {
            const float uM = p.u[(int64_t)b * (p.M + 1) + p.M];
            float* zr = zbase + ((int64_t)(b % p.group_batch) * (p.M + 1) + p.M) * (p.N + 1);
            for (int j = tid; j <= p.N; j += 256) zr[j] = ((p.alpha + uM) + vb[j]) - p.norm;
        }
        return;
    }

    float z[4][KT][4];
    float vv[KT][4];
    int col[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        col[k] = 4 * (lane + 64 * k);
        f32x4 t = (FULL || col[k] < p.ldS) ? *reinterpret_cast<const f32x4*>(vb + col[k]) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) vv[k][e] = t[e];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = min(row0 + r, p.M - 1);
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            f32x4 t = (FULL || col[k] < p.ldS) ? *reinterpret_cast<const f32x4*>(Sb + (int64_t)row * p.ldS + col[k])
                                       : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) z[r][k][e] = t[e];
        }
    }
    const float vN = vb[p.N];
    float ur[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool rvalid = row0 + r < p.M;
        if (!FINAL) {
            // u_i = log_mu - LSE_j(S_ij + v_j  U  alpha + v_N)
            float mx = p.alpha + vN;
#pragma unroll
            for (int k = 0; k < KT; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (FULL || col[k] + e < p.N) mx = fmaxf(mx, z[r][k][e] + vv[k][e]);
            mx = wave_max(mx);
            float sm = 0.f;
#pragma unroll
            for (int k = 0; k < KT; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (FULL || col[k] + e < p.N) sm += __expf(z[r][k][e] + vv[k][e] - mx);
            sm = wave_sum(sm) + __expf(p.alpha + vN - mx);
            ur[r] = p.norm - (mx + __logf(sm));
            if (rvalid && lane == 0) p.u[(int64_t)b * (p.M + 1) + row0 + r] = ur[r];
        } else {
            ur[r] = p.u[(int64_t)b * (p.M + 1) + min(row0 + r, p.M - 1)];
        }
        if (!rvalid) ur[r] = -INFINITY;  // ragged last chunk: row does not exist
    }

    float* lm = lds + (wave * 2 + 0) * (KT * 256);
    float* ls = lds + (wave * 2 + 1) * (KT * 256);
    if (!FINAL) {
        // (max, sum-exp) of this wave's u values: feeds the dustbin column v_N in sinkhorn_combine
        const float uwm = fmaxf(fmaxf(ur[0], ur[1]), fmaxf(ur[2], ur[3]));
        const float uwm_s = (uwm == -INFINITY) ? 0.f : uwm;
        const float uws = __expf(ur[0] - uwm_s) + __expf(ur[1] - uwm_s) + __expf(ur[2] - uwm_s) + __expf(ur[3] - uwm_s);
        // column partials over this wave's 4 rows: (max, sum exp) of S_ij + u_i
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            f32x4 m4, s4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float y0 = z[0][k][e] + ur[0], y1 = z[1][k][e] + ur[1], y2 = z[2][k][e] + ur[2], y3 = z[3][k][e] + ur[3];
                float m = fmaxf(fmaxf(y0, y1), fmaxf(y2, y3));
                float mm = (m == -INFINITY) ? 0.f : m;
                m4[e] = m;
                s4[e] = __expf(y0 - mm) + __expf(y1 - mm) + __expf(y2 - mm) + __expf(y3 - mm);
            }
            *reinterpret_cast<f32x4*>(lm + col[k]) = m4;
            *reinterpret_cast<f32x4*>(ls + col[k]) = s4;
        }
        __syncthreads();
        // fold the 4 waves; thread owns 4 consecutive columns per 1024-column group
        for (int c = tid * 4; c < p.ldS; c += 1024) {
            f32x4 M4 = *reinterpret_cast<const f32x4*>(lds + c);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                f32x4 t = *reinterpret_cast<const f32x4*>(lds + (w * 2) * (KT * 256) + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) M4[e] = fmaxf(M4[e], t[e]);
            }
            f32x4 S4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                f32x4 tm = *reinterpret_cast<const f32x4*>(lds + (w * 2) * (KT * 256) + c);
                f32x4 ts = *reinterpret_cast<const f32x4*>(lds + (w * 2 + 1) * (KT * 256) + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) S4[e] += ts[e] * __expf(tm[e] - M4[e]);  // exp(-inf) = 0 for empty waves
            }
            const int64_t o = ((int64_t)b * p.chunks + chunk) * p.ldS + c;
            *reinterpret_cast<f32x4*>(p.pm + o) = M4;
            *reinterpret_cast<f32x4*>(p.ps + o) = S4;
        }
        __syncthreads();  // the fold is done with the LDS image: reuse its head for the u partials
        if (lane == 0) { lds[wave] = uwm; lds[4 + wave] = uws; }
        __syncthreads();
        if (tid == 0) {
            const float M4 = fmaxf(fmaxf(lds[0], lds[1]), fmaxf(lds[2], lds[3]));
            float S4 = 0.f;
            for (int w = 0; w < 4; ++w) S4 += lds[4 + w] * __expf(lds[w] - M4);
            p.upm[(int64_t)b * p.chunks + chunk] = M4;
            p.ups[(int64_t)b * p.chunks + chunk] = S4;
        }
    } else {
        // final: write logZ rows, row arg-max (first max wins), column partial arg-max
        int* li = reinterpret_cast<int*>(ls);
        float* const zbase = p.logZ[b / p.group_batch];
        const int bl = b % p.group_batch;
        float cm[KT][4];
        int ci[KT][4];
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) { cm[k][e] = -INFINITY; ci[k][e] = 0; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + r;
            const bool rvalid = row < p.M;  // wave-uniform
            float best = -INFINITY;
            int bj = 0x7fffffff;
            float* zr = zbase ? zbase + ((int64_t)bl * (p.M + 1) + row) * (p.N + 1) : nullptr;
#pragma unroll
            for (int k = 0; k < KT; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = col[k] + e;
                    if (FULL || j < p.N) {
                        // same association as the reference: ((couplings + u) + v) - norm
                        const float zz = ((z[r][k][e] + ur[r]) + vv[k][e]) - p.norm;
                        if (rvalid) {
                            if (zr) zr[j] = zz;
                            if (zz > best) { best = zz; bj = j; }
                            if (zz > cm[k][e]) { cm[k][e] = zz; ci[k][e] = row; }
                        }
                    }
                }
            if (rvalid) {
                if (zr && lane == 0) zr[p.N] = ((p.alpha + ur[r]) + vN) - p.norm;
                // wave arg-max, lowest index on ties (torch CPU max semantics)
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    float ob = __shfl_xor(best, o);
                    int oj = __shfl_xor(bj, o);
                    if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
                }
                if (lane == 0) {
                    p.max0[(int64_t)b * p.M + row] = best;
                    p.idx0[(int64_t)b * p.M + row] = bj;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) { lm[col[k] + e] = cm[k][e]; li[col[k] + e] = ci[k][e]; }
        __syncthreads();
        for (int c = tid; c < p.ldS; c += 256) {
            float bm = lds[c];
            int bi = reinterpret_cast<int*>(lds + (KT * 256))[c];
#pragma unroll
            for (int w = 1; w < 4; ++w) {  // waves own increasing rows: strict > keeps the first
                float m = lds[(w * 2) * (KT * 256) + c];
                int i = reinterpret_cast<int*>(lds + (w * 2 + 1) * (KT * 256))[c];
                if (m > bm) { bm = m; bi = i; }
            }
            const int64_t o = ((int64_t)b * p.chunks + chunk) * p.ldS + c;
            p.pv[o] = bm;
            p.pi[o] = bi;
        }
    }
}

// Fold the column partials into v.  grid (ceil(ldV/64), B), 64 columns x 4 chunk ranges per workgroup; the partial rows
// are read as coalesced 256-byte wave loads.  Every workgroup first recomputes the
// dustbin-ROW potential u_M of this iteration from the previous v (N+1 values, L2-resident);
// workgroup x == 0 also produces the dustbin-COLUMN potential v_N from the u partials.
// v is double-buffered (reads p.v, writes p.v_next) because workgroups of one launch overlap.
__global__ __launch_bounds__(256) void sinkhorn_combine(SkParams p) {
    __shared__ float red[8];
    const int b = blockIdx.y, tid = threadIdx.x;
    const float* vprev = p.v + (int64_t)b * p.ldV;
    float* vb = p.v_next + (int64_t)b * p.ldV;
    // u_M = log_mu_M - (alpha + LSE(v_0..v_N)),  log_mu_M = log N + norm
    float vm = -INFINITY;
    for (int j = tid; j <= p.N; j += 256) vm = fmaxf(vm, vprev[j]);
    vm = block_max(vm, red);
    float vs = 0.f;
    for (int j = tid; j <= p.N; j += 256) vs += __expf(vprev[j] - vm);
    vs = block_sum(vs, red);
    const float uM = (__logf((float)p.N) + p.norm) - (p.alpha + vm + __logf(vs));
    // v_j = log_nu - LSE_i(S_ij + u_i  U  alpha + u_M).  64 columns per workgroup, the chunk list of a column split over
    // 4 threads (4x the loads in flight, 4x the workgroups: the plain one-thread-per-column form ran 160 workgroups on
    // 256 CUs and was latency-bound at 9.7 us); the 4 partial (max, sum) pairs are merged in a fixed order.
    __shared__ float pm4[4][64], ps4[4][64];
    const int part = tid >> 6, cl = tid & 63;
    const int j = blockIdx.x * 64 + cl;
    if (j < p.N) {
        const int cps = (p.chunks + 3) >> 2, c0 = part * cps, c1 = min(p.chunks, c0 + cps);
        float Mx = part == 0 ? p.alpha + uM : -INFINITY, Sx = part == 0 ? 1.f : 0.f;
        const float* pm = p.pm + (int64_t)b * p.chunks * p.ldS + j;
        const float* ps = p.ps + (int64_t)b * p.chunks * p.ldS + j;
#pragma unroll 8
        for (int ch = c0; ch < c1; ++ch) {
            const float m = pm[(int64_t)ch * p.ldS], s = ps[(int64_t)ch * p.ldS];
            const float nm = fmaxf(Mx, m);
            Sx = Sx * __expf(Mx - nm) + s * __expf(m - nm);
            Mx = nm;
        }
        pm4[part][cl] = Mx;
        ps4[part][cl] = Sx;
    }
    __syncthreads();
    if (part == 0) {
        if (j < p.N) {
            float Mx = pm4[0][cl], Sx = ps4[0][cl];
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const float m = pm4[q][cl], s2 = ps4[q][cl];
                if (s2 > 0.f) {  // an empty part (fewer than 4 chunks) contributes nothing
                    const float nm = fmaxf(Mx, m);
                    Sx = Sx * __expf(Mx - nm) + s2 * __expf(m - nm);
                    Mx = nm;
                }
            }
            vb[j] = p.norm - (Mx + __logf(Sx));
        } else if (j > p.N && j < p.ldV) {
            vb[j] = 0.f;
        }
    }
    if (blockIdx.x == 0) {
        // v_N = log_nu_N - (alpha + LSE(u_0..u_M)),  log_nu_N = log M + norm
        float um = (tid == 0) ? uM : -INFINITY;
        for (int c = tid; c < p.chunks; c += 256) um = fmaxf(um, p.upm[(int64_t)b * p.chunks + c]);
        um = block_max(um, red);
        float us = (tid == 0) ? __expf(uM - um) : 0.f;
        for (int c = tid; c < p.chunks; c += 256)
            us += p.ups[(int64_t)b * p.chunks + c] * __expf(p.upm[(int64_t)b * p.chunks + c] - um);
        us = block_sum(us, red);
        if (tid == 0) {
            vb[p.N] = (__logf((float)p.M) + p.norm) - (p.alpha + um + __logf(us));
            p.u[(int64_t)b * (p.M + 1) + p.M] = uM;  // read by the final sweep
        }
    }
}

__global__ void sinkhorn_init(SkParams p, int B) {
    const int b = blockIdx.x;
    float* vb = p.v + (int64_t)b * p.ldV;
    for (int j = threadIdx.x; j < p.ldV; j += blockDim.x) vb[j] = 0.f;
}

// degenerate iters == 0: u = 0 (the reference returns couplings + 0 + 0 - norm)
__global__ void sinkhorn_zero_u(SkParams p) {
    const int b = blockIdx.x;
    float* ub = p.u + (int64_t)b * (p.M + 1);
    for (int i = threadIdx.x; i <= p.M; i += blockDim.x) ub[i] = 0.f;
}

struct MatchParams {
    int M, N, chunks;
    int64_t ldS;
    const float* max0;  // [B][M]
    const int* idx0;    // [B][M]
    const float* pv;    // [B][chunks][ldS]
    const int* pi;
    const int* idx1_in;  // [B][N] when the column arg-max is already final (dense path), else null
    float thr;
    int group_batch;
    int64_t* m0[kMaxGroups];
    int64_t* m1[kMaxGroups];
    float* ms0[kMaxGroups];
    float* ms1[kMaxGroups];
};

// Mutual check (match block of SuperGlue.forward).  One workgroup per pair, indices in LDS; 1024 threads: the kernel is a chain of
// dependent loads per column (the chunk partials of the column arg-max) on as few workgroups as there are pairs - 256 threads
// took 35 us for 32 pairs of 1024 keypoints.
constexpr int MF_THREADS = 1024;
__global__ __launch_bounds__(MF_THREADS) void match_finalize(MatchParams p) {
    extern __shared__ int sidx[];  // idx0 [M] | idx1 [N] | valid0 [M]
    int* i0 = sidx;
    int* i1 = sidx + p.M;
    int* v0 = sidx + p.M + p.N;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int grp = b / p.group_batch, bl = b % p.group_batch;
    int64_t* const om0 = p.m0[grp];
    int64_t* const om1 = p.m1[grp];
    float* const oms0 = p.ms0[grp];
    float* const oms1 = p.ms1[grp];
    for (int i = tid; i < p.M; i += MF_THREADS) i0[i] = p.idx0[(int64_t)b * p.M + i];
    for (int j = tid; j < p.N; j += MF_THREADS) {
        if (p.idx1_in) {
            i1[j] = p.idx1_in[(int64_t)b * p.N + j];
        } else {
            const int64_t o = (int64_t)b * p.chunks * p.ldS + j;
            float bm = p.pv[o];
            int bi = p.pi[o];
            // chunks own increasing rows: strict > keeps the first.  Both arrays are read unconditionally, 8 chunks at a
            // time, so the loads of a group are in flight together (the loop used to be one dependent L2 round trip per chunk)
            int ch = 1;
            for (; ch + 8 <= p.chunks; ch += 8) {
                float m[8];
                int ix[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    m[q] = p.pv[o + (int64_t)(ch + q) * p.ldS];
                    ix[q] = p.pi[o + (int64_t)(ch + q) * p.ldS];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (m[q] > bm) { bm = m[q]; bi = ix[q]; }
            }
            for (; ch < p.chunks; ++ch) {
                const float m = p.pv[o + (int64_t)ch * p.ldS];
                if (m > bm) { bm = m; bi = p.pi[o + (int64_t)ch * p.ldS]; }
            }
            i1[j] = bi;
        }
    }
    __syncthreads();
    for (int i = tid; i < p.M; i += MF_THREADS) {
        const int j = i0[i];
        const bool mutual = i1[j] == i;
        const float sc = mutual ? __expf(p.max0[(int64_t)b * p.M + i]) : 0.f;
        const bool valid = mutual && sc > p.thr;
        v0[i] = valid;
        if (oms0) oms0[(int64_t)bl * p.M + i] = sc;
        if (om0) om0[(int64_t)bl * p.M + i] = valid ? (int64_t)j : (int64_t)-1;
    }
    __syncthreads();
    for (int j = tid; j < p.N; j += MF_THREADS) {
        const int i = i1[j];
        const bool mutual = i0[i] == j;
        // mscores1 = where(mutual1, mscores0.gather(idx1), 0): mscores0[i] is exp(max0[i]) iff i is mutual
        const bool mut_i = i1[i0[i]] == i;
        const float sc = (mutual && mut_i) ? __expf(p.max0[(int64_t)b * p.M + i]) : 0.f;
        if (oms1) oms1[(int64_t)bl * p.N + j] = sc;
        if (om1) om1[(int64_t)bl * p.N + j] = (mutual && v0[i]) ? (int64_t)i : (int64_t)-1;
    }
}

// ---- dense-logZ arg-max (stand-alone e2emv_extract_matches) ----
__global__ __launch_bounds__(256) void dense_row_argmax(const float* Z, int M, int N, float* max0, int* idx0) {
    const int b = blockIdx.y, row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* zr = Z + ((int64_t)b * (M + 1) + row) * (N + 1);
    float best = -INFINITY;
    int bj = 0x7fffffff;
    for (int j = lane; j < N; j += 64) {
        float zz = zr[j];
        if (zz > best) { best = zz; bj = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o);
        int oj = __shfl_xor(bj, o);
        if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
    }
    if (lane == 0) { max0[(int64_t)b * M + row] = best; idx0[(int64_t)b * M + row] = bj; }
}
__global__ __launch_bounds__(256) void dense_col_argmax(const float* Z, int M, int N, int* idx1) {
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const float* zc = Z + (int64_t)b * (M + 1) * (N + 1) + j;
    float best = -INFINITY;
    int bi = 0;
    for (int i = 0; i < M; ++i) {
        float zz = zc[(int64_t)i * (N + 1)];
        if (zz > best) { best = zz; bi = i; }
    }
    idx1[(int64_t)b * N + j] = bi;
}

__global__ void pad_copy_rows(const float* src, int64_t rows, int N, float* dst, int64_t ld) {
    const int64_t r = blockIdx.x;
    for (int j = threadIdx.x; j < ld; j += blockDim.x) dst[r * ld + j] = j < N ? src[r * N + j] : 0.f;
}

// =====================================================================================================================
// Resident Sinkhorn: ALL iterations in one launch, the score matrix read from HBM ONCE per call, and NO transcendental
// per matrix element per iteration.
//
// The reference iterates in the log domain: u_i = log mu_i - LSE_j(S_ij + v_j), v_j = log nu_j - LSE_i(S_ij + u_i) - two
// exps per matrix element per iteration.  The same recurrence in the exponential domain with a per-row shift
// m_i = max(alpha, max_j S_ij):   K_ij = exp(S_ij - m_i) <= 1 (computed once),  a_i = exp(u_i + m_i),  b_j = exp(v_j),
//     a_i = mu_i / (sum_j K_ij b_j + r_i b_N),      r_i = exp(alpha - m_i)         (dustbin column)
//     b_j = nu_j / (sum_i K_ij a_i + a_M),          a_M = exp(u_M + alpha) = mu_M / (sum_j b_j + b_N)   (dustbin row)
//     b_N = nu_N / (sum_i r_i a_i + a_M)
// is one multiply-add per element per half-iteration.  u = log a - m and v = log b are handed to the final sweep, which
// evaluates logZ = ((S + u) + v) - norm from the scores exactly like the streaming path.  Every product is <= the value
// the log-domain form exponentiates after its max shift, so nothing can overflow where the reference does not; a row
// or column whose whole mass falls below fp32's range (potentials moving by > 80 nats) shows up as a zero / non-finite
// scaling, is counted in the sticky error word and poisons the outputs - E2EMV_SINKHORN=stream runs such inputs.
//
// A workgroup (8 waves) keeps 32 rows of K in registers (wave = 4 rows, lane = 4*KT columns - the sweep kernel's layout)
// for the whole call; the G = ceil(M / 32) workgroups of a problem exchange, per iteration, only column sums.  The
// exchange is a reduce-scatter + all-gather between the workgroups of ONE problem (other problems are independent and
// never wait for each other):
//   A. every workgroup publishes its N partial column sums; workgroup w adds the slice [w*cs, (w+1)*cs) over the G
//      producers in fixed order (16 lanes per column, each lane a fixed producer subset, xor-butterfly -> bit-
//      reproducible) and gets b_j for its slice;
//   B. the b slices are published and every workgroup reads all N of them back (into LDS: b is read four columns at a
//      time where it is used, the registers hold K).
// The dustbin scalings need no extra hop: a_M is a function of b (every wave sees all of b), b_N of the G partial sums of
// r_i a_i, which every workgroup adds up for itself.
// Transport = 8-byte {tag = epoch, value} granules written by one relaxed agent-scope store and polled with relaxed
// agent-scope loads (MI355X guide, Guideline 16 R2: the data is the flag; no fence, no cache-policy dependence, correct
// for any workgroup -> XCD placement).  Buffers are zeroed by a memset node before every launch, epochs count up within
// the launch, every spin is bounded (a give-up poisons the outputs with NaN and sets *timeout).  Single buffering of A
// and B is safe: a producer rewrites its stage-A granules only after it has received every stage-B slice of the
// iteration, which each consumer publishes after it has read all of stage A (and symmetrically for stage B); the
// dustbin statistics are double-buffered by epoch parity because a workgroup without a column slice publishes nothing
// the others wait for.
// Residency: the grid is at most (workgroups the occupancy query admits per CU, capped at 2) x CUs, so every workgroup of
// the launch is resident and a problem's workgroups can wait for each other; problems beyond the resident set are
// processed by the same workgroups in rounds.  16 problems of 1024 x 1024 are resident at a time (64 MB of registers).
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

__device__ __forceinline__ void granule_store(u64* p, unsigned tag, float v) {
    __hip_atomic_store((gu64*)(p), ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 granule_load(const u64* p) {
    return __hip_atomic_load((const gu64*)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// merge two (max, sum-exp) pairs; an empty pair is (-inf, 0)
__device__ __forceinline__ void lse_merge(float& M, float& S, float m, float s) {
    const float nm = fmaxf(M, m);
    const float ref = (nm == -INFINITY) ? 0.f : nm;
    S = S * __expf(M - ref) + s * __expf(m - ref);
    M = nm;
}

// Wave-wide reductions on the DPP cross-lane path (no LDS round trips): quad swaps, half-row / row mirrors, then the row
// broadcasts; the total is read from lane 63 as a scalar.  Fixed association -> bit-reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float identity, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, dpp_move<0xB1, 0xF>(v, v));              // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_move<0x4E, 0xF>(v, v));              // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_move<0x141, 0xF>(v, v));             // row_half_mirror
    v = fmaxf(v, dpp_move<0x140, 0xF>(v, v));             // row_mirror: every lane of a 16-lane row holds the row's max
    v = fmaxf(v, dpp_move<0x142, 0xA>(-INFINITY, v));     // row_bcast:15 into rows 1 and 3
    v = fmaxf(v, dpp_move<0x143, 0xC>(-INFINITY, v));     // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));  // the builtin is typed int: bit-cast, never convert
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_move<0xB1, 0xF>(v, v);
    v += dpp_move<0x4E, 0xF>(v, v);
    v += dpp_move<0x141, 0xF>(v, v);
    v += dpp_move<0x140, 0xF>(v, v);
    v += dpp_move<0x142, 0xA>(0.f, v);
    v += dpp_move<0x143, 0xC>(0.f, v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));  // the builtin is typed int: bit-cast, never convert
}

// Four wave-wide sums at once: lane l returns the sum of x[l & 3] over the wave.  The first two steps are butterflies that
// halve the number of live vectors (a lane keeps the operand of its own class and sends the other), then one vector is reduced
// over the four quads of a row (rotations by 4 and 8) and over the four rows (gfx950's row / half-wave swaps): 12 cross-lane
// operations for four sums instead of 24, and the sums arrive in four LANES - what follows (a division per row) runs once.
// (in two parts: the butterflies leave ONE register per four rows - what a pass over many rows keeps until all row sums exist)
__device__ __forceinline__ float wave_sum4_quads(float x0, float x1, float x2, float x3, int lane) {
    const bool o1 = (lane & 1) != 0, o2 = (lane & 2) != 0;
    const float u01 = (o1 ? x1 : x0) + dpp_move<0xB1, 0xF>(0.f, o1 ? x0 : x1);  // quad_perm [1,0,3,2]
    const float u23 = (o1 ? x3 : x2) + dpp_move<0xB1, 0xF>(0.f, o1 ? x2 : x3);
    return (o2 ? u23 : u01) + dpp_move<0x4E, 0xF>(0.f, o2 ? u01 : u23);         // quad_perm [2,3,0,1]: lane l = x[l & 3] over its quad
}
__device__ __forceinline__ float wave_sum4_rows(float t) {
    t += dpp_move<0x124, 0xF>(0.f, t);                                           // row_ror:4
    t += dpp_move<0x128, 0xF>(0.f, t);                                           // row_ror:8
    auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    t = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
    auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
    return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
}
__device__ __forceinline__ float wave_sum4_dpp(float x0, float x1, float x2, float x3, int lane) {
    return wave_sum4_rows(wave_sum4_quads(x0, x1, x2, x3, lane));
}

// rows per workgroup: 8 waves x 4 rows up to 1024 columns, 8 x 2 rows up to 2048 (64 matrix values per lane either way)
// rows per workgroup: 8 waves x RW rows.  RW = 4 (64 matrix values per lane at 1024 columns, two workgroups per CU) or RW = 8 at
// 1024 columns (128 values per lane, one workgroup per CU: the same number of resident problems, HALF as many workgroups in a
// problem's exchange and half the granule traffic)
static int g_skr_rw8 = -1;
static inline int skr_rw(int64_t ldS) {
    if (g_skr_rw8 < 0) g_skr_rw8 = dbg_knob("E2EMV_SKR_RW", 8) == 8 ? 1 : 0;
    return (g_skr_rw8 && ldS > 512 && ldS <= 1024) ? 8 : 4;
}
static inline int skr_rows(int64_t ldS) { return 8 * skr_rw(ldS); }
constexpr unsigned SKR_SPIN_LIMIT = 1u << 21;
constexpr unsigned SKR_GAVE_UP_NAN = 0x7fc0dead;  // potentials of a problem whose inter-workgroup wait gave up

struct SkResParams {
    const float* S;     // [B][M][ldS]
    int64_t ldS;
    int M, N, B, iters;
    float alpha, norm;
    int G;              // workgroups per problem = ceil(M / 32)
    int n_res;          // problems resident at a time (grid = n_res * G)
    int cs;             // columns per reduce-scatter slice = ceil(N / G)
    u64* bufA;          // [n_res][G consumer][G producer][cs]      partial column sums (granules)
    u64* bufB;          // [n_res][G * cs]                          b granules
    u64* bufU;          // [n_res][2][G]                            sum of r_i a_i over a workgroup's rows, by epoch parity
    unsigned* timeout;  // [1]
    unsigned long long* dbg;  // optional [16 iterations][G][8] 100 MHz timestamps of resident problem 0 (E2EMV_SKR_DEBUG)
    int flags;          // experiment switch (E2EMV_SKR_FLAGS): 1 = no s_sleep in the polls
    float* u;           // [B][M+1]  out: row potentials (u[M] = dustbin row)
    float* v;           // [B][ldV]  out: column potentials (v[N] = dustbin column)
    int64_t ldV;
};

// polls until every lane's granules carry `epoch`; returns false after a give-up (then `dead` is set for the workgroup's
// later polls).  Lane-local granule count n <= NMAX (0 for idle lanes), granule i at base[off[i]].
template <int NMAX>
__device__ __forceinline__ bool granule_wait(const u64* base, const int (&off)[NMAX], int n, unsigned epoch, unsigned (&val)[NMAX],
                                             unsigned* timeout, bool& dead, bool nap = true) {
    if (dead) {
#pragma unroll
        for (int i = 0; i < NMAX; ++i) val[i] = 0x7fc00000u;  // NaN
        return false;
    }
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NMAX; ++i)
            if (i < n) {
                const u64 g = granule_load(base + off[i]);
                val[i] = (unsigned)g;
                ok = ok && (unsigned)(g >> 32) == epoch;
            }
        if (__all(ok)) return true;
        if ((spins & 255u) == 255u) {
            const unsigned flag = __hip_atomic_load((__attribute__((address_space(1))) unsigned*)(timeout), __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
            if (flag || spins >= SKR_SPIN_LIMIT) {
                if ((threadIdx.x & 63) == 0) {
                    if (!flag) atomicAdd(timeout + 4, 1u);  // diagnostic count of give-ups (the rescue pass below re-solves the problem)
                    atomicOr(timeout, 1u);
                }
                dead = true;
#pragma unroll
                for (int i = 0; i < NMAX; ++i) val[i] = 0x7fc00000u;
                return false;
            }
        }
        if (nap) __builtin_amdgcn_s_sleep(1);
    }
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef unsigned skr_u32x4 __attribute__((ext_vector_type(4)));

// Two granules of ADJACENT columns in one 16-byte write-through store / load (aux 16 = sc1): an 8-byte sc1 store is one
// fabric write, 2.7x the time per byte of a 16-byte one (guide, price list), and the exchange is what the kernel waits for.
// Each 8-byte half is a self-validating granule {value, tag}: the two halves need not arrive together.
__device__ __forceinline__ void granule_store2(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned tag, float v0, float v1) {
    const skr_u32x4 g = {__float_as_uint(v0), tag, __float_as_uint(v1), tag};
    __builtin_amdgcn_raw_buffer_store_b128(g, r, byte_off, 0, 16);
}
// polls until both granules of every lane-local pair carry `epoch` (same give-up protocol as granule_wait)
template <int NMAX>
__device__ __forceinline__ bool granule_wait2(__amdgpu_buffer_rsrc_t r, const unsigned (&off)[NMAX], int n, unsigned epoch, unsigned (&val)[NMAX][2],
                                              unsigned* timeout, bool& dead, bool nap = true) {
    if (dead) {
#pragma unroll
        for (int i = 0; i < NMAX; ++i) { val[i][0] = 0x7fc00000u; val[i][1] = 0x7fc00000u; }
        return false;
    }
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NMAX; ++i)
            if (i < n) {
                const skr_u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(r, off[i], 0, 16);
                val[i][0] = g[0]; val[i][1] = g[2];
                ok = ok && g[1] == epoch && g[3] == epoch;
            }
        if (__all(ok)) return true;
        if ((spins & 255u) == 255u) {
            const unsigned flag = __hip_atomic_load((__attribute__((address_space(1))) unsigned*)(timeout), __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
            if (flag || spins >= SKR_SPIN_LIMIT) {
                if ((threadIdx.x & 63) == 0) {
                    if (!flag) atomicAdd(timeout + 4, 1u);
                    atomicOr(timeout, 1u);
                }
                dead = true;
#pragma unroll
                for (int i = 0; i < NMAX; ++i) { val[i][0] = 0x7fc00000u; val[i][1] = 0x7fc00000u; }
                return false;
            }
        }
        if (nap) __builtin_amdgcn_s_sleep(1);
    }
}

// exp(x) for x <= 0 with the product x*log2(e) carried in two pieces: relative error ~2e-7 also for |x| ~ 80 (the plain
// fast exp loses |x| * 1e-7).  Runs once per matrix element per call.
__device__ __forceinline__ float exp_accurate(float x) {
    const float L2E_HI = 1.44269502162933349609f, L2E_LO = 1.92596299112661746e-8f;
    const float y = x * L2E_HI;
    const float r = fmaf(x, L2E_HI, -y) + x * L2E_LO;   // what the rounded product lost, in log2 units
    return __builtin_amdgcn_exp2f(y) * fmaf(r, 0.693147180559945f, 1.0f);
}

template <int KT, bool FULL, bool PAIR = false, int RW = 4>
__global__ __launch_bounds__(512, (KT <= 4 && RW == 4) ? 4 : 2) void sinkhorn_resident(SkResParams p) {
    constexpr int W = KT * 256;                       // padded column count held by a wave
    // RW rows per wave: 64 matrix values per lane at RW = 4 (KT <= 4: two workgroups per
                                                      // CU), 128 at KT = 8 (one per CU - half as many workgroups exchange)
    constexpr int ROWS = 8 * RW;                      // rows per workgroup
    constexpr int CPT = (W + 511) / 512;              // columns a thread folds / publishes
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* fold = lds;                                // [8 waves][W] partial column sums
    float* vbuf = lds + 8 * W;                        // [W + 4]: b of the current iteration (+ b_N at [W])
    float* red = vbuf + W + 4;                        // [32] small reductions
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS bases and row numbers stay out of the VGPRs
    const int grp = blockIdx.x / p.G, w = blockIdx.x % p.G;
    const int G = p.G, cs = p.cs, N = p.N, M = p.M;
    const int row0 = w * ROWS + wave * RW;
    u64* const bufA = p.bufA + (int64_t)grp * G * G * cs;
    u64* const bufB = p.bufB + (int64_t)grp * G * cs;
    u64* const bufU2 = p.bufU + (int64_t)grp * 2 * G;  // [epoch parity][G]
    bool dead = false;
    const bool nap = !(p.flags & 1);
    int col[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) col[k] = 4 * (lane + 64 * k);
    // marginals in the linear domain (log_mu = norm, log_mu_M = log N + norm, ...; norm = -log(M + N))
    const float mu = 1.0f / (float)(M + N), muM = (float)N / (float)(M + N), nuN = (float)M / (float)(M + N);
    // stage-A destination of the columns this thread folds: consumer region wc = c / cs, producer slot w, column jl
    int dstA[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = tid + 512 * i, wc = c / cs, jl = c - wc * cs;
        dstA[i] = (wc * G + w) * cs + jl;
    }
    // pair mode: a thread owns CPT ADJACENT columns and every granule travels as half of a 16-byte pair (needs an even
    // column slice per consumer so that a pair never straddles two consumer regions)
    // (PAIR is chosen by the launcher: KT >= 4 and cs even)
    constexpr bool pair = PAIR && (CPT % 2 == 0);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(bufA, 0, G * G * cs * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(bufB, 0, G * cs * 8, 0x00020000);
    unsigned round = 0;
    for (int b = grp; b < p.B; b += p.n_res, ++round) {
        const unsigned ebase = round * (unsigned)p.iters;  // epochs run on without a gap: their parity alternates
        const float* Sb = p.S + (int64_t)b * M * p.ldS;
        // ---- load the 4 rows of this wave, shift by the row maximum, exponentiate once
        f32x2 K[RW][KT][2];
        float mrow[RW], rK[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int row = min(row0 + r, M - 1);
            float zz[KT][4];
            float mx = p.alpha;
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const f32x4 t = (FULL || col[k] < p.ldS) ? *reinterpret_cast<const f32x4*>(Sb + (int64_t)row * p.ldS + col[k])
                                                         : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    zz[k][e] = t[e];
                    if (FULL || col[k] + e < N) mx = fmaxf(mx, t[e]);
                }
            }
            mx = wave_max_dpp(mx);
            const bool rvalid = row0 + r < M;  // ragged tail: the row does not exist -> K = 0, a = 0
            mrow[r] = mx;
            rK[r] = rvalid ? exp_accurate(p.alpha - mx) : 0.f;
#pragma unroll
            for (int k = 0; k < KT; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float kv = (rvalid && (FULL || col[k] + e < N)) ? exp_accurate(zz[k][e] - mx) : 0.f;
                    K[r][k][e >> 1][e & 1] = kv;
                }
        }
        // b = exp(v) = 1, b_N = 1 (v starts at 0)
        for (int c = tid; c < W + 4; c += 512) vbuf[c] = (c < N || c == W) ? 1.f : 0.f;
        __syncthreads();
        float bN = 1.f, aM = 0.f;
        float a[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) a[r] = 0.f;

        for (int it = 0; it < p.iters; ++it) {
            const unsigned epoch = ebase + (unsigned)it + 1u;
            // the thread index, made opaque once per iteration: the exchange addresses below are then recomputed (a few
            // integer ops) instead of being hoisted out of the loop as dozens of 64-bit loop invariants that would spill
            int tq = tid;
            asm volatile("" : "+v"(tq));
            u64* const bufU = bufU2 + (epoch & 1u) * (unsigned)G;
            unsigned long long* const dbg = (p.dbg && grp == 0 && round == 0 && it < 16 && tid == 0) ? p.dbg + ((int64_t)it * G + w) * 8 : nullptr;
            if (dbg) dbg[0] = __builtin_amdgcn_s_memrealtime();
            // ---- row half-iteration: a_i = mu / (sum_j K_ij b_j + r_i b_N) for the wave's 4 rows; a_M from sum_j b_j
            {
                f32x2 acc[RW], accb = {0.f, 0.f};
#pragma unroll
                for (int r = 0; r < RW; ++r) acc[r] = f32x2{0.f, 0.f};
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(vbuf + col[k]);  // 0 beyond N
                    const f32x2 blo = {b4[0], b4[1]}, bhi = {b4[2], b4[3]};
                    accb += blo + bhi;
#pragma unroll
                    for (int r = 0; r < RW; ++r) {
                        acc[r] = __builtin_elementwise_fma(K[r][k][0], blo, acc[r]);
                        acc[r] = __builtin_elementwise_fma(K[r][k][1], bhi, acc[r]);
                    }
                }
                // four row sums per reduction, arriving in lanes (l & 3): ONE division gives the a_i of four rows, read back as scalars
                static_assert(RW % 4 == 0, "rows of a wave in groups of four");
#pragma unroll
                for (int g = 0; g < RW / 4; ++g) {
                    const float s4 = wave_sum4_dpp(acc[4 * g][0] + acc[4 * g][1], acc[4 * g + 1][0] + acc[4 * g + 1][1],
                                                   acc[4 * g + 2][0] + acc[4 * g + 2][1], acc[4 * g + 3][0] + acc[4 * g + 3][1], lane);
                    const int q = lane & 3;
                    const float rk = q == 0 ? rK[4 * g] : (q == 1 ? rK[4 * g + 1] : (q == 2 ? rK[4 * g + 2] : rK[4 * g + 3]));
                    const float a4 = (row0 + 4 * g + q < M) ? mu / fmaf(rk, bN, s4) : 0.f;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) a[4 * g + qq] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a4), qq));
                }
                aM = muM / (wave_sum_dpp(accb[0] + accb[1]) + bN);
            }
            // ---- column half-iteration, this wave's part: sum over its 4 rows of K_ij a_i -> LDS
            {
                float* lf = fold + wave * W;
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    f32x2 lo = K[0][k][0] * f32x2{a[0], a[0]}, hi = K[0][k][1] * f32x2{a[0], a[0]};
#pragma unroll
                    for (int r = 1; r < RW; ++r) {
                        lo = __builtin_elementwise_fma(K[r][k][0], f32x2{a[r], a[r]}, lo);
                        hi = __builtin_elementwise_fma(K[r][k][1], f32x2{a[r], a[r]}, hi);
                    }
                    *reinterpret_cast<f32x4*>(lf + col[k]) = f32x4{lo[0], lo[1], hi[0], hi[1]};
                }
                float ra = rK[0] * a[0];
#pragma unroll
                for (int r = 1; r < RW; ++r) ra = fmaf(rK[r], a[r], ra);
                if (lane == 0) red[wave] = ra;  // dustbin column
            }
            if (dbg) dbg[1] = __builtin_amdgcn_s_memrealtime();
            __syncthreads();
            // ---- fold the 8 waves, publish the workgroup's partial column sums (stage A) and its dustbin-column sum
            if constexpr (pair) {
                float fT[CPT];
#pragma unroll
                for (int i = 0; i < CPT; ++i) {
                    const int c = CPT * tq + i;
                    float T = fold[c];
#pragma unroll
                    for (int wv = 1; wv < 8; ++wv) T += fold[wv * W + c];
                    fT[i] = T;  // (columns >= N hold zeros: K is zero there)
                }
#pragma unroll
                for (int i = 0; i < CPT; i += 2) {
                    const int c = CPT * tq + i, wc = c / cs, jl = c - wc * cs;
                    if (FULL || c < N) granule_store2(rsA, (unsigned)((wc * G + w) * cs + jl) * 8u, epoch, fT[i], fT[i + 1]);
                }
                if (tq == 0) {
                    float U = red[0];
                    for (int wv = 1; wv < 8; ++wv) U += red[wv];
                    granule_store(bufU + w, epoch, U);
                }
            } else {
                float fT[CPT];
#pragma unroll
                for (int i = 0; i < CPT; ++i) {
                    const int c = tq + 512 * i;
                    fT[i] = 0.f;
                    if (c < W && (FULL || c < N)) {
                        float T = fold[c];
#pragma unroll
                        for (int wv = 1; wv < 8; ++wv) T += fold[wv * W + c];
                        fT[i] = T;
                    }
                }
#pragma unroll
                for (int i = 0; i < CPT; ++i) {  // all stores after all LDS work: nothing waits behind a write-through store
                    const int c = tq + 512 * i;
                    if (c < W && (FULL || c < N)) granule_store(bufA + dstA[i], epoch, fT[i]);
                }
                if (tq == 0) {
                    float U = red[0];
                    for (int wv = 1; wv < 8; ++wv) U += red[wv];
                    granule_store(bufU + w, epoch, U);
                }
            }
            if (dbg) dbg[2] = __builtin_amdgcn_s_memrealtime();
            // ---- stage A consume: my slice of columns over all producers -> b_j = nu / (sum + a_M), published as stage B
            if constexpr (pair) {
                const int q = tq & 15, cg = tq >> 4;
                const unsigned base_b = (unsigned)(w * G * cs) * 8u;  // my consumer region: [producer][cs]
                for (int j0 = 0; j0 < cs; j0 += 64) {
                    const int jl = j0 + 2 * cg, c = w * cs + jl;
                    const bool act = jl < cs && c < N;
                    float T0 = 0.f, T1 = 0.f;
                    for (int g0 = 0; g0 < G; g0 += 32) {  // wave-uniform trip count; two 16-byte loads in flight per lane
                        unsigned off[2];
                        unsigned val[2][2];
                        int n = 0;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int g = g0 + q + 16 * i;
                            off[i] = base_b;
                            if (act && g < G) { off[i] = base_b + (unsigned)(g * cs + jl) * 8u; n = i + 1; }
                        }
                        granule_wait2<2>(rsA, off, n, epoch, val, p.timeout, dead, nap);
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            if (i < n) { T0 += __uint_as_float(val[i][0]); T1 += __uint_as_float(val[i][1]); }
                    }
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) { T0 += __shfl_xor(T0, o); T1 += __shfl_xor(T1, o); }
                    if (act && q == 0) granule_store2(rsB, (unsigned)c * 8u, epoch, mu / (T0 + aM), mu / (T1 + aM));  // nu_j = mu
                }
            } else {
                const int q = tq & 15, cg = tq >> 4;
                const u64* base = bufA + (int64_t)w * G * cs;  // my consumer region: [producer][cs]
                for (int j0 = 0; j0 < cs; j0 += 32) {
                    const int jl = j0 + cg, c = w * cs + jl;
                    const bool act = jl < cs && c < N;
                    float T = 0.f;
                    for (int g0 = 0; g0 < G; g0 += 64) {  // wave-uniform trip count
                        int off[4];
                        unsigned val[4];
                        int n = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int g = g0 + q + 16 * i;
                            off[i] = 0;
                            if (act && g < G) { off[i] = g * cs + jl; n = i + 1; }
                        }
                        granule_wait<4>(base, off, n, epoch, val, p.timeout, dead, nap);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (i < n) T += __uint_as_float(val[i]);
                    }
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) T += __shfl_xor(T, o);
                    if (act && q == 0) granule_store(bufB + c, epoch, mu / (T + aM));  // nu_j = mu
                }
            }
            if (dbg) dbg[3] = __builtin_amdgcn_s_memrealtime();
            // ---- b_N = nu_N / (sum_i r_i a_i + a_M) from the G workgroup sums (wave 0)
            if (wave == 0) {
                float U = 0.f;
                for (int g0 = 0; g0 < G; g0 += 64) {
                    const int g = g0 + lane;
                    int off[1] = {g < G ? g : 0};
                    unsigned val[1];
                    granule_wait<1>(bufU, off, g < G ? 1 : 0, epoch, val, p.timeout, dead, nap);
                    if (g < G) U += __uint_as_float(val[0]);
                }
                U = wave_sum_dpp(U);
                if (lane == 0) vbuf[W] = nuN / (U + aM);
            }
            if (dbg) dbg[4] = __builtin_amdgcn_s_memrealtime();
            // ---- stage B consume: all of b into LDS
            if constexpr (pair) {
                for (int c0 = 0; c0 < W; c0 += 1024) {  // wave-uniform trip count
                    const int ca = c0 + 2 * tq;
                    unsigned off[1] = {ca < N ? (unsigned)ca * 8u : 0u};
                    unsigned val[1][2];
                    granule_wait2<1>(rsB, off, ca < N ? 1 : 0, epoch, val, p.timeout, dead, nap);
                    if (ca < W) *reinterpret_cast<f32x2*>(vbuf + ca) = f32x2{ca < N ? __uint_as_float(val[0][0]) : 0.f,
                                                                            ca + 1 < N ? __uint_as_float(val[0][1]) : 0.f};
                }
            } else
            for (int c0 = 0; c0 < W; c0 += 1024) {  // wave-uniform trip count
                const int ca = c0 + tq, cb = c0 + 512 + tq;
                int off[2] = {ca < N ? ca : 0, cb < N ? cb : 0};
                unsigned val[2];
                granule_wait<2>(bufB, off, cb < N ? 2 : (ca < N ? 1 : 0), epoch, val, p.timeout, dead, nap);
                if (ca < W) vbuf[ca] = ca < N ? __uint_as_float(val[0]) : 0.f;
                if (cb < W) vbuf[cb] = cb < N ? __uint_as_float(val[1]) : 0.f;
            }
            if (dbg) dbg[5] = __builtin_amdgcn_s_memrealtime();
            if (__syncthreads_or(dead ? 1 : 0)) dead = true;
            if (dbg) dbg[6] = __builtin_amdgcn_s_memrealtime();
            bN = vbuf[W];
        }

        // ---- hand the potentials to the final sweep (logZ, fused arg-max): u = log a - m of this workgroup's rows, and
        // from workgroup 0 the dustbin-row potential and v = log b.  A scaling that left fp32's range (zero, infinite,
        // NaN) or a give-up in the exchange is counted in the sticky error word; its NaN / inf reaches the outputs.
        {
            // (a give-up marks ITS problem with a NaN of its own payload: the rescue pass books the problem as a timeout - contention,
            // says nothing about the model - only when it finds that mark; a scaling that left fp32's range yields inf / the default NaN)
            const float qnan = __uint_as_float(SKR_GAVE_UP_NAN);
            float* ub = p.u + (int64_t)b * (M + 1);
            bool bad = false;
#pragma unroll
            for (int r = 0; r < RW; ++r)
                if (row0 + r < M) {
                    bad = bad || !(a[r] > 0.f) || !(a[r] < INFINITY);
                    if (lane == 0) ub[row0 + r] = dead ? qnan : __logf(a[r]) - mrow[r];
                }
            if (w == 0) {
                float* vb = p.v + (int64_t)b * p.ldV;
                for (int j = tid; j < p.ldV; j += 512) {
                    const float bj = j < N ? vbuf[j] : (j == N ? bN : 1.f);
                    bad = bad || !(bj > 0.f) || !(bj < INFINITY);
                    vb[j] = dead ? qnan : __logf(bj);
                }
                bad = bad || !(aM > 0.f) || !(aM < INFINITY);
                if (tid == 0) ub[M] = dead ? qnan : __logf(aM) - p.alpha;
            }
            if (__syncthreads_or(bad ? 1 : 0) && tid == 0) atomicAdd(p.timeout + 4, 1u);  // also: LDS is reused by the next problem
        }
    }
}


// ---- 128 rows per workgroup: ALL problems of a 32-pair batch resident at once (round 5) -----------------------------------------
// sinkhorn_resident keeps 64 rows x 1024 columns per workgroup (one per CU) in registers: 16 problems of 1024 x 1024 fill the chip,
// a batch of 32 runs as two rounds of 100 iterations, and an iteration is bound by the exchange, not by the arithmetic.  Here a
// workgroup owns 128 rows - 8 workgroups per problem, 32 problems resident, ONE round - which needs 512 KB of couplings per CU,
// the size of the register file.  What makes it fit:
//   * FOUR waves per workgroup, one per SIMD: a wave then has 512 registers per lane (256 VGPRs + 256 AGPRs); 24 of its 32 rows live
//     there (384 values per lane - hipcc parks what does not fit the VGPRs in the accumulator file, one v_accvgpr_read per use),
//     8 rows in LDS (128 KB per workgroup);
//   * ONE pass over the couplings per iteration: a_i depends on row i's sum alone (rows are whole inside a wave), so a row's
//     column contribution K_ij a_i is accumulated right behind its row sum - no a[] array, no second sweep over K (the row
//     kernel's two half-iterations read K twice: twice the accumulator-file reads and LDS traffic here);
//   * the fold buffer holds the 4 waves' partial column sums (16 KB).
// Exchange, epochs, give-up protocol, rescue: the row kernel's (pair mode), with the lane mappings of 256 threads.  An iteration
// costs about twice the arithmetic per CU and the same two hops, for half as many rounds: chosen by the launcher when it saves
// rounds (more than 16 problems of 513 ... 1024 columns).
template <int... I, class F>
__device__ __forceinline__ void sk_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sk_static_for(F&& f) { sk_static_for_impl(std::make_integer_sequence<int, N>{}, f); }
// rows of a wave: 12 in vector registers v64 .. v255, 12 in accumulation registers a64 .. a255, 8 in LDS
constexpr int SK128_RV = 12, SK128_RA = 12, SK128_RL = 8, SK128_RR = SK128_RV + SK128_RA, SK128_R0 = 64;
constexpr int sk128_base(int r) { return SK128_R0 + 16 * (r < SK128_RV ? r : r - SK128_RV); }
#include "sinkhorn128_rows.h"
// per-phase timestamps (tools/skr_timing.py): only in the measurement build - in these two kernels a 64-bit pointer kept over the
// iteration costs registers the compiler's window does not have (it went to scratch and was reloaded at every stamp)
#ifdef E2EMV_STAMPS
#define SK_STAMP_PTR() unsigned long long* const dbg = (p.dbg && grp == 0 && round == 0 && it < 16 && tid == 0) ? p.dbg + ((int64_t)it * G + w) * 8 : nullptr
#define SK_STAMP(i) do { if (dbg) dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SK_STAMP_PTR() do {} while (0)
#define SK_STAMP(i) do {} while (0)
#endif
template <bool FULL>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(56))) void sinkhorn_resident128(SkResParams p) {
    constexpr int KT = 4, W = 1024, RW = SK128_RR + SK128_RL, ROWS = 4 * RW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* klds = lds;                       // [4 waves][RL rows][W]: the LDS-resident rows of K
    float* fold = klds + 4 * SK128_RL * W;   // [4][W] partial column sums
    float* vbuf = fold + 4 * W;              // [W + 4]: b of the current iteration (+ b_N at [W])
    float* red = vbuf + W + 4;               // [32]
    float* rks = red + 32;                   // [ROWS] r_i = exp(alpha - rowmax_i)
    float* mrs = rks + ROWS;                 // [ROWS] rowmax_i
    float* asv = mrs + ROWS;                 // [ROWS] a_i of the last iteration (for the potentials)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = blockIdx.x / p.G, w = blockIdx.x % p.G;
    const int G = p.G, cs = p.cs, N = p.N, M = p.M;
    const int row0 = w * ROWS + wave * RW;
    u64* const bufA = p.bufA + (int64_t)grp * G * G * cs;
    u64* const bufB = p.bufB + (int64_t)grp * G * cs;
    u64* const bufU2 = p.bufU + (int64_t)grp * 2 * G;
    bool dead = false;
    const bool nap = !(p.flags & 1);
    const float mu = 1.0f / (float)(M + N), muM = (float)N / (float)(M + N), nuN = (float)M / (float)(M + N);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(bufA, 0, G * G * cs * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(bufB, 0, G * cs * 8, 0x00020000);
    float* const kl = klds + wave * SK128_RL * W + 4 * lane;  // this lane's first chunk of the wave's LDS rows (chunk k: + 256 k)
    unsigned round = 0;
    for (int b = grp; b < p.B; b += p.n_res, ++round) {
        const unsigned ebase = round * (unsigned)p.iters;
        const float* Sb = p.S + (int64_t)b * M * p.ldS;
        // ---- load the wave's 32 rows, shift by the row maximum, exponentiate once; rows 24 - 31 go to LDS
        // 24 of the wave's 32 rows live in registers the compiler does not allocate: amdgpu_num_vgpr(56) confines it to v0 - v55
        // (and a0 - a55 as its spill space); v56 - v63 are the row pass's temporaries, v64 - v255 hold rows 0 - 11 and a64 - a255
        // rows 12 - 23, as [16 r + 4 k + e].  hipcc's allocator cannot keep 384 values in place for a whole call (it spills
        // exactly the long-lived ones: profiles/r5_sinkhorn_blocks.log), so these registers are written (v_mov / v_accvgpr_write,
        // once per problem) and read (the row pass in sinkhorn128_rows.h, twice per iteration) by number.  The clobber sizes the
        // wave's allocation at 256 + 256 registers; tests/test_host_and_abi.py disassembles the kernel and checks that nothing
        // outside these statements touches a register above v55 / a55.
        asm volatile("" ::: "v255", "a255");
        sk_static_for<RW>([&](auto r_c) {
            constexpr int r = decltype(r_c)::value;
            const int row = FULL ? row0 + r : min(row0 + r, M - 1);
            f32x4 zz[KT];
            float mx = p.alpha;
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const int c = 4 * (lane + 64 * k);
                zz[k] = (FULL || c < p.ldS) ? *reinterpret_cast<const f32x4*>(Sb + (int64_t)row * p.ldS + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (FULL || c + e < N) mx = fmaxf(mx, zz[k][e]);
            }
            mx = wave_max_dpp(mx);
            const bool rvalid = FULL || row0 + r < M;
            if (lane == 0) {
                mrs[wave * RW + r] = mx;
                rks[wave * RW + r] = rvalid ? exp_accurate(p.alpha - mx) : 0.f;
            }
            sk_static_for<KT>([&](auto k_c) {
                constexpr int k = decltype(k_c)::value;
                const int c = 4 * (lane + 64 * k);
                f32x4 kv;
#pragma unroll
                for (int e = 0; e < 4; ++e) kv[e] = (rvalid && (FULL || c + e < N)) ? exp_accurate(zz[k][e] - mx) : 0.f;
                if constexpr (r < SK128_RV) {
                    const float k0 = kv[0], k1 = kv[1], k2 = kv[2], k3 = kv[3];
                    asm volatile("v_mov_b32 v[%4], %0\n\tv_mov_b32 v[%4+1], %1\n\tv_mov_b32 v[%4+2], %2\n\tv_mov_b32 v[%4+3], %3"
                                 :: "v"(k0), "v"(k1), "v"(k2), "v"(k3), "n"(sk128_base(r) + 4 * k));
                } else if constexpr (r < SK128_RR) {
                    const float k0 = kv[0], k1 = kv[1], k2 = kv[2], k3 = kv[3];
                    asm volatile("v_accvgpr_write_b32 a[%4], %0\n\tv_accvgpr_write_b32 a[%4+1], %1\n\tv_accvgpr_write_b32 a[%4+2], %2\n\tv_accvgpr_write_b32 a[%4+3], %3"
                                 :: "v"(k0), "v"(k1), "v"(k2), "v"(k3), "n"(sk128_base(r) + 4 * k));
                } else {
                    *reinterpret_cast<f32x4*>(kl + (r - SK128_RR) * W + 256 * k) = kv;
                }
            });
            if (r & 1) __builtin_amdgcn_sched_barrier(0);  // two rows in flight
        });
        for (int c = tid; c < W + 4; c += 256) vbuf[c] = (c < N || c == W) ? 1.f : 0.f;
        __syncthreads();
        float bN = 1.f, aM = 0.f;

        for (int it = 0; it < p.iters; ++it) {
            const unsigned epoch = ebase + (unsigned)it + 1u;
            int tq = tid;
            asm volatile("" : "+v"(tq));
            u64* const bufU = bufU2 + (epoch & 1u) * (unsigned)G;
            const bool last = it + 1 == p.iters;
            SK_STAMP_PTR();
            SK_STAMP(0);
            // ---- the wave's 32 rows in three phases, so that no latency-bound chain stands between two streams of multiply-adds:
            //   (1) row sums of all rows (asm), four rows folded into one register by two butterflies;
            //   (2) the 8 reductions over quads and rows and the 8 divisions - independent chains, interleaved by the compiler;
            //   (3) column sums of all rows (asm) with the a_i as scalars.  LDS rows are read in both (1) and (3).
            f32x2 cl[KT], ch[KT];  // partial column sums of this lane's 16 columns (pairs 0 - 1 | 2 - 3 of each chunk)
            float ra = 0.f;         // sum of r_i a_i over the wave's rows (dustbin column)
            {
                const unsigned kl_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)kl;
                float t4[RW / 4];   // group g: lane l holds the sum of row 4 g + (l & 3) over the lane's quad
                {
                    f32x2 blo[KT], bhi[KT];
                    f32x2 accb = {0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(vbuf + 4 * (lane + 64 * k));  // 0 beyond N
                        blo[k] = f32x2{b4[0], b4[1]};
                        bhi[k] = f32x2{b4[2], b4[3]};
                        accb += blo[k] + bhi[k];
                    }
                    aM = muM / (wave_sum_dpp(accb[0] + accb[1]) + bN);
                    // every asm statement of the register rows also fetches ONE LDS row, consumed right behind it (the LDS latency
                    // hides under the statement's multiply-adds): rows 0 - 2 with the vector-register groups, 3 - 7 with the first five
                    // pair statements of the accumulation-register groups
                    float x[SK128_RL];
                    auto lds_row_sum = [&](const f32x4 (&t)[4], int j) __attribute__((always_inline)) {
                        f32x2 acc = {0.f, 0.f};
#pragma unroll
                        for (int k = 0; k < KT; ++k) {
                            acc = __builtin_elementwise_fma(f32x2{t[k][0], t[k][1]}, blo[k], acc);
                            acc = __builtin_elementwise_fma(f32x2{t[k][2], t[k][3]}, bhi[k], acc);
                        }
                        x[j] = acc[0] + acc[1];
                        asm volatile("" : "+v"(x[j]));
                    };
                    sk_static_for<SK128_RR / 4>([&](auto g_c) {
                        constexpr int g = decltype(g_c)::value, r0 = 4 * g;
                        constexpr int B0 = sk128_base(r0), B1 = sk128_base(r0 + 1), B2 = sk128_base(r0 + 2), B3 = sk128_base(r0 + 3);
                        f32x2 acc[4];
                        f32x4 t[4];
                        if constexpr (r0 < SK128_RV) {
                            sk128_rs4v_l<B0, B1, B2, B3, g * W * 4>(acc, blo, bhi, t, kl_a);
                            lds_row_sum(t, g);
                        } else {
                            constexpr int j0 = SK128_RV / 4 + 2 * (g - SK128_RV / 4);  // LDS rows of this group's two pair statements
                            sk128_rs2a_l<B0, B1, j0 * W * 4>(acc[0], acc[1], blo, bhi, t, kl_a);
                            lds_row_sum(t, j0);
                            if constexpr (j0 + 1 < SK128_RL) {
                                sk128_rs2a_l<B2, B3, (j0 + 1) * W * 4>(acc[2], acc[3], blo, bhi, t, kl_a);
                                lds_row_sum(t, j0 + 1);
                            } else {
                                sk128_rs2a<B2, B3>(acc[2], acc[3], blo, bhi);
                            }
                        }
                        t4[g] = wave_sum4_quads(acc[0][0] + acc[0][1], acc[1][0] + acc[1][1], acc[2][0] + acc[2][1], acc[3][0] + acc[3][1], lane);
                        asm volatile("" : "+v"(t4[g]));
                    });
                    static_assert(SK128_RV / 4 + 2 * (SK128_RA / 4) - 1 >= SK128_RL, "an asm statement per LDS row");
                    t4[SK128_RR / 4] = wave_sum4_quads(x[0], x[1], x[2], x[3], lane);
                    t4[SK128_RR / 4 + 1] = wave_sum4_quads(x[4], x[5], x[6], x[7], lane);
                }
                // (2) a_i of four rows per division; the scalars for phase 3; the dustbin statistic sum_i r_i a_i per lane class
                float as[RW];
                float ra4 = 0.f;
                int l3 = lane & 3;
                asm volatile("" : "+v"(l3));  // (per iteration: the 8 LDS addresses below are otherwise hoisted out of the loop and spilled)
#pragma unroll
                for (int g = 0; g < RW / 4; ++g) {
                    const float s_r = wave_sum4_rows(t4[g]);
                    const int rl = wave * RW + 4 * g + l3;
                    const float rk = rks[rl];
                    const float ar = (FULL || row0 + 4 * g + l3 < M) ? mu / fmaf(rk, bN, s_r) : 0.f;
                    ra4 = fmaf(rk, ar, ra4);
                    if (last) asv[rl] = ar;  // for the potentials (16 lanes write the same value to the same word)
#pragma unroll
                    for (int q = 0; q < 4; ++q) as[4 * g + q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ar), q));
                }
                ra = (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ra4), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ra4), 1)))
                     + (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ra4), 2)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ra4), 3)));
                // (3)
#pragma unroll
                for (int k = 0; k < KT; ++k) { cl[k] = f32x2{0.f, 0.f}; ch[k] = f32x2{0.f, 0.f}; }
                sk_static_for<SK128_RR / 4>([&](auto g_c) {
                    constexpr int r0 = 4 * decltype(g_c)::value;
                    constexpr int B0 = sk128_base(r0), B1 = sk128_base(r0 + 1), B2 = sk128_base(r0 + 2), B3 = sk128_base(r0 + 3);
                    const f32x2 a2[4] = {f32x2{as[r0], as[r0]}, f32x2{as[r0 + 1], as[r0 + 1]}, f32x2{as[r0 + 2], as[r0 + 2]}, f32x2{as[r0 + 3], as[r0 + 3]}};
                    if constexpr (r0 < SK128_RV) sk128_rc4v<B0, B1, B2, B3>(cl, ch, a2);
                    else sk128_rc4a<B0, B1, B2, B3>(cl, ch, a2);
                });
                sk_static_for<SK128_RL / 2>([&](auto g_c) {  // LDS rows again, two per statement (32 registers: b is dead by now)
                    constexpr int r = 2 * decltype(g_c)::value;
                    f32x4 t0, t1, t2, t3, t4_, t5, t6, t7;
                    asm volatile("ds_read_b128 %0, %8 offset:%9\n\tds_read_b128 %1, %8 offset:%9+1024\n\t"
                                 "ds_read_b128 %2, %8 offset:%9+2048\n\tds_read_b128 %3, %8 offset:%9+3072\n\t"
                                 "ds_read_b128 %4, %8 offset:%9+4096\n\tds_read_b128 %5, %8 offset:%9+4096+1024\n\t"
                                 "ds_read_b128 %6, %8 offset:%9+4096+2048\n\tds_read_b128 %7, %8 offset:%9+4096+3072\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4_), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(kl_a), "n"(r * W * 4));
                    const f32x4 t[2][KT] = {{t0, t1, t2, t3}, {t4_, t5, t6, t7}};
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x2 a2 = {as[SK128_RR + r + j], as[SK128_RR + r + j]};
#pragma unroll
                        for (int k = 0; k < KT; ++k) {
                            cl[k] = __builtin_elementwise_fma(f32x2{t[j][k][0], t[j][k][1]}, a2, cl[k]);
                            ch[k] = __builtin_elementwise_fma(f32x2{t[j][k][2], t[j][k][3]}, a2, ch[k]);
                        }
                    }
                    // (here, not sunk to the end of the pass with the rows kept in scratch until then)
                    asm volatile("" : "+v"(cl[0]), "+v"(cl[1]), "+v"(cl[2]), "+v"(cl[3]), "+v"(ch[0]), "+v"(ch[1]), "+v"(ch[2]), "+v"(ch[3]));
                });
            }
            // ---- the 4 waves' partial column sums -> LDS
            {
                float* lf = fold + wave * W + 4 * lane;
#pragma unroll
                for (int k = 0; k < KT; ++k) *reinterpret_cast<f32x4*>(lf + 256 * k) = f32x4{cl[k][0], cl[k][1], ch[k][0], ch[k][1]};
                if (lane == 0) red[wave] = ra;
                SK_STAMP(1);
                __syncthreads();
            }
            // ---- publish the workgroup's partial column sums (stage A, 16-byte pairs: 4 adjacent columns per thread) and its dustbin sum
            {
                const int c = 4 * tq;
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(fold + c), t1 = *reinterpret_cast<const f32x4*>(fold + W + c);
                const f32x4 t2 = *reinterpret_cast<const f32x4*>(fold + 2 * W + c), t3 = *reinterpret_cast<const f32x4*>(fold + 3 * W + c);
                const f32x4 T = (t0 + t1) + (t2 + t3);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int cc = c + 2 * h, wc = cc / cs, jl = cc - wc * cs;
                    if (FULL || cc < N) granule_store2(rsA, (unsigned)((wc * G + w) * cs + jl) * 8u, epoch, T[2 * h], T[2 * h + 1]);
                }
                if (tq == 0) {
                    const float U = (red[0] + red[1]) + (red[2] + red[3]);
                    granule_store(bufU + w, epoch, U);
                }
            }
            SK_STAMP(2);
            // ---- stage A consume: my slice of columns over all producers -> b_j = mu / (sum + a_M), published as stage B
            {
                const int q = tq & 3, cg = tq >> 2;  // 4 lanes per column pair, each two producers: q, q + 4 (, + 8, + 12)
                const unsigned base_b = (unsigned)(w * G * cs) * 8u;
                for (int j0 = 0; j0 < cs; j0 += 128) {
                    const int jl = j0 + 2 * cg, c = w * cs + jl;
                    const bool act = jl < cs && c < N;
                    float T0 = 0.f, T1 = 0.f;
                    for (int g0 = 0; g0 < G; g0 += 8) {
                        unsigned off[2];
                        unsigned val[2][2];
                        int n = 0;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int g = g0 + q + 4 * i;
                            off[i] = base_b;
                            if (act && g < G) { off[i] = base_b + (unsigned)(g * cs + jl) * 8u; n = i + 1; }
                        }
                        granule_wait2<2>(rsA, off, n, epoch, val, p.timeout, dead, nap);
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            if (i < n) { T0 += __uint_as_float(val[i][0]); T1 += __uint_as_float(val[i][1]); }
                    }
#pragma unroll
                    for (int o = 2; o > 0; o >>= 1) { T0 += __shfl_xor(T0, o); T1 += __shfl_xor(T1, o); }
                    if (act && q == 0) granule_store2(rsB, (unsigned)c * 8u, epoch, mu / (T0 + aM), mu / (T1 + aM));
                }
            }
            SK_STAMP(3);
            // ---- b_N = nu_N / (sum_i r_i a_i + a_M) from the G workgroup sums (wave 0)
            if (wave == 0) {
                float U = 0.f;
                for (int g0 = 0; g0 < G; g0 += 64) {
                    const int g = g0 + lane;
                    int off[1] = {g < G ? g : 0};
                    unsigned val[1];
                    granule_wait<1>(bufU, off, g < G ? 1 : 0, epoch, val, p.timeout, dead, nap);
                    if (g < G) U += __uint_as_float(val[0]);
                }
                U = wave_sum_dpp(U);
                if (lane == 0) vbuf[W] = nuN / (U + aM);
            }
            SK_STAMP(4);
            // ---- stage B consume: all of b into LDS (2 pairs per thread)
            {
                unsigned off[2];
                unsigned val[2][2];
                int n = 0;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ca = 2 * tq + 512 * i;
                    off[i] = 0u;
                    if (ca < N) { off[i] = (unsigned)ca * 8u; n = i + 1; }
                }
                granule_wait2<2>(rsB, off, n, epoch, val, p.timeout, dead, nap);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ca = 2 * tq + 512 * i;
                    *reinterpret_cast<f32x2*>(vbuf + ca) = f32x2{ca < N ? __uint_as_float(val[i][0]) : 0.f, ca + 1 < N ? __uint_as_float(val[i][1]) : 0.f};
                }
            }
            SK_STAMP(5);
            if (__syncthreads_or(dead ? 1 : 0)) dead = true;
            SK_STAMP(6);
            bN = vbuf[W];
        }

        // ---- potentials for the final sweep (as in sinkhorn_resident)
        {
            const float qnan = __uint_as_float(SKR_GAVE_UP_NAN);
            float* ub = p.u + (int64_t)b * (M + 1);
            bool bad = false;
            if (tid < ROWS && w * ROWS + tid < M) {
                const float ar = asv[tid];
                bad = bad || !(ar > 0.f) || !(ar < INFINITY);
                ub[w * ROWS + tid] = dead ? qnan : __logf(ar) - mrs[tid];
            }
            if (w == 0) {
                float* vb = p.v + (int64_t)b * p.ldV;
                for (int j = tid; j < p.ldV; j += 256) {
                    const float bj = j < N ? vbuf[j] : (j == N ? bN : 1.f);
                    bad = bad || !(bj > 0.f) || !(bj < INFINITY);
                    vb[j] = dead ? qnan : __logf(bj);
                }
                bad = bad || !(aM > 0.f) || !(aM < INFINITY);
                if (tid == 0) ub[M] = dead ? qnan : __logf(aM) - p.alpha;
            }
            if (__syncthreads_or(bad ? 1 : 0) && tid == 0) atomicAdd(p.timeout + 4, 1u);
        }
    }
}


// ---- the same construction for 1025 .. 2048 columns: 64 rows per workgroup ---------------------------------------------------
// A row of K is 32 registers per lane here ([32 r + 16 h + 4 k + e]: h = the column half, chunk 4 h + k covers columns
// 4 (lane + 64 (4 h + k)) .. + 3).  A wave holds 16 rows: 6 in v64 .. v255, 6 in a64 .. a255, 4 in LDS (128 KB for the
// workgroup), so a problem of 2048 rows is 32 workgroups (64 with the 32-row workgroups of sinkhorn_resident<8>) and eight
// problems are resident instead of four.  The compiler's 56 registers cannot hold b (32) and the column partials (32) at once:
//   * row sums: per column half - b of the half (16 registers), the rows in pairs (sk_rs2v / sk128_rs2a, LDS rows through 16
//     registers), each row's partial sum added into ONE register per row;
//   * four 4-way reductions, a_i of four rows per division (as in sinkhorn_resident128);
//   * column sums: per half 16 registers of partials, both halves kept (b is dead by then) until the fold;
//   * the fold of the 4 waves goes through 16 KB (LDS is full): waves 2 and 3 write, waves 0 and 1 add theirs and write
//     back, then all threads publish fold[0] + fold[1].
constexpr int SK2K_RV = 6, SK2K_RA = 6, SK2K_RL = 4, SK2K_RR = SK2K_RV + SK2K_RA, SK2K_RW = SK2K_RR + SK2K_RL;
constexpr int sk2k_base(int r, int h) { return 64 + 32 * (r < SK2K_RV ? r : r - SK2K_RV) + 16 * h; }
template <bool FULL>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(56))) void sinkhorn_resident2k(SkResParams p) {
    constexpr int W = 2048, RW = SK2K_RW, ROWS = 4 * RW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* klds = lds;                       // [4 waves][RL rows][W]: the LDS-resident rows of K
    float* fold = klds + 4 * SK2K_RL * W;    // [2][W] partial column sums
    float* vbuf = fold + 2 * W;              // [W + 4]: b of the current iteration (+ b_N at [W])
    float* red = vbuf + W + 4;               // [32]
    float* rks = red + 32;                   // [ROWS] r_i = exp(alpha - rowmax_i)
    float* mrs = rks + ROWS;                 // [ROWS] rowmax_i
    float* asv = mrs + ROWS;                 // [ROWS] a_i of the last iteration (for the potentials)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = blockIdx.x / p.G, w = blockIdx.x % p.G;
    const int G = p.G, cs = p.cs, N = p.N, M = p.M;
    const int row0 = w * ROWS + wave * RW;
    u64* const bufA = p.bufA + (int64_t)grp * G * G * cs;
    u64* const bufB = p.bufB + (int64_t)grp * G * cs;
    u64* const bufU2 = p.bufU + (int64_t)grp * 2 * G;
    bool dead = false;
    const bool nap = !(p.flags & 1);
    const float mu = 1.0f / (float)(M + N), muM = (float)N / (float)(M + N), nuN = (float)M / (float)(M + N);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(bufA, 0, G * G * cs * 8, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(bufB, 0, G * cs * 8, 0x00020000);
    float* const kl = klds + wave * SK2K_RL * W + 4 * lane;  // this lane's first chunk of the wave's LDS rows (chunk c: + 256 c)
    unsigned round = 0;
    for (int b = grp; b < p.B; b += p.n_res, ++round) {
        const unsigned ebase = round * (unsigned)p.iters;
        const float* Sb = p.S + (int64_t)b * M * p.ldS;
        asm volatile("" ::: "v255", "a255");  // (the wave is allocated 256 + 256 registers: see sinkhorn_resident128)
        // ---- load the wave's 16 rows, shift by the row maximum, exponentiate once
        sk_static_for<RW>([&](auto r_c) {
            constexpr int r = decltype(r_c)::value;
            const int row = FULL ? row0 + r : min(row0 + r, M - 1);
            f32x4 zz[8];
            float mx = p.alpha;
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                const int c = 4 * (lane + 64 * c8);
                zz[c8] = (FULL || c < p.ldS) ? *reinterpret_cast<const f32x4*>(Sb + (int64_t)row * p.ldS + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (FULL || c + e < N) mx = fmaxf(mx, zz[c8][e]);
            }
            mx = wave_max_dpp(mx);
            const bool rvalid = FULL || row0 + r < M;
            if (lane == 0) {
                mrs[wave * RW + r] = mx;
                rks[wave * RW + r] = rvalid ? exp_accurate(p.alpha - mx) : 0.f;
            }
            sk_static_for<8>([&](auto c_c) {
                constexpr int c8 = decltype(c_c)::value, h = c8 >> 2, k = c8 & 3;
                const int c = 4 * (lane + 64 * c8);
                f32x4 kv;
#pragma unroll
                for (int e = 0; e < 4; ++e) kv[e] = (rvalid && (FULL || c + e < N)) ? exp_accurate(zz[c8][e] - mx) : 0.f;
                if constexpr (r < SK2K_RV) {
                    const float k0 = kv[0], k1 = kv[1], k2 = kv[2], k3 = kv[3];
                    asm volatile("v_mov_b32 v[%4], %0\n\tv_mov_b32 v[%4+1], %1\n\tv_mov_b32 v[%4+2], %2\n\tv_mov_b32 v[%4+3], %3"
                                 :: "v"(k0), "v"(k1), "v"(k2), "v"(k3), "n"(sk2k_base(r, h) + 4 * k));
                } else if constexpr (r < SK2K_RR) {
                    const float k0 = kv[0], k1 = kv[1], k2 = kv[2], k3 = kv[3];
                    asm volatile("v_accvgpr_write_b32 a[%4], %0\n\tv_accvgpr_write_b32 a[%4+1], %1\n\tv_accvgpr_write_b32 a[%4+2], %2\n\tv_accvgpr_write_b32 a[%4+3], %3"
                                 :: "v"(k0), "v"(k1), "v"(k2), "v"(k3), "n"(sk2k_base(r, h) + 4 * k));
                } else {
                    *reinterpret_cast<f32x4*>(kl + (r - SK2K_RR) * W + 256 * c8) = kv;
                }
            });
            __builtin_amdgcn_sched_barrier(0);  // one row in flight
        });
        for (int c = tid; c < W + 4; c += 256) vbuf[c] = (c < N || c == W) ? 1.f : 0.f;
        __syncthreads();
        float bN = 1.f, aM = 0.f;
        const unsigned kl_a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)kl;

        for (int it = 0; it < p.iters; ++it) {
            const unsigned epoch = ebase + (unsigned)it + 1u;
            int tq = tid;
            asm volatile("" : "+v"(tq));
            u64* const bufU = bufU2 + (epoch & 1u) * (unsigned)G;
            const bool last = it + 1 == p.iters;
            SK_STAMP_PTR();
            SK_STAMP(0);
            // ---- row sums, per column half: one register per row
            float accp[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) accp[r] = 0.f;
            f32x2 accb = {0.f, 0.f};
            sk_static_for<2>([&](auto h_c) {
                constexpr int h = decltype(h_c)::value;
                f32x2 blo[4], bhi[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(vbuf + 4 * (lane + 64 * (4 * h + k)));  // 0 beyond N
                    blo[k] = f32x2{b4[0], b4[1]};
                    bhi[k] = f32x2{b4[2], b4[3]};
                    accb += blo[k] + bhi[k];
                }
                // register rows in pairs; the first four pair statements also fetch one LDS row (half) each, consumed right behind them
                auto lds_row_sum = [&](const f32x4 (&t)[4], int j) __attribute__((always_inline)) {
                    f32x2 acc = {0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        acc = __builtin_elementwise_fma(f32x2{t[k][0], t[k][1]}, blo[k], acc);
                        acc = __builtin_elementwise_fma(f32x2{t[k][2], t[k][3]}, bhi[k], acc);
                    }
                    accp[SK2K_RR + j] += acc[0] + acc[1];
                    asm volatile("" : "+v"(accp[SK2K_RR + j]));
                };
                sk_static_for<SK2K_RV / 2>([&](auto p_c) {
                    constexpr int pp = decltype(p_c)::value, r0 = 2 * pp;
                    f32x2 acc[4];
                    f32x4 t[4];
                    sk_rs2v_l<sk2k_base(r0, h), sk2k_base(r0 + 1, h), (pp * W + 1024 * h) * 4>(acc, blo, bhi, t, kl_a);
                    accp[r0] += (acc[0][0] + acc[0][1]) + (acc[2][0] + acc[2][1]);
                    accp[r0 + 1] += (acc[1][0] + acc[1][1]) + (acc[3][0] + acc[3][1]);
                    asm volatile("" : "+v"(accp[r0]), "+v"(accp[r0 + 1]));
                    lds_row_sum(t, pp);
                });
                sk_static_for<SK2K_RA / 2>([&](auto p_c) {
                    constexpr int pp = decltype(p_c)::value, r0 = SK2K_RV + 2 * pp;
                    f32x2 a0, a1;
                    if constexpr (SK2K_RV / 2 + pp < SK2K_RL) {
                        f32x4 t[4];
                        sk128_rs2a_l<sk2k_base(r0, h), sk2k_base(r0 + 1, h), ((SK2K_RV / 2 + pp) * W + 1024 * h) * 4>(a0, a1, blo, bhi, t, kl_a);
                        lds_row_sum(t, SK2K_RV / 2 + pp);
                    } else {
                        sk128_rs2a<sk2k_base(r0, h), sk2k_base(r0 + 1, h)>(a0, a1, blo, bhi);
                    }
                    accp[r0] += a0[0] + a0[1];
                    accp[r0 + 1] += a1[0] + a1[1];
                    asm volatile("" : "+v"(accp[r0]), "+v"(accp[r0 + 1]));
                });
            });
            aM = muM / (wave_sum_dpp(accb[0] + accb[1]) + bN);
            // ---- a_i: four rows per reduction and division; the scalars a_i for the column sums
            float as[RW];
            float ra4 = 0.f;
#pragma unroll
            for (int g = 0; g < RW / 4; ++g) {
                const float s_r = wave_sum4_dpp(accp[4 * g], accp[4 * g + 1], accp[4 * g + 2], accp[4 * g + 3], lane);
                const int rl = wave * RW + 4 * g + (lane & 3);
                const float rk = rks[rl];
                const float ar = (FULL || row0 + 4 * g + (lane & 3) < M) ? mu / fmaf(rk, bN, s_r) : 0.f;
                ra4 = fmaf(rk, ar, ra4);
                if (last) asv[rl] = ar;
#pragma unroll
                for (int q = 0; q < 4; ++q) as[4 * g + q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ar), q));
            }
            const float ra = (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ra4), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ra4), 1)))
                             + (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ra4), 2)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ra4), 3)));
            // ---- column sums, per half; both halves stay in registers until the fold
            f32x2 cl[2][4], ch[2][4];
            sk_static_for<2>([&](auto h_c) {
                constexpr int h = decltype(h_c)::value;
#pragma unroll
                for (int k = 0; k < 4; ++k) { cl[h][k] = f32x2{0.f, 0.f}; ch[h][k] = f32x2{0.f, 0.f}; }
                auto lds_row_cols = [&](const f32x4 (&t)[4], int j) __attribute__((always_inline)) {
                    const f32x2 a2 = {as[SK2K_RR + j], as[SK2K_RR + j]};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        cl[h][k] = __builtin_elementwise_fma(f32x2{t[k][0], t[k][1]}, a2, cl[h][k]);
                        ch[h][k] = __builtin_elementwise_fma(f32x2{t[k][2], t[k][3]}, a2, ch[h][k]);
                    }
                    asm volatile("" : "+v"(cl[h][0]), "+v"(cl[h][1]), "+v"(cl[h][2]), "+v"(cl[h][3]), "+v"(ch[h][0]), "+v"(ch[h][1]), "+v"(ch[h][2]), "+v"(ch[h][3]));
                };
                sk_static_for<SK2K_RV / 2>([&](auto p_c) {
                    constexpr int pp = decltype(p_c)::value, r0 = 2 * pp;
                    f32x4 t[4];
                    sk_rc2v_l<sk2k_base(r0, h), sk2k_base(r0 + 1, h), (pp * W + 1024 * h) * 4>(cl[h], ch[h], f32x2{as[r0], as[r0]}, f32x2{as[r0 + 1], as[r0 + 1]}, t, kl_a);
                    lds_row_cols(t, pp);
                });
                sk_static_for<SK2K_RA / 2>([&](auto p_c) {
                    constexpr int pp = decltype(p_c)::value, r0 = SK2K_RV + 2 * pp;
                    if constexpr (SK2K_RV / 2 + pp < SK2K_RL) {
                        f32x4 t[4];
                        sk_rc2a_l<sk2k_base(r0, h), sk2k_base(r0 + 1, h), ((SK2K_RV / 2 + pp) * W + 1024 * h) * 4>(cl[h], ch[h], f32x2{as[r0], as[r0]}, f32x2{as[r0 + 1], as[r0 + 1]}, t, kl_a);
                        lds_row_cols(t, SK2K_RV / 2 + pp);
                    } else {
                        sk_rc2a<sk2k_base(r0, h), sk2k_base(r0 + 1, h)>(cl[h], ch[h], f32x2{as[r0], as[r0]}, f32x2{as[r0 + 1], as[r0 + 1]});
                    }
                });
            });
            // ---- fold of the 4 waves through 16 KB: waves 2, 3 write; waves 0, 1 add theirs and write back
            {
                float* lf = fold + (wave & 1) * W + 4 * lane;
                if (wave >= 2) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(lf + 256 * (4 * h + k)) = f32x4{cl[h][k][0], cl[h][k][1], ch[h][k][0], ch[h][k][1]};
                }
                if (lane == 0) red[wave] = ra;
                SK_STAMP(1);
                __syncthreads();
                if (wave < 2) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const f32x4 o = *reinterpret_cast<const f32x4*>(lf + 256 * (4 * h + k));
                            *reinterpret_cast<f32x4*>(lf + 256 * (4 * h + k)) = f32x4{cl[h][k][0] + o[0], cl[h][k][1] + o[1], ch[h][k][0] + o[2], ch[h][k][1] + o[3]};
                        }
                }
                __syncthreads();
            }
            // ---- publish the workgroup's partial column sums (stage A, 16-byte pairs: 8 adjacent columns per thread) and its dustbin sum
            {
                const int c = 8 * tq;
#pragma unroll
                for (int q4 = 0; q4 < 2; ++q4) {
                    const f32x4 T = *reinterpret_cast<const f32x4*>(fold + c + 4 * q4) + *reinterpret_cast<const f32x4*>(fold + W + c + 4 * q4);
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int cc = c + 4 * q4 + 2 * hh, wc = cc / cs, jl = cc - wc * cs;
                        if (FULL || cc < N) granule_store2(rsA, (unsigned)((wc * G + w) * cs + jl) * 8u, epoch, T[2 * hh], T[2 * hh + 1]);
                    }
                }
                if (tq == 0) {
                    const float U = (red[0] + red[1]) + (red[2] + red[3]);
                    granule_store(bufU + w, epoch, U);
                }
            }
            SK_STAMP(2);
            // ---- stage A consume: my slice of columns over all producers -> b_j = mu / (sum + a_M), published as stage B
            {
                // 8 lanes per column pair, each two producers per wait (four per wait - all 32 producers in one round trip - was
                // slower: 4.1 against 2.9 us for this stage, the polls themselves load the memory system)
                const int q = tq & 7, cg = tq >> 3;
                const unsigned base_b = (unsigned)(w * G * cs) * 8u;
                for (int j0 = 0; j0 < cs; j0 += 64) {
                    const int jl = j0 + 2 * cg, c = w * cs + jl;
                    const bool act = jl < cs && c < N;
                    float T0 = 0.f, T1 = 0.f;
                    for (int g0 = 0; g0 < G; g0 += 16) {
                        unsigned off[2];
                        unsigned val[2][2];
                        int n = 0;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int g = g0 + q + 8 * i;
                            off[i] = base_b;
                            if (act && g < G) { off[i] = base_b + (unsigned)(g * cs + jl) * 8u; n = i + 1; }
                        }
                        granule_wait2<2>(rsA, off, n, epoch, val, p.timeout, dead, nap);
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            if (i < n) { T0 += __uint_as_float(val[i][0]); T1 += __uint_as_float(val[i][1]); }
                    }
#pragma unroll
                    for (int o = 4; o > 0; o >>= 1) { T0 += __shfl_xor(T0, o); T1 += __shfl_xor(T1, o); }
                    if (act && q == 0) granule_store2(rsB, (unsigned)c * 8u, epoch, mu / (T0 + aM), mu / (T1 + aM));
                }
            }
            SK_STAMP(3);
            // ---- b_N = nu_N / (sum_i r_i a_i + a_M) from the G workgroup sums (wave 0)
            if (wave == 0) {
                float U = 0.f;
                for (int g0 = 0; g0 < G; g0 += 64) {
                    const int g = g0 + lane;
                    int off[1] = {g < G ? g : 0};
                    unsigned val[1];
                    granule_wait<1>(bufU, off, g < G ? 1 : 0, epoch, val, p.timeout, dead, nap);
                    if (g < G) U += __uint_as_float(val[0]);
                }
                U = wave_sum_dpp(U);
                if (lane == 0) vbuf[W] = nuN / (U + aM);
            }
            SK_STAMP(4);
            // ---- stage B consume: all of b into LDS (4 pairs per thread, one wait)
            {
                unsigned off[4];
                unsigned val[4][2];
                int n = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ca = 2 * tq + 512 * i;
                    off[i] = 0u;
                    if (ca < N) { off[i] = (unsigned)ca * 8u; n = i + 1; }
                }
                granule_wait2<4>(rsB, off, n, epoch, val, p.timeout, dead, nap);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ca = 2 * tq + 512 * i;
                    *reinterpret_cast<f32x2*>(vbuf + ca) = f32x2{ca < N ? __uint_as_float(val[i][0]) : 0.f, ca + 1 < N ? __uint_as_float(val[i][1]) : 0.f};
                }
            }
            SK_STAMP(5);
            if (__syncthreads_or(dead ? 1 : 0)) dead = true;
            SK_STAMP(6);
            bN = vbuf[W];
        }

        // ---- potentials for the final sweep (as in sinkhorn_resident)
        {
            const float qnan = __uint_as_float(SKR_GAVE_UP_NAN);
            float* ub = p.u + (int64_t)b * (M + 1);
            bool bad = false;
            if (tid < ROWS && w * ROWS + tid < M) {
                const float ar = asv[tid];
                bad = bad || !(ar > 0.f) || !(ar < INFINITY);
                ub[w * ROWS + tid] = dead ? qnan : __logf(ar) - mrs[tid];
            }
            if (w == 0) {
                float* vb = p.v + (int64_t)b * p.ldV;
                for (int j = tid; j < p.ldV; j += 256) {
                    const float bj = j < N ? vbuf[j] : (j == N ? bN : 1.f);
                    bad = bad || !(bj > 0.f) || !(bj < INFINITY);
                    vb[j] = dead ? qnan : __logf(bj);
                }
                bad = bad || !(aM > 0.f) || !(aM < INFINITY);
                if (tid == 0) ub[M] = dead ? qnan : __logf(aM) - p.alpha;
            }
            if (__syncthreads_or(bad ? 1 : 0) && tid == 0) atomicAdd(p.timeout + 4, 1u);
        }
    }
}

// ---- rescue pass behind the resident kernel -----------------------------------------------------------------------------
// One workgroup per problem looks at the potentials the resident kernel left.  All finite (every call of an ordinary
// network): return - the pass costs one launch of B idle workgroups.  Otherwise (a scaling left fp32's range in the
// exponential domain, or an inter-workgroup wait gave up under contention) this workgroup re-solves ITS problem alone in
// the log domain, upstream's u = log_mu - LSE_j(C + v), v = log_nu - LSE_i(C + u): no range limit, no inter-workgroup
// wait, scores streamed from L2 / HBM twice per iteration (milliseconds per problem - a rare path).  flags[3] counts the
// rescued problems; flags[1] the problems whose potentials are non-finite even so (non-finite scores: a real error,
// reported by e2emv_sync).
__global__ __launch_bounds__(1024) void sinkhorn_rescue(SkParams p, int iters, unsigned* flags) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = p.M, N = p.N;
    float* ub = p.u + (int64_t)b * (M + 1);
    float* vb = p.v + (int64_t)b * p.ldV;
    bool bad = false, gave_up = false;
    for (int i = tid; i <= M; i += 1024) { const float x = ub[i]; bad = bad || !(fabsf(x) < INFINITY); gave_up = gave_up || __float_as_uint(x) == SKR_GAVE_UP_NAN; }
    for (int j = tid; j <= N; j += 1024) { const float x = vb[j]; bad = bad || !(fabsf(x) < INFINITY); gave_up = gave_up || __float_as_uint(x) == SKR_GAVE_UP_NAN; }
    if (!__syncthreads_or(bad ? 1 : 0)) return;
    const int timed_out = __syncthreads_or(gave_up ? 1 : 0);  // THIS problem's reason (the launch-global flag says nothing about it)
    float* su = lds;            // [M + 1]
    float* sv = lds + (M + 1);  // [N + 1]
    const float* Sb = p.S + (int64_t)b * M * p.ldS;
    for (int j = tid; j <= N; j += 1024) sv[j] = 0.f;
    __syncthreads();
    const float log_mu_bin = __logf((float)N) + p.norm, log_nu_bin = __logf((float)M) + p.norm;
    for (int it = 0; it < iters; ++it) {
        for (int i = wave; i <= M; i += 16) {  // one wave per row
            float mx = -INFINITY;
            for (int j = lane; j <= N; j += 64) mx = fmaxf(mx, ((i < M && j < N) ? Sb[(int64_t)i * p.ldS + j] : p.alpha) + sv[j]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            float sm = 0.f;
            for (int j = lane; j <= N; j += 64) sm += __expf(((i < M && j < N) ? Sb[(int64_t)i * p.ldS + j] : p.alpha) + sv[j] - mx);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
            if (lane == 0) su[i] = (i < M ? p.norm : log_mu_bin) - (mx + __logf(sm));
        }
        __syncthreads();
        for (int j = tid; j <= N; j += 1024) {  // one thread per column, running maximum
            float mx = -INFINITY, sm = 0.f;
            for (int i = 0; i <= M; ++i) {
                const float x = ((i < M && j < N) ? Sb[(int64_t)i * p.ldS + j] : p.alpha) + su[i];
                if (x > mx) { sm = sm * __expf(mx - x) + 1.f; mx = x; } else { sm += __expf(x - mx); }
            }
            sv[j] = (j < N ? p.norm : log_nu_bin) - (mx + __logf(sm));
        }
        __syncthreads();
    }
    if (iters <= 0) {
        for (int i = tid; i <= M; i += 1024) su[i] = 0.f;
        __syncthreads();
    }
    bad = false;
    for (int i = tid; i <= M; i += 1024) { ub[i] = su[i]; bad = bad || !(fabsf(su[i]) < INFINITY); }
    for (int j = tid; j < p.ldV; j += 1024) {
        const float x = j <= N ? sv[j] : 0.f;
        vb[j] = x;
        bad = bad || !(fabsf(x) < INFINITY);
    }
    const int still = __syncthreads_or(bad ? 1 : 0);
    // [1] non-finite even in the log domain (non-finite scores: an error); otherwise rescued - [6] when a wait of the resident
    // kernel gave up on THIS problem (contention: says nothing about the model), [3] when not (a scaling left fp32's range)
    if (tid == 0) atomicAdd(flags + (still ? 1 : (timed_out ? 6 : 3)), 1u);
}

// streaming chain: the potentials of a problem with non-finite scores are non-finite - counted like the resident path's
// (flags[1], reported by e2emv_sync / check_finite)
__global__ __launch_bounds__(256) void sinkhorn_check_finite(SkParams p, unsigned* flags) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* ub = p.u + (int64_t)b * (p.M + 1);
    const float* vb = p.v + (int64_t)b * p.ldV;
    bool bad = false;
    for (int i = tid; i <= p.M; i += 256) bad = bad || !(fabsf(ub[i]) < INFINITY);
    for (int j = tid; j <= p.N; j += 256) bad = bad || !(fabsf(vb[j]) < INFINITY);
    if (__syncthreads_or(bad ? 1 : 0) && tid == 0) atomicAdd(flags + 1, 1u);
}

static int round_up(int x, int m) { return (x + m - 1) / m * m; }

// geometry of the resident kernel for a problem size and `slots` co-resident workgroups
struct ResidentPlan {
    int G, n_res, cs;
    size_t bytesA, bytesB, bytesU;
};
static int round_up(int x, int m);
static ResidentPlan resident_plan(int B, int M, int N, int slots, int rows_per_wg = 0) {
    ResidentPlan r;
    const int rows = rows_per_wg > 0 ? rows_per_wg : skr_rows(round_up(N, 4));
    r.G = (M + rows - 1) / rows;
    r.n_res = std::max(1, std::min(B, slots / std::max(r.G, 1)));
    r.cs = (N + r.G - 1) / r.G;
    auto al = [](size_t n) { return (n + 255) & ~size_t(255); };
    r.bytesA = al((size_t)r.n_res * r.G * r.G * r.cs * 8);
    r.bytesB = al((size_t)r.n_res * r.G * r.cs * 8);
    r.bytesU = al((size_t)r.n_res * 2 * r.G * 8);
    return r;
}
static size_t resident_ws_bytes(int B, int M, int N, int slots) {
    if (round_up(N, 4) > 2048) return 0;
    // n_res grows with the slot count until it reaches B: the largest plan is the one at `slots`
    const ResidentPlan r = resident_plan(B, M, N, slots);
    return r.bytesA + r.bytesB + r.bytesU + 256;
}

size_t sinkhorn_ws_bytes(int B, int M, int N) {
    const int ldS = round_up(N, 4);
    const int chunks = (M + SK_ROWS - 1) / SK_ROWS;
    size_t f = 0;
    auto al = [](size_t n) { return (n * 4 + 255) & ~size_t(255); };
    f += al((size_t)B * (M + 1));                 // u
    f += 2 * al((size_t)B * (ldS + 4));           // v ping-pong
    f += 2 * al((size_t)B * chunks * ldS);        // pm / ps (re-used as pv / pi)
    f += 2 * al((size_t)B * chunks);              // upm / ups
    f += 2 * al((size_t)B * M);                   // max0, idx0
    f += resident_ws_bytes(B, M, N, 1024);        // granule buffers of the resident kernel (upper bound: 4 workgroups / CU)
    return f;
}

static int KT_of(int64_t ldS) {
    const int kt = (int)((ldS + 255) / 256);
    return kt <= 2 ? kt : (kt <= 4 ? 4 : 8);
}

static void hipLaunchKernelGGL_ptr(const void* fn, dim3 grid, dim3 block, size_t lds, hipStream_t s, SkResParams& par) {  // (block: 512, or 256 for sinkhorn_resident128)
    void* args[] = {&par};
    (void)hipLaunchKernel(fn, grid, block, args, lds, s);
}

template <int KT>
static void launch_sweeps(const SkParams& p, int B, bool final, hipStream_t s) {
    const size_t lds = sizeof(float) * 8 * KT * 256;
    const bool full = p.N == p.ldS && p.N == KT * 256;
    if (!final) {
        if (full) hipLaunchKernelGGL((sinkhorn_sweep<KT, false, true>), dim3(p.chunks, B), dim3(256), lds, s, p);
        else hipLaunchKernelGGL((sinkhorn_sweep<KT, false, false>), dim3(p.chunks, B), dim3(256), lds, s, p);
    } else {
        if (full) hipLaunchKernelGGL((sinkhorn_sweep<KT, true, true>), dim3(p.chunks + 1, B), dim3(256), lds, s, p);
        else hipLaunchKernelGGL((sinkhorn_sweep<KT, true, false>), dim3(p.chunks + 1, B), dim3(256), lds, s, p);
    }
}

// every polled word of a resident launch starts from 0 (epochs count from 1), and so does the launch's give-up flag
__global__ __launch_bounds__(256) void skr_zero_kernel(uint4* buf, size_t n16, unsigned* flag) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) buf[i] = uint4{0u, 0u, 0u, 0u};
    if (blockIdx.x == 0 && threadIdx.x == 0) flag[0] = 0u;
}

// ---- the launcher's plan, shared by launch_sinkhorn and the e2emv_sinkhorn_plan query (bench.py reports it instead of re-deriving it)
struct SkKernel { const void* fn = nullptr; int rows = 0, threads = 512, wg_per_cu = 0; size_t lds = 0; bool big = false; };
struct SkSegment { int b0 = 0, n = 0; SkKernel k; int resident = 0; };
struct SkPlan { int n_seg = 0; SkSegment seg[2]; };  // n_seg == 0: the log-domain launch chain

// `count`: this is a real call (the demotion counters of the context advance); false for the query
static int plan_sinkhorn(e2emv_ctx* ctx, int B, int M, int N, int64_t ldS, int iters, bool count, SkPlan& plan) {
    plan = SkPlan{};
    bool resident = iters >= 1 && ldS <= 2048;
    // which kernel: the context's pin (e2emv_set_sinkhorn_kernel; initialised ONCE from E2EMV_SINKHORN when the context is made)
    const int pin = ctx->sinkhorn_kernel;
    if (pin == E2EMV_SINKHORN_STREAM) resident = false;
    // once a call of this context reported scores outside the exponential-domain kernel's range (e2emv_sync / e2emv_get_stats),
    // the model at hand is served by the log-domain chain: slower, no range limit
    // (demotion needs two observed range events; after 16 calls on the chain the resident kernel gets another try - a model whose
    // scores really are out of its range is demoted again by the next event)
    if (ctx->sinkhorn_stream) {
        if (!count) resident = false;
        else if (++ctx->sk_stream_calls > 16) { ctx->sinkhorn_stream = false; ctx->sk_stream_calls = 0; ctx->sk_range_strikes = 1; }
        else resident = false;
    }
    if (!resident) return E2EMV_OK;
    // A call is served by one or two resident launches (segments of the batch): the kernels with K in registers addressed by number
    // (sinkhorn_resident128 at 513 .. 1024 columns, sinkhorn_resident2k at 1025 .. 2048: twice the rows per workgroup, twice the
    // problems resident, a round 1.6 - 1.75 times as long - measured 0.83 - 0.94 against 0.51 - 0.53 ms per 100 iterations at
    // 1024 x 1024, 1.21 against 0.80 at 2048 x 2048) take every FULL round of theirs, the remainder goes to whichever is cheaper:
    // rounds of the compiler-allocated kernel, or one more round of the big one.  80 problems of 1024 x 1024 = 64 + 16.
    // Pin rows64: never the big kernels; rows128: the big kernel for the whole batch whenever the shape allows it - with a pin a
    // problem's result does not depend on its batch neighbours (tests compare a problem alone with the same problem in a batch).
    SkKernel kbase, kbig;
    const bool full = N == ldS && N == KT_of(ldS) * 256;
    // granule pairs (16-byte exchange stores / loads): a thread must own an even number of columns and a consumer's column
    // slice must be even
    const int G0 = (M + skr_rows(ldS) - 1) / skr_rows(ldS);
    const bool pairs = KT_of(ldS) >= 4 && ((N + G0 - 1) / G0) % 2 == 0 && dbg_knob("E2EMV_SKR_PAIR", 1) != 0;
    const void* kfn = nullptr;
    switch (KT_of(ldS)) {
        case 1: kfn = full ? (const void*)sinkhorn_resident<1, true> : (const void*)sinkhorn_resident<1, false>; break;
        case 2: kfn = full ? (const void*)sinkhorn_resident<2, true> : (const void*)sinkhorn_resident<2, false>; break;
        case 4:
            if (skr_rw(ldS) == 8) {
                if (pairs) kfn = full ? (const void*)sinkhorn_resident<4, true, true, 8> : (const void*)sinkhorn_resident<4, false, true, 8>;
                else kfn = full ? (const void*)sinkhorn_resident<4, true, false, 8> : (const void*)sinkhorn_resident<4, false, false, 8>;
            } else if (pairs) kfn = full ? (const void*)sinkhorn_resident<4, true, true> : (const void*)sinkhorn_resident<4, false, true>;
            else kfn = full ? (const void*)sinkhorn_resident<4, true> : (const void*)sinkhorn_resident<4, false>;
            break;
        default:
            if (pairs) kfn = full ? (const void*)sinkhorn_resident<8, true, true> : (const void*)sinkhorn_resident<8, false, true>;
            else kfn = full ? (const void*)sinkhorn_resident<8, true> : (const void*)sinkhorn_resident<8, false>;
            break;
    }
    kbase.fn = kfn; kbase.rows = skr_rows(ldS); kbase.threads = 512;
    kbase.lds = sizeof(float) * (size_t)(9 * KT_of(ldS) * 256 + 4 + 32);
    if (KT_of(ldS) == 4) {
        const int G128 = (M + 127) / 128, cs128 = (N + G128 - 1) / G128;
        if (cs128 % 2 == 0 && G128 <= 16) {
            kbig.fn = (full && M % 128 == 0) ? (const void*)sinkhorn_resident128<true> : (const void*)sinkhorn_resident128<false>;  // (full rows AND columns)
            kbig.rows = 128;
            kbig.lds = sizeof(float) * (size_t)(4 * SK128_RL * 1024 + 4 * 1024 + 1024 + 4 + 32 + 3 * 128);
        }
    } else if (KT_of(ldS) == 8) {
        const int G2k = (M + 63) / 64, cs2k = (N + G2k - 1) / G2k;
        if (cs2k % 2 == 0 && G2k <= 64) {
            kbig.fn = (full && M % 64 == 0) ? (const void*)sinkhorn_resident2k<true> : (const void*)sinkhorn_resident2k<false>;
            kbig.rows = 64;
            kbig.lds = sizeof(float) * (size_t)(4 * SK2K_RL * 2048 + 2 * 2048 + 2048 + 4 + 32 + 3 * 64);
        }
    }
    kbig.threads = 256; kbig.big = true;
    static std::map<std::pair<int, const void*>, int> occupancy;  // (device, kernel) -> resident workgroups per CU
    static std::mutex occupancy_mu;
    for (SkKernel* k : {&kbase, &kbig}) {
        if (!k->fn) continue;
        std::lock_guard<std::mutex> lk(occupancy_mu);
        auto it = occupancy.find({ctx->device, k->fn});
        if (it == occupancy.end()) {
            int nb = 0;
            if (k->lds > 48 * 1024 && ensure_dynamic_lds(ctx, k->fn, k->lds) != E2EMV_OK) {
                (void)hipGetLastError();  // a device with less LDS: the streaming chain serves the call
                nb = 0;
            } else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k->fn, k->threads, k->lds) != hipSuccess) {
                (void)hipGetLastError();
                nb = 0;
            }
            it = occupancy.emplace(std::make_pair(ctx->device, k->fn), std::min(nb, 2)).first;
        }
        k->wg_per_cu = it->second;
        const int G = (M + k->rows - 1) / k->rows;
        if (k->wg_per_cu < 1 || G > k->wg_per_cu * ctx->num_cus) k->fn = nullptr;  // cannot hold a problem's workgroups at once
    }
    if (!kbase.fn) { if (kbig.fn) kbase = kbig; else return E2EMV_OK; }
    auto res_of = [&](const SkKernel& k) { return std::max(1, (k.wg_per_cu * ctx->num_cus) / std::max((M + k.rows - 1) / k.rows, 1)); };
    auto add = [&](int b0, int n, const SkKernel& k) {
        SkSegment& sg = plan.seg[plan.n_seg++];
        sg.b0 = b0; sg.n = n; sg.k = k; sg.resident = std::min(n, res_of(k));
    };
    if (!kbig.fn || kbase.big || pin == E2EMV_SINKHORN_ROWS64) {
        add(0, B, kbase);
    } else if (pin == E2EMV_SINKHORN_ROWS128) {
        add(0, B, kbig);
    } else {
        const int res_big = res_of(kbig), res_base = res_of(kbase);
        int n_big = (B / res_big) * res_big;   // every full round of the big kernel (it holds twice the problems at < 2 x the time)
        const int rem = B - n_big;
        if (rem > 0 && 8 * ((rem + res_base - 1) / res_base) > 13) n_big = B;  // the remainder too: one round of 1.6 against two or more of 1
        if (n_big > 0) add(0, n_big, kbig);
        if (n_big < B) add(n_big, B - n_big, kbase);
    }
    return E2EMV_OK;
}


int launch_sinkhorn(e2emv_ctx* ctx, int B, int M, int N, const float* S, int64_t ldS, float alpha, int iters,
                    float match_thr, const SinkhornOut& out, char* ws, hipStream_t s) {
    if (B <= 0 || M <= 0 || N <= 0) return set_err(ctx, E2EMV_ESHAPE, "sinkhorn: empty problem");
    if (ldS % 4 || ldS < N || ((uintptr_t)S % 16)) return set_err(ctx, E2EMV_ESHAPE, "sinkhorn: score rows must be 16-byte aligned");
    if (ldS > 2048) return set_err(ctx, E2EMV_ESHAPE, "sinkhorn: N=%d > 2048 keypoints not supported", N);
    if ((size_t)(2 * M + N) * 4 > 60000) return set_err(ctx, E2EMV_ESHAPE, "sinkhorn: M=%d too large", M);
    SkParams p{};
    p.S = S; p.ldS = ldS; p.M = M; p.N = N;
    p.chunks = (M + SK_ROWS - 1) / SK_ROWS;
    p.alpha = alpha;
    p.norm = -logf((float)(M + N));
    auto al = [](size_t n) { return (n * 4 + 255) & ~size_t(255); };
    char* w = ws;
    p.u = (float*)w; w += al((size_t)B * (M + 1));
    p.ldV = ldS + 4;
    float* v0 = (float*)w; w += al((size_t)B * p.ldV);
    float* v1 = (float*)w; w += al((size_t)B * p.ldV);
    p.pm = (float*)w; w += al((size_t)B * p.chunks * ldS);
    p.ps = (float*)w; w += al((size_t)B * p.chunks * ldS);
    p.upm = (float*)w; w += al((size_t)B * p.chunks);
    p.ups = (float*)w; w += al((size_t)B * p.chunks);
    p.max0 = (float*)w; w += al((size_t)B * M);
    p.idx0 = (int*)w; w += al((size_t)B * M);
    p.pv = p.pm;
    p.pi = (int*)p.ps;
    const int gb = out.group_batch > 0 ? out.group_batch : B;
    if (out.n_groups < 1 || out.n_groups > kMaxGroups || gb * out.n_groups != B)
        return set_err(ctx, E2EMV_EINVAL, "sinkhorn: %d groups x %d != batch %d", out.n_groups, gb, B);
    p.group_batch = gb;
    bool want_match = false;
    for (int g = 0; g < kMaxGroups; ++g) {
        p.logZ[g] = g < out.n_groups ? out.logZ[g] : nullptr;
        if (g < out.n_groups && (out.m0[g] || out.m1[g] || out.ms0[g] || out.ms1[g])) want_match = true;
    }
    p.v = v0;
    p.v_next = v1;
    const int KT = (int)((ldS + 255) / 256);
    auto sweep = [&](bool final) {
        switch (KT) {
            case 1: launch_sweeps<1>(p, B, final, s); break;
            case 2: launch_sweeps<2>(p, B, final, s); break;
            case 3: case 4: launch_sweeps<4>(p, B, final, s); break;
            default: launch_sweeps<8>(p, B, final, s); break;
        }
    };

    // ---- resident path: all iterations in one launch (S read once); the streaming chain below is the fallback for
    // iters == 0, for the `stream` pin and for devices that cannot hold a problem's workgroups at once (plan_sinkhorn above)
    SkPlan plan;
    if (int rc_p = plan_sinkhorn(ctx, B, M, N, ldS, iters, true, plan)) return rc_p;
    bool resident = plan.n_seg > 0;
    if (resident) {
        if (int rc_f = ensure_flags(ctx)) return rc_f;
        const SkSegment* seg = plan.seg;
        const int n_seg = plan.n_seg;
        const char* dbg_path = dbg_env("E2EMV_SKR_DEBUG");
        for (int si = 0; si < n_seg; ++si) {
            const SkKernel& k = seg[si].k;
            const int b0 = seg[si].b0, nb = seg[si].n;
            const ResidentPlan rp = resident_plan(nb, M, N, k.wg_per_cu * ctx->num_cus, k.rows);
            SkResParams rpar{};
            rpar.S = S + (int64_t)b0 * M * ldS; rpar.ldS = ldS; rpar.M = M; rpar.N = N; rpar.B = nb; rpar.iters = iters;
            rpar.alpha = alpha; rpar.norm = p.norm;
            rpar.G = rp.G; rpar.n_res = rp.n_res; rpar.cs = rp.cs;
            char* gw = w;  // granule buffers follow the streaming path's arrays in the workspace (segments run one after the other)
            rpar.bufA = (u64*)gw; gw += rp.bytesA;
            rpar.bufB = (u64*)gw; gw += rp.bytesB;
            rpar.bufU = (u64*)gw; gw += rp.bytesU;
            rpar.timeout = ctx->d_flags;
            unsigned long long* d_dbg = nullptr;
            const size_t dbg_bytes = (size_t)16 * rp.G * 8 * sizeof(unsigned long long);
            if (dbg_path && si == 0 && hipMalloc((void**)&d_dbg, dbg_bytes) == hipSuccess) (void)hipMemsetAsync(d_dbg, 0, dbg_bytes, s);
            rpar.dbg = d_dbg;
            rpar.flags = dbg_knob("E2EMV_SKR_FLAGS", rpar.flags);
            rpar.u = p.u + (int64_t)b0 * (M + 1); rpar.v = p.v + (int64_t)b0 * p.ldV; rpar.ldV = p.ldV;
            // every polled word starts from 0 in every launch (epochs count from 1)
            // (ONE launch for the granule buffers and the give-up flag: two hipMemsetAsync were two fill kernels of ~14 us each per segment)
            hipLaunchKernelGGL(skr_zero_kernel, dim3((unsigned)std::min<size_t>(1024, ((rp.bytesA + rp.bytesB + rp.bytesU) / 16 + 255) / 256)), dim3(256), 0, s,
                               reinterpret_cast<uint4*>(w), (rp.bytesA + rp.bytesB + rp.bytesU) / 16, ctx->d_flags);
            E2EMV_CHECK_LAUNCH(ctx, "skr_zero_kernel");
            hipLaunchKernelGGL_ptr(k.fn, dim3((unsigned)(rp.n_res * rp.G)), dim3(k.threads), k.lds, s, rpar);
            E2EMV_CHECK_LAUNCH(ctx, "sinkhorn_resident");
            if (d_dbg) {  // development aid: per-phase timestamps of resident problem 0, appended as text
                std::vector<unsigned long long> h(dbg_bytes / 8);
                (void)hipStreamSynchronize(s);
                (void)hipMemcpy(h.data(), d_dbg, dbg_bytes, hipMemcpyDeviceToHost);
                (void)hipFree(d_dbg);
                if (FILE* f = fopen(dbg_path, "a")) {
                    fprintf(f, "# B=%d M=%d N=%d iters=%d G=%d n_res=%d flags=%d\n", nb, M, N, iters, rp.G, rp.n_res, rpar.flags);
                    for (int it = 0; it < 16 && it < iters; ++it)
                        for (int g = 0; g < rp.G; ++g) {
                            fprintf(f, "%d %d", it, g);
                            for (int kk = 0; kk < 7; ++kk) fprintf(f, " %llu", h[((size_t)it * rp.G + g) * 8 + kk]);
                            fprintf(f, "\n");
                        }
                    fclose(f);
                }
            }
        }
        if (seg[0].k.big) ++ctx->stat_sinkhorn_rows128;
        // problems the exponential-domain kernel could not finish are re-solved in the log domain before anything reads u, v
        hipLaunchKernelGGL(sinkhorn_rescue, dim3(B), dim3(1024), sizeof(float) * (size_t)(M + N + 2), s, p, iters, ctx->d_flags);
        E2EMV_CHECK_LAUNCH(ctx, "sinkhorn_rescue");
    }
    if (!resident) {
        hipLaunchKernelGGL(sinkhorn_init, dim3(B), dim3(256), 0, s, p, B);
        if (iters <= 0) hipLaunchKernelGGL(sinkhorn_zero_u, dim3(B), dim3(256), 0, s, p);
        for (int it = 0; it < iters; ++it) {
            sweep(false);
            hipLaunchKernelGGL(sinkhorn_combine, dim3((unsigned)((p.ldV + 63) / 64), B), dim3(256), 0, s, p);
            std::swap(p.v, p.v_next);
        }
        if (ensure_flags(ctx) == E2EMV_OK) hipLaunchKernelGGL(sinkhorn_check_finite, dim3(B), dim3(256), 0, s, p, ctx->d_flags);
    }
    sweep(true);  // logZ + fused arg-max from the final potentials (one more read of the scores)
    E2EMV_CHECK_LAUNCH(ctx, "sinkhorn kernels");
    if (want_match) {
        MatchParams mp{};
        mp.M = M; mp.N = N; mp.chunks = p.chunks; mp.ldS = ldS;
        mp.max0 = p.max0; mp.idx0 = p.idx0; mp.pv = p.pv; mp.pi = p.pi; mp.idx1_in = nullptr;
        mp.thr = match_thr;
        mp.group_batch = gb;
        for (int g = 0; g < out.n_groups; ++g) {
            mp.m0[g] = out.m0[g]; mp.m1[g] = out.m1[g]; mp.ms0[g] = out.ms0[g]; mp.ms1[g] = out.ms1[g];
        }
        hipLaunchKernelGGL(match_finalize, dim3(B), dim3(MF_THREADS), sizeof(int) * (2 * M + N), s, mp);
        E2EMV_CHECK_LAUNCH(ctx, "match_finalize");
    }
    return E2EMV_OK;
}

}  // namespace e2emv

using namespace e2emv;

extern "C" int e2emv_set_sinkhorn_kernel(e2emv_ctx* ctx, int kernel) {
    if (!ctx) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    if (kernel < E2EMV_SINKHORN_AUTO || kernel > E2EMV_SINKHORN_STREAM) return set_err(ctx, E2EMV_EINVAL, "set_sinkhorn_kernel: %d (E2EMV_SINKHORN_AUTO .. _STREAM)", kernel);
    ctx->sinkhorn_kernel = kernel;
    return E2EMV_OK;
}

extern "C" int e2emv_sinkhorn_plan(e2emv_ctx* ctx, int B, int M, int N, int iters, int* plan_out, int n) {
    if (!ctx || !plan_out || n < 1) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    (void)hipSetDevice(ctx->device);
    if (B <= 0 || M <= 0 || N <= 0 || iters < 0) return set_err(ctx, E2EMV_ESHAPE, "sinkhorn_plan: bad sizes");
    SkPlan plan;
    if (int rc = plan_sinkhorn(ctx, B, M, N, round_up(N, 4), iters, false, plan)) return rc;
    for (int i = 0; i < n; ++i) plan_out[i] = 0;
    plan_out[0] = plan.n_seg;
    for (int si = 0; si < plan.n_seg && 1 + 4 * (si + 1) <= n; ++si) {
        const SkSegment& sg = plan.seg[si];
        plan_out[1 + 4 * si + 0] = sg.k.rows;                               // rows of a problem per workgroup
        plan_out[1 + 4 * si + 1] = sg.n;                                    // problems of the batch this launch takes
        plan_out[1 + 4 * si + 2] = sg.resident;                             // problems resident at a time
        plan_out[1 + 4 * si + 3] = (sg.n + sg.resident - 1) / sg.resident;  // rounds
    }
    return E2EMV_OK;
}

extern "C" int e2emv_sinkhorn(e2emv_ctx* ctx, int B, int M, int N, const float* d_scores, float bin_score, int iters,
                              float* d_logZ, void* stream) {
    if (!ctx || !d_scores || !d_logZ) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (B <= 0 || M <= 0 || N <= 0 || iters < 0) return set_err(ctx, E2EMV_ESHAPE, "sinkhorn: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    const int ldS = round_up(N, 4);
    const bool need_copy = (N % 4 != 0) || ((uintptr_t)d_scores % 16 != 0);
    size_t need = sinkhorn_ws_bytes(B, M, N) + (need_copy ? (((size_t)B * M * ldS * 4 + 255) & ~size_t(255)) : 0);
    int rc = ws_reserve(ctx, need);
    if (rc) return rc;
    char* ws = ctx->d_ws;
    const float* S = d_scores;
    prof_begin(ctx, PS_SINKHORN, s);
    if (need_copy) {
        float* Sp = (float*)ws;
        ws += ((size_t)B * M * ldS * 4 + 255) & ~size_t(255);
        hipLaunchKernelGGL(pad_copy_rows, dim3((unsigned)((int64_t)B * M)), dim3(256), 0, s, d_scores, (int64_t)B * M, N, Sp, (int64_t)ldS);
        S = Sp;
    }
    SinkhornOut out;
    out.logZ[0] = d_logZ;
    rc = launch_sinkhorn(ctx, B, M, N, S, ldS, bin_score, iters, 0.f, out, ws, s);
    prof_end(ctx, s);
    return rc;
}

extern "C" int e2emv_extract_matches(e2emv_ctx* ctx, int B, int M, int N, const float* d_logZ, float match_threshold,
                                     int64_t* d_matches0, int64_t* d_matches1, float* d_mscores0, float* d_mscores1,
                                     void* stream) {
    if (!ctx || !d_logZ) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (B <= 0 || M <= 0 || N <= 0) return set_err(ctx, E2EMV_ESHAPE, "extract_matches: bad sizes");
    if ((size_t)(2 * M + N) * 4 > 60000) return set_err(ctx, E2EMV_ESHAPE, "extract_matches: too many keypoints");
    hipStream_t s = (hipStream_t)stream;
    auto al = [](size_t n) { return (n * 4 + 255) & ~size_t(255); };
    int rc = ws_reserve(ctx, 2 * al((size_t)B * M) + al((size_t)B * N));
    if (rc) return rc;
    char* w = ctx->d_ws;
    float* max0 = (float*)w; w += al((size_t)B * M);
    int* idx0 = (int*)w; w += al((size_t)B * M);
    int* idx1 = (int*)w;
    prof_begin(ctx, PS_MATCH, s);
    hipLaunchKernelGGL(dense_row_argmax, dim3((M + 3) / 4, B), dim3(256), 0, s, d_logZ, M, N, max0, idx0);
    hipLaunchKernelGGL(dense_col_argmax, dim3((N + 255) / 256, B), dim3(256), 0, s, d_logZ, M, N, idx1);
    MatchParams mp{};
    mp.M = M; mp.N = N; mp.chunks = 0; mp.ldS = 0;
    mp.max0 = max0; mp.idx0 = idx0; mp.idx1_in = idx1; mp.thr = match_threshold;
    mp.group_batch = B;
    mp.m0[0] = d_matches0; mp.m1[0] = d_matches1; mp.ms0[0] = d_mscores0; mp.ms1[0] = d_mscores1;
    hipLaunchKernelGGL(match_finalize, dim3(B), dim3(MF_THREADS), sizeof(int) * (2 * M + N), s, mp);
    prof_end(ctx, s);
    E2EMV_CHECK_LAUNCH(ctx, "extract_matches kernels");
    return E2EMV_OK;
}
