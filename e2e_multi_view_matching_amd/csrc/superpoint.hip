// SuperPoint front-end on the device (SURVEY.md 8(f) "next" row 4): image -> keypoints, scores, 256-d descriptors in
// the layout the matcher consumes.
//
// The reference imports models.models.superpoint.SuperPoint from an absent submodule; call sites helpers.py:73-96
// (run_super_point), configs train.py:335-341, eval_pairs.py:197-202, eval_multi_view.py:135-140.  Algorithm = upstream
// magicleap SuperPoint (SuperGluePretrainedNetwork models/superpoint.py), restated in oracle/superpoint.py and pinned
// against the HuggingFace port.
//
// Layout: activations are NHWC fp32 (channels contiguous), so every 3x3 convolution is an implicit GEMM on the fp32
// matrix cores through the CONV mode of gemm_nt_kernel (gemm.hip): M = pixels, N = Cout, K = 9*Cin ordered (ky, kx, cin),
// zero padding resolved in the operand fetch, bias + ReLU fused in the epilogue; the 1x1 heads are plain GEMMs.
//   conv1a (Cin = 1)            direct kernel, 16 lanes x 4 channels per pixel (HBM-bound: 256 B written per pixel)
//   2x2 max-pool                 fused into the epilogue of conv1b / conv2b / conv3b (quad-major row order, 2 lane
//                                shuffles): the full-resolution conv_b outputs never reach memory
//   detector head                convPa (conv GEMM) -> convPb (GEMM, N = 65) -> softmax over 65 + 8x8 depth-to-space
//   simple_nms                   5 max-pools of radius r: tile kernel, row-max then column-max through LDS, -inf padding
//   threshold / borders / top-k  one workgroup per image: ordered compaction, 4-pass radix select of the k-th score,
//                                bitonic sort of the <= 4096 kept keys (score desc, pixel index asc)
//   descriptors                  convDa -> convDb, then per keypoint: bilinear grid_sample(align_corners) of the
//                                L2-normalised cells (norms recomputed for the 4 neighbours), renormalise, write [B,256,K]
#include <cmath>
#include <cstring>

#include "common.h"

namespace e2emv {

typedef __attribute__((ext_vector_type(4))) float sp_f4;

static const char* kSpNames[12] = {"conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convPb", "convDa", "convDb"};
static const int kSpCin[12] = {1, 64, 64, 64, 64, 128, 128, 128, 128, 256, 128, 256};
static const int kSpCout[12] = {64, 64, 64, 64, 128, 128, 128, 128, 256, 65, 256, 256};
static const int kSpK[12] = {3, 3, 3, 3, 3, 3, 3, 3, 3, 1, 3, 1};

// ---- conv1a: 1 -> 64 channels, 3x3, pad 1, bias, ReLU; out NHWC ----------------------------------------------------------
__global__ __launch_bounds__(256) void sp_conv1a_kernel(const float* __restrict__ img, const float* __restrict__ w /*[64][9]*/,
                                                        const float* __restrict__ b, float* __restrict__ out, int H, int W, int64_t npix) {
    __shared__ float sw[64 * 9 + 64];
    for (int i = threadIdx.x; i < 64 * 9 + 64; i += 256) sw[i] = i < 576 ? w[i] : b[i - 576];
    __syncthreads();
    const int64_t pix = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (pix >= npix) return;
    const int c0 = (threadIdx.x & 15) * 4;
    const int64_t hw = (int64_t)H * W;
    const int64_t im = pix / hw;
    const int rem = (int)(pix - im * hw), y = rem / W, x = rem - y * W;
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        v[t] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? img[im * hw + (int64_t)yy * W + xx] : 0.f;
    }
    sp_f4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc = fmaf(v[t], sw[(c0 + e) * 9 + t], acc);
        o[e] = relu_nan(acc + sw[576 + c0 + e]);
    }
    *reinterpret_cast<sp_f4*>(out + pix * 64 + c0) = o;
}

// ---- softmax over the 65 detector channels, drop the dustbin, 8x8 depth-to-space --------------------------------------
__global__ __launch_bounds__(256) void sp_softmax_d2s_kernel(const float* __restrict__ s65, float* __restrict__ score, int Hc, int Wc, int64_t cells) {
    const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (cell >= cells) return;
    const float* src = s65 + cell * 65;
    float v[65], m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 65; ++i) { v[i] = src[i]; m = fmaxf(m, v[i]); }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 65; ++i) { v[i] = expf(v[i] - m); sum += v[i]; }
    const int64_t im = cell / ((int64_t)Hc * Wc);
    const int rem = (int)(cell - im * Hc * Wc), cy = rem / Wc, cx = rem - cy * Wc;
    const int W = Wc * 8;
    float* dst = score + (im * Hc * 8 + cy * 8) * (int64_t)W + cx * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sp_f4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = v[8 * i + e] / sum; b[e] = v[8 * i + 4 + e] / sum; }
        *reinterpret_cast<sp_f4*>(dst + (int64_t)i * W) = a;
        *reinterpret_cast<sp_f4*>(dst + (int64_t)i * W + 4) = b;
    }
}

// ---- simple_nms: one (2r+1)^2 max-pool per launch, -inf outside the image ------------------------------------------------
// mode 0: mask = (s == mp(s))
// mode 1: supp = mp(mask) > 0 ; ss = supp ? 0 : s
// mode 2: mask |= (ss == mp(ss)) & !supp ; if last: out = mask ? s : 0
constexpr int kNmsT = 32;
struct NmsArgs {
    const float* in;   // the map that is pooled (s, mask or ss)
    const float* s;
    float* mask;
    float* supp;
    float* ss;
    float* out;
    int H, W, r, mode, last;
};
__global__ __launch_bounds__(256) void sp_nms_kernel(NmsArgs a) {
    extern __shared__ float lds[];  // tile[(T+2r)][(T+2r)] | rowmax[(T+2r)][T]
    const int r = a.r, R = kNmsT + 2 * r;
    float* tile = lds;
    float* rmax = lds + R * R;
    const int64_t base = (int64_t)blockIdx.z * a.H * a.W;
    const int y0 = blockIdx.y * kNmsT, x0 = blockIdx.x * kNmsT;
    for (int i = threadIdx.x; i < R * R; i += 256) {
        const int ty = i / R, tx = i - ty * R, y = y0 + ty - r, x = x0 + tx - r;
        tile[i] = ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) ? a.in[base + (int64_t)y * a.W + x] : -INFINITY;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R * kNmsT; i += 256) {
        const int ty = i / kNmsT, tx = i - ty * kNmsT;
        float m = -INFINITY;
        for (int d = 0; d <= 2 * r; ++d) m = fmaxf(m, tile[ty * R + tx + d]);
        rmax[i] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kNmsT * kNmsT; i += 256) {
        const int ty = i / kNmsT, tx = i - ty * kNmsT, y = y0 + ty, x = x0 + tx;
        if (y >= a.H || x >= a.W) continue;
        float m = -INFINITY;
        for (int d = 0; d <= 2 * r; ++d) m = fmaxf(m, rmax[(ty + d) * kNmsT + tx]);
        const int64_t g = base + (int64_t)y * a.W + x;
        if (a.mode == 0) {
            a.mask[g] = (a.s[g] == m) ? 1.f : 0.f;
        } else if (a.mode == 1) {
            const bool sp = m > 0.f;
            a.supp[g] = sp ? 1.f : 0.f;
            a.ss[g] = sp ? 0.f : a.s[g];
        } else {
            const bool nm = (a.ss[g] == m) && (a.supp[g] == 0.f);
            const bool mk = (a.mask[g] != 0.f) || nm;
            a.mask[g] = mk ? 1.f : 0.f;
            if (a.last) a.out[g] = mk ? a.s[g] : 0.f;
        }
    }
}

// ---- keypoint selection: one workgroup per image -------------------------------------------------------------------------
constexpr int kSelThreads = 1024;
constexpr int kSelMaxK = 4096;
struct SelArgs {
    const float* nms;   // [B][H][W] NMS-ed scores
    int H, W, border, K, fill_random;
    float thr;
    uint32_t seed;
    int* list_idx;      // [B][H*W] scratch: candidate pixel indices in row-major order
    float* list_sc;     // [B][H*W]
    float* kpts;        // [B][K][2] (x, y)
    float* scores;      // [B][K]
    int* count;         // [B]
};

__device__ __forceinline__ int sel_block_excl_scan(int v, int* s_wave /*[17]*/, int& total) {
    // exclusive prefix sum of v over the 1024 threads (wave scan + 16 wave totals)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int off = 0, tot = 0;
    for (int w = 0; w < kSelThreads / 64; ++w) {
        const int t = s_wave[w];
        if (w < wave) off += t;
        tot += t;
    }
    total = tot;
    return off + incl - v;
}

__global__ __launch_bounds__(kSelThreads) void sp_select_kernel(SelArgs a) {
    __shared__ unsigned long long keys[kSelMaxK];
    __shared__ int s_wave[17];
    __shared__ int hist[256];
    __shared__ int s_misc[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t hw = (int64_t)a.H * a.W;
    const float* nms = a.nms + b * hw;
    int* lidx = a.list_idx + b * hw;
    float* lsc = a.list_sc + b * hw;
    // 1. ordered compaction of the candidates (score > threshold, inside the border band)
    int n = 0;
    for (int64_t base = 0; base < hw; base += kSelThreads) {
        const int64_t i = base + tid;
        float s = 0.f;
        bool f = false;
        if (i < hw) {
            s = nms[i];
            const int y = (int)(i / a.W), x = (int)(i - (int64_t)y * a.W);
            f = s > a.thr && y >= a.border && y < a.H - a.border && x >= a.border && x < a.W - a.border;
        }
        int tot;
        const int pos = sel_block_excl_scan(f ? 1 : 0, s_wave, tot);
        if (f) { lidx[n + pos] = (int)i; lsc[n + pos] = s; }
        n += tot;
    }
    __syncthreads();
    const int K = a.K;
    float* okp = a.kpts + (int64_t)b * K * 2;
    float* osc = a.scores + (int64_t)b * K;
    int kept;
    if (n <= K) {  // nothing is cut: row-major order (top_k_keypoints returns early, upstream superpoint.py)
        for (int i = tid; i < n; i += kSelThreads) {
            const int idx = lidx[i];
            okp[2 * i] = (float)(idx % a.W);
            okp[2 * i + 1] = (float)(idx / a.W);
            osc[i] = lsc[i];
        }
        kept = n;
    } else {
        // 2. radix select of the K-th largest score (scores are positive floats: their bit patterns order like uints)
        unsigned prefix = 0, pmask = 0;
        int want = K;  // rank (1-based, from the top) still searched inside the current prefix class
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            for (int i = tid; i < 256; i += kSelThreads) hist[i] = 0;
            __syncthreads();
            for (int i = tid; i < n; i += kSelThreads) {
                const unsigned u = __float_as_uint(lsc[i]);
                if ((u & pmask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int acc = 0, bin = 255;
                for (; bin >= 0; --bin) {
                    if (acc + hist[bin] >= want) break;
                    acc += hist[bin];
                }
                s_misc[0] = bin;
                s_misc[1] = want - acc;
            }
            __syncthreads();
            prefix |= (unsigned)s_misc[0] << shift;
            pmask |= 255u << shift;
            want = s_misc[1];
            __syncthreads();
        }
        const unsigned tbits = prefix;  // the K-th largest score; `want` of the entries equal to it are kept (lowest index first)
        // 3. ordered pick: everything above, then the first `want` equal ones
        int taken = 0, eq_seen = 0;
        for (int base = 0; base < n; base += kSelThreads) {
            const int i = base + tid;
            unsigned u = 0;
            if (i < n) u = __float_as_uint(lsc[i]);
            const bool gt = i < n && u > tbits, eq = i < n && u == tbits;
            int tot_eq;
            const int eq_pos = sel_block_excl_scan(eq ? 1 : 0, s_wave, tot_eq);
            const bool take = gt || (eq && eq_seen + eq_pos < want);
            int tot;
            const int pos = sel_block_excl_scan(take ? 1 : 0, s_wave, tot);
            if (take) keys[taken + pos] = ((unsigned long long)u << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)lidx[i]);
            taken += tot;
            eq_seen += tot_eq;
        }
        __syncthreads();
        // 4. bitonic sort, descending on (score, -index); pad to a power of two with zeros (sort to the end)
        int P = 1;
        while (P < K) P <<= 1;
        for (int i = K + tid; i < P; i += kSelThreads) keys[i] = 0ull;
        __syncthreads();
        for (int size = 2; size <= P; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < P / 2; t += kSelThreads) {
                    const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                    const bool desc = ((lo & size) == 0);
                    const unsigned long long x = keys[lo], y = keys[hi];
                    if ((x < y) == desc) { keys[lo] = y; keys[hi] = x; }
                }
                __syncthreads();
            }
        for (int i = tid; i < K; i += kSelThreads) {
            const unsigned long long k = keys[i];
            const int idx = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
            okp[2 * i] = (float)(idx % a.W);
            okp[2 * i + 1] = (float)(idx / a.W);
            osc[i] = __uint_as_float((unsigned)(k >> 32));
        }
        kept = K;
    }
    // 5. the remaining slots: zeros, or (fork option fill_with_random_keypoints) hashed pseudo-random pixels, score 0
    for (int i = kept + tid; i < K; i += kSelThreads) {
        float x = 0.f, y = 0.f;
        if (a.fill_random) {
            unsigned h = a.seed ^ (0x9E3779B9u * (unsigned)(b + 1)) ^ (0x85EBCA6Bu * (unsigned)(i + 1));
            h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
            const int bw = max(1, a.W - 2 * a.border), bh = max(1, a.H - 2 * a.border);
            x = (float)(a.border + (int)(h % (unsigned)bw));
            y = (float)(a.border + (int)((h / (unsigned)bw) % (unsigned)bh));
        }
        okp[2 * i] = x; okp[2 * i + 1] = y; osc[i] = 0.f;
    }
    if (tid == 0) a.count[b] = a.fill_random ? K : kept;
}

// ---- descriptor sampling: one wave per keypoint, lane = 4 channels ------------------------------------------------------
__global__ __launch_bounds__(64) void sp_sample_kernel(const float* __restrict__ dense /*[B][Hc][Wc][256]*/, const float* __restrict__ kpts,
                                                       const int* __restrict__ count, float* __restrict__ out /*[B][256][K]*/, int Hc, int Wc, int K) {
    const int k = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    float* dst = out + ((int64_t)b * 256 + 4 * lane) * K + k;
    if (k >= count[b]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[(int64_t)e * K] = 0.f;
        return;
    }
    const float kx = kpts[((int64_t)b * K + k) * 2], ky = kpts[((int64_t)b * K + k) * 2 + 1];
    // sample_descriptors (upstream superpoint.py): same operation order in fp32
    const float s = 8.f;
    float gx = (kx - s / 2 + 0.5f) / (Wc * s - s / 2 - 0.5f), gy = (ky - s / 2 + 0.5f) / (Hc * s - s / 2 - 0.5f);
    gx = gx * 2 - 1; gy = gy * 2 - 1;
    const float ix = ((gx + 1) / 2) * (Wc - 1), iy = ((gy + 1) / 2) * (Hc - 1);  // grid_sample, align_corners=True
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const float wts[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};  // nw, ne, sw, se
    sp_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int xx = x0 + (c & 1), yy = y0 + (c >> 1);
        if ((unsigned)xx >= (unsigned)Wc || (unsigned)yy >= (unsigned)Hc) continue;  // zeros padding
        const sp_f4 v = *reinterpret_cast<const sp_f4*>(dense + (((int64_t)b * Hc + yy) * Wc + xx) * 256 + 4 * lane);
        float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n2 += __shfl_xor(n2, o);
        const float inv = wts[c] / fmaxf(sqrtf(n2), 1e-12f);  // F.normalize(dense, dim=1) then the bilinear weight
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(v[e], inv, acc[e]);
    }
    float n2 = acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2] + acc[3] * acc[3];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n2 += __shfl_xor(n2, o);
    const float inv = 1.f / fmaxf(sqrtf(n2), 1e-12f);
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[(int64_t)e * K] = acc[e] * inv;
}

}  // namespace e2emv

using namespace e2emv;

// ---- weights --------------------------------------------------------------------------------------------------------------
extern "C" int e2emv_superpoint_commit(e2emv_ctx* ctx) {
    if (!ctx) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    size_t total = 0;
    for (int l = 0; l < 12; ++l) total += (size_t)kSpCout[l] * kSpCin[l] * kSpK[l] * kSpK[l] + ((kSpCout[l] + 3) & ~3);
    std::vector<float> host(total + 64, 0.f);
    size_t off = 0;
    size_t w_off[12], b_off[12];
    for (int l = 0; l < 12; ++l) {
        const std::string wn = std::string("superpoint.") + kSpNames[l] + ".weight", bn = std::string("superpoint.") + kSpNames[l] + ".bias";
        auto wi = ctx->raw.find(wn), bi = ctx->raw.find(bn);
        if (wi == ctx->raw.end() || bi == ctx->raw.end()) return set_err(ctx, E2EMV_ESTATE, "superpoint: weight %s / %s missing", wn.c_str(), bn.c_str());
        const int co = kSpCout[l], ci = kSpCin[l], k = kSpK[l];
        if (wi->second.data.size() != (size_t)co * ci * k * k || bi->second.data.size() != (size_t)co)
            return set_err(ctx, E2EMV_ESHAPE, "superpoint: %s must be [%d,%d,%d,%d]", wn.c_str(), co, ci, k, k);
        off = (off + 3) & ~size_t(3);
        w_off[l] = off;
        const float* src = wi->second.data.data();  // [co][ci][ky][kx]
        for (int o = 0; o < co; ++o)
            for (int t = 0; t < k * k; ++t)
                for (int c = 0; c < ci; ++c) host[off + ((size_t)o * k * k + t) * ci + c] = src[((size_t)o * ci + c) * k * k + t];
        off += (size_t)co * ci * k * k;
        off = (off + 3) & ~size_t(3);
        b_off[l] = off;
        std::memcpy(&host[off], bi->second.data.data(), sizeof(float) * co);
        off += co;
    }
    if (ctx->d_sparena) { (void)hipFree(ctx->d_sparena); ctx->d_sparena = nullptr; }
    E2EMV_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_sparena), sizeof(float) * host.size()));
    E2EMV_HIP(ctx, hipMemcpy(ctx->d_sparena, host.data(), sizeof(float) * host.size(), hipMemcpyHostToDevice));
    for (int l = 0; l < 12; ++l) { ctx->sp_w[l] = ctx->d_sparena + w_off[l]; ctx->sp_b[l] = ctx->d_sparena + b_off[l]; }
    ctx->sp_committed = true;
    E2EMV_NULL_STREAM_FENCE(ctx);
    return E2EMV_OK;
}

namespace {

int sp_conv3x3(e2emv_ctx* ctx, int layer, const float* in, float* out, int imgs, int H, int W, hipStream_t s, bool pool = false) {
    GemmArgs g;
    g.M = imgs * H * W; g.N = kSpCout[layer]; g.K = 9 * kSpCin[layer];
    g.A = in; g.lda = kSpCin[layer];
    g.W = ctx->sp_w[layer]; g.ldw = g.K;
    g.bias = ctx->sp_b[layer];
    g.C = out; g.ldc = g.N;
    g.relu = true;
    g.conv_h = H; g.conv_w = W; g.conv_c = kSpCin[layer];
    g.conv_pool = pool;
    prof_begin(ctx, PS_GEMM, s);
    const int rc = launch_gemm_nt(ctx, g, s);
    prof_end(ctx, s);
    return rc;
}

// Images whose height / width is not a multiple of 8 (upstream: the convolutions see the whole image, every max-pool floors):
// the encoder runs on the zero-padded multiple-of-8 grid and, after every convolution, the part of its output that lies outside
// the VALID size of that level (H, H/2, H/4 floored like the pools) is set to zero - exactly the zero padding upstream's next
// convolution would see there; behind the third pool the valid block is cropped out and everything after runs on it unpadded.
typedef float sp_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void sp_zero_outside_kernel(float* buf, int Hp, int Wp, int C4, int hv, int wv, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = i / C4;
        const int x = (int)(pix % Wp), y = (int)((pix / Wp) % Hp);
        if (y >= hv || x >= wv) reinterpret_cast<sp_f32x4*>(buf)[i] = sp_f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
__global__ __launch_bounds__(256) void sp_crop_kernel(const float* src, float* dst, int Hs, int Ws, int hv, int wv, int C4, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        const int64_t pix = i / C4;
        const int x = (int)(pix % wv), y = (int)((pix / wv) % hv);
        const int64_t b = pix / ((int64_t)wv * hv);
        reinterpret_cast<sp_f32x4*>(dst)[i] = reinterpret_cast<const sp_f32x4*>(src)[((b * Hs + y) * Ws + x) * C4 + c];
    }
}

int sp_conv1x1(e2emv_ctx* ctx, int layer, const float* in, float* out, int64_t rows, hipStream_t s) {
    GemmArgs g;
    g.M = (int)rows; g.N = kSpCout[layer]; g.K = kSpCin[layer];
    g.A = in; g.lda = g.K;
    g.W = ctx->sp_w[layer]; g.ldw = g.K;
    g.bias = ctx->sp_b[layer];
    g.C = out; g.ldc = g.N;
    prof_begin(ctx, PS_GEMM, s);
    const int rc = launch_gemm_nt(ctx, g, s);
    prof_end(ctx, s);
    return rc;
}

}  // namespace

extern "C" int e2emv_superpoint_forward(e2emv_ctx* ctx, const e2emv_superpoint_desc* d, const float* d_images, float* d_kpts, float* d_scores,
                                        float* d_desc, int32_t* d_count, float* d_score_map, void* stream) {
    if (!ctx) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    if (!d || !d_images || !d_kpts || !d_scores || !d_desc || !d_count) return set_err(ctx, E2EMV_EINVAL, "superpoint_forward: NULL argument");
    if (!ctx->sp_committed) return set_err(ctx, E2EMV_ESTATE, "superpoint_forward: weights not committed (e2emv_superpoint_commit)");
    const int B = d->batch, K = d->max_keypoints, r = d->nms_radius;
    const int Hp = d->height, Wp = d->width;  // the (padded) grid the images are stored on
    if (B < 1 || Hp < 16 || Wp < 16 || Hp % 8 || Wp % 8) return set_err(ctx, E2EMV_ESHAPE, "superpoint_forward: image %dx%d must be a multiple of 8 (>= 16)", Hp, Wp);
    const int Hv = d->valid_height > 0 ? d->valid_height : Hp, Wv = d->valid_width > 0 ? d->valid_width : Wp;  // the image itself
    if (Hv > Hp || Wv > Wp || Hp - Hv >= 8 || Wp - Wv >= 8 || Hv < 16 || Wv < 16)
        return set_err(ctx, E2EMV_ESHAPE, "superpoint_forward: valid size %dx%d must lie within 7 pixels below the padded %dx%d", Hv, Wv, Hp, Wp);
    const bool padded = Hv != Hp || Wv != Wp;
    const int H = Hv / 8 * 8, W = Wv / 8 * 8;  // size of the score map = what every stage behind the encoder works on
    if (K < 1 || K > kSelMaxK) return set_err(ctx, E2EMV_ESHAPE, "superpoint_forward: max_keypoints %d outside 1..%d", K, kSelMaxK);
    if (r < 0 || r > 16 || d->remove_borders < 0) return set_err(ctx, E2EMV_EINVAL, "superpoint_forward: nms_radius %d / remove_borders %d", r, d->remove_borders);
    if ((int64_t)B * Hp * Wp >= (int64_t(1) << 31)) return set_err(ctx, E2EMV_ESHAPE, "superpoint_forward: batch too large for one call (B*H*W < 2^31)");
    hipStream_t s = (hipStream_t)stream;
    const int64_t npix_p = (int64_t)B * Hp * Wp;
    const int64_t npix = (int64_t)B * H * W, cells = npix / 64;
    const int Hc = H / 8, Wc = W / 8;
    auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
    // workspace: two full-resolution 64-channel buffers (ping-pong for the whole encoder), heads, 5 score-sized maps, lists
    const size_t big = al(sizeof(float) * npix_p * 64);
    const size_t sz_x = al(sizeof(float) * cells * 128), sz_p = al(sizeof(float) * cells * 256), sz_65 = al(sizeof(float) * cells * 65);
    const size_t sz_map = al(sizeof(float) * npix);
    const size_t bytes = 2 * big + sz_x + 2 * sz_p + sz_65 + 7 * sz_map;
    int rc = ws_reserve(ctx, bytes);
    if (rc) return rc;
    char* p = ctx->d_ws;
    auto take = [&](size_t b) { char* q = p; p += b; return q; };
    float* A = (float*)take(big); float* Bf = (float*)take(big);
    float* X = (float*)take(sz_x); float* P1 = (float*)take(sz_p); float* P2 = (float*)take(sz_p); float* S65 = (float*)take(sz_65);
    float* score = (float*)take(sz_map); float* mask = (float*)take(sz_map); float* supp = (float*)take(sz_map); float* ss = (float*)take(sz_map);
    float* nms = d_score_map ? d_score_map : (float*)take(sz_map);
    if (d_score_map) take(sz_map);
    int* lidx = (int*)take(sz_map); float* lsc = (float*)take(sz_map);

    // ---- encoder ----
    prof_begin(ctx, PS_INGEST, s);
    hipLaunchKernelGGL(sp_conv1a_kernel, dim3((unsigned)((npix_p + 15) / 16)), dim3(256), 0, s, d_images, ctx->sp_w[0], ctx->sp_b[0], A, Hp, Wp, npix_p);
    E2EMV_CHECK_LAUNCH(ctx, "sp_conv1a_kernel");
    prof_end(ctx, s);
    auto zero_outside = [&](float* buf, int lvl, int C) {  // zero what lies outside the valid size of encoder level `lvl` (1, 2, 4 = stride)
        if (!padded) return;
        const int hp = Hp / lvl, wp = Wp / lvl, hv = Hv / lvl, wv = Wv / lvl;
        if (hv == hp && wv == wp) return;
        const int64_t total = (int64_t)B * hp * wp * (C / 4);
        hipLaunchKernelGGL(sp_zero_outside_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0, s, buf, hp, wp, C / 4, hv, wv, total);
    };
    // conv_b of blocks 1-3 write their 2x2 max-pooled output directly (fused epilogue)
    zero_outside(A, 1, 64);
    if ((rc = sp_conv3x3(ctx, 1, A, Bf, B, Hp, Wp, s, true))) return rc;
    zero_outside(Bf, 2, 64);
    if ((rc = sp_conv3x3(ctx, 2, Bf, A, B, Hp / 2, Wp / 2, s))) return rc;
    zero_outside(A, 2, 64);
    if ((rc = sp_conv3x3(ctx, 3, A, Bf, B, Hp / 2, Wp / 2, s, true))) return rc;
    zero_outside(Bf, 4, 64);
    if ((rc = sp_conv3x3(ctx, 4, Bf, A, B, Hp / 4, Wp / 4, s))) return rc;
    zero_outside(A, 4, 128);
    if ((rc = sp_conv3x3(ctx, 5, A, Bf, B, Hp / 4, Wp / 4, s, true))) return rc;
    if (padded && (Hc != Hp / 8 || Wc != Wp / 8)) {  // the valid block of the third pool's output, unpadded from here on
        const int64_t total = (int64_t)B * Hc * Wc * 32;
        hipLaunchKernelGGL(sp_crop_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 8192)), dim3(256), 0, s, (const float*)Bf, A, Hp / 8, Wp / 8, Hc, Wc, 32, total);
        E2EMV_CHECK_LAUNCH(ctx, "sp_crop_kernel");
        std::swap(A, Bf);
    }
    if ((rc = sp_conv3x3(ctx, 6, Bf, A, B, Hc, Wc, s))) return rc;
    if ((rc = sp_conv3x3(ctx, 7, A, X, B, Hc, Wc, s))) return rc;
    // ---- detector head ----
    if ((rc = sp_conv3x3(ctx, 8, X, P1, B, Hc, Wc, s))) return rc;
    if ((rc = sp_conv1x1(ctx, 9, P1, S65, cells, s))) return rc;
    prof_begin(ctx, PS_MATCH, s);
    hipLaunchKernelGGL(sp_softmax_d2s_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, S65, score, Hc, Wc, cells);
    E2EMV_CHECK_LAUNCH(ctx, "sp_softmax_d2s_kernel");
    {
        NmsArgs n{};
        n.s = score; n.mask = mask; n.supp = supp; n.ss = ss; n.out = nms; n.H = H; n.W = W; n.r = r;
        const dim3 grid((W + kNmsT - 1) / kNmsT, (H + kNmsT - 1) / kNmsT, B);
        const int R = kNmsT + 2 * r;
        const size_t lds = sizeof(float) * (R * R + R * kNmsT);
        auto run = [&](const float* in, int mode, int last) {
            n.in = in; n.mode = mode; n.last = last;
            hipLaunchKernelGGL(sp_nms_kernel, grid, dim3(256), lds, s, n);
        };
        run(score, 0, 0);
        for (int it = 0; it < 2; ++it) {
            run(mask, 1, 0);
            run(ss, 2, it == 1);
        }
        E2EMV_CHECK_LAUNCH(ctx, "sp_nms_kernel");
    }
    {
        SelArgs a{};
        a.nms = nms; a.H = H; a.W = W; a.border = d->remove_borders; a.K = K; a.fill_random = d->fill_random ? 1 : 0;
        a.thr = d->keypoint_threshold; a.seed = d->seed;
        a.list_idx = lidx; a.list_sc = lsc; a.kpts = d_kpts; a.scores = d_scores; a.count = d_count;
        hipLaunchKernelGGL(sp_select_kernel, dim3(B), dim3(kSelThreads), 0, s, a);
        E2EMV_CHECK_LAUNCH(ctx, "sp_select_kernel");
    }
    prof_end(ctx, s);
    // ---- descriptor head ----
    if ((rc = sp_conv3x3(ctx, 10, X, P1, B, Hc, Wc, s))) return rc;
    if ((rc = sp_conv1x1(ctx, 11, P1, P2, cells, s))) return rc;
    prof_begin(ctx, PS_MISC, s);
    hipLaunchKernelGGL(sp_sample_kernel, dim3(K, B), dim3(64), 0, s, P2, d_kpts, d_count, d_desc, Hc, Wc, K);
    E2EMV_CHECK_LAUNCH(ctx, "sp_sample_kernel");
    prof_end(ctx, s);
    return E2EMV_OK;
}
