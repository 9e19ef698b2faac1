// Multi-head attention of the f16x2 arithmetic mode on PLANE operands (p2.h):  out = softmax(q k^T / sqrt(64)) v
//
// The arithmetic and the transposed flash structure are those of attention_h2f_kernel (attention3.hip: S^T = mfma(K, Q),
// O^T += mfma(V^T, P^T), three fp16 products per block, lazy running maximum).  What changed is where the operands come
// from: the q|k|v projection (gemm_p2.hip, P2_OUT_QKV) already wrote
//   q | k   as plain planes [row][2D] (q pre-scaled by 2^6 log2(e)/8), one 256-byte piece per (row, head), and
//   V^T     as plain planes [image][head][64 dims][n_rows keys] (v pre-scaled by 2^4, keys of a group of 16 stored in
//           the order of the transposed-score registers),
// so this kernel has no split, no transposing stores and no ds_write in its key loop: a 64-key tile of K (64 keys x
// 256 B) and of V^T (64 dims x 256 B) goes from global memory straight into LDS with 32 buffer_load_dwordx4 ... lds,
// double-buffered, one barrier per tile; Q fragments are loaded once, straight into registers.  LDS image: row r of a
// tile keeps its 16 chunks of 16 bytes at position c ^ (r & 15) (swizzle applied on the source address), which makes
// every ds_read_b128 of a fragment conflict-free.  The output is written as scaled planes, the operand format of MLP0.
#include <algorithm>

#include "attention_p2.h"

namespace e2emv {

typedef __attribute__((ext_vector_type(16))) float ap_f32x16;

constexpr int AP_KV = 64, AP_HD = 64;
constexpr int AP_TILEB = 64 * 256;       // one operand tile: 64 rows x 256 B
constexpr int AP_BUFB = 2 * AP_TILEB;    // K | V^T
constexpr float AP_SINV = 1.f / P2_QS, AP_PLOG = 10.f, AP_LAZY = 5.f;


template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void attention_p2_kernel(AttnP2Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem_ap[];
    constexpr int QT = 32 * NW;
    constexpr int LPW = 16 / NW;  // LDS-direct loads per wave, tile and operand

    const int lin = blockIdx.x;
    const int xcd = lin & 7, idx = lin >> 3;
    const int g = xcd * p.gper + idx / p.nq;
    if (g >= p.groups) return;
    const int qt = idx % p.nq;
    const int img = g / p.H, head = g % p.H;
    const int b = img / p.T, t = img % p.T;
    if (qt * QT >= p.nv[t]) return;  // shorter image of a ragged tuple: no queries in this tile

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned row_b = 8u * (unsigned)p.D;  // bytes per q|k row: 2D columns x 4 B

    // ---- Q fragments (B operand): lane (query l31, lh) holds d = 16 s + 8 lh .. + 7 of both planes
    const int q_row = qt * QT + wave * 32 + l31;
    const bool q_ok = q_row < p.n_rows;  // (n_rows is a multiple of 128: the last 256-query tile may be half empty)
    p2_u32x4 Qf[2][4];
    {
        const char* qp = reinterpret_cast<const char*>(p.qk) + ((int64_t)img * p.n_rows + (q_ok ? q_row : p.n_rows - 1)) * row_b + head * 256;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
                Qf[pl][s] = *reinterpret_cast<const p2_u32x4*>(qp + ((s >> 1) * 8 + pl * 4 + 2 * (s & 1) + lh) * 16);
    }

    ap_f32x16 O0, O1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { O0[r] = 0.f; O1[r] = 0.f; }
    float m_run = -1e30f, l_run = 0.f;   // m_run: running maximum in TRUE base-2 logits (the score units change with k's exponent)
    // tile exponents: q's (this wave's 32 queries lie in one 64-row block) goes into the logit scale together with the key
    // tile's; the O accumulator lives at the exponent of the CURRENT V tile and is rescaled (exactly) when that changes
    const int q_blk = __builtin_amdgcn_readfirstlane(((int64_t)img * p.n_rows + min(qt * QT + wave * 32, p.n_rows - 1)) >> 6);
    const int e_q = p.EQK ? p.EQK[q_blk * 8 + head] : 0;
    int e_o = 0;
    bool o_started = false;

    const int n_src = p.cross ? p.T - 1 : 1;
    auto src_t = [&](int si) { return !p.cross ? t : (si < t ? si : si + 1); };
    int n_tiles = 0;
    for (int si = 0; si < n_src; ++si) n_tiles += (p.nv[src_t(si)] + AP_KV - 1) / AP_KV;
    struct TilePos { int si, kt; };
    auto advance_pos = [&](TilePos& tp) {
        if (++tp.kt * AP_KV >= p.nv[src_t(tp.si)] && tp.si + 1 < n_src) { ++tp.si; tp.kt = 0; }
    };

    // ---- loader: one load = 4 rows x 256 B; lane -> (row lane >> 4, LDS position lane & 15), source chunk = position ^ (row & 15)
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.qk), 0, (int)p.qk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.vt), 0, (int)p.vt_bytes, 0x00020000);
    unsigned k_vo[LPW], v_vo[LPW];
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
        const int row = 4 * (wave * LPW + i) + (lane >> 4);
        const unsigned c = (unsigned)((lane & 15) ^ (row & 15));
        k_vo[i] = (unsigned)row * row_b + 4u * (unsigned)p.D + (unsigned)head * 256u + c * 16u;
        v_vo[i] = (unsigned)row * (unsigned)p.n_rows * 4u + c * 16u;
    }
    auto issue = [&](int buf, int tp_si, int tp_kt) {
        const int tt = src_t(tp_si);
        const unsigned k_so = (unsigned)((b * p.T + tt) * p.n_rows + tp_kt * AP_KV) * row_b;
        const unsigned v_so = (unsigned)(((b * p.T + tt) * p.H + head) * AP_HD) * (unsigned)p.n_rows * 4u + (unsigned)tp_kt * 256u;
        char* dst = smem_ap + buf * AP_BUFB + wave * LPW * 1024;
#pragma unroll
        for (int i = 0; i < LPW; ++i)
            p2_glds16(rsK, dst + i * 1024, k_vo[i], k_so);
#pragma unroll
        for (int i = 0; i < LPW; ++i)
            p2_glds16(rsV, dst + AP_TILEB + i * 1024, v_vo[i], v_so);
    };

    constexpr int PA[3] = {1, 0, 0};  // plane of the A operand (K or V^T), smallest terms first
    constexpr int PB[3] = {0, 1, 0};  // plane of the B operand (Q or P)
    const int kz = l31 & 15;          // swizzle of this lane's fragment rows (keys l31 / l31 + 32, dims l31 / l31 + 32)

    // tile exponents of a key tile (k's, v's): fetched by lane 0 right behind the tile's loads, i.e. a whole tile before
    // they are needed, and broadcast with readfirstlane - no scalar-memory latency at the head of a tile
    auto fetch_e = [&](int si, int kt, int& ek, int& evv) {
        ek = 0; evv = 0;
        if (p.EQK && lane == 0) {
            const int blk = ((b * p.T + src_t(si)) * p.n_rows + kt * AP_KV) >> 6;
            ek = p.EQK[blk * 8 + 4 + head];
            evv = p.EVt ? p.EVt[blk * 4 + head] : 0;
        }
    };
    TilePos cur{0, 0}, nxt{0, 0};
    issue(0, nxt.si, nxt.kt);
    int ek_n = 0, ev_n = 0;
    fetch_e(nxt.si, nxt.kt, ek_n, ev_n);
    int buf = 0;
    for (int tile = 0; tile < n_tiles; ++tile) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's pieces have landed (issued a whole tile ago)
        __syncthreads();                                   // ... everybody's; and everybody is done with the other buffer
        cur = nxt;
        const int e_k = __builtin_amdgcn_readfirstlane(ek_n), e_v = __builtin_amdgcn_readfirstlane(ev_n);
        if (tile + 1 < n_tiles) {
            advance_pos(nxt);
            issue(buf ^ 1, nxt.si, nxt.kt);
            fetch_e(nxt.si, nxt.kt, ek_n, ev_n);
        }
        const char* Kt = smem_ap + buf * AP_BUFB;
        const char* Vt = Kt + AP_TILEB;
        const int valid_in_tile = p.nv[src_t(cur.si)] - cur.kt * AP_KV;
        const float sinv = AP_SINV * p2_exp2i(e_q + e_k);
        if (p.EVt) {
            if (o_started && e_v != e_o) {
                const float f = e_o - e_v < -126 ? 0.f : p2_exp2i(e_o - e_v);
#pragma unroll
                for (int r = 0; r < 16; ++r) { O0[r] *= f; O1[r] *= f; }
            }
            e_o = e_v;
            o_started = true;
        }
        // S^T of a 32-key block: 12 MFMAs on one accumulator (the first takes the zero C operand)
        auto qk = [&](int sub) {
            ap_f32x16 S;
            __builtin_amdgcn_s_setprio(3);  // the matrix-core bursts of a wave go ahead of the other waves' softmax arithmetic (measured: -1.5 %)
            const char* kp = Kt + (sub * 32 + l31) * 256;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                p2_f16x8 kf[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    kf[pl] = *reinterpret_cast<const p2_f16x8*>(kp + ((((s >> 1) * 8 + pl * 4 + 2 * (s & 1) + lh) ^ kz) << 4));
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (s == 0 && q == 0) {
                        const ap_f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        S = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[PA[q]], __builtin_bit_cast(p2_f16x8, Qf[PB[q]][s]), zero, 0, 0, 0);
                    } else {
                        S = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[PA[q]], __builtin_bit_cast(p2_f16x8, Qf[PB[q]][s]), S, 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            return S;
        };
        // online softmax of the block + its P V products
        auto softmax_pv = [&](int sub, ap_f32x16 S) {
            if (valid_in_tile < sub * 32 + 32) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= valid_in_tile) S[r] = -INFINITY;
                }
            }
            float mx = S[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
            {   // the other half of the key block sits in lane ^ 32: v_permlane32_swap instead of an LDS-crossbar shuffle (-1.5 %).
                // Inline asm with two distinct registers: after the swap a = [lo, lo], b = [hi, hi] (hipcc 7.2 folds the two
                // results of the builtin into one when both operands are the same value and loses the upper half).
                float a = mx, b = mx;
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
                mx = fmaxf(a, b);
            }
            // S (and m) are in units of 1 / P2_QS of a base-2 logit; P carries the factor 2^AP_PLOG (cancels in O / l).  Lazy
            // running maximum as in attention_h2f_kernel: m_run moves (and O, l are rescaled) only when a row's new maximum
            // exceeds it by more than 2^AP_LAZY; P then stays below 2^(AP_PLOG + AP_LAZY) = 32768, inside fp16's range
            mx *= sinv;
            float m_new = m_run, alpha = 1.f;
            const bool grow = mx - m_run > AP_LAZY;
            if (__builtin_amdgcn_ballot_w64(grow) != 0) {  // wave-uniform
                m_new = fmaxf(m_run, mx);
                alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
                for (int r = 0; r < 16; ++r) { O0[r] *= alpha; O1[r] *= alpha; }
            }
            const float e0 = AP_PLOG - m_new;
            float ps = 0.f;
            p2_u32x4 Pf[2][2];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], sinv, e0));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r + 1], sinv, e0));
                ps += p0 + p1;
                const P2Pair pr = p2_split_plain(p0, p1);
                Pf[0][r >> 3][(r & 7) >> 1] = pr.hi; Pf[1][r >> 3][(r & 7) >> 1] = pr.lo;
            }
            l_run = l_run * alpha + ps;
            m_run = m_new;
            const char* vp = Vt + l31 * 256;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int uu = 2 * sub + u;  // 16-key step inside the tile
                p2_f16x8 v0[2], v1[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const int off = ((((uu >> 1) * 8 + pl * 4 + 2 * (uu & 1) + lh) ^ kz) << 4);
                    v0[pl] = *reinterpret_cast<const p2_f16x8*>(vp + off);
                    v1[pl] = *reinterpret_cast<const p2_f16x8*>(vp + 32 * 256 + off);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    O0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0[PA[q]], __builtin_bit_cast(p2_f16x8, Pf[PB[q]][u]), O0, 0, 0, 0);
                    O1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1[PA[q]], __builtin_bit_cast(p2_f16x8, Pf[PB[q]][u]), O1, 0, 0, 0);
                }
            }
        };
        // (Measured and dropped: both score blocks first, so that the MFMAs of the second run under the softmax arithmetic of
        // the first - 16 more registers, 232 us against 224 us at 32 pairs x 1024 keypoints: the two workgroups of a CU
        // already fill each other's gaps.)
        if (NW == 8 && valid_in_tile > 32) {
            // both 32-key score blocks with their MFMAs INTERLEAVED: two independent accumulator chains instead of one chain of 12
            // dependent MFMAs per block (-1.6 %; one block after the other - same registers - had measured +2.5 %)
            ap_f32x16 S0, S1;
            __builtin_amdgcn_s_setprio(3);
            const char* kp0 = Kt + l31 * 256;
            const char* kp1 = Kt + (32 + l31) * 256;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                p2_f16x8 k0[2], k1[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const int off = ((((s >> 1) * 8 + pl * 4 + 2 * (s & 1) + lh) ^ kz) << 4);
                    k0[pl] = *reinterpret_cast<const p2_f16x8*>(kp0 + off);
                    k1[pl] = *reinterpret_cast<const p2_f16x8*>(kp1 + off);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const p2_f16x8 qf = __builtin_bit_cast(p2_f16x8, Qf[PB[q]][s]);
                    if (s == 0 && q == 0) {
                        const ap_f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        S0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0[PA[q]], qf, zero, 0, 0, 0);
                        S1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1[PA[q]], qf, zero, 0, 0, 0);
                    } else {
                        S0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0[PA[q]], qf, S0, 0, 0, 0);
                        S1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1[PA[q]], qf, S1, 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            softmax_pv(0, S0);
            softmax_pv(1, S1);
        } else {
            softmax_pv(0, qk(0));
            if (valid_in_tile > 32) softmax_pv(1, qk(1));
        }
        buf ^= 1;
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / (l_tot * P2_VS);
    if (!q_ok) return;
    // |O / l| <= max |v'| < 2^15 / P2_VS: the output planes inherit the exponent of the last V tile (every wave of the
    // workgroup - and of the other query tiles of this image and head - walks the same tiles: the same value)
    if (p.EO && lane == 0) p.EO[(((int64_t)img * p.n_rows + q_row) >> 6) * 4 + head] = e_o;
    // scaled planes of the output row: lane (query, lh) owns dims (r & 3) + 8 (r >> 2) + 4 lh of each 32-dim block
    uint16_t* op = p.out + p2_index((int64_t)img * p.n_rows + q_row, head * AP_HD + 4 * lh, p.D);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const P2Pair a0 = p2_split_scaled(O0[gq * 4] * inv, O0[gq * 4 + 1] * inv), a1 = p2_split_scaled(O0[gq * 4 + 2] * inv, O0[gq * 4 + 3] * inv);
        const P2Pair c0 = p2_split_scaled(O1[gq * 4] * inv, O1[gq * 4 + 1] * inv), c1 = p2_split_scaled(O1[gq * 4 + 2] * inv, O1[gq * 4 + 3] * inv);
        *reinterpret_cast<p2_u32x2*>(op + 8 * gq) = p2_u32x2{a0.hi, a1.hi};
        *reinterpret_cast<p2_u32x2*>(op + 8 * gq + 32) = p2_u32x2{a0.lo, a1.lo};
        *reinterpret_cast<p2_u32x2*>(op + 64 + 8 * gq) = p2_u32x2{c0.hi, c1.hi};
        *reinterpret_cast<p2_u32x2*>(op + 64 + 8 * gq + 32) = p2_u32x2{c0.lo, c1.lo};
    }
}

int launch_attention_p2(e2emv_ctx* ctx, int B, int T, int n_rows, const int* nv, int D, int H, const uint16_t* qk,
                        const uint16_t* vt, int cross, uint16_t* outp, hipStream_t s, const int* EQK, const int* EVt, int* EO) {
    int n_valid = 0;
    for (int t = 0; t < T; ++t) {
        if (nv[t] <= 0 || nv[t] > n_rows) return set_err(ctx, E2EMV_ESHAPE, "attention_p2: image %d has %d keypoints (n_rows %d)", t, nv[t], n_rows);
        n_valid = std::max(n_valid, nv[t]);
    }
    if (D != H * AP_HD || D != 256) return set_err(ctx, E2EMV_ESHAPE, "attention_p2: needs 4 heads of 64 (D=%d H=%d)", D, H);
    if (n_rows % 128 || n_valid <= 0) return set_err(ctx, E2EMV_ESHAPE, "attention_p2: n_rows=%d must be a multiple of 128", n_rows);
    if (cross && T < 2) return set_err(ctx, E2EMV_ESHAPE, "attention_p2: cross layer needs T >= 2");
    if (!qk || !vt || !outp || (uintptr_t)qk % 16 || (uintptr_t)vt % 16 || (uintptr_t)outp % 16) return set_err(ctx, E2EMV_EINVAL, "attention_p2: null / unaligned buffer");
    const int64_t rows = (int64_t)B * T * n_rows;
    if (rows * 8 * D >= ((int64_t)1 << 31)) return set_err(ctx, E2EMV_ESHAPE, "attention_p2: q|k planes larger than 2 GB (32-bit byte offsets)");
    AttnP2Params p{};
    p.qk = qk; p.vt = vt; p.out = outp;
    p.EQK = EQK; p.EVt = EVt; p.EO = EO;
    p.qk_bytes = (unsigned)(rows * 8 * D); p.vt_bytes = (unsigned)(rows * 4 * D);
    p.B = B; p.T = T; p.n_rows = n_rows; p.D = D; p.H = H;
    for (int t = 0; t < E2EMV_MAX_TUPLE; ++t) p.nv[t] = t < T ? nv[t] : 0;
    p.cross = cross;
    p.groups = B * T * H;
    p.gper = (p.groups + 7) / 8;
    static int nw_knob = -1;
    if (nw_knob < 0) nw_knob = dbg_knob("E2EMV_AP2_NW", 0);
    if (nw_knob == 4 || nw_knob == 8 || nw_knob == 1) ctx->attn_p2_nw = nw_knob;
    // above 256 keys: one wave per SIMD with the overlap of matrix and vector work written into the wave's instruction
    // stream (attention_p2w.hip, attn_p2_nw == 1 forces it); below, and for A/B runs (4 / 8), the two-waves-per-SIMD kernel here
    if (ctx->attn_p2_nw == 1 || (ctx->attn_p2_nw == 0 && ctx->attn_wide && n_valid > 256)) return launch_attention_p2w(ctx, p, n_valid, s);
    const int nw = ctx->attn_p2_nw == 4 || ctx->attn_p2_nw == 8 ? ctx->attn_p2_nw : (n_valid > 256 ? 8 : 4);  // measured: 222 / 224 us at 1024 keys, 202 / 209 at 2048
    const size_t lds = 2 * AP_BUFB;
    const void* fn = nw == 8 ? reinterpret_cast<const void*>(attention_p2_kernel<8>) : reinterpret_cast<const void*>(attention_p2_kernel<4>);
    p.nq = (n_valid + 32 * nw - 1) / (32 * nw);
    if (int rc = ensure_dynamic_lds(ctx, fn, lds)) return rc;
    void* args[] = {&p};
    E2EMV_HIP(ctx, hipLaunchKernel(fn, dim3(8 * p.gper * p.nq), dim3(64 * nw), args, lds, s));
    E2EMV_CHECK_LAUNCH(ctx, "attention_p2_kernel");
    return E2EMV_OK;
}

}  // namespace e2emv
