// Plane-in / plane-out "NT" GEMM of the f16x2 arithmetic mode:  C[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]) (+ R[m][n])
//
// Same arithmetic as gemm_h2.hip (x = x_hi + 2^-11 x_lo', weights 2^s W = w_hi + w_lo, three fp16 MFMA products per
// block into one fp32 accumulator), but the activation arrives ALREADY as the two planes (p2.h): whoever produced it
// split it once, in its epilogue.  What that buys in the K loop, per 256 x 256 x 32 step and workgroup:
//   * no fp32 -> plane split (40 VALU per thread) and no ds_write at all: the 64 KB of operand planes go from global
//     memory straight into LDS, 64 buffer_load_dwordx4 ... lds per step (8 per wave), each a full 128-byte line per row;
//   * LDS double-buffered (2 x 64 KB), ONE barrier per K step: the loads of step g + 1 are issued right after the barrier
//     that opens step g and have that whole step (48 MFMAs per wave) to land;
//   * bank-conflict-free fragment reads without padding (XOR swizzle applied on the source address, p2.h).
// The epilogue goes through per-wave LDS slabs (32 rows x 36 floats) so that bias / ReLU / residual / the plane split
// work on 8 consecutive columns of one row per lane and every store covers 64-byte row segments; the residual is read
// back from its planes (22 bits) in the same layout.  Output kinds: fp32, scaled planes (the next GEMM's operand), or
// the attention operands q | k (plain planes, q pre-scaled) + V^T (transposed plain planes, keys in the order of the
// transposed-score registers) - attention_p2.hip then never touches an fp32 q|k|v matrix.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "gemm_p2_core.h"

namespace e2emv {

// DBG (instantiated only in -DE2EMV_STAMPS builds, tools/p2_stamps.py): 1 no MFMA, 2 no operand loads after the first two K
// steps, 4 no epilogue, 8 s_memtime stamps per K step, 16 loads of step g + 1 issued one per MFMA group, 32 every wave issues
// its loads before it computes (no opposite orders on a SIMD), 256 vmcnt(0) at every step (no store overlap)
template <int OUT, bool HAS_R = false, int DBG = 0>
__global__ __launch_bounds__(512, 1) void gemm_p2_kernel(GemmP2Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem_p2[];

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
    const int per_xcd = (p.total + 7) / 8;
    const int t_begin = xcd * per_xcd;
    const int t_end = min(t_begin + per_xcd, p.total);
    int tile = t_begin + slot;
    if (tile >= t_end) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int nk = p.K / P2_BK, nk1 = p.K1 / P2_BK;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.A2), 0, (int)p.a2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.W), 0, (int)p.w_bytes, 0x00020000);

    // ---- loader: wave w moves rows 32 w .. 32 w + 31 of the activation tile and of the weight tile, 4 + 4 LDS-direct loads
    // per K step; one load = 8 rows x 128 B, lane -> (row lane >> 3, LDS position lane & 7), source chunk = position ^ swizzle
    const int ld_r = lane >> 3, ld_p = lane & 7;
    unsigned a_vo[4], a2_vo[4], w_vo[4];
    auto setup = [&](int t) {
        const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 32 * wave + 8 * i + ld_r;
            const unsigned c = (unsigned)(ld_p ^ ((row >> 1) & 7));
            const unsigned gm = (unsigned)min(tm * P2_BM + row, p.M - 1);
            a_vo[i] = gm * p.lda_b + c * 16u;
            a2_vo[i] = gm * p.lda2_b + c * 16u;
            const unsigned gn = (unsigned)min(tn * P2_BN + row, p.N - 1);
            w_vo[i] = gn * p.ldw_b + c * 16u;
        }
    };
    // piece i of a K step: 0..3 activation rows, 4..7 weight rows
    auto issue_piece = [&](int buf, int kt, int i, unsigned dep) {
        char* dst = smem_p2 + buf * P2_BUFB + 32 * wave * P2_ROWB;
        if (i < 4) {
            if (kt < nk1) p2_glds16(rsA, dst + i * 1024, a_vo[i] + dep, (unsigned)kt * 128u);
            else p2_glds16(rsA2, dst + i * 1024, a2_vo[i] + dep, (unsigned)(kt - nk1) * 128u);
        } else {
            p2_glds16(rsW, dst + P2_TILEB + (i - 4) * 1024, w_vo[i - 4] + dep, (unsigned)kt * 128u);
        }
    };
    auto issue = [&](int buf, int kt, unsigned dep) {
        char* dst = smem_p2 + buf * P2_BUFB + 32 * wave * P2_ROWB;
        if (kt < nk1) {
            const unsigned so = (unsigned)kt * 128u;
#pragma unroll
            for (int i = 0; i < 4; ++i) p2_glds16(rsA, dst + i * 1024, a_vo[i] + dep, so);
        } else {
            const unsigned so = (unsigned)(kt - nk1) * 128u;
#pragma unroll
            for (int i = 0; i < 4; ++i) p2_glds16(rsA2, dst + i * 1024, a2_vo[i] + dep, so);
        }
        const unsigned sw = (unsigned)kt * 128u;
#pragma unroll
        for (int i = 0; i < 4; ++i) p2_glds16(rsW, dst + P2_TILEB + i * 1024, w_vo[i] + dep, sw);
    };

    // ---- fragments: lane (row l31, k half lh); chunk index c = 4 plane + 2 ks + lh, stored at position c ^ ((l31 >> 1) & 7)
    const int swz = (l31 >> 1) & 7;
    p2_f32x16 acc[4][2];
    // (`late`, DBG & 16 in measurement builds only: the loads of the next step go out one per MFMA group, each behind a
    // scheduling-only dependency on that group's accumulator - an empty asm, no instruction)
    auto compute = [&](int buf, auto FIRST, bool late, int ld_buf, int ld_k) {
        constexpr bool first_step = decltype(FIRST)::value;
        const char* xs = smem_p2 + buf * P2_BUFB + (wr * 64 + l31) * P2_ROWB;
        const char* ws = smem_p2 + buf * P2_BUFB + P2_TILEB + (wc * 128 + l31) * P2_ROWB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            p2_f16x8 x[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    x[t][pl] = *reinterpret_cast<const p2_f16x8*>(xs + t * 32 * P2_ROWB + (((4 * pl + 2 * ks + lh) ^ swz) << 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p2_f16x8 w[3];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    w[pl] = *reinterpret_cast<const p2_f16x8*>(ws + j * 32 * P2_ROWB + (((4 * pl + 2 * ks + lh) ^ swz) << 4));
                w[2] = w[0] * (_Float16)(1.f / 2048.f);  // 2^-11 w_hi (exact wherever it matters: gemm_h2.hip)
                constexpr int PW[3] = {1, 2, 0}, PX[3] = {0, 1, 0};  // x_hi w_lo, x_lo' (2^-11 w_hi), x_hi w_hi: smallest first
                if (DBG & 1) {  // operand pipeline only
                    asm volatile("" :: "v"(x[0][0]), "v"(x[0][1]), "v"(x[1][0]), "v"(x[1][1]), "v"(w[0]), "v"(w[1]), "v"(w[2]));
                    if ((DBG & 16) && late) issue_piece(ld_buf, ld_k, 4 * ks + j, 0u);
                    continue;
                }
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (first_step && ks == 0 && q == 0) {
                            const p2_f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[PW[q]], x[i][PX[q]], zero, 0, 0, 0);
                        } else {
                            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[PW[q]], x[i][PX[q]], acc[j][i], 0, 0, 0);
                        }
                    }
                if ((DBG & 16) && late) {
                    unsigned dep = 0;
                    asm("" : "+v"(dep) : "v"(acc[j][0]));
                    issue_piece(ld_buf, ld_k, 4 * ks + j, dep);
                }
            }
        }
    };

    // ---- the K step of the product kernel: gp_kstep (gemm_p2_core.h), software-pipelined inside the wave; `compute` above is kept
    // for the measurement build's ablations

    // ---- epilogue: gp_epilogue (gemm_p2_core.h)

    // ---- pipeline.  Step g = (tile, kt) in execution order lives in LDS buffer g & 1.  A step opens with "my loads of this
    // step have landed" + ONE barrier (everybody's have; everybody is done reading the other buffer); the loads of step
    // g + 1 then go into the other buffer.  The two waves that share a SIMD (w and w + 4) take OPPOSITE orders: waves 4-7
    // issue their 8 loads first and compute after, waves 0-3 compute first and issue after (s_memtime stamps: with both
    // issuing first the matrix pipe idled ~700-1000 cycles per step).
    // Across an epilogue the load position runs TWO steps ahead: the loads of step L + 1 went out during the tile's last
    // step L as always, those of step L + 2 go out right after it (its buffer is free: the slabs live behind the tile
    // buffers) - both BEFORE the 32 stores of the epilogue.  vmcnt retires in issue order, so step L + 1 waits with
    // vmcnt(40) (8 loads of L + 2 and 32 stores may stay in flight) and step L + 2 with vmcnt(32): the store burst of the
    // epilogue - every CU of the chip reaches it at the same time - drains under two K steps of the next tile instead of
    // in front of them.  Barriers are raw s_barrier: __syncthreads() would add vmcnt(0) while LDS-direct loads are in flight.
    int ld_tile = tile, ld_kt = 0;
    bool ld_valid = true;
    auto advance = [&]() {
        if (ld_kt + 1 < nk) { ++ld_kt; return; }
        if (ld_tile + slots < t_end) {
            ld_tile += slots;
            ld_kt = 0;
            setup(ld_tile);
        } else {
            ld_valid = false;
        }
    };
    setup(tile);
    issue(0, 0, 0u);
    advance();
    const bool issue_first = (DBG & 32) ? true : wave >= 4;
    const bool overlap = nk >= 3 && !(DBG & (256 | 64 | 4 | 2));
    // tile exponents of the A operand (p2.h): the accumulators live at the exponent of the CURRENT K block; when it changes
    // they are rescaled (exact: a power of two) - a wave-uniform branch that an ordinary network never takes
    // The exponents of a tile's K blocks are fetched ONCE, by one vector load (lane i: block i of this wave's 64 rows; the
    // next tile's before the epilogue of this one), and read with v_readlane in the loop: no memory operation there.
    // Lanes 32, 33 / 34, 35 of the same load bring the exponents / max |x| of the two residual blocks this wave adds in the
    // tile's epilogue (a dependent load there would cost an L2 / HBM round trip per tile with the matrix pipe idle).
    const bool has_e = p.EA != nullptr || (HAS_R && p.ER != nullptr);
    int e_run = 0, cur_kt = 0;
    int ev = 0, ev_next = 0;
    auto fetch_e = [&](int t) {
        int v = 0;
        if (has_e) {
            const int tm_ = t / p.tiles_n, tn_ = t - tm_ * p.tiles_n;
            const int erow = tm_ * 4 + wr;
            const int nb1 = nk1 >> 1, nb = (nk + 1) >> 1;
            if (erow * 64 < p.M) {
                if (lane < nb) {
                    if (p.EA) v = lane < nb1 ? p.EA[erow * p.eld_a + lane] : (p.EA2 ? p.EA2[erow * p.eld_a2 + lane - nb1] : 0);
                } else if (HAS_R && p.ER && lane >= 32 && lane < 36) {
                    const int cb = tn_ * 4 + wc * 2 + (lane & 1);
                    if (cb < p.eld_r) {
                        if (lane < 34) v = p.ER[erow * p.eld_r + cb];
                        else if (p.AR) v = __builtin_bit_cast(int, p.AR[erow * p.eld_r + cb]);
                    }
                }
            }
        }
        return v;
    };
    ev = fetch_e(tile);
    int since = 8;       // K steps since the last epilogue
    bool ahead = false;  // the loads of the step after next were issued before that epilogue
    int buf = 0, dbg_n = 0, dbg_steps = 0;
    if (DBG & 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    }
    auto step = [&](auto FIRST) {
        long long t0 = 0, t1 = 0, t2 = 0;
        if (DBG & 8) t0 = clock64();
        if (since == 0 && ahead) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
        else if (since <= 1 && overlap) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");  // (since == 0 without `ahead`: 8 loads, then the stores)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (DBG & 8) t1 = clock64();
        if (has_e) {
            const int e_step = p.EA ? __builtin_amdgcn_readlane(ev, cur_kt >> 1) : 0;
            if (!decltype(FIRST)::value && e_step != e_run) {
                asm volatile("s_nop 15");  // (the previous step's asm MFMAs -> the VALU below: hipcc does not pad an asm's results)
                const int d = e_run - e_step;
                const float f = d < -126 ? 0.f : p2_exp2i(d);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[j][i][r] *= f;
            }
            e_run = e_step;
            // the NEXT tile's exponents: loaded beside the operand loads of K step 2, picked up at the head of step 3, right
            // behind that step's vmcnt(0) - the wait hipcc attaches to the use is then free.  (Fetched at the tile boundary the
            // wait would sit behind the epilogue's 32 stores and drain them.)
            if (nk >= 4 && tile + slots < t_end) {
                if (cur_kt == 2) ev_next = fetch_e(tile + slots);
                if (cur_kt == 3) asm volatile("" : "+v"(ev_next));
            }
        }
        ++cur_kt;
        const bool ldv = ld_valid && !(since == 0 && ahead) && !((DBG & 2) && dbg_steps >= 1);
        ++dbg_steps;
        if (!(DBG & 16) && issue_first && ldv) issue(buf ^ 1, ld_kt, 0u);
        if (DBG & 8) t2 = clock64();
        if constexpr ((DBG & ~8) == 0) gp_kstep<decltype(FIRST)::value>(smem_p2, buf, wr, wc, l31, lh, acc);  // (8 = stamps around the real stream)
        else compute(buf, FIRST, (DBG & 16) && ldv, buf ^ 1, ld_kt);  // ONE call site: two would double the accumulator live ranges
        if (!(DBG & 16) && !issue_first && ldv) {
            unsigned dep = 0;
            if (!(DBG & 1)) asm("" : "+v"(dep) : "v"(acc[3][1]));  // scheduling-only: keeps the loads behind the MFMAs
            issue(buf ^ 1, ld_kt, dep);
        }
        if (ldv) advance();
        if (since == 0) ahead = false;
        ++since;
        if (DBG & 8) {
            const long long t3 = clock64();
            if (p.dbg && lane == 0 && dbg_n < 48 && (blockIdx.x == 0 || blockIdx.x == 101)) {
                long long* o = p.dbg + ((blockIdx.x ? 1 : 0) * 8 + wave) * 50 * 4 + dbg_n * 4;
                o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
                ++dbg_n;
            }
        }
        buf ^= 1;
    };
    for (;;) {
        step(std::true_type{});
        for (int kt = 1; kt < nk; ++kt) step(std::false_type{});
        long long e0 = 0;
        if (DBG & 8) e0 = clock64();
        if (overlap && ld_valid && !(DBG & 2)) {
            // the buffer of the step just computed is free once every wave is through it: the loads of the step after next
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            unsigned dep = 0;
            if (!(DBG & 1)) asm("" : "+v"(dep) : "v"(acc[3][1]));
            issue(buf ^ 1, ld_kt, dep);
            advance();
            ahead = true;
        }
        cur_kt = 0;
        if (has_e && nk < 4 && tile + slots < t_end) ev_next = fetch_e(tile + slots);
        if (!(DBG & 4)) { gp_acc_fence(acc); gp_epilogue<OUT, HAS_R, DBG>(p, smem_p2, acc, wave, tile / p.tiles_n, tile % p.tiles_n, e_run, ev); }
        else asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]), "v"(acc[0][1]), "v"(acc[1][1]), "v"(acc[2][1]), "v"(acc[3][1]));
        ev = ev_next;
        since = overlap ? 0 : 8;
        if ((DBG & 8) && p.dbg && lane == 0 && dbg_n < 48 && (blockIdx.x == 0 || blockIdx.x == 101)) {
            long long* o = p.dbg + ((blockIdx.x ? 1 : 0) * 8 + wave) * 50 * 4 + dbg_n * 4;
            o[0] = -1; o[1] = e0; o[2] = clock64(); o[3] = 0;
            ++dbg_n;
        }
        tile += slots;
        if (tile >= t_end) break;
    }
}

// validates one GEMM and fills its kernel parameter block (shared with the chained launch, gemm_p2c.hip)
int fill_gemm_p2_params(e2emv_ctx* ctx, const GemmP2Args& a, GemmP2Params& p) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: empty problem");
    const int K1 = a.A2 ? a.K1 : a.K;
    if (a.K % 32 || K1 % 32 || K1 <= 0 || K1 > a.K || (K1 < a.K && !a.A2))
        return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: K=%d K1=%d must be multiples of 32", a.K, K1);
    if (!a.A || !a.W || a.lda % 32 || a.lda < K1 || (a.A2 && (a.lda2 % 32 || a.lda2 < a.K - K1)))
        return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: operand planes need whole 32-column blocks (lda=%lld lda2=%lld)", (long long)a.lda, (long long)a.lda2);
    if ((uintptr_t)a.A % 16 || (a.A2 && (uintptr_t)a.A2 % 16) || (uintptr_t)a.W % 16 || (a.bias && (uintptr_t)a.bias % 16) || (a.Rp && (uintptr_t)a.Rp % 16))
        return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: operands must be 16-byte aligned");
    if (a.Rp && (a.ldr % 32 || a.ldr < a.N)) return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: residual planes need ldr %% 32 == 0");
    const int64_t a_bytes = (int64_t)a.M * a.lda * 4, a2_bytes = a.A2 ? (int64_t)a.M * a.lda2 * 4 : 16, w_bytes = (int64_t)a.N * a.K * 4;
    if (a_bytes >= ((int64_t)1 << 31) || a2_bytes >= ((int64_t)1 << 31) || w_bytes >= ((int64_t)1 << 31))
        return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: operand larger than 2 GB (M=%d lda=%lld): 32-bit byte offsets", a.M, (long long)a.lda);
    p = GemmP2Params{};
    p.A = a.A; p.A2 = a.A2 ? a.A2 : a.A; p.W = a.W;
    p.a_bytes = (unsigned)a_bytes; p.a2_bytes = (unsigned)(a.A2 ? a2_bytes : a_bytes); p.w_bytes = (unsigned)w_bytes;
    p.lda_b = (unsigned)(a.lda * 4); p.lda2_b = (unsigned)((a.A2 ? a.lda2 : a.lda) * 4); p.ldw_b = (unsigned)(a.K * 4);
    p.bias = a.bias; p.Rp = a.Rp; p.ldr = a.ldr;
    p.C32 = a.C32; p.Cp = a.Cp; p.Vt = a.Vt; p.ldc = a.ldc;
    p.M = a.M; p.N = a.N; p.K = a.K; p.K1 = K1;
    p.tiles_n = (a.N + P2_BN - 1) / P2_BN;
    p.total = ((a.M + P2_BM - 1) / P2_BM) * p.tiles_n;
    p.relu = a.relu ? 1 : 0;
    p.out_scale = a.out_scale;
    p.col_scale[0] = p.col_scale[1] = p.col_scale[2] = 1.f;
    p.n_rows = a.n_rows; p.heads = a.heads;
    p.EA = a.EA; p.EA2 = a.EA2; p.ER = a.ER; p.EC = a.EC; p.EVt = a.EVt; p.AR = a.AR; p.AC = a.AC;
    p.eld_a = (int)(a.lda / 64); p.eld_a2 = (int)(a.lda2 / 64); p.eld_r = (int)(a.ldr / 64); p.eld_c = (int)(a.ldc / 64);
    p.bias_amax = a.bias ? a.bias_amax : 0.f;
    if (a.EA && (a.lda % 64 || K1 % 64 || (a.A2 && (a.lda2 % 64 || (a.K - K1) % 64)) || a.M % 64))
        return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: tile exponents need 64-column / 64-row blocks (lda=%lld K1=%d M=%d)", (long long)a.lda, K1, a.M);
    if ((a.EC || a.EVt) && (a.M % 64 || (a.out == P2_OUT_PLANES && a.ldc % 64))) return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: output exponents need 64 x 64 blocks");
    if (a.ER && (a.ldr % 64 || a.M % 64)) return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: residual exponents need 64 x 64 blocks");
    if (int rc = ensure_flags(ctx)) return rc;
    p.stats = ctx->d_flags + 2;
    switch (a.out) {
        case P2_OUT_F32:
            if (!a.C32 || a.N % 4 || a.ldc % 4 || (uintptr_t)a.C32 % 16) return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: fp32 output needs N %% 4 == 0, ldc %% 4 == 0");
            break;
        case P2_OUT_PLANES:
            if (!a.Cp || a.N % 8 || a.ldc % 32 || a.ldc < a.N || (uintptr_t)a.Cp % 16) return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: plane output needs N %% 8 == 0, ldc %% 32 == 0");
            break;
        case P2_OUT_QKV:
            if (!a.Cp || !a.Vt || a.N != 3 * P2_BN || a.heads != 4 || a.n_rows <= 0 || a.n_rows % 32 || a.M % a.n_rows || a.relu || a.Rp ||
                (uintptr_t)a.Cp % 16 || (uintptr_t)a.Vt % 16)
                return set_err(ctx, E2EMV_ESHAPE, "gemm_p2: q|k|v output needs N = 768 (4 heads of 64), n_rows %% 32 == 0, M %% n_rows == 0");
            p.ldc = 2 * P2_BN;
            p.eld_c = 8;
            p.col_scale[0] = 0.125f * 1.4426950408889634f * P2_QS;  // log2(e) / sqrt(64), then the plane pre-scale
            p.col_scale[2] = P2_VS;
            break;
        default: return set_err(ctx, E2EMV_EINVAL, "gemm_p2: unknown output kind %d", a.out);
    }
    if (!ctx->d_dummy) E2EMV_HIP(ctx, hipMalloc((void**)&ctx->d_dummy, 4096));
    p.dummy = ctx->d_dummy;
    p.dbg = nullptr;
    return E2EMV_OK;
}

int launch_gemm_p2(e2emv_ctx* ctx, const GemmP2Args& a, hipStream_t s) {
    GemmP2Params p;
    if (int rc = fill_gemm_p2_params(ctx, a, p)) return rc;
    const void* fn = nullptr;
    switch (a.out) {
        case P2_OUT_F32: fn = a.Rp ? reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_F32, true>) : reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_F32, false>); break;
        case P2_OUT_PLANES: fn = a.Rp ? reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, true>) : reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false>); break;
        default: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_QKV, false>); break;
    }
    const int per_xcd = (p.total + 7) / 8;
    const int sl = std::min(per_xcd, std::max(1, ctx->num_cus / 8));
    const size_t lds = P2_LDSB;
#ifdef E2EMV_STAMPS
    // measurement build only (tools/p2_stamps.py): E2EMV_P2_DBG selects an ablation / the stamped variant of the planes kernel
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("E2EMV_P2_DBG"); dbg = e ? atoi(e) : 0; }
    static long long* d_buf = nullptr;
    const size_t nb = sizeof(long long) * 2 * 8 * 50 * 4;
    if (dbg && a.out == P2_OUT_PLANES && !a.Rp) {
        switch (dbg) {
            case 1: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 1>); break;
            case 2: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 2>); break;
            case 3: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 3>); break;
            case 4: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 4>); break;
            case 6: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 6>); break;
            case 8: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 8>); break;
            case 16: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 16>); break;
            case 17: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 17>); break;
            case 24: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 24>); break;
            case 32: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 32>); break;
            case 64: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 64>); break;
            case 128: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 128>); break;
            case 256: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 256>); break;
            case 36: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 36>); break;
            case 40: fn = reinterpret_cast<const void*>(gemm_p2_kernel<P2_OUT_PLANES, false, 40>); break;
            default: break;
        }
        if (dbg & 8) {
            if (!d_buf) E2EMV_HIP(ctx, hipMalloc((void**)&d_buf, nb));
            E2EMV_HIP(ctx, hipMemsetAsync(d_buf, 0, nb, s));
            p.dbg = d_buf;
        }
    }
#endif
    if (int rc = ensure_dynamic_lds(ctx, fn, lds)) return rc;
    void* args[] = {&p};
    E2EMV_HIP(ctx, hipLaunchKernel(fn, dim3(8 * sl), dim3(512), args, lds, s));
    E2EMV_CHECK_LAUNCH(ctx, "gemm_p2_kernel");
#ifdef E2EMV_STAMPS
    if (p.dbg) {
        E2EMV_HIP(ctx, hipStreamSynchronize(s));
        std::vector<long long> h(2 * 8 * 50 * 4);
        E2EMV_HIP(ctx, hipMemcpy(h.data(), d_buf, nb, hipMemcpyDeviceToHost));
        static int printed = 0;
        if (printed++ < 2)
            for (int wg = 0; wg < 2; ++wg)
                for (int w = 0; w < 8; w += 5) {
                    const long long* o = &h[((size_t)wg * 8 + w) * 50 * 4];
                    fprintf(stderr, "gemm_p2 M=%d N=%d K=%d wg %d wave %d: per K step wait+barrier | issue | compute | total   (epilogue rows: -1)\n", p.M, p.N, p.K, wg ? 101 : 0, w);
                    for (int i = 0; i < 44; ++i) {
                        const long long* t = o + i * 4;
                        if (!t[0]) break;
                        if (t[0] == -1) { fprintf(stderr, "  %2d: epilogue %lld\n", i, t[2] - t[1]); continue; }
                        fprintf(stderr, "  %2d: %5lld %5lld %5lld | %5lld\n", i, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[3] - t[0]);
                    }
                }
    }
#endif
    return E2EMV_OK;
}

}  // namespace e2emv
