// The row-local GEMMs of a GNN layer CHAINED in one persistent launch (f16x2 kernel generation 5).
//
// With D = 256 a block of 256 keypoint rows is closed under everything in a layer except the attention:
//     MLP0 (2 column tiles, K = [x | attention] = 512)  ->  MLP1 + residual (1 tile: the complete x_new of the rows)
//     ->  q | k | v of the NEXT layer (3 tiles, K = 256)        [last layer: -> final_proj (1 tile)]
// gemm_p2.hip runs these as three launches of 1 - 3 tiles per workgroup; every launch pays its ramp (first operand tiles of
// 256 workgroups at once) and its tail, and every tile's epilogue - a 67 MB store burst that all CUs reach together - has
// nothing to run under but the next launch's ramp.  Here ONE workgroup walks the 6 tiles of its row block back to back:
//   * tile shape, bytes per MAC, weight traffic, the K step and the epilogues are gemm_p2's (gemm_p2_core.h): the results
//     are bit-identical to the three launches (tests/test_gpu_round5.py compares them);
//   * the hand-off between the GEMMs goes through HBM / L2 exactly as before - the producer tile stores its planes, the
//     consumer tile loads them - but inside ONE workgroup: "my stores have retired" (s_waitcnt vmcnt) + the workgroup barrier
//     is all the synchronisation there is (the CU's vector L1 is write-through and coherent for the waves of a workgroup; no
//     grid barrier, no flags, no other workgroup ever touches these rows);
//   * 5 of the 6 store bursts drain under the K loop of the following tile (the counted-vmcnt pipeline of gemm_p2), the
//     operand loads of the next tile run two K steps ahead across the epilogue as before.
// Dependencies between consecutive tiles of a row block (dep_kt of a stage = the first K step of its FIRST tile that reads
// what the tile right before it stored):
//   * none (MLP0's tiles; the 2nd, 3rd tile of a stage): loads and the exponent fetch run ahead as in gemm_p2;
//   * soft, dep_kt >= 4 (MLP1: K steps 0 - 7 read the hidden columns of MLP0's FIRST tile, whose stores retired - counted
//     vmcnt(0) of K step 2 + the barrier of step 3 - long before; steps 8 - 15 read the second tile's, stored by the epilogue
//     right in front): the loads still run ahead - the steps that depend are issued behind step 3's barrier by
//     construction - and only the exponents of the K blocks that depend are fetched again, at step 4;
//   * hard, dep_kt == 0 (q | k | v / final_proj read x_new from K step 0 on): no run-ahead; after the producer's epilogue
//     the workgroup waits for its stores (vmcnt(0)), meets at a barrier and starts the consumer like a first tile.  This is
//     the one exposed store drain per row block and launch (gemm_p2 exposes one per launch and tile round).
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "gemm_p2_core.h"

namespace e2emv {

constexpr int P2C_MAX_STAGES = 4, P2C_MAX_TILES = 16;
constexpr int P2C_INDEP = P2_CHAIN_INDEP;

struct GemmP2ChainParams {
    GemmP2Params st[P2C_MAX_STAGES];
    int kind[P2C_MAX_STAGES];       // P2_OUT_* | 4 = with residual
    int dep_kt[P2C_MAX_STAGES];     // see above; P2C_INDEP = the stage's operands come from earlier launches (the kernel reads t_info)
    int first[P2C_MAX_STAGES + 1];  // index of a stage's first tile among the tiles of a row block; [n_stages] = tiles per row block
    int n_stages, row_blocks;
    int t_info[P2C_MAX_TILES];      // per tile of a row block: stage | column tile << 8 | hard << 16 | soft << 17 (one scalar load)
};

// DBG (instantiated only in -DE2EMV_STAMPS builds, tools/p2c_stamps.py; results wrong but for 8, 16, 1024): 4 no epilogue, 8 s_memtime
// stamps per tile (K loop | epilogue | hand-off) of two workgroups, 16 stamps per K step, 512 no wait at the hard hand-off, 1024
// activation loads non-temporal, 2048 odd column tiles walk K backwards
template <int DBG>
__global__ __launch_bounds__(512, 1) void gemm_p2_chain_kernel(GemmP2ChainParams cp_by_value) {
    extern __shared__ __attribute__((aligned(16))) char smem_p2c[];
    // The stage of a tile is a run-time index into the parameter block.  Indexing the by-value argument makes hipcc copy the
    // whole block to scratch (and every field a vector load); the same bytes read through the kernarg segment pointer
    // (constant address space, the block is the kernel's only argument: offset 0) stay scalar loads.
    (void)cp_by_value;
    const GemmP2ChainParams& cp = *(const GemmP2ChainParams*)(const __attribute__((address_space(4))) GemmP2ChainParams*)__builtin_amdgcn_kernarg_segment_ptr();

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // Nothing derived from the lane index lives longer than it must (gp_lane_now): the K loop's fragment addresses are derived
    // anew per tile (so the epilogue between two K loops has their registers), the loader's offsets per K step, the exponent
    // fetch's per fetch.  The four epilogues of this kernel need every register gemm_p2's single one has.
    int l31 = 0, lh = 0;
    unsigned rc_t = 0;  // the loader's lane constants (lane_rc below), per tile like l31 / lh
    const int nst = cp.n_stages, tpr = cp.first[nst];
    // workgroup b walks row blocks b, b + gridDim.x, ...; flat tile index f = (row block of mine) * tpr + tile within the block
    const int n_my = (cp.row_blocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = n_my * tpr;
    if (total <= 0) return;
    auto is_hard = [&](int f) { return f > 0 && (cp.t_info[f % tpr] & (1 << 16)) != 0; };  // reads, from K step 0 on, what the tile before it stores

    // ---- loader (gemm_p2's, with the operand descriptors of the LOAD position's stage): wave w moves rows 32 w .. 32 w + 31 of
    // the activation tile and of the weight tile, 4 + 4 LDS-direct loads per K step; load i = rows 8 i .. 8 i + 7 of the wave's
    // 32, lane -> (row lane >> 3, LDS position lane & 7), source chunk = position ^ ((row >> 1) & 7).  Whole tiles only (M, N
    // multiples of 256: the launcher checks), so row 8 i + r of a wave sits 8 i rows behind row r - a wave-uniform byte offset
    // that rides in the load's scalar offset together with the tile's base - and its swizzle is that of row r with bit 2
    // flipped for odd i: the lane's part of an offset is a handful of VALU instructions per K step, recomputed there (gemm_p2
    // holds twelve offset registers across the whole kernel).
    unsigned a_base = 0, w_base = 0;  // byte offset of row 0 of this wave's 32 rows of the load position's tiles (scalar)
    unsigned lda_l = 0, ldw_l = 0;    // row strides in bytes
    __amdgpu_buffer_rsrc_t rsA, rsA2, rsW;
    int nk_l = 0, nk1_l = 0;
    bool rev_l = false;  // (measurement variant 2048: odd column tiles walk K backwards - their first steps re-read what the tile before read last)
    auto setup = [&](int f) {
        const int rbi = f / tpr, r = f - rbi * tpr, ti = cp.t_info[r], s = ti & 255;
        const int tm = (int)blockIdx.x + rbi * (int)gridDim.x, tn = (ti >> 8) & 255;
        const GemmP2Params& q = cp.st[s];
        rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(q.A), 0, (int)q.a_bytes, 0x00020000);
        rsA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(q.A2), 0, (int)q.a2_bytes, 0x00020000);
        rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(q.W), 0, (int)q.w_bytes, 0x00020000);
        nk_l = q.K / P2_BK;
        nk1_l = q.K1 / P2_BK;
        rev_l = (DBG & 2048) && (tn & 1);
        lda_l = q.lda_b;
        ldw_l = q.ldw_b;
        a_base = (((DBG & 4096) ? 0u : (unsigned)tm * P2_BM) + 32u * wave) * lda_l;  // (4096: every workgroup reads row block 0)
        w_base = ((unsigned)tn * P2_BN + 32u * wave) * ldw_l;
    };
    // (ld_r | c16 << 8) of a lane: its row among the 8 of a load and its swizzled 16-byte chunk - two VGPRs' worth of lane
    // constants in one, made per tile for the K loop's steps and on the spot for the issues outside it
    auto lane_rc = [&]() {
        const unsigned ln = (unsigned)gp_lane_now(), ld_r = ln >> 3, ld_p = ln & 7;
        return ld_r | ((ld_p ^ (ld_r >> 1)) * 16u) << 8;
    };
    auto issue = [&](int buf, int kt, unsigned rc) {
        if ((DBG & 2048) && rev_l) kt = nk_l - 1 - kt;
        char* dst = smem_p2c + buf * P2_BUFB + 32 * wave * P2_ROWB;
        const unsigned ld_r = rc & 255u, c16 = rc >> 8;
        // (24-bit multiplies: full rate - a 32-bit v_mad_u64_u32 per operand stood in front of every step's loads)
        const unsigned a0 = __umul24(ld_r, lda_l) + c16, a1 = a0 ^ 64u, w0 = __umul24(ld_r, ldw_l) + c16, w1 = w0 ^ 64u;
        const unsigned a_step = 8u * lda_l, w_step = 8u * ldw_l;
        // (E2EMV_STAMPS builds: DBG & 1024 = the activation stream with the non-temporal hint)
#define P2C_LDA(rs, d, v, so) do { if constexpr ((DBG & 1024) != 0) p2_glds16_nt(rs, d, v, so); else p2_glds16(rs, d, v, so); } while (0)
        if (kt < nk1_l) {
            const unsigned so = a_base + (unsigned)kt * 128u;
            P2C_LDA(rsA, dst, a0, so);
            P2C_LDA(rsA, dst + 1024, a1, so + a_step);
            P2C_LDA(rsA, dst + 2048, a0, so + 2 * a_step);
            P2C_LDA(rsA, dst + 3072, a1, so + 3 * a_step);
        } else {
            const unsigned so = a_base + (unsigned)(kt - nk1_l) * 128u;
            P2C_LDA(rsA2, dst, a0, so);
            P2C_LDA(rsA2, dst + 1024, a1, so + a_step);
            P2C_LDA(rsA2, dst + 2048, a0, so + 2 * a_step);
            P2C_LDA(rsA2, dst + 3072, a1, so + 3 * a_step);
        }
#undef P2C_LDA
        const unsigned sw = w_base + (unsigned)kt * 128u;
        p2_glds16(rsW, dst + P2_TILEB, w0, sw);
        p2_glds16(rsW, dst + P2_TILEB + 1024, w1, sw + w_step);
        p2_glds16(rsW, dst + P2_TILEB + 2048, w0, sw + 2 * w_step);
        p2_glds16(rsW, dst + P2_TILEB + 3072, w1, sw + 3 * w_step);
    };

    // ---- tile exponents (p2.h) - SCALAR here.  gemm_p2 fetches a tile's exponents with one vector load (lane i: K block i) and
    // reads them with v_readlane; in this kernel that register was one too many beside the four epilogues (hipcc spilled it, or
    // put an s_waitcnt vmcnt(0) for it in front of every K step's readlane - in the middle of the counted load / store pipeline).
    // The exponents are wave-uniform and small (P2_EMIN .. P2_EMAX: a signed byte): a tile's <= 8 K-block exponents are fetched
    // ONCE per tile by scalar loads (asm statements that carry their own s_waitcnt lgkmcnt: nothing of the vector memory
    // counter, nothing pending that hipcc could copy or spill early), packed into one 64-bit scalar, and a K block's exponent
    // is a shift + sign extension - the K loop contains no memory operation for the side-band.  The scalar cache is not coherent
    // with the epilogues' vector stores: it is invalidated where a tile reads exponents the tile before it stored (hard
    // hand-off; step 4 of a soft tile, whose K blocks >= dep_kt / 2 are fetched again there - the producer's stores retired at
    // step 2, everybody's by step 3's barrier).
    typedef int p2c_i4 __attribute__((ext_vector_type(4)));
    typedef int p2c_i2 __attribute__((ext_vector_type(2)));
    auto sload4 = [](const int* p) {
        p2c_i4 r;
        asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(p) : "memory");
        return r;
    };
    auto sload2 = [](const int* p) {
        p2c_i2 r;
        asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(p) : "memory");
        return r;
    };
    auto pack4 = [](p2c_i4 v) { return (unsigned)((v[0] & 255) | (v[1] & 255) << 8 | (v[2] & 255) << 16 | (v[3] & 255) << 24); };
    // packed exponents of tile f's A operand for this wave's 64-row block (K blocks 0 - 3 | 4 - 7); the launcher admits only
    // segment shapes of 4 | 4, 8 and 4 blocks
    auto fetch_ew = [&](int f, bool low, bool high, unsigned long long old) {
        const int rbi = f / tpr, r = f - rbi * tpr, ti = cp.t_info[r], s = ti & 255;
        const int erow = ((int)blockIdx.x + rbi * (int)gridDim.x) * 4 + wr;
        const GemmP2Params& q = cp.st[s];
        if (!q.EA) return 0ull;
        const int nb1 = (q.K1 / P2_BK) >> 1, nb = (q.K / P2_BK + 1) >> 1;
        unsigned lo = (unsigned)old, hi = (unsigned)(old >> 32);
        if (low) lo = pack4(sload4(q.EA + erow * q.eld_a));
        if (high) {
            hi = 0;
            if (nb1 == 8) hi = pack4(sload4(q.EA + erow * q.eld_a + 4));
            else if (nb > nb1 && q.EA2) hi = pack4(sload4(q.EA2 + erow * q.eld_a2));
        }
        return (unsigned long long)hi << 32 | lo;
    };
    auto e_of = [](unsigned long long ew, int kb) { return (int)(signed char)(ew >> (8 * kb)); };

    p2_f32x16 acc[4][2];
    int lf = 0, lkt = 0;                        // load position: (flat tile, K step)
    bool ld_valid = true, ld_blocked = false;   // blocked: the next tile is hard-dependent - its loads wait for this tile's epilogue
    auto advance = [&]() {
        if (__builtin_expect(lkt + 1 < nk_l, 1)) { ++lkt; return; }
        if (lf + 1 >= total) { ld_valid = false; return; }
        if (is_hard(lf + 1)) { ld_blocked = true; return; }
        ++lf;
        lkt = 0;
        setup(lf);
    };
    setup(0);
    issue(0, 0, lane_rc());
    advance();
    unsigned long long ew = fetch_ew(0, true, true, 0), ew_next = 0;  // packed exponents of the current / the next tile
    int er_c[2] = {0, 0}, ar_c[2] = {0, 0};  // residual blocks of this wave's two 64-column chunks: exponents / max |x| bits
    const bool issue_first = wave >= 4;  // the two waves of a SIMD take opposite orders (gemm_p2.hip)
    int e_run = 0, cur_kt = 0;
    int since = 8;       // K steps since the last epilogue
    bool ahead = false;  // the loads of the step after next were issued before that epilogue
    int buf = 0;
    // per-tile state of the compute position (wave-uniform)
    bool has_e = false, soft = false, prefetch_next = false;
    int f = 0;
    int n_st = 0;  // (DBG & 16: per K step stamps of workgroup 0 / 101, waves 0 and 5)
    auto step = [&](auto FIRST) {
        long long t0 = 0, t1 = 0, t2 = 0;
        if (DBG & 16) t0 = __builtin_amdgcn_s_memtime();
        if (since == 0 && ahead) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
        else if (since <= 1) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (DBG & 32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (measurement: the K loop without its per-step barrier - races, time only)
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (DBG & 16) t1 = __builtin_amdgcn_s_memtime();
        if (has_e && (cur_kt & 1) == 0) {  // a new K block
            // (everything that runs once per tile is marked unlikely: the K step's own path from the barrier to its first MFMA is
            // matrix-pipe idle time, and a cold block hipcc leaves inside it costs a taken branch + an instruction fetch)
            if (__builtin_expect(soft && cur_kt == 4, 0)) {
                asm volatile("s_dcache_inv\n\tbuffer_inv sc0\n\ts_waitcnt lgkmcnt(0)" ::: "memory");  // (vector L1 as at the hard hand-off)
                ew = fetch_ew(f, false, true, ew);
            }
            const int e_step = e_of(ew, cur_kt >> 1);
            if (!decltype(FIRST)::value && __builtin_expect(e_step != e_run, 0)) {
                asm volatile("s_nop 15");  // (the previous step's asm MFMAs -> the VALU below: hipcc does not pad an asm's results)
                const int d = e_run - e_step;
                const float fs = d < -126 ? 0.f : p2_exp2i(d);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[j][i][r] *= fs;
            }
            e_run = e_step;
            if (__builtin_expect(cur_kt == 2, 0)) {  // once per tile, beside the operand loads of K step 2
                if (prefetch_next) ew_next = fetch_ew(f + 1, true, true, 0);
                const int rr_ = f % tpr, ti_ = cp.t_info[rr_], s_ = ti_ & 255;
                const GemmP2Params& q_ = cp.st[s_];
                if ((cp.kind[s_] & 4) && q_.ER) {  // the residual blocks this wave adds in the tile's epilogue (old data: any time)
                    const int erow = ((int)blockIdx.x + (f / tpr) * (int)gridDim.x) * 4 + wr;
                    const int cb = ((ti_ >> 8) & 255) * 4 + wc * 2;  // (two adjacent 64-column blocks: one 8-byte load each)
                    const p2c_i2 e2 = sload2(q_.ER + erow * q_.eld_r + cb);
                    er_c[0] = e2[0]; er_c[1] = e2[1];
                    if (q_.AR) {
                        const p2c_i2 a2 = sload2(reinterpret_cast<const int*>(q_.AR) + erow * q_.eld_r + cb);
                        ar_c[0] = a2[0]; ar_c[1] = a2[1];
                    }
                }
            }
        }
        ++cur_kt;
        const bool ldv = ld_valid && !ld_blocked && !(since == 0 && ahead);
        const bool ldi = ldv && !(DBG & 2);  // (2: no operand loads in the K loop)
        if (issue_first && ldi) issue(buf ^ 1, lkt, rc_t);
        if (DBG & 16) t2 = __builtin_amdgcn_s_memtime();
        gp_kstep<decltype(FIRST)::value, DBG & 1>(smem_p2c, buf, wr, wc, l31, lh, acc);
        if (!issue_first && ldi) {
            unsigned rc = rc_t;
            asm("" : "+v"(rc) : "v"(acc[3][1]));  // scheduling-only: keeps the loads behind the MFMAs
            issue(buf ^ 1, lkt, rc);
        }
        if (ldv) advance();
        if (since == 0) ahead = false;
        ++since;
        buf ^= 1;
        if (DBG & 16) {
            const GemmP2Params& q0 = cp.st[0];
            if (q0.dbg && gp_lane_now() == 0 && (blockIdx.x == 0 || blockIdx.x == 101) && (wave == 0 || wave == 5) && n_st < 80) {
                long long* o = q0.dbg + (((blockIdx.x ? 1 : 0) * 2 + (wave ? 1 : 0)) * 80 + n_st) * 4;
                o[0] = t0; o[1] = t1; o[2] = t2; o[3] = __builtin_amdgcn_s_memtime();
            }
            ++n_st;
        }
    };
    for (f = 0; f < total; ++f) {
        const int rbi = f / tpr, r = f - rbi * tpr, ti = cp.t_info[r], s = ti & 255;
        const int tm = (int)blockIdx.x + rbi * (int)gridDim.x, tn = (ti >> 8) & 255;
        const GemmP2Params& q = cp.st[s];
        const int nk = q.K / P2_BK;
        {
            const int ln = gp_lane_now();
            l31 = ln & 31;
            lh = ln >> 5;
            rc_t = lane_rc();
        }
        has_e = q.EA != nullptr || ((cp.kind[s] & 4) && q.ER != nullptr);
        soft = f > 0 && (ti & (1 << 17)) != 0;
        prefetch_next = f + 1 < total && !is_hard(f + 1);
        long long ts0 = 0, ts1 = 0, ts2 = 0;
        if (DBG & 8) ts0 = __builtin_amdgcn_s_memtime();
        step(std::true_type{});
        for (int kt = 1; kt < nk; ++kt) step(std::false_type{});
        if (DBG & 8) ts1 = __builtin_amdgcn_s_memtime();
        if (ld_valid && !ld_blocked) {
            // the buffer of the step just computed is free once every wave is through it: the loads of the step after next
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            unsigned rc = lane_rc();
            asm("" : "+v"(rc) : "v"(acc[3][1]));
            issue(buf ^ 1, lkt, rc);
            advance();
            ahead = true;
        }
        cur_kt = 0;
        gp_acc_fence(acc);
        // the epilogue's view of the side-band: lanes 32 - 35 of one register (gp_epilogue reads them with v_readlane)
        int ev = 0;
        if ((cp.kind[s] & 4) && q.ER) {
            asm("v_writelane_b32 %0, %1, 32\n\tv_writelane_b32 %0, %2, 33\n\tv_writelane_b32 %0, %3, 34\n\tv_writelane_b32 %0, %4, 35"
                : "+v"(ev) : "s"(er_c[0]), "s"(er_c[1]), "s"(ar_c[0]), "s"(ar_c[1]));
        }
        if (!(DBG & 4)) {
            constexpr int EDBG = DBG & (64 | 128 | 8192);  // no stores | stores into 1 MB | residual from a slab
            switch (cp.kind[s]) {
                case P2_OUT_PLANES: gp_epilogue<P2_OUT_PLANES, false, EDBG>(q, smem_p2c, acc, wave, tm, tn, e_run, ev); break;
                case P2_OUT_PLANES | 4:
                    // MLP1: the tile behind it is hard-dependent - nothing has run ahead into the tile buffers; once every wave is through
                    // the last K step they carry this epilogue's residual ring (gemm_p2_core.h, RLDS).  (E2EMV_P2C_DBG 32768, measurement
                    // build: the residual through registers as before)
                    if (ld_blocked && !(DBG & 32768)) {
                        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                        gp_epilogue<P2_OUT_PLANES, true, EDBG, true>(q, smem_p2c, acc, wave, tm, tn, e_run, ev);
                    } else {
                        gp_epilogue<P2_OUT_PLANES, true, EDBG>(q, smem_p2c, acc, wave, tm, tn, e_run, ev);
                    }
                    break;
                case P2_OUT_QKV: gp_epilogue<P2_OUT_QKV, false, EDBG>(q, smem_p2c, acc, wave, tm, tn, e_run, ev); break;
                default: gp_epilogue<P2_OUT_F32, false, EDBG>(q, smem_p2c, acc, wave, tm, tn, e_run, ev); break;
            }
        } else {
            asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]), "v"(acc[0][1]), "v"(acc[1][1]), "v"(acc[2][1]), "v"(acc[3][1]));
        }
        if (DBG & 8) ts2 = __builtin_amdgcn_s_memtime();
        ew = ew_next;
        since = 0;
        if (ld_blocked) {
            // hard hand-off: every wave's stores of this tile have retired, then everybody's have; the consumer starts like a first tile
            // (buffer_inv sc0: the consumer's loads must not be served from vector-L1 lines older than the producer waves' stores.  In
            // this build - CU mode, no tgsplit: build.py refuses the flag - all waves of the workgroup share one write-through L1 and
            // the invalidate is redundant; it is issued anyway, once per row block, so that the hand-off does not rest on that)
            if (DBG & 512) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\tbuffer_inv sc0" ::: "memory");
            ld_blocked = false;
            ++lf;
            lkt = 0;
            setup(lf);
            asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
            ew = fetch_ew(lf, true, true, 0);
            issue(buf, 0, lane_rc());
            advance();
            since = 8;
            ahead = false;
        }
        if ((DBG & 8) && q.dbg && gp_lane_now() == 0 && (blockIdx.x == 0 || blockIdx.x == 101) && f < 12) {
            long long* o = q.dbg + (((blockIdx.x ? 1 : 0) * 8 + wave) * 12 + f) * 4;
            o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = __builtin_amdgcn_s_memtime();
        }
    }
}

// `a[0 .. n)`: the GEMMs of the chain in execution order, all over the same M rows; dep_kt[i] as described at the top.
int launch_gemm_p2_chain(e2emv_ctx* ctx, const GemmP2Args* a, const int* dep_kt, int n, hipStream_t s) {
    if (n < 1 || n > P2C_MAX_STAGES) return set_err(ctx, E2EMV_EINVAL, "gemm_p2 chain: %d stages (1 .. %d)", n, P2C_MAX_STAGES);
    GemmP2ChainParams cp{};
    cp.n_stages = n;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        if (int rc = fill_gemm_p2_params(ctx, a[i], cp.st[i])) return rc;
        if (a[i].M != a[0].M) return set_err(ctx, E2EMV_ESHAPE, "gemm_p2 chain: stage %d has %d rows, stage 0 %d", i, a[i].M, a[0].M);
        if (a[i].K < 4 * P2_BK) return set_err(ctx, E2EMV_ESHAPE, "gemm_p2 chain: K = %d (the store / load overlap needs >= 4 K steps)", a[i].K);
        if (a[i].out == P2_OUT_F32 && a[i].Rp) return set_err(ctx, E2EMV_ESHAPE, "gemm_p2 chain: fp32 output with a residual is not a chain stage");
        if (dep_kt[i] != 0 && dep_kt[i] < 4) return set_err(ctx, E2EMV_EINVAL, "gemm_p2 chain: dep_kt %d (0, >= 4 or independent)", dep_kt[i]);
        // a softly dependent stage is one tile, followed by a hard one; the first exponent that depends is requested at step
        // dep_kt - 2, behind the scalar-cache invalidate of step 4
        if (dep_kt[i] >= 4 && dep_kt[i] != P2C_INDEP && (a[i].N != P2_BN || i + 1 >= n || dep_kt[i + 1] != 0 || dep_kt[i] < 8 || (dep_kt[i] & 1)))
            return set_err(ctx, E2EMV_EINVAL, "gemm_p2 chain: a softly dependent stage must be one tile wide, depend from an even K step >= 8 on and be followed by a hard-dependent one");
        if (i == 0 && dep_kt[i] != P2C_INDEP) return set_err(ctx, E2EMV_EINVAL, "gemm_p2 chain: the first stage cannot depend on a tile before it");
        // a soft dependency may only reach back over the tile right in front: the tiles before that one must have retired their
        // stores, which the K loop of one tile (>= 4 steps) guarantees
        if ((cp.st[i].EA != nullptr) != (cp.st[0].EA != nullptr)) return set_err(ctx, E2EMV_EINVAL, "gemm_p2 chain: tile exponents on all stages or on none");
        // the scalar side-band of the kernel: K segments of 4 | 4, 8 or 4 exponent blocks, residual blocks in adjacent pairs
        if (a[i].EA) {
            const int nb1 = (a[i].A2 ? a[i].K1 : a[i].K) / 64, nb = a[i].K / 64;
            if (!((nb1 == 4 && (nb == 4 || nb == 8)) || (nb1 == 8 && nb == 8)) || (a[i].ER && (a[i].ldr / 64) % 2))
                return set_err(ctx, E2EMV_ESHAPE, "gemm_p2 chain: stage %d: tile exponents need K segments of 256 | 256, 512 or 256 columns (K=%d K1=%d)", i, a[i].K, a[i].K1);
        }
        // whole tiles only, one row stride for both K segments (the one-register loader of the kernel)
        if (a[i].M % P2_BM || a[i].N % P2_BN || (a[i].A2 && a[i].lda2 != a[i].lda) || a[i].lda * 4 >= (1 << 24) || (int64_t)a[i].K * 4 >= (1 << 24))
            return set_err(ctx, E2EMV_ESHAPE, "gemm_p2 chain: stage %d needs M, N multiples of 256 and lda2 == lda (M=%d N=%d)", i, a[i].M, a[i].N);
        cp.kind[i] = a[i].out | (a[i].Rp ? 4 : 0);
        cp.dep_kt[i] = dep_kt[i];
        cp.first[i] = tiles;
        for (int t = 0; t < cp.st[i].tiles_n; ++t) {
            if (tiles + t >= P2C_MAX_TILES) return set_err(ctx, E2EMV_ESHAPE, "gemm_p2 chain: more than %d tiles per row block", P2C_MAX_TILES);
            cp.t_info[tiles + t] = i | t << 8 | ((t == 0 && dep_kt[i] == 0) ? 1 << 16 : 0) | ((t == 0 && dep_kt[i] >= 4 && dep_kt[i] != P2C_INDEP) ? 1 << 17 : 0);
        }
        tiles += cp.st[i].tiles_n;
    }
    cp.first[n] = tiles;
    cp.row_blocks = (a[0].M + P2_BM - 1) / P2_BM;
    const int grid = std::min(cp.row_blocks, std::max(1, ctx->num_cus));
    const void* fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<0>);
#ifdef E2EMV_STAMPS
    // measurement build only (tools/p2c_stamps.py): E2EMV_P2C_DBG selects an ablation / the stamped variant
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("E2EMV_P2C_DBG"); dbg = e ? atoi(e) : 0; }
    static long long* d_buf = nullptr;
    const size_t nb = sizeof(long long) * 4 * 80 * 4;  // (>= 2 * 8 * 12 * 4)
    switch (dbg) {
        case 4: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<4>); break;
        case 8: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<8>); break;
        case 512: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<512>); break;
        case 516: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<516>); break;
        case 1024: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<1024>); break;
        case 1028: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<1028>); break;
        case 2048: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<2048>); break;
        case 16: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<16>); break;
        // round 6: 1 no MFMAs, 2 no operand loads in the K loop, 4096 every workgroup's activation rows = row block 0 (L2-resident),
        // 64 epilogues without stores, 128 epilogue stores into 1 MB, 8192 the residual from an L2-resident slab
        case 1: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<1>); break;
        case 2: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<2>); break;
        case 4096: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<4096>); break;
        case 4100: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<4100>); break;
        case 64: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<64>); break;
        case 128: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<128>); break;
        case 8192: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<8192>); break;
        case 72: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<72>); break;      // per-tile stamps of the variants without stores / with L2-resident activations
        case 4104: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<4104>); break;
        case 4160: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<4160>); break;  // no stores AND L2-resident activations
        case 32: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<32>); break;      // no per-step barrier
        case 36: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<36>); break;
        case 32768: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<32768>); break;  // MLP1's residual through registers (the round-5 form)
        case 32776: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<32776>); break;  // ... with per-tile stamps
        case 12416: fn = reinterpret_cast<const void*>(gemm_p2_chain_kernel<12416>); break;  // 4096 + 8192 + 128: no HBM traffic but the weights
        default: break;
    }
    if ((dbg & 8) || dbg == 16) {
        if (!d_buf) E2EMV_HIP(ctx, hipMalloc((void**)&d_buf, nb));
        E2EMV_HIP(ctx, hipMemsetAsync(d_buf, 0, nb, s));
        for (int i = 0; i < n; ++i) cp.st[i].dbg = d_buf;
    }
#endif
    if (int rc = ensure_dynamic_lds(ctx, fn, P2_LDSB)) return rc;
    void* args[] = {&cp};
    E2EMV_HIP(ctx, hipLaunchKernel(fn, dim3(grid), dim3(512), args, P2_LDSB, s));
    E2EMV_CHECK_LAUNCH(ctx, "gemm_p2_chain_kernel");
#ifdef E2EMV_STAMPS
    if (dbg == 16) {
        static int printed16 = 0;
        if (printed16++ < 2) {
            E2EMV_HIP(ctx, hipStreamSynchronize(s));
            std::vector<long long> h(4 * 80 * 4);
            E2EMV_HIP(ctx, hipMemcpy(h.data(), d_buf, nb, hipMemcpyDeviceToHost));
            for (int q = 0; q < 4; ++q) {
                fprintf(stderr, "gemm_p2_chain wg %d wave %d: per K step  wait+barrier | exponents+issue | compute(+issue) | total   (ticks)\n", (q >> 1) ? 101 : 0, (q & 1) ? 5 : 0);
                const long long* o = &h[(size_t)q * 80 * 4];
                for (int i = 0; i < 80 && o[4 * i]; ++i)
                    fprintf(stderr, "  %2d: %5lld %5lld %5lld | %5lld   (gap to next %lld)\n", i, o[4 * i + 1] - o[4 * i], o[4 * i + 2] - o[4 * i + 1], o[4 * i + 3] - o[4 * i + 2],
                            o[4 * i + 3] - o[4 * i], i + 1 < 80 && o[4 * i + 4] ? o[4 * i + 4] - o[4 * i + 3] : 0);
            }
        }
    }
    if (dbg & 8) {
        static int printed = 0;
        if (printed++ < 1) {
            E2EMV_HIP(ctx, hipStreamSynchronize(s));
            std::vector<long long> h(2 * 8 * 12 * 4);
            E2EMV_HIP(ctx, hipMemcpy(h.data(), d_buf, nb, hipMemcpyDeviceToHost));
            for (int wg = 0; wg < 2; ++wg)
                for (int w = 0; w < 8; w += 5) {
                    fprintf(stderr, "gemm_p2_chain wg %d wave %d: per tile  K loop | ahead-issue + epilogue | hand-off   (s_memtime ticks, ~0.54 ns each)\n", wg ? 101 : 0, w);
                    const long long* o = &h[((size_t)wg * 8 + w) * 12 * 4];
                    for (int f = 0; f < tiles && f < 12; ++f)
                        fprintf(stderr, "  tile %d: %6lld | %6lld | %6lld   (start +%lld)\n", f, o[4 * f + 1] - o[4 * f], o[4 * f + 2] - o[4 * f + 1],
                                o[4 * f + 3] - o[4 * f + 2], o[4 * f] - o[0]);
                }
        }
    }
#endif
    return E2EMV_OK;
}

}  // namespace e2emv
