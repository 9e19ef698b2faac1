// Shared core of the plane GEMMs (gemm_p2.hip: one GEMM per launch; gemm_p2c.hip: the row-local GEMMs of a GNN layer chained in
// one persistent launch): tile constants, the kernel parameter block, the pipelined K step and the epilogue.  See gemm_p2.hip
// for the arithmetic and the layout.
#pragma once
#include <type_traits>

#include "p2.h"

namespace e2emv {

typedef __attribute__((ext_vector_type(16))) float p2_f32x16;

constexpr int P2_BM = 256, P2_BN = 256, P2_BK = 32;
constexpr int P2_ROWB = 128;                    // bytes of one tile row per K step: 32 hi halves | 32 lo halves
constexpr int P2_TILEB = P2_BM * P2_ROWB;       // 32 KB per operand tile
constexpr int P2_BUFB = 2 * P2_TILEB;           // A tile | W tile
constexpr int P2_SLABB = 32 * 32 * 4;            // one epilogue slab per wave (32 rows x 32 floats), behind the tile buffers
constexpr int P2_LDSB = 2 * P2_BUFB + 8 * P2_SLABB;  // 160 KB: the whole LDS of a CU

struct GemmP2Params {
    const uint16_t* A;
    const uint16_t* A2;
    const uint16_t* W;
    unsigned a_bytes, a2_bytes, w_bytes;  // extents for the buffer descriptors
    unsigned lda_b, lda2_b, ldw_b;        // row strides in bytes
    const float* bias;
    const uint16_t* Rp;
    float* C32;
    uint16_t* Cp;
    uint16_t* Vt;
    int64_t ldc, ldr;
    int M, N, K, K1;
    int tiles_n, total, relu;
    float out_scale;
    float col_scale[3];
    int n_rows, heads;
    const int* EA;   // tile exponents (p2.h); null = all zero
    const int* EA2;
    const int* ER;
    int* EC;
    int* EVt;
    const float* AR;  // max |value| of the residual's 64 x 64 blocks (true units) - the bound that picks the output exponent
    float* AC;        // the same of the output (written when the output is a later residual: x)
    int eld_a, eld_a2, eld_r, eld_c;  // entries per 64-row block
    float bias_amax;
    unsigned* stats;  // [0]: number of output blocks that needed a non-zero exponent
    char* dummy;     // 4 KB: target of the stores of rows / columns beyond the matrix (a wave always issues all its stores)
    long long* dbg;  // E2EMV_STAMPS builds only: phase timestamps of two workgroups
};

// The lane index, recomputed where it is needed (two VALU instructions; volatile: not hoisted, not merged with other copies).
// Code that runs once per tile beside a K loop with no register to spare must not keep lane-derived constants alive across
// that loop: hipcc spilled them, and a scratch reload brings an s_waitcnt vmcnt(0) with it - in the middle of the counted
// load / store pipeline.
__device__ __forceinline__ int gp_lane_now() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// validates one GEMM and fills its parameter block (gemm_p2.hip)
int fill_gemm_p2_params(e2emv_ctx* ctx, const GemmP2Args& a, GemmP2Params& p);

// MFMAs of the pipelined K step as asm statements (order = source order; see compute_p below)
__device__ __forceinline__ void gp_mfma(p2_f32x16& c, p2_f16x8 a, p2_f16x8 b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void gp_mfma0(p2_f32x16& c, p2_f16x8 a, p2_f16x8 b) {  // zero C operand: the first product of a tile
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ unsigned gp_pk_mul(unsigned x, unsigned k) {  // two fp16 products (2^-11 w_hi)
    unsigned d;
    asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(d) : "v"(x), "v"(k));
    return d;
}

// ---- epilogue arithmetic, one instruction where hipcc needs several (the epilogue of a tile is ~1000 VALU instructions per
// wave with the matrix pipe idle: its instruction count is its time)
// max(a, |x|, |y|): fmaxf(fabsf()) compiles to a canonicalising v_max_f32 |x|, |x| per value in front of the maximum
__device__ __forceinline__ float gp_amax3(float a, float x, float y) {
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(a) : "v"(x), "v"(y));
    return a;
}
// ReLU of 8 values as x <- x + |x| = 2 max(x, 0), in place, skipped when `on` is 0 - ONE statement with the branch inside: a
// source-level `if` around in-place updates makes hipcc copy all 8 registers on both arms (12 v_mov per arm), a select
// costs 2 instructions per value.  NaN stays NaN (ReLU must not hide one: common.h relu_nan; -inf, which has no finite
// origin, turns NaN instead of 0); the factor 1/2 goes into the scale that follows (a power of two: exact)
__device__ __forceinline__ void gp_relu2x8(p2_f32x4& v0, p2_f32x4& v1, int on) {
    float a = v0[0], b = v0[1], c = v0[2], d = v0[3], e = v1[0], f = v1[1], g = v1[2], h = v1[3];
    asm("s_cmp_eq_u32 %8, 0\n\t"
        "s_cbranch_scc1 1f\n\t"
        "v_add_f32_e64 %0, %0, |%0|\n\t"
        "v_add_f32_e64 %1, %1, |%1|\n\t"
        "v_add_f32_e64 %2, %2, |%2|\n\t"
        "v_add_f32_e64 %3, %3, |%3|\n\t"
        "v_add_f32_e64 %4, %4, |%4|\n\t"
        "v_add_f32_e64 %5, %5, |%5|\n\t"
        "v_add_f32_e64 %6, %6, |%6|\n\t"
        "v_add_f32_e64 %7, %7, |%7|\n"
        "1:"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(on) : "scc");
    v0 = p2_f32x4{a, b, c, d};
    v1 = p2_f32x4{e, f, g, h};
}
// fp32 values of a pair of scaled-plane elements, hi + 2^-11 lo' (exact), straight from the packed halves
__device__ __forceinline__ p2_f32x2 gp_join_scaled(unsigned hi, unsigned lo) {
    float x0, x1;
    asm("v_fma_mix_f32 %0, %2, %4, %3 op_sel_hi:[1,0,1]\n\tv_fma_mix_f32 %1, %2, %4, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]"
        : "=&v"(x0), "=&v"(x1) : "v"(lo), "v"(hi), "s"(1.f / 2048.f));
    return {x0, x1};
}
// 8 values -> 4 + 4 packed plane words.  a = the values, b = 2048 a (scaled planes; both products of ONE fp32 value with
// powers of two) or b = a (plain planes, K = -1): hi = fp16(a), lo = fp16(b - K' hi) with K' = 2048 or 1
template <bool SCALED>
__device__ __forceinline__ void gp_split8(const p2_f32x4& a0, const p2_f32x4& a1, const p2_f32x4& b0, const p2_f32x4& b1, p2_u32x4& hi, p2_u32x4& lo) {
    const p2_f32x2 q0 = {a0[0], a0[1]}, q1 = {a0[2], a0[3]}, q2 = {a1[0], a1[1]}, q3 = {a1[2], a1[3]};
    const unsigned h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(q0, p2_f16x2)), h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(q1, p2_f16x2));
    const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(q2, p2_f16x2)), h3 = __builtin_bit_cast(unsigned, __builtin_convertvector(q3, p2_f16x2));
    unsigned l0, l1, l2, l3;
    // (the four low words first, their high halves behind them: no v_fma_mixhi reads the word the instruction before it wrote)
    asm("v_fma_mixlo_f16 %0, %4, %16, %8 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %5, %16, %10 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %2, %6, %16, %12 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %3, %7, %16, %14 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %4, %16, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %5, %16, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %2, %6, %16, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %3, %7, %16, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "s_nop 1"
        : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "v"(h0), "v"(h1), "v"(h2), "v"(h3), "v"(b0[0]), "v"(b0[1]), "v"(b0[2]), "v"(b0[3]), "v"(b1[0]), "v"(b1[1]), "v"(b1[2]), "v"(b1[3]),
          "s"(SCALED ? -2048.f : -1.f));
    hi = p2_u32x4{h0, h1, h2, h3};
    lo = p2_u32x4{l0, l1, l2, l3};
}

// ---- the same K step, software-pipelined inside the wave (the default; `compute` above is kept for the measurement
// build's ablations).  hipcc's schedule of `compute` reads a group's weight fragments right in front of its MFMAs and
// waits for them at once (lgkmcnt(0) behind the ds_reads): every group of 6 MFMAs opened with an exposed LDS round trip
// that only the SIMD's other wave could fill.  Here the 8 groups of a step (2 k-halves x 4 weight row blocks) run as one
// stream: the fragments of group g + 1 (and the activation fragments of the next k-half) are read under the MFMAs of
// group g, the four v_pk_mul_f16 that make 2^-11 w_hi sit behind the group's first MFMA; MFMAs and multiplies are asm
// statements (source order = machine order, fenced per slot), so the distances the hardware needs are kept by
// placement: multiplies -> the MFMA that reads them: one MFMA and two ds_reads apart.
// (acc[j][i]: weight row block j of the wave's 128 output columns x activation row block i of its 64 rows; wr / wc = the wave's
// row / column position in the 4 x 2 wave grid, l31 / lh = lane & 31 / lane >> 5)
// KDBG (measurement builds only): 1 = the fragment reads and multiplies without the MFMAs
template <bool first_step, int KDBG = 0>
__device__ __forceinline__ void gp_kstep(const char* smem, int buf, int wr, int wc, int l31, int lh, p2_f32x16 (&acc)[4][2]) {
    const int swz = (l31 >> 1) & 7;
    const char* xs = smem + buf * P2_BUFB + (wr * 64 + l31) * P2_ROWB;
    const char* ws = smem + buf * P2_BUFB + P2_TILEB + (wc * 128 + l31) * P2_ROWB;
    auto rd_x = [&](int ks, int t, int pl) { return *reinterpret_cast<const p2_f16x8*>(xs + t * 32 * P2_ROWB + (((4 * pl + 2 * ks + lh) ^ swz) << 4)); };
    auto rd_w = [&](int ks, int j, int pl) { return *reinterpret_cast<const p2_f16x8*>(ws + j * 32 * P2_ROWB + (((4 * pl + 2 * ks + lh) ^ swz) << 4)); };
    auto gp_mfma = [](p2_f32x16& c, p2_f16x8 a, p2_f16x8 b) {
        if constexpr (KDBG & 1) asm volatile("" :: "v"(a), "v"(b)); else e2emv::gp_mfma(c, a, b);
    };
    auto gp_mfma0 = [](p2_f32x16& c, p2_f16x8 a, p2_f16x8 b) {
        if constexpr (KDBG & 1) asm volatile("" :: "v"(a), "v"(b)); else e2emv::gp_mfma0(c, a, b);
    };
    p2_f16x8 xb[2][2][2];  // [k-half parity][row block][plane]
    p2_f16x8 wb[2][2];     // [group parity][plane]
    unsigned k2048 = 0x10001000u;  // two fp16 2^-11
    asm volatile("" : "+v"(k2048));
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) xb[0][t][pl] = rd_x(0, t, pl);
    wb[0][1] = rd_w(0, 0, 1);
    wb[0][0] = rd_w(0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int ks = g >> 2, j = g & 3, gp = g & 1;
        const p2_u32x4 wh = __builtin_bit_cast(p2_u32x4, wb[gp][0]);
        p2_u32x4 w2u;
        const bool z = first_step && ks == 0;
        // slot 0: x_hi w_lo of row block 0; then 2^-11 w_hi
        if (z) gp_mfma0(acc[j][0], wb[gp][1], xb[ks][0][0]); else gp_mfma(acc[j][0], wb[gp][1], xb[ks][0][0]);
#pragma unroll
        for (int e = 0; e < 4; ++e) w2u[e] = gp_pk_mul(wh[e], k2048);
        __builtin_amdgcn_sched_barrier(0);
        // slot 1: x_hi w_lo of row block 1; the next group's weight fragments
        if (z) gp_mfma0(acc[j][1], wb[gp][1], xb[ks][1][0]); else gp_mfma(acc[j][1], wb[gp][1], xb[ks][1][0]);
        if (g < 7) {
            wb[gp ^ 1][1] = rd_w((g + 1) >> 2, (g + 1) & 3, 1);
            wb[gp ^ 1][0] = rd_w((g + 1) >> 2, (g + 1) & 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const p2_f16x8 w2 = __builtin_bit_cast(p2_f16x8, w2u);
        // slots 2, 3: x_lo' (2^-11 w_hi); the next k-half's activation fragments behind them (groups 2 and 3)
        gp_mfma(acc[j][0], w2, xb[ks][0][1]);
        if (ks == 0 && j == 2) { xb[1][0][0] = rd_x(1, 0, 0); xb[1][1][0] = rd_x(1, 1, 0); }
        __builtin_amdgcn_sched_barrier(0);
        gp_mfma(acc[j][1], w2, xb[ks][1][1]);
        if (ks == 0 && j == 3) { xb[1][0][1] = rd_x(1, 0, 1); xb[1][1][1] = rd_x(1, 1, 1); }
        __builtin_amdgcn_sched_barrier(0);
        // slots 4, 5: x_hi w_hi
        gp_mfma(acc[j][0], wb[gp][0], xb[ks][0][0]);
        __builtin_amdgcn_sched_barrier(0);
        gp_mfma(acc[j][1], wb[gp][0], xb[ks][1][0]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- epilogue (see the header).  Each wave owns ONE slab of 32 rows x 32 floats behind the two tile buffers (16-byte
// chunk c of row r at position c ^ (r & 7): conflict-free b128 writes, 2-way reads).  The 8 blocks (32 rows x 32
// columns) of a wave are software-pipelined through registers: block b + 1 goes through the slab and its residual
// loads are issued while block b is finished (bias / ReLU / residual / split) and stored.  Two rules keep the store
// stream asynchronous (gfx950 retires loads AND stores in issue order on one counter, vmcnt):
//   * no load is issued behind a store whose completion we do not want to wait for: the bias is fetched once, up
//     front, and the residual of block b + 1 before the stores of block b;
//   * every wave issues EXACTLY 32 store instructions per tile (rows / columns beyond the matrix go to a dummy line
//     instead of being skipped), so the K loop of the next tile can wait with a COUNTED vmcnt for its operand loads,
//     which were issued before these stores, and leave the stores in flight (see the pipeline below).
// `p` = the parameter block of the GEMM this tile belongs to, (tm, tn) = the tile's row / column block; e_run = the exponent the
// accumulators carry, wave = the wave's index in the workgroup (scalar), ev = the tile's exponent fetch (lanes 32.. : the residual blocks' exponents / maxima)
// The last K step's asm MFMAs -> their first VALU readers in the epilogue: the accumulators pass through the statement, nothing
// that reads them moves above it.  Called ONCE in front of the epilogue(s): inside gp_epilogue, the chained kernel's four
// epilogue arms would each redefine all 128 accumulator registers and meet in 128 phi nodes behind the switch (hipcc then
// spilled hundreds of registers inside the K loop).
__device__ __forceinline__ void gp_acc_fence(p2_f32x16 (&acc)[4][2]) {
    asm volatile("s_nop 15" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
}

// RLDS (round 6; the chained kernel's MLP1, whose successor tile is hard-dependent: nothing runs ahead, both tile buffers are idle): the
// residual blocks arrive in LDS instead of in registers - a ring of four 4 KB block slots in the wave's own 16 KB of the tile buffers,
// filled by LDS-direct loads: blocks 0 - 3 go out at the head of the epilogue, IN FRONT of every store, block b + 4 when block b's slot has
// been read.  In the register form a block's residual load sits in the in-order memory queue behind the stores of the block before it and
// the wave cannot go on until it returns: eight store + load round trips per tile (MLP1's epilogue: 24.5 us against 10 for the others);
// here a load is waited for four blocks after it was issued, with a counted vmcnt that leaves everything younger in flight.
// (M a multiple of 256 and whole column tiles: the chained kernel's own conditions.)
template <int OUT, bool HAS_R, int DBG, bool RLDS = false>
__device__ __forceinline__ void gp_epilogue(const GemmP2Params& p, char* smem, const p2_f32x16 (&acc)[4][2], int wave, int tm, int tn, int e_run, int ev) {
    static_assert(!RLDS || (HAS_R && OUT == P2_OUT_PLANES), "the LDS residual ring belongs to the plane epilogue with a residual");
    // (the lane index is recomputed per tile: everything the epilogue derives from it - slab positions, store offsets, masks -
    // is then recomputed per tile, a few dozen integer instructions, instead of being hoisted out of the tile loop and held in
    // registers across the K loop, where there are none to spare)
    const int lane = gp_lane_now();
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    char* sl = smem + 2 * P2_BUFB + wave * P2_SLABB;
    const int o_r = lane >> 2, o_c = (lane & 3) * 8;  // row-contiguous view: 16 rows per pass, 8 columns per lane
    const int o_z = o_r & 7;
    const float cs = OUT == P2_OUT_QKV ? p.col_scale[min(tn, 2)] : 1.f;
    char* dummy = p.dummy + lane * 64;
    const float os = p.out_scale * p2_exp2i(e_run);  // the accumulators carry the exponent of the last K block
    // ---- tile exponents of the output: one per 64 columns of this wave's 64 rows, from an upper bound of the values
    const int erow = tm * 4 + wr;
    float rsc[2] = {1.f, 1.f};   // 2^e of the residual blocks
    float osc[2] = {1.f, 1.f};   // 2^-e of the output blocks
    float iosc[2] = {1.f, 1.f};  // 2^e
    float amx[2][2] = {{0.f, 0.f}, {0.f, 0.f}};  // max |final value| of the output blocks (this lane's share, two chains)
    if ((OUT != P2_OUT_F32 && (p.EC || p.EVt)) || (HAS_R && p.ER)) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const int cb = tn * 4 + wc * 2 + ch;  // 64-column block of the output
            // (the residual block's exponent and max |x| came with the tile's exponent fetch, lanes 32.. - no memory access here)
            int er = 0;
            float ar = 0.f;
            if (HAS_R && p.ER) {
                er = __builtin_amdgcn_readlane(ev, 32 + ch);
                ar = p.AR ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(ev, 34 + ch)) : 65536.f * p2_exp2i(er);
            }
            rsc[ch] = p2_exp2i(er);
            int* E = (OUT == P2_OUT_QKV && tn == 2) ? p.EVt : p.EC;
            if (OUT == P2_OUT_F32 || !E) continue;
            float am = 0.f, am1 = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int r = 0; r < 16; r += 4) {
                        am = gp_amax3(am, acc[2 * ch + jj][i][r], acc[2 * ch + jj][i][r + 1]);
                        am1 = gp_amax3(am1, acc[2 * ch + jj][i][r + 2], acc[2 * ch + jj][i][r + 3]);
                    }
            am = p2_wave_max(fmaxf(am, am1));
            const float bound = am * os + p.bias_amax + ar;
            const int e = p2_pick_exponent(bound * cs);
            osc[ch] = p2_exp2i(-e);
            iosc[ch] = p2_exp2i(e);
            const int ecb = (OUT == P2_OUT_QKV && tn == 2) ? wc * 2 + ch : cb;
            const int eld = (OUT == P2_OUT_QKV && tn == 2) ? 4 : p.eld_c;
            if (lane == 0 && erow * 64 < p.M && ecb < eld) {
                E[erow * eld + ecb] = e;
                if (e != 0 && p.stats) atomicAdd(p.stats, 1u);
            }
        }
    }
    // chunk (2 g + lh) ^ (l31 & 7) of row l31 = the address of chunk lh ^ (l31 & 7) with bit 5 / 6 flipped by g: ONE address
    // register, three v_xor per block (held as four addresses they were spilled in the chained kernel, and a scratch reload in
    // the epilogue waits for the residual loads and the stores in front of it)
    const unsigned sw0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sl + (unsigned)(l31 * 128 + ((lh ^ (l31 & 7)) << 4));
    auto slab_write = [&](int i, int j) {
        unsigned a = sw0;
        asm volatile("" : "+v"(a));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            p2_f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[j][i][4 * g + e];
            typedef __attribute__((address_space(3))) p2_f32x4* lds_f32x4_t;
            *reinterpret_cast<lds_f32x4_t>((uintptr_t)(a ^ (32u * g))) = v;
        }
    };
    if (OUT == P2_OUT_QKV && tn == 2) {
        // V^T: lane -> (dim d, 16-byte chunk q of the 32-key block) = 8 keys in accumulator order
        const int dl0 = lane >> 2, q = lane & 3;
        const int rb = 16 * (q >> 1) + 4 * (q & 1);
        // the bias of this lane's 8 dims (4 column blocks x 2 passes) BEFORE the first store: gfx950 retires loads and stores
        // in issue order, a load behind a store waits for that store (one bias load per pass drained the stores 16 times
        // per tile)
        float vb[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int n = tn * P2_BN + wc * 128 + j * 32 + dl0 + 16 * pass;
                vb[j][pass] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
            }
        const unsigned vt_rd0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)sl
                                + (unsigned)(rb * 128 + ((((dl0 >> 2) ^ (rb & 7)) << 4) | ((dl0 & 3) << 2)));
        const int64_t row2 = 2 * (int64_t)p.n_rows;  // halves per dim row of V^T
        uint16_t* const vt_lane = p.Vt + dl0 * row2 + q * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m0 = tm * P2_BM + wr * 64 + i * 32;
            const int img = m0 / p.n_rows, key0 = m0 - img * p.n_rows;
            const int64_t off_i = (int64_t)img * p.heads * 64 * row2 + (key0 >> 5) * 64;  // (wave-uniform)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                slab_write(i, j);
                // (v os + b) f with f = 16 x 2^-e: the power of two goes into both operands of ONE fma - the same value
                const float f = cs * osc[j >> 1], of = os * f;
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int dl = dl0 + 16 * pass;
                    // slab[key row][dim dl]: row rb + t (t = 0..3; + 8: immediate offset) holds it in chunk (dl >> 2) ^ (row & 7),
                    // i.e. address_t = (address_0 ^ 16 t) + 128 t, pass 1 = pass 0 ^ 64.  Recomputed from ONE register per pass
                    // (the empty asm keeps hipcc from hoisting 8 addresses out of the loops: they were spilled, and a
                    // scratch reload behind a store waits for that store)
                    unsigned a0 = vt_rd0 ^ (pass ? 64u : 0u);
                    asm volatile("" : "+v"(a0));
                    float v[8];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        typedef __attribute__((address_space(3))) const float* lds_f32_t;
                        const unsigned at = (a0 ^ (16u * t)) + 128u * t;
                        v[t] = *reinterpret_cast<lds_f32_t>((uintptr_t)at);
                        v[4 + t] = *reinterpret_cast<lds_f32_t>((uintptr_t)(at + 1024u));
                    }
                    const float bf = vb[j][pass] * f;
                    p2_u32x4 hi, lo;
                    p2_f32x4 w0, w1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { w0[e] = __builtin_fmaf(v[e], of, bf); w1[e] = __builtin_fmaf(v[4 + e], of, bf); }
                    gp_split8<false>(w0, w1, w0, w1, hi, lo);
                    const int dim = (wc * 2 + (j >> 1)) * 64 + (j & 1) * 32 + 16 * pass;  // (+ dl0: in vt_lane)
                    uint16_t* dst = vt_lane + off_i + dim * row2;
                    if (m0 >= p.M || tn * P2_BN + wc * 128 + j * 32 + dl >= p.N) dst = reinterpret_cast<uint16_t*>(dummy);
                    if (DBG & 64) { asm volatile("" :: "v"(hi), "v"(lo), "v"(dst)); continue; }          // measurement: no stores
                    if (DBG & 128) dst = p.Vt + ((dst - p.Vt) & ((1 << 19) - 1) & ~63ll);                 // measurement: 1 MB target
                    *reinterpret_cast<p2_u32x4*>(dst) = hi;
                    *reinterpret_cast<p2_u32x4*>(dst + 32) = lo;
                }
            }
        }
        return;
    }
    // bias: without a residual the epilogue issues NO load behind its first store (all four column blocks up front, 32
    // registers); with one, the bias of block b + 1 travels with its residual loads (the registers go to the residual)
    constexpr bool RREG = HAS_R && !RLDS;  // the residual through registers, a block ahead (gemm_p2; every chained tile but MLP1)
    constexpr int NB = RREG ? 2 : 4;
    p2_f32x4 bias8[NB][2];
    auto load_bias = [&](int slot, int j) {
        const int n = tn * P2_BN + wc * 128 + j * 32 + o_c;
        bias8[slot][0] = bias8[slot][1] = p2_f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias && n < p.N) bias8[slot][0] = *reinterpret_cast<const p2_f32x4*>(p.bias + n);
        if (p.bias && n + 4 < p.N) bias8[slot][1] = *reinterpret_cast<const p2_f32x4*>(p.bias + n + 4);
    };
    if (!RREG) {
#pragma unroll
        for (int j = 0; j < 4; ++j) load_bias(j, j);
    }
    // ---- RLDS: the ring.  Slot s = this wave's 16 KB of the tile buffers + 4 KB s; a block = 32 rows x 128 B (one 32-column plane block per
    // row), moved by 4 LDS-direct loads of 8 rows; chunk c of row r sits at position c ^ (((r >> 1) & 1) << 2) (the hi / lo halves of odd row
    // pairs swapped: conflict-free 16-byte reads in the row-contiguous view)
    char* const rl_base = smem + wave * 16384;
    __amdgpu_buffer_rsrc_t rsR;
    unsigned rl_vo = 0;
    if constexpr (RLDS) {
        rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.Rp), 0, (int)((unsigned)p.M * (unsigned)p.ldr * 4u), 0x00020000);
        const unsigned r8 = (unsigned)lane >> 3, pos = (unsigned)lane & 7u;
        rl_vo = r8 * (unsigned)p.ldr * 4u + ((pos ^ (((r8 >> 1) & 1u) << 2)) * 16u);
    }
    auto rl_issue = [&](int bb) {
        if constexpr (RLDS) {
            const int i = bb >> 2, j = bb & 3;
            const unsigned m0 = (unsigned)(tm * P2_BM + wr * 64 + i * 32);
            const unsigned so = m0 * (unsigned)p.ldr * 4u + (unsigned)(tn * 8 + wc * 4 + j) * 128u;
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) p2_glds16(rsR, rl_base + (bb & 3) * 4096 + q * 1024, rl_vo, so + (unsigned)q * 8u * (unsigned)p.ldr * 4u);
            asm volatile("" ::: "memory");
        }
    };
    if constexpr (RLDS) {  // (behind the bias loads, in front of everything else this epilogue puts into the memory queue)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) rl_issue(bb);
    }
    p2_f32x4 rv[2][2][2];              // [parity][pass][half]: the block in the row-contiguous view
    p2_u32x4 rr[RREG ? 2 : 1][2][2];   // [parity][pass][plane]: its residual
    // this lane's first row / column of the tile in the row-contiguous view; block (i, j), pass: row0 + 32 i + 16 pass,
    // columns col0 + 32 j .. + 7.  Addresses = one per-tile base + wave-uniform steps (per-store index arithmetic in 64 bits
    // was a tenth of the epilogue's instructions)
    const int row0 = tm * P2_BM + wr * 64 + o_r, col0 = tn * P2_BN + wc * 128 + o_c;
    const int rows_left = p.M - row0;  // row 32 i + 16 pass of the lane exists iff it is < rows_left
    const int64_t rstep = (OUT == P2_OUT_F32 ? 16 : 32) * p.ldc;  // 16 rows further, in elements of the output
    float* const c32_t = OUT == P2_OUT_F32 ? p.C32 + (int64_t)row0 * p.ldc + col0 : nullptr;
    uint16_t* const cp_t = OUT == P2_OUT_F32 ? nullptr : p.Cp + p2_index(row0, col0, p.ldc);
    // ReLU as x + |x| (gp_relu2): the factor 1/2 is folded into the plane scale where one follows directly
    const bool relu = OUT != P2_OUT_QKV && p.relu;
    const bool fold = relu && OUT == P2_OUT_PLANES && !HAS_R;
    const float unfold = relu && !fold ? 0.5f : 1.f;
    auto stage = [&](auto BB) {  // block b -> slab -> registers; residual loads issued
        constexpr int b = decltype(BB)::value;
        constexpr int i = b >> 2, j = b & 3;
        slab_write(i, j);
        if (RREG) load_bias(b & 1, j);
        const uint16_t* rcol = nullptr;
        if constexpr (RREG) {
            const int nc = min(col0 + j * 32, p.N - 8);
            rcol = p.Rp + ((nc >> 5) * 64 + (nc & 31));
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = o_r + 16 * pass;
            const int c0 = 2 * (lane & 3);
            rv[b & 1][pass][0] = *reinterpret_cast<const p2_f32x4*>(sl + r * 128 + ((c0 ^ o_z) << 4));
            rv[b & 1][pass][1] = *reinterpret_cast<const p2_f32x4*>(sl + r * 128 + (((c0 + 1) ^ o_z) << 4));
            if constexpr (RREG) {
                int m = min(row0 + i * 32 + 16 * pass, p.M - 1);
                if (DBG & 8192) m &= 255;  // measurement: the residual from an L2-resident slab
                const uint16_t* rp = rcol + (int64_t)m * (2 * p.ldr);
                rr[b & 1][pass][0] = *reinterpret_cast<const p2_u32x4*>(rp);
                rr[b & 1][pass][1] = *reinterpret_cast<const p2_u32x4*>(rp + 32);
            }
        }
    };
    auto finish = [&](auto BB) {
        constexpr int b = decltype(BB)::value;
        constexpr int i = b >> 2, j = b & 3;
        const bool col_ok = col0 + j * 32 < p.N;
        // plane scale of the block (wave-uniform): 2^-e of its tile exponent (x the column scale of q / k), x 1/2 behind x + |x|
        const float fa = OUT == P2_OUT_F32 ? 1.f : cs * osc[j >> 1] * (fold ? 0.5f : 1.f);
        // scaled planes: fa is a power of two and goes into the operands of the first fma (the same values; ReLU and the
        // residual sum commute with it) - the bias once per use of its registers, the accumulator scale as a uniform
        float osf = os, rsf = 1.f;
        if constexpr (HAS_R) rsf = rsc[j >> 1];
        // RLDS: this block's residual from its ring slot.  Issue order of the wave's memory queue: [bias] L0 L1 L2 L3 | L4 S0 | L5 S1 | L6 S2 | L7 S3 |
        // S4 | S5 | S6 | S7 (L = 4 loads, S = 4 stores): the wait for L_b leaves everything younger in flight
        p2_u32x4 rq[2][2];
        if constexpr (RLDS) {
            constexpr int CNT[8] = {12, 16, 20, 24, 28, 24, 20, 16};
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CNT[b]) : "memory");
            const unsigned sw = (((unsigned)o_r >> 1) & 1u) << 2;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const char* rp_ = rl_base + (b & 3) * 4096 + (o_r + 16 * pass) * 128;
                rq[pass][0] = *reinterpret_cast<const p2_u32x4*>(rp_ + ((((unsigned)lane & 3u) ^ sw) << 4));
                rq[pass][1] = *reinterpret_cast<const p2_u32x4*>(rp_ + (((4u + ((unsigned)lane & 3u)) ^ sw) << 4));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rq[0][0]), "+v"(rq[0][1]), "+v"(rq[1][0]), "+v"(rq[1][1]) :: "memory");  // in registers before the slot is refilled
            if constexpr (b < 4) rl_issue(b + 4);
        }
        if constexpr (OUT == P2_OUT_PLANES) {
            osf = os * fa;
            rsf *= fa;
            if (RREG || i == 0) { bias8[RREG ? (b & 1) : j][0] *= fa; bias8[RREG ? (b & 1) : j][1] *= fa; }
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const bool ok = col_ok && i * 32 + 16 * pass < rows_left;
            p2_f32x4 v0 = rv[b & 1][pass][0] * osf + bias8[RREG ? (b & 1) : j][0];
            p2_f32x4 v1 = rv[b & 1][pass][1] * osf + bias8[RREG ? (b & 1) : j][1];
            if constexpr (OUT != P2_OUT_QKV) {
                gp_relu2x8(v0, v1, p.relu);
                if constexpr (OUT == P2_OUT_F32 || HAS_R) { v0 *= unfold; v1 *= unfold; }  // (no plane scale to fold the 1/2 into)
            }
            if constexpr (HAS_R) {
                const p2_u32x4 rh = RLDS ? rq[pass][0] : rr[RREG ? (b & 1) : 0][pass][0], rl = RLDS ? rq[pass][1] : rr[RREG ? (b & 1) : 0][pass][1];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const p2_f32x2 a = gp_join_scaled(rh[e], rl[e]), c = gp_join_scaled(rh[2 + e], rl[2 + e]);
                    v0[2 * e] += a[0] * rsf; v0[2 * e + 1] += a[1] * rsf;
                    v1[2 * e] += c[0] * rsf; v1[2 * e + 1] += c[1] * rsf;
                }
            }
            if (OUT == P2_OUT_PLANES && p.AC && ok) {  // (of the values in the block's units, 2^-e: back to true units once, below)
                amx[j >> 1][0] = gp_amax3(amx[j >> 1][0], v0[0], v0[1]); amx[j >> 1][1] = gp_amax3(amx[j >> 1][1], v0[2], v0[3]);
                amx[j >> 1][0] = gp_amax3(amx[j >> 1][0], v1[0], v1[1]); amx[j >> 1][1] = gp_amax3(amx[j >> 1][1], v1[2], v1[3]);
            }
            if (OUT == P2_OUT_F32) {
                float* cp = c32_t + (2 * i + pass) * rstep + j * 32;
                float* cq = cp + 4;
                if (!ok) cp = reinterpret_cast<float*>(dummy);
                if (!(ok && col0 + j * 32 + 4 < p.N)) cq = reinterpret_cast<float*>(dummy + 16);
                *reinterpret_cast<p2_f32x4*>(cp) = v0;
                *reinterpret_cast<p2_f32x4*>(cq) = v1;
            } else {
                p2_u32x4 hi, lo;
                if (OUT == P2_OUT_QKV) {
                    const p2_f32x4 a0 = v0 * fa, a1 = v1 * fa;  // (q: not a power of two - its own product)
                    gp_split8<false>(a0, a1, a0, a1, hi, lo);
                } else {
                    const p2_f32x4 b0 = v0 * 2048.f, b1 = v1 * 2048.f;
                    gp_split8<true>(v0, v1, b0, b1, hi, lo);
                }
                uint16_t* cp = cp_t + (2 * i + pass) * rstep + j * 64;
                if (!ok) cp = reinterpret_cast<uint16_t*>(dummy);
                if (DBG & 64) { asm volatile("" :: "v"(hi), "v"(lo), "v"(cp)); continue; }          // measurement: no stores
                if (DBG & 128) cp = p.Cp + ((cp - p.Cp) & ((1 << 19) - 1) & ~63ll);                   // measurement: 1 MB target
                *reinterpret_cast<p2_u32x4*>(cp) = hi;
                *reinterpret_cast<p2_u32x4*>(cp + 32) = lo;
            }
        }
    };
#define P2_BLK(b) std::integral_constant<int, b>{}
    stage(P2_BLK(0));
    stage(P2_BLK(1)); finish(P2_BLK(0));
    stage(P2_BLK(2)); finish(P2_BLK(1));
    stage(P2_BLK(3)); finish(P2_BLK(2));
    stage(P2_BLK(4)); finish(P2_BLK(3));
    stage(P2_BLK(5)); finish(P2_BLK(4));
    stage(P2_BLK(6)); finish(P2_BLK(5));
    stage(P2_BLK(7)); finish(P2_BLK(6));
    finish(P2_BLK(7));
#undef P2_BLK
    if (OUT == P2_OUT_PLANES && p.AC) {
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            const float a = p2_wave_max(fmaxf(amx[ch][0], amx[ch][1])) * iosc[ch];
            const int cb = tn * 4 + wc * 2 + ch;
            if (lane == 0 && erow * 64 < p.M && cb < p.eld_c) p.AC[erow * p.eld_c + cb] = a;
        }
    }
}

}  // namespace e2emv
