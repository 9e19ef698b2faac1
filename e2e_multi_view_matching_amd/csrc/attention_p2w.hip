// Multi-head attention on plane operands, ONE wave per SIMD:  out = softmax(q k^T / sqrt(64)) v
//
// Same operands, arithmetic and output as attention_p2.hip (three fp16 plane products per block on
// v_mfma_f32_32x32x16_f16, S^T = mfma(K, Q), O^T += mfma(V^T, P^T), P split in the score registers' layout, tile exponents).
// What differs is who overlaps the matrix pipe with the softmax arithmetic.  attention_p2 leaves it to the two waves of a
// SIMD; measured, the SIMD runs their matrix and vector phases at the SUM of the two.  Here a workgroup is 4
// waves = one per SIMD, 512 registers each, a wave owns 64 queries as two STREAMS of 32 (a = 0, 1), and the overlap is
// written into the instruction stream: per 64-key tile j a wave runs two segments of 48 MFMAs,
//
//     segment X(j):  matrix pipe  P_0(j) V(j) -> O_0,   K(j+1) Q_0 -> S_0(j+1)      vector pipe  softmax of S_1(j)   -> P_1(j)
//     segment Y(j):  matrix pipe  P_1(j) V(j) -> O_1,   K(j+1) Q_1 -> S_1(j+1)      vector pipe  softmax of S_0(j+1) -> P_0(j+1)
//
// one MFMA per slot with three vector instructions of the OTHER stream's softmax, one fragment read and (now and then) one
// LDS-direct load behind it, fenced by sched_barrier so that the order of the source is the order of the machine code.
// The softmax of a tile has no row maximum on its fast path: the numerators are formed against the running maximum of the
// stream, their sum is checked at the end of the segment (every p <= sum < 60000 stays inside fp16), and only a tile
// that fails the check - the first one, and one whose scores outgrow the running maximum by more than ~2^5 - is redone on
// the slow path (true maximum, O and l rescaled).  O and l are exact in either case: the scale cancels.
//
// LDS: three regions of 32 KB, region j % 3 = [ V(j) | K(j+1) ], filled by LDS-direct loads TWO tiles ahead (issued in
// segment X(j-2), awaited with a counted vmcnt at the one barrier of a tile, slot 36 of segment Y(j-1): the workgroups of
// an (image, head) walk its keys in lock step, so every tile is an L2 miss for all of them - one tile of lead did not cover
// it); fragment reads run two groups of 6 MFMAs ahead, across segment and tile boundaries.
#include <algorithm>
#include <type_traits>
#include <vector>

#include "attention_p2.h"

namespace e2emv {

typedef __attribute__((ext_vector_type(16))) float aw_f32x16;

constexpr int AW_QT = 256;              // queries per workgroup
constexpr int AW_TILEB = 64 * 256;      // one operand tile: 64 rows x 256 B
constexpr int AW_REGB = 2 * AW_TILEB;   // region: V(j) | K(j+1)
constexpr int AW_PART_F = 68;           // floats of a key-split part's record per query: O[64], m, l (+ 2: 16-byte rows)
// Softmax numerators carry the factor 2^AW_PLOG against the stream's running maximum.  7 (attention_p2: 10) leaves the fast
// path 8.9 bits of growth of a row's maximum before its sum check sends the tile through the slow path - measured with 10:
// a third of the tiles of a 1024-key problem had SOME query of the workgroup beyond 5.9 bits, and a slow path costs every
// wave of the workgroup ~1500 cycles at the tile's barrier.  Numerators below 2^-10 of a row's maximum lose bits of their
// low plane (fp16 subnormals) - 2^-21 of the row sum at worst.
constexpr float AW_SINV = 1.f / P2_QS, AW_PLOG = 7.f, AW_LIMIT = 60000.f;

__device__ __forceinline__ void aw_wait_q(p2_f16x8 (&q)[2][2][4]) {
    asm volatile("s_waitcnt vmcnt(16)"
                 : "+a"(q[0][0][0]), "+a"(q[0][0][1]), "+a"(q[0][0][2]), "+a"(q[0][0][3]), "+a"(q[0][1][0]), "+a"(q[0][1][1]), "+a"(q[0][1][2]), "+a"(q[0][1][3]),
                   "+a"(q[1][0][0]), "+a"(q[1][0][1]), "+a"(q[1][0][2]), "+a"(q[1][0][3]), "+a"(q[1][1][0]), "+a"(q[1][1][1]), "+a"(q[1][1][2]), "+a"(q[1][1][3])
                 :: "memory");
}

template <int I, int N, class F>
__device__ __forceinline__ void aw_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        aw_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ unsigned aw_cvt_pk(float v0, float v1) {
    const p2_f32x2 vv = {v0, v1};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(vv, p2_f16x2));
}
// low plane of a pair: fp16(v - hi), the exact fp32 residual rounded once (no trailing wait state: the consumer is an MFMA of
// the NEXT segment)
// The vector instructions of the fast-path softmax are asm volatile like the MFMAs: one statement = one instruction, in the
// order of the source (left to itself hipcc packs the row sums into dependent v_pk_add_f32 chains and moves them together).
// Hazards that hipcc would pad and an asm statement must keep by placement (segment code below): a v_exp_f32 result is
// read no sooner than two instructions later; v_fma_mixhi_f16 (which keeps the low half of its destination) comes a slot
// after the v_fma_mixlo_f16 that wrote it.
__device__ __forceinline__ float aw_fma(float a, float b, float c) {
    float d;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float aw_exp2(float a) {
    float d;
    asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(a));
    return d;
}
__device__ __forceinline__ void aw_acc(float& sum, float a) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(a)); }
__device__ __forceinline__ unsigned aw_cvt_pk_v(float v0, float v1) {
    unsigned d;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(v0), "v"(v1));
    return d;
}
__device__ __forceinline__ unsigned aw_mixlo(unsigned hi, float v0) {  // the two halves of aw_lo as separate instructions
    unsigned lo;
    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(v0));
    return lo;
}
__device__ __forceinline__ unsigned aw_mixhi(unsigned lo, unsigned hi, float v1) {
    asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(v1));
    return lo;
}
__device__ __forceinline__ unsigned aw_lo(unsigned hi, float v0, float v1) {
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(lo) : "v"(hi), "v"(v0), "v"(v1));
    return lo;
}

// The MFMAs are inline asm so that the register FILE of every operand is ours: score accumulators in VGPRs (the softmax reads
// them with VALU instructions), O accumulators and the Q fragments resident in the accumulator file (left to itself hipcc
// keeps the scores there too and spills Q into it, one v_accvgpr_read per use).  hipcc neither schedules nor pads an asm
// statement: the distances MFMA result -> VALU reader are kept by construction (comments at the segment boundaries).
__device__ __forceinline__ void aw_mfma_s(aw_f32x16& c, p2_f16x8 a, p2_f16x8 b) {   // S += K Q^T: C in VGPRs, B (Q) in AGPRs
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
}
__device__ __forceinline__ void aw_mfma_s0(aw_f32x16& c, p2_f16x8 a, p2_f16x8 b) {  // S = K Q^T (zero C operand)
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(b));
}
__device__ __forceinline__ void aw_mfma_o(aw_f32x16& c, p2_f16x8 a, p2_u32x4 b) {   // O += V^T P^T: C in AGPRs
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// the 8 fragments of a query row (2 planes x 4 steps of 16 dims), loaded from global memory straight INTO the accumulator
// file, where they stay.  An asm load is not counted by hipcc: aw_wait_q, which names every destination, is the wait (it
// sits behind the first tiles' LDS-direct loads: one latency for all of the prologue's loads)
__device__ __forceinline__ void aw_load_q(const char* qp, p2_f16x8 (&q)[2][4]) {
    asm volatile(
        "global_load_dwordx4 %0, %8, off\n\t"
        "global_load_dwordx4 %1, %8, off offset:32\n\t"
        "global_load_dwordx4 %2, %8, off offset:128\n\t"
        "global_load_dwordx4 %3, %8, off offset:160\n\t"
        "global_load_dwordx4 %4, %8, off offset:64\n\t"
        "global_load_dwordx4 %5, %8, off offset:96\n\t"
        "global_load_dwordx4 %6, %8, off offset:192\n\t"
        "global_load_dwordx4 %7, %8, off offset:224"
        : "=&a"(q[0][0]), "=&a"(q[0][1]), "=&a"(q[0][2]), "=&a"(q[0][3]), "=&a"(q[1][0]), "=&a"(q[1][1]), "=&a"(q[1][2]), "=&a"(q[1][3])
        : "v"(qp) : "memory");
}

// ABL (measurement only, wrong results): 1 no softmax arithmetic in the slots, 2 no MFMAs, 4 no fragment reads, 8 no LDS-direct
// loads, 16 no barrier
// MULTI: more than one source image per query image (cross layers of tuples with T > 2); without it the tile walk is a counter
// PART: the instantiation that walks the key-split PARTS of a launch's leftover items (its own launch behind the whole items': the
// whole-item kernel stays the code it was - carrying the part logic as run-time flags cost it 6 % at configs[1])
template <bool HAS_E, int ABL = 0, bool MULTI = true, bool PART = false>
__global__ __launch_bounds__(256, 1) void attention_p2w_kernel(AttnP2Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem_aw[];

    // PART: items from p.n_full on are walked in PARTS (key split, round 6): when the items do not fill the last round of
    // workgroups (T = 5: 640 items on 256 CUs - 3 rounds for 2.5), each of the r leftover items is walked by n_split workgroups, part
    // ks taking tiles [ks T / n_split, (ks + 1) T / n_split) of the item's T key tiles and leaving (m, l, O) of its keys in p.part for
    // attention_p2w_combine.  A part keeps its item's place in the XCD interleave (its keys are the L2 working set of that XCD).
    int lin = blockIdx.x, ks = 0;
    constexpr bool part = PART;
    if constexpr (PART) {  // (grid = leftover items x n_split)
        const int h = blockIdx.x, tpx = (8 * p.gper * p.nq - p.n_full) >> 3;  // leftover items per XCD
        ks = (h >> 3) / tpx;
        lin = p.n_full + 8 * ((h >> 3) % tpx) + (h & 7);
    }
    const int xcd = lin & 7, idx = lin >> 3;
    const int g = xcd * p.gper + idx / p.nq;
    if (g >= p.groups) return;
    const int qt = idx % p.nq;
    const int img = g / p.H, head = g % p.H;
    const int b = img / p.T, t = img % p.T;
    if (qt * AW_QT >= p.nv[t]) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned row_b = 8u * (unsigned)p.D;
    int n_stamp = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr ((ABL & 32) != 0) {
            const long long tnow = __builtin_amdgcn_s_memtime();
            if (p.dbg && lane == 0 && n_stamp < 40 && (blockIdx.x == 0 || blockIdx.x == 517)) p.dbg[((blockIdx.x ? 1 : 0) * 4 + wave) * 40 + n_stamp] = tnow;
            ++n_stamp;
        }
    };
    stamp();  // 0: start

    // ---- Q fragments of the two streams (B operand): lane (query l31, lh) holds d = 16 s + 8 lh .. + 7 of both planes
    int q_row[2];
    bool q_ok[2];
    p2_f16x8 Qf[2][2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        q_row[a] = qt * AW_QT + wave * 64 + a * 32 + l31;
        q_ok[a] = q_row[a] < p.n_rows;
        const char* qp = reinterpret_cast<const char*>(p.qk) + ((int64_t)img * p.n_rows + (q_ok[a] ? q_row[a] : p.n_rows - 1)) * row_b + head * 256;
        aw_load_q(qp + lh * 16, Qf[a]);  // chunk ((s >> 1) * 8 + pl * 4 + 2 * (s & 1) + lh) of the row's 16
    }
    stamp();  // 1: Q loads issued
    const int q_blk = __builtin_amdgcn_readfirstlane(((int64_t)img * p.n_rows + min(qt * AW_QT + wave * 64, p.n_rows - 1)) >> 6);
    const int e_q = HAS_E ? p.EQK[q_blk * 8 + head] : 0;

    // ---- the tiles of this image's sources, in order.  A cursor = (source, tile in it, that source's keypoint count); the
    // count is re-read from the kernel arguments only when a cursor moves to the next source
    const int n_src = p.cross ? p.T - 1 : 1;
    auto src_t = [&](int si) __attribute__((always_inline)) { return !p.cross ? t : (si < t ? si : si + 1); };
    int n_tiles = 0;
    for (int si = 0; si < n_src; ++si) n_tiles += (p.nv[src_t(si)] + 63) / 64;
    int t0 = 0;  // first tile of this workgroup's share (a whole item: all of them)
    if constexpr (PART) {
        t0 = ks * n_tiles / p.n_split;
        n_tiles = (ks + 1) * n_tiles / p.n_split - t0;
        if (n_tiles <= 0) return;  // (fewer tiles than parts: the combine pass skips the part the same way)
    }
    struct Cur { int si, kt, nv; unsigned kb, vb; int eb; };  // kb / vb: byte offsets of the source's first K / V^T tile, eb: its first 64-row block
    auto cur_src = [&](Cur& c) __attribute__((always_inline)) {
        const int im = b * p.T + src_t(c.si);
        c.kt = 0;
        c.nv = p.nv[src_t(c.si)];
        c.kb = (unsigned)(im * p.n_rows) * row_b;
        c.vb = (unsigned)((im * p.H + head) * 64) * (unsigned)p.n_rows * 4u;
        c.eb = (im * p.n_rows) >> 6;
    };
    auto cur_next = [&](Cur& c) __attribute__((always_inline)) {
        ++c.kt;
        if constexpr (MULTI) {
            if (c.kt * 64 >= c.nv && c.si + 1 < n_src) { ++c.si; cur_src(c); }
        }
    };
    auto cur_init = [&]() __attribute__((always_inline)) {
        Cur c;
        c.si = 0;
        cur_src(c);
        if constexpr (PART)
            for (int i = 0; i < t0; ++i) cur_next(c);  // (a part starts at its first tile)
        return c;
    };

    // ---- loader: one piece = 4 rows x 256 B; lane -> (row lane >> 4, LDS position lane & 15), source chunk = position ^ (row & 15)
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.qk), 0, (int)p.qk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.vt), 0, (int)p.vt_bytes, 0x00020000);
    unsigned k_vo[4], v_vo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * (wave * 4 + i) + (lane >> 4);
        const unsigned c = (unsigned)((lane & 15) ^ (row & 15));
        k_vo[i] = (unsigned)row * row_b + 4u * (unsigned)p.D + (unsigned)head * 256u + c * 16u;
        v_vo[i] = (unsigned)row * (unsigned)p.n_rows * 4u + c * 16u;
    }
    auto k_so = [&](const Cur& c) __attribute__((always_inline)) { return c.kb + (unsigned)c.kt * 64u * row_b; };
    auto v_so = [&](const Cur& c) __attribute__((always_inline)) { return c.vb + (unsigned)c.kt * 256u; };
    auto k_piece = [&](int region, int i, unsigned so) __attribute__((always_inline)) { p2_glds16(rsK, smem_aw + region * AW_REGB + AW_TILEB + (wave * 4 + i) * 1024, k_vo[i], so); };
    auto v_piece = [&](int region, int i, unsigned so) __attribute__((always_inline)) { p2_glds16(rsV, smem_aw + region * AW_REGB + (wave * 4 + i) * 1024, v_vo[i], so); };
    // tile exponents travel in lane 0 of a VGPR, fetched right before the tile's pieces
    // Tile exponents (k's and v's of every 64-key tile this workgroup walks): copied into LDS once, in the prologue, and read
    // from there as {e_k, e_v} pairs a segment ahead of their use.  (Fetched from global memory inside the loop they sat on the
    // same in-order counter as the tiles' pieces, and every use - or copy - of one drained the loads behind it.)
    // (loaded first of all - the oldest operations of the prologue's one queue - and written to LDS behind the tiles' pieces)
    int* const els = reinterpret_cast<int*>(smem_aw + 3 * AW_REGB);
    int ek_mine = 0, ev_mine = 0;
    if (HAS_E && tid < n_tiles) {  // (n_tiles <= 7 sources x 32 tiles)
        int si = 0, kt = PART ? tid + t0 : tid;
        if constexpr (MULTI) {
            for (; si + 1 < n_src; ++si) {
                const int nt = (p.nv[src_t(si)] + 63) / 64;
                if (kt < nt) break;
                kt -= nt;
            }
        }
        const int blk = (((b * p.T + src_t(si)) * p.n_rows) >> 6) + kt;
        ek_mine = p.EQK[blk * 8 + 4 + head];
        if (p.EVt) ev_mine = p.EVt[blk * 4 + head];
    }
    typedef __attribute__((address_space(3))) const p2_u32x2* lds_pair_t;
    const unsigned els0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_aw + 3u * AW_REGB;
    auto read_e = [&](int tl) __attribute__((always_inline)) {  // {e_k, e_v} of tile tl (every lane the same address)
        return *reinterpret_cast<lds_pair_t>((uintptr_t)(els0 + 8u * (unsigned)min(tl, n_tiles - 1)));
    };

    // ---- fragment addressing: R[s][pl] = region base + row l31 + swizzled chunk of (16-column step s, plane pl); the row
    // block (32 rows = 8 KB) and V | K go into the instruction's immediate offset
    const int kz = l31 & 15;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_aw;
    unsigned R[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
            R[s][pl] = lds0 + (unsigned)(l31 * 256 + ((((s >> 1) * 8 + pl * 4 + 2 * (s & 1) + lh) ^ kz) << 4));
    // (R holds LDS addresses, not offsets from smem_aw: no address arithmetic left beside the read)
    typedef __attribute__((address_space(3))) const p2_f16x8* lds_frag_t;
    auto frag = [&](int s, int pl, int imm) __attribute__((always_inline)) { return *reinterpret_cast<lds_frag_t>((uintptr_t)(R[s][pl] + (unsigned)imm)); };

    constexpr int PA[3] = {1, 0, 0};  // plane of the A operand (K or V^T), smallest terms first
    constexpr int PB[3] = {0, 1, 0};  // plane of the B operand (Q or P)

    aw_f32x16 O[2][2], S[2][2];
    p2_u32x4 Pf[2][2][4];  // [stream][plane][16-key step]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[a][d][r] = 0.f;
    float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};
    int e_o = 0;
    bool o_started = false;
    unsigned n_slow = 0;  // softmaxes this wave redid on the slow path (one atomic per wave behind the epilogue: an atomic per
                          // event sat in the loop's in-order memory queue - 8192 of them on one address cost a ragged launch 45 us)

    // ---- slow path of a tile's softmax (first tile; a tile whose fast-path sum left fp16's range): true row maximum, O and l
    // of the stream rescaled, numerators against the new maximum
    auto sm_slow = [&](auto BB, float sinv, int valid) __attribute__((always_inline)) {
        constexpr int B = decltype(BB)::value;
        asm volatile("s_nop 15" : "+v"(S[B][0]), "+v"(S[B][1]), "+a"(O[B][0]), "+a"(O[B][1]));  // (MFMA results -> VALU; nothing of this path moves above it)
        if (valid < 64) {  // keys beyond the source's keypoint count
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh >= valid) S[B][kb][r] = -INFINITY;
        }
        float mx = S[B][0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[B][kb][r]);
        {
            float x = mx, y = mx;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
            mx = fmaxf(x, y);
        }
        mx *= sinv;
        const float m_new = fmaxf(m_run[B], mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run[B] - m_new);
#pragma unroll
        for (int r = 0; r < 16; ++r) { O[B][0][r] *= alpha; O[B][1][r] *= alpha; }
        const float e0 = AW_PLOG - m_new;
        float ps = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int kb = k >> 3, r = 2 * (k & 7), u = 2 * kb + ((k & 7) >> 2), e = k & 3;
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[B][kb][r], sinv, e0));
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[B][kb][r + 1], sinv, e0));
            ps += p0 + p1;
            const P2Pair pr = p2_split_plain(p0, p1);
            Pf[B][0][u][e] = pr.hi; Pf[B][1][u][e] = pr.lo;
        }
        l_run[B] = l_run[B] * alpha + ps;
        m_run[B] = m_new;
        asm volatile("s_nop 4" : "+a"(O[B][0]), "+a"(O[B][1]));  // (v_accvgpr_write of the rescaled O -> the next MFMA's C operand)
    };

    // ---- prologue: K(0) into the K half of region 2 (the place of "tile -1"), V(0) | K(1) into region 0, V(1) | K(2) into
    // region 1.  Past the last tile a cursor stays where it is: the tile is loaded once more, into a place nobody reads (no
    // branches around the pieces, and the counted wait of the loop always sees the same number of operations).
    Cur cK = cur_init(), cV = cur_init(), cS = cur_init();
    int nK = 0, nV = 0;  // tiles issued so far
    auto adv_k = [&]() __attribute__((always_inline)) { if (++nK < n_tiles) cur_next(cK); };
    auto adv_v = [&]() __attribute__((always_inline)) { if (++nV < n_tiles) cur_next(cV); };
    {
        const unsigned so = k_so(cK);
#pragma unroll
        for (int i = 0; i < 4; ++i) k_piece(2, i, so);
        adv_k();
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const unsigned sov = v_so(cV), sok = k_so(cK);
#pragma unroll
        for (int i = 0; i < 4; ++i) v_piece(r, i, sov);
#pragma unroll
        for (int i = 0; i < 4; ++i) k_piece(r, i, sok);
        adv_v();
        adv_k();
    }
    if (HAS_E && tid < n_tiles) { els[2 * tid] = ek_mine; els[2 * tid + 1] = ev_mine; }
    // one queue, in issue order: exponents, Q fragments, K(0) | V(0), K(1) | V(1), K(2).  The first scores need the first three:
    // the 16 youngest pieces stay in flight (every workgroup of the chip starts at the same time: the prologue's loads are a
    // 37 MB burst, its tail lands under the first scores and the first softmax)
    if constexpr ((ABL & 256) != 0) {  // measurement (wrong results): what the prologue's wait for Q / K(0) costs a launch
        asm volatile("s_waitcnt vmcnt(63)" : "+a"(Qf[0][0][0]), "+a"(Qf[0][0][1]), "+a"(Qf[0][0][2]), "+a"(Qf[0][0][3]), "+a"(Qf[0][1][0]), "+a"(Qf[0][1][1]), "+a"(Qf[0][1][2]), "+a"(Qf[0][1][3]),
                     "+a"(Qf[1][0][0]), "+a"(Qf[1][0][1]), "+a"(Qf[1][0][2]), "+a"(Qf[1][0][3]), "+a"(Qf[1][1][0]), "+a"(Qf[1][1][1]), "+a"(Qf[1][1][2]), "+a"(Qf[1][1][3]) :: "memory");
    } else {
        aw_wait_q(Qf);
    }
    __syncthreads();
    p2_u32x2 e_pair = {0u, 0u};  // {e_k, e_v} of the tile whose exponents are needed next (all lanes equal)
    if constexpr (HAS_E) e_pair = read_e(0);
    stamp();  // 2: Q and the first tiles landed
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) R[s][pl] += 2 * AW_REGB;  // region 2
    {
        const float sinv = AW_SINV * p2_exp2i(e_q + __builtin_amdgcn_readfirstlane((int)e_pair[0]));
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                p2_f16x8 kf[2][2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) kf[kb][pl] = frag(s, pl, AW_TILEB + kb * 8192);
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        if (s == 0 && q == 0) aw_mfma_s0(S[a][kb], kf[kb][PA[q]], Qf[a][PB[q]][s]);
                        else aw_mfma_s(S[a][kb], kf[kb][PA[q]], Qf[a][PB[q]][s]);
                    }
            }
        // K(0) sits where X(0) puts K(3): everybody is through with it before anybody's pieces go out; and
        // V(0) | K(1) have landed, everybody's (8 pieces - V(1) | K(2) - may still be in flight: the barrier of tile 0 waits for them)
        if constexpr ((ABL & 256) != 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        sm_slow(std::integral_constant<int, 0>{}, sinv, cS.nv - cS.kt * 64);
        // stream 1's first tile takes the fast path in X(0): its running maximum starts at the tile's row maxima (a ragged
        // first tile is left to the slow path, which masks)
        if (cS.nv - cS.kt * 64 >= 64) {
            asm volatile("s_nop 15" : "+v"(S[1][0]), "+v"(S[1][1]));
            float mx = S[1][0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[1][kb][r]);
            float x = mx, y = mx;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
            m_run[1] = fmaxf(x, y) * sinv;
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) R[s][pl] -= 2 * AW_REGB;  // region 0
    int reg_cur = 0, seg_tile = 0;
    stamp();  // 3: prologue scores + slow softmax done

    // fragment registers: group g of a segment (6 MFMAs: one 16-key step of V^T, or one 16-dim step of K) uses F[g & 3]; the
    // reads run TWO groups (12 MFMAs) ahead of their use - one group ahead the matrix pipe waited for LDS (measured)
    p2_f16x8 F[4][4];  // [buffer][2 x + pl], x = row block (dims 0-31 / 32-63 of V^T, keys 0-31 / 32-63 of K)
    // read number n (0..3) of a group, in the order the group's MFMAs need them: (x0, lo) (x1, lo) (x0, hi) (x1, hi)
    auto read_frag = [&](int buf, int n, int s, int kpart) __attribute__((always_inline)) {
        const int x = n & 1, pl = n < 2 ? 1 : 0;
        F[buf][2 * x + pl] = frag(s, pl, kpart * AW_TILEB + x * 8192);
    };
    // groups 0 and 1 of segment X(0): V(0), 16-key steps 0 and 1
#pragma unroll
    for (int n = 0; n < 4; ++n) { read_frag(0, n, 0, 0); read_frag(1, n, 1, 0); }

    // ---- one segment.  A = the stream on the matrix pipe, B = 1 - A the stream whose softmax runs beside it; LAST = the last
    // tile (no K(j+1) left: 24 MFMAs; segment Y then has no softmax either)
    auto seg = [&](auto AA, auto LL) __attribute__((always_inline)) {
        constexpr int A = decltype(AA)::value, B = 1 - A;
        constexpr bool LAST = decltype(LL)::value;
        constexpr bool HAS_SM = !(LAST && A == 1);
        constexpr int NSLOT = LAST ? 24 : 48, NG = NSLOT / 6;
        constexpr int MPS = LAST ? 2 : 1;  // softmax micro-steps per slot

        if (A == 0) {
            // V(j)'s exponent: both streams' O accumulators live at the exponent of the current V tile
            if (HAS_E && p.EVt) {
                const int e_v = __builtin_amdgcn_readfirstlane((int)e_pair[1]);
                if (__builtin_expect(o_started && e_v != e_o, 0)) {
                    const float f = e_o - e_v < -126 ? 0.f : p2_exp2i(e_o - e_v);
                    asm volatile("s_nop 15" : "+a"(O[0][0]), "+a"(O[0][1]), "+a"(O[1][0]), "+a"(O[1][1]));
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { O[a][0][r] *= f; O[a][1][r] *= f; }
                    asm volatile("s_nop 4" : "+a"(O[0][0]), "+a"(O[0][1]), "+a"(O[1][0]), "+a"(O[1][1]));
                }
                e_o = e_v;
                o_started = true;
            }
        }
        // the tile whose softmax runs here: S_1(j) in X (cS at tile j), S_0(j+1) in Y (cS at tile j + 1)
        float sinv = AW_SINV, e0 = 0.f, ps = 0.f;
        int valid = 64;
        if (HAS_SM) {
            // X(j): e_pair = tile j (read in X(j-1)); then tile j + 1 is read, for Y(j) and for X(j+1)
            if (HAS_E) sinv = AW_SINV * p2_exp2i(e_q + __builtin_amdgcn_readfirstlane((int)e_pair[0]));
            if (HAS_E && A == 0 && !LAST) e_pair = read_e(seg_tile + 1);
            e0 = AW_PLOG - m_run[B];
            valid = cS.nv - cS.kt * 64;
        }
        // a ragged tile (the last one of a source with nv % 64 != 0): scores of the keys beyond nv to -inf in front of the
        // fast path (p = 0; the S of this stream are a segment old: no MFMA in flight writes them)
        if (HAS_SM && !(ABL & 1) && __builtin_expect(valid < 64, 0)) {
            asm volatile("s_nop 7" : "+v"(S[B][0]), "+v"(S[B][1]));
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh >= valid) S[B][kb][r] = -INFINITY;
            asm volatile("" : "+v"(S[B][0]), "+v"(S[B][1]));
        }
        // fast-path softmax of the tile, 3 instructions per slot, ordered so that NO instruction reads the result of one of the
        // two before it (one wave per SIMD: a dependent pair stalls the issue, nobody fills the gap) - pair k = scores 2k, 2k + 1:
        //   slot 3k      x0 = S sinv + e0            lo(k-1).lo = p0 - hi     x1 = S sinv + e0
        //   slot 3k + 1  p0 = exp2 x0                lo(k-1).hi = p1 - hi     p1 = exp2 x1
        //   slot 3k + 2  psA += p0                   hi(k) = (fp16 p0, p1)    psB += p1
        bool slow = false;
        float xa[16][2], pa[16][2], psA = 0.f, psB = 0.f, sinv_v = sinv;
        asm volatile("" : "+v"(sinv_v), "+v"(e0));
        unsigned hi_k[16], lo_k[16];
        auto micro = [&](auto MM) __attribute__((always_inline)) {
            constexpr int m = decltype(MM)::value;
            constexpr int k = m / 3, ph = m % 3;
            constexpr int kb = (k & 15) >> 3, r = 2 * (k & 7);
            if constexpr (m == 48) {         // the last pair's low plane, first half
                lo_k[15] = aw_mixlo(hi_k[15], pa[15][0]);
            } else if constexpr (m == 49) {  // ... second half
                Pf[B][1][3][3] = aw_mixhi(lo_k[15], hi_k[15], pa[15][1]);
            } else if constexpr (ph == 0) {
                xa[k][0] = aw_fma((ABL & 64) ? e0 : S[B][kb][r], sinv_v, e0);  // (ABL 64: the softmax does not read MFMA results)
                if constexpr (k > 0) lo_k[k - 1] = aw_mixlo(hi_k[k - 1], pa[k - 1][0]);
                xa[k][1] = aw_fma((ABL & 64) ? e0 : S[B][kb][r + 1], sinv_v, e0);
            } else if constexpr (ph == 1) {
                pa[k][0] = aw_exp2(xa[k][0]);
                if constexpr (k > 0) {
                    constexpr int k1 = k - 1, u = 2 * (k1 >> 3) + ((k1 & 7) >> 2), e = k1 & 3;
                    Pf[B][1][u][e] = aw_mixhi(lo_k[k1], hi_k[k1], pa[k1][1]);
                }
                pa[k][1] = aw_exp2(xa[k][1]);
            } else {
                constexpr int u = 2 * (k >> 3) + ((k & 7) >> 2), e = k & 3;
                aw_acc(psA, pa[k][0]);
                hi_k[k] = aw_cvt_pk_v(pa[k][0], pa[k][1]);
                Pf[B][0][u][e] = hi_k[k];
                aw_acc(psB, pa[k][1]);
            }
        };

        unsigned so_v = 0, so_k = 0;
        if (A == 0 && !LAST) {  // the pieces of V(j+2) | K(j+3) go out in X(j), two tiles ahead of their use, into region (j + 2) % 3
            so_v = v_so(cV);
            adv_v();
            so_k = k_so(cK);
            adv_k();
        }
        const int reg_ld = reg_cur == 0 ? 2 : reg_cur - 1;  // (reg_cur + 2) % 3

        long long tslot[9], tm[3] = {0, 0, 0};
        aw_for<0, NSLOT>([&](auto II) __attribute__((always_inline)) {
            constexpr int i = decltype(II)::value;
            constexpr int gq = i / 6, w = i % 6;
            if constexpr ((ABL & 128) != 0 && !LAST && w == 0) asm volatile("s_memtime %0" : "=s"(tslot[gq]));
            // (1) the slot's MFMA
            if constexpr (ABL & 2) {
            } else if constexpr (i < 24) {
                constexpr int u = i / 6, q = (i % 6) / 2, db = i % 2;
                aw_mfma_o(O[A][db], F[u & 3][2 * db + PA[q]], Pf[A][PB[q]][u]);
            } else {
                constexpr int s = (i - 24) / 6, q = (i % 6) / 2, kb = i % 2;
                if constexpr (s == 0 && q == 0) aw_mfma_s0(S[A][kb], F[(4 + s) & 3][2 * kb + PA[q]], Qf[A][PB[q]][s]);
                else aw_mfma_s(S[A][kb], F[(4 + s) & 3][2 * kb + PA[q]], Qf[A][PB[q]][s]);
            }
            // (2) the other stream's softmax.  A full segment runs its 50 micro-steps in slots 0 .. 45 (two in four of the first
            // nine), sums up in slot 46 and evaluates the range check in slot 47 - INSIDE the stream: behind the last MFMA of a
            // segment comes one branch on a scalar flag and the first MFMA of the next, not the tail of a softmax (measured:
            // ~300 cycles of idle matrix pipe per segment boundary).
            if constexpr (HAS_SM && !(ABL & 1)) {
                if constexpr (LAST) {
                    aw_for<0, MPS>([&](auto JJ) __attribute__((always_inline)) { micro(std::integral_constant<int, MPS * i + decltype(JJ)::value>{}); });
                } else if constexpr (i < 9) {
                    // slots 2, 4, 6, 8 take two micro-steps: the sums / high plane of a pair and the first step of the next pair
                    // (independent of each other but for the mixlo, which reads the high plane formed two instructions before)
                    constexpr int m0 = i + i / 2 - (i > 0 && i % 2 == 0 ? 1 : 0);  // 0 1 2 4 5 7 8 10 11
                    micro(std::integral_constant<int, m0>{});
                    if constexpr (i >= 2 && i % 2 == 0) micro(std::integral_constant<int, m0 + 1>{});
                } else if constexpr (i < 46) {
                    micro(std::integral_constant<int, i + 4>{});
                } else if constexpr (i == 46) {
                    ps = psA;
                    aw_acc(ps, psB);
                } else if constexpr (i == 47) {
                    slow = ((ABL & 28) == 28 || (ABL & 512)) ? false : __builtin_amdgcn_ballot_w64(!(ps < AW_LIMIT)) != 0;
                    l_run[B] += slow ? 0.f : ps;
                }
            }
            // (3) the tile's barrier (segment Y only, head of group 6): the pieces issued in X(j) have landed, everybody's; nobody
            // reads region j % 3 any more (the fragments of this segment's last groups were read in slots 25..34)
            if constexpr (A == 1 && !LAST && i == 36) {
                // counted: the 8 pieces issued in X(j) - the tile after next - stay in flight
                if constexpr ((ABL & 128) != 0) {  // measurement: what the wait is for
                    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm[0]));
                    asm volatile("s_waitcnt vmcnt(8)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm[1]));
                    asm volatile("s_barrier\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm[2]));
                }
                if constexpr (ABL & 16) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const unsigned d = reg_cur == 2 ? (unsigned)(-2 * AW_REGB) : (unsigned)AW_REGB;
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) R[s][pl] += d;
                reg_cur = reg_cur == 2 ? 0 : reg_cur + 1;
            }
            // (4) fragment reads of the group after next (slots 1..4 of a group); past this segment's last group they are
            // groups 0 / 1 of the NEXT segment - behind the barrier where that is the next tile
            if constexpr (w >= 1 && w <= 4 && !(ABL & 4)) {
                constexpr int gn = gq + 2;
                if constexpr (gn < NG) {
                    if constexpr (gn < 4) read_frag(gn & 3, w - 1, gn, 0);       // V(j), 16-key step gn
                    else read_frag(gn & 3, w - 1, gn - 4, 1);                    // K(j+1), 16-dim step gn - 4
                } else if constexpr (!(LAST && A == 1)) {
                    read_frag((gn - NG) & 3, w - 1, gn - NG, 0);                 // V of the next segment, steps 0 / 1
                }
            }
            // (5) LDS-direct loads of the next tile (segment X): one piece every fourth slot (a piece costs the wave 60 - 180
            // cycles of issue: bunched they starve the matrix pipe), the last one 55 slots ahead of the barrier
            if constexpr (A == 0 && !LAST && i < 32 && i % 4 == 1 && !(ABL & 8)) {
                if constexpr (i < 16) v_piece(reg_ld, i / 4, so_v);
                else k_piece(reg_ld, i / 4 - 4, so_k);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr ((ABL & 128) != 0 && !LAST) {
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tslot[8]));
            if (p.dbg && lane == 0 && seg_tile == 2 && (blockIdx.x == 0 || blockIdx.x == 517))
            {
                for (int q = 0; q < 9; ++q) p.dbg[320 + (((blockIdx.x ? 1 : 0) * 4 + wave) * 2 + A) * 9 + q] = tslot[q];
                if (A == 1) for (int q = 0; q < 3; ++q) p.dbg[320 + 144 + ((blockIdx.x ? 1 : 0) * 4 + wave) * 3 + q] = tm[q];
            }
        }

        if constexpr (HAS_SM && LAST && !(ABL & 1)) {
            // (the peeled last tile: its softmax ran two micro-steps per slot over 24 slots; split of the last pair and the range
            // check behind the stream)
            lo_k[15] = aw_mixlo(hi_k[15], pa[15][0]);
            ps = psA + psB;
            Pf[B][1][3][3] = aw_mixhi(lo_k[15], hi_k[15], pa[15][1]);
            const bool slow = __builtin_amdgcn_ballot_w64(!(ps < AW_LIMIT)) != 0;
            l_run[B] += slow ? 0.f : ps;
            if (__builtin_expect(slow, 0)) {
                sm_slow(std::integral_constant<int, B>{}, sinv, valid);
                ++n_slow;
            }
        }
        if constexpr (HAS_SM && !LAST && !(ABL & 1)) {
            if (__builtin_expect(slow, 0)) {  // (out of line: the fast path falls through)
                sm_slow(std::integral_constant<int, B>{}, sinv, valid);
                ++n_slow;
            }
        }
        if (A == 0) cur_next(cS);  // X used tile j, Y uses tile j + 1
    };

    for (int j = 0; j + 1 < n_tiles; ++j) {
        seg_tile = j;
        seg(std::integral_constant<int, 0>{}, std::false_type{});
        seg(std::integral_constant<int, 1>{}, std::false_type{});
        stamp();  // 4 .. 4 + n_tiles - 2: after tile j
    }
    stamp();  // 10: loop done
    seg(std::integral_constant<int, 0>{}, std::true_type{});  // the last tile, peeled: no K(j+1) left
    seg(std::integral_constant<int, 1>{}, std::true_type{});
    stamp();  // 11: last tile done

    // ---- epilogue: scaled planes of the two streams' output rows
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the pieces the last iterations issued past the last tile: nothing lands in LDS behind this workgroup)
    asm volatile("s_nop 15" : "+a"(O[0][0]), "+a"(O[0][1]), "+a"(O[1][0]), "+a"(O[1][1]));  // (the last MFMAs' results -> VALU)
    if constexpr (PART) {
        // a part's result: per query O (64 floats, at the exponent e_o of its last V tile), the running maximum and the sum of its keys
        // - record (leftover item, part, query) of AW_PART_F floats; the wave's e_o beside them
        const int64_t rec = ((int64_t)(lin - p.n_full) * p.n_split + ks) * AW_QT;
        if (lane == 0) p.part_e[rec / 64 + wave] = e_o;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float l_tot = l_run[a] + __shfl_xor(l_run[a], 32);
            float* o = p.part + (rec + wave * 64 + a * 32 + l31) * AW_PART_F;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<p2_f32x4*>(o + 32 * d + 8 * gq + 4 * lh) = p2_f32x4{O[a][d][4 * gq], O[a][d][4 * gq + 1], O[a][d][4 * gq + 2], O[a][d][4 * gq + 3]};
            if (lh == 0) { o[64] = m_run[a]; o[65] = l_tot; }
        }
        if (p.stats && n_slow && lane == 0) atomicAdd(p.stats, n_slow);
        return;
    }
    if (p.EO && lane == 0 && q_ok[0]) p.EO[(((int64_t)img * p.n_rows + q_row[0]) >> 6) * 4 + head] = e_o;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const float l_tot = l_run[a] + __shfl_xor(l_run[a], 32);
        const float inv = 1.f / (l_tot * P2_VS);
        // lane (query, lh) owns dims 8 g + 4 lh .. + 3 of each 32-dim block (g = 0..3): 8 bytes per plane.  The two halves of a
        // wave trade groups (v_permlane32_swap) so that a lane stores 16 contiguous bytes - dims 8 g .. + 7 of group 2 kp (lower
        // half) / 2 kp + 1 (upper half): half the store instructions for the same bytes (the tail is bound by their issue)
        uint16_t* op = p.out + p2_index((int64_t)img * p.n_rows + q_row[a], head * 64, p.D) + 8 * lh;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const int r0 = 8 * kp, r1 = 8 * kp + 4;  // first accumulator registers of groups 2 kp and 2 kp + 1
                const P2Pair x0 = p2_split_scaled(O[a][d][r0] * inv, O[a][d][r0 + 1] * inv), x1 = p2_split_scaled(O[a][d][r0 + 2] * inv, O[a][d][r0 + 3] * inv);
                const P2Pair y0 = p2_split_scaled(O[a][d][r1] * inv, O[a][d][r1 + 1] * inv), y1 = p2_split_scaled(O[a][d][r1 + 2] * inv, O[a][d][r1 + 3] * inv);
                const auto h0 = __builtin_amdgcn_permlane32_swap(x0.hi, y0.hi, false, false), h1 = __builtin_amdgcn_permlane32_swap(x1.hi, y1.hi, false, false);
                const auto l0 = __builtin_amdgcn_permlane32_swap(x0.lo, y0.lo, false, false), l1 = __builtin_amdgcn_permlane32_swap(x1.lo, y1.lo, false, false);
                uint16_t* o = op + d * 64 + kp * 16;
                if (q_ok[a]) {
                    *reinterpret_cast<p2_u32x4*>(o) = p2_u32x4{h0[0], h1[0], h0[1], h1[1]};
                    *reinterpret_cast<p2_u32x4*>(o + 32) = p2_u32x4{l0[0], l1[0], l0[1], l1[1]};
                }
            }
    }
    if (p.stats && n_slow && lane == 0) atomicAdd(p.stats, n_slow);
    stamp();  // 12: stores issued
    if constexpr ((ABL & 32) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp();  // 13: stores done
    }
}

// The parts of the key-split items -> output rows.  Part k of a query holds O_k = sum_j 2^(s_j - m_k + c) v_j over ITS keys (at the
// exponent e_k of its last V tile), l_k = sum_j 2^(s_j - m_k + c) and m_k; with m = max m_k: out = sum_k 2^(m_k - m) 2^(e_k - e) O_k /
// (16 sum_k 2^(m_k - m) l_k) at e = max e_k - the arithmetic of the kernel's own epilogue behind the rescaling of its slow path.
// One workgroup = one leftover item's 256 queries; thread = (query, 16 dims).
__global__ __launch_bounds__(1024) void attention_p2w_combine(AttnP2Params p) {
    const int item = blockIdx.x, lin = p.n_full + item;
    const int xcd = lin & 7, idx = lin >> 3;
    const int g = xcd * p.gper + idx / p.nq;
    if (g >= p.groups) return;
    const int qt = idx % p.nq, img = g / p.H, head = g % p.H, t = img % p.T;
    if (qt * AW_QT >= p.nv[t]) return;
    const int n_src = p.cross ? p.T - 1 : 1;
    int n_tiles = 0;
    for (int si = 0; si < n_src; ++si) n_tiles += (p.nv[!p.cross ? t : (si < t ? si : si + 1)] + 63) / 64;
    const int q = threadIdx.x >> 2, dq = (threadIdx.x & 3) * 16;
    const int row = qt * AW_QT + q;
    float m = -3.0e38f;
    int e = -1000;
    for (int k = 0; k < p.n_split; ++k) {
        if ((k + 1) * n_tiles / p.n_split - k * n_tiles / p.n_split <= 0) continue;
        const int64_t rec = ((int64_t)item * p.n_split + k) * AW_QT;
        m = fmaxf(m, p.part[(rec + q) * AW_PART_F + 64]);
        e = max(e, p.part_e[rec / 64 + (q >> 6)]);
    }
    float acc[16], l = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k = 0; k < p.n_split; ++k) {
        if ((k + 1) * n_tiles / p.n_split - k * n_tiles / p.n_split <= 0) continue;
        const int64_t rec = ((int64_t)item * p.n_split + k) * AW_QT;
        const float* o = p.part + (rec + q) * AW_PART_F;
        const int ek = p.part_e[rec / 64 + (q >> 6)];
        const float a = __builtin_amdgcn_exp2f(o[64] - m);
        const float ae = a * (ek - e < -126 ? 0.f : p2_exp2i(ek - e));
        l += a * o[65];
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            const p2_f32x4 v = *reinterpret_cast<const p2_f32x4*>(o + dq + i);
            acc[i] += ae * v[0]; acc[i + 1] += ae * v[1]; acc[i + 2] += ae * v[2]; acc[i + 3] += ae * v[3];
        }
    }
    if (row >= p.n_rows) return;
    if (p.EO && (threadIdx.x & 255) == 0) p.EO[(((int64_t)img * p.n_rows + row) >> 6) * 4 + head] = e;
    const float inv = 1.f / (l * P2_VS);
    uint16_t* op = p.out + p2_index((int64_t)img * p.n_rows + row, head * 64 + dq, p.D);  // (dq, dq + 8: inside one 32-column block)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        p2_u32x4 hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const P2Pair pr = p2_split_scaled(acc[8 * h + 2 * i] * inv, acc[8 * h + 2 * i + 1] * inv);
            hi[i] = pr.hi; lo[i] = pr.lo;
        }
        *reinterpret_cast<p2_u32x4*>(op + 8 * h) = hi;
        *reinterpret_cast<p2_u32x4*>(op + 8 * h + 32) = lo;
    }
}

int launch_attention_p2w(e2emv_ctx* ctx, AttnP2Params& p, int n_valid, hipStream_t s) {
    p.nq = (n_valid + AW_QT - 1) / AW_QT;
    if (int rc = ensure_flags(ctx)) return rc;
    p.stats = ctx->d_flags + 5;
    const size_t lds = 3 * AW_REGB + 2048;  // + the tile exponents (<= 7 sources x 32 tiles x 8 bytes)
    const bool multi = p.cross && p.T > 2;
    const void* fn = p.EQK ? (multi ? reinterpret_cast<const void*>(attention_p2w_kernel<true, 0, true>) : reinterpret_cast<const void*>(attention_p2w_kernel<true, 0, false>))
                           : (multi ? reinterpret_cast<const void*>(attention_p2w_kernel<false, 0, true>) : reinterpret_cast<const void*>(attention_p2w_kernel<false, 0, false>));
#ifdef E2EMV_STAMPS
    switch (ctx->attn_abl) {  // measurement build: ablations of the main loop (e2emv_attention_p2 flags bits 4..7 + bit 3)
        case 1: fn = reinterpret_cast<const void*>(attention_p2w_kernel<true, 1, false>); break;
        case 2: fn = reinterpret_cast<const void*>(attention_p2w_kernel<true, 2, false>); break;
        case 3: fn = reinterpret_cast<const void*>(attention_p2w_kernel<true, 3, false>); break;
        case 4: fn = reinterpret_cast<const void*>(attention_p2w_kernel<true, 4, false>); break;
        case 5: fn = reinterpret_cast<const void*>(attention_p2w_kernel<true, 5, false>); break;
        case 8: fn = reinterpret_cast<const void*>(attention_p2w_kernel<true, 8, false>); break;
        case 13: fn = reinterpret_cast<const void*>(attention_p2w_kernel<true, 13, false>); break;
        case 16: fn = reinterpret_cast<const void*>(attention_p2w_kernel<true, 16, false>); break;
        case 29: fn = reinterpret_cast<const void*>(attention_p2w_kernel<true, 29, false>); break;
        case 12: fn = reinterpret_cast<const void*>(attention_p2w_kernel<true, 28, false>); break;   // MFMAs + softmax only
        case 11: fn = reinterpret_cast<const void*>(attention_p2w_kernel<true, 92, false>); break;   // ... the softmax fed from a constant
        case 10: fn = reinterpret_cast<const void*>(attention_p2w_kernel<true, 30, false>); break;   // softmax only
        case 14: fn = reinterpret_cast<const void*>(attention_p2w_kernel<true, 32, false>); break;  // timestamps
        case 9: fn = reinterpret_cast<const void*>(attention_p2w_kernel<true, 160, false>); break;
        case 7: fn = reinterpret_cast<const void*>(attention_p2w_kernel<true, 256 + 512, false>); break;  // no wait for Q / K(0) in the prologue, never the slow path
        case 6: fn = reinterpret_cast<const void*>(attention_p2w_kernel<true, 512, false>); break;        // never the slow path (the arm to compare it with)  // ... and inside the segments of tile 2
        default: break;
    }
    static long long* d_stamps = nullptr;
    const size_t nb = sizeof(long long) * (2 * 4 * 40 + 2 * 4 * 2 * 9 + 2 * 4 * 3);
    p.dbg = nullptr;
    if (ctx->attn_abl == 14 || ctx->attn_abl == 9) {
        if (!d_stamps) E2EMV_HIP(ctx, hipMalloc((void**)&d_stamps, nb));
        E2EMV_HIP(ctx, hipMemsetAsync(d_stamps, 0, nb, s));
        p.dbg = d_stamps;
    }
#endif
    if (int rc = ensure_dynamic_lds(ctx, fn, lds)) return rc;
    void* args[] = {&p};
    // ---- the last round of workgroups.  n_items = R CUs + r: with 0 < r <= CUs / 2 the r leftover items would hold the chip for a whole
    // round at < half its width (T = 5, 1024 keypoints: 640 items, 3 rounds for 2.5 of work) - they are split along the keys into
    // n_split = floor(CUs / r) parts each (<= 8, <= the fewest key tiles of an item) + one small combine launch.  Fewer items than CUs
    // (R = 0: a pair or two per call) split the same way.  e2emv_attention_p2 flags bit 12 (tests): never.
    const int n_items = 8 * p.gper * p.nq, cus = std::max(8, ctx->num_cus / 8 * 8);
    const int r = n_items % cus;
    int min_tiles = 1 << 30;
    for (int t = 0; t < p.T; ++t) {  // key tiles of the item with the fewest: a part must own at least one
        int nt = 0;
        for (int si = 0; si < p.T; ++si)
            if (p.cross ? si != t : si == t) nt += (p.nv[si] + 63) / 64;
        min_tiles = std::min(min_tiles, nt);
    }
    p.n_full = n_items;
    p.n_split = 1;
    p.part = nullptr;
    p.part_e = nullptr;
    if (ctx->attn_key_split && r > 0 && 2 * r <= cus && r % 8 == 0) {
        const int k = std::min(std::min(cus / r, 8), min_tiles);
        if (k >= 2) {
            const size_t n_rec = (size_t)r * k * AW_QT;
            const size_t need = n_rec * AW_PART_F * sizeof(float) + (n_rec / 64) * sizeof(int);
            if (need > ctx->attn_part_bytes) {
                if (ctx->d_attn_part) { E2EMV_HIP(ctx, hipStreamSynchronize(s)); (void)hipFree(ctx->d_attn_part); ctx->d_attn_part = nullptr; ctx->attn_part_bytes = 0; }
                E2EMV_HIP(ctx, hipMalloc((void**)&ctx->d_attn_part, need));
                ctx->attn_part_bytes = need;
            }
            p.n_full = n_items - r;
            p.n_split = k;
            p.part = ctx->d_attn_part;
            p.part_e = reinterpret_cast<int*>(ctx->d_attn_part + n_rec * AW_PART_F);
        }
    }
    if (p.n_full > 0) {
        E2EMV_HIP(ctx, hipLaunchKernel(fn, dim3(p.n_full), dim3(256), args, lds, s));
        E2EMV_CHECK_LAUNCH(ctx, "attention_p2w_kernel");
    }
    if (p.n_split > 1) {
        const void* fp = p.EQK ? (multi ? reinterpret_cast<const void*>(attention_p2w_kernel<true, 0, true, true>) : reinterpret_cast<const void*>(attention_p2w_kernel<true, 0, false, true>))
                               : (multi ? reinterpret_cast<const void*>(attention_p2w_kernel<false, 0, true, true>) : reinterpret_cast<const void*>(attention_p2w_kernel<false, 0, false, true>));
        if (int rc = ensure_dynamic_lds(ctx, fp, lds)) return rc;
        E2EMV_HIP(ctx, hipLaunchKernel(fp, dim3((n_items - p.n_full) * p.n_split), dim3(256), args, lds, s));
        E2EMV_CHECK_LAUNCH(ctx, "attention_p2w_kernel (parts)");
        hipLaunchKernelGGL(attention_p2w_combine, dim3(n_items - p.n_full), dim3(1024), 0, s, p);
        E2EMV_CHECK_LAUNCH(ctx, "attention_p2w_combine");
    }
#ifdef E2EMV_STAMPS
    if (p.dbg) {
        E2EMV_HIP(ctx, hipStreamSynchronize(s));
        std::vector<long long> h(2 * 4 * 40 + 2 * 4 * 2 * 9 + 2 * 4 * 3);
        E2EMV_HIP(ctx, hipMemcpy(h.data(), p.dbg, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        static int printed = 0;
        if (printed++ < 2)
            for (int wg = 0; wg < 2; ++wg)
                for (int w = 0; w < 4; ++w) {
                    const long long* o = &h[((size_t)wg * 4 + w) * 40];
                    fprintf(stderr, "attention_p2w wg %d wave %d: cycles since start at the stamps (0 start, 1 Q loads issued, 2 Q and first tiles landed, 3 prologue done, then after every tile of the loop, loop done, last tile done, stores issued, stores done)\n  ", wg ? 517 : 0, w);
                    for (int i = 0; i < 40 && (i == 0 || o[i]); ++i) fprintf(stderr, " %lld", o[i] - o[0]);
                    fprintf(stderr, "\n   per tile:");
                    for (int i = 4; i < 40 && o[i + 4]; ++i) fprintf(stderr, " %lld", o[i] - o[i - 1]);
                    fprintf(stderr, "\n");
                    if (ctx->attn_abl == 9)
                        for (int a = 0; a < 2; ++a) {
                            const long long* t = &h[320 + (((size_t)wg * 4 + w) * 2 + a) * 9];
                            fprintf(stderr, "   tile 2 segment %c, first slot at %lld, behind the last slot at %lld; cycles per group of 6 slots:", a ? 'Y' : 'X', t[0] - o[0], t[8] - o[0]);
                            for (int q = 0; q < 8; ++q) fprintf(stderr, " %lld", t[q + 1] - t[q]);
                            fprintf(stderr, "\n");
                            if (a == 1) {
                                const long long* m = &h[320 + 144 + ((size_t)wg * 4 + w) * 3];
                                fprintf(stderr, "   its barrier: reached at %lld, pieces landed +%lld, barrier passed +%lld\n", m[0] - o[0], m[1] - m[0], m[2] - m[1]);
                            }
                        }
                }
    }
#endif
    return E2EMV_OK;
}

}  // namespace e2emv
