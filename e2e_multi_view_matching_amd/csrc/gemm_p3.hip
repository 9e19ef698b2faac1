// Plane-in / plane-out "NT" GEMM of the bf16x3 arithmetic mode (round 6):  C[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]) (+ R[m][n])
//
// The arithmetic of gemm_x3.hip - every fp32 operand as three bf16 planes x = x1 + x2 + x3 (24 significant bits, bf16 keeps fp32's
// exponent range: no scaling, no exponent side-band), a product as the 6 largest of the 9 plane products, each exact in fp32,
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16 - on the SKELETON of gemm_p2.hip: the activation arrives already as its planes
// ("P3", below), the planes go from global memory straight into LDS (buffer_load ... lds, no VGPR staging, no split in the K loop -
// gemm_x3 spends ~1/3 of a K step converting), LDS double-buffered with ONE barrier per K step, the K step software-pipelined inside
// the wave, the epilogue's stores counted and left in flight under the next tile's first steps.  bf16x3 is the split mode that is
// NOT narrower than fp32 (3 x 8 bits); it pays 6 MFMA products per flop where f16x2 pays 3.
//
// P3 layout of an fp32 matrix X [rows][C] (C % 16 == 0): row m = C / 16 blocks of 96 B:
//     block b = [ p1(m, 16 b .. 16 b + 15) : 16 bf16 | p2(...) : 16 bf16 | p3(...) : 16 bf16 ],   x = p1 + p2 + p3
// i.e. one K step of a consumer (16 columns of the three planes) is ONE contiguous 96-byte piece per row.  LDS image of a tile
// row = its 6 16-byte chunks (chunk = 2 plane + k half), chunk c stored at position c ^ ((row >> 3) & 1): rows r and r + 8 sit 768 B
// = 3 x 256 B apart, i.e. on the same banks - the swap of the two halves of a plane separates them (conflict-free ds_read_b128).
// The swizzle is applied on the SOURCE address of the LDS-direct load, whose destination is lane-linear: lane l of a wave's load
// number t lands at chunk 64 t + l of the tile = (row (64 t + l) / 6, position (64 t + l) % 6).
#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

#include "p2.h"

namespace e2emv {

typedef __attribute__((ext_vector_type(16))) float p3_f32x16;

constexpr int P3_BM = 256, P3_BN = 256, P3_BK = 16;
constexpr int P3_ROWB = 96;                      // bytes of one tile row per K step
constexpr int P3_TILEB = P3_BM * P3_ROWB;        // 24 KB per operand tile
constexpr int P3_BUFB = 2 * P3_TILEB;            // A tile | W tile
constexpr int P3_SLABB = 32 * 32 * 4;            // one epilogue slab per wave
constexpr int P3_LDSB = 2 * P3_BUFB + 8 * P3_SLABB;  // 128 KB

struct GemmP3Params {
    const uint16_t* A;
    const uint16_t* A2;
    const uint16_t* W;
    unsigned a_bytes, a2_bytes, w_bytes;
    unsigned lda_b, lda2_b, ldw_b;  // row strides in bytes (6 per column)
    const float* bias;
    const uint16_t* R;
    float* C32;
    uint16_t* C3;
    int64_t ldc, ldr;
    int M, N, K, K1;
    int tiles_n, total, relu;
    char* dummy;  // 4 KB: target of the stores of rows / columns beyond the matrix (a wave always issues all its stores)
};

// offset (in halves) of plane 0 of element (m, k) in a P3 matrix of C columns; plane pl sits 16 pl halves further
__host__ __device__ __forceinline__ int64_t p3_index(int64_t m, int k, int64_t C) { return m * 3 * C + (k >> 4) * 48 + (k & 15); }

__device__ __forceinline__ void p3_mfma(p3_f32x16& c, p2_u32x4 a, p2_u32x4 b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void p3_mfma0(p3_f32x16& c, p2_u32x4 a, p2_u32x4 b) {  // zero C operand: the first product of a tile
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}

// two fp32 -> the three packed bf16 planes (round to nearest even at every stage, as the host split of the weights and gemm3.hip's)
__device__ __forceinline__ void p3_split2(float v0, float v1, unsigned& p1, unsigned& p2, unsigned& p3) {
    asm("" : "+v"(v0), "+v"(v1));  // (one conversion per plane: the residual is taken against the value that is stored)
    const __bf16 a0 = (__bf16)v0, a1 = (__bf16)v1;
    const float r0 = v0 - (float)a0, r1 = v1 - (float)a1;
    const __bf16 b0 = (__bf16)r0, b1 = (__bf16)r1;
    const float s0 = r0 - (float)b0, s1 = r1 - (float)b1;
    const __bf16 c0 = (__bf16)s0, c1 = (__bf16)s1;
    p1 = (unsigned)__builtin_bit_cast(uint16_t, a0) | (unsigned)__builtin_bit_cast(uint16_t, a1) << 16;
    p2 = (unsigned)__builtin_bit_cast(uint16_t, b0) | (unsigned)__builtin_bit_cast(uint16_t, b1) << 16;
    p3 = (unsigned)__builtin_bit_cast(uint16_t, c0) | (unsigned)__builtin_bit_cast(uint16_t, c1) << 16;
}
// packed pair of the three planes -> fp32 (exact: the planes do not overlap)
__device__ __forceinline__ p2_f32x2 p3_join2(unsigned p1, unsigned p2, unsigned p3) {
    auto lo = [](unsigned u) { return __builtin_bit_cast(float, u << 16); };
    auto hi = [](unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); };
    return {(lo(p3) + lo(p2)) + lo(p1), (hi(p3) + hi(p2)) + hi(p1)};
}

// ---- K step, software-pipelined inside the wave like gp_kstep (gemm_p2_core.h): 4 groups (weight row blocks j) of 12 MFMAs - the 6
// plane products x 2 activation row blocks, smallest terms first: x3 w1, x2 w2, x1 w3, x2 w1, x1 w2, x1 w1 - the three weight
// fragments of group j + 1 read under the MFMAs of group j.  No multiplies, 18 fragment reads per 48 MFMAs (gemm_p2: 24 reads + 32
// v_pk_mul_f16).
template <bool first_step>
__device__ __forceinline__ void p3_kstep(const char* smem, int buf, int wr, int wc, int l31, int lh, p3_f32x16 (&acc)[4][2]) {
    const int sw = (l31 >> 3) & 1;
    const char* xs = smem + buf * P3_BUFB + (wr * 64 + l31) * P3_ROWB;
    const char* ws = smem + buf * P3_BUFB + P3_TILEB + (wc * 128 + l31) * P3_ROWB;
    auto rd_x = [&](int t, int pl) { return *reinterpret_cast<const p2_u32x4*>(xs + t * 32 * P3_ROWB + (((2 * pl + lh) ^ sw) << 4)); };
    auto rd_w = [&](int j, int pl) { return *reinterpret_cast<const p2_u32x4*>(ws + j * 32 * P3_ROWB + (((2 * pl + lh) ^ sw) << 4)); };
    p2_u32x4 xb[2][3], wb[2][3];
#pragma unroll
    for (int pl = 2; pl >= 0; --pl) {  // (the planes the first MFMAs need first)
        xb[0][pl] = rd_x(0, pl);
        xb[1][pl] = rd_x(1, pl);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) wb[0][pl] = rd_w(0, pl);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int PW[6] = {0, 1, 2, 0, 1, 0}, PX[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int gp = g & 1;
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (first_step && q == 0) p3_mfma0(acc[g][i], wb[gp][PW[q]], xb[i][PX[q]]);
                else p3_mfma(acc[g][i], wb[gp][PW[q]], xb[i][PX[q]]);
                // the next group's weight fragments: one behind each of slots 1, 3, 5 (w1, the plane of four of its six products, first)
                if (g < 3 && i == 1 && q < 3) wb[gp ^ 1][q] = rd_w(g + 1, q);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
}

__device__ __forceinline__ void p3_acc_fence(p3_f32x16 (&acc)[4][2]) {
    asm volatile("s_nop 15" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
}

// stores a wave issues per tile (the K loop's counted waits leave exactly these in flight): planes 3 x 16 B, fp32 2 x 16 B per pass
template <bool PLANES>
constexpr int p3_stores() { return 8 * 2 * (PLANES ? 3 : 2); }

// ---- epilogue: gp_epilogue's scheme (gemm_p2_core.h) without the exponent side-band.  A wave's 8 blocks of 32 x 32 go through its LDS
// slab into the row-contiguous view (lane = row o_r of 16, 8 consecutive columns), software-pipelined: block b + 1 through the slab
// and its residual loads issued while block b is finished (bias, ReLU, residual, split) and stored.
template <bool PLANES, bool HAS_R>
__device__ __forceinline__ void p3_epilogue(const GemmP3Params& p, char* smem, const p3_f32x16 (&acc)[4][2], int wave, int lane, int tm, int tn) {
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    char* sl = smem + 2 * P3_BUFB + wave * P3_SLABB;
    const int o_r = lane >> 2, o_c = (lane & 3) * 8;
    const int o_z = o_r & 7;
    char* dummy = p.dummy + lane * 48;  // (a pass stores up to 80 B behind it: 63 x 48 + 96 < 4 KB)
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
    constexpr int NB = HAS_R ? 2 : 4;
    p2_f32x4 bias8[NB][2];
    auto load_bias = [&](int slot, int j) {
        const int n = tn * P3_BN + wc * 128 + j * 32 + o_c;
        bias8[slot][0] = bias8[slot][1] = p2_f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias && n < p.N) bias8[slot][0] = *reinterpret_cast<const p2_f32x4*>(p.bias + n);
        if (p.bias && n + 4 < p.N) bias8[slot][1] = *reinterpret_cast<const p2_f32x4*>(p.bias + n + 4);
    };
    if (!HAS_R) {
#pragma unroll
        for (int j = 0; j < 4; ++j) load_bias(j, j);
    }
    p2_f32x4 rv[2][2][2];              // [parity][pass][half]
    p2_u32x4 rr[HAS_R ? 2 : 1][2][3];  // [parity][pass][plane]: the block's residual
    const int row0 = tm * P3_BM + wr * 64 + o_r, col0 = tn * P3_BN + wc * 128 + o_c;
    const int rows_left = p.M - row0;
    float* const c32_t = PLANES ? nullptr : p.C32 + (int64_t)row0 * p.ldc + col0;
    uint16_t* const c3_t = PLANES ? p.C3 + p3_index(row0, col0, p.ldc) : nullptr;
    const int64_t rstep = (PLANES ? 48 : 16) * p.ldc;  // 16 rows further, in elements of the output
    auto stage = [&](auto BB) {
        constexpr int b = decltype(BB)::value;
        constexpr int i = b >> 2, j = b & 3;
        slab_write(i, j);
        if (HAS_R) load_bias(b & 1, j);
        const uint16_t* rcol = nullptr;
        if constexpr (HAS_R) {
            const int nc = min(col0 + j * 32, p.N - 8);
            rcol = p.R + ((nc >> 4) * 48 + (nc & 15));
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = o_r + 16 * pass;
            const int c0 = 2 * (lane & 3);
            rv[b & 1][pass][0] = *reinterpret_cast<const p2_f32x4*>(sl + r * 128 + ((c0 ^ o_z) << 4));
            rv[b & 1][pass][1] = *reinterpret_cast<const p2_f32x4*>(sl + r * 128 + (((c0 + 1) ^ o_z) << 4));
            if constexpr (HAS_R) {
                const int m = min(row0 + i * 32 + 16 * pass, p.M - 1);
                const uint16_t* rp = rcol + (int64_t)m * (3 * p.ldr);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) rr[b & 1][pass][pl] = *reinterpret_cast<const p2_u32x4*>(rp + 16 * pl);
            }
        }
    };
    auto finish = [&](auto BB) {
        constexpr int b = decltype(BB)::value;
        constexpr int i = b >> 2, j = b & 3;
        const bool col_ok = col0 + j * 32 < p.N;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const bool ok = col_ok && i * 32 + 16 * pass < rows_left;
            p2_f32x4 v0 = rv[b & 1][pass][0] + bias8[HAS_R ? (b & 1) : j][0];
            p2_f32x4 v1 = rv[b & 1][pass][1] + bias8[HAS_R ? (b & 1) : j][1];
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = relu_nan(v0[e]); v1[e] = relu_nan(v1[e]); }
            }
            if constexpr (HAS_R) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const p2_f32x2 a = p3_join2(rr[b & 1][pass][0][e], rr[b & 1][pass][1][e], rr[b & 1][pass][2][e]);
                    const p2_f32x2 c = p3_join2(rr[b & 1][pass][0][2 + e], rr[b & 1][pass][1][2 + e], rr[b & 1][pass][2][2 + e]);
                    v0[2 * e] += a[0]; v0[2 * e + 1] += a[1];
                    v1[2 * e] += c[0]; v1[2 * e + 1] += c[1];
                }
            }
            if constexpr (!PLANES) {
                float* cp = c32_t + (2 * i + pass) * rstep + j * 32;
                float* cq = cp + 4;
                if (!ok) cp = reinterpret_cast<float*>(dummy);
                if (!(ok && col0 + j * 32 + 4 < p.N)) cq = reinterpret_cast<float*>(dummy + 16);
                *reinterpret_cast<p2_f32x4*>(cp) = v0;
                *reinterpret_cast<p2_f32x4*>(cq) = v1;
            } else {
                unsigned u1[4], u2[4], u3[4];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    p3_split2(v0[2 * e], v0[2 * e + 1], u1[e], u2[e], u3[e]);
                    p3_split2(v1[2 * e], v1[2 * e + 1], u1[2 + e], u2[2 + e], u3[2 + e]);
                }
                const p2_u32x4 q1 = {u1[0], u1[1], u1[2], u1[3]}, q2 = {u2[0], u2[1], u2[2], u2[3]}, q3 = {u3[0], u3[1], u3[2], u3[3]};
                uint16_t* cp = c3_t + (2 * i + pass) * rstep + j * 96;  // (32 columns = two 16-column blocks of 48 halves)
                if (!ok) cp = reinterpret_cast<uint16_t*>(dummy);
                *reinterpret_cast<p2_u32x4*>(cp) = q1;
                *reinterpret_cast<p2_u32x4*>(cp + 16) = q2;
                *reinterpret_cast<p2_u32x4*>(cp + 32) = q3;
            }
        }
    };
#define P3_BLK(b) std::integral_constant<int, b>{}
    stage(P3_BLK(0));
    stage(P3_BLK(1)); finish(P3_BLK(0));
    stage(P3_BLK(2)); finish(P3_BLK(1));
    stage(P3_BLK(3)); finish(P3_BLK(2));
    stage(P3_BLK(4)); finish(P3_BLK(3));
    stage(P3_BLK(5)); finish(P3_BLK(4));
    stage(P3_BLK(6)); finish(P3_BLK(5));
    stage(P3_BLK(7)); finish(P3_BLK(6));
    finish(P3_BLK(7));
#undef P3_BLK
}

template <bool PLANES, bool HAS_R>
__global__ __launch_bounds__(512, 1) void gemm_p3_kernel(GemmP3Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem_p3[];

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
    const int nk = p.K / P3_BK, nk1 = p.K1 / P3_BK;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.A2), 0, (int)p.a2_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.W), 0, (int)p.w_bytes, 0x00020000);

    // ---- loader: an operand tile of a K step = 256 rows x 6 chunks = 1536 chunks of 16 B = 24 LDS-direct loads, 3 per wave and operand
    unsigned a_vo[3], a2_vo[3], w_vo[3];
    auto setup = [&](int t) {
        const int tm = t / p.tiles_n, tn = t - tm * p.tiles_n;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned c = 64u * (3u * (unsigned)wave + (unsigned)i) + (unsigned)lane;
            const unsigned row = c / 6u, pos = c - 6u * row;
            const unsigned q = pos ^ ((row >> 3) & 1u);
            const unsigned gm = (unsigned)min(tm * P3_BM + (int)row, p.M - 1);
            a_vo[i] = gm * p.lda_b + q * 16u;
            a2_vo[i] = gm * p.lda2_b + q * 16u;
            const unsigned gn = (unsigned)min(tn * P3_BN + (int)row, p.N - 1);
            w_vo[i] = gn * p.ldw_b + q * 16u;
        }
    };
    auto issue = [&](int buf, int kt, unsigned dep) {
        char* dst = smem_p3 + buf * P3_BUFB + 3 * wave * 1024;
        if (kt < nk1) {
            const unsigned so = (unsigned)kt * (unsigned)P3_ROWB;
#pragma unroll
            for (int i = 0; i < 3; ++i) p2_glds16(rsA, dst + i * 1024, a_vo[i] + dep, so);
        } else {
            const unsigned so = (unsigned)(kt - nk1) * (unsigned)P3_ROWB;
#pragma unroll
            for (int i = 0; i < 3; ++i) p2_glds16(rsA2, dst + i * 1024, a2_vo[i] + dep, so);
        }
        const unsigned sw = (unsigned)kt * (unsigned)P3_ROWB;
#pragma unroll
        for (int i = 0; i < 3; ++i) p2_glds16(rsW, dst + P3_TILEB + i * 1024, w_vo[i] + dep, sw);
    };

    p3_f32x16 acc[4][2];

    // ---- pipeline: gemm_p2.hip's.  Step g lives in LDS buffer g & 1; a step opens with "my loads of this step have landed" + ONE raw
    // barrier, then the loads of step g + 1 go into the other buffer (waves 4 - 7 issue first and compute after, waves 0 - 3 the other
    // way round).  Across an epilogue the load position runs TWO steps ahead, both issued BEFORE the epilogue's stores; vmcnt retires in
    // issue order, so the first step behind an epilogue waits with vmcnt(6 + stores) and the second with vmcnt(stores): the store burst
    // drains under two K steps of the next tile.
    constexpr int NST = p3_stores<PLANES>();
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
    const bool issue_first = wave >= 4;
    const bool overlap = nk >= 3;
    int since = 8;       // K steps since the last epilogue
    bool ahead = false;  // the loads of the step after next were issued before that epilogue
    int buf = 0;
    auto step = [&](auto FIRST) {
        if (since == 0 && ahead) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(6 + NST) : "memory");
        else if (since <= 1 && overlap) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const bool ldv = ld_valid && !(since == 0 && ahead);
        if (issue_first && ldv) issue(buf ^ 1, ld_kt, 0u);
        p3_kstep<decltype(FIRST)::value>(smem_p3, buf, wr, wc, l31, lh, acc);
        if (!issue_first && ldv) {
            unsigned dep = 0;
            asm("" : "+v"(dep) : "v"(acc[3][1]));  // scheduling-only: keeps the loads behind the MFMAs
            issue(buf ^ 1, ld_kt, dep);
        }
        if (ldv) advance();
        if (since == 0) ahead = false;
        ++since;
        buf ^= 1;
    };
    for (;;) {
        step(std::true_type{});
        for (int kt = 1; kt < nk; ++kt) step(std::false_type{});
        if (overlap && ld_valid) {
            // the buffer of the step just computed is free once every wave is through it: the loads of the step after next
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            unsigned dep = 0;
            asm("" : "+v"(dep) : "v"(acc[3][1]));
            issue(buf ^ 1, ld_kt, dep);
            advance();
            ahead = true;
        }
        p3_acc_fence(acc);
        p3_epilogue<PLANES, HAS_R>(p, smem_p3, acc, wave, lane, tile / p.tiles_n, tile % p.tiles_n);
        since = overlap ? 0 : 8;
        tile += slots;
        if (tile >= t_end) break;
    }
}

int launch_gemm_p3(e2emv_ctx* ctx, const GemmP3Args& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0 || a.K <= 0) return set_err(ctx, E2EMV_ESHAPE, "gemm_p3: empty problem");
    const int K1 = a.A2 ? a.K1 : a.K;
    if (a.K % 16 || K1 % 16 || K1 <= 0 || K1 > a.K || (K1 < a.K && !a.A2)) return set_err(ctx, E2EMV_ESHAPE, "gemm_p3: K=%d K1=%d must be multiples of 16", a.K, K1);
    if (!a.A || !a.W || a.lda % 16 || a.lda < K1 || (a.A2 && (a.lda2 % 16 || a.lda2 < a.K - K1)))
        return set_err(ctx, E2EMV_ESHAPE, "gemm_p3: operand planes need whole 16-column blocks (lda=%lld lda2=%lld)", (long long)a.lda, (long long)a.lda2);
    if ((uintptr_t)a.A % 16 || (a.A2 && (uintptr_t)a.A2 % 16) || (uintptr_t)a.W % 16 || (a.bias && (uintptr_t)a.bias % 16) || (a.R && (uintptr_t)a.R % 16))
        return set_err(ctx, E2EMV_ESHAPE, "gemm_p3: operands must be 16-byte aligned");
    if (a.R && (a.ldr % 16 || a.ldr < a.N)) return set_err(ctx, E2EMV_ESHAPE, "gemm_p3: residual planes need ldr %% 16 == 0");
    if (a.K / 16 < 3) return set_err(ctx, E2EMV_ESHAPE, "gemm_p3: K = %d (>= 48)", a.K);
    const int64_t a_bytes = (int64_t)a.M * a.lda * 6, a2_bytes = a.A2 ? (int64_t)a.M * a.lda2 * 6 : 16, w_bytes = (int64_t)a.N * a.K * 6;
    if (a_bytes >= ((int64_t)1 << 31) || a2_bytes >= ((int64_t)1 << 31) || w_bytes >= ((int64_t)1 << 31))
        return set_err(ctx, E2EMV_ESHAPE, "gemm_p3: operand larger than 2 GB (M=%d lda=%lld): 32-bit byte offsets", a.M, (long long)a.lda);
    GemmP3Params p{};
    p.A = a.A; p.A2 = a.A2 ? a.A2 : a.A; p.W = a.W;
    p.a_bytes = (unsigned)a_bytes; p.a2_bytes = (unsigned)(a.A2 ? a2_bytes : a_bytes); p.w_bytes = (unsigned)w_bytes;
    p.lda_b = (unsigned)(a.lda * 6); p.lda2_b = (unsigned)((a.A2 ? a.lda2 : a.lda) * 6); p.ldw_b = (unsigned)(a.K * 6);
    p.bias = a.bias; p.R = a.R; p.ldr = a.ldr;
    p.C32 = a.C32; p.C3 = a.C3; p.ldc = a.ldc;
    p.M = a.M; p.N = a.N; p.K = a.K; p.K1 = K1;
    p.tiles_n = (a.N + P3_BN - 1) / P3_BN;
    p.total = ((a.M + P3_BM - 1) / P3_BM) * p.tiles_n;
    p.relu = a.relu ? 1 : 0;
    const bool planes = a.C3 != nullptr;
    if (planes) {
        if (a.N % 8 || a.ldc % 16 || a.ldc < a.N || (uintptr_t)a.C3 % 16) return set_err(ctx, E2EMV_ESHAPE, "gemm_p3: plane output needs N %% 8 == 0, ldc %% 16 == 0");
    } else if (!a.C32 || a.N % 4 || a.ldc % 4 || (uintptr_t)a.C32 % 16) {
        return set_err(ctx, E2EMV_ESHAPE, "gemm_p3: fp32 output needs N %% 4 == 0, ldc %% 4 == 0");
    }
    if (!ctx->d_dummy) E2EMV_HIP(ctx, hipMalloc((void**)&ctx->d_dummy, 4096));
    p.dummy = ctx->d_dummy;
    const void* fn = planes ? (a.R ? reinterpret_cast<const void*>(gemm_p3_kernel<true, true>) : reinterpret_cast<const void*>(gemm_p3_kernel<true, false>))
                            : (a.R ? reinterpret_cast<const void*>(gemm_p3_kernel<false, true>) : reinterpret_cast<const void*>(gemm_p3_kernel<false, false>));
    const int per_xcd = (p.total + 7) / 8;
    const int sl = std::min(per_xcd, std::max(1, ctx->num_cus / 8));
    if (int rc = ensure_dynamic_lds(ctx, fn, P3_LDSB)) return rc;
    void* args[] = {&p};
    E2EMV_HIP(ctx, hipLaunchKernel(fn, dim3(8 * sl), dim3(512), args, P3_LDSB, s));
    E2EMV_CHECK_LAUNCH(ctx, "gemm_p3_kernel");
    return E2EMV_OK;
}

// ---- conversions fp32 <-> P3 (thread = one row, 8 consecutive columns)
__global__ __launch_bounds__(256) void to_planes3_kernel(const float* src, int64_t rows, int C, int64_t ld_src, uint16_t* dst) {
    const int per_row = C / 8;
    const int64_t total = rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / per_row;
        const int n = (int)(i - m * per_row) * 8;
        const float* sp = src + m * ld_src + n;
        const p2_f32x4 v0 = *reinterpret_cast<const p2_f32x4*>(sp), v1 = *reinterpret_cast<const p2_f32x4*>(sp + 4);
        unsigned u1[4], u2[4], u3[4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            p3_split2(v0[2 * e], v0[2 * e + 1], u1[e], u2[e], u3[e]);
            p3_split2(v1[2 * e], v1[2 * e + 1], u1[2 + e], u2[2 + e], u3[2 + e]);
        }
        const p2_u32x4 q1 = {u1[0], u1[1], u1[2], u1[3]}, q2 = {u2[0], u2[1], u2[2], u2[3]}, q3 = {u3[0], u3[1], u3[2], u3[3]};
        uint16_t* dp = dst + p3_index(m, n, C);
        *reinterpret_cast<p2_u32x4*>(dp) = q1;
        *reinterpret_cast<p2_u32x4*>(dp + 16) = q2;
        *reinterpret_cast<p2_u32x4*>(dp + 32) = q3;
    }
}

__global__ __launch_bounds__(256) void from_planes3_kernel(const uint16_t* src, int64_t rows, int C, float* dst, int64_t ld_dst) {
    const int per_row = C / 8;
    const int64_t total = rows * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / per_row;
        const int n = (int)(i - m * per_row) * 8;
        const uint16_t* sp = src + p3_index(m, n, C);
        const p2_u32x4 q1 = *reinterpret_cast<const p2_u32x4*>(sp), q2 = *reinterpret_cast<const p2_u32x4*>(sp + 16), q3 = *reinterpret_cast<const p2_u32x4*>(sp + 32);
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const p2_f32x2 a = p3_join2(q1[e], q2[e], q3[e]);
            o[2 * e] = a[0]; o[2 * e + 1] = a[1];
        }
        float* dp = dst + m * ld_dst + n;
        *reinterpret_cast<p2_f32x4*>(dp) = p2_f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<p2_f32x4*>(dp + 4) = p2_f32x4{o[4], o[5], o[6], o[7]};
    }
}

static int p3_grid_for(int64_t items) { return (int)std::min<int64_t>((items + 255) / 256, 256 * 8); }

int launch_to_planes3(e2emv_ctx* ctx, const float* src, int64_t rows, int C, int64_t ld_src, uint16_t* dst, hipStream_t s) {
    if (rows <= 0 || C <= 0 || C % 16 || ld_src % 4 || (uintptr_t)src % 16 || (uintptr_t)dst % 16)
        return set_err(ctx, E2EMV_ESHAPE, "to_planes3: C=%d must be a multiple of 16, rows 16-byte aligned", C);
    hipLaunchKernelGGL(to_planes3_kernel, dim3(p3_grid_for(rows * (C / 8))), dim3(256), 0, s, src, rows, C, ld_src, dst);
    E2EMV_CHECK_LAUNCH(ctx, "to_planes3_kernel");
    return E2EMV_OK;
}

int launch_from_planes3(e2emv_ctx* ctx, const uint16_t* src, int64_t rows, int C, float* dst, int64_t ld_dst, hipStream_t s) {
    if (rows <= 0 || C <= 0 || C % 16 || ld_dst % 4 || (uintptr_t)src % 16 || (uintptr_t)dst % 16)
        return set_err(ctx, E2EMV_ESHAPE, "from_planes3: C=%d must be a multiple of 16, rows 16-byte aligned", C);
    hipLaunchKernelGGL(from_planes3_kernel, dim3(p3_grid_for(rows * (C / 8))), dim3(256), 0, s, src, rows, C, dst, ld_dst);
    E2EMV_CHECK_LAUNCH(ctx, "from_planes3_kernel");
    return E2EMV_OK;
}

// host: fp32 weights [rows][cols] -> P3 planes appended to `out` (offset returned); the split of add_split3 (ctx.hip)
size_t add_split_p3(std::vector<uint16_t>& out, const std::vector<float>& w, int rows, int cols) {
    auto f2bf = [](float f) {
        uint32_t u;
        memcpy(&u, &f, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    };
    auto bf2f = [](uint16_t h) {
        const uint32_t u = (uint32_t)h << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    };
    const size_t off = (out.size() + 127) & ~size_t(127);
    out.resize(off + (size_t)rows * 3 * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            const float v = w[(size_t)r * cols + c];
            const uint16_t a = f2bf(v);
            const float r1 = v - bf2f(a);
            const uint16_t b = f2bf(r1);
            const float r2 = r1 - bf2f(b);
            const uint16_t d = f2bf(r2);
            uint16_t* o = &out[off + (size_t)p3_index(r, c, cols)];
            o[0] = a; o[16] = b; o[32] = d;
        }
    return off;
}

}  // namespace e2emv
