// "P2" activation format of the f16x2 arithmetic mode: an fp32 matrix X [rows][C] (C % 32 == 0) is kept in HBM as two fp16
// planes, 4 bytes per element like the fp32 it replaces, laid out so that one K step of a consumer is ONE contiguous,
// cache-line sized piece per row:
//
//   row m = C/32 blocks of 128 B;  block b = [ hi(m, 32b .. 32b+31) : 32 halves | lo(m, 32b .. 32b+31) : 32 halves ]
//
//   GEMM operands ("scaled" planes):     x = hi + 2^-11 lo',  hi = fp16(x),  lo' = fp16(2^11 (x - hi))
//   attention operands ("plain" planes): x = hi + lo,         hi = fp16(x),  lo  = fp16(x - hi)      (after a power-of-two
//                                        pre-scale that keeps lo normal: attention_p2.hip)
//
// The PRODUCER of an activation emits the planes in its epilogue (gemm_p2.hip, attention_p2.hip); consumers move the
// 128-byte pieces straight from global memory into LDS (buffer_load ... lds, no VGPR staging, no split in the K loop).
// LDS image of a tile row = the 8 16-byte chunks of the piece, chunk c stored at position c ^ ((row >> 1) & 7): every
// ds_read_b128 of an MFMA fragment (32 rows, one chunk index) then touches 16 different 16-byte bank slots per lane group.
// The swizzle is applied on the SOURCE address of the LDS-direct load (the destination of such a load is lane-linear).
#pragma once
#include "common.h"

namespace e2emv {

typedef __attribute__((ext_vector_type(2))) float p2_f32x2;
typedef __attribute__((ext_vector_type(4))) float p2_f32x4;
typedef __attribute__((ext_vector_type(2))) _Float16 p2_f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 p2_f16x8;
typedef __attribute__((ext_vector_type(2))) unsigned p2_u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned p2_u32x4;

struct P2Pair {
    unsigned hi, lo;  // two packed fp16 each
};

// (v0, v1) -> scaled planes.  hi = one v_cvt_pk_f16_f32; lo' = v_fma_mix{lo,hi}_f16(hi as fp16, -2048, 2048 v): the exact
// fp32 residual rounded once.  The values are made opaque first: left alone hipcc 7.2 may select the high plane twice
// (cvt_pk for the stored copy, fma_mix for the one the residual is taken against) and the two differ near fp16 ties
// (tests/test_gpu_kernels.py::test_attention_split_kernels_near_fp16_ties).
__device__ __forceinline__ P2Pair p2_split_scaled(float v0, float v1) {
    asm("" : "+v"(v0), "+v"(v1));
    const p2_f32x2 vv = {v0, v1};
    const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(vv, p2_f16x2));
    const float s0 = v0 * 2048.f, s1 = v1 * 2048.f;
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, %4, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %4, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 1"
        : "=&v"(lo) : "v"(hi), "v"(s0), "v"(s1), "s"(-2048.f));
    return {hi, lo};
}

// (v0, v1) -> plain planes hi + lo
__device__ __forceinline__ P2Pair p2_split_plain(float v0, float v1) {
    asm("" : "+v"(v0), "+v"(v1));
    const p2_f32x2 vv = {v0, v1};
    const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(vv, p2_f16x2));
    unsigned lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 1"
        : "=&v"(lo) : "v"(hi), "v"(v0), "v"(v1));
    return {hi, lo};
}

// packed pair of scaled planes -> fp32
__device__ __forceinline__ p2_f32x2 p2_join_scaled(unsigned hi, unsigned lo) {
    const p2_f16x2 h = __builtin_bit_cast(p2_f16x2, hi), l = __builtin_bit_cast(p2_f16x2, lo);
    return {(float)h[0] + (float)l[0] * (1.f / 2048.f), (float)h[1] + (float)l[1] * (1.f / 2048.f)};
}
__device__ __forceinline__ p2_f32x2 p2_join_plain(unsigned hi, unsigned lo) {
    const p2_f16x2 h = __builtin_bit_cast(p2_f16x2, hi), l = __builtin_bit_cast(p2_f16x2, lo);
    return {(float)h[0] + (float)l[0], (float)h[1] + (float)l[1]};
}

// One LDS-direct load: 64 lanes x 16 bytes from rsrc[voffset + soffset] to the 1 KiB at the wave-uniform LDS address dst
// (lane l lands at dst + 16 l).  Kept in a non-template function: inside a kernel TEMPLATE's dependent lambda the builtin
// makes hipcc 7.2 drop the host-side instantiation of the whole kernel without a diagnostic (undefined __device_stub__).
__device__ __forceinline__ void p2_glds16(__amdgpu_buffer_rsrc_t rsrc, char* dst, unsigned voffset, unsigned soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voffset, soffset, 0, 0);
}
// the same with the non-temporal hint (aux = 2: nt): a stream ONE CU reads - it should not push the weight tiles every CU
// re-reads out of the XCD's L2
__device__ __forceinline__ void p2_glds16_nt(__amdgpu_buffer_rsrc_t rsrc, char* dst, unsigned voffset, unsigned soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, voffset, soffset, 0, 2);
}

// ---- range side-band: tile exponents -------------------------------------------------------------------------------------
// fp16 planes have fp16's exponent range.  Every plane tensor therefore carries one int32 exponent e per block of 64 rows x
// 64 columns (E [rows / 64][C / 64]): the planes hold X 2^-e.  A producer picks e from an upper bound of the block's
// magnitudes; inside the dead zone 2^-6 ... 2^15 it is 0 (every block of an ordinary network: the consumers then take
// their plain paths, which cost nothing), above it the block's maximum is brought to [2^14, 2^15), below it to ~2^10.
// Consumers undo it exactly (powers of two): a GEMM rescales its accumulators when the exponent changes between K
// blocks and folds the last one into its output scale, the attention folds q's and k's into the logit scale and v's into
// the O accumulator.  Range of the mode: 2^-16 ... 2^60 on top of fp16's own, i.e. |x| < 2^75; precision is relative to
// the largest element of a 64 x 64 block (22 bits down to 2^-14 of it, absolute 2^-36 of it below).
constexpr int P2_EMIN = -16, P2_EMAX = 60;
__host__ __device__ __forceinline__ float p2_exp2i(int e) {  // 2^e, |e| <= 126
    return __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
}
// max over the 64 lanes of a wave on the DPP cross-lane path (no LDS crossbar round trips); every lane gets the result
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float p2_dpp(float identity, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float p2_wave_max(float v) {  // v >= 0
    v = fmaxf(v, p2_dpp<0xB1, 0xF>(v, v));    // quad_perm [1,0,3,2]
    v = fmaxf(v, p2_dpp<0x4E, 0xF>(v, v));    // quad_perm [2,3,0,1]
    v = fmaxf(v, p2_dpp<0x141, 0xF>(v, v));   // row_half_mirror
    v = fmaxf(v, p2_dpp<0x140, 0xF>(v, v));   // row_mirror
    v = fmaxf(v, p2_dpp<0x142, 0xA>(0.f, v)); // row_bcast:15 into rows 1 and 3
    v = fmaxf(v, p2_dpp<0x143, 0xC>(0.f, v)); // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int p2_pick_exponent(float bound) {
    if (!(bound < 3.0e38f)) return 0;  // inf / NaN: the values go through as they are
    const int lg = (int)((__builtin_bit_cast(unsigned, bound) >> 23) & 255u) - 127;  // floor(log2 bound) for normal numbers
    int e = 0;
    if (lg >= 15) e = lg - 14;
    else if (lg < -6 && bound > 0.f) e = lg - 10;
    return e < P2_EMIN ? P2_EMIN : (e > P2_EMAX ? P2_EMAX : e);
}

// offset (in halves) of the hi half of element (m, k) in a P2 matrix of C columns; the lo half sits 32 halves further
__host__ __device__ __forceinline__ int64_t p2_index(int64_t m, int k, int64_t C) { return m * 2 * C + (k >> 5) * 64 + (k & 31); }

// attention operand pre-scales (powers of two, cancel exactly): q x 2^6 on top of log2(e)/sqrt(d); v x 2^4; the softmax
// numerators carry 2^10 (attention_p2.hip)
constexpr float P2_QS = 64.f, P2_VS = 16.f;
// position of key k (0..15 inside its group of 16) in the stored V^T planes: bits 2 and 3 swapped, so that a 16-byte chunk
// holds keys {0-3, 8-11} or {4-7, 12-15} - the key order of the transposed-score accumulator registers
__host__ __device__ __forceinline__ int p2_vt_pos(int k) { return (k & 3) | ((k & 4) << 1) | ((k & 8) >> 1); }

// ---- launchers (gemm_p2.hip / attention_p2.hip / p2_tools.hip) ----
enum { P2_OUT_F32 = 0, P2_OUT_PLANES = 1, P2_OUT_QKV = 2 };
struct GemmP2Args {
    int M = 0, N = 0, K = 0, K1 = 0;
    const uint16_t* A = nullptr;   // P2 scaled planes [M][lda columns]
    int64_t lda = 0;
    const uint16_t* A2 = nullptr;  // second K segment (k >= K1)
    int64_t lda2 = 0;
    const uint16_t* W = nullptr;   // P2 planes of 2^s W [N][K columns] (ctx.hip: add_split_p2)
    float out_scale = 1.f;         // 2^-s
    const float* bias = nullptr;
    const uint16_t* Rp = nullptr;  // residual, P2 scaled planes [M][ldr columns]
    int64_t ldr = 0;
    bool relu = false;
    int out = P2_OUT_F32;
    float* C32 = nullptr;          // P2_OUT_F32: [M][ldc]
    uint16_t* Cp = nullptr;        // P2_OUT_PLANES: scaled planes [M][ldc columns]; P2_OUT_QKV: q | k plain planes [M][2D]
    int64_t ldc = 0;
    uint16_t* Vt = nullptr;        // P2_OUT_QKV: V^T plain planes [M / n_rows][heads][64][n_rows]
    int n_rows = 0, heads = 0;
    // tile exponents (see above; any pointer may be null = all zero / not wanted).  EA [M/64][lda/64], EA2 [M/64][lda2/64],
    // ER [M/64][ldr/64]; outputs EC [M/64][ldc/64] (P2_OUT_QKV: q | k -> [M/64][8]) and EVt [M/64][4] (V^T, indexed by key rows)
    const int* EA = nullptr;
    const int* EA2 = nullptr;
    const int* ER = nullptr;
    int* EC = nullptr;
    int* EVt = nullptr;
    const float* AR = nullptr;     // [M/64][ldr/64] max |value| of the residual's blocks (picks the output exponent; null: 2^(16 + e))
    float* AC = nullptr;           // [M/64][ldc/64] the same of a plane output that is a later residual
    float bias_amax = 0.f;         // upper bound of |bias| (0 when there is none)
};
int launch_gemm_p2(e2emv_ctx* ctx, const GemmP2Args& a, hipStream_t s);
// gemm_p2c.hip: the GEMMs a[0 .. n) over the same M rows, chained per 256-row block in ONE launch.  dep_kt[i] = the first K step
// of stage i's first tile that reads what the tile right before it stored (0: from the start; >= 4; P2_CHAIN_INDEP: nothing)
constexpr int P2_CHAIN_INDEP = 1 << 30;
int launch_gemm_p2_chain(e2emv_ctx* ctx, const GemmP2Args* a, const int* dep_kt, int n, hipStream_t s);
// fp32 [rows][C] (row stride ld_src floats) -> P2 scaled planes [rows][C]
// E (optional) [rows/64][C/64]: the tile exponents are computed from the data (rows % 64 == 0, C % 64 == 0 then)
int launch_to_planes(e2emv_ctx* ctx, const float* src, int64_t rows, int C, int64_t ld_src, uint16_t* dst, hipStream_t s, int* E = nullptr,
                     float* AM = nullptr);  // AM (optional, with E): the blocks' max |value|
int launch_from_planes(e2emv_ctx* ctx, const uint16_t* src, int64_t rows, int C, float* dst, int64_t ld_dst, hipStream_t s, const int* E = nullptr);
// [mdesc_i[n] | mdesc_j[matches[n]]] of B samples ([B][n_rows][2 D]) straight to scaled planes + tile exponents (the conf head's GEMM operand)
int launch_conf_gather_planes(e2emv_ctx* ctx, const float* mdesc_i, const float* mdesc_j, int64_t tuple_stride, const int64_t* matches, int N, int n_rows,
                              int B, int D, uint16_t* dst, int* E, hipStream_t s);
// host: fp32 weights [rows][cols] -> P2 planes of 2^s W appended to `out` (offset returned), *out_scale = 2^-s
size_t add_split_p2(std::vector<uint16_t>& out, const std::vector<float>& w, int rows, int cols, float* out_scale);
// softmax(q k^T / sqrt(64)) v on plane operands: qk = q | k plain planes [n_img * n_rows][2D], vt = V^T plain planes;
// out = P2 scaled planes [n_img * n_rows][D]
// EQK [rows/64][8], EVt [rows/64][4]: tile exponents of the operands, EO [rows/64][4] of the output (null = zero / not wanted)
int launch_attention_p2(e2emv_ctx* ctx, int B, int T, int n_rows, const int* n_valid_img, int D, int H, const uint16_t* qk,
                        const uint16_t* vt, int cross, uint16_t* outp, hipStream_t s, const int* EQK = nullptr, const int* EVt = nullptr,
                        int* EO = nullptr);

}  // namespace e2emv
