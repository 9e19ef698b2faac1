// Parameters shared by the two attention kernels on plane operands (attention_p2.hip: two waves per SIMD, 32 queries per wave;
// attention_p2w.hip: one wave per SIMD, 64 queries per wave, software-pipelined inside the wave).
#pragma once
#include "p2.h"

namespace e2emv {

struct AttnP2Params {
    const uint16_t* qk;   // [n_img*n_rows][2D] plain planes, q | k
    const uint16_t* vt;   // [n_img][H][64][n_rows] plain planes
    uint16_t* out;        // [n_img*n_rows][D] scaled planes
    unsigned qk_bytes, vt_bytes;
    const int* EQK;       // tile exponents (p2.h) of q | k [rows/64][8] and of V^T [rows/64][4] (by key rows); null = all zero
    const int* EVt;
    int* EO;              // exponents of the output [rows/64][4]; null = not wanted
    int B, T, n_rows, D, H, cross;
    int nv[E2EMV_MAX_TUPLE];
    int nq, groups, gper;
    int n_full, n_split;  // attention_p2w key split (launcher): items < n_full are whole, the rest are walked by n_split workgroups each
    float* part;          // ... their (O, m, l) records and
    int* part_e;          // ... the V exponent each wave's O sits at
    unsigned* stats;      // attention_p2w: [0] += (wave, stream, tile) softmaxes redone on the slow path (beyond a stream's first tile)
    long long* dbg;       // measurement build: timestamps of two workgroups (attention_p2w)
};

// attention_p2w.hip: p.nq / p.gper are filled in by the launcher (256 queries per workgroup)
int launch_attention_p2w(e2emv_ctx* ctx, AttnP2Params& p, int n_valid, hipStream_t s);

}  // namespace e2emv
