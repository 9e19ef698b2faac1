// Ingest kernels of the matcher forward (forward.hip), shared with the training forward (train.hip).
#pragma once
#include "common.h"

namespace e2emv {

struct IngestParams {
    const float* kpts[E2EMV_MAX_TUPLE];
    const float* ksc[E2EMV_MAX_TUPLE];
    const void* desc[E2EMV_MAX_TUPLE];
    float img_w[E2EMV_MAX_TUPLE], img_h[E2EMV_MAX_TUPLE];
    int B, T, n_rows, D, c0, f16;
    int Nimg[E2EMV_MAX_TUPLE];  // keypoints of image t
    const float* w0;
    const float* b0;
    float* x0;
    float* h0;
    float* inp;  // optional [img][n_rows][4]: the normalised (x, y, score, 0) the encoder sees (training tape)
};

// [B][D][N] (N contiguous) -> [img][n_rows][D] (D contiguous); rows >= N := 0
__global__ void ingest_transpose(IngestParams p);
// keypoint normalisation + kenc layer 0 (3 -> c0, BN folded, ReLU); rows >= N := 0
__global__ void ingest_kenc0(IngestParams p);

}  // namespace e2emv
