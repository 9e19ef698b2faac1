// Internal declarations shared by the HIP translation units of libe2emv.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <type_traits>
#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/e2emv.h"

namespace e2emv {

// ReLU with torch.relu's NaN behaviour: relu(NaN) = NaN.  fmaxf(NaN, 0) = 0 would turn a non-finite activation (an fp16
// overflow in the f16x2 mode, a NaN in the inputs) into a silent zero instead of letting it reach the scores, where
// e2emv_sync reports it.
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }


// profile slots (kernel families)
enum ProfSlot {
    PS_INGEST = 0,
    PS_GEMM,
    PS_ATTN,
    PS_SCORE,
    PS_SINKHORN,
    PS_MATCH,
    PS_CONF,
    PS_W8PT,
    PS_MISC,
    // the layer GEMMs of the plane kernels, one slot per kernel instantiation (bench.py: roofline.per_kernel); "gemm" keeps the rest
    PS_GEMM_QKV,
    PS_GEMM_MLP0,
    PS_GEMM_MLP1,
    PS_GEMM_CHAIN,
    PS_COUNT
};

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

struct LayerWeights {
    // all device pointers into the weight arena; GEMM weights are [out][in] row-major
    float* w_qkv = nullptr;   // [3D][D]   rows head-major: q | k | v
    float* b_qkv = nullptr;   // [3D]
    float* w_merge = nullptr; // [D][D]    input columns head-major
    float* b_merge = nullptr;
    float* w_mlp0 = nullptr;  // [2D][2D]  BN folded
    float* b_mlp0 = nullptr;
    float* w_mlp1 = nullptr;  // [D][2D]
    float* b_mlp1 = nullptr;
    int type = 0;             // 0 self, 1 cross
    // bf16x3 planes ("S3": [out][3][in]) of the three big GEMMs
    uint16_t* w3_qkv = nullptr;
    uint16_t* w3_mlp0 = nullptr;
    uint16_t* w3_mlp1 = nullptr;
    // f16x2 planes ([out][{hi, lo}][in] of 2^s W, gemm_h2.hip) of the same GEMMs and their 2^-s
    uint16_t* wh_qkv = nullptr;
    uint16_t* wh_mlp0 = nullptr;
    uint16_t* wh_mlp1 = nullptr;
    float hs_qkv = 0.f, hs_mlp0 = 0.f, hs_mlp1 = 0.f;
    // the same 2^s W as P2 planes (p2.h: 32-column blocks of {hi, lo}) for gemm_p2.hip; the scales are hs_* above
    uint16_t* wp_qkv = nullptr;
    uint16_t* wp_mlp0 = nullptr;
    uint16_t* wp_mlp1 = nullptr;
    float ba_qkv = 0.f, ba_mlp0 = 0.f, ba_mlp1 = 0.f;  // max |bias| of the three GEMMs (bounds for the tile exponents, p2.h)
};

// Family timing as a CHAIN of events on the launch stream: one event where the family changes (it ends the previous
// interval and starts the next) and one when an API call returns - ~45 events per forward pass instead of two per launch
// (170), which cost 5-9 % of a 12.8 ms step.  Node k: its event and the family / launch count of the interval that starts
// there (slot -1: nothing of ours - the time up to the next API call).
struct ProfEvent {
    hipEvent_t ev;
    int slot;
    int launches;
};

}  // namespace e2emv

struct e2emv_ctx {
    int device = 0;
    int num_cus = 256;
    std::string err;
    // raw host weights as handed over by e2emv_set_weight
    std::map<std::string, e2emv::HostTensor> raw;
    // committed model
    bool committed = false;
    bool fuse_merge = true;  // fold attn.merge into MLP0 at commit time (E2EMV_NO_FUSE_MERGE=1 disables)
    e2emv_model_desc model{};
    float* d_warena = nullptr;
    size_t warena_floats = 0;
    uint16_t* d_w3arena = nullptr;
    size_t w3arena_elems = 0;
    int precision = 0;  // E2EMV_PRECISION_F32 | _BF16X3 | _F16X2 (dense GNN contractions)
    int64_t split_min_rows = -1;  // split-operand kernels from this many keypoint rows per call (-1: half a 128-row tile per CU)
    bool h2_legacy = false;  // f16x2 mode on the round-2 kernels (fp32 activations split inside gemm_h2 / attention_h2f): the A/B arm of the plane path
    int gemm_chain = 1;      // f16x2 kernel generation 5 (default): MLP0 -> MLP1 -> next q|k|v chained in one launch (gemm_p2c.hip); 0 = generation 4 (a launch per GEMM), 2 = chained whenever the shapes allow (e2emv_set_f16x2_kernels(105))
    bool attn_wide = true;   // f16x2 kernel generations 4, 5: attention_p2w above 256 keys; false = generation 3 (attention_p2 everywhere)
    bool attn_key_split = true;  // attention_p2w: a half-empty last round of workgroups is split along the keys (attention_p2w.hip; e2emv_attention_p2 flags bit 12 clears it for one call)
    int attn_abl = 0;        // measurement build: ablation of attention_p2w's main loop
    int attn_p2_nw = 0;      // attention on planes: 0 = by key count, 4 | 8 = attention_p2 with that many waves, 1 = attention_p2w (micro-benchmarks)
    bool b3_planes = false;  // bf16x3 mode: q|k|v handed to the attention as planes from the GEMM epilogue (E2EMV_B3_PLANES=1)
    // keypoint encoder: layer 0 (3->c0) used by the ingest kernel, the rest through the GEMM
    float* kenc_w0 = nullptr;  // [c0][3] folded
    float* kenc_b0 = nullptr;  // [c0]
    std::vector<float*> kenc_w, kenc_b;  // layers 1..n (folded), [out][in]
    std::vector<const uint16_t*> kenc_wh;  // the same layers as fp16 x 2 weight planes (null: fan-in < 128, stays on the fp32 kernel)
    std::vector<float> kenc_hs;
    const uint16_t* wh_final = nullptr;  // final_proj / conf_mlp.0 as fp16 x 2 weight planes
    const uint16_t* wh_conf0 = nullptr;
    const uint16_t* wp_final = nullptr;  // ... and in the plane kernels' weight format (gemm_p2)
    const uint16_t* wp_conf0 = nullptr;
    float hs_final = 0.f, hs_conf0 = 0.f, ba_final = 0.f, ba_conf0 = 0.f;
    std::vector<int> kenc_dims;          // [3, c0, c1, ..., D]
    std::vector<e2emv::LayerWeights> layers;
    float* w_final = nullptr;
    float* b_final = nullptr;
    float bin_score = 1.f;
    float* w_conf0 = nullptr;  // [D][2D] folded
    float* b_conf0 = nullptr;
    float* w_conf1 = nullptr;  // [D]
    float b_conf1 = 0.f;
    // SuperPoint front-end (superpoint.hip): packed conv weights [Cout][ky][kx][Cin] + biases
    bool sp_committed = false;
    float* d_sparena = nullptr;
    float* sp_w[12] = {nullptr};
    float* sp_b[12] = {nullptr};
    // partial results of the key-split attention (small problems), grown on demand
    float* d_attn_part = nullptr;
    size_t attn_part_bytes = 0;
    // device flags: [0] give-up flag of the running resident Sinkhorn launch, [1] sticky count of give-ups
    unsigned* d_flags = nullptr;
    uint64_t stat_sinkhorn_bad = 0;      // Sinkhorn problems with non-finite scores so far (e2emv_get_stats)
    uint64_t stat_sinkhorn_rescued = 0;  // problems the log-domain rescue pass re-solved behind the resident kernel
    int sk_range_strikes = 0;        // observed calls with range rescues (2 -> sinkhorn_stream)
    int sk_stream_calls = 0;         // calls served by the chain since the demotion (16 -> the resident kernel is tried again)
    uint64_t stat_sinkhorn_timeouts = 0;
    uint64_t stat_sinkhorn_rows128 = 0;  // calls served by sinkhorn_resident128 (128 rows per workgroup)
    int sinkhorn_kernel = 0;         // E2EMV_SINKHORN_* pin (e2emv_set_sinkhorn_kernel; initial value from the E2EMV_SINKHORN variable, read once at e2emv_create)
    bool sinkhorn_stream = false;    // set by the first such report: later calls run the log-domain launch chain
    char* d_dummy = nullptr;  // 4 KB scratch line: target of masked-out stores of kernels that must issue a fixed number of stores (gemm_p2.hip)
    // workspace arena
    char* d_ws = nullptr;
    size_t ws_bytes = 0;
    // One call at a time per context: every compute entry point holds `mu` while it carves the shared workspace and
    // enqueues its kernels (Python threads / several torch streams on one device would otherwise interleave inside the
    // arena).  Work of consecutive calls is ordered by the stream; when the caller switches streams the previous one is
    // drained first, because the arena contents of the earlier call may still be in use there.
    std::recursive_mutex mu;
    void* train = nullptr;  // e2emv::TrainState (train.hip)
    // matched descriptors (final_proj output) of the last forward_joint call, in the workspace: [md_imgs][md_rows][md_dim] fp32
    const float* last_mdesc = nullptr;
    int md_imgs = 0, md_rows = 0, md_n = 0, md_dim = 0;
    hipStream_t last_stream = nullptr;
    bool have_last_stream = false;
    // profiling
    bool prof = false;
    std::vector<e2emv::ProfEvent> prof_events;
    hipStream_t prof_stream = nullptr;
    std::vector<hipEvent_t> event_pool;
    float prof_ms[E2EMV_PROF_SLOTS] = {0};
    int64_t prof_n[E2EMV_PROF_SLOTS] = {0};
};

namespace e2emv {

int set_err(e2emv_ctx* ctx, int code, const char* fmt, ...);

#define E2EMV_HIP(ctx, expr)                                                                      \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return e2emv::set_err(ctx, E2EMV_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                  __FILE__, __LINE__);                                            \
    } while (0)

#define E2EMV_CHECK_LAUNCH(ctx, what)                                                             \
    do {                                                                                          \
        hipError_t _e = hipGetLastError();                                                        \
        if (_e != hipSuccess)                                                                     \
            return e2emv::set_err(ctx, E2EMV_EHIP, "launch of %s failed: %s", what, hipGetErrorString(_e)); \
    } while (0)

// RAII guard of a compute entry point: serialises calls on the context and orders the shared workspace across streams.
struct CallGuard {
    std::unique_lock<std::recursive_mutex> lk;
    e2emv_ctx* c;
    CallGuard(e2emv_ctx* ctx, void* stream) : lk(ctx->mu), c(ctx) {
        hipStream_t s = (hipStream_t)stream;
        (void)hipSetDevice(ctx->device);
        if (ctx->have_last_stream && ctx->last_stream != s) (void)hipStreamSynchronize(ctx->last_stream);
        ctx->last_stream = s;
        ctx->have_last_stream = true;
    }
    ~CallGuard();  // closes an open profiling interval (ctx.hip)
};
#define E2EMV_ENTER(ctx, stream) e2emv::CallGuard _call_guard(ctx, stream)
#define E2EMV_LOCK(ctx) std::unique_lock<std::recursive_mutex> _call_lock((ctx)->mu)

// The synchronous-API uploads and memsets of the commit / allocation paths (hipMemcpy from pageable memory, hipMemset) are
// work of the legacy NULL stream: hipMemcpy may return once the data is staged, hipMemset is asynchronous - and the caller's
// streams (torch's) are non-blocking, i.e. not ordered with the null stream.  Every entry point that used them ends with
// this fence, so that the kernels the caller enqueues next find the bytes in place (seen as wrong first results of a second
// context created while another stream kept the GPU busy).
#define E2EMV_NULL_STREAM_FENCE(ctx) E2EMV_HIP(ctx, hipStreamSynchronize(nullptr))

// workspace: makes ctx->d_ws at least `bytes` large, growing the arena (synchronising) when needed; each
// top-level entry point carves it with 256-byte aligned offsets.
int ws_reserve(e2emv_ctx* ctx, size_t bytes);

// Ablation / profiling knobs (forced tile shapes, kernels with a phase removed - results may be wrong) exist only in the
// measurement build (-DE2EMV_STAMPS: `python tools/p2_stamps.py --build` makes libe2emv_stamps.so).  The release library
// compiles them to their defaults and never reads these environment variables.
#ifdef E2EMV_STAMPS
inline int dbg_knob(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
inline const char* dbg_env(const char* name) { return getenv(name); }
#else
constexpr int dbg_knob(const char*, int dflt) { return dflt; }
constexpr const char* dbg_env(const char*) { return nullptr; }
#endif
void train_free(e2emv_ctx* ctx);  // train.hip
// ctx->d_flags (device words: [0] Sinkhorn give-up flag, [1] its sticky count, [2] plane blocks that needed a tile exponent,
// [3] Sinkhorn problems rescued after a range event, [6] after a timeout, [4] give-ups of the resident kernel's waits,
// [5] (wave, tile) softmaxes attention_p2w redid on its slow path)
int ensure_flags(e2emv_ctx* ctx);
// hipFuncAttributeMaxDynamicSharedMemorySize for kernels that use more than the default dynamic LDS: once per
// (device, kernel), thread-safe (ctx.hip)
int ensure_dynamic_lds(e2emv_ctx* ctx, const void* kernel, size_t bytes);

// profiling: prof_begin before every launch of a family (an event is recorded only when the family changes), prof_end is a
// no-op kept for symmetry at the call sites; the entry point's CallGuard closes the last interval
void prof_begin(e2emv_ctx* ctx, int slot, hipStream_t s);
inline void prof_end(e2emv_ctx*, hipStream_t) {}
void prof_close(e2emv_ctx* ctx);

// ---- kernel launchers (defined in the respective .hip files) ----
struct GemmArgs {
    int batch = 1, M = 0, N = 0, K = 0, K1 = 0;
    const float* A = nullptr;
    int64_t lda = 0, sA = 0;
    const float* A2 = nullptr;
    int64_t lda2 = 0, sA2 = 0;
    const float* W = nullptr;
    int64_t ldw = 0, sW = 0;
    const float* bias = nullptr;
    const float* R = nullptr;
    int64_t ldr = 0, sR = 0;
    float* C = nullptr;
    int64_t ldc = 0, sC = 0;
    uint16_t* C3 = nullptr;  // optional bf16x3-plane output (S3 [M][3][ldc3]); C may then be null
    int64_t ldc3 = 0;
    uint16_t* Vt = nullptr;  // optional: columns >= vt_n0 go out transposed as bf16x3 planes (attention3's V^T)
    int vt_n0 = 0, n_rows = 0;
    int q_cols = 0;          // columns < q_cols are scaled by q_scale
    float q_scale = 1.f;
    float scale = 1.f;
    bool relu = false;
    // implicit-GEMM 3x3 convolution (pad 1, stride 1) over NHWC activations: A = input [imgs*H*W][conv_c], K = 9*conv_c
    int conv_h = 0, conv_w = 0, conv_c = 0;
    bool conv_pool = false;  // fuse the following 2x2 / stride-2 max-pool (output [imgs*(H/2)*(W/2)][N])
};
int launch_gemm_nt(e2emv_ctx* ctx, const GemmArgs& a, hipStream_t s);

int launch_attention(e2emv_ctx* ctx, int B, int T, int n_rows, const int* n_valid_img, int D, int H, const float* qkv,
                     int cross, float* out, hipStream_t s);

// ---- bf16x3 split-operand path (gemm3.hip / attention3.hip): fp32-class accuracy on the bf16 pipe ----
// "S3" = matrix stored as three bf16 planes per row: row r = [plane0 | plane1 | plane2], each ld wide.
struct Gemm3Args {
    int M = 0, N = 0, K = 0, K1 = 0;
    const uint16_t* A = nullptr;
    int64_t lda = 0;
    const uint16_t* A2 = nullptr;
    int64_t lda2 = 0;
    const uint16_t* W = nullptr;
    int64_t ldw = 0;
    const float* bias = nullptr;
    const float* R = nullptr;
    int64_t ldr = 0;
    float* C32 = nullptr;
    int64_t ldc32 = 0;
    uint16_t* C3 = nullptr;
    int64_t ldc3 = 0;
    uint16_t* Vt = nullptr;
    int vt_n0 = 0, n_rows = 0;
    int q_cols = 0;
    float q_scale = 1.f;
    bool relu = false;
};
int launch_gemm3(e2emv_ctx* ctx, const Gemm3Args& a, hipStream_t s);
// gemm_x3.hip: fp32 activations (a.A / a.A2, a.bias, a.R, a.C, a.relu as for launch_gemm_nt) x pre-split weights W3 (S3 [N][3][ldw3])
// host: fp32 weights -> the f16x2 planes appended to `out` (offset returned), *out_scale = 2^-s (ctx.hip)
size_t add_split_h2(std::vector<uint16_t>& out, const std::vector<float>& w, int rows, int cols, float* out_scale);
// h2_out_scale != 0 selects the fp16 x 2 form (gemm_h2.hip): W3 = fp16 planes [N][{hi, lo}][ldw3] of 2^s W, h2_out_scale = 2^-s
int launch_gemm_h2(e2emv_ctx* ctx, const GemmArgs& a, const uint16_t* WH, int64_t ldw, float out_scale, hipStream_t s);
int launch_gemm_x3(e2emv_ctx* ctx, const GemmArgs& a, const uint16_t* W3, int64_t ldw3, hipStream_t s, float h2_out_scale = 0.f);
int launch_split3(e2emv_ctx* ctx, const float* src, int64_t rows, int C, int64_t ld_src, uint16_t* dst, int64_t ld,
                  hipStream_t s);
// the same attention fed with the fp32 q|k|v matrix (planes produced inside the kernel)
int launch_attention3f(e2emv_ctx* ctx, int B, int T, int n_rows, const int* n_valid_img, int D, int H, const float* qkv,
                       int cross, float* out32, hipStream_t s, bool h2 = false);
int launch_attention3(e2emv_ctx* ctx, int B, int T, int n_rows, const int* n_valid_img, int D, int H, const uint16_t* qk,
                      const uint16_t* vt, int cross, uint16_t* out3, float* out32, hipStream_t s);

// Sinkhorn on an internal score buffer S [n_groups * group_batch][M][ldS] (ldS % 4 == 0).  Batch
// element bb belongs to output group bb / group_batch (= the image pair of a tuple): each group
// has its own dense logZ [group_batch][M+1][N+1] (optional) and match outputs.
constexpr int kMaxGroups = E2EMV_MAX_TUPLE * (E2EMV_MAX_TUPLE - 1) / 2;
struct SinkhornOut {
    int n_groups = 1;
    int group_batch = 0;  // 0: = B (single group)
    float* logZ[kMaxGroups] = {nullptr};
    int64_t* m0[kMaxGroups] = {nullptr};
    int64_t* m1[kMaxGroups] = {nullptr};
    float* ms0[kMaxGroups] = {nullptr};
    float* ms1[kMaxGroups] = {nullptr};
};
size_t sinkhorn_ws_bytes(int B, int M, int N);
int launch_sinkhorn(e2emv_ctx* ctx, int B, int M, int N, const float* S, int64_t ldS, float alpha, int iters,
                    float match_thr, const SinkhornOut& out, char* ws, hipStream_t s);

}  // namespace e2emv
