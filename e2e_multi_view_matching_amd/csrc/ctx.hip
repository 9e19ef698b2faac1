// Context lifecycle, memory helpers, weight ingestion (BN folding + head-major re-ordering),
// workspace arena and the HIP-event profiling hooks of libe2emv.so.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "common.h"
#include "p2.h"

namespace e2emv {

int set_err(e2emv_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

int ensure_dynamic_lds(e2emv_ctx* ctx, const void* kernel, size_t bytes) {
    static std::map<std::pair<int, const void*>, size_t> done;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    size_t& have = done[{ctx->device, kernel}];
    if (have >= bytes) return E2EMV_OK;
    E2EMV_HIP(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
    return E2EMV_OK;
}

int ensure_flags(e2emv_ctx* ctx) {
    if (ctx->d_flags) return E2EMV_OK;
    E2EMV_HIP(ctx, hipMalloc((void**)&ctx->d_flags, 256));
    E2EMV_HIP(ctx, hipMemset(ctx->d_flags, 0, 256));
    E2EMV_NULL_STREAM_FENCE(ctx);
    return E2EMV_OK;
}

int ws_reserve(e2emv_ctx* ctx, size_t bytes) {
    // every entry point that carves the arena comes through here first: whatever the previous call left in it (the matched
    // descriptors e2emv_get_descriptors hands out) is about to be overwritten, or freed by the regrow below
    ctx->last_mdesc = nullptr;
    if (bytes <= ctx->ws_bytes) return E2EMV_OK;
    E2EMV_HIP(ctx, hipDeviceSynchronize());
    if (ctx->d_ws) E2EMV_HIP(ctx, hipFree(ctx->d_ws));
    ctx->d_ws = nullptr;
    ctx->ws_bytes = 0;
    size_t want = bytes + bytes / 8 + (size_t(1) << 20);
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) {
        (void)hipGetLastError();
        return set_err(ctx, E2EMV_ENOMEM, "workspace allocation of %zu bytes failed", want);
    }
    // padded rows of activation buffers must start finite (see forward.hip)
    E2EMV_HIP(ctx, hipMemset(p, 0, want));
    E2EMV_NULL_STREAM_FENCE(ctx);
    ctx->d_ws = static_cast<char*>(p);
    ctx->ws_bytes = want;
    return E2EMV_OK;
}

static hipEvent_t take_event(e2emv_ctx* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

void prof_begin(e2emv_ctx* ctx, int slot, hipStream_t s) {
    if (!ctx->prof) return;
    if (!ctx->prof_events.empty() && ctx->prof_events.back().slot == slot && ctx->prof_stream == s) {
        ++ctx->prof_events.back().launches;  // same family as the launch before: the interval goes on
        return;
    }
    ProfEvent pe;
    pe.ev = take_event(ctx);
    pe.slot = slot;
    pe.launches = 1;
    (void)hipEventRecord(pe.ev, s);
    ctx->prof_stream = s;
    ctx->prof_events.push_back(pe);
}

void prof_close(e2emv_ctx* ctx) {
    if (!ctx->prof || ctx->prof_events.empty() || ctx->prof_events.back().slot < 0) return;
    ProfEvent pe;
    pe.ev = take_event(ctx);
    pe.slot = -1;
    pe.launches = 0;
    (void)hipEventRecord(pe.ev, ctx->prof_stream);
    ctx->prof_events.push_back(pe);
}

CallGuard::~CallGuard() { prof_close(c); }

namespace {
inline uint16_t f2h(float f) {
    const _Float16 h = (_Float16)f;  // round to nearest even
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}
inline float h2f(uint16_t u) {
    _Float16 h;
    memcpy(&h, &u, 2);
    return (float)h;
}
}  // namespace

// weights [rows][cols] fp32 -> fp16 planes [rows][{hi, lo}][cols] of 2^s W appended to `out` (the "f16x2" weight format
// of gemm_h2.hip); s = the power of two that brings max |w| into [2^13, 2^14), *out_scale = 2^-s.  With that scale lo
// (and the 2^-11 hi the kernel derives) stay normal fp16 numbers for every |w| >= 2^-16 max |w|.
size_t add_split_h2(std::vector<uint16_t>& out, const std::vector<float>& w, int rows, int cols, float* out_scale) {
    float mx = 0.f;
    for (float v : w) mx = std::max(mx, std::fabs(v));
    int e = 0;
    if (mx > 0.f && std::isfinite(mx)) (void)std::frexp(mx, &e);  // mx = m 2^e, m in [0.5, 1)
    const int sh = 14 - e;
    const float sc = std::ldexp(1.f, sh);
    *out_scale = std::ldexp(1.f, -sh);
    size_t off = (out.size() + 127) & ~size_t(127);
    out.resize(off + (size_t)rows * 2 * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            const float v = w[(size_t)r * cols + c] * sc;
            const uint16_t hi = f2h(v);
            uint16_t* o = &out[off + (size_t)r * 2 * cols];
            o[c] = hi; o[cols + c] = f2h(v - h2f(hi));
        }
    return off;
}

}  // namespace e2emv

using namespace e2emv;

static const char* kProfNames[PS_COUNT] = {"ingest", "gemm", "attention", "score_gemm", "sinkhorn",
                                           "match", "conf", "w8pt", "misc", "gemm_qkv", "gemm_mlp0", "gemm_mlp1", "gemm_chain"};

extern "C" {

int e2emv_version(void) { return E2EMV_ABI_VERSION; }

int e2emv_create(e2emv_ctx** out, int device) {
    if (!out) return E2EMV_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return E2EMV_EHIP;
    }
    if (device < 0 || device >= n) return E2EMV_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return E2EMV_EHIP;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return E2EMV_EHIP;
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) return E2EMV_EHIP;  // kernels are gfx950-only
    e2emv_ctx* ctx = new (std::nothrow) e2emv_ctx();
    if (!ctx) return E2EMV_ENOMEM;
    ctx->device = device;
    ctx->num_cus = p.multiProcessorCount;
    if (dbg_knob("E2EMV_NO_FUSE_MERGE", 0) == 1) ctx->fuse_merge = false;
    if (dbg_knob("E2EMV_B3_PLANES", 0) == 1) ctx->b3_planes = true;
    if (const char* e = getenv("E2EMV_F16X2_KERNELS")) {  // r4: a launch per GEMM (the chain's A/B arm and the T = 5 path)
#ifdef E2EMV_STAMPS
        // the superseded generations are arms of the measurement build only (round 6: the product selects 5, 105 or 4)
        ctx->h2_legacy = strcmp(e, "r2") == 0;
        ctx->attn_wide = strcmp(e, "r3") != 0 && !ctx->h2_legacy;
        if (strcmp(e, "r2") == 0 || strcmp(e, "r3") == 0) ctx->gemm_chain = 0;
#endif
        if (strcmp(e, "r4") == 0) ctx->gemm_chain = 0;
    }

    if (const char* e = getenv("E2EMV_SINKHORN")) {  // the Sinkhorn kernel pin (e2emv_set_sinkhorn_kernel): read here, once
        ctx->sinkhorn_kernel = strcmp(e, "rows64") == 0 ? E2EMV_SINKHORN_ROWS64 : strcmp(e, "rows128") == 0 ? E2EMV_SINKHORN_ROWS128
                               : strcmp(e, "stream") == 0 ? E2EMV_SINKHORN_STREAM : E2EMV_SINKHORN_AUTO;
    }

    // default arithmetic of the dense GNN contractions: the split-operand fp16 x 2 path (22-bit operands, fp32 accumulate;
    // every parity test runs in all three modes at the same bar); E2EMV_PRECISION=bf16x3 selects the 24-bit bf16 x 3
    // split, =f32 the exact fp32-MFMA kernels
    ctx->precision = ctx->fuse_merge ? E2EMV_PRECISION_F16X2 : E2EMV_PRECISION_F32;
    if (const char* e = getenv("E2EMV_PRECISION")) {
        ctx->precision = E2EMV_PRECISION_F32;
        if (ctx->fuse_merge && strcmp(e, "bf16x3") == 0) ctx->precision = E2EMV_PRECISION_BF16X3;
        if (ctx->fuse_merge && strcmp(e, "f16x2") == 0) ctx->precision = E2EMV_PRECISION_F16X2;
    }
    *out = ctx;
    return E2EMV_OK;
}

void e2emv_destroy(e2emv_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    e2emv::train_free(ctx);
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    if (ctx->d_warena) (void)hipFree(ctx->d_warena);
    if (ctx->d_w3arena) (void)hipFree(ctx->d_w3arena);
    if (ctx->d_sparena) (void)hipFree(ctx->d_sparena);
    if (ctx->d_attn_part) (void)hipFree(ctx->d_attn_part);
    if (ctx->d_flags) (void)hipFree(ctx->d_flags);
    if (ctx->d_dummy) (void)hipFree(ctx->d_dummy);
    for (auto& pe : ctx->prof_events) (void)hipEventDestroy(pe.ev);
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    delete ctx;
}

const char* e2emv_last_error(const e2emv_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int e2emv_malloc(e2emv_ctx* ctx, void** d_ptr, size_t bytes) {
    if (!ctx || !d_ptr) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    (void)hipSetDevice(ctx->device);
    if (hipMalloc(d_ptr, bytes ? bytes : 1) != hipSuccess) {
        (void)hipGetLastError();
        return set_err(ctx, E2EMV_ENOMEM, "hipMalloc(%zu) failed", bytes);
    }
    return E2EMV_OK;
}

int e2emv_free(e2emv_ctx* ctx, void* d_ptr) {
    if (!ctx) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    if (d_ptr) E2EMV_HIP(ctx, hipFree(d_ptr));
    return E2EMV_OK;
}

int e2emv_h2d(e2emv_ctx* ctx, void* d_dst, const void* src, size_t bytes, void* stream) {
    if (!ctx || (!d_dst && bytes) || (!src && bytes)) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    E2EMV_HIP(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return E2EMV_OK;
}

int e2emv_d2h(e2emv_ctx* ctx, void* dst, const void* d_src, size_t bytes, void* stream) {
    if (!ctx || (!dst && bytes) || (!d_src && bytes)) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    E2EMV_HIP(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return E2EMV_OK;
}

// Rescue counts the host has just read from the device: `range` problems whose scalings left the exponential-domain kernel's
// range (a property of the model: the second observation moves the context to the log-domain chain), `timeouts` problems given
// up because an inter-workgroup wait ran out (contention, e.g. a co-tenant process: counted, never a reason to demote)
static void note_rescues(e2emv_ctx* ctx, unsigned range, unsigned timeouts) {
    const unsigned zero = 0;
    if (range) {
        (void)hipMemcpy(ctx->d_flags + 3, &zero, sizeof(zero), hipMemcpyHostToDevice);
        ctx->stat_sinkhorn_rescued += range;
        if (++ctx->sk_range_strikes >= 2) { ctx->sinkhorn_stream = true; ctx->sk_stream_calls = 0; }
    }
    if (timeouts) {
        (void)hipMemcpy(ctx->d_flags + 6, &zero, sizeof(zero), hipMemcpyHostToDevice);
        ctx->stat_sinkhorn_rescued += timeouts;
        ctx->stat_sinkhorn_timeouts += timeouts;
    }
}

int e2emv_sync(e2emv_ctx* ctx, void* stream) {
    if (!ctx) return E2EMV_EINVAL;
    E2EMV_ENTER(ctx, stream);
    E2EMV_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    if (ctx->d_flags) {
        // [1]: Sinkhorn problems whose potentials are non-finite even after the log-domain rescue pass = non-finite scores
        // [3]: problems the rescue pass re-solved (finite outputs; a context that keeps needing it moves to the log-domain chain)
        unsigned f[7] = {0, 0, 0, 0, 0, 0, 0};
        E2EMV_HIP(ctx, hipMemcpy(f, ctx->d_flags, sizeof(f), hipMemcpyDeviceToHost));
        note_rescues(ctx, f[3], f[6]);
        if (f[1]) {
            unsigned zero = 0;
            (void)hipMemcpy(ctx->d_flags + 1, &zero, sizeof(zero), hipMemcpyHostToDevice);
            ctx->stat_sinkhorn_bad += f[1];
            return set_err(ctx, E2EMV_EHIP, "sinkhorn: %u problem(s) with non-finite scores (an activation left the range of the arithmetic "
                           "mode, or the inputs were non-finite) - the outputs of those problems are NaN/inf", f[1]);
        }
    }
    E2EMV_NULL_STREAM_FENCE(ctx);  // (the flag resets above)
    return E2EMV_OK;
}

int e2emv_set_weight(e2emv_ctx* ctx, const char* key, const float* data, const int64_t* shape, int ndim) {
    if (!ctx || !key || !data || ndim < 0 || ndim > 4 || (ndim && !shape)) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    std::string k(key);
    if (k.rfind("module.", 0) == 0) k = k.substr(7);
    HostTensor t;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] < 0) return set_err(ctx, E2EMV_ESHAPE, "negative dim in '%s'", key);
        t.shape.push_back(shape[i]);
        n *= shape[i];
    }
    t.data.assign(data, data + n);
    const bool sp = k.rfind("superpoint.", 0) == 0;  // front-end weights (superpoint.hip) live beside the matcher's
    ctx->raw[k] = std::move(t);
    if (sp) ctx->sp_committed = false; else ctx->committed = false;
    return E2EMV_OK;
}

}  // extern "C"

namespace {

struct Packer {
    std::vector<float> host;
    size_t add(const std::vector<float>& v) {
        size_t off = (host.size() + 63) & ~size_t(63);  // 256-B aligned segments
        host.resize(off);
        host.insert(host.end(), v.begin(), v.end());
        return off;
    }
};

// fp32 -> bf16 (round to nearest even) and back, host side - same arithmetic as the device split
inline uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// weights [rows][cols] fp32 -> S3 [rows][3][cols] bf16 planes appended to `out`; returns the offset
size_t add_split3(std::vector<uint16_t>& out, const std::vector<float>& w, int rows, int cols) {
    size_t off = (out.size() + 127) & ~size_t(127);
    out.resize(off + (size_t)rows * 3 * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            const float v = w[(size_t)r * cols + c];
            const uint16_t a = f2bf(v);
            const float r1 = v - bf2f(a);
            const uint16_t b = f2bf(r1);
            const float r2 = r1 - bf2f(b);
            const uint16_t d = f2bf(r2);
            uint16_t* o = &out[off + (size_t)r * 3 * cols];
            o[c] = a; o[cols + c] = b; o[2 * cols + c] = d;
        }
    return off;
}

const HostTensor* find(e2emv_ctx* ctx, const std::string& k) {
    auto it = ctx->raw.find(k);
    return it == ctx->raw.end() ? nullptr : &it->second;
}

// conv weight [out][in](,1) -> checked copy
int get_conv(e2emv_ctx* ctx, const std::string& prefix, int out, int in, std::vector<float>& w,
             std::vector<float>& b) {
    const HostTensor* tw = find(ctx, prefix + ".weight");
    const HostTensor* tb = find(ctx, prefix + ".bias");
    if (!tw || !tb) return set_err(ctx, E2EMV_ESTATE, "missing weight '%s.{weight,bias}'", prefix.c_str());
    if ((int64_t)tw->data.size() != (int64_t)out * in || (int64_t)tb->data.size() != out)
        return set_err(ctx, E2EMV_ESHAPE, "'%s': expected [%d,%d], got %zu elements", prefix.c_str(), out, in,
                       tw->data.size());
    w = tw->data;
    b = tb->data;
    return E2EMV_OK;
}

// fold eval-mode BatchNorm1d `bn` (if present) into conv (w [out][in], b [out])
int fold_bn(e2emv_ctx* ctx, const std::string& bn, int out, int in, std::vector<float>& w, std::vector<float>& b) {
    const HostTensor* mean = find(ctx, bn + ".running_mean");
    if (!mean) return E2EMV_OK;  // fork without BN: nothing to fold
    const HostTensor* var = find(ctx, bn + ".running_var");
    const HostTensor* g = find(ctx, bn + ".weight");
    const HostTensor* be = find(ctx, bn + ".bias");
    if (!var || !g || !be || (int)mean->data.size() != out || (int)var->data.size() != out ||
        (int)g->data.size() != out || (int)be->data.size() != out)
        return set_err(ctx, E2EMV_ESHAPE, "BatchNorm '%s' incomplete or wrong size", bn.c_str());
    for (int o = 0; o < out; ++o) {
        // same association as the unfolded op order: (x - mean) / sqrt(var + eps) * g + beta
        double s = (double)g->data[o] / std::sqrt((double)var->data[o] + 1e-5);
        for (int i = 0; i < in; ++i) w[(size_t)o * in + i] = (float)((double)w[(size_t)o * in + i] * s);
        b[o] = (float)(((double)b[o] - (double)mean->data[o]) * s + (double)be->data[o]);
    }
    return E2EMV_OK;
}

}  // namespace

extern "C" int e2emv_commit_weights(e2emv_ctx* ctx, const e2emv_model_desc* m) {
    if (!ctx || !m) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    const int D = m->desc_dim, H = m->num_heads;
    if (D <= 0 || H <= 0 || D % H != 0 || D / H != 64 || D % 64 != 0)
        return set_err(ctx, E2EMV_ESHAPE, "descriptor_dim %d / num_heads %d: head dim must be 64", D, H);
    if (m->n_kenc < 1 || m->n_kenc > E2EMV_MAX_KENC || m->n_layers < 0 || m->n_layers > E2EMV_MAX_LAYERS)
        return set_err(ctx, E2EMV_ESHAPE, "bad layer counts");
    (void)hipSetDevice(ctx->device);
    const int d = D / H;
    Packer pk;
    int rc;
    std::vector<float> w, b;
    // ---- keypoint encoder ----
    std::vector<int> dims = {3};
    for (int i = 0; i < m->n_kenc; ++i) dims.push_back(m->kenc[i]);
    dims.push_back(D);
    for (size_t i = 1; i + 1 < dims.size(); ++i)
        if (dims[i] % 32 != 0) return set_err(ctx, E2EMV_ESHAPE, "keypoint_encoder width %d not a multiple of 32", dims[i]);
    std::vector<uint16_t> pk3;  // split (bf16 x 3 / fp16 x 2) planes of the GEMM weights
    std::vector<size_t> kw_off, kb_off, kwh_off;
    std::vector<float> kwh_hs;
    const int nk = (int)dims.size() - 1;
    for (int i = 0; i < nk; ++i) {
        std::string p = "kenc.encoder." + std::to_string(3 * i);
        if ((rc = get_conv(ctx, p, dims[i + 1], dims[i], w, b))) return rc;
        if (i < nk - 1 && (rc = fold_bn(ctx, "kenc.encoder." + std::to_string(3 * i + 1), dims[i + 1], dims[i], w, b)))
            return rc;
        kw_off.push_back(pk.add(w));
        kb_off.push_back(pk.add(b));
        kwh_hs.push_back(0.f);
        kwh_off.push_back(dims[i] >= 128 ? add_split_h2(pk3, w, dims[i + 1], dims[i], &kwh_hs.back()) : (size_t)-1);
    }
    // ---- GNN layers ----
    struct LOff {
        size_t wqkv, bqkv, wm, bm, w0, b0, w1, b1;
        size_t w3qkv, w3m0, w3m1, whqkv, whm0, whm1, wpqkv, wpm0, wpm1;
        float hsqkv, hsm0, hsm1, baqkv, bam0, bam1;
    };
    std::vector<LOff> loff(m->n_layers);
    for (int l = 0; l < m->n_layers; ++l) {
        std::string base = "gnn.layers." + std::to_string(l);
        std::vector<float> wqkv((size_t)3 * D * D), bqkv((size_t)3 * D);
        for (int p = 0; p < 3; ++p) {
            if ((rc = get_conv(ctx, base + ".attn.proj." + std::to_string(p), D, D, w, b))) return rc;
            for (int h = 0; h < H; ++h)
                for (int dd = 0; dd < d; ++dd) {
                    int src = dd * H + h, dst = p * D + h * d + dd;  // upstream channel -> head-major
                    memcpy(&wqkv[(size_t)dst * D], &w[(size_t)src * D], sizeof(float) * D);
                    bqkv[dst] = b[src];
                }
        }
        loff[l].wqkv = pk.add(wqkv);
        loff[l].bqkv = pk.add(bqkv);
        loff[l].w3qkv = add_split3(pk3, wqkv, 3 * D, D);
        loff[l].whqkv = add_split_h2(pk3, wqkv, 3 * D, D, &loff[l].hsqkv);
        loff[l].wpqkv = add_split_p2(pk3, wqkv, 3 * D, D, &loff[l].hsqkv);
        loff[l].baqkv = 0.f;
        for (float v : bqkv) loff[l].baqkv = std::max(loff[l].baqkv, std::fabs(v));
        if ((rc = get_conv(ctx, base + ".attn.merge", D, D, w, b))) return rc;
        std::vector<float> wm((size_t)D * D);
        for (int o = 0; o < D; ++o)
            for (int h = 0; h < H; ++h)
                for (int dd = 0; dd < d; ++dd) wm[(size_t)o * D + h * d + dd] = w[(size_t)o * D + dd * H + h];
        loff[l].wm = pk.add(wm);
        loff[l].bm = pk.add(b);
        if ((rc = get_conv(ctx, base + ".mlp.0", 2 * D, 2 * D, w, b))) return rc;
        if ((rc = fold_bn(ctx, base + ".mlp.1", 2 * D, 2 * D, w, b))) return rc;
        if (ctx->fuse_merge) {
            // MLP0([x | merge(o)]) = W0x x + (W0m Wmerge) o + (b0 + W0m bmerge): the merge conv is
            // linear and feeds nothing else, so it is folded into MLP0's second K segment (fp64 on
            // the host).  Saves one GEMM (2 N D^2 flops) and one activation round trip per layer.
            std::vector<double> acc((size_t)2 * D * D, 0.0);
            const float* bmerge = &pk.host[loff[l].bm];
            for (int o = 0; o < 2 * D; ++o) {
                const float* w0m = &w[(size_t)o * 2 * D + D];
                double* ao = &acc[(size_t)o * D];
                double bb = b[o];
                for (int k = 0; k < D; ++k) {
                    const double wk = w0m[k];
                    const float* wr = &wm[(size_t)k * D];
                    for (int c = 0; c < D; ++c) ao[c] += wk * (double)wr[c];
                    bb += wk * (double)bmerge[k];
                }
                b[o] = (float)bb;
            }
            for (int o = 0; o < 2 * D; ++o)
                for (int c = 0; c < D; ++c) w[(size_t)o * 2 * D + D + c] = (float)acc[(size_t)o * D + c];
        }
        loff[l].w0 = pk.add(w);
        loff[l].b0 = pk.add(b);
        loff[l].w3m0 = add_split3(pk3, w, 2 * D, 2 * D);
        loff[l].whm0 = add_split_h2(pk3, w, 2 * D, 2 * D, &loff[l].hsm0);
        loff[l].wpm0 = add_split_p2(pk3, w, 2 * D, 2 * D, &loff[l].hsm0);
        loff[l].bam0 = 0.f;
        for (float v : b) loff[l].bam0 = std::max(loff[l].bam0, std::fabs(v));
        if ((rc = get_conv(ctx, base + ".mlp.3", D, 2 * D, w, b))) return rc;
        loff[l].w1 = pk.add(w);
        loff[l].b1 = pk.add(b);
        loff[l].w3m1 = add_split3(pk3, w, D, 2 * D);
        loff[l].whm1 = add_split_h2(pk3, w, D, 2 * D, &loff[l].hsm1);
        loff[l].wpm1 = add_split_p2(pk3, w, D, 2 * D, &loff[l].hsm1);
        loff[l].bam1 = 0.f;
        for (float v : b) loff[l].bam1 = std::max(loff[l].bam1, std::fabs(v));
    }
    if ((rc = get_conv(ctx, "final_proj", D, D, w, b))) return rc;
    size_t wf = pk.add(w), bf = pk.add(b);
    float hs_final = 0.f, hs_conf0 = 0.f;
    const size_t whf = add_split_h2(pk3, w, D, D, &hs_final);
    const size_t wpf = (D % 32 == 0) ? add_split_p2(pk3, w, D, D, &hs_final) : (size_t)-1;
    float ba_final = 0.f, ba_conf0 = 0.f;
    for (float v : b) ba_final = std::max(ba_final, std::fabs(v));
    size_t whc0 = 0, wpc0 = (size_t)-1;
    const HostTensor* bs = find(ctx, "bin_score");
    if (!bs || bs->data.size() != 1) return set_err(ctx, E2EMV_ESTATE, "missing scalar 'bin_score'");
    size_t wc0 = 0, bc0 = 0, wc1 = 0;
    float bc1 = 0.f;
    if (m->conf_mlp) {
        if ((rc = get_conv(ctx, "conf_mlp.0", D, 2 * D, w, b))) return rc;
        if ((rc = fold_bn(ctx, "conf_mlp.1", D, 2 * D, w, b))) return rc;
        wc0 = pk.add(w);
        bc0 = pk.add(b);
        whc0 = add_split_h2(pk3, w, D, 2 * D, &hs_conf0);
        if (D % 32 == 0) wpc0 = add_split_p2(pk3, w, D, 2 * D, &hs_conf0);
        for (float v : b) ba_conf0 = std::max(ba_conf0, std::fabs(v));
        if ((rc = get_conv(ctx, "conf_mlp.3", 1, D, w, b))) return rc;
        wc1 = pk.add(w);
        bc1 = b[0];
    }
    // ---- upload ----
    if (pk.host.size() > ctx->warena_floats) {
        E2EMV_HIP(ctx, hipDeviceSynchronize());
        if (ctx->d_warena) E2EMV_HIP(ctx, hipFree(ctx->d_warena));
        ctx->d_warena = nullptr;
        ctx->warena_floats = 0;
        void* p = nullptr;
        if (hipMalloc(&p, pk.host.size() * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            return set_err(ctx, E2EMV_ENOMEM, "weight arena allocation failed");
        }
        ctx->d_warena = (float*)p;
        ctx->warena_floats = pk.host.size();
    }
    if (pk3.size() > ctx->w3arena_elems) {
        E2EMV_HIP(ctx, hipDeviceSynchronize());
        if (ctx->d_w3arena) E2EMV_HIP(ctx, hipFree(ctx->d_w3arena));
        ctx->d_w3arena = nullptr;
        ctx->w3arena_elems = 0;
        void* p3 = nullptr;
        if (hipMalloc(&p3, std::max<size_t>(pk3.size(), 1) * sizeof(uint16_t)) != hipSuccess) {
            (void)hipGetLastError();
            return set_err(ctx, E2EMV_ENOMEM, "bf16x3 weight arena allocation failed");
        }
        ctx->d_w3arena = (uint16_t*)p3;
        ctx->w3arena_elems = pk3.size();
    }
    E2EMV_HIP(ctx, hipDeviceSynchronize());  // no forward may be in flight while weights change
    E2EMV_HIP(ctx, hipMemcpy(ctx->d_warena, pk.host.data(), pk.host.size() * sizeof(float), hipMemcpyHostToDevice));
    if (!pk3.empty())
        E2EMV_HIP(ctx, hipMemcpy(ctx->d_w3arena, pk3.data(), pk3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    float* base = ctx->d_warena;
    ctx->kenc_dims = dims;
    ctx->kenc_w0 = base + kw_off[0];
    ctx->kenc_b0 = base + kb_off[0];
    ctx->kenc_w.clear();
    ctx->kenc_b.clear();
    ctx->kenc_wh.clear();
    ctx->kenc_hs.clear();
    for (int i = 1; i < nk; ++i) {
        ctx->kenc_w.push_back(base + kw_off[i]);
        ctx->kenc_b.push_back(base + kb_off[i]);
        ctx->kenc_wh.push_back(kwh_off[i] == (size_t)-1 ? nullptr : ctx->d_w3arena + kwh_off[i]);
        ctx->kenc_hs.push_back(kwh_hs[i]);
    }
    ctx->wh_final = ctx->d_w3arena + whf;
    ctx->hs_final = hs_final;
    ctx->wh_conf0 = m->conf_mlp ? ctx->d_w3arena + whc0 : nullptr;
    ctx->hs_conf0 = hs_conf0;
    ctx->wp_final = wpf == (size_t)-1 ? nullptr : ctx->d_w3arena + wpf;
    ctx->wp_conf0 = (m->conf_mlp && wpc0 != (size_t)-1) ? ctx->d_w3arena + wpc0 : nullptr;
    ctx->ba_final = ba_final;
    ctx->ba_conf0 = ba_conf0;
    ctx->layers.assign(m->n_layers, LayerWeights());
    for (int l = 0; l < m->n_layers; ++l) {
        LayerWeights& L = ctx->layers[l];
        L.w_qkv = base + loff[l].wqkv;
        L.b_qkv = base + loff[l].bqkv;
        L.w_merge = base + loff[l].wm;
        L.b_merge = base + loff[l].bm;
        L.w_mlp0 = base + loff[l].w0;
        L.b_mlp0 = base + loff[l].b0;
        L.w_mlp1 = base + loff[l].w1;
        L.b_mlp1 = base + loff[l].b1;
        L.type = m->layer_types[l] ? 1 : 0;
        L.w3_qkv = ctx->d_w3arena + loff[l].w3qkv;
        L.w3_mlp0 = ctx->d_w3arena + loff[l].w3m0;
        L.w3_mlp1 = ctx->d_w3arena + loff[l].w3m1;
        L.wh_qkv = ctx->d_w3arena + loff[l].whqkv; L.hs_qkv = loff[l].hsqkv;
        L.wh_mlp0 = ctx->d_w3arena + loff[l].whm0; L.hs_mlp0 = loff[l].hsm0;
        L.wh_mlp1 = ctx->d_w3arena + loff[l].whm1; L.hs_mlp1 = loff[l].hsm1;
        L.wp_qkv = ctx->d_w3arena + loff[l].wpqkv;
        L.wp_mlp0 = ctx->d_w3arena + loff[l].wpm0;
        L.wp_mlp1 = ctx->d_w3arena + loff[l].wpm1;
        L.ba_qkv = loff[l].baqkv; L.ba_mlp0 = loff[l].bam0; L.ba_mlp1 = loff[l].bam1;
    }
    ctx->w_final = base + wf;
    ctx->b_final = base + bf;
    ctx->bin_score = bs->data[0];
    if (m->conf_mlp) {
        ctx->w_conf0 = base + wc0;
        ctx->b_conf0 = base + bc0;
        ctx->w_conf1 = base + wc1;
        ctx->b_conf1 = bc1;
    } else {
        ctx->w_conf0 = ctx->b_conf0 = ctx->w_conf1 = nullptr;
    }
    ctx->model = *m;
    ctx->committed = true;
    E2EMV_NULL_STREAM_FENCE(ctx);  // (the arenas and the weight planes uploaded above)
    return E2EMV_OK;
}

extern "C" {

int e2emv_set_precision(e2emv_ctx* ctx, int precision) {
    if (!ctx) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    if (precision != E2EMV_PRECISION_F32 && precision != E2EMV_PRECISION_BF16X3 && precision != E2EMV_PRECISION_F16X2)
        return set_err(ctx, E2EMV_EINVAL, "unknown precision %d", precision);
    if (precision != E2EMV_PRECISION_F32 && !ctx->fuse_merge)
        return set_err(ctx, E2EMV_ESTATE, "bf16x3 needs the merge conv folded into MLP0 (unset E2EMV_NO_FUSE_MERGE)");
    ctx->precision = precision;
    return E2EMV_OK;
}

int e2emv_get_stats(e2emv_ctx* ctx, uint64_t* stats, int n, int reset) {
    if (!ctx || !stats || n < 0) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    (void)hipSetDevice(ctx->device);
    E2EMV_HIP(ctx, hipDeviceSynchronize());
    unsigned f[7] = {0, 0, 0, 0, 0, 0, 0};
    if (ctx->d_flags) E2EMV_HIP(ctx, hipMemcpy(f, ctx->d_flags, sizeof(f), hipMemcpyDeviceToHost));
    ctx->stat_sinkhorn_bad += f[1];
    if (ctx->d_flags) note_rescues(ctx, f[3], f[6]);
    const uint64_t v[6] = {(uint64_t)f[2], ctx->stat_sinkhorn_bad, ctx->stat_sinkhorn_rescued, (uint64_t)f[5], ctx->stat_sinkhorn_timeouts, ctx->stat_sinkhorn_rows128};
    for (int i = 0; i < n; ++i) stats[i] = i < 6 ? v[i] : 0;
    if (ctx->d_flags && f[1]) E2EMV_HIP(ctx, hipMemset(ctx->d_flags + 1, 0, sizeof(unsigned)));  // moved into the host-side total
    if (reset) {
        ctx->stat_sinkhorn_bad = 0;
        ctx->stat_sinkhorn_rescued = 0;
        ctx->stat_sinkhorn_timeouts = 0;
        ctx->stat_sinkhorn_rows128 = 0;
        ctx->sk_range_strikes = 0;
        ctx->sk_stream_calls = 0;
        ctx->sinkhorn_stream = false;  // (a reset also returns the Sinkhorn to the resident kernel)
        if (ctx->d_flags) E2EMV_HIP(ctx, hipMemset(ctx->d_flags + 2, 0, sizeof(unsigned)));
        if (ctx->d_flags) E2EMV_HIP(ctx, hipMemset(ctx->d_flags + 5, 0, sizeof(unsigned)));
    }
    E2EMV_NULL_STREAM_FENCE(ctx);
    return E2EMV_OK;
}

int e2emv_set_f16x2_kernels(e2emv_ctx* ctx, int generation) {
    if (!ctx) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    const bool always = generation == 105;  // generation 5 with the GEMM chain on every shape that allows it (tests, A/B runs)
    if (always) generation = 5;
    if (generation < 2 || generation > 5) return set_err(ctx, E2EMV_EINVAL, "f16x2 kernel generation %d (2 .. 5)", generation);
#ifndef E2EMV_STAMPS
    if (generation < 4)
        return set_err(ctx, E2EMV_EINVAL, "f16x2 kernel generations 2 and 3 are A/B arms of the measurement build (tools/p2_stamps.py --build); this "
                                          "library selects 5 (default), 105 or 4");
#endif
    ctx->h2_legacy = generation == 2;
    ctx->attn_wide = generation >= 4;
    ctx->gemm_chain = generation >= 5 ? (always ? 2 : 1) : 0;
    return E2EMV_OK;
}

int e2emv_set_attention_key_split(e2emv_ctx* ctx, int on) {
    if (!ctx) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    ctx->attn_key_split = on != 0;
    return E2EMV_OK;
}

int e2emv_set_split_min_rows(e2emv_ctx* ctx, int64_t min_rows) {
    if (!ctx) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    ctx->split_min_rows = min_rows < 0 ? -1 : min_rows;
    return E2EMV_OK;
}

int e2emv_get_precision(e2emv_ctx* ctx, int* precision) {
    if (!ctx || !precision) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    *precision = ctx->precision;
    return E2EMV_OK;
}

int e2emv_profile(e2emv_ctx* ctx, int enable) {
    if (!ctx) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    if (!enable) prof_close(ctx);
    ctx->prof = enable != 0;
    return E2EMV_OK;
}

int e2emv_profile_read(e2emv_ctx* ctx, float* ms, int64_t* launches, int n_slots, int reset) {
    if (!ctx) return E2EMV_EINVAL;
    E2EMV_LOCK(ctx);
    prof_close(ctx);
    E2EMV_HIP(ctx, hipDeviceSynchronize());
    for (size_t i = 0; i < ctx->prof_events.size(); ++i) {
        const ProfEvent& pe = ctx->prof_events[i];
        float t = 0.f;
        if (pe.slot >= 0 && pe.slot < E2EMV_PROF_SLOTS && i + 1 < ctx->prof_events.size()) {
            if (hipEventElapsedTime(&t, pe.ev, ctx->prof_events[i + 1].ev) == hipSuccess) {
                ctx->prof_ms[pe.slot] += t;
                ctx->prof_n[pe.slot] += pe.launches;
            } else {
                (void)hipGetLastError();
            }
        }
        ctx->event_pool.push_back(pe.ev);
    }
    ctx->prof_events.clear();
    for (int i = 0; i < n_slots && i < E2EMV_PROF_SLOTS; ++i) {
        if (ms) ms[i] = ctx->prof_ms[i];
        if (launches) launches[i] = ctx->prof_n[i];
    }
    if (reset)
        for (int i = 0; i < E2EMV_PROF_SLOTS; ++i) {
            ctx->prof_ms[i] = 0.f;
            ctx->prof_n[i] = 0;
        }
    return E2EMV_OK;
}

const char* e2emv_profile_name(int slot) { return (slot >= 0 && slot < PS_COUNT) ? kProfNames[slot] : ""; }

}  // extern "C"
