// What one wave per SIMD hides beside its MFMAs on gfx950 (compile and run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_fill.hip -o /tmp/mfma_fill && /tmp/mfma_fill).
// A 256-thread workgroup per CU (one wave per SIMD, as attention_p2w.hip), a stream of v_mfma_f32_32x32x16_f16 on NCH
// accumulator chains taken round-robin, NF filler instructions of a kind behind every MFMA; everything is asm volatile, so the
// source order is the machine order.  Reports cycles per MFMA slot (s_memtime of one wave; 32 = the matrix pipe's pace).
#include <hip/hip_runtime.h>

#include <cstdio>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// CD: 0 accumulators in AGPRs, 1 in VGPRs.  BSRC: 0 B operand in VGPRs, 1 in AGPRs.
// KIND: 0 v_fma_f32, 1 v_exp_f32, 2 the softmax mix of attention_p2w (fma fma exp | exp add cvt | add mixlo mixhi), 3 one
//       ds_read_b128 + (NF - 1) v_fma_f32, 4 s_nop 0
template <int NCH, int CD, int BSRC, int NF, int KIND>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* ticks, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed + i;
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f16x8 xa, xb;
    for (int e = 0; e < 8; ++e) { xa[e] = (_Float16)(seed + threadIdx.x * 1e-3f + e); xb[e] = (_Float16)(seed - e); }
    f16x8 xb_a;
    asm volatile("; def %0" : "=a"(xb_a));  // (contents do not matter for timing)
    float f[12];
    for (int i = 0; i < 12; ++i) f[i] = seed * 0.001f + i * 0.01f + threadIdx.x * 1e-6f;
    const float k1 = 0.999f, k2 = 1e-3f;
    f32x4 frag = {0.f, 0.f, 0.f, 0.f};
    const unsigned laddr = (threadIdx.x & 63) * 16;
    unsigned pk = 0;
    const unsigned pk0 = 0;  // two fp16 zeros
    const float tiny = 3e-6f + seed * 1e-9f, neg = -100.f - seed;  // 3e-6: an fp16 denormal
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 48; ++s) {
            f32x16& c = acc[s % NCH];
            if (CD == 0 && BSRC == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(xa), "v"(xb));
            if (CD == 0 && BSRC == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(xa), "a"(xb_a));
            if (CD == 1 && BSRC == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(xa), "v"(xb));
            if (CD == 1 && BSRC == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(xa), "a"(xb_a));
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                float& x = f[(s * NF + n) % 12];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(k1), "v"(k2));
                if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                if (KIND == 2) {
                    const int m = (s * NF + n) % 9;
                    if (m == 0 || m == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(k1), "v"(k2));
                    else if (m == 2 || m == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                    else if (m == 4 || m == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(k2));
                    else if (m == 5) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(x), "v"(k1));
                    else if (m == 7) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(pk) : "v"(pk), "v"(x));
                    else asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(pk) : "v"(pk), "v"(x));
                }
                if (KIND == 3) {
                    if (n == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(frag) : "v"(laddr));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(k1), "v"(k2));
                }
                if (KIND == 4) asm volatile("s_nop 0");
                // 5 / 6: v_fma_mixlo_f16 + v_fma_mixhi_f16 with results that are fp16 DENORMALS / normal numbers; 7 / 8: v_cvt_pk_f16_f32
                // the same; 9: v_exp_f32 of a very negative number (result 0)
                if (KIND == 5) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(pk) : "v"(pk0), "v"(tiny));
                if (KIND == 6) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(pk) : "v"(pk0), "v"(k1));
                if (KIND == 7) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(pk) : "v"(tiny));
                if (KIND == 8) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(pk) : "v"(k1));
                if (KIND == 9) asm volatile("v_exp_f32 %0, %1" : "=v"(x) : "v"(neg));
            }
            if (KIND == 3 && s % 6 == 5) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(frag));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 12; ++i) s += f[i];
    s += frag[0] + (float)pk + lds[threadIdx.x];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int NCH, int CD, int BSRC, int NF, int KIND>
void run(float* d, long long* dt) {
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe<NCH, CD, BSRC, NF, KIND><<<256, 256>>>(d, dt, 50, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<NCH, CD, BSRC, NF, KIND><<<256, 256>>>(d, dt, iters, 1.f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    long long t;
    hipMemcpy(&t, dt, sizeof(t), hipMemcpyDeviceToHost);
    static const char* kinds[] = {"v_fma_f32", "v_exp_f32", "softmax mix", "ds_read_b128 + v_fma", "s_nop 0", "fma_mixlo+hi -> f16 denormal", "fma_mixlo+hi -> f16 normal", "cvt_pk_f16 -> denormal", "cvt_pk_f16 -> normal", "v_exp_f32(-100)"};
    printf("chains %d  C/D %s  B %s  %d x %-22s : %6.1f ticks / MFMA slot   (%.3f ms; %.1f ns / slot; %.0f TFLOP/s)\n", NCH, CD ? "VGPR" : "AGPR",
           BSRC ? "AGPR" : "VGPR", NF, kinds[KIND], (double)t / (48.0 * iters), ms, ms * 1e6 / (48.0 * iters), 32768.0 * 48 * iters * 4 * 256 / ms / 1e9);
}

int main() {
    float* d;
    long long* dt;
    hipMalloc(&d, 256 * 256 * sizeof(float));
    hipMalloc(&dt, 64);
    run<2, 0, 0, 2, 5>(d, dt); run<2, 0, 0, 2, 6>(d, dt); run<2, 0, 0, 2, 7>(d, dt); run<2, 0, 0, 2, 8>(d, dt); run<2, 0, 0, 2, 9>(d, dt);
    run<2, 0, 0, 0, 0>(d, dt);
    run<4, 0, 0, 0, 0>(d, dt);
    run<2, 1, 0, 0, 0>(d, dt);
    run<2, 1, 1, 0, 0>(d, dt);
    run<2, 0, 1, 0, 0>(d, dt);
    for (int cd = 0; cd < 2; ++cd) {
        if (cd == 0) {
            run<2, 0, 0, 1, 0>(d, dt); run<2, 0, 0, 2, 0>(d, dt); run<2, 0, 0, 3, 0>(d, dt); run<2, 0, 0, 4, 0>(d, dt); run<2, 0, 0, 5, 0>(d, dt); run<2, 0, 0, 6, 0>(d, dt);
            run<4, 0, 0, 3, 0>(d, dt); run<4, 0, 0, 5, 0>(d, dt);
            run<2, 0, 0, 1, 1>(d, dt); run<2, 0, 0, 2, 1>(d, dt); run<2, 0, 0, 3, 1>(d, dt);
            run<2, 0, 0, 3, 2>(d, dt); run<2, 0, 0, 4, 2>(d, dt); run<4, 0, 0, 3, 2>(d, dt);
            run<2, 0, 0, 1, 3>(d, dt); run<2, 0, 0, 4, 3>(d, dt);
            run<2, 0, 0, 1, 4>(d, dt); run<2, 0, 0, 3, 4>(d, dt);
        } else {
            run<2, 1, 0, 1, 0>(d, dt); run<2, 1, 0, 3, 0>(d, dt); run<2, 1, 0, 5, 0>(d, dt);
            run<2, 1, 1, 3, 2>(d, dt); run<2, 1, 1, 4, 3>(d, dt);
        }
    }
    return 0;
}
