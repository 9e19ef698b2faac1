#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float identity, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ float wave_sum_dpp(float v, float* steps) {
    v += dpp_move<0xB1, 0xF>(v, v); steps[0*64+threadIdx.x]=v;
    v += dpp_move<0x4E, 0xF>(v, v); steps[1*64+threadIdx.x]=v;
    v += dpp_move<0x141, 0xF>(v, v); steps[2*64+threadIdx.x]=v;
    v += dpp_move<0x140, 0xF>(v, v); steps[3*64+threadIdx.x]=v;
    v += dpp_move<0x142, 0xA>(0.f, v); steps[4*64+threadIdx.x]=v;
    v += dpp_move<0x143, 0xC>(0.f, v); steps[5*64+threadIdx.x]=v;
    return __builtin_amdgcn_readlane(v, 63);
}
__global__ void k(float* out, float* steps) {
    float v = (float)(threadIdx.x + 1);
    out[threadIdx.x] = wave_sum_dpp(v, steps);
}
int main() {
    float *d, *s; hipMalloc(&d, 64*4); hipMalloc(&s, 6*64*4);
    k<<<1,64>>>(d, s);
    float h[64], hs[6*64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); hipMemcpy(hs, s, 6*256, hipMemcpyDeviceToHost);
    printf("total (expect 2080): %g %g\n", h[0], h[63]);
    for (int st = 0; st < 6; ++st) { printf("step %d:", st); for (int l = 0; l < 64; l += 1) printf(" %g", hs[st*64+l]); printf("\n"); }
    return 0;
}
