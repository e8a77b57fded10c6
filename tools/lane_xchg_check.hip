// Diagnostic: the DPP / permlane lane exchanges used by the wave reductions against __shfl_xor (the LDS crossbar they replace).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ float dppF(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false)); }
__device__ float xor16(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
    const int lane = threadIdx.x & 63;
    return __int_as_float(((lane >> 4) & 1) ? r[0] : r[1]);
}
__device__ float xor32(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    const int lane = threadIdx.x & 63;
    return __int_as_float((lane >> 5) ? r[0] : r[1]);
}
__global__ void k(float* o) {
    const int lane = threadIdx.x;
    const float v = (float)(lane * 7 % 64) + 0.5f;
    o[lane] = dppF<0xB1>(v) - __shfl_xor(v, 1, 64);
    o[64 + lane] = dppF<0x4E>(v) - __shfl_xor(v, 2, 64);
    o[128 + lane] = dppF<0x124>(v) - __shfl(v, (lane & 48) | ((lane + 4) & 15), 64);
    o[192 + lane] = dppF<0x128>(v) - __shfl_xor(v, 8, 64);
    o[256 + lane] = xor16(v) - __shfl_xor(v, 16, 64);
    o[320 + lane] = xor32(v) - __shfl_xor(v, 32, 64);
}
int main() {
    float* d; hipMalloc(&d, 384 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[6] = {"quad_perm xor1", "quad_perm xor2", "row_ror:4 (either direction is fine for a reduction)", "row_ror:8 = xor8", "permlane16_swap = xor16", "permlane32_swap = xor32"};
    for (int t = 0; t < 6; ++t) { int bad = 0; for (int i = 0; i < 64; ++i) bad += h[t * 64 + i] != 0.0f; printf("%-60s %s (%d lanes differ)\n", nm[t], bad ? "DIFFERS" : "ok", bad); }
    return 0;
}
