// Diagnostic (GPU box): on which SIMD of its CU does wave k of a four-wave workgroup land?  hipcc --offload-arch=gfx950 -O2 tools/simd_census.hip -o /tmp/simd_census
// The kernel has the first pass's footprint (256 threads, 26.8 KB of LDS, ~80 VGPRs are not reproduced -- placement is by wave slot).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void census(unsigned* hist, unsigned* cuHist, int spin) {
    __shared__ float pad[6698];
    const int w = threadIdx.x >> 6;
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID, all 32 bits
    const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, se = (hw >> 13) & 7;
    pad[threadIdx.x] = (float)hw;
    __syncthreads();
    float acc = pad[(threadIdx.x * 7) & 255];
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;   // stay resident so that the CU fills to six workgroups
    if ((threadIdx.x & 63) == 0) { atomicAdd(&hist[w * 4 + simd], 1u); atomicAdd(&cuHist[(se * 16 + cu) & 127], 1u); }
    if (acc == 123.456f) hist[31] = 1;
}
int main() {
    unsigned *h, *c; hipMalloc(&h, 32 * 4); hipMalloc(&c, 128 * 4); hipMemset(h, 0, 128); hipMemset(c, 0, 512);
    census<<<1821, 256>>>(h, c, 20000); hipDeviceSynchronize();
    unsigned hh[32]; hipMemcpy(hh, h, 128, hipMemcpyDeviceToHost);
    for (int w = 0; w < 4; ++w) printf("wave %d of the workgroup: SIMD0 %u SIMD1 %u SIMD2 %u SIMD3 %u\n", w, hh[w * 4], hh[w * 4 + 1], hh[w * 4 + 2], hh[w * 4 + 3]);
    return 0;
}
