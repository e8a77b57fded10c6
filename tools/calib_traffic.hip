// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for pdbatch's access pattern (MI355X_MICROARCH.md, HBM section:
// "calibrate on a known byte count in your own access pattern").  Three kernels with known bytes per launch:
//   rec_copy : one 64-lane block per 2208-byte record, 16 B per lane in, 16 B per lane out (the step kernel's phases 0/7)
//   rec_read : record in only (sum written by one lane per block)
//   row_write: 26 lanes x 4 B = 104-byte output rows only
// Build: hipcc --offload-arch=gfx950 -O3 tools/calib_traffic.hip -o gpurun_out/calib_traffic ; run under rocprofv3 --pmc.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REC 2208
__global__ void rec_copy(const uint4* __restrict__ in, uint4* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * (REC / 16);
    for (int i = threadIdx.x; i < REC / 16; i += 64) { uint4 v = in[base + i]; v.x += 1; out[base + i] = v; }
}
__global__ void rec_read(const uint4* __restrict__ in, unsigned* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * (REC / 16);
    unsigned s = 0;
    for (int i = threadIdx.x; i < REC / 16; i += 64) { const uint4 v = in[base + i]; s += v.x ^ v.y ^ v.z ^ v.w; }
    if (s == 0x12345678u) out[blockIdx.x] = s;   // practically never
}
__global__ void row_write(float* __restrict__ out, float x) {
    if (threadIdx.x < 26) out[(size_t)blockIdx.x * 26 + threadIdx.x] = x + threadIdx.x;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096, reps = 20;
    uint4 *a, *b; float* rows; unsigned* flag;
    hipMalloc(&a, (size_t)n * REC); hipMalloc(&b, (size_t)n * REC); hipMalloc(&rows, (size_t)n * 104); hipMalloc(&flag, (size_t)n * 4);
    hipMemset(a, 1, (size_t)n * REC); hipMemset(b, 0, (size_t)n * REC);
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(rec_copy, dim3(n), dim3(64), 0, 0, a, b);
        hipLaunchKernelGGL(rec_read, dim3(n), dim3(64), 0, 0, a, flag);
        hipLaunchKernelGGL(row_write, dim3(n), dim3(64), 0, 0, rows, (float)r);
        hipLaunchKernelGGL(rec_copy, dim3(n), dim3(64), 0, 0, a, a);   // in place, like the step kernel
    }
    hipDeviceSynchronize();
    printf("known bytes per launch: rec_copy read %zu write %zu | rec_read read %zu | row_write write %zu\n", (size_t)n * REC, (size_t)n * REC, (size_t)n * REC, (size_t)n * 104);
    return 0;
}
