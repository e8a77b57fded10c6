// Issue-rate micro-benchmark for the VALU-occupancy figure of DESIGN.md section 3 / bench.py's roofline block.
// Independent streams of one instruction (v_fma_f32, v_fma_f64, v_rcp_f32 + Newton = the division sequence's core, ds_read_b128
// broadcast + 4 v_fma_f32 = the factorisation's inner step) at 1, 2, 4, 6 and 8 waves per SIMD, every CU busy.  Reports the
// shader cycles per wave-instruction per SIMD (s_memtime) so that "VALU issue busy" can be priced with MEASURED costs instead of
// an assumed 4 cycles; run under rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES the same
// launches calibrate what those counters tally per instruction.
//   hipcc --offload-arch=gfx950 -O3 -o tools/issue_rate tools/issue_rate.hip && tools/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITER 2048
#define UNROLL 16

template <int KIND>
__global__ void __launch_bounds__(256) stream_kernel(float* out, unsigned long long* cyc, int iters) {
    float a[UNROLL];
    double d[UNROLL / 2];
    __shared__ __attribute__((aligned(16))) float lds[64];
    if (threadIdx.x < 64) lds[threadIdx.x] = 1.0f + threadIdx.x * 1e-7f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) a[i] = 1.0f + threadIdx.x * 1e-6f + i;
#pragma unroll
    for (int i = 0; i < UNROLL / 2; ++i) d[i] = 1.0 + threadIdx.x * 1e-9 + i;
    const float m = 0.999999f, c = 1e-7f;
    const double md = 0.999999, cd = 1e-9;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (KIND == 1) {
#pragma unroll
            for (int i = 0; i < UNROLL / 2; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(md), "v"(cd));
        } else if (KIND == 2) {
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        } else if (KIND == 3) {   // the factorisation's inner step: one 16-byte broadcast read from LDS, four fma
#pragma unroll
            for (int i = 0; i < UNROLL; i += 4) {
                float4 v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(0));
                asm volatile("s_waitcnt lgkmcnt(0)\n v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %6, %1\n v_fma_f32 %2, %4, %7, %2\n v_fma_f32 %3, %4, %8, %3"
                             : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]) : "v"(m), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            }
        } else if (KIND == 4) {   // v_readlane broadcast + fma (the substitutions' step)
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) { int s; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(a[i])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "s"(s)); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < UNROLL / 2; ++i) s += (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x & 63];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int perIter, int cus) {
    for (int wps : {1, 2, 4, 6, 8}) {
        const int blocks = cus * wps;   // 256 threads = 4 waves = one per SIMD; wps blocks per CU
        float* out; unsigned long long* cyc;
        hipMalloc(&out, sizeof(float) * blocks * 256); hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        stream_kernel<KIND><<<blocks, 256>>>(out, cyc, 64);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        stream_kernel<KIND><<<blocks, 256>>>(out, cyc, ITER);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        const double insts = (double)ITER * perIter;          // wave-instructions per wave
        // cycles the SIMD spends per wave-instruction = wave's elapsed cycles / (its instructions * waves sharing the SIMD)
        printf("%-28s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD (wave sees %.2f), kernel %.3f ms, clock %.2f GHz\n", name, wps, med / insts / wps, med / insts, ms,
               med / (ms * 1e6));
        hipFree(out); hipFree(cyc);
    }
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs\n", p.gcnArchName, cus);
    run<0>("v_fma_f32", UNROLL, cus);
    run<1>("v_fma_f64", UNROLL / 2, cus);
    run<2>("v_rcp_f32", UNROLL, cus);
    run<3>("ds_read_b128 + 4 v_fma_f32", UNROLL / 4 * 5, cus);
    run<4>("v_readlane + v_fma_f32", UNROLL * 2, cus);
    return 0;
}
