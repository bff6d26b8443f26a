// Microbenchmark: issue rate of v_mfma_f32_16x16x4_f32 for different accumulator patterns (one wave per SIMD, 4 waves per CU).
// hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PAT>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc)
{
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (PAT == 0) {          // one chain: 4 dependent MFMAs
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            } else if (PAT == 1) {   // two chains alternating
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            } else if (PAT == 2) {   // four chains round robin
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
            } else {                 // two chains in blocks of two
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int PAT> void run(const char* name, int waves_per_simd)
{
    float* out; unsigned long long* cyc; unsigned long long h = 0;
    hipMalloc(&out, 4096 * 256 * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 20000, blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = one per SIMD) x waves_per_simd
    hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, out, 100, cyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 16;
    printf("%-28s waves/SIMD %d: %.1f clk/MFMA/wave (s_memtime %.1f)  %.1f TFLOP/s\n", name, waves_per_simd, ms * 1e-3 * 2.4e9 / n, (double)h / n,
           n * 2048.0 * blocks * 4 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int w = 1; w <= 2; ++w) {
        run<0>("1 chain", w); run<1>("2 chains alternating", w); run<2>("4 chains round robin", w); run<3>("2 chains, blocks of 2", w);
    }
    return 0;
}
