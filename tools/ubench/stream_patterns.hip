// stream_patterns.hip — HBM read bandwidth of MI355X for the access patterns a per-wave sequential FIR can generate.
// build: hipcc -O3 --offload-arch=gfx950 -o stream_patterns stream_patterns.hip ; run: ./stream_patterns [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// P0: classic contiguous sweep, 16 B per lane, grid-stride
__global__ __launch_bounds__(256) void p0(const float4* in, size_t n4, float* out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
// per-wave sequential segments.  LANES active lanes x BYTES per lane per load, U loads in flight.
// map 0: wave w -> segment w (neighbouring waves = neighbouring segments)
// map 1: wave w -> (seg = w / B, stream = w % B), address = stream * row + seg * SEG  (neighbouring waves 'row' bytes apart)
template <int LANES, int VEC, int U>
__global__ __launch_bounds__(256) void pseq(const float* in, size_t seg_floats, size_t nwaves, int map, size_t B, size_t row_floats, float* out)
{
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nwaves) return;
    const int lane = threadIdx.x & 63;
    const int l = lane < LANES ? lane : LANES - 1;
    const float* p = map == 0 ? in + w * seg_floats : in + (w % B) * row_floats + (w / B) * seg_floats;
    const size_t blk = (size_t)LANES * VEC;           // floats per block
    const size_t nblk = seg_floats / blk;
    float acc = 0.f;
    if constexpr (VEC == 2) {
        float2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float2*>(p + (size_t)u * blk + l * 2);
        for (size_t t = 0; t < nblk; t += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc += v[u].x * v[u].y;
                const size_t tn = t + u + U < nblk ? t + u + U : nblk - 1;
                v[u] = *reinterpret_cast<const float2*>(p + tn * blk + l * 2);
            }
        }
    } else {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)u * blk + l * 4);
        for (size_t t = 0; t < nblk; t += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc += v[u].x * v[u].y + v[u].z * v[u].w;
                const size_t tn = t + u + U < nblk ? t + u + U : nblk - 1;
                v[u] = *reinterpret_cast<const float4*>(p + tn * blk + l * 4);
            }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <class F> static double timeit(F f, int reps = 5)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv)
{
    const size_t gib = argc > 1 ? atoi(argv[1]) : 16;
    const size_t bytes = gib << 30, nf = bytes / 4;
    float* in; float* out; CK(hipMalloc(&in, bytes + (1 << 20))); CK(hipMalloc(&out, 256));
    CK(hipMemset(in, 0x11, bytes));
    auto rep = [&](const char* name, double ms, double useful) { printf("%-58s %8.3f ms  %7.1f GB/s\n", name, ms, useful / ms / 1e6); };
    rep("P0 contiguous sweep, dwordx4, grid 256*8", timeit([&] { hipLaunchKernelGGL(p0, dim3(2048), dim3(256), 0, 0, (const float4*)in, nf / 4, out); }), bytes);
    rep("P0 contiguous sweep, dwordx4, grid 256*32", timeit([&] { hipLaunchKernelGGL(p0, dim3(8192), dim3(256), 0, 0, (const float4*)in, nf / 4, out); }), bytes);
    const size_t row = 262144 * 2;          // floats per stream row (262144 cf32)
    const size_t B = nf / row;              // streams
    for (int segk : {200, 50}) {
        const size_t seg400 = (size_t)segk * 1024 / 400 * 100;      // floats, multiple of a 100-float (400 B) block
        {   // 400-byte blocks, 50 lanes x 8 B
            const size_t per_row = row / seg400, nw = per_row * B;
            char nm[128];
            snprintf(nm, sizeof nm, "P2 wave-seq 50 lanes x8B, seg %zu KB, waves in row order", seg400 * 4 / 1024);
            // map 0 with seg = contiguous: treat whole buffer as nw segments (rows not respected; pure pattern test)
            rep(nm, timeit([&] { hipLaunchKernelGGL((pseq<50, 2, 8>), dim3((nw + 3) / 4), dim3(256), 0, 0, in, seg400, nw, 0, B, row, out); }), (double)nw * seg400 * 4);
            snprintf(nm, sizeof nm, "P3 wave-seq 50 lanes x8B, seg %zu KB, neighbours 2 MB apart", seg400 * 4 / 1024);
            rep(nm, timeit([&] { hipLaunchKernelGGL((pseq<50, 2, 8>), dim3((nw + 3) / 4), dim3(256), 0, 0, in, seg400, nw, 1, B, row, out); }), (double)nw * seg400 * 4);
            snprintf(nm, sizeof nm, "P3b same, 16 loads in flight");
            rep(nm, timeit([&] { hipLaunchKernelGGL((pseq<50, 2, 16>), dim3((nw + 3) / 4), dim3(256), 0, 0, in, seg400, nw, 1, B, row, out); }), (double)nw * seg400 * 4);
        }
        {   // 512-byte blocks, 64 lanes x 8 B
            const size_t seg = (size_t)segk * 1024 / 512 * 128, per_row = row / seg, nw = per_row * B;
            rep("P4 wave-seq 64 lanes x8B (aligned 512 B), 2 MB apart", timeit([&] { hipLaunchKernelGGL((pseq<64, 2, 8>), dim3((nw + 3) / 4), dim3(256), 0, 0, in, seg, nw, 1, B, row, out); }), (double)nw * seg * 4);
            rep("P4b same, row order", timeit([&] { hipLaunchKernelGGL((pseq<64, 2, 8>), dim3((nw + 3) / 4), dim3(256), 0, 0, in, seg, nw, 0, B, row, out); }), (double)nw * seg * 4);
        }
        {   // 1 KiB blocks, 64 lanes x 16 B
            const size_t seg = (size_t)segk * 1024 / 1024 * 256, per_row = row / seg, nw = per_row * B;
            rep("P1 wave-seq 64 lanes x16B (1 KiB), 2 MB apart, 8 in flight", timeit([&] { hipLaunchKernelGGL((pseq<64, 4, 8>), dim3((nw + 3) / 4), dim3(256), 0, 0, in, seg, nw, 1, B, row, out); }), (double)nw * seg * 4);
            rep("P1b same, row order", timeit([&] { hipLaunchKernelGGL((pseq<64, 4, 8>), dim3((nw + 3) / 4), dim3(256), 0, 0, in, seg, nw, 0, B, row, out); }), (double)nw * seg * 4);
            rep("P1c 2 MB apart, 4 in flight", timeit([&] { hipLaunchKernelGGL((pseq<64, 4, 4>), dim3((nw + 3) / 4), dim3(256), 0, 0, in, seg, nw, 1, B, row, out); }), (double)nw * seg * 4);
        }
    }
    return 0;
}
