// stream_lds.hip — HBM read bandwidth of MI355X through LDS-DMA (global_load_lds_dwordx4) for the access patterns a streaming FIR
// front end can use.  Round-3 question: the phase-lane front end (k_decim_pl) tops out at ~5.1 TB/s with per-wave 400-byte VGPR
// loads; what do (a) wave-private LDS rings filled by 1 KiB DMA pieces and (b) workgroup-cooperative contiguous tiles reach,
// as a function of pieces in flight, waves per CU, segment length and the nt policy?
// build: hipcc -O3 --offload-arch=gfx950 -o stream_lds stream_lds.hip ; run: ./stream_lds [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// one LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS[lds_dst + 16 lane]; M0 saved and restored in the statement
template <int NT>
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst)
{
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// ---- reference patterns (VGPR loads) -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void p0(const float4* in, size_t n4, float* out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
// the k_decim_pl pattern: per-wave sequential, 50 lanes x 8 B per load, U loads in flight; unit u = (stream, segment), consecutive
// units = consecutive segments of one stream
template <int U>
__global__ __launch_bounds__(256) void pseq50(const float2* in, size_t seg_samples, size_t nunits, float* out)
{
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nunits) return;
    const int lane = threadIdx.x & 63, l = lane < 50 ? lane : 49;
    const float2* p = in + w * seg_samples + l;
    const size_t nblk = seg_samples / 50;
    float acc = 0.f;
    float2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[(size_t)u * 50];
    for (size_t t = 0; t < nblk; t += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc = fmaf(v[u].x, v[u].y, acc);
            const size_t tn = t + u + U < nblk ? t + u + U : nblk - 1;
            v[u] = p[tn * 50];
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// ---- W1: wave-private LDS ring, per-wave sequential segment ---------------------------------------------------------------------
// Every wave streams its own segment of `pieces` 1 KiB pieces through a private ring of RP pieces with PD pieces in flight, and
// reads each landed piece back as the phase-lane kernel would (ds_read_b64, lane-contiguous) + KF dependent FMAs per sample.
template <int PD, int RP, int NT, int KF, int NW>
__global__ __launch_bounds__(NW * 64) void w1(const unsigned char* in, size_t pieces, size_t nunits, float* out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    static_assert((RP & (RP - 1)) == 0 && RP > PD, "ring");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t u = (size_t)blockIdx.x * NW + wave;
    if (u >= nunits) return;
    unsigned char* ring = smem + wave * (RP * 1024);
    const uint32_t rbase = (uint32_t)(uintptr_t)(lds_ptr_t)ring;
    const unsigned char* g = in + u * pieces * 1024 + lane * 16;
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < PD; ++q) glds16<NT>(g + (size_t)(q < (int)pieces ? q : 0) * 1024, rbase + q * 1024);
    for (size_t t = 0; t < pieces; ++t) {
        const size_t tn = t + PD < pieces ? t + PD : pieces - 1;          // (re-reads the last piece past the end: harmless)
        glds16<NT>(g + tn * 1024, rbase + (uint32_t)((t + PD) & (RP - 1)) * 1024);
        wait_vm<PD>();
        const unsigned char* pc = ring + (t & (RP - 1)) * 1024 + lane * 8;
        const float2 a = *reinterpret_cast<const float2*>(pc);
        const float2 b = *reinterpret_cast<const float2*>(pc + 512);
#pragma unroll
        for (int k = 0; k < KF; ++k) { acc = fmaf(a.x, a.y, acc); acc = fmaf(b.x, b.y, acc); }
    }
    wait_vm<0>();
    if (acc == 12345.678f) out[0] = acc;
}

// ---- W2: workgroup-cooperative contiguous tiles ---------------------------------------------------------------------------------
// A workgroup of NW waves streams tiles of NW * PPT pieces (contiguous bytes): wave w issues pieces w, w + NW, ... of the tile, so
// the workgroup's DMA requests of one round are adjacent KiB.  NB LDS buffers, ND tiles in flight (ND < NB), one s_barrier per tile.
// mode 0: tile index = blockIdx + k gridDim (the whole chip sweeps one contiguous window); mode 1: workgroup g owns a contiguous run.
template <int PPT, int NB, int ND, int NT, int KF, int NW>
__global__ __launch_bounds__(NW * 64) void w2(const unsigned char* in, size_t ntiles, int mode, float* out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int TILE = NW * PPT * 1024;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t sbase = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    const size_t G = gridDim.x;
    const size_t per = (ntiles + G - 1) / G;
    const size_t nmine = mode == 0 ? (ntiles > blockIdx.x ? (ntiles - blockIdx.x + G - 1) / G : 0)
                                   : (per * blockIdx.x < ntiles ? (per * (blockIdx.x + 1) <= ntiles ? per : ntiles - per * blockIdx.x) : 0);
    auto tile_addr = [&](size_t k) -> const unsigned char* {
        const size_t t = mode == 0 ? blockIdx.x + k * G : per * blockIdx.x + k;
        return in + t * (size_t)TILE;
    };
    auto issue = [&](size_t k) {
        const size_t kk = k < nmine ? k : (nmine ? nmine - 1 : 0);
        const unsigned char* g = tile_addr(kk) + lane * 16;
        const uint32_t dst = sbase + (uint32_t)(k % NB) * TILE;
#pragma unroll
        for (int j = 0; j < PPT; ++j) glds16<NT>(g + (size_t)(j * NW + wave) * 1024, dst + (uint32_t)(j * NW + wave) * 1024);
    };
    float acc = 0.f;
    if (nmine == 0) return;
#pragma unroll
    for (int k = 0; k < ND; ++k) issue(k);
    for (size_t k = 0; k < nmine; ++k) {
        issue(k + ND);
        wait_vm<ND * PPT>();                       // this wave's pieces of tile k have landed
        __builtin_amdgcn_s_barrier();              // ... and everybody else's
        // consume: wave w reads the contiguous 1/NW of the tile (PPT KiB), lane-contiguous 8-byte reads
        const unsigned char* pc = smem + (k % NB) * TILE + wave * (PPT * 1024) + lane * 8;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const float2 a = *reinterpret_cast<const float2*>(pc + j * 1024);
            const float2 b = *reinterpret_cast<const float2*>(pc + j * 1024 + 512);
#pragma unroll
            for (int q = 0; q < KF; ++q) { acc = fmaf(a.x, a.y, acc); acc = fmaf(b.x, b.y, acc); }
        }
        if (NB - ND < 2) __builtin_amdgcn_s_barrier();   // the buffer refilled next round is the one just read
    }
    wait_vm<0>();
    if (acc == 12345.678f) out[0] = acc;
}

// fill with pseudo-random floats of unit scale (xorshift per element): HBM power -- and with it the sustained bandwidth -- may depend
// on the data; the round-3 front ends all stop at 5.06 TB/s on random IQ while constant data streams at 7.07 TB/s
__global__ __launch_bounds__(256) void fill_random(float* p, size_t n, uint32_t seed)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x << 13; x ^= x >> 17; x ^= x << 5; x *= 0x9E3779B1u; x ^= x >> 15;
        p[i] = ((float)(int32_t)x) * (0.05f / 2147483648.0f);
    }
}
static void rep(const char* name, double ms, double useful);
template <class F> static double timeit(F f, int reps);
// ---- W3: the data flow of k_decim_pm: wave-private ring, consumption in GROUPS of GB bytes ---------------------------------------
// A group is copied out of the ring (ds_read_b64 x 13 per lane) once it has landed; that frees its slots and the wave issues the
// next pieces in a burst, then "computes" (CW dependent FMAs per lane) before it waits for the next group.  PFK > 0: every group the
// wave also touches one dword per 128-byte line PFK KiB ahead of the DMA front (a sparse global_load whose result is dropped), so
// that HBM requests are in flight beyond what the LDS ring can hold and the DMA pieces hit L2.
template <int RP, int NT, int CW, int PFK, int NW>
__global__ __launch_bounds__(NW * 64) void w3(const unsigned char* in, size_t seg_bytes, size_t nunits, float* out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr uint32_t GB = 6400;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t u = (size_t)blockIdx.x * NW + wave;
    if (u >= nunits) return;
    unsigned char* ring = smem + wave * (RP * 1024);
    const uint32_t rbase = (uint32_t)(uintptr_t)(lds_ptr_t)ring;
    const unsigned char* base = in + u * seg_bytes;
    const unsigned char* g = base + lane * 16;
    const uint32_t ngrp = (uint32_t)(seg_bytes / GB);
    const uint32_t npieces = (ngrp * GB + 1023u) >> 10;
    uint32_t issued = 0;
    auto issue_upto = [&](uint32_t want) {
        while (issued < want) { glds16<NT>(g + (size_t)issued * 1024, rbase + (issued & (RP - 1)) * 1024u); ++issued; }
    };
    float acc = 0.f, sink = 0.f;
    issue_upto(npieces < RP ? npieces : RP);
    uint32_t gbytes = 0;
    const uint32_t lane_byte = ((uint32_t)(lane & 15) * 400u + (uint32_t)(lane >> 4) * 8u);
    for (uint32_t gi = 0; gi < ngrp; ++gi) {
        uint32_t need = (gbytes + GB + 1023u) >> 10;
        need = need < npieces ? need : npieces;
        switch (issued - need) {
        case 0: wait_vm<0>(); break; case 1: wait_vm<1>(); break; case 2: wait_vm<2>(); break; case 3: wait_vm<3>(); break;
        case 4: wait_vm<4>(); break; case 5: wait_vm<5>(); break; case 6: wait_vm<6>(); break; case 7: wait_vm<7>(); break;
        case 8: wait_vm<8>(); break; case 9: wait_vm<9>(); break; case 10: wait_vm<10>(); break; default: wait_vm<11>(); break;
        }
        float2 x[13];
#pragma unroll
        for (int s2 = 0; s2 < 13; ++s2) x[s2] = *reinterpret_cast<const float2*>(ring + ((gbytes + lane_byte + 32u * s2) & (RP * 1024u - 1u)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t cap = ((gbytes + GB) >> 10) + RP;
        issue_upto(cap < npieces ? cap : npieces);
        if (PFK > 0) {   // sparse touch PFK KiB ahead of the DMA front: 64 lanes x 128 B = 8 KiB of lines per instruction
            const size_t ahead = (size_t)issued * 1024 + (size_t)PFK * 1024;
            if (ahead + 8192 <= seg_bytes) sink += *reinterpret_cast<const volatile float*>(base + ahead + (size_t)lane * 128);
        }
#pragma unroll
        for (int s2 = 0; s2 < 13; ++s2) {
#pragma unroll
            for (int k = 0; k < CW; ++k) acc = fmaf(x[s2].x, x[s2].y, acc);
        }
        gbytes += GB;
    }
    wait_vm<0>();
    if (acc == 12345.678f || sink == 1.2345f) out[0] = acc + sink;
}
// ---- W4: W3's flow with a group of GB bytes, NS read-outs per lane, and per group NM f32 MFMAs (16x16x4, NA independent accumulators)
// beside CW dependent FMAs per read-out: what a phase-major front end for the 25:1 / 100:1 filters (42 block lags = 3 lag tiles)
// would ask of the matrix pipe and of the memory path at the same time.
typedef float f32x4_u __attribute__((ext_vector_type(4)));
template <int RP, int GB, int NS, int NM, int NA, int CW, int NW>
__global__ __launch_bounds__(NW * 64) void w4(const unsigned char* in, size_t seg_bytes, size_t nunits, float* out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t u = (size_t)blockIdx.x * NW + wave;
    if (u >= nunits) return;
    unsigned char* ring = smem + wave * (RP * 1024);
    const uint32_t rbase = (uint32_t)(uintptr_t)(lds_ptr_t)ring;
    const unsigned char* g = in + u * seg_bytes + lane * 16;
    const uint32_t ngrp = (uint32_t)(seg_bytes / GB);
    const uint32_t npieces = (ngrp * (uint32_t)GB + 1023u) >> 10;
    uint32_t issued = 0;
    auto issue_upto = [&](uint32_t want) {
        while (issued < want) { glds16<1>(g + (size_t)issued * 1024, rbase + (issued & (RP - 1)) * 1024u); ++issued; }
    };
    f32x4_u z[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) z[a] = f32x4_u{0.f, 0.f, 0.f, 0.f};
    float acc = 0.f;
    issue_upto(npieces < RP ? npieces : RP);
    uint32_t gbytes = 0;
    const uint32_t lane_byte = ((uint32_t)(lane & 15) * (uint32_t)(GB / 16) + (uint32_t)(lane >> 4) * 8u);
    for (uint32_t gi = 0; gi < ngrp; ++gi) {
        uint32_t need = (gbytes + GB + 1023u) >> 10;
        need = need < npieces ? need : npieces;
        switch (issued - need) {
        case 0: wait_vm<0>(); break; case 1: wait_vm<1>(); break; case 2: wait_vm<2>(); break; case 3: wait_vm<3>(); break;
        case 4: wait_vm<4>(); break; case 5: wait_vm<5>(); break; case 6: wait_vm<6>(); break; case 7: wait_vm<7>(); break;
        case 8: wait_vm<8>(); break; case 9: wait_vm<9>(); break; case 10: wait_vm<10>(); break; case 11: wait_vm<11>(); break;
        case 12: wait_vm<12>(); break; case 13: wait_vm<13>(); break; case 14: wait_vm<14>(); break; default: wait_vm<15>(); break;
        }
        float2 x[NS];
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) x[s2] = *reinterpret_cast<const float2*>(ring + ((gbytes + lane_byte + 32u * s2) & (RP * 1024u - 1u)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t cap = ((gbytes + GB) >> 10) + RP;
        issue_upto(cap < npieces ? cap : npieces);
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
            float v = x[s2].x;
#pragma unroll
            for (int k = 0; k < CW; ++k) v = fmaf(v, x[s2].y, 0.25f);
            acc += v;
#pragma unroll
            for (int k = 0; k < NM / NS; ++k) z[(s2 * (NM / NS) + k) % NA] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, x[s2].y, z[(s2 * (NM / NS) + k) % NA], 0, 0, 0);
        }
        gbytes += GB;
    }
    wait_vm<0>();
    float zs = 0.f;
#pragma unroll
    for (int a = 0; a < NA; ++a) zs += z[a][0] + z[a][1] + z[a][2] + z[a][3];
    if (acc + zs == 12345.678f) out[0] = acc;
}
template <int RP, int GB, int NS, int NM, int NA, int CW, int NW>
static void run_w4(const unsigned char* in, size_t bytes, int pad_kib, float* out)
{
    const auto kern = w4<RP, GB, NS, NM, NA, CW, NW>;
    const size_t lds = (size_t)NW * RP * 1024 + (size_t)pad_kib * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t seg = (size_t)(204800 / GB) * GB;
    const size_t nunits = bytes / seg;
    char nm[200];
    snprintf(nm, sizeof nm, "W4 pm flow: group %d B, ring %d KiB, %d MFMA (%d accumulators) + %d x %d fma per group, %d waves/WG, LDS/WG %zu KiB",
             GB, RP, NM, NA, NS, CW, NW, lds / 1024);
    rep(nm, timeit([&] { hipLaunchKernelGGL(kern, dim3((nunits + NW - 1) / NW), dim3(NW * 64), lds, 0, in, seg, nunits, out); }, 4), (double)nunits * seg);
}
template <int RP, int NT, int CW, int PFK, int NW>
static void run_w3(const unsigned char* in, size_t bytes, int pad_kib, float* out)
{
    const auto kern = w3<RP, NT, CW, PFK, NW>;
    const size_t lds = (size_t)NW * RP * 1024 + (size_t)pad_kib * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t seg = 32 * 6400;   // 200 KiB
    const size_t nunits = bytes / seg;
    char nm[160];
    snprintf(nm, sizeof nm, "W3 pm flow: ring %d KiB, nt %d, %d fma per sample slot, L2 touch %d KiB ahead, %d waves/WG, LDS/WG %zu KiB",
             RP, NT, CW, PFK, NW, lds / 1024);
    rep(nm, timeit([&] { hipLaunchKernelGGL(kern, dim3((nunits + NW - 1) / NW), dim3(NW * 64), lds, 0, in, seg, nunits, out); }, 4), (double)nunits * seg);
}
template <class F> static double timeit(F f, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); CK(hipEventDestroy(a)); CK(hipEventDestroy(b)); return ms / reps;
}
static void rep(const char* name, double ms, double useful) { printf("%-86s %8.3f ms  %7.1f GB/s\n", name, ms, useful / ms / 1e6); fflush(stdout); }

template <int PD, int RP, int NT, int KF, int NW>
static void run_w1(const unsigned char* in, size_t bytes, size_t seg_kib, int pad_kib, float* out)
{
    const auto kern = w1<PD, RP, NT, KF, NW>;
    const size_t lds = (size_t)NW * RP * 1024 + (size_t)pad_kib * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t nunits = bytes / (seg_kib * 1024);
    char nm[160];
    snprintf(nm, sizeof nm, "W1 wave ring: %d in flight, ring %d KiB, nt %d, %d fma/sample, %d waves/WG, seg %zu KiB, LDS/WG %zu KiB",
             PD, RP, NT, 2 * KF / 2, NW, seg_kib, lds / 1024);
    rep(nm, timeit([&] { hipLaunchKernelGGL(kern, dim3((nunits + NW - 1) / NW), dim3(NW * 64), lds, 0, in, seg_kib, nunits, out); }, 4), (double)nunits * seg_kib * 1024);
}
template <int PPT, int NB, int ND, int NT, int KF, int NW>
static void run_w2(const unsigned char* in, size_t bytes, int grid_per_cu, int mode, int pad_kib, float* out)
{
    const auto kern = w2<PPT, NB, ND, NT, KF, NW>;
    constexpr size_t TILE = (size_t)NW * PPT * 1024;
    const size_t lds = NB * TILE + (size_t)pad_kib * 1024;
    if (lds > 160 * 1024) return;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t ntiles = bytes / TILE;
    char nm[160];
    snprintf(nm, sizeof nm, "W2 WG tiles: %d waves x %d KiB = %zu KiB tile, %d bufs, %d in flight, nt %d, %d fma, grid %d/CU, mode %d, LDS %zu KiB",
             NW, PPT, TILE / 1024, NB, ND, NT, KF, grid_per_cu, mode, lds / 1024);
    rep(nm, timeit([&] { hipLaunchKernelGGL(kern, dim3(256 * grid_per_cu), dim3(NW * 64), lds, 0, in, ntiles, mode, out); }, 4), (double)ntiles * TILE);
}

int main(int argc, char** argv)
{
    const size_t gib = argc > 1 ? atoi(argv[1]) : 16;
    const int fill = argc > 2 ? atoi(argv[2]) : 0;      // 0: constant bytes, 1: pseudo-random floats
    const int quick = argc > 3 ? atoi(argv[3]) : 0;     // 1: only the headline patterns
    const size_t bytes = gib << 30;
    unsigned char* in; float* out; CK(hipMalloc(&in, bytes + (4 << 20))); CK(hipMalloc(&out, 256));
    if (fill == 0) CK(hipMemset(in, 0x11, bytes + (4 << 20)));
    else { hipLaunchKernelGGL(fill_random, dim3(256 * 32), dim3(256), 0, 0, (float*)in, (bytes + (4 << 20)) / 4, 12345u); CK(hipDeviceSynchronize()); }
    printf("# %zu GiB, fill = %s\n", gib, fill ? "pseudo-random floats" : "constant 0x11");
    rep("P0 contiguous sweep, dwordx4 to VGPR, grid 256*8", timeit([&] { hipLaunchKernelGGL(p0, dim3(2048), dim3(256), 0, 0, (const float4*)in, bytes / 16, out); }, 4), bytes);
    rep("P0 contiguous sweep, dwordx4 to VGPR, grid 256*16", timeit([&] { hipLaunchKernelGGL(p0, dim3(4096), dim3(256), 0, 0, (const float4*)in, bytes / 16, out); }, 4), bytes);
    {
        const size_t seg = 25600, nunits = bytes / 8 / seg;   // 512 blocks of 50 samples = 200 KiB
        rep("P2 k_decim_pl pattern: wave-seq 50 lanes x 8 B, 8 in flight, seg 200 KB", timeit([&] { hipLaunchKernelGGL(pseq50<8>, dim3((nunits + 3) / 4), dim3(256), 0, 0, (const float2*)in, seg, nunits, out); }, 4), (double)nunits * seg * 8);
    }
    if (quick == 3) {   // phase-major front ends for the long filters: 42 MFMA per 16 blocks of 25 samples, 150 per 16 blocks of 100
        run_w4<8, 6400, 13, 26, 2, 8, 4>(in, bytes, 6, out);      // = k_decim_pm's shape (C1)
        run_w4<8, 3200, 7, 42, 6, 8, 4>(in, bytes, 6, out);       // 25:1, ring holds 2.5 groups
        run_w4<8, 3200, 7, 42, 6, 8, 4>(in, bytes, 0, out);       // 5 WGs/CU
        run_w4<8, 3200, 7, 0, 6, 8, 4>(in, bytes, 6, out);        // no MFMA
        run_w4<8, 3200, 7, 42, 6, 0, 4>(in, bytes, 6, out);       // no VALU work
        run_w4<4, 3200, 7, 42, 6, 8, 4>(in, bytes, 0, out);       // 4 KiB rings: 16 KiB/WG, 8+ WGs/CU
        run_w4<16, 12800, 25, 150, 6, 8, 4>(in, bytes, 6, out);   // 100:1: 12.8 KB groups, 16 KiB rings, 2 WGs/CU
        run_w4<16, 12800, 25, 150, 6, 8, 2>(in, bytes, 3, out);   // 2-wave workgroups: 35 KiB/WG, 4 WGs/CU = 8 waves
        run_w4<16, 12800, 25, 0, 6, 8, 4>(in, bytes, 6, out);
        run_w4<8, 6400, 13, 78, 6, 8, 4>(in, bytes, 6, out);      // 50:1 front end (2091 taps): 78 MFMA per 6400 B
        return 0;
    }
    if (quick == 4) {   // LDS bank conflicts of the raw read-out: block strides of 200 / 800 bytes (D = 25 / 100) against padded strides
                        // of 208 / 816 bytes (4 x odd dwords: the 32 lanes of a half-wave hit 32 different bank pairs); compare GROUPS per second
        run_w4<8, 3200, 7, 42, 6, 8, 4>(in, bytes, 6, out);
        run_w4<8, 3328, 7, 42, 6, 8, 4>(in, bytes, 6, out);
        run_w4<16, 12800, 25, 150, 6, 8, 4>(in, bytes, 6, out);
        run_w4<16, 13056, 25, 150, 6, 8, 4>(in, bytes, 6, out);
        run_w4<8, 3200, 7, 42, 6, 8, 4>(in, bytes, 6, out);
        run_w4<8, 3328, 7, 42, 6, 8, 4>(in, bytes, 6, out);
        return 0;
    }
    if (quick == 2) {   // the data flow of k_decim_pm and ways to keep more bytes in flight
        run_w3<8, 1, 1, 0, 4>(in, bytes, 6, out);      // as k_decim_pm: 4 WGs/CU x 4 waves, 38 KiB/WG
        run_w3<8, 1, 10, 0, 4>(in, bytes, 6, out);
        run_w3<8, 1, 30, 0, 4>(in, bytes, 6, out);
        run_w3<8, 0, 10, 0, 4>(in, bytes, 6, out);
        run_w3<8, 1, 10, 8, 4>(in, bytes, 6, out);     // + L2 touch 8 KiB ahead
        run_w3<8, 1, 10, 16, 4>(in, bytes, 6, out);
        run_w3<8, 1, 10, 32, 4>(in, bytes, 6, out);
        run_w3<8, 1, 30, 16, 4>(in, bytes, 6, out);
        run_w3<8, 0, 10, 16, 4>(in, bytes, 6, out);
        run_w3<16, 1, 10, 0, 4>(in, bytes, 6, out);    // 16 KiB rings: 2 WGs/CU = 8 waves
        run_w3<16, 1, 10, 16, 4>(in, bytes, 6, out);
        run_w3<8, 1, 10, 0, 4>(in, bytes, 0, out);     // 32 KiB/WG: 5 WGs/CU
        run_w3<8, 1, 10, 16, 4>(in, bytes, 0, out);
        return 0;
    }
    if (quick) {
        run_w1<4, 8, 0, 1, 4>(in, bytes, 200, 0, out);
        run_w1<4, 8, 1, 1, 4>(in, bytes, 200, 0, out);
        run_w1<7, 8, 1, 10, 4>(in, bytes, 200, 0, out);
        run_w2<2, 3, 2, 1, 1, 16>(in, bytes, 1, 0, 0, out);
        run_w2<2, 3, 2, 1, 1, 16>(in, bytes, 1, 1, 0, out);
        return 0;
    }
    // W1: in flight / waves per CU (through LDS per WG) / segment length / nt / compute
    run_w1<2, 4, 0, 1, 4>(in, bytes, 200, 0, out);     // 16 KiB/WG: 8+ WGs/CU = 32 waves, 2 KiB in flight each
    run_w1<4, 8, 0, 1, 4>(in, bytes, 200, 0, out);     // 32 KiB/WG: 5 WGs/CU = 20 waves x 4 KiB
    run_w1<4, 8, 1, 1, 4>(in, bytes, 200, 0, out);
    run_w1<4, 8, 0, 1, 4>(in, bytes, 200, 8, out);     // 40 KiB/WG: 4 WGs/CU = 16 waves
    run_w1<4, 8, 0, 1, 4>(in, bytes, 200, 24, out);    // 56 KiB/WG -> 2 WGs/CU = 8 waves
    run_w1<7, 8, 0, 1, 4>(in, bytes, 200, 0, out);     // 7 KiB in flight per wave
    run_w1<7, 8, 1, 1, 4>(in, bytes, 200, 0, out);
    run_w1<7, 8, 0, 1, 4>(in, bytes, 200, 24, out);    // 8 waves/CU x 7 KiB
    run_w1<7, 8, 1, 1, 4>(in, bytes, 200, 24, out);
    run_w1<12, 16, 0, 1, 4>(in, bytes, 200, 0, out);   // 64 KiB/WG: 2 WGs/CU = 8 waves x 12 KiB
    run_w1<12, 16, 1, 1, 4>(in, bytes, 200, 0, out);
    run_w1<4, 8, 0, 1, 4>(in, bytes, 50, 0, out);
    run_w1<4, 8, 0, 1, 4>(in, bytes, 16, 0, out);
    run_w1<7, 8, 0, 1, 4>(in, bytes, 16, 0, out);
    run_w1<4, 8, 0, 10, 4>(in, bytes, 200, 0, out);    // FE-like VALU load: 20 fma per sample
    run_w1<7, 8, 1, 10, 4>(in, bytes, 200, 0, out);
    run_w1<4, 8, 0, 1, 8>(in, bytes, 200, 0, out);     // 8-wave workgroups
    run_w1<4, 8, 0, 1, 16>(in, bytes, 200, 0, out);    // 16-wave workgroups: 128 KiB, 1 WG/CU
    run_w1<4, 8, 1, 1, 16>(in, bytes, 200, 0, out);
    // W2: cooperative tiles
    for (int mode = 0; mode < 2; ++mode) {
        run_w2<1, 3, 2, 0, 1, 4>(in, bytes, 8, mode, 0, out);     // 4 KiB tiles, 12 KiB LDS
        run_w2<2, 3, 2, 0, 1, 4>(in, bytes, 8, mode, 0, out);     // 8 KiB tiles
        run_w2<4, 3, 2, 0, 1, 4>(in, bytes, 4, mode, 0, out);     // 16 KiB tiles, 48 KiB LDS -> 3 WGs/CU
        run_w2<4, 3, 2, 1, 1, 4>(in, bytes, 4, mode, 0, out);
        run_w2<4, 4, 3, 0, 1, 4>(in, bytes, 2, mode, 0, out);     // 64 KiB LDS -> 2 WGs/CU, 48 KiB in flight per WG
        run_w2<2, 3, 2, 0, 1, 8>(in, bytes, 4, mode, 0, out);     // 8 waves x 2 = 16 KiB tiles
        run_w2<2, 3, 2, 1, 1, 8>(in, bytes, 4, mode, 0, out);
        run_w2<2, 4, 3, 0, 1, 8>(in, bytes, 2, mode, 0, out);
        run_w2<1, 3, 2, 0, 1, 16>(in, bytes, 2, mode, 0, out);    // 16 waves x 1 = 16 KiB tiles, 1 KiB per wave
        run_w2<2, 3, 2, 0, 1, 16>(in, bytes, 1, mode, 0, out);    // 32 KiB tiles, 96 KiB LDS, 1 WG/CU = 16 waves
        run_w2<2, 3, 2, 1, 1, 16>(in, bytes, 1, mode, 0, out);
        run_w2<2, 3, 2, 1, 10, 16>(in, bytes, 1, mode, 0, out);   // with FE-like VALU load
        run_w2<2, 4, 3, 1, 1, 16>(in, bytes, 1, mode, 0, out);    // 128 KiB LDS, 64 KiB in flight
        run_w2<2, 4, 2, 1, 1, 16>(in, bytes, 1, mode, 0, out);    // one barrier per tile (NB = ND + 2)
        run_w2<1, 4, 2, 0, 1, 16>(in, bytes, 2, mode, 0, out);
        run_w2<4, 4, 2, 0, 1, 4>(in, bytes, 2, mode, 0, out);
        run_w2<2, 5, 3, 1, 1, 16>(in, bytes, 1, mode, 0, out);    // 160 KiB LDS
    }
    return 0;
}
