// MFMA issue-rate micro-benchmark for the structures the conv kernels use (registers only unless LDS is asked for):
// a "stage" = NACC accumulators x CH chained v_mfma_f32_16x16x32_bf16 (or 32x32x16), 8 waves per workgroup, one workgroup per
// CU (two waves per SIMD), with optional extras between the chains -- a scalar branch around a load (BR), a workgroup
// barrier per stage (BAR), ds_read_b128 fragment reads two chains ahead (LDS), scalar/vector bookkeeping per stage (BK).
// Prints wall-clock TFLOP/s, the in-kernel clock and the cycles one SIMD spends per MFMA (16 = 16x16x32 pipe rate).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH, int NACC, bool BR, bool BAR, bool LDS, int BK, int MODE = 0>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, float* sink, const uint4* gsrc, int iters, unsigned seed, int flag) {
    __shared__ uint4 lds[LDS ? 4096 : 64];
    const int lane = threadIdx.x & 63;
    uint4 a[3], b[3];
    for (int i = 0; i < 3; ++i) {
        unsigned v = (seed * (i + 1) * 2654435761u + lane * 40503u) & 0x3f7f3f7f;
        a[i] = make_uint4(v | 0x3c003c00, v ^ 0x01230123, v + 0x00010001, v);
        b[i] = make_uint4(v ^ 0x02460246, v | 0x3d003d00, v, v + 0x00030003);
    }
    if (LDS) {
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = a[i % 3];
        __syncthreads();
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 extra = a[0];
    int bk = seed;
    uint4 f[3][3];
    for (int q = 0; q < 3; ++q)
        for (int p = 0; p < 3; ++p) f[q][p] = b[p];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int it = 0; it < iters; ++it) {
        if (BK > 0 && MODE == 1 && wave < 4) {
#pragma unroll
            for (int q = 0; q < BK; ++q) bk = __builtin_amdgcn_readfirstlane(bk * 1664525 + 1013904223 + q) ^ (bk >> 3);
        }
        if (LDS) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                f[0][p] = lds[(lane + p * 1024) & 4095];
                f[1][p] = lds[(lane + 64 + p * 1024) & 4095];
            }
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (LDS && i + 2 < NACC) {
#pragma unroll
                for (int p = 0; p < 3; ++p) f[(i + 2) % 3][p] = lds[(lane + (i + 2) * 64 + p * 1024) & 4095];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < CH; ++c)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[c % 3]),
                                                               __builtin_bit_cast(bf16x8, LDS ? f[i % 3][c % 3] : b[(c + i) % 3]), acc[i], 0, 0, 0);
            if (BR) {
                if (flag) extra = gsrc[lane + i * 64];
            }
            if (BK > 0 && MODE == 2) {
#pragma unroll
                for (int q = 0; q < (BK + NACC - 1) / NACC; ++q) bk = __builtin_amdgcn_readfirstlane(bk * 1664525 + 1013904223 + q) ^ (bk >> 3);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BK > 0 && (MODE == 0 || (MODE == 1 && wave >= 4))) {
#pragma unroll
            for (int q = 0; q < BK; ++q) bk = __builtin_amdgcn_readfirstlane(bk * 1664525 + 1013904223 + q) ^ (bk >> 3);
        }
        if (BAR) __builtin_amdgcn_s_barrier();
    }
    float tot = (float)bk + __builtin_bit_cast(float, extra.x);
    for (int i = 0; i < NACC; ++i) tot += acc[i][0] + acc[i][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
    }
    if (tot == 12345.678f) sink[threadIdx.x] = tot;
}

template <int CH, int NACC, bool BR, bool BAR, bool LDS, int BK, int MODE = 0>
static void run(const char* what, unsigned long long* d_out, float* d_sink, const uint4* d_src) {
    const int iters = 1000, waves = 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 30; ++rep)
        hipLaunchKernelGGL((k<CH, NACC, BR, BAR, LDS, BK, MODE>), dim3(256), dim3(64 * waves), 0, 0, d_out, d_sink, d_src, iters, 12345u + rep, 0);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 30; ++rep)
        hipLaunchKernelGGL((k<CH, NACC, BR, BAR, LDS, BK, MODE>), dim3(256), dim3(64 * waves), 0, 0, d_out, d_sink, d_src, iters, 777u + rep, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    const double mfmas = 30.0 * 256 * waves * (double)iters * CH * NACC;
    const double ghz = (double)h[0] / ((double)h[1] / 100.0) / 1e3;
    printf("%-58s %5.0f TFLOP/s  clock %.2f GHz  %5.2f cycles per MFMA per SIMD\n", what, mfmas * 16384.0 / (ms * 1e-3) * 1e-12, ghz,
           ms * 1e-3 * ghz * 1e9 / (mfmas / 1024.0));
}

int main() {
    unsigned long long* d_out;
    float* d_sink;
    uint4* d_src;
    if (hipMalloc(&d_out, 64) != hipSuccess || hipMalloc(&d_sink, 4096) != hipSuccess || hipMalloc(&d_src, 1 << 20) != hipSuccess) return 1;
    hipMemset(d_src, 0, 1 << 20);
    run<6, 18, false, false, false, 0>("chains of 6 x 18 accumulators", d_out, d_sink, d_src);
    run<6, 18, true, false, false, 0>("+ scalar branch around a load after every chain", d_out, d_sink, d_src);
    run<6, 18, false, true, false, 0>("+ workgroup barrier per 108 MFMAs", d_out, d_sink, d_src);
    run<6, 18, false, false, false, 60>("+ 60 dependent scalar ops per 108 MFMAs", d_out, d_sink, d_src);
    run<6, 18, false, false, true, 0>("+ 3 ds_read_b128 per chain, two chains ahead", d_out, d_sink, d_src);
    run<6, 18, true, true, true, 60>("all of the above", d_out, d_sink, d_src);
    run<6, 18, false, true, false, 60>("barrier + scalar ops after the MFMAs", d_out, d_sink, d_src);
    run<6, 18, false, true, false, 60, 1>("barrier + scalar ops, anti-phase on the two waves of a SIMD", d_out, d_sink, d_src);
    run<6, 18, false, true, false, 60, 2>("barrier + scalar ops dealt between the chains", d_out, d_sink, d_src);
    run<6, 18, false, true, true, 0>("barrier + LDS reads", d_out, d_sink, d_src);
    run<6, 18, true, true, true, 60, 1>("all, scalar ops anti-phase", d_out, d_sink, d_src);
    run<6, 18, true, true, true, 60, 2>("all, scalar ops dealt between the chains", d_out, d_sink, d_src);
    return 0;
}
