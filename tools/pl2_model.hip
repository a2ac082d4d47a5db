// Model of one k_conv_pl2 stage stream (conv_mfma_pl2.hip) without the convolution around it, for what-if timing of the
// tile / wave arrangement.  A workgroup multiplies a (16*RB*WM) x (16*CB*WN) tile per 64-deep stage:
//   waves = WM x WN, each wave RB row blocks x CB column blocks, chains of 6 v_mfma_f32_16x16x32_bf16 per (row block, column
//   block, 32-deep half); the activation stage (rows x 128 B x 3 planes) arrives by LDS-DMA into one of two buffers, its
//   fragments are read with ds_read_b128 two chains ahead and shared by the CB column blocks; the weight fragments
//   (CB x 3 planes x 2 halves per wave) are loaded global -> registers a stage ahead; one s_waitcnt + barrier per stage;
//   every load is issued between two chains.
//   hipcc -O3 --offload-arch=gfx950 tools/pl2_model.hip -o pl2_model && ./pl2_model
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vptr;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

template <int WM, int WN, int RB, int CB, int TPC>      // TPC: workgroups meant to share a CU (LDS budget check only)
__global__ __launch_bounds__(64 * WM * WN, (TPC * WM * WN + 3) / 4) void k(unsigned long long* out, float* sink, const unsigned char* src,
                                                                         const uint4* wsrc, int stages) {
    constexpr int ROWS = 16 * RB * WM, PLANE = ROWS * 128, STAGE = 3 * PLANE, NW = WM * WN;
    constexpr int PIECES = STAGE / 1024, PPW = (PIECES + NW - 1) / NW;      // DMA pieces per wave
    constexpr int NB = CB * 6;                                             // weight loads per wave per stage
    constexpr int CHAINS = 2 * RB * CB;
    static_assert(2 * STAGE * TPC <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_vptr)smem;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN;
    for (int i = t; i < 2 * STAGE / 16; i += 64 * NW) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0x3c003c00u + i, 0x3d003d00u, 0x3c803c80u, 0x3d803d80u);
    __syncthreads();
    const unsigned char* gp = src + (size_t)blockIdx.x * STAGE + (size_t)lane * 16;
    const uint4* wp = wsrc + (size_t)wave * 64 * 64 + lane;
    f32x4 acc[RB][CB];
    for (int i = 0; i < RB; ++i)
        for (int j = 0; j < CB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 bc[CB][2][3], bn[CB][2][3];
    for (int j = 0; j < CB; ++j)
        for (int h = 0; h < 2; ++h)
            for (int p = 0; p < 3; ++p) bc[j][h][p] = bn[j][h][p] = make_uint4(0x3c003c00u + lane, 0x3d003d00u + j, 0x3c803c80u + h, 0x3d803d80u + p);
    uint4 f[3][3];
    const int fbase = wm * (RB * 2048) + (lane & 15) * 128 + (((lane >> 4) ^ ((lane & 15) >> 1)) * 16);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int st = 0, bk = lane;
    for (int s = 0; s < stages; ++s) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < 12; ++q) bk = __builtin_amdgcn_readfirstlane(bk * 1664525 + 1013904223 + q) ^ (bk >> 3);      // stage bookkeeping
        auto aload = [&](int slot, int idx) __attribute__((always_inline)) {      // idx = half * RB + rb
            const int half = idx / RB, rb = idx - half * RB;
            const unsigned char* p = smem + st * STAGE + (fbase ^ (half * 64)) + rb * 2048;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) f[slot][pl] = *reinterpret_cast<const uint4*>(p + pl * PLANE);
        };
        int lslot = 0;
        auto loads = [&](int n) __attribute__((always_inline)) {                  // n load slots of the following stage
#pragma unroll
            for (int q = 0; q < n; ++q, ++lslot) {
                if (lslot < NB) {
                    const int j = lslot / 6, r = lslot % 6;
                    bn[j][r / 3][r % 3] = wp[(size_t)lslot * 64];
                } else if (lslot < NB + PPW) {
                    const int piece = wave + NW * (lslot - NB);
                    if (piece < PIECES) glds16(gp + (size_t)piece * 1024, lds_base + (st ^ 1) * STAGE + piece * 1024);
                }
            }
        };
        constexpr int PER = (NB + PPW + 2 * RB - 1) / (2 * RB);
        aload(0, 0);
        aload(1, 1);
#pragma unroll
        for (int idx = 0; idx < 2 * RB; ++idx) {
            if (idx + 2 < 2 * RB) aload((idx + 2) % 3, idx + 2);
            __builtin_amdgcn_sched_barrier(0);
            const int half = idx / RB, rb = idx - half * RB;
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                f32x4 c = acc[rb][j];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bc[j][half][0]), __builtin_bit_cast(bf16x8, f[idx % 3][2]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bc[j][half][2]), __builtin_bit_cast(bf16x8, f[idx % 3][0]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bc[j][half][1]), __builtin_bit_cast(bf16x8, f[idx % 3][1]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bc[j][half][0]), __builtin_bit_cast(bf16x8, f[idx % 3][1]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bc[j][half][1]), __builtin_bit_cast(bf16x8, f[idx % 3][0]), c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bc[j][half][0]), __builtin_bit_cast(bf16x8, f[idx % 3][0]), c, 0, 0, 0);
                acc[rb][j] = c;
            }
            loads(PER);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int p = 0; p < 3; ++p) bc[j][h][p] = bn[j][h][p];
        st ^= 1;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float tot = (float)bk;
    for (int i = 0; i < RB; ++i)
        for (int j = 0; j < CB; ++j) tot += acc[i][j][0] + acc[i][j][3];
    if (t == 0 && blockIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
    }
    if (tot == 12345.678f) sink[t] = tot;
}

template <int WM, int WN, int RB, int CB, int TPC>
static void run(const char* what, unsigned long long* d_out, float* d_sink, const unsigned char* d_src, const uint4* d_w) {
    const int stages = 400;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * TPC, blk = 64 * WM * WN;
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL((k<WM, WN, RB, CB, TPC>), dim3(grid), dim3(blk), 0, 0, d_out, d_sink, d_src, d_w, stages);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL((k<WM, WN, RB, CB, TPC>), dim3(grid), dim3(blk), 0, 0, d_out, d_sink, d_src, d_w, stages);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] / 100.0) / 1e3;
    const double mfma = 10.0 * grid * (WM * WN) * (double)stages * 2 * RB * CB * 6;
    const double ideal = 2.0 * RB * CB * 6 * 16.0 * (WM * WN * TPC) / 4.0;          // matrix-pipe cycles per stage per SIMD
    printf("%-60s tile %3d x %3d  %5.0f TFLOP/s  clock %.2f GHz  %6.0f cycles per stage (%4.0f = matrix pipe)\n", what, 16 * RB * WM, 16 * CB * WN,
           mfma * 16384.0 / (ms * 1e-3) * 1e-12, ghz, (double)h[0] / stages, ideal);
}

int main() {
    unsigned long long* d_out;
    float* d_sink;
    unsigned char* d_src;
    uint4* d_w;
    const size_t nsrc = (size_t)512 * 61440 + 65536;
    if (hipMalloc(&d_out, 64) != hipSuccess || hipMalloc(&d_sink, 8192) != hipSuccess || hipMalloc(&d_src, nsrc) != hipSuccess ||
        hipMalloc(&d_w, 16 << 20) != hipSuccess)
        return 1;
    hipMemset(d_src, 0x3c, nsrc);
    hipMemset(d_w, 0x3c, 16 << 20);
    run<1, 8, 9, 1, 1>("production: 8 waves x (9 row blocks x 1 column block)", d_out, d_sink, d_src, d_w);
    run<2, 4, 5, 2, 1>("8 waves as 2 x 4, each 5 row blocks x 2 column blocks", d_out, d_sink, d_src, d_w);
    run<1, 4, 9, 2, 1>("4 waves (one per SIMD), each 9 row blocks x 2 column blocks", d_out, d_sink, d_src, d_w);
    run<1, 8, 9, 2, 1>("8 waves, each 9 row blocks x 2 column blocks (256 columns)", d_out, d_sink, d_src, d_w);
    run<1, 4, 5, 2, 2>("two workgroups per CU: 4 waves, 5 row blocks x 2 column blocks", d_out, d_sink, d_src, d_w);
    run<1, 4, 6, 2, 2>("two workgroups per CU: 4 waves, 6 row blocks x 2 column blocks", d_out, d_sink, d_src, d_w);
    return 0;
}
