// Model of one k_wgrad_plw step (conv_wgrad_pl.hip) without the convolution around it, for what-if timing:
// per 32-pixel step a workgroup of 8 waves DMAs 72 KB (9 x 1 KB per wave, global_load_lds_dwordx4) into one of two LDS stage
// buffers, waits, barriers, and each wave multiplies a 64 x 64 block: 8 blocks of 6 chained v_mfma_f32_32x32x16_bf16 fed by
// 6 ds_read_b64_tr_b16 pairs per block read one block ahead.  Switches remove or move one ingredient at a time.
//   hipcc -O3 --offload-arch=gfx950 tools/wgrad_model.hip -o wgrad_model && ./wgrad_model
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
typedef __attribute__((address_space(3))) void* lds_vptr;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

// DMA: 0 none, 1 burst after the barrier, 2 one after each block, 3 waves 0-3 only (18 each)
// RD: 0 no LDS reads, 1 transposing reads one block ahead
// SMALL: 1 -> v_mfma_f32_16x16x32_bf16 (twice as many, same FLOPs)
template <int DMA, int RD, int SMALL, int BK = 0>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, float* sink, const unsigned char* src, int steps, long long wg_stride, long long step_stride) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * 73728];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_vptr)smem;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    for (int i = t; i < 2 * 73728 / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0x3c003c00u + i, 0x3d003d00u, 0x3c803c80u, 0x3d803d80u);
    __syncthreads();
    const unsigned char* gp0 = src + (size_t)blockIdx.x * wg_stride + (size_t)lane * 16 + (size_t)wave * 1024;
    const unsigned char* gp = gp0;
    f32x16 acc[4];
    f32x4 acs[16];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 16; ++i) acs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 F[4][3];
    for (int g = 0; g < 4; ++g)
        for (int p = 0; p < 3; ++p) F[g][p] = make_uint4(0x3c003c00u + lane, 0x3d003d00u, 0x3c803c80u + g, 0x3d803d80u + p);
    const int tg = lane >> 4, ti = lane & 15, tq = ti >> 2, tp = ti & 3, th = tg >> 1, tc = (tg & 1) * 16 + tp * 4;
    auto frag = [&](int st, int img, int pl, int ks, int col0) __attribute__((always_inline)) -> uint4 {
        const unsigned char* p = smem + st * 73728 + (img * 3 + pl) * 8192 + (ks * 16 + th * 8 + tq) * 256 + (((col0 + tc) * 2) ^ (tq * 64));
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 1024));
        uint2 a2 = __builtin_bit_cast(uint2, lo), b2 = __builtin_bit_cast(uint2, hi);
        return make_uint4(a2.x, a2.y, b2.x, b2.y);
    };
    auto ld = [&](int st, int g, int ks) __attribute__((always_inline)) {       // group g: 0,1 = A blocks, 2,3 = B blocks
        if (RD == 0) return;
#pragma unroll
        for (int p = 0; p < 3; ++p) F[g][p] = frag(st, g < 2 ? 0 : 1 + (wave & 1), p, ks, (g & 1) * 32 + (wave >> 1) * 16);
    };
    auto mm = [&](int mb, int nb) __attribute__((always_inline)) {
        if (SMALL) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                f32x4 c = acs[(mb * 2 + nb) * 4 + h];
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, F[mb][q]), __builtin_bit_cast(bf16x8, F[2 + nb][(q + h) % 3]), c, 0, 0, 0);
                acs[(mb * 2 + nb) * 4 + h] = c;
            }
        } else {
            f32x16 c = acc[mb * 2 + nb];
#pragma unroll
            for (int q = 0; q < 6; ++q)
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F[mb][q % 3]), __builtin_bit_cast(bf16x8, F[2 + nb][(q + 1) % 3]), c, 0, 0, 0);
            acc[mb * 2 + nb] = c;
        }
    };
    // bookkeeping of the kernel's next(): pixel index -> three gather addresses with bounds selects (64-bit multiplies)
    int kc = 0, pw = lane & 31, ph = wave, pn = 0;
    const unsigned char* g3[3] = {gp, gp + 8192, gp + 16384};
    long long st3[3] = {8192, 8192, 8192};
    const int ldx = 1024 + (steps & 1), Wd = 33 + (steps & 2);
    auto next = [&]() __attribute__((always_inline)) {
        ++kc;
        int ow = pw + 32 % Wd;
        const int c1 = ow >= Wd ? 1 : 0;
        pw = ow - (c1 ? Wd : 0);
        int oh = ph + c1;
        const int c2 = oh >= Wd ? 1 : 0;
        ph = oh - (c2 ? Wd : 0);
        pn += c2;
        const int p = kc * 32 + (lane >> 4) + 4 * wave;
        const bool pin = p < steps * 32;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int ih = ph + i - 1, iw = pw + i - 1;
            const bool v = pin && (unsigned)ih < (unsigned)Wd && (unsigned)iw < (unsigned)Wd;
            const size_t off = ((size_t)((pn * Wd + ih) * Wd + iw) * ldx * 2) & 0xffff;
            g3[i] = v ? gp + (off & ~(size_t)15) % 8192 + (size_t)i * 24576 : gp + (size_t)i * 24576;
            st3[i] = v ? 8192 : 8192;
        }
    };
    auto dma = [&](int i, int st) __attribute__((always_inline)) {
        if (BK) glds16(g3[i / 3] + (i % 3) * st3[i / 3], lds_base + st * 73728 + i * 8192 + wave * 1024);
        else glds16(gp + (size_t)i * 8192, lds_base + st * 73728 + i * 8192 + wave * 1024);
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (DMA) {
#pragma unroll
        for (int i = 0; i < 9; ++i) dma(i, 0);
    }
    int st = 0;
    for (int s = 0; s < steps; ++s) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        gp = gp0 + (size_t)((s + 1) % steps) * step_stride;      // the step after next streams from here
        if (BK == 1) next();
        if (DMA == 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) dma(i, st ^ 1);
        }
        if (DMA == 3 && wave < 4) {
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                dma(i, st ^ 1);
                glds16(gp + (size_t)i * 8192 + 4096, lds_base + (st ^ 1) * 73728 + i * 8192 + (wave + 4) * 1024);
            }
        }
        int slot = 0;
        auto dmas = [&](int blk) __attribute__((always_inline)) {
            if (DMA == 2) {
                const int upto = (9 * (blk + 1) + 7) / 8;
#pragma unroll
                for (; slot < upto; ++slot) dma(slot, st ^ 1);
            }
        };
#define SB() __builtin_amdgcn_sched_barrier(0)
        ld(st, 0, 0); ld(st, 2, 0);
        SB(); ld(st, 3, 0); SB(); mm(0, 0); dmas(0); SB();
        SB(); ld(st, 1, 0); SB(); mm(0, 1); dmas(1); SB();
        SB(); ld(st, 0, 1); SB(); mm(1, 1); dmas(2); SB();
        SB(); ld(st, 3, 1); SB(); mm(1, 0); dmas(3); SB();
        SB(); ld(st, 2, 1); SB(); mm(0, 1); dmas(4); SB();
        SB(); ld(st, 1, 1); SB(); mm(0, 0); dmas(5); SB();
        SB(); mm(1, 0); dmas(6);
        if (BK == 2) next();
        mm(1, 1); dmas(7); SB();
#undef SB
        st ^= 1;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float tot = 0.f;
    for (int i = 0; i < 4; ++i) tot += acc[i][0] + acc[i][9];
    for (int i = 0; i < 16; ++i) tot += acs[i][1];
    if (t == 0 && blockIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
    }
    if (tot == 12345.678f) sink[t] = tot;
}

template <int DMA, int RD, int SMALL, int BK = 0>
static void run(const char* what, unsigned long long* d_out, float* d_sink, const unsigned char* d_src, long long wg_stride = 73728, long long step_stride = 0) {
    const int steps = 400;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL((k<DMA, RD, SMALL, BK>), dim3(256), dim3(512), 0, 0, d_out, d_sink, d_src, steps, wg_stride, step_stride);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL((k<DMA, RD, SMALL, BK>), dim3(256), dim3(512), 0, 0, d_out, d_sink, d_src, steps, wg_stride, step_stride);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] / 100.0) / 1e3;
    const double flop = 20.0 * 256 * 8 * (double)steps * 48 * 32768.0;
    printf("%-64s %5.0f TFLOP/s  clock %.2f GHz  %6.0f cycles per step (3072 = matrix pipe)\n", what, flop / (ms * 1e-3) * 1e-12, ghz,
           (double)h[0] / steps);
}

int main() {
    unsigned long long* d_out;
    float* d_sink;
    unsigned char* d_src;
    const size_t nsrc = (size_t)256 * 400 * 73728 + 65536;      // 7.5 GB: every step of every workgroup its own 72 KB
    if (hipMalloc(&d_out, 64) != hipSuccess || hipMalloc(&d_sink, 4096) != hipSuccess || hipMalloc(&d_src, nsrc) != hipSuccess) return 1;
    hipMemset(d_src, 0x3c, nsrc);
    run<0, 0, 0>("MFMA + barrier only", d_out, d_sink, d_src);
    run<0, 1, 0>("+ transposing LDS reads one block ahead", d_out, d_sink, d_src);
    run<1, 0, 0>("+ DMA burst (no LDS reads)", d_out, d_sink, d_src);
    run<1, 1, 0>("+ both (the kernel's step)", d_out, d_sink, d_src);
    run<2, 1, 0>("both, DMA dealt over the blocks", d_out, d_sink, d_src);
    run<3, 1, 0>("both, DMA issued by waves 0-3 only", d_out, d_sink, d_src);
    run<1, 1, 0, 1>("kernel's step + address bookkeeping after the barrier", d_out, d_sink, d_src);
    run<1, 1, 0, 2>("kernel's step + address bookkeeping inside the last blocks", d_out, d_sink, d_src);
    run<2, 1, 0, 2>("DMA dealt + address bookkeeping inside the last blocks", d_out, d_sink, d_src);
    run<1, 1, 0>("kernel's step, operands streamed from a 470-MB footprint (8 steps per workgroup wrap)", d_out, d_sink, d_src, 8LL * 73728 * 25, 73728LL * 25 / 25);
    run<1, 1, 0>("kernel's step, every step its own 72 KB of a 7.5-GB buffer (HBM)", d_out, d_sink, d_src, 400LL * 73728, 73728);
    run<2, 1, 0>("DMA dealt, every step its own 72 KB (HBM)", d_out, d_sink, d_src, 400LL * 73728, 73728);
    run<0, 0, 1>("16x16x32: MFMA + barrier only", d_out, d_sink, d_src);
    run<1, 1, 1>("16x16x32: DMA burst + LDS reads", d_out, d_sink, d_src);
    run<2, 1, 1>("16x16x32: DMA dealt + LDS reads", d_out, d_sink, d_src);
    return 0;
}
