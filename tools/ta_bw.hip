// TA / L1 feed-rate micro-benchmark with an L2-resident source: LDS-DMA (or register loads) of row segments
// (SEG bytes per row, rows `pitch` bytes apart) vs contiguous KB pieces.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_vptr;
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int SEG, int MODE>   // MODE 0: glds, 1: register loads
__global__ __launch_bounds__(256) void k(const unsigned char* x, size_t pitch, size_t chunk_step, size_t band_stride, int nbands,
                                         int iters, int pieces, int reps, float* out) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * 24 * 1024];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_vptr)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int LPR = SEG / 16, RPP = 64 / LPR;
    const unsigned char* p = x + (size_t)(blockIdx.x % nbands) * band_stride + (size_t)(lane / LPR) * pitch + (lane % LPR) * 16;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int r = 0; r < reps; ++r)
    for (int it = 0; it < iters; ++it) {
        for (int j = wave; j < pieces; j += 4) {
            const unsigned char* q = p + (size_t)it * chunk_step + (size_t)j * RPP * pitch;
            if (MODE == 0) glds16(q, lds_base + (it & 1) * 24576 + j * 1024);
            else { uint4 v = *(const uint4*)q; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
    }
    if (acc.x == 0x12345 && out) out[0] = acc.y + smem[threadIdx.x];
}
template <int SEG, int MODE>
static float run(int wgs, const unsigned char* x, size_t pitch, size_t cs, size_t bs, int nb, int iters, int pieces, int reps) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k<SEG, MODE>), dim3(wgs), dim3(256), 0, 0, x, pitch, cs, bs, nb, iters, pieces, reps, nullptr);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b);
    }
    return ms;
}
int main() {
    const size_t bytes = 64ull << 20;
    unsigned char* x = nullptr;
    if (hipMalloc(&x, bytes) != hipSuccess || !x) { printf("malloc failed\n"); return 1; }
    if (hipMemset(x, 1, bytes) != hipSuccess) return 1;
    const int pieces = 24, nbands = 2, reps = 4;
    for (int wgs : {256, 512, 768}) for (int mode = 0; mode < 2; ++mode) for (int seg : {64, 128, 1024}) {
        const size_t pitch = seg == 1024 ? 1024 : 4096;
        const int rpp = 1024 / seg;
        const int iters = seg == 1024 ? 32 : (int)(pitch / seg);
        const size_t cs = seg == 1024 ? (size_t)pieces * 1024 : (size_t)seg;
        const size_t bs = seg == 1024 ? (size_t)iters * pieces * 1024 : (size_t)pieces * rpp * pitch;
        const size_t last = (size_t)(nbands - 1) * bs + (size_t)(iters - 1) * cs + (size_t)(pieces - 1) * rpp * pitch + (size_t)(rpp - 1) * pitch + seg;
        if (last > bytes) { printf("skip %d: needs %zu\n", seg, last); continue; }
        float ms;
        if (mode == 0) ms = seg == 64 ? run<64, 0>(wgs, x, pitch, cs, bs, nbands, iters, pieces, reps) : seg == 128 ? run<128, 0>(wgs, x, pitch, cs, bs, nbands, iters, pieces, reps) : run<1024, 0>(wgs, x, pitch, cs, bs, nbands, iters, pieces, reps);
        else ms = seg == 64 ? run<64, 1>(wgs, x, pitch, cs, bs, nbands, iters, pieces, reps) : seg == 128 ? run<128, 1>(wgs, x, pitch, cs, bs, nbands, iters, pieces, reps) : run<1024, 1>(wgs, x, pitch, cs, bs, nbands, iters, pieces, reps);
        const double tot = (double)wgs * reps * iters * pieces * 1024;
        printf("wgs %4d %s seg %4d footprint %.1f MB: %.1f us  %.2f TB/s  %.1f B/clk/CU@2.4GHz\n", wgs, mode ? "regs" : "glds", seg, nbands * bs / 1e6,
               ms * 1e3, tot / ms * 1e-9, tot / 256 / (ms * 1e-3 * 2.4e9));
    }
    return 0;
}
