// Declarations shared by the implicit-GEMM convolution kernels (conv_mfma.hip: general-K kernels and
// the C entry points; conv_mfma_u.hip: the tap-uniform fast path).
#pragma once
#include <stdlib.h>
#include "common.h"
#include "planes.h"

namespace iswm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Statistics of the BatchNorm backward pass that consumes a data gradient, taken in the data-gradient kernel's epilogue
// (iswm_conv2d_dgrad_pl2_bn): per tile row and channel  sum(dz), sum(dz * xhat)  with  dz = dx * [ReLU pattern],
// xhat = (y - mean) * invstd  of the PRODUCER stage whose activation dx is the gradient of.
struct BnFuse {
    const float* y;            // raw conv output of the producer stage at the data gradient's pixels [P][ldy]
    const float* mean;
    const float* invstd;
    const float* mscale;       // relu == 2: pattern recomputed as (y - mean) * mscale + mshift > 0
    const float* mshift;
    double* part;              // [2][tiles][C]; nullptr = off
    int ldy, relu;
    // relu == 3: the producer stage is a RESIDUAL stage (out = relu(bn(y) + identity)): its ReLU pattern is read from the hi
    // plane of its saved output (`mask`, pitch ldm bf16 elements) and the epilogue stores the MASKED gradient dz = dx * [out > 0]
    // -- that tensor is at once the input of the producer's BatchNorm backward and the gradient of its identity branch
    const unsigned short* mask;
    int ldm;
};

struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    float* stats;
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, dil, ldx, ldy;
    int M;       // GEMM rows
    int Ktot;    // GEMM K (fwd/dgrad) or flattened N (wgrad)
    int MT, NT;  // tile counts
    int nsplit;  // wgrad: K splits
    int psplit;  // wgrad: pixels per split (multiple of 32)
    int accumulate;  // dgrad: dx += result instead of dx = result
    long long xps;   // plane kernels (conv_mfma_pl2*.hip): byte stride between the bf16 planes of the gathered operand
    int porder;                // strided dgrad of the planes kernels: the four parity quarters of the M tiles, heaviest first (2 bits each)
    int abl;                   // timing ablations of the planes kernels (ISWM_PL2_ABL: 1 no weight loads, 2 no activation DMA, 4 no stage barrier, 8 no fragment reads): wrong results by design
    BnFuse bnf;                // planes data gradient: fused BatchNorm-backward statistics (part == nullptr: off)
    unsigned long long* dbg;   // diagnostic builds only (iswm_set_debug_buffer): per-stage s_memtime stamps of workgroup 0, wave 0
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

constexpr int KC_PITCH = 36;  // floats per LDS row of a K-contiguous operand tile (32 + 4 pad)


// ---- bf16x6 helpers (see conv_mfma_x6.hip) ------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                   0, 0, 0);
}

// GEMM row m -> pixel (n, rh, rw) of an N x RH x RW grid.  par == false: row-major.  par == true (data gradient of
// a stride-2 conv): rows are grouped by the PARITY class (rh & 1, rw & 1) of the pixel -- classes (0,0), (0,1),
// (1,0), (1,1), each row-major over (n, rh >> 1, rw >> 1) -- because a tap (kh, kw) only reaches input pixels of one
// parity class: with uniform-parity tiles the per-tile tap culling drops the 3/4 (1x1) or ~5/9..8/9 (3x3) of the taps
// that would gather nothing but zeros.
__device__ __forceinline__ void x6_row_pixel(int m, int N, int RH, int RW, bool par, int& n, int& rh, int& rw) {
    if (!par) {
        const int RHW = RH * RW;
        n = m / RHW;
        const int rem = m - n * RHW;
        rh = rem / RW;
        rw = rem - rh * RW;
        return;
    }
    const int H0 = (RH + 1) >> 1, H1 = RH >> 1, W0 = (RW + 1) >> 1, W1 = RW >> 1;
    const int o1 = N * H0 * W0, o2 = o1 + N * H0 * W1, o3 = o2 + N * H1 * W0;
    int ph, pw, r;
    if (m < o1) { ph = 0; pw = 0; r = m; }
    else if (m < o2) { ph = 0; pw = 1; r = m - o1; }
    else if (m < o3) { ph = 1; pw = 0; r = m - o2; }
    else { ph = 1; pw = 1; r = m - o3; }
    const int Hc = ph ? H1 : H0, Wc = pw ? W1 : W0, S = Hc * Wc;
    n = r / S;
    const int rem = r - n * S;
    const int i = rem / Wc;
    rh = 2 * i + ph;
    rw = 2 * (rem - i * Wc) + pw;
}

// the 7x7 stride-2 stem as a GEMM of its own (conv_stem.hip), conv math bf16x6
bool stem_geometry(const ConvArgs& a);
int stem_tile_rows();
bool launch_stem_fwd(ConvArgs a, hipStream_t s);
size_t stem_wgrad_workspace(const ConvArgs& a);
bool launch_stem_wgrad(ConvArgs a, float* dw, float* workspace, hipStream_t s);

// bf16x6 path (conv_mfma_x6.hip): fp32-accurate products from six bf16 MFMAs
bool launch_conv_fwd_x6(ConvArgs a, hipStream_t s, int bm, int bn);
bool launch_conv_dgrad_x6(ConvArgs a, hipStream_t s, int bm, int bn);
bool launch_conv_x6_pk(ConvArgs a, hipStream_t s, bool dgrad, int bm, int planes);
size_t packed_weight_bytes_x6(int Cout, int T, int Cin, bool dgrad, int planes);
void launch_pack_weights_batch(const void* jobs_dev, int njobs, int total_blocks, int planes, hipStream_t s);
int pack_job_blocks_x6(int Cout, int T, int Cin, bool dgrad);
void launch_pack_weights_x6(const float* w, void* packed, int Cout, int T, int Cin, bool dgrad, int planes, hipStream_t s);
void launch_transpose_ohwi(const float* w, float* wt, int Cout, int T, int Cin, hipStream_t s);

void launch_join_planes(const unsigned short* in, int ldp, int64_t ps, int64_t M, int C, float* x, int ldx, hipStream_t s);
void launch_split_planes(const float* x, int64_t M, int C, int ldx, unsigned short* out, int ldp, int64_t pstride_elems,
                         int planes, hipStream_t s);


// Weight packing for k_conv_pl2 (16x16x32 fragments): packed[((cb * K32 + k32) * NP + plane) * 64 + lane] (uint4) holds, for
// column cb*16 + (lane & 15), the 8 bf16 of that plane at k = k32*32 + 8*(lane >> 4) .. +7.
//   fwd  : column = cout, k = (tap, cin);   dgrad: column = cin, k = (tap, cout) (implicit transpose).
// Columns >= NC are zero; cb runs to ceil(NC/128)*8.
template <bool DGRAD, int NP>
__device__ __forceinline__ void pack_weights_pl2_body(const float* __restrict__ w, uint4* __restrict__ packed, int Cout, int T,
                                                      int Cin, int K32, int idx) {
    const int lane = idx & 63, f = idx >> 6;
    const int k32 = f % K32, cb = f / K32;
    const int col = cb * 16 + (lane & 15), k0 = k32 * 32 + 8 * (lane >> 4);
    const int NC = DGRAD ? Cin : Cout, GC = DGRAD ? Cout : Cin;
    float v[8];
    const int tap = k0 / GC, g0 = k0 - tap * GC;      // 8 consecutive k never straddle a tap (GC % 64 == 0)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float x = 0.f;
        if (col < NC) x = DGRAD ? w[((size_t)(g0 + i) * T + tap) * Cin + col] : w[((size_t)col * T + tap) * Cin + g0 + i];
        v[i] = x;
    }
    uint4* o = packed + (size_t)f * (64 * NP) + lane;
    if constexpr (NP == 1) {
        const uint2 r0 = round_bf16x4(make_float4(v[0], v[1], v[2], v[3])), r1 = round_bf16x4(make_float4(v[4], v[5], v[6], v[7]));
        o[0] = make_uint4(r0.x, r0.y, r1.x, r1.y);
    } else {
        uint2 h0, m0, l0, h1, m1, l1;
        split3(make_float4(v[0], v[1], v[2], v[3]), h0, m0, l0);
        split3(make_float4(v[4], v[5], v[6], v[7]), h1, m1, l1);
        o[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        o[64] = make_uint4(m0.x, m0.y, m1.x, m1.y);
        o[128] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

// second-generation planes kernel: (16*rbw) x 128 tiles, one workgroup per CU (conv_mfma_pl2.hip)
int pack_job_blocks_pl2(int Cout, int T, int Cin, bool dgrad);
int conv_pl2_pick_rbw(int64_t M, int cols);
void conv_pl2_plan(int64_t M, int cols, int K, bool wide_ok, int* rbw, int* wide);      // tile height and 128- / 256-column form
bool launch_conv_pl2w(ConvArgs a, hipStream_t s, bool dgrad, int planes, int rbw);  // conv_mfma_pl2w.hip
bool launch_conv_pl2(ConvArgs a, hipStream_t s, bool dgrad, int planes, int rbw);
size_t packed_weight_bytes_pl2(int Cout, int T, int Cin, bool dgrad, int planes);
void launch_pack_weights_pl2(const float* w, void* packed, int Cout, int T, int Cin, bool dgrad, int planes, hipStream_t s);

// halo-patch bf16x6 kernel for stride-1 KxK filters (conv_mfma_x6p.hip)
struct PatchArgs {
    const float* x;      // gathered tensor (fwd: input x, dgrad: dy), NHWC with pitch ldg
    const uint4* wpk;    // packed weights (k_pack_weights_x6)
    const float* bias;   // fwd only, may be null
    float* y;            // output tensor (fwd: y, dgrad: dx), pitch ldo
    float* stats;        // fwd only: [2][MT][NC] tile sums / centred M2, then [MT] valid-row counts; may be null
    int N, RH, RW;       // pixel grid of the GEMM rows (fwd: Ho x Wo, dgrad: H x W)
    int GH, GW;          // pixel grid of the gathered tensor
    int GC, NC;          // gathered channels (GEMM K per tap), output columns
    int KH, KW, dil;
    int orgh, orgw;      // gathered coordinate of halo (0,0) = patch origin + org
    int flip;            // 1: tap (kh, kw) reads halo offset ((KH-1-kh)*dil, (KW-1-kw)*dil)  (data gradient)
    int PH, PW, HH, HW;  // patch and halo dims
    int TPY, TPX;        // patches per image
    int ldg, ldo;
    int MT, NT;
    int accumulate;
};
bool conv_patch_plan(int RH, int RW, int KH, int KW, int dil, int* PH, int* PW);
void launch_conv_x6_patch(PatchArgs a, bool dgrad, int planes, hipStream_t s);

// tap-uniform fast path (conv_mfma_u.hip); each returns false when the geometry does not qualify
bool launch_conv_fwd_u(ConvArgs a, hipStream_t s);
void conv_pick_tile(int64_t M, int cols, int* bm, int* bn);
void conv_pick_tile_x6(int64_t M, int cols, int K, bool dgrad, bool pointwise, int* bm, int* bn);
int conv_fwd_tile_rows(int64_t M, int Cin, int Cout);   // rows per forward M tile == rows per BN partial
bool launch_conv_dgrad_u(ConvArgs a, hipStream_t s);

}  // namespace iswm
