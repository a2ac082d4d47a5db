// Declarations shared by the implicit-GEMM convolution kernels (conv_mfma.hip: general-K kernels and
// the C entry points; conv_mfma_u.hip: the tap-uniform fast path).
#pragma once
#include "common.h"

namespace iswm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

constexpr int KC_PITCH = 36;  // floats per LDS row of a K-contiguous operand tile (32 + 4 pad)


// tap-uniform fast path (conv_mfma_u.hip); each returns false when the geometry does not qualify
bool launch_conv_fwd_u(ConvArgs a, hipStream_t s);
void conv_pick_tile(int64_t M, int cols, int* bm, int* bn);
int conv_fwd_tile_rows(int64_t M, int Cin, int Cout);   // rows per forward M tile == rows per BN partial
bool launch_conv_dgrad_u(ConvArgs a, hipStream_t s);

}  // namespace iswm
