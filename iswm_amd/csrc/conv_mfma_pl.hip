// Producer-side helpers of the pre-split ("planes") activation layout.
//
// An activation that feeds a convolution is stored by its PRODUCER (the BatchNorm / pooling / resize pass that writes
// it) as three bf16 tensors [plane][pixel][ld] (hi, mid, lo; hi + mid + lo == the fp32 value, bit-exactly), or as one
// round-to-nearest-even bf16 plane in the mixed-precision mode.  The convolutions that consume the planes are in
// conv_mfma_pl2.hip / conv_mfma_pl2w.hip / conv_mfma_pl2t.hip / conv_wgrad_pl.hip; this file holds the stand-alone
// split and join passes (iswm_split_planes / iswm_join_planes) for tensors that arrive in fp32 (the network input of a
// planes consumer, test operands) and for reading a planes tensor back.
// (The first planes convolution kernel -- the round-1 tiling fed by LDS-DMA, measured no faster than the in-kernel split,
// DESIGN.md section 3.2 -- lived here until round 3; profiles/r02_pl_check.txt keeps its numbers.)
#include <stdlib.h>

#include "conv_common.h"

namespace iswm {

// exact 3-way split of an fp32 [M][C] (pitch ldx) matrix into bf16 planes [NP][M][ldp]  (NP == 1: round to nearest)
template <int NP>
__global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ x, int64_t M, int C4, int ldx,
                                                      unsigned short* __restrict__ out, int ldp, int64_t pstride) {
    const int64_t total = M * C4;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / C4;
        const int c4 = (int)(i - m * C4);
        const float4 v = ldg4(x + m * ldx + c4 * 4);
        unsigned short* o = out + m * ldp + c4 * 4;
        if constexpr (NP == 1) {
            *reinterpret_cast<uint2*>(o) = round_bf16x4(v);
        } else {
            uint2 h, md, l;
            split3(v, h, md, l);
            *reinterpret_cast<uint2*>(o) = h;
            *reinterpret_cast<uint2*>(o + pstride) = md;
            *reinterpret_cast<uint2*>(o + 2 * pstride) = l;
        }
    }
}

// planes -> fp32 (x = hi + mid + lo, exact); one plane: the bf16 value
__global__ __launch_bounds__(256) void k_join_planes(const unsigned short* __restrict__ in, int ldp, int64_t ps, int64_t M, int C4,
                                                     float* __restrict__ x, int ldx) {
    const int64_t total = M * C4;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / C4;
        const int c4 = (int)(i - m * C4);
        *reinterpret_cast<float4*>(x + m * ldx + c4 * 4) = ld4x(in, m * ldp + c4 * 4, ps);
    }
}

void launch_join_planes(const unsigned short* in, int ldp, int64_t ps, int64_t M, int C, float* x, int ldx, hipStream_t s) {
    hipLaunchKernelGGL(k_join_planes, dim3(stream_grid(M * (C / 4), 256)), dim3(256), 0, s, in, ldp, ps, M, C / 4, x, ldx);
}

void launch_split_planes(const float* x, int64_t M, int C, int ldx, unsigned short* out, int ldp, int64_t pstride_elems,
                         int planes, hipStream_t s) {
    const int grid = stream_grid(M * (C / 4), 256);
    if (planes == 1) hipLaunchKernelGGL(k_split_planes<1>, dim3(grid), dim3(256), 0, s, x, M, C / 4, ldx, out, ldp, pstride_elems);
    else hipLaunchKernelGGL(k_split_planes<3>, dim3(grid), dim3(256), 0, s, x, M, C / 4, ldx, out, ldp, pstride_elems);
}

}  // namespace iswm
