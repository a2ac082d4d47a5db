// Bilinear resize with align_corners=False (F.interpolate at network/_deeplab.py:58 and
// network/utils.py:22), forward and backward, plus the NCHW <-> NHWC layout changes at the
// model boundary.  Index/weight arithmetic restates ATen's area_pixel_compute_source_index:
//   scale = in/out (float); src = scale*(dst+0.5)-0.5, clamped at 0; i0 = (int)src;
//   i1 = i0 + (i0 < in-1); l1 = src - i0; l0 = 1 - l1.
// Backward is a GATHER (each input pixel visits the output pixels that can reference it and
// re-derives their weights with the same float formula), so it needs no atomics and is
// bit-reproducible.  All kernels are HBM/L2-bound.
#include "rowmap.h"

namespace iswm {

struct Lerp {
    int i0, i1;
    float l0, l1;
};

__device__ __forceinline__ Lerp src_index(float scale, int dst, int in_size) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    Lerp r;
    r.i0 = (int)src;
    if (r.i0 > in_size - 1) r.i0 = in_size - 1;
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

// candidate output range that can touch input index i (with one index of slack each side)
__device__ __forceinline__ void out_range(float inv_scale, int i, int out_size, int& lo, int& hi) {
    float a = ((float)i - 0.5f) * inv_scale - 0.5f;
    float b = ((float)i + 1.5f) * inv_scale - 0.5f;
    lo = (int)floorf(a) - 1;
    hi = (int)ceilf(b) + 1;
    if (lo < 0) lo = 0;
    if (hi > out_size - 1) hi = out_size - 1;
}

__device__ __forceinline__ float weight_for(const Lerp& l, int i) {
    return (l.i0 == i ? l.l0 : 0.f) + (l.i1 == i ? l.l1 : 0.f);
}

__global__ __launch_bounds__(256) void k_bilinear_fwd(const float* __restrict__ x, int N, int Hi, int Wi, int C4,
                                                      int ldx, void* __restrict__ y, int64_t yps, int Ho, int Wo, int ldy,
                                                      float sh, float sw, int CQ, int RL) {
    RowThread rt = row_thread(C4, CQ, RL);
    if (!rt.active) return;
    const int c = rt.c4 * 4;
    const int64_t Mo = (int64_t)N * Ho * Wo;
    for (int64_t r = rt.row0; r < Mo; r += rt.rstep) {
        int n = (int)(r / (Ho * Wo));
        int rem = (int)(r - (int64_t)n * Ho * Wo);
        int oh = rem / Wo, ow = rem - oh * Wo;
        Lerp lh = src_index(sh, oh, Hi), lw = src_index(sw, ow, Wi);
        const float* base = x + (size_t)n * Hi * Wi * ldx + c;
        float4 a = ld4(base + ((size_t)lh.i0 * Wi + lw.i0) * ldx);
        float4 b = ld4(base + ((size_t)lh.i0 * Wi + lw.i1) * ldx);
        float4 d = ld4(base + ((size_t)lh.i1 * Wi + lw.i0) * ldx);
        float4 e = ld4(base + ((size_t)lh.i1 * Wi + lw.i1) * ldx);
        float4 o;
        o.x = lh.l0 * (lw.l0 * a.x + lw.l1 * b.x) + lh.l1 * (lw.l0 * d.x + lw.l1 * e.x);
        o.y = lh.l0 * (lw.l0 * a.y + lw.l1 * b.y) + lh.l1 * (lw.l0 * d.y + lw.l1 * e.y);
        o.z = lh.l0 * (lw.l0 * a.z + lw.l1 * b.z) + lh.l1 * (lw.l0 * d.z + lw.l1 * e.z);
        o.w = lh.l0 * (lw.l0 * a.w + lw.l1 * b.w) + lh.l1 * (lw.l0 * d.w + lw.l1 * e.w);
        st4x(y, (int64_t)r * ldy + c, yps, o);
    }
}

__global__ __launch_bounds__(256) void k_bilinear_bwd(const float* __restrict__ dy, int N, int Hi, int Wi, int C4,
                                                      int lddy, int Ho, int Wo, float* __restrict__ dx, int lddx,
                                                      float sh, float sw, int CQ, int RL) {
    RowThread rt = row_thread(C4, CQ, RL);
    if (!rt.active) return;
    const int c = rt.c4 * 4;
    const int64_t Mi = (int64_t)N * Hi * Wi;
    const float ish = 1.f / sh, isw = 1.f / sw;
    for (int64_t r = rt.row0; r < Mi; r += rt.rstep) {
        int n = (int)(r / (Hi * Wi));
        int rem = (int)(r - (int64_t)n * Hi * Wi);
        int ih = rem / Wi, iw = rem - ih * Wi;
        int hlo, hhi, wlo, whi;
        out_range(ish, ih, Ho, hlo, hhi);
        out_range(isw, iw, Wo, wlo, whi);
        float4 g = make_float4(0, 0, 0, 0);
        const float* base = dy + (size_t)n * Ho * Wo * lddy + c;
        for (int oh = hlo; oh <= hhi; ++oh) {
            float wh = weight_for(src_index(sh, oh, Hi), ih);
            if (wh == 0.f) continue;
            for (int ow = wlo; ow <= whi; ++ow) {
                float ww = weight_for(src_index(sw, ow, Wi), iw);
                if (ww == 0.f) continue;
                float4 d = ld4(base + ((size_t)oh * Wo + ow) * lddy);
                float w = wh * ww;
                g.x += w * d.x; g.y += w * d.y; g.z += w * d.z; g.w += w * d.w;
            }
        }
        st4(dx + (size_t)r * lddx + c, g);
    }
}

// logits: NHWC (pitch ldx, first C channels) -> NCHW, one thread per output pixel
__global__ __launch_bounds__(256) void k_bilinear_to_nchw_fwd(const float* __restrict__ x, int N, int Hi, int Wi,
                                                              int C, int ldx, float* __restrict__ y, int Ho,
                                                              int Wo, float sh, float sw) {
    const int64_t Mo = (int64_t)N * Ho * Wo;
    const int64_t HWo = (int64_t)Ho * Wo;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < Mo; r += (int64_t)gridDim.x * blockDim.x) {
        int n = (int)(r / HWo);
        int rem = (int)(r - n * HWo);
        int oh = rem / Wo, ow = rem - oh * Wo;
        Lerp lh = src_index(sh, oh, Hi), lw = src_index(sw, ow, Wi);
        const float* base = x + (size_t)n * Hi * Wi * ldx;
        const float* pa = base + ((size_t)lh.i0 * Wi + lw.i0) * ldx;
        const float* pb = base + ((size_t)lh.i0 * Wi + lw.i1) * ldx;
        const float* pd = base + ((size_t)lh.i1 * Wi + lw.i0) * ldx;
        const float* pe = base + ((size_t)lh.i1 * Wi + lw.i1) * ldx;
        for (int c0 = 0; c0 < C; c0 += 4) {
            float4 a = ld4(pa + c0), b = ld4(pb + c0), d = ld4(pd + c0), e = ld4(pe + c0);
            float o[4];
            o[0] = lh.l0 * (lw.l0 * a.x + lw.l1 * b.x) + lh.l1 * (lw.l0 * d.x + lw.l1 * e.x);
            o[1] = lh.l0 * (lw.l0 * a.y + lw.l1 * b.y) + lh.l1 * (lw.l0 * d.y + lw.l1 * e.y);
            o[2] = lh.l0 * (lw.l0 * a.z + lw.l1 * b.z) + lh.l1 * (lw.l0 * d.z + lw.l1 * e.z);
            o[3] = lh.l0 * (lw.l0 * a.w + lw.l1 * b.w) + lh.l1 * (lw.l0 * d.w + lw.l1 * e.w);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c0 + k < C) y[((size_t)n * C + c0 + k) * HWo + rem] = o[k];
        }
    }
}

// dlogits NCHW -> d(low-res) NHWC with pitch lddx (channels >= C are written as zero)
__global__ __launch_bounds__(256) void k_bilinear_to_nchw_bwd(const float* __restrict__ dy, int N, int Hi, int Wi,
                                                              int C, int lddx, int Ho, int Wo,
                                                              float* __restrict__ dx, float sh, float sw) {
    const int64_t Mi = (int64_t)N * Hi * Wi;
    const int64_t HWo = (int64_t)Ho * Wo;
    const float ish = 1.f / sh, isw = 1.f / sw;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < Mi; r += (int64_t)gridDim.x * blockDim.x) {
        int n = (int)(r / (Hi * Wi));
        int rem = (int)(r - (int64_t)n * Hi * Wi);
        int ih = rem / Wi, iw = rem - ih * Wi;
        int hlo, hhi, wlo, whi;
        out_range(ish, ih, Ho, hlo, hhi);
        out_range(isw, iw, Wo, wlo, whi);
        for (int c0 = 0; c0 < lddx; c0 += 4) {
            float g[4] = {0.f, 0.f, 0.f, 0.f};
            if (c0 < C) {
                for (int oh = hlo; oh <= hhi; ++oh) {
                    float wh = weight_for(src_index(sh, oh, Hi), ih);
                    if (wh == 0.f) continue;
                    for (int ow = wlo; ow <= whi; ++ow) {
                        float ww = weight_for(src_index(sw, ow, Wi), iw);
                        if (ww == 0.f) continue;
                        float w = wh * ww;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (c0 + k < C) g[k] += w * dy[((size_t)n * C + c0 + k) * HWo + (size_t)oh * Wo + ow];
                    }
                }
            }
            st4(dx + (size_t)r * lddx + c0, make_float4(g[0], g[1], g[2], g[3]));
        }
    }
}

__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float* __restrict__ x, int N, int C, int64_t HW,
                                                      float* __restrict__ y, int Cp) {
    const int64_t M = (int64_t)N * HW;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < M; r += (int64_t)gridDim.x * blockDim.x) {
        int64_t n = r / HW, p = r - n * HW;
        for (int c0 = 0; c0 < Cp; c0 += 4) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (c0 + k < C) ? x[(n * C + c0 + k) * HW + p] : 0.f;
            st4(y + r * Cp + c0, make_float4(v[0], v[1], v[2], v[3]));
        }
    }
}

__global__ __launch_bounds__(256) void k_nhwc_to_nchw(const float* __restrict__ x, int N, int C, int64_t HW,
                                                      int ldx, float* __restrict__ y) {
    const int64_t M = (int64_t)N * HW;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < M; r += (int64_t)gridDim.x * blockDim.x) {
        int64_t n = r / HW, p = r - n * HW;
        for (int c = 0; c < C; ++c) y[(n * C + c) * HW + p] = x[r * ldx + c];
    }
}

}  // namespace iswm

using namespace iswm;

extern "C" int iswm_bilinear_fwd(const float* x, int N, int Hi, int Wi, int C, int ldx, float* y, int Ho, int Wo,
                                 int ldy, iswm_stream_t stream) {
    return iswm_bilinear_fwd_pl(x, N, Hi, Wi, C, ldx, y, 0, Ho, Wo, ldy, stream);
}

extern "C" int iswm_bilinear_fwd_pl(const float* x, int N, int Hi, int Wi, int C, int ldx, void* y, int64_t y_ps, int Ho,
                                    int Wo, int ldy, iswm_stream_t stream) {
    ISWM_REQUIRE(y_ps == 0 || y_ps == -1 || y_ps >= (int64_t)N * Ho * Wo * ldy, "bilinear_fwd: bad plane stride");
    ISWM_REQUIRE(x && y && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C % 4 == 0 && ldx % 4 == 0 &&
                     ldy % 4 == 0 && ldx >= C && ldy >= C,
                 "bilinear_fwd: bad argument");
    RowPlan p = plan_rows((int64_t)N * Ho * Wo, C);
    hipLaunchKernelGGL(k_bilinear_fwd, dim3(p.rowblocks, p.colblocks), dim3(256), 0, (hipStream_t)stream, x, N, Hi,
                       Wi, p.C4, ldx, y, y_ps, Ho, Wo, ldy, (float)Hi / (float)Ho, (float)Wi / (float)Wo, p.CQ, p.RL);
    return check_launch("bilinear_fwd");
}

extern "C" int iswm_bilinear_bwd(const float* dy, int N, int Hi, int Wi, int C, int lddy, int Ho, int Wo, float* dx,
                                 int lddx, iswm_stream_t stream) {
    ISWM_REQUIRE(dy && dx && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C % 4 == 0 && lddx % 4 == 0 &&
                     lddy % 4 == 0 && lddx >= C && lddy >= C,
                 "bilinear_bwd: bad argument");
    RowPlan p = plan_rows((int64_t)N * Hi * Wi, C);
    hipLaunchKernelGGL(k_bilinear_bwd, dim3(p.rowblocks, p.colblocks), dim3(256), 0, (hipStream_t)stream, dy, N, Hi,
                       Wi, p.C4, lddy, Ho, Wo, dx, lddx, (float)Hi / (float)Ho, (float)Wi / (float)Wo, p.CQ, p.RL);
    return check_launch("bilinear_bwd");
}

extern "C" int iswm_bilinear_nhwc_to_nchw_fwd(const float* x, int N, int Hi, int Wi, int C, int ldx, float* y,
                                              int Ho, int Wo, iswm_stream_t stream) {
    ISWM_REQUIRE(x && y && N > 0 && C > 0 && ldx % 4 == 0 && ldx >= ((C + 3) / 4) * 4, "bilinear_to_nchw_fwd: bad argument");
    int64_t Mo = (int64_t)N * Ho * Wo;
    hipLaunchKernelGGL(k_bilinear_to_nchw_fwd, dim3(stream_grid(Mo, 256)), dim3(256), 0, (hipStream_t)stream, x, N,
                       Hi, Wi, C, ldx, y, Ho, Wo, (float)Hi / (float)Ho, (float)Wi / (float)Wo);
    return check_launch("bilinear_to_nchw_fwd");
}

extern "C" int iswm_bilinear_nhwc_to_nchw_bwd(const float* dy, int N, int Hi, int Wi, int C, int lddx, int Ho,
                                              int Wo, float* dx, iswm_stream_t stream) {
    ISWM_REQUIRE(dy && dx && N > 0 && C > 0 && lddx % 4 == 0 && lddx >= ((C + 3) / 4) * 4, "bilinear_to_nchw_bwd: bad argument");
    int64_t Mi = (int64_t)N * Hi * Wi;
    hipLaunchKernelGGL(k_bilinear_to_nchw_bwd, dim3(stream_grid(Mi, 256)), dim3(256), 0, (hipStream_t)stream, dy, N,
                       Hi, Wi, C, lddx, Ho, Wo, dx, (float)Hi / (float)Ho, (float)Wi / (float)Wo);
    return check_launch("bilinear_to_nchw_bwd");
}

extern "C" int iswm_nchw_to_nhwc(const float* x, int N, int C, int HW, float* y, int Cp, iswm_stream_t stream) {
    ISWM_REQUIRE(x && y && N > 0 && C > 0 && Cp % 4 == 0 && Cp >= C, "nchw_to_nhwc: bad argument");
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3(stream_grid((int64_t)N * HW, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       N, C, (int64_t)HW, y, Cp);
    return check_launch("nchw_to_nhwc");
}

extern "C" int iswm_nhwc_to_nchw(const float* x, int N, int C, int HW, int ldx, float* y, iswm_stream_t stream) {
    ISWM_REQUIRE(x && y && N > 0 && C > 0 && ldx >= C, "nhwc_to_nchw: bad argument");
    hipLaunchKernelGGL(k_nhwc_to_nchw, dim3(stream_grid((int64_t)N * HW, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       N, C, (int64_t)HW, ldx, y);
    return check_launch("nhwc_to_nchw");
}
