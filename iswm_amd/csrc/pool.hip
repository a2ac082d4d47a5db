// Pooling kernels over NHWC tensors (all HBM-bound):
//   * MaxPool2d(3, stride 2, pad 1) of the ResNet stem, network/backbone/resnet.py:148,204,
//     with the ATen tie rule (first maximum in window scan order wins; NaN propagates);
//   * AdaptiveAvgPool2d(1) of ASPPPooling, network/_deeplab.py:133, and the 1x1 -> HxW
//     bilinear upsample that follows it (a broadcast), network/_deeplab.py:141.
#include "rowmap.h"

namespace iswm {

__global__ __launch_bounds__(256) void k_maxpool_fwd(const float* __restrict__ x, int N, int H, int W, int C4,
                                                     void* __restrict__ y, int64_t yps, uint8_t* __restrict__ idx, int Ho,
                                                     int Wo, int CQ, int RL) {
    RowThread rt = row_thread(C4, CQ, RL);
    if (!rt.active) return;
    const int C = C4 * 4, c = rt.c4 * 4;
    const int64_t Mo = (int64_t)N * Ho * Wo;
    for (int64_t r = rt.row0; r < Mo; r += rt.rstep) {
        int n = (int)(r / (Ho * Wo));
        int rem = (int)(r - (int64_t)n * Ho * Wo);
        int oh = rem / Wo, ow = rem - oh * Wo;
        const int hs = oh * 2 - 1, ws = ow * 2 - 1;
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int bi[4] = {-1, -1, -1, -1};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            int ih = hs + kh;
            if (ih < 0 || ih >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                int iw = ws + kw;
                if (iw < 0 || iw >= W) continue;
                float4 v = ld4(x + ((size_t)(n * H + ih) * W + iw) * C + c);
                const int tap = kh * 3 + kw;
                // ATen: take v if (v > max) or isnan(v); the first in-bounds tap always wins over -inf
                if (bi[0] < 0 || v.x > best.x || v.x != v.x) { best.x = v.x; bi[0] = tap; }
                if (bi[1] < 0 || v.y > best.y || v.y != v.y) { best.y = v.y; bi[1] = tap; }
                if (bi[2] < 0 || v.z > best.z || v.z != v.z) { best.z = v.z; bi[2] = tap; }
                if (bi[3] < 0 || v.w > best.w || v.w != v.w) { best.w = v.w; bi[3] = tap; }
            }
        }
        st4x(y, (int64_t)r * C + c, yps, best);
        uchar4 u = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2],
                               (unsigned char)bi[3]);
        *reinterpret_cast<uchar4*>(idx + (size_t)r * C + c) = u;
    }
}

// gather form: every input pixel sums the (at most 4) windows that cover it and selected it
__global__ __launch_bounds__(256) void k_maxpool_bwd(const float* __restrict__ dy,
                                                     const uint8_t* __restrict__ idx, int N, int H, int W,
                                                     int C4, int Ho, int Wo, float* __restrict__ dx, int CQ,
                                                     int RL) {
    RowThread rt = row_thread(C4, CQ, RL);
    if (!rt.active) return;
    const int C = C4 * 4, c = rt.c4 * 4;
    const int64_t Mi = (int64_t)N * H * W;
    for (int64_t r = rt.row0; r < Mi; r += rt.rstep) {
        int n = (int)(r / (H * W));
        int rem = (int)(r - (int64_t)n * H * W);
        int ih = rem / W, iw = rem - ih * W;
        float4 g = make_float4(0, 0, 0, 0);
        const int oh_lo = ih >> 1, oh_hi = (ih + 1) >> 1;   // ceil((ih-1)/2) .. floor((ih+1)/2)
        const int ow_lo = iw >> 1, ow_hi = (iw + 1) >> 1;
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            if (oh >= Ho) continue;
            const int kh = ih - (oh * 2 - 1);
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                if (ow >= Wo) continue;
                const int kw = iw - (ow * 2 - 1);
                const unsigned char tap = (unsigned char)(kh * 3 + kw);
                size_t o = ((size_t)(n * Ho + oh) * Wo + ow) * C + c;
                uchar4 u = *reinterpret_cast<const uchar4*>(idx + o);
                float4 d = ld4(dy + o);
                if (u.x == tap) g.x += d.x;
                if (u.y == tap) g.y += d.y;
                if (u.z == tap) g.z += d.z;
                if (u.w == tap) g.w += d.w;
            }
        }
        st4(dx + (size_t)r * C + c, g);
    }
}

// y[n, c] = mean_p x[n, p, c];  grid (1, colblocks, N)
__global__ __launch_bounds__(256) void k_gap_fwd(const void* __restrict__ x, int64_t xps, int HW, int C4, int ldx,
                                                 float* __restrict__ y, int CQ, int RL) {
    __shared__ float red[256 * 4];
    RowThread rt = row_thread(C4, CQ, RL);
    const int n = blockIdx.z;
    float4 s = make_float4(0, 0, 0, 0);
    if (rt.active) {
        const int64_t p = (int64_t)n * HW * ldx + rt.c4 * 4;
        // four rows in flight per thread (a thread walks HW / RL ~ 136 rows of the 33 x 33 map: one dependent load chain per row
        // left the pass at 1.5 TB/s), summed in a fixed order
        float4 s1 = s, s2 = s, s3 = s;
        int r = rt.rl;
        for (; r + 3 * RL < HW; r += 4 * RL) {
            const float4 v0 = ld4x(x, p + (int64_t)r * ldx, xps), v1 = ld4x(x, p + (int64_t)(r + RL) * ldx, xps);
            const float4 v2 = ld4x(x, p + (int64_t)(r + 2 * RL) * ldx, xps), v3 = ld4x(x, p + (int64_t)(r + 3 * RL) * ldx, xps);
            s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
            s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
            s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
            s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
        }
        for (; r < HW; r += RL) {
            const float4 v = ld4x(x, p + (int64_t)r * ldx, xps);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        s.x = (s.x + s1.x) + (s2.x + s3.x); s.y = (s.y + s1.y) + (s2.y + s3.y);
        s.z = (s.z + s1.z) + (s2.z + s3.z); s.w = (s.w + s1.w) + (s2.w + s3.w);
    }
    const int t = threadIdx.x;
    st4(&red[t * 4], s);
    __syncthreads();
    if (rt.active && rt.rl == 0) {
        for (int k = 1; k < RL; ++k) {
            float4 a = ld4(&red[(t + k * CQ) * 4]);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        const float inv = 1.f / (float)HW;
        st4(y + (size_t)n * C4 * 4 + rt.c4 * 4, make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv));
    }
}

// dst[n, p, c] (=|+=) v[n, c] * mul;  grid (rowblocks, colblocks, N)
template <bool ACC>
__global__ __launch_bounds__(256) void k_bcast(const float* __restrict__ v, int HW, int C4, float mul,
                                               void* __restrict__ dst, int ldd, int64_t dps, int CQ, int RL) {
    RowThread rt = row_thread(C4, CQ, RL);
    if (!rt.active) return;
    const int n = blockIdx.z, c = rt.c4 * 4;
    float4 s = ld4(v + (size_t)n * C4 * 4 + c);
    s = make_float4(s.x * mul, s.y * mul, s.z * mul, s.w * mul);
    const int64_t p = (int64_t)n * HW * ldd + c;
    for (int64_t r = rt.row0; r < HW; r += rt.rstep) {
        if (ACC) {
            float* q = reinterpret_cast<float*>(dst) + p + r * ldd;      // accumulation targets are fp32 gradients
            float4 o = ld4(q);
            st4(q, make_float4(o.x + s.x, o.y + s.y, o.z + s.z, o.w + s.w));
        } else {
            st4x(dst, p + r * ldd, dps, s);
        }
    }
}

// dv[n, c] = sum_p dy[n, p, c]  (same shape of work as k_gap_fwd without the 1/HW)
__global__ __launch_bounds__(256) void k_bcast_bwd(const float* __restrict__ dy, int HW, int C4, int ldd,
                                                   float* __restrict__ dv, int CQ, int RL) {
    __shared__ float red[256 * 4];
    RowThread rt = row_thread(C4, CQ, RL);
    const int n = blockIdx.z;
    float4 s = make_float4(0, 0, 0, 0);
    if (rt.active) {
        const float* p = dy + (size_t)n * HW * ldd + rt.c4 * 4;
        float4 s1 = s, s2 = s, s3 = s;                   // four rows in flight, as k_gap_fwd
        int r = rt.rl;
        for (; r + 3 * RL < HW; r += 4 * RL) {
            const float4 v0 = ld4(p + (size_t)r * ldd), v1 = ld4(p + (size_t)(r + RL) * ldd);
            const float4 v2 = ld4(p + (size_t)(r + 2 * RL) * ldd), v3 = ld4(p + (size_t)(r + 3 * RL) * ldd);
            s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
            s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
            s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
            s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
        }
        for (; r < HW; r += RL) {
            const float4 v = ld4(p + (size_t)r * ldd);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        s.x = (s.x + s1.x) + (s2.x + s3.x); s.y = (s.y + s1.y) + (s2.y + s3.y);
        s.z = (s.z + s1.z) + (s2.z + s3.z); s.w = (s.w + s1.w) + (s2.w + s3.w);
    }
    const int t = threadIdx.x;
    st4(&red[t * 4], s);
    __syncthreads();
    if (rt.active && rt.rl == 0) {
        for (int k = 1; k < RL; ++k) {
            float4 a = ld4(&red[(t + k * CQ) * 4]);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        st4(dv + (size_t)n * C4 * 4 + rt.c4 * 4, s);
    }
}

// small-CQ plan for the per-image reductions: more row lanes per block
static RowPlan plan_reduce(int C) {
    RowPlan p;
    p.C4 = C / 4;
    p.CQ = p.C4 < 32 ? p.C4 : 32;
    p.RL = 256 / p.CQ;
    p.colblocks = (p.C4 + p.CQ - 1) / p.CQ;
    p.rowblocks = 1;
    return p;
}

}  // namespace iswm

using namespace iswm;

extern "C" int iswm_maxpool3x3s2_fwd(const float* x, int N, int H, int W, int C, float* y, uint8_t* idx, int Ho,
                                     int Wo, iswm_stream_t stream) {
    return iswm_maxpool3x3s2_fwd_pl(x, N, H, W, C, y, 0, idx, Ho, Wo, stream);
}

extern "C" int iswm_maxpool3x3s2_fwd_pl(const float* x, int N, int H, int W, int C, void* y, int64_t y_ps, uint8_t* idx,
                                        int Ho, int Wo, iswm_stream_t stream) {
    ISWM_REQUIRE(x && y && idx && C % 4 == 0 && N > 0, "maxpool_fwd: bad argument");
    ISWM_REQUIRE(y_ps == 0 || y_ps == -1 || y_ps >= (int64_t)N * Ho * Wo * C, "maxpool_fwd: bad plane stride");
    const int64_t yps = y_ps;
    ISWM_REQUIRE(Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1, "maxpool_fwd: bad output size");
    RowPlan p = plan_rows((int64_t)N * Ho * Wo, C);
    hipLaunchKernelGGL(k_maxpool_fwd, dim3(p.rowblocks, p.colblocks), dim3(256), 0, (hipStream_t)stream, x, N, H,
                       W, p.C4, y, yps, idx, Ho, Wo, p.CQ, p.RL);
    return check_launch("maxpool_fwd");
}

extern "C" int iswm_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, int N, int H, int W, int C, int Ho,
                                     int Wo, float* dx, iswm_stream_t stream) {
    ISWM_REQUIRE(dy && idx && dx && C % 4 == 0 && N > 0, "maxpool_bwd: bad argument");
    RowPlan p = plan_rows((int64_t)N * H * W, C);
    hipLaunchKernelGGL(k_maxpool_bwd, dim3(p.rowblocks, p.colblocks), dim3(256), 0, (hipStream_t)stream, dy, idx,
                       N, H, W, p.C4, Ho, Wo, dx, p.CQ, p.RL);
    return check_launch("maxpool_bwd");
}

extern "C" int iswm_gap_fwd(const float* x, int N, int HW, int C, int ldx, float* y, iswm_stream_t stream) {
    return iswm_gap_fwd_pl(x, 0, N, HW, C, ldx, y, stream);
}

extern "C" int iswm_gap_fwd_pl(const void* x, int64_t x_ps, int N, int HW, int C, int ldx, float* y, iswm_stream_t stream) {
    ISWM_REQUIRE(x && y && C % 4 == 0 && ldx % 4 == 0 && ldx >= C && N > 0 && HW > 0, "gap_fwd: bad argument");
    RowPlan p = plan_reduce(C);
    hipLaunchKernelGGL(k_gap_fwd, dim3(1, p.colblocks, N), dim3(256), 0, (hipStream_t)stream, x, x_ps, HW, p.C4, ldx, y,
                       p.CQ, p.RL);
    return check_launch("gap_fwd");
}

extern "C" int iswm_gap_bwd(const float* dy, int N, int HW, int C, float* dx, int lddx, int accumulate,
                            iswm_stream_t stream) {
    ISWM_REQUIRE(dy && dx && C % 4 == 0 && lddx % 4 == 0 && lddx >= C && N > 0 && HW > 0, "gap_bwd: bad argument");
    RowPlan p = plan_rows(HW, C);
    dim3 grid(p.rowblocks, p.colblocks, N);
    const float mul = 1.f / (float)HW;
    if (accumulate)
        hipLaunchKernelGGL((k_bcast<true>), grid, dim3(256), 0, (hipStream_t)stream, dy, HW, p.C4, mul, dx, lddx,
                           (int64_t)0, p.CQ, p.RL);
    else
        hipLaunchKernelGGL((k_bcast<false>), grid, dim3(256), 0, (hipStream_t)stream, dy, HW, p.C4, mul, dx, lddx,
                           (int64_t)0, p.CQ, p.RL);
    return check_launch("gap_bwd");
}

extern "C" int iswm_bcast_fwd(const float* v, int N, int HW, int C, float* y, int ldy, iswm_stream_t stream) {
    return iswm_bcast_fwd_pl(v, N, HW, C, y, ldy, 0, stream);
}

extern "C" int iswm_bcast_fwd_pl(const float* v, int N, int HW, int C, void* y, int ldy, int64_t y_ps, iswm_stream_t stream) {
    ISWM_REQUIRE(v && y && C % 4 == 0 && ldy % 4 == 0 && ldy >= C && N > 0 && HW > 0, "bcast_fwd: bad argument");
    RowPlan p = plan_rows(HW, C);
    hipLaunchKernelGGL((k_bcast<false>), dim3(p.rowblocks, p.colblocks, N), dim3(256), 0, (hipStream_t)stream, v,
                       HW, p.C4, 1.f, y, ldy, y_ps, p.CQ, p.RL);
    return check_launch("bcast_fwd");
}

extern "C" int iswm_bcast_bwd(const float* dy, int lddy, int N, int HW, int C, float* dv, iswm_stream_t stream) {
    ISWM_REQUIRE(dy && dv && C % 4 == 0 && lddy % 4 == 0 && lddy >= C && N > 0 && HW > 0, "bcast_bwd: bad argument");
    RowPlan p = plan_reduce(C);
    hipLaunchKernelGGL(k_bcast_bwd, dim3(1, p.colblocks, N), dim3(256), 0, (hipStream_t)stream, dy, HW, p.C4, lddy,
                       dv, p.CQ, p.RL);
    return check_launch("bcast_bwd");
}
