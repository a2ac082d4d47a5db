// Depthwise (groups == channels) convolution -- the first half of the reference's AtrousSeparableConvolution
// (network/_deeplab.py:95-119): forward, data gradient and weight gradient on pitched NHWC fp32.
// HBM-bound streaming kernels: one thread owns 4 consecutive channels of one pixel and walks the KH*KW taps
// (the 4-channel weight columns come from L1/L2: the whole filter is C*KH*KW floats).  The weight tensor is the
// torch parameter as is: w[c][kh][kw]  (shape [C,1,KH,KW]).  Channels >= Cw of a wider (zero-padded) buffer get
// zero weights.  The weight gradient reduces over pixels in two fixed-order stages (no atomics).
#include "common.h"

namespace iswm {

struct DwArgs {
    const float* x;      // input  [N,H,W,ldx]
    const float* w;      // [Cw][KH*KW]
    const float* bias;   // [Cw] or null
    float* y;            // output [N,Ho,Wo,ldy]
    int N, H, W, C, Cw, Ho, Wo, KH, KW, stride, pad, dil, ldx, ldy;
    int accumulate;
};

__device__ __forceinline__ float4 dw_w4(const float* w, int c, int T, int tap, int Cw) {
    float4 r;
    r.x = c + 0 < Cw ? w[(size_t)(c + 0) * T + tap] : 0.f;
    r.y = c + 1 < Cw ? w[(size_t)(c + 1) * T + tap] : 0.f;
    r.z = c + 2 < Cw ? w[(size_t)(c + 2) * T + tap] : 0.f;
    r.w = c + 3 < Cw ? w[(size_t)(c + 3) * T + tap] : 0.f;
    return r;
}

__global__ __launch_bounds__(256) void k_dwconv_fwd(const DwArgs a) {
    const int C4 = a.C >> 2, T = a.KH * a.KW;
    const long long total = (long long)a.N * a.Ho * a.Wo * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        long long p = i / C4;
        const int ow = (int)(p % a.Wo);
        p /= a.Wo;
        const int oh = (int)(p % a.Ho), n = (int)(p / a.Ho);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) {
            acc.x = c + 0 < a.Cw ? a.bias[c + 0] : 0.f;
            acc.y = c + 1 < a.Cw ? a.bias[c + 1] : 0.f;
            acc.z = c + 2 < a.Cw ? a.bias[c + 2] : 0.f;
            acc.w = c + 3 < a.Cw ? a.bias[c + 3] : 0.f;
        }
        for (int kh = 0; kh < a.KH; ++kh) {
            const int ih = oh * a.stride - a.pad + kh * a.dil;
            if ((unsigned)ih >= (unsigned)a.H) continue;
            for (int kw = 0; kw < a.KW; ++kw) {
                const int iw = ow * a.stride - a.pad + kw * a.dil;
                if ((unsigned)iw >= (unsigned)a.W) continue;
                const float4 xv = *reinterpret_cast<const float4*>(a.x + ((size_t)(n * a.H + ih) * a.W + iw) * a.ldx + c);
                const float4 wv = dw_w4(a.w, c, T, kh * a.KW + kw, a.Cw);
                acc.x = fmaf(xv.x, wv.x, acc.x);
                acc.y = fmaf(xv.y, wv.y, acc.y);
                acc.z = fmaf(xv.z, wv.z, acc.z);
                acc.w = fmaf(xv.w, wv.w, acc.w);
            }
        }
        *reinterpret_cast<float4*>(a.y + ((size_t)(n * a.Ho + oh) * a.Wo + ow) * a.ldy + c) = acc;
    }
}

// dx[n,ih,iw,c] (=|+=) sum over taps with (ih + pad - kh*dil) divisible by stride: dy[n,oh,ow,c] * w[c,kh,kw]
// here a.x = dy [N,Ho,Wo,ldx], a.y = dx [N,H,W,ldy]
__global__ __launch_bounds__(256) void k_dwconv_dgrad(const DwArgs a) {
    const int C4 = a.C >> 2, T = a.KH * a.KW;
    const long long total = (long long)a.N * a.H * a.W * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        long long p = i / C4;
        const int iw = (int)(p % a.W);
        p /= a.W;
        const int ih = (int)(p % a.H), n = (int)(p / a.H);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int kh = 0; kh < a.KH; ++kh) {
            const int th = ih + a.pad - kh * a.dil;
            if (th < 0 || th % a.stride) continue;
            const int oh = th / a.stride;
            if (oh >= a.Ho) continue;
            for (int kw = 0; kw < a.KW; ++kw) {
                const int tw = iw + a.pad - kw * a.dil;
                if (tw < 0 || tw % a.stride) continue;
                const int ow = tw / a.stride;
                if (ow >= a.Wo) continue;
                const float4 g = *reinterpret_cast<const float4*>(a.x + ((size_t)(n * a.Ho + oh) * a.Wo + ow) * a.ldx + c);
                const float4 wv = dw_w4(a.w, c, T, kh * a.KW + kw, a.Cw);
                acc.x = fmaf(g.x, wv.x, acc.x);
                acc.y = fmaf(g.y, wv.y, acc.y);
                acc.z = fmaf(g.z, wv.z, acc.z);
                acc.w = fmaf(g.w, wv.w, acc.w);
            }
        }
        float4* o = reinterpret_cast<float4*>(a.y + ((size_t)(n * a.H + ih) * a.W + iw) * a.ldy + c);
        if (a.accumulate) {
            const float4 old = *o;
            acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
        }
        *o = acc;
    }
}

// stage 1: partial[chunk][tap][c] = sum over the chunk's output pixels of dy[p][c] * x[pin(p, tap)][c]
// block = 16 channel groups (64 channels) x 16 pixel lanes; grid = (channel blocks, chunks, taps)
__global__ __launch_bounds__(256) void k_dwconv_wgrad_partial(const float* __restrict__ x, const float* __restrict__ dy,
                                                              DwArgs a, int chunk_pixels, double* __restrict__ partial) {
    __shared__ float4 red[16][16];
    const int C4 = a.C >> 2, T = a.KH * a.KW;
    const int cg = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c4 = blockIdx.x * 16 + cg, chunk = blockIdx.y, tap = blockIdx.z;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const long long P = (long long)a.N * a.Ho * a.Wo;
    const long long p0 = (long long)chunk * chunk_pixels, p1 = p0 + chunk_pixels < P ? p0 + chunk_pixels : P;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < C4)
        for (long long p = p0 + pl; p < p1; p += 16) {
            const int ow = (int)(p % a.Wo);
            const long long q = p / a.Wo;
            const int oh = (int)(q % a.Ho), n = (int)(q / a.Ho);
            const int ih = oh * a.stride - a.pad + kh * a.dil, iw = ow * a.stride - a.pad + kw * a.dil;
            if ((unsigned)ih >= (unsigned)a.H || (unsigned)iw >= (unsigned)a.W) continue;
            const float4 g = *reinterpret_cast<const float4*>(dy + (size_t)p * a.ldy + c4 * 4);
            const float4 xv = *reinterpret_cast<const float4*>(x + ((size_t)(n * a.H + ih) * a.W + iw) * a.ldx + c4 * 4);
            acc.x = fmaf(g.x, xv.x, acc.x);
            acc.y = fmaf(g.y, xv.y, acc.y);
            acc.z = fmaf(g.z, xv.z, acc.z);
            acc.w = fmaf(g.w, xv.w, acc.w);
        }
    red[pl][cg] = acc;
    __syncthreads();
    if (pl == 0 && c4 < C4) {
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        for (int k = 0; k < 16; ++k) {      // fixed order
            s0 += red[k][cg].x; s1 += red[k][cg].y; s2 += red[k][cg].z; s3 += red[k][cg].w;
        }
        double* o = partial + ((size_t)chunk * T + tap) * a.C + c4 * 4;
        o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
    }
}

// stage 2: dw[c][tap] = sum over chunks (fixed order, double)
__global__ __launch_bounds__(256) void k_dwconv_wgrad_final(const double* __restrict__ partial, int chunks, int T, int C,
                                                            int Cw, float* __restrict__ dw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Cw * T) return;
    const int c = i / T, tap = i - c * T;
    double s = 0.0;
    for (int k = 0; k < chunks; ++k) s += partial[((size_t)k * T + tap) * C + c];
    dw[i] = (float)s;
}

static int dw_chunks(long long P) {
    long long c = (P + 4095) / 4096;
    if (c > 256) c = 256;
    if (c < 1) c = 1;
    return (int)c;
}

}  // namespace iswm

using namespace iswm;

static int dw_validate(const iswm_conv_desc* d, int Cw, const char* what) {
    ISWM_REQUIRE(d, "%s: null descriptor", what);
    ISWM_REQUIRE(d->Cin == d->Cout && d->Cin % 4 == 0, "%s: depthwise needs Cin == Cout, a multiple of 4", what);
    ISWM_REQUIRE(Cw > 0 && Cw <= d->Cin, "%s: weight channels %d outside (0, %d]", what, Cw, d->Cin);
    ISWM_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0,
                 "%s: bad geometry", what);
    ISWM_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1 &&
                     d->Wo == (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1,
                 "%s: Ho/Wo do not match the geometry", what);
    ISWM_REQUIRE(d->ldx >= d->Cin && d->ldy >= d->Cin && d->ldx % 4 == 0 && d->ldy % 4 == 0, "%s: bad pitch", what);
    return 0;
}

static DwArgs dw_args(const iswm_conv_desc* d, int Cw) {
    DwArgs a{};
    a.N = d->N; a.H = d->H; a.W = d->W; a.C = d->Cin; a.Cw = Cw; a.Ho = d->Ho; a.Wo = d->Wo; a.KH = d->KH; a.KW = d->KW;
    a.stride = d->stride; a.pad = d->pad; a.dil = d->dil; a.ldx = d->ldx; a.ldy = d->ldy;
    return a;
}

/* w: the depthwise parameter [Cw][1][KH][KW] as stored by torch; the activation may carry C = d->Cin >= Cw channels
 * (zero-padded buffers), the extra channels see zero weights. */
extern "C" int iswm_dwconv2d_fwd(const iswm_conv_desc* d, const float* x, const float* w, int Cw, const float* bias,
                                 float* y, iswm_stream_t stream) {
    if (int e = dw_validate(d, Cw, "dwconv_fwd")) return e;
    ISWM_REQUIRE(x && w && y && aligned16(x) && aligned16(y), "dwconv_fwd: bad pointer");
    DwArgs a = dw_args(d, Cw);
    a.x = x; a.w = w; a.bias = bias; a.y = y;
    const long long total = (long long)d->N * d->Ho * d->Wo * (d->Cin / 4);
    hipLaunchKernelGGL(k_dwconv_fwd, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("dwconv_fwd");
}

extern "C" int iswm_dwconv2d_dgrad(const iswm_conv_desc* d, const float* dy, const float* w, int Cw, float* dx,
                                   int accumulate, iswm_stream_t stream) {
    if (int e = dw_validate(d, Cw, "dwconv_dgrad")) return e;
    ISWM_REQUIRE(dy && w && dx && aligned16(dy) && aligned16(dx), "dwconv_dgrad: bad pointer");
    DwArgs a = dw_args(d, Cw);
    a.x = dy; a.w = w; a.y = dx; a.accumulate = accumulate;
    a.ldx = d->ldy; a.ldy = d->ldx;      // kernel naming: a.x = dy (pitch ldy), a.y = dx (pitch ldx)
    const long long total = (long long)d->N * d->H * d->W * (d->Cin / 4);
    hipLaunchKernelGGL(k_dwconv_dgrad, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("dwconv_dgrad");
}

extern "C" size_t iswm_dwconv2d_wgrad_workspace(const iswm_conv_desc* d) {
    if (!d) return 0;
    return (size_t)dw_chunks((long long)d->N * d->Ho * d->Wo) * d->KH * d->KW * d->Cin * sizeof(double);
}

/* dw[Cw][KH*KW] = sum over pixels of dy * gathered x (fixed summation order) */
extern "C" int iswm_dwconv2d_wgrad(const iswm_conv_desc* d, const float* x, const float* dy, int Cw, float* dw,
                                   void* workspace, size_t workspace_bytes, iswm_stream_t stream) {
    if (int e = dw_validate(d, Cw, "dwconv_wgrad")) return e;
    ISWM_REQUIRE(x && dy && dw && workspace && aligned16(x) && aligned16(dy) && aligned16(workspace),
                 "dwconv_wgrad: bad pointer");
    ISWM_REQUIRE(workspace_bytes >= iswm_dwconv2d_wgrad_workspace(d), "dwconv_wgrad: workspace too small");
    DwArgs a = dw_args(d, Cw);
    const long long P = (long long)d->N * d->Ho * d->Wo;
    const int chunks = dw_chunks(P), T = d->KH * d->KW;
    const int chunk_pixels = (int)((P + chunks - 1) / chunks);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_dwconv_wgrad_partial, dim3((d->Cin / 4 + 15) / 16, chunks, T), dim3(256), 0, s, x, dy, a,
                       chunk_pixels, (double*)workspace);
    hipLaunchKernelGGL(k_dwconv_wgrad_final, dim3((Cw * T + 255) / 256), dim3(256), 0, s, (const double*)workspace, chunks,
                       T, d->Cin, Cw, dw);
    return check_launch("dwconv_wgrad");
}
