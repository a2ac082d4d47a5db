// Small HBM-bound helpers: channel-slice copies (torch.cat and its backward,
// network/_deeplab.py:59,171), in-place add / scale, and nn.Dropout(0.1)
// (network/_deeplab.py:165) with a counter-based Philox4x32-10 mask.
#include <stdarg.h>
#include <string.h>

#include "rowmap.h"

namespace iswm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

__global__ __launch_bounds__(256) void k_copy_channels(const float* __restrict__ src, int lds, float* __restrict__ dst,
                                                       int ldd, int64_t M, int C4, int CQ, int RL) {
    RowThread rt = row_thread(C4, CQ, RL);
    if (!rt.active) return;
    const int c = rt.c4 * 4;
    for (int64_t r = rt.row0; r < M; r += rt.rstep) st4(dst + r * ldd + c, ld4(src + r * lds + c));
}

__global__ __launch_bounds__(256) void k_add_inplace(float* __restrict__ dst, const float* __restrict__ src,
                                                     int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<float4*>(dst)[i], b = reinterpret_cast<const float4*>(src)[i];
        reinterpret_cast<float4*>(dst)[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] += src[n4 * 4 + threadIdx.x];
}

__global__ __launch_bounds__(256) void k_scale_inplace(float* __restrict__ x, int64_t n,
                                                       const float* __restrict__ scalar, float mul) {
    const float f = (scalar ? scalar[0] : 1.f) * mul;
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<float4*>(x)[i];
        reinterpret_cast<float4*>(x)[i] = make_float4(a.x * f, a.y * f, a.z * f, a.w * f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) x[n4 * 4 + threadIdx.x] *= f;
}

// Conv2d whose Cin / Cout is not a multiple of 4 (the 3-channel stem, the 304-wide decoder input, the 48-wide projection, the
// num_classes-wide classifier; nn.Conv2d at network/_deeplab.py:44-52, backbone/resnet.py:137): the kernels take a zero-padded
// OHWI copy of the parameter and hand back a padded OHWI gradient.  w / grad: the OIHW parameter with its own element strides.
__global__ __launch_bounds__(256) void k_pad_weights(const float* __restrict__ w, int cout, int cin, int kh, int kw, int64_t so,
                                                     int64_t si, int64_t sh, int64_t sw, int cin_p, int64_t total,
                                                     float* __restrict__ out) {
    const int64_t idx = blockIdx.x * 256ll + threadIdx.x;
    if (idx >= total) return;
    const int i = (int)(idx % cin_p);
    int64_t r = idx / cin_p;
    const int x = (int)(r % kw);
    r /= kw;
    const int y = (int)(r % kh);
    const int o = (int)(r / kh);
    out[idx] = (o < cout && i < cin) ? w[o * so + i * si + y * sh + x * sw] : 0.f;
}

__global__ __launch_bounds__(256) void k_unpad_weights(const float* __restrict__ dw, int cin, int kh, int kw, int cin_p,
                                                       int64_t total, float* __restrict__ grad, int64_t so, int64_t si,
                                                       int64_t sh, int64_t sw) {
    const int64_t idx = blockIdx.x * 256ll + threadIdx.x;          // over the UNPADDED [cout][kh][kw][cin]
    if (idx >= total) return;
    const int i = (int)(idx % cin);
    int64_t r = idx / cin;
    const int x = (int)(r % kw);
    r /= kw;
    const int y = (int)(r % kh);
    const int64_t o = r / kh;
    grad[o * so + i * si + y * sh + x * sw] = dw[((o * kh + y) * kw + x) * cin_p + i];
}

// zero the 8-byte-aligned byte range [b0, b1) of every row (pitch ld_bytes) of each of nplanes planes: the zero channels that pad
// a concatenation buffer / a few-channel gradient up to the kernels' channel granule
__global__ __launch_bounds__(256) void k_zero_cols(unsigned char* __restrict__ y, int64_t M, int64_t ld_bytes, int b0, int w8,
                                                   int64_t plane_bytes, int nplanes) {
    const int64_t total = M * w8 * nplanes;
    for (int64_t idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % w8);
        int64_t r = idx / w8;
        const int64_t m = r % M;
        const int pl = (int)(r / M);
        *reinterpret_cast<uint2*>(y + pl * plane_bytes + m * ld_bytes + b0 + 8 * c) = make_uint2(0u, 0u);
    }
}

// Philox4x32-10 (Salmon et al., SC'11): counter = (element group, offset), key = seed
__device__ __forceinline__ uint4 philox4x32(uint4 ctr, uint2 key) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
        uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += W0;
        key.y += W1;
    }
    return ctr;
}

__global__ __launch_bounds__(256) void k_dropout_fwd(const float* __restrict__ x, float* __restrict__ y,
                                                     uint8_t* __restrict__ mask, int64_t n, float p, uint64_t seed,
                                                     uint64_t offset) {
    const float scale = 1.f / (1.f - p);
    const int64_t n4 = (n + 3) >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        uint4 r = philox4x32(make_uint4((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)),
                             make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int64_t e = i * 4 + k;
            if (e < n) {
                float u = (float)(rr[k] >> 8) * (1.f / 16777216.f);  // uniform [0,1)
                uint8_t keep = u >= p;
                mask[e] = keep;
                y[e] = keep ? x[e] * scale : 0.f;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_dropout_bwd(const float* __restrict__ dy, const uint8_t* __restrict__ mask,
                                                     float* __restrict__ dx, int64_t n, float scale) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = mask[i] ? dy[i] * scale : 0.f;
}

}  // namespace iswm

using namespace iswm;

extern "C" const char* iswm_last_error(void) { return g_err; }
extern "C" int iswm_version(void) { return 100; }

extern "C" int iswm_copy_channels(const float* src, int lds, float* dst, int ldd, int64_t M, int C,
                                  iswm_stream_t stream) {
    ISWM_REQUIRE(src && dst && M > 0 && C > 0 && C % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && lds >= C && ldd >= C,
                 "copy_channels: bad argument");
    ISWM_REQUIRE(aligned16(src) && aligned16(dst), "copy_channels: pointers must be 16-byte aligned");
    RowPlan p = plan_rows(M, C);
    hipLaunchKernelGGL(k_copy_channels, dim3(p.rowblocks, p.colblocks), dim3(256), 0, (hipStream_t)stream, src, lds,
                       dst, ldd, M, p.C4, p.CQ, p.RL);
    return check_launch("copy_channels");
}

extern "C" int iswm_pad_weights(const float* w, int cout, int cin, int kh, int kw, const int64_t* strides, int cout_p, int cin_p,
                                float* out_ohwi, iswm_stream_t stream) {
    ISWM_REQUIRE(w && strides && out_ohwi && cout > 0 && cin > 0 && kh > 0 && kw > 0 && cout_p >= cout && cin_p >= cin,
                 "pad_weights: bad argument");
    const int64_t total = (int64_t)cout_p * kh * kw * cin_p;
    hipLaunchKernelGGL(k_pad_weights, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, cout, cin, kh, kw,
                       strides[0], strides[1], strides[2], strides[3], cin_p, total, out_ohwi);
    return check_launch("pad_weights");
}

extern "C" int iswm_unpad_weights(const float* dw_ohwi, int cout_p, int cin_p, int cout, int cin, int kh, int kw, float* grad,
                                  const int64_t* strides, iswm_stream_t stream) {
    ISWM_REQUIRE(dw_ohwi && strides && grad && cout > 0 && cin > 0 && kh > 0 && kw > 0 && cout_p >= cout && cin_p >= cin,
                 "unpad_weights: bad argument");
    const int64_t total = (int64_t)cout * kh * kw * cin;
    hipLaunchKernelGGL(k_unpad_weights, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dw_ohwi, cin, kh, kw,
                       cin_p, total, grad, strides[0], strides[1], strides[2], strides[3]);
    return check_launch("unpad_weights");
}

extern "C" int iswm_zero_cols(void* y, int64_t M, int64_t ld_bytes, int byte0, int byte1, int64_t plane_bytes, int nplanes,
                              iswm_stream_t stream) {
    ISWM_REQUIRE(y && M > 0 && nplanes > 0 && byte0 >= 0 && byte1 > byte0 && byte1 <= ld_bytes, "zero_cols: bad argument");
    ISWM_REQUIRE(((uintptr_t)y | (uintptr_t)ld_bytes | (uintptr_t)byte0 | (uintptr_t)byte1 | (uintptr_t)plane_bytes) % 8 == 0,
                 "zero_cols: pointer, pitch, range and plane stride must be multiples of 8 bytes");
    const int w8 = (byte1 - byte0) / 8;
    hipLaunchKernelGGL(k_zero_cols, dim3(stream_grid(M * w8 * nplanes, 256)), dim3(256), 0, (hipStream_t)stream,
                       (unsigned char*)y, M, ld_bytes, byte0, w8, plane_bytes, nplanes);
    return check_launch("zero_cols");
}

extern "C" int iswm_fill_zero(void* p, size_t bytes, iswm_stream_t stream) {
    ISWM_REQUIRE(p && bytes > 0, "fill_zero: bad argument");
    if (hipMemsetAsync(p, 0, bytes, (hipStream_t)stream) != hipSuccess) {
        set_error("fill_zero: hipMemsetAsync failed");
        return 1;
    }
    return 0;
}

extern "C" int iswm_add_inplace(float* dst, const float* src, int64_t n, iswm_stream_t stream) {
    ISWM_REQUIRE(dst && src && n > 0 && aligned16(dst) && aligned16(src), "add_inplace: bad argument");
    hipLaunchKernelGGL(k_add_inplace, dim3(stream_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, dst, src,
                       n);
    return check_launch("add_inplace");
}

extern "C" int iswm_scale_inplace(float* x, int64_t n, const float* scalar_dev, float host_mul,
                                  iswm_stream_t stream) {
    ISWM_REQUIRE(x && n > 0 && aligned16(x), "scale_inplace: bad argument");
    hipLaunchKernelGGL(k_scale_inplace, dim3(stream_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x, n,
                       scalar_dev, host_mul);
    return check_launch("scale_inplace");
}

extern "C" int iswm_dropout_fwd(const float* x, float* y, uint8_t* mask, int64_t n, float p, uint64_t seed,
                                uint64_t offset, iswm_stream_t stream) {
    ISWM_REQUIRE(x && y && mask && n > 0 && p >= 0.f && p < 1.f, "dropout_fwd: bad argument");
    hipLaunchKernelGGL(k_dropout_fwd, dim3(stream_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                       mask, n, p, seed, offset);
    return check_launch("dropout_fwd");
}

extern "C" int iswm_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, int64_t n, float p,
                                iswm_stream_t stream) {
    ISWM_REQUIRE(dy && mask && dx && n > 0 && p >= 0.f && p < 1.f, "dropout_bwd: bad argument");
    hipLaunchKernelGGL(k_dropout_bwd, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, mask, dx, n,
                       1.f / (1.f - p));
    return check_launch("dropout_bwd");
}
