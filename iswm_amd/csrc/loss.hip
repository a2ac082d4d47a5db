// Fused class-weighted cross-entropy / focal loss over per-pixel NCHW logits, forward and
// backward in ONE pass over the logits (HBM-bound: per pixel C*4 B logits + 1|8 B label read,
// C*4 B gradient written; 24 B/pixel at C = 2 with int64 labels).
//
//   mode 0  nn.CrossEntropyLoss(weight=w, ignore_index, 'mean')     train.py:454-459
//           L = sum_i w[y_i] nll_i / sum_i w[y_i]      over y_i != ignore_index
//   mode 1  FocalLoss(size_average=True)                            utils/loss.py:23-35
//           ce_i = w[y_i] nll_i (0 if ignored), pt = exp(-ce), f = alpha (1-pt)^gamma ce,
//           L = mean over ALL pixels (ignored ones included in the denominator)
//   mode 2  FocalLoss(size_average=False):  L = sum_i f_i
//
// The kernel writes the gradient UNNORMALISED (dL_i/dz without the 1/sum_w or 1/npix
// factor) plus per-block partial sums; iswm_loss_finalize reduces those in a fixed order
// (double precision, bit-reproducible) and iswm_loss_bwd_scale applies upstream/normaliser.
// Under data parallelism the host all-reduces sums[2] between the two so that the
// normaliser is the GLOBAL sum of weights, as the reference's gathered-logits loss has it.
#include "common.h"

namespace iswm {

// one pixel: value terms and the unnormalised gradient coefficient
__device__ __forceinline__ void loss_pixel(const float* zc, int C, long long y, const float* __restrict__ cw, int ignore_index,
                                           float alpha, float gamma, int mode, float& s1, float& s2, float& coef, float& lse,
                                           bool& valid) {
    float m = zc[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, zc[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(zc[c] - m);
    lse = m + logf(se);
    valid = (y != (long long)ignore_index) && y >= 0 && y < C;
    coef = 0.f;  // dL_i/dce_i * w[y]
    if (valid) {
        const float w = cw ? cw[y] : 1.f;
        float zy = zc[0];
        for (int c = 1; c < C; ++c) zy = (c == (int)y) ? zc[c] : zy;
        const float nll = lse - zy;
        if (mode == 0) {
            s1 += w * nll;
            s2 += w;
            coef = w;
        } else {
            const float ce = w * nll;
            float f, dfdce;
            if (gamma == 0.f) {
                f = alpha * ce;
                dfdce = alpha;
            } else {
                const float pt = expf(-ce);
                const float om = 1.f - pt;
                const float pw = powf(om, gamma);
                f = alpha * pw * ce;
                dfdce = (ce > 0.f && om > 0.f) ? alpha * (pw + gamma * powf(om, gamma - 1.f) * pt * ce) : 0.f;
            }
            s1 += f;
            s2 += w;
            coef = dfdce * w;
        }
    }
}

// UNR independent pixels per thread and iteration (consecutive threads take consecutive pixels: every load and store of
// a wave is one contiguous 256-B run): 4x the bytes in flight of the one-pixel loop, which ran at 1.7 TB/s.
// CT > 0: class count known at compile time (2: the binary internal-wave masks), logits held in registers.
template <typename LabelT, int CT>
__global__ __launch_bounds__(256) void k_loss_fwd(const float* __restrict__ logits,
                                                  const LabelT* __restrict__ labels, int Crt, int64_t HW,
                                                  int64_t npix, const float* __restrict__ cw, int ignore_index,
                                                  float alpha, float gamma, int mode,
                                                  float* __restrict__ grad, float* __restrict__ partials,
                                                  int nblocks) {
    constexpr int UNR = 4, CMAX = CT > 0 ? CT : 8;
    const int C = CT > 0 ? CT : Crt;
    __shared__ float red[2][4];
    float s1 = 0.f, s2 = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i0 < npix; i0 += UNR * stride) {
        if (CT > 0 || C <= CMAX) {
            float z[UNR][CMAX];
            long long y[UNR];
            int64_t off[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int64_t i = i0 + u * stride;
                const bool in = i < npix;
                const int64_t ii = in ? i : 0;
                const int64_t b = ii / HW, p = ii - b * HW;
                off[u] = in ? b * C * HW + p : -1;
                y[u] = in ? (long long)labels[ii] : (long long)ignore_index;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) z[u][c] = (in && c < C) ? logits[b * C * HW + c * HW + p] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (off[u] < 0) continue;
                float coef, lse;
                bool valid;
                loss_pixel(z[u], C, y[u], cw, ignore_index, alpha, gamma, mode, s1, s2, coef, lse, valid);
#pragma unroll
                for (int c = 0; c < CMAX; ++c)
                    if (c < C) {
                        const float pc = expf(z[u][c] - lse);
                        grad[off[u] + c * HW] = valid ? coef * (pc - (c == (int)y[u] ? 1.f : 0.f)) : 0.f;
                    }
            }
        } else {                                    // many classes: stream the class axis from memory
            for (int u = 0; u < UNR; ++u) {
                const int64_t i = i0 + u * stride;
                if (i >= npix) break;
                const int64_t b = i / HW, p = i - b * HW;
                const float* zp = logits + b * C * HW + p;
                float m = zp[0];
                for (int c = 1; c < C; ++c) m = fmaxf(m, zp[c * HW]);
                float se = 0.f;
                for (int c = 0; c < C; ++c) se += expf(zp[c * HW] - m);
                const float lse = m + logf(se);
                const long long yy = (long long)labels[i];
                const bool valid = (yy != (long long)ignore_index) && yy >= 0 && yy < C;
                float coef = 0.f;
                if (valid) {
                    const float zy = zp[yy * HW];
                    const float w = cw ? cw[yy] : 1.f;
                    const float nll = lse - zy;
                    if (mode == 0) {
                        s1 += w * nll;
                        s2 += w;
                        coef = w;
                    } else {
                        const float ce = w * nll;
                        float f, dfdce;
                        if (gamma == 0.f) {
                            f = alpha * ce;
                            dfdce = alpha;
                        } else {
                            const float pt = expf(-ce);
                            const float om = 1.f - pt;
                            const float pw = powf(om, gamma);
                            f = alpha * pw * ce;
                            dfdce = (ce > 0.f && om > 0.f) ? alpha * (pw + gamma * powf(om, gamma - 1.f) * pt * ce) : 0.f;
                        }
                        s1 += f;
                        s2 += w;
                        coef = dfdce * w;
                    }
                }
                float* g = grad + b * C * HW + p;
                for (int c = 0; c < C; ++c) {
                    const float pc = expf(zp[c * HW] - lse);
                    g[c * HW] = valid ? coef * (pc - (c == (int)yy ? 1.f : 0.f)) : 0.f;
                }
            }
        }
    }
    // block reduction: wave shuffle, then across the 4 waves (fixed order)
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_down(s1, o);
        s2 += __shfl_down(s2, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = s1;
        red[1][wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partials[nblocks + blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// one wave: lane l sums partials l, l + 64, ... in double (fixed order), then a shuffle tree -- the single-thread loop over
// the partials took longer than the loss pass itself
__global__ __launch_bounds__(64) void k_loss_finalize(const float* __restrict__ partials, int blocks, int mode, double npix,
                                                      float* sums, float* loss) {
    const int lane = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int i = lane; i < blocks; i += 64) {
        a += (double)partials[i];
        b += (double)partials[blocks + i];
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o);
        b += __shfl_down(b, o);
    }
    if (lane == 0) {
        sums[0] = (float)a;
        sums[1] = (float)b;
        if (loss) loss[0] = mode == 0 ? (float)(a / b) : (mode == 1 ? (float)(a / npix) : (float)a);
    }
}

__global__ __launch_bounds__(256) void k_loss_bwd_scale(float* __restrict__ grad, int64_t n,
                                                        const float* __restrict__ sums,
                                                        const float* __restrict__ upstream, int mode,
                                                        float inv_npix) {
    float f = upstream ? upstream[0] : 1.f;
    f *= mode == 0 ? 1.f / sums[1] : (mode == 1 ? inv_npix : 1.f);
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<float4*>(grad)[i];
        reinterpret_cast<float4*>(grad)[i] = make_float4(v.x * f, v.y * f, v.z * f, v.w * f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) grad[n4 * 4 + threadIdx.x] *= f;
}

__global__ __launch_bounds__(256) void k_argmax_nchw(const float* __restrict__ logits, int C, int64_t HW,
                                                     int64_t npix, int64_t* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW, p = i - b * HW;
        const float* z = logits + b * C * HW + p;
        float best = z[0];
        int bi = 0;
        for (int c = 1; c < C; ++c) {
            float v = z[c * HW];
            if (v > best) {  // strict: ties keep the lowest class index (torch.max semantics)
                best = v;
                bi = c;
            }
        }
        out[i] = bi;
    }
}

}  // namespace iswm

using namespace iswm;

extern "C" int iswm_loss_blocks(int64_t npix) {
    int64_t b = (npix + 1023) / 1024;      // 4 pixels per thread and iteration
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int iswm_loss_fwd(const float* logits, const void* labels, int label_bytes, int B, int C, int64_t HW,
                             const float* class_weight, int ignore_index, float alpha, float gamma, int mode,
                             float* grad_unnorm, float* partials, iswm_stream_t stream) {
    ISWM_REQUIRE(logits && labels && grad_unnorm && partials, "loss_fwd: null pointer");
    ISWM_REQUIRE(B > 0 && C > 0 && HW > 0 && mode >= 0 && mode <= 2, "loss_fwd: bad shape or mode");
    ISWM_REQUIRE(label_bytes == 1 || label_bytes == 8, "loss_fwd: labels must be uint8 or int64");
    const int64_t npix = (int64_t)B * HW;
    const int blocks = iswm_loss_blocks(npix);
    hipStream_t s = (hipStream_t)stream;
#define LLAUNCH(T, CT)                                                                                             \
    hipLaunchKernelGGL((k_loss_fwd<T, CT>), dim3(blocks), dim3(256), 0, s, logits, (const T*)labels, C, HW, npix,  \
                       class_weight, ignore_index, alpha, gamma, mode, grad_unnorm, partials, blocks)
    if (label_bytes == 1) {
        if (C == 2) LLAUNCH(uint8_t, 2);
        else LLAUNCH(uint8_t, 0);
    } else {
        if (C == 2) LLAUNCH(int64_t, 2);
        else LLAUNCH(int64_t, 0);
    }
#undef LLAUNCH
    return check_launch("loss_fwd");
}

extern "C" int iswm_loss_finalize(const float* partials, int blocks, int mode, int64_t npix, float* sums,
                                  float* loss, iswm_stream_t stream) {
    ISWM_REQUIRE(partials && sums && blocks > 0, "loss_finalize: bad argument");
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(64), 0, (hipStream_t)stream, partials, blocks, mode,
                       (double)npix, sums, loss);
    return check_launch("loss_finalize");
}

extern "C" int iswm_loss_bwd_scale(float* grad, int64_t n, const float* sums, const float* upstream, int mode,
                                   int64_t npix, iswm_stream_t stream) {
    ISWM_REQUIRE(grad && sums && n > 0 && aligned16(grad), "loss_bwd_scale: bad argument");
    hipLaunchKernelGGL(k_loss_bwd_scale, dim3(stream_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, grad,
                       n, sums, upstream, mode, 1.f / (float)npix);
    return check_launch("loss_bwd_scale");
}

extern "C" int iswm_argmax_nchw(const float* logits, int B, int C, int64_t HW, int64_t* out, iswm_stream_t stream) {
    ISWM_REQUIRE(logits && out && B > 0 && C > 0 && HW > 0, "argmax: bad argument");
    const int64_t npix = (int64_t)B * HW;
    hipLaunchKernelGGL(k_argmax_nchw, dim3(stream_grid(npix, 256)), dim3(256), 0, (hipStream_t)stream, logits, C,
                       HW, npix, out);
    return check_launch("argmax");
}
