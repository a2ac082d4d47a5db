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

template <typename LabelT>
__global__ __launch_bounds__(256) void k_loss_fwd(const float* __restrict__ logits,
                                                  const LabelT* __restrict__ labels, int C, int64_t HW,
                                                  int64_t npix, const float* __restrict__ cw, int ignore_index,
                                                  float alpha, float gamma, int mode,
                                                  float* __restrict__ grad, float* __restrict__ partials,
                                                  int nblocks) {
    __shared__ float red[2][4];
    float s1 = 0.f, s2 = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npix;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW, p = i - b * HW;
        const float* z = logits + b * C * HW + p;
        float* g = grad + b * C * HW + p;
        float m = z[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, z[c * HW]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(z[c * HW] - m);
        const float lse = m + logf(se);
        const long long y = (long long)labels[i];
        const bool valid = (y != (long long)ignore_index) && y >= 0 && y < C;
        float coef = 0.f;  // dL_i/dce_i * w[y]
        if (valid) {
            const float w = cw ? cw[y] : 1.f;
            const float nll = lse - z[y * HW];
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
        for (int c = 0; c < C; ++c) {
            float pc = expf(z[c * HW] - lse);
            g[c * HW] = valid ? coef * (pc - (c == (int)y ? 1.f : 0.f)) : 0.f;
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

__global__ void k_loss_finalize(const float* __restrict__ partials, int blocks, int mode, double npix, float* sums,
                                float* loss) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double a = 0.0, b = 0.0;
    for (int i = 0; i < blocks; ++i) {
        a += (double)partials[i];
        b += (double)partials[blocks + i];
    }
    sums[0] = (float)a;
    sums[1] = (float)b;
    if (loss) loss[0] = mode == 0 ? (float)(a / b) : (mode == 1 ? (float)(a / npix) : (float)a);
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
    int64_t b = (npix + 255) / 256;
    if (b > 1024) b = 1024;
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
    if (label_bytes == 1)
        hipLaunchKernelGGL((k_loss_fwd<uint8_t>), dim3(blocks), dim3(256), 0, s, logits, (const uint8_t*)labels, C,
                           HW, npix, class_weight, ignore_index, alpha, gamma, mode, grad_unnorm, partials, blocks);
    else
        hipLaunchKernelGGL((k_loss_fwd<int64_t>), dim3(blocks), dim3(256), 0, s, logits, (const int64_t*)labels, C,
                           HW, npix, class_weight, ignore_index, alpha, gamma, mode, grad_unnorm, partials, blocks);
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
