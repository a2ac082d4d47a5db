// Fused optimizer steps over one flat fp32 arena (all parameters of the model live in one
// allocation, gradients in a second, optimizer state in a third -- a single launch updates
// the 40-59 M parameters of DeepLabV3+-ResNet50/101).  HBM-bound: SGD-momentum reads p,g,m and
// writes p,m = 20 B/param; Adam reads p,g,m,v and writes p,m,v = 28 B/param.
//
// Arithmetic restates torch.optim.SGD(momentum, nesterov, weight_decay) and
// torch.optim.Adam / AdamW (defaults) exactly as setup_optimizer builds them, train.py:421-444.
// The learning rate (and Adam's bias corrections) are read from device memory so a captured
// hipGraph can be replayed while the host-side CosineAnnealingLR (train.py:446-452) changes them.
#include "common.h"

namespace iswm {

__global__ __launch_bounds__(256) void k_sgd(float* __restrict__ p, const float* __restrict__ g,
                                             float* __restrict__ buf, int64_t n4, int64_t n,
                                             const float* __restrict__ lr_dev, float mu, float wd, int nesterov) {
    const float lr = lr_dev[0];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 bv = reinterpret_cast<float4*>(buf)[i];
        float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d = gg[k] + wd * pp[k];      // g + wd*p
            bb[k] = mu * bb[k] + d;            // buf = mu*buf + g   (buf starts at 0 == torch's first-step clone)
            float u = nesterov ? d + mu * bb[k] : bb[k];
            pp[k] = pp[k] - lr * u;
        }
        reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        reinterpret_cast<float4*>(buf)[i] = make_float4(bb[0], bb[1], bb[2], bb[3]);
    }
    if (blockIdx.x == 0) {
        int64_t i = n4 * 4 + threadIdx.x;
        if (i < n) {
            float d = g[i] + wd * p[i];
            float b = mu * buf[i] + d;
            buf[i] = b;
            p[i] = p[i] - lr * (nesterov ? d + mu * b : b);
        }
    }
}

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr, float bc1, float bc2s,
                                         float b1, float b2, float eps, float wd, int decoupled) {
    if (decoupled) p = p * (1.f - lr * wd);
    else g = g + wd * p;
    m = m + (g - m) * (1.f - b1);                 // exp_avg.lerp_(g, 1-b1)
    v = v * b2 + (1.f - b2) * g * g;              // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    float denom = sqrtf(v) / bc2s + eps;
    p = p - (lr / bc1) * (m / denom);
}

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v, int64_t n,
                                              const float* __restrict__ hyper, float b1, float b2, float eps,
                                              float wd, int decoupled) {
    const float lr = hyper[0], bc1 = hyper[1], bc2s = sqrtf(hyper[2]);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pp = p[i], mm = m[i], vv = v[i];
        adam_one(pp, g[i], mm, vv, lr, bc1, bc2s, b1, b2, eps, wd, decoupled);
        p[i] = pp;
        m[i] = mm;
        v[i] = vv;
    }
}

}  // namespace iswm

using namespace iswm;

extern "C" int iswm_sgd_step(float* p, const float* g, float* buf, int64_t n, const float* lr_dev, float momentum,
                             float weight_decay, int nesterov, iswm_stream_t stream) {
    ISWM_REQUIRE(p && g && buf && lr_dev && n > 0, "sgd_step: bad argument");
    ISWM_REQUIRE(aligned16(p) && aligned16(g) && aligned16(buf), "sgd_step: arenas must be 16-byte aligned");
    hipLaunchKernelGGL(k_sgd, dim3(stream_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, p, g, buf,
                       n / 4, n, lr_dev, momentum, weight_decay, nesterov);
    return check_launch("sgd_step");
}

extern "C" int iswm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper_dev,
                              float beta1, float beta2, float eps, float weight_decay, int decoupled,
                              iswm_stream_t stream) {
    ISWM_REQUIRE(p && g && m && v && hyper_dev && n > 0, "adam_step: bad argument");
    hipLaunchKernelGGL(k_adam, dim3(stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       hyper_dev, beta1, beta2, eps, weight_decay, decoupled);
    return check_launch("adam_step");
}
