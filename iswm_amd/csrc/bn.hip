// BatchNorm2d (training + eval) with fused ReLU / residual add, forward and backward,
// over pitched NHWC tensors.  Replaces nn.BatchNorm2d + nn.ReLU + FloatFunctional.add of
// network/backbone/resnet.py:99-120 and network/_deeplab.py:38-39,125-126,135-136,150-151,163-164.
//
// All of these are HBM-bound (a few flops per 16-byte access); see rowmap.h for the
// thread mapping.  Statistics are reduced in two deterministic stages (per-tile partial
// sums, then one double-precision pass over the tiles) -- no float atomics, so results
// are bit-reproducible run to run.
#include <stdlib.h>

#include "rowmap.h"

namespace iswm {

// ---- per-channel tile statistics: sum and centred sum of squares over a row range ---------
// tile t covers rows [t*R, min(M, (t+1)*R)); two passes over the tile (the second re-reads it
// through L2) so that M2 is taken about the tile mean -- see iswm_bn_finalize.
__global__ __launch_bounds__(256) void k_colstat(const float* __restrict__ x, int64_t M, int C4, int ld,
                                                 int CQ, int RL, int tiles, int C, int64_t R,
                                                 float* __restrict__ partials) {
    __shared__ float red[256 * 4];
    __shared__ float mean_s[256 * 4];
    RowThread rt = row_thread(C4, CQ, RL);
    const int t = threadIdx.x;
    const int64_t r_begin = (int64_t)blockIdx.x * R;
    const int64_t r_end = min(M, r_begin + R);
    const float cnt = (float)(r_end - r_begin);
    const float* p = x + rt.c4 * 4;
    float4 s = make_float4(0, 0, 0, 0);
    if (rt.active)
        for (int64_t r = r_begin + rt.rl; r < r_end; r += RL) {
            float4 v = ld4(p + r * ld);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    st4(&red[t * 4], s);
    __syncthreads();
    if (rt.active && rt.rl == 0) {
        for (int k = 1; k < RL; ++k) {
            float4 a = ld4(&red[(t + k * CQ) * 4]);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        st4(&partials[(size_t)blockIdx.x * C + rt.c4 * 4], s);
        st4(&mean_s[t * 4], make_float4(s.x / cnt, s.y / cnt, s.z / cnt, s.w / cnt));
    }
    __syncthreads();
    float4 q = make_float4(0, 0, 0, 0);
    if (rt.active) {
        const float4 mu = ld4(&mean_s[(t - rt.rl * CQ) * 4]);
        for (int64_t r = r_begin + rt.rl; r < r_end; r += RL) {
            float4 v = ld4(p + r * ld);
            float dx = v.x - mu.x, dy = v.y - mu.y, dz = v.z - mu.z, dw = v.w - mu.w;
            q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
        }
    }
    st4(&red[t * 4], q);
    __syncthreads();
    if (rt.active && rt.rl == 0) {
        for (int k = 1; k < RL; ++k) {
            float4 a = ld4(&red[(t + k * CQ) * 4]);
            q.x += a.x; q.y += a.y; q.z += a.z; q.w += a.w;
        }
        st4(&partials[(size_t)(tiles + blockIdx.x) * C + rt.c4 * 4], q);
    }
}

// ---- merge tiles -> mean/var, running stats, scale/shift ------------------------------------
// partials[0][t][c] = S_t (sum over the tile's rows), partials[1][t][c] = M2_t (sum of squared
// deviations from the tile mean); tile t holds n_t = min(R, count - t*R) rows.
// mean = sum S_t / N;  M2 = sum [M2_t + n_t (S_t/n_t - mean)^2]   (Chan et al. pairwise update)
// block = CPB channels x (256 / CPB) tile lanes: many tiles (large maps) want more lanes per channel
template <int CPB>
__global__ __launch_bounds__(256) void k_bn_finalize(const float* __restrict__ partials, int tiles, int C,
                                                     double count, double R, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* running_mean,
                                                     float* running_var, float momentum, float eps,
                                                     float* scale, float* shift, float* save_mean,
                                                     float* save_invstd) {
    constexpr int TL = 256 / CPB;
    __shared__ double red[TL][CPB + 1];
    __shared__ double mean_s[CPB];
    const int cl = threadIdx.x % CPB, tl = threadIdx.x / CPB;
    const int c = blockIdx.x * CPB + cl;
    // (these kernels are pure latency: a few dependent L2 round trips between two launches.  Both partials of up to
    // KEEP tiles per thread are loaded ONCE, all in flight together, and the second pass runs out of registers: one round
    // trip for the usual 121-242 tiles at 64 tile lanes per channel, and for the 1849 tiles of a 129 x 129 map at 256)
    constexpr int KEEP = 8;
    float sv[KEEP], qv[KEEP];
    double s = 0.0;
    if (c < C) {
#pragma unroll
        for (int i = 0; i < KEEP; ++i) {
            const int k = tl + i * TL;
            sv[i] = k < tiles ? partials[(size_t)k * C + c] : 0.f;
            qv[i] = k < tiles ? partials[(size_t)(tiles + k) * C + c] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < KEEP; ++i) s += (double)sv[i];
        for (int k = tl + KEEP * TL; k < tiles; k += TL) s += (double)partials[(size_t)k * C + c];
    }
    red[tl][cl] = s;
    __syncthreads();
    if (tl == 0) {
        for (int k = 1; k < TL; ++k) s += red[k][cl];
        mean_s[cl] = s / count;
    }
    __syncthreads();
    const double mean = mean_s[cl];
    double m2 = 0.0;
    if (c < C) {
        auto term = [&](int k, float sk, float qk) -> double {
            const double nt = R > 0.0 ? fmin(R, count - (double)k * R) : (double)partials[(size_t)2 * tiles * C + k];
            const double d = (double)sk / nt - mean;
            return (double)qk + nt * d * d;
        };
#pragma unroll
        for (int i = 0; i < KEEP; ++i) {
            const int k = tl + i * TL;
            if (k < tiles) m2 += term(k, sv[i], qv[i]);
        }
        for (int k = tl + KEEP * TL; k < tiles; k += TL) m2 += term(k, partials[(size_t)k * C + c], partials[(size_t)(tiles + k) * C + c]);
    }
    __syncthreads();
    red[tl][cl] = m2;
    __syncthreads();
    if (tl == 0 && c < C) {
        for (int k = 1; k < TL; ++k) m2 += red[k][cl];
        double var = m2 / count;
        if (var < 0.0) var = 0.0;
        float invstd = (float)(1.0 / sqrt(var + (double)eps));
        float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        float sc = g * invstd;
        scale[c] = sc;
        shift[c] = b;   // bn_apply evaluates (y - mean)*scale + beta: no cancellation when |mean| >> std
        save_mean[c] = (float)mean;
        save_invstd[c] = invstd;
        if (running_mean) {
            double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

__global__ void k_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* rm,
                                 const float* rv, float eps, float* scale, float* shift, float* save_mean,
                                 float* save_invstd) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float invstd = 1.f / sqrtf(rv[c] + eps);
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b;
    save_mean[c] = rm[c];
    save_invstd[c] = invstd;
}

// ---- out = act(y*scale + shift (+ residual)) -------------------------------------------------
// RELU: 0 none, 1 ReLU, 2 ReLU6 (clamp to [0, 6])
template <int RELU, bool RES>
__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ y, int64_t M, int C4, int ldy,
                                                  const float* __restrict__ scale,
                                                  const float* __restrict__ shift,
                                                  const float* __restrict__ mean,
                                                  const void* __restrict__ res, int ldr, int64_t rps,
                                                  void* __restrict__ out, int ldo, int64_t ops, int CQ, int RL) {
    RowThread rt = row_thread(C4, CQ, RL);
    if (!rt.active) return;
    const int c = rt.c4 * 4;
    const float4 sc = ld4(scale + c), sh = ld4(shift + c), mu = ld4(mean + c);
    for (int64_t r = rt.row0; r < M; r += rt.rstep) {
        float4 v = ld4(y + r * ldy + c);
        float4 o;
        o.x = (v.x - mu.x) * sc.x + sh.x; o.y = (v.y - mu.y) * sc.y + sh.y;
        o.z = (v.z - mu.z) * sc.z + sh.z; o.w = (v.w - mu.w) * sc.w + sh.w;
        if (RES) {
            float4 q = ld4x(res, r * ldr + c, rps);
            o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
        }
        if (RELU) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        if (RELU == 2) {
            o.x = fminf(o.x, 6.f); o.y = fminf(o.y, 6.f); o.z = fminf(o.z, 6.f); o.w = fminf(o.w, 6.f);
        }
        st4x(out, r * ldo + c, ops, o);
    }
}

// 8 channels per thread (two float4 of y, 16-byte plane stores): the form used whenever the output is pre-split
template <int RELU, bool RES>
__global__ __launch_bounds__(256) void k_bn_apply8(const float* __restrict__ y, int64_t M, int C8, int ldy,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   const float* __restrict__ mean, const void* __restrict__ res, int ldr,
                                                   int64_t rps, void* __restrict__ out, int ldo, int64_t ops, int CQ, int RL) {
    RowThread rt = row_thread(C8, CQ, RL);
    if (!rt.active) return;
    const int c = rt.c4 * 8;
    const float4 sc0 = ld4(scale + c), sc1 = ld4(scale + c + 4), sh0 = ld4(shift + c), sh1 = ld4(shift + c + 4);
    const float4 mu0 = ld4(mean + c), mu1 = ld4(mean + c + 4);
    auto one = [&](int64_t r, const float4 v0, const float4 v1, const float4 q0, const float4 q1) {
        float4 o0, o1;
        o0.x = (v0.x - mu0.x) * sc0.x + sh0.x; o0.y = (v0.y - mu0.y) * sc0.y + sh0.y;
        o0.z = (v0.z - mu0.z) * sc0.z + sh0.z; o0.w = (v0.w - mu0.w) * sc0.w + sh0.w;
        o1.x = (v1.x - mu1.x) * sc1.x + sh1.x; o1.y = (v1.y - mu1.y) * sc1.y + sh1.y;
        o1.z = (v1.z - mu1.z) * sc1.z + sh1.z; o1.w = (v1.w - mu1.w) * sc1.w + sh1.w;
        if (RES) {
            o0.x += q0.x; o0.y += q0.y; o0.z += q0.z; o0.w += q0.w;
            o1.x += q1.x; o1.y += q1.y; o1.z += q1.z; o1.w += q1.w;
        }
        if (RELU) {
            o0.x = fmaxf(o0.x, 0.f); o0.y = fmaxf(o0.y, 0.f); o0.z = fmaxf(o0.z, 0.f); o0.w = fmaxf(o0.w, 0.f);
            o1.x = fmaxf(o1.x, 0.f); o1.y = fmaxf(o1.y, 0.f); o1.z = fmaxf(o1.z, 0.f); o1.w = fmaxf(o1.w, 0.f);
        }
        if (RELU == 2) {
            o0.x = fminf(o0.x, 6.f); o0.y = fminf(o0.y, 6.f); o0.z = fminf(o0.z, 6.f); o0.w = fminf(o0.w, 6.f);
            o1.x = fminf(o1.x, 6.f); o1.y = fminf(o1.y, 6.f); o1.z = fminf(o1.z, 6.f); o1.w = fminf(o1.w, 6.f);
        }
        st8x(out, r * ldo + c, ops, o0, o1);
    };
    // two rows per iteration: twice the loads in flight per thread (the one-row loop ran at 3.8 TB/s)
    int64_t r = rt.row0;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (; r + rt.rstep < M; r += 2 * rt.rstep) {
        const int64_t r2 = r + rt.rstep;
        const float4 a0 = ld4(y + r * ldy + c), a1 = ld4(y + r * ldy + c + 4);
        const float4 b0 = ld4(y + r2 * ldy + c), b1 = ld4(y + r2 * ldy + c + 4);
        float4 qa0 = z4, qa1 = z4, qb0 = z4, qb1 = z4;
        if (RES) {
            ld8x(res, r * ldr + c, rps, qa0, qa1);
            ld8x(res, r2 * ldr + c, rps, qb0, qb1);
        }
        one(r, a0, a1, qa0, qa1);
        one(r2, b0, b1, qb0, qb1);
    }
    if (r < M) {
        const float4 a0 = ld4(y + r * ldy + c), a1 = ld4(y + r * ldy + c + 4);
        float4 qa0 = z4, qa1 = z4;
        if (RES) ld8x(res, r * ldr + c, rps, qa0, qa1);
        one(r, a0, a1, qa0, qa1);
    }
}

// ---- backward stage 1: partial sums of dz and dz*xhat ----------------------------------------
// Sums and xhat are formed in double: ATen's CPU BatchNorm backward accumulates in double
// (acc_type<float, false>), and over a handful of samples (the ASPP image-pooling branch
// normalises over N x 1 x 1) dy = dz - mean(dz) - xhat*mean(dz*xhat) cancels to ~eps/(var+eps) of
// its terms, so fp32 rounding of xhat would show up at the 1e-3 level.  The kernels stay HBM-bound.
// RELU: 0 = no activation, 1 = ReLU mask read from the saved output (out > 0), 2 = ReLU mask RECOMPUTED from y with the
// forward's own expression (y - mean) * scale + shift > 0 (bit-identical to k_bn_apply, which has no residual in this
// case) -- the saved output is then never read: 8 instead of 12 bytes per element in this pass
template <int RELU, bool DBL>
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const float* __restrict__ dout, int ldd,
                                                       const void* __restrict__ out, int ldo, int64_t ops,
                                                       const float* __restrict__ y, int ldy, int64_t M,
                                                       int C4, int C, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd,
                                                       const float* __restrict__ mscale,
                                                       const float* __restrict__ mshift, int CQ, int RL,
                                                       int tiles, double* __restrict__ partials) {
    __shared__ double red[2 * 256 * 4];
    RowThread rt = row_thread(C4, CQ, RL);
    double s[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (rt.active) {
        const int c = rt.c4 * 4;
        const float4 mu = ld4(mean + c), is = ld4(invstd + c);
        float4 msc = make_float4(0.f, 0.f, 0.f, 0.f), msh = msc;
        if (RELU == 2) {
            msc = ld4(mscale + c);
            msh = ld4(mshift + c);
        }
        // fp32 partial sums over runs of 8 rows (4 loads x 3 tensors in flight per step), flushed into
        // double: keeps the double-accumulated result to ~1e-7 while staying load-bound, not DP-latency-bound
        for (int64_t r = rt.row0; r < M; r += 8 * rt.rstep) {
            float f[4] = {0.f, 0.f, 0.f, 0.f}, f2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t rr = r + u * rt.rstep;
                if (rr < M) {
                    float4 g = ld4(dout + rr * ldd + c);
                    float4 v = ld4(y + rr * ldy + c);
                    if (RELU == 1) {
                        float4 o = ld4x_hi(out, rr * ldo + c, ops);
                        g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
                        g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
                    } else if (RELU == 3) {         // ReLU6: the gradient passes where 0 < out < 6
                        float4 o = ld4x_hi(out, rr * ldo + c, ops);
                        g.x = (o.x > 0.f && o.x < 6.f) ? g.x : 0.f; g.y = (o.y > 0.f && o.y < 6.f) ? g.y : 0.f;
                        g.z = (o.z > 0.f && o.z < 6.f) ? g.z : 0.f; g.w = (o.w > 0.f && o.w < 6.f) ? g.w : 0.f;
                    } else if (RELU == 2) {
                        g.x = (v.x - mu.x) * msc.x + msh.x > 0.f ? g.x : 0.f;
                        g.y = (v.y - mu.y) * msc.y + msh.y > 0.f ? g.y : 0.f;
                        g.z = (v.z - mu.z) * msc.z + msh.z > 0.f ? g.z : 0.f;
                        g.w = (v.w - mu.w) * msc.w + msh.w > 0.f ? g.w : 0.f;
                    }
                    if (DBL) {   // few rows per channel (image-pooling branch): everything in double
                        const double gd[4] = {g.x, g.y, g.z, g.w}, vd[4] = {v.x, v.y, v.z, v.w};
                        const double mud[4] = {mu.x, mu.y, mu.z, mu.w}, isd[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            s[k] += gd[k];
                            s2[k] += gd[k] * ((vd[k] - mud[k]) * isd[k]);
                        }
                    } else {
                        f[0] += g.x; f[1] += g.y; f[2] += g.z; f[3] += g.w;
                        f2[0] += g.x * ((v.x - mu.x) * is.x); f2[1] += g.y * ((v.y - mu.y) * is.y);
                        f2[2] += g.z * ((v.z - mu.z) * is.z); f2[3] += g.w * ((v.w - mu.w) * is.w);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s[k] += (double)f[k];
                s2[k] += (double)f2[k];
            }
        }
    }
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[t * 4 + k] = s[k];
        red[(256 + t) * 4 + k] = s2[k];
    }
    __syncthreads();
    if (rt.active && rt.rl == 0) {
        for (int j = 1; j < RL; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s[k] += red[(t + j * CQ) * 4 + k];
                s2[k] += red[(256 + t + j * CQ) * 4 + k];
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            partials[(size_t)blockIdx.x * C + rt.c4 * 4 + k] = s[k];
            partials[(size_t)(tiles + blockIdx.x) * C + rt.c4 * 4 + k] = s2[k];
        }
    }
}

// dbeta = sum dz, dgamma = sum dz*xhat; sums[2][C] keeps them in double for stage 2
// block = CPB channels x (256 / CPB) tile lanes
template <typename T, int CPB>
__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const T* __restrict__ partials, int tiles, int C,
                                                         float* dgamma, float* dbeta, double* sums) {
    constexpr int TL = 256 / CPB;
    __shared__ double red[2][TL][CPB + 1];
    const int cl = threadIdx.x % CPB, tl = threadIdx.x / CPB;
    const int c = blockIdx.x * CPB + cl;
    double s = 0.0, s2 = 0.0;
    if (c < C) {
        int k = tl;
        for (; k + TL < tiles; k += 2 * TL) {          // four loads in flight per thread
            const T a0 = partials[(size_t)k * C + c], a1 = partials[(size_t)(k + TL) * C + c];
            const T b0 = partials[(size_t)(tiles + k) * C + c], b1 = partials[(size_t)(tiles + k + TL) * C + c];
            s += (double)a0 + (double)a1;
            s2 += (double)b0 + (double)b1;
        }
        for (; k < tiles; k += TL) {
            s += (double)partials[(size_t)k * C + c];
            s2 += (double)partials[(size_t)(tiles + k) * C + c];
        }
    }
    red[0][tl][cl] = s;
    red[1][tl][cl] = s2;
    __syncthreads();
    if (tl == 0 && c < C) {
        for (int k = 1; k < TL; ++k) {
            s += red[0][k][cl];
            s2 += red[1][k][cl];
        }
        dbeta[c] = (float)s;
        dgamma[c] = (float)s2;
        if (sums) {
            sums[c] = s;
            sums[C + c] = s2;
        }
    }
}

// ---- backward stage 2 ---------------------------------------------------------------------------
template <int RELU, bool TRAIN, bool DRES>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ dout, int ldd,
                                                      const void* __restrict__ out, int ldo, int64_t ops,
                                                      const float* __restrict__ y, int ldy, int64_t M,
                                                      int C4, int C, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ mscale,
                                                      const float* __restrict__ mshift,
                                                      const double* __restrict__ sums, double inv_count,
                                                      void* __restrict__ dy, int lddy, int64_t dyps,
                                                      float* __restrict__ dres, int lddres, int CQ, int RL) {
    RowThread rt = row_thread(C4, CQ, RL);
    if (!rt.active) return;
    const int c = rt.c4 * 4;
    const float4 mu = ld4(mean + c), is = ld4(invstd + c);
    const float4 ga = gamma ? ld4(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    const double mud[4] = {mu.x, mu.y, mu.z, mu.w}, isd[4] = {is.x, is.y, is.z, is.w};
    const double gi[4] = {(double)ga.x * is.x, (double)ga.y * is.y, (double)ga.z * is.z, (double)ga.w * is.w};
    float4 msc = make_float4(0.f, 0.f, 0.f, 0.f), msh = msc;
    if (RELU == 2) {
        msc = ld4(mscale + c);
        msh = ld4(mshift + c);
    }
    double k1[4] = {0, 0, 0, 0}, k2[4] = {0, 0, 0, 0};
    if (TRAIN) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            k1[k] = sums[c + k] * inv_count;
            k2[k] = sums[C + c + k] * inv_count;
        }
    }
    for (int64_t r = rt.row0; r < M; r += rt.rstep) {
        float4 g = ld4(dout + r * ldd + c);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (TRAIN || RELU == 2) v = ld4(y + r * ldy + c);
        if (RELU == 1) {
            float4 o = ld4x_hi(out, r * ldo + c, ops);
            g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
            g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
        } else if (RELU == 3) {
            float4 o = ld4x_hi(out, r * ldo + c, ops);
            g.x = (o.x > 0.f && o.x < 6.f) ? g.x : 0.f; g.y = (o.y > 0.f && o.y < 6.f) ? g.y : 0.f;
            g.z = (o.z > 0.f && o.z < 6.f) ? g.z : 0.f; g.w = (o.w > 0.f && o.w < 6.f) ? g.w : 0.f;
        } else if (RELU == 2) {
            g.x = (v.x - mu.x) * msc.x + msh.x > 0.f ? g.x : 0.f;
            g.y = (v.y - mu.y) * msc.y + msh.y > 0.f ? g.y : 0.f;
            g.z = (v.z - mu.z) * msc.z + msh.z > 0.f ? g.z : 0.f;
            g.w = (v.w - mu.w) * msc.w + msh.w > 0.f ? g.w : 0.f;
        }
        if (DRES) st4(dres + r * lddres + c, g);
        const double gd[4] = {g.x, g.y, g.z, g.w};
        float d[4];
        if (TRAIN) {
            const double vd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                d[k] = (float)(gi[k] * (gd[k] - k1[k] - (vd[k] - mud[k]) * isd[k] * k2[k]));
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = (float)(gi[k] * gd[k]);
        }
        st4x(dy, r * lddy + c, dyps, make_float4(d[0], d[1], d[2], d[3]));
    }
}

static int chk_rows(const char* what, int64_t M, int C, int ld) {
    ISWM_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 && ld >= C, "%s: bad shape M=%lld C=%d ld=%d", what,
                 (long long)M, C, ld);
    return 0;
}

}  // namespace iswm

using namespace iswm;

extern "C" int iswm_colstat_tiles(int64_t M) {
    // one workgroup per tile (per 256/1024-channel column block): ~32 rows each keeps >= 500 workgroups
    // in flight on the 33x33 stages (M = 17 424) -- 69 tiles of 256 rows ran at 2.7 TB/s
    int64_t t = (M + 31) / 32;
    if (t > 1024) t = 1024;
    if (t < 1) t = 1;
    return (int)t;
}

extern "C" int64_t iswm_colstat_tile_rows(int64_t M) {
    const int64_t tiles = iswm_colstat_tiles(M);
    return (M + tiles - 1) / tiles;
}

extern "C" int iswm_colstat(const float* x, int64_t M, int C, int ld, float* partials, iswm_stream_t stream) {
    if (int e = chk_rows("colstat", M, C, ld)) return e;
    ISWM_REQUIRE(x && partials && aligned16(x) && aligned16(partials), "colstat: bad pointer");
    const int64_t R = iswm_colstat_tile_rows(M);
    const int tiles = (int)((M + R - 1) / R);
    RowPlan p = plan_rows(M, C, tiles);
    hipLaunchKernelGGL(k_colstat, dim3(p.rowblocks, p.colblocks), dim3(256), 0, (hipStream_t)stream, x, M, p.C4,
                       ld, p.CQ, p.RL, tiles, C, R, partials);
    return check_launch("colstat");
}

extern "C" int iswm_bn_finalize(const float* partials, int tiles, int C, int64_t count, int64_t tile_rows,
                                const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                                float eps, float* scale, float* shift, float* save_mean, float* save_invstd,
                                iswm_stream_t stream) {
    ISWM_REQUIRE(partials && scale && shift && save_mean && save_invstd && tiles > 0 && C > 0 && count > 0 &&
                     tile_rows >= 0 && (tile_rows == 0 || (int64_t)tiles == (count + tile_rows - 1) / tile_rows),
                 "bn_finalize: bad argument (tiles %d, count %lld, tile_rows %lld)", tiles, (long long)count,
                 (long long)tile_rows);
    if (tiles > 512)
        hipLaunchKernelGGL((k_bn_finalize<1>), dim3(C), dim3(256), 0, (hipStream_t)stream, partials, tiles, C,
                           (double)count, (double)tile_rows, gamma, beta, running_mean, running_var, momentum, eps,
                           scale, shift, save_mean, save_invstd);
    else if (tiles > 32)
        hipLaunchKernelGGL((k_bn_finalize<4>), dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, partials, tiles, C,
                           (double)count, (double)tile_rows, gamma, beta, running_mean, running_var, momentum, eps,
                           scale, shift, save_mean, save_invstd);
    else
        hipLaunchKernelGGL((k_bn_finalize<16>), dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, partials, tiles,
                           C, (double)count, (double)tile_rows, gamma, beta, running_mean, running_var, momentum, eps,
                           scale, shift, save_mean, save_invstd);
    return check_launch("bn_finalize");
}

extern "C" int iswm_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float eps, float* scale, float* shift,
                                   float* save_mean, float* save_invstd, iswm_stream_t stream) {
    ISWM_REQUIRE(C > 0 && running_mean && running_var && scale && shift && save_mean && save_invstd,
                 "bn_eval_coeffs: bad argument");
    hipLaunchKernelGGL(k_bn_eval_coeffs, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, gamma, beta,
                       running_mean, running_var, eps, scale, shift, save_mean, save_invstd);
    return check_launch("bn_eval_coeffs");
}

static int chk_ps(const char* what, const void* p, int64_t M, int ld, int64_t ps) {
    ISWM_REQUIRE(ps == 0 || ps == -1 || (ps >= M * ld && ps % 4 == 0), "%s: bad plane stride %lld", what, (long long)ps);
    ISWM_REQUIRE(ps == 0 ? aligned16(p) : ((reinterpret_cast<uintptr_t>(p) & 7) == 0), "%s: misaligned tensor", what);
    return 0;
}

extern "C" int iswm_bn_apply(const float* y, int64_t M, int C, int ldy, const float* scale, const float* shift,
                             const float* mean, const float* residual, int ldr, int relu, float* out, int ldo,
                             iswm_stream_t stream) {
    return iswm_bn_apply_pl(y, M, C, ldy, scale, shift, mean, residual, ldr, 0, relu, out, ldo, 0, stream);
}

extern "C" int iswm_bn_apply_pl(const float* y, int64_t M, int C, int ldy, const float* scale, const float* shift,
                                const float* mean, const void* residual, int ldr, int64_t res_ps, int relu, void* out,
                                int ldo, int64_t out_ps, iswm_stream_t stream) {
    if (int e = chk_rows("bn_apply", M, C, ldy)) return e;
    ISWM_REQUIRE(y && scale && shift && mean && out && ldo % 4 == 0 && ldo >= C, "bn_apply: bad argument");
    ISWM_REQUIRE(!residual || (ldr % 4 == 0 && ldr >= C), "bn_apply: bad residual pitch");
    if (int e = chk_ps("bn_apply(out)", out, M, ldo, out_ps)) return e;
    if (residual) if (int e = chk_ps("bn_apply(residual)", residual, M, ldr, res_ps)) return e;
    const int64_t rps = res_ps, ops = out_ps;
    hipStream_t s = (hipStream_t)stream;
    if (out_ps != 0 && C % 8 == 0 && ldy % 8 == 0 && ldo % 8 == 0 && aligned16(out) && (out_ps < 0 || out_ps % 8 == 0) &&
        (!residual || (ldr % 8 == 0 && aligned16(residual) && (res_ps <= 0 || res_ps % 8 == 0)))) {
        RowPlan p = plan_rows(M, C / 2);       // C / 8 groups of 8 channels
        dim3 grid(p.rowblocks, p.colblocks), blk(256);
#define LAUNCH8(R, S) \
    hipLaunchKernelGGL((k_bn_apply8<R, S>), grid, blk, 0, s, y, M, p.C4, ldy, scale, shift, mean, residual, ldr, rps, out, \
                       ldo, ops, p.CQ, p.RL)
        ISWM_REQUIRE(relu == 0 || relu == 1 || relu == 6, "bn_apply: relu must be 0 (none), 1 (ReLU) or 6 (ReLU6)");
        if (relu == 6 && residual) LAUNCH8(2, true);
        else if (relu == 6) LAUNCH8(2, false);
        else if (relu && residual) LAUNCH8(1, true);
        else if (relu) LAUNCH8(1, false);
        else if (residual) LAUNCH8(0, true);
        else LAUNCH8(0, false);
#undef LAUNCH8
        return check_launch("bn_apply8");
    }
    RowPlan p = plan_rows(M, C);
    dim3 grid(p.rowblocks, p.colblocks), blk(256);
#define LAUNCH(R, S) \
    hipLaunchKernelGGL((k_bn_apply<R, S>), grid, blk, 0, s, y, M, p.C4, ldy, scale, shift, mean, residual, ldr, rps, out, \
                       ldo, ops, \
                       p.CQ, p.RL)
    ISWM_REQUIRE(relu == 0 || relu == 1 || relu == 6, "bn_apply: relu must be 0 (none), 1 (ReLU) or 6 (ReLU6)");
    if (relu == 6 && residual) LAUNCH(2, true);
    else if (relu == 6) LAUNCH(2, false);
    else if (relu && residual) LAUNCH(1, true);
    else if (relu) LAUNCH(1, false);
    else if (residual) LAUNCH(0, true);
    else LAUNCH(0, false);
#undef LAUNCH
    return check_launch("bn_apply");
}

extern "C" size_t iswm_bn_bwd_workspace(int64_t M, int C) {
    // double partials[2][tiles][C] + double sums[2][C]
    return ((size_t)2 * iswm_colstat_tiles(M) * C + (size_t)2 * C) * sizeof(double);
}

extern "C" int iswm_bn_backward(const float* dout, int ldd, const float* out, int ldo, const float* y, int ldy,
                                int64_t M, int C, const float* mean, const float* invstd, const float* gamma,
                                const float* mask_scale, const float* mask_shift,
                                int relu, int training, float* dgamma, float* dbeta, float* dy, int lddy,
                                float* dres, int lddres, void* workspace, size_t workspace_bytes,
                                iswm_stream_t stream) {
    return iswm_bn_backward_pl(dout, ldd, out, ldo, 0, y, ldy, M, C, mean, invstd, gamma, mask_scale, mask_shift, relu,
                               training, dgamma, dbeta, dy, lddy, 0, dres, lddres, workspace, workspace_bytes, stream);
}

// ready_partials != nullptr: the reduction pass was done by the producer of dout (iswm_conv2d_dgrad_pl2_bn)
static int bn_backward_impl(const float* dout, int ldd, const void* out, int ldo, int64_t out_ps, const float* y,
                            int ldy, int64_t M, int C, const float* mean, const float* invstd, const float* gamma,
                            const float* mask_scale, const float* mask_shift, int relu, int training,
                            float* dgamma, float* dbeta, void* dy, int lddy, int64_t dy_ps, float* dres,
                            int lddres, void* workspace, size_t workspace_bytes, const double* ready_partials,
                            int ready_tiles, iswm_stream_t stream) {
    if (int e = chk_rows("bn_backward", M, C, ldy)) return e;
    if (int e = chk_ps("bn_backward(dy)", dy, M, lddy, dy_ps)) return e;
    if (out) if (int e = chk_ps("bn_backward(out)", out, M, ldo, out_ps)) return e;
    const int64_t ops = out_ps, dyps = dy_ps;
    // ReLU without a residual: the sign pattern is recomputed from y when the forward's scale / shift are given
    ISWM_REQUIRE(relu == 0 || relu == 1 || relu == 6, "bn_backward: relu must be 0 (none), 1 (ReLU) or 6 (ReLU6)");
    const bool relu6 = relu == 6;
    const bool masky = relu == 1 && !dres && mask_scale && mask_shift;
    ISWM_REQUIRE(dout && y && mean && invstd && dgamma && dbeta && dy && workspace && (!relu || out || masky),
                 "bn_backward: null pointer");
    ISWM_REQUIRE(ldd % 4 == 0 && ldd >= C && lddy % 4 == 0 && lddy >= C && (!relu || (ldo % 4 == 0 && ldo >= C)) &&
                     (!dres || (lddres % 4 == 0 && lddres >= C)),
                 "bn_backward: bad pitch");
    ISWM_REQUIRE(workspace_bytes >= iswm_bn_bwd_workspace(M, C) && aligned16(workspace),
                 "bn_backward: workspace too small");
    const int tiles = ready_partials ? ready_tiles : iswm_colstat_tiles(M);
    const double* partials = ready_partials ? ready_partials : (const double*)workspace;
    double* sums = (double*)workspace + (size_t)2 * iswm_colstat_tiles(M) * C;
    hipStream_t s = (hipStream_t)stream;
    if (!ready_partials) {
        double* partials = (double*)workspace;
        RowPlan p = plan_rows(M, C, tiles);
        dim3 grid(p.rowblocks, p.colblocks), blk(256);
#define RLAUNCH(R, D)                                                                                           \
    hipLaunchKernelGGL((k_bn_bwd_reduce<R, D>), grid, blk, 0, s, dout, ldd, out, ldo, ops, y, ldy, M, p.C4, C, mean, invstd, \
                       mask_scale, mask_shift, p.CQ, p.RL, tiles, partials)
        const bool dbl = M <= 8192;
        if (relu6 && dbl) RLAUNCH(3, true);
        else if (relu6) RLAUNCH(3, false);
        else if (masky && dbl) RLAUNCH(2, true);
        else if (masky) RLAUNCH(2, false);
        else if (relu && dbl) RLAUNCH(1, true);
        else if (relu) RLAUNCH(1, false);
        else if (dbl) RLAUNCH(0, true);
        else RLAUNCH(0, false);
#undef RLAUNCH
        if (int e = check_launch("bn_bwd_reduce")) return e;
    }
    if (tiles > 32)
        hipLaunchKernelGGL((k_bn_bwd_finalize<double, 4>), dim3((C + 3) / 4), dim3(256), 0, s, partials, tiles, C,
                           dgamma, dbeta, sums);
    else
        hipLaunchKernelGGL((k_bn_bwd_finalize<double, 16>), dim3((C + 15) / 16), dim3(256), 0, s, partials, tiles, C,
                           dgamma, dbeta, sums);
    if (int e = check_launch("bn_bwd_finalize")) return e;
    const double inv = 1.0 / (double)M;
    // (an 8-channel form of the backward apply pass -- 16-byte plane stores -- measured 1.7 ms/step SLOWER: its double-precision
    // per-channel coefficients cost the occupancy the wider stores buy)
    RowPlan p = plan_rows(M, C);
    dim3 grid(p.rowblocks, p.colblocks), blk(256);
#define LAUNCH(R, T, D)                                                                                          \
    hipLaunchKernelGGL((k_bn_bwd_apply<R, T, D>), grid, blk, 0, s, dout, ldd, out, ldo, ops, y, ldy, M, p.C4, C, mean,    \
                       invstd, gamma, mask_scale, mask_shift, sums, inv, dy, lddy, dyps, dres, lddres, p.CQ, p.RL)
    const int key = (relu ? 4 : 0) | (training ? 2 : 0) | (dres ? 1 : 0);
    if (relu6) {
        if (training && dres) LAUNCH(3, true, true);
        else if (training) LAUNCH(3, true, false);
        else if (dres) LAUNCH(3, false, true);
        else LAUNCH(3, false, false);
    } else if (masky) {
        if (training) LAUNCH(2, true, false);
        else LAUNCH(2, false, false);
    } else switch (key) {
        case 0: LAUNCH(0, false, false); break;
        case 1: LAUNCH(0, false, true); break;
        case 2: LAUNCH(0, true, false); break;
        case 3: LAUNCH(0, true, true); break;
        case 4: LAUNCH(1, false, false); break;
        case 5: LAUNCH(1, false, true); break;
        case 6: LAUNCH(1, true, false); break;
        default: LAUNCH(1, true, true); break;
    }
#undef LAUNCH
    return check_launch("bn_bwd_apply");
}

extern "C" int iswm_bn_backward_pl(const float* dout, int ldd, const void* out, int ldo, int64_t out_ps, const float* y,
                                   int ldy, int64_t M, int C, const float* mean, const float* invstd, const float* gamma,
                                   const float* mask_scale, const float* mask_shift, int relu, int training,
                                   float* dgamma, float* dbeta, void* dy, int lddy, int64_t dy_ps, float* dres,
                                   int lddres, void* workspace, size_t workspace_bytes, iswm_stream_t stream) {
    return bn_backward_impl(dout, ldd, out, ldo, out_ps, y, ldy, M, C, mean, invstd, gamma, mask_scale, mask_shift, relu,
                            training, dgamma, dbeta, dy, lddy, dy_ps, dres, lddres, workspace, workspace_bytes, nullptr, 0,
                            stream);
}

/* iswm_bn_backward_pl whose first pass (sum dz, sum dz * xhat over the pixels) was already taken by the kernel that produced
 * dout: partials = [2][tiles][C] doubles as written by iswm_conv2d_dgrad_pl2_bn.  Finalize + apply only. */
extern "C" int iswm_bn_backward_stats_pl(const float* dout, int ldd, const void* out, int ldo, int64_t out_ps,
                                         const float* y, int ldy, int64_t M, int C, const float* mean,
                                         const float* invstd, const float* gamma, const float* mask_scale,
                                         const float* mask_shift, int relu, int training, float* dgamma, float* dbeta,
                                         void* dy, int lddy, int64_t dy_ps, float* dres, int lddres,
                                         const double* partials, int tiles, void* workspace, size_t workspace_bytes,
                                         iswm_stream_t stream) {
    ISWM_REQUIRE(partials && tiles > 0 && aligned16(partials), "bn_backward_stats: bad partials");
    return bn_backward_impl(dout, ldd, out, ldo, out_ps, y, ldy, M, C, mean, invstd, gamma, mask_scale, mask_shift, relu,
                            training, dgamma, dbeta, dy, lddy, dy_ps, dres, lddres, workspace, workspace_bytes, partials,
                            tiles, stream);
}

/* column sums of per-tile partials (bias gradient of a conv with bias): out[c] = sum_t partials[0][t][c] */
extern "C" int iswm_colsum_finalize(const float* partials, int tiles, int C, float* out, float* scratch,
                                    iswm_stream_t stream) {
    ISWM_REQUIRE(partials && out && scratch && tiles > 0 && C > 0, "colsum_finalize: bad argument");
    hipLaunchKernelGGL((k_bn_bwd_finalize<float, 4>), dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, partials,
                       tiles, C, scratch, out, (double*)nullptr);
    return check_launch("colsum_finalize");
}
