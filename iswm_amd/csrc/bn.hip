// BatchNorm2d (training + eval) with fused ReLU / residual add, forward and backward,
// over pitched NHWC tensors.  Replaces nn.BatchNorm2d + nn.ReLU + FloatFunctional.add of
// network/backbone/resnet.py:99-120 and network/_deeplab.py:38-39,125-126,135-136,150-151,163-164.
//
// All of these are HBM-bound (a few flops per 16-byte access); see rowmap.h for the
// thread mapping.  Statistics are reduced in two deterministic stages (per-tile partial
// sums, then one double-precision pass over the tiles) -- no float atomics, so results
// are bit-reproducible run to run.
#include "rowmap.h"

namespace iswm {

// ---- per-channel sum / sum of squares over rows --------------------------------------------
__global__ __launch_bounds__(256) void k_colstat(const float* __restrict__ x, int64_t M, int C4, int ld,
                                                 int CQ, int RL, int tiles, int C,
                                                 float* __restrict__ partials) {
    __shared__ float red[2 * 256 * 4];
    RowThread rt = row_thread(C4, CQ, RL);
    float4 s = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
    if (rt.active) {
        const float* p = x + rt.c4 * 4;
        for (int64_t r = rt.row0; r < M; r += rt.rstep) {
            float4 v = ld4(p + r * ld);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
        }
    }
    const int t = threadIdx.x;
    st4(&red[t * 4], s);
    st4(&red[(256 + t) * 4], s2);
    __syncthreads();
    if (rt.active && rt.rl == 0) {
        for (int k = 1; k < RL; ++k) {
            float4 a = ld4(&red[(t + k * CQ) * 4]), b = ld4(&red[(256 + t + k * CQ) * 4]);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
        }
        st4(&partials[(size_t)blockIdx.x * C + rt.c4 * 4], s);
        st4(&partials[(size_t)(tiles + blockIdx.x) * C + rt.c4 * 4], s2);
    }
}

// ---- reduce tiles -> mean/var, running stats, scale/shift ------------------------------------
// block = 16 channels x 16 tile lanes
__global__ __launch_bounds__(256) void k_bn_finalize(const float* __restrict__ partials, int tiles, int C,
                                                     double count, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* running_mean,
                                                     float* running_var, float momentum, float eps,
                                                     float* scale, float* shift, float* save_mean,
                                                     float* save_invstd) {
    __shared__ double red[2][16][17];
    const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0.0, s2 = 0.0;
    if (c < C)
        for (int k = tl; k < tiles; k += 16) {
            s += (double)partials[(size_t)k * C + c];
            s2 += (double)partials[(size_t)(tiles + k) * C + c];
        }
    red[0][tl][cl] = s;
    red[1][tl][cl] = s2;
    __syncthreads();
    if (tl == 0 && c < C) {
        for (int k = 1; k < 16; ++k) {
            s += red[0][k][cl];
            s2 += red[1][k][cl];
        }
        double mean = s / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        float invstd = (float)(1.0 / sqrt(var + (double)eps));
        float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        float sc = g * invstd;
        scale[c] = sc;
        shift[c] = b - (float)mean * sc;
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
    shift[c] = b - rm[c] * sc;
    save_mean[c] = rm[c];
    save_invstd[c] = invstd;
}

// ---- out = act(y*scale + shift (+ residual)) -------------------------------------------------
template <bool RELU, bool RES>
__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ y, int64_t M, int C4, int ldy,
                                                  const float* __restrict__ scale,
                                                  const float* __restrict__ shift,
                                                  const float* __restrict__ res, int ldr,
                                                  float* __restrict__ out, int ldo, int CQ, int RL) {
    RowThread rt = row_thread(C4, CQ, RL);
    if (!rt.active) return;
    const int c = rt.c4 * 4;
    const float4 sc = ld4(scale + c), sh = ld4(shift + c);
    for (int64_t r = rt.row0; r < M; r += rt.rstep) {
        float4 v = ld4(y + r * ldy + c);
        float4 o;
        o.x = v.x * sc.x + sh.x; o.y = v.y * sc.y + sh.y; o.z = v.z * sc.z + sh.z; o.w = v.w * sc.w + sh.w;
        if (RES) {
            float4 q = ld4(res + r * ldr + c);
            o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
        }
        if (RELU) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        st4(out + r * ldo + c, o);
    }
}

// ---- backward stage 1: partial sums of dz and dz*xhat ----------------------------------------
template <bool RELU>
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const float* __restrict__ dout, int ldd,
                                                       const float* __restrict__ out, int ldo,
                                                       const float* __restrict__ y, int ldy, int64_t M,
                                                       int C4, int C, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, int CQ, int RL,
                                                       int tiles, float* __restrict__ partials) {
    __shared__ float red[2 * 256 * 4];
    RowThread rt = row_thread(C4, CQ, RL);
    float4 s = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
    if (rt.active) {
        const int c = rt.c4 * 4;
        const float4 mu = ld4(mean + c), is = ld4(invstd + c);
        for (int64_t r = rt.row0; r < M; r += rt.rstep) {
            float4 g = ld4(dout + r * ldd + c);
            if (RELU) {
                float4 o = ld4(out + r * ldo + c);
                g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
                g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
            }
            float4 v = ld4(y + r * ldy + c);
            s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
            s2.x += g.x * ((v.x - mu.x) * is.x); s2.y += g.y * ((v.y - mu.y) * is.y);
            s2.z += g.z * ((v.z - mu.z) * is.z); s2.w += g.w * ((v.w - mu.w) * is.w);
        }
    }
    const int t = threadIdx.x;
    st4(&red[t * 4], s);
    st4(&red[(256 + t) * 4], s2);
    __syncthreads();
    if (rt.active && rt.rl == 0) {
        for (int k = 1; k < RL; ++k) {
            float4 a = ld4(&red[(t + k * CQ) * 4]), b = ld4(&red[(256 + t + k * CQ) * 4]);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
        }
        st4(&partials[(size_t)blockIdx.x * C + rt.c4 * 4], s);
        st4(&partials[(size_t)(tiles + blockIdx.x) * C + rt.c4 * 4], s2);
    }
}

// dbeta = sum dz, dgamma = sum dz*xhat
__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const float* __restrict__ partials, int tiles, int C,
                                                         float* dgamma, float* dbeta) {
    __shared__ double red[2][16][17];
    const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0.0, s2 = 0.0;
    if (c < C)
        for (int k = tl; k < tiles; k += 16) {
            s += (double)partials[(size_t)k * C + c];
            s2 += (double)partials[(size_t)(tiles + k) * C + c];
        }
    red[0][tl][cl] = s;
    red[1][tl][cl] = s2;
    __syncthreads();
    if (tl == 0 && c < C) {
        for (int k = 1; k < 16; ++k) {
            s += red[0][k][cl];
            s2 += red[1][k][cl];
        }
        dbeta[c] = (float)s;
        dgamma[c] = (float)s2;
    }
}

// ---- backward stage 2 ---------------------------------------------------------------------------
template <bool RELU, bool TRAIN, bool DRES>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ dout, int ldd,
                                                      const float* __restrict__ out, int ldo,
                                                      const float* __restrict__ y, int ldy, int64_t M,
                                                      int C4, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ dgamma,
                                                      const float* __restrict__ dbeta, float inv_count,
                                                      float* __restrict__ dy, int lddy,
                                                      float* __restrict__ dres, int lddres, int CQ, int RL) {
    RowThread rt = row_thread(C4, CQ, RL);
    if (!rt.active) return;
    const int c = rt.c4 * 4;
    const float4 mu = ld4(mean + c), is = ld4(invstd + c);
    float4 ga = gamma ? ld4(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    float4 k1 = make_float4(0, 0, 0, 0), k2 = make_float4(0, 0, 0, 0);
    if (TRAIN) {
        float4 db = ld4(dbeta + c), dg = ld4(dgamma + c);
        k1 = make_float4(db.x * inv_count, db.y * inv_count, db.z * inv_count, db.w * inv_count);
        k2 = make_float4(dg.x * inv_count, dg.y * inv_count, dg.z * inv_count, dg.w * inv_count);
    }
    const float4 gi = make_float4(ga.x * is.x, ga.y * is.y, ga.z * is.z, ga.w * is.w);
    for (int64_t r = rt.row0; r < M; r += rt.rstep) {
        float4 g = ld4(dout + r * ldd + c);
        if (RELU) {
            float4 o = ld4(out + r * ldo + c);
            g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f;
            g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
        }
        if (DRES) st4(dres + r * lddres + c, g);
        float4 d;
        if (TRAIN) {
            float4 v = ld4(y + r * ldy + c);
            d.x = gi.x * (g.x - k1.x - (v.x - mu.x) * is.x * k2.x);
            d.y = gi.y * (g.y - k1.y - (v.y - mu.y) * is.y * k2.y);
            d.z = gi.z * (g.z - k1.z - (v.z - mu.z) * is.z * k2.z);
            d.w = gi.w * (g.w - k1.w - (v.w - mu.w) * is.w * k2.w);
        } else {
            d = make_float4(gi.x * g.x, gi.y * g.y, gi.z * g.z, gi.w * g.w);
        }
        st4(dy + r * lddy + c, d);
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
    int64_t t = (M + 255) / 256;
    if (t > 1024) t = 1024;
    if (t < 1) t = 1;
    return (int)t;
}

extern "C" int iswm_colstat(const float* x, int64_t M, int C, int ld, float* partials, iswm_stream_t stream) {
    if (int e = chk_rows("colstat", M, C, ld)) return e;
    ISWM_REQUIRE(x && partials && aligned16(x) && aligned16(partials), "colstat: bad pointer");
    const int tiles = iswm_colstat_tiles(M);
    RowPlan p = plan_rows(M, C, tiles);
    hipLaunchKernelGGL(k_colstat, dim3(p.rowblocks, p.colblocks), dim3(256), 0, (hipStream_t)stream, x, M, p.C4,
                       ld, p.CQ, p.RL, tiles, C, partials);
    return check_launch("colstat");
}

extern "C" int iswm_bn_finalize(const float* partials, int tiles, int C, int64_t count, const float* gamma,
                                const float* beta, float* running_mean, float* running_var, float momentum,
                                float eps, float* scale, float* shift, float* save_mean, float* save_invstd,
                                iswm_stream_t stream) {
    ISWM_REQUIRE(partials && scale && shift && save_mean && save_invstd && tiles > 0 && C > 0 && count > 0,
                 "bn_finalize: bad argument");
    hipLaunchKernelGGL(k_bn_finalize, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, partials, tiles, C,
                       (double)count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift,
                       save_mean, save_invstd);
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

extern "C" int iswm_bn_apply(const float* y, int64_t M, int C, int ldy, const float* scale, const float* shift,
                             const float* residual, int ldr, int relu, float* out, int ldo,
                             iswm_stream_t stream) {
    if (int e = chk_rows("bn_apply", M, C, ldy)) return e;
    ISWM_REQUIRE(y && scale && shift && out && ldo % 4 == 0 && ldo >= C, "bn_apply: bad argument");
    ISWM_REQUIRE(!residual || (ldr % 4 == 0 && ldr >= C), "bn_apply: bad residual pitch");
    RowPlan p = plan_rows(M, C);
    dim3 grid(p.rowblocks, p.colblocks), blk(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(R, S) \
    hipLaunchKernelGGL((k_bn_apply<R, S>), grid, blk, 0, s, y, M, p.C4, ldy, scale, shift, residual, ldr, out, ldo, \
                       p.CQ, p.RL)
    if (relu && residual) LAUNCH(true, true);
    else if (relu) LAUNCH(true, false);
    else if (residual) LAUNCH(false, true);
    else LAUNCH(false, false);
#undef LAUNCH
    return check_launch("bn_apply");
}

extern "C" int iswm_bn_bwd_reduce(const float* dout, int ldd, const float* out, int ldo, const float* y, int ldy,
                                  int64_t M, int C, const float* mean, const float* invstd, int relu,
                                  float* partials, iswm_stream_t stream) {
    if (int e = chk_rows("bn_bwd_reduce", M, C, ldy)) return e;
    ISWM_REQUIRE(dout && y && mean && invstd && partials && (!relu || out), "bn_bwd_reduce: null pointer");
    ISWM_REQUIRE(ldd % 4 == 0 && ldd >= C && (!relu || (ldo % 4 == 0 && ldo >= C)), "bn_bwd_reduce: bad pitch");
    const int tiles = iswm_colstat_tiles(M);
    RowPlan p = plan_rows(M, C, tiles);
    dim3 grid(p.rowblocks, p.colblocks), blk(256);
    hipStream_t s = (hipStream_t)stream;
    if (relu)
        hipLaunchKernelGGL((k_bn_bwd_reduce<true>), grid, blk, 0, s, dout, ldd, out, ldo, y, ldy, M, p.C4, C, mean,
                           invstd, p.CQ, p.RL, tiles, partials);
    else
        hipLaunchKernelGGL((k_bn_bwd_reduce<false>), grid, blk, 0, s, dout, ldd, out, ldo, y, ldy, M, p.C4, C, mean,
                           invstd, p.CQ, p.RL, tiles, partials);
    return check_launch("bn_bwd_reduce");
}

extern "C" int iswm_bn_bwd_finalize(const float* partials, int tiles, int C, float* dgamma, float* dbeta,
                                    iswm_stream_t stream) {
    ISWM_REQUIRE(partials && dgamma && dbeta && tiles > 0 && C > 0, "bn_bwd_finalize: bad argument");
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, partials, tiles,
                       C, dgamma, dbeta);
    return check_launch("bn_bwd_finalize");
}

extern "C" int iswm_bn_bwd_apply(const float* dout, int ldd, const float* out, int ldo, const float* y, int ldy,
                                 int64_t M, int C, const float* mean, const float* invstd, const float* gamma,
                                 const float* dgamma, const float* dbeta, int relu, int training, float* dy,
                                 int lddy, float* dres, int lddres, iswm_stream_t stream) {
    if (int e = chk_rows("bn_bwd_apply", M, C, ldy)) return e;
    ISWM_REQUIRE(dout && y && mean && invstd && dy && (!relu || out) && (!training || (dgamma && dbeta)),
                 "bn_bwd_apply: null pointer");
    ISWM_REQUIRE(ldd % 4 == 0 && ldd >= C && lddy % 4 == 0 && lddy >= C && (!dres || (lddres % 4 == 0 && lddres >= C)),
                 "bn_bwd_apply: bad pitch");
    RowPlan p = plan_rows(M, C);
    dim3 grid(p.rowblocks, p.colblocks), blk(256);
    hipStream_t s = (hipStream_t)stream;
    const float inv = 1.f / (float)M;
#define LAUNCH(R, T, D)                                                                                          \
    hipLaunchKernelGGL((k_bn_bwd_apply<R, T, D>), grid, blk, 0, s, dout, ldd, out, ldo, y, ldy, M, p.C4, mean, invstd, \
                       gamma, dgamma, dbeta, inv, dy, lddy, dres, lddres, p.CQ, p.RL)
    const int key = (relu ? 4 : 0) | (training ? 2 : 0) | (dres ? 1 : 0);
    switch (key) {
        case 0: LAUNCH(false, false, false); break;
        case 1: LAUNCH(false, false, true); break;
        case 2: LAUNCH(false, true, false); break;
        case 3: LAUNCH(false, true, true); break;
        case 4: LAUNCH(true, false, false); break;
        case 5: LAUNCH(true, false, true); break;
        case 6: LAUNCH(true, true, false); break;
        default: LAUNCH(true, true, true); break;
    }
#undef LAUNCH
    return check_launch("bn_bwd_apply");
}
