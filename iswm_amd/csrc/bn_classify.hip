// The tail of the DeepLab heads -- ... Conv2d(256, 256, 3) -> BatchNorm2d -> ReLU -> Conv2d(256, num_classes, 1)
// (network/_deeplab.py:44-52 DeepLabHeadV3Plus.classifier, :84-90 DeepLabHead) -- with the 1x1 classifier folded into the
// BatchNorm passes of the stage in front of it.
//
// The classifier reads and its backward re-reads / produces 256-channel tensors at the decoder's resolution (129 x 129 x 16 images:
// 409 MB as planes, 272 MB in fp32) for 2 output channels: separate kernels moved ~2.3 GB per step around it.  Folded:
//   forward   logits[p][k] = bias[k] + sum_c relu((y[p][c] - mean[c]) * scale[c] + shift[c]) * Wc[k][c]
//             one pass over the raw conv output y; the 256-channel activation is never stored (backward recomputes it from y);
//   backward  the gradient of that activation is dact[p][c] = sum_k dlogit[p][k] * Wc[k][c]: formed on the fly from the 16-byte
//             dlogit row inside BOTH BatchNorm-backward passes (no 272-MB tensor), and the reduce pass also takes
//             dWc[k][c] = sum_p dlogit[p][k] * act[p][c] (the classifier's weight gradient) along.
// C = 256 only (a row is one wave: 64 lanes x 4 channels), up to 4 classes (Wc zero-padded to [4][C], dlogit / logits rows of
// 4 floats); anything else keeps the unfused path.  Same per-element expressions as bn.hip (k_bn_apply8, k_bn_bwd_reduce<2>,
// k_bn_bwd_apply<2>): sums in double from fp32 8-row runs, dy in double.
#include "rowmap.h"

namespace iswm {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// one wave per row: lane -> channels 4 lane .. 4 lane + 3
__global__ __launch_bounds__(256) void k_bn_apply_cls(const float* __restrict__ y, int64_t M, int ldy,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      const float* __restrict__ mean, const float* __restrict__ wc,
                                                      const float* __restrict__ bias, float* __restrict__ logits, int ldl) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = 4 * lane;
    const float4 sc = ld4(scale + c), sh = ld4(shift + c), mu = ld4(mean + c);
    const float4 w0 = ld4(wc + c), w1 = ld4(wc + 256 + c), w2 = ld4(wc + 512 + c), w3 = ld4(wc + 768 + c);
    const float4 b = bias ? ld4(bias) : make_float4(0.f, 0.f, 0.f, 0.f);
    auto act = [&](const float4 v) __attribute__((always_inline)) -> float4 {
        float4 a;
        a.x = fmaxf((v.x - mu.x) * sc.x + sh.x, 0.f); a.y = fmaxf((v.y - mu.y) * sc.y + sh.y, 0.f);
        a.z = fmaxf((v.z - mu.z) * sc.z + sh.z, 0.f); a.w = fmaxf((v.w - mu.w) * sc.w + sh.w, 0.f);
        return a;
    };
    auto dot = [&](const float4 a, const float4 w) __attribute__((always_inline)) -> float {
        return (a.x * w.x + a.y * w.y) + (a.z * w.z + a.w * w.w);
    };
    const int64_t step = (int64_t)gridDim.x * 4;
    int64_t r = (int64_t)blockIdx.x * 4 + wave;
    for (; r + step < M; r += 2 * step) {                 // two rows in flight per wave
        const float4 va = ld4(y + r * ldy + c), vb = ld4(y + (r + step) * ldy + c);
        const float4 aa = act(va), ab = act(vb);
        float4 la, lb;
        la.x = wave_sum(dot(aa, w0)); la.y = wave_sum(dot(aa, w1)); la.z = wave_sum(dot(aa, w2)); la.w = wave_sum(dot(aa, w3));
        lb.x = wave_sum(dot(ab, w0)); lb.y = wave_sum(dot(ab, w1)); lb.z = wave_sum(dot(ab, w2)); lb.w = wave_sum(dot(ab, w3));
        if (lane == 0) {
            st4(logits + r * ldl, make_float4(la.x + b.x, la.y + b.y, la.z + b.z, la.w + b.w));
            st4(logits + (r + step) * ldl, make_float4(lb.x + b.x, lb.y + b.y, lb.z + b.z, lb.w + b.w));
        }
    }
    if (r < M) {
        const float4 aa = act(ld4(y + r * ldy + c));
        float4 la;
        la.x = wave_sum(dot(aa, w0)); la.y = wave_sum(dot(aa, w1)); la.z = wave_sum(dot(aa, w2)); la.w = wave_sum(dot(aa, w3));
        if (lane == 0) st4(logits + r * ldl, make_float4(la.x + b.x, la.y + b.y, la.z + b.z, la.w + b.w));
    }
}

// partial sums per row block: bn[2][tiles][C] = (sum dz, sum dz * xhat), wpart[tiles][4][C] = sum dlogit[k] * act
__global__ __launch_bounds__(256) void k_bn_bwd_reduce_cls(const float* __restrict__ dl, int ldl, const float* __restrict__ wc,
                                                           const float* __restrict__ y, int ldy, int64_t M, int C4, int C,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ mscale, const float* __restrict__ mshift,
                                                           int CQ, int RL, int tiles, double* __restrict__ partials,
                                                           double* __restrict__ wpart) {
    __shared__ double red[6 * 256 * 4];
    RowThread rt = row_thread(C4, CQ, RL);
    double s[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, sw[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) sw[k][j] = 0.0;
    if (rt.active) {
        const int c = rt.c4 * 4;
        const float4 mu = ld4(mean + c), is = ld4(invstd + c), msc = ld4(mscale + c), msh = ld4(mshift + c);
        const float4 w0 = ld4(wc + c), w1 = ld4(wc + C + c), w2 = ld4(wc + 2 * C + c), w3 = ld4(wc + 3 * C + c);
        for (int64_t r = rt.row0; r < M; r += 8 * rt.rstep) {
            float f[4] = {0.f, 0.f, 0.f, 0.f}, f2[4] = {0.f, 0.f, 0.f, 0.f}, fw[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) fw[k][j] = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t rr = r + u * rt.rstep;
                if (rr < M) {
                    const float4 d = ld4(dl + rr * ldl);
                    const float4 v = ld4(y + rr * ldy + c);
                    const float bn[4] = {(v.x - mu.x) * msc.x + msh.x, (v.y - mu.y) * msc.y + msh.y, (v.z - mu.z) * msc.z + msh.z,
                                         (v.w - mu.w) * msc.w + msh.w};
                    const float xh[4] = {(v.x - mu.x) * is.x, (v.y - mu.y) * is.y, (v.z - mu.z) * is.z, (v.w - mu.w) * is.w};
                    const float ga[4] = {d.x * w0.x + d.y * w1.x + d.z * w2.x + d.w * w3.x, d.x * w0.y + d.y * w1.y + d.z * w2.y + d.w * w3.y,
                                         d.x * w0.z + d.y * w1.z + d.z * w2.z + d.w * w3.z, d.x * w0.w + d.y * w1.w + d.z * w2.w + d.w * w3.w};
                    const float dk[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool on = bn[j] > 0.f;
                        const float g = on ? ga[j] : 0.f, a = on ? bn[j] : 0.f;
                        f[j] += g;
                        f2[j] += g * xh[j];
#pragma unroll
                        for (int k = 0; k < 4; ++k) fw[k][j] += dk[k] * a;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[j] += (double)f[j];
                s2[j] += (double)f2[j];
#pragma unroll
                for (int k = 0; k < 4; ++k) sw[k][j] += (double)fw[k][j];
            }
        }
    }
    const int t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[t * 4 + j] = s[j];
        red[(256 + t) * 4 + j] = s2[j];
#pragma unroll
        for (int k = 0; k < 4; ++k) red[((2 + k) * 256 + t) * 4 + j] = sw[k][j];
    }
    __syncthreads();
    if (rt.active && rt.rl == 0) {
        for (int q = 1; q < RL; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[j] += red[(t + q * CQ) * 4 + j];
                s2[j] += red[(256 + t + q * CQ) * 4 + j];
#pragma unroll
                for (int k = 0; k < 4; ++k) sw[k][j] += red[((2 + k) * 256 + t + q * CQ) * 4 + j];
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            partials[(size_t)blockIdx.x * C + rt.c4 * 4 + j] = s[j];
            partials[(size_t)(tiles + blockIdx.x) * C + rt.c4 * 4 + j] = s2[j];
#pragma unroll
            for (int k = 0; k < 4; ++k) wpart[((size_t)blockIdx.x * 4 + k) * C + rt.c4 * 4 + j] = sw[k][j];
        }
    }
}

// out[i] = sum over tiles of part[tile][i] (fixed order); also dgamma / dbeta / sums of the BatchNorm from its two partial arrays
__global__ __launch_bounds__(256) void k_cls_finalize(const double* __restrict__ partials, const double* __restrict__ wpart, int tiles,
                                                      int C, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      double* __restrict__ sums, float* __restrict__ dwc) {
    // block = 4 columns x 64 tile lanes; columns 0 .. 6C-1: [sum dz | sum dz xhat | dWc rows 0..3]
    __shared__ double red[64][5];
    const int cl = threadIdx.x & 3, tl = threadIdx.x >> 2;
    const int col = blockIdx.x * 4 + cl;
    double s = 0.0;
    if (col < 6 * C) {
        const int which = col / C, c = col - which * C;
        for (int k = tl; k < tiles; k += 64) {
            if (which == 0) s += partials[(size_t)k * C + c];
            else if (which == 1) s += partials[(size_t)(tiles + k) * C + c];
            else s += wpart[((size_t)k * 4 + (which - 2)) * C + c];
        }
    }
    red[tl][cl] = s;
    __syncthreads();
    if (tl == 0 && col < 6 * C) {
        for (int k = 1; k < 64; ++k) s += red[k][cl];
        const int which = col / C, c = col - which * C;
        if (which == 0) {
            dbeta[c] = (float)s;
            sums[c] = s;
        } else if (which == 1) {
            dgamma[c] = (float)s;
            sums[C + c] = s;
        } else {
            dwc[(size_t)(which - 2) * C + c] = (float)s;
        }
    }
}

template <bool TRAIN>
__global__ __launch_bounds__(256) void k_bn_bwd_apply_cls(const float* __restrict__ dl, int ldl, const float* __restrict__ wc,
                                                          const float* __restrict__ y, int ldy, int64_t M, int C4, int C,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ mscale,
                                                          const float* __restrict__ mshift, const double* __restrict__ sums,
                                                          double inv_count, void* __restrict__ dy, int lddy, int64_t dyps, int CQ,
                                                          int RL) {
    RowThread rt = row_thread(C4, CQ, RL);
    if (!rt.active) return;
    const int c = rt.c4 * 4;
    const float4 mu = ld4(mean + c), is = ld4(invstd + c), msc = ld4(mscale + c), msh = ld4(mshift + c);
    const float4 ga = gamma ? ld4(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 w0 = ld4(wc + c), w1 = ld4(wc + C + c), w2 = ld4(wc + 2 * C + c), w3 = ld4(wc + 3 * C + c);
    const double mud[4] = {mu.x, mu.y, mu.z, mu.w}, isd[4] = {is.x, is.y, is.z, is.w};
    const double gi[4] = {(double)ga.x * is.x, (double)ga.y * is.y, (double)ga.z * is.z, (double)ga.w * is.w};
    double k1[4] = {0, 0, 0, 0}, k2[4] = {0, 0, 0, 0};
    if (TRAIN) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            k1[k] = sums[c + k] * inv_count;
            k2[k] = sums[C + c + k] * inv_count;
        }
    }
    for (int64_t r = rt.row0; r < M; r += rt.rstep) {
        const float4 d = ld4(dl + r * ldl);
        const float4 v = ld4(y + r * ldy + c);
        float g[4] = {d.x * w0.x + d.y * w1.x + d.z * w2.x + d.w * w3.x, d.x * w0.y + d.y * w1.y + d.z * w2.y + d.w * w3.y,
                      d.x * w0.z + d.y * w1.z + d.z * w2.z + d.w * w3.z, d.x * w0.w + d.y * w1.w + d.z * w2.w + d.w * w3.w};
        g[0] = (v.x - mu.x) * msc.x + msh.x > 0.f ? g[0] : 0.f;
        g[1] = (v.y - mu.y) * msc.y + msh.y > 0.f ? g[1] : 0.f;
        g[2] = (v.z - mu.z) * msc.z + msh.z > 0.f ? g[2] : 0.f;
        g[3] = (v.w - mu.w) * msc.w + msh.w > 0.f ? g[3] : 0.f;
        const double vd[4] = {v.x, v.y, v.z, v.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = TRAIN ? (float)(gi[k] * ((double)g[k] - k1[k] - (vd[k] - mud[k]) * isd[k] * k2[k])) : (float)(gi[k] * (double)g[k]);
        st4x(dy, r * lddy + c, dyps, make_float4(o[0], o[1], o[2], o[3]));
    }
}

}  // namespace iswm

using namespace iswm;

extern "C" int iswm_colstat_tiles(int64_t M);

static bool cls_shape_ok(int64_t M, int C, int ldy, int ldl) {
    return M > 0 && C == 256 && ldy % 4 == 0 && ldy >= C && ldl == 4;
}

/* logits[p][0..3] = bias4 + Wc4 . relu(bn(y[p]));  Wc4 = the classifier's weight zero-padded to [4][C], bias4 to [4] (or null) */
extern "C" int iswm_bn_apply_classify(const float* y, int64_t M, int C, int ldy, const float* scale, const float* shift,
                                      const float* mean, const float* wc4, const float* bias4, float* logits, int ldl,
                                      iswm_stream_t stream) {
    ISWM_REQUIRE(y && scale && shift && mean && wc4 && logits, "bn_apply_classify: null pointer");
    ISWM_REQUIRE(cls_shape_ok(M, C, ldy, ldl), "bn_apply_classify: C must be 256 and logits rows 4 floats (C %d ldy %d ldl %d)", C, ldy, ldl);
    ISWM_REQUIRE(aligned16(y) && aligned16(wc4) && aligned16(logits) && (!bias4 || aligned16(bias4)), "bn_apply_classify: alignment");
    int64_t blocks = (M + 7) / 8;                 // 4 waves per block, >= 2 rows per wave
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_bn_apply_cls, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, M, ldy, scale, shift, mean,
                       wc4, bias4, logits, ldl);
    return check_launch("bn_apply_classify");
}

extern "C" size_t iswm_bn_classify_bwd_workspace(int64_t M, int C) {
    // double partials[2][tiles][C] + sums[2][C] + wpart[tiles][4][C]
    return ((size_t)6 * iswm_colstat_tiles(M) * C + (size_t)2 * C) * sizeof(double);
}

/* BatchNorm (+ReLU) backward of the stage in front of the classifier, fed by dlogit [M][4] and Wc4 [4][C] instead of a 256-channel
 * gradient tensor; also returns the classifier's weight gradient dwc4 [4][C].  mask_scale / mask_shift: the forward's scale / shift. */
extern "C" int iswm_bn_backward_classify(const float* dlogit, int ldl, const float* wc4, const float* y, int ldy, int64_t M, int C,
                                         const float* mean, const float* invstd, const float* gamma, const float* mask_scale,
                                         const float* mask_shift, int training, float* dgamma, float* dbeta, float* dwc4,
                                         void* dy, int lddy, int64_t dy_ps, void* workspace, size_t workspace_bytes,
                                         iswm_stream_t stream) {
    ISWM_REQUIRE(dlogit && wc4 && y && mean && invstd && mask_scale && mask_shift && dgamma && dbeta && dwc4 && dy && workspace,
                 "bn_backward_classify: null pointer");
    ISWM_REQUIRE(cls_shape_ok(M, C, ldy, ldl) && lddy % 4 == 0 && lddy >= C, "bn_backward_classify: bad shape");
    ISWM_REQUIRE(dy_ps == 0 || dy_ps == -1 || (dy_ps >= M * lddy && dy_ps % 4 == 0), "bn_backward_classify: bad plane stride");
    ISWM_REQUIRE(workspace_bytes >= iswm_bn_classify_bwd_workspace(M, C) && aligned16(workspace) && aligned16(dlogit) && aligned16(wc4),
                 "bn_backward_classify: workspace too small or misaligned");
    const int tiles = iswm_colstat_tiles(M);
    double* partials = (double*)workspace;
    double* sums = partials + (size_t)2 * tiles * C;
    double* wpart = sums + (size_t)2 * C;
    hipStream_t s = (hipStream_t)stream;
    {
        RowPlan p = plan_rows(M, C, tiles);
        dim3 grid(p.rowblocks, p.colblocks), blk(256);
        hipLaunchKernelGGL(k_bn_bwd_reduce_cls, grid, blk, 0, s, dlogit, ldl, wc4, y, ldy, M, p.C4, C, mean, invstd, mask_scale,
                           mask_shift, p.CQ, p.RL, tiles, partials, wpart);
        if (int e = check_launch("bn_bwd_reduce_cls")) return e;
    }
    hipLaunchKernelGGL(k_cls_finalize, dim3((6 * C + 3) / 4), dim3(256), 0, s, partials, wpart, tiles, C, dgamma, dbeta, sums, dwc4);
    if (int e = check_launch("cls_finalize")) return e;
    RowPlan p = plan_rows(M, C);
    dim3 grid(p.rowblocks, p.colblocks), blk(256);
    const double inv = 1.0 / (double)M;
    if (training)
        hipLaunchKernelGGL((k_bn_bwd_apply_cls<true>), grid, blk, 0, s, dlogit, ldl, wc4, y, ldy, M, p.C4, C, mean, invstd, gamma,
                           mask_scale, mask_shift, sums, inv, dy, lddy, dy_ps, p.CQ, p.RL);
    else
        hipLaunchKernelGGL((k_bn_bwd_apply_cls<false>), grid, blk, 0, s, dlogit, ldl, wc4, y, ldy, M, p.C4, C, mean, invstd, gamma,
                           mask_scale, mask_shift, sums, inv, dy, lddy, dy_ps, p.CQ, p.RL);
    return check_launch("bn_bwd_apply_cls");
}
