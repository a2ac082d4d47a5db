// Implicit-GEMM 2-D convolution for gfx950 on the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32: forward, data gradient and weight gradient.
//
// Replaces the nn.Conv2d calls of the reference hot path (network/backbone/resnet.py:27-35,
// 144,184-187; network/_deeplab.py:37,44-51,124,134,149,162) and their autograd backward.
//
// Design:
//   * activations NHWC, weights OHWI: the GEMM K axis (tap, channel) is contiguous in
//     memory for both operands of the forward pass, so tiles are staged with 16-byte
//     loads and no im2col buffer ever exists;
//   * one workgroup = 4 waves (2x2), block tile 128 x {128,64} x 32, each wave owns a
//     64 x {64,32} sub-tile as 32x32 MFMA accumulators; two workgroups per CU;
//   * global -> register -> LDS staging, double-buffered in LDS, one barrier per K chunk:
//     the loads of chunk k+1 are issued before the MFMAs of chunk k and written to the
//     other LDS buffer after them;
//   * an operand whose K axis is contiguous in memory ("KC": activations / OHWI weights
//     in fwd, dy in dgrad) is kept [row][k] in LDS with a 36-float row pitch and read as
//     ds_read_b128 (conflict-free: 16 rows x 144 B hit 16 distinct 16-byte slots); one
//     b128 read feeds 4 MFMAs (lane half h supplies k = 4h+j for MFMA j);
//   * an operand whose ROW axis is contiguous ("RC": weights in dgrad, dy and x in
//     wgrad) is kept [k][row] and read with ds_read_b32 (32 consecutive floats per half);
//   * padding taps are zero-filled at load time; dilation is just a tap offset;
//   * the forward epilogue optionally emits per-tile per-channel sum / sum-of-squares for
//     the training-mode BatchNorm that follows every conv (deterministic two-stage stats);
//   * wgrad flattens (tap, cin) into the GEMM N axis and splits the pixel (K) axis across
//     workgroups into slabs that a second kernel sums in a fixed order (bit-reproducible).
#include <stdlib.h>
#include <string.h>

#include "conv_common.h"

namespace iswm {

// ------------------------------------------------------------------------------------------
// forward:  y[m, co] = sum_k A[m, k] * W[co, k],  m = (n, oh, ow),  k = (kh, kw, ci)
// ------------------------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(256, 2) void k_conv_fwd(const ConvArgs a) {
    constexpr int BM = 128;
    constexpr int NB = BN / 64;    // 32-wide MFMA column tiles per wave
    constexpr int BROWS = BN / 32; // weight rows staged per thread
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * KC_PITCH];
    float* As = smem;
    float* Bs = smem + 2 * BM * KC_PITCH;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = L / a.NT, nt = L - mt * a.NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const int q = t & 7, r0 = t >> 3;

    const int HoWo = a.Ho * a.Wo;
    int ihb[4], iwb[4], pb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int m = m0 + r0 + 32 * j;
        if (m < a.M) {
            int n = m / HoWo, rem = m - n * HoWo;
            int oh = rem / a.Wo, ow = rem - oh * a.Wo;
            ihb[j] = oh * a.stride - a.pad;
            iwb[j] = ow * a.stride - a.pad;
            pb[j] = n * a.H * a.W;
        } else {
            ihb[j] = -(1 << 28);
            iwb[j] = 0;
            pb[j] = 0;
        }
    }
    const int Cin4 = a.Cin >> 2, K4 = a.Ktot >> 2;
    const float* wrow[BROWS];
    bool wok[BROWS];
#pragma unroll
    for (int j = 0; j < BROWS; ++j) {
        int n = n0 + r0 + 32 * j;
        wok[j] = n < a.Cout;
        wrow[j] = a.w + (size_t)(wok[j] ? n : 0) * a.Ktot;
    }

    float4 ra[4], rb[BROWS];
    auto gload = [&](int kc) {
        const int k4 = kc * 8 + q;
        const bool kv = k4 < K4;
        const int tap = k4 / Cin4, c4 = k4 - tap * Cin4;
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        const int dh = kh * a.dil, dw = kw * a.dil;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int ih = ihb[j] + dh, iw = iwb[j] + dw;
            bool ok = kv && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            ra[j] = ok ? ldg4(a.x + (size_t)(pb[j] + ih * a.W + iw) * a.ldx + c4 * 4)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < BROWS; ++j)
            rb[j] = (kv && wok[j]) ? ldg4(wrow[j] + k4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(&As[(buf * BM + r0 + 32 * j) * KC_PITCH + q * 4]) = ra[j];
#pragma unroll
        for (int j = 0; j < BROWS; ++j)
            *reinterpret_cast<float4*>(&Bs[(buf * BN + r0 + 32 * j) * KC_PITCH + q * 4]) = rb[j];
    };

    f32x16 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nK = (a.Ktot + 31) >> 5;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kc = 0; kc < nK; ++kc) {
        const int cur = kc & 1;
        const bool more = kc + 1 < nK;
        if (more) gload(kc + 1);
        const float* Ab = &As[(cur * BM + wm * 64 + li) * KC_PITCH + lh * 4];
        const float* Bb = &Bs[(cur * BN + wn * (BN / 2) + li) * KC_PITCH + lh * 4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float af[2][4], bf[NB][4];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                *reinterpret_cast<float4*>(af[mb]) =
                    *reinterpret_cast<const float4*>(Ab + mb * 32 * KC_PITCH + g * 8);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                *reinterpret_cast<float4*>(bf[nb]) =
                    *reinterpret_cast<const float4*>(Bb + nb * 32 * KC_PITCH + g * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = mfma32(af[mb][j], bf[nb][j], acc[mb][nb]);
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = n0 + wn * (BN / 2) + nb * 32 + li;
        const bool cok = col < a.Cout;
        const float bv = (a.bias != nullptr && cok) ? a.bias[col] : 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (cok && row < a.M) a.y[(size_t)row * a.ldy + col] = acc[mb][nb][r] + bv;
            }
        }
    }
    if (a.stats != nullptr) {
        // Per-tile BatchNorm statistics, numerically centred: column sum S_t first, then the sum of
        // squared deviations from the TILE mean (M2_t).  iswm_bn_finalize merges tiles with the
        // pairwise (Chan) update in double, so the batch variance never suffers the E[x^2]-mean^2
        // cancellation -- this is what keeps 100 stacked train-mode BN layers within 1e-3 of the CPU.
        float* red = smem;  // [4][BN]: sum(wm=0), sum(wm=1), M2(wm=0), M2(wm=1)
        const int cnt = min(BM, a.M - m0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float s = 0.f;   // rows past M were staged as zeros, so they add nothing
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[mb][nb][r];
            s += __shfl_xor(s, 32);
            if (lh == 0) red[wm * BN + wn * (BN / 2) + nb * 32 + li] = s;
        }
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int c = wn * (BN / 2) + nb * 32 + li;
            const float mean = (red[c] + red[BN + c]) / (float)cnt;
            float q = 0.f;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = m0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float dv = acc[mb][nb][r] - mean;
                    q += row < a.M ? dv * dv : 0.f;
                }
            q += __shfl_xor(q, 32);
            if (lh == 0) red[(2 + wm) * BN + c] = q;
        }
        __syncthreads();
        if (t < BN && n0 + t < a.Cout) {
            a.stats[(size_t)mt * a.Cout + n0 + t] = red[t] + red[BN + t];
            a.stats[(size_t)(a.MT + mt) * a.Cout + n0 + t] = red[2 * BN + t] + red[3 * BN + t];
        }
    }
}

// ------------------------------------------------------------------------------------------
// dgrad:  dx[m, ci] = sum_k dyG[m, k] * W[k, ci],  m = (n, ih, iw),  k = (kh, kw, co)
// A: gather from dy (K-contiguous).  B: OHWI weights read with ci contiguous (row-contiguous).
// ------------------------------------------------------------------------------------------
template <int BN>
__global__ __launch_bounds__(256, 2) void k_conv_dgrad(const ConvArgs a) {
    constexpr int BM = 128;
    constexpr int NB = BN / 64;
    constexpr int BQ = BN / 4;         // float4 columns of the B tile
    constexpr int BKR = 256 / BQ;      // k rows covered per pass
    constexpr int BPASS = 32 / BKR;    // passes per thread
    __shared__ __attribute__((aligned(16))) float smem[2 * BM * KC_PITCH + 2 * 32 * BN];
    float* As = smem;
    float* Bs = smem + 2 * BM * KC_PITCH;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = L / a.NT, nt = L - mt * a.NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const int q = t & 7, r0 = t >> 3;
    const int bq = t % BQ, bk0 = t / BQ;

    // rows are INPUT pixels here: a.H/a.W input dims, a.Ho/a.Wo the dims of dy
    const int HW = a.H * a.W;
    int thb[4], twb[4], pb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int m = m0 + r0 + 32 * j;
        if (m < a.M) {
            int n = m / HW, rem = m - n * HW;
            int ih = rem / a.W, iw = rem - ih * a.W;
            thb[j] = ih + a.pad;
            twb[j] = iw + a.pad;
            pb[j] = n * a.Ho * a.Wo;
        } else {
            thb[j] = -(1 << 28);
            twb[j] = 0;
            pb[j] = 0;
        }
    }
    const int Co4 = a.Cout >> 2, K4 = a.Ktot >> 2;
    const int taps = a.KH * a.KW;
    const bool nok = n0 + bq * 4 < a.Cin;

    float4 ra[4], rb[BPASS];
    auto gload = [&](int kc) {
        const int k4 = kc * 8 + q;
        const bool kv = k4 < K4;
        const int tap = k4 / Co4, c4 = k4 - tap * Co4;
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        const int dh = kh * a.dil, dw = kw * a.dil;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int th = thb[j] - dh, tw = twb[j] - dw;
            int oh = th, ow = tw;
            bool ok = kv && th >= 0 && tw >= 0;
            if (a.stride != 1) {
                oh = th / a.stride;
                ow = tw / a.stride;
                ok = ok && (oh * a.stride == th) && (ow * a.stride == tw);
            }
            ok = ok && oh < a.Ho && ow < a.Wo;
            ra[j] = ok ? ldg4(a.x + (size_t)(pb[j] + oh * a.Wo + ow) * a.ldx + c4 * 4)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            int k = kc * 32 + bk0 + BKR * j;
            bool ok = nok && k < a.Ktot;
            int tp = k / a.Cout, co = k - tp * a.Cout;
            rb[j] = ok ? ldg4(a.w + ((size_t)co * taps + tp) * a.Cin + n0 + bq * 4)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(&As[(buf * BM + r0 + 32 * j) * KC_PITCH + q * 4]) = ra[j];
#pragma unroll
        for (int j = 0; j < BPASS; ++j)
            *reinterpret_cast<float4*>(&Bs[(buf * 32 + bk0 + BKR * j) * BN + bq * 4]) = rb[j];
    };

    f32x16 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nK = (a.Ktot + 31) >> 5;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kc = 0; kc < nK; ++kc) {
        const int cur = kc & 1;
        const bool more = kc + 1 < nK;
        if (more) gload(kc + 1);
        const float* Ab = &As[(cur * BM + wm * 64 + li) * KC_PITCH + lh * 4];
        const float* Bb = &Bs[(cur * 32 + lh * 4) * BN + wn * (BN / 2) + li];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float af[2][4], bf[NB][4];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                *reinterpret_cast<float4*>(af[mb]) =
                    *reinterpret_cast<const float4*>(Ab + mb * 32 * KC_PITCH + g * 8);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[nb][j] = Bb[(g * 8 + j) * BN + nb * 32];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = mfma32(af[mb][j], bf[nb][j], acc[mb][nb]);
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = n0 + wn * (BN / 2) + nb * 32 + li;
        const bool cok = col < a.Cin;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (cok && row < a.M) {
                    float* o = &a.y[(size_t)row * a.ldy + col];
                    *o = a.accumulate ? *o + acc[mb][nb][r] : acc[mb][nb][r];
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// wgrad:  dw[co, j] = sum_p dy[p, co] * xG[p, j],  j = (kh, kw, ci) flattened, p = (n, oh, ow)
// Both operands row-contiguous; the pixel axis is split across blockIdx.y into slabs.
// ------------------------------------------------------------------------------------------
static __device__ __attribute__((aligned(16))) float g_zero_row_w[64];   // target of out-of-bounds rows

// MODE 0: any geometry.  MODE 1: stride 1 and Ho == H, Wo == W ("same" convs: every 3x3 of the net
// except the two strided ones) -- the gathered pixel of output pixel p is p + dh*W + dw, so addresses
// advance by a constant and only the bounds test needs (oh, ow).  MODE 2: 1x1 stride 1 -- no bounds.
// X6: bf16x6 arithmetic (conv_mfma_x6.hip).  Both operands arrive with the GEMM K axis (pixels) STRIDED in
// memory, so the three bf16 planes are kept [k][row] in LDS (natural 8-byte writes) and the k-contiguous MFMA
// fragments are produced by gfx950's transposing LDS read ds_read_b64_tr_b16 (4 k x 16 rows per 16 lanes);
// row pitch 2*rows + 64 B puts the 4 k-rows of one read on disjoint bank quarters.
// NP (X6 only): bf16 planes per operand -- 3 = exact split / six MFMAs (bf16x6), 1 = rounded operands / one MFMA (bf16)
template <int BM, int BN, int MODE, bool X6, int NP = 3>
__global__ __launch_bounds__(256, 2) void k_conv_wgrad(const ConvArgs a) {
    constexpr int MB = BM / 64, NB = BN / 64;
    constexpr int AQ = BM / 4, AKR = 256 / AQ, APASS = 32 / AKR;
    constexpr int BQ = BN / 4, BKR = 256 / BQ, BPASS = 32 / BKR;
    // X6 row pitches (bytes).  128-column planes are packed (256 B rows, 48 KB per workgroup -> 3 per CU) and
    // the column offset is XORed with 64*(k&3), which lands the 4 k-rows of one transposing read on disjoint
    // bank quarters exactly as the +64 B padding does for the 64-column planes.
    constexpr int XPA = BM == 128 ? 256 : BM * 2 + 64, XPB = BN == 128 ? 256 : BN * 2 + 64;
    constexpr int SWA = BM == 128 ? 64 : 0, SWB = BN == 128 ? 64 : 0;   // swizzle step (bytes)
    constexpr int XPLA = 32 * XPA, XPLB = 32 * XPB;                  // X6 plane sizes (bytes)
    constexpr int SMEM_BYTES = X6 ? NP * (XPLA + XPLB) : 2 * 32 * (BM + BN) * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];
    float* smem = reinterpret_cast<float*>(smem_raw);
    float* As = smem;
    float* Bs = smem + 2 * 32 * BM;
    unsigned char* Ax = smem_raw;               // X6: [3][32][XPA]
    unsigned char* Bx = smem_raw + NP * XPLA;   // X6: [NP][32][XPB]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = L / a.NT, nt = L - mt * a.NT;
    const int m0 = mt * BM, n0 = nt * BN;
    const int split = blockIdx.y;
    const int P = a.M;  // pixels of dy
    const int p_begin = split * a.psplit;
    const int p_end = min(P, p_begin + a.psplit);

    const int aq = t % AQ, ak0 = t / AQ;
    const int bq = t % BQ, bk0 = t / BQ;
    const bool aok = m0 + aq * 4 < a.Cout;

    // this thread's B column: one (tap, ci4) for the whole K loop
    const int Cin4 = a.Cin >> 2;
    const int n4 = (n0 >> 2) + bq;
    const bool bok = n4 < (a.Ktot >> 2);
    const int tap = n4 / Cin4, c4 = n4 - tap * Cin4;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    const int dh = kh * a.dil - a.pad, dw = kw * a.dil - a.pad;

    // pixel coordinates (n, oh, ow) of this thread's B rows for the CURRENT chunk; advanced by 32 pixels
    // per chunk with a branch-free carry (falls back to division on tiny maps)
    int bn_[BPASS], boh[BPASS], bow[BPASS];
    const int HoWo = a.Ho * a.Wo;
    const int d_oh = 32 / a.Wo, d_ow = 32 - d_oh * a.Wo;
    const bool fast_adv = d_oh + 1 <= a.Ho;
    auto decode = [&](int kc) {
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            int p = p_begin + kc * 32 + bk0 + BKR * j;
            int n = p / HoWo, rem = p - n * HoWo;
            bn_[j] = n;
            boh[j] = rem / a.Wo;
            bow[j] = rem - boh[j] * a.Wo;
        }
    };
    auto advance = [&](int kc) {
        if (MODE == 2) return;
        if (!fast_adv) {
            decode(kc);
            return;
        }
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            int ow = bow[j] + d_ow;
            int c1 = ow >= a.Wo ? 1 : 0;
            bow[j] = ow - (c1 ? a.Wo : 0);
            int oh = boh[j] + d_oh + c1;
            int c2 = oh >= a.Ho ? 1 : 0;
            boh[j] = oh - (c2 ? a.Ho : 0);
            bn_[j] += c2;
        }
    };
    decode(0);

    // Row culling: when every column of this tile belongs to ONE filter tap (kh fixed), output rows whose
    // input row oh*stride + kh*dil - pad is outside the image contribute nothing -- whole 32-pixel chunks
    // inside such rows (the ASPP rates on a 33x33 map) are skipped.
    const int n4_last = (min(n0 + BN, a.Ktot) >> 2) - 1;
    const int tap_last = n4_last / Cin4;
    const int tap_first = (n0 >> 2) / Cin4;
    const int kh_u = tap_first / a.KW;
    const int off_u = kh_u * a.dil - a.pad;
    // valid oh: 0 <= oh*stride + off_u <= H-1
    const int oh_lo = off_u >= 0 ? 0 : (-off_u + a.stride - 1) / a.stride;
    const int oh_hi = (a.H - 1 - off_u) >= 0 ? min(a.Ho - 1, (a.H - 1 - off_u) / a.stride) : -1;
    const bool cull = (tap_first / a.KW == tap_last / a.KW) && (oh_lo > 0 || oh_hi < a.Ho - 1) && HoWo >= 64;
    auto skip = [&](int kc) -> bool {   // block-uniform
        if (!cull) return false;
        const int p0 = p_begin + kc * 32, p1 = min(p0 + 31, p_end - 1);
        const int n0_ = p0 / HoWo, oh0 = (p0 - n0_ * HoWo) / a.Wo;
        const int n1_ = p1 / HoWo, oh1 = (p1 - n1_ * HoWo) / a.Wo;
        if (n0_ == n1_) return oh1 < oh_lo || oh0 > oh_hi;          // rows oh0..oh1 of one image
        if (n1_ == n0_ + 1) return oh0 > oh_hi && oh1 < oh_lo;      // tail of one image + head of the next
        return false;
    };

    float4 ra[APASS], rb[BPASS];
    // All loads are unconditional: rows past the end of the split / outside the image read the zero row
    // (B side), and the A side then only needs a valid address (0 x finite = 0), so it clamps its pixel.
    const float* abase = aok ? a.y + m0 + aq * 4 : g_zero_row_w;
    const int a_ld = aok ? a.ldy : 0;
    const int shift = dh * a.W + dw;     // MODE 1/2: input pixel = output pixel + shift
    auto gload = [&](int kc) {
#pragma unroll
        for (int j = 0; j < APASS; ++j) {
            int p = min(p_begin + kc * 32 + ak0 + AKR * j, P - 1);
            ra[j] = ldg4(abase + (size_t)p * a_ld);
        }
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            int p = p_begin + kc * 32 + bk0 + BKR * j;
            bool ok = bok && p < p_end;
            const float* src;
            if (MODE == 2) {
                src = a.x + (size_t)p * a.ldx + c4 * 4;
            } else if (MODE == 1) {
                int ih = boh[j] + dh, iw = bow[j] + dw;
                ok = ok && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
                src = a.x + (size_t)(p + shift) * a.ldx + c4 * 4;
            } else {
                int ih = boh[j] * a.stride + dh, iw = bow[j] * a.stride + dw;
                ok = ok && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
                src = a.x + (size_t)((bn_[j] * a.H + ih) * a.W + iw) * a.ldx + c4 * 4;
            }
            rb[j] = ldg4(ok ? src : g_zero_row_w);
        }
    };
    auto lstore = [&](int buf) {
        if constexpr (X6) {
#pragma unroll
            for (int j = 0; j < APASS; ++j) {
                const int kr = ak0 + AKR * j;
                unsigned char* p = Ax + kr * XPA + ((aq * 8) ^ ((kr & 3) * SWA));
                if constexpr (NP == 1) {
                    *reinterpret_cast<uint2*>(p) = round_bf16x4(ra[j]);
                } else {
                    uint2 h, m, l;
                    split3(ra[j], h, m, l);
                    *reinterpret_cast<uint2*>(p) = h;
                    *reinterpret_cast<uint2*>(p + XPLA) = m;
                    *reinterpret_cast<uint2*>(p + 2 * XPLA) = l;
                }
            }
#pragma unroll
            for (int j = 0; j < BPASS; ++j) {
                const int kr = bk0 + BKR * j;
                unsigned char* p = Bx + kr * XPB + ((bq * 8) ^ ((kr & 3) * SWB));
                if constexpr (NP == 1) {
                    *reinterpret_cast<uint2*>(p) = round_bf16x4(rb[j]);
                } else {
                    uint2 h, m, l;
                    split3(rb[j], h, m, l);
                    *reinterpret_cast<uint2*>(p) = h;
                    *reinterpret_cast<uint2*>(p + XPLB) = m;
                    *reinterpret_cast<uint2*>(p + 2 * XPLB) = l;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < APASS; ++j)
                *reinterpret_cast<float4*>(&As[(buf * 32 + ak0 + AKR * j) * BM + aq * 4]) = ra[j];
#pragma unroll
            for (int j = 0; j < BPASS; ++j)
                *reinterpret_cast<float4*>(&Bs[(buf * 32 + bk0 + BKR * j) * BN + bq * 4]) = rb[j];
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nK = (p_end - p_begin + 31) >> 5;
    int kc = 0;
    bool first = true;
    auto next = [&]() -> bool {   // move to the next chunk that has work (coordinates follow kc)
        for (;;) {
            if (!first) {
                ++kc;
                if (kc < nK) advance(kc);
            }
            first = false;
            if (kc >= nK) return false;
            if (!skip(kc)) return true;
        }
    };
    if constexpr (X6) {
        // transposing-read lane roles: 16-lane group g = lane>>4 covers rows 16*(g&1).. of the 32-row MFMA tile
        // for k half h = g>>1; lane 4q+p of the group addresses k-row q, columns 4p..4p+3
        const int tg = lane >> 4, ti = lane & 15, tq = ti >> 2, tp = ti & 3;
        const int th = tg >> 1, tc = (tg & 1) * 16 + tp * 4;
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
        auto tr_frag = [&](const unsigned char* plane, int pitch, int sw, int col0, int ks) -> uint4 {
            const unsigned char* p = plane + (ks * 16 + th * 8 + tq) * pitch + (((col0 + tc) * 2) ^ (tq * sw));
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 4 * pitch));
            uint2 a2 = __builtin_bit_cast(uint2, lo), b2 = __builtin_bit_cast(uint2, hi);
            return make_uint4(a2.x, a2.y, b2.x, b2.y);
        };
        bool more = next();
        if (more) gload(kc);
        while (more) {
            lstore(0);
            __syncthreads();
            const bool more2 = next();
            if (more2) gload(kc);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 ah[MB], am[MB], al[MB], bh[NB], bm[NB], bl[NB];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int c0 = wm * (BM / 2) + mb * 32;
                    ah[mb] = tr_frag(Ax, XPA, SWA, c0, ks);
                    if constexpr (NP == 3) {
                        am[mb] = tr_frag(Ax + XPLA, XPA, SWA, c0, ks);
                        al[mb] = tr_frag(Ax + 2 * XPLA, XPA, SWA, c0, ks);
                    }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int c0 = wn * (BN / 2) + nb * 32;
                    bh[nb] = tr_frag(Bx, XPB, SWB, c0, ks);
                    if constexpr (NP == 3) {
                        bm[nb] = tr_frag(Bx + XPLB, XPB, SWB, c0, ks);
                        bl[nb] = tr_frag(Bx + 2 * XPLB, XPB, SWB, c0, ks);
                    }
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        f32x16 c = acc[mb][nb];
                        if constexpr (NP == 3) {
                            c = mfma_bf16(al[mb], bh[nb], c);
                            c = mfma_bf16(ah[mb], bl[nb], c);
                            c = mfma_bf16(am[mb], bm[nb], c);
                            c = mfma_bf16(am[mb], bh[nb], c);
                            c = mfma_bf16(ah[mb], bm[nb], c);
                        }
                        c = mfma_bf16(ah[mb], bh[nb], c);
                        acc[mb][nb] = c;
                    }
            }
            __syncthreads();
            more = more2;
        }
    } else {
        bool more = next();
        if (more) {
            gload(kc);
            lstore(0);
        }
        __syncthreads();
        int cur = 0;
        while (more) {
            const bool more2 = next();
            if (more2) gload(kc);
            const float* Ab = &As[(cur * 32 + lh * 4) * BM + wm * (BM / 2) + li];
            const float* Bb = &Bs[(cur * 32 + lh * 4) * BN + wn * (BN / 2) + li];
    #pragma unroll
            for (int g = 0; g < 4; ++g) {
                float af[MB][4], bf[NB][4];
    #pragma unroll
                for (int mb = 0; mb < MB; ++mb)
    #pragma unroll
                    for (int j = 0; j < 4; ++j) af[mb][j] = Ab[(g * 8 + j) * BM + mb * 32];
    #pragma unroll
                for (int nb = 0; nb < NB; ++nb)
    #pragma unroll
                    for (int j = 0; j < 4; ++j) bf[nb][j] = Bb[(g * 8 + j) * BN + nb * 32];
    #pragma unroll
                for (int j = 0; j < 4; ++j)
    #pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
    #pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[mb][nb] = mfma32(af[mb][j], bf[nb][j], acc[mb][nb]);
            }
            if (more2) lstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
            more = more2;
        }
    }
    float* out = a.stats + (size_t)split * a.Cout * a.Ktot;  // slab (or dw itself when nsplit == 1)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = n0 + wn * (BN / 2) + nb * 32 + li;
        const bool cok = col < a.Ktot;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm * (BM / 2) + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (cok && row < a.Cout) out[(size_t)row * a.Ktot + col] = acc[mb][nb][r];
            }
    }
}

__global__ void k_reduce_slabs(const float* __restrict__ slabs, float* __restrict__ dst, int64_t n4,
                               int nsplit) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        // a small weight (a 1x1 on a large map: 4 096 float4, 64-128 splits) leaves each thread a long chain of dependent
        // round trips: eight loads in flight per thread, summed in the fixed order 1, 2, 3, ...
        const float4* p = reinterpret_cast<const float4*>(slabs) + i;
        float4 s = p[0];
        int k = 1;
        for (; k + 7 < nsplit; k += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(k + u) * n4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s.x += v[u].x;
                s.y += v[u].y;
                s.z += v[u].z;
                s.w += v[u].w;
            }
        }
        for (; k < nsplit; ++k) {
            const float4 v = p[(int64_t)k * n4];
            s.x += v.x;
            s.y += v.y;
            s.z += v.z;
            s.w += v.w;
        }
        reinterpret_cast<float4*>(dst)[i] = s;
    }
}

void launch_reduce_slabs(const float* slabs, float* dst, int64_t n4, int nsplit, hipStream_t s) {
    hipLaunchKernelGGL(k_reduce_slabs, dim3(stream_grid(n4, 256)), dim3(256), 0, s, slabs, dst, n4, nsplit);
}

static int validate(const iswm_conv_desc* d) {
    ISWM_REQUIRE(d != nullptr, "conv: null descriptor");
    ISWM_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "conv: empty tensor");
    ISWM_REQUIRE(d->Cin % 4 == 0 && d->Cout % 4 == 0, "conv: Cin (%d) and Cout (%d) must be multiples of 4",
                 d->Cin, d->Cout);
    ISWM_REQUIRE(d->ldx % 4 == 0 && d->ldy % 4 == 0 && d->ldx >= d->Cin && d->ldy >= d->Cout,
                 "conv: bad pixel pitch ldx=%d ldy=%d", d->ldx, d->ldy);
    ISWM_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0, "conv: bad geometry");
    int ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
    int wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
    ISWM_REQUIRE(ho == d->Ho && wo == d->Wo, "conv: output size %dx%d does not match geometry (%dx%d)", d->Ho,
                 d->Wo, ho, wo);
    ISWM_REQUIRE((int64_t)d->N * d->H * d->W * d->ldx < (1ll << 31) &&
                     (int64_t)d->N * d->Ho * d->Wo * d->ldy < (1ll << 31),
                 "conv: tensor exceeds 2^31 elements");
    return 0;
}

// Tile-width choice for the forward / dgrad kernels.  A CU works through ceil(tiles/256) tiles
// (co-resident workgroups share its SIMDs), so a 128x128 grid of 274 tiles (the 33x33 stages with 256
// output channels) costs two full tile times where 128x64 tiles cost three half tile times.
// Narrow tiles re-read the activation panel once more and pay ~10 % in MFMA:staging ratio.
static bool use_narrow_tile(int64_t MT, int cols) {
    if (cols <= 64 || (cols % 128 != 0 && cols % 128 <= 64)) return true;
    const int64_t t128 = MT * ((cols + 127) / 128), t64 = MT * ((cols + 63) / 64);
    const double c128 = (double)((t128 + 255) / 256) * 128.0;
    const double c64 = (double)((t64 + 255) / 256) * 64.0 * 1.10;
    return c64 < c128;
}

static ConvArgs base_args(const iswm_conv_desc* d) {
    ConvArgs a{};
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin;
    a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    a.ldx = d->ldx; a.ldy = d->ldy;
    return a;
}

struct WgradPlan {
    int bm, bn, MT, NT, nsplit, psplit;
};

static WgradPlan plan_wgrad(const iswm_conv_desc* d, bool x6) {
    WgradPlan p;
    const int Ktot = d->KH * d->KW * d->Cin;
    p.bm = (d->Cout % 128 == 0) ? 128 : 64;
    p.bn = (p.bm == 128 && (Ktot % 128 == 0 || Ktot >= 1024)) ? 128 : 64;
    if (p.bn == 64) p.bm = 64;  // instantiated shapes: 128x128 and 64x64
    p.MT = (d->Cout + p.bm - 1) / p.bm;
    p.NT = (Ktot + p.bn - 1) / p.bn;
    const int64_t P = (int64_t)d->N * d->Ho * d->Wo;
    const int64_t tiles = (int64_t)p.MT * p.NT;
    // Split count: minimise  rounds x (chunks per workgroup) x chunk time  +  slab write/read time, where a
    // round is 512 co-resident workgroups (2 per CU) and a 128x128x32 chunk takes ~4.5 us when two
    // workgroups share a CU.  tiles*splits just above a multiple of 512 costs a whole extra round.
    // (Three bf16x6 workgroups per CU fit in LDS and registers but measured no faster than two: r01 notes.)
    (void)x6;
    const int64_t slots = 512;
    const double chunk_us = 4.5 * (double)(p.bm * p.bn) / 16384.0;
    const double slab_us = (double)d->Cout * Ktot * 8.0 / 4.0e6;   // one slab written + read at ~4 TB/s
    int64_t maxs = (P + 255) / 256;                                 // at least 256 pixels per split
    if (maxs > 64) maxs = 64;
    double best = 1e300;
    p.psplit = (int)((P + 31) / 32 * 32);
    p.nsplit = 1;
    for (int64_t ns = 1; ns <= maxs; ++ns) {
        int64_t ps = ((P + ns - 1) / ns + 31) / 32 * 32;
        int64_t nsp = (P + ps - 1) / ps;
        int64_t rounds = (tiles * nsp + slots - 1) / slots;
        double t = (double)rounds * (double)(ps / 32 + 3) * chunk_us + (nsp > 1 ? (double)nsp * slab_us + 5.0 : 0.0);
        if (t < best) {
            best = t;
            p.psplit = (int)ps;
            p.nsplit = (int)nsp;
        }
    }
    return p;
}

}  // namespace iswm

using namespace iswm;

// 0: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32);  1 (default): bf16x6 split on the bf16 matrix cores
static int g_conv_math = -1;
static int conv_math() {
    if (g_conv_math < 0) {
        const char* e = getenv("ISWM_CONV_MATH");
        g_conv_math = (e && (!strcmp(e, "f32") || !strcmp(e, "0"))) ? 0
                      : (e && (!strcmp(e, "bf16") || !strcmp(e, "2"))) ? 2 : 1;   // default: bf16x6
    }
    return g_conv_math;
}
extern "C" int iswm_set_conv_math(int mode) {
    ISWM_REQUIRE(mode >= 0 && mode <= 2, "set_conv_math: mode must be 0 (f32), 1 (bf16x6) or 2 (bf16)");
    g_conv_math = mode;
    return 0;
}
extern "C" int iswm_get_conv_math(void) { return conv_math(); }
// bf16 planes per operand of the packed / weight-gradient kernels under the current math: 3 (bf16x6) or 1 (bf16)
static int math_planes() { return conv_math() == 2 ? 1 : 3; }

// Halo-patch kernel applicability (stride-1 KxK, bf16x6, packed weights); ISWM_X6_PATCH=0 disables it.
static int g_x6_patch = -1;
static bool patch_plan(const iswm_conv_desc* d, bool dgrad, int* PH, int* PW) {
    if (g_x6_patch < 0) {
        const char* e = getenv("ISWM_X6_PATCH");
        g_x6_patch = (e && e[0] == '0') ? 0 : 1;
    }
    if (!g_x6_patch || d->stride != 1 || d->KH * d->KW <= 1) return false;
    return dgrad ? conv_patch_plan(d->H, d->W, d->KH, d->KW, d->dil, PH, PW)
                 : conv_patch_plan(d->Ho, d->Wo, d->KH, d->KW, d->dil, PH, PW);
}

static PatchArgs patch_args(const iswm_conv_desc* d, bool dgrad, int PH, int PW) {
    PatchArgs p{};
    p.N = d->N;
    p.KH = d->KH; p.KW = d->KW; p.dil = d->dil;
    p.PH = PH; p.PW = PW;
    if (!dgrad) {
        p.RH = d->Ho; p.RW = d->Wo; p.GH = d->H; p.GW = d->W; p.GC = d->Cin; p.NC = d->Cout;
        p.orgh = -d->pad; p.orgw = -d->pad; p.flip = 0; p.ldg = d->ldx; p.ldo = d->ldy;
    } else {
        p.RH = d->H; p.RW = d->W; p.GH = d->Ho; p.GW = d->Wo; p.GC = d->Cout; p.NC = d->Cin;
        p.orgh = d->pad - d->dil * (d->KH - 1); p.orgw = d->pad - d->dil * (d->KW - 1);
        p.flip = 1; p.ldg = d->ldy; p.ldo = d->ldx;
    }
    return p;
}

namespace iswm { int wgrad_pl_is_wide(const iswm_conv_desc* d); int wgrad_pl_kernel_kind(const iswm_conv_desc* d); }

// may this geometry run the 256-column planes kernel (conv_mfma_pl2w.hip)?  bf16x6 only; strided data gradients keep the
// parity-ordered rows of k_conv_pl2
static bool pl2_wide_ok(const iswm_conv_desc* d, bool dgrad) {
    if (math_planes() != 3 || (dgrad && d->stride != 1)) return false;
    return !dgrad || d->KH * d->KW * d->Cout >= 512;          // the data gradient needs 8 stages per tile to pay
}
static int pl2_K(const iswm_conv_desc* d, bool dgrad) { return d->KH * d->KW * (dgrad ? d->Cout : d->Cin); }

extern "C" int iswm_conv2d_kernel_name(const iswm_conv_desc* d, int kind, char* buf, int buflen) {
    ISWM_REQUIRE(d && buf && buflen > 0 && kind >= 0 && kind <= 7, "kernel_name: bad argument");
    if (kind == 7) {   // iswm_conv2d_wgrad_planes
        const int kk = wgrad_pl_kernel_kind(d);
        snprintf(buf, buflen, kk == 2 ? "k_wgrad_pls<%d, 0, 4>" : kk == 1 ? "k_wgrad_plw<%d, false, 0>" : "k_wgrad_pl<%d>", math_planes());
        return 0;
    }
    if (kind >= 5) {   // 5 / 6: iswm_conv2d_fwd_pl2 / iswm_conv2d_dgrad_pl2
        const bool dg = kind == 6;
        const int cols = dg ? d->Cin : d->Cout;
        int rbw, wide;
        conv_pl2_plan(dg ? (int64_t)d->N * d->H * d->W : (int64_t)d->N * d->Ho * d->Wo, cols, pl2_K(d, dg), pl2_wide_ok(d, dg), &rbw, &wide);
        if (wide) snprintf(buf, buflen, "k_conv_pl2w<%d, %d, %s>", rbw, math_planes(), dg ? "true" : "false");
        else if (cols <= 64) snprintf(buf, buflen, "k_conv_pl2<%d, 2, %d, %s, false, 0>", rbw / 2, math_planes(), dg ? "true" : "false");
        else snprintf(buf, buflen, "k_conv_pl2<%d, 1, %d, %s, false, 0>", rbw, math_planes(), dg ? "true" : "false");
        return 0;
    }
    if (kind >= 3) {   // 3 / 4: iswm_conv2d_fwd_packed / iswm_conv2d_dgrad_packed
        const bool dg = kind == 4;
        int pbm, pbn;
        if (patch_plan(d, dg, &pbm, &pbn)) {
            snprintf(buf, buflen, "k_conv_x6_patch<%s, %d>", dg ? "true" : "false", math_planes());
            return 0;
        }
        conv_pick_tile_x6(dg ? (int64_t)d->N * d->H * d->W : (int64_t)d->N * d->Ho * d->Wo, dg ? d->Cin : d->Cout,
                          d->KH * d->KW * (dg ? d->Cout : d->Cin), dg, d->KH * d->KW == 1, &pbm, &pbn);
        snprintf(buf, buflen, "k_conv_x6<%d, 64, %s, true, %d>", pbm, dg ? "true" : "false", math_planes());
        return 0;
    }
    int bm, bn;
    if (kind == 0) {
        const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
        if (d->Cin % 32 == 0) {
            if (conv_math() == 1) conv_pick_tile_x6(M, d->Cout, d->KH * d->KW * d->Cin, false, d->KH * d->KW == 1, &bm, &bn);
            else conv_pick_tile(M, d->Cout, &bm, &bn);
            snprintf(buf, buflen, conv_math() == 1 ? "k_conv_x6<%d, %d, false, false, 3>" : "k_conv_fwd_u<%d, %d>", bm, bn);
        } else if (conv_math() == 1 && stem_geometry(base_args(d))) {
            snprintf(buf, buflen, "k_stem_fwd<%d>", stem_tile_rows() / 16);
        } else {
            snprintf(buf, buflen, "k_conv_fwd<%d>", use_narrow_tile((M + 127) / 128, d->Cout) ? 64 : 128);
        }
    } else if (kind == 1) {
        const int64_t M = (int64_t)d->N * d->H * d->W;
        if (d->Cout % 32 == 0) {
            if (conv_math() == 1) conv_pick_tile_x6(M, d->Cin, d->KH * d->KW * d->Cout, true, d->KH * d->KW == 1, &bm, &bn);
            else conv_pick_tile(M, d->Cin, &bm, &bn);
            if (conv_math() == 1) snprintf(buf, buflen, "k_conv_x6<%d, %d, true, false, 3>", bm, bn);
            else snprintf(buf, buflen, "k_conv_dgrad_u<%d, %d>", bm, bn);
        } else {
            snprintf(buf, buflen, "k_conv_dgrad<%d>", use_narrow_tile((M + 127) / 128, d->Cin) ? 64 : 128);
        }
    } else {
        if (conv_math() == 1 && stem_geometry(base_args(d))) {
            snprintf(buf, buflen, "k_stem_wgrad");
            return 0;
        }
        WgradPlan p = plan_wgrad(d, conv_math() >= 1);
        const bool same = d->stride == 1 && d->Ho == d->H && d->Wo == d->W;
        const int mode = (same && d->KH == 1 && d->KW == 1 && d->pad == 0) ? 2 : (same ? 1 : 0);
        snprintf(buf, buflen, "k_conv_wgrad<%d, %d, %d, %s, %d>", p.bm, p.bn, mode, conv_math() >= 1 ? "true" : "false",
                 conv_math() == 2 ? 1 : 3);
    }
    return 0;
}

extern "C" int iswm_conv2d_stat_tile_rows(const iswm_conv_desc* d) {
    if (!d) return 0;
    if (conv_math() == 1 && stem_geometry(base_args(d))) return stem_tile_rows();
    if (conv_math() == 1 && d->Cin % 32 == 0) {
        int bm, bn;
        conv_pick_tile_x6((int64_t)d->N * d->Ho * d->Wo, d->Cout, d->KH * d->KW * d->Cin, false, d->KH * d->KW == 1, &bm, &bn);
        return bm;
    }
    return conv_fwd_tile_rows((int64_t)d->N * d->Ho * d->Wo, d->Cin, d->Cout);
}

extern "C" int iswm_conv2d_stat_tiles(const iswm_conv_desc* d) {
    if (!d) return 0;
    const int64_t M = (int64_t)d->N * d->Ho * d->Wo, R = iswm_conv2d_stat_tile_rows(d);
    return (int)((M + R - 1) / R);
}

extern "C" int iswm_conv2d_fwd(const iswm_conv_desc* d, const float* x, const float* w, const float* bias,
                               float* y, float* stat_partials, iswm_stream_t stream) {
    if (int e = validate(d)) return e;
    ISWM_REQUIRE(x && w && y, "conv_fwd: null pointer");
    ISWM_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y), "conv_fwd: pointers must be 16-byte aligned");
    ConvArgs a = base_args(d);
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.stats = stat_partials;
    a.M = d->N * d->Ho * d->Wo;
    a.Ktot = d->KH * d->KW * d->Cin;
    a.MT = (a.M + 127) / 128;
    hipStream_t s = (hipStream_t)stream;
    if (conv_math() == 1 && launch_stem_fwd(a, s)) return check_launch("stem_fwd");
    if (conv_math() == 1 && d->Cin % 32 == 0) {
        int bm, bn;
        conv_pick_tile_x6(a.M, d->Cout, a.Ktot, false, d->KH * d->KW == 1, &bm, &bn);
        if (launch_conv_fwd_x6(a, s, bm, bn)) return check_launch("conv_fwd_x6");
    }
    if (launch_conv_fwd_u(a, s)) return check_launch("conv_fwd_u");
    if (use_narrow_tile(a.MT, d->Cout)) {
        a.NT = (d->Cout + 63) / 64;
        hipLaunchKernelGGL(k_conv_fwd<64>, dim3(a.MT * a.NT), dim3(256), 0, s, a);
    } else {
        a.NT = (d->Cout + 127) / 128;
        hipLaunchKernelGGL(k_conv_fwd<128>, dim3(a.MT * a.NT), dim3(256), 0, s, a);
    }
    return check_launch("conv_fwd");
}

extern "C" int iswm_conv2d_dgrad(const iswm_conv_desc* d, const float* dy, const float* w, float* dx,
                                 int accumulate, iswm_stream_t stream) {
    if (int e = validate(d)) return e;
    ISWM_REQUIRE(dy && w && dx, "conv_dgrad: null pointer");
    ISWM_REQUIRE(aligned16(dy) && aligned16(w) && aligned16(dx), "conv_dgrad: pointers must be 16-byte aligned");
    ConvArgs a = base_args(d);
    // kernel naming: a.x = gathered operand (dy, pitch ldy), a.y = output (dx, pitch ldx)
    a.x = dy; a.w = w; a.y = dx; a.accumulate = accumulate;
    a.ldx = d->ldy; a.ldy = d->ldx;
    a.M = d->N * d->H * d->W;
    a.Ktot = d->KH * d->KW * d->Cout;
    a.MT = (a.M + 127) / 128;
    hipStream_t s = (hipStream_t)stream;
    if (launch_conv_dgrad_u(a, s)) return check_launch("conv_dgrad_u");
    if (use_narrow_tile(a.MT, d->Cin)) {
        a.NT = (d->Cin + 63) / 64;
        hipLaunchKernelGGL(k_conv_dgrad<64>, dim3(a.MT * a.NT), dim3(256), 0, s, a);
    } else {
        a.NT = (d->Cin + 127) / 128;
        hipLaunchKernelGGL(k_conv_dgrad<128>, dim3(a.MT * a.NT), dim3(256), 0, s, a);
    }
    return check_launch("conv_dgrad");
}

extern "C" int iswm_transpose_weights(const iswm_conv_desc* d, const float* w, float* wt, iswm_stream_t stream) {
    if (int e = validate(d)) return e;
    ISWM_REQUIRE(w && wt && w != wt, "transpose_weights: bad pointer");
    launch_transpose_ohwi(w, wt, d->Cout, d->KH * d->KW, d->Cin, (hipStream_t)stream);
    return check_launch("transpose_weights");
}

/* 1 when iswm_conv2d_dgrad_wt (bf16x6 data gradient on transposed weights) applies to this geometry under
 * the current conv math */
extern "C" int iswm_conv2d_dgrad_wants_wt(const iswm_conv_desc* d) {
    return (d && conv_math() == 1 && d->Cout % 32 == 0) ? 1 : 0;
}

extern "C" int iswm_conv2d_dgrad_wt(const iswm_conv_desc* d, const float* dy, const float* wt, float* dx,
                                    int accumulate, iswm_stream_t stream) {
    if (int e = validate(d)) return e;
    ISWM_REQUIRE(dy && wt && dx, "conv_dgrad_wt: null pointer");
    ISWM_REQUIRE(aligned16(dy) && aligned16(wt) && aligned16(dx), "conv_dgrad_wt: pointers must be 16-byte aligned");
    ISWM_REQUIRE(d->Cout % 32 == 0, "conv_dgrad_wt: Cout must be a multiple of 32");
    ConvArgs a = base_args(d);
    a.x = dy; a.w = wt; a.y = dx; a.accumulate = accumulate;
    a.ldx = d->ldy; a.ldy = d->ldx;
    a.M = d->N * d->H * d->W;
    a.Ktot = d->KH * d->KW * d->Cout;
    int bm, bn;
    conv_pick_tile_x6(a.M, d->Cin, a.Ktot, true, d->KH * d->KW == 1, &bm, &bn);
    launch_conv_dgrad_x6(a, (hipStream_t)stream, bm, bn);
    return check_launch("conv_dgrad_x6");
}

/* BN-partials layout of iswm_conv2d_fwd_packed: *tile_rows == 0 means the tiles are image patches with varying
 * row counts, stored as floats after the two planes (partials + 2*tiles*Cout). */
extern "C" int iswm_conv2d_fwd_packed_stat_layout(const iswm_conv_desc* d, int* tiles, int* tile_rows) {
    ISWM_REQUIRE(d && tiles && tile_rows, "fwd_packed_stat_layout: null pointer");
    int PH, PW;
    if (patch_plan(d, false, &PH, &PW)) {
        *tiles = d->N * ((d->Ho + PH - 1) / PH) * ((d->Wo + PW - 1) / PW);
        *tile_rows = 0;
    } else {
        int bm, bn;     // the tile launch_conv_x6_pk will use
        const int64_t M = (int64_t)d->N * d->Ho * d->Wo;
        conv_pick_tile_x6(M, d->Cout, d->KH * d->KW * d->Cin, false, d->KH * d->KW == 1, &bm, &bn);
        *tile_rows = bm;
        *tiles = (int)((M + bm - 1) / bm);
    }
    return 0;
}

/* ---- bf16x6 with pre-split, fragment-ordered weights ("packed"): kind 0 = forward, 1 = data gradient ---- */
extern "C" size_t iswm_conv2d_packed_weight_bytes(const iswm_conv_desc* d, int kind) {
    if (!d || conv_math() < 1 || (kind != 0 && kind != 1)) return 0;
    const int gc = kind ? d->Cout : d->Cin;
    if (gc % 32 != 0) return 0;
    return packed_weight_bytes_x6(d->Cout, d->KH * d->KW, d->Cin, kind == 1, math_planes());
}

extern "C" int iswm_conv2d_pack_weights(const iswm_conv_desc* d, int kind, const float* w, void* packed,
                                        iswm_stream_t stream) {
    if (int e = validate(d)) return e;
    ISWM_REQUIRE(kind == 0 || kind == 1, "pack_weights: kind must be 0 (forward) or 1 (data gradient)");
    ISWM_REQUIRE(w && packed && aligned16(packed), "pack_weights: bad pointer");
    ISWM_REQUIRE((kind ? d->Cout : d->Cin) % 32 == 0, "pack_weights: gathered channel count must be a multiple of 32");
    launch_pack_weights_x6(w, packed, d->Cout, d->KH * d->KW, d->Cin, kind == 1, math_planes(), (hipStream_t)stream);
    return check_launch("pack_weights");
}

/* ---- batched packing: every conv of a model in one launch ---- */
extern "C" size_t iswm_packed_weight_bytes(int Cout, int taps, int Cin, int kind) {
    if (Cout <= 0 || taps <= 0 || Cin <= 0 || kind < 0 || kind > 3 || conv_math() < 1) return 0;
    if (kind >= 2) {        // planes kernels: gathered channel count a multiple of 64
        if (((kind == 3) ? Cout : Cin) % 64 != 0) return 0;
        return packed_weight_bytes_pl2(Cout, taps, Cin, kind == 3, math_planes());
    }
    if ((kind ? Cout : Cin) % 32 != 0) return 0;
    return packed_weight_bytes_x6(Cout, taps, Cin, kind == 1, math_planes());
}

extern "C" int iswm_pack_job_blocks(int Cout, int taps, int Cin, int kind) {
    if (iswm_packed_weight_bytes(Cout, taps, Cin, kind) == 0) return 0;
    if (kind >= 2) return pack_job_blocks_pl2(Cout, taps, Cin, kind == 3);
    return pack_job_blocks_x6(Cout, taps, Cin, kind == 1);
}

extern "C" int iswm_pack_weights_batch(const iswm_pack_job* jobs_dev, int njobs, int total_blocks,
                                       iswm_stream_t stream) {
    ISWM_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0, "pack_weights_batch: bad argument");
    static_assert(sizeof(iswm_pack_job) == 40, "iswm_pack_job layout");
    launch_pack_weights_batch(jobs_dev, njobs, total_blocks, math_planes(), (hipStream_t)stream);
    return check_launch("pack_weights_batch");
}

extern "C" int iswm_conv2d_fwd_packed(const iswm_conv_desc* d, const float* x, const void* wpk, const float* bias,
                                      float* y, float* stat_partials, iswm_stream_t stream) {
    if (int e = validate(d)) return e;
    ISWM_REQUIRE(x && wpk && y, "conv_fwd_packed: null pointer");
    ISWM_REQUIRE(aligned16(x) && aligned16(wpk) && aligned16(y), "conv_fwd_packed: pointers must be 16-byte aligned");
    ISWM_REQUIRE(d->Cin % 32 == 0, "conv_fwd_packed: Cin must be a multiple of 32");
    ConvArgs a = base_args(d);
    a.x = x; a.w = reinterpret_cast<const float*>(wpk); a.bias = bias; a.y = y; a.stats = stat_partials;
    a.M = d->N * d->Ho * d->Wo;
    a.Ktot = d->KH * d->KW * d->Cin;
    int PH, PW;
    if (patch_plan(d, false, &PH, &PW)) {
        PatchArgs p = patch_args(d, false, PH, PW);
        p.x = x; p.wpk = reinterpret_cast<const uint4*>(wpk); p.bias = bias; p.y = y; p.stats = stat_partials;
        launch_conv_x6_patch(p, false, math_planes(), (hipStream_t)stream);
        return check_launch("conv_fwd_patch");
    }
    int bm, bn;
    conv_pick_tile_x6(a.M, d->Cout, a.Ktot, false, d->KH * d->KW == 1, &bm, &bn);
    launch_conv_x6_pk(a, (hipStream_t)stream, false, bm, math_planes());
    return check_launch("conv_fwd_packed");
}

extern "C" int iswm_conv2d_dgrad_packed(const iswm_conv_desc* d, const float* dy, const void* wpk, float* dx,
                                        int accumulate, iswm_stream_t stream) {
    if (int e = validate(d)) return e;
    ISWM_REQUIRE(dy && wpk && dx, "conv_dgrad_packed: null pointer");
    ISWM_REQUIRE(aligned16(dy) && aligned16(wpk) && aligned16(dx), "conv_dgrad_packed: pointers must be 16-byte aligned");
    ISWM_REQUIRE(d->Cout % 32 == 0, "conv_dgrad_packed: Cout must be a multiple of 32");
    ConvArgs a = base_args(d);
    a.x = dy; a.w = reinterpret_cast<const float*>(wpk); a.y = dx; a.accumulate = accumulate;
    a.ldx = d->ldy; a.ldy = d->ldx;
    a.M = d->N * d->H * d->W;
    a.Ktot = d->KH * d->KW * d->Cout;
    int PH, PW;
    if (patch_plan(d, true, &PH, &PW)) {
        PatchArgs p = patch_args(d, true, PH, PW);
        p.x = dy; p.wpk = reinterpret_cast<const uint4*>(wpk); p.y = dx; p.accumulate = accumulate;
        launch_conv_x6_patch(p, true, math_planes(), (hipStream_t)stream);
        return check_launch("conv_dgrad_patch");
    }
    int bm, bn;
    conv_pick_tile_x6(a.M, d->Cin, a.Ktot, true, d->KH * d->KW == 1, &bm, &bn);
    launch_conv_x6_pk(a, (hipStream_t)stream, true, bm, math_planes());
    return check_launch("conv_dgrad_packed");
}

/* ---- activations pre-split into bf16 planes (see include/iswm_hip.h "planes") ---- */
extern "C" int iswm_split_planes(const float* x, int64_t M, int C, int ldx, void* planes, int ldp, int64_t plane_stride,
                                 iswm_stream_t stream) {
    ISWM_REQUIRE(x && planes && M > 0 && C > 0, "split_planes: bad argument");
    ISWM_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldx >= C && ldp % 4 == 0 && ldp >= C, "split_planes: C %d ldx %d ldp %d", C, ldx, ldp);
    ISWM_REQUIRE(aligned16(x) && plane_stride % 4 == 0 && plane_stride >= M * ldp,
                 "split_planes: planes must be 16-byte aligned and disjoint");
    launch_split_planes(x, M, C, ldx, (unsigned short*)planes, ldp, plane_stride, math_planes(), (hipStream_t)stream);
    return check_launch("split_planes");
}

extern "C" int iswm_join_planes(const void* planes, int ldp, int64_t plane_stride, int64_t M, int C, float* x, int ldx,
                                iswm_stream_t stream) {
    ISWM_REQUIRE(x && planes && M > 0 && C > 0, "join_planes: bad argument");
    ISWM_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ldx >= C && ldp % 4 == 0 && ldp >= C, "join_planes: C %d ldx %d ldp %d", C, ldx, ldp);
    ISWM_REQUIRE(plane_stride == -1 || plane_stride >= M * ldp, "join_planes: bad plane stride");
    launch_join_planes((const unsigned short*)planes, ldp, plane_stride, M, C, x, ldx, (hipStream_t)stream);
    return check_launch("join_planes");
}

namespace iswm { extern unsigned long long* g_conv_dbg; }
/* diagnostics: a device buffer of >= 512 uint64 that workgroup 0 of the planes conv kernels fills with per-stage shader-clock
 * stamps (tools/pl2_timeline.py); NULL (the default) switches the stamps off */
extern "C" int iswm_set_debug_buffer(void* buf) {
    iswm::g_conv_dbg = reinterpret_cast<unsigned long long*>(buf);
    return 0;
}

/* second-generation planes kernels (conv_mfma_pl2.hip): kind 0 forward, 1 data gradient */
extern "C" size_t iswm_conv2d_pl2_weight_bytes(const iswm_conv_desc* d, int kind) {
    if (!d || conv_math() < 1 || (kind != 0 && kind != 1)) return 0;
    const int gc = kind ? d->Cout : d->Cin;
    if (gc % 64 != 0) return 0;
    return packed_weight_bytes_pl2(d->Cout, d->KH * d->KW, d->Cin, kind == 1, math_planes());
}

extern "C" int iswm_conv2d_pl2_pack_weights(const iswm_conv_desc* d, int kind, const float* w, void* packed,
                                            iswm_stream_t stream) {
    if (int e = validate(d)) return e;
    ISWM_REQUIRE(kind == 0 || kind == 1, "pl2_pack_weights: kind must be 0 (forward) or 1 (data gradient)");
    ISWM_REQUIRE(w && packed && aligned16(packed), "pl2_pack_weights: bad pointer");
    ISWM_REQUIRE((kind ? d->Cout : d->Cin) % 64 == 0, "pl2_pack_weights: gathered channel count must be a multiple of 64");
    launch_pack_weights_pl2(w, packed, d->Cout, d->KH * d->KW, d->Cin, kind == 1, math_planes(), (hipStream_t)stream);
    return check_launch("pl2_pack_weights");
}

extern "C" int iswm_conv2d_pl2_tile_rows(const iswm_conv_desc* d, int kind) {
    if (!d) return 0;
    int rbw, wide;
    if (kind) conv_pl2_plan((int64_t)d->N * d->H * d->W, d->Cin, pl2_K(d, true), pl2_wide_ok(d, true), &rbw, &wide);
    else conv_pl2_plan((int64_t)d->N * d->Ho * d->Wo, d->Cout, pl2_K(d, false), pl2_wide_ok(d, false), &rbw, &wide);
    return 16 * rbw;
}

extern "C" int iswm_conv2d_fwd_pl2(const iswm_conv_desc* d, const void* xp, int64_t plane_stride, const void* wpk,
                                   const float* bias, float* y, float* stat_partials, iswm_stream_t stream) {
    if (int e = validate(d)) return e;
    ISWM_REQUIRE(xp && wpk && y, "conv_fwd_pl2: null pointer");
    ISWM_REQUIRE(aligned16(xp) && aligned16(wpk) && aligned16(y), "conv_fwd_pl2: pointers must be 16-byte aligned");
    ISWM_REQUIRE(d->Cin % 64 == 0 && d->ldx % 8 == 0 && (plane_stride % 8 == 0 || (plane_stride == -1 && math_planes() == 1)),
                 "conv_fwd_pl2: Cin %% 64, ldx %% 8, plane stride %% 8 (or -1: one rounded plane under conv math bf16)");
    ConvArgs a = base_args(d);
    a.x = reinterpret_cast<const float*>(xp); a.w = reinterpret_cast<const float*>(wpk); a.bias = bias; a.y = y;
    a.stats = stat_partials;
    a.xps = plane_stride * 2;
    a.M = d->N * d->Ho * d->Wo;
    a.Ktot = d->KH * d->KW * d->Cin;
    int rbw, wide;
    conv_pl2_plan(a.M, d->Cout, pl2_K(d, false), pl2_wide_ok(d, false), &rbw, &wide);
    ISWM_REQUIRE(wide ? launch_conv_pl2w(a, (hipStream_t)stream, false, math_planes(), rbw)
                      : launch_conv_pl2(a, (hipStream_t)stream, false, math_planes(), rbw),
                 "conv_fwd_pl2: no kernel for this configuration");
    return check_launch("conv_fwd_pl2");
}

static int dgrad_pl2_impl(const iswm_conv_desc* d, const void* dyp, int64_t plane_stride, const void* wpk, float* dx,
                          int accumulate, const BnFuse* f, iswm_stream_t stream) {
    if (int e = validate(d)) return e;
    ISWM_REQUIRE(dyp && wpk && dx, "conv_dgrad_pl2: null pointer");
    ISWM_REQUIRE(aligned16(dyp) && aligned16(wpk) && aligned16(dx), "conv_dgrad_pl2: pointers must be 16-byte aligned");
    ISWM_REQUIRE(d->Cout % 64 == 0 && d->ldy % 8 == 0 && (plane_stride % 8 == 0 || (plane_stride == -1 && math_planes() == 1)),
                 "conv_dgrad_pl2: Cout %% 64, ldy %% 8, plane stride %% 8 (or -1: one rounded plane under conv math bf16)");
    ConvArgs a = base_args(d);
    a.x = reinterpret_cast<const float*>(dyp); a.w = reinterpret_cast<const float*>(wpk); a.y = dx; a.accumulate = accumulate;
    a.ldx = d->ldy; a.ldy = d->ldx;
    a.xps = plane_stride * 2;
    a.M = d->N * d->H * d->W;
    a.Ktot = d->KH * d->KW * d->Cout;
    if (f) a.bnf = *f;
    int rbw, wide;
    conv_pl2_plan(a.M, d->Cin, pl2_K(d, true), pl2_wide_ok(d, true), &rbw, &wide);
    ISWM_REQUIRE(wide ? launch_conv_pl2w(a, (hipStream_t)stream, true, math_planes(), rbw)
                      : launch_conv_pl2(a, (hipStream_t)stream, true, math_planes(), rbw),
                 "conv_dgrad_pl2: no kernel for this configuration");
    return check_launch("conv_dgrad_pl2");
}

extern "C" int iswm_conv2d_dgrad_pl2(const iswm_conv_desc* d, const void* dyp, int64_t plane_stride, const void* wpk,
                                     float* dx, int accumulate, iswm_stream_t stream) {
    return dgrad_pl2_impl(d, dyp, plane_stride, wpk, dx, accumulate, nullptr, stream);
}

/* tile rows of the planes data gradient = first dimension of the statistics it can emit for the consumer BatchNorm backward */
extern "C" int iswm_conv2d_dgrad_pl2_stat_tiles(const iswm_conv_desc* d) {
    if (!d || d->Cin <= 0) return 0;
    const int64_t M = (int64_t)d->N * d->H * d->W;
    int rbw, wide;
    conv_pl2_plan(M, d->Cin, pl2_K(d, true), pl2_wide_ok(d, true), &rbw, &wide);
    const int wm = d->Cin <= 64 ? 2 : 1;                 // narrow tiles: (rbw / 2) blocks x 2 wave rows
    const int64_t mt = (M + rbw * 16 - 1) / (rbw * 16);
    return (int)(mt * wm);
}

/* iswm_conv2d_dgrad_pl2 that also emits, per tile row and input channel, the two sums the BatchNorm backward of the stage
 * that PRODUCED the conv's input needs over the finished dx (after accumulation):  partials[0][t][c] = sum dz,
 * partials[1][t][c] = sum dz * xhat,  dz = dx * [ReLU pattern], xhat = (y - mean) * invstd.  relu: 0 none, 2 pattern
 * recomputed as (y - mean) * mask_scale + mask_shift > 0 (as iswm_bn_backward does), 3 the producer is a RESIDUAL stage:
 * pattern = (hi plane of its saved output, mask_hi, pitch ld_mask bf16 elements) > 0, and dx is STORED MASKED (dz): that
 * tensor is both the dout of the producer's BatchNorm backward (call it with relu = 0) and the gradient of its identity
 * branch, so neither the reduction pass nor a separate `dres` tensor exists for that stage.  y: the producer's raw conv output
 * [N*H*W][ldy], Cin channels.  partials: 2 * tiles * Cin doubles, tiles = iswm_conv2d_dgrad_pl2_stat_tiles(d).  Feed them to
 * iswm_bn_backward_pl with partial_tiles = tiles: it then skips its own reduction pass over dout and y. */
extern "C" int iswm_conv2d_dgrad_pl2_bn(const iswm_conv_desc* d, const void* dyp, int64_t plane_stride, const void* wpk,
                                        float* dx, int accumulate, const float* y, int ldy, const float* mean,
                                        const float* invstd, const float* mask_scale, const float* mask_shift, int relu,
                                        const void* mask_hi, int ld_mask, double* partials, int tiles,
                                        iswm_stream_t stream) {
    ISWM_REQUIRE(d && y && mean && invstd && partials, "conv_dgrad_pl2_bn: null pointer");
    ISWM_REQUIRE(relu == 0 || (relu == 2 && mask_scale && mask_shift) ||
                     (relu == 3 && mask_hi && ld_mask % 4 == 0 && ld_mask >= d->Cin && (((uintptr_t)mask_hi) & 7) == 0),
                 "conv_dgrad_pl2_bn: relu must be 0, 2 (with mask_scale / mask_shift) or 3 (with the producer's saved output planes)");
    ISWM_REQUIRE(d->Cin % 4 == 0 && ldy % 4 == 0 && ldy >= d->Cin && aligned16(y) && aligned16(mean) && aligned16(invstd),
                 "conv_dgrad_pl2_bn: Cin %% 4, ldy %% 4, 16-byte aligned pointers");
    ISWM_REQUIRE(tiles == iswm_conv2d_dgrad_pl2_stat_tiles(d), "conv_dgrad_pl2_bn: tiles %d != %d", tiles,
                 iswm_conv2d_dgrad_pl2_stat_tiles(d));
    BnFuse f{};
    f.y = y; f.ldy = ldy; f.mean = mean; f.invstd = invstd; f.mscale = mask_scale; f.mshift = mask_shift; f.relu = relu;
    f.part = partials;
    f.mask = reinterpret_cast<const unsigned short*>(mask_hi); f.ldm = ld_mask;
    return dgrad_pl2_impl(d, dyp, plane_stride, wpk, dx, accumulate, &f, stream);
}

extern "C" size_t iswm_conv2d_wgrad_workspace(const iswm_conv_desc* d) {
    if (!d) return 0;
    if (conv_math() == 1 && stem_geometry(base_args(d))) {
        const size_t stem = stem_wgrad_workspace(base_args(d));          // 0: the stem's own weight gradient is switched off
        if (stem > 0) return stem;
    }
    WgradPlan p = plan_wgrad(d, conv_math() >= 1);
    if (p.nsplit <= 1) return 0;
    return (size_t)p.nsplit * d->Cout * d->KH * d->KW * d->Cin * sizeof(float);
}

extern "C" int iswm_conv2d_wgrad(const iswm_conv_desc* d, const float* x, const float* dy, float* dw,
                                 float* workspace, size_t workspace_bytes, iswm_stream_t stream) {
    if (int e = validate(d)) return e;
    ISWM_REQUIRE(x && dy && dw, "conv_wgrad: null pointer");
    ISWM_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dw), "conv_wgrad: pointers must be 16-byte aligned");
    WgradPlan p = plan_wgrad(d, conv_math() >= 1);
    const size_t need = iswm_conv2d_wgrad_workspace(d);
    ISWM_REQUIRE(workspace_bytes >= need && (need == 0 || (workspace && aligned16(workspace))),
                 "conv_wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
    ConvArgs a = base_args(d);
    a.x = x; a.y = const_cast<float*>(dy);
    a.M = d->N * d->Ho * d->Wo;
    a.Ktot = d->KH * d->KW * d->Cin;
    if (conv_math() == 1 && launch_stem_wgrad(a, dw, workspace, (hipStream_t)stream)) return check_launch("stem_wgrad");
    a.MT = p.MT; a.NT = p.NT; a.nsplit = p.nsplit; a.psplit = p.psplit;
    a.stats = (p.nsplit > 1) ? workspace : dw;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(p.MT * p.NT, p.nsplit);
    const bool same = d->stride == 1 && d->Ho == d->H && d->Wo == d->W;
    const int mode = (same && d->KH == 1 && d->KW == 1 && d->pad == 0) ? 2 : (same ? 1 : 0);
    const bool x6 = conv_math() >= 1, one = conv_math() == 2;
#define WLAUNCH(BM_, BN_, X_, NP_)                                                                          \
    do {                                                                                                    \
        if (mode == 2) hipLaunchKernelGGL((k_conv_wgrad<BM_, BN_, 2, X_, NP_>), grid, dim3(256), 0, s, a);      \
        else if (mode == 1) hipLaunchKernelGGL((k_conv_wgrad<BM_, BN_, 1, X_, NP_>), grid, dim3(256), 0, s, a); \
        else hipLaunchKernelGGL((k_conv_wgrad<BM_, BN_, 0, X_, NP_>), grid, dim3(256), 0, s, a);                \
    } while (0)
    if (p.bm == 128 && one) WLAUNCH(128, 128, true, 1);
    else if (p.bm == 128 && x6) WLAUNCH(128, 128, true, 3);
    else if (p.bm == 128) WLAUNCH(128, 128, false, 3);
    else if (one) WLAUNCH(64, 64, true, 1);
    else if (x6) WLAUNCH(64, 64, true, 3);
    else WLAUNCH(64, 64, false, 3);
#undef WLAUNCH
    if (int e = check_launch("conv_wgrad")) return e;
    if (p.nsplit > 1) {
        int64_t n4 = (int64_t)d->Cout * a.Ktot / 4;
        hipLaunchKernelGGL(k_reduce_slabs, dim3(stream_grid(n4, 256)), dim3(256), 0, s, workspace, dw, n4,
                           p.nsplit);
        return check_launch("conv_wgrad_reduce");
    }
    return 0;
}
