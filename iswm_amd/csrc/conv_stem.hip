// The ResNet stem convolution -- nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False), network/backbone/resnet.py:137,
// forward :145 -- as a bf16x6 GEMM of its own.
//
// With 3 (padded: 4) input channels no K chunk of the generic kernels lines up with a filter tap, so until round 3 the stem ran
// on the round-1 fp32-MFMA kernel with a per-element gather (0.40 ms, 48 TFLOP/s) and its weight gradient on the generic bf16x6
// one (0.61 ms).  Its geometry makes a much simpler GEMM: in NHWC4 the 7 input pixels a filter ROW touches are 28 contiguous
// floats, and with stride 2 the window of output pixel ow starts at input column 2 ow - 3.  K is laid out as 7 steps of 32 =
// (kernel row kh) x (8 input pixels x 4 channels); the 8th pixel carries zero weights (12.5 % padding of K).
//
//   forward   D[cout][pixel] += W[cout][kh: 32] . X[kh: 32][pixel]        v_mfma_f32_16x16x32_bf16, weights = row operand
//
// A lane of the column (activation) operand holds k = 8 g .. 8 g + 7 of pixel (lane & 15), g = lane >> 4: TWO adjacent input
// pixels, i.e. 32 contiguous bytes of the image -- loaded straight from global memory into registers (two float4, each replaced by
// a zero line when its pixel is outside the image), split exactly into three bf16 pieces on the VALU, multiplied.  No LDS stage
// and no barrier in the loop: activations are used by one wave only (all 64 output channels belong to the same wave).  The split
// weights (4 column blocks x 7 steps x 3 planes x 1 KB = 84 KB) sit in LDS for the lifetime of the persistent workgroup, which
// builds them itself from the OHWI fp32 parameter (no pack launch), and are read once per kernel row for all the wave's pixels.
// One wave = RB row blocks of 16 output pixels x 64 channels; 8 waves per workgroup walk the wave-tiles round-robin.
// Epilogue: lane = pixel, 4 registers = 4 consecutive channels -> float4 stores; BatchNorm partials per wave-tile (centred).
#include "conv_common.h"

namespace iswm {

typedef float stem_f32x4 __attribute__((ext_vector_type(4)));

static __device__ __attribute__((aligned(16))) float g_stem_zero[4];         // a zero pixel

__device__ __forceinline__ stem_f32x4 stem_mfma(uint4 a, uint4 b, stem_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int STEM_RB = 6;                  // 96 output pixels per wave-tile: 96 accumulator registers
constexpr int STEM_TILE = 16 * STEM_RB;

// a.x: NHWC4 fp32 image, a.w: OHWI [64][7][7][4] fp32, a.y: [M][ldy] fp32, a.stats: [2][MT][64] or null (MT = wave-tiles)
template <int RB>
__global__ __launch_bounds__(512) void k_stem_fwd(const ConvArgs a) {
    __shared__ __attribute__((aligned(16))) uint4 wl[4 * 7 * 3 * 64];         // [column block][kh][plane][lane]
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = lane >> 4, lp = lane & 15;

    for (int e = t; e < 4 * 7 * 64; e += 512) {
        const int l = e & 63, ck = e >> 6, kh = ck % 7, cb = ck / 7;
        const int co = cb * 16 + (l & 15), kw = 2 * (l >> 4);
        const float* wp = a.w + ((size_t)(co * 7 + kh) * 7 + kw) * 4;
        const float4 v0 = ldg4(wp);
        const float4 v1 = kw + 1 < 7 ? ldg4(wp + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        uint2 h0, m0, l0, h1, m1, l1;
        split3(v0, h0, m0, l0);
        split3(v1, h1, m1, l1);
        uint4* dst = wl + (size_t)(ck * 3) * 64 + l;
        dst[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        dst[64] = make_uint4(m0.x, m0.y, m1.x, m1.y);
        dst[128] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
    __syncthreads();

    const int H = a.H, W = a.W, HoWo = a.Ho * a.Wo;
    const float4* const x4 = reinterpret_cast<const float4*>(a.x);
    const float4* const zero4 = reinterpret_cast<const float4*>(g_stem_zero);
    const int nwt = a.MT;
    for (int wt = blockIdx.x * 8 + wave; wt < nwt; wt += gridDim.x * 8) {
        const int m0 = wt * (16 * RB);
        int ih0[RB], pix[RB];
        bool okA[RB], okB[RB], valid[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int m = m0 + rb * 16 + lp;
            valid[rb] = m < a.M;
            const int mm = valid[rb] ? m : 0;
            const int n = mm / HoWo, rem = mm - n * HoWo;
            const int oh = rem / a.Wo, ow = rem - oh * a.Wo;
            const int iw = 2 * ow - 3 + 2 * g;
            ih0[rb] = 2 * oh - 3;
            pix[rb] = (n * H + ih0[rb]) * W + iw;
            okA[rb] = valid[rb] && (unsigned)iw < (unsigned)W;
            okB[rb] = valid[rb] && (unsigned)(iw + 1) < (unsigned)W;
        }
        stem_f32x4 acc[RB][4];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = stem_f32x4{0.f, 0.f, 0.f, 0.f};

        // (fetching the pixels a kernel row ahead into the registers the split frees was measured: no faster -- 214 vs 202 us at the
        // RB = 5 it needs to stay clear of spills; two waves per SIMD cover the load latency already)
#pragma unroll 1
        for (int kh = 0; kh < 7; ++kh) {
            float4 ra[RB], rc[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const bool rowok = (unsigned)(ih0[rb] + kh) < (unsigned)H;
                const int p = pix[rb] + kh * W;
                ra[rb] = *((rowok && okA[rb]) ? x4 + p : zero4);
                rc[rb] = *((rowok && okB[rb]) ? x4 + p + 1 : zero4);
            }
            uint4 wf[4][3];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wf[cb][pl] = wl[((cb * 7 + kh) * 3 + pl) * 64 + lane];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                uint2 h0, m0_, l0, h1, m1, l1;
                split3(ra[rb], h0, m0_, l0);
                split3(rc[rb], h1, m1, l1);
                const uint4 xh = make_uint4(h0.x, h0.y, h1.x, h1.y), xm = make_uint4(m0_.x, m0_.y, m1.x, m1.y),
                            xl = make_uint4(l0.x, l0.y, l1.x, l1.y);
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    stem_f32x4 c = acc[rb][cb];
                    c = stem_mfma(wf[cb][0], xl, c);          // smallest terms first, as everywhere (conv_mfma_pl2.hip)
                    c = stem_mfma(wf[cb][2], xh, c);
                    c = stem_mfma(wf[cb][1], xm, c);
                    c = stem_mfma(wf[cb][0], xm, c);
                    c = stem_mfma(wf[cb][1], xh, c);
                    c = stem_mfma(wf[cb][0], xh, c);
                    acc[rb][cb] = c;
                }
            }
        }

        // ---- epilogue: lane -> pixel lp of each row block, its 4 registers -> channels 16 cb + 4 g + r
        if (a.bias != nullptr) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const float4 bv = *reinterpret_cast<const float4*>(a.bias + cb * 16 + 4 * g);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    acc[rb][cb][0] += bv.x; acc[rb][cb][1] += bv.y; acc[rb][cb][2] += bv.z; acc[rb][cb][3] += bv.w;
                }
            }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            if (valid[rb]) {
                float* o = a.y + (size_t)(m0 + rb * 16 + lp) * a.ldy + 4 * g;
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    *reinterpret_cast<float4*>(o + cb * 16) = make_float4(acc[rb][cb][0], acc[rb][cb][1], acc[rb][cb][2], acc[rb][cb][3]);
            }
        }
        if (a.stats != nullptr) {
            // per wave-tile, numerically centred: column sum first, then the squared deviations from the TILE mean (k_conv_fwd)
            const float cnt = (float)min(16 * RB, a.M - m0);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[r] += valid[rb] ? acc[rb][cb][r] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[r] += __shfl_xor(s[r], 1);
                    s[r] += __shfl_xor(s[r], 2);
                    s[r] += __shfl_xor(s[r], 4);
                    s[r] += __shfl_xor(s[r], 8);
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dv = acc[rb][cb][r] - s[r] / cnt;
                        q[r] += valid[rb] ? dv * dv : 0.f;
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    q[r] += __shfl_xor(q[r], 1);
                    q[r] += __shfl_xor(q[r], 2);
                    q[r] += __shfl_xor(q[r], 4);
                    q[r] += __shfl_xor(q[r], 8);
                }
                if (lp == 0) {
                    const int col = cb * 16 + 4 * g;
                    *reinterpret_cast<float4*>(&a.stats[(size_t)wt * 64 + col]) = make_float4(s[0], s[1], s[2], s[3]);
                    *reinterpret_cast<float4*>(&a.stats[(size_t)(nwt + wt) * 64 + col]) = make_float4(q[0], q[1], q[2], q[3]);
                }
            }
        }
    }
}

// ---- weight gradient:  dW[cout][kh][kw][c] = sum over output pixels p of dy[p][cout] * x[n, 2 oh - 3 + kh, 2 ow - 3 + kw, c]
// GEMM rows = the 64 output channels, columns = the 7 x 32 (kernel row, 8 pixels x 4 channels) slots of the forward's K layout,
// K = output pixels.  Both operands are K-strided in memory (channel contiguous), so they go through LDS as [pixel][channel] bf16
// images and come out as pixel-contiguous fragments by ds_read_b64_tr_b16 -- the image layout and the fragment reader of
// k_wgrad_pls (conv_wgrad_pl.hip): rows of 256 B = 128 channels, 16-byte group q of row r at byte (16 q) ^ (64 (r & 3)).
//   image 0 = [dy: 64 channels | kernel row 0 | kernel row 1],  image 1 = kernel rows 2..5,  image 2 = kernel row 6
// A workgroup walks its share of the pixels in 16-pixel stages: every thread fetches up to three float4 of the NEXT stage
// (dy: one pixel's 4 channels; x: one input pixel, zero outside the image) before the multiply of the current one, then splits
// them exactly into the three bf16 pieces and writes the other LDS buffer: one barrier per stage.  Waves 0..6 each own one
// kernel row: a 64 x 32 block of dW as two 32x32x16 accumulators.  Every workgroup writes its partial dW [64][224] and a second
// kernel sums the partials in a fixed order into the OHWI gradient (bit-reproducible, no float atomics).
constexpr int SW_KS = 16;                      // pixels per stage
constexpr int SW_PLANE = SW_KS * 256, SW_IMG = 3 * SW_PLANE, SW_STAGE = 3 * SW_IMG;

__global__ __launch_bounds__(512) void k_stem_wgrad(const ConvArgs a, float* __restrict__ slabs, int per) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * SW_STAGE];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int H = a.H, W = a.W, HoWo = a.Ho * a.Wo;
    const int p_begin = blockIdx.x * per, p_end = min(a.M, p_begin + per);
    const int nK = p_end > p_begin ? (p_end - p_begin + SW_KS - 1) / SW_KS : 0;
    const float4* const x4 = reinterpret_cast<const float4*>(a.x);
    const float4 zf = make_float4(0.f, 0.f, 0.f, 0.f);

    // fetch roles.  dy: threads 0..255 -> pixel t >> 4, channels 4 (t & 15).  x: items j = t, t + 512 (< 896): pixel j / 56,
    // kernel row (j % 56) >> 3, window pixel (j % 56) & 7
    const int dpx = t >> 4, dc4 = t & 15;
    int xpx[2], xkh[2], xq[2];
    bool xon[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = t + 512 * i;
        xon[i] = j < 16 * 56;
        const int jj = xon[i] ? j : 0;
        xpx[i] = jj / 56;
        const int rem = jj - xpx[i] * 56;
        xkh[i] = rem >> 3;
        xq[i] = rem & 7;
    }
    float4 rd = zf, rx[2] = {zf, zf};
    const float4* const zero4 = reinterpret_cast<const float4*>(g_stem_zero);
    // this thread's x pixels as (n, oh, ow), advanced by 16 output pixels per stage (Wo > 16: at most one carry each) -- two
    // integer divisions per item and stage cost more VALU time than the stage's matrix work
    int xn[2], xoh[2], xow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = min(p_begin + xpx[i], a.M - 1);
        xn[i] = p / HoWo;
        const int rem = p - xn[i] * HoWo;
        xoh[i] = rem / a.Wo;
        xow[i] = rem - xoh[i] * a.Wo;
    }
    auto fetch = [&](int s2) __attribute__((always_inline)) {          // branch-free: a pixel that is not there reads the zero pixel;
        const int p0 = p_begin + s2 * SW_KS;                           // called for s2 = 0, 1, 2, ... in order
        {
            const int p = p0 + dpx;
            const bool ok = t < 256 && p < p_end;
            rd = *(ok ? reinterpret_cast<const float4*>(a.y + (size_t)p * a.ldy + dc4 * 4) : zero4);          // a.y = dy
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = p0 + xpx[i];
            const int ih = 2 * xoh[i] - 3 + xkh[i], iw = 2 * xow[i] - 3 + xq[i];
            const bool ok = xon[i] && p < p_end && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
            rx[i] = *(ok ? x4 + ((size_t)(xn[i] * H + ih) * W + iw) : zero4);
            int ow = xow[i] + SW_KS;
            const int c1 = ow >= a.Wo ? 1 : 0;
            xow[i] = ow - (c1 ? a.Wo : 0);
            const int oh = xoh[i] + c1;
            const int c2 = oh >= a.Ho ? 1 : 0;
            xoh[i] = c2 ? 0 : oh;
            xn[i] += c2;
        }
    };
    auto put = [&](unsigned char* img, int row, int col, const float4 v) __attribute__((always_inline)) {
        uint2 h, m, l;
        split3(v, h, m, l);
        unsigned char* q = img + row * 256 + ((col * 2) ^ ((row & 3) * 64));
        *reinterpret_cast<uint2*>(q) = h;
        *reinterpret_cast<uint2*>(q + SW_PLANE) = m;
        *reinterpret_cast<uint2*>(q + 2 * SW_PLANE) = l;
    };
    auto store = [&](int buf) __attribute__((always_inline)) {
        unsigned char* st = smem + buf * SW_STAGE;
        if (t < 256) put(st, dpx, dc4 * 4, rd);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (xon[i]) {
                const int kh = xkh[i];
                const int img = kh < 2 ? 0 : (kh < 6 ? 1 : 2);
                const int col = (kh < 2 ? 64 + 32 * kh : (kh < 6 ? 32 * (kh - 2) : 0)) + 4 * xq[i];
                put(st + img * SW_IMG, xpx[i], col, rx[i]);
            }
        }
    };

    // fragment reader of k_wgrad_pls: 32 channels [col0, col0 + 32) x the stage's 16 pixels -> the 32x32x16 operand
    const int tg = lane >> 4, ti = lane & 15, tq = ti >> 2, tp = ti & 3;
    const int th = tg >> 1, tc = (tg & 1) * 16 + tp * 4;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    auto tr_frag = [&](const unsigned char* plane, int col0) __attribute__((always_inline)) -> uint4 {
        const unsigned char* q = plane + (th * 8 + tq) * 256 + (((col0 + tc) * 2) ^ (tq * 64));
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(q));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(q + 4 * 256));
        uint2 a2 = __builtin_bit_cast(uint2, lo), b2 = __builtin_bit_cast(uint2, hi);
        return make_uint4(a2.x, a2.y, b2.x, b2.y);
    };
    const int kh_w = wave;                                    // this wave's kernel row (wave 7 only fetches)
    const int bimg = kh_w < 2 ? 0 : (kh_w < 6 ? 1 : 2);
    const int bcol = kh_w < 2 ? 64 + 32 * kh_w : (kh_w < 6 ? 32 * (kh_w - 2) : 0);
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    if (nK > 0) {
        fetch(0);
        store(0);
    }
    __syncthreads();
    for (int s2 = 0; s2 < nK; ++s2) {
        const int buf = s2 & 1;
        if (s2 + 1 < nK) fetch(s2 + 1);
        if (wave < 7) {
            const unsigned char* st = smem + buf * SW_STAGE;
            uint4 FB[3], FA[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                FB[pl] = tr_frag(st + bimg * SW_IMG + pl * SW_PLANE, bcol);
                FA[0][pl] = tr_frag(st + pl * SW_PLANE, 0);
                FA[1][pl] = tr_frag(st + pl * SW_PLANE, 32);
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                f32x16 c = acc[mb];
                c = mfma_bf16(FA[mb][2], FB[0], c);     // smallest terms first
                c = mfma_bf16(FA[mb][0], FB[2], c);
                c = mfma_bf16(FA[mb][1], FB[1], c);
                c = mfma_bf16(FA[mb][1], FB[0], c);
                c = mfma_bf16(FA[mb][0], FB[1], c);
                c = mfma_bf16(FA[mb][0], FB[0], c);
                acc[mb] = c;
            }
        }
        if (s2 + 1 < nK) store(buf ^ 1);
        __syncthreads();
    }
    if (wave < 7) {
        float* out = slabs + (size_t)blockIdx.x * 64 * 224;
        const int li = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                out[(size_t)row * 224 + kh_w * 32 + li] = acc[mb][r];
            }
    }
}

// dw[co][kh][kw][c] (OHWI, 4 channels) = sum over the workgroups' partials, in order
__global__ __launch_bounds__(256) void k_stem_wgrad_reduce(const float* __restrict__ slabs, int nslab, float* __restrict__ dw) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 64 * 196) return;
    const int co = idx / 196, rem = idx - co * 196;
    const int kh = rem / 28, r2 = rem - kh * 28;
    const float* p = slabs + (size_t)co * 224 + kh * 32 + r2;
    float s = 0.f;
    int b = 0;
    for (; b + 3 < nslab; b += 4) {
        const float v0 = p[(size_t)b * 14336], v1 = p[(size_t)(b + 1) * 14336], v2 = p[(size_t)(b + 2) * 14336],
                    v3 = p[(size_t)(b + 3) * 14336];
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; b < nslab; ++b) s += p[(size_t)b * 14336];
    dw[idx] = s;
}

// the geometry this file covers (conv math bf16x6 only: the exact-fp32 mode keeps the fp32 MFMA kernels)
bool stem_geometry(const ConvArgs& a) {
    static int on = -1;
    if (on < 0) on = (getenv("ISWM_STEM") && getenv("ISWM_STEM")[0] == '0') ? 0 : 1;       // tuning switch: 0 = the generic kernels
    return on && a.Cin == 4 && a.Cout == 64 && a.KH == 7 && a.KW == 7 && a.stride == 2 && a.pad == 3 && a.dil == 1 && a.ldx == 4 &&
           a.ldy % 4 == 0 && a.ldy >= 64 && (long long)a.N * a.H * a.W < (1ll << 29) && a.Ho == (a.H - 1) / 2 + 1 &&
           a.Wo == (a.W - 1) / 2 + 1 && a.Wo > 16;
}

int stem_tile_rows() { return STEM_TILE; }

bool launch_stem_fwd(ConvArgs a, hipStream_t s) {
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    if (!stem_geometry(a)) return false;
    a.MT = (a.M + STEM_TILE - 1) / STEM_TILE;
    const int wgs = (a.MT + 7) / 8;
    hipLaunchKernelGGL((k_stem_fwd<STEM_RB>), dim3(wgs < ncu ? wgs : ncu), dim3(512), 0, s, a);
    return true;
}

static int stem_wgrad_grid(const ConvArgs& a, int* per) {
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    const long long M = (long long)a.N * a.Ho * a.Wo;
    int grid = (int)((M + SW_KS - 1) / SW_KS < ncu ? (M + SW_KS - 1) / SW_KS : ncu);
    if (grid < 1) grid = 1;
    *per = (int)(((M + grid - 1) / grid + SW_KS - 1) / SW_KS * SW_KS);
    return (int)((M + *per - 1) / *per);
}

static bool stem_wgrad_on() {
    static int on = -1;
    if (on < 0) on = (getenv("ISWM_STEM_WG") && getenv("ISWM_STEM_WG")[0] == '0') ? 0 : 1;      // tuning switch
    return on != 0;
}

size_t stem_wgrad_workspace(const ConvArgs& a) {
    if (!stem_wgrad_on()) return 0;          // 0: the generic weight gradient runs (and sizes its own workspace)
    int per;
    return (size_t)stem_wgrad_grid(a, &per) * 64 * 224 * sizeof(float);
}

// a.x: NHWC4 fp32 image, a.y: dy [M][ldy] fp32; dw: OHWI [64][7][7][4]; workspace >= stem_wgrad_workspace
bool launch_stem_wgrad(ConvArgs a, float* dw, float* workspace, hipStream_t s) {
    if (!stem_geometry(a) || !stem_wgrad_on()) return false;
    int per;
    const int grid = stem_wgrad_grid(a, &per);
    a.M = a.N * a.Ho * a.Wo;
    hipLaunchKernelGGL(k_stem_wgrad, dim3(grid), dim3(512), 0, s, a, workspace, per);
    hipLaunchKernelGGL(k_stem_wgrad_reduce, dim3((64 * 196 + 255) / 256), dim3(256), 0, s, workspace, grid, dw);
    return true;
}

}  // namespace iswm
