// Tap-uniform fast path of the implicit-GEMM convolution (forward and data gradient).
//
// When the channel count of the gathered operand is a multiple of the K chunk (32) -- every conv of
// the network except the 3-channel stem, the 304-channel decoder input and the 48/16-wide projections --
// a K chunk never straddles two filter taps, so the K loop becomes (tap, channel-chunk):
//   * per TAP each thread derives its gather pointers once (ih/iw, bounds, 64-bit address); per CHUNK it
//     only adds a constant stride -- no integer division and no per-row predicate in the inner loop.
//     Rows that fall into the zero padding point at a device zero buffer with stride 0, so the loads are
//     unconditional (no exec-mask branches) and land in L1;
//   * a tap whose gather is out of bounds for EVERY row of the tile (the ASPP rates on a 33x33 map: 36-60 %
//     of the taps) is skipped with one block-wide vote -- the all-padding taps are never multiplied;
//   * the lower per-chunk VALU cost makes 64x64 tiles with 4 workgroups per CU viable, which removes
//     most of the tile-quantisation loss on the 33x33 stages (137 row tiles x 2 column tiles = 274
//     workgroups of 128x128 for 512 slots).
// MFMA scheme, LDS layouts and epilogues are those of conv_mfma.hip.
#include <stdlib.h>

#include "conv_common.h"

namespace iswm {

__device__ __attribute__((aligned(16))) float g_zero_row[64];   // zero-initialised: target of padded rows

template <int BM, int BN>
struct TileCfg {
    static constexpr int MB = BM / 64;        // 32x32 MFMA tiles per wave along M
    static constexpr int NB = BN / 64;
    static constexpr int AR = BM / 32;        // A rows staged per thread
    static constexpr int BR = BN / 32;
    static constexpr int OCC = (BM == 64 && BN == 64) ? 4 : 2;   // workgroups per CU (LDS / VGPR budget)
};

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(256, (TileCfg<BM, BN>::OCC)) void k_conv_fwd_u(const ConvArgs a) {
    using T = TileCfg<BM, BN>;
    constexpr int MB = T::MB, NB = T::NB, AR = T::AR, BR = T::BR;
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
    int ihb[AR], iwb[AR], pb[AR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
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
    const float* wbase[BR];
    bool wok[BR];
#pragma unroll
    for (int j = 0; j < BR; ++j) {
        int n = n0 + r0 + 32 * j;
        wok[j] = n < a.Cout;
        wbase[j] = a.w + (size_t)(wok[j] ? n : 0) * a.Ktot + q * 4;
    }
    const int taps = a.KH * a.KW;
    const int nCC = a.Cin >> 5;

    const float* aptr[AR];
    int astep[AR];
    const float* bptr[BR];
    int bstep[BR];
    // derive this thread's gather pointers for one tap; returns whether ANY row of the tile is in bounds
    auto setup_tap = [&](int tap) -> bool {
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        const int dh = kh * a.dil, dw = kw * a.dil;
        int any = 0;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            int ih = ihb[j] + dh, iw = iwb[j] + dw;
            bool ok = (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            aptr[j] = ok ? a.x + (size_t)(pb[j] + ih * a.W + iw) * a.ldx + q * 4 : g_zero_row + q * 4;
            astep[j] = ok ? 32 : 0;
            any |= ok;
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            bptr[j] = wok[j] ? wbase[j] + (size_t)tap * a.Cin : g_zero_row + q * 4;
            bstep[j] = wok[j] ? 32 : 0;
        }
        return __syncthreads_or(any) != 0;
    };
    int tap = -1, cc = nCC - 1;
    auto next = [&]() -> bool {   // advance to the next (tap, channel chunk) that has work; block-uniform
        if (++cc < nCC) return true;
        cc = 0;
        do {
            if (++tap >= taps) return false;
        } while (!setup_tap(tap));
        return true;
    };

    float4 ra[AR], rb[BR];
    auto gload = [&]() {
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            ra[j] = ldg4(aptr[j]);
            aptr[j] += astep[j];
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            rb[j] = ldg4(bptr[j]);
            bptr[j] += bstep[j];
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < AR; ++j)
            *reinterpret_cast<float4*>(&As[(buf * BM + r0 + 32 * j) * KC_PITCH + q * 4]) = ra[j];
#pragma unroll
        for (int j = 0; j < BR; ++j)
            *reinterpret_cast<float4*>(&Bs[(buf * BN + r0 + 32 * j) * KC_PITCH + q * 4]) = rb[j];
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bool more = next();
    if (more) {
        gload();
        lstore(0);
    }
    __syncthreads();
    int cur = 0;
    while (more) {
        const bool more2 = next();
        if (more2) gload();
        const float* Ab = &As[(cur * BM + wm * (BM / 2) + li) * KC_PITCH + lh * 4];
        const float* Bb = &Bs[(cur * BN + wn * (BN / 2) + li) * KC_PITCH + lh * 4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float af[MB][4], bf[NB][4];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                *reinterpret_cast<float4*>(af[mb]) =
                    *reinterpret_cast<const float4*>(Ab + mb * 32 * KC_PITCH + g * 8);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                *reinterpret_cast<float4*>(bf[nb]) =
                    *reinterpret_cast<const float4*>(Bb + nb * 32 * KC_PITCH + g * 8);
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

    // ---- epilogue (as conv_mfma.hip): C/D map col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = n0 + wn * (BN / 2) + nb * 32 + li;
        const bool cok = col < a.Cout;
        const float bv = (a.bias != nullptr && cok) ? a.bias[col] : 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm * (BM / 2) + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (cok && row < a.M) a.y[(size_t)row * a.ldy + col] = acc[mb][nb][r] + bv;
            }
    }
    if (a.stats != nullptr) {
        // per-tile BatchNorm statistics {S_t, M2_t about the tile mean}; tiles are BM rows here, so the
        // host passes tile_rows = BM to iswm_bn_finalize
        float* red = smem;  // [4][BN]
        const int cnt = min(BM, a.M - m0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float s = 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
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
            float qv = 0.f;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = m0 + wm * (BM / 2) + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float dv = acc[mb][nb][r] - mean;
                    qv += row < a.M ? dv * dv : 0.f;
                }
            qv += __shfl_xor(qv, 32);
            if (lh == 0) red[(2 + wm) * BN + c] = qv;
        }
        __syncthreads();
        if (t < BN && n0 + t < a.Cout) {
            a.stats[(size_t)mt * a.Cout + n0 + t] = red[t] + red[BN + t];
            a.stats[(size_t)(a.MT + mt) * a.Cout + n0 + t] = red[2 * BN + t] + red[3 * BN + t];
        }
    }
}

// ------------------------------------------------------------------------------------------
// data gradient: rows are INPUT pixels, K = (tap, cout); a.x = dy (pitch a.ldx), a.y = dx (pitch a.ldy)
// ------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(256, (TileCfg<BM, BN>::OCC)) void k_conv_dgrad_u(const ConvArgs a) {
    using T = TileCfg<BM, BN>;
    constexpr int MB = T::MB, NB = T::NB, AR = T::AR;
    constexpr int BQ = BN / 4, BKR = 256 / BQ, BPASS = 32 / BKR;
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

    const int HW = a.H * a.W;
    int thb[AR], twb[AR], pb[AR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
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
    const int taps = a.KH * a.KW;
    const int nCC = a.Cout >> 5;
    const bool nok = n0 + bq * 4 < a.Cin;
    const size_t bchunk = (size_t)32 * taps * a.Cin;   // weight floats between consecutive cout chunks

    const float* aptr[AR];
    int astep[AR];
    const float* bptr[BPASS];
    auto setup_tap = [&](int tap) -> bool {
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        const int dh = kh * a.dil, dw = kw * a.dil;
        int any = 0;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            int th = thb[j] - dh, tw = twb[j] - dw;
            int oh = th, ow = tw;
            bool ok = th >= 0 && tw >= 0;
            if (a.stride != 1) {
                oh = th / a.stride;
                ow = tw / a.stride;
                ok = ok && (oh * a.stride == th) && (ow * a.stride == tw);
            }
            ok = ok && oh < a.Ho && ow < a.Wo;
            aptr[j] = ok ? a.x + (size_t)(pb[j] + oh * a.Wo + ow) * a.ldx + q * 4 : g_zero_row + q * 4;
            astep[j] = ok ? 32 : 0;
            any |= ok;
        }
#pragma unroll
        for (int j = 0; j < BPASS; ++j)
            bptr[j] = nok ? a.w + ((size_t)(bk0 + BKR * j) * taps + tap) * a.Cin + n0 + bq * 4 : g_zero_row;
        return __syncthreads_or(any) != 0;
    };
    const size_t bstep = nok ? bchunk : 0;
    int tap = -1, cc = nCC - 1;
    auto next = [&]() -> bool {
        if (++cc < nCC) return true;
        cc = 0;
        do {
            if (++tap >= taps) return false;
        } while (!setup_tap(tap));
        return true;
    };

    float4 ra[AR], rb[BPASS];
    auto gload = [&]() {
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            ra[j] = ldg4(aptr[j]);
            aptr[j] += astep[j];
        }
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            rb[j] = ldg4(bptr[j]);
            bptr[j] += bstep;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < AR; ++j)
            *reinterpret_cast<float4*>(&As[(buf * BM + r0 + 32 * j) * KC_PITCH + q * 4]) = ra[j];
#pragma unroll
        for (int j = 0; j < BPASS; ++j)
            *reinterpret_cast<float4*>(&Bs[(buf * 32 + bk0 + BKR * j) * BN + bq * 4]) = rb[j];
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bool more = next();
    if (more) {
        gload();
        lstore(0);
    }
    __syncthreads();
    int cur = 0;
    while (more) {
        const bool more2 = next();
        if (more2) gload();
        const float* Ab = &As[(cur * BM + wm * (BM / 2) + li) * KC_PITCH + lh * 4];
        const float* Bb = &Bs[(cur * 32 + lh * 4) * BN + wn * (BN / 2) + li];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float af[MB][4], bf[NB][4];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                *reinterpret_cast<float4*>(af[mb]) =
                    *reinterpret_cast<const float4*>(Ab + mb * 32 * KC_PITCH + g * 8);
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
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = n0 + wn * (BN / 2) + nb * 32 + li;
        const bool cok = col < a.Cin;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm * (BM / 2) + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (cok && row < a.M) {
                    float* o = &a.y[(size_t)row * a.ldy + col];
                    *o = a.accumulate ? *o + acc[mb][nb][r] : acc[mb][nb][r];
                }
            }
    }
}

// Tile choice: a CU works through ceil(tiles / 256) tiles (co-resident workgroups share its SIMDs), so
// the estimated time is rounds x tile area / intrinsic efficiency of the shape.
struct TilePick {
    int bm, bn;
};

static TilePick pick_tile(int64_t M, int cols) {
    if (const char* e = getenv("ISWM_TILE")) {          // tuning override: "128x128" | "128x64" | "64x64"
        int bm = 0, bn = 0;
        if (sscanf(e, "%dx%d", &bm, &bn) == 2 && (bm == 128 || bm == 64) && (bn == 128 || bn == 64) &&
            !(bm == 64 && bn == 128)) {
            if (bn == 128 && (cols <= 64 || (cols % 128 != 0 && cols % 128 <= 64))) bn = 64;
            return TilePick{bm, bn};
        }
    }
    struct Cand {
        int bm, bn;
        double eff;
    };
    const Cand cands[3] = {{128, 128, 1.00}, {128, 64, 0.92}, {64, 64, 0.80}};
    double best = 1e300;
    TilePick p{128, 128};
    for (const Cand& c : cands) {
        if (c.bn == 128 && (cols <= 64 || (cols % 128 != 0 && cols % 128 <= 64))) continue;
        int64_t tiles = ((M + c.bm - 1) / c.bm) * ((cols + c.bn - 1) / c.bn);
        double cost = (double)((tiles + 255) / 256) * c.bm * c.bn / c.eff;
        if (cost < best) {
            best = cost;
            p = {c.bm, c.bn};
        }
    }
    return p;
}

void conv_pick_tile(int64_t M, int cols, int* bm, int* bn) {
    TilePick p = pick_tile(M, cols);
    *bm = p.bm;
    *bn = p.bn;
}

int conv_fwd_tile_rows(int64_t M, int Cin, int Cout) {
    if (Cin % 32 != 0) return 128;          // general-K kernel: 128-row tiles
    return pick_tile(M, Cout).bm;
}

bool launch_conv_fwd_u(ConvArgs a, hipStream_t s) {
    if (a.Cin % 32 != 0) return false;
    TilePick p = pick_tile(a.M, a.Cout);
    a.MT = (a.M + p.bm - 1) / p.bm;
    a.NT = (a.Cout + p.bn - 1) / p.bn;
    dim3 grid(a.MT * a.NT), blk(256);
    if (p.bm == 128 && p.bn == 128) hipLaunchKernelGGL((k_conv_fwd_u<128, 128>), grid, blk, 0, s, a);
    else if (p.bm == 128) hipLaunchKernelGGL((k_conv_fwd_u<128, 64>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((k_conv_fwd_u<64, 64>), grid, blk, 0, s, a);
    return true;
}

bool launch_conv_dgrad_u(ConvArgs a, hipStream_t s) {
    if (a.Cout % 32 != 0) return false;
    TilePick p = pick_tile(a.M, a.Cin);
    a.MT = (a.M + p.bm - 1) / p.bm;
    a.NT = (a.Cin + p.bn - 1) / p.bn;
    dim3 grid(a.MT * a.NT), blk(256);
    if (p.bm == 128 && p.bn == 128) hipLaunchKernelGGL((k_conv_dgrad_u<128, 128>), grid, blk, 0, s, a);
    else if (p.bm == 128) hipLaunchKernelGGL((k_conv_dgrad_u<128, 64>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((k_conv_dgrad_u<64, 64>), grid, blk, 0, s, a);
    return true;
}

}  // namespace iswm
