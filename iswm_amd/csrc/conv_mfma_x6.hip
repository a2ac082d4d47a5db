// fp32-accurate implicit-GEMM convolution on the BF16 matrix cores ("bf16x6").
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate.  An fp32 value splits EXACTLY into three
// bf16 pieces by truncation (24-bit significand = 8 + 8 + 8 bits):  x = hi + mid + lo.  A product then is
//   a*b = ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm   + O(2^-24 |a*b|)
// i.e. six bf16 MFMAs (products of bf16 pairs are exact in fp32, accumulation is fp32) reproduce the fp32
// product to ~1.2e-7 relative -- the rounding level of fp32 itself -- at 16/6 = 2.7x the fp32 MFMA rate.
// The dropped terms (am*bl, al*bm, al*bl) are below 2^-24 of the product.
//
// Structure = conv_mfma_u.hip (tap-uniform K loop, pointer/stride gather, padding-tap culling), except:
//   * the loader splits each staged float4 into three packed bf16x4 (4 VALU ops per element + 3 v_perm per
//     pair) and writes three bf16 planes to LDS: [plane][row][32 k] with an 80-byte row pitch, which makes
//     the 16-byte fragment reads of v_mfma_f32_32x32x16_bf16 (lane r = l&31, h = l>>5 reads k = 8h..8h+7 of
//     row r) bank-conflict free (5r mod 16 is a permutation);
//   * LDS holds ONE K chunk (3 planes x 2 operands x 128 rows x 80 B = 61 KB -> 2 workgroups per CU); the next
//     chunk is prefetched into registers while the current one is multiplied (two barriers per chunk).
#include <stdlib.h>

#include "conv_common.h"

namespace iswm {

static __device__ __attribute__((aligned(16))) float g_zero_row_x[64];

constexpr int X6_PITCH = 80;   // bytes per LDS row of one bf16 plane (32 k x 2 B + 16 B pad)

// DGRAD == false: forward.  rows = output pixels, A = x gathered per tap, B = OHWI weights [cout][(tap, cin)].
// DGRAD == true : data gradient.  rows = INPUT pixels, A = dy gathered per tap (a.x = dy, pitch a.ldx),
//                 B = TRANSPOSED weights [cin][(tap, cout)] (iswm_transpose_weights), output a.y = dx.
// In both cases the GEMM K axis (tap, gathered channel) is contiguous in memory for A and B.
// BD == true ("B direct"): the weight operand was split and laid out in MFMA fragment order ahead of time
//   (k_pack_weights_x6: [col block of 32][k block of 16][plane][lane] x 16 B), so each wave loads its B fragments
//   straight from global/L2 into registers -- no split arithmetic, LDS write or LDS read for B.  a.w = packed.
// NP: bf16 planes per operand.  3 = the exact split, six MFMAs per product (fp32-grade, "bf16x6");  1 = operands rounded
//     to nearest bf16, one MFMA per product, fp32 accumulation ("bf16": mixed precision, packed weights only).
template <int BM, int BN, bool DGRAD, bool BD, int NP = 3>
__global__ __launch_bounds__(256, (BM == 128 && BN == 128) ? 2 : (BM == 128 ? 3 : 4)) void k_conv_x6(const ConvArgs a) {
    static_assert(NP == 3 || (NP == 1 && BD), "the one-plane arithmetic exists for packed weights only");
    const int GC = DGRAD ? a.Cout : a.Cin;     // channels of the gathered operand (per tap)
    const int NC = DGRAD ? a.Cin : a.Cout;     // output columns
    constexpr int MB = BM / 64, NB = BN / 64, AR = BM / 32, BR = BN / 32;
    constexpr int PLANE_A = BM * X6_PITCH, PLANE_B = BN * X6_PITCH;          // bytes
    __shared__ __attribute__((aligned(16))) unsigned char smem[NP * (PLANE_A + (BD ? 0 : PLANE_B))];
    unsigned char* As = smem;                  // [NP][BM][80 B]
    unsigned char* Bs = smem + NP * PLANE_A;   // [NP][BN][80 B]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    int mt = L / a.NT;
    const int nt = L - mt * a.NT;
    if (DGRAD && a.stride == 2 && a.nsplit == 0) {
        // parity-ordered rows (below): the four parity classes occupy consecutive quarters of the M tiles and do very
        // different amounts of work (a 1x1 reaches one class only), while xcd_remap hands each XCD one contiguous run
        // of tiles -- deal the tiles out so every run holds all four quarters:  mt = 4*idx + k  ->  quarter k, slot idx
        const int qn = a.MT >> 2, rem = a.MT & 3, k = mt & 3, idx = mt >> 2;
        mt = k * qn + (k < rem ? k : rem) + idx;
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int q = t & 7, r0 = t >> 3;

    // row -> pixel of the tensor the rows live in (fwd: output Ho x Wo; dgrad: input H x W)
    const int RH = DGRAD ? a.H : a.Ho, RW = DGRAD ? a.W : a.Wo;
    const int GH = DGRAD ? a.Ho : a.H, GW = DGRAD ? a.Wo : a.W;     // gathered tensor dims
    const bool par = DGRAD && a.stride == 2 && a.nsplit == 0;     // a.nsplit != 0: tuning switch, row-major rows
    __shared__ int rowpix[DGRAD ? BM : 1];       // parity order: output pixel of each tile row
    if (DGRAD && par && t < BM) {
        const int m = m0 + t;
        int n = 0, rh = 0, rw = 0;
        if (m < a.M) x6_row_pixel(m, a.N, RH, RW, true, n, rh, rw);
        rowpix[DGRAD ? t : 0] = (n * RH + rh) * RW + rw;
    }
    int ihb[AR], iwb[AR], pb[AR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        int m = m0 + r0 + 32 * j;
        if (m < a.M) {
            int n, rh, rw;
            x6_row_pixel(m, a.N, RH, RW, par, n, rh, rw);
            ihb[j] = DGRAD ? rh + a.pad : rh * a.stride - a.pad;
            iwb[j] = DGRAD ? rw + a.pad : rw * a.stride - a.pad;
            pb[j] = n * GH * GW;
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
        wok[j] = n < NC;
        wbase[j] = a.w + (size_t)(wok[j] ? n : 0) * a.Ktot + q * 4;
    }
    const int taps = a.KH * a.KW;
    const int nCC = GC >> 5;
    // BD: this wave's packed fragments: column block (n0 + wn*BN/2)/32 + nb, 192 uint4 per (column block, k16)
    const uint4* wpk = reinterpret_cast<const uint4*>(a.w) + (size_t)((n0 + wn * (BN / 2)) >> 5) * (a.Ktot >> 4) * (64 * NP) + lane;
    const size_t wpk_nb = (size_t)(a.Ktot >> 4) * (64 * NP);

    const float* aptr[AR];
    int astep[AR];
    const float* bptr[BR];
    int bstep[BR];
    auto setup_tap = [&](int tap) -> bool {
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        const int dh = kh * a.dil, dw = kw * a.dil;
        int any = 0;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            int gh, gw;
            bool ok;
            if (DGRAD) {
                int th = ihb[j] - dh, tw = iwb[j] - dw;
                gh = th;
                gw = tw;
                ok = th >= 0 && tw >= 0;
                if (a.stride != 1) {
                    gh = th / a.stride;
                    gw = tw / a.stride;
                    ok = ok && (gh * a.stride == th) && (gw * a.stride == tw);
                }
                ok = ok && gh < GH && gw < GW;
            } else {
                gh = ihb[j] + dh;
                gw = iwb[j] + dw;
                ok = (unsigned)gh < (unsigned)GH && (unsigned)gw < (unsigned)GW;
            }
            aptr[j] = ok ? a.x + (size_t)(pb[j] + gh * GW + gw) * a.ldx + q * 4 : g_zero_row_x + q * 4;
            astep[j] = ok ? 32 : 0;
            any |= ok;
        }
        if constexpr (!BD) {
#pragma unroll
            for (int j = 0; j < BR; ++j) {
                bptr[j] = wok[j] ? wbase[j] + (size_t)tap * GC : g_zero_row_x + q * 4;
                bstep[j] = wok[j] ? 32 : 0;
            }
        }
        return __syncthreads_or(any) != 0;
    };
    int tap = -1, cc = nCC - 1;
    auto next = [&]() -> bool {
        if (++cc < nCC) return true;
        cc = 0;
        do {
            if (++tap >= taps) return false;
        } while (!setup_tap(tap));
        return true;
    };

    float4 ra[AR], rb[BR];
    auto gload = [&](float4 (&ra)[AR]) {
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            ra[j] = ldg4(aptr[j]);
            aptr[j] += astep[j];
        }
        if constexpr (!BD) {
#pragma unroll
            for (int j = 0; j < BR; ++j) {
                rb[j] = ldg4(bptr[j]);
                bptr[j] += bstep[j];
            }
        }
    };
    uint4 bfr[2][NB][NP];  // BD: [k half][column block][plane]
    auto bload = [&](int k16) {   // fragments of the chunk whose first k16 block is k16 = (tap * nCC + cc) * 2
        const uint4* p = wpk + (size_t)k16 * (64 * NP);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) bfr[ks][nb][pl] = p[nb * wpk_nb + (ks * NP + pl) * 64];
    };
    auto lstore = [&](const float4 (&ra)[AR]) {   // split into bf16 planes and write 8 B per plane per row
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            unsigned char* p = As + (r0 + 32 * j) * X6_PITCH + q * 8;
            if constexpr (NP == 1) {
                *reinterpret_cast<uint2*>(p) = round_bf16x4(ra[j]);
            } else {
                uint2 h, m, l;
                split3(ra[j], h, m, l);
                *reinterpret_cast<uint2*>(p) = h;
                *reinterpret_cast<uint2*>(p + PLANE_A) = m;
                *reinterpret_cast<uint2*>(p + 2 * PLANE_A) = l;
            }
        }
        if constexpr (!BD)
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            uint2 h, m, l;
            split3(rb[j], h, m, l);
            unsigned char* p = Bs + (r0 + 32 * j) * X6_PITCH + q * 8;
            *reinterpret_cast<uint2*>(p) = h;
            *reinterpret_cast<uint2*>(p + PLANE_B) = m;
            *reinterpret_cast<uint2*>(p + 2 * PLANE_B) = l;
        }
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto mfma_phase = [&]() {
        // fragment base: row (wave tile row + lane&31), k offset 8*(lane>>5) elements = 16 B
        const unsigned char* Ab = As + (wm * (BM / 2) + li) * X6_PITCH + lh * 16;
        const unsigned char* Bb = Bs + (wn * (BN / 2) + li) * X6_PITCH + lh * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 ah[MB], am[MB], al[MB], bh[NB], bm[NB], bl[NB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const unsigned char* p = Ab + mb * 32 * X6_PITCH + ks * 32;
                ah[mb] = *reinterpret_cast<const uint4*>(p);
                if constexpr (NP == 3) {
                    am[mb] = *reinterpret_cast<const uint4*>(p + PLANE_A);
                    al[mb] = *reinterpret_cast<const uint4*>(p + 2 * PLANE_A);
                }
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                if constexpr (BD) {
                    bh[nb] = bfr[ks][nb][0];
                    if constexpr (NP == 3) {
                        bm[nb] = bfr[ks][nb][1];
                        bl[nb] = bfr[ks][nb][2];
                    }
                } else {
                    const unsigned char* p = Bb + nb * 32 * X6_PITCH + ks * 32;
                    bh[nb] = *reinterpret_cast<const uint4*>(p);
                    bm[nb] = *reinterpret_cast<const uint4*>(p + PLANE_B);
                    bl[nb] = *reinterpret_cast<const uint4*>(p + 2 * PLANE_B);
                }
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f32x16 c = acc[mb][nb];
                    if constexpr (NP == 3) {
                        c = mfma_bf16(al[mb], bh[nb], c);     // smallest terms first
                        c = mfma_bf16(ah[mb], bl[nb], c);
                        c = mfma_bf16(am[mb], bm[nb], c);
                        c = mfma_bf16(am[mb], bh[nb], c);
                        c = mfma_bf16(ah[mb], bm[nb], c);
                    }
                    c = mfma_bf16(ah[mb], bh[nb], c);
                    acc[mb][nb] = c;
                }
        }
    };
    {
        bool more = next();
        // a tile no tap reaches (odd-parity tiles of a strided 1x1 data gradient) adds nothing: leave dx untouched
        if (DGRAD && a.accumulate && !more) return;
        if (more) gload(ra);
        while (more) {
            // B fragments of this chunk straight from L2 (issued before the split so they land behind it); fetching
            // them a chunk ahead or keeping two activation chunks in flight measured no faster
            if constexpr (BD) bload((tap * nCC + cc) * 2);
            lstore(ra);
            __syncthreads();
            const bool more2 = next();
            if (more2) gload(ra);
            mfma_phase();
            __syncthreads();
            more = more2;
        }
    }

    // ---- epilogue: identical C/D map to the fp32 kernels
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = n0 + wn * (BN / 2) + nb * 32 + li;
        const bool cok = col < NC;
        const float bv = (!DGRAD && a.bias != nullptr && cok) ? a.bias[col] : 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wm * (BM / 2) + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int row = m0 + lr;
                if (cok && row < a.M) {
                    const int pix = (DGRAD && par) ? rowpix[DGRAD ? lr : 0] : row;
                    float* o = &a.y[(size_t)pix * a.ldy + col];
                    *o = (DGRAD && a.accumulate) ? *o + acc[mb][nb][r] : acc[mb][nb][r] + bv;
                }
            }
    }
    if (!DGRAD && a.stats != nullptr) {
        float* red = reinterpret_cast<float*>(smem);  // [4][BN]
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

// Tile choice for the bf16x6 forward / data-gradient kernels, from forced-tile sweeps over every
// ResNet-101/DeepLabV3+ geometry (profiles/r01_x6_tile_sweep.txt): 128x64 runs 3 blocks per CU
// (168 VGPRs) and wins almost everywhere; 64x64 (4 blocks per CU) wins when there are too few
// 128x64 tiles to give every CU three, and for write-dominated 1x1 data gradients (small K, wide
// output); 128x128 (2 blocks per CU) only pays for very large M.
void conv_pick_tile_x6(int64_t M, int cols, int K, bool dgrad, bool pointwise, int* bm, int* bn) {
    if (const char* e = getenv("ISWM_TILE")) {          // tuning override: "128x128" | "128x64" | "64x64"
        int m = 0, n = 0;
        if (sscanf(e, "%dx%d", &m, &n) == 2 && (m == 128 || m == 64) && (n == 128 || n == 64) && !(m == 64 && n == 128)) {
            if (n == 128 && (cols <= 64 || (cols % 128 != 0 && cols % 128 <= 64))) n = 64;
            *bm = m; *bn = n;
            return;
        }
    }
    const int64_t mt128 = (M + 127) / 128;
    const int64_t tiles64 = mt128 * ((cols + 63) / 64);
    if (tiles64 < 384 || (dgrad && pointwise && 2 * K <= cols)) {
        *bm = 64; *bn = 64;
    } else if (!dgrad && cols % 128 == 0 && M >= 131072 && K >= 1024) {
        *bm = 128; *bn = 128;
    } else {
        *bm = 128; *bn = 64;
    }
}

bool launch_conv_fwd_x6(ConvArgs a, hipStream_t s, int bm, int bn) {
    if (a.Cin % 32 != 0) return false;
    a.MT = (a.M + bm - 1) / bm;
    a.NT = (a.Cout + bn - 1) / bn;
    dim3 grid(a.MT * a.NT), blk(256);
    if (bm == 128 && bn == 128) hipLaunchKernelGGL((k_conv_x6<128, 128, false, false>), grid, blk, 0, s, a);
    else if (bm == 128) hipLaunchKernelGGL((k_conv_x6<128, 64, false, false>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((k_conv_x6<64, 64, false, false>), grid, blk, 0, s, a);
    return true;
}

// packed-weight ("B direct") launchers: a.w = k_pack_weights_x6 output.  Tiles: 128x64 or 64x64.
bool launch_conv_x6_pk(ConvArgs a, hipStream_t s, bool dgrad, int bm, int planes) {
    static int parity = -1;
    if (parity < 0) {
        const char* e = getenv("ISWM_X6_PARITY");
        parity = (e && e[0] == '0') ? 0 : 1;
    }
    a.nsplit = parity ? 0 : 1;
    const int nc = dgrad ? a.Cin : a.Cout;
    a.MT = (a.M + bm - 1) / bm;
    a.NT = (nc + 63) / 64;
    dim3 grid(a.MT * a.NT), blk(256);
    if (planes == 1) {
        if (dgrad) {
            if (bm == 128) hipLaunchKernelGGL((k_conv_x6<128, 64, true, true, 1>), grid, blk, 0, s, a);
            else hipLaunchKernelGGL((k_conv_x6<64, 64, true, true, 1>), grid, blk, 0, s, a);
        } else {
            if (bm == 128) hipLaunchKernelGGL((k_conv_x6<128, 64, false, true, 1>), grid, blk, 0, s, a);
            else hipLaunchKernelGGL((k_conv_x6<64, 64, false, true, 1>), grid, blk, 0, s, a);
        }
        return true;
    }
    if (dgrad) {
        if (bm == 128) hipLaunchKernelGGL((k_conv_x6<128, 64, true, true>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((k_conv_x6<64, 64, true, true>), grid, blk, 0, s, a);
    } else {
        if (bm == 128) hipLaunchKernelGGL((k_conv_x6<128, 64, false, true>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((k_conv_x6<64, 64, false, true>), grid, blk, 0, s, a);
    }
    return true;
}

// Split + fragment-order packing of a conv weight for the B-direct kernels.
//   fwd  : column = cout, k = (tap, cin)   from w[cout][tap][cin]
//   dgrad: column = cin,  k = (tap, cout)  from the same tensor (implicit transpose)
// packed[((cb * K16 + k16) * 3 + plane) * 64 + lane] (uint4) holds, for column cb*32 + (lane & 31), the 8 bf16 of
// plane {hi, mid, lo} at k = k16*16 + 8*(lane >> 5) .. +7.  Columns >= NC are zero; cb runs to ceil(NC/64)*2.
template <bool DGRAD, int NP>
__device__ __forceinline__ void pack_weights_body(const float* __restrict__ w, uint4* __restrict__ packed, int Cout, int T,
                                                  int Cin, int K16, int idx) {
    const int lane = idx & 63, f = idx >> 6;
    const int k16 = f % K16, cb = f / K16;
    const int col = cb * 32 + (lane & 31), k0 = k16 * 16 + 8 * (lane >> 5);
    const int NC = DGRAD ? Cin : Cout, GC = DGRAD ? Cout : Cin;
    float v[8];
    const int tap = k0 / GC, g0 = k0 - tap * GC;      // 8 consecutive k never straddle a tap (GC % 32 == 0)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float x = 0.f;
        if (col < NC) x = DGRAD ? w[((size_t)(g0 + i) * T + tap) * Cin + col] : w[((size_t)col * T + tap) * Cin + g0 + i];
        v[i] = x;
    }
    uint4* o = packed + (size_t)f * (64 * NP) + lane;
    if constexpr (NP == 1) {
        const uint2 r0 = round_bf16x4(make_float4(v[0], v[1], v[2], v[3])), r1 = round_bf16x4(make_float4(v[4], v[5], v[6], v[7]));
        o[0] = make_uint4(r0.x, r0.y, r1.x, r1.y);
    } else {
        uint2 h0, m0, l0, h1, m1, l1;
        split3(make_float4(v[0], v[1], v[2], v[3]), h0, m0, l0);
        split3(make_float4(v[4], v[5], v[6], v[7]), h1, m1, l1);
        o[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        o[64] = make_uint4(m0.x, m0.y, m1.x, m1.y);
        o[128] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

template <bool DGRAD, int NP>
__global__ __launch_bounds__(256) void k_pack_weights_x6(const float* __restrict__ w, uint4* __restrict__ packed,
                                                         int Cout, int T, int Cin, int K16, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    pack_weights_body<DGRAD, NP>(w, packed, Cout, T, Cin, K16, idx);
}

// All conv weights of a model in ONE launch (a training step otherwise issues ~220 tiny pack kernels, each mostly
// launch latency): jobs[] is sorted by first_block; a workgroup finds its job by binary search.
struct PackJob {             // mirrors iswm_pack_job
    const float* w;
    void* packed;
    int Cout, T, Cin, kind;
    int first_block, reserved;
};

template <int NP>
__global__ __launch_bounds__(256) void k_pack_weights_batch(const PackJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    const int b = blockIdx.x;
    while (lo < hi) {                       // last job with first_block <= b
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= b) lo = mid;
        else hi = mid - 1;
    }
    const PackJob j = jobs[lo];
    const int idx = (b - j.first_block) * 256 + threadIdx.x;
    if (j.kind >= 2) {          // kinds 2 / 3: the 16x16x32 fragment order of the planes kernels (conv_mfma_pl2.hip)
        const bool dg = j.kind == 3;
        const int NC = dg ? j.Cin : j.Cout, GC = dg ? j.Cout : j.Cin;
        const int cbs = ((NC + 127) / 128) * 8, K32 = j.T * GC / 32;
        if (idx >= cbs * K32 * 64) return;
        if (dg) pack_weights_pl2_body<true, NP>(j.w, (uint4*)j.packed, j.Cout, j.T, j.Cin, K32, idx);
        else pack_weights_pl2_body<false, NP>(j.w, (uint4*)j.packed, j.Cout, j.T, j.Cin, K32, idx);
        return;
    }
    const int NC = j.kind ? j.Cin : j.Cout, GC = j.kind ? j.Cout : j.Cin;
    const int cbs = ((NC + 63) / 64) * 2, K16 = j.T * GC / 16;
    if (idx >= cbs * K16 * 64) return;
    if (j.kind) pack_weights_body<true, NP>(j.w, (uint4*)j.packed, j.Cout, j.T, j.Cin, K16, idx);
    else pack_weights_body<false, NP>(j.w, (uint4*)j.packed, j.Cout, j.T, j.Cin, K16, idx);
}

void launch_pack_weights_batch(const void* jobs_dev, int njobs, int total_blocks, int planes, hipStream_t s) {
    if (planes == 1) hipLaunchKernelGGL(k_pack_weights_batch<1>, dim3(total_blocks), dim3(256), 0, s, (const PackJob*)jobs_dev, njobs);
    else hipLaunchKernelGGL(k_pack_weights_batch<3>, dim3(total_blocks), dim3(256), 0, s, (const PackJob*)jobs_dev, njobs);
}

int pack_job_blocks_x6(int Cout, int T, int Cin, bool dgrad) {
    const int NC = dgrad ? Cin : Cout, GC = dgrad ? Cout : Cin;
    const long long total = (long long)((NC + 63) / 64) * 2 * (T * GC / 16) * 64;
    return (int)((total + 255) / 256);
}

size_t packed_weight_bytes_x6(int Cout, int T, int Cin, bool dgrad, int planes) {
    const int NC = dgrad ? Cin : Cout, GC = dgrad ? Cout : Cin;
    const size_t cbs = (size_t)((NC + 63) / 64) * 2, K16 = (size_t)T * GC / 16;
    return cbs * K16 * 64 * planes * sizeof(uint4);
}

void launch_pack_weights_x6(const float* w, void* packed, int Cout, int T, int Cin, bool dgrad, int planes, hipStream_t s) {
    const int NC = dgrad ? Cin : Cout, GC = dgrad ? Cout : Cin;
    const int cbs = ((NC + 63) / 64) * 2, K16 = T * GC / 16;
    const int total = cbs * K16 * 64;
    dim3 grid((total + 255) / 256), blk(256);
    uint4* o = (uint4*)packed;
    if (planes == 1) {
        if (dgrad) hipLaunchKernelGGL((k_pack_weights_x6<true, 1>), grid, blk, 0, s, w, o, Cout, T, Cin, K16, total);
        else hipLaunchKernelGGL((k_pack_weights_x6<false, 1>), grid, blk, 0, s, w, o, Cout, T, Cin, K16, total);
    } else {
        if (dgrad) hipLaunchKernelGGL((k_pack_weights_x6<true, 3>), grid, blk, 0, s, w, o, Cout, T, Cin, K16, total);
        else hipLaunchKernelGGL((k_pack_weights_x6<false, 3>), grid, blk, 0, s, w, o, Cout, T, Cin, K16, total);
    }
}

// a: as prepared by iswm_conv2d_dgrad (a.x = dy with pitch a.ldx, a.y = dx with pitch a.ldy, a.M = N*H*W),
// a.w = TRANSPOSED weights [Cin][taps][Cout]
bool launch_conv_dgrad_x6(ConvArgs a, hipStream_t s, int bm, int bn) {
    if (a.Cout % 32 != 0) return false;
    a.MT = (a.M + bm - 1) / bm;
    a.NT = (a.Cin + bn - 1) / bn;
    dim3 grid(a.MT * a.NT), blk(256);
    if (bm == 128 && bn == 128) hipLaunchKernelGGL((k_conv_x6<128, 128, true, false>), grid, blk, 0, s, a);
    else if (bm == 128) hipLaunchKernelGGL((k_conv_x6<128, 64, true, false>), grid, blk, 0, s, a);
    else hipLaunchKernelGGL((k_conv_x6<64, 64, true, false>), grid, blk, 0, s, a);
    return true;
}

// wt[ci][t][co] = w[co][t][ci]: per tap a [Cout x Cin] -> [Cin x Cout] transpose through a 32x33 LDS tile
__global__ __launch_bounds__(256) void k_transpose_ohwi(const float* __restrict__ w, float* __restrict__ wt, int Cout,
                                                        int T, int Cin) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z, ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        int co = co0 + r, ci = ci0 + tx;
        tile[r][tx] = (co < Cout && ci < Cin) ? w[((size_t)co * T + t) * Cin + ci] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int ci = ci0 + r, co = co0 + tx;
        if (ci < Cin && co < Cout) wt[((size_t)ci * T + t) * Cout + co] = tile[tx][r];
    }
}

void launch_transpose_ohwi(const float* w, float* wt, int Cout, int T, int Cin, hipStream_t s) {
    dim3 grid((Cin + 31) / 32, (Cout + 31) / 32, T);
    hipLaunchKernelGGL(k_transpose_ohwi, grid, dim3(256), 0, s, w, wt, Cout, T, Cin);
}

}  // namespace iswm
