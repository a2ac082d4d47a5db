// bf16x6 implicit-GEMM convolution over PRE-SPLIT activations ("planes").
//
// The per-tap kernels of conv_mfma_x6.hip gather fp32 activations, split every value into its three bf16
// pieces on the VALU and write three LDS planes with ds_write -- work that is repeated by every column tile
// and every tap that touches the value, and that the measurements show to serialise with the MFMAs.  Here
// the PRODUCER of an activation (the BatchNorm / pooling / resize pass that writes it) stores the exact
// split once:  three bf16 tensors [plane][pixel][ld] (hi, mid, lo; hi + mid + lo == the fp32 value,
// bit-exactly).  The convolution's A operand then goes HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4:
// no VGPRs, no VALU, no ds_write) and the loop is LDS fragment reads + MFMAs only; the weight operand comes
// pre-split in fragment order straight into registers as before (k_pack_weights_x6).
//
// LDS image of one K chunk (32 channels): [plane][row][64 B], rows unpadded because an LDS-DMA instruction
// writes 64 lanes x 16 B linearly.  Bank conflicts of the 16-byte fragment reads are removed on the SOURCE
// side: lane (row, g) of a DMA fetches the 16-byte channel group g ^ ((row >> 2) & 3) of its row, and the
// fragment read of channel group kg of row r takes slot kg ^ ((r >> 2) & 3).
// Two stages (double buffer), ONE barrier per chunk: chunk c+1 is in flight while chunk c is multiplied.
#include <stdlib.h>

#include "conv_common.h"

namespace iswm {

static __device__ __attribute__((aligned(128))) unsigned short g_zero_row_pl[64];   // 128 B of zeros

typedef __attribute__((address_space(3))) void* lds_vptr;

// One LDS-DMA instruction: every lane fetches 16 bytes from its own global address; the wave's 1 KB lands at LDS byte
// address lds_dst (wave-uniform) + 16 * lane.  Issued from inline asm so that hipcc does not count it: its waitcnt
// pass would otherwise drain the DMA (vmcnt(0)) in front of LDS reads it cannot prove disjoint, which serialises the
// prefetch with the multiply.  Completion is waited for explicitly (vmcnt) before the barrier that publishes the stage.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// DGRAD / tiles / packed weights: as k_conv_x6<.., BD = true>.  a.x = base of plane 0 (bf16), a.ldx = pixel pitch
// in bf16 elements, a.xps = plane stride in BYTES.
template <int BM, int BN, bool DGRAD, int NP, int ABL = 0>   // ABL: timing ablations (1: no activation DMA, 2: no weight loads, 3: neither)
__global__ __launch_bounds__(256, BM == 128 ? 3 : 4) void k_conv_pl(const ConvArgs a) {
    const int GC = DGRAD ? a.Cout : a.Cin;     // channels of the gathered operand (per tap)
    const int NC = DGRAD ? a.Cin : a.Cout;     // output columns
    constexpr int MB = BM / 64, NB = BN / 64;  // 32x32 blocks per wave (wave tile BM/2 x BN/2)
    constexpr int RB = BM / 64;                // 16-row DMA pieces per wave per plane
    constexpr int PLANE = BM * 64;             // bytes of one plane of one stage
    constexpr int STAGE = NP * PLANE;
    constexpr int ROWPIX = DGRAD ? BM * 4 : 0;
    constexpr int SMEM = 2 * STAGE + ROWPIX < 4 * BN * 4 ? 4 * BN * 4 : 2 * STAGE + ROWPIX;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];   // ONE LDS object (stages | rowpix)
    int* rowpix = reinterpret_cast<int*>(smem + 2 * STAGE);
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_vptr)smem;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    int mt = L / a.NT;
    const int nt = L - mt * a.NT;
    const bool par = DGRAD && a.stride == 2 && a.nsplit == 0;
    if (par) {      // deal the four parity quarters of the M tiles across the XCD runs (see k_conv_x6)
        const int qn = a.MT >> 2, rem = a.MT & 3, k = mt & 3, idx = mt >> 2;
        mt = k * qn + (k < rem ? k : rem) + idx;
    }
    const int m0 = mt * BM, n0 = nt * BN;

    const int RH = DGRAD ? a.H : a.Ho, RW = DGRAD ? a.W : a.Wo;
    const int GH = DGRAD ? a.Ho : a.H, GW = DGRAD ? a.Wo : a.W;
    if (DGRAD && par && t < BM) {
        const int m = m0 + t;
        int n = 0, rh = 0, rw = 0;
        if (m < a.M) x6_row_pixel(m, a.N, RH, RW, true, n, rh, rw);
        rowpix[DGRAD ? t : 0] = (n * RH + rh) * RW + rw;
    }
    // DMA role of this lane: row lr of a 16-row piece, 16-byte channel group gs of the 64-byte chunk row
    const int lr = lane >> 2;
    const int gs = (lane & 3) ^ ((lr >> 2) & 3);
    int ihb[RB], iwb[RB], pb[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int m = m0 + 16 * (wave + 4 * j) + lr;
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
    const int taps = a.KH * a.KW;
    const int nCC = GC >> 5;
    const uint4* wpk = reinterpret_cast<const uint4*>(a.w) + (size_t)((n0 + wn * (BN / 2)) >> 5) * (a.Ktot >> 4) * (64 * NP) + lane;
    const size_t wpk_nb = (size_t)(a.Ktot >> 4) * (64 * NP);
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x);
    const unsigned char* zrow = reinterpret_cast<const unsigned char*>(g_zero_row_pl) + gs * 16;

    const unsigned char* aptr[RB];
    int astep[RB];
    long long pst[RB];
    auto setup_tap = [&](int tap) -> bool {
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        const int dh = kh * a.dil, dw = kw * a.dil;
        int any = 0;
#pragma unroll
        for (int j = 0; j < RB; ++j) {
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
            aptr[j] = ok ? xb + ((size_t)(pb[j] + gh * GW + gw) * a.ldx) * 2 + gs * 16 : zrow;
            astep[j] = ok ? 64 : 0;
            pst[j] = ok ? a.xps : 0;
            any |= ok;
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

    // stage the chunk the pointers stand on (LDS-DMA, 16 B per lane, 1 KB per instruction) and advance them
    auto issueA = [&](int stage) {
#pragma unroll
        for (int j = 0; j < RB; ++j) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
                if (!(ABL & 1)) glds16(aptr[j] + p * pst[j], lds_base + stage * STAGE + p * PLANE + (wave + 4 * j) * 1024);
            aptr[j] += astep[j];
        }
    };
    struct BFrag {
        uint4 v[2][NB][NP];    // [k half][column block][plane]
    };
    auto bload = [&](BFrag& b, int k16) {
        const uint4* p = wpk + (size_t)k16 * (64 * NP);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    if (!(ABL & 2)) b.v[ks][nb][pl] = p[nb * wpk_nb + (ks * NP + pl) * 64];
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fsw = (li >> 2) & 3;
    auto compute = [&](int stage, const BFrag& b) {
        const unsigned char* Ab = smem + stage * STAGE + (wm * (BM / 2) + li) * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int off = ((2 * ks + lh) ^ fsw) * 16;
            uint4 ah[MB], am[MB], al[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const unsigned char* p = Ab + mb * 32 * 64 + off;
                ah[mb] = *reinterpret_cast<const uint4*>(p);
                if constexpr (NP == 3) {
                    am[mb] = *reinterpret_cast<const uint4*>(p + PLANE);
                    al[mb] = *reinterpret_cast<const uint4*>(p + 2 * PLANE);
                }
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f32x16 c = acc[mb][nb];
                    if constexpr (NP == 3) {
                        c = mfma_bf16(al[mb], b.v[ks][nb][0], c);     // smallest terms first
                        c = mfma_bf16(ah[mb], b.v[ks][nb][2], c);
                        c = mfma_bf16(am[mb], b.v[ks][nb][1], c);
                        c = mfma_bf16(am[mb], b.v[ks][nb][0], c);
                        c = mfma_bf16(ah[mb], b.v[ks][nb][1], c);
                    }
                    c = mfma_bf16(ah[mb], b.v[ks][nb][0], c);
                    acc[mb][nb] = c;
                }
        }
    };
    {
        BFrag b0, b1;
        if (ABL & 2) {
            b0.v[0][0][0] = make_uint4(a.N, a.H, a.W, lane);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) b1.v[ks][nb][pl] = b0.v[ks][nb][pl] = b0.v[0][0][0];
        }
        bool more = next();
        // a tile no tap reaches (odd-parity tiles of a strided 1x1 data gradient) adds nothing: leave dx untouched
        if (DGRAD && a.accumulate && !more) return;
        if (more) {
            issueA(0);
            bload(b0, (tap * nCC + cc) * 2);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int stage = 0;
        auto step = [&](const BFrag& bc, BFrag& bn) -> bool {
            const bool more2 = next();
            if (more2) {
                issueA(stage ^ 1);
                bload(bn, (tap * nCC + cc) * 2);
            }
            compute(stage, bc);
            __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): this wave's DMA pieces of the next chunk landed
            __builtin_amdgcn_s_barrier();                          // ... everyone's did, and this stage is free again
            asm volatile("" ::: "memory");
            stage ^= 1;
            return more2;
        };
        if (more)
            for (;;) {
                if (!step(b0, b1)) break;
                if (!step(b1, b0)) break;
            }
    }

    // ---- epilogue: identical C/D map to the other conv kernels
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = n0 + wn * (BN / 2) + nb * 32 + li;
        const bool cok = col < NC;
        const float bv = (!DGRAD && a.bias != nullptr && cok) ? a.bias[col] : 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lrw = wm * (BM / 2) + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int row = m0 + lrw;
                if (cok && row < a.M) {
                    const int pix = (DGRAD && par) ? rowpix[DGRAD ? lrw : 0] : row;
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

bool launch_conv_pl(ConvArgs a, hipStream_t s, bool dgrad, int bm, int planes) {
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
            if (bm == 128) hipLaunchKernelGGL((k_conv_pl<128, 64, true, 1>), grid, blk, 0, s, a);
            else hipLaunchKernelGGL((k_conv_pl<64, 64, true, 1>), grid, blk, 0, s, a);
        } else {
            if (bm == 128) hipLaunchKernelGGL((k_conv_pl<128, 64, false, 1>), grid, blk, 0, s, a);
            else hipLaunchKernelGGL((k_conv_pl<64, 64, false, 1>), grid, blk, 0, s, a);
        }
        return true;
    }
    static int abl = -1;
    if (abl < 0) {
        const char* e = getenv("ISWM_PL_ABL");
        abl = e ? atoi(e) : 0;
    }
    if (abl && !dgrad && bm == 128) {
        if (abl == 1) hipLaunchKernelGGL((k_conv_pl<128, 64, false, 3, 1>), grid, blk, 0, s, a);
        else if (abl == 2) hipLaunchKernelGGL((k_conv_pl<128, 64, false, 3, 2>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((k_conv_pl<128, 64, false, 3, 3>), grid, blk, 0, s, a);
        return true;
    }
    if (dgrad) {
        if (bm == 128) hipLaunchKernelGGL((k_conv_pl<128, 64, true, 3>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((k_conv_pl<64, 64, true, 3>), grid, blk, 0, s, a);
    } else {
        if (bm == 128) hipLaunchKernelGGL((k_conv_pl<128, 64, false, 3>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((k_conv_pl<64, 64, false, 3>), grid, blk, 0, s, a);
    }
    return true;
}

// exact 3-way split of an fp32 [M][C] (pitch ldx) matrix into bf16 planes [NP][M][ldp]  (NP == 1: round to nearest)
template <int NP>
__global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ x, int64_t M, int C4, int ldx,
                                                      unsigned short* __restrict__ out, int ldp, int64_t pstride) {
    const int64_t total = M * C4;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / C4;
        const int c4 = (int)(i - m * C4);
        const float4 v = ldg4(x + m * ldx + c4 * 4);
        unsigned short* o = out + m * ldp + c4 * 4;
        if constexpr (NP == 1) {
            *reinterpret_cast<uint2*>(o) = round_bf16x4(v);
        } else {
            uint2 h, md, l;
            split3(v, h, md, l);
            *reinterpret_cast<uint2*>(o) = h;
            *reinterpret_cast<uint2*>(o + pstride) = md;
            *reinterpret_cast<uint2*>(o + 2 * pstride) = l;
        }
    }
}

// planes -> fp32 (x = hi + mid + lo, exact); one plane: the bf16 value
__global__ __launch_bounds__(256) void k_join_planes(const unsigned short* __restrict__ in, int ldp, int64_t ps, int64_t M, int C4,
                                                     float* __restrict__ x, int ldx) {
    const int64_t total = M * C4;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / C4;
        const int c4 = (int)(i - m * C4);
        *reinterpret_cast<float4*>(x + m * ldx + c4 * 4) = ld4x(in, m * ldp + c4 * 4, ps);
    }
}

void launch_join_planes(const unsigned short* in, int ldp, int64_t ps, int64_t M, int C, float* x, int ldx, hipStream_t s) {
    hipLaunchKernelGGL(k_join_planes, dim3(stream_grid(M * (C / 4), 256)), dim3(256), 0, s, in, ldp, ps, M, C / 4, x, ldx);
}

void launch_split_planes(const float* x, int64_t M, int C, int ldx, unsigned short* out, int ldp, int64_t pstride_elems,
                         int planes, hipStream_t s) {
    const int grid = stream_grid(M * (C / 4), 256);
    if (planes == 1) hipLaunchKernelGGL(k_split_planes<1>, dim3(grid), dim3(256), 0, s, x, M, C / 4, ldx, out, ldp, pstride_elems);
    else hipLaunchKernelGGL(k_split_planes<3>, dim3(grid), dim3(256), 0, s, x, M, C / 4, ldx, out, ldp, pstride_elems);
}

}  // namespace iswm
