// bf16x6 implicit-GEMM convolution over planes, (16*rbw) x 256 tiles: k_conv_pl2 (conv_mfma_pl2.hip) with TWO 16-column
// blocks per wave.
//
// Why: in k_conv_pl2 a wave reads three activation fragments from LDS (ds_read_b128) and issues one load slot per SIX
// MFMAs; v_mfma_f32_16x16x32_bf16 holds the SIMD's vector issue for 8 of its 16 cycles (MI355X_MICROARCH.md, constants
// table), so with two waves per SIMD the issue port is ~85 % booked by the stage loop itself and any stall shows
// (profiles/r02_pl2_model.txt: 4 636 modelled / 5 000-5 400 measured cycles per stage against 3 456 of matrix work).
// Here every activation fragment feeds TWELVE MFMAs (two column blocks): half the LDS reads, half the activation DMA and
// the same weight traffic per MFMA.  The price is the tile count: 256 output columns per workgroup, so the planner
// (conv_pl2_plan) only takes this kernel where tiles still cover the 256 CUs -- Cout (forward) / Cin (data gradient)
// >= 512 on the 33 x 33 maps, and the 129 x 129 decoder convolutions.
//
// Registers: 2 * RBW accumulators (72 at RBW 9) + the weight fragments of ONE stage (2 column blocks x 2 k halves x 3
// planes = 48) -- the fragments are refreshed IN PLACE: while the second k half of a stage is multiplied the first half's
// registers take the next stage's fragments, and vice versa (no second fragment set, no 24-register copy per stage).
#include <stdlib.h>

#include "conv_common.h"

namespace iswm {

static __device__ __attribute__((aligned(128))) unsigned short g_zero_row_pl2w[64];   // 128 B of zeros
static __device__ float4 g_dump_pl2w[64];        // where the epilogue's out-of-range lanes store (never read)

typedef __attribute__((address_space(3))) void* lds_vptr2w;
typedef float f32x4w __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16bw(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

__device__ __forceinline__ f32x4w mfma16w(uint4 a, uint4 b, f32x4w c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Same operand conventions as k_conv_pl2 (ConvArgs; weights packed by k_pack_weights_pl2, whose column blocks run to
// ceil(NC / 128) * 8: a.cbs).  Tile = (16 * RBW) rows x 256 columns; wave w owns columns 32 w .. 32 w + 31 and ALL rows.
// Row-major rows only (no parity classes: strided data gradients stay on k_conv_pl2).  PERSISTENT like k_conv_pl2.
// (Static s_setprio 1 for waves 4-7, and priorities swapped between the two waves of a SIMD every row block, were measured
// equal or slower: profiles/r03_pl2w_ab.txt.)
template <int RBW, int NP, bool DGRAD>
__global__ __launch_bounds__(512, 2) void k_conv_pl2w(const ConvArgs a) {
    const int GC = DGRAD ? a.Cout : a.Cin;
    const int NC = DGRAD ? a.Cin : a.Cout;
    constexpr int BM = 16 * RBW, RG = BM / 8;
    constexpr int PLANE = BM * 128;
    constexpr int STAGE = NP * PLANE;
    constexpr int NRG = (RG + 7) / 8;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_vptr2w)smem;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int RH = DGRAD ? a.H : a.Ho, RW = DGRAD ? a.W : a.Wo;
    const int GH = DGRAD ? a.Ho : a.H, GW = DGRAD ? a.Wo : a.W;
    const int taps = a.KH * a.KW;
    const int nCC = GC >> 6;
    const int K32 = a.Ktot >> 5;
    const int ntiles = a.MT * a.NT;
    const int cbs = ((NC + 127) >> 7) << 3;         // packed column blocks
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x);
    const int gs = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    const unsigned char* zrow = reinterpret_cast<const unsigned char*>(g_zero_row_pl2w) + gs * 16;

    // ---- issue side
    int i_tile = xcd_remap(blockIdx.x, gridDim.x);
    int i_m0 = 0, i_n0 = 0;
    int ihb[NRG], iwb[NRG], pb[NRG];
    const uint4* wpk[2] = {nullptr, nullptr};        // issue side: this wave's two packed column blocks
    auto load_tile = [&](int tile) __attribute__((always_inline)) {
        const int mt = tile / a.NT;
        const int nt = tile - mt * a.NT;
        i_m0 = mt * BM;
        i_n0 = nt * 256;
#pragma unroll
        for (int i = 0; i < NRG; ++i) {
            const int m = i_m0 + 8 * (wave + 8 * i) + (lane >> 3);
            if (wave + 8 * i < RG && m < a.M) {
                int n, rh, rw;
                x6_row_pixel(m, a.N, RH, RW, false, n, rh, rw);
                ihb[i] = DGRAD ? rh + a.pad : rh * a.stride - a.pad;
                iwb[i] = DGRAD ? rw + a.pad : rw * a.stride - a.pad;
                pb[i] = n * GH * GW;
            } else {
                ihb[i] = -(1 << 28);
                iwb[i] = 0;
                pb[i] = 0;
            }
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            // columns past the packed range (a 256-wide tile over NC % 256 == 128 columns) re-read the last block; their
            // results are never stored
            const int cbi = min((i_n0 >> 4) + 2 * wave + cb, cbs - 1);
            wpk[cb] = reinterpret_cast<const uint4*>(a.w) + (size_t)cbi * K32 * (64 * NP) + lane;
        }
    };
    const unsigned char* aptr[NRG];
    int astep[NRG];
    long long pst[NRG];
    auto setup_tap = [&](int tap) __attribute__((always_inline)) -> bool {
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        const int dh = kh * a.dil, dw = kw * a.dil;
        int any = 0;
#pragma unroll
        for (int i = 0; i < NRG; ++i) {
            int gh, gw;
            bool ok;
            if (DGRAD) {
                gh = ihb[i] - dh;
                gw = iwb[i] - dw;
                ok = gh >= 0 && gw >= 0 && gh < GH && gw < GW;          // stride 1 only
            } else {
                gh = ihb[i] + dh;
                gw = iwb[i] + dw;
                ok = (unsigned)gh < (unsigned)GH && (unsigned)gw < (unsigned)GW;
            }
            aptr[i] = ok ? xb + ((size_t)(pb[i] + gh * GW + gw) * a.ldx) * 2 + gs * 16 : zrow;
            astep[i] = ok ? 128 : 0;
            pst[i] = ok ? a.xps : 0;
            any |= ok;
        }
        if (a.pad < 4) return true;                       // the vote only where padding is deep (ASPP rates)
        return __syncthreads_or(any) != 0;
    };
    int tap = -1, cc = nCC - 1;
    auto next_in_tile = [&]() __attribute__((always_inline)) -> bool {
        if (++cc < nCC) return true;
        cc = 0;
        do {
            if (++tap >= taps) return false;
        } while (!setup_tap(tap));
        return true;
    };
    auto issueA = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NRG; ++i) {
            if (wave + 8 * i < RG) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    glds16bw(aptr[i] + p * pst[i], lds_base + st * STAGE + p * PLANE + (wave + 8 * i) * 1024);
            }
            aptr[i] += astep[i];
        }
    };

    f32x4w acc[2][RBW];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < RBW; ++i) acc[cb][i] = f32x4w{0.f, 0.f, 0.f, 0.f};

    // weight fragments of the stage in hand: [column block][32-deep k half][plane]
    uint4 B[2][2][NP];
    auto bload_half = [&](const uint4* const (&wp)[2], int k32, int h) __attribute__((always_inline)) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const uint4* p = wp[cb] + (size_t)k32 * (64 * NP);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) B[cb][h][pl] = p[(h * NP + pl) * 64];
        }
    };

    const int fbase = (lane & 15) * 128 + (((lane >> 4) ^ ((lane & 15) >> 1)) * 16);
    struct AFrag {
        uint4 v[NP];
    };
    // compute side of the weight stream: the packed blocks and k32 index of the stage being multiplied
    const uint4* wpk_c[2] = {nullptr, nullptr};
    int k32_c = 0;

    // Multiply stage `st`.  Load slots, two per row block until they run out (the last row blocks of each k half stay free
    // so that what was issued has landed when the half ends):
    //   first k half : the SECOND half's weight fragments of THIS stage (their registers were busy until the previous
    //                  stage ended), then the first part of the next stage's activation DMA;
    //   second k half: the FIRST half's weight fragments of the NEXT stage (in place), then the rest of the DMA.
    auto compute = [&](int st, bool more, int k32n) __attribute__((always_inline)) {
        auto aload = [&](AFrag& f, int idx) __attribute__((always_inline)) {
            const int half = idx / RBW, rb = idx - half * RBW;
            const unsigned char* p = smem + st * STAGE + (fbase ^ (half * 64)) + rb * 2048;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) f.v[pl] = *reinterpret_cast<const uint4*>(p + pl * PLANE);
        };
        auto mul = [&](const AFrag& f, int idx) __attribute__((always_inline)) {
            const int half = idx / RBW, rb = idx - half * RBW;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                f32x4w c = acc[cb][rb];
                if constexpr (NP == 3) {
                    c = mfma16w(B[cb][half][0], f.v[2], c);     // smallest terms first
                    c = mfma16w(B[cb][half][2], f.v[0], c);
                    c = mfma16w(B[cb][half][1], f.v[1], c);
                    c = mfma16w(B[cb][half][0], f.v[1], c);
                    c = mfma16w(B[cb][half][1], f.v[0], c);
                }
                c = mfma16w(B[cb][half][0], f.v[0], c);
                acc[cb][rb] = c;
            }
        };
        constexpr int NB = 2 * NP;                      // weight loads per k half (two column blocks)
        constexpr int NA = NRG * NP;                    // activation DMA instructions per stage
        constexpr int NA0 = NA / 2, NA1 = NA - NA0;     // ... issued in the first / second k half
        const unsigned char* asrc[NRG];
        long long apl[NRG];
#pragma unroll
        for (int i = 0; i < NRG; ++i) {
            asrc[i] = more ? aptr[i] : zrow;
            apl[i] = more ? pst[i] : 0;
        }
        const uint4* wc0 = wpk_c[0] + (size_t)k32_c * (64 * NP);
        const uint4* wc1 = wpk_c[1] + (size_t)k32_c * (64 * NP);
        const uint4* wn0 = wpk[0] + (size_t)k32n * (64 * NP);
        const uint4* wn1 = wpk[1] + (size_t)k32n * (64 * NP);
        auto dma = [&](int j) __attribute__((always_inline)) {       // j-th DMA instruction of the next stage
            const int i = j / NP, pp = j - i * NP;
            if (8 * i + 8 <= RG || wave + 8 * i < RG)
                glds16bw(asrc[i] + pp * apl[i], lds_base + (st ^ 1) * STAGE + pp * PLANE + (wave + 8 * i) * 1024);
            if (pp == NP - 1) aptr[i] += astep[i];
        };
        auto slot = [&](int half, int s) __attribute__((always_inline)) {
            if (s < NB) {
                const int cb = s / NP, pl = s - cb * NP;
                if (half == 0) B[cb][1][pl] = (cb ? wc1 : wc0)[(NP + pl) * 64];          // this stage, second half
                else B[cb][0][pl] = (cb ? wn1 : wn0)[pl * 64];                           // next stage, first half
            } else {
                const int j = s - NB;
                if (half == 0) { if (j < NA0) dma(j); }
                else { if (j < NA1) dma(NA0 + j); }
            }
        };
        AFrag f[3];
        aload(f[0], 0);
        if (2 * RBW > 1) aload(f[1], 1);
#pragma unroll
        for (int idx = 0; idx < 2 * RBW; ++idx) {
            if (idx + 2 < 2 * RBW) aload(f[(idx + 2) % 3], idx + 2);
            __builtin_amdgcn_sched_barrier(0);
            mul(f[idx % 3], idx);
            const int half = idx / RBW, rb = idx - half * RBW;
            slot(half, 2 * rb);
            slot(half, 2 * rb + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        static_assert(2 * RBW >= NB + NA1, "load slots do not fit the row blocks of a k half");
    };

    // ---- epilogue of the tile (m0, n0): lane -> pixel (lane & 15) of a row block, its 4 registers -> 4 consecutive channels
    const int lq = lane >> 4, lp = lane & 15;
    auto epilogue = [&](int tile, int m0, int n0, bool zero) __attribute__((always_inline)) {
        const int mt = tile / a.NT;
        const bool bnf = DGRAD && a.bnf.part != nullptr;
        const bool mk = bnf && a.bnf.relu == 3;
        if (DGRAD && a.accumulate && zero && !bnf) return;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int col = n0 + 32 * wave + 16 * cb + 4 * lq;
            const bool cok = col < NC;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!DGRAD && a.bias != nullptr && cok) bv = *reinterpret_cast<const float4*>(a.bias + col);
            float4 f_mu = make_float4(0.f, 0.f, 0.f, 0.f), f_is = f_mu, f_sc = f_mu, f_sh = f_mu;
            float fs[4] = {0.f, 0.f, 0.f, 0.f}, fq[4] = {0.f, 0.f, 0.f, 0.f};
            if (bnf && cok) {
                f_mu = *reinterpret_cast<const float4*>(a.bnf.mean + col);
                f_is = *reinterpret_cast<const float4*>(a.bnf.invstd + col);
                if (a.bnf.relu == 2) {
                    f_sc = *reinterpret_cast<const float4*>(a.bnf.mscale + col);
                    f_sh = *reinterpret_cast<const float4*>(a.bnf.mshift + col);
                }
            }
            if constexpr (!DGRAD) {
                // FORWARD: BRANCH-FREE (see k_conv_pl2's epilogue: a predicate around each store made hipcc wait vmcnt(0) before
                // every one of them, i.e. every store waited for the previous store's acknowledgement): out-of-range lanes
                // store into a per-lane dump slot, selected by address
                float4* const dump = g_dump_pl2w + lane;
#pragma unroll
                for (int rb = 0; rb < RBW; ++rb) {
                    const int row = m0 + rb * 16 + lp;
                    const bool ok = cok && row < a.M;
                    float4 v = zero ? make_float4(0.f, 0.f, 0.f, 0.f)
                                    : make_float4(acc[cb][rb][0], acc[cb][rb][1], acc[cb][rb][2], acc[cb][rb][3]);
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    float4* o = ok ? reinterpret_cast<float4*>(&a.y[(size_t)(ok ? row : 0) * a.ldy + col]) : dump;
                    *o = v;
                }
            } else {
                // DATA GRADIENT: the predicated form (the branch-free one spills at RBW 9 / 10: 72-80 accumulators + 48 weight
                // registers leave no room for the selected addresses)
                // everything the epilogue reads is fetched before its first store (two halves of the row blocks)
                constexpr int EH = (RBW + 1) / 2;
    #pragma unroll
                for (int h0 = 0; h0 < RBW; h0 += EH) {
                    float4 oldv[EH], yvv[EH];
                    uint2 mkv[EH];
                    if constexpr (DGRAD) {
    #pragma unroll
                        for (int j = 0; j < EH; ++j) {
                            const int rb = h0 + j;
                            const int row = m0 + rb * 16 + lp;
                            oldv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                            yvv[j] = oldv[j];
                            mkv[j] = make_uint2(0u, 0u);
                            if (rb < RBW && cok && row < a.M) {
                                if (a.accumulate) oldv[j] = *reinterpret_cast<const float4*>(&a.y[(size_t)row * a.ldy + col]);
                                if (bnf) yvv[j] = *reinterpret_cast<const float4*>(a.bnf.y + (size_t)row * a.bnf.ldy + col);
                                if (mk) mkv[j] = *reinterpret_cast<const uint2*>(a.bnf.mask + (size_t)row * a.bnf.ldm + col);
                            }
                        }
                    }
    #pragma unroll
                    for (int j = 0; j < EH; ++j) {
                        const int rb = h0 + j;
                        if (rb >= RBW) continue;
                        const int row = m0 + rb * 16 + lp;
                        if (cok && row < a.M) {
                            float4* o = reinterpret_cast<float4*>(&a.y[(size_t)row * a.ldy + col]);
                            float4 v = zero ? make_float4(0.f, 0.f, 0.f, 0.f)
                                            : make_float4(acc[cb][rb][0], acc[cb][rb][1], acc[cb][rb][2], acc[cb][rb][3]);
                            if (DGRAD && a.accumulate) {
                                const float4 old = oldv[j];
                                v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
                            } else {
                                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                            }
                            if (mk) {
                                // residual producer: pattern from the hi plane of its saved output; the masked gradient is stored
                                const float4 ov = bf16x4_to_f32(mkv[j]);
                                v.x = ov.x > 0.f ? v.x : 0.f; v.y = ov.y > 0.f ? v.y : 0.f;
                                v.z = ov.z > 0.f ? v.z : 0.f; v.w = ov.w > 0.f ? v.w : 0.f;
                                *o = v;
                            } else if (!(DGRAD && a.accumulate && zero)) *o = v;
                            if (bnf) {
                                // same expressions as k_bn_bwd_reduce (bn.hip) and k_conv_pl2
                                const float4 yv = yvv[j];
                                float4 g = v;
                                if (a.bnf.relu == 2) {
                                    g.x = (yv.x - f_mu.x) * f_sc.x + f_sh.x > 0.f ? g.x : 0.f;
                                    g.y = (yv.y - f_mu.y) * f_sc.y + f_sh.y > 0.f ? g.y : 0.f;
                                    g.z = (yv.z - f_mu.z) * f_sc.z + f_sh.z > 0.f ? g.z : 0.f;
                                    g.w = (yv.w - f_mu.w) * f_sc.w + f_sh.w > 0.f ? g.w : 0.f;
                                }
                                fs[0] += g.x; fs[1] += g.y; fs[2] += g.z; fs[3] += g.w;
                                fq[0] += g.x * ((yv.x - f_mu.x) * f_is.x); fq[1] += g.y * ((yv.y - f_mu.y) * f_is.y);
                                fq[2] += g.z * ((yv.z - f_mu.z) * f_is.z); fq[3] += g.w * ((yv.w - f_mu.w) * f_is.w);
                            }
                        }
                    }
                }
            }
            if (bnf) {
                double ds[4], dq[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ds[r] = (double)fs[r];
                    dq[r] = (double)fq[r];
#pragma unroll
                    for (int m = 1; m < 16; m <<= 1) {
                        ds[r] += __shfl_xor(ds[r], m);
                        dq[r] += __shfl_xor(dq[r], m);
                    }
                }
                if (lp == 0 && cok) {
                    const size_t T = (size_t)a.MT, prow = (size_t)mt;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        a.bnf.part[prow * NC + col + r] = ds[r];
                        a.bnf.part[(T + prow) * NC + col + r] = dq[r];
                    }
                }
            }
            if (!DGRAD && a.stats != nullptr) {
                const int cnt = min(BM, a.M - m0);
                float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int rb = 0; rb < RBW; ++rb) {
                    const bool ok = !zero && m0 + rb * 16 + lp < a.M;
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[r] += ok ? acc[cb][rb][r] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[r] += __shfl_xor(s[r], 1);
                    s[r] += __shfl_xor(s[r], 2);
                    s[r] += __shfl_xor(s[r], 4);
                    s[r] += __shfl_xor(s[r], 8);
                }
                float qv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int rb = 0; rb < RBW; ++rb) {
                    const bool ok = !zero && m0 + rb * 16 + lp < a.M;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dv = acc[cb][rb][r] - s[r] / (float)cnt;
                        qv[r] += ok ? dv * dv : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    qv[r] += __shfl_xor(qv[r], 1);
                    qv[r] += __shfl_xor(qv[r], 2);
                    qv[r] += __shfl_xor(qv[r], 4);
                    qv[r] += __shfl_xor(qv[r], 8);
                }
                if (lp == 0 && cok) {
                    *reinterpret_cast<float4*>(&a.stats[(size_t)mt * a.Cout + col]) = make_float4(s[0], s[1], s[2], s[3]);
                    *reinterpret_cast<float4*>(&a.stats[(size_t)(a.MT + mt) * a.Cout + col]) = make_float4(qv[0], qv[1], qv[2], qv[3]);
                }
            }
        }
        if (!zero) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int i = 0; i < RBW; ++i) acc[cb][i] = f32x4w{0.f, 0.f, 0.f, 0.f};
        }
    };

    // ---- the stage stream
    int c_tile = i_tile, c_m0, c_n0;
    load_tile(i_tile);
    c_m0 = i_m0;
    c_n0 = i_n0;
    auto next_tile = [&]() __attribute__((always_inline)) -> bool {
        i_tile += gridDim.x;
        if (i_tile >= ntiles) return false;
        load_tile(i_tile);
        tap = -1;
        cc = nCC - 1;
        return true;
    };
    auto next_tile_stage = [&]() __attribute__((always_inline)) -> bool {
        for (;;) {
            if (!next_tile()) return false;
            if (next_in_tile()) return true;
            epilogue(i_tile, i_m0, i_n0, true);
        }
    };
    {
        bool have = next_in_tile();
        if (!have) {
            epilogue(i_tile, i_m0, i_n0, true);
            have = next_tile_stage();
            c_tile = i_tile; c_m0 = i_m0; c_n0 = i_n0;
        }
        if (have) {
            k32_c = tap * (GC >> 5) + 2 * cc;
            wpk_c[0] = wpk[0];
            wpk_c[1] = wpk[1];
            bload_half(wpk, k32_c, 0);          // the first stage's first-half fragments; its second half loads in-stage
            issueA(0);
        }
        int st = 0;
        while (have) {
            __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            bool more = next_in_tile();
            const bool last = !more;
            if (last) more = next_tile_stage();
            const int k32n = more ? tap * (GC >> 5) + 2 * cc : 0;
            compute(st, more, k32n);
            if (last) {
                epilogue(c_tile, c_m0, c_n0, false);
                c_tile = i_tile; c_m0 = i_m0; c_n0 = i_n0;
            }
            st ^= 1;
            k32_c = k32n;
            wpk_c[0] = wpk[0];
            wpk_c[1] = wpk[1];
            have = more;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
}

// launch the 256-column kernel; rbw in {8, 9, 10}.  Returns false when there is no instantiation.
bool launch_conv_pl2w(ConvArgs a, hipStream_t s, bool dgrad, int planes, int rbw) {
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    if (planes != 3 || (dgrad && a.stride != 1)) return false;
    const int nc = dgrad ? a.Cin : a.Cout;
    a.MT = (a.M + rbw * 16 - 1) / (rbw * 16);
    a.NT = (nc + 255) / 256;
    a.psplit = 1;
    const int tiles = a.MT * a.NT;
    dim3 grid(tiles < ncu ? tiles : ncu), blk(512);
#define PL2W_LAUNCH(R)                                                                     \
    do {                                                                                   \
        if (dgrad) hipLaunchKernelGGL((k_conv_pl2w<R, 3, true>), grid, blk, 0, s, a);       \
        else hipLaunchKernelGGL((k_conv_pl2w<R, 3, false>), grid, blk, 0, s, a);            \
    } while (0)
    if (rbw == 8) PL2W_LAUNCH(8);
    else if (rbw == 9) PL2W_LAUNCH(9);
    else if (rbw == 10) PL2W_LAUNCH(10);
    else return false;
#undef PL2W_LAUNCH
    return true;
}

}  // namespace iswm
