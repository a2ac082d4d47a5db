// Weight gradient over pre-split operands ("planes"):  dw[co][(tap, ci)] = sum_p dy[p][co] * x[p + tap][ci].
//
// GEMM rows = output channels, columns = the flattened (tap, input channel) axis, K = pixels.  Both operands are stored
// pixel-major, i.e. with the GEMM K axis STRIDED and the row/column axis contiguous -- which is exactly the order an
// LDS-DMA wants: a K step of 32 pixels x 128 channels is 32 rows of 256 B per plane, fetched in 4-row x 256-B pieces
// (whole cache lines) with no VGPRs, no split arithmetic and no ds_write.  The k-contiguous MFMA fragments come out of
// that [k][channel] image through gfx950's transposing LDS read (ds_read_b64_tr_b16); the 16-byte channel group g of k-row
// k sits in slot g ^ (4 * (k & 3)) (applied to the DMA source), which puts the four k-rows of one transposing read on
// disjoint bank quarters.
// Tile 128 x 128, 8 waves: waves 0-3 multiply the first 16 pixels of every 32-pixel step, waves 4-7 the second 16 (each
// wave a 64 x 64 quarter of the tile), and the two halves are added through LDS at the end -- two waves per SIMD keep the
// matrix pipe busy while the other issues its DMA.  One workgroup per CU; the pixel axis is split across workgroups
// (slabs summed in a fixed order by k_reduce_slabs: bit-reproducible).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "conv_common.h"

namespace iswm {

extern unsigned long long* g_conv_dbg;

static __device__ __attribute__((aligned(256))) unsigned short g_zero_row_wg[128];   // 256 B of zeros

typedef __attribute__((address_space(3))) void* lds_vptr3;

__device__ __forceinline__ void glds16w(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

struct WgArgs {
    const unsigned short* dy;   // planes of dy [P][ldy]
    const unsigned short* x;    // planes of x  [N*H*W][ldx]
    float* out;                 // slabs [nsplit][Cout][Ktot] (or dw itself when nsplit == 1)
    long long dyps, xps;        // plane strides in BYTES
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, dil, ldx, ldy;
    int P, Ktot, MT, NT, nsplit, psplit;
    int abl;                    // timing ablations (ISWM_WG_ABL): 1 no DMA, 2 no multiply
    int always;                 // 1: every gathered pixel is in bounds (1x1 stride-1 pad-0): no per-step culling vote
    unsigned long long* dbg;    // iswm_set_debug_buffer: per-step shader-clock stamps of workgroup 0 (tools/wgrad_timeline.py)
    int vote;                   // 1: skip 32-pixel steps whose gathered pixels are ALL padding (workgroup-wide vote, one more
                                // barrier per step: only worth it when the filter reaches far -- ASPP rates)
    // k_wgrad_pls, deep padding: every 256-column tile lies inside ONE filter tap (Cin % 256 == 0), and a tap (dh, dw) only
    // pairs output pixels of the rectangle [oh0, oh1) x [ow0, ow1) with in-bounds input pixels.  rect = 1: a tile walks
    // exactly the N x (oh1 - oh0) x (ow1 - ow0) pixels of its tap's rectangle (split nsplit ways) instead of all N x Ho x Wo
    // with a vote -- no padding pixel is ever fetched or multiplied.  tap_order: taps by descending rectangle size (the heavy
    // tiles are dispatched first).
    int rect;
    unsigned char tap_order[32];
};

template <int NP>
__global__ __launch_bounds__(512, 2) void k_wgrad_pl(const WgArgs a) {
    constexpr int PLANE = 32 * 256;            // bytes of one plane of one operand of one stage
    constexpr int OPER = NP * PLANE;
    constexpr int STAGE = 2 * OPER;            // A image then B image
    constexpr int NST = 3;                     // stage buffers: two steps in flight behind the one being multiplied
    constexpr int SMEM = NST * STAGE < 65536 ? 65536 : NST * STAGE;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_vptr3)smem;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int kh = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    // split-major order over a flat grid, dealt to the XCDs in contiguous runs: the workgroups of one pixel split (all tiles
    // over the same dy / x pixel range) share an L2 -- each operand byte then leaves HBM / Infinity Cache about once instead
    // of once per tile row / column
    const int tiles = a.MT * a.NT;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int split = L / tiles, tile = L - split * tiles;
    const int mt = tile / a.NT, nt = tile - mt * a.NT;
    const int m0 = mt * 128, n0 = nt * 128;
    const int p_begin = split * a.psplit;
    const int p_end = min(a.P, p_begin + a.psplit);
    const int nK = (p_end - p_begin + 31) >> 5;

    // ---- DMA role: k-row r = 4 * wave + (lane >> 4) of the 32-pixel step, LDS slot lane & 15, source channel group gsrc
    const int kr = 4 * wave + (lane >> 4);
    const int gsrc = (lane & 15) ^ (4 * ((lane >> 4) & 3));
    const unsigned char* zrow = reinterpret_cast<const unsigned char*>(g_zero_row_wg) + (lane & 15) * 16;
    // A (dy): channel m0 + 8 * gsrc
    const int ach = m0 + 8 * gsrc;
    const bool aok = ach < a.Cout;
    const unsigned char* abase = reinterpret_cast<const unsigned char*>(a.dy) + (size_t)ach * 2;
    // B (x): column n0 + 8 * gsrc of the (tap, ci) axis -> one tap and channel for the whole loop
    const int bcol = n0 + 8 * gsrc;
    const bool bok = bcol < a.Ktot;
    const int tap = bok ? bcol / a.Cin : 0, bch = bok ? bcol - tap * a.Cin : 0;
    const int tkh = tap / a.KW, tkw = tap - tkh * a.KW;
    const int dh = tkh * a.dil - a.pad, dw = tkw * a.dil - a.pad;
    const unsigned char* bbase = reinterpret_cast<const unsigned char*>(a.x) + (size_t)bch * 2;
    const int HoWo = a.Ho * a.Wo;

    const unsigned char* asrc = zrow;
    const unsigned char* bsrc = zrow;
    long long apst = 0, bpst = 0;
    int kc = -1;
    // pixel (n, oh, ow) of this lane's k-row in the CURRENT step, advanced by 32 pixels per step with a branch-free carry
    // (integer divisions per step cost more VALU issue than the step's 24 MFMAs leave room for); tiny maps re-divide
    int pn, poh, pow_;
    {
        const int p = p_begin + kr;
        pn = p / HoWo;
        const int rem = p - pn * HoWo;
        poh = rem / a.Wo;
        pow_ = rem - poh * a.Wo;
    }
    const int d_oh = 32 / a.Wo, d_ow = 32 - d_oh * a.Wo;
    const bool fast_adv = d_oh + 1 <= a.Ho;
    // move to the next 32-pixel step in which some gathered pixel is in bounds; leaves the DMA sources on it
    auto next = [&]() __attribute__((always_inline)) -> bool {
        for (;;) {
            if (++kc >= nK) return false;
            const int p = p_begin + kc * 32 + kr;
            const bool pin = p < p_end;
            asrc = (pin && aok) ? abase + (size_t)p * a.ldy * 2 : zrow;
            apst = (pin && aok) ? a.dyps : 0;
            if (a.always) {              // 1x1, stride 1, no padding: the gathered pixel IS p
                const bool v = pin && bok;
                bsrc = v ? bbase + (size_t)p * a.ldx * 2 : zrow;
                bpst = v ? a.xps : 0;
                return true;
            }
            if (kc > 0) {
                if (fast_adv) {
                    int ow = pow_ + d_ow;
                    const int c1 = ow >= a.Wo ? 1 : 0;
                    pow_ = ow - (c1 ? a.Wo : 0);
                    int oh = poh + d_oh + c1;
                    const int c2 = oh >= a.Ho ? 1 : 0;
                    poh = oh - (c2 ? a.Ho : 0);
                    pn += c2;
                } else {
                    pn = p / HoWo;
                    const int rem = p - pn * HoWo;
                    poh = rem / a.Wo;
                    pow_ = rem - poh * a.Wo;
                }
            }
            const int ih = poh * a.stride + dh, iw = pow_ * a.stride + dw;
            const bool v = pin && bok && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
            bsrc = v ? bbase + (size_t)((pn * a.H + ih) * a.W + iw) * a.ldx * 2 : zrow;
            bpst = v ? a.xps : 0;
            if (!a.vote) return true;
            if (__syncthreads_or(v ? 1 : 0)) return true;
        }
    };
    auto issue = [&](int st) __attribute__((always_inline)) {
        if (a.abl & 1) return;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) glds16w(asrc + pl * apst, lds_base + st * STAGE + pl * PLANE + wave * 1024);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) glds16w(bsrc + pl * bpst, lds_base + st * STAGE + OPER + pl * PLANE + wave * 1024);
    };

    // ---- transposing fragment reads (see conv_mfma.hip k_conv_wgrad): 16-lane group g = lane >> 4 covers rows
    // 16 * (g & 1).. of the 32-row MFMA block for k half g >> 1; lane 4q + p of the group addresses k-row q, columns 4p..
    const int tg = lane >> 4, ti = lane & 15, tq = ti >> 2, tp = ti & 3;
    const int th = tg >> 1, tc = (tg & 1) * 16 + tp * 4;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    auto tr_frag = [&](const unsigned char* plane, int col0) __attribute__((always_inline)) -> uint4 {
        const unsigned char* p = plane + (kh * 16 + th * 8 + tq) * 256 + (((col0 + tc) * 2) ^ (tq * 64));
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 4 * 256));
        uint2 a2 = __builtin_bit_cast(uint2, lo), b2 = __builtin_bit_cast(uint2, hi);
        return make_uint4(a2.x, a2.y, b2.x, b2.y);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int st) __attribute__((always_inline)) {
        const unsigned char* Ax = smem + st * STAGE;
        const unsigned char* Bx = Ax + OPER;
        uint4 af[2][NP], bf[2][NP];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) af[mb][pl] = tr_frag(Ax + pl * PLANE, wm * 64 + mb * 32);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) bf[nb][pl] = tr_frag(Bx + pl * PLANE, wn * 64 + nb * 32);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                f32x16 c = acc[mb][nb];
                if constexpr (NP == 3) {
                    c = mfma_bf16(af[mb][2], bf[nb][0], c);     // smallest terms first
                    c = mfma_bf16(af[mb][0], bf[nb][2], c);
                    c = mfma_bf16(af[mb][1], bf[nb][1], c);
                    c = mfma_bf16(af[mb][1], bf[nb][0], c);
                    c = mfma_bf16(af[mb][0], bf[nb][1], c);
                }
                c = mfma_bf16(af[mb][0], bf[nb][0], c);
                acc[mb][nb] = c;
            }
    };

    {
        // three stage buffers: the DMA of a step is issued two steps ahead of its multiply (the ~1-2 us it takes to
        // land is otherwise exposed at every barrier: a 128x128x32 step is only ~0.7 us of matrix work)
        int nissued = 0, wr = 0, rd = 0;
        bool src_more = true;
        auto advance = [&]() __attribute__((always_inline)) -> bool {      // uniform: holds the culling vote
            if (src_more && next()) return true;
            src_more = false;
            return false;
        };
        auto post = [&]() __attribute__((always_inline)) {
            issue(wr);
            wr = wr == NST - 1 ? 0 : wr + 1;
            ++nissued;
        };
        if (advance()) post();
        if (advance()) post();
        while (nissued > 0) {
            // the oldest step in flight must have landed; a younger one (2 * NP DMA instructions per wave) may stay in flight
            if (nissued >= 2) {
                if constexpr (NP == 3) __builtin_amdgcn_s_waitcnt(0x0F76);      // vmcnt(6)
                else __builtin_amdgcn_s_waitcnt(0x0F72);                        // vmcnt(2)
            } else {
                __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0)
            }
            __builtin_amdgcn_s_barrier();        // everyone's pieces landed; the buffer multiplied last time is free
            asm volatile("" ::: "memory");
            if (advance()) post();
            if (!(a.abl & 2)) compute(rd);
            rd = rd == NST - 1 ? 0 : rd + 1;
            --nissued;
        }
    }

    // ---- combine the two k halves through LDS, then write the tile (slab of this split)
    __syncthreads();
    float* xch = reinterpret_cast<float*>(smem);        // [wm][wn][mb][nb][r][lane]
    if (kh == 1) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[((((wm * 2 + wn) * 2 + mb) * 2 + nb) * 16 + r) * 64 + lane] = acc[mb][nb][r];
    }
    __syncthreads();
    if (kh == 0) {
        float* out = a.out + (size_t)split * a.Cout * a.Ktot;
        const int li = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int col = n0 + wn * 64 + nb * 32 + li;
            const bool cok = col < a.Ktot;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float v = acc[mb][nb][r] + xch[((((wm * 2 + wn) * 2 + mb) * 2 + nb) * 16 + r) * 64 + lane];
                    if (cok && row < a.Cout) out[(size_t)row * a.Ktot + col] = v;
                }
        }
    }
}


// ---- 128 x 256 tile ("wide"): the same DMA / transposing-read machinery on a tile twice as wide along the (tap, ci) axis.
// Per 32-pixel step a CU now multiplies 128 x 256 x 32 (3 072 matrix cycles per SIMD instead of 1 536) behind ONE barrier,
// and fetches 72 KB for it (23 B/clk at full matrix rate instead of 31; a single workgroup per CU is fed ~30 B/clk,
// profiles/r02_ta_bw.txt).  Eight waves as 2 x 4, each a 64 x 64 block over the whole step: no k-half split, so no LDS
// combine at the end.  Two stage buffers of [A | B0 | B1] images (144 KB).
// DBG: per-step stamps and the in-kernel clock (tools/wgrad_timeline.py); ABL: timing ablations (1 no DMA, 2 no multiply).
// Both are compile-time: an untaken scalar branch between the barrier and the first MFMA of a step is not free
// (tools/mfma_rate.hip), and the production instantiation carries none of them.
template <int NP, bool DBG = false, int ABL = 0>
__global__ __launch_bounds__(512, 2) void k_wgrad_plw(const WgArgs a) {
    constexpr int PLANE = 32 * 256;            // bytes of one plane of one 128-channel image of one stage
    constexpr int IMG = NP * PLANE;
    constexpr int STAGE = 3 * IMG;             // A image, then the two B images
    constexpr int NST = 2;
    constexpr int SMEM = NST * STAGE < 65536 ? 65536 : NST * STAGE;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_vptr3)smem;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int tiles = a.MT * a.NT;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int split = L / tiles, tile = L - split * tiles;
    const int mt = tile / a.NT, nt = tile - mt * a.NT;
    const int m0 = mt * 128, n0 = nt * 256;
    const int p_begin = split * a.psplit;
    const int p_end = min(a.P, p_begin + a.psplit);
    const int nK = (p_end - p_begin + 31) >> 5;

    const int kr = 4 * wave + (lane >> 4);
    const int gsrc = (lane & 15) ^ (4 * ((lane >> 4) & 3));
    const unsigned char* zrow = reinterpret_cast<const unsigned char*>(g_zero_row_wg) + (lane & 15) * 16;
    const int ach = m0 + 8 * gsrc;
    const bool aok = ach < a.Cout;
    const unsigned char* abase = reinterpret_cast<const unsigned char*>(a.dy) + (size_t)ach * 2;
    // the two B images: columns n0 + 8 * gsrc and n0 + 128 + 8 * gsrc of the (tap, ci) axis
    bool bok[2];
    int dh[2], dw[2];
    const unsigned char* bbase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int bcol = n0 + 128 * i + 8 * gsrc;
        bok[i] = bcol < a.Ktot;
        const int tap = bok[i] ? bcol / a.Cin : 0, bch = bok[i] ? bcol - tap * a.Cin : 0;
        const int tkh = tap / a.KW, tkw = tap - tkh * a.KW;
        dh[i] = tkh * a.dil - a.pad;
        dw[i] = tkw * a.dil - a.pad;
        bbase[i] = reinterpret_cast<const unsigned char*>(a.x) + (size_t)bch * 2;
    }
    const int HoWo = a.Ho * a.Wo;

    const unsigned char* asrc = zrow;
    const unsigned char* bsrc[2] = {zrow, zrow};
    long long apst = 0, bpst[2] = {0, 0};
    int kc = -1;
    int pn, poh, pow_;
    {
        const int p = p_begin + kr;
        pn = p / HoWo;
        const int rem = p - pn * HoWo;
        poh = rem / a.Wo;
        pow_ = rem - poh * a.Wo;
    }
    const int d_oh = 32 / a.Wo, d_ow = 32 - d_oh * a.Wo;
    const bool fast_adv = d_oh + 1 <= a.Ho;
    auto next = [&]() __attribute__((always_inline)) -> bool {
        for (;;) {
            if (++kc >= nK) return false;
            const int p = p_begin + kc * 32 + kr;
            const bool pin = p < p_end;
            asrc = (pin && aok) ? abase + (size_t)p * a.ldy * 2 : zrow;
            apst = (pin && aok) ? a.dyps : 0;
            if (a.always) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bool v = pin && bok[i];
                    bsrc[i] = v ? bbase[i] + (size_t)p * a.ldx * 2 : zrow;
                    bpst[i] = v ? a.xps : 0;
                }
                return true;
            }
            if (kc > 0) {
                if (fast_adv) {
                    int ow = pow_ + d_ow;
                    const int c1 = ow >= a.Wo ? 1 : 0;
                    pow_ = ow - (c1 ? a.Wo : 0);
                    int oh = poh + d_oh + c1;
                    const int c2 = oh >= a.Ho ? 1 : 0;
                    poh = oh - (c2 ? a.Ho : 0);
                    pn += c2;
                } else {
                    pn = p / HoWo;
                    const int rem = p - pn * HoWo;
                    poh = rem / a.Wo;
                    pow_ = rem - poh * a.Wo;
                }
            }
            bool any = false;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ih = poh * a.stride + dh[i], iw = pow_ * a.stride + dw[i];
                const bool v = pin && bok[i] && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
                bsrc[i] = v ? bbase[i] + (size_t)((pn * a.H + ih) * a.W + iw) * a.ldx * 2 : zrow;
                bpst[i] = v ? a.xps : 0;
                any = any || v;
            }
            if (!a.vote) return true;
            if (__syncthreads_or(any ? 1 : 0)) return true;
        }
    };
    // DMA instruction i of a step: image i / NP (A, B0, B1), plane i % NP
    auto dma = [&](int i, int st) __attribute__((always_inline)) {
        const int img = i / NP, pl = i - img * NP;
        const unsigned dst = lds_base + st * STAGE + img * IMG + pl * PLANE + wave * 1024;
        if (img == 0) glds16w(asrc + pl * apst, dst);
        else glds16w(bsrc[img - 1] + pl * bpst[img - 1], dst);
    };
    auto issue = [&](int st) __attribute__((always_inline)) {
        if (ABL & 1) return;
#pragma unroll
        for (int i = 0; i < 3 * NP; ++i) dma(i, st);
    };

    const int tg = lane >> 4, ti = lane & 15, tq = ti >> 2, tp = ti & 3;
    const int th = tg >> 1, tc = (tg & 1) * 16 + tp * 4;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    auto tr_frag = [&](const unsigned char* plane, int ks, int col0) __attribute__((always_inline)) -> uint4 {
        const unsigned char* p = plane + (ks * 16 + th * 8 + tq) * 256 + (((col0 + tc) * 2) ^ (tq * 64));
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 4 * 256));
        uint2 a2 = __builtin_bit_cast(uint2, lo), b2 = __builtin_bit_cast(uint2, hi);
        return make_uint4(a2.x, a2.y, b2.x, b2.y);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // One step = 8 blocks of 6 MFMAs (2 k halves x 2 x 2 blocks of 32 x 32).  Fragments live in four register groups (the
    // two A row blocks, the two B column blocks); every group is re-read ONE block before the block that needs it, in an
    // order in which the group being overwritten is already dead -- so the transposing reads run under the previous
    // block's MFMAs instead of in a read phase of their own.  Only the first two groups of a step are exposed.
    uint4 F[4][NP];        // 0, 1: A row blocks;  2, 3: B column blocks
    auto ldA = [&](int st, int mb, int ks) __attribute__((always_inline)) {
        const unsigned char* Ax = smem + st * STAGE;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) F[mb][pl] = tr_frag(Ax + pl * PLANE, ks, wm * 64 + mb * 32);
    };
    auto ldB = [&](int st, int nb, int ks) __attribute__((always_inline)) {
        const unsigned char* Bx = smem + st * STAGE + (1 + (wn >> 1)) * IMG;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) F[2 + nb][pl] = tr_frag(Bx + pl * PLANE, ks, (wn & 1) * 64 + nb * 32);
    };
    auto mm = [&](int mb, int nb) __attribute__((always_inline)) {
        f32x16 c = acc[mb][nb];
        if constexpr (NP == 3) {
            c = mfma_bf16(F[mb][2], F[2 + nb][0], c);     // smallest terms first
            c = mfma_bf16(F[mb][0], F[2 + nb][2], c);
            c = mfma_bf16(F[mb][1], F[2 + nb][1], c);
            c = mfma_bf16(F[mb][1], F[2 + nb][0], c);
            c = mfma_bf16(F[mb][0], F[2 + nb][1], c);
        }
        c = mfma_bf16(F[mb][0], F[2 + nb][0], c);
        acc[mb][nb] = c;
    };
    // Where the next step's DMA instructions are issued -- as a burst after the barrier, dealt over the 8 blocks, or over
    // blocks 0-3 by one wave of a SIMD and 4-7 by the other -- measured the same within noise (profiles/r02_notes.md), so
    // the burst (least code) stays.
    auto compute = [&](int st) __attribute__((always_inline)) {
#define ISWM_SB() __builtin_amdgcn_sched_barrier(0)
        ldA(st, 0, 0); ldB(st, 0, 0);
        ISWM_SB(); ldB(st, 1, 0); ISWM_SB(); mm(0, 0); ISWM_SB();
        ISWM_SB(); ldA(st, 1, 0); ISWM_SB(); mm(0, 1); ISWM_SB();
        ISWM_SB(); ldA(st, 0, 1); ISWM_SB(); mm(1, 1); ISWM_SB();     // A0 <- second k half
        ISWM_SB(); ldB(st, 1, 1); ISWM_SB(); mm(1, 0); ISWM_SB();     // B1 <- second k half
        ISWM_SB(); ldB(st, 0, 1); ISWM_SB(); mm(0, 1); ISWM_SB();     // B0 <- second k half
        ISWM_SB(); ldA(st, 1, 1); ISWM_SB(); mm(0, 0); ISWM_SB();     // A1 <- second k half
        ISWM_SB(); mm(1, 0); ISWM_SB();
        ISWM_SB(); mm(1, 1); ISWM_SB();
        ISWM_SB();
#undef ISWM_SB
    };

    if (DBG && a.dbg != nullptr && blockIdx.x == 0 && t == 0) {          // in-kernel clock
        a.dbg[500] = __builtin_amdgcn_s_memtime();
        a.dbg[501] = __builtin_amdgcn_s_memrealtime();
    }
    {
        int dbg_n = 0;
        const bool dbg = DBG && a.dbg != nullptr && blockIdx.x == 0 && (wave == 0 || wave == 4);
        auto stamp = [&](int k) __attribute__((always_inline)) {
            if constexpr (DBG) {
                if (dbg && dbg_n < 40 && lane == 0) a.dbg[(wave ? 256 : 0) + dbg_n * 6 + k] = __builtin_amdgcn_s_memtime();
            }
        };
        bool have = next();
        if (have) issue(0);
        int st = 0;
        while (have) {
            stamp(0);
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's pieces of the step have landed
            stamp(1);
            __builtin_amdgcn_s_barrier();            // everyone's have; the other buffer is free again
            asm volatile("" ::: "memory");
            stamp(2);
            const bool more = next();
            if (more) issue(st ^ 1);
            stamp(3);
            if (!(ABL & 2)) compute(st);
            stamp(4);
            st ^= 1;
            have = more;
            ++dbg_n;
        }
        stamp(0);
    }
    if (DBG && a.dbg != nullptr && blockIdx.x == 0 && t == 0) {
        a.dbg[502] = __builtin_amdgcn_s_memtime();
        a.dbg[503] = __builtin_amdgcn_s_memrealtime();
    }

    if (a.nsplit > 1) {
        // slab of this (split, tile) in ACCUMULATOR order: [wave][(mb, nb, q)][lane] float4 -- 16 whole-line stores per lane
        // instead of 64 four-byte ones; k_reduce_slabs_frag sums the splits in that order and scatters the result into dw
        float4* slab = reinterpret_cast<float4*>(a.out) + ((size_t)split * tiles + tile) * 8192 + (size_t)wave * 1024 + lane;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    slab[((mb * 2 + nb) * 4 + q) * 64] =
                        make_float4(acc[mb][nb][4 * q], acc[mb][nb][4 * q + 1], acc[mb][nb][4 * q + 2], acc[mb][nb][4 * q + 3]);
        return;
    }
    float* out = a.out;
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int col = n0 + wn * 64 + nb * 32 + li;
        const bool cok = col < a.Ktot;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (cok && row < a.Cout) out[(size_t)row * a.Ktot + col] = acc[mb][nb][r];
            }
    }
}

// ---- 128 x 256 tile, LOADER and MULTIPLIER waves ("specialised").
// In k_wgrad_plw all eight waves do everything in lockstep: after the step's barrier each of them advances its pixel, builds
// three source addresses, issues 9 DMA instructions, reads its first fragments -- and only then starts its 48 MFMAs; the two
// waves of a SIMD cannot cover each other because the barrier keeps them in phase, so every one of those sections lengthens
// the step (profiles/r02_wgrad_model.txt: 3 086 cycles of matrix work per step, ~6 100 measured).  A deeper DMA ring by
// itself does not help (tried: 16-pixel steps through four buffers, all waves still doing everything: 11-13 % SLOWER, isolated
// and in the network -- the step is not waiting for memory, profiles/r03_wgrad_ring.txt).
// Here the roles are split.  Waves 4-7 are LOADERS: each stands on 4 of a stage's 16 pixel rows, does the pixel bookkeeping
// and issues the stage's 36 DMA pieces (9 per loader), three stages ahead through a ring of four 36-KB buffers, and waits
// for a stage to land before it joins that stage's barrier.  Waves 0-3 are MULTIPLIERS, one per SIMD: a 64 x 128 block of the
// tile each (128 accumulator registers), 48 MFMAs of 32x32x16 per 16-pixel stage with the transposing fragment reads in
// the gaps; between two barriers a multiplier executes nothing but reads and MFMAs.
// Stages whose gathered pixels are all padding are not skipped (no workgroup-wide vote: it would put the multipliers back
// in the loaders' lockstep) -- the deep-padding ASPP shapes stay on k_wgrad_plw.
// MW: multiplier waves -- 4 (one per SIMD, 64 x 128 each) or 8 (two per SIMD, 64 x 64 each; 12 waves per workgroup)
template <int NP, int ABL = 0, int MW = 4>
__global__ __launch_bounds__(64 * (MW + 4), (MW + 4) / 4) void k_wgrad_pls(const WgArgs a) {
    constexpr int KS = 16;                     // pixels per stage
    constexpr int PLANE = KS * 256;            // bytes of one plane of one 128-channel image of one stage
    constexpr int IMG = NP * PLANE;
    constexpr int STAGE = 3 * IMG;             // A image, then the two B images
    constexpr int NST = 4;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_vptr3)smem;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tiles = a.MT * a.NT;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    int split = L / tiles, tile = L - split * tiles;
    int mt = tile / a.NT;
    int nt = tile - mt * a.NT;
    if (a.rect == 2) {
        // eight 256-channel column blocks, eight XCDs: XCD x (workgroups x, x + 8, ...) takes column block x of EVERY tap, walked
        // split-major (image range), then taps heaviest first, then the row tiles -- the taps of a block read the same rows of
        // the input shifted by the dilation, and now find them in their XCD's L2 (1.7 MB per image and block).  With the generic
        // order an XCD held all eight blocks of five taps (13 MB per image): every tap re-read the input through the fabric,
        // 6-10x the operand bytes (profiles/r03_pmc/traffic_by_geometry.txt).
        const int idx = blockIdx.x >> 3, per_split = (a.NT >> 3) * a.MT;
        split = idx / per_split;
        const int rem = idx - split * per_split;
        mt = rem % a.MT;
        nt = (rem / a.MT) * 8 + (blockIdx.x & 7);              // (tap slot) * chunks + column block
    }
    // rect mode: this tile's tap, its rectangle of output pixels and the part of it this split walks
    int r_oh0 = 0, r_ow0 = 0, r_h = a.Ho, r_w = a.Wo, r_dh = 0, r_dw = 0;
    int p_begin = split * a.psplit;
    int p_end = min(a.P, p_begin + a.psplit);
    if (a.rect) {
        const int chunks = a.Cin >> 8;                         // 256-column tiles per tap
        const int slot = nt / chunks;
        const int tap = a.tap_order[slot];
        nt = tap * chunks + (nt - slot * chunks);
        const int tkh = tap / a.KW, tkw = tap - tkh * a.KW;
        r_dh = tkh * a.dil - a.pad;
        r_dw = tkw * a.dil - a.pad;
        r_oh0 = max(0, -r_dh);
        r_ow0 = max(0, -r_dw);
        r_h = max(0, min(a.Ho, a.H - r_dh) - r_oh0);
        r_w = max(0, min(a.Wo, a.W - r_dw) - r_ow0);
        const int Pt = a.N * r_h * r_w;
        const int per = (((Pt + a.nsplit - 1) / a.nsplit) + KS - 1) / KS * KS;
        p_begin = split * per;
        p_end = min(Pt, p_begin + per);
    }
    const int m0 = mt * 128, n0 = nt * 256;
    const int nK = p_end > p_begin ? (p_end - p_begin + KS - 1) / KS : 0;

    if (wave >= MW) {
        // ================= loader =================
        const int rg = wave - MW;                              // 4-row group of the stage
        const int kr = 4 * rg + (lane >> 4);
        const int gsrc = (lane & 15) ^ (4 * ((lane >> 4) & 3));
        const unsigned char* zrow = reinterpret_cast<const unsigned char*>(g_zero_row_wg) + (lane & 15) * 16;
        const int ach = m0 + 8 * gsrc;
        const bool aok = ach < a.Cout;
        const unsigned char* abase = reinterpret_cast<const unsigned char*>(a.dy) + (size_t)ach * 2;
        bool bok[2];
        int dh[2], dw[2];
        const unsigned char* bbase[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int bcol = n0 + 128 * i + 8 * gsrc;
            bok[i] = bcol < a.Ktot;
            const int tap = bok[i] ? bcol / a.Cin : 0, bch = bok[i] ? bcol - tap * a.Cin : 0;
            const int tkh = tap / a.KW, tkw = tap - tkh * a.KW;
            dh[i] = tkh * a.dil - a.pad;
            dw[i] = tkw * a.dil - a.pad;
            bbase[i] = reinterpret_cast<const unsigned char*>(a.x) + (size_t)bch * 2;
        }
        // the pixel grid this tile walks: all of N x Ho x Wo, or (rect) N x r_h x r_w with origin (r_oh0, r_ow0)
        const int GW_ = max(1, r_w), GHW = max(1, r_h * r_w);
        int pn, poh, pow_;                                     // image, row and column INSIDE the grid
        {
            const int p = p_begin + kr;
            pn = p / GHW;
            const int rem = p - pn * GHW;
            poh = rem / GW_;
            pow_ = rem - poh * GW_;
        }
        const int d_oh = KS / GW_, d_ow = KS - d_oh * GW_;
        const bool fast_adv = d_oh + 1 <= r_h;
        // DMA of stage kc into its ring slot
        auto issue = [&](int kc) __attribute__((always_inline)) {
            const int p = p_begin + kc * KS + kr;
            const bool pin = p < p_end;
            const unsigned char* asrc;
            long long apst;
            const unsigned char* bsrc[2];
            long long bpst[2];
            if (a.rect) {
                if (kc > 0) {
                    if (fast_adv) {
                        int ow = pow_ + d_ow;
                        const int c1 = ow >= GW_ ? 1 : 0;
                        pow_ = ow - (c1 ? GW_ : 0);
                        int oh = poh + d_oh + c1;
                        const int c2 = oh >= r_h ? 1 : 0;
                        poh = oh - (c2 ? r_h : 0);
                        pn += c2;
                    } else {
                        pn = p / GHW;
                        const int rem = p - pn * GHW;
                        poh = rem / GW_;
                        pow_ = rem - poh * GW_;
                    }
                }
                const int oh = r_oh0 + poh, ow = r_ow0 + pow_;
                const bool va = pin && aok;
                asrc = va ? abase + (size_t)((pn * a.Ho + oh) * a.Wo + ow) * a.ldy * 2 : zrow;
                apst = va ? a.dyps : 0;
                const size_t xpix = (size_t)((pn * a.H + oh + r_dh) * a.W + ow + r_dw);      // in bounds by construction
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bool v = pin && bok[i];
                    bsrc[i] = v ? bbase[i] + xpix * a.ldx * 2 : zrow;
                    bpst[i] = v ? a.xps : 0;
                }
                const unsigned dstr = lds_base + (kc & 3) * STAGE + rg * 1024;
                if constexpr (ABL & 1) return;
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) glds16w(asrc + pl * apst, dstr + pl * PLANE);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) glds16w(bsrc[i] + pl * bpst[i], dstr + (1 + i) * IMG + pl * PLANE);
                return;
            }
            asrc = (pin && aok) ? abase + (size_t)p * a.ldy * 2 : zrow;
            apst = (pin && aok) ? a.dyps : 0;
            if (a.always) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bool v = pin && bok[i];
                    bsrc[i] = v ? bbase[i] + (size_t)p * a.ldx * 2 : zrow;
                    bpst[i] = v ? a.xps : 0;
                }
            } else {
                if (kc > 0) {
                    if (fast_adv) {
                        int ow = pow_ + d_ow;
                        const int c1 = ow >= GW_ ? 1 : 0;
                        pow_ = ow - (c1 ? GW_ : 0);
                        int oh = poh + d_oh + c1;
                        const int c2 = oh >= r_h ? 1 : 0;
                        poh = oh - (c2 ? r_h : 0);
                        pn += c2;
                    } else {
                        pn = p / GHW;
                        const int rem = p - pn * GHW;
                        poh = rem / GW_;
                        pow_ = rem - poh * GW_;
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ih = poh * a.stride + dh[i], iw = pow_ * a.stride + dw[i];
                    const bool v = pin && bok[i] && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
                    bsrc[i] = v ? bbase[i] + (size_t)((pn * a.H + ih) * a.W + iw) * a.ldx * 2 : zrow;
                    bpst[i] = v ? a.xps : 0;
                }
            }
            const unsigned dst = lds_base + (kc & 3) * STAGE + rg * 1024;
            if constexpr (ABL & 1) return;              // timing ablation: no DMA
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) glds16w(asrc + pl * apst, dst + pl * PLANE);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) glds16w(bsrc[i] + pl * bpst[i], dst + (1 + i) * IMG + pl * PLANE);
        };
        const int pre = nK < 3 ? nK : 3;
        for (int kc = 0; kc < pre; ++kc) issue(kc);
        for (int s2 = 0; s2 < nK; ++s2) {
            // stage s2 must have landed; stages issued beyond it (<= 2) may stay in flight: 3 NP pieces each
            const int beyond = min(nK - 1, s2 + 2) - s2;
            if constexpr (NP == 3) {
                if (beyond >= 2) __builtin_amdgcn_s_waitcnt(0x4F72);        // vmcnt(18)
                else if (beyond == 1) __builtin_amdgcn_s_waitcnt(0x0F79);   // vmcnt(9)
                else __builtin_amdgcn_s_waitcnt(0x0F70);
            } else {
                if (beyond >= 2) __builtin_amdgcn_s_waitcnt(0x0F76);        // vmcnt(6)
                else if (beyond == 1) __builtin_amdgcn_s_waitcnt(0x0F73);   // vmcnt(3)
                else __builtin_amdgcn_s_waitcnt(0x0F70);
            }
            __builtin_amdgcn_s_barrier();          // B(s2): the multipliers are done with stage s2 - 1, its buffer is free
            asm volatile("" ::: "memory");
            if (s2 + 3 < nK) issue(s2 + 3);
        }
        return;
    }

    // ================= multiplier =================
    constexpr int NBW = MW == 4 ? 4 : 2;               // 32-column blocks per multiplier
    const int wm = MW == 4 ? wave >> 1 : wave >> 2;    // rows 64 wm ..
    const int wn4 = MW == 4 ? 2 * (wave & 1) : (wave & 3);        // first 64-column group of this wave (of 4)
    const int tg = lane >> 4, ti = lane & 15, tq = ti >> 2, tp = ti & 3;
    const int th = tg >> 1, tc = (tg & 1) * 16 + tp * 4;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
    auto tr_frag = [&](const unsigned char* plane, int col0) __attribute__((always_inline)) -> uint4 {
        const unsigned char* p = plane + (th * 8 + tq) * 256 + (((col0 + tc) * 2) ^ (tq * 64));
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 4 * 256));
        uint2 a2 = __builtin_bit_cast(uint2, lo), b2 = __builtin_bit_cast(uint2, hi);
        return make_uint4(a2.x, a2.y, b2.x, b2.y);
    };
    f32x16 acc[2][NBW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NBW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    uint4 FA[2][NP], FB[NBW][NP];
    auto ldA = [&](int slot, int mb) __attribute__((always_inline)) {
        const unsigned char* Ax = smem + slot * STAGE;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) FA[mb][pl] = tr_frag(Ax + pl * PLANE, wm * 64 + mb * 32);
    };
    auto ldB = [&](int slot, int nb) __attribute__((always_inline)) {
        const unsigned char* Bx = smem + slot * STAGE + (1 + (wn4 >> 1)) * IMG;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) FB[nb][pl] = tr_frag(Bx + pl * PLANE, (wn4 & 1) * 64 + nb * 32);
    };
    auto mm = [&](int mb, int nb) __attribute__((always_inline)) {
        f32x16 c = acc[mb][nb];
        if constexpr (ABL & 2) {                        // timing ablation: one MFMA per block instead of six
            c = mfma_bf16(FA[mb][0] ^ FA[mb][1] ^ FA[mb][2], FB[nb][0] ^ FB[nb][1] ^ FB[nb][2], c);
            acc[mb][nb] = c;
            return;
        }
        if constexpr (NP == 3) {
            c = mfma_bf16(FA[mb][2], FB[nb][0], c);     // smallest terms first
            c = mfma_bf16(FA[mb][0], FB[nb][2], c);
            c = mfma_bf16(FA[mb][1], FB[nb][1], c);
            c = mfma_bf16(FA[mb][1], FB[nb][0], c);
            c = mfma_bf16(FA[mb][0], FB[nb][1], c);
        }
        c = mfma_bf16(FA[mb][0], FB[nb][0], c);
        acc[mb][nb] = c;
    };
#define ISWM_SB() __builtin_amdgcn_sched_barrier(0)
    for (int s2 = 0; s2 < nK; ++s2) {
        __builtin_amdgcn_s_barrier();              // B(s2): stage s2 has landed
        asm volatile("" ::: "memory");
        const int slot = s2 & 3;
        // blocks of 6 MFMAs; every fragment group is read one block before the block that needs it
        ldA(slot, 0); ldB(slot, 0);
        if constexpr (NBW == 4) {
            ISWM_SB(); ldB(slot, 1); ISWM_SB(); mm(0, 0); ISWM_SB();
            ISWM_SB(); ldB(slot, 2); ISWM_SB(); mm(0, 1); ISWM_SB();
            ISWM_SB(); ldB(slot, 3); ISWM_SB(); mm(0, 2); ISWM_SB();
            ISWM_SB(); ldA(slot, 1); ISWM_SB(); mm(0, 3); ISWM_SB();
            ISWM_SB(); mm(1, 3); ISWM_SB();
            ISWM_SB(); mm(1, 2); ISWM_SB();
            ISWM_SB(); mm(1, 1); ISWM_SB();
            ISWM_SB(); mm(1, 0); ISWM_SB();
        } else {
            ISWM_SB(); ldB(slot, 1); ISWM_SB(); mm(0, 0); ISWM_SB();
            ISWM_SB(); ldA(slot, 1); ISWM_SB(); mm(0, 1); ISWM_SB();
            ISWM_SB(); mm(1, 1); ISWM_SB();
            ISWM_SB(); mm(1, 0); ISWM_SB();
        }
        // all reads of the stage were waited for by the MFMAs that consumed them: the next barrier may free its buffer
    }
#undef ISWM_SB

    // slab / result layout of k_wgrad_plw (k_reduce_slabs_frag): its "wave" w8 = 4 wm + (64-column group), block (mb, nb & 1);
    // the slab sits at the tile's OUTPUT position (rect mode walks the column tiles in tap_order)
    if (a.nsplit > 1) {
        float4* slab = reinterpret_cast<float4*>(a.out) + ((size_t)split * tiles + (mt * a.NT + nt)) * 8192 + lane;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const int w8 = wm * 4 + wn4 + (nb >> 1);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    slab[(size_t)w8 * 1024 + ((mb * 2 + (nb & 1)) * 4 + q) * 64] =
                        make_float4(acc[mb][nb][4 * q], acc[mb][nb][4 * q + 1], acc[mb][nb][4 * q + 2], acc[mb][nb][4 * q + 3]);
            }
        return;
    }
    float* out = a.out;
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int col = n0 + wn4 * 64 + nb * 32 + li;
        const bool cok = col < a.Ktot;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (cok && row < a.Cout) out[(size_t)row * a.Ktot + col] = acc[mb][nb][r];
            }
    }
}

// sum of the accumulator-order slabs of k_wgrad_plw over the splits (fixed order: bit-reproducible), scattered into dw:
// float4 f of tile t holds rows m0 + 64 wm + 32 mb + 8 q + 4 (lane >> 5) + 0..3 of column n0 + 64 wn + 32 nb + (lane & 31)
__global__ __launch_bounds__(256) void k_reduce_slabs_frag(const float4* __restrict__ slabs, float* __restrict__ dw, int tiles,
                                                           int NT, int nsplit, int Cout, int Ktot) {
    const int64_t total = (int64_t)tiles * 8192;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float4 s = slabs[i];
        int k = 1;
        for (; k + 3 < nsplit; k += 4) {           // four splits in flight per thread
            const float4 v0 = slabs[i + (int64_t)k * total], v1 = slabs[i + (int64_t)(k + 1) * total];
            const float4 v2 = slabs[i + (int64_t)(k + 2) * total], v3 = slabs[i + (int64_t)(k + 3) * total];
            s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
            s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
            s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
            s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
        }
        for (; k < nsplit; ++k) {
            const float4 v = slabs[i + (int64_t)k * total];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int tile = (int)(i >> 13), f = (int)(i & 8191);
        const int wave = f >> 10, j = (f >> 6) & 15, lane = f & 63;
        const int mt = tile / NT, nt = tile - mt * NT;
        const int wm = wave >> 2, wn = wave & 3, mb = j >> 3, nb = (j >> 2) & 1, q = j & 3;
        const int col = nt * 256 + wn * 64 + nb * 32 + (lane & 31);
        const int row = mt * 128 + wm * 64 + mb * 32 + 8 * q + 4 * (lane >> 5);
        if (col < Ktot) {
            float* o = dw + (size_t)row * Ktot + col;
            if (row < Cout) o[0] = s.x;
            if (row + 1 < Cout) o[(size_t)Ktot] = s.y;
            if (row + 2 < Cout) o[(size_t)2 * Ktot] = s.z;
            if (row + 3 < Cout) o[(size_t)3 * Ktot] = s.w;
        }
    }
}

// tile width along the (tap, ci) axis: 256 when that wastes little
static int wgrad_pl_wide(int Ktot, int taps, int64_t P) {
    static int force = -2;
    if (force == -2) force = getenv("ISWM_WG_WIDE") ? atoi(getenv("ISWM_WG_WIDE")) : -1;
    if (force >= 0) return force && Ktot > 128;
    if (Ktot <= 128) return 0;
    // the slabs of a launch are ~256 workgroups x one tile whatever the shape: 32 MB instead of 16 MB.  K x K filters and
    // the 33 x 33 maps have the work per launch to pay for that (measured 1.05-1.4x); a 1 x 1 on a large map splits
    // 64-128 ways over 1-4 tiles and does not (0.8-0.95x)
    if (taps == 1 && P > 20000) return 0;
    const int pad256 = (Ktot + 255) / 256 * 256, pad128 = (Ktot + 127) / 128 * 128;
    return pad256 * 10 <= pad128 * 14;
}

int wgrad_pl_is_wide(const iswm_conv_desc* d) {
    return wgrad_pl_wide(d->KH * d->KW * d->Cin, d->KH * d->KW, (int64_t)d->N * d->Ho * d->Wo);
}

// pixels per split (multiple of 32) and split count: minimise  rounds x (steps per workgroup + fixed cost) + slab traffic
void plan_wgrad_pl(int Cout, int Ktot, int taps, int64_t P, int* nsplit, int* psplit) {
    const int wide = wgrad_pl_wide(Ktot, taps, P);
    const int tn = wide ? 256 : 128;
    const int64_t tiles = (int64_t)((Cout + 127) / 128) * ((Ktot + tn - 1) / tn);
    const double step_us = wide ? 1.45 : 0.75;                    // one 128 x tn x 32 step of a CU
    const double slab_us = (double)Cout * Ktot * 8.0 / 4.0e6;     // one slab written + read at ~4 TB/s
    int64_t maxs = (P + 255) / 256;
    if (maxs > 256) maxs = 256;
    if (const char* e = getenv("ISWM_WGPL_SPLIT")) {
        const int64_t v = atoi(e);
        if (v >= 1 && v <= maxs) {
            const int64_t ps = ((P + v - 1) / v + 31) / 32 * 32;
            *psplit = (int)ps;
            *nsplit = (int)((P + ps - 1) / ps);
            return;
        }
    }
    double best = 1e300;
    *psplit = (int)((P + 31) / 32 * 32);
    *nsplit = 1;
    for (int64_t ns = 1; ns <= maxs; ++ns) {
        const int64_t ps = ((P + ns - 1) / ns + 31) / 32 * 32;
        const int64_t nsp = (P + ps - 1) / ps;
        const int64_t rounds = (tiles * nsp + 255) / 256;
        const double tm = (double)rounds * ((double)(ps / 32) + 6.0) * step_us + (nsp > 1 ? (double)nsp * slab_us + 5.0 : 0.0);
        if (tm < best) {
            best = tm;
            *psplit = (int)ps;
            *nsplit = (int)nsp;
        }
    }
}

void launch_wgrad_pl(const WgArgs& a, int planes, int wide, hipStream_t s) {
    dim3 grid(a.MT * a.NT * a.nsplit), blk(512);
    static int spec = -1;
    if (spec < 0) spec = getenv("ISWM_WG_SPEC") ? atoi(getenv("ISWM_WG_SPEC")) : 1;
    if (wide && spec && !a.vote && a.abl == 0 && a.dbg == nullptr) {
        static int sabl = -1;
        if (sabl < 0) sabl = getenv("ISWM_WGS_ABL") ? atoi(getenv("ISWM_WGS_ABL")) : 0;
        if (planes == 1) hipLaunchKernelGGL(k_wgrad_pls<1>, grid, blk, 0, s, a);
        else if (sabl == 1) hipLaunchKernelGGL((k_wgrad_pls<3, 1>), grid, blk, 0, s, a);
        else if (sabl == 2) hipLaunchKernelGGL((k_wgrad_pls<3, 2>), grid, blk, 0, s, a);
        else if (sabl == 3) hipLaunchKernelGGL((k_wgrad_pls<3, 3>), grid, blk, 0, s, a);
        else if (sabl == 8) hipLaunchKernelGGL((k_wgrad_pls<3, 0, 8>), grid, dim3(768), 0, s, a);
        else if (sabl == 9) hipLaunchKernelGGL((k_wgrad_pls<3, 1, 8>), grid, dim3(768), 0, s, a);
        else hipLaunchKernelGGL(k_wgrad_pls<3>, grid, blk, 0, s, a);
        return;
    }
    if (wide) {
        if (planes == 1) hipLaunchKernelGGL(k_wgrad_plw<1>, grid, blk, 0, s, a);
        else if (a.abl == 1) hipLaunchKernelGGL((k_wgrad_plw<3, false, 1>), grid, blk, 0, s, a);
        else if (a.abl == 2) hipLaunchKernelGGL((k_wgrad_plw<3, false, 2>), grid, blk, 0, s, a);
        else if (a.abl == 3) hipLaunchKernelGGL((k_wgrad_plw<3, false, 3>), grid, blk, 0, s, a);
        else if (a.dbg != nullptr) hipLaunchKernelGGL((k_wgrad_plw<3, true, 0>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL(k_wgrad_plw<3>, grid, blk, 0, s, a);
    } else {
        if (planes == 1) hipLaunchKernelGGL(k_wgrad_pl<1>, grid, blk, 0, s, a);
        else hipLaunchKernelGGL(k_wgrad_pl<3>, grid, blk, 0, s, a);
    }
}

}  // namespace iswm

using namespace iswm;

namespace iswm {
void launch_reduce_slabs(const float* slabs, float* dst, int64_t n4, int nsplit, hipStream_t s);
}

// every precondition of the planes weight gradient in ONE place: iswm_conv2d_wgrad_planes_ok() answers with it (callers
// fall back to the fp32-input weight gradient) and the entry point refuses with its message.  nullptr = acceptable.
static const char* wg_refusal(const iswm_conv_desc* d) {
    if (d == nullptr) return "wgrad_planes: null descriptor";
    if (!(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0)) return "wgrad_planes: empty tensor";
    if (!(d->Cin % 8 == 0 && d->Cout % 8 == 0 && d->ldx % 8 == 0 && d->ldy % 8 == 0 && d->ldx >= d->Cin && d->ldy >= d->Cout))
        return "wgrad_planes: channel counts and pitches must be multiples of 8";
    if (!(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0)) return "wgrad_planes: bad geometry";
    const int ho = (d->H + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
    if (!(ho == d->Ho && wo == d->Wo)) return "wgrad_planes: output size does not match geometry";
    if (!((int64_t)d->N * d->H * d->W * d->ldx < (1ll << 30) && (int64_t)d->N * d->Ho * d->Wo * d->ldy < (1ll << 30)))
        return "wgrad_planes: tensor too large (2^30 elements per operand)";
    return nullptr;
}

static int wg_validate(const iswm_conv_desc* d) {
    const char* why = wg_refusal(d);
    ISWM_REQUIRE(why == nullptr, "%s (Cin %d Cout %d ldx %d ldy %d)", why, d ? d->Cin : 0, d ? d->Cout : 0, d ? d->ldx : 0,
                 d ? d->ldy : 0);
    return 0;
}

extern "C" int iswm_conv2d_wgrad_planes_ok(const iswm_conv_desc* d) {
    return (wg_refusal(d) == nullptr && iswm_get_conv_math() >= 1) ? 1 : 0;
}

// Tap-rectangle mode of k_wgrad_pls (WgArgs::rect): deep padding, stride 1, whole taps per 256-column tile.  Returns the mean
// number of pixels a tile walks (what the split planner balances) and fills the taps by descending rectangle size.
static bool wgrad_rect_mode(const iswm_conv_desc* d, int64_t* p_eff, unsigned char* order, bool* similar = nullptr) {
    static int on = -1;
    if (on < 0) on = getenv("ISWM_WG_RECT") ? atoi(getenv("ISWM_WG_RECT")) : 1;
    const int taps = d->KH * d->KW;
    if (!on || d->pad < 4 || d->stride != 1 || d->Cin % 256 != 0 || taps <= 1 || taps > 32 || iswm_get_conv_math() != 1) return false;
    int64_t area[32], sum = 0;
    for (int t = 0; t < taps; ++t) {
        const int dh = (t / d->KW) * d->dil - d->pad, dw = (t % d->KW) * d->dil - d->pad;
        const int h = std::max(0, std::min(d->Ho, d->H - dh) - std::max(0, -dh));
        const int w = std::max(0, std::min(d->Wo, d->W - dw) - std::max(0, -dw));
        area[t] = (int64_t)d->N * h * w;
        sum += area[t];
    }
    if (order) {
        for (int t = 0; t < taps; ++t) order[t] = (unsigned char)t;
        for (int i = 0; i < taps; ++i)
            for (int j = i + 1; j < taps; ++j)
                if (area[order[j]] > area[order[i]]) std::swap(order[i], order[j]);
    }
    if (p_eff) *p_eff = std::max<int64_t>(32, sum / taps);
    if (similar) {              // are the taps' rectangles of similar size (smallest >= 35 % of the largest)?
        int64_t lo = area[0], hi = area[0];
        for (int t = 1; t < taps; ++t) {
            lo = std::min(lo, area[t]);
            hi = std::max(hi, area[t]);
        }
        *similar = 100 * lo >= 35 * hi;
    }
    return true;
}

// which kernel iswm_conv2d_wgrad_planes launches for this geometry (iswm_conv2d_kernel_name): 0 k_wgrad_pl (128 x 128 tiles),
// 1 k_wgrad_plw (128 x 256, all waves load and multiply, culling vote), 2 k_wgrad_pls (loader / multiplier waves)
namespace iswm {
int wgrad_pl_kernel_kind(const iswm_conv_desc* d) {
    if (!wgrad_pl_is_wide(d)) return 0;
    static int spec = -1;
    if (spec < 0) spec = getenv("ISWM_WG_SPEC") ? atoi(getenv("ISWM_WG_SPEC")) : 1;
    const bool rect = wgrad_rect_mode(d, nullptr, nullptr);
    const bool vote = d->pad >= 4 && !rect;
    return (spec && !vote) ? 2 : 1;
}
}  // namespace iswm

extern "C" size_t iswm_conv2d_wgrad_planes_workspace(const iswm_conv_desc* d) {
    if (!d) return 0;
    int ns, ps;
    const int Ktot = d->KH * d->KW * d->Cin;
    int64_t P = (int64_t)d->N * d->Ho * d->Wo;
    wgrad_rect_mode(d, &P, nullptr);                  // the planner balances what the tiles really walk
    plan_wgrad_pl(d->Cout, Ktot, d->KH * d->KW, P, &ns, &ps);
    if (ns <= 1) return 0;
    if (wgrad_pl_wide(Ktot, d->KH * d->KW, (int64_t)d->N * d->Ho * d->Wo))      // whole 128 x 256 tiles in accumulator order
        return (size_t)ns * ((d->Cout + 127) / 128) * ((Ktot + 255) / 256) * 32768 * sizeof(float);
    return (size_t)ns * d->Cout * Ktot * sizeof(float);
}

extern "C" int iswm_conv2d_wgrad_planes(const iswm_conv_desc* d, const void* xp, int64_t x_ps, const void* dyp, int64_t dy_ps,
                                        float* dw, float* workspace, size_t workspace_bytes, iswm_stream_t stream) {
    if (int e = wg_validate(d)) return e;
    ISWM_REQUIRE(xp && dyp && dw && aligned16(xp) && aligned16(dyp) && aligned16(dw), "wgrad_planes: bad pointer");
    const int planes = iswm_get_conv_math() == 2 ? 1 : 3;
    ISWM_REQUIRE(planes == 1 || (x_ps % 8 == 0 && dy_ps % 8 == 0 && x_ps > 0 && dy_ps > 0), "wgrad_planes: bad plane stride");
    WgArgs a{};
    a.dy = (const unsigned short*)dyp; a.x = (const unsigned short*)xp;
    a.dyps = dy_ps * 2; a.xps = x_ps * 2;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil; a.ldx = d->ldx; a.ldy = d->ldy;
    a.P = d->N * d->Ho * d->Wo;
    a.Ktot = d->KH * d->KW * d->Cin;
    a.MT = (d->Cout + 127) / 128;
    const int wide = wgrad_pl_wide(a.Ktot, d->KH * d->KW, a.P);
    a.NT = wide ? (a.Ktot + 255) / 256 : (a.Ktot + 127) / 128;
    int64_t p_plan = a.P;
    bool similar = false;
    a.rect = (wide && wgrad_rect_mode(d, &p_plan, a.tap_order, &similar)) ? 1 : 0;
    {
        // one column block per XCD (k_wgrad_pls) where the taps walk their rectangles at a similar pace (rates 6 and 12 on the
        // 33 x 33 map: fabric fetch -43 % / -27 %, kernel -4.6 % / -2.2 %; at rate 18 the corner taps are a fifth of the centre
        // tap, the XCD's taps drift apart by images and the order only unbalances the rounds: +3 %) -- profiles/r03_rect_xcd.txt
        static int rx = -2;
        if (rx == -2) rx = getenv("ISWM_WG_RECT_XCD") ? atoi(getenv("ISWM_WG_RECT_XCD")) : -1;    // tuning switch: 0 never, 1 always
        if (a.rect && (d->Cin >> 8) == 8 && (rx == 1 || (rx < 0 && similar))) a.rect = 2;
    }
    plan_wgrad_pl(d->Cout, a.Ktot, d->KH * d->KW, p_plan, &a.nsplit, &a.psplit);
    static int abl = -1;
    if (abl < 0) abl = getenv("ISWM_WG_ABL") ? atoi(getenv("ISWM_WG_ABL")) : 0;
    a.abl = abl;
    a.dbg = g_conv_dbg;
    {
        static int fv = -2;
        if (fv == -2) fv = getenv("ISWM_WG_VOTE") ? atoi(getenv("ISWM_WG_VOTE")) : -1;
        a.vote = fv >= 0 ? fv : (d->pad >= 4 ? 1 : 0);
        if (a.rect) a.vote = 0;                       // nothing to vote on: the tile walks in-bounds pixels only
    }
    a.always = (d->KH == 1 && d->KW == 1 && d->pad == 0 && d->stride == 1) ? 1 : 0;
    const size_t need = iswm_conv2d_wgrad_planes_workspace(d);
    ISWM_REQUIRE(workspace_bytes >= need && (need == 0 || (workspace && aligned16(workspace))),
                 "wgrad_planes: workspace too small (%zu < %zu)", workspace_bytes, need);
    a.out = a.nsplit > 1 ? workspace : dw;
    launch_wgrad_pl(a, planes, wide, (hipStream_t)stream);
    if (int e = check_launch("wgrad_planes")) return e;
    if (a.nsplit > 1) {
        if (wide) {
            const int tiles = a.MT * a.NT;
            hipLaunchKernelGGL(k_reduce_slabs_frag, dim3(stream_grid((int64_t)tiles * 8192, 256)), dim3(256), 0,
                               (hipStream_t)stream, reinterpret_cast<const float4*>(workspace), dw, tiles, a.NT, a.nsplit,
                               d->Cout, a.Ktot);
        } else {
            launch_reduce_slabs(workspace, dw, (int64_t)d->Cout * a.Ktot / 4, a.nsplit, (hipStream_t)stream);
        }
        return check_launch("wgrad_planes_reduce");
    }
    return 0;
}
