// bf16x6 implicit-GEMM convolution over pre-split activations ("planes"), second generation: one
// (16*rbw) x 128 tile per compute unit.
//
// What the measurements behind this layout say (profiles/r02_notes.md):
//   * removing the in-kernel split (conv_mfma_pl.hip vs conv_mfma_x6.hip) changes nothing by itself: the
//     128x64-tile kernels are bound by what the CU's vector-memory path (TA / L1) can feed -- 192*(BM+BN)
//     operand bytes per 32-deep K step against 0.094*BM*BN matrix cycles -- and by tile quantisation
//     (a 33x33x16 map is 137 row tiles of 128: 548 workgroups leave a third of the chip idle in the last round);
//   * LDS-DMA of 64-byte row segments (32 bf16 channels) runs at HALF the rate of 128-byte segments
//     (12-17 vs 18-33 TB/s chip-wide from L2): a stage must hold 64 channels so every row is a whole line.
// Hence:
//   * tile = 16*rbw rows (rbw chosen per geometry so that tiles ~ k * 256 CUs: 144 rows for 33x33x16) x 128
//     columns, 8 waves, ONE workgroup per CU; L1 traffic per MAC is 2.3x lower than with 128x64 tiles;
//   * wave w owns columns 16w..16w+15 and ALL rows: its weight fragments come straight from global memory
//     (fragment-ordered packing, 16 B per lane, 3 loads per 32-deep step) with no redundancy between waves and no
//     LDS; the activation tile is shared through LDS and read by every wave (125 B/clk, half the LDS rate);
//   * activation stage = 64 channels = three planes of [row][128 B], two stages (<= 120 KB), filled by LDS-DMA
//     in 8-row x 128-B pieces; the 16-byte group g of row r sits in slot g ^ ((r >> 1) & 7), which makes the
//     16x16x32 fragment reads (ds_read_b128) conflict-free on unpadded rows (swizzle applied to the DMA SOURCE);
//   * v_mfma_f32_16x16x32_bf16: 16-row granularity for the tile height, and the chip holds a higher clock on it;
//   * one barrier per 32-deep step; everything a step needs was issued a full step earlier.
#include <stdlib.h>

#include "conv_common.h"

namespace iswm {

static __device__ __attribute__((aligned(128))) unsigned short g_zero_row_pl2[64];   // 128 B of zeros
static __device__ float4 g_dump_pl2[64];         // where the epilogue's out-of-range lanes store (never read)

typedef __attribute__((address_space(3))) void* lds_vptr2;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16b(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

__device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int PL2_RBWMAX = 10;     // tile height up to 160 rows

unsigned long long* g_conv_dbg = nullptr;        // iswm_set_debug_buffer

// a.x = plane 0 of the gathered operand (bf16), a.ldx = its pixel pitch in bf16 elements, a.xps = plane stride (bytes)
// a.w = weights packed by k_pack_weights_pl2;  a.MT, a.NT tile counts;  a.psplit != 0: row-major rows for strided dgrad.
// Tile = (16 * RBW * WM) rows x (128 / WM) columns; wave (wm, wn) owns rows [wm*16*RBW, +16*RBW) and columns 16*wn..+15.
// PERSISTENT: the grid is min(tiles, CUs) workgroups; a workgroup walks tiles  it * gridDim + xcd_remap(blockIdx)  and its
// stage pipeline runs across tile boundaries, so a tile's epilogue overlaps the DMA of the next tile's first stage.
template <int RBW, int WM, int NP, bool DGRAD, bool DBG = false, int ABL = 0>
__global__ __launch_bounds__(512, 2) void k_conv_pl2(const ConvArgs a) {
    const int GC = DGRAD ? a.Cout : a.Cin;     // channels of the gathered operand (per tap)
    const int NC = DGRAD ? a.Cin : a.Cout;     // output columns
    constexpr int WN = 8 / WM, BN = 16 * WN, BM = 16 * RBW * WM, RG = BM / 8;
    constexpr int PLANE = BM * 128;                   // bytes of one plane of one stage
    constexpr int STAGE = NP * PLANE;
    constexpr int NRG = (RG + 7) / 8;                 // 8-row DMA groups per wave
    constexpr int XTRA = (DGRAD ? BM * 4 : 0) + (WM > 1 ? 4 * 128 * 4 : 0);
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE + XTRA];
    int* rowpix = reinterpret_cast<int*>(smem + 2 * STAGE);
    float* red = reinterpret_cast<float*>(smem + 2 * STAGE + (DGRAD ? BM * 4 : 0));
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_vptr2)smem;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const bool par = DGRAD && a.stride == 2 && a.psplit == 0;
    const int RH = DGRAD ? a.H : a.Ho, RW = DGRAD ? a.W : a.Wo;
    const int GH = DGRAD ? a.Ho : a.H, GW = DGRAD ? a.Wo : a.W;
    const int taps = a.KH * a.KW;
    const int nCC = GC >> 6;                   // 64-channel stages per tap
    const int K32 = a.Ktot >> 5;
    const int ntiles = a.MT * a.NT;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x);
    // DMA role of this lane: row (lane >> 3) of an 8-row group, source 16-byte group gs of the 128-byte row
    const int gs = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    const unsigned char* zrow = reinterpret_cast<const unsigned char*>(g_zero_row_pl2) + gs * 16;

    // ---- issue side: the tile / tap / channel stage the DMA pointers stand on
    int i_tile = xcd_remap(blockIdx.x, gridDim.x);      // < ntiles (grid <= ntiles)
    int i_m0 = 0, i_n0 = 0;
    int ihb[NRG], iwb[NRG], pb[NRG];
    const uint4* wpk = nullptr;
    // Strided data gradient: rows are sorted by the parity class of their pixel (x6_row_pixel) and the tile sequence walks
    // the four quarters of the M tiles HEAVIEST FIRST (a.porder, two bits per rank: a class is reached by 1, 2 or 4 of a 3x3's
    // taps, by one or none of a 1x1's).  The persistent grid hands out tiles round-robin, so every workgroup gets its share
    // of the heavy ones in the first rounds; dealing the quarters tile by tile (as the one-tile-per-workgroup kernels of
    // round 1 do across XCD runs) gave each workgroup tiles of ONE class -- a quarter of the CUs did all of a 1x1's work.
    auto par_mt = [&](int sq) __attribute__((always_inline)) -> int {
        const int qn = a.MT >> 2, rem = a.MT & 3;
        int k = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            k = (a.porder >> (2 * j)) & 3;
            const int sz = qn + (k < rem ? 1 : 0);
            if (sq < sz || j == 3) break;
            sq -= sz;
        }
        return k * qn + (k < rem ? k : rem) + sq;
    };
    auto load_tile = [&](int tile) __attribute__((always_inline)) {
        int mt = tile / a.NT;
        const int nt = tile - mt * a.NT;
        if (par) mt = par_mt(mt);
        i_m0 = mt * BM;
        i_n0 = nt * BN;
#pragma unroll
        for (int i = 0; i < NRG; ++i) {
            const int m = i_m0 + 8 * (wave + 8 * i) + (lane >> 3);
            if (wave + 8 * i < RG && m < a.M) {
                int n, rh, rw;
                x6_row_pixel(m, a.N, RH, RW, par, n, rh, rw);
                ihb[i] = DGRAD ? rh + a.pad : rh * a.stride - a.pad;
                iwb[i] = DGRAD ? rw + a.pad : rw * a.stride - a.pad;
                pb[i] = n * GH * GW;
            } else {
                ihb[i] = -(1 << 28);
                iwb[i] = 0;
                pb[i] = 0;
            }
        }
        // this wave's packed weight fragments: column block (n0 >> 4) + wn; 64*NP uint4 per (column block, k32)
        wpk = reinterpret_cast<const uint4*>(a.w) + (size_t)((i_n0 >> 4) + wn) * K32 * (64 * NP) + lane;
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
                int th = ihb[i] - dh, tw = iwb[i] - dw;
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
                gh = ihb[i] + dh;
                gw = iwb[i] + dw;
                ok = (unsigned)gh < (unsigned)GH && (unsigned)gw < (unsigned)GW;
            }
            aptr[i] = ok ? xb + ((size_t)(pb[i] + gh * GW + gw) * a.ldx) * 2 + gs * 16 : zrow;
            astep[i] = ok ? 128 : 0;
            pst[i] = ok ? a.xps : 0;
            any |= ok;
        }
        // tiles no tap reaches only exist where the padding is deep (ASPP rates); elsewhere the workgroup-wide vote (a barrier
        // and an LDS round trip per tap, with every wave waiting) would buy nothing: a tap that reads only the zero row just
        // multiplies zeros
        if (a.pad < 4 && a.stride == 1) return true;
        return __syncthreads_or(any) != 0;
    };
    int tap = -1, cc = nCC - 1;
    auto next_in_tile = [&]() __attribute__((always_inline)) -> bool {       // advance (tap, cc) to the next 64-channel stage of this tile with work
        if (++cc < nCC) return true;
        cc = 0;
        do {
            if (++tap >= taps) return false;
        } while (!setup_tap(tap));
        return true;
    };

    // LDS-DMA of the whole stage the pointers stand on (8-row x 128-B pieces), then advance the pointers
    auto issueA = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NRG; ++i) {
            if (wave + 8 * i < RG) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    glds16b(aptr[i] + p * pst[i], lds_base + st * STAGE + p * PLANE + (wave + 8 * i) * 1024);
            }
            aptr[i] += astep[i];
        }
    };
    struct BFrag {
        uint4 v[2][NP];      // [32-deep half of the stage][plane]
    };
    auto bload = [&](BFrag& b, int k32) __attribute__((always_inline)) {
        const uint4* p = wpk + (size_t)k32 * (64 * NP);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) b.v[h][pl] = p[(h * NP + pl) * 64];
    };

    f32x4 acc[RBW];
#pragma unroll
    for (int i = 0; i < RBW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment address of this lane inside a plane: row (lane & 15) of a 16-row block, k group (lane >> 4) [+4 in the
    // second half of the stage -> slot ^ 4 -> byte ^ 64]
    const int fbase = wm * (RBW * 2048) + (lane & 15) * 128 + (((lane >> 4) ^ ((lane & 15) >> 1)) * 16);
    struct AFrag {
        uint4 v[NP];
    };
    // Multiply stage `st`; when `more`, the loads of the FOLLOWING stage (weight fragments into bn, activation DMA into stage
    // buffer st ^ 1) are issued one or two at a time BETWEEN the row blocks.  Issued in one burst at the top of the stage
    // they serialise with the multiply: with one workgroup per CU the vector-memory path takes ~34 cycles per 1-KB
    // instruction (30 B/clk/CU, tools/ta_bw.hip) and a wave sits in the issue queue instead of feeding the matrix pipe.
    auto compute = [&](int st, const BFrag& b, bool more, BFrag& bn, int k32n) __attribute__((always_inline)) {
        // fragments of row block i+1 are read while block i is multiplied; the scheduling fences keep hipcc from
        // hoisting all 2 * RBW * NP fragment reads of a stage to its top (216 VGPRs for RBW = 9)
        auto aload = [&](AFrag& f, int idx) __attribute__((always_inline)) {          // idx = half * RBW + rb
            if (ABL & 8) return;                    // ablation: multiply whatever the registers hold
            const int half = idx / RBW, rb = idx - half * RBW;
            const unsigned char* p = smem + st * STAGE + (fbase ^ (half * 64)) + rb * 2048;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) f.v[pl] = *reinterpret_cast<const uint4*>(p + pl * PLANE);
        };
        auto mul = [&](const AFrag& f, int idx) __attribute__((always_inline)) {
            const int half = idx / RBW, rb = idx - half * RBW;
            f32x4 c = acc[rb];
            if constexpr (NP == 3) {
                // weights are the MFMA's row operand: lane = pixel, 4 registers = 4 consecutive channels (16-byte stores)
                c = mfma16(b.v[half][0], f.v[2], c);     // smallest terms first: bh*al, bl*ah, bm*am, bh*am, bm*ah, bh*ah
                c = mfma16(b.v[half][2], f.v[0], c);
                c = mfma16(b.v[half][1], f.v[1], c);
                c = mfma16(b.v[half][0], f.v[1], c);
                c = mfma16(b.v[half][1], f.v[0], c);
            }
            c = mfma16(b.v[half][0], f.v[0], c);
            acc[rb] = c;
        };
        constexpr int NB_SLOTS = 2 * NP, NA_SLOTS = NRG * NP, NSLOTS = NB_SLOTS + NA_SLOTS;
        // spread evenly over the row blocks (front-loading them -- two per block over the first half of the stage -- measured
        // 8-12 % slower: the issue burst is what hurts)
        constexpr int PER = (NSLOTS + 2 * RBW - 1) / (2 * RBW);          // load slots per row block
        // The following stage's loads are issued unconditionally -- a scalar branch per row block costs more than the load it
        // would skip once per tile stream (tools/mfma_rate.hip): after the last stage they re-read the current weight block
        // and DMA the zero row into the idle stage buffer (drained by the s_waitcnt at the end of the kernel).
        const uint4* wn_ = wpk + (size_t)k32n * (64 * NP);
        const unsigned char* asrc[NRG];
        long long apl[NRG];
#pragma unroll
        for (int i = 0; i < NRG; ++i) {
            asrc[i] = more ? aptr[i] : zrow;
            apl[i] = more ? pst[i] : 0;
        }
        auto slot = [&](int sidx) __attribute__((always_inline)) {
            if (sidx < NB_SLOTS) {
                if (!(ABL & 1)) bn.v[sidx / NP][sidx % NP] = wn_[sidx * 64];
            } else if (sidx < NSLOTS) {
                const int i = (sidx - NB_SLOTS) / NP, pp = (sidx - NB_SLOTS) % NP;
                if (!(ABL & 2)) {
                    if (8 * i + 8 <= RG || wave + 8 * i < RG)
                        glds16b(asrc[i] + pp * apl[i], lds_base + (st ^ 1) * STAGE + pp * PLANE + (wave + 8 * i) * 1024);
                }
                if (pp == NP - 1) aptr[i] += astep[i];          // this row group's pointer moves on to the stage after
            }
        };
        // fragments are read TWO row blocks ahead of their multiply, with the order pinned: left to itself hipcc sinks two
        // of a block's three reads behind the 4th MFMA of the previous block and waits for them (lgkmcnt(0)) two MFMAs
        // later -- 32 cycles of cover for a ~100-cycle LDS round trip, at every row block
        AFrag f[3];
        if (ABL & 8) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) f[i].v[pl] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
        }
        aload(f[0], 0);
        if (2 * RBW > 1) aload(f[1], 1);
#pragma unroll
        for (int idx = 0; idx < 2 * RBW; ++idx) {
            if (idx + 2 < 2 * RBW) aload(f[(idx + 2) % 3], idx + 2);
            __builtin_amdgcn_sched_barrier(0);
            mul(f[idx % 3], idx);
#pragma unroll
            for (int q = 0; q < PER; ++q) slot(idx * PER + q);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- epilogue of the tile (m0, n0).  D = W . X^T per 16x16 block: lane -> pixel (lane & 15) of the row block, its 4
    // registers -> channels 16*wn + 4*(lane >> 4) + r: one 16-byte store per row block
    const int lq = lane >> 4, lp = lane & 15;
    auto epilogue = [&](int tile, int m0, int n0, bool zero) __attribute__((always_inline)) {     // zero: a tile no tap reaches
        int mt = tile / a.NT;
        if (par) mt = par_mt(mt);
        const int col = n0 + 16 * wn + 4 * lq;
        const bool cok = col < NC;
        const bool bnf = DGRAD && a.bnf.part != nullptr;
        const bool mk = bnf && a.bnf.relu == 3;
        if (DGRAD && a.accumulate && zero && !bnf) return;           // ... adds nothing
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!DGRAD && a.bias != nullptr && cok) bv = *reinterpret_cast<const float4*>(a.bias + col);
        if (DGRAD && par) {
            __syncthreads();
            if (t < BM) {
                const int m = m0 + t;
                int n = 0, rh = 0, rw = 0;
                if (m < a.M) x6_row_pixel(m, a.N, RH, RW, true, n, rh, rw);
                rowpix[DGRAD ? t : 0] = (n * RH + rh) * RW + rw;
            }
            __syncthreads();
        }
        float4 f_mu = bv, f_is = bv, f_sc = bv, f_sh = bv;
        float fs[4] = {0.f, 0.f, 0.f, 0.f}, fq[4] = {0.f, 0.f, 0.f, 0.f};
        if (bnf && cok) {
            f_mu = *reinterpret_cast<const float4*>(a.bnf.mean + col);
            f_is = *reinterpret_cast<const float4*>(a.bnf.invstd + col);
            if (a.bnf.relu == 2) {
                f_sc = *reinterpret_cast<const float4*>(a.bnf.mscale + col);
                f_sh = *reinterpret_cast<const float4*>(a.bnf.mshift + col);
            }
        }
        // everything the epilogue READS (the gradient accumulated so far, the producer's conv output for the fused
        // statistics) is fetched for all row blocks before the first store: hipcc cannot move a load above a store that
        // may alias it, and one load-then-store per row block costs a memory round trip per block
        // (in two halves of the row blocks: 2 x RBW x 4 more live registers do not fit beside the RBW = 10 accumulators)
        // BRANCH-FREE on purpose.  With a predicate around each store (`if (cok && row < M) *o = v`) hipcc gives every store a
        // basic block of its own, and its waitcnt pass -- conservative at the joins -- then puts `s_waitcnt vmcnt(0)` in front of
        // EVERY store of the main-loop epilogue (the bias / old-gradient loads it must wait for are younger than the next
        // stage's weight loads).  Stores count in vmcnt, so each store waited for the previous one to be acknowledged: 9
        // serial write round trips per tile, ~3 us -- a third of a K = 256 tile.  Out-of-range lanes now read a zero line and
        // store into a per-lane dump slot instead, selected by address: one block, one wait, back-to-back stores.
        const float4* const zero4 = reinterpret_cast<const float4*>(g_zero_row_pl2);
        const uint2* const zero2 = reinterpret_cast<const uint2*>(g_zero_row_pl2);
        float4* const dump = g_dump_pl2 + lane;
        const bool acc_old = DGRAD && a.accumulate;
        const bool relu2 = bnf && a.bnf.relu == 2;
        constexpr int EH = (RBW + 1) / 2;
#pragma unroll
        for (int h0 = 0; h0 < RBW; h0 += EH) {
            float4 oldv[EH], yvv[EH];
            uint2 mkv[EH];
            if constexpr (DGRAD) {
                // each kind of read under ONE uniform branch per batch (a plain data gradient issues none of them)
                int pixj[EH];
                bool okj[EH];
#pragma unroll
                for (int j = 0; j < EH; ++j) {
                    const int rb = h0 + j;
                    const int lrw = (wm * RBW + rb) * 16 + lp;
                    const int row = m0 + lrw;
                    okj[j] = rb < RBW && cok && row < a.M;
                    pixj[j] = okj[j] ? (par ? rowpix[lrw] : row) : 0;
                    oldv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    yvv[j] = oldv[j];
                    mkv[j] = make_uint2(0u, 0u);
                }
                if (acc_old) {
#pragma unroll
                    for (int j = 0; j < EH; ++j)
                        oldv[j] = *(okj[j] ? reinterpret_cast<const float4*>(&a.y[(size_t)pixj[j] * a.ldy + col]) : zero4);
                }
                if (bnf) {
#pragma unroll
                    for (int j = 0; j < EH; ++j)
                        yvv[j] = *(okj[j] ? reinterpret_cast<const float4*>(a.bnf.y + (size_t)pixj[j] * a.bnf.ldy + col) : zero4);
                }
                if (mk) {
#pragma unroll
                    for (int j = 0; j < EH; ++j)
                        mkv[j] = *(okj[j] ? reinterpret_cast<const uint2*>(a.bnf.mask + (size_t)pixj[j] * a.bnf.ldm + col) : zero2);
                }
            }
#pragma unroll
            for (int j = 0; j < EH; ++j) {
                const int rb = h0 + j;
                if (rb >= RBW) continue;
                const int lrw = (wm * RBW + rb) * 16 + lp;
                const int row = m0 + lrw;
                const bool ok = cok && row < a.M;
                const int pix = ok ? ((DGRAD && par) ? rowpix[DGRAD ? lrw : 0] : row) : 0;
                float4 v = zero ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(acc[rb][0], acc[rb][1], acc[rb][2], acc[rb][3]);
                float4 addv = bv;
                if constexpr (DGRAD) {
                    addv.x = acc_old ? oldv[j].x : bv.x; addv.y = acc_old ? oldv[j].y : bv.y;
                    addv.z = acc_old ? oldv[j].z : bv.z; addv.w = acc_old ? oldv[j].w : bv.w;
                }
                v.x += addv.x; v.y += addv.y; v.z += addv.z; v.w += addv.w;
                if constexpr (DGRAD) {
                    // residual producer (mk): the pattern comes from the hi plane of its saved output (as k_bn_bwd_reduce<1> reads
                    // it); what is stored is the masked gradient
                    const float4 ov = bf16x4_to_f32(mkv[j]);
                    v.x = (!mk || ov.x > 0.f) ? v.x : 0.f; v.y = (!mk || ov.y > 0.f) ? v.y : 0.f;
                    v.z = (!mk || ov.z > 0.f) ? v.z : 0.f; v.w = (!mk || ov.w > 0.f) ? v.w : 0.f;
                }
                const bool st = ok && (mk || !(acc_old && zero));
                float4* o = st ? reinterpret_cast<float4*>(&a.y[(size_t)pix * a.ldy + col]) : dump;
                *o = v;
                if constexpr (DGRAD) {
                    // same expressions as k_bn_bwd_reduce (bn.hip): the ReLU pattern from the forward's own formula
                    const float4 yv = yvv[j];
                    float4 g = v;
                    g.x = (!relu2 || (yv.x - f_mu.x) * f_sc.x + f_sh.x > 0.f) ? g.x : 0.f;
                    g.y = (!relu2 || (yv.y - f_mu.y) * f_sc.y + f_sh.y > 0.f) ? g.y : 0.f;
                    g.z = (!relu2 || (yv.z - f_mu.z) * f_sc.z + f_sh.z > 0.f) ? g.z : 0.f;
                    g.w = (!relu2 || (yv.w - f_mu.w) * f_sc.w + f_sh.w > 0.f) ? g.w : 0.f;
                    g.x = ok ? g.x : 0.f; g.y = ok ? g.y : 0.f; g.z = ok ? g.z : 0.f; g.w = ok ? g.w : 0.f;
                    fs[0] += g.x; fs[1] += g.y; fs[2] += g.z; fs[3] += g.w;
                    fq[0] += g.x * ((yv.x - f_mu.x) * f_is.x); fq[1] += g.y * ((yv.y - f_mu.y) * f_is.y);
                    fq[2] += g.z * ((yv.z - f_mu.z) * f_is.z); fq[3] += g.w * ((yv.w - f_mu.w) * f_is.w);
                }
            }
        }
        if (bnf) {
            // <= RBW pixels per lane in fp32, the 16 lanes of a channel quad and everything after in double
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
                const size_t T = (size_t)a.MT * WM, prow = (size_t)mt * WM + wm;
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
                const bool ok = !zero && m0 + (wm * RBW + rb) * 16 + lp < a.M;
#pragma unroll
                for (int r = 0; r < 4; ++r) s[r] += ok ? acc[rb][r] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[r] += __shfl_xor(s[r], 1);
                s[r] += __shfl_xor(s[r], 2);
                s[r] += __shfl_xor(s[r], 4);
                s[r] += __shfl_xor(s[r], 8);
            }
            if constexpr (WM > 1) {                  // columns are shared by WM waves: combine through LDS
                __syncthreads();
                if (lp == 0) *reinterpret_cast<float4*>(&red[wm * 128 + wn * 16 + 4 * lq]) = make_float4(s[0], s[1], s[2], s[3]);
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 4; ++r) s[r] = 0.f;
#pragma unroll
                for (int k = 0; k < WM; ++k)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[r] += red[k * 128 + wn * 16 + 4 * lq + r];
            }
            float qv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) {
                const bool ok = !zero && m0 + (wm * RBW + rb) * 16 + lp < a.M;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dv = acc[rb][r] - s[r] / (float)cnt;
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
            if constexpr (WM > 1) {
                __syncthreads();
                if (lp == 0) *reinterpret_cast<float4*>(&red[wm * 128 + wn * 16 + 4 * lq]) = make_float4(qv[0], qv[1], qv[2], qv[3]);
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 4; ++r) qv[r] = 0.f;
#pragma unroll
                for (int k = 0; k < WM; ++k)
#pragma unroll
                    for (int r = 0; r < 4; ++r) qv[r] += red[k * 128 + wn * 16 + 4 * lq + r];
            }
            if (lp == 0 && cok && wm == 0) {
                *reinterpret_cast<float4*>(&a.stats[(size_t)mt * a.Cout + col]) = make_float4(s[0], s[1], s[2], s[3]);
                *reinterpret_cast<float4*>(&a.stats[(size_t)(a.MT + mt) * a.Cout + col]) = make_float4(qv[0], qv[1], qv[2], qv[3]);
            }
        }
        if (!zero) {
#pragma unroll
            for (int i = 0; i < RBW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    // ---- the stage stream: one pipeline step per 64-channel stage, running across this workgroup's tiles
    int c_tile = i_tile, c_m0, c_n0;          // compute side: the tile the accumulators belong to
    load_tile(i_tile);
    c_m0 = i_m0;
    c_n0 = i_n0;
    auto next_tile = [&]() __attribute__((always_inline)) -> bool {          // issue side: step to this workgroup's next tile
        i_tile += gridDim.x;
        if (i_tile >= ntiles) return false;
        load_tile(i_tile);
        tap = -1;
        cc = nCC - 1;
        return true;
    };
    // move the issue side to the first stage of the next tile that has one; tiles without any stage (no tap reaches
    // them: odd-parity tiles of a strided 1x1 data gradient) get their zero epilogue on the way
    auto next_tile_stage = [&]() __attribute__((always_inline)) -> bool {
        for (;;) {
            if (!next_tile()) return false;
            if (next_in_tile()) return true;
            epilogue(i_tile, i_m0, i_n0, true);
        }
    };
    // diagnostic build: shader clock and 100 MHz reference clock at both ends of workgroup 0 -> the clock the chip actually
    // held while this kernel ran (tools/pl2_timeline.py; the chip lowers it under matrix load, MI355X_MICROARCH.md 'DVFS')
    if (DBG && a.dbg != nullptr && blockIdx.x == 0 && t == 0) {
        a.dbg[500] = __builtin_amdgcn_s_memtime();
        a.dbg[501] = __builtin_amdgcn_s_memrealtime();
    }
    {
        BFrag bc, bn;
        bool have = next_in_tile();
        if (!have) {
            epilogue(i_tile, i_m0, i_n0, true);
            have = next_tile_stage();
            c_tile = i_tile; c_m0 = i_m0; c_n0 = i_n0;
        }
        if (have) {
            bload(bc, tap * (GC >> 5) + 2 * cc);
            issueA(0);
        }
        int st = 0;
        // one step per stage: publish stage `st`, start the loads of the following stage (possibly of the next tile),
        // multiply stage `st`, and run the epilogue when it was the last stage of its tile
        int dbg_n = 0;
        // stamps exist only in the diagnostic instantiation (DBG): even as untaken branches they sit between the barrier
        // and the first MFMA of every stage, where nothing overlaps them (tools/mfma_rate.hip)
        const bool dbg = DBG && a.dbg != nullptr && blockIdx.x == 0 && (wave == 0 || wave == 4);
        auto stamp = [&](int k) __attribute__((always_inline)) {
            if constexpr (DBG) {
                if (dbg && dbg_n < 40 && lane == 0) a.dbg[(wave ? 256 : 0) + dbg_n * 6 + k] = __builtin_amdgcn_s_memtime();
            }
        };
        while (have) {
            stamp(0);
            __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): this wave's DMA pieces and weight fragments landed
            stamp(1);
            if (!(ABL & 4)) __builtin_amdgcn_s_barrier();                // ... everyone's did; the other stage buffer is free
            asm volatile("" ::: "memory");
            stamp(2);
            bool more = next_in_tile();
            const bool last = !more;                      // the stage in hand is the last of its tile
            if (last) more = next_tile_stage();
            stamp(3);
            compute(st, bc, more, bn, more ? tap * (GC >> 5) + 2 * cc : 0);
            stamp(4);
            if (last) {
                epilogue(c_tile, c_m0, c_n0, false);
                c_tile = i_tile; c_m0 = i_m0; c_n0 = i_n0;
            }
            st ^= 1;
            bc = bn;
            stamp(5);
            ++dbg_n;
            have = more;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the idle loads of the last stage
    if (DBG && a.dbg != nullptr && blockIdx.x == 0 && t == 0) {
        a.dbg[502] = __builtin_amdgcn_s_memtime();
        a.dbg[503] = __builtin_amdgcn_s_memrealtime();
    }
}

// Tile height for a GEMM of M rows x cols columns on 256 CUs (one 128-column workgroup per CU at a time):
// rbw 16-row blocks, chosen to minimise  rounds x (rows per tile + fixed per-tile cost in row equivalents).
int conv_pl2_pick_rbw(int64_t M, int cols) {
    if (const char* e = getenv("ISWM_PL2_RBW")) {
        const int v = atoi(e);
        if (v >= 8 && v <= PL2_RBWMAX && !(cols <= 64 && v == 9)) return v;      // 64-column tiles have no 9-block form
    }
    const int64_t NT = cols <= 64 ? 1 : (cols + 127) / 128;
    int best = PL2_RBWMAX;
    double bestc = 1e300;
    for (int rbw = 8; rbw <= PL2_RBWMAX; ++rbw) {
        if (cols <= 64 && rbw == 9) continue;        // 64-column tiles: 2 wave rows x 4 or 5 blocks
        const int64_t MT = (M + rbw * 16 - 1) / (rbw * 16);
        const int64_t rounds = (MT * NT + 255) / 256;
        const double c = (double)rounds * (rbw * 16 + 24.0);
        if (c < bestc - 1e-9) {
            bestc = c;
            best = rbw;
        }
    }
    return best;
}

// Tile plan: height (rbw 16-row blocks) and width.  256-column tiles (k_conv_pl2w: two column blocks per wave, ~15 % less
// time per MFMA) are taken where the tile count still covers the chip:  cost = rounds x (rows + fixed) x (2 x 0.85 if wide).
void conv_pl2_plan(int64_t M, int cols, int K, bool wide_ok, int* rbw_out, int* wide_out) {
    static int force = -2;
    static double eff = 0.85;
    if (force == -2) {
        force = getenv("ISWM_PL2_WIDE") ? atoi(getenv("ISWM_PL2_WIDE")) : -1;        // 0: never, 1: wherever possible
        if (const char* e = getenv("ISWM_PL2W_EFF")) eff = atof(e) > 0 ? atof(e) : eff;
    }
    *rbw_out = conv_pl2_pick_rbw(M, cols);
    *wide_out = 0;
    // short K: a tile is a few stages and then an epilogue of twice the size -- measured slower below 4 stages (forward)
    // / 8 stages (data gradient, whose accumulating epilogue also reads) [profiles/r03_pl2w_ab.txt]
    if (!wide_ok || cols < 256 || force == 0 || (force != 1 && K < 256)) return;
    const int64_t NT = (cols + 127) / 128, NTW = (cols + 255) / 256;
    double best = 1e300;
    {
        const int rbw = *rbw_out;
        const int64_t MT = (M + rbw * 16 - 1) / (rbw * 16);
        best = (double)((MT * NT + 255) / 256) * (rbw * 16 + 24.0);
        if (force == 1) best = 1e300;
    }
    for (int rbw = 8; rbw <= PL2_RBWMAX; ++rbw) {
        const int64_t MT = (M + rbw * 16 - 1) / (rbw * 16);
        const double c = (double)((MT * NTW + 255) / 256) * (rbw * 16 + 24.0) * 2.0 * eff;
        if (c < best - 1e-9) {
            best = c;
            *rbw_out = rbw;
            *wide_out = 1;
        }
    }
}

// rbw = 16-row blocks per tile (8..10 instantiated for 128-column tiles; 2 x 4/5 for 64-column tiles)
bool launch_conv_pl2(ConvArgs a, hipStream_t s, bool dgrad, int planes, int rbw) {
    static int parity = -1, ncu = 0;
    if (parity < 0) {
        const char* e = getenv("ISWM_X6_PARITY");
        parity = (e && e[0] == '0') ? 0 : 1;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
        if (const char* g = getenv("ISWM_PL2_GRID")) ncu = atoi(g) > 0 ? atoi(g) : ncu;
    }
    a.psplit = parity ? 0 : 1;
    {   // quarters of the parity-sorted rows, heaviest first (class k = 2 * (row parity) + column parity)
        int wgt[4], ord[4] = {0, 1, 2, 3};
        for (int k = 0; k < 4; ++k) {
            int ch = 0, cw = 0;
            for (int kh = 0; kh < a.KH; ++kh) ch += (((k >> 1) + a.pad - kh * a.dil) & 1) == 0;
            for (int kw = 0; kw < a.KW; ++kw) cw += (((k & 1) + a.pad - kw * a.dil) & 1) == 0;
            wgt[k] = ch * cw;
        }
        for (int i = 0; i < 4; ++i)
            for (int j = i + 1; j < 4; ++j)
                if (wgt[ord[j]] > wgt[ord[i]]) { const int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
        a.porder = ord[0] | (ord[1] << 2) | (ord[2] << 4) | (ord[3] << 6);
    }
    a.dbg = g_conv_dbg;
    static int abl = -1;
    if (abl < 0) abl = getenv("ISWM_PL2_ABL") ? atoi(getenv("ISWM_PL2_ABL")) : 0;
    a.abl = abl;
    const int nc = dgrad ? a.Cin : a.Cout;
    const bool narrow = nc <= 64;
    a.MT = (a.M + rbw * 16 - 1) / (rbw * 16);
    a.NT = narrow ? 1 : (nc + 127) / 128;
    const int tiles = a.MT * a.NT;
    dim3 grid(tiles < ncu ? tiles : ncu), blk(512);
    if (planes == 1) {          // ONE bf16 plane per operand (conv math "bf16": activations stored rounded, one MFMA per product)
#define PL2_LAUNCH1(R, W)                                                                      \
    do {                                                                                       \
        if (dgrad) hipLaunchKernelGGL((k_conv_pl2<R, W, 1, true>), grid, blk, 0, s, a);         \
        else hipLaunchKernelGGL((k_conv_pl2<R, W, 1, false>), grid, blk, 0, s, a);              \
    } while (0)
        if (narrow) {
            if (rbw == 8) PL2_LAUNCH1(4, 2);
            else if (rbw == 10) PL2_LAUNCH1(5, 2);
            else return false;
        } else {
            if (rbw == 8) PL2_LAUNCH1(8, 1);
            else if (rbw == 9) PL2_LAUNCH1(9, 1);
            else if (rbw == 10) PL2_LAUNCH1(10, 1);
            else return false;
        }
#undef PL2_LAUNCH1
        return true;
    }
    if (planes != 3) return false;
#define PL2_LAUNCH(R, W)                                                                       \
    do {                                                                                       \
        if (dgrad) hipLaunchKernelGGL((k_conv_pl2<R, W, 3, true>), grid, blk, 0, s, a);         \
        else hipLaunchKernelGGL((k_conv_pl2<R, W, 3, false>), grid, blk, 0, s, a);              \
    } while (0)
    if (narrow) {
        if (rbw == 8) PL2_LAUNCH(4, 2);
        else if (rbw == 10) PL2_LAUNCH(5, 2);
        else return false;
    } else {
        if (rbw == 8) PL2_LAUNCH(8, 1);
        else if (rbw == 9 && a.abl != 0) {               // ISWM_PL2_ABL: timing ablations are compile-time variants
#define PL2_ABL_LAUNCH(A)                                                                                        \
    do {                                                                                                         \
        if (dgrad) hipLaunchKernelGGL((k_conv_pl2<9, 1, 3, true, false, A>), grid, blk, 0, s, a);                 \
        else hipLaunchKernelGGL((k_conv_pl2<9, 1, 3, false, false, A>), grid, blk, 0, s, a);                      \
    } while (0)
            if (a.abl == 1) PL2_ABL_LAUNCH(1);
            else if (a.abl == 2) PL2_ABL_LAUNCH(2);
            else if (a.abl == 3) PL2_ABL_LAUNCH(3);
            else if (a.abl == 7) PL2_ABL_LAUNCH(7);
            else if (a.abl == 15) PL2_ABL_LAUNCH(15);
            else return false;
#undef PL2_ABL_LAUNCH
        } else if (rbw == 9 && a.dbg != nullptr) {          // iswm_set_debug_buffer: the stamped instantiation
            if (dgrad) hipLaunchKernelGGL((k_conv_pl2<9, 1, 3, true, true>), grid, blk, 0, s, a);
            else hipLaunchKernelGGL((k_conv_pl2<9, 1, 3, false, true>), grid, blk, 0, s, a);
        } else if (rbw == 9) PL2_LAUNCH(9, 1);
        else if (rbw == 10) PL2_LAUNCH(10, 1);
        else return false;
    }
#undef PL2_LAUNCH
    return true;
}

// Weight packing for k_conv_pl2: packed[((cb * K32 + k32) * NP + plane) * 64 + lane] (uint4) holds, for column
// cb*16 + (lane & 15), the 8 bf16 of that plane at k = k32*32 + 8*(lane >> 4) .. +7  (the 16x16x32 B fragment).
//   fwd  : column = cout, k = (tap, cin);   dgrad: column = cin, k = (tap, cout) (implicit transpose).
// Columns >= NC are zero; cb runs to ceil(NC/128)*8.
template <bool DGRAD, int NP>
__global__ __launch_bounds__(256) void k_pack_weights_pl2(const float* __restrict__ w, uint4* __restrict__ packed, int Cout,
                                                          int T, int Cin, int K32, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    pack_weights_pl2_body<DGRAD, NP>(w, packed, Cout, T, Cin, K32, idx);
}

int pack_job_blocks_pl2(int Cout, int T, int Cin, bool dgrad) {
    const int NC = dgrad ? Cin : Cout, GC = dgrad ? Cout : Cin;
    const long long total = (long long)((NC + 127) / 128) * 8 * (T * GC / 32) * 64;
    return (int)((total + 255) / 256);
}

size_t packed_weight_bytes_pl2(int Cout, int T, int Cin, bool dgrad, int planes) {
    const int NC = dgrad ? Cin : Cout, GC = dgrad ? Cout : Cin;
    const size_t cbs = (size_t)((NC + 127) / 128) * 8, K32 = (size_t)T * GC / 32;
    return cbs * K32 * 64 * planes * sizeof(uint4);
}

void launch_pack_weights_pl2(const float* w, void* packed, int Cout, int T, int Cin, bool dgrad, int planes, hipStream_t s) {
    const int NC = dgrad ? Cin : Cout, GC = dgrad ? Cout : Cin;
    const int cbs = ((NC + 127) / 128) * 8, K32 = T * GC / 32;
    const int total = cbs * K32 * 64;
    dim3 grid((total + 255) / 256), blk(256);
    uint4* o = (uint4*)packed;
    if (planes == 1) {
        if (dgrad) hipLaunchKernelGGL((k_pack_weights_pl2<true, 1>), grid, blk, 0, s, w, o, Cout, T, Cin, K32, total);
        else hipLaunchKernelGGL((k_pack_weights_pl2<false, 1>), grid, blk, 0, s, w, o, Cout, T, Cin, K32, total);
    } else {
        if (dgrad) hipLaunchKernelGGL((k_pack_weights_pl2<true, 3>), grid, blk, 0, s, w, o, Cout, T, Cin, K32, total);
        else hipLaunchKernelGGL((k_pack_weights_pl2<false, 3>), grid, blk, 0, s, w, o, Cout, T, Cin, K32, total);
    }
}

}  // namespace iswm
