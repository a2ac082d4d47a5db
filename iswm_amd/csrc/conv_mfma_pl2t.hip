// ASPP as ONE launch: the parallel branches of network/_deeplab.py:143-172 (1x1 + three atrous 3x3 convolutions over the
// same 2048-channel map) run from one tile table -- iswm_aspp_fwd / iswm_aspp_bwd.
//
// What the separate launches lose (profiles/r02_conv_table.txt): each branch is 242 tiles on 256 CUs -- ONE tile per CU for
// the whole kernel -- so skipping the taps that only read padding helps a launch only as far as its slowest tile; with
// row-run tiles (144 consecutive pixels = 4.4 image rows) a tap is dead for a tile only when whole ROWS are out of reach,
// which leaves 64 / 76 / 88 % of the nominal work at rates 18 / 12 / 6 where 40 / 57 / 77 % is in bounds; and the data
// gradient of the four branches reads and rewrites the 2048-channel input gradient three times.
// Here:
//   * ROWS ARE SORTED BY TAP SET.  For a filter with pad = dilation a pixel's set of in-bounds taps is the product of a row
//     class and a column class (three each), so the pixels of the batch fall into <= 9 classes per branch; the plan
//     (aspp_plan) orders the GEMM rows class by class and records, per 144-row tile, the union of its rows' tap sets.  A tile
//     then runs exactly its taps' stages -- no vote, no padding multiplied except in the few tiles that straddle two classes.
//   * ONE TILE TABLE over all branches, heaviest tiles first and dealt in serpentine order to the persistent grid: ~970 tiles
//     of 1 .. 9 taps (forward) balance where 242 could not.
//   * THE DATA GRADIENT IS ONE GEMM: dx = sum over the 28 taps of the four branches; the gathered operand is the
//     concatenated gradient buffer [P][4 x 256] and a tap names its channel slice.  dx is written once.
// The stage loop is k_conv_pl2's (conv_mfma_pl2.hip) at RBW = 9: LDS-DMA of 64-channel activation rows with a source-side
// swizzle, weight fragments from packed global memory one stage ahead, v_mfma_f32_16x16x32_bf16, pinned fragment reads.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "conv_common.h"

namespace iswm {

static __device__ __attribute__((aligned(128))) unsigned short g_zero_row_pl2t[64];
static __device__ float4 g_dump_pl2t[64];        // where the epilogue's out-of-range lanes store (never read)

typedef __attribute__((address_space(3))) void* lds_vptr2t;
typedef float f32x4t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16bt(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

__device__ __forceinline__ f32x4t mfma16t(uint4 a, uint4 b, f32x4t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int ASPP_MAXB = 4;          // branches
constexpr int ASPP_MAXT = 32;         // taps over all branches (1 + 9 + 9 + 9 = 28)
constexpr int ASPP_RBW = 9;           // 144-row tiles
constexpr int ASPP_MAGIC = 0x41535031;

// ---- the plan (host-built, copied to the device by the caller; identical bytes on both sides) ----
struct AsppTap {
    int dh, dw;          // gathered pixel = row pixel + (dh, dw)
    int branch;          // whose weights / whose channel slice of the gathered tensor (data gradient)
    int k32base;         // first 32-deep k block of this tap inside the branch's packed weights
};
struct AsppJob {
    int tap_begin, ntaps;    // this job's taps in AsppPlan::taps
    int branch;              // forward: the branch this job computes (its weights, output, statistics); data gradient: -1
    int rowmap_off;          // first entry of this job's row -> pixel map (ints from AsppPlan::rowmap_off)
    int tile_begin_unused;
    int pad0, pad1, pad2;
};
struct AsppPlan {
    int magic, kind;         // kind 0 forward, 1 data gradient
    int nbranch, ntaps, njobs, ntiles;
    int N, H, W, M, MT, NT;  // rows = N*H*W pixels, 144-row tiles, 128-column tiles per job
    int GC, NC;              // gathered channels per tap, output columns per job
    int rowmap_off, tiles_off;   // byte offsets of int rowmap[njobs][M] and int4 tiles[ntiles] from the start of the plan
    int total_bytes, pad;
    AsppTap taps[ASPP_MAXT];
    AsppJob jobs[ASPP_MAXB];
};

struct AsppArgs {
    const AsppPlan* plan;
    const unsigned short* x;      // plane 0 of the gathered tensor
    long long xps;                // plane stride in BYTES
    int ldx;                      // its pixel pitch (bf16 elements)
    int goff[ASPP_MAXB];          // data gradient: channel offset of branch b's slice in the gathered tensor (forward: 0)
    const uint4* w[ASPP_MAXB];    // packed weights per branch (k_pack_weights_pl2 layout)
    int K32[ASPP_MAXB];           // 32-deep k blocks per column block of branch b's packed weights
    float* y[ASPP_MAXB];          // forward: output per job; data gradient: y[0] = dx
    float* stats[ASPP_MAXB];      // forward: [2][MT][NC] BatchNorm partials per job (may be null)
    int ldy;                      // output pixel pitch (floats)
    int accumulate;               // data gradient: dx += result
};

// PERSISTENT: grid = min(tiles, CUs); workgroup g walks table entries g, g + grid, ...
template <bool DGRAD>
__global__ __launch_bounds__(512, 2) void k_conv_pl2t(const AsppArgs a) {
    constexpr int RBW = ASPP_RBW, NP = 3;
    constexpr int BM = 16 * RBW, RG = BM / 8;
    constexpr int PLANE = BM * 128;
    constexpr int STAGE = NP * PLANE;
    constexpr int NRG = (RG + 7) / 8;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_vptr2t)smem;
    const AsppPlan& P = *a.plan;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wn = wave;
    const int GH = P.H, GW = P.W;
    const int nCC = P.GC >> 6;
    const int ntiles = P.ntiles;
    const int* rowmap = reinterpret_cast<const int*>(reinterpret_cast<const unsigned char*>(a.plan) + P.rowmap_off);
    const int4* tiles = reinterpret_cast<const int4*>(reinterpret_cast<const unsigned char*>(a.plan) + P.tiles_off);
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x);
    const int gs = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    const unsigned char* zrow = reinterpret_cast<const unsigned char*>(g_zero_row_pl2t) + gs * 16;

    // ---- issue side
    // workgroups b and b + 8 share an XCD: give every XCD a contiguous run of table entries -- the column tiles of one row
    // tile (adjacent entries: same activation rows) then share one L2 instead of being dealt across the eight
    int i_tile = xcd_remap(blockIdx.x, gridDim.x);
    int i_job = 0, i_mt = 0, i_m0 = 0, i_n0 = 0;
    unsigned i_mask = 0;                 // taps of the tile still to run
    int ihb[NRG], iwb[NRG], pb[NRG];
    auto load_tile = [&](int ti) __attribute__((always_inline)) {
        const int4 e = tiles[ti];
        i_job = __builtin_amdgcn_readfirstlane(e.x);
        i_mt = __builtin_amdgcn_readfirstlane(e.y);
        i_n0 = __builtin_amdgcn_readfirstlane(e.z) * 128;
        i_mask = (unsigned)__builtin_amdgcn_readfirstlane(e.w);
        i_m0 = i_mt * BM;
        const int* rm = rowmap + P.jobs[i_job].rowmap_off;
#pragma unroll
        for (int i = 0; i < NRG; ++i) {
            const int m = i_m0 + 8 * (wave + 8 * i) + (lane >> 3);
            if (wave + 8 * i < RG && m < P.M) {
                const int code = rm[m];
                ihb[i] = (code >> 10) & 1023;
                iwb[i] = code & 1023;
                pb[i] = (code >> 20) * GH * GW;
            } else {
                ihb[i] = -(1 << 28);
                iwb[i] = 0;
                pb[i] = 0;
            }
        }
    };
    const unsigned char* aptr[NRG];
    int astep[NRG];
    long long pst[NRG];
    const uint4* wtap = nullptr;         // issue side: packed weights of the tap in hand for this wave's column block
    auto setup_tap = [&](int tg) __attribute__((always_inline)) {
        const AsppTap tp = P.taps[tg];
        const int b = tp.branch;
        const int goff = DGRAD ? a.goff[b] : 0;
#pragma unroll
        for (int i = 0; i < NRG; ++i) {
            const int gh = ihb[i] + tp.dh, gw = iwb[i] + tp.dw;
            const bool ok = (unsigned)gh < (unsigned)GH && (unsigned)gw < (unsigned)GW;
            aptr[i] = ok ? xb + ((size_t)(pb[i] + gh * GW + gw) * a.ldx + goff) * 2 + gs * 16 : zrow;
            astep[i] = ok ? 128 : 0;
            pst[i] = ok ? a.xps : 0;
        }
        wtap = a.w[b] + ((size_t)((i_n0 >> 4) + wn) * a.K32[b] + tp.k32base) * (64 * NP) + lane;
    };
    int cc = nCC - 1;
    auto next_in_tile = [&]() __attribute__((always_inline)) -> bool {       // advance to the next 64-channel stage of this tile
        if (++cc < nCC) return true;
        cc = 0;
        if (i_mask == 0) return false;
        const int bit = __builtin_ctz(i_mask);
        i_mask &= i_mask - 1;
        setup_tap(P.jobs[i_job].tap_begin + bit);
        return true;
    };
    auto issueA = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NRG; ++i) {
            if (wave + 8 * i < RG) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    glds16bt(aptr[i] + p * pst[i], lds_base + st * STAGE + p * PLANE + (wave + 8 * i) * 1024);
            }
            aptr[i] += astep[i];
        }
    };
    struct BFrag {
        uint4 v[2][NP];
    };
    auto bload = [&](BFrag& b, const uint4* p) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) b.v[h][pl] = p[(h * NP + pl) * 64];
    };

    f32x4t acc[RBW];
#pragma unroll
    for (int i = 0; i < RBW; ++i) acc[i] = f32x4t{0.f, 0.f, 0.f, 0.f};

    const int fbase = (lane & 15) * 128 + (((lane >> 4) ^ ((lane & 15) >> 1)) * 16);
    struct AFrag {
        uint4 v[NP];
    };
    // multiply stage `st`; the following stage's loads (weights from wn_, activation DMA) are issued between the row blocks
    auto compute = [&](int st, const BFrag& b, bool more, BFrag& bn, const uint4* wn_) __attribute__((always_inline)) {
        auto aload = [&](AFrag& f, int idx) __attribute__((always_inline)) {
            const int half = idx / RBW, rb = idx - half * RBW;
            const unsigned char* p = smem + st * STAGE + (fbase ^ (half * 64)) + rb * 2048;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) f.v[pl] = *reinterpret_cast<const uint4*>(p + pl * PLANE);
        };
        auto mul = [&](const AFrag& f, int idx) __attribute__((always_inline)) {
            const int half = idx / RBW, rb = idx - half * RBW;
            f32x4t c = acc[rb];
            c = mfma16t(b.v[half][0], f.v[2], c);     // smallest terms first
            c = mfma16t(b.v[half][2], f.v[0], c);
            c = mfma16t(b.v[half][1], f.v[1], c);
            c = mfma16t(b.v[half][0], f.v[1], c);
            c = mfma16t(b.v[half][1], f.v[0], c);
            c = mfma16t(b.v[half][0], f.v[0], c);
            acc[rb] = c;
        };
        constexpr int NB_SLOTS = 2 * NP, NA_SLOTS = NRG * NP, NSLOTS = NB_SLOTS + NA_SLOTS;
        constexpr int PER = (NSLOTS + 2 * RBW - 1) / (2 * RBW);
        const unsigned char* asrc[NRG];
        long long apl[NRG];
#pragma unroll
        for (int i = 0; i < NRG; ++i) {
            asrc[i] = more ? aptr[i] : zrow;
            apl[i] = more ? pst[i] : 0;
        }
        auto slot = [&](int sidx) __attribute__((always_inline)) {
            if (sidx < NB_SLOTS) {
                bn.v[sidx / NP][sidx % NP] = wn_[sidx * 64];
            } else if (sidx < NSLOTS) {
                const int i = (sidx - NB_SLOTS) / NP, pp = (sidx - NB_SLOTS) % NP;
                if (8 * i + 8 <= RG || wave + 8 * i < RG)
                    glds16bt(asrc[i] + pp * apl[i], lds_base + (st ^ 1) * STAGE + pp * PLANE + (wave + 8 * i) * 1024);
                if (pp == NP - 1) aptr[i] += astep[i];
            }
        };
        AFrag f[3];
        aload(f[0], 0);
        aload(f[1], 1);
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

    // ---- epilogue of tile (job, mt, n0): lane -> row (lane & 15) of a row block, 4 registers -> 4 consecutive columns
    const int lq = lane >> 4, lp = lane & 15;
    auto epilogue = [&](int job, int mt, int n0, bool zero) __attribute__((always_inline)) {
        const int m0 = mt * BM;
        const int col = n0 + 16 * wn + 4 * lq;
        const bool cok = col < P.NC;
        const int* rm = rowmap + P.jobs[job].rowmap_off;
        float* yb = DGRAD ? a.y[0] : a.y[P.jobs[job].branch];
        if (DGRAD && a.accumulate && zero) return;
        // every read of the epilogue (row -> pixel map, the gradient accumulated so far) before its first store
        int pix[RBW];
        float4 oldv[RBW];
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) {
            const int row = m0 + rb * 16 + lp;
            pix[rb] = -1;
            oldv[rb] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < P.M) {
                const int code = rm[row];
                pix[rb] = ((code >> 20) * GH + ((code >> 10) & 1023)) * GW + (code & 1023);
            }
        }
        // branch-free (see k_conv_pl2's epilogue): out-of-range lanes read a zero line / store into a dump slot by address select
        const float4* const zero4 = reinterpret_cast<const float4*>(g_zero_row_pl2t);
        float4* const dump = g_dump_pl2t + lane;
        if constexpr (DGRAD) {
            const bool acc_old = a.accumulate != 0;
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) {
                const bool ok = cok && pix[rb] >= 0 && acc_old;
                const float4* po = ok ? reinterpret_cast<const float4*>(&yb[(size_t)(ok ? pix[rb] : 0) * a.ldy + col]) : zero4;
                oldv[rb] = *po;
            }
        }
#pragma unroll
        for (int rb = 0; rb < RBW; ++rb) {
            const bool ok = cok && pix[rb] >= 0;
            float4 v = zero ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(acc[rb][0], acc[rb][1], acc[rb][2], acc[rb][3]);
            v.x += oldv[rb].x; v.y += oldv[rb].y; v.z += oldv[rb].z; v.w += oldv[rb].w;
            float4* o = ok ? reinterpret_cast<float4*>(&yb[(size_t)(ok ? pix[rb] : 0) * a.ldy + col]) : dump;
            *o = v;
        }
        if (!DGRAD && a.stats[P.jobs[job].branch] != nullptr) {
            float* stp = a.stats[P.jobs[job].branch];
            const int cnt = min(BM, P.M - m0);
            float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) {
                const bool ok = !zero && pix[rb] >= 0;
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
            float qv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rb = 0; rb < RBW; ++rb) {
                const bool ok = !zero && pix[rb] >= 0;
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
            if (lp == 0 && cok) {
                *reinterpret_cast<float4*>(&stp[(size_t)mt * P.NC + col]) = make_float4(s[0], s[1], s[2], s[3]);
                *reinterpret_cast<float4*>(&stp[(size_t)(P.MT + mt) * P.NC + col]) = make_float4(qv[0], qv[1], qv[2], qv[3]);
            }
        }
        if (!zero) {
#pragma unroll
            for (int i = 0; i < RBW; ++i) acc[i] = f32x4t{0.f, 0.f, 0.f, 0.f};
        }
    };

    // ---- the stage stream across this workgroup's tiles
    int c_job, c_mt, c_n0;
    load_tile(i_tile);
    c_job = i_job; c_mt = i_mt; c_n0 = i_n0;
    auto next_tile = [&]() __attribute__((always_inline)) -> bool {
        i_tile += gridDim.x;
        if (i_tile >= ntiles) return false;
        load_tile(i_tile);
        cc = nCC - 1;
        return true;
    };
    auto next_tile_stage = [&]() __attribute__((always_inline)) -> bool {
        for (;;) {
            if (!next_tile()) return false;
            if (next_in_tile()) return true;
            epilogue(i_job, i_mt, i_n0, true);          // a tile no tap reaches: zeros
        }
    };
    {
        BFrag bc, bn;
        bool have = next_in_tile();
        if (!have) {
            epilogue(i_job, i_mt, i_n0, true);
            have = next_tile_stage();
            c_job = i_job; c_mt = i_mt; c_n0 = i_n0;
        }
        if (have) {
            bload(bc, wtap + (size_t)(2 * cc) * (64 * NP));
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
            const uint4* wn_ = more ? wtap + (size_t)(2 * cc) * (64 * NP) : a.w[0] + lane;
            compute(st, bc, more, bn, wn_);
            if (last) {
                epilogue(c_job, c_mt, c_n0, false);
                c_job = i_job; c_mt = i_mt; c_n0 = i_n0;
            }
            st ^= 1;
            bc = bn;
            have = more;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
}

// ---------------------------------------------------------------------------------------------------------------------
// the plan builder (host)
// ---------------------------------------------------------------------------------------------------------------------
struct AsppGeom {
    int nbranch;
    int ksize[ASPP_MAXB], dil[ASPP_MAXB];        // square filters, stride 1, pad = dil * (k - 1) / 2
};

static bool aspp_geom_ok(const iswm_conv_desc* d, int nbranch, const int* ksize, const int* dil, int kind) {
    if (!d || nbranch < 1 || nbranch > ASPP_MAXB || !ksize || !dil) return false;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->H > 1023 || d->W > 1023 || d->N > 2047) return false;
    if (d->Ho != d->H || d->Wo != d->W || d->stride != 1) return false;
    const int gc = kind ? d->Cout : d->Cin, nc = kind ? d->Cin : d->Cout;
    if (gc % 64 != 0 || nc % 4 != 0 || nc <= 0) return false;
    int taps = 0;
    for (int b = 0; b < nbranch; ++b) {
        if (ksize[b] < 1 || (ksize[b] & 1) == 0 || dil[b] < 1) return false;
        taps += ksize[b] * ksize[b];
    }
    return taps <= ASPP_MAXT && (int64_t)d->N * d->H * d->W < (1ll << 30);
}

static size_t aspp_plan_size(const iswm_conv_desc* d, int nbranch, int kind) {
    const int64_t M = (int64_t)d->N * d->H * d->W;
    const int64_t MT = (M + 16 * ASPP_RBW - 1) / (16 * ASPP_RBW);
    const int nc = kind ? d->Cin : d->Cout;
    const int64_t NT = (nc + 127) / 128;
    const int njobs = kind ? 1 : nbranch;
    size_t bytes = (sizeof(AsppPlan) + 15) / 16 * 16;
    bytes += (size_t)njobs * M * sizeof(int);
    bytes = (bytes + 15) / 16 * 16;
    bytes += (size_t)njobs * MT * NT * sizeof(int4);
    return bytes;
}

// rows of one job: pixels sorted by their set of in-bounds taps (heaviest set first); mask[m] = that set
static void aspp_sort_rows(int N, int H, int W, const AsppTap* taps, int ntaps, int* rowmap, std::vector<unsigned>& rowmask) {
    const int64_t M = (int64_t)N * H * W;
    std::vector<unsigned> hm(H), wm(W);        // per image row / column: taps in bounds along that axis
    for (int h = 0; h < H; ++h) {
        unsigned m = 0;
        for (int t = 0; t < ntaps; ++t)
            if ((unsigned)(h + taps[t].dh) < (unsigned)H) m |= 1u << t;
        hm[h] = m;
    }
    for (int w = 0; w < W; ++w) {
        unsigned m = 0;
        for (int t = 0; t < ntaps; ++t)
            if ((unsigned)(w + taps[t].dw) < (unsigned)W) m |= 1u << t;
        wm[w] = m;
    }
    struct Key { unsigned mask; int code; };
    std::vector<Key> keys((size_t)M);
    size_t i = 0;
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w) keys[i++] = Key{hm[h] & wm[w], (n << 20) | (h << 10) | w};
    std::stable_sort(keys.begin(), keys.end(), [](const Key& x, const Key& y) {
        const int px = __builtin_popcount(x.mask), py = __builtin_popcount(y.mask);
        if (px != py) return px > py;
        return x.mask < y.mask;                // equal sets stay together; inside a set: batch-major row-major (stable)
    });
    rowmask.resize((size_t)M);
    for (size_t k = 0; k < (size_t)M; ++k) {
        rowmap[k] = keys[k].code;
        rowmask[k] = keys[k].mask;
    }
}

static int aspp_build_plan(const iswm_conv_desc* d, int nbranch, const int* ksize, const int* dil, int kind, void* out, int ncu) {
    const int64_t M64 = (int64_t)d->N * d->H * d->W;
    const int M = (int)M64, BM = 16 * ASPP_RBW;
    const int MT = (M + BM - 1) / BM;
    const int nc = kind ? d->Cin : d->Cout, gc = kind ? d->Cout : d->Cin;
    const int NT = (nc + 127) / 128;
    const int njobs = kind ? 1 : nbranch;
    const size_t total = aspp_plan_size(d, nbranch, kind);
    memset(out, 0, total);
    AsppPlan* P = reinterpret_cast<AsppPlan*>(out);
    P->magic = ASPP_MAGIC; P->kind = kind; P->nbranch = nbranch; P->njobs = njobs;
    P->N = d->N; P->H = d->H; P->W = d->W; P->M = M; P->MT = MT; P->NT = NT; P->GC = gc; P->NC = nc;
    size_t off = (sizeof(AsppPlan) + 15) / 16 * 16;
    P->rowmap_off = (int)off;
    off += (size_t)njobs * M * sizeof(int);
    off = (off + 15) / 16 * 16;
    P->tiles_off = (int)off;
    P->total_bytes = (int)total;
    // taps: forward gathers x at (oh + kh*dil - pad, ...); the data gradient gathers dy at (ih + pad - kh*dil, ...)
    int nt = 0;
    for (int b = 0; b < nbranch; ++b) {
        const int k = ksize[b], pad = dil[b] * (k - 1) / 2;
        if (!kind) {
            P->jobs[b].tap_begin = nt; P->jobs[b].ntaps = k * k; P->jobs[b].branch = b; P->jobs[b].rowmap_off = b * M;
        }
        for (int kh = 0; kh < k; ++kh)
            for (int kw = 0; kw < k; ++kw) {
                AsppTap& t = P->taps[nt++];
                t.dh = kind ? pad - kh * dil[b] : kh * dil[b] - pad;
                t.dw = kind ? pad - kw * dil[b] : kw * dil[b] - pad;
                t.branch = b;
                t.k32base = (kh * k + kw) * (gc >> 5);
            }
    }
    P->ntaps = nt;
    if (kind) {
        P->jobs[0].tap_begin = 0; P->jobs[0].ntaps = nt; P->jobs[0].branch = -1; P->jobs[0].rowmap_off = 0;
    }
    int* rowmap = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(out) + P->rowmap_off);
    int4* tiles = reinterpret_cast<int4*>(reinterpret_cast<unsigned char*>(out) + P->tiles_off);
    struct T { int job, mt, nt; unsigned mask; int weight; };
    std::vector<T> all;
    std::vector<unsigned> rowmask;
    for (int j = 0; j < njobs; ++j) {
        const AsppJob& jb = P->jobs[j];
        aspp_sort_rows(d->N, d->H, d->W, P->taps + jb.tap_begin, jb.ntaps, rowmap + jb.rowmap_off, rowmask);
        for (int mt = 0; mt < MT; ++mt) {
            unsigned mask = 0;
            for (int m = mt * BM; m < std::min(M, (mt + 1) * BM); ++m) mask |= rowmask[(size_t)m];
            for (int n = 0; n < NT; ++n) all.push_back(T{j, mt, n, mask, __builtin_popcount(mask)});
        }
    }
    // heaviest first (a tile's time is its tap count), then by position so that neighbours in the table share operands
    std::stable_sort(all.begin(), all.end(), [](const T& x, const T& y) { return x.weight > y.weight; });
    // the persistent grid hands out entries round-robin: reverse every other band of `grid` entries (serpentine) so that
    // the workgroup that got the heaviest tile of one band gets the lightest of the next
    const int grid = std::max(1, std::min((int)all.size(), ncu));
    for (size_t b0 = grid; b0 < all.size(); b0 += 2 * (size_t)grid)
        std::reverse(all.begin() + b0, all.begin() + std::min(all.size(), b0 + grid));
    P->ntiles = (int)all.size();
    for (size_t k = 0; k < all.size(); ++k) tiles[k] = make_int4(all[k].job, all[k].mt, all[k].nt, (int)all[k].mask);
    return 0;
}

static int device_cus() {
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    return ncu;
}

}  // namespace iswm

using namespace iswm;

extern "C" size_t iswm_aspp_plan_bytes(const iswm_conv_desc* d, int nbranch, const int* ksize, const int* dil, int kind) {
    if ((kind != 0 && kind != 1) || !aspp_geom_ok(d, nbranch, ksize, dil, kind) || iswm_get_conv_math() != 1) return 0;
    return aspp_plan_size(d, nbranch, kind);
}

/* fills `host_plan` (iswm_aspp_plan_bytes bytes of HOST memory); the caller copies it to the device once per geometry.
 * grid_hint: workgroups the launch will use (0 = the current device's compute units, 256 when there is none) */
extern "C" int iswm_aspp_plan(const iswm_conv_desc* d, int nbranch, const int* ksize, const int* dil, int kind, void* host_plan,
                              int grid_hint) {
    ISWM_REQUIRE(kind == 0 || kind == 1, "aspp_plan: kind must be 0 (forward) or 1 (data gradient)");
    ISWM_REQUIRE(aspp_geom_ok(d, nbranch, ksize, dil, kind),
                 "aspp_plan: needs 1..4 odd square stride-1 same-size branches, <= 32 taps, gathered channels %% 64 == 0");
    ISWM_REQUIRE(host_plan != nullptr, "aspp_plan: null buffer");
    return aspp_build_plan(d, nbranch, ksize, dil, kind, host_plan, grid_hint > 0 ? grid_hint : 256);
}

static int aspp_check_plan_args(const iswm_conv_desc* d, int nbranch, const void* plan_dev, const char* what) {
    ISWM_REQUIRE(d && plan_dev && aligned16(plan_dev), "%s: null / unaligned plan", what);
    ISWM_REQUIRE(nbranch >= 1 && nbranch <= ASPP_MAXB, "%s: 1..4 branches", what);
    ISWM_REQUIRE(iswm_get_conv_math() == 1, "%s: bf16x6 conv math only", what);
    return 0;
}

/* y_b = conv(x, w_b) for every branch b in ONE launch; x: planes [N*H*W][ldx]; wpk[b]: branch b's weights packed by
 * iswm_conv2d_pl2_pack_weights(kind 0) for its own descriptor (Cin -> Cout, k_b x k_b); y[b]: fp32 [N*H*W][ldy] (d->ldy);
 * stats[b]: [2][tiles][Cout] BatchNorm partials with tiles = ceil(N*H*W / 144) rows of 144 (null: none).  The arrays are HOST arrays
 * of device pointers. */
extern "C" int iswm_aspp_fwd(const iswm_conv_desc* d, int nbranch, const int* ksize, const int* dil, const void* plan_dev,
                             const void* xp, int64_t x_ps, const void* const* wpk, float* const* y, float* const* stats,
                             iswm_stream_t stream) {
    if (int e = aspp_check_plan_args(d, nbranch, plan_dev, "aspp_fwd")) return e;
    ISWM_REQUIRE(aspp_geom_ok(d, nbranch, ksize, dil, 0) && d->Cout % 4 == 0, "aspp_fwd: unsupported geometry");
    ISWM_REQUIRE(xp && wpk && y && aligned16(xp) && d->ldx % 8 == 0 && x_ps % 8 == 0 && x_ps > 0 && d->ldy % 4 == 0 && d->ldy >= d->Cout,
                 "aspp_fwd: bad operand (ldx %% 8, plane stride %% 8, ldy %% 4)");
    AsppArgs a{};
    a.plan = reinterpret_cast<const AsppPlan*>(plan_dev);
    a.x = reinterpret_cast<const unsigned short*>(xp);
    a.xps = x_ps * 2;
    a.ldx = d->ldx;
    a.ldy = d->ldy;
    for (int b = 0; b < nbranch; ++b) {
        ISWM_REQUIRE(wpk[b] && y[b] && aligned16(wpk[b]) && aligned16(y[b]), "aspp_fwd: null / unaligned branch pointer");
        a.w[b] = reinterpret_cast<const uint4*>(wpk[b]);
        a.K32[b] = ksize[b] * ksize[b] * (d->Cin >> 5);
        a.y[b] = y[b];
        a.stats[b] = stats ? stats[b] : nullptr;
    }
    const int64_t M = (int64_t)d->N * d->H * d->W;
    const int64_t tiles = ((M + 16 * ASPP_RBW - 1) / (16 * ASPP_RBW)) * ((d->Cout + 127) / 128) * nbranch;
    const int ncu = device_cus();
    hipLaunchKernelGGL(k_conv_pl2t<false>, dim3((unsigned)std::min<int64_t>(tiles, ncu)), dim3(512), 0, (hipStream_t)stream, a);
    return check_launch("aspp_fwd");
}

/* dx (=|+=) sum_b conv^T(dy_b, w_b) in ONE launch.  dyp: planes of the concatenated output gradient [N*H*W][ld_dy] with branch b
 * at channels [b * Cout, (b + 1) * Cout); wpk[b]: branch b's weights packed by iswm_conv2d_pl2_pack_weights(kind 1); dx: fp32
 * [N*H*W][d->ldx].  When x planes, dw pointers and a workspace are given the four weight gradients run too
 * (iswm_conv2d_wgrad_planes per branch: tap rectangles for the atrous ones) -- the whole backward of the branch convolutions. */
extern "C" int iswm_aspp_bwd(const iswm_conv_desc* d, int nbranch, const int* ksize, const int* dil, const void* plan_dev,
                             const void* dyp, int64_t dy_ps, int ld_dy, const void* const* wpk, float* dx, int accumulate,
                             const void* xp, int64_t x_ps, float* const* dw, float* workspace, size_t workspace_bytes,
                             iswm_stream_t stream) {
    if (int e = aspp_check_plan_args(d, nbranch, plan_dev, "aspp_bwd")) return e;
    ISWM_REQUIRE(aspp_geom_ok(d, nbranch, ksize, dil, 1) && d->Cin % 4 == 0, "aspp_bwd: unsupported geometry");
    ISWM_REQUIRE(dyp && wpk && dx && aligned16(dyp) && aligned16(dx) && ld_dy % 8 == 0 && ld_dy >= nbranch * d->Cout && dy_ps % 8 == 0 &&
                     dy_ps > 0 && d->ldx % 4 == 0 && d->ldx >= d->Cin,
                 "aspp_bwd: bad operand (ld_dy %% 8 and >= nbranch * Cout, plane stride %% 8, ldx %% 4)");
    AsppArgs a{};
    a.plan = reinterpret_cast<const AsppPlan*>(plan_dev);
    a.x = reinterpret_cast<const unsigned short*>(dyp);
    a.xps = dy_ps * 2;
    a.ldx = ld_dy;
    a.ldy = d->ldx;
    a.y[0] = dx;
    a.accumulate = accumulate;
    for (int b = 0; b < nbranch; ++b) {
        ISWM_REQUIRE(wpk[b] && aligned16(wpk[b]), "aspp_bwd: null / unaligned weights");
        a.w[b] = reinterpret_cast<const uint4*>(wpk[b]);
        a.K32[b] = ksize[b] * ksize[b] * (d->Cout >> 5);
        a.goff[b] = b * d->Cout;
    }
    const int64_t M = (int64_t)d->N * d->H * d->W;
    const int64_t tiles = ((M + 16 * ASPP_RBW - 1) / (16 * ASPP_RBW)) * ((d->Cin + 127) / 128);
    const int ncu = device_cus();
    hipLaunchKernelGGL(k_conv_pl2t<true>, dim3((unsigned)std::min<int64_t>(tiles, ncu)), dim3(512), 0, (hipStream_t)stream, a);
    if (int e = check_launch("aspp_bwd(data)")) return e;
    if (!dw) return 0;
    ISWM_REQUIRE(xp && aligned16(xp), "aspp_bwd: the weight gradients need the input planes");
    for (int b = 0; b < nbranch; ++b) {
        if (!dw[b]) continue;
        iswm_conv_desc db = *d;
        db.KH = db.KW = ksize[b];
        db.dil = dil[b];
        db.pad = dil[b] * (ksize[b] - 1) / 2;
        db.ldy = ld_dy;
        const unsigned short* dyb = reinterpret_cast<const unsigned short*>(dyp) + (size_t)b * d->Cout;
        if (int e = iswm_conv2d_wgrad_planes(&db, xp, x_ps, dyb, dy_ps, dw[b], workspace, workspace_bytes, stream)) return e;
    }
    return 0;
}
