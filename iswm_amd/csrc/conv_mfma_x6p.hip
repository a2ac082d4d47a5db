// bf16x6 implicit-GEMM convolution for stride-1 KxK filters, "halo patch" form.
//
// The tap-uniform kernels (conv_mfma_x6.hip) re-gather and re-split the activation operand once per filter
// tap: a 3x3 conv splits every input value 9 times per column tile and pays two barriers per 32-channel
// chunk of every tap.  Here a workgroup owns a PH x PW patch of output pixels (PH*PW <= 128 GEMM rows) of one
// image and, per 32-channel chunk, stages the (PH + dil*(KH-1)) x (PW + dil*(KW-1)) input halo ONCE: gather
// -> exact 3-way bf16 split -> three LDS planes [halo pixel][32 k] (80-byte rows).  All KH*KW taps are then
// multiplied out of that one staging -- the A fragment of tap (kh, kw) for patch pixel (py, px) is simply the
// LDS row (py + kh*dil)*HW + px + kw*dil -- so split arithmetic, LDS writes, global gathers and barriers all
// drop by ~KH*KW / (halo / patch) (5-6x for 3x3), leaving a loop of LDS fragment reads and MFMAs.
// The weight operand comes pre-split in fragment order (k_pack_weights_x6) straight from L2 into registers,
// double buffered across taps.
//
// Tile: 128 rows x 64 columns, 256 threads = 2 x 2 waves of 64 x 32.  LDS 3 x 208 x 80 B = 49.9 KB -> 3 per CU.
// Forward and data gradient share the kernel: the data gradient of a stride-1 conv is a conv of dy with the
// transposed weights (packed kind 1) and mirrored taps.
#include "conv_common.h"

namespace iswm {

static __device__ __attribute__((aligned(16))) float g_zero_row_p[32];

constexpr int XP_PITCH = 80;        // bytes per LDS row of one plane
constexpr int XP_HPMAX = 208;       // halo pixels per patch (LDS rows)
constexpr int XP_PASSES = (XP_HPMAX + 31) / 32;
constexpr int XP_PLANE = XP_HPMAX * XP_PITCH;


// NP: bf16 planes per operand -- 3 = exact split, six MFMAs per product (bf16x6); 1 = operands rounded to bf16, one MFMA
template <bool DGRAD, int NP>
__global__ __launch_bounds__(256, 3) void k_conv_x6_patch(const PatchArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char As[NP * XP_PLANE];
    __shared__ int rowpix[128];          // output pixel index of each tile row, -1 = no such pixel
    __shared__ float red[4 * 64];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = L / a.NT, nt = L - mt * a.NT;
    const int n0 = nt * 64;
    const int tpi = a.TPY * a.TPX;
    const int img = mt / tpi, pidx = mt - img * tpi;
    const int pyi = pidx / a.TPX, pxi = pidx - pyi * a.TPX;
    const int oy = pyi * a.PH, ox = pxi * a.PW;
    const int HP = a.HH * a.HW, PP = a.PH * a.PW;
    const int q = t & 7, r0 = t >> 3;

    // tile row -> output pixel (for the epilogue) and validity count
    int myvalid = 0;
    if (t < 128) {
        int pix = -1;
        if (t < PP) {
            int py = t / a.PW, px = t - py * a.PW;
            if (oy + py < a.RH && ox + px < a.RW) pix = (img * a.RH + oy + py) * a.RW + ox + px;
        }
        rowpix[t] = pix;
        myvalid = pix >= 0;
    }
    const int cnt = __syncthreads_count(myvalid);

    // halo staging: thread (q, r0) owns halo pixels r0 + 32 j, channels 4q..4q+3 of the current 32-channel chunk
    const float* aptr[XP_PASSES];
    int astep[XP_PASSES];
#pragma unroll
    for (int j = 0; j < XP_PASSES; ++j) {
        const int hp = r0 + 32 * j;
        bool ok = hp < HP;
        const int hy = hp / a.HW, hx = hp - hy * a.HW;
        const int gy = oy + a.orgh + hy, gx = ox + a.orgw + hx;
        ok = ok && (unsigned)gy < (unsigned)a.GH && (unsigned)gx < (unsigned)a.GW;
        aptr[j] = ok ? a.x + ((size_t)(img * a.GH + gy) * a.GW + gx) * a.ldg + q * 4 : g_zero_row_p + q * 4;
        astep[j] = ok ? 32 : 0;
    }
    const int npass = (HP + 31) >> 5;
    float4 ra[XP_PASSES];
    auto gload = [&]() {
#pragma unroll
        for (int j = 0; j < XP_PASSES; ++j)
            if (j < npass) {
                ra[j] = ldg4(aptr[j]);
                aptr[j] += astep[j];
            }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < XP_PASSES; ++j)
            if (j < npass && r0 + 32 * j < XP_HPMAX) {
                unsigned char* p = As + (r0 + 32 * j) * XP_PITCH + q * 8;
                if constexpr (NP == 1) {
                    *reinterpret_cast<uint2*>(p) = round_bf16x4(ra[j]);
                } else {
                    uint2 h, m, l;
                    split3(ra[j], h, m, l);
                    *reinterpret_cast<uint2*>(p) = h;
                    *reinterpret_cast<uint2*>(p + XP_PLANE) = m;
                    *reinterpret_cast<uint2*>(p + 2 * XP_PLANE) = l;
                }
            }
    };

    // A fragment rows of this lane: tile row wm*64 + mb*32 + li -> halo row of tap (0,0)
    int hb[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int r = wm * 64 + mb * 32 + li;
        const int py = r / a.PW, px = r - py * a.PW;
        hb[mb] = r < PP ? (py * a.HW + px) * XP_PITCH + lh * 16 : lh * 16;
    }

    const int taps = a.KH * a.KW;
    const int nCC = a.GC >> 5;
    // packed weights: 192 uint4 per (column block of 32, k16); this wave's column block = (n0 + wn*32) / 32
    const uint4* wpk = a.wpk + (size_t)((n0 >> 5) + wn) * ((size_t)taps * a.GC >> 4) * (64 * NP) + lane;

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    struct BFrag {
        uint4 v[2][NP];  // [k half][plane]
    };
    auto bload = [&](BFrag& b, int tap, int cc) {
        const uint4* p = wpk + (size_t)((tap * nCC + cc) * 2) * (64 * NP);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) b.v[ks][pl] = p[(ks * NP + pl) * 64];
    };
    auto tap_off = [&](int tap) -> int {
        int kh = tap / a.KW, kw = tap - kh * a.KW;
        if (a.flip) {
            kh = a.KH - 1 - kh;
            kw = a.KW - 1 - kw;
        }
        return (kh * a.dil * a.HW + kw * a.dil) * XP_PITCH;
    };
    auto compute = [&](const BFrag& b, int tap) {
        const int off = tap_off(tap);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 ah[2], am[2], al[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const unsigned char* p = As + hb[mb] + off + ks * 32;
                ah[mb] = *reinterpret_cast<const uint4*>(p);
                if constexpr (NP == 3) {
                    am[mb] = *reinterpret_cast<const uint4*>(p + XP_PLANE);
                    al[mb] = *reinterpret_cast<const uint4*>(p + 2 * XP_PLANE);
                }
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                f32x16 c = acc[mb];
                if constexpr (NP == 3) {
                    c = mfma_bf16(al[mb], b.v[ks][0], c);     // smallest terms first
                    c = mfma_bf16(ah[mb], b.v[ks][2], c);
                    c = mfma_bf16(am[mb], b.v[ks][1], c);
                    c = mfma_bf16(am[mb], b.v[ks][0], c);
                    c = mfma_bf16(ah[mb], b.v[ks][1], c);
                }
                c = mfma_bf16(ah[mb], b.v[ks][0], c);
                acc[mb] = c;
            }
        }
    };

    gload();
    for (int cc = 0; cc < nCC; ++cc) {
        lstore();
        __syncthreads();
        if (cc + 1 < nCC) gload();
        BFrag b0, b1;
        bload(b0, 0, cc);
        for (int tap = 0; tap < taps; tap += 2) {
            if (tap + 1 < taps) bload(b1, tap + 1, cc);
            compute(b0, tap);
            if (tap + 2 < taps) bload(b0, tap + 2, cc);
            if (tap + 1 < taps) compute(b1, tap + 1);
        }
        __syncthreads();
    }

    // ---- epilogue
    const int col = n0 + wn * 32 + li;
    const bool cok = col < a.NC;
    const float bv = (!DGRAD && a.bias != nullptr && cok) ? a.bias[col] : 0.f;
    unsigned vmask = 0;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int pix = rowpix[row];
            if (pix >= 0) {
                vmask |= 1u << (mb * 16 + r);
                if (cok) {
                    float* o = &a.y[(size_t)pix * a.ldo + col];
                    *o = (DGRAD && a.accumulate) ? *o + acc[mb][r] : acc[mb][r] + bv;
                }
            }
        }
    if (!DGRAD && a.stats != nullptr) {
        float s = 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += (vmask >> (mb * 16 + r)) & 1u ? acc[mb][r] : 0.f;
        s += __shfl_xor(s, 32);
        if (lh == 0) red[wm * 64 + wn * 32 + li] = s;
        __syncthreads();
        const int c = wn * 32 + li;
        const float mean = (red[c] + red[64 + c]) / (float)cnt;
        float qv = 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float dv = acc[mb][r] - mean;
                qv += (vmask >> (mb * 16 + r)) & 1u ? dv * dv : 0.f;
            }
        qv += __shfl_xor(qv, 32);
        if (lh == 0) red[(2 + wm) * 64 + c] = qv;
        __syncthreads();
        if (t < 64 && n0 + t < a.NC) {
            a.stats[(size_t)mt * a.NC + n0 + t] = red[t] + red[64 + t];
            a.stats[(size_t)(a.MT + mt) * a.NC + n0 + t] = red[128 + t] + red[192 + t];
        }
        if (t == 0 && nt == 0) a.stats[(size_t)2 * a.MT * a.NC + mt] = (float)cnt;
    }
}

// Patch shape for an RH x RW pixel grid: PH*PW <= 128 rows, halo <= XP_HPMAX LDS rows; fewest tiles, then
// smallest halo.  Returns false when no shape reaches 80 % row utilisation (the caller keeps the tap-uniform
// kernel) or the filter is pointwise.
bool conv_patch_plan(int RH, int RW, int KH, int KW, int dil, int* PH, int* PW) {
    if (KH * KW <= 1) return false;
    const int eh = dil * (KH - 1), ew = dil * (KW - 1);
    long best_tiles = -1, best_halo = 0;
    for (int pw = 1; pw <= RW && pw <= 128; ++pw) {
        int ph = 128 / pw;
        if (ph > RH) ph = RH;
        while (ph >= 1 && (long)(ph + eh) * (pw + ew) > XP_HPMAX) --ph;
        if (ph < 1) continue;
        // the same tile count may be reachable with fewer rows per patch: shrink ph while it stays equal
        const long ty = (RH + ph - 1) / ph, tx = (RW + pw - 1) / pw;
        const int ph2 = (int)((RH + ty - 1) / ty);
        const long tiles = ty * tx, halo = (long)(ph2 + eh) * (pw + ew);
        if (best_tiles < 0 || tiles < best_tiles || (tiles == best_tiles && halo < best_halo)) {
            best_tiles = tiles;
            best_halo = halo;
            *PH = ph2;
            *PW = pw;
        }
    }
    if (best_tiles < 0) return false;
    return (double)RH * RW >= 0.80 * (double)best_tiles * 128.0;
}

// a: fields x, wpk, bias, y, stats, N, RH, RW, GH, GW, GC, NC, KH, KW, dil, orgh, orgw, flip, ldg, ldo, accumulate
// set by the caller; PH/PW from conv_patch_plan.
void launch_conv_x6_patch(PatchArgs a, bool dgrad, int planes, hipStream_t s) {
    a.HH = a.PH + a.dil * (a.KH - 1);
    a.HW = a.PW + a.dil * (a.KW - 1);
    a.TPY = (a.RH + a.PH - 1) / a.PH;
    a.TPX = (a.RW + a.PW - 1) / a.PW;
    a.MT = a.N * a.TPY * a.TPX;
    a.NT = (a.NC + 63) / 64;
    dim3 grid(a.MT * a.NT), blk(256);
    if (planes == 1) {
        if (dgrad) hipLaunchKernelGGL((k_conv_x6_patch<true, 1>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((k_conv_x6_patch<false, 1>), grid, blk, 0, s, a);
    } else {
        if (dgrad) hipLaunchKernelGGL((k_conv_x6_patch<true, 3>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((k_conv_x6_patch<false, 3>), grid, blk, 0, s, a);
    }
}

}  // namespace iswm
