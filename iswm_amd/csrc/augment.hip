// Training-input pipeline on the device: the reference's per-sample PIL chain
//   ExtRandomScale (bilinear / nearest resize) -> ExtRandomCrop(pad_if_needed) -> ExtRandomHorizontalFlip
//   -> ExtToTensor -> ExtNormalize                        (train.py:355-362, utils/ext_transforms.py:94-115,212-396)
// as ONE kernel over the whole batch, reading the uint8 source tiles and writing the normalised fp32 NCHW batch
// plus the uint8 label batch.  Only the crop window is ever computed.
//
// Bit-exactness with Pillow (the arithmetic the reference runs on, Resample.c / Geometry.c):
//   * image: Pillow's 8-bit antialiased BILINEAR resample is a horizontal pass into a uint8 temporary followed by a
//     vertical pass, each a fixed-point (22 fractional bits) weighted sum rounded and clipped to uint8.  The
//     per-output-column / per-output-row bounds and integer weights are computed on the host in double exactly as
//     precompute_coeffs / normalize_coeffs_8bpc do and passed in a table; the kernel evaluates the same two integer
//     sums (the horizontal values of the rows a pixel needs are recomputed on the fly -- at most 5 x 5 taps);
//   * label: Pillow's NEAREST resize walks source coordinates by repeated double addition; the host performs the
//     same additions and passes the integer source index per output column / row;
//   * ToTensor / Normalize: v / 255.0f, then (x - mean) / std in IEEE fp32, as torch does.
#include "common.h"

namespace iswm {

constexpr int AUG_PRECISION_BITS = 22;   // Pillow: 32 - 8 - 2

struct AugSample {           // mirrors iswm_aug_sample
    long long img_off;       // byte offset of the uint8 HWC (3 channel) source image
    long long lbl_off;       // byte offset of the uint8 HW source label
    int src_h, src_w;
    int rs_h, rs_w;          // size after ExtRandomScale
    int pad;                 // border added on every side by pad_if_needed (0 if none)
    int crop_i, crop_j;      // crop origin in the padded image
    int flip;
    int tab_off;             // offset (ints) of this sample's tables in `tables`
    int ksize_h, ksize_v;    // taps per output column / row
    int reserved;
};

__device__ __forceinline__ int clip8(int v) {
    v >>= AUG_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// tables of one sample (ints): xintab[rs_w] | yintab[rs_h] | hbounds[rs_w][2] | hk[rs_w][ksize_h] |
//                              vbounds[rs_h][2] | vk[rs_h][ksize_v]
__global__ __launch_bounds__(256) void k_augment(const unsigned char* __restrict__ images,
                                                 const unsigned char* __restrict__ labels,
                                                 const AugSample* __restrict__ samples, const int* __restrict__ tables,
                                                 int B, int TH, int TW, float m0, float m1, float m2, float s0, float s1,
                                                 float s2, float* __restrict__ out, unsigned char* __restrict__ out_lbl) {
    const long long npix = (long long)B * TH * TW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / ((long long)TH * TW));
        const int rem = (int)(i - (long long)b * TH * TW);
        const int y = rem / TW, x = rem - y * TW;
        const AugSample s = samples[b];
        const int xs = s.flip ? TW - 1 - x : x;
        const int ry = s.crop_i + y - s.pad, rx = s.crop_j + xs - s.pad;     // coordinates in the resized image
        int v0 = 0, v1 = 0, v2 = 0, lab = 0;                                  // padding: image 0, label 0
        if (ry >= 0 && ry < s.rs_h && rx >= 0 && rx < s.rs_w) {
            const int* tab = tables + s.tab_off;
            const int* xintab = tab;
            const int* yintab = xintab + s.rs_w;
            const int* hb = yintab + s.rs_h;
            const int* hk = hb + 2 * s.rs_w;
            const int* vb = hk + (size_t)s.rs_w * s.ksize_h;
            const int* vk = vb + 2 * s.rs_h;
            lab = labels[s.lbl_off + (long long)yintab[ry] * s.src_w + xintab[rx]];
            const int xmin = hb[2 * rx], xcnt = hb[2 * rx + 1];
            const int ymin = vb[2 * ry], ycnt = vb[2 * ry + 1];
            const int* kx = hk + (size_t)rx * s.ksize_h;
            const int* ky = vk + (size_t)ry * s.ksize_v;
            int a0 = 1 << (AUG_PRECISION_BITS - 1), a1 = a0, a2 = a0;
            for (int r = 0; r < ycnt; ++r) {
                const unsigned char* row = images + s.img_off + ((long long)(ymin + r) * s.src_w + xmin) * 3;
                int h0 = 1 << (AUG_PRECISION_BITS - 1), h1 = h0, h2 = h0;
                for (int c = 0; c < xcnt; ++c) {
                    const int k = kx[c];
                    h0 += row[3 * c] * k;
                    h1 += row[3 * c + 1] * k;
                    h2 += row[3 * c + 2] * k;
                }
                const int k = ky[r];
                a0 += clip8(h0) * k;
                a1 += clip8(h1) * k;
                a2 += clip8(h2) * k;
            }
            v0 = clip8(a0);
            v1 = clip8(a1);
            v2 = clip8(a2);
        }
        const size_t plane = (size_t)TH * TW;
        float* o = out + (size_t)b * 3 * plane + (size_t)y * TW + x;
        o[0] = __fdiv_rn(__fdiv_rn((float)v0, 255.0f) - m0, s0);
        o[plane] = __fdiv_rn(__fdiv_rn((float)v1, 255.0f) - m1, s1);
        o[2 * plane] = __fdiv_rn(__fdiv_rn((float)v2, 255.0f) - m2, s2);
        out_lbl[i] = (unsigned char)lab;
    }
}

}  // namespace iswm

using namespace iswm;

extern "C" int iswm_augment_batch(const unsigned char* images, const unsigned char* labels, const void* samples,
                                  const int* tables, int B, int crop_h, int crop_w, const float* mean3,
                                  const float* std3, float* out_nchw, unsigned char* out_labels, iswm_stream_t stream) {
    ISWM_REQUIRE(images && labels && samples && tables && out_nchw && out_labels && mean3 && std3,
                 "augment_batch: null pointer");
    ISWM_REQUIRE(B > 0 && crop_h > 0 && crop_w > 0, "augment_batch: bad size");
    static_assert(sizeof(AugSample) == 64, "iswm_aug_sample layout");
    const long long npix = (long long)B * crop_h * crop_w;
    hipLaunchKernelGGL(k_augment, dim3(stream_grid(npix, 256)), dim3(256), 0, (hipStream_t)stream, images, labels,
                       (const AugSample*)samples, tables, B, crop_h, crop_w, mean3[0], mean3[1], mean3[2], std3[0],
                       std3[1], std3[2], out_nchw, out_labels);
    return check_launch("augment_batch");
}
