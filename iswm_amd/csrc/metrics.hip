// Validation metrics on the device: the n_classes x n_classes confusion matrix of (ground truth, prediction)
// pairs -- StreamMetrics._fast_hist of the reference (metrics/stream_metrics.py:24-31) -- either from an
// argmax mask or fused with the argmax over NCHW logits (train.py:644,659), so no mask leaves the GPU.
// Integer counting only: per-workgroup LDS histogram, then one 64-bit atomic add per non-zero bin -- integer
// addition is associative, so the result is exact and order independent (bit-equal to numpy.bincount).
#include "common.h"

namespace iswm {

constexpr int CM_MAX_CLASSES = 32;   // LDS histogram of up to 32 x 32 bins

template <typename LT, typename PT, bool FUSED>
__global__ __launch_bounds__(256) void k_confusion(const LT* __restrict__ labels, const PT* __restrict__ preds,
                                                   const float* __restrict__ logits, int C, int64_t HW, int64_t npix,
                                                   int nc, unsigned long long* __restrict__ hist) {
    __shared__ unsigned int sh[CM_MAX_CLASSES * CM_MAX_CLASSES];
    const int bins = nc * nc;
    for (int i = threadIdx.x; i < bins; i += 256) sh[i] = 0;
    __syncthreads();
    // a workgroup counts at most 2^32-1 pixels: the grid is sized so that it never does
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
        const long long t = (long long)labels[i];
        long long p;
        if (FUSED) {
            const int64_t b = i / HW, s = i - b * HW;
            const float* z = logits + (size_t)b * C * HW + s;
            float best = z[0];
            int bi = 0;
            for (int c = 1; c < C; ++c) {
                float v = z[(size_t)c * HW];
                if (v > best) {          // ties keep the lowest index, as torch.max does
                    best = v;
                    bi = c;
                }
            }
            p = bi;
        } else {
            p = (long long)preds[i];
        }
        // mask = (label_true >= 0) & (label_true < n_classes); a prediction outside [0, n_classes) would index
        // past the matrix in numpy as well -- it is dropped here
        if (t >= 0 && t < nc && p >= 0 && p < nc) atomicAdd(&sh[(int)t * nc + (int)p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += 256)
        if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}

}  // namespace iswm

using namespace iswm;

static int cm_grid(int64_t npix) {
    int64_t g = (npix + 256 * 16 - 1) / (256 * 16);
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}

/* hist[n_classes][n_classes] (int64) += bincount(n_classes * label + pred) over pixels whose label is in
 * [0, n_classes).  label_dtype / pred_dtype: 0 = uint8, 1 = int64. */
extern "C" int iswm_confusion_matrix(const void* labels, int label_dtype, const void* preds, int pred_dtype, int64_t npix,
                                     int n_classes, int64_t* hist, iswm_stream_t stream) {
    ISWM_REQUIRE(labels && preds && hist && npix > 0, "confusion_matrix: bad argument");
    ISWM_REQUIRE(n_classes >= 1 && n_classes <= CM_MAX_CLASSES, "confusion_matrix: n_classes must be in [1, %d]",
                 CM_MAX_CLASSES);
    ISWM_REQUIRE((label_dtype == 0 || label_dtype == 1) && (pred_dtype == 0 || pred_dtype == 1),
                 "confusion_matrix: dtype codes are 0 (uint8) and 1 (int64)");
    dim3 grid(cm_grid(npix)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* h = reinterpret_cast<unsigned long long*>(hist);
#define CM_LAUNCH(LT, PT)                                                                                        \
    hipLaunchKernelGGL((k_confusion<LT, PT, false>), grid, blk, 0, s, (const LT*)labels, (const PT*)preds, nullptr, 0, \
                       (int64_t)0, npix, n_classes, h)
    if (label_dtype == 0 && pred_dtype == 0) CM_LAUNCH(unsigned char, unsigned char);
    else if (label_dtype == 0) CM_LAUNCH(unsigned char, long long);
    else if (pred_dtype == 0) CM_LAUNCH(long long, unsigned char);
    else CM_LAUNCH(long long, long long);
#undef CM_LAUNCH
    return check_launch("confusion_matrix");
}

/* the same with the prediction taken as argmax over the C channels of NCHW logits (ties -> lowest index) */
extern "C" int iswm_confusion_matrix_logits(const void* labels, int label_dtype, const float* logits, int B, int C,
                                            int64_t HW, int n_classes, int64_t* hist, iswm_stream_t stream) {
    ISWM_REQUIRE(labels && logits && hist && B > 0 && C > 0 && HW > 0, "confusion_matrix_logits: bad argument");
    ISWM_REQUIRE(n_classes >= 1 && n_classes <= CM_MAX_CLASSES, "confusion_matrix_logits: n_classes must be in [1, %d]",
                 CM_MAX_CLASSES);
    ISWM_REQUIRE(label_dtype == 0 || label_dtype == 1, "confusion_matrix_logits: dtype codes are 0 (uint8) and 1 (int64)");
    const int64_t npix = (int64_t)B * HW;
    dim3 grid(cm_grid(npix)), blk(256);
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* h = reinterpret_cast<unsigned long long*>(hist);
    if (label_dtype == 0)
        hipLaunchKernelGGL((k_confusion<unsigned char, unsigned char, true>), grid, blk, 0, s, (const unsigned char*)labels,
                           (const unsigned char*)nullptr, logits, C, HW, npix, n_classes, h);
    else
        hipLaunchKernelGGL((k_confusion<long long, unsigned char, true>), grid, blk, 0, s, (const long long*)labels,
                           (const unsigned char*)nullptr, logits, C, HW, npix, n_classes, h);
    return check_launch("confusion_matrix_logits");
}
