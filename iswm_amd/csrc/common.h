// Shared helpers for the gfx950 kernels of libiswm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/iswm_hip.h"

namespace iswm {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return 2;
    }
    return 0;
}

#define ISWM_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            iswm::set_error(__VA_ARGS__);  \
            return 1;                      \
        }                                  \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// memory-bound kernels: cap the grid and grid-stride the rest (256 CUs x 8 blocks)
inline int stream_grid(int64_t work_items, int block) {
    int64_t g = (work_items + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

// Blocks b and b+8 share an XCD (and its L2) under round-robin dispatch; give each
// XCD a contiguous run of logical tile ids so neighbouring tiles hit one L2.
// Bijective for any grid size.  Placement affects speed only, never results.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7;
    int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

}  // namespace iswm
