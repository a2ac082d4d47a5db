// Thread mapping shared by the HBM-bound per-channel kernels over pitched NHWC
// tensors viewed as [M rows][C channels]:  each thread owns ONE float4 channel group
// for its whole life (so per-channel coefficients are loaded once and no index
// division happens per element) and walks rows with a fixed stride.  A row of threads
// reads CQ consecutive float4 = one coalesced run.
#pragma once
#include "common.h"
#include "planes.h"

namespace iswm {

struct RowPlan {
    int C4;         // float4 groups per row
    int CQ;         // channel groups per block (<= 256)
    int RL;         // row lanes per block = 256 / CQ
    int colblocks;  // gridDim.y
    int rowblocks;  // gridDim.x
};

inline RowPlan plan_rows(int64_t M, int C, int fixed_rowblocks = 0) {
    RowPlan p;
    p.C4 = C / 4;
    p.CQ = p.C4 < 256 ? p.C4 : 256;
    p.RL = 256 / p.CQ;
    p.colblocks = (p.C4 + p.CQ - 1) / p.CQ;
    if (fixed_rowblocks > 0) {
        p.rowblocks = fixed_rowblocks;
    } else {
        int64_t rb = (M + (int64_t)p.RL * 4 - 1) / ((int64_t)p.RL * 4);
        int64_t cap = 8192 / p.colblocks;
        if (cap < 1) cap = 1;
        if (rb > cap) rb = cap;
        if (rb < 1) rb = 1;
        p.rowblocks = (int)rb;
    }
    return p;
}

struct RowThread {
    int c4;         // this thread's float4 channel group
    int rl;         // row lane inside the block
    bool active;
    int64_t row0;   // first row
    int64_t rstep;  // row stride
};

__device__ __forceinline__ RowThread row_thread(int C4, int CQ, int RL) {
    RowThread r;
    const int t = threadIdx.x;
    r.rl = t / CQ;
    const int cq = t - r.rl * CQ;
    r.c4 = blockIdx.y * CQ + cq;
    r.active = (r.rl < RL) && (r.c4 < C4);
    r.row0 = (int64_t)blockIdx.x * RL + r.rl;
    r.rstep = (int64_t)gridDim.x * RL;
    return r;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

}  // namespace iswm
