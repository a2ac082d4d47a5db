// "Planes": an fp32 activation stored as its exact 3-way bf16 truncation split (hi, mid, lo; hi + mid + lo == x
// bit-exactly), three bf16 tensors [plane][row][ld], `ps` elements apart -- the operand format of the bf16x6
// convolution kernels (conv_mfma_pl2.hip).  The memory-bound kernels read and write it through ld4x / st4x:
//   ps == 0 : plain fp32 tensor (pitch ld floats)
//   ps  > 0 : three bf16 planes (pitch ld bf16 elements, plane stride ps elements)
//   ps == -1: ONE plane holding the value rounded to nearest bf16 (conv math "bf16", mixed precision)
#pragma once
#include "common.h"

namespace iswm {

// exact 3-way truncation split of 4 floats into packed bf16x4 planes
__device__ __forceinline__ void split3(const float4 v, uint2& hi, uint2& mid, uint2& lo) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned xb = __float_as_uint(x[i]);
        float r1 = x[i] - __uint_as_float(xb & 0xFFFF0000u);
        unsigned rb = __float_as_uint(r1);
        float r2 = r1 - __uint_as_float(rb & 0xFFFF0000u);
        h[i] = xb;
        m[i] = rb;
        l[i] = __float_as_uint(r2);
    }
    // pack the upper halves of two words: {src0 = odd element, src1 = even element}
    hi = make_uint2(__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u));
    mid = make_uint2(__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u));
    lo = make_uint2(__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u));
}


// round-to-nearest-even conversion of 4 floats to packed bf16x4 (conv math "bf16": one plane, one MFMA per product)
__device__ __forceinline__ uint2 round_bf16x4(const float4 v) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned u = __float_as_uint(x[i]);
        r[i] = u + 0x7FFFu + ((u >> 16) & 1u);        // finite inputs; NaN payloads are not preserved
    }
    return make_uint2(__builtin_amdgcn_perm(r[1], r[0], 0x07060302u), __builtin_amdgcn_perm(r[3], r[2], 0x07060302u));
}


__device__ __forceinline__ float4 bf16x4_to_f32(uint2 u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xFFFF0000u));
}

// element offset `off` = row * ld + channel (channel % 4 == 0)
__device__ __forceinline__ float4 ld4x(const void* base, int64_t off, int64_t ps) {
    if (ps == 0) return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
    const unsigned short* p = reinterpret_cast<const unsigned short*>(base) + off;
    float4 h = bf16x4_to_f32(*reinterpret_cast<const uint2*>(p));
    if (ps > 0) {
        const float4 m = bf16x4_to_f32(*reinterpret_cast<const uint2*>(p + ps));
        const float4 l = bf16x4_to_f32(*reinterpret_cast<const uint2*>(p + 2 * ps));
        h.x = (h.x + m.x) + l.x; h.y = (h.y + m.y) + l.y; h.z = (h.z + m.z) + l.z; h.w = (h.w + m.w) + l.w;   // exact
    }
    return h;
}

__device__ __forceinline__ void st4x(void* base, int64_t off, int64_t ps, float4 v) {
    if (ps == 0) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + off) = v;
        return;
    }
    unsigned short* p = reinterpret_cast<unsigned short*>(base) + off;
    if (ps > 0) {
        uint2 h, m, l;
        split3(v, h, m, l);
        *reinterpret_cast<uint2*>(p) = h;
        *reinterpret_cast<uint2*>(p + ps) = m;
        *reinterpret_cast<uint2*>(p + 2 * ps) = l;
    } else {
        *reinterpret_cast<uint2*>(p) = round_bf16x4(v);
    }
}

// 8 consecutive channels (channel % 8 == 0, 16-byte plane accesses: the memory-bound passes run ~1.3x faster than with 8-byte ones)
__device__ __forceinline__ void ld8x(const void* base, int64_t off, int64_t ps, float4& a, float4& b) {
    if (ps == 0) {
        const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
        a = p[0];
        b = p[1];
        return;
    }
    const unsigned short* p = reinterpret_cast<const unsigned short*>(base) + off;
    const uint4 h = *reinterpret_cast<const uint4*>(p);
    a = bf16x4_to_f32(make_uint2(h.x, h.y));
    b = bf16x4_to_f32(make_uint2(h.z, h.w));
    if (ps > 0) {
        const uint4 m = *reinterpret_cast<const uint4*>(p + ps), l = *reinterpret_cast<const uint4*>(p + 2 * ps);
        const float4 ma = bf16x4_to_f32(make_uint2(m.x, m.y)), mb = bf16x4_to_f32(make_uint2(m.z, m.w));
        const float4 la = bf16x4_to_f32(make_uint2(l.x, l.y)), lb = bf16x4_to_f32(make_uint2(l.z, l.w));
        a.x = (a.x + ma.x) + la.x; a.y = (a.y + ma.y) + la.y; a.z = (a.z + ma.z) + la.z; a.w = (a.w + ma.w) + la.w;
        b.x = (b.x + mb.x) + lb.x; b.y = (b.y + mb.y) + lb.y; b.z = (b.z + mb.z) + lb.z; b.w = (b.w + mb.w) + lb.w;
    }
}

__device__ __forceinline__ void st8x(void* base, int64_t off, int64_t ps, float4 a, float4 b) {
    if (ps == 0) {
        float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + off);
        p[0] = a;
        p[1] = b;
        return;
    }
    unsigned short* p = reinterpret_cast<unsigned short*>(base) + off;
    if (ps > 0) {
        uint2 h0, m0, l0, h1, m1, l1;
        split3(a, h0, m0, l0);
        split3(b, h1, m1, l1);
        *reinterpret_cast<uint4*>(p) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        *reinterpret_cast<uint4*>(p + ps) = make_uint4(m0.x, m0.y, m1.x, m1.y);
        *reinterpret_cast<uint4*>(p + 2 * ps) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    } else {
        const uint2 r0 = round_bf16x4(a), r1 = round_bf16x4(b);
        *reinterpret_cast<uint4*>(p) = make_uint4(r0.x, r0.y, r1.x, r1.y);
    }
}

__device__ __forceinline__ void ld8x_hi(const void* base, int64_t off, int64_t ps, float4& a, float4& b) {
    if (ps == 0) {
        const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
        a = p[0];
        b = p[1];
        return;
    }
    const uint4 h = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(base) + off);
    a = bf16x4_to_f32(make_uint2(h.x, h.y));
    b = bf16x4_to_f32(make_uint2(h.z, h.w));
}

// the hi plane alone (sign / range tests on a saved activation): truncation keeps sign and x > 0, x < 6 for x in [0, 6]
__device__ __forceinline__ float4 ld4x_hi(const void* base, int64_t off, int64_t ps) {
    if (ps == 0) return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
    return bf16x4_to_f32(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off));
}

}  // namespace iswm
