// MFMA fragment helpers shared by the GEMM and window-attention kernels (gfx950).
//
// Fragment convention (16x16 output tile, 32-deep k-step): lane l = 16*g + c holds, for tile row
// (A) or tile column (B) `c`, the 8 k-values k = 8g .. 8g+7.  Both operands use the same k
// assignment, so any MFMA whose hardware k-order differs is still a correct dot product.
// Accumulator (C/D) layout: acc[r] = D[row = 4g + r][col = c].
#pragma once
#include "common.h"

template <typename T>
struct Frag;
template <>
struct Frag<bf16> {
    bf16x8 v;
};
template <>
struct Frag<float> {
    float v[8];
};

__device__ __forceinline__ void mma(const Frag<bf16>& a, const Frag<bf16>& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ void mma(const Frag<float>& a, const Frag<float>& b, f32x4& c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
}

typedef short s16x8 __attribute__((ext_vector_type(8)));


// k-contiguous LDS image [rows][LD]: 8 elements at (row r0+c, k = k0+8g..)
template <typename T>
__device__ __forceinline__ Frag<T> frag_kc(const T* lds, int LD, int r0, int k0, int c, int g) {
    Frag<T> f;
    const T* p = lds + (r0 + c) * LD + k0 + 8 * g;
    if constexpr (sizeof(T) == 2) {
        f.v = *reinterpret_cast<const bf16x8*>(p);
    } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f.v[j] = a[j];
            f.v[4 + j] = b[j];
        }
    }
    return f;
}

// k-strided LDS image [k][LD] (rows of the tile are the contiguous dim): element (row r0+c,
// k = k0+8g+j).  bf16 uses the gfx950 transpose read: the 16 lanes of group g fetch the 4(k) x
// 16(row) block cooperatively -- lane c supplies the 8 bytes at (k = k0+8g+4h+(c>>2), rows
// r0+4(c&3)..+3) and receives column r0+c (guide T10: result[j] = chunk fetched by lane
// 4j+(c>>2) of the group, element c&3).
template <typename T, bool USE_TR>
__device__ __forceinline__ Frag<T> frag_ks(const T* lds, int LD, int r0, int k0, int c, int g) {
    Frag<T> f;
    if constexpr (sizeof(T) == 2 && USE_TR) {
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const T* p0 = lds + (k0 + 8 * g + (c >> 2)) * LD + r0 + 4 * (c & 3);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * LD));
        const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        f.v = __builtin_bit_cast(bf16x8, both);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) f.v[j] = lds[(k0 + 8 * g + j) * LD + r0 + c];
    }
    return f;
}
