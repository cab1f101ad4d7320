// Shared pieces of the 16-token-per-wave fused kernels (mlp_fused16.hip, attn_branch.hip): LDS weight images with their
// conflict-free swizzles, LDS-DMA plumbing, bf16 packing / row swaps, in-register LayerNorm statistics.  See mlp_fused16.hip for
// the fragment conventions.  Everything here has internal linkage (included inside each translation unit once).
#pragma once
#include "common.h"

namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int HCH = 32;  // hidden units per chunk

template <int C>
struct Cfg16 {
    static constexpr int KS = C / 32;             // k-steps of a product over the channels
    static constexpr int MT = C / 16;             // 16-channel output tiles
    static constexpr int UA = C / 8;              // 16-byte units per image-A row
    static constexpr int A_BYTES = HCH * C * 2;   // image A: [32 hidden][C]   (rows of W1p, or of W2Tp)
    static constexpr int B_BYTES = C * HCH * 2;   // image B: [C][32 hidden]   (columns of W2, or of W1T)
    static constexpr int PA = A_BYTES / 1024;     // 1 KiB DMA pieces
    static constexpr int PB = B_BYTES / 1024;
    // image A unit swizzle (XOR inside aligned blocks of 4 / 8 / 16 units; conflict-free for the fragment read below)
    __device__ __forceinline__ static int swa(int unit, int row) {
        if constexpr (C == 96) return (unit & ~3) | ((unit ^ ((row >> 2) & 3)) & 3);
        else if constexpr (C == 192) return (unit & ~7) | ((unit ^ ((((row >> 3) & 3) << 1) | ((row >> 1) & 1))) & 7);
        else return (unit & ~15) | ((unit ^ (((row & 3) << 2) | ((-(row >> 3)) & 3))) & 15);
    }
    __device__ __forceinline__ static int swb(int unit, int row) { return unit ^ ((row >> 1) & 3); }
    __device__ __forceinline__ static int voff_a(int piece, int lane) {
        const int p = piece * 64 + lane;
        const int r = p / UA, u = p % UA;
        return (r * C + swa(u, r) * 8) * 2;  // (XOR is an involution: image unit u holds source unit swa(u))
    }
    __device__ __forceinline__ static int voff_b(int piece, int lane) {
        const int p = piece * 64 + lane;
        const int r = p / 4, u = p % 4;
        return (r * 4 * C + swb(u, r) * 8) * 2;
    }
    // fragment byte offsets inside an image
    __device__ __forceinline__ static int frag_a(int mt, int ks, int c, int g) {  // MFMA row c of hidden tile mt
        const int row = 8 * (c >> 2) + 4 * mt + (c & 3);
        return (row * UA + swa(4 * ks + g, row)) * 16;
    }
    __device__ __forceinline__ static int frag_b(int mt, int c, int g) {  // channel 16 mt + c, hidden 8g .. 8g+7
        const int row = 16 * mt + c;
        return row * 64 + swb(g, row) * 16;
    }
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* base, long bytes) {
    const long capped = bytes > 0xfffffff0L ? 0xfffffff0L : (bytes < 0 ? 0 : bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)capped, 0x00020000);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// raw barrier of the chunk loops (NOT __syncthreads(): its fence drains vmcnt, i.e. the LDS-DMA in flight and the stores)
__device__ __forceinline__ void chunk_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(unsigned, v);
}

// lanes g (even) and g + 1 each hold two 4-element groups x | y.  Afterwards the even lane holds (own x, partner's x) and the
// odd lane (partner's y, own y): v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of
// its second.
__device__ __forceinline__ void row_swap(unsigned& x, unsigned& y) {
    const u32x2 r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void gelu_both(float v, float& g, float& dg) {  // exact erf-GELU and GELU' from one exponential
    const float av = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * av);
    const float e = __expf(-0.5f * v * v);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erfv = copysignf(1.f - poly * e, v);
    const float cdf = 0.5f * (1.f + erfv);
    g = v * cdf;
    dg = cdf + v * 0.39894228040143268f * e;
}

// LayerNorm statistics of a token whose C channels are spread over the four lanes (c, g = 0..3); v: this lane's C / 4 values
template <int N>
__device__ __forceinline__ void row_stats(const float (&v)[N], float inv_c, float eps, float& mean, float& rstd) {
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) s1 += v[i];
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    mean = s1 * inv_c;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float d = v[i] - mean;
        s2 += d * d;
    }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    rstd = rsqrtf(s2 * inv_c + eps);
}

// bf16 store of a [16 tok][C] tile held as MT x f32x4 per lane (channels 16 mt + 4g + r): 16 bytes per lane, 64 contiguous bytes
// per token and instruction (tiles mt, mt + 1 -> channels 16 mt .. 16 mt + 31)
template <int MT>
__device__ __forceinline__ void store_bf16_tiles(bf16* __restrict__ dst_row, const float (&v)[MT][4], int g, bool ok) {
#pragma unroll
    for (int mt = 0; mt < MT; mt += 2) {
        unsigned x0 = pack2(v[mt][0], v[mt][1]), x1 = pack2(v[mt][2], v[mt][3]);
        unsigned y0 = pack2(v[mt + 1][0], v[mt + 1][1]), y1 = pack2(v[mt + 1][2], v[mt + 1][3]);
        row_swap(x0, y0);
        row_swap(x1, y1);
        // even g: tile mt, channels 4g .. 4g+7;  odd g: tile mt + 1, channels 4(g-1) .. 4(g-1)+7
        const int ch = 16 * (mt + (g & 1)) + 4 * (g & ~1);
        if (ok) *reinterpret_cast<u32x4*>(dst_row + ch) = u32x4{x0, x1, y0, y1};
    }
}


}  // namespace
