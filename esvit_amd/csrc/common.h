// esvit_amd device/host common helpers (gfx950 / CDNA4 only).
//
// Everything in csrc/ is written for one target: MI355X (gfx950), wave64,
// MFMA 16x16x32 bf16 / 16x16x4 f32, 160 KiB LDS per CU.  No portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define ESVIT_F32 0
#define ESVIT_BF16 1

#define ESVIT_OK 0
#define ESVIT_ERR_ARG (-1)
#define ESVIT_ERR_HIP (-2)
#define ESVIT_ERR_UNSUPPORTED (-3)

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
// error plumbing (host)
// ---------------------------------------------------------------------------
void esvit_set_error(const char* fmt, ...);

#define ESVIT_CHECK_ARG(cond, ...)                         \
    do {                                                   \
        if (!(cond)) {                                     \
            esvit_set_error(__VA_ARGS__);                  \
            return ESVIT_ERR_ARG;                          \
        }                                                  \
    } while (0)

#define ESVIT_CHECK_LAUNCH(name)                                                        \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            esvit_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return ESVIT_ERR_HIP;                                                       \
        }                                                                               \
    } while (0)

// ---------------------------------------------------------------------------
// element access helpers: activations are stored either as f32 or bf16
// ---------------------------------------------------------------------------
template <typename T>
struct ElemTraits;
template <>
struct ElemTraits<float> {
    static constexpr int VEC = 4;  // elements per 16-byte vector
    static constexpr int DT = ESVIT_F32;
};
template <>
struct ElemTraits<bf16> {
    static constexpr int VEC = 8;
    static constexpr int DT = ESVIT_BF16;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// 16-byte vector of T with float views of each lane
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
    f32x4 v;
    static constexpr int N = 4;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <>
struct Vec16<bf16> {
    bf16x8 v;
    static constexpr int N = 8;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
};

template <typename T>
__device__ __forceinline__ Vec16<T> ld16(const T* p) {
    Vec16<T> r;
    r.v = *reinterpret_cast<const decltype(r.v)*>(p);
    return r;
}
template <typename T>
__device__ __forceinline__ void st16(T* p, const Vec16<T>& r) {
    *reinterpret_cast<decltype(r.v)*>(p) = r.v;
}
template <typename T>
__device__ __forceinline__ Vec16<T> zero16() {
    Vec16<T> r;
#pragma unroll
    for (int i = 0; i < Vec16<T>::N; ++i) r.set(i, 0.f);
    return r;
}

// ---------------------------------------------------------------------------
// 16-byte raw-buffer store with the uniform part of the address added to the per-lane offset instead of riding in the instruction's
// SGPR offset field.  Measured on MI355X (tools/probe/diag_attn64b.py, tools/isa_store_hazard.py): with every CU fully occupied, a
// `buffer_store_dwordx4 ... offen` with an SGPR offset that is DIRECTLY followed by a VALU write of its first data register stored the
// new register value in a quarter of the lanes -- the case LLVM's hazard recogniser exempts from the "VMEM store of more than 64 bits,
// then a write of its data VGPRs" wait state.  With a constant-zero offset field the compiler inserts that wait state itself.
// (Out-of-range lanes stay out of range: the offsets are unsigned 32-bit and the ranges are below 2 GiB.)
// ---------------------------------------------------------------------------
typedef unsigned int esvit_u32x4 __attribute__((ext_vector_type(4)));
template <int AUX = 0, typename V>
__device__ __forceinline__ void buffer_store_b128(const V& data, __amdgpu_buffer_rsrc_t rsrc, unsigned lane_offset, unsigned uniform_offset) {
    static_assert(sizeof(V) == 16, "buffer_store_b128 takes 16 bytes");
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(esvit_u32x4, data), rsrc, (int)(lane_offset + uniform_offset), 0, AUX);
}

// ---------------------------------------------------------------------------
// 16-byte row stores straight from TRANSPOSED accumulators (a product formed as D^T = B^T A^T leaves lane (c, g) with, for its
// column c -- a token / window slot --, the 4 consecutive rows 16 t + 4g + r of every 16-row tile t -- channels).  fp32: each tile
// is a 16-byte vector as it is.  bf16: two tiles t, t + 1 are packed to 2 x 2 dwords and one pair is exchanged between lanes g and
// g ^ 1 (v_permlane16_swap), after which an even g holds channels 16 t + 4g .. + 7 and an odd g channels 16 (t + 1) + 4 (g - 1) .. + 7.
// ---------------------------------------------------------------------------
typedef unsigned int esvit_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned esvit_pack2_bf16(float a, float b) {
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const bf16x2_ v = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ esvit_u32x4 esvit_pack_tile_pair_bf16(const f32x4& t0, const f32x4& t1) {
    unsigned x0 = esvit_pack2_bf16(t0[0], t0[1]), x1 = esvit_pack2_bf16(t0[2], t0[3]);
    unsigned y0 = esvit_pack2_bf16(t1[0], t1[1]), y1 = esvit_pack2_bf16(t1[2], t1[3]);
    const esvit_u32x2 a = __builtin_amdgcn_permlane16_swap(x0, y0, false, false), b = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
    return esvit_u32x4{a[0], b[0], a[1], b[1]};
}
// first channel (relative to tile t) of the 8 consecutive channels a lane holds after esvit_pack_tile_pair_bf16
__device__ __forceinline__ int esvit_tile_pair_ch0(int g) { return 16 * (g & 1) + 4 * (g & ~1); }

// ---------------------------------------------------------------------------
// butterfly steps across the four 16-lane rows of a wave on the VALU
// ---------------------------------------------------------------------------
// v_permlane16_swap / v_permlane32_swap exchange the odd 16- / 32-lane rows of their first operand with the even rows of their
// second: fed the same value twice, the two results are v[l] and v[l ^ 16] (v[l ^ 32]) in some order on every lane -- a butterfly
// step on the VALU (__shfl_xor goes through the LDS crossbar).  The instruction is written in inline asm WITH ITS OWN WAIT STATES:
//  * through the builtin, hipcc 7.2 folds `bitcast<float>(r[1])` of the result pair to `bitcast<float>(r[0])`
//    (tools/probe/permlane_swap_fold.hip: `v_add_f32 v1, v1, v1` after the swap; the integer uses -- esvit_pack_tile_pair_bf16,
//    fused16.h: row_swap -- are right);
//  * with the builtin and empty asm statements around it as a workaround, the compiler put ONE wait state between the swap and
//    the VALU instruction that reads its results; in attn_bwd3's reductions that made the step's gradients differ between
//    identical runs on some boxes (tests/test_dist_gpu.py, round 5: 0 / 10 failures with ds_bpermute reductions, 12 / 14 with
//    these) -- a read-after-swap hazard the hazard recogniser does not cover.  Four wait states on either side here.
// Used by the fused attention branch only; the older attention kernels keep their ds_bpermute reductions.
__device__ __forceinline__ void swap16(float v, float& x, float& y) {
    x = v;
    y = v;
    asm volatile("s_nop 3\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 3" : "+v"(x), "+v"(y));
}
__device__ __forceinline__ void swap32(float v, float& x, float& y) {
    x = v;
    y = v;
    asm volatile("s_nop 3\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 3" : "+v"(x), "+v"(y));
}
__device__ __forceinline__ float xor16_sum(float v) {
    float x, y;
    swap16(v, x, y);
    return x + y;
}
__device__ __forceinline__ float xor32_sum(float v) {
    float x, y;
    swap32(v, x, y);
    return x + y;
}
__device__ __forceinline__ float xor16_max(float v) {
    float x, y;
    swap16(v, x, y);
    return fmaxf(x, y);
}
__device__ __forceinline__ float xor32_max(float v) {
    float x, y;
    swap32(v, x, y);
    return fmaxf(x, y);
}

// ---------------------------------------------------------------------------
// wave / block reductions (wave = 64 lanes)
// ---------------------------------------------------------------------------
// The dispatcher places workgroup b on XCD b % 8 (eight XCDs, one L2 each).  Bijective renumbering that gives every XCD a
// contiguous range of unit ids, so units that touch the same cache lines (the heads of one window: a head's 64-byte
// q/k/v slice is half a 128-byte line) or the same operand panel meet in one L2 instead of being fetched once per XCD.
__device__ __forceinline__ int xcd_contiguous_id(int b, int total) {
    const int q = total / 8, r = total % 8;
    const int xcd = b % 8, idx = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), every lane gets the result: four VALU adds with DPP operands (quad
// permutes, then the half-row and row mirrors) -- __shfl_xor compiles the 4- and 8-lane steps to ds_bpermute (LDS crossbar round trips)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// reduce over a power-of-two lane group of width G (G <= 64)
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x == NT (multiple of 64); scratch >= NT/64 floats
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) r += scratch[i];
    return r;
}
template <int NT>
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) scratch[w] = v;
    __syncthreads();
    float r = scratch[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

// erf to fp32 round-off without the branches of libm's erff: Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);  // v_rcp_f32: 1 ulp, far inside the 1.5e-7 of the fit
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.f - poly * __expf(-ax * ax);
    return copysignf(r, x);
}
// erf-GELU (reference: nn.GELU default, "none" approximation) and its derivative
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erf_fast(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {  // one exponential: exp(-x^2 / 2) serves the erf fit and the density (as mlp_fused16's gelu_both)
    const float ax = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
    const float e = __expf(-0.5f * x * x);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float cdf = 0.5f * (1.f + copysignf(1.f - poly * e, x));
    return cdf + x * 0.39894228040143268f * e;
}

// QuickGELU of the CvT feed-forward (cvt_v4_transformer.py:44-46) and its derivative
__device__ __forceinline__ float qgelu_f(float x) { return x * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float qgelu_grad_f(float x) {
    const float s = __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * x));
    return s * (1.f + 1.702f * x * (1.f - s));
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// One-time raise of a kernel's dynamic-LDS cap, per DEVICE: the attribute lives with the device's copy of the code object, so a
// process that drives a second GPU (tests, a single-process launcher) must set it there too.  `mask`: one static word per kernel
// instantiation at the call site (bit = device ordinal).
template <typename K>
static inline void esvit_raise_lds(K kern, int bytes, unsigned long long& mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(mask & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        mask |= bit;
    }
}


// out[c] (+)= sum_b ws[b*ld + c], c < ncols  (elementwise.hip) -- second stage of every two-stage column reduction
int esvit_partial_reduce(const float* ws, int nblk, int ncols, long ld, float* out, int accumulate, hipStream_t stream);
