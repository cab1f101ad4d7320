// Fused Swin MLP forward (swin_transformer.py:331 + 31-37):
//
//     y = x + rowscale * ( GELU( LN(x) W1^T + b1 ) W2^T + b2 )          x, y: fp32 [M, C];  W1: [4C, C];  W2: [C, 4C]
//
// for the narrow stages (C = 96, 192), where the unfused sequence LayerNorm -> fc1 (+GELU) -> fc2 (+residual) is bound
// by the HBM round trips of the 4C-wide hidden activation, not by MFMA work: unfused 40 B per token-channel forward
// (32 for the teacher, which saves nothing), fused 26 B with everything the backward needs still written (LN output,
// statistics, pre-activation, GELU output: the backward kernels are unchanged) and 8 B for the teacher.
//
// Work split.  A workgroup is 4 waves; a wave OWNS 32 token rows for the whole MLP, so nothing but the weight tiles is
// shared between waves.  MFMA shape: v_mfma_f32_32x32x16_bf16.  For a 32-wide chunk of the hidden dimension the wave
// computes the TRANSPOSED pre-activation  P^T[hidden 32][token 32] = W1_chunk[32 x C] * LN(x)^T  with LN(x) as the B
// operand, held in registers for the whole tile (lane (n, h): token n, channels 16s + 8h .. +7).  In the accumulator
// layout (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) lane (n, h) then holds, for ITS token n, the hidden
// units {8m + 4h + e}: after bias + GELU + rounding these 16 values ARE two A-operand fragments of the second GEMM
// y[token][c] += H[token][hidden] W2[c][hidden] -- an MFMA only needs both operands to agree on which k sits in which
// (lane >> 5, j) slot, so W2 is read with the matching permutation (two 8-byte LDS reads per fragment) and H never
// leaves the registers (the trick the 14x14 attention kernels use for P, window_attn_big.hip).
//
// Weights stream L2 -> LDS by LDS-DMA (buffer_load ... lds) in 32-hidden chunks (W1 rows [32 x C], W2 columns [C x 32]),
// double-buffered, one workgroup barrier per chunk; bank conflicts are removed by XOR-swizzling the 16-byte chunk index on
// the source address and on the fragment read (guide rule 21).  The loop body contains no vector-memory LOAD besides the
// DMA (the fc1 bias arrives through scalar loads), so the counted waits at the end of a chunk are exact.  Side outputs for
// the backward leave through wave-private, XOR-swizzled LDS slabs as 128-byte row pieces (two chunks at a time).
// LDS: 2 x (W1 + W2 chunk) + 2 slabs x 4 KiB x 4 waves = 80 KiB at C = 192 -> two workgroups per CU.
#include "common.h"
#include "../../include/esvit_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8v __attribute__((ext_vector_type(8)));

constexpr int MLP_WAVES = 4;
constexpr int MLP_ROWS = 32 * MLP_WAVES;  // token rows per workgroup
constexpr int HCH = 32;                   // hidden units per chunk
constexpr int SLAB = 32 * 128;            // bytes: [32 tokens][64 hidden] bf16

template <int C>
struct MlpCfg {
    static constexpr int KS1 = C / 16;            // k-steps of GEMM1 (k = channel)
    static constexpr int NT2 = C / 32;            // 32-channel output tiles of GEMM2
    static constexpr int W1_BYTES = HCH * C * 2;  // [32 hidden][C]
    static constexpr int W2_BYTES = C * HCH * 2;  // [C][32 hidden]
    static constexpr int N1 = (W1_BYTES / 1024 + MLP_WAVES - 1) / MLP_WAVES;  // DMA instructions per wave (1 KiB each)
    static constexpr int N2 = (W2_BYTES / 1024 + MLP_WAVES - 1) / MLP_WAVES;
    static constexpr int WBUF = W1_BYTES + W2_BYTES;
    static constexpr int LDS_BYTES = 2 * WBUF + MLP_WAVES * 2 * SLAB;
    static constexpr int M1 = C == 192 ? 7 : 3;  // swizzle mask of the W1 image (chunks per row: 24 = 3 x 8, 12 = 3 x 4)
    // W1 image: rows of 2C bytes.  One A-fragment read = 32 rows x 16 bytes at one chunk index: the 384-byte pitch (C = 192)
    // alternates two bank phases -> XOR the chunk with (row >> 1) & 7; the 192-byte pitch (C = 96) cycles four -> (row >> 2) & 3.
    __device__ __forceinline__ static int sw1(int row) { return C == 192 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
    __device__ __forceinline__ static int pos1(int chunk, int row) { return (chunk & ~M1) | ((chunk ^ sw1(row)) & M1); }
    // W2 image: rows of 64 bytes = 4 chunks; a fragment read touches 32 consecutive rows at one 8-byte slot -> XOR the
    // chunk with (row >> 2) & 3 (rows n and n + 16 still share a bank: 2-way on an 8-byte read, off the critical path)
    __device__ __forceinline__ static int sw2(int row) { return (row >> 2) & 3; }
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* base, long bytes) {
    const long capped = bytes > 0xfffffff0L ? 0xfffffff0L : (bytes < 0 ? 0 : bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)capped, 0x00020000);
}

template <typename F>
__device__ __forceinline__ void static_for2(F&& f) {
    f(std::integral_constant<int, 0>{});
    f(std::integral_constant<int, 1>{});
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// staged side output: token row n, 8-byte slot (4 hidden units) `slot8` of the 128-byte row; the 16-byte chunk index is
// XOR-ed with (row >> 1) & 7 so that both the 8-byte writes (32 rows, one slot) and the 16-byte reads (2 rows x 8 chunks)
// of a 16-lane group fall on 16 different 16-byte slots of the 256-byte bank row
__device__ __forceinline__ int slab_off(int row, int slot8) {
    return row * 128 + ((((slot8 >> 1) ^ (row >> 1)) & 7) * 16) + (slot8 & 1) * 8;
}

template <int C, bool SAVE>
__device__ __forceinline__ void mlp_fused_fwd_body(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const bf16* __restrict__ W1, const float* __restrict__ b1, const bf16* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ rowscale, long M, float* __restrict__ y, bf16* __restrict__ h_out, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, bf16* __restrict__ pre_out, bf16* __restrict__ act_out) {
    using Cfg = MlpCfg<C>;
    constexpr int H4 = 4 * C;
    constexpr int NPAIR = H4 / (2 * HCH);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, hh = lane >> 5;
    const long row0 = (long)blockIdx.x * MLP_ROWS + wave * 32;  // first token row of this wave
    const long row = row0 + n;
    const bool row_ok = row < M;
    const long rrow = row_ok ? row : (M - 1);  // out-of-range lanes compute on a valid row and store nothing
    const bool full_tile = row0 + 32 <= M;     // wave-uniform

    // ---- weight chunk DMA: per-lane source byte offsets of this wave's instructions for chunk 0; chunk q adds a scalar ----
    const __amdgpu_buffer_rsrc_t r1 = mk_rsrc(W1, (long)H4 * C * 2), r2 = mk_rsrc(W2, (long)C * H4 * 2);
    int voff1[Cfg::N1], voff2[Cfg::N2];
#pragma unroll
    for (int i = 0; i < Cfg::N1; ++i) {
        const int p = (wave * Cfg::N1 + i) * 64 + lane;  // 16-byte chunk index inside the W1 image
        const int r = p / (C / 8), cp = p % (C / 8);     // image row (hidden unit of the chunk), chunk position in the row
        voff1[i] = (r * C + Cfg::pos1(cp, r) * 8) * 2;   // (XOR is an involution: image position cp holds source chunk pos1(cp))
    }
#pragma unroll
    for (int i = 0; i < Cfg::N2; ++i) {
        const int p = (wave * Cfg::N2 + i) * 64 + lane;
        const int r = p / 4, cp = p % 4;                 // image row (output channel), chunk position (8 hidden units each)
        voff2[i] = (r * H4 + (cp ^ Cfg::sw2(r)) * 8) * 2;
    }
    auto issue_chunk = [&](int q, int buf) {
        char* w1 = smem + buf * Cfg::WBUF;
        char* w2 = w1 + Cfg::W1_BYTES;
        const int so1 = q * HCH * C * 2, so2 = q * HCH * 2;
#pragma unroll
        for (int i = 0; i < Cfg::N1; ++i)
            if ((wave * Cfg::N1 + i) * 1024 < Cfg::W1_BYTES)  // wave-uniform: the last wave may own fewer pieces
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_void*)(w1 + (wave * Cfg::N1 + i) * 1024), 16, voff1[i], so1, 0, 0);
#pragma unroll
        for (int i = 0; i < Cfg::N2; ++i)
            if ((wave * Cfg::N2 + i) * 1024 < Cfg::W2_BYTES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_void*)(w2 + (wave * Cfg::N2 + i) * 1024), 16, voff2[i], so2, 0, 0);
    };
    issue_chunk(0, 0);

    // ---- LayerNorm of this lane's half row, straight into the B-operand fragments of GEMM1 ----
    bf16x8 xb[Cfg::KS1];
    {
        const float* xr = x + rrow * C + 8 * hh;
        float xv[Cfg::KS1][8];
        float s1 = 0.f;
#pragma unroll
        for (int s = 0; s < Cfg::KS1; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 16 * s), b = *reinterpret_cast<const f32x4*>(xr + 16 * s + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[s][e] = a[e];
                xv[s][4 + e] = b[e];
                s1 += a[e] + b[e];
            }
        }
        s1 += __shfl_xor(s1, 32, 64);
        const float mean = s1 * (1.f / C);
        float s2 = 0.f;
#pragma unroll
        for (int s = 0; s < Cfg::KS1; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = xv[s][e] - mean;
                s2 += d * d;
            }
        s2 += __shfl_xor(s2, 32, 64);
        const float rstd = rsqrtf(s2 * (1.f / C) + eps);
        if constexpr (SAVE) {
            if (row_ok && hh == 0) {
                mean_out[row] = mean;
                rstd_out[row] = rstd;
            }
        }
#pragma unroll
        for (int s = 0; s < Cfg::KS1; ++s) {
            const float* gp = gamma + 16 * s + 8 * hh;
            const float* bp = beta + 16 * s + 8 * hh;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(bp), c1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xb[s][e] = (bf16)((xv[s][e] - mean) * rstd * g0[e] + c0[e]);
                xb[s][4 + e] = (bf16)((xv[s][4 + e] - mean) * rstd * g1[e] + c1[e]);
            }
            if constexpr (SAVE) {
                if (row_ok) *reinterpret_cast<bf16x8*>(h_out + row * C + 16 * s + 8 * hh) = xb[s];
            }
        }
    }

    f32x16 acc2[Cfg::NT2];
#pragma unroll
    for (int t = 0; t < Cfg::NT2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;

    char* slab_pre = smem + 2 * Cfg::WBUF + wave * 2 * SLAB;  // wave-private staging of the two side outputs
    char* slab_act = slab_pre + SLAB;

    wait_vm<0>();     // this wave's part of chunk 0 has landed (and its LN loads / stores are done)
    __syncthreads();  // ... everybody else's too

    for (int qp = 0; qp < NPAIR; ++qp) {
        static_for2([&](auto halfc) {
            constexpr int half = decltype(halfc)::value;  // 0: even chunk, 1: odd chunk of the pair
            const int q = 2 * qp + half;
            const int buf = half;                          // chunk q lives in buffer q & 1
            if (q + 1 < 2 * NPAIR) issue_chunk(q + 1, buf ^ 1);  // that buffer was released by the barrier ending chunk q - 1
            const char* w1 = smem + buf * Cfg::WBUF;
            const char* w2 = w1 + Cfg::W1_BYTES;

            // ---- GEMM1: P^T[32 hidden][32 tokens] = W1_chunk * LN(x)^T ----
            f32x16 acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
            for (int s = 0; s < Cfg::KS1; ++s) {  // A fragment: hidden row n of the chunk, channels 16s + 8hh .. +7
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(w1 + (n * (C / 8) + Cfg::pos1(2 * s + hh, n)) * 16);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb[s], acc1, 0, 0, 0);
            }
            // ---- bias, GELU, rounding: register 4m + e holds hidden unit 32q + 8m + 4hh + e of token n ----
            bf16x8 hf[2];
            bf16x4 pre4[4];
            const float* bq = b1 + q * HCH;  // wave-uniform address: scalar loads (no vmcnt traffic inside the loop)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float blo = bq[8 * m + e], bhi = bq[8 * m + 4 + e];
                    const float v = acc1[4 * m + e] + (hh ? bhi : blo);
                    if constexpr (SAVE) pre4[m][e] = (bf16)v;
                    hf[m >> 1][4 * (m & 1) + e] = (bf16)gelu_f(v);
                }
            }
            // ---- GEMM2: y[32 tokens][C] += H[32 tokens][32 hidden] W2_chunk^T; k-slot (hh, j) <-> hidden 16t + 4hh + j (j < 4),
            //      16t + 8 + 4hh + (j - 4) (j >= 4) ----
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int nt = 0; nt < Cfg::NT2; ++nt) {
                    const int c = 32 * nt + n;                         // B fragment column = output channel
                    const int sl0 = 4 * t + hh, sl1 = 4 * t + 2 + hh;  // 8-byte slots (4 hidden units each) of the 64-byte row
                    const char* rowp = w2 + c * 64;
                    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(rowp + (((sl0 >> 1) ^ Cfg::sw2(c)) * 16) + (sl0 & 1) * 8);
                    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(rowp + (((sl1 >> 1) ^ Cfg::sw2(c)) * 16) + (sl1 & 1) * 8);
                    const bf16x8 b = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[t], b, acc2[nt], 0, 0, 0);
                }
            }
            // ---- side outputs for the backward: pre-activation and GELU output, 64 hidden (128-byte row pieces) at a time ----
            if constexpr (SAVE) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int m = 0; m < 4; ++m) {  // slot of hidden 32 half + 8m + 4hh .. +3 in the 64-wide staged row
                    const int slot8 = 8 * half + 2 * m + hh;
                    *reinterpret_cast<bf16x4*>(slab_pre + slab_off(n, slot8)) = pre4[m];
                    const bf16x4 a4 = {hf[m >> 1][4 * (m & 1)], hf[m >> 1][4 * (m & 1) + 1], hf[m >> 1][4 * (m & 1) + 2], hf[m >> 1][4 * (m & 1) + 3]};
                    *reinterpret_cast<bf16x4*>(slab_act + slab_off(n, slot8)) = a4;
                }
                __builtin_amdgcn_wave_barrier();
                if constexpr (half == 1) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int r = 8 * p + (lane >> 3), cc = lane & 7;
                        const int off = r * 128 + ((cc ^ (r >> 1)) & 7) * 16;
                        const bf16x8 vp = *reinterpret_cast<const bf16x8*>(slab_pre + off);
                        const bf16x8 va = *reinterpret_cast<const bf16x8*>(slab_act + off);
                        if (row0 + r < M) {
                            *reinterpret_cast<bf16x8*>(pre_out + (row0 + r) * H4 + 64 * qp + 8 * cc) = vp;
                            *reinterpret_cast<bf16x8*>(act_out + (row0 + r) * H4 + 64 * qp + 8 * cc) = va;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            // the DMA of chunk q + 1 was issued before this chunk's 8 side-output stores (a full tile issues exactly 8): loads
            // and stores retire in order, so at most 8 outstanding means the DMA has landed while the stores may still fly
            if (SAVE && half == 1 && full_tile) wait_vm<8>();
            else wait_vm<0>();
            __syncthreads();  // chunk q + 1 landed for every wave; every wave is done reading chunk q
        });
    }

    // ---- epilogue: y = x + rowscale * (acc2 + b2); lane: channel 32nt + n, tokens (r & 3) + 8 (r >> 2) + 4hh ----
    float rs[16];
    long trow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long tr = row0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        trow[r] = tr < M ? tr : -1;
        rs[r] = (rowscale && tr < M) ? rowscale[tr] : 1.f;
    }
#pragma unroll
    for (int nt = 0; nt < Cfg::NT2; ++nt) {
        const int c = 32 * nt + n;
        const float bb = b2[c];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (trow[r] >= 0) y[trow[r] * C + c] = x[trow[r] * C + c] + rs[r] * (acc2[nt][r] + bb);
        }
    }
}

template <int C, bool SAVE>
__global__ __launch_bounds__(MLP_WAVES * 64, 2) void mlp_fused_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const bf16* __restrict__ W1, const float* __restrict__ b1, const bf16* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ rowscale, long M, float* __restrict__ y, bf16* __restrict__ h_out, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, bf16* __restrict__ pre_out, bf16* __restrict__ act_out) {
    mlp_fused_fwd_body<C, SAVE>(x, gamma, beta, eps, W1, b1, W2, b2, rowscale, M, y, h_out, mean_out, rstd_out, pre_out, act_out);
}

template <int C>
int launch_mlp(const float* x, const float* gamma, const float* beta, float eps, const void* W1, const float* b1, const void* W2,
               const float* b2, const float* rowscale, long M, float* y, void* h_out, float* mean_out, float* rstd_out, void* pre_out,
               void* act_out, hipStream_t stream) {
    const bool save = h_out != nullptr;
    const int grid = ceil_div(M, MLP_ROWS);
    const size_t lds = MlpCfg<C>::LDS_BYTES;
    if (save) {
        auto kern = mlp_fused_fwd_kernel<C, true>;
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            done = true;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(MLP_WAVES * 64), lds, stream, x, gamma, beta, eps, (const bf16*)W1, b1, (const bf16*)W2, b2,
                           rowscale, M, y, (bf16*)h_out, mean_out, rstd_out, (bf16*)pre_out, (bf16*)act_out);
    } else {
        auto kern = mlp_fused_fwd_kernel<C, false>;
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            done = true;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(MLP_WAVES * 64), lds, stream, x, gamma, beta, eps, (const bf16*)W1, b1, (const bf16*)W2, b2,
                           rowscale, M, y, (bf16*)nullptr, (float*)nullptr, (float*)nullptr, (bf16*)nullptr, (bf16*)nullptr);
    }
    ESVIT_CHECK_LAUNCH("esvit_mlp_fused_fwd");
    return ESVIT_OK;
}

}  // namespace

extern "C" int esvit_mlp_fused_supported(int dtype, int C) { return dtype == ESVIT_BF16 && (C == 96 || C == 192); }

extern "C" int esvit_mlp_fused_fwd(int dtype, const float* x, const float* gamma, const float* beta, float eps, const void* W1,
                                   const float* b1, const void* W2, const float* b2, const float* rowscale, int64_t M, int C,
                                   float* y, void* h_out, float* mean_out, float* rstd_out, void* pre_out, void* act_out,
                                   esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(esvit_mlp_fused_supported(dtype, C), "esvit_mlp_fused_fwd: bf16 activations and C in {96, 192} only (C=%d)", C);
    ESVIT_CHECK_ARG(x && gamma && beta && W1 && b1 && W2 && b2 && y && M > 0, "esvit_mlp_fused_fwd: null pointer / empty input");
    const bool any = h_out || mean_out || rstd_out || pre_out || act_out, all = h_out && mean_out && rstd_out && pre_out && act_out;
    ESVIT_CHECK_ARG(!any || all, "esvit_mlp_fused_fwd: the five side outputs are written together or not at all");
    ESVIT_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)W1 % 16 == 0) && ((uintptr_t)W2 % 16 == 0) &&
                        ((uintptr_t)gamma % 16 == 0) && ((uintptr_t)beta % 16 == 0),
                    "esvit_mlp_fused_fwd: operands must be 16-byte aligned");
    ESVIT_CHECK_ARG((long)M * 4 * C * 2 < 0x7fffffff00L, "esvit_mlp_fused_fwd: too many rows");
    if (C == 96) return launch_mlp<96>(x, gamma, beta, eps, W1, b1, W2, b2, rowscale, M, y, h_out, mean_out, rstd_out, pre_out, act_out, stream);
    return launch_mlp<192>(x, gamma, beta, eps, W1, b1, W2, b2, rowscale, M, y, h_out, mean_out, rstd_out, pre_out, act_out, stream);
}
