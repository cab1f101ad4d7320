// Fused Swin MLP forward for inference-mode passes (swin_transformer.py:331 + 31-37):
//
//     y = x + rowscale * ( GELU( LN(x) W1^T + b1 ) W2^T + b2 )          x, y: fp32 [M, C];  W1: [4C, C];  W2: [C, 4C]
//
// for the narrow stages (C = 96, 192) of the TEACHER, which saves nothing for a backward: the unfused sequence LayerNorm ->
// fc1 (+GELU) -> fc2 (+residual) moves 32 B per token-channel through HBM, most of it the 4C-wide hidden activation; fused
// it is 8 B.  (A variant that also wrote the LayerNorm output / pre-activation / GELU output the student's backward reads
// was measured: those 18 B per token-channel of side outputs leave it no faster than the unfused kernels --
// profiles/r02_mlp_fused.jsonl -- so the student keeps the unfused path.)
//
// Work split.  A workgroup is 4 waves; a wave OWNS 32 token rows for the whole MLP, so nothing but the weight tiles is
// shared between waves.  MFMA shape: v_mfma_f32_32x32x16_bf16.  For a 32-wide chunk of the hidden dimension the wave
// computes the TRANSPOSED pre-activation  P^T[hidden 32][token 32] = W1_chunk[32 x C] * LN(x)^T  with LN(x) as the B
// operand, held in registers for the whole tile (lane (n, h): token n, channels 16s + 8h .. +7).  In the accumulator
// layout (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) lane (n, h) then holds, for ITS token n, the
// MFMA rows {8m + 4h + e}: after bias + GELU + rounding these 16 values ARE two A-operand fragments of the second GEMM
// y[token][c] += H[token][hidden] W2[c][hidden], so H never leaves the registers (the trick the 14x14 attention kernels use
// for P, window_attn_big.hip).  Which hidden unit an MFMA row computes is free -- it is just the W1 row the A fragment
// reads -- so row i is given hidden unit rho(i) (bits 2 and 3 of i swapped): each lane half then owns 8 CONSECUTIVE hidden
// units per k-step and W2 is read in its natural order, one 16-byte LDS read per fragment.
//
// Weights stream L2 -> LDS by LDS-DMA (buffer_load ... lds) in 32-hidden chunks (W1 rows [32 x C], W2 columns [C x 32]),
// three buffers (chunk q + 2 is requested while chunk q is computed), one workgroup barrier per chunk; bank conflicts are
// removed by XOR-swizzling the 16-byte chunk index on the source address and on the fragment read (guide rule 21; measured
// SQ_LDS_BANK_CONFLICT = 0).  The loop body contains no vector-memory operation besides the DMA (the fc1 bias arrives
// through scalar loads), so the counted wait at the end of a chunk is exact.  LDS: 3 x (W1 + W2 chunk) = 36 / 72 KiB.
#include "common.h"
#include "../../include/esvit_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8v __attribute__((ext_vector_type(8)));

constexpr int MLP_WAVES = 4;
constexpr int MLP_ROWS = 32 * MLP_WAVES;  // token rows per workgroup
constexpr int HCH = 32;                   // hidden units per chunk

template <int C>
struct MlpCfg {
    static constexpr int KS1 = C / 16;            // k-steps of GEMM1 (k = channel)
    static constexpr int NT2 = C / 32;            // 32-channel output tiles of GEMM2
    static constexpr int W1_BYTES = HCH * C * 2;  // [32 hidden][C]
    static constexpr int W2_BYTES = C * HCH * 2;  // [C][32 hidden]
    static constexpr int P1 = W1_BYTES / 1024;    // 1 KiB DMA pieces of the W1 image, then of the W2 image
    static constexpr int PPW = (W1_BYTES + W2_BYTES) / 1024 / MLP_WAVES;  // pieces (= DMA instructions) per wave and chunk
    static_assert((W1_BYTES + W2_BYTES) % (1024 * MLP_WAVES) == 0, "every wave issues the same number of DMA instructions");
    static constexpr int WBUF = W1_BYTES + W2_BYTES;
    static constexpr int NBUF = 3;                                   // chunk q + 2 is requested while chunk q is computed
    static constexpr int LDS_BYTES = NBUF * WBUF;
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

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int C>
__device__ __forceinline__ void mlp_fused_fwd_body(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const bf16* __restrict__ W1, const float* __restrict__ b1, const bf16* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ rowscale, long M, float* __restrict__ y) {
    using Cfg = MlpCfg<C>;
    constexpr int H4 = 4 * C;
    constexpr int NCHUNK = H4 / HCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, hh = lane >> 5;
    const long row0 = (long)blockIdx.x * MLP_ROWS + wave * 32;  // first token row of this wave
    const long row = row0 + n;
    const bool row_ok = row < M;
    const long rrow = row_ok ? row : (M - 1);  // out-of-range lanes compute on a valid row and store nothing

    // ---- weight chunk DMA: per-lane source byte offsets of this wave's instructions for chunk 0; chunk q adds a scalar ----
    const __amdgpu_buffer_rsrc_t r1 = mk_rsrc(W1, (long)H4 * C * 2), r2 = mk_rsrc(W2, (long)C * H4 * 2);
    int voff[Cfg::PPW];
#pragma unroll
    for (int i = 0; i < Cfg::PPW; ++i) {
        const int piece = wave * Cfg::PPW + i;  // wave-uniform
        if (piece < Cfg::P1) {
            const int p = piece * 64 + lane;              // 16-byte chunk index inside the W1 image
            const int r = p / (C / 8), cp = p % (C / 8);  // image row (hidden unit of the chunk), chunk position in the row
            voff[i] = (r * C + Cfg::pos1(cp, r) * 8) * 2;  // (XOR is an involution: image position cp holds source chunk pos1(cp))
        } else {
            const int p = (piece - Cfg::P1) * 64 + lane;
            const int r = p / 4, cp = p % 4;              // image row (output channel), chunk position (8 hidden units each)
            voff[i] = (r * H4 + (cp ^ Cfg::sw2(r)) * 8) * 2;
        }
    }
    auto issue_chunk = [&](int q, int buf) {
        char* img = smem + buf * Cfg::WBUF;  // W1 image, then W2 image
        const int so1 = q * HCH * C * 2, so2 = q * HCH * 2;
#pragma unroll
        for (int i = 0; i < Cfg::PPW; ++i) {
            const int piece = wave * Cfg::PPW + i;
            if (piece < Cfg::P1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_void*)(img + piece * 1024), 16, voff[i], so1, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_void*)(img + piece * 1024), 16, voff[i], so2, 0, 0);
        }
    };
    issue_chunk(0, 0);
    issue_chunk(1, 1);

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
        }
    }

    f32x16 acc2[Cfg::NT2];
#pragma unroll
    for (int t = 0; t < Cfg::NT2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;

    wait_vm<0>();     // this wave's parts of chunks 0 and 1 have landed (and its LN loads / stores are done)
    __syncthreads();  // ... everybody else's too

    // MFMA row i of the transposed pre-activation tile holds hidden unit rho(i) of the chunk, rho swapping bits 2 and 3 of i:
    // lane (n, hh) then owns, for k-step t of the second GEMM, the EIGHT CONSECUTIVE hidden units 16t + 8hh .. +7 (registers
    // 8t .. 8t+7), i.e. the standard "half hh holds k = 8hh + j" operand convention -- W2 is read in its natural order with one
    // 16-byte LDS read per fragment.
    const int rho_n = (n & ~12) | ((n & 4) << 1) | ((n & 8) >> 1);
    int buf = 0;
    for (int q = 0; q < NCHUNK; ++q) {
        const bool more = q + 2 < NCHUNK;
        if (more) issue_chunk(q + 2, buf == 0 ? 2 : buf - 1);  // the buffer of chunk q - 1, released by the barrier that ended it
        const char* w1 = smem + buf * Cfg::WBUF;
        const char* w2 = w1 + Cfg::W1_BYTES;

        // ---- GEMM1: P^T[32 hidden][32 tokens] = W1_chunk * LN(x)^T (two accumulators: half the dependent-MFMA chain) ----
        f32x16 acc1a, acc1b;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1a[r] = acc1b[r] = 0.f;
#pragma unroll
        for (int s = 0; s < Cfg::KS1; ++s) {  // A fragment: W1 row rho(n) of the chunk, channels 16s + 8hh .. +7
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(w1 + (rho_n * (C / 8) + Cfg::pos1(2 * s + hh, rho_n)) * 16);
            if (s & 1) acc1b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb[s], acc1b, 0, 0, 0);
            else acc1a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb[s], acc1a, 0, 0, 0);
        }
        // ---- bias, GELU, rounding: register 8t + e (e < 8) holds hidden unit 32q + 16t + 8hh + e of token n ----
        bf16x8 hf[2];
        const float* bq = b1 + q * HCH;  // wave-uniform address: scalar loads (no vector-memory load inside the loop)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // MFMA row of register 8t + e: (r & 3) + 8 (r >> 2) + 4hh with r = 8t + e  ->  hidden rho(row) = 16t + 8hh + e
                const float blo = bq[16 * t + e], bhi = bq[16 * t + 8 + e];
                const float v = acc1a[8 * t + e] + acc1b[8 * t + e] + (hh ? bhi : blo);
                hf[t][e] = (bf16)gelu_f(v);
            }
        // ---- GEMM2: y[32 tokens][C] += H[32 tokens][32 hidden] W2_chunk^T ----
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int nt = 0; nt < Cfg::NT2; ++nt) {
                const int c = 32 * nt + n;  // B fragment: output channel c, hidden 16t + 8hh .. +7 of the chunk
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(w2 + c * 64 + (((2 * t + hh) ^ Cfg::sw2(c)) * 16));
                acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf[t], b, acc2[nt], 0, 0, 0);
            }
        }
        // chunk q + 1 must have landed before the barrier; the DMA of chunk q + 2, issued after it, may stay in flight
        if (more) wait_vm<Cfg::PPW>();
        else wait_vm<0>();
        __syncthreads();  // chunk q + 1 landed for every wave; every wave is done reading chunk q
        buf = buf == 2 ? 0 : buf + 1;
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

// waves per SIMD the register allocation is asked to fit: C = 96 needs ~128 registers -> 4; C = 192 (96 accumulator + 48
// operand registers) -> 2
template <int C>
__global__ __launch_bounds__(MLP_WAVES * 64, (C == 96 ? 4 : 2)) void mlp_fused_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const bf16* __restrict__ W1, const float* __restrict__ b1, const bf16* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ rowscale, long M, float* __restrict__ y) {
    mlp_fused_fwd_body<C>(x, gamma, beta, eps, W1, b1, W2, b2, rowscale, M, y);
}

template <int C>
int launch_mlp(const float* x, const float* gamma, const float* beta, float eps, const void* W1, const float* b1, const void* W2,
               const float* b2, const float* rowscale, long M, float* y, hipStream_t stream) {
    const int grid = ceil_div(M, MLP_ROWS);
    const size_t lds = MlpCfg<C>::LDS_BYTES;
    auto kern = mlp_fused_fwd_kernel<C>;
    static bool done = false;  // one-time raise of the dynamic LDS cap (idempotent)
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MLP_WAVES * 64), lds, stream, x, gamma, beta, eps, (const bf16*)W1, b1, (const bf16*)W2, b2,
                       rowscale, M, y);
    ESVIT_CHECK_LAUNCH("esvit_mlp_fused_fwd");
    return ESVIT_OK;
}

}  // namespace

int esvit_i_mlp_fused_supported(int dtype, int C) { return dtype == ESVIT_BF16 && (C == 96 || C == 192); }  // esvit_query

extern "C" int esvit_mlp_fused_fwd(int dtype, const float* x, const float* gamma, const float* beta, float eps, const void* W1,
                                   const float* b1, const void* W2, const float* b2, const float* rowscale, int64_t M, int C,
                                   float* y, esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(esvit_i_mlp_fused_supported(dtype, C), "esvit_mlp_fused_fwd: bf16 activations and C in {96, 192} only (C=%d)", C);
    ESVIT_CHECK_ARG(x && gamma && beta && W1 && b1 && W2 && b2 && y && M > 0, "esvit_mlp_fused_fwd: null pointer / empty input");
    ESVIT_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)W1 % 16 == 0) && ((uintptr_t)W2 % 16 == 0) &&
                        ((uintptr_t)gamma % 16 == 0) && ((uintptr_t)beta % 16 == 0),
                    "esvit_mlp_fused_fwd: operands must be 16-byte aligned");
    if (C == 96) return launch_mlp<96>(x, gamma, beta, eps, W1, b1, W2, b2, rowscale, M, y, stream);
    return launch_mlp<192>(x, gamma, beta, eps, W1, b1, W2, b2, rowscale, M, y, stream);
}
