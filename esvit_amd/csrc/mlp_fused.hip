// Fused Swin MLP branch for the narrow stages (C = 96, 192), forward AND backward (swin_transformer.py:331 + 31-37):
//
//     y = x + rowscale * ( GELU( LN(x) W1^T + b1 ) W2^T + b2 )          x, y: fp32 [M, C];  W1: [4C, C];  W2: [C, 4C]
//
// The unfused sequence LayerNorm -> fc1 (+GELU) -> fc2 (+residual) moves 40 B per token-channel through HBM in the forward
// (most of it the 4C-wide hidden activation, written twice for the backward) and 64 B in the backward.  Here:
//
//   esvit_mlp_fused_fwd   8 B per token-channel (x in, y out); NOTHING hidden-sized is saved -- the backward recomputes it.
//                         Optionally it also emits LayerNorm(y) with the NEXT block's norm1 parameters (bf16) and its row
//                         statistics, so that block's LayerNorm launch disappears.
//   esvit_mlp_fused_bwd   recomputes LN(x) and the pre-activation chunk by chunk, forms dA = (dy W2) o GELU'(A) and
//                         dH = dA W1 with the hidden tile in registers, applies the LayerNorm backward in registers and writes
//                         dL/dx (fp32), its activation-dtype copy, and -- once, coalesced -- the three operands the two
//                         weight-gradient GEMMs need: GELU(A), dA (bf16 [M, 4C]) and xhat = (x - mean) rstd (bf16 [M, C]).
//                         The LayerNorm parameter gradients come out of the fc1 weight gradient (esvit_ln_fold_finish below),
//                         so no cross-lane column reduction exists in the kernel.
//
// THIS FILE holds the first generation of the forward (32 token rows per wave, v_mfma_f32_32x32x16_bf16), which is what runs at
// C = 192, the entry points, and esvit_ln_fold_finish.  The forward at C = 96 and the backward at both widths run the second
// generation (16 token rows per wave, mlp_fused16.hip: half the registers, 64-byte row pieces): measured on the rows of the
// B = 128 step (profiles/r03_mlp_fused_generations.jsonl) forward 96: 580 -> 553 us (teacher rows 347 -> 304), backward kernel 96:
// 1427 -> 1366, backward kernel 192: 960 -> 893, but forward 192: 432 -> 643 (twice the LDS weight bytes per token and a chunk
// too short to hide its DMA), so that one stays here.
//
// Work split.  A workgroup is 4 waves; a wave OWNS 32 token rows for the whole branch, so nothing but the weight tiles is
// shared between waves.  MFMA shape: v_mfma_f32_32x32x16_bf16.  Every product is formed TRANSPOSED -- hidden / channel index
// on the MFMA rows, the wave's 32 tokens on the MFMA columns:
//     P^T [32 hidden][32 tok] = W1_chunk [32 x C]  * LN(x)^T          (A: weight rows from LDS, B: token fragments in registers)
//     y^T [C][32 tok]        += W2_chunk [C x 32]  * GELU(P)^T        (B: the accumulator registers of the line above)
// In the accumulator layout (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) lane (n, h) holds, for ITS token n,
// MFMA rows {8m + 4h + e}: after bias + GELU + rounding these 16 values ARE two B-operand fragments of the next product, so the
// hidden tile never leaves the registers (the trick the 14x14 attention kernels use for P, window_attn_big.hip).  Which hidden
// unit an MFMA row computes is free -- it is just the W1 row the A fragment reads -- so row i is given hidden unit rho(i) (bits
// 2 and 3 of i swapped): each lane half then owns 8 CONSECUTIVE hidden units per k-step and the second weight is read in its
// natural order, one 16-byte LDS read per fragment.  Because the OUTPUT tiles are transposed as well, a lane ends up with 16 of
// every 32 channels of its own token (its partner lane n + 32 has the other 16): row statistics (the next LayerNorm, the
// LayerNorm backward) are in-lane sums plus one cross-half exchange, and all global accesses are 16-byte vectors.
//
// Weights stream L2 -> LDS by LDS-DMA (buffer_load ... lds) in 32-hidden chunks, NBUF buffers, one workgroup barrier per
// chunk; bank conflicts are removed by XOR-swizzling the 16-byte chunk index on the source address and on the fragment read
// (guide rule 21).  The forward loop contains no vector-memory operation besides the DMA; the backward loop also stores the two
// hidden tiles, and its counted wait accounts for them (vmcnt retires in order).
#include "common.h"
#include "../../include/esvit_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int MLP_WAVES = 4;
constexpr int MLP_ROWS = 32 * MLP_WAVES;  // token rows per workgroup
constexpr int HCH = 32;                   // hidden units per chunk

template <int C>
struct MlpCfg {
    static constexpr int KS1 = C / 16;            // k-steps of a product over the channels
    static constexpr int NT2 = C / 32;            // 32-channel output tiles
    static constexpr int W1_BYTES = HCH * C * 2;  // image A: [32 hidden][C]      (rows of W1, or of W2^T)
    static constexpr int W2_BYTES = C * HCH * 2;  // image B: [C][32 hidden]      (columns of W2, or of W1^T)
    static constexpr int PA = W1_BYTES / 1024;    // 1 KiB DMA pieces per image
    static constexpr int PB = W2_BYTES / 1024;
    static constexpr int M1 = C == 192 ? 7 : 3;   // swizzle mask of image A (chunks per row: 24 = 3 x 8, 12 = 3 x 4)
    // image A: rows of 2C bytes.  One fragment read = 32 rows x 16 bytes at one chunk index: the 384-byte pitch (C = 192)
    // alternates two bank phases -> XOR the chunk with (row >> 1) & 7; the 192-byte pitch (C = 96) cycles four -> (row >> 2) & 3.
    __device__ __forceinline__ static int sw1(int row) { return C == 192 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
    __device__ __forceinline__ static int pos1(int chunk, int row) { return (chunk & ~M1) | ((chunk ^ sw1(row)) & M1); }
    // image B: rows of 64 bytes = 4 chunks; a fragment read touches 32 consecutive rows at one 16-byte slot -> XOR the
    // chunk with (row >> 2) & 3
    __device__ __forceinline__ static int sw2(int row) { return (row >> 2) & 3; }
    // per-lane source byte offset of DMA piece `piece` of an image (chunk 0); chunk q adds a scalar offset
    __device__ __forceinline__ static int voff_a(int piece, int lane) {
        const int p = piece * 64 + lane;              // 16-byte chunk index inside the image
        const int r = p / (C / 8), cp = p % (C / 8);  // image row (hidden unit of the chunk), chunk position in the row
        return (r * C + pos1(cp, r) * 8) * 2;         // (XOR is an involution: image position cp holds source chunk pos1(cp))
    }
    __device__ __forceinline__ static int voff_b(int piece, int lane) {
        const int p = piece * 64 + lane;
        const int r = p / 4, cp = p % 4;              // image row (channel), chunk position (8 hidden units each)
        return (r * 4 * C + (cp ^ sw2(r)) * 8) * 2;
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

// Workgroup barrier of the chunk loops.  NOT __syncthreads(): its fence makes hipcc emit s_waitcnt vmcnt(0) in front of the
// s_barrier, which drains the LDS-DMA of the chunk after next (and, in the backward, the hidden-tile stores to HBM) at every
// chunk -- the loop would run at memory latency.  The data hazards are covered explicitly: the counted vmcnt before the barrier
// (this wave's DMA pieces of the next chunk have landed), lgkmcnt(0) (this wave's LDS reads of the current chunk have returned).
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

// lanes n and n + 32 each hold two 4-channel groups (x: channels 8j' + 4h .., y: channels 8(j'+1) + 4h ..): exchange so that
// the low lane holds channels 8j' .. 8j' + 7 and the high lane 8(j'+1) .. 8(j'+1) + 7 (v_permlane32_swap)
__device__ __forceinline__ void pair_swap(unsigned& x, unsigned& y) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}

// exact erf-GELU and its derivative from ONE exponential: erf's exp(-u^2) with u = v / sqrt(2) is the Gaussian of GELU'
__device__ __forceinline__ void gelu_both(float v, float& g, float& dg) {
    const float av = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * av);
    const float e = __expf(-0.5f * v * v);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erfv = copysignf(1.f - poly * e, v);
    const float cdf = 0.5f * (1.f + erfv);
    g = v * cdf;
    dg = cdf + v * 0.39894228040143268f * e;
}

// ------------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------------
template <int C>
struct FwdCfg : MlpCfg<C> {
    using B = MlpCfg<C>;
    static constexpr int NP = B::PA + B::PB;                  // DMA pieces per chunk
    static constexpr int PPW = NP / MLP_WAVES;                // per wave
    static_assert(NP % MLP_WAVES == 0, "every wave issues the same number of DMA instructions");
    static constexpr int WBUF = B::W1_BYTES + B::W2_BYTES;
    static constexpr int NBUF = 3;                            // chunk q + 2 is requested while chunk q is computed
    static constexpr int LDS_BYTES = NBUF * WBUF;
};

template <int C, bool LNN>
__device__ __forceinline__ void mlp_fused_fwd_body(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const bf16* __restrict__ W1, const float* __restrict__ b1, const bf16* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ rowscale, long M, float* __restrict__ y, const float* __restrict__ gamma_n,
    const float* __restrict__ beta_n, bf16* __restrict__ xw_n, float* __restrict__ mean_n, float* __restrict__ rstd_n) {
    using Cfg = FwdCfg<C>;
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

    // ---- weight chunk DMA ----
    const __amdgpu_buffer_rsrc_t r1 = mk_rsrc(W1, (long)H4 * C * 2), r2 = mk_rsrc(W2, (long)C * H4 * 2);
    int voff[Cfg::PPW];
#pragma unroll
    for (int i = 0; i < Cfg::PPW; ++i) {
        const int piece = wave * Cfg::PPW + i;  // wave-uniform
        voff[i] = piece < Cfg::PA ? Cfg::voff_a(piece, lane) : Cfg::voff_b(piece - Cfg::PA, lane);
    }
    auto issue_chunk = [&](int q, int buf) {
        char* img = smem + buf * Cfg::WBUF;  // W1 image, then W2 image
        const int so1 = q * HCH * C * 2, so2 = q * HCH * 2;
#pragma unroll
        for (int i = 0; i < Cfg::PPW; ++i) {
            const int piece = wave * Cfg::PPW + i;
            if (piece < Cfg::PA) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_void*)(img + piece * 1024), 16, voff[i], so1, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_void*)(img + piece * 1024), 16, voff[i], so2, 0, 0);
        }
    };
    issue_chunk(0, 0);
    issue_chunk(1, 1);

    // ---- LayerNorm of this lane's half row, straight into the B-operand fragments of the first product ----
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

    f32x16 acc2[Cfg::NT2];  // y^T: tile mt, register r <-> channel 32 mt + (r & 3) + 8 (r >> 2) + 4 hh of token n
#pragma unroll
    for (int t = 0; t < Cfg::NT2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;

    wait_vm<0>();     // this wave's parts of chunks 0 and 1 have landed (and its LN loads are done)
    __syncthreads();  // ... everybody else's too

    const int rho_n = (n & ~12) | ((n & 4) << 1) | ((n & 8) >> 1);
    int buf = 0;
    for (int q = 0; q < NCHUNK; ++q) {
        const bool more = q + 2 < NCHUNK;
        if (more) issue_chunk(q + 2, buf == 0 ? 2 : buf - 1);  // the buffer of chunk q - 1, released by the barrier that ended it
        const char* w1 = smem + buf * Cfg::WBUF;
        const char* w2 = w1 + Cfg::W1_BYTES;

        // ---- P^T[32 hidden][32 tokens] = W1_chunk * LN(x)^T (the first step takes a literal zero accumulator: no zero-fill) ----
        f32x16 acc1;
#pragma unroll
        for (int s = 0; s < Cfg::KS1; ++s) {  // A fragment: W1 row rho(n) of the chunk, channels 16s + 8hh .. +7
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(w1 + (rho_n * (C / 8) + Cfg::pos1(2 * s + hh, rho_n)) * 16);
            if (s == 0) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb[s], f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
            else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb[s], acc1, 0, 0, 0);
        }
        // ---- bias, GELU, rounding: register 8t + e (e < 8) holds hidden unit 32q + 16t + 8hh + e of token n ----
        bf16x8 hf[2];
        const float* bq = b1 + q * HCH;  // wave-uniform address: scalar loads (no vector-memory load inside the loop)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float blo = bq[16 * t + e], bhi = bq[16 * t + 8 + e];
                hf[t][e] = (bf16)gelu_f(acc1[8 * t + e] + (hh ? bhi : blo));
            }
        // ---- y^T[C][32 tokens] += W2_chunk[C x 32 hidden] * H^T ----
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int mt = 0; mt < Cfg::NT2; ++mt) {
                const int c = 32 * mt + n;  // A fragment: output channel c, hidden 16t + 8hh .. +7 of the chunk
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(w2 + c * 64 + (((2 * t + hh) ^ Cfg::sw2(c)) * 16));
                acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hf[t], acc2[mt], 0, 0, 0);
            }
        }
        // chunk q + 1 must have landed before the barrier; the DMA of chunk q + 2, issued after it, may stay in flight
        if (more) wait_vm<Cfg::PPW>();
        else wait_vm<0>();
        chunk_barrier();  // chunk q + 1 landed for every wave; every wave is done reading chunk q
        buf = buf == 2 ? 0 : buf + 1;
    }

    // ---- epilogue: y = x + rowscale * (acc2 + b2); this lane: token n, channels 32 mt + 8 j + 4 hh + (0..3) ----
    const float rs = rowscale ? rowscale[rrow] : 1.f;
    float s1 = 0.f;
#pragma unroll
    for (int mt = 0; mt < Cfg::NT2; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c0 = 32 * mt + 8 * j + 4 * hh;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + rrow * C + c0);
            const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + c0);
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] = xv[i] + rs * (acc2[mt][4 * j + i] + bb[i]);
                if constexpr (LNN) {
                    acc2[mt][4 * j + i] = o[i];
                    s1 += o[i];
                }
            }
            if (row_ok) *reinterpret_cast<f32x4*>(y + row * C + c0) = o;
        }
    if constexpr (LNN) {  // LayerNorm(y) with the next block's norm1 parameters, bf16, + its row statistics
        s1 += __shfl_xor(s1, 32, 64);
        const float mean = s1 * (1.f / C);
        float s2 = 0.f;
#pragma unroll
        for (int mt = 0; mt < Cfg::NT2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = acc2[mt][r] - mean;
                s2 += d * d;
            }
        s2 += __shfl_xor(s2, 32, 64);
        const float rstd = rsqrtf(s2 * (1.f / C) + eps);
        if (row_ok && hh == 0) {
            mean_n[row] = mean;
            rstd_n[row] = rstd;
        }
#pragma unroll
        for (int mt = 0; mt < Cfg::NT2; ++mt)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                unsigned w[2][2];  // [j & 1][dword]
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * jp + jj;
                    const int c0 = 32 * mt + 8 * j + 4 * hh;
                    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma_n + c0), b = *reinterpret_cast<const f32x4*>(beta_n + c0);
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = (acc2[mt][4 * j + i] - mean) * rstd * g[i] + b[i];
                    w[jj][0] = pack2(v[0], v[1]);
                    w[jj][1] = pack2(v[2], v[3]);
                }
                pair_swap(w[0][0], w[1][0]);
                pair_swap(w[0][1], w[1][1]);
                // low lane: channels 32 mt + 16 jp + 0..7, high lane: + 8..15
                if (row_ok) *reinterpret_cast<u32x4*>(xw_n + row * C + 32 * mt + 16 * jp + 8 * hh) = u32x4{w[0][0], w[0][1], w[1][0], w[1][1]};
            }
    }
}

template <int C, bool LNN>
__global__ __launch_bounds__(MLP_WAVES * 64, (C == 96 ? 4 : 2)) void mlp_fused_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    const bf16* __restrict__ W1, const float* __restrict__ b1, const bf16* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ rowscale, long M, float* __restrict__ y, const float* __restrict__ gamma_n,
    const float* __restrict__ beta_n, bf16* __restrict__ xw_n, float* __restrict__ mean_n, float* __restrict__ rstd_n) {
    mlp_fused_fwd_body<C, LNN>(x, gamma, beta, eps, W1, b1, W2, b2, rowscale, M, y, gamma_n, beta_n, xw_n, mean_n, rstd_n);
}

template <int C, bool LNN>
int launch_mlp(const float* x, const float* gamma, const float* beta, float eps, const void* W1, const float* b1, const void* W2,
               const float* b2, const float* rowscale, long M, float* y, const float* gamma_n, const float* beta_n, void* xw_n,
               float* mean_n, float* rstd_n, hipStream_t stream) {
    const int grid = ceil_div(M, MLP_ROWS);
    const size_t lds = FwdCfg<C>::LDS_BYTES;
    auto kern = mlp_fused_fwd_kernel<C, LNN>;
    static unsigned long long lds_set = 0;  // one-time raise of the dynamic LDS cap (per device; idempotent)
    esvit_raise_lds(kern, (int)lds, lds_set);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(MLP_WAVES * 64), lds, stream, x, gamma, beta, eps, (const bf16*)W1, b1, (const bf16*)W2, b2,
                       rowscale, M, y, gamma_n, beta_n, (bf16*)xw_n, mean_n, rstd_n);
    ESVIT_CHECK_LAUNCH("esvit_mlp_fused_fwd");
    return ESVIT_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// small helpers of the fused branch
// ------------------------------------------------------------------------------------------------------------------------
// LayerNorm folded out of a weight gradient.  For y = LN(x) W^T + b with LN(x) = xhat o gamma + beta the GEMM was run on
// xhat:  G = dY^T xhat  [J, C],  db = colsum(dY)  [J].  Then
//     dW = G o gamma (per column) + db (x) beta,     dgamma[c] = sum_j W[j, c] G[j, c],     dbeta[c] = sum_j db[j] W[j, c]
// (W: the fp32 master).  One workgroup per 32 columns, 32 x 32 threads; G is overwritten by dW.
__global__ __launch_bounds__(1024) void ln_fold_finish_kernel(float* __restrict__ G, const float* __restrict__ db, const float* __restrict__ W,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, int J, int Cc,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    __shared__ float sg[32][33], sb[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    float ag = 0.f, ab = 0.f;
    if (c < Cc) {
        const float gm = gamma[c], bt = beta[c];
        for (int j = ty; j < J; j += 32) {
            const long o = (long)j * Cc + c;
            const float g = G[o], w = W[o], d = db[j];
            ag += w * g;
            ab += d * w;
            G[o] = g * gm + d * bt;
        }
    }
    sg[ty][tx] = ag;
    sb[ty][tx] = ab;
    __syncthreads();
    if (ty == 0 && c < Cc) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            a += sg[i][tx];
            b += sb[i][tx];
        }
        dgamma[c] = accumulate ? dgamma[c] + a : a;
        dbeta[c] = accumulate ? dbeta[c] + b : b;
    }
}

}  // namespace

// esvit_query(ESVIT_Q_MLP_FUSED): bit 0 = esvit_mlp_fused_fwd exists, bit 1 = esvit_mlp_fused_bwd exists
int esvit_i_mlp_fused_supported(int dtype, int C) {
    if (dtype != ESVIT_BF16) return 0;
    if (C == 384) return 1;  // mlp_fused32p.hip: forward (inference pass; the training pass has its own entry point, esvit_mlp_fused_fwd_train)
    return (C == 96 || C == 128 || C == 192 || C == 256) ? 3 : 0;
}

// third generation (mlp_fused32p.hip): C = 384
int esvit_i_mlp32p_fwd(const float* x, const float* gamma, const float* beta, float eps, const void* W1, const float* b1, const void* W2,
                       const float* b2, const float* rowscale, long M, int C, float* y, void* a1, void* a1g, void* h, float* mean, float* rstd,
                       hipStream_t stream);

// second generation (mlp_fused16.hip)
int esvit_i_mlp16_fwd(const float* x, const float* gamma, const float* beta, float eps, const void* W1p, const float* b1, const void* W2,
                      const float* b2, const float* rowscale, long M, int C, float* y, const float* gn, const float* bn, void* xw, float* mn,
                      float* rn, hipStream_t stream);
int esvit_i_mlp16_bwd(const float* x, const float* gy, const float* rs_mlp, const float* rs_out, const float* gamma, const float* beta, float eps,
                      const void* W1p, const void* W2Tp, const void* W1T, const float* b1, long M, int C, float* gx, void* gxa, void* xhat,
                      void* a1g, void* da1, hipStream_t stream);

#define AL16(p_) (((uintptr_t)(p_) % 16) == 0)

extern "C" int esvit_mlp_fused_fwd(int dtype, const float* x, const float* gamma, const float* beta, float eps, const void* W1,
                                   const float* b1, const void* W2, const float* b2, const float* rowscale, int64_t M, int C,
                                   float* y, const float* gamma_next, const float* beta_next, void* xw_next, float* mean_next,
                                   float* rstd_next, esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(esvit_i_mlp_fused_supported(dtype, C) & 1, "esvit_mlp_fused_fwd: bf16 activations and C in {96, 128, 192, 256, 384} only (C=%d)", C);
    ESVIT_CHECK_ARG(x && gamma && beta && W1 && b1 && W2 && b2 && y && M > 0, "esvit_mlp_fused_fwd: null pointer / empty input");
    ESVIT_CHECK_ARG(AL16(x) && AL16(y) && AL16(W1) && AL16(W2) && AL16(gamma) && AL16(beta) && AL16(b1) && AL16(b2),
                    "esvit_mlp_fused_fwd: operands must be 16-byte aligned");
    const bool lnn = gamma_next != nullptr;
    if (lnn)
        ESVIT_CHECK_ARG(beta_next && xw_next && mean_next && rstd_next && AL16(gamma_next) && AL16(beta_next) && AL16(xw_next),
                        "esvit_mlp_fused_fwd: the next-LayerNorm outputs come together (gamma, beta, xw, mean, rstd; 16-byte aligned)");
    if (C == 384) {
        ESVIT_CHECK_ARG(!lnn, "esvit_mlp_fused_fwd: C = 384 does not emit the next LayerNorm");
        return esvit_i_mlp32p_fwd(x, gamma, beta, eps, W1, b1, W2, b2, rowscale, M, C, y, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
    }
    if (C == 96 || C == 128 || C == 256)  // 16 tokens per wave; W1 in the permuted channel order (ESVIT_MLP_W1_FWD of esvit_mlp_fused_weight)
        return esvit_i_mlp16_fwd(x, gamma, beta, eps, W1, b1, W2, b2, rowscale, M, C, y, gamma_next, beta_next, xw_next, mean_next, rstd_next, stream);
    return lnn ? launch_mlp<192, true>(x, gamma, beta, eps, W1, b1, W2, b2, rowscale, M, y, gamma_next, beta_next, xw_next, mean_next, rstd_next, stream)
               : launch_mlp<192, false>(x, gamma, beta, eps, W1, b1, W2, b2, rowscale, M, y, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int esvit_mlp_fused_fwd_train(int dtype, const float* x, const float* gamma, const float* beta, float eps, const void* W1,
                                         const float* b1, const void* W2, const float* b2, const float* rowscale, int64_t M, int C, float* y,
                                         void* a1, void* a1g, void* h, float* mean, float* rstd, esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(dtype == ESVIT_BF16 && C == 384, "esvit_mlp_fused_fwd_train: bf16 activations and C = 384 only (C=%d)", C);
    ESVIT_CHECK_ARG(x && gamma && beta && W1 && b1 && W2 && b2 && y && a1 && a1g && h && mean && rstd && M > 0, "esvit_mlp_fused_fwd_train: null pointer / empty input");
    ESVIT_CHECK_ARG(AL16(x) && AL16(y) && AL16(W1) && AL16(W2) && AL16(gamma) && AL16(beta) && AL16(b1) && AL16(b2) && AL16(a1) && AL16(a1g) && AL16(h),
                    "esvit_mlp_fused_fwd_train: operands must be 16-byte aligned");
    return esvit_i_mlp32p_fwd(x, gamma, beta, eps, W1, b1, W2, b2, rowscale, M, C, y, a1, a1g, h, mean, rstd, stream);
}

extern "C" int esvit_mlp_fused_bwd(int dtype, const float* x, const float* gy, const float* rowscale_mlp, const float* rowscale_out,
                                   const float* gamma, const float* beta, float eps, const void* W1, const void* W2T, const void* W1T,
                                   const float* b1, int64_t M, int C, float* gx, void* gx_act, void* xhat, void* a1g, void* da1,
                                   esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(esvit_i_mlp_fused_supported(dtype, C) & 2, "esvit_mlp_fused_bwd: bf16 activations and C in {96, 128, 192, 256} only (C=%d)", C);
    ESVIT_CHECK_ARG(x && gy && gamma && beta && W1 && W2T && W1T && b1 && gx && gx_act && xhat && a1g && da1 && M > 0,
                    "esvit_mlp_fused_bwd: null pointer / empty input");
    ESVIT_CHECK_ARG(AL16(x) && AL16(gy) && AL16(gx) && AL16(gx_act) && AL16(xhat) && AL16(a1g) && AL16(da1) && AL16(W1) && AL16(W2T) &&
                        AL16(W1T) && AL16(gamma) && AL16(beta) && AL16(b1),
                    "esvit_mlp_fused_bwd: operands must be 16-byte aligned");
    return esvit_i_mlp16_bwd(x, gy, rowscale_mlp, rowscale_out, gamma, beta, eps, W1, W2T, W1T, b1, M, C, gx, gx_act, xhat, a1g, da1, stream);
}

extern "C" int esvit_ln_fold_finish(float* G, const float* db, const float* W, const float* gamma, const float* beta, int J, int C,
                                    float* dgamma, float* dbeta, int accumulate, esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(G && db && W && gamma && beta && dgamma && dbeta && J > 0 && C > 0, "esvit_ln_fold_finish: bad arguments");
    hipLaunchKernelGGL(ln_fold_finish_kernel, dim3(ceil_div(C, 32)), dim3(1024), 0, stream, G, db, W, gamma, beta, J, C, dgamma, dbeta, accumulate);
    ESVIT_CHECK_LAUNCH("esvit_ln_fold_finish");
    return ESVIT_OK;
}
