// Fused Swin MLP branch, second generation: 16 token rows per wave on v_mfma_f32_16x16x32_bf16 (see mlp_fused.hip for the
// branch itself, the transposed-product idea and the LayerNorm fold; this file only changes the work split).
//
// Why.  The first generation gave a wave 32 tokens (32x32x16 MFMA).  Its forward spent half of its time in the prologue /
// epilogue -- a lane there reads 16-byte pieces of ITS token row, 32 bytes of a row per instruction, so every 128-byte line is
// requested four times -- and its backward held 194 / 394 registers (two waves / ONE wave per SIMD at C = 96 / 192), far too few
// waves to overlap the I/O phases of one workgroup with the hidden-chunk loops of another (profiles/r03_mlp_fused_ablation.txt).
// With 16 tokens per wave a lane (c = token, g = lane >> 4) carries a quarter of the 32-deep k-step, so
//   * the four lanes of a token cover 64 contiguous bytes per global instruction (rows of fp32 x / dy, of the bf16 hidden tiles,
//     of every output), twice the piece size and half the instructions;
//   * the register footprint halves: ~80 (forward) / ~115-165 (backward) registers -> 6 / 4-3 waves per SIMD;
//   * the price is weight traffic from LDS: a 16-token wave reads the same 1 KiB weight fragments as a 32-token wave did, i.e.
//     twice the LDS bytes per token (still below the LDS port: the loop is VALU-bound on GELU).
//
// Fragment conventions (mfma.h): A [16 x 32]: lane (c, g) holds row c, k-slots 8g .. 8g+7; B [32 x 16]: lane (c, g) holds column
// c, k-slots 8g .. 8g+7; D [16 x 16]: lane (c, g) holds D[4g + r][c], r < 4.  The wave's tokens are the MFMA COLUMNS (c) of every
// product (all products are formed transposed):
//   P^T [32 hidden][16 tok] = W1_chunk * LN(x)^T     two 16-row tiles; MFMA row i of tile t is hidden unit 8 (i >> 2) + 4 t + (i & 3)
//                                                    of the chunk, so lane (c, g) ends up with hidden units 8g .. 8g+7 of its token:
//                                                    exactly the k-slots of a B fragment of the next product
//   y^T [C][16 tok]        += W2_chunk * GELU(P)^T   C / 16 row tiles, one k-step (the chunk's 32 hidden units)
// The k-slot -> channel assignment of the FIRST product is free as long as both operands agree: slot 8g + e of k-step s is
// channel 32s + 4g + e (e < 4) or 32s + 16 + 4g + (e - 4), so a lane's two fp32 loads per k-step are 16 bytes each and the four
// lanes of a token read 64 contiguous bytes; the weight copies the kernels stream carry the same permutation in their columns
// (esvit_cast_weight, perm32).  Outputs leave as 16-byte vectors: fp32 tiles as they are (channels 16t + 4g .. +3), bf16 tiles
// after a v_permlane16_swap between lanes g and g ^ 1 (8 consecutive channels per lane).
//
// Weights stream L2 -> LDS by LDS-DMA in 32-hidden chunks (plus one 1 KiB piece carrying the chunk's 32 fc1 biases), NBUF
// buffers, one raw workgroup barrier per chunk, counted vmcnt waits; swizzles verified conflict-free against the ds_read_b128
// lane grouping of MI355X_MICROARCH.md (image A: per-C XOR of the 16-byte unit index, image B: unit ^ ((row >> 1) & 3)).
#include "common.h"
#include "fused16.h"
#include "../../include/esvit_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------------
template <int C, int NW, bool LNN>
__device__ __forceinline__ void fwd16_body(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                           const bf16* __restrict__ W1p, const float* __restrict__ b1, const bf16* __restrict__ W2,
                                           const float* __restrict__ b2, const float* __restrict__ rowscale, long M, float* __restrict__ y,
                                           const float* __restrict__ gamma_n, const float* __restrict__ beta_n, bf16* __restrict__ xw_n,
                                           float* __restrict__ mean_n, float* __restrict__ rstd_n) {
    using Cf = Cfg16<C>;
    constexpr int H4 = 4 * C, NCHUNK = H4 / HCH;
    constexpr int NP = Cf::PA + Cf::PB + 1;            // DMA pieces per chunk: W1 rows | W2 columns | biases
    constexpr int PPW = (NP + NW - 1) / NW;
    constexpr int WBUF = Cf::A_BYTES + Cf::B_BYTES + 1024;
    constexpr int NBUF = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const long row = ((long)blockIdx.x * NW + wave) * 16 + c;
    const bool ok = row < M;
    const long rrow = ok ? row : (M - 1);

    const __amdgpu_buffer_rsrc_t r1 = mk_rsrc(W1p, (long)H4 * C * 2), r2 = mk_rsrc(W2, (long)C * H4 * 2), r3 = mk_rsrc(b1, (long)H4 * 4);
    int voff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int piece = wave + NW * i;  // wave-uniform
        voff[i] = piece < Cf::PA ? Cf::voff_a(piece, lane) : (piece < Cf::PA + Cf::PB ? Cf::voff_b(piece - Cf::PA, lane) : (lane & 7) * 16);
    }
    const bool full = ((NP - wave + NW - 1) / NW) == PPW;  // this wave issues PPW (not PPW - 1) pieces per chunk
    auto issue_chunk = [&](int q, int buf) {
        char* img = smem + buf * WBUF;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wave + NW * i;
            if (piece < Cf::PA) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_void*)(img + piece * 1024), 16, voff[i], q * HCH * C * 2, 0, 0);
            else if (piece < Cf::PA + Cf::PB) __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_void*)(img + piece * 1024), 16, voff[i], q * HCH * 2, 0, 0);
            else if (piece < NP) __builtin_amdgcn_raw_ptr_buffer_load_lds(r3, (lds_void*)(img + piece * 1024), 16, voff[i], q * HCH * 4, 0, 0);
        }
    };
    issue_chunk(0, 0);
    issue_chunk(1, 1);

    // ---- LayerNorm of this lane's quarter row, straight into the B fragments of the first product ----
    bf16x8 xb[Cf::KS];
    {
        float xv[2 * Cf::KS * 4];
        const float* xr = x + rrow * C + 4 * g;
#pragma unroll
        for (int s = 0; s < Cf::KS; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 32 * s), b = *reinterpret_cast<const f32x4*>(xr + 32 * s + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[8 * s + e] = a[e];
                xv[8 * s + 4 + e] = b[e];
            }
        }
        float mean, rstd;
        row_stats(xv, 1.f / C, eps, mean, rstd);
#pragma unroll
        for (int s = 0; s < Cf::KS; ++s) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + 32 * s + 4 * g), g1 = *reinterpret_cast<const f32x4*>(gamma + 32 * s + 16 + 4 * g);
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(beta + 32 * s + 4 * g), c1 = *reinterpret_cast<const f32x4*>(beta + 32 * s + 16 + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xb[s][e] = (bf16)((xv[8 * s + e] - mean) * rstd * g0[e] + c0[e]);
                xb[s][4 + e] = (bf16)((xv[8 * s + 4 + e] - mean) * rstd * g1[e] + c1[e]);
            }
        }
    }

    f32x4 acc2[Cf::MT];  // y^T: tile mt, element r <-> channel 16 mt + 4g + r of token c
#pragma unroll
    for (int t = 0; t < Cf::MT; ++t) acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    wait_vm<0>();
    __syncthreads();

    int buf = 0;
    for (int q = 0; q < NCHUNK; ++q) {
        const bool more = q + 2 < NCHUNK;
        if (more) issue_chunk(q + 2, buf == 0 ? 2 : buf - 1);
        const char* wa = smem + buf * WBUF;
        const char* wb = wa + Cf::A_BYTES;
        const float* sb = reinterpret_cast<const float*>(wb + Cf::B_BYTES);

        f32x4 p0, p1;
#pragma unroll
        for (int s = 0; s < Cf::KS; ++s) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(wa + Cf::frag_a(0, s, c, g));
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(wa + Cf::frag_a(1, s, c, g));
            if (s == 0) {
                p0 = mfma16(a0, xb[s], f32x4{0.f, 0.f, 0.f, 0.f});
                p1 = mfma16(a1, xb[s], f32x4{0.f, 0.f, 0.f, 0.f});
            } else {
                p0 = mfma16(a0, xb[s], p0);
                p1 = mfma16(a1, xb[s], p1);
            }
        }
        // bias + GELU: this lane holds hidden units 8g .. 8g+7 of token c
        const f32x4 bl = *reinterpret_cast<const f32x4*>(sb + 8 * g), bh = *reinterpret_cast<const f32x4*>(sb + 8 * g + 4);
        bf16x8 hf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hf[r] = (bf16)gelu_f(p0[r] + bl[r]);
            hf[4 + r] = (bf16)gelu_f(p1[r] + bh[r]);
        }
#pragma unroll
        for (int mt = 0; mt < Cf::MT; ++mt) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(wb + Cf::frag_b(mt, c, g));
            acc2[mt] = mfma16(a, hf, acc2[mt]);
        }
        if (more) {
            if (full) wait_vm<PPW>();
            else wait_vm<(PPW > 1 ? PPW - 1 : 0)>();
        } else {
            wait_vm<0>();
        }
        chunk_barrier();
        buf = buf == 2 ? 0 : buf + 1;
    }

    // ---- epilogue: y = x + rowscale * (acc2 + b2) ----
    const float rs = rowscale ? rowscale[rrow] : 1.f;
    float yv[Cf::MT * 4];
#pragma unroll
    for (int mt = 0; mt < Cf::MT; ++mt) {
        const int c0 = 16 * mt + 4 * g;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + rrow * C + c0);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + c0);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[r] = xv[r] + rs * (acc2[mt][r] + bb[r]);
            yv[4 * mt + r] = o[r];
        }
        if (ok) *reinterpret_cast<f32x4*>(y + row * C + c0) = o;
    }
    if constexpr (LNN) {
        float mean, rstd;
        row_stats(yv, 1.f / C, eps, mean, rstd);
        if (ok && g == 0) {
            mean_n[row] = mean;
            rstd_n[row] = rstd;
        }
        float w[Cf::MT][4];
#pragma unroll
        for (int mt = 0; mt < Cf::MT; ++mt) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma_n + 16 * mt + 4 * g), bt = *reinterpret_cast<const f32x4*>(beta_n + 16 * mt + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) w[mt][r] = (yv[4 * mt + r] - mean) * rstd * gm[r] + bt[r];
        }
        store_bf16_tiles<Cf::MT>(xw_n + row * C, w, g, ok);
    }
}

template <int C, int NW, bool LNN, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void fwd16_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, const bf16* __restrict__ W1p, const float* __restrict__ b1,
                                                            const bf16* __restrict__ W2, const float* __restrict__ b2,
                                                            const float* __restrict__ rowscale, long M, float* __restrict__ y,
                                                            const float* __restrict__ gamma_n, const float* __restrict__ beta_n,
                                                            bf16* __restrict__ xw_n, float* __restrict__ mean_n, float* __restrict__ rstd_n) {
    fwd16_body<C, NW, LNN>(x, gamma, beta, eps, W1p, b1, W2, b2, rowscale, M, y, gamma_n, beta_n, xw_n, mean_n, rstd_n);
}

// ------------------------------------------------------------------------------------------------------------------------
// backward (data-gradient path)
// ------------------------------------------------------------------------------------------------------------------------
template <int C, int NW, int NBUF>
__device__ __forceinline__ void bwd16_body(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ rs_mlp,
                                           const float* __restrict__ rs_out, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                           const bf16* __restrict__ W1p, const bf16* __restrict__ W2Tp, const bf16* __restrict__ W1T,
                                           const float* __restrict__ b1, long M, float* __restrict__ gx, bf16* __restrict__ gxa,
                                           bf16* __restrict__ xhat, bf16* __restrict__ a1g, bf16* __restrict__ da1) {
    using Cf = Cfg16<C>;
    constexpr int H4 = 4 * C, NCHUNK = H4 / HCH;
    constexpr int NP = 2 * Cf::PA + Cf::PB + 1;  // W1 rows | W2^T rows | W1^T columns | biases
    constexpr int PPW = (NP + NW - 1) / NW;
    constexpr int WBUF = 2 * Cf::A_BYTES + Cf::B_BYTES + 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const long row0 = ((long)blockIdx.x * NW + wave) * 16;
    const long row = row0 + c;
    const bool ok = row < M;
    const long rrow = ok ? row : (M - 1);
    const bool wave_live = row0 < M;  // (a wave without a valid row issues no stores)

    const __amdgpu_buffer_rsrc_t r1 = mk_rsrc(W1p, (long)H4 * C * 2), r2 = mk_rsrc(W2Tp, (long)H4 * C * 2), r3 = mk_rsrc(W1T, (long)C * H4 * 2),
                                 r4 = mk_rsrc(b1, (long)H4 * 4);
    int voff[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int piece = wave + NW * i;
        voff[i] = piece < 2 * Cf::PA ? Cf::voff_a(piece < Cf::PA ? piece : piece - Cf::PA, lane)
                                     : (piece < 2 * Cf::PA + Cf::PB ? Cf::voff_b(piece - 2 * Cf::PA, lane) : (lane & 7) * 16);
    }
    const bool full = ((NP - wave + NW - 1) / NW) == PPW;
    auto issue_chunk = [&](int q, int buf) {
        char* img = smem + buf * WBUF;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wave + NW * i;
            if (piece < Cf::PA) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_void*)(img + piece * 1024), 16, voff[i], q * HCH * C * 2, 0, 0);
            else if (piece < 2 * Cf::PA) __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_void*)(img + piece * 1024), 16, voff[i], q * HCH * C * 2, 0, 0);
            else if (piece < 2 * Cf::PA + Cf::PB) __builtin_amdgcn_raw_ptr_buffer_load_lds(r3, (lds_void*)(img + piece * 1024), 16, voff[i], q * HCH * 2, 0, 0);
            else if (piece < NP) __builtin_amdgcn_raw_ptr_buffer_load_lds(r4, (lds_void*)(img + piece * 1024), 16, voff[i], q * HCH * 4, 0, 0);
        }
    };
    issue_chunk(0, 0);
    if constexpr (NBUF == 3) issue_chunk(1, 1);

    // ---- prologue: both inputs are requested first (the xhat stores then retire last: nothing waits behind them) ----
    bf16x8 xb[Cf::KS], dyb[Cf::KS];
    float mean, rstd;
    {
        float xv[2 * Cf::KS * 4], gv[2 * Cf::KS * 4];
        const float* xr = x + rrow * C + 4 * g;
        const float* gr = gy + rrow * C + 4 * g;
#pragma unroll
        for (int s = 0; s < Cf::KS; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 32 * s), b = *reinterpret_cast<const f32x4*>(xr + 32 * s + 16);
            const f32x4 p = *reinterpret_cast<const f32x4*>(gr + 32 * s), q = *reinterpret_cast<const f32x4*>(gr + 32 * s + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[8 * s + e] = a[e];
                xv[8 * s + 4 + e] = b[e];
                gv[8 * s + e] = p[e];
                gv[8 * s + 4 + e] = q[e];
            }
        }
        row_stats(xv, 1.f / C, eps, mean, rstd);
        const float sm = rs_mlp ? rs_mlp[rrow] : 1.f;
#pragma unroll
        for (int s = 0; s < Cf::KS; ++s) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + 32 * s + 4 * g), g1 = *reinterpret_cast<const f32x4*>(gamma + 32 * s + 16 + 4 * g);
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(beta + 32 * s + 4 * g), c1 = *reinterpret_cast<const f32x4*>(beta + 32 * s + 16 + 4 * g);
            float h[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (xv[8 * s + e] - mean) * rstd;
                h[4 + e] = (xv[8 * s + 4 + e] - mean) * rstd;
                xb[s][e] = (bf16)(h[e] * g0[e] + c0[e]);
                xb[s][4 + e] = (bf16)(h[4 + e] * g1[e] + c1[e]);
                dyb[s][e] = (bf16)(sm * gv[8 * s + e]);
                dyb[s][4 + e] = (bf16)(sm * gv[8 * s + 4 + e]);
            }
            // xhat out: channels 32s + 4g .. +3 and 32s + 16 + 4g .. +3 -> 8 consecutive channels per lane after the row swap
            unsigned x0 = pack2(h[0], h[1]), x1 = pack2(h[2], h[3]), y0 = pack2(h[4], h[5]), y1 = pack2(h[6], h[7]);
            row_swap(x0, y0);
            row_swap(x1, y1);
            const int ch = 32 * s + 16 * (g & 1) + 4 * (g & ~1);
            if (ok) *reinterpret_cast<u32x4*>(xhat + row * C + ch) = u32x4{x0, x1, y0, y1};
        }
    }

    f32x4 acc3[Cf::MT];  // dH^T
#pragma unroll
    for (int t = 0; t < Cf::MT; ++t) acc3[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    wait_vm<0>();
    __syncthreads();

    bf16* a1g_row = a1g + row * H4 + 8 * g;
    bf16* da1_row = da1 + row * H4 + 8 * g;
    int buf = 0;
    for (int q = 0; q < NCHUNK; ++q) {
        const bool more = q + NBUF - 1 < NCHUNK;
        if (more) issue_chunk(q + NBUF - 1, NBUF == 2 ? (buf ^ 1) : (buf == 0 ? 2 : buf - 1));
        const char* wa = smem + buf * WBUF;       // W1 rows of the chunk (permuted columns)
        const char* wt = wa + Cf::A_BYTES;        // W2^T rows of the chunk (permuted columns)
        const char* wc = wt + Cf::A_BYTES;        // W1^T columns of the chunk [C][32 hidden]
        const float* sb = reinterpret_cast<const float*>(wc + Cf::B_BYTES);

        f32x4 p0, p1, q0, q1;
#pragma unroll
        for (int s = 0; s < Cf::KS; ++s) {
            const int o0 = Cf::frag_a(0, s, c, g), o1 = Cf::frag_a(1, s, c, g);
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(wa + o0), a1 = *reinterpret_cast<const bf16x8*>(wa + o1);
            const bf16x8 t0 = *reinterpret_cast<const bf16x8*>(wt + o0), t1 = *reinterpret_cast<const bf16x8*>(wt + o1);
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            p0 = mfma16(a0, xb[s], s == 0 ? z : p0);
            p1 = mfma16(a1, xb[s], s == 0 ? z : p1);
            q0 = mfma16(t0, dyb[s], s == 0 ? z : q0);
            q1 = mfma16(t1, dyb[s], s == 0 ? z : q1);
        }
        const f32x4 bl = *reinterpret_cast<const f32x4*>(sb + 8 * g), bh = *reinterpret_cast<const f32x4*>(sb + 8 * g + 4);
        bf16x8 hf, df;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float gl, dg;
            gelu_both(p0[r] + bl[r], gl, dg);
            hf[r] = (bf16)gl;
            df[r] = (bf16)(q0[r] * dg);
            gelu_both(p1[r] + bh[r], gl, dg);
            hf[4 + r] = (bf16)gl;
            df[4 + r] = (bf16)(q1[r] * dg);
        }
        if (ok) {  // hidden units 32q + 8g .. +7 of token c: the four lanes of a token write 64 contiguous bytes
            *reinterpret_cast<bf16x8*>(a1g_row + q * HCH) = hf;
            *reinterpret_cast<bf16x8*>(da1_row + q * HCH) = df;
        }
#pragma unroll
        for (int mt = 0; mt < Cf::MT; ++mt) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(wc + Cf::frag_b(mt, c, g));
            acc3[mt] = mfma16(a, df, acc3[mt]);
        }
        // chunk q + 1 must have landed.  Only the LOADS this wave issued after that chunk's DMA are allowed to stay in flight (the DMA pieces of
        // chunk q + 2): loads retire in order among themselves, but stores retire out of order with respect to loads, so an allowance that
        // counted the side-output stores as well (rounds 4-5) could be met with chunk q + 1 still on its way (profiles/r06_gemm_astat_probe.txt:
        // seen in a probe kernel; never here, the chunk has a whole iteration to land).  Same step time (47.36 vs 47.37 ms, same box).
        if constexpr (NBUF == 2) {
            wait_vm<0>();
        } else {
            if (!wave_live) wait_vm<0>();
            else if (more) {
                if (full) wait_vm<PPW>();
                else wait_vm<PPW - 1>();
            } else wait_vm<0>();
        }
        chunk_barrier();
        if constexpr (NBUF == 2) buf ^= 1;
        else buf = buf == 2 ? 0 : buf + 1;
    }

    // ---- epilogue: LayerNorm backward.  This lane: token c, channels 16 mt + 4g + r ----
    float xh[Cf::MT][4], gd[Cf::MT][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < Cf::MT; ++mt) {
        const int c0 = 16 * mt + 4 * g;
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + rrow * C + c0);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xh[mt][r] = (xv[r] - mean) * rstd;
            gd[mt][r] = acc3[mt][r] * gm[r];
            s1 += gd[mt][r];
            s2 += gd[mt][r] * xh[mt][r];
        }
    }
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float m1 = s1 * (1.f / C), m2 = s2 * (1.f / C);
    const float ro = rs_out ? rs_out[rrow] : 1.f;
#pragma unroll
    for (int mt = 0; mt < Cf::MT; ++mt) {
        const int c0 = 16 * mt + 4 * g;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gy + rrow * C + c0);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[r] = gv[r] + rstd * (gd[mt][r] - m1 - xh[mt][r] * m2);
            gd[mt][r] = ro * o[r];
        }
        if (ok) *reinterpret_cast<f32x4*>(gx + row * C + c0) = o;
    }
    store_bf16_tiles<Cf::MT>(gxa + row * C, gd, g, ok);
}

template <int C, int NW, int NBUF, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void bwd16_kernel(const float* __restrict__ x, const float* __restrict__ gy, const float* __restrict__ rs_mlp,
                                                            const float* __restrict__ rs_out, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, const bf16* __restrict__ W1p,
                                                            const bf16* __restrict__ W2Tp, const bf16* __restrict__ W1T,
                                                            const float* __restrict__ b1, long M, float* __restrict__ gx, bf16* __restrict__ gxa,
                                                            bf16* __restrict__ xhat, bf16* __restrict__ a1g, bf16* __restrict__ da1) {
    bwd16_body<C, NW, NBUF>(x, gy, rs_mlp, rs_out, gamma, beta, eps, W1p, W2Tp, W1T, b1, M, gx, gxa, xhat, a1g, da1);
}

// dst (bf16) = cast of src (fp32 [R, S]), optionally transposed ([S, R]) and / or with the 32-block column permutation the 16-token
// kernels use for their channel operand: position 32b + 8g + e of a row holds column 32b + 4g + e (e < 4) / 32b + 16 + 4g + e - 4
__global__ __launch_bounds__(256) void cast_weight_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int R, int S, int transpose, int perm32) {
    const int D1 = transpose ? R : S;  // row length of dst
    const long total = (long)R * S;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int p = (int)(i % D1);
        const long r = i / D1;
        int ch = p;
        if (perm32) {
            const int b = p >> 5, gg = (p >> 3) & 3, e = p & 7;
            ch = 32 * b + (e < 4 ? 4 * gg + e : 16 + 4 * gg + (e - 4));
        }
        dst[i] = (bf16)(transpose ? src[(long)ch * S + r] : src[r * S + ch]);
    }
}

}  // namespace

#define AL16(p_) (((uintptr_t)(p_) % 16) == 0)

// geometry per width: waves per workgroup, waves per SIMD asked of the register allocator
//   forward : C = 96: 8 waves, 6 per SIMD (3 workgroups x 39 KiB);  C = 192: 6 waves, 3 per SIMD (2 x 75 KiB)
//   backward: C = 96: 6 waves, 3 per SIMD, three buffers (2 x 57 KiB);  C = 192: 4 waves, 2 per SIMD, two buffers (2 x 74 KiB)
int esvit_i_mlp16_fwd(const float* x, const float* gamma, const float* beta, float eps, const void* W1p, const float* b1, const void* W2,
                      const float* b2, const float* rowscale, long M, int C, float* y, const float* gn, const float* bn, void* xw, float* mn,
                      float* rn, hipStream_t stream) {
    const bool lnn = gn != nullptr;
#define LAUNCH_F(C_, NW_, WPS_)                                                                                                          \
    do {                                                                                                                                 \
        constexpr size_t lds = 3 * (Cfg16<C_>::A_BYTES + Cfg16<C_>::B_BYTES + 1024);                                                     \
        const int grid = ceil_div(M, 16 * NW_);                                                                                          \
        if (lnn) {                                                                                                                       \
            auto k = fwd16_kernel<C_, NW_, true, WPS_>;                                                                                  \
            static unsigned long long lds_set = 0;                                                                                       \
            esvit_raise_lds(k, (int)lds, lds_set);                                                                                       \
            hipLaunchKernelGGL(k, dim3(grid), dim3(NW_ * 64), lds, stream, x, gamma, beta, eps, (const bf16*)W1p, b1, (const bf16*)W2,   \
                               b2, rowscale, M, y, gn, bn, (bf16*)xw, mn, rn);                                                           \
        } else {                                                                                                                         \
            auto k = fwd16_kernel<C_, NW_, false, WPS_>;                                                                                 \
            static unsigned long long lds_set = 0;                                                                                       \
            esvit_raise_lds(k, (int)lds, lds_set);                                                                                       \
            hipLaunchKernelGGL(k, dim3(grid), dim3(NW_ * 64), lds, stream, x, gamma, beta, eps, (const bf16*)W1p, b1, (const bf16*)W2,   \
                               b2, rowscale, M, y, nullptr, nullptr, nullptr, nullptr, nullptr);                                         \
        }                                                                                                                                \
    } while (0)
    // (C = 384 was measured as well -- 8 waves, 2 per SIMD, 147 KiB of weight buffers: 385 us against 468 unfused on the student's
    // stage-2 rows, 258 against 245 on the teacher's; without a backward of that width it has no user and is not instantiated)
    // (C = 128 / 256: the narrow stages of Swin-B, round 6 -- the general swizzle of fused16.h covers every C whose rows are whole 256-byte bank rows)
    if (C == 96) LAUNCH_F(96, 8, 6);
    else if (C == 128) LAUNCH_F(128, 8, 4);
    else if (C == 256) LAUNCH_F(256, 8, 2);
    else LAUNCH_F(192, 6, 3);
#undef LAUNCH_F
    ESVIT_CHECK_LAUNCH("esvit_mlp_fused_fwd(16)");
    return ESVIT_OK;
}

int esvit_i_mlp16_bwd(const float* x, const float* gy, const float* rs_mlp, const float* rs_out, const float* gamma, const float* beta, float eps,
                      const void* W1p, const void* W2Tp, const void* W1T, const float* b1, long M, int C, float* gx, void* gxa, void* xhat,
                      void* a1g, void* da1, hipStream_t stream) {
#define LAUNCH_B(C_, NW_, NB_, WPS_)                                                                                                     \
    do {                                                                                                                                 \
        constexpr size_t lds = NB_ * (2 * Cfg16<C_>::A_BYTES + Cfg16<C_>::B_BYTES + 1024);                                               \
        const int grid = ceil_div(M, 16 * NW_);                                                                                          \
        auto k = bwd16_kernel<C_, NW_, NB_, WPS_>;                                                                                       \
        static unsigned long long lds_set = 0;                                                                                           \
        esvit_raise_lds(k, (int)lds, lds_set);                                                                                           \
        hipLaunchKernelGGL(k, dim3(grid), dim3(NW_ * 64), lds, stream, x, gy, rs_mlp, rs_out, gamma, beta, eps, (const bf16*)W1p,        \
                           (const bf16*)W2Tp, (const bf16*)W1T, b1, M, gx, (bf16*)gxa, (bf16*)xhat, (bf16*)a1g, (bf16*)da1);             \
    } while (0)
    if (C == 96) LAUNCH_B(96, 6, 3, 3);
    else if (C == 128) LAUNCH_B(128, 6, 3, 3);
    else if (C == 256) LAUNCH_B(256, 8, 2, 2);
    else LAUNCH_B(192, 4, 2, 2);
#undef LAUNCH_B
    ESVIT_CHECK_LAUNCH("esvit_mlp_fused_bwd(16)");
    return ESVIT_OK;
}

extern "C" int esvit_mlp_fused_weight(int kind, const float* src, void* dst_bf16, int C, esvit_stream_t s_) {
    // the four weight copies of the fused branch; which generation consumes a copy decides its channel order (mlp_fused.hip entry points)
    ESVIT_CHECK_ARG(C == 96 || C == 128 || C == 192 || C == 256 || (C == 384 && kind == ESVIT_MLP_W1_FWD),
                    "esvit_mlp_fused_weight: C in {96, 128, 192, 256} (and the forward's copy at 384) only (C=%d)", C);
    switch (kind) {
        case ESVIT_MLP_W1_FWD: return esvit_cast_weight(src, dst_bf16, 4 * C, C, 0, C == 96 || C == 128 || C == 256, s_);   // fc1.weight [4C, C] for the forward (16-token kernels: permuted; 192 / 384: the plain cast)
        case ESVIT_MLP_W1_BWD: return esvit_cast_weight(src, dst_bf16, 4 * C, C, 0, 1, s_);          // ... for the backward
        case ESVIT_MLP_W1T_BWD: return esvit_cast_weight(src, dst_bf16, 4 * C, C, 1, 0, s_);         // fc1.weight^T [C, 4C]
        case ESVIT_MLP_W2T_BWD: return esvit_cast_weight(src, dst_bf16, C, 4 * C, 1, 1, s_);         // fc2.weight^T [4C, C] from fc2.weight [C, 4C]
        default: ESVIT_CHECK_ARG(false, "esvit_mlp_fused_weight: bad kind %d", kind);
    }
    return ESVIT_OK;
}

extern "C" int esvit_cast_weight(const float* src, void* dst_bf16, int R, int S, int transpose, int perm32, esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(src && dst_bf16 && R > 0 && S > 0, "esvit_cast_weight: bad arguments");
    ESVIT_CHECK_ARG(!perm32 || ((transpose ? R : S) % 32 == 0), "esvit_cast_weight: the permuted dimension must be a multiple of 32");
    const long total = (long)R * S;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(cast_weight_kernel, dim3(grid), dim3(256), 0, stream, src, (bf16*)dst_bf16, R, S, transpose, perm32);
    ESVIT_CHECK_LAUNCH("esvit_cast_weight");
    return ESVIT_OK;
}
