// Fused Swin MLP branch for the WIDE stage (C = 384: stage 2 of Swin-T / Swin-S, six or eighteen of the blocks), third generation:
//
//     y = x + rowscale * ( GELU( LN(x) W1^T + b1 ) W2^T + b2 )          x, y: fp32 [M, C];  W1: [4C, C];  W2: [C, 4C]     (swin_transformer.py:31-37, 331)
//
// one wave per SIMD, 32 token rows per wave, v_mfma_f32_32x32x16_bf16, the two products SOFTWARE-PIPELINED against each other and
// against the GELU between them.  Why a third structure (mlp_fused.hip / mlp_fused16.hip serve C = 96 / 192):
//   * at C = 384 a wave that owns its tokens for the whole branch carries y^T [384 ch][32 tok] = 192 accumulator registers plus the
//     bf16 LayerNorm output [32 tok][384 ch] = 96 registers: one wave per SIMD, the whole 512-register file.  (16 tokens per wave would
//     halve both but reads one 1 KiB weight fragment from LDS per MFMA = the LDS port's peak, measured at 18 % of the MFMA peak in round 3.)
//   * with one wave per SIMD nothing overlaps unless the instruction stream does: in iteration q the wave issues, interleaved in
//     program order,   the MFMAs of  P(q+1)^T = W1_chunk(q+1) LN(x)^T   (24, A fragments from LDS, accumulating into VGPRs),
//                      the VALU of   H(q) = GELU(P(q) + b1)             (16 values per lane; the bias is the MFMA's C operand),
//                      the MFMAs of  y^T += W2_chunk(q-1) H(q-1)^T      (24, accumulating into AccVGPRs),
//     i.e. per 32-hidden chunk 48 MFMAs (1536 cycles of the matrix pipe) beside ~200 VALU, 52 LDS reads and 12 LDS-DMA pieces: ~325 of the
//     384 four-cycle issue slots.  GELU is the erf form to 1.5e-5 (a (4,3) rational of the normal CDF, one v_rcp: 12 VALU per value; the
//     Abramowitz-Stegun form of common.h costs 16 + two transcendentals and would make the loop issue-bound).
//   * operand delivery matches the 256 x 256 GEMM tile: a workgroup (4 waves, 128 tokens) streams 48 KiB of weights per 6.3 MFLOP.
//
// Fragment conventions (32x32x16: A [32 x 16]: lane (n, hh) holds row n, k-slots 8hh .. 8hh+7; B [16 x 32]: lane (n, hh) holds column n,
// the same k-slots; D: lane (n, hh) register r holds D[(r & 3) + 8 (r >> 2) + 4 hh][n]).  The wave's tokens are the MFMA COLUMNS of every
// product.  Which hidden unit / output channel an MFMA ROW computes is free (it is the weight row its A fragment reads):
//   P^T: MFMA row rho = 16 t + 8 e2 + 4 hh + e0  <->  hidden unit 16 hh + 8 t + 4 e2 + e0 of the chunk: register 8 t + e of lane (n, hh)
//        is hidden unit 16 hh + 8 t + e -- after GELU + rounding the 16 registers ARE the two B fragments of the second product (k-step t,
//        slot 8 hh + e), and a lane owns 16 CONSECUTIVE hidden units of its token (32-byte side-output stores);
//   y^T: MFMA row rho = 8 b + 4 hh + r of tile mt  <->  channel 32 mt + 16 hh + 4 b + r: a lane owns 16 consecutive channels per tile
//        (64-byte pieces of x / y rows, the two lanes of a token one 128-byte line).
// LDS images (written by LDS-DMA: linear destination, permuted SOURCE address, the same involution on the read):
//   W1 chunk  [32 rows rho][C / 8 units of 16 B], unit u of row rho at position (u & ~15) | ((u ^ rho) & 15)      (24 KiB)
//   W2 chunk  [C / 32 tiles][32 rows rho][4 units], unit u = 2 t + hh at position u ^ ((rho >> 2) & 3)            (24 KiB)
//   both conflict-free for ds_read_b128's lane groups (MI355X_MICROARCH.md: {0-3,12-15,20-27} ...: 16 distinct 16-byte slots each).
// Rings: two slots per image; iteration q reads W1(q+1) | b1(q+1) | W2(q-1) from slot (q+1) & 1 and requests W1(q+2) | b1(q+2) | W2(q) into
// the other one; one workgroup barrier per iteration.
#include "common.h"
#include "../../include/esvit_hip.h"
#include <utility>

#ifdef ESVIT_P32_PROBE
// tools/probe only (tools/probe/build_p32.sh, tools/p32_timeline.py): s_memtime stamps of workgroup 0 / wave 0 around every iteration --
// [start, loop body done, waits done, barrier passed] -- and ablation switches of the iteration body
__device__ long* g_p32_tl = nullptr;
#ifndef P32_TL_BLOCK
#define P32_TL_BLOCK 0
#endif
#ifdef P32_STAMPS
#define P32_TL(it, slot) do { if (g_p32_tl && blockIdx.x == P32_TL_BLOCK && threadIdx.x == 0) g_p32_tl[(it) * 4 + (slot)] = (long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define P32_TL(it, slot) do { } while (0)
#endif
#else
#define P32_TL(it, slot) do { } while (0)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

constexpr int P32_WAVES = 4;
constexpr int P32_ROWS = 32 * P32_WAVES;
constexpr int P32_HCH = 32;

template <typename F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    sfor_impl(f, std::make_integer_sequence<int, N>{});
}

template <int C>
struct P32Cfg {
    static constexpr int KS1 = C / 16;                 // k-steps of the first product
    static constexpr int NT2 = C / 32;                 // 32-channel output tiles
    static constexpr int UA = C / 8;                   // 16-byte units per W1 image row
    static constexpr int W_BYTES = P32_HCH * C * 2;    // either image
    static constexpr int PW = W_BYTES / 1024;          // DMA pieces per image
    static constexpr int PPW = PW / P32_WAVES;         // ... per wave
    static_assert(PW % P32_WAVES == 0 && UA % 16 == 0, "C a multiple of 128");
    static constexpr int SLOT1 = W_BYTES + 1024;       // W1 image + the bias piece
    static constexpr int OFF2 = 2 * SLOT1;             // W2 ring
    static constexpr int LDS = 2 * SLOT1 + 2 * W_BYTES;
    __device__ __forceinline__ static int hid(int rho) { return 16 * ((rho >> 2) & 1) + 8 * (rho >> 4) + 4 * ((rho >> 3) & 1) + (rho & 3); }
    __device__ __forceinline__ static int chan(int rho) { return 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); }
    // per-lane source byte offset of DMA piece `piece` of the W1 image (chunk 0 of fc1.weight [4C, C] bf16)
    __device__ __forceinline__ static int voff1(int piece, int lane) {
        const int U = piece * 64 + lane, rho = U / UA, pu = U % UA;
        const int u = (pu & ~15) | ((pu ^ rho) & 15);
        return (hid(rho) * C + 8 * u) * 2;
    }
    // ... of the W2 image (chunk 0 of fc2.weight [C, 4C] bf16)
    __device__ __forceinline__ static int voff2(int piece, int lane) {
        const int U = piece * 64 + lane, R = U >> 2, mt = R >> 5, rho = R & 31;
        const int u = (U & 3) ^ ((rho >> 2) & 3);
        return ((32 * mt + chan(rho)) * 4 * C + 16 * (u & 1) + 8 * (u >> 1)) * 2;
    }
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t p32_rsrc(const void* base, long bytes) {
    const long capped = bytes > 0xfffffff0L ? 0xfffffff0L : (bytes < 0 ? 0 : bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)capped, 0x00020000);
}

// y^T tile += A * B with the accumulator pinned to the AccVGPR half of the register file (192 of them: the allocator must not copy them)
__device__ __forceinline__ void mma_acc32(const bf16x8& a, const bf16x8& b, f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// P^T (+)= A * B, same pinning (the GELU of the next iteration reads it through v_accvgpr_read: written a whole iteration earlier)
__device__ __forceinline__ void mma_p32(const bf16x8& a, const bf16x8& b, f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// the first MFMA of a chain takes the biases as its C operand: when the compiler has moved them into the accumulator registers with
// v_accvgpr_write (instead of reading them from LDS straight into AccVGPRs) the write -> MFMA-operand wait states must be inside the asm
// statement (hipcc pads nothing there; without them the MFMA read the previous chunk's values: a1 off by O(1) in the training variant)
__device__ __forceinline__ void mma_p32_first(const bf16x8& a, const bf16x8& b, f32x16& c) {
    asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// erf-GELU: x Phi(x), Phi(x) - 1/2 = u P(u^2) / Q(u^2) on |u| <= 4.5 (u = x clamped; max |error| of Phi 3.4e-6 = its tail beyond the clamp)
struct Gelu43 {
    static constexpr float P0 = 3.989448530e-01f, P1 = 2.708297554e-02f, P2 = 3.837898957e-03f, P3 = 3.371665041e-05f;
    static constexpr float Q1 = 2.345878969e-01f, Q2 = 2.366735537e-02f, Q3 = 1.174744135e-03f;
    static constexpr int STAGES = 12;
    float u, s, p, q;
    // stage k of the evaluation of one value (one VALU instruction each; v: the pre-activation; the result is left in p)
    // Every result passes through an empty asm statement: the statement is ordered against the MFMA statements around it, so the
    // instruction stays in the MFMA shadow it was written into (plain arithmetic floats freely through instruction selection whatever
    // sched_barrier says, and hipcc gathered the GELU into blocks of 20 VALU instructions between runs of bare MFMAs).
    template <int K>
    __device__ __forceinline__ void stage(const float v) {
        if constexpr (K == 0) { u = __builtin_amdgcn_fmed3f(v, -4.5f, 4.5f); pin(u); }
        else if constexpr (K == 1) { s = u * u; pin(s); }
        else if constexpr (K == 2) { p = __builtin_fmaf(P3, s, P2); pin(p); }
        else if constexpr (K == 3) { q = __builtin_fmaf(Q3, s, Q2); pin(q); }
        else if constexpr (K == 4) { p = __builtin_fmaf(p, s, P1); pin(p); }
        else if constexpr (K == 5) { q = __builtin_fmaf(q, s, Q1); pin(q); }
        else if constexpr (K == 6) { p = __builtin_fmaf(p, s, P0); pin(p); }
        else if constexpr (K == 7) { q = __builtin_fmaf(q, s, 1.0f); pin(q); }
        else if constexpr (K == 8) { q = __builtin_amdgcn_rcpf(q); pin(q); }
        else if constexpr (K == 9) { p = p * u; pin(p); }
        else if constexpr (K == 10) { p = __builtin_fmaf(p, q, 0.5f); pin(p); }
        else { p = p * v; pin(p); }
    }
    __device__ __forceinline__ static void pin(float& x) { asm volatile("" : "+v"(x)); }
};

struct P32Args {
    const float* x;
    const float* gamma;
    const float* beta;
    float eps;
    const bf16* W1;
    const float* b1;
    const bf16* W2;
    const float* b2;
    const float* rowscale;
    long M;
    float* y;
    // training pass (SIDE): what the unfused backward reads
    bf16* a1;    // pre-activation  [M, 4C]
    bf16* a1g;   // GELU            [M, 4C]
    bf16* h;     // LayerNorm(x)    [M, C]
    float* mean;
    float* rstd;
};

template <int C, bool SIDE>
__device__ __forceinline__ void mlp32p_fwd_body(const P32Args& A) {
    using Cf = P32Cfg<C>;
    constexpr int H4 = 4 * C, NCH = H4 / P32_HCH;
    static_assert(NCH % 2 == 0, "the steady loop runs two chunks per trip");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, hh = lane >> 5;
    const long row = (long)blockIdx.x * P32_ROWS + wave * 32 + n;
    const bool row_ok = row < A.M;
    const long rrow = row_ok ? row : (A.M - 1);  // out-of-range lanes compute on a valid row and store nothing

    // ---- weight chunk DMA ----
    const __amdgpu_buffer_rsrc_t r1 = p32_rsrc(A.W1, (long)H4 * C * 2), r2 = p32_rsrc(A.W2, (long)C * H4 * 2), r3 = p32_rsrc(A.b1, (long)H4 * 4);
    int vo1[Cf::PPW], vo2[Cf::PPW];
#pragma unroll
    for (int i = 0; i < Cf::PPW; ++i) {
        vo1[i] = Cf::voff1(wave + P32_WAVES * i, lane);
        vo2[i] = Cf::voff2(wave + P32_WAVES * i, lane);
    }
    const int vo3 = (lane & 31) * 4;  // the chunk's 32 biases, one copy per wave (4-byte pieces: no wave-dependent branch inside the iteration)
    // piece i of this wave's share of W1(q) | b1(q) (into W1 slot `slot`) resp. W2(q) (into W2 slot `slot`)
    auto dma1 = [&](auto ic, int q, int slot) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        char* dst = smem + slot * Cf::SLOT1 + (wave + P32_WAVES * i) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_void*)dst, 16, vo1[i], q * (P32_HCH * C * 2), 0, 0);
    };
    auto dma2 = [&](auto ic, int q, int slot) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        char* dst = smem + Cf::OFF2 + slot * Cf::W_BYTES + (wave + P32_WAVES * i) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_void*)dst, 16, vo2[i], q * (P32_HCH * 2), 0, 0);
    };
    auto dma3 = [&](int q, int slot) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r3, (lds_void*)(smem + slot * Cf::SLOT1 + Cf::W_BYTES + wave * 256), 4, vo3, q * (P32_HCH * 4), 0, 0);
    };
    sfor<Cf::PPW>([&](auto ic) { dma1(ic, 0, 0); });
    dma3(0, 0);

    // ---- LayerNorm of this lane's half rows, straight into the B fragments of the first product: k-step s, slot 8 hh + e = channel 16 s + 8 hh + e ----
    bf16x8 xb[Cf::KS1];
    {
        const float* xr = A.x + rrow * C + 8 * hh;
        float xv[Cf::KS1][8];
        float s1 = 0.f;
#pragma unroll
        for (int s = 0; s < Cf::KS1; ++s) {
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
        for (int s = 0; s < Cf::KS1; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = xv[s][e] - mean;
                s2 += d * d;
            }
        s2 += __shfl_xor(s2, 32, 64);
        const float rstd = rsqrtf(s2 * (1.f / C) + A.eps);
        if constexpr (SIDE) {
            if (row_ok && hh == 0) {
                A.mean[row] = mean;
                A.rstd[row] = rstd;
            }
        }
#pragma unroll
        for (int s = 0; s < Cf::KS1; ++s) {
            const float* gp = A.gamma + 16 * s + 8 * hh;
            const float* bp = A.beta + 16 * s + 8 * hh;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(bp), c1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xb[s][e] = (bf16)((xv[s][e] - mean) * rstd * g0[e] + c0[e]);
                xb[s][4 + e] = (bf16)((xv[s][4 + e] - mean) * rstd * g1[e] + c1[e]);
            }
            if constexpr (SIDE) {
                if (row_ok) *reinterpret_cast<bf16x8*>(A.h + row * C + 16 * s + 8 * hh) = xb[s];
            }
        }
    }

    // side outputs (training pass): one descriptor per tensor, per-lane byte offset of this lane's 32 bytes of chunk 0 (out of range for rows past M)
    const __amdgpu_buffer_rsrc_t ra1 = p32_rsrc(A.a1, SIDE ? A.M * H4 * 2 : 0), ra1g = p32_rsrc(A.a1g, SIDE ? A.M * H4 * 2 : 0);
    const unsigned side_off = row_ok ? (unsigned)(row * (H4 * 2) + 32 * hh) : 0x80000000u;  // (the tensors are below 2 GiB: the host checks)

    f32x16 acc[Cf::NT2];  // y^T: tile mt, register i <-> channel 32 mt + 16 hh + i of token n
#pragma unroll
    for (int t = 0; t < Cf::NT2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- LDS read addresses (bytes): W1 fragment of k-step s at a1b[(2s & 15) >> 1] + (2s & ~15) * 16 (+ slot); W2 fragment (mt, t) at a2b[t] + mt * 2048 ----
    int a1b[8], a2b[2];
    {
        const int k = (hh ^ n) & 15;
#pragma unroll
        for (int j = 0; j < 8; ++j) a1b[j] = n * (C * 2) + (((2 * j) ^ k) * 16);
#pragma unroll
        for (int t = 0; t < 2; ++t) a2b[t] = Cf::OFF2 + n * 64 + (((2 * t + hh) ^ ((n >> 2) & 3)) * 16);
    }
    const int a3b = Cf::W_BYTES + wave * 256 + hh * 64;  // this half's 16 biases (this wave's copy): hidden units 16 hh .. 16 hh + 15 of the chunk

    auto frag1 = [&](auto sc, int slot) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        return *reinterpret_cast<const bf16x8*>(smem + a1b[((2 * s) & 15) >> 1] + ((2 * s) & ~15) * 16 + slot * Cf::SLOT1);
    };
    auto frag2 = [&](auto jc, int slot) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;  // MFMA j of the second product: tile j >> 1, k-step j & 1
        return *reinterpret_cast<const bf16x8*>(smem + a2b[j & 1] + (j >> 1) * 2048 + slot * Cf::W_BYTES);
    };

    auto iter_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // One iteration (chunk index q; PAR = (q + 1) & 1 is the ring slot it READS).  G1: P_out = W1(q+1) LN(x)^T + b1(q+1);  GL: H_out = GELU(P_in)
    // (+ side outputs of chunk q);  G2: acc += W2(q-1) H_in^T.  Requests W1(q+2) | b1(q+2) (when ISSUE1) and W2(q) (when ISSUE2) into slot PAR ^ 1.
    constexpr int D = 4;  // fragment reads run D MFMA pairs ahead of their use
    auto iteration = [&](auto parc, auto g1c, auto glc, auto g2c, auto i1c, auto i2c, const int q, f32x16& P_out, f32x16& P_in, bf16x8 (&H_out)[2],
                         const bf16x8 (&H_in)[2]) __attribute__((always_inline)) {
        constexpr int PAR = decltype(parc)::value;
        // (probe builds: an ablated part leaves its inputs consumed and its outputs opaque, so that nothing else is folded away)
        constexpr bool G1 = decltype(g1c)::value, GL = decltype(glc)::value, G2 = decltype(g2c)::value;
#if defined(ESVIT_P32_PROBE) && defined(P32_NO_G1)
        constexpr bool G1_MFMA = false;
#else
        constexpr bool G1_MFMA = true;
#endif
#if defined(ESVIT_P32_PROBE) && defined(P32_NO_GELU)
        constexpr bool GL_VALU = false;
#else
        constexpr bool GL_VALU = true;
#endif
#if defined(ESVIT_P32_PROBE) && defined(P32_NO_G2)
        constexpr bool G2_MFMA = false;
#else
        constexpr bool G2_MFMA = true;
#endif
#if defined(ESVIT_P32_PROBE) && defined(P32_NO_DMA)
        constexpr bool ISSUE1 = false, ISSUE2 = false;
#else
        constexpr bool ISSUE1 = decltype(i1c)::value, ISSUE2 = decltype(i2c)::value;
#endif
        constexpr int NJ = Cf::KS1;  // MFMA pairs of the iteration (24 + 24 at C = 384)
        static_assert(2 * Cf::NT2 == Cf::KS1, "the two products of a chunk have the same MFMA count");
        P32_TL(q + 1, 0);
        // P_in was written by MFMAs the compiler cannot see (asm): this statement keeps every read of it (v_accvgpr_read) on this side of the
        // previous iteration's barrier, an MFMA latency and more after the last of them (hipcc otherwise read the first registers right
        // behind the last MFMA in the peeled tail iterations: chunks 46 / 47 of a1 were off by O(1))
        asm volatile("" : "+a"(P_in));
        bf16x8 f1[NJ], f2[NJ];  // (SSA values: D + 1 of each are live at a time)
        f32x16 bias;
        if constexpr (G1) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(smem + a3b + PAR * Cf::SLOT1 + 16 * v);
#pragma unroll
                for (int e = 0; e < 4; ++e) bias[4 * v + e] = b4[e];
            }
        }
        sfor<D>([&](auto jc) {
            if constexpr (G1) f1[decltype(jc)::value] = frag1(jc, PAR);
            if constexpr (G2) f2[decltype(jc)::value] = frag2(jc, PAR);
        });
        Gelu43 ge[2];
        float gv[16];
        constexpr int OPS = 8 * (2 * Gelu43::STAGES + 1);  // eight pairs of values: 2 x 12 stages + one packing
        constexpr int NH = 2 * NJ;                         // MFMAs of the iteration: even h = first product, odd h = second product (k-step / tile h >> 1)
        constexpr int OPS_PER_H = (OPS + NH - 1) / NH;
        __builtin_amdgcn_sched_barrier(0);
        // One MFMA per half-slot followed by its fillers (the wave is alone on its SIMD: what is not placed in an MFMA's shadow is not hidden):
        // a fragment read D pairs ahead, at most one LDS-DMA request, OPS_PER_H VALU operations of the GELU.
        sfor<NH>([&](auto hc) {
            constexpr int h = decltype(hc)::value, j = h >> 1;
            constexpr bool first = (h & 1) == 0;
            if constexpr (first) {
                if constexpr (G1) {
                    if constexpr (G1_MFMA) {
                        if constexpr (j == 0) {
                            P_out = bias;
                            mma_p32_first(f1[j], xb[j], P_out);
                        } else {
                            mma_p32(f1[j], xb[j], P_out);
                        }
                    } else {
                        if constexpr (j == 0) P_out = bias;
                        asm volatile("" : "+v"(P_out) : "v"(f1[j]), "v"(xb[j]));
                    }
                }
                if constexpr (j + D < NJ && G1) f1[j + D] = frag1(std::integral_constant<int, j + D>{}, PAR);
                // the requests of the next chunks go out in the first half of the iteration (they must have landed at its end)
                if constexpr (j < Cf::PPW && ISSUE1) dma1(std::integral_constant<int, j>{}, q + 2, PAR ^ 1);
                if constexpr (j == Cf::PPW && ISSUE1) dma3(q + 2, PAR ^ 1);
            } else {
                if constexpr (G2) {
                    if constexpr (G2_MFMA) mma_acc32(f2[j], H_in[j & 1], acc[j >> 1]);
                    else asm volatile("" : "+a"(acc[j >> 1]) : "v"(f2[j]), "v"(H_in[j & 1]));
                }
                if constexpr (j + D < NJ && G2) f2[j + D] = frag2(std::integral_constant<int, j + D>{}, PAR);
                if constexpr (j < Cf::PPW && ISSUE2) dma2(std::integral_constant<int, j>{}, q, PAR ^ 1);
            }
            if constexpr (GL && !GL_VALU) {
                if constexpr (h < 8) {
                    H_out[h >> 2][2 * (h & 3)] = (bf16)P_in[2 * h];
                    H_out[h >> 2][2 * (h & 3) + 1] = (bf16)P_in[2 * h + 1];
                }
            }
            if constexpr (GL && GL_VALU) {
                // VALU of this half-slot: operations [h * OPS_PER_H, (h + 1) * OPS_PER_H) of the flattened list (pair-major: values 2k, 2k + 1 alternate)
                sfor<OPS_PER_H>([&](auto oc) {
                    constexpr int o = h * OPS_PER_H + decltype(oc)::value;
                    if constexpr (o < OPS) {
                        constexpr int pair = o / (2 * Gelu43::STAGES + 1), w = o % (2 * Gelu43::STAGES + 1);
                        if constexpr (w < 2 * Gelu43::STAGES) {
                            constexpr int el = 2 * pair + (w & 1), st = w >> 1;
                            ge[w & 1].template stage<st>(P_in[el]);
                            if constexpr (st == Gelu43::STAGES - 1) gv[el] = ge[w & 1].p;
                        } else {
                            constexpr int e0 = 2 * pair;
                            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                            bf16x2_t pk = {(bf16)gv[e0], (bf16)gv[e0 + 1]};
                            unsigned pku = __builtin_bit_cast(unsigned, pk);
                            asm volatile("" : "+v"(pku));
                            pk = __builtin_bit_cast(bf16x2_t, pku);
                            H_out[e0 >> 3][e0 & 7] = pk[0];
                            H_out[e0 >> 3][(e0 & 7) + 1] = pk[1];
                        }
                    }
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        constexpr bool STORES = GL && SIDE;
        if constexpr (STORES) {
            // chunk q of the side outputs: hidden units 32 q + 16 hh + 0 .. 15 of token n (two 16-byte stores per tensor and lane).  Buffer stores:
            // rows past M fall outside the descriptors' ranges, so every wave issues exactly four store instructions per iteration.  (The
            // measurements of this variant were taken with "vmcnt(4)" below -- the stores left in flight; that is only sound if stores and loads
            // retire in ONE order, and they do not: profiles/r06_gemm_astat_probe.txt.  The wait is vmcnt(0) now.)
            const unsigned so = side_off + (unsigned)(q * P32_HCH * 2);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bf16x8 pre;
#pragma unroll
                for (int e = 0; e < 8; ++e) pre[e] = (bf16)P_in[8 * t + e];
#if defined(ESVIT_P32_PROBE) && defined(P32_NO_SIDE_STORE)
                asm volatile("" :: "v"(pre), "v"(H_out[t]), "v"(so));
#elif defined(ESVIT_P32_PROBE) && defined(P32_SIDE_LINEAR)
                // (timing probe, wrong layout: the same bytes as whole 1 KiB lines per wave-instruction -- what coalesced stores would cost)
                {
                    const unsigned lo = (unsigned)((((long)blockIdx.x * P32_WAVES + wave) * NCH + q) * 2 + t) * 1024u + (unsigned)lane * 16u;
                    buffer_store_b128(pre, ra1, lo, 0);
                    buffer_store_b128(H_out[t], ra1g, lo, 0);
                }
#elif defined(ESVIT_P32_PROBE) && defined(P32_SIDE_NT)
                buffer_store_b128<2>(pre, ra1, so + 16 * t, 0);  // (timing probe: non-temporal)
                buffer_store_b128<2>(H_out[t], ra1g, so + 16 * t, 0);
#elif defined(ESVIT_P32_PROBE) && defined(P32_SIDE_SC1)
                buffer_store_b128<16>(pre, ra1, so + 16 * t, 0);  // (timing probe: write-through)
                buffer_store_b128<16>(H_out[t], ra1g, so + 16 * t, 0);
#else
                buffer_store_b128(pre, ra1, so + 16 * t, 0);
                buffer_store_b128(H_out[t], ra1g, so + 16 * t, 0);
#endif
            }
        }
        P32_TL(q + 1, 1);
#if defined(ESVIT_P32_PROBE) && defined(P32_NO_DMA_WAIT)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
#if defined(ESVIT_P32_PROBE) && defined(P32_NO_SIDE_STORE)
        if constexpr (STORES) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
        if constexpr (STORES) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
        P32_TL(q + 1, 2);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        P32_TL(q + 1, 3);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    f32x16 P0, P1;
    bf16x8 H0[2], H1[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) P0[r] = P1[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) H0[t][e] = H1[t][e] = (bf16)0.f;

    iter_barrier();  // W1(0) | b1(0) have landed for every wave (and this wave's x loads are done)
    // q = -1: first product of chunk 0 (reads slot 0), requests W1(1)
    iteration(I0{}, T_{}, F_{}, F_{}, T_{}, F_{}, -1, P0, P1, H1, H0);
    // q = 0: first product of chunk 1 (slot 1) beside GELU(0); requests W1(2), W2(0)
    iteration(I1{}, T_{}, T_{}, F_{}, T_{}, T_{}, 0, P1, P0, H0, H1);
    // steady state: q odd reads slot 0, writes P0 / H1, reads P1 / H0
#pragma unroll 1
    for (int q = 1; q < NCH - 3; q += 2) {
        iteration(I0{}, T_{}, T_{}, T_{}, T_{}, T_{}, q, P0, P1, H1, H0);
        iteration(I1{}, T_{}, T_{}, T_{}, T_{}, T_{}, q + 1, P1, P0, H0, H1);
    }
    // q = NCH - 3 (odd): the last request of W1 (chunk NCH - 1); q = NCH - 2: no W1 left to request
    iteration(I0{}, T_{}, T_{}, T_{}, T_{}, T_{}, NCH - 3, P0, P1, H1, H0);
    iteration(I1{}, T_{}, T_{}, T_{}, F_{}, T_{}, NCH - 2, P1, P0, H0, H1);
    // q = NCH - 1: GELU of the last chunk, second product of chunk NCH - 2; requests W2(NCH - 1)
    iteration(I0{}, F_{}, T_{}, T_{}, F_{}, T_{}, NCH - 1, P0, P1, H1, H0);
    // q = NCH: second product of the last chunk
    iteration(I1{}, F_{}, F_{}, T_{}, F_{}, F_{}, NCH, P1, P0, H0, H1);

    // ---- epilogue: y = x + rowscale * (acc + b2); this lane: token n, channels 32 mt + 16 hh + (0 .. 15) ----
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (the last MFMAs were issued from asm: their results are read below ...
#pragma unroll
    for (int mt = 0; mt < Cf::NT2; ++mt) asm volatile("" : "+a"(acc[mt]));  // ... and no read of them may be scheduled above the wait states)
    const float rs = A.rowscale ? A.rowscale[rrow] : 1.f;
#pragma unroll
    for (int mt = 0; mt < Cf::NT2; ++mt) {
        const int c0 = 32 * mt + 16 * hh;
        f32x4 xv[4], bb[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            xv[v] = *reinterpret_cast<const f32x4*>(A.x + rrow * C + c0 + 4 * v);
            bb[v] = *reinterpret_cast<const f32x4*>(A.b2 + c0 + 4 * v);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = xv[v][e] + rs * (acc[mt][4 * v + e] + bb[v][e]);
            if (row_ok) *reinterpret_cast<f32x4*>(A.y + row * C + c0 + 4 * v) = o;
        }
    }
}

// (the body lives in a __device__ function: with the LDS-DMA builtin inside the __global__ function itself hipcc 7.2 emitted no host stub for it)
template <int C, bool SIDE>
__global__ __launch_bounds__(P32_WAVES * 64, 1) void mlp32p_fwd_kernel(const P32Args A) {
    mlp32p_fwd_body<C, SIDE>(A);
}

template <int C, bool SIDE>
int launch_p32(const P32Args& a, hipStream_t stream) {
    auto kern = mlp32p_fwd_kernel<C, SIDE>;
    static unsigned long long lds_set = 0;
    esvit_raise_lds(kern, P32Cfg<C>::LDS, lds_set);
    hipLaunchKernelGGL(kern, dim3(ceil_div(a.M, P32_ROWS)), dim3(P32_WAVES * 64), P32Cfg<C>::LDS, stream, a);
    ESVIT_CHECK_LAUNCH("esvit_mlp_fused_fwd(32p)");
    return ESVIT_OK;
}

}  // namespace

// C = 384.  W1 / W2: the plain activation-dtype casts of fc1.weight [4C, C] / fc2.weight [C, 4C].  a1 .. rstd: all null (inference pass) or all
// given (training pass: the operands of the unfused backward)
int esvit_i_mlp32p_fwd(const float* x, const float* gamma, const float* beta, float eps, const void* W1, const float* b1, const void* W2,
                       const float* b2, const float* rowscale, long M, int C, float* y, void* a1, void* a1g, void* h, float* mean, float* rstd,
                       hipStream_t stream) {
    if (C != 384) return ESVIT_ERR_UNSUPPORTED;
    if (a1) {
        if ((long)M * 4 * C * 2 >= 0x7ff00000L) return ESVIT_ERR_UNSUPPORTED;  // (the side outputs are addressed through 32-bit buffer offsets, rows past M at 2 GiB)
        P32Args a{x, gamma, beta, eps, (const bf16*)W1, b1, (const bf16*)W2, b2, rowscale, M, y, (bf16*)a1, (bf16*)a1g, (bf16*)h, mean, rstd};
        return launch_p32<384, true>(a, stream);
    }
    P32Args a{x, gamma, beta, eps, (const bf16*)W1, b1, (const bf16*)W2, b2, rowscale, M, y, nullptr, nullptr, nullptr, nullptr, nullptr};
    return launch_p32<384, false>(a, stream);
}

#ifdef ESVIT_P32_PROBE
extern "C" __attribute__((visibility("default"))) int p32_probe_set_timeline(long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_p32_tl), &buf, sizeof(buf));
}
extern "C" __attribute__((visibility("default"))) int p32_probe_fwd(const float* x, const float* gamma, const float* beta, float eps, const void* W1, const float* b1,
                                                                    const void* W2, const float* b2, long M, float* y, void* a1, void* a1g, void* h, float* mean,
                                                                    float* rstd, void* stream) {
    return esvit_i_mlp32p_fwd(x, gamma, beta, eps, W1, b1, W2, b2, nullptr, M, 384, y, a1, a1g, h, mean, rstd, reinterpret_cast<hipStream_t>(stream));
}
void esvit_set_error(const char*, ...) {}
#endif
