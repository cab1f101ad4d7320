// Fused attention branch of a Swin block (swin_transformer.py:283-330 with 120-152): for 7x7 windows of C = 96 / 192 channels
// (stages 0 / 1, head_dim 32) ONE kernel computes
//     y = x + rowscale * ( proj( window_attention( qkv( LayerNorm(x) ) ) ) + b_proj )
// with the window's tokens resident on the chip: the LayerNorm output, qkv and the attention output never reach HBM (8 B per
// token-channel instead of 32).  A training pass that still runs the unfused backward can ask for the tensors that backward
// reads (LayerNorm output + statistics, qkv, attention output) as side outputs of the same pass.
//
// Work split.  A workgroup owns NWIN windows at a time, four waves per window; wave (j, w) owns the 16 slots 16w .. 16w+15 of
// window j (49 tokens in 64 slots) as the MFMA COLUMNS of every product it forms (fragment conventions: mlp_fused16.hip / fused16.h):
//   q^T, k^T [32 x 16]      = W_{q,k}[head] * LN(x)^T          A: weight rows from LDS (perm32 columns), B: LN(x) in registers;
//                                                              MFMA row i of tile t <-> head channel 8 (i >> 2) + 4 t + (i & 3), so a
//                                                              lane (c, g) ends up with channels 8g .. 8g+7 of its token: q^T IS the B
//                                                              fragment of the score product, k leaves as one 16-byte LDS store
//   V [16 x 32]             = LN(x) * W_v[head]^T              the same fragments with the MFMA operands exchanged: a lane holds 4
//                                                              slots of one channel -> one 8-byte store into the V^T image
//   S^T [64 keys x 16 q]    = K * (scale log2e q)^T + bias     A: K rows from the window's LDS image (all four waves wrote it)
//   softmax over keys, base 2: in registers + two VALU butterfly steps (v_permlane16_swap / v_permlane32_swap, common.h)
//   O^T [32 x 16]           = V^T * P^T                        A: two 8-byte reads of the V^T image per fragment, in the key order the
//                                                              score accumulators carry; B: P (unnormalised) from registers
//   y^T [C x 16]           += Wproj[:, head] * O^T             A: the head's 32 columns of Wproj from LDS (perm32 columns: the two
//                                                              16-channel tiles of O^T are the permuted k-slots), B: O * (rowscale /
//                                                              softmax sum) from registers
// The accumulator of y^T starts as x + rowscale * b_proj and O is scaled by the row's DropPath factor, so the finished accumulator IS
// the output row.  Zero-pad slots (win2tok = -1) and the idle slots 49 .. 63 carry LN(x) = 0, i.e. q / k / v = bias as in the
// reference (padding happens after norm1); idle keys are masked by the -1e30 columns of the fragment-order bias.
//
// Pipeline.  The (window group, head) pairs of a workgroup are ONE sequence of steps; step s runs the attention + projection of
// pair s-1 (piece A) and the q | k | v products of pair s (piece Q; at a window's first head also the LayerNorm of its rows, piece N)
// between two workgroup barriers.  K / V^T images are double-buffered per window, so one barrier per step orders everything.  The
// window slots of a workgroup run the pieces in opposite orders (A N Q / N Q A): the two waves of a SIMD belong to different slots,
// so one is in its MFMA / LDS-bound piece while the other runs the VALU-bound softmax.  Weights: when the slices of all heads fit
// beside the images (C = 96: 77 KB) they are requested once, during the workgroup's first window, and stay resident; otherwise
// (C = 192) they stream L2 -> LDS by LDS-DMA per head through two-deep rings, one step ahead.  V is kept as a V^T image written from a
// NON-transposed product (the same weight fragments with the MFMA operands exchanged), so P V needs no transposing LDS read.
//
// Memory operations are counted, not waited for.  The token rows of the NEXT window (HBM) are prefetched into registers, a few
// 16-byte loads at the top of every step after the weight requests; they, the output stores and the side-output stores are the
// YOUNGEST operations of a step; its closing `s_waitcnt vmcnt(n)` leaves the row LOADS in flight and waits for everything else, the stores
// included (stores retire out of order with respect to loads: an allowance that counts them is unsound; with resident weights there is
// nothing to wait for after the first window).  For the count -- and hipcc's own inserted waits -- to be exact, every memory
// operation of the loop is unconditional: rows that do not exist are addressed out of range (row_off), a wave without a DMA piece
// issues one into a spare KiB.  The fragment-order bias of all heads lives in registers: a load inside the step would have to be
// waited for behind the HBM prefetch (vmcnt retires in order).  DESIGN.md 4.2b has the measurements behind each of these.
#include "common.h"
#include "fused16.h"
#include "../../include/esvit_hip.h"

struct ABParams {
    const float* x;         // [rows, C] fp32: rows of this resolution group (image b, token t -> row b * L + t)
    const float* gamma;
    const float* beta;
    float eps;
    const bf16* Wqkv;       // [3C, C] bf16, perm32 columns
    const float* bqkv;      // [3C]
    const bf16* Wproj;      // [C, C] bf16, perm32 columns
    const float* bproj;     // [C]
    const float* bias_frag; // [nH][AB_FRAG]
    const int* win2tok;     // [nW * N]
    const int* region_ids;  // [nW * N] or null
    const float* rowscale;  // [rows] or null
    float* y;               // [rows, C] fp32
    bf16* xw;               // side outputs (all or none): LayerNorm(x) [rows, C]
    bf16* qkv;              //   [rows, 3C]
    bf16* ao;               //   [rows, C]
    float* mean;            //   [rows]
    float* rstd;            //   [rows]
    int nW, Bw, N, L;
    long rows;
    float scale;
    unsigned* timeline;     // tools/attn_branch_timeline.py builds only (ESVIT_AB_TIMELINE): per-phase cycle stamps, else unused
};

// per-phase cycle stamps of wave 0 and the last wave of workgroups 0 and 1, first AB_TL_STEPS steps, kept in LDS and written out at
// the end of the kernel (tools/attn_branch_timeline.py; never defined in the product build)
#ifdef ESVIT_AB_TIMELINE
#define AB_TL_STEPS 48
#define AB_TL_BYTES (AB_TL_STEPS * 2 * 12 * 4)
#define TL(i_)                                                                                                                     \
    do {                                                                                                                           \
        if (tl_on && tl_step < AB_TL_STEPS) tl_lds[(tl_step * 2 + tl_w) * 12 + (i_)] = (unsigned)__builtin_readcyclecounter();     \
    } while (0)
#else
#define AB_TL_BYTES 0
#define TL(i_)
#endif

// scheduling fences between the pieces of a step (register pressure against overlap: tools/bench_attn_branch.py decides)
#ifdef ESVIT_AB_NO_SCHED_BARRIER
#define AB_SCHED_BARRIER()
#else
#define AB_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif

namespace {

constexpr int AB_FRAG = 16 * 256;   // floats per head of the fragment-order bias (window_attn.hip: FRAG_ELEMS)
constexpr int K_BYTES = 64 * 64;    // K image [64 slots][32 d] bf16: 64-byte rows, 16-byte units XOR-swizzled by (row >> 1) & 3
constexpr int VT_LD = 136;          // V^T image [32 d][64 slots] bf16: 128-byte rows padded to 136 (8-byte fragment reads, conflict-free)
constexpr int KV_BYTES = K_BYTES + ((32 * VT_LD + 1023) / 1024) * 1024;  // one K | V^T pair
constexpr unsigned AB_OOB = 0x7ffffff0u;
constexpr float LOG2E = 1.4426950408889634f;

template <int I>
struct IC {
    static constexpr int value = I;
};
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

// sum / max over the four lanes (c, g = 0..3) of a token
__device__ __forceinline__ float tok_sum(float v) { return xor32_sum(xor16_sum(v)); }
__device__ __forceinline__ float tok_max(float v) { return xor32_max(xor16_max(v)); }

__device__ __forceinline__ void mem_fence_compiler() { asm volatile("" ::: "memory"); }

// byte offset of a buffer access that must be ISSUED whether or not its row exists (the step's vmcnt bookkeeping counts it): `off`
// for row >= 0, an out-of-range offset otherwise (loads return zeros, stores are dropped).  Branch-free and opaque to the
// optimiser, which otherwise turns the select into two predicated copies of the access.
__device__ __forceinline__ unsigned row_off(int row, unsigned off) {
    const unsigned m = (unsigned)(row >> 31);
    unsigned v = (off & ~m) | (AB_OOB & m);
    asm volatile("" : "+v"(v));
    return v;
}

template <int C, int NWIN>
struct ABCfg {
    using Cf = Cfg16<C>;
    static constexpr int NH = C / 32;
    static constexpr int NW = 4 * NWIN;
    static constexpr int QBUF = 3 * Cf::A_BYTES + 1024;  // q | k | v row images of a head, then its 96 biases (fp32)
    static constexpr int PBUF = Cf::B_BYTES;             // [C][32]: the head's columns of Wproj
    static constexpr int CONST_BYTES = ((3 * C * 4 + 1023) / 1024) * 1024;  // gamma | beta | b_proj
    // ring depth of the weight slices: every head resident (requested once, during the workgroup's first window) when that fits
    // beside the K / V images, else two slots and a request per step
    static constexpr bool RES = CONST_BYTES + NH * (QBUF + PBUF) + NWIN * 2 * KV_BYTES + 1024 + AB_TL_BYTES <= 160 * 1024;
    static constexpr int RING = RES ? NH : 2;
    static constexpr int OFF_Q = CONST_BYTES;
    static constexpr int OFF_P = OFF_Q + RING * QBUF;
    static constexpr int OFF_KV = OFF_P + RING * PBUF;   // [NWIN][2 buffers][K | V^T]
    static constexpr int OFF_DUMMY = OFF_KV + NWIN * 2 * KV_BYTES;  // landing KiB of the DMA pieces a wave issues only to keep every wave's count equal
    static constexpr int LDS = OFF_DUMMY + 1024 + AB_TL_BYTES;
    static constexpr int NPQ = 3 * Cf::PA + 1;           // DMA pieces of a q | k | v slice (+ the bias piece)
    static constexpr int PPQ = (NPQ + NW - 1) / NW;
    static constexpr int PPP = (Cf::PB + NW - 1) / NW;
    static constexpr int NXL = 2 * Cf::KS;               // 16-byte row loads per lane and window
    static constexpr int PF = (NXL + NH - 2) / (NH - 1); // of them per step (steps 1 .. NH-1 of the previous window)
};


template <int C, int NWIN, bool SAVE>
__device__ __forceinline__ void attn_branch_fwd_body(const ABParams& p) {
    using AC = ABCfg<C, NWIN>;
    using Cf = Cfg16<C>;
    constexpr int NH = AC::NH, NW = AC::NW, KS = Cf::KS, MT = Cf::MT, NXL = AC::NXL, PF = AC::PF;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int wj = wave >> 2, w = wave & 3;  // window slot of the workgroup, 16-slot tile of the window
    const int slot = 16 * w + c;

#ifdef ESVIT_AB_TIMELINE
    unsigned* tl_lds = reinterpret_cast<unsigned*>(smem + AC::OFF_DUMMY + 1024);
    const bool tl_on = p.timeline && blockIdx.x < 2 && lane == 0 && (wave == 0 || wave == NW - 1);
    const int tl_w = wave == 0 ? 0 : 1;
    int tl_step = 0;
#endif
    // ---- constants into LDS ----
    {
        float* cst = reinterpret_cast<float*>(smem);
        for (int i = threadIdx.x; i < 3 * C; i += NW * 64) cst[i] = i < C ? p.gamma[i] : (i < 2 * C ? p.beta[i - C] : p.bproj[i - 2 * C]);
    }

    // ---- per-lane LDS offsets, computed once (every other address is one of these plus a uniform ring offset plus an immediate) ----
    // weight-row fragment (tile t, k-step ks) inside an image A.  The unit swizzle of Cfg16 is an XOR of the low unit bits with a
    // function of the row: at C = 96 it leaves the k-step additive (+ 64 B per k-step), at C = 192 the k-step's parity takes part
    // (two registers per tile, + 128 B per pair of k-steps)
    constexpr int FAP = C == 96 ? 1 : 2;
    int fa_[2][FAP];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < FAP; ++q) fa_[t][q] = Cf::frag_a(t, q, c, g);
    auto fa = [&](int t, int ks) -> int { return C == 96 ? fa_[t][0] + 64 * ks : fa_[t][ks & (FAP - 1)] + 128 * (ks >> 1); };
    const int fb = Cf::frag_b(0, c, g);  // Wproj column fragment of tile 0 (tile mt: + 1024 mt)
    const int lc = 4 * g * 4;            // this lane's 4 channels of a 16-channel tile inside gamma / beta / b_proj
    // K / V images [64 slots][64 B], unit u of row r at r * 64 + ((u ^ ((r >> 1) & 3)) << 4)
    const int kr = c * 64 + ((g ^ ((c >> 1) & 3)) << 4);                    // K fragment of key tile 0 (tile i: + 1024 i)
    const int kw = slot * 64 + ((g ^ ((slot >> 1) & 3)) << 4);              // where this lane's 8 channels of k / v go
    // V^T fragment of O^T = V^T P^T: row d = 16 dt + c, k-slots 8g + e <-> keys 32 ks + 4g + e (e < 4) / 32 ks + 16 + 4g + e - 4, the
    // order the score accumulators hold P in: two 8-byte reads (+ 16 VT_LD dt, + 64 ks, + 32 for the second)
    const int vr = c * VT_LD + 8 * g;
    // where the V product of the q | k | v step leaves its accumulators (tile t, column c <-> head channel 8 (c >> 2) + 4t + (c & 3),
    // rows = this wave's slots 16w + 4g .. +3): + 4 VT_LD t
    const int vw = (8 * (c >> 2) + (c & 3)) * VT_LD + (16 * w + 4 * g) * 2;
    const int vb = (8 * (c >> 2) + (c & 3)) * 4;  // this lane's channel inside the head's 32 v biases (+ 16 t bytes)

    // ---- weight streaming ----
    const __amdgpu_buffer_rsrc_t rp = mk_rsrc(p.Wproj, (long)C * C * 2);
    const __amdgpu_buffer_rsrc_t rbf = mk_rsrc(p.bias_frag, (long)NH * AB_FRAG * 4);
    // Every wave issues exactly PPQ + PPP pieces per step and no memory operation of the loop sits under a branch: the step's
    // closing wait counts operations, and hipcc's own wait insertion (for the register loads) is exact only on branch-free code.
    // A piece beyond the slice reads through an out-of-range offset (zeros) into the dummy KiB.
    int voffq[AC::PPQ], voffp[AC::PPP];
#pragma unroll
    for (int i = 0; i < AC::PPQ; ++i) {
        const int piece = wave + NW * i;
        voffq[i] = piece < 3 * Cf::PA ? Cf::voff_a(piece % Cf::PA, lane)
                                      : (piece == 3 * Cf::PA ? ((lane >> 3) < 3 ? (lane >> 3) : 2) * C * 4 + (lane & 7) * 16 : (int)AB_OOB);
    }
#pragma unroll
    for (int i = 0; i < AC::PPP; ++i) {
        const int piece = wave + NW * i;
        const int pp = piece * 64 + lane;
        const int r = pp >> 2, u = pp & 3;
        voffp[i] = piece < Cf::PB ? (r * C + Cf::swb(u, r) * 8) * 2 : (int)AB_OOB;
    }
    auto issue_q = [&](int h, int buf) {  // q | k | v rows + biases of head h -> ring slot buf
#pragma unroll
        for (int i = 0; i < AC::PPQ; ++i) {
            const int piece = wave + NW * i;  // wave-uniform
            const int part = piece / Cf::PA;  // 0..2: q | k | v rows, 3: the bias piece (another descriptor, same instruction)
            const bool isb = piece == 3 * Cf::PA;
            const int dst = piece < AC::NPQ ? AC::OFF_Q + buf * AC::QBUF + piece * 1024 : AC::OFF_DUMMY;
            const __amdgpu_buffer_rsrc_t r = mk_rsrc(isb ? (const void*)p.bqkv : (const void*)p.Wqkv, isb ? 3L * C * 4 : 3L * C * C * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(smem + dst), 16, voffq[i], isb ? h * 32 * 4 : (part * C + h * 32) * C * 2, 0, 0);
        }
    };
    auto issue_p = [&](int h, int buf) {  // columns 32h .. 32h+31 of Wproj -> ring slot buf
#pragma unroll
        for (int i = 0; i < AC::PPP; ++i) {
            const int piece = wave + NW * i;
            const int dst = piece < Cf::PB ? AC::OFF_P + buf * AC::PBUF + piece * 1024 : AC::OFF_DUMMY;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, (lds_void*)(smem + dst), 16, voffp[i], h * 32 * 2, 0, 0);
        }
    };

    // ---- window schedule ----
    const int groups = (p.Bw + NWIN - 1) / NWIN;
    const int iters = (groups + gridDim.x - 1) / gridDim.x;
    const int steps = iters * NH;
    const bool masked = p.region_ids != nullptr;
    const __amdgpu_buffer_rsrc_t rx = mk_rsrc(p.x, p.rows * C * 4), ry = mk_rsrc(p.y, p.rows * C * 4);
    const __amdgpu_buffer_rsrc_t rrs = mk_rsrc(p.rowscale, p.rowscale ? p.rows * 4 : 0);
    __amdgpu_buffer_rsrc_t rxw = rrs, rqkv = rrs, rao = rrs, rmean = rrs, rrstd = rrs;
    if constexpr (SAVE) {
        rxw = mk_rsrc(p.xw, p.rows * C * 2);
        rqkv = mk_rsrc(p.qkv, p.rows * 3L * C * 2);
        rao = mk_rsrc(p.ao, p.rows * C * 2);
        rmean = mk_rsrc(p.mean, p.rows * 4);
        rrstd = mk_rsrc(p.rstd, p.rows * 4);
    }

    // window `it` of this wave: two unconditional loads (clamped indices, see above) whose results are only LOOKED AT a window
    // later (map_row / map_reg) -- a use right behind the loads would make hipcc wait for them and, in-order, for the weight
    // slices requested just before
    const int* regs_or_map = masked ? p.region_ids : p.win2tok;
    const float inv_nw = 1.0f / (float)p.nW;
    auto issue_map = [&](int it, int& tok, int& rg, int& base) {
        const int bw = (it * (int)gridDim.x + (int)blockIdx.x) * NWIN + wj;  // (wave-uniform)
        const bool act = it < iters && bw < p.Bw;
        // image / window of bw without an integer division (bw < 2^22, checked by the host): a float quotient, corrected by one
        int img = (int)(((float)bw + 0.5f) * inv_nw);
        int wi = bw - img * p.nW;
        if (wi < 0) { wi += p.nW; --img; }
        if (wi >= p.nW) { wi -= p.nW; ++img; }
        if (!act) wi = 0;
        tok = p.win2tok[(long)wi * p.N + (slot < p.N ? slot : 0)];
        rg = regs_or_map[(long)wi * p.N + (lane < p.N ? lane : 0)];
        base = act ? img * p.L : -1;
    };
    auto map_row = [&](int tok, int base) -> int { return (base >= 0 && slot < p.N && tok >= 0) ? base + tok : -1; };  // row of this lane's slot or -1
    auto map_reg = [&](int rg, int base) -> int { return (masked && base >= 0 && lane < p.N) ? rg : -1; };             // region id of slot `lane` or -1

    f32x4 xn[NXL];   // the rows of the NEXT window (landing), tile mt <-> channels 16 mt + 4g .. +3
    float rs_n = 1.f;
    int row_cur = -1, row_nxt, reg_cur = -1, reg_nxt, tok_nn, rg_nn, base_nn;
    float rs_cur = 1.f;
    unsigned mbits = 0;  // bit 4i + r: key 16i + 4g + r lies in another shift region than query `slot`

    auto xoff = [&](int row, int i) -> unsigned { return row_off(row, (unsigned)row * (C * 4) + (16 * i + 4 * g) * 4); };

    bf16x8 xb[KS];
    f32x4 accy[MT];
    bf16x8 qf = {}, qf_n;
    // fragment-order bias of every head (16 keys x this lane's query each), resident: a load inside the loop would have to be
    // waited for within its step and, vmcnt retiring in order, would drag the HBM row prefetch issued before it along
    f32x4 bias_r[NH][4];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            bias_r[h][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbf, (w * 64 + lane) * 16, (h * AB_FRAG + i * 1024) * 4, 0)) * LOG2E;
    u32x4 sv_q = {}, sv_k = {}, sv_v = {};  // side-output rows of the step (SAVE)

    // ---- prologue: first slice, first window's rows ----
    issue_q(0, 0);
    issue_map(0, tok_nn, rg_nn, base_nn);
    row_nxt = map_row(tok_nn, base_nn);
    reg_nxt = map_reg(rg_nn, base_nn);
    issue_map(1, tok_nn, rg_nn, base_nn);
#pragma unroll
    for (int i = 0; i < NXL; ++i) xn[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff(row_nxt, i), 0, 0));
    const bool has_rs = p.rowscale != nullptr;
    rs_n = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, row_off(row_nxt, (unsigned)row_nxt * 4), 0, 0));
    wait_vm<0>();
    __syncthreads();

    // the attention + projection half of a step: pair (window of `row_cur`, head hp); kvo: LDS offset of the window's K image of
    // that pair, po: of the pair's Wproj columns; hn: the head whose bias fragment is requested for the next step
    auto attend = [&](auto Hp, int kvo, int po, u32x4& sv_o) {
        constexpr int hp = decltype(Hp)::value;
        f32x4 pr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(smem + kvo + kr + 1024 * i);
            pr[i] = mfma16(kf, qf, bias_r[hp][i]);  // (the bias rides in as the C operand: no copy, no add)
        }
        if (mbits) {  // shift mask: -100 where key and query lie in different regions (only windows that straddle a region border)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) pr[i][r] += ((mbits >> (4 * i + r)) & 1u) ? -100.f * LOG2E : 0.f;
        }
        TL(2);
        // (the fragments of the two products behind the softmax are requested now: their LDS latency hides behind its arithmetic)
        u32x2 vf[2][2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const char* a0 = smem + kvo + K_BYTES + vr + 16 * VT_LD * dt + 64 * ks;
                vf[ks][dt][0] = *reinterpret_cast<const u32x2*>(a0);
                vf[ks][dt][1] = *reinterpret_cast<const u32x2*>(a0 + 32);
            }
        // softmax over the 64 keys in the base-2 domain (q and the bias carry log2(e)): one v_exp_f32 per score; P stays
        // unnormalised (<= 1) and 1 / sum rides in the scale of O
        float m = -3.0e38f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, pr[i][r]);
        m = tok_max(m);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(pr[i][r] - m);
                pr[i][r] = e;
                sum += e;
            }
        sum = tok_sum(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
        TL(3);
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 pf;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pf[e] = (bf16)pr[2 * ks][e];
                pf[4 + e] = (bf16)pr[2 * ks + 1][e];
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const u32x2 lo = vf[ks][dt][0], hi = vf[ks][dt][1];
                o[dt] = mfma16(__builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]}), pf, o[dt]);
            }
        }
        TL(4);
        if constexpr (SAVE) {  // attention output row, channels 32 hp + (d = 4g + e | 16 + 4g + e) -> 8 consecutive after the row swap
            unsigned x0 = pack2(o[0][0] * inv, o[0][1] * inv), x1 = pack2(o[0][2] * inv, o[0][3] * inv);
            unsigned y0 = pack2(o[1][0] * inv, o[1][1] * inv), y1 = pack2(o[1][2] * inv, o[1][3] * inv);
            row_swap(x0, y0);
            row_swap(x1, y1);
            sv_o = u32x4{x0, x1, y0, y1};
        }
        const float osc = inv * rs_cur;
        bf16x8 of;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            of[e] = (bf16)(o[0][e] * osc);
            of[4 + e] = (bf16)(o[1][e] * osc);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(smem + po + fb + 1024 * mt);
            accy[mt] = mfma16(a, of, accy[mt]);
        }
        TL(5);
    };
    // uniform ring offsets of a step, opaque to the optimiser (it would otherwise keep one set of lane addresses per ring slot)
    auto ring = [&](int slot, int par, int& qo, int& po, int& kvo) {
        qo = AC::OFF_Q + slot * AC::QBUF;
        po = AC::OFF_P + slot * AC::PBUF;
        kvo = AC::OFF_KV + (wj * 2 + par) * KV_BYTES;
        asm volatile("" : "+s"(qo), "+s"(po), "+s"(kvo));
    };

    for (int it = 0; it < iters; ++it) {
        static_for<0, NH>([&](auto Hc) {
            constexpr int h = decltype(Hc)::value;
            constexpr int hprev = h == 0 ? NH - 1 : h - 1, hnext = h + 1 == NH ? 0 : h + 1;
            const int s = it * NH + h;
            const int par = s & 1;
            const int rs0 = AC::RES ? h : par, rs1 = AC::RES ? hprev : (par ^ 1), rsn = AC::RES ? hnext : (par ^ 1);  // ring slots: this / previous / next pair
            int qo, po, kvo, qo1, po1, kvo1;
            ring(rs0, par, qo, po, kvo);          // this pair: q | k | v slice read, K / V images written, Wproj columns requested
            ring(rs1, par ^ 1, qo1, po1, kvo1);   // the previous pair: K / V images and Wproj columns read
            // ---- top: the map of window it + 2, the next slices, then rows of the next window.  The row loads come from HBM and are
            // the youngest loads of the step: the closing wait lets them (and the stores) fly on, so they have until the next
            // step's close -- issued at the end of a step they would have to land within ONE step, behind which every step waited ----
            TL(0);
            if constexpr (h == 1) issue_map(it + 2, tok_nn, rg_nn, base_nn);  // (the previous answer was taken at h == 0)
            if (!AC::RES || it == 0) {
                if (!AC::RES || h + 1 < NH) issue_q(hnext, rsn);  // (past the last pair: a slice nobody reads)
                issue_p(h, rs0);
            }
            constexpr int pf0 = h == 0 ? NXL : (h - 1) * PF, pf1 = h == 0 ? NXL : (h * PF < NXL ? h * PF : NXL);
            static_for<pf0, (pf0 < pf1 ? pf1 : pf0)>([&](auto Ic) {
                constexpr int i = decltype(Ic)::value;
                xn[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff(row_nxt, i), 0, 0));
            });
            if constexpr (h == 1) rs_n = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, row_off(row_nxt, (unsigned)row_nxt * 4), 0, 0));
            mem_fence_compiler();
            TL(1);

            // The step's three pieces.  A: attention + projection of the previous pair (and, at h == 0, the finished window's rows
            // out + the new window's accumulator start).  N (h == 0 only): LayerNorm of the new window's rows.  Q: q, k, v of this
            // pair.  The window slots of a workgroup run them in opposite orders (even: A N Q, odd: N Q A): the two waves of a SIMD
            // belong to different slots, so one is in its MFMA / LDS-bound piece while the other runs the VALU-bound softmax.
            u32x4 sv_o = {};
            const bool have_prev = s > 0;
            const int row_prev = have_prev ? row_cur : -1;  // the previous pair's row (row_cur changes in piece A at h == 0)
            int row_q = row_cur;                            // this pair's row (side outputs)
            if constexpr (h == 0) row_q = row_nxt;

            auto piece_A = [&]() {
                attend(IC<hprev>{}, kvo1, po1, sv_o);  // (s == 0: on images nobody wrote; nothing of it is kept)
                AB_SCHED_BARRIER();
                mem_fence_compiler();
                if constexpr (SAVE) {
                    const unsigned vo = row_off(row_prev, (unsigned)row_prev * (C * 2) + (32 * hprev + 16 * (g & 1) + 4 * (g & ~1)) * 2);
                    buffer_store_b128(sv_o, rao, vo, 0);
                }
                if constexpr (h == 0) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const unsigned vo = row_off(row_prev, (unsigned)row_prev * (C * 4) + (16 * mt + 4 * g) * 4);
                        buffer_store_b128(accy[mt], ry, vo, 0);
                    }
                    // the new window takes over: accumulator start x + rowscale * b_proj, DropPath factor, shift-mask bits
                    row_cur = row_nxt;
                    reg_cur = reg_nxt;
                    rs_cur = has_rs ? rs_n : 1.f;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const f32x4 bp = *reinterpret_cast<const f32x4*>(smem + lc + (2 * C + 16 * mt) * 4);
                        accy[mt] = xn[mt] + rs_cur * bp;
                    }
                    mbits = 0;
                    if (masked) {
                        const int rq_ = __shfl(reg_cur, slot, 64);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) mbits |= (__shfl(reg_cur, 16 * i + 4 * g + r, 64) != rq_ ? 1u : 0u) << (4 * i + r);
                    }
                    row_nxt = map_row(tok_nn, base_nn);
                    reg_nxt = map_reg(rg_nn, base_nn);
                }
                AB_SCHED_BARRIER();
            };
            auto piece_N = [&]() {
                if constexpr (h == 0) {
                    float mean, rstd;
                    {
                        float s1 = 0.f;
#pragma unroll
                        for (int i = 0; i < NXL; ++i) s1 += xn[i][0] + xn[i][1] + xn[i][2] + xn[i][3];
                        mean = tok_sum(s1) * (1.f / C);
                        float s2 = 0.f;
#pragma unroll
                        for (int i = 0; i < NXL; ++i)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float d = xn[i][e] - mean;
                                s2 += d * d;
                            }
                        rstd = rsqrtf(tok_sum(s2) * (1.f / C) + p.eps);
                    }
                    const float live = row_q >= 0 ? 1.f : 0.f;  // zero-pad and idle slots: LayerNorm output 0 (the reference pads after norm1)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        float hv[8];
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const f32x4 gm = *reinterpret_cast<const f32x4*>(smem + lc + (32 * ks + 16 * t) * 4);
                            const f32x4 bt = *reinterpret_cast<const f32x4*>(smem + lc + (C + 32 * ks + 16 * t) * 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                hv[4 * t + e] = live * ((xn[2 * ks + t][e] - mean) * rstd * gm[e] + bt[e]);
                                xb[ks][4 * t + e] = (bf16)hv[4 * t + e];
                            }
                        }
                        if constexpr (SAVE) {
                            unsigned x0 = pack2(hv[0], hv[1]), x1 = pack2(hv[2], hv[3]), y0 = pack2(hv[4], hv[5]), y1 = pack2(hv[6], hv[7]);
                            row_swap(x0, y0);
                            row_swap(x1, y1);
                            const unsigned vo = row_off(row_q, (unsigned)row_q * (C * 2) + (32 * ks + 16 * (g & 1) + 4 * (g & ~1)) * 2);
                            buffer_store_b128(u32x4{x0, x1, y0, y1}, rxw, vo, 0);
                        }
                    }
                    if constexpr (SAVE) {
                        const unsigned vo = row_off(g == 0 ? row_q : -1, (unsigned)row_q * 4);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mean), rmean, vo, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rstd), rrstd, vo, 0, 0);
                    }
                    AB_SCHED_BARRIER();
                }
            };
            auto piece_Q = [&]() {
                const float* sb = reinterpret_cast<const float*>(smem + qo + 3 * Cf::A_BYTES);
                // software pipeline over the three parts: the weight fragments of part p + 1 are requested before the MFMAs of part p
                bf16x8 wf[2][2 * KS];
                auto load_part = [&](int part, bf16x8 (&dst)[2 * KS]) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        dst[2 * ks] = *reinterpret_cast<const bf16x8*>(smem + qo + fa(0, ks) + part * Cf::A_BYTES);
                        dst[2 * ks + 1] = *reinterpret_cast<const bf16x8*>(smem + qo + fa(1, ks) + part * Cf::A_BYTES);
                    }
                };
                load_part(0, wf[0]);
                static_for<0, 3>([&](auto Pc) {
                    constexpr int part = decltype(Pc)::value;
                    constexpr bool TRANSPOSED = part < 2 || SAVE;  // q, k (and the side-output copy of v): [32 channels][16 slots]
                    if constexpr (part < 2) load_part(part + 1, wf[(part + 1) & 1]);
                    const bf16x8 (&wc)[2 * KS] = wf[part & 1];
                    // (biases as the accumulators' initial values: channels 8g .. 8g+7 of the part for the transposed tiles, one
                    // channel per lane for the V tiles)
                    f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0, v0 = a0, v1 = a0;
                    if constexpr (TRANSPOSED) {
                        a0 = *reinterpret_cast<const f32x4*>(sb + 32 * part + 8 * g);
                        a1 = *reinterpret_cast<const f32x4*>(sb + 32 * part + 8 * g + 4);
                    }
                    if constexpr (part == 2) {
                        const float bv0 = *reinterpret_cast<const float*>(smem + qo + 3 * Cf::A_BYTES + 64 * 4 + vb);
                        const float bv1 = *reinterpret_cast<const float*>(smem + qo + 3 * Cf::A_BYTES + 64 * 4 + vb + 16);
                        v0 = f32x4{bv0, bv0, bv0, bv0};
                        v1 = f32x4{bv1, bv1, bv1, bv1};
                    }
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        if constexpr (TRANSPOSED) {
                            a0 = mfma16(wc[2 * ks], xb[ks], a0);
                            a1 = mfma16(wc[2 * ks + 1], xb[ks], a1);
                        }
                        if constexpr (part == 2) {  // V [16 slots][32 channels]: the same fragments with the operands exchanged
                            v0 = mfma16(xb[ks], wc[2 * ks], v0);
                            v1 = mfma16(xb[ks], wc[2 * ks + 1], v1);
                        }
                    }
                    u32x4 pk = {};
                    if constexpr (TRANSPOSED && (part != 0 || SAVE)) pk = u32x4{pack2(a0[0], a0[1]), pack2(a0[2], a0[3]), pack2(a1[0], a1[1]), pack2(a1[2], a1[3])};
                    if constexpr (part == 0) {
                        const float qs = p.scale * LOG2E;  // scores in the base-2 domain
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            qf_n[e] = (bf16)(a0[e] * qs);
                            qf_n[4 + e] = (bf16)(a1[e] * qs);
                        }
                        if constexpr (SAVE) sv_q = pk;
                    } else if constexpr (part == 1) {
                        *reinterpret_cast<u32x4*>(smem + kvo + kw) = pk;
                        if constexpr (SAVE) sv_k = pk;
                    } else {
                        *reinterpret_cast<u32x2*>(smem + kvo + K_BYTES + vw) = u32x2{pack2(v0[0], v0[1]), pack2(v0[2], v0[3])};
                        *reinterpret_cast<u32x2*>(smem + kvo + K_BYTES + vw + 4 * VT_LD) = u32x2{pack2(v1[0], v1[1]), pack2(v1[2], v1[3])};
                        if constexpr (SAVE) sv_v = pk;
                    }
                });
                mem_fence_compiler();
                if constexpr (SAVE) {  // side-output rows of this pair
                    const unsigned vo = row_off(row_q, (unsigned)row_q * (3 * C * 2) + (32 * h + 8 * g) * 2);
                    buffer_store_b128(sv_q, rqkv, vo, 0);
                    buffer_store_b128(sv_k, rqkv, vo, C * 2);
                    buffer_store_b128(sv_v, rqkv, vo, 2 * C * 2);
                }
            };
            if (wj & 1) {
                piece_N();
                TL(6);
                piece_Q();
                TL(7);
                piece_A();
            } else {
                piece_A();
                piece_N();
                TL(6);
                piece_Q();
                TL(7);
            }
            // everything older than the row loads of this step has landed (the weight slices above all); the row loads and the
            // stores stay in flight through the next step.  Resident weights: nothing to wait for after the first window
            constexpr int NPFL = (pf0 < pf1 ? pf1 - pf0 : 0);
            // (only the row LOADS count: loads retire in order among themselves, stores retire out of order with respect to loads, so the
            // allowance of rounds 5-6 -- row loads + this step's output / side-output stores -- could be met with a weight slice still on its
            // way; profiles/r06_gemm_astat_probe.txt.  Same step time.)
            constexpr int LATE = NPFL;
            if (!AC::RES || it == 0) wait_vm<LATE>();
            TL(8);
            chunk_barrier();
            TL(9);
#ifdef ESVIT_AB_TIMELINE
            ++tl_step;
#endif
            qf = qf_n;
        });
    }
    // ---- drain: the last pair's attention, the last window's rows ----
    {
        int qo, po, kvo;
        ring(AC::RES ? NH - 1 : ((steps & 1) ^ 1), (steps & 1) ^ 1, qo, po, kvo);
        u32x4 sv_o = {};
        attend(IC<NH - 1>{}, kvo, po, sv_o);
        if constexpr (SAVE) {
            const unsigned vo = row_off(row_cur, (unsigned)row_cur * (C * 2) + (32 * (NH - 1) + 16 * (g & 1) + 4 * (g & ~1)) * 2);
            buffer_store_b128(sv_o, rao, vo, 0);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const unsigned vo = row_off(row_cur, (unsigned)row_cur * (C * 4) + (16 * mt + 4 * g) * 4);
            buffer_store_b128(accy[mt], ry, vo, 0);
        }
    }
#ifdef ESVIT_AB_TIMELINE
    __syncthreads();
    if (p.timeline && blockIdx.x < 2)
        for (int i = threadIdx.x; i < AB_TL_BYTES / 4; i += NW * 64) p.timeline[blockIdx.x * (AB_TL_BYTES / 4) + i] = tl_lds[i];
#endif
}

template <int C, int NWIN, bool SAVE>
__global__ __launch_bounds__(4 * NWIN * 64, (ABCfg<C, NWIN>::LDS * 2 <= 160 * 1024 ? 2 : 1)) void attn_branch_fwd_kernel(const ABParams p) {
    attn_branch_fwd_body<C, NWIN, SAVE>(p);
}

template <int C, int NWIN, bool SAVE>
int launch_ab(const ABParams& prm, hipStream_t stream) {
    using AC = ABCfg<C, NWIN>;
    auto k = attn_branch_fwd_kernel<C, NWIN, SAVE>;
    static unsigned long long lds_set = 0;
    esvit_raise_lds(k, AC::LDS, lds_set);
    const int groups = (prm.Bw + NWIN - 1) / NWIN;
    const int per_cu = AC::LDS * 2 <= 160 * 1024 ? 2 : 1;  // persistent workgroups: as many as the LDS of the 256 CUs holds
    const int grid = groups < 256 * per_cu ? groups : 256 * per_cu;
    hipLaunchKernelGGL(k, dim3(grid), dim3(AC::NW * 64), AC::LDS, stream, prm);
    return ESVIT_OK;
}

}  // namespace

int esvit_i_fill_bias_frag(const float* rel_table, int ws, int N, int nH, float* bias_frag_ws, hipStream_t stream);

#ifdef ESVIT_AB_TIMELINE
static unsigned* g_ab_timeline = nullptr;
extern "C" __attribute__((visibility("default"))) void esvit_attn_branch_timeline(void* buf) { g_ab_timeline = (unsigned*)buf; }
#endif

extern "C" int esvit_attn_branch_fwd(int dtype, const float* x, const float* gamma, const float* beta, float eps, const void* Wqkv_p,
                                     const float* bqkv, const void* Wproj_p, const float* bproj, const int32_t* win2tok, int L,
                                     const float* rel_table, int ws, float* bias_frag_ws, const int32_t* region_ids, int nW, int nB, int N,
                                     int nH, float scale, const float* rowscale, float* y, void* xw, void* qkv, void* ao, float* mean,
                                     float* rstd, esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(dtype == ESVIT_BF16, "esvit_attn_branch_fwd: bf16 activations only");
    ESVIT_CHECK_ARG(x && gamma && beta && Wqkv_p && bqkv && Wproj_p && bproj && win2tok && bias_frag_ws && y && L > 0 && nW > 0 && nB > 0,
                    "esvit_attn_branch_fwd: bad arguments");
    ESVIT_CHECK_ARG((nH == 3 || nH == 6) && N > 0 && N <= 64 && N == ws * ws, "esvit_attn_branch_fwd: C = 32 nH in {96, 192}, windows of <= 64 tokens (nH=%d N=%d)", nH, N);
    const bool save = xw || qkv || ao || mean || rstd;
    ESVIT_CHECK_ARG(!save || (xw && qkv && ao && mean && rstd), "esvit_attn_branch_fwd: the side outputs come all together or not at all");
    const int C = 32 * nH;
    const long rows = (long)nB * L;
    ESVIT_CHECK_ARG(rows * 3 * C * 2 < 0x7fff0000L && rows * C * 4 < 0x7fff0000L, "esvit_attn_branch_fwd: the rows of one call must fit 2 GiB buffer ranges");
    ESVIT_CHECK_ARG((long)nB * nW < (1L << 22), "esvit_attn_branch_fwd: at most 2^22 windows per call");
    ESVIT_CHECK_ARG((((uintptr_t)x | (uintptr_t)y | (uintptr_t)Wqkv_p | (uintptr_t)Wproj_p | (uintptr_t)xw | (uintptr_t)qkv | (uintptr_t)ao | (uintptr_t)bias_frag_ws) & 15) == 0,
                    "esvit_attn_branch_fwd: x, y, the weights, the bias fragments and the side outputs are read / written in 16-byte pieces: align them");
    if (rel_table) {
        int rc = esvit_i_fill_bias_frag(rel_table, ws, N, nH, bias_frag_ws, stream);
        if (rc != ESVIT_OK) return rc;
    }
    ABParams prm;
    prm.x = x; prm.gamma = gamma; prm.beta = beta; prm.eps = eps;
    prm.Wqkv = (const bf16*)Wqkv_p; prm.bqkv = bqkv; prm.Wproj = (const bf16*)Wproj_p; prm.bproj = bproj;
    prm.bias_frag = bias_frag_ws; prm.win2tok = win2tok; prm.region_ids = region_ids; prm.rowscale = rowscale;
    prm.y = y; prm.xw = (bf16*)xw; prm.qkv = (bf16*)qkv; prm.ao = (bf16*)ao; prm.mean = mean; prm.rstd = rstd;
#ifdef ESVIT_AB_TIMELINE
    prm.timeline = g_ab_timeline;
#else
    prm.timeline = nullptr;
#endif
    prm.nW = nW; prm.Bw = nB * nW; prm.N = N; prm.L = L; prm.rows = rows; prm.scale = scale;
    int rc;
    // windows per workgroup, measured on the B = 128 row counts (tools/bench_attn_branch.py): C = 96: two (eight waves, two per SIMD, 216-233
    // registers; one: 303 us, three: 283 us against 227 us on the 224-crop rows); C = 192: one (four waves at 472-487 registers: the q | k | v
    // weights of a head alone are 37 KB of fragments, two windows per workgroup spill)
    if (nH == 3) rc = save ? launch_ab<96, 2, true>(prm, stream) : launch_ab<96, 2, false>(prm, stream);
    else rc = save ? launch_ab<192, 1, true>(prm, stream) : launch_ab<192, 1, false>(prm, stream);
    if (rc != ESVIT_OK) return rc;
    ESVIT_CHECK_LAUNCH("esvit_attn_branch_fwd");
    return ESVIT_OK;
}
