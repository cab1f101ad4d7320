// Window attention for up to 224 tokens per window: the 14x14 windows (N = 196, head_dim 32) of the W=14 configurations of the reference
// (swin_*_patch4_window14_224.yaml) and, as the head_dim-64 instances, the 197 tokens of a 224^2 crop of the
// monolithic ViTs (one "window" per image, zero bias table).  Same mathematics and token-ordered I/O as window_attn.hip,
// but a 196x196 score tile does not fit one wave's registers, so the work is blocked flash-style:
//
//   forward        one workgroup per (window, head): K, V staged once in LDS; every wave owns 32-query blocks, forms
//                  S^T = K (scale Q)^T for its block (14 x 2 MFMA tiles), softmax in registers, writes P to LDS and
//                  multiplies by V (transpose read).  The per-query log-sum-exp is saved for the backward.
//   backward dQ    wave <-> fixed (head, query block), looping over windows so the relative-position-bias gradient of
//                  its 32 x 196 strip accumulates in registers (no atomics); dP^T = V dO^T, dS, dQ = scale dS K.
//   backward dK,dV one workgroup per (window, head): scale*Q and dO staged once; every wave owns 32-key blocks, rebuilds
//                  P^T from the saved log-sum-exp, delta = rowsum(dO o O), dV = P^T dO, dK = dS^T (scale Q).
//
// Tokens are padded to 224 = 14 MFMA tiles (keys >= 196 get -1e30, never any probability).  Small per-window tables live
// in LDS: slot->token map and the shift-mask region label of every slot.  The relative-position bias arrives precomputed
// in MFMA fragment order (relpos_bias_frag_big_kernel), one 16-byte load per lane per 16x16 score tile.
#include "common.h"
#include "mfma.h"
#include "../../include/esvit_hip.h"

namespace {

constexpr int NT = 14;         // 16-wide tiles per window side
constexpr int NPB = NT * 16;   // 224 padded tokens
constexpr int NQB = NPB / 32;  // 7 blocks of 32 queries / keys
constexpr int TAB_FLOATS = 768;
typedef int i32x4 __attribute__((ext_vector_type(4)));

// HD = head_dim: 32 (the Swin W=14 models) or 64 (the 197-token crops of the monolithic ViTs, bf16 only: two fp32 [224][68] images
// plus the per-wave images exceed a CU's LDS)
template <typename T, int HD>
struct BigCfg {
    static constexpr int VEC = ElemTraits<T>::VEC;
    static constexpr int LDQ = HD + VEC;    // [*][LDQ] images with d contiguous
    static constexpr int LDP = NPB + VEC;   // [32][LDP] P / dS image of one query block
    // waves of the dK/dV kernel.  fp32 parity mode: half the waves, same LDS budget; head_dim 64: one wave per 32-key block (7), the
    // workgroup is alone on its CU anyway (two [224][72] images)
    static constexpr int WAVES = sizeof(T) == 2 ? (HD == 32 ? 4 : 7) : 2;
    static constexpr int FULL = NPB * LDQ;  // one [224][LDQ] image
    static constexpr int BLK = 32 * LDQ;    // one [32][LDQ] image
    static constexpr int PIMG = 32 * LDP;
    static constexpr int TABLE_BYTES = TAB_FLOATS * 4 + 2 * NPB * 4 + 2 * NPB * 4;  // tab, tok, packed, lse, delta
};

struct BigTables {
    float* tab;
    int* tok;
    int* pk;      // a(t) | region << 16
    float* lse;
    float* delta;
};
__device__ __forceinline__ BigTables carve_tables(char* p) {
    BigTables t;
    t.tab = reinterpret_cast<float*>(p);
    t.tok = reinterpret_cast<int*>(p + TAB_FLOATS * 4);
    t.pk = t.tok + NPB;
    t.lse = reinterpret_cast<float*>(t.pk + NPB);
    t.delta = t.lse + NPB;
    return t;
}

// per-window tables (all threads of the block)
__device__ __forceinline__ void load_window_tables(const BigTables& tb, const int* __restrict__ win2tok, const int* __restrict__ region_ids,
                                                   int w, int N, int ws, bool active) {
    const int w2 = 2 * ws - 1;
    for (int t = threadIdx.x; t < NPB; t += blockDim.x) {
        int tok = -1, pk = 0;
        if (t < N) {
            if (active) tok = win2tok[(long)w * N + t];
            const int reg = region_ids ? region_ids[(long)w * N + t] : 0;
            pk = ((t / ws) * w2 + t % ws) | (reg << 16);
        }
        tb.tok[t] = tok;
        tb.pk[t] = pk;
    }
}

// which of the window's NT query tiles hold a token at all (bit j: some slot 16 j .. 16 j + 15 is live) -- wave-uniform, from the
// slot -> token table in LDS.  The windows of the 96^2 crops are mostly padding from stage 1 on (a 6 x 6 map in a 14 x 14 window: 36
// live slots in tiles 0..4 of 14): a pad-slot QUERY produces nothing (its output row is cropped away, its dO row is zero, so its dS
// row is zero), only pad-slot KEYS take part (swin_transformer.py:292-300 pads after norm1: their k, v are the qkv bias).  The three
// kernels skip query tiles without a live slot; the results are the same to the bit (the skipped products add zeros).
__device__ __forceinline__ unsigned live_query_tiles(const int* tok_lds, int lane) {
    bool l = false;
    if (lane < NPB / 4) {
        const i32x4 t = *reinterpret_cast<const i32x4*>(tok_lds + 4 * lane);
        l = (t[0] & t[1] & t[2] & t[3]) >= 0;  // some token index is non-negative
    }
    const unsigned long long b = __ballot(l);
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) m |= (((b >> (4 * j)) & 0xfull) != 0 ? 1u : 0u) << j;
    return m;
}

// stage NROWS window slots (first slot s0) of a token-ordered matrix into a [NROWS][LDQ] image; NTHR threads cooperate.
// Two phases: every global load of the thread is issued before the first LDS store, so a thread waits ONE memory round
// trip per call instead of one per 16-byte piece (the first version looped load -> store and paid 4-7 serial round trips
// per (window, head) with one wave per SIMD to hide them).
template <typename T, int NROWS, int NTHR, int HD>
struct SlotStage {
    static constexpr int VEC = BigCfg<T, HD>::VEC, LDQ = BigCfg<T, HD>::LDQ, VPR = HD / VEC;
    static constexpr int ITERS = (NROWS * VPR + NTHR - 1) / NTHR;
    Vec16<T> x[ITERS];

    // the 16-byte piece a pad slot holds in this thread's column of the image: the qkv bias of the head (swin_transformer.py:292-300 pads
    // after norm1, so a pad token's q, k, v are the bias).  NTHR is a multiple of the pieces per row, so a thread stages the same
    // column piece in every iteration and for every window: loaded ONCE per kernel (the first version read it element by element for
    // every pad slot -- eight scalar loads per piece, and four of five slots of a 96^2 crop's 14 x 14 window are pad slots).
    static_assert(NTHR % VPR == 0, "a thread keeps its column piece");
    static __device__ __forceinline__ Vec16<T> pad_piece(const float* __restrict__ pad, int tid) {
        Vec16<T> r;
#pragma unroll
        for (int e = 0; e < Vec16<T>::N; ++e) r.set(e, pad[(tid % VPR) * VEC + e]);
        return r;
    }
    __device__ __forceinline__ void load(const T* __restrict__ g, long row_stride, const int* tok_lds, long tok_base, int s0, int N,
                                         const Vec16<T>* padv, int tid) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int v = tid + it * NTHR;
            const int rl = v / VPR, dv = v % VPR;
            const int t = s0 + rl;
            x[it] = zero16<T>();
            if (v < NROWS * VPR && t < N) {
                const int tok = tok_lds[t];
                if (tok >= 0) x[it] = ld16<T>(g + (tok_base + tok) * row_stride + dv * VEC);
                else if (padv) x[it] = *padv;
            }
        }
    }
    __device__ __forceinline__ void store(T* lds, float scale, int tid) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int v = tid + it * NTHR;
            if (v >= NROWS * VPR) continue;
            const int rl = v / VPR, dv = v % VPR;
            Vec16<T> y = x[it];
            if (scale != 1.f) {
#pragma unroll
                for (int e = 0; e < Vec16<T>::N; ++e) y.set(e, y.get(e) * scale);
            }
            st16<T>(lds + rl * LDQ + dv * VEC, y);
        }
    }
};

// bias_frag[h][((ki*NT + qj)*64 + lane)*4 + r] = table[a(q) - a(key) + off][h] for q = 16qj + c, key = 16ki + 4g + r
// (0 for padded queries, -1e30 for padded keys): one 16-byte load per lane per 16x16 score tile replaces four LDS table
// gathers and ~40 VALU instructions -- the first version of these kernels spent 40-50 VALU instructions per MFMA on it.
__global__ void relpos_bias_frag_big_kernel(const float* __restrict__ table, int ws, int N, int nH, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int FE = NT * NT * 256;
    if (i >= 2L * nH * FE) return;
    const bool tr = i >= (long)nH * FE;  // second half: score tiles with rows = queries (the dK/dV kernel), see below
    const long i2 = tr ? i - (long)nH * FE : i;
    const int h = (int)(i2 / FE), e = (int)(i2 % FE);
    const int r = e & 3, lane = (e >> 2) & 63, f = e >> 8;
    const int c = lane & 15, g = lane >> 4;
    // first half:  tile f = ki*NT + qj holds S^T: row = key 16ki + 4g + r, column = query 16qj + c
    // second half: tile f = qj*NT + ki holds S:   row = query 16qj + 4g + r, column = key 16ki + c
    const int q = tr ? 16 * (f / NT) + 4 * g + r : 16 * (f % NT) + c;
    const int key = tr ? 16 * (f % NT) + c : 16 * (f / NT) + 4 * g + r;
    float v = 0.f;
    if (key >= N) v = -1.0e30f;
    else if (q < N) {
        const int w2 = 2 * ws - 1;
        v = table[(long)((q / ws - key / ws + ws - 1) * w2 + (q % ws - key % ws + ws - 1)) * nH + h];
    }
    out[i] = v;
}


// 32 result rows (window slots s0 .. s0 + 31) x HD from TRANSPOSED accumulators (acc[ti][j][r] = result[d = 16 j + 4g + r][slot s0 +
// 16 ti + c]: the producing MFMAs take their two operands exchanged, whose fragments have the same register layout): 16-byte row pieces
// straight from the registers (common.h: esvit_pack_tile_pair_bf16) instead of an LDS transpose with 2-byte scattered writes.  Pad-slot
// rows are summed (as stored) into padacc[4 HD / 16]: head channels 16 j + 4g + r of this lane's slots.
template <typename T, int HD>
__device__ __forceinline__ void store_block_rows_t(const f32x4 (&acc)[2][HD / 16], float mul, T* __restrict__ dst, long row_stride, const int* tok_lds,
                                                   long tok_base, int s0, int N, bool active, float* padacc, int c, int g) {
    constexpr int DT = HD / 16;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int t = s0 + 16 * ti + c;
        const bool live = active && t < N;  // (no branch around the lane exchange below)
        const int tok = live ? tok_lds[t] : -1;
        f32x4 v[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) {
            v[j] = acc[ti][j] * mul;
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[j][r] = (float)(bf16)v[j][r];
            }
        }
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int j = 0; j < DT; j += 2) {
                const esvit_u32x4 x = esvit_pack_tile_pair_bf16(v[j], v[j + 1]);  // (every lane takes part in the exchange)
                if (tok >= 0) *reinterpret_cast<esvit_u32x4*>(dst + (tok_base + tok) * row_stride + 16 * j + esvit_tile_pair_ch0(g)) = x;
            }
        } else {
#pragma unroll
            for (int j = 0; j < DT; ++j)
                if (tok >= 0) *reinterpret_cast<f32x4*>(dst + (tok_base + tok) * row_stride + 16 * j + 4 * g) = v[j];
        }
        if (live && tok < 0 && padacc) {
#pragma unroll
            for (int j = 0; j < DT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) padacc[4 * j + r] += v[j][r];
        }
    }
}

// one 16-slot tile (slots s0 .. s0 + 15) of result rows from transposed accumulators acc[j][r] = result[d = 16 j + 4g + r][slot s0 + c]
template <typename T, int HD>
__device__ __forceinline__ void store_tile_rows_t(const f32x4 (&acc)[HD / 16], float mul, T* __restrict__ dst, long row_stride, const int* tok_lds,
                                                  long tok_base, int s0, int N, bool active, int c, int g) {
    constexpr int DT = HD / 16;
    const int t = s0 + c;
    const bool live = active && t < N;
    const int tok = live ? tok_lds[t] : -1;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < DT; j += 2) {
            const esvit_u32x4 x = esvit_pack_tile_pair_bf16(acc[j] * mul, acc[j + 1] * mul);
            if (tok >= 0) *reinterpret_cast<esvit_u32x4*>(dst + (tok_base + tok) * row_stride + 16 * j + esvit_tile_pair_ch0(g)) = x;
        }
    } else {
#pragma unroll
        for (int j = 0; j < DT; ++j)
            if (tok >= 0) *reinterpret_cast<f32x4*>(dst + (tok_base + tok) * row_stride + 16 * j + 4 * g) = acc[j] * mul;
    }
}

// -------------------------------------------------------------------------------------------------------------
// forward, second generation (default).  Same mathematics; what changed is where the data waits:
//   * P never goes through LDS.  The S^T tiles leave lane (c, g) with, for query c, the keys 16i + 4g + r -- for a 32-key
//     chunk that is 8 keys {32ks + 4g + e, 32ks + 16 + 4g + e}.  An MFMA only needs A and B to agree on which key sits in
//     which k-slot, so those 8 values ARE the A fragment of P V if V's B fragment is read with the same key permutation
//     (frag_v_perm: two transpose reads 16 key rows apart).  That removes 56 LDS stores + 14 LDS fragment reads + a wave
//     barrier per query block and, more importantly, the 59 KB of per-wave P images: the workgroup needs 53 KB instead of
//     112 KB, so two (bf16) workgroups share a CU and one's softmax overlaps the other's MFMAs and loads.
//   * K, V and the first Q block are requested together (one round trip), the Q block of the second pass is requested
//     before the first pass computes, and the shift-mask labels of the keys are packed into 14 registers once per window
//     instead of being re-read from LDS for every score tile.
// -------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ Frag<T> frag_v_perm(const T* Vs, int LD, int d0, int ks, int c, int g) {
    Frag<T> f;
    if constexpr (sizeof(T) == 2) {
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const T* p0 = Vs + (32 * ks + 4 * g + (c >> 2)) * LD + d0 + 4 * (c & 3);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 16 * LD));
        const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        f.v = __builtin_bit_cast(bf16x8, both);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f.v[e] = Vs[(32 * ks + 4 * g + e) * LD + d0 + c];
            f.v[4 + e] = Vs[(32 * ks + 16 + 4 * g + e) * LD + d0 + c];
        }
    }
    return f;
}

template <typename T>
__device__ __forceinline__ Frag<T> frag_p_regs(const f32x4& lo, const f32x4& hi) {
    Frag<T> f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if constexpr (sizeof(T) == 2) {
            f.v[e] = (bf16)lo[e];
            f.v[4 + e] = (bf16)hi[e];
        } else {
            f.v[e] = lo[e];
            f.v[4 + e] = hi[e];
        }
    }
    return f;
}

// forward, third variant (default): the second generation with ONE 16-query tile per wave and pass.  A workgroup is seven
// waves, wave w takes query tiles w and w + 7 (no idle wave in the second pass, the 32-query blocks left one of four idle);
// the score strip is 56 registers instead of 112, so the kernel fits four waves per SIMD and two workgroups (14 waves) share
// a CU -- the second generation's SQ counters still showed 59 % of the wave cycles parked on loads with two waves per SIMD.
constexpr int FWD3_WAVES = 7;

template <typename T, bool WANT_ATTN, int HD>
__global__ __launch_bounds__(FWD3_WAVES * 64, sizeof(T) == 2 ? (HD == 32 ? 4 : 2) : 1) void attn_big_fwd3_kernel(
    const T* __restrict__ qkv, const float* __restrict__ qkv_bias, const int* __restrict__ win2tok, int L,
    const float* __restrict__ bias_frag, int ws, const int* __restrict__ region_ids, int nW, int Bw, int N, int nH,
    float scale, T* __restrict__ out, float* __restrict__ lse_out, float* __restrict__ attn_out) {
    using Cfg = BigCfg<T, HD>;
    constexpr int LDQ = Cfg::LDQ, VEC = Cfg::VEC, VPR = HD / VEC;
    constexpr int KS = HD / 32, DT = HD / 16;  // k-steps of a q.k product, 16-wide tiles of an output row
    constexpr int TILE = 16 * LDQ;
    static_assert(NT == 2 * FWD3_WAVES, "two query tiles per wave");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const BigTables tb = carve_tables(smem_raw);
    T* Ks = reinterpret_cast<T*>(smem_raw + Cfg::TABLE_BYTES);
    T* Vs = Ks + Cfg::FULL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    T* Qs = Vs + Cfg::FULL + wave * TILE;  // this wave's [16][LDQ] image: Q tile, then the output transpose

    const int unit = xcd_contiguous_id(blockIdx.x, gridDim.x);  // (bw, h); the heads of a window share an XCD
    const int bw = unit / nH, h = unit % nH;
    const int C = nH * HD;
    const long tok_base = (long)(bw / nW) * L;
    const T* src = qkv + h * HD;
    const bool masked = region_ids != nullptr;
    const float* bias_h = bias_frag + (long)h * (NT * NT * 256);

    load_window_tables(tb, win2tok, region_ids, bw % nW, N, ws, true);
    __syncthreads();
    SlotStage<T, 16, 64, HD> sq;
    const Vec16<T> padq = SlotStage<T, 16, 64, HD>::pad_piece(qkv_bias + h * HD, lane);
    {
        SlotStage<T, NPB, FWD3_WAVES * 64, HD> sk, sv;
        const Vec16<T> padk = SlotStage<T, NPB, FWD3_WAVES * 64, HD>::pad_piece(qkv_bias + C + h * HD, threadIdx.x);
        const Vec16<T> padv = SlotStage<T, NPB, FWD3_WAVES * 64, HD>::pad_piece(qkv_bias + 2 * C + h * HD, threadIdx.x);
        sk.load(src + C, 3L * C, tb.tok, tok_base, 0, N, &padk, threadIdx.x);
        sv.load(src + 2 * C, 3L * C, tb.tok, tok_base, 0, N, &padv, threadIdx.x);
        sq.load(src, 3L * C, tb.tok, tok_base, 16 * wave, N, &padq, lane);
        sk.store(Ks, 1.f, threadIdx.x);
        sv.store(Vs, 1.f, threadIdx.x);
        sq.store(Qs, scale, lane);
    }
    // query tiles without a live slot are skipped (live_query_tiles); the probabilities export computes every row it writes
    const unsigned live = WANT_ATTN ? ((1u << NT) - 1) : live_query_tiles(tb.tok, lane);
    __syncthreads();  // K, V complete (whole workgroup); everything below is private to the wave

#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int q0 = 16 * (wave + pass * FWD3_WAVES);
        const bool tile_live = (live >> (wave + pass * FWD3_WAVES)) & 1;
        if (pass > 0 && tile_live) {
            __builtin_amdgcn_wave_barrier();
            sq.store(Qs, scale, lane);
            __builtin_amdgcn_wave_barrier();
        }
        Frag<T> qf[KS];
        if (tile_live) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[ks] = frag_kc<T>(Qs, LDQ, 0, 32 * ks, c, g);
        }
        if (pass == 0 && ((live >> (wave + FWD3_WAVES)) & 1)) sq.load(src, 3L * C, tb.tok, tok_base, 16 * (wave + FWD3_WAVES), N, &padq, lane);
        if (!tile_live) {
            if (g == 0 && lse_out) lse_out[(long)unit * NPB + q0 + c] = 0.f;  // (a placeholder: the backward kernels skip these tiles or replace the value, attn_big_bwd_dkv2)
            continue;
        }
        const int rq = masked ? ((tb.pk[q0 + c] >> 16) & 0xff) : 0;
        f32x4 p[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            f32x4 b = *reinterpret_cast<const f32x4*>(bias_h + ((i * NT + (q0 >> 4)) * 64 + lane) * 4);
            if (masked) {
                const i32x4 pk4 = *reinterpret_cast<const i32x4*>(tb.pk + 16 * i + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] += (((pk4[r] >> 16) & 0xff) != rq) ? -100.f : 0.f;
            }
            p[i] = b;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) mma(frag_kc<T>(Ks, LDQ, 16 * i, 32 * ks, c, g), qf[ks], p[i]);
        }
        float m = -3.0e38f;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, p[i][r]);
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(p[i][r] - m);
                p[i][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
#pragma unroll
        for (int i = 0; i < NT; ++i) p[i] *= inv;
        if (g == 0 && lse_out) lse_out[(long)unit * NPB + q0 + c] = m + __logf(sum);
        if constexpr (WANT_ATTN) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = q0 + c, key = 16 * i + 4 * g + r;
                    if (q < N && key < N) attn_out[((long)unit * N + q) * N + key] = p[i][r];
                }
        }
        f32x4 o[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NPB / 32; ++ks) {
            const Frag<T> pf = frag_p_regs<T>(p[2 * ks], p[2 * ks + 1]);
#pragma unroll
            for (int j = 0; j < DT; ++j) mma(frag_v_perm<T>(Vs, LDQ, 16 * j, ks, c, g), pf, o[j]);  // O^T [d][query]: operands exchanged
        }
        store_tile_rows_t<T, HD>(o, 1.f, out + h * HD, (long)C, tb.tok, tok_base, q0, N, true, c, g);
    }
}

// -------------------------------------------------------------------------------------------------------------
// backward, second generation (default).  Same split (dQ + bias gradient | dK, dV) and the same mathematics; changes:
//   * dS / P never go through LDS: a score tile leaves its 4 accumulator rows per lane in exactly the k-slots an MFMA A
//     operand wants if the other operand is read with the matching row permutation (frag_v_perm) -- rows = keys for
//     dQ = dS K (tiles oriented S^T as in the forward), rows = queries for dV = P^T dO and dK = dS^T Q (tiles oriented S;
//     the bias arrives in a second fragment buffer with that orientation).  That frees the 59 / 72 KB of per-wave images
//     and lets two workgroups share a CU.
//   * both kernels rebuild P from the saved log-sum-exp (no max / sum passes, no shuffles).
//   * the relative-position-bias gradient stays in accumulator registers across the windows a wave visits (fragment
//     layout, reduced over `parts` afterwards).  Reducing it into its (2ws-1)^2-entry table in LDS with ds_add_f32 instead
//     (112 LDS atomics per wave and window) was tried and measured 3x slower -- the LDS atomic unit saturates
//     (profiles/r01_kernel_stats_w14_lds_atomic_dq.csv) -- and a 32-query strip per wave needs > 256 registers with those
//     accumulators, hence dq4 below: eight waves, one 16-query tile each.
// -------------------------------------------------------------------------------------------------------------

// dQ + bias gradient, fourth variant (default).  The bias gradient has to live in accumulator registers across the windows
// a wave visits; with a 32-query strip per wave that is 112 registers and the kernel cannot run two waves per SIMD
// (the first-generation dQ kernel: one wave per SIMD, every load round trip exposed).  Here a workgroup is EIGHT waves and a wave owns ONE
// query tile (16 queries): 56 bias-gradient + 56 P + 28 packed-dS registers, two waves per SIMD, while K and V are still
// staged once per 8 (6) query tiles.  Same fragment-layout bias-gradient workspace as the first generation.
constexpr int DQ4_WAVES = 8;
constexpr int DQ4_GROUPS = (NT + DQ4_WAVES - 1) / DQ4_WAVES;

template <typename T, int HD>
__global__ __launch_bounds__(DQ4_WAVES * 64) void attn_big_bwd_dq4_kernel(
    const T* __restrict__ qkv, const float* __restrict__ qkv_bias, const int* __restrict__ win2tok, int L, const T* __restrict__ dout,
    const T* __restrict__ fout, const float* __restrict__ lse_in, const float* __restrict__ bias_frag, int ws,
    const int* __restrict__ region_ids, int nW, int Bw, int N, int nH, float scale, int parts, T* __restrict__ dqkv,
    float* __restrict__ dbias_ws) {
    using Cfg = BigCfg<T, HD>;
    constexpr int LDQ = Cfg::LDQ, VEC = Cfg::VEC, VPR = HD / VEC;
    constexpr int KS = HD / 32, DT = HD / 16;
    // the relative-position-bias gradient (56 accumulator registers) exists for the windowed Swin models only: head_dim 64 is the
    // monolithic ViT, which has no bias table -- dbias_ws is left untouched there
    constexpr bool WANT_DB = HD == 32;
    constexpr int TILE = 16 * LDQ;  // one [16][LDQ] image
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const BigTables tb = carve_tables(smem_raw);
    T* Ks = reinterpret_cast<T*>(smem_raw + Cfg::TABLE_BYTES);
    T* Vs = Ks + Cfg::FULL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    T* Qs = Vs + Cfg::FULL + wave * (3 * TILE);
    T* Os = Qs + TILE;  // dO rows of this wave's queries
    T* Fs = Os + TILE;  // forward output rows

    const int unit = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int grp = unit % DQ4_GROUPS;
    const int ph = unit / DQ4_GROUPS;  // (part, h)
    const int h = ph % nH, part = ph / nH;
    const int qt = grp * DQ4_WAVES + wave;   // query tile of this wave
    const bool wave_ok = qt < NT;
    const int q0 = wave_ok ? 16 * qt : 0;
    const int C = nH * HD;
    const bool masked = region_ids != nullptr;
    const T* src = qkv + h * HD;
    const float* bias_h = bias_frag + (long)h * (NT * NT * 256);

    f32x4 db[WANT_DB ? NT : 1];
#pragma unroll
    for (int i = 0; i < (WANT_DB ? NT : 1); ++i) db[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const Vec16<T> padk = SlotStage<T, NPB, DQ4_WAVES * 64, HD>::pad_piece(qkv_bias + C + h * HD, threadIdx.x);
    const Vec16<T> padv = SlotStage<T, NPB, DQ4_WAVES * 64, HD>::pad_piece(qkv_bias + 2 * C + h * HD, threadIdx.x);
    const Vec16<T> padq = SlotStage<T, 16, 64, HD>::pad_piece(qkv_bias + h * HD, lane);
    const int iters = (Bw + parts - 1) / parts;
    for (int it = 0; it < iters; ++it) {
        const int bw = part + it * parts;
        const bool win_ok = bw < Bw;
        const int bwc = win_ok ? bw : 0;
        const bool active = win_ok && wave_ok;
        const long tok_base = (long)(bwc / nW) * L;
        __syncthreads();  // previous window's reads are complete
        load_window_tables(tb, win2tok, region_ids, bwc % nW, N, ws, win_ok);
        __syncthreads();
        const unsigned live = live_query_tiles(tb.tok, lane);
        if (((live >> (grp * DQ4_WAVES)) & ((1u << DQ4_WAVES) - 1)) == 0) continue;  // no query of this workgroup's tiles is live (whole workgroup: the table is shared)
        const bool tile_live = wave_ok && ((live >> qt) & 1);
        float lq = 0.f;
        {
            SlotStage<T, NPB, DQ4_WAVES * 64, HD> sk, sv;
            SlotStage<T, 16, 64, HD> sq, so, sf;
            sk.load(src + C, 3L * C, tb.tok, tok_base, 0, N, &padk, threadIdx.x);
            sv.load(src + 2 * C, 3L * C, tb.tok, tok_base, 0, N, &padv, threadIdx.x);
            if (tile_live) {
                sq.load(src, 3L * C, tb.tok, tok_base, q0, N, &padq, lane);
                so.load(dout + h * HD, (long)C, tb.tok, tok_base, q0, N, nullptr, lane);
                sf.load(fout + h * HD, (long)C, tb.tok, tok_base, q0, N, nullptr, lane);
                lq = lse_in[((long)bwc * nH + h) * NPB + q0 + c];
            }
            sk.store(Ks, 1.f, threadIdx.x);
            sv.store(Vs, 1.f, threadIdx.x);
            if (tile_live) {
                sq.store(Qs, scale, lane);
                so.store(Os, 1.f, lane);
                sf.store(Fs, 1.f, lane);
            }
        }
        __syncthreads();
        if (!tile_live) continue;  // (no workgroup barrier below this line)
        Frag<T> qf[KS], of[KS];
        const int rq = masked ? ((tb.pk[q0 + c] >> 16) & 0xff) : 0;
        float d = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = frag_kc<T>(Qs, LDQ, 0, 32 * ks, c, g);
            of[ks] = frag_kc<T>(Os, LDQ, 0, 32 * ks, c, g);
            const Frag<T> ff = frag_kc<T>(Fs, LDQ, 0, 32 * ks, c, g);
#pragma unroll
            for (int e = 0; e < 8; ++e) d += (float)of[ks].v[e] * (float)ff.v[e];
        }
        d += __shfl_xor(d, 16, 64);
        d += __shfl_xor(d, 32, 64);
        // P^T tiles of this query tile (rows = keys) from the saved log-sum-exp
        f32x4 pj[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            f32x4 b = *reinterpret_cast<const f32x4*>(bias_h + ((i * NT + (q0 >> 4)) * 64 + lane) * 4);
            if (masked) {
                const i32x4 pk4 = *reinterpret_cast<const i32x4*>(tb.pk + 16 * i + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] += (((pk4[r] >> 16) & 0xff) != rq) ? -100.f : 0.f;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) mma(frag_kc<T>(Ks, LDQ, 16 * i, 32 * ks, c, g), qf[ks], b);
#pragma unroll
            for (int r = 0; r < 4; ++r) b[r] = __expf(b[r] - lq);
            pj[i] = b;
        }
        // dP^T = V dO^T, dS = P o (dP - delta): accumulate the bias gradient, pack dS for the dQ product
        Frag<T> sfr[NPB / 32];
#pragma unroll
        for (int ks = 0; ks < NPB / 32; ++ks) {
            f32x4 ds2[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = 2 * ks + u;
                f32x4 dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kd = 0; kd < KS; ++kd) mma(frag_kc<T>(Vs, LDQ, 16 * i, 32 * kd, c, g), of[kd], dp);
                ds2[u] = pj[i] * (dp - d);
                if constexpr (WANT_DB) {
                    if (active) db[i] += ds2[u];
                }
            }
            sfr[ks] = frag_p_regs<T>(ds2[0], ds2[1]);
        }
        f32x4 acc[DT];
#pragma unroll
        for (int j = 0; j < DT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NPB / 32; ++ks)
#pragma unroll
            for (int j = 0; j < DT; ++j) mma(frag_v_perm<T>(Ks, LDQ, 16 * j, ks, c, g), sfr[ks], acc[j]);  // dQ^T [d][query]: operands exchanged
        store_tile_rows_t<T, HD>(acc, scale, dqkv + h * HD, 3L * C, tb.tok, tok_base, q0, N, active, c, g);
    }
    if (WANT_DB && wave_ok) {
        // frag layout of the NPB x NPB bias gradient: ((ki*NT + qj)*64 + lane)*4 + r
        float* wsp = dbias_ws + ((long)part * nH + h) * (NT * NT * 256);
#pragma unroll
        for (int i = 0; i < (WANT_DB ? NT : 0); ++i) *reinterpret_cast<f32x4*>(wsp + ((i * NT + qt) * 64 + lane) * 4) = db[WANT_DB ? i : 0];
    }
}

template <typename T, int HD>
__global__ __launch_bounds__((BigCfg<T, HD>::WAVES * 64), ((sizeof(T) == 2 && HD == 32) ? 2 : 1)) void attn_big_bwd_dkv2_kernel(
    const T* __restrict__ qkv, const float* __restrict__ qkv_bias, const int* __restrict__ win2tok, int L, const T* __restrict__ dout,
    const T* __restrict__ fout, const float* __restrict__ lse_in, const float* __restrict__ bias_frag_s, int ws,
    const int* __restrict__ region_ids, int nW, int Bw, int N, int nH, float scale, T* __restrict__ dqkv, float* __restrict__ dpad_ws) {
    using Cfg = BigCfg<T, HD>;
    constexpr int LDQ = Cfg::LDQ, WAVES = Cfg::WAVES;
    constexpr int KS = HD / 32, DT = HD / 16;
    constexpr int PASSES = (NQB + WAVES - 1) / WAVES;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const BigTables tb = carve_tables(smem_raw);
    T* Qs = reinterpret_cast<T*>(smem_raw + Cfg::TABLE_BYTES);  // scale*Q, [224][LDQ]
    T* Os = Qs + Cfg::FULL;                                      // dO,      [224][LDQ]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    T* Kb = Os + Cfg::FULL + wave * (2 * Cfg::BLK);
    T* Vb = Kb + Cfg::BLK;

    const int unit = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int bw = unit / nH, h = unit % nH;
    const int C = nH * HD;
    const long tok_base = (long)(bw / nW) * L;
    const T* src = qkv + h * HD;
    const bool masked = region_ids != nullptr;
    const float* bias_h = bias_frag_s + (long)h * (NT * NT * 256);

    load_window_tables(tb, win2tok, region_ids, bw % nW, N, ws, true);
    __syncthreads();
    const unsigned live = HD == 32 ? live_query_tiles(tb.tok, lane) : (1u << NT) - 1;
    SlotStage<T, 32, 64, HD> sk, sv;
    const Vec16<T> padk_piece = SlotStage<T, 32, 64, HD>::pad_piece(qkv_bias + C + h * HD, lane);
    const Vec16<T> padv_piece = SlotStage<T, 32, 64, HD>::pad_piece(qkv_bias + 2 * C + h * HD, lane);
    {
        SlotStage<T, NPB, WAVES * 64, HD> sq, so;
        const Vec16<T> padq_piece = SlotStage<T, NPB, WAVES * 64, HD>::pad_piece(qkv_bias + h * HD, threadIdx.x);
        sq.load(src, 3L * C, tb.tok, tok_base, 0, N, &padq_piece, threadIdx.x);
        so.load(dout + h * HD, (long)C, tb.tok, tok_base, 0, N, nullptr, threadIdx.x);
        sk.load(src + C, 3L * C, tb.tok, tok_base, 32 * wave, N, &padk_piece, lane);
        sv.load(src + 2 * C, 3L * C, tb.tok, tok_base, 32 * wave, N, &padv_piece, lane);
        // per-query statistics: saved log-sum-exp and delta = sum_d dO[q,d] * O[q,d]
        for (int t = threadIdx.x; t < NPB; t += WAVES * 64) {
            float l = 0.f, d = 0.f;
            if (t < N) {
                l = lse_in[(long)unit * NPB + t];
                const int tok = tb.tok[t];
                if (tok >= 0) {
                    const T* orow = fout + (tok_base + tok) * (long)C + h * HD;
                    const T* grow = dout + (tok_base + tok) * (long)C + h * HD;
#pragma unroll
                    for (int vv = 0; vv < HD / Cfg::VEC; ++vv) {
                        const Vec16<T> ov = ld16<T>(orow + vv * Cfg::VEC), gv = ld16<T>(grow + vv * Cfg::VEC);
#pragma unroll
                        for (int e = 0; e < Cfg::VEC; ++e) d += ov.get(e) * gv.get(e);
                    }
                }
            }
            // a dead query tile (no live token: the forward skipped it and left lse = 0) contributes P = exp(s - lse) = 0 in BOTH forms of
            // the block below -- the full form walks dead tiles too, and with lse = 0 an s > 88 there would be inf * (dO = 0) = NaN
            if (!((live >> (t >> 4)) & 1)) l = 3.0e38f;
            tb.lse[t] = l;
            tb.delta[t] = d;
        }
        sq.store(Qs, scale, threadIdx.x);
        so.store(Os, 1.f, threadIdx.x);
        sk.store(Kb, 1.f, lane);
        sv.store(Vb, 1.f, lane);
    }
    float padk[4 * (HD / 16)], padv[4 * (HD / 16)];  // pad-slot rows of dK / dV: head channels 16 j + 4g + r of this lane's slots
#pragma unroll
    for (int e = 0; e < 4 * (HD / 16); ++e) padk[e] = padv[e] = 0.f;
    __syncthreads();  // Q, dO, lse, delta staged by the whole workgroup; Kb / Vb are private to the wave

#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        const int kb = wave + pass * WAVES;
        const bool valid = kb < NQB;
        const int k0 = valid ? 32 * kb : 0;
        if (pass > 0) {
            __builtin_amdgcn_wave_barrier();
            sk.store(Kb, 1.f, lane);
            sv.store(Vb, 1.f, lane);
            __builtin_amdgcn_wave_barrier();
        }
        Frag<T> kf[2][KS], vf[2][KS];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int kd = 0; kd < KS; ++kd) {
                kf[a][kd] = frag_kc<T>(Kb, LDQ, 16 * a, 32 * kd, c, g);
                vf[a][kd] = frag_kc<T>(Vb, LDQ, 16 * a, 32 * kd, c, g);
            }
        if (pass + 1 < PASSES) {  // next pass's key / value rows travel while this pass computes
            const int kn = wave + (pass + 1) * WAVES;
            const int kn0 = kn < NQB ? 32 * kn : 0;
            sk.load(src + C, 3L * C, tb.tok, tok_base, kn0, N, &padk_piece, lane);
            sv.load(src + 2 * C, 3L * C, tb.tok, tok_base, kn0, N, &padv_piece, lane);
        }
        int rkey[2] = {0, 0};
        if (masked) {
            rkey[0] = (tb.pk[k0 + c] >> 16) & 0xff;
            rkey[1] = (tb.pk[k0 + 16 + c] >> 16) & 0xff;
        }
        // P, dV, dS, dK over the window's query tiles.  full: every query tile; otherwise only the tiles of `live` (live_query_tiles) --
        // the branches around single tiles cost the full windows their instruction-level parallelism, so this form is taken only for
        // windows that are mostly padding (a constant per call site: two specialised copies after inlining).
        auto block = [&](const bool full) __attribute__((always_inline)) {
            // P block, oriented S: rows = queries (14 tiles), columns = this block's 32 keys (2 tiles)
            f32x4 p[2][NT];
        #pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (!full && !((live >> j) & 1)) {
                    p[0][j] = p[1][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    continue;
                }
                Frag<T> qf[KS];
        #pragma unroll
                for (int kd = 0; kd < KS; ++kd) qf[kd] = frag_kc<T>(Qs, LDQ, 16 * j, 32 * kd, c, g);
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(tb.lse + 16 * j + 4 * g);
                const i32x4 pk4 = *reinterpret_cast<const i32x4*>(tb.pk + 16 * j + 4 * g);
        #pragma unroll
                for (int a = 0; a < 2; ++a) {
                    f32x4 b = *reinterpret_cast<const f32x4*>(bias_h + ((j * NT + (k0 >> 4) + a) * 64 + lane) * 4);
                    if (masked) {
        #pragma unroll
                        for (int r = 0; r < 4; ++r) b[r] += (((pk4[r] >> 16) & 0xff) != rkey[a]) ? -100.f : 0.f;
                    }
        #pragma unroll
                    for (int kd = 0; kd < KS; ++kd) mma(qf[kd], kf[a][kd], b);
        #pragma unroll
                    for (int r = 0; r < 4; ++r) b[r] = (16 * j + 4 * g + r < N) ? __expf(b[r] - l4[r]) : 0.f;  // padded queries carry no gradient
                    p[a][j] = b;
                }
            }
            // dV[key][d] = sum_q P[q][key] dO[q][d]
            {
                f32x4 av[2][DT];
        #pragma unroll
                for (int a = 0; a < 2; ++a)
        #pragma unroll
                    for (int j = 0; j < DT; ++j) av[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        #pragma unroll
                for (int ks = 0; ks < NPB / 32; ++ks) {
                    if (!full && !((live >> (2 * ks)) & 3)) continue;  // both query tiles of this k-step are dead: P = 0
                    Frag<T> pf[2];
        #pragma unroll
                    for (int a = 0; a < 2; ++a) pf[a] = frag_p_regs<T>(p[a][2 * ks], p[a][2 * ks + 1]);
        #pragma unroll
                    for (int j = 0; j < DT; ++j) {
                        const Frag<T> oj = frag_v_perm<T>(Os, LDQ, 16 * j, ks, c, g);
        #pragma unroll
                        for (int a = 0; a < 2; ++a) mma(oj, pf[a], av[a][j]);  // dV^T [d][key]: operands exchanged
                    }
                }
                store_block_rows_t<T, HD>(av, 1.f, dqkv + 2 * C + h * HD, 3L * C, tb.tok, tok_base, k0, N, valid, padv, c, g);
            }
            // dS = P o (dP - delta), dP[q][key] = sum_d dO[q][d] V[key][d]; dS overwrites P
        #pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (!full && !((live >> j) & 1)) continue;  // (p stays 0)
                Frag<T> of[KS];
        #pragma unroll
                for (int kd = 0; kd < KS; ++kd) of[kd] = frag_kc<T>(Os, LDQ, 16 * j, 32 * kd, c, g);
                const f32x4 dl4 = *reinterpret_cast<const f32x4*>(tb.delta + 16 * j + 4 * g);
        #pragma unroll
                for (int a = 0; a < 2; ++a) {
                    f32x4 dp = {0.f, 0.f, 0.f, 0.f};
        #pragma unroll
                    for (int kd = 0; kd < KS; ++kd) mma(of[kd], vf[a][kd], dp);
                    p[a][j] = p[a][j] * (dp - dl4);
                }
            }
            // dK[key][d] = sum_q dS[q][key] (scale q)[q][d]
            {
                f32x4 ak[2][DT];
        #pragma unroll
                for (int a = 0; a < 2; ++a)
        #pragma unroll
                    for (int j = 0; j < DT; ++j) ak[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        #pragma unroll
                for (int ks = 0; ks < NPB / 32; ++ks) {
                    if (!full && !((live >> (2 * ks)) & 3)) continue;
                    Frag<T> sf[2];
        #pragma unroll
                    for (int a = 0; a < 2; ++a) sf[a] = frag_p_regs<T>(p[a][2 * ks], p[a][2 * ks + 1]);
        #pragma unroll
                    for (int j = 0; j < DT; ++j) {
                        const Frag<T> qj = frag_v_perm<T>(Qs, LDQ, 16 * j, ks, c, g);
        #pragma unroll
                        for (int a = 0; a < 2; ++a) mma(qj, sf[a], ak[a][j]);  // dK^T [d][key]
                    }
                }
                store_block_rows_t<T, HD>(ak, 1.f, dqkv + C + h * HD, 3L * C, tb.tok, tok_base, k0, N, valid, padk, c, g);
            }
        };
        // (head_dim 64 = the monolithic ViTs: 197 of 224 slots live in every "window", the second form would only cost registers)
        if constexpr (HD == 32) {
            if (__builtin_popcount(live) <= NT - 4) block(false);
            else block(true);
        } else {
            block(true);
        }
    }
    // sum over the 16 slots of a lane group (DPP row adds); lane c == 0 of every group holds channels 16 j + 4g + r
#pragma unroll
    for (int e = 0; e < 4 * (HD / 16); ++e) {
        padk[e] = row16_sum(padk[e]);
        padv[e] = row16_sum(padv[e]);
    }
    if (c == 0) {  // one slab row per (unit, wave): [k | v][nH][hd]
        float* pw = dpad_ws + ((long)unit * WAVES + wave) * 2 * C + h * HD;
#pragma unroll
        for (int j = 0; j < HD / 16; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pw[16 * j + 4 * g + r] = padk[4 * j + r];
                pw[C + 16 * j + 4 * g + r] = padv[4 * j + r];
            }
    }
}

template <typename T, int HD>
size_t fwd3_lds() {
    using Cfg = BigCfg<T, HD>;
    return Cfg::TABLE_BYTES + (size_t)(2 * Cfg::FULL + FWD3_WAVES * 16 * Cfg::LDQ) * sizeof(T);
}
template <typename T, int HD>
size_t dq4_lds() {
    using Cfg = BigCfg<T, HD>;
    return Cfg::TABLE_BYTES + (size_t)(2 * Cfg::FULL + DQ4_WAVES * 3 * 16 * Cfg::LDQ) * sizeof(T);
}
template <typename T, int HD>
size_t dkv2_lds() {
    using Cfg = BigCfg<T, HD>;
    return Cfg::TABLE_BYTES + (size_t)(2 * Cfg::FULL + Cfg::WAVES * 2 * Cfg::BLK) * sizeof(T);
}

inline int big_parts(int Bw, int nH) {
    // the dQ workgroups are persistent (each walks its share of the windows): parts * nH * DQ4_GROUPS of them are launched, rounded
    // DOWN to 1024 so that the last round of resident workgroups is not a handful of stragglers (cf. bwd_parts in window_attn.hip)
    int parts = 512 / nH;
    if (parts > Bw) parts = Bw;
    return parts < 1 ? 1 : parts;
}

}  // namespace

#define STREAM(s_) hipStream_t stream = reinterpret_cast<hipStream_t>(s_)

int esvit_big_frag_elems() { return NT * NT * 256; }
int esvit_big_npb() { return NPB; }
int esvit_big_parts(int Bw, int nH) { return big_parts(Bw, nH); }
int esvit_big_pad_rows(int Bw, int nH, int dtype) { (void)dtype; return Bw * nH * FWD3_WAVES; }  // >= waves per workgroup of every dK/dV variant

static int fill_bias_frag_big(const float* rel_table, int ws, int N, int nH, float* bias_frag_ws, hipStream_t stream) {
    const long n = 2L * nH * NT * NT * 256;  // S^T-oriented tiles of every head, then S-oriented tiles of every head
    hipLaunchKernelGGL(relpos_bias_frag_big_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, rel_table, ws, N, nH, bias_frag_ws);
    ESVIT_CHECK_LAUNCH("relpos_bias(frag, 14x14)");
    return ESVIT_OK;
}

template <typename T, int HD>
static int big_fwd_launch(const void* qkv, const float* qkv_bias, const int32_t* win2tok, int L, const float* rel_table, int ws,
                          const int32_t* region_ids, int nW, int Bw, int N, int nH, float scale, void* out, float* lse, float* attn_out,
                          hipStream_t stream) {
    const size_t lds = fwd3_lds<T, HD>();
    if (attn_out) {
        auto kern = attn_big_fwd3_kernel<T, true, HD>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(Bw * nH), dim3(FWD3_WAVES * 64), lds, stream, (const T*)qkv, qkv_bias, win2tok, L, rel_table, ws,
                           region_ids, nW, Bw, N, nH, scale, (T*)out, lse, attn_out);
    } else {
        auto kern = attn_big_fwd3_kernel<T, false, HD>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(Bw * nH), dim3(FWD3_WAVES * 64), lds, stream, (const T*)qkv, qkv_bias, win2tok, L, rel_table, ws,
                           region_ids, nW, Bw, N, nH, scale, (T*)out, lse, attn_out);
    }
    ESVIT_CHECK_LAUNCH("window_attn_fwd(14x14)");
    return ESVIT_OK;
}

int esvit_big_attn_fwd(int dtype, const void* qkv, const float* qkv_bias, const int32_t* win2tok, int L, const float* rel_table, int ws,
                       float* bias_frag_ws, const int32_t* region_ids, int nW, int nB, int N, int nH, int hd, float scale, void* out, float* lse,
                       float* attn_out, hipStream_t stream) {
    ESVIT_CHECK_ARG(N <= NPB, "window_attn: window %d too large", ws);
    ESVIT_CHECK_ARG(hd == 32 || (hd == 64 && dtype == ESVIT_BF16), "window_attn: %d tokens at head_dim %d: 32, or 64 in bf16", N, hd);
    ESVIT_CHECK_ARG(bias_frag_ws != nullptr, "esvit_window_attn_fwd: the bias_frag_ws scratch is required");
    {
        int rc = rel_table ? fill_bias_frag_big(rel_table, ws, N, nH, bias_frag_ws, stream) : ESVIT_OK;
        if (rc != ESVIT_OK) return rc;
    }
    const int Bw = nB * nW;
    if (hd == 64)
        return big_fwd_launch<bf16, 64>(qkv, qkv_bias, win2tok, L, bias_frag_ws, ws, region_ids, nW, Bw, N, nH, scale, out, lse, attn_out, stream);
    if (dtype == ESVIT_BF16)
        return big_fwd_launch<bf16, 32>(qkv, qkv_bias, win2tok, L, bias_frag_ws, ws, region_ids, nW, Bw, N, nH, scale, out, lse, attn_out, stream);
    return big_fwd_launch<float, 32>(qkv, qkv_bias, win2tok, L, bias_frag_ws, ws, region_ids, nW, Bw, N, nH, scale, out, lse, attn_out, stream);
}

template <typename T, int HD>
static int big_bwd_launch(const void* qkv, const float* qkv_bias, const int32_t* win2tok, int L, const void* dout, const void* fout,
                          const float* lse, const float* rel_table, int ws, const int32_t* region_ids, int nW, int Bw, int N,
                          int nH, float scale, void* dqkv, float* dbias_ws, float* dpad_ws, hipStream_t stream) {
    using Cfg = BigCfg<T, HD>;
    const int parts = big_parts(Bw, nH);
#ifdef ESVIT_BIG_BWD_FORK  // probe (tools/probe/ab_big_bwd_fork.sh): the dK / dV kernel on a second stream beside the dQ kernel (disjoint outputs)
    static hipStream_t s2 = nullptr;
    static hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    if (!s2) {
        (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
        (void)hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&ev_join, hipEventDisableTiming);
    }
    (void)hipEventRecord(ev_fork, stream);
    (void)hipStreamWaitEvent(s2, ev_fork, 0);
    hipStream_t stream_kv = s2;
#else
    hipStream_t stream_kv = stream;
#endif
    {
        auto kern = attn_big_bwd_dq4_kernel<T, HD>;
        const size_t lds = dq4_lds<T, HD>();
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(parts * nH * DQ4_GROUPS), dim3(DQ4_WAVES * 64), lds, stream, (const T*)qkv, qkv_bias, win2tok, L,
                           (const T*)dout, (const T*)fout, lse, rel_table, ws, region_ids, nW, Bw, N, nH, scale, parts, (T*)dqkv, dbias_ws);
        ESVIT_CHECK_LAUNCH("window_attn_bwd(14x14, dQ)");
    }
    {
        auto kern = attn_big_bwd_dkv2_kernel<T, HD>;
        const size_t lds = dkv2_lds<T, HD>();
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(Bw * nH), dim3(Cfg::WAVES * 64), lds, stream_kv, (const T*)qkv, qkv_bias, win2tok, L, (const T*)dout,
                           (const T*)fout, lse, rel_table + (long)nH * NT * NT * 256, ws, region_ids, nW, Bw, N, nH, scale, (T*)dqkv, dpad_ws);
        ESVIT_CHECK_LAUNCH("window_attn_bwd(14x14, dK dV)");
    }
#ifdef ESVIT_BIG_BWD_FORK
    (void)hipEventRecord(ev_join, s2);
    (void)hipStreamWaitEvent(stream, ev_join, 0);
#endif
    return ESVIT_OK;
}

int esvit_big_attn_bwd(int dtype, const void* qkv, const float* qkv_bias, const int32_t* win2tok, int L, const void* dout,
                       const void* fout, const float* lse, const float* rel_table_, int ws, float* bias_frag_ws, const int32_t* region_ids,
                       int nW, int nB, int N, int nH, int hd, float scale, void* dqkv, float* dbias_ws, float* dpad_ws, hipStream_t stream) {
    ESVIT_CHECK_ARG(N <= NPB, "window_attn: window %d too large", ws);
    ESVIT_CHECK_ARG(hd == 32 || (hd == 64 && dtype == ESVIT_BF16), "window_attn: %d tokens at head_dim %d: 32, or 64 in bf16", N, hd);
    ESVIT_CHECK_ARG(fout && lse, "esvit_window_attn_bwd: 14x14 windows need the forward output and log-sum-exp");
    ESVIT_CHECK_ARG(bias_frag_ws != nullptr, "esvit_window_attn_bwd: the bias_frag_ws scratch is required");
    {
        int rc = rel_table_ ? fill_bias_frag_big(rel_table_, ws, N, nH, bias_frag_ws, stream) : ESVIT_OK;
        if (rc != ESVIT_OK) return rc;
    }
    const float* rel_table = bias_frag_ws;  // the kernels read the frag-layout bias
    const int Bw = nB * nW;
    if (hd == 64)
        return big_bwd_launch<bf16, 64>(qkv, qkv_bias, win2tok, L, dout, fout, lse, rel_table, ws, region_ids, nW, Bw, N, nH, scale, dqkv, dbias_ws,
                                        dpad_ws, stream);
    if (dtype == ESVIT_BF16)
        return big_bwd_launch<bf16, 32>(qkv, qkv_bias, win2tok, L, dout, fout, lse, rel_table, ws, region_ids, nW, Bw, N, nH, scale, dqkv, dbias_ws,
                                        dpad_ws, stream);
    return big_bwd_launch<float, 32>(qkv, qkv_bias, win2tok, L, dout, fout, lse, rel_table, ws, region_ids, nW, Bw, N, nH, scale, dqkv, dbias_ws,
                                     dpad_ws, stream);
}

int esvit_i_relpos_fold(const float* dbias_ws, int parts, int nt, const int64_t* index, int N, int nH, float* dtable, hipStream_t stream);

int esvit_big_relpos_bias_bwd(const float* dbias_ws, int parts, const int64_t* index, int N, int nH, int table_rows, float* dtable,
                              int accumulate, hipStream_t stream) {
    hipError_t e = accumulate ? hipSuccess : hipMemsetAsync(dtable, 0, (size_t)table_rows * nH * sizeof(float), stream);
    if (e != hipSuccess) {
        esvit_set_error("esvit_relpos_bias_bwd: memset failed: %s", hipGetErrorString(e));
        return ESVIT_ERR_HIP;
    }
    return esvit_i_relpos_fold(dbias_ws, parts, NT, index, N, nH, dtable, stream);  // fold of the per-workgroup slabs + scatter, one launch
}
