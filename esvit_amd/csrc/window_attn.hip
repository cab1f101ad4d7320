// Shifted-window attention core (swin_transformer.py:126-152) for 7x7 windows (N = 49 <= 64,
// head_dim 32) on CDNA4 MFMA tiles, forward and backward.
//
// One wave owns one (window, head): q, k, v (49 x 32 each) are staged in LDS, the score tile is
// computed TRANSPOSED, S^T = K (scale*Q)^T, so that after the MFMA each lane holds, for its query
// column q = 16j + c, the keys {16i + 4g + r}: the softmax reductions over keys are then
// in-register plus two cross-lane steps (xor 16, xor 32) instead of a 16-lane butterfly per
// row.  Relative-position bias and the shift mask arrive pre-arranged in exactly this fragment
// order ("frag layout", one 16-byte load per lane per 16x16 tile), with -1e30 in the key
// columns >= N so padding never needs a branch.
//
// frag layout of an NP x NP (NP = 64) matrix X[q][key]:
//     X_frag[((ki*4 + qj)*64 + lane)*4 + r] = X[q = 16*qj + c][key = 16*ki + 4*g + r],  lane = 16*g + c
//
// Backward recomputes P, then  dV = P^T dO,  dP^T = V dO^T,  dS = P o (dP - rowsum(P o dP)),
// dQ = scale * dS K,  dK = dS^T (scale*Q).  Operands whose reduction index is the LDS row index
// (P^T, dS^T, and the [token][d] images used as B operands) are read with ds_read_b64_tr_b16, so
// nothing is transposed through memory.  The bias gradient is accumulated in registers across
// all windows a wave processes and written once per wave to a partial slab (no atomics).
#include "common.h"
#include "mfma.h"
#include "../../include/esvit_hip.h"

namespace {

constexpr int NP = 64;  // padded tokens per window
constexpr int HD = 32;  // head dim
constexpr int NF = (NP / 16) * (NP / 16);
constexpr int FRAG_ELEMS = NF * 256;  // floats per (head) or (window) frag-layout matrix

template <typename T, int HDIM>
struct AttnCfgH {
    static constexpr int VEC = ElemTraits<T>::VEC;
    static constexpr int LDQ = HDIM + VEC;  // [NP][LDQ] images of q, k, v, dO
    static constexpr int LDP = NP + VEC;    // [NP][LDP] image of P / dS, [HDIM][LDP] image of V^T
    static constexpr int QK_ELEMS = NP * LDQ;
    static constexpr int P_ELEMS = NP * LDP;
    static constexpr int VT_ELEMS = HDIM * LDP;
    static constexpr int R1 = (2 * QK_ELEMS > P_ELEMS) ? 2 * QK_ELEMS : P_ELEMS;  // Q,K overlaid by P
    static constexpr int FWD_PER_WAVE = R1 + VT_ELEMS;
    static constexpr int BWD_PER_WAVE = 2 * QK_ELEMS + P_ELEMS;
    static constexpr int BWD3_PER_PAIR = 5 * QK_ELEMS + P_ELEMS;  // Q, K, V, dO, out-staging, P/dS
};
template <typename T>
using AttnCfg = AttnCfgH<T, HD>;  // head_dim 32 (Swin); the third-generation kernels also take 64 (CvT: dim / heads)

template <typename T>
__device__ __forceinline__ void store_frag4(T* p, f32x4 v) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<f32x4*>(p) = v;
    } else {
        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(p) = o;
    }
}

template <typename T>
__device__ __forceinline__ f32x4 load_frag4(const T* p) {
    if constexpr (sizeof(T) == 4) {
        return *reinterpret_cast<const f32x4*>(p);
    } else {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
}

// -------------------------------------------------------------------------------------------------
// Forward, second generation (the default): the first-generation kernel above spent 64% of its wave cycles parked
// on a serial chain (slot map -> row loads -> LDS -> compute -> store) with one window per wave.  Here a wave owns one
// head and a strided set of windows, requests the q, k, v rows of window it+1 (and the slot map of window it+2) while
// it computes window `it`, synchronises only with itself, and stores 16-byte rows.  Same math, same LDS images.
typedef unsigned int u32x4_f __attribute__((ext_vector_type(4)));

// P V with P in registers (fourth-generation forward): operand helpers, see window_attn_big.hip for the derivation
template <typename T>
__device__ __forceinline__ Frag<T> frag_v_perm64(const T* Vs, int LD, int d0, int ks, int c, int g) {
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
__device__ __forceinline__ Frag<T> frag_p_regs64(const f32x4& lo, const f32x4& hi) {
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

// Forward (the fourth generation; its predecessors are gone): a wave pair per (window, head), P kept in registers and V staged in its natural [key][d]
// layout -- no P image, no 2-byte transposed V stores (16 per thread and window), one block barrier less per window.
template <typename T, bool WANT_ATTN, int HDIM>
__global__ __launch_bounds__(128, (HDIM == 32 ? 3 : 2)) void attn_fwd4_kernel(const T* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                          const int* __restrict__ win2tok, int L, const float* __restrict__ bias_frag,
                                                          const int* __restrict__ region_ids, int nW, int Bw, int N, int nH, float scale,
                                                          int parts, T* __restrict__ out, float* __restrict__ attn_out) {
    using Cfg = AttnCfgH<T, HDIM>;
    constexpr int KS = HDIM / 32, DT = HDIM / 16;  // k-steps over the head dim, 16-wide output column tiles
    constexpr int LDQ = Cfg::LDQ, VEC = Cfg::VEC, ES = sizeof(T);
    constexpr int VPR = HDIM / VEC, NV = NP * VPR / 128, LSTEP = 128 / VPR;
    constexpr int NS = 32 * VPR / 64, SSTEP = 64 / VPR;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    T* base = reinterpret_cast<T*>(smem_raw);
    T* Qs = base;                      // [NP][LDQ]; wave w's own 32 query rows double as its output staging image
    T* Ks = base + Cfg::QK_ELEMS;      // [NP][LDQ]
    T* Vs = base + 2 * Cfg::QK_ELEMS;  // [NP][LDQ], natural layout: P V reads it with transpose reads (frag_v_perm64)

    const long unit = xcd_contiguous_id(blockIdx.x, gridDim.x);  // the heads of a part share an XCD (and its L2)
    const bool unit_ok = unit < (long)parts * nH;
    const int h = (int)(unit % nH);
    const int part = (int)(unit / nH);
    const int C = nH * HDIM;
    const bool masked = region_ids != nullptr;
    const int lrow0 = tid / VPR, dv = tid % VPR;
    const int srow0 = 32 * w + lane / VPR;
    const float* bias_f = bias_frag + (long)h * FRAG_ELEMS;

    Vec16<T> padq, padk_v, padv_v;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        padq.set(e, qkv_bias[h * HDIM + dv * VEC + e]);
        padq.set(e, padq.get(e) * scale);
        padk_v.set(e, qkv_bias[C + h * HDIM + dv * VEC + e]);
        padv_v.set(e, qkv_bias[2 * C + h * HDIM + dv * VEC + e]);
    }

    const int iters = (Bw + parts - 1) / parts;
    auto win_of = [&](int it, bool& act) -> int {
        const int bw = part + it * parts;
        act = unit_ok && it < iters && bw < Bw;
        return act ? bw : 0;
    };
    auto load_map = [&](int it, int& tok, int& reg) {
        bool act;
        const int bw = win_of(it, act);
        tok = (act && lane < N) ? win2tok[(long)(bw % nW) * N + lane] : -1;
        reg = (masked && act && lane < N) ? region_ids[(long)(bw % nW) * N + lane] : -1;
    };
    struct Win {
        u32x4_f q[NV], k[NV], v[NV];
        int ltok[NV];
        int mytok, myreg, bw;
        long tok_base;
        bool active;
    };
    auto issue_rows = [&](int it, int mytok, int myreg, Win& x) {
        x.bw = win_of(it, x.active);
        x.mytok = mytok;
        x.myreg = myreg;
        x.tok_base = (long)(x.bw / nW) * L;
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(qkv + x.tok_base * 3L * C), 0, (int)(L * 3L * C * ES), 0x00020000);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int tok = __shfl(mytok, lrow0 + LSTEP * i, 64);
            x.ltok[i] = tok;
            const int vq = tok >= 0 ? tok * 3 * C * ES + dv * 16 : OOB;
            x.q[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, vq, h * HDIM * ES, 0);
            x.k[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, vq, (C + h * HDIM) * ES, 0);
            x.v[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, vq, (2 * C + h * HDIM) * ES, 0);
        }
    };

    Win cur, nxt;
    int tok1, reg1, tok2 = -1, reg2 = -1;
    load_map(0, tok1, reg1);
    issue_rows(0, tok1, reg1, cur);
    load_map(1, tok1, reg1);

    for (int it = 0; it < iters; ++it) {
        __syncthreads();  // the other wave is done with the previous window's P / V images
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int t = lrow0 + LSTEP * i;
            const bool padslot = cur.active && t < N && cur.ltok[i] < 0;
            Vec16<T> xq, xk, xv;
            xq.v = __builtin_bit_cast(decltype(xq.v), cur.q[i]);
            xk.v = __builtin_bit_cast(decltype(xk.v), cur.k[i]);
            xv.v = __builtin_bit_cast(decltype(xv.v), cur.v[i]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) xq.set(e, xq.get(e) * scale);
            if (padslot) {
                xq = padq;
                xk = padk_v;
                xv = padv_v;
            }
            st16<T>(Qs + t * LDQ + dv * VEC, xq);
            st16<T>(Ks + t * LDQ + dv * VEC, xk);
            st16<T>(Vs + t * LDQ + dv * VEC, xv);
        }
        const bool active = cur.active;
        const int mytok = cur.mytok, myreg = cur.myreg;
        const long tok_base = cur.tok_base;
        const long unit_wh = (long)cur.bw * nH + h;
        __syncthreads();  // images complete
        issue_rows(it + 1, tok1, reg1, nxt);
        load_map(it + 2, tok2, reg2);

        f32x4 p[4][2];
        {
            Frag<T> kf[4][KS], qf[2][KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) kf[i][ks] = frag_kc<T>(Ks, LDQ, 16 * i, 32 * ks, c, g);
#pragma unroll
                for (int jl = 0; jl < 2; ++jl) qf[jl][ks] = frag_kc<T>(Qs, LDQ, 16 * (2 * w + jl), 32 * ks, c, g);
            }
            int rq[2];
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) rq[jl] = __shfl(myreg, 16 * (2 * w + jl) + c, 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int rk[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) rk[r] = __shfl(myreg, 16 * i + 4 * g + r, 64);
#pragma unroll
                for (int jl = 0; jl < 2; ++jl) {
                    f32x4 b = *reinterpret_cast<const f32x4*>(bias_f + ((i * 4 + 2 * w + jl) * 64 + lane) * 4);
                    if (masked) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) b[r] += (rk[r] != rq[jl]) ? -100.f : 0.f;
                    }
                    p[i][jl] = b;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) mma(kf[i][ks], qf[jl][ks], p[i][jl]);
                }
            }
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) {
                float m = -3.0e38f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) m = fmaxf(m, p[i][jl][r]);
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __expf(p[i][jl][r] - m);
                        p[i][jl][r] = e;
                        sum += e;
                    }
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                const float inv = 1.f / sum;
#pragma unroll
                for (int i = 0; i < 4; ++i) p[i][jl] *= inv;
            }
        }
        if constexpr (WANT_ATTN) {
            if (attn_out && active) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int q = 16 * (2 * w + jl) + c, key = 16 * i + 4 * g + r;
                            if (q < N && key < N) attn_out[((unit_wh * N) + q) * N + key] = p[i][jl][r];
                        }
            }
        }
        // P V straight from the score accumulators: lane (c, g) holds, for query c, keys 16i + 4g + r, i.e. for a 32-key chunk
        // the keys {32ks + 4g + e, 32ks + 16 + 4g + e}; V's fragment is read with the same key permutation.
        // The product is formed TRANSPOSED (the two MFMA operands exchanged: their fragment layouts are symmetric): O^T [d][query],
        // lane (c, g) = head channels 16 dt + 4g + r of query 16 il + c -- a row piece that leaves as a 16-byte vector without the
        // LDS transpose (2-byte scattered writes) the [query][d] orientation needed.
        f32x4 o[2][DT];
#pragma unroll
        for (int il = 0; il < 2; ++il)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[il][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {  // 64 keys = two 32-deep steps
            Frag<T> vf[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) vf[dt] = frag_v_perm64<T>(Vs, LDQ, 16 * dt, ks, c, g);
#pragma unroll
            for (int il = 0; il < 2; ++il) {
                const Frag<T> pf = frag_p_regs64<T>(p[2 * ks][il], p[2 * ks + 1][il]);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) mma(vf[dt], pf, o[il][dt]);
            }
        }
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + tok_base * (long)C, 0, (int)(L * (long)C * ES), 0x00020000);
#pragma unroll
        for (int il = 0; il < 2; ++il) {
            const int tok = __shfl(mytok, 32 * w + 16 * il + c, 64);
            const bool ok = active && tok >= 0;
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int dt = 0; dt < DT; dt += 2) {
                    const esvit_u32x4 x = esvit_pack_tile_pair_bf16(o[il][dt], o[il][dt + 1]);
                    buffer_store_b128(x, ro, ok ? tok * C * ES + (16 * dt + esvit_tile_pair_ch0(g)) * ES : OOB, h * HDIM * ES);
                }
            } else {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) buffer_store_b128(o[il][dt], ro, ok ? tok * C * ES + (16 * dt + 4 * g) * ES : OOB, h * HDIM * ES);
            }
        }
        cur = nxt;
        tok1 = tok2;
        reg1 = reg2;
    }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// -------------------------------------------------------------------------------------------------
// Backward: a wave pair (one 128-thread workgroup) owns one head and a strided set of windows.  The rows of window it+1
// are requested into registers with buffer loads (pad / idle slots read zeros through an out-of-range offset -- no branch
// around a load) and the slot map of window it+2 while window `it` is computed; all operand images stay resident in LDS,
// the head's bias fragment lives in registers across windows; dQ, dK, dV leave through an LDS transpose as 16-byte rows
// (dropped for pad slots, whose dK / dV rows are summed for the qkv-bias gradient).  Each (window, head) is shared by a
// PAIR of waves.  The LDS images of a window-head (34.6 KiB in bf16) limit a CU to four of them, so one wave per
// window-head means one wave per SIMD and nothing to hide LDS / MFMA / transcendental latency behind.  Splitting the
// 64 query columns (and the 64 key rows of the dK / dV outputs) between two waves halves every wave's register
// footprint (<= 256 VGPRs -> two waves per SIMD) and its dependent instruction chains, at the price of five
// two-wave barriers per window.  Wave w owns query tiles {2w, 2w+1} for P, dS, dQ and the bias gradient, and key
// tiles {2w, 2w+1} for dK and dV; everything else (prefetch, buffer loads / stores, pad-slot sums) is as in bwd2.
template <typename T, bool USE_TR, int HDIM>
__global__ __launch_bounds__(128, (HDIM == 32 ? 2 : 1)) void attn_bwd3_kernel(const T* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                          const int* __restrict__ win2tok, int L, const T* __restrict__ dout,
                                                          const float* __restrict__ bias_frag, const int* __restrict__ region_ids,
                                                          int nW, int Bw, int N, int nH, float scale, int parts,
                                                          T* __restrict__ dqkv, float* __restrict__ dbias_ws,
                                                          float* __restrict__ dpad_ws) {
    using Cfg = AttnCfgH<T, HDIM>;
    constexpr int KS = HDIM / 32, DT = HDIM / 16;  // k-steps over the head dim, 16-wide output column tiles
    constexpr int LDQ = Cfg::LDQ, LDP = Cfg::LDP, VEC = Cfg::VEC, ES = sizeof(T);
    constexpr int VPR = HDIM / VEC;          // 16-byte vectors per [slot][HDIM] row
    constexpr int NV = NP * VPR / 128;     // row vectors per thread per matrix (loads: 128 threads cover 64 slots)
    constexpr int LSTEP = 128 / VPR;       // slot stride between a thread's load vectors
    constexpr int NS = 32 * VPR / 64;      // row vectors per lane per 32-row output tile (stores)
    constexpr int SSTEP = 64 / VPR;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    T* base = reinterpret_cast<T*>(smem_raw);
    T* Qs = base;
    T* Ks = base + Cfg::QK_ELEMS;
    T* Vs = base + 2 * Cfg::QK_ELEMS;
    T* Os = base + 3 * Cfg::QK_ELEMS;
    T* Sg = base + 4 * Cfg::QK_ELEMS + w * (32 * LDQ);  // this wave's half of the output staging image
    T* Ps = base + 5 * Cfg::QK_ELEMS;

    const long unit = xcd_contiguous_id(blockIdx.x, gridDim.x);  // one (head, part) per workgroup; the heads of a part share an XCD
    const bool unit_ok = unit < (long)parts * nH;
    const int h = (int)(unit % nH);
    const int part = (int)(unit / nH);
    const int C = nH * HDIM;
    const bool masked = region_ids != nullptr;
    const int lrow0 = tid / VPR, dv = tid % VPR;         // load rows: lrow0 + LSTEP*i
    const int srow0 = 32 * w + lane / VPR;                // store rows: srow0 + SSTEP*i  (dv is the same: 64 % VPR == 0)

    f32x4 bias_r[4][2];
    {
        const float* bias_f = bias_frag + (long)h * FRAG_ELEMS;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) bias_r[i][jl] = *reinterpret_cast<const f32x4*>(bias_f + ((i * 4 + 2 * w + jl) * 64 + lane) * 4);
    }
    Vec16<T> padq, padk_v, padv_v;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        padq.set(e, qkv_bias[h * HDIM + dv * VEC + e]);
        padq.set(e, padq.get(e) * scale);
        padk_v.set(e, qkv_bias[C + h * HDIM + dv * VEC + e]);
        padv_v.set(e, qkv_bias[2 * C + h * HDIM + dv * VEC + e]);
    }
    f32x4 db[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jl = 0; jl < 2; ++jl) db[i][jl] = f32x4{0.f, 0.f, 0.f, 0.f};
    float padk[4 * DT], padv[4 * DT];  // pad-slot rows of dK / dV summed: head channels 16 dt + 4g + r of this lane's slots
#pragma unroll
    for (int e = 0; e < 4 * DT; ++e) padk[e] = padv[e] = 0.f;

    const int iters = (Bw + parts - 1) / parts;
    auto win_of = [&](int it, bool& act) -> int {
        const int bw = part + it * parts;
        act = unit_ok && it < iters && bw < Bw;
        return act ? bw : 0;
    };
    auto load_map = [&](int it, int& tok, int& reg) {
        bool act;
        const int bw = win_of(it, act);
        tok = (act && lane < N) ? win2tok[(long)(bw % nW) * N + lane] : -1;
        reg = (masked && act && lane < N) ? region_ids[(long)(bw % nW) * N + lane] : -1;
    };
    struct Win {
        u32x4 q[NV], k[NV], v[NV], o[NV];
        int ltok[NV];  // token rows of the slots this thread loads
        int mytok, myreg;
        long tok_base;
        bool active;
    };
    auto issue_rows = [&](int it, int mytok, int myreg, Win& x) {
        const int bw = win_of(it, x.active);
        x.mytok = mytok;
        x.myreg = myreg;
        x.tok_base = (long)(bw / nW) * L;
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(qkv + x.tok_base * 3L * C), 0, (int)(L * 3L * C * ES), 0x00020000);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dout + x.tok_base * (long)C), 0, (int)(L * (long)C * ES), 0x00020000);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int tok = __shfl(mytok, lrow0 + LSTEP * i, 64);
            x.ltok[i] = tok;
            const int vq = tok >= 0 ? tok * 3 * C * ES + dv * 16 : OOB;
            const int vo = tok >= 0 ? tok * C * ES + dv * 16 : OOB;
            x.q[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, vq, h * HDIM * ES, 0);
            x.k[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, vq, (C + h * HDIM) * ES, 0);
            x.v[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, vq, (2 * C + h * HDIM) * ES, 0);
            x.o[i] = __builtin_amdgcn_raw_buffer_load_b128(ro, vo, h * HDIM * ES, 0);
        }
    };

    Win cur, nxt;
    int tok1, reg1, tok2 = -1, reg2 = -1;
    load_map(0, tok1, reg1);
    issue_rows(0, tok1, reg1, cur);
    load_map(1, tok1, reg1);

    for (int it = 0; it < iters; ++it) {
        __syncthreads();  // the other wave is done with the previous window's images
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int t = lrow0 + LSTEP * i;
            const bool padslot = cur.active && t < N && cur.ltok[i] < 0;
            Vec16<T> xq, xk, xv, xo;
            xq.v = __builtin_bit_cast(decltype(xq.v), cur.q[i]);
            xk.v = __builtin_bit_cast(decltype(xk.v), cur.k[i]);
            xv.v = __builtin_bit_cast(decltype(xv.v), cur.v[i]);
            xo.v = __builtin_bit_cast(decltype(xo.v), cur.o[i]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) xq.set(e, xq.get(e) * scale);
            if (padslot) {
                xq = padq;
                xk = padk_v;
                xv = padv_v;
            }
            st16<T>(Qs + t * LDQ + dv * VEC, xq);
            st16<T>(Ks + t * LDQ + dv * VEC, xk);
            st16<T>(Vs + t * LDQ + dv * VEC, xv);
            st16<T>(Os + t * LDQ + dv * VEC, xo);
        }
        const int mytok = cur.mytok, myreg = cur.myreg;
        const bool active = cur.active;
        const long tok_base = cur.tok_base;
        int stok[2];  // token rows of the two slots (32w + 16 il + c) whose result rows this lane stores
#pragma unroll
        for (int il = 0; il < 2; ++il) stok[il] = __shfl(mytok, 32 * w + 16 * il + c, 64);
        __syncthreads();  // images complete
        issue_rows(it + 1, tok1, reg1, nxt);
        load_map(it + 2, tok2, reg2);

        // ---- phase 1: this wave's two query tiles of P ----
        {
            f32x4 p[4][2];
            Frag<T> kf[4][KS], qf[2][KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) kf[i][ks] = frag_kc<T>(Ks, LDQ, 16 * i, 32 * ks, c, g);
#pragma unroll
                for (int jl = 0; jl < 2; ++jl) qf[jl][ks] = frag_kc<T>(Qs, LDQ, 16 * (2 * w + jl), 32 * ks, c, g);
            }
            int rq[2];
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) rq[jl] = __shfl(myreg, 16 * (2 * w + jl) + c, 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int rk[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) rk[r] = __shfl(myreg, 16 * i + 4 * g + r, 64);
#pragma unroll
                for (int jl = 0; jl < 2; ++jl) {
                    f32x4 b = bias_r[i][jl];
                    if (masked) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) b[r] += (rk[r] != rq[jl]) ? -100.f : 0.f;
                    }
                    p[i][jl] = b;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) mma(kf[i][ks], qf[jl][ks], p[i][jl]);
                }
            }
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) {
                float m = -3.0e38f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) m = fmaxf(m, p[i][jl][r]);
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __expf(p[i][jl][r] - m);
                        p[i][jl][r] = e;
                        s += e;
                    }
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                const float inv = 1.f / s;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    p[i][jl] *= inv;
                    store_frag4<T>(Ps + (16 * (2 * w + jl) + c) * LDP + 16 * i + 4 * g, p[i][jl]);
                }
            }
        }
        __syncthreads();  // P complete (both waves' query tiles)

        // 32 result rows (slots 32w .. 32w+31) x HDIM from TRANSPOSED accumulators (acc[il][dt][r] = result[d = 16 dt + 4g + r][slot
        // 32w + 16 il + c]: the producing MFMAs take their operands exchanged): 16-byte row pieces straight from the registers --
        // no LDS transpose.  Pad-slot rows (dropped by the out-of-range offset) are summed into `padacc` [4 DT] (the values as stored)
        auto emit = [&](const f32x4 (&acc)[2][DT], float mul, int col0, float* padacc) {
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dqkv + tok_base * 3L * C, 0, (int)(L * 3L * C * ES), 0x00020000);
#pragma unroll
            for (int il = 0; il < 2; ++il) {
                const int t = 32 * w + 16 * il + c;
                const int tok = stok[il];
                const bool ok = active && tok >= 0;
                f32x4 v[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    v[dt] = acc[il][dt] * mul;
                    if constexpr (sizeof(T) == 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[dt][r] = (float)(bf16)v[dt][r];
                    }
                }
                if constexpr (sizeof(T) == 2) {
#pragma unroll
                    for (int dt = 0; dt < DT; dt += 2) {
                        const esvit_u32x4 x = esvit_pack_tile_pair_bf16(v[dt], v[dt + 1]);
                        buffer_store_b128(x, rd, ok ? tok * 3 * C * ES + (16 * dt + esvit_tile_pair_ch0(g)) * ES : OOB, (col0 + h * HDIM) * ES);
                    }
                } else {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) buffer_store_b128(v[dt], rd, ok ? tok * 3 * C * ES + (16 * dt + 4 * g) * ES : OOB, (col0 + h * HDIM) * ES);
                }
                if (padacc && active && t < N && tok < 0) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) padacc[4 * dt + r] += v[dt][r];
                }
            }
        };

        // ---- phase 2: dV rows of this wave's key tiles = P^T dO (all queries) ----
        {
            f32x4 acc[2][DT];
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) acc[il][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {  // reduction over the 64 queries
                Frag<T> bo[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) bo[dt] = frag_ks<T, USE_TR>(Os, LDQ, 16 * dt, 32 * ks, c, g);
#pragma unroll
                for (int il = 0; il < 2; ++il) {
                    const Frag<T> a = frag_ks<T, USE_TR>(Ps, LDP, 16 * (2 * w + il), 32 * ks, c, g);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) mma(bo[dt], a, acc[il][dt]);  // (transposed: emit)
                }
            }
            emit(acc, 1.f, 2 * C, padv);
        }
        __syncthreads();  // both waves have read P: its rows may now be overwritten with dS
        // ---- dP^T = V dO^T and dS = P o (dP - delta) for this wave's query tiles ----
        {
            Frag<T> vf[4][KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i) vf[i][ks] = frag_kc<T>(Vs, LDQ, 16 * i, 32 * ks, c, g);
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) {
                const int j = 2 * w + jl;
                Frag<T> of[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) of[ks] = frag_kc<T>(Os, LDQ, 16 * j, 32 * ks, c, g);
                f32x4 dpj[4], pj[4];
                float d = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dpj[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) mma(vf[i][ks], of[ks], dpj[i]);
                    pj[i] = load_frag4<T>(Ps + (16 * j + c) * LDP + 16 * i + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) d += pj[i][r] * dpj[i][r];
                }
                d += __shfl_xor(d, 16, 64);
                d += __shfl_xor(d, 32, 64);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 ds = pj[i] * (dpj[i] - d);
                    if (active) db[i][jl] += ds;
                    store_frag4<T>(Ps + (16 * j + c) * LDP + 16 * i + 4 * g, ds);
                }
            }
        }
        __syncthreads();  // dS complete

        // ---- phase 3: dQ rows of this wave's query tiles = scale dS K;  dK rows of its key tiles = dS^T (scale q) ----
        {
            f32x4 aq[2][DT], ak[2][DT];
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    aq[il][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    ak[il][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {  // reduction over the 64 keys (dQ) / the 64 queries (dK)
                Frag<T> kb[DT], qb[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    kb[dt] = frag_ks<T, USE_TR>(Ks, LDQ, 16 * dt, 32 * ks, c, g);
                    qb[dt] = frag_ks<T, USE_TR>(Qs, LDQ, 16 * dt, 32 * ks, c, g);
                }
#pragma unroll
                for (int il = 0; il < 2; ++il) {
                    const Frag<T> a = frag_kc<T>(Ps, LDP, 16 * (2 * w + il), 32 * ks, c, g);
                    const Frag<T> at = frag_ks<T, USE_TR>(Ps, LDP, 16 * (2 * w + il), 32 * ks, c, g);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        mma(kb[dt], a, aq[il][dt]);  // (transposed: emit)
                        mma(qb[dt], at, ak[il][dt]);
                    }
                }
            }
            emit(aq, scale, 0, nullptr);
            emit(ak, 1.f, C, padk);
        }
        cur = nxt;
        tok1 = tok2;
        reg1 = reg2;
    }

    if (unit_ok) {
        float* ws = dbias_ws + ((long)part * nH + h) * FRAG_ELEMS;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) *reinterpret_cast<f32x4*>(ws + ((i * 4 + 2 * w + jl) * 64 + lane) * 4) = db[i][jl];
    }
    // pad-row sums: over the 16 slots of a lane group (DPP row reduction), then over the two waves through LDS
#pragma unroll
    for (int e = 0; e < 4 * DT; ++e) {
        padk[e] = row16_sum(padk[e]);
        padv[e] = row16_sum(padv[e]);
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_raw);  // [2 waves][k|v][HDIM]
    if (c == 0) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                red[(w * 2 + 0) * HDIM + 16 * dt + 4 * g + r] = padk[4 * dt + r];
                red[(w * 2 + 1) * HDIM + 16 * dt + 4 * g + r] = padv[4 * dt + r];
            }
    }
    __syncthreads();
    if (unit_ok && tid < 2 * HDIM) {
        const int kv = tid / HDIM, d = tid % HDIM;
        dpad_ws[(long)part * 2 * C + kv * C + h * HDIM + d] = red[(0 * 2 + kv) * HDIM + d] + red[(1 * 2 + kv) * HDIM + d];
    }
}


// bias_frag[h][frag] from the (2ws-1)^2 x nH table (swin_transformer.py:133-135); the relative-position index is computed in
// closed form (a(q) - a(key) + (ws-1) 2ws), same values as the relative_position_index buffer; key columns >= N get -1e30
__global__ void relpos_bias_frag_from_table_kernel(const float* __restrict__ table, int ws, int N, int nH, float* __restrict__ bias_frag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nH * FRAG_ELEMS) return;
    const int h = i / FRAG_ELEMS, e = i % FRAG_ELEMS;
    const int r = e & 3, lane = (e >> 2) & 63, f = e >> 8;
    const int c = lane & 15, g = lane >> 4;
    const int q = 16 * (f & 3) + c, key = 16 * (f >> 2) + 4 * g + r;
    float v = 0.f;
    if (key >= N) v = -1.0e30f;
    else if (q < N) {
        const int w2 = 2 * ws - 1;
        const int idx = (q / ws - key / ws + ws - 1) * w2 + (q % ws - key % ws + ws - 1);
        v = table[(long)idx * nH + h];
    }
    bias_frag[i] = v;
}

// dtable[index[q,key]][h] += sum over the `parts` per-wave slabs of ws[part][h][frag(q,key)], in ONE launch (the bias-table gradient sits
// between the large kernels of the backward chain: every launch there is ~10 us of step time, profiles/r06_finish_offchain_ab.txt; rounds 1-5
// ran a partial_reduce launch and a scatter launch): block = 32 frag-layout columns x 8 slices of the slabs (the shape of
// partial_reduce_kernel), then the eight slice sums of a column are folded and scattered by its first thread.
// nt = 16-key tiles per window side (4 for <= 64 tokens, NT for 14 x 14): column e of head h is (q, key) with
// f = e / 256 = (key / 16) * nt + q / 16, lane = (e / 4) % 64 = ((key % 16) / 4) * 16 + q % 16, r = e % 4 = key % 4.
__global__ __launch_bounds__(256) void relpos_bias_bwd_fold_kernel(const float* __restrict__ ws, int parts, int nt, const long* __restrict__ index,
                                                                    int N, int nH, float* __restrict__ dtable) {
    __shared__ float sm[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int fe = nt * nt * 256;
    const long ld = (long)nH * fe;
    const int c = blockIdx.x * 32 + tx;  // < nH * fe: the launch covers whole heads (fe % 32 == 0)
    const int h = c / fe, e = c % fe;
    const int f = e >> 8, lane = (e >> 2) & 63, r = e & 3;
    const int key = (f / nt) * 16 + (lane >> 4) * 4 + r, q = (f % nt) * 16 + (lane & 15);
    const bool live = q < N && key < N;
    float s = 0.f;
    if (live) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int b = ty;
        for (; b + 24 < parts; b += 32) {
            s0 += ws[(long)b * ld + c];
            s1 += ws[(long)(b + 8) * ld + c];
            s2 += ws[(long)(b + 16) * ld + c];
            s3 += ws[(long)(b + 24) * ld + c];
        }
        for (; b < parts; b += 8) s0 += ws[(long)b * ld + c];
        s = (s0 + s1) + (s2 + s3);
    }
    sm[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && live) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sm[k][tx];
        atomicAdd(dtable + index[q * N + key] * nH + h, t);
    }
}

inline int bwd_parts(int Bw, int nH) {
    // as many wave pairs as the chip keeps resident and NOT ONE MORE: 256 CUs x 4 two-wave workgroups (256 registers, 34.6 KB
    // of LDS each); parts * nH workgroups are launched, so round DOWN -- a 1025th workgroup would start only when another has
    // finished all its windows and run a second round alone
    int parts = 1024 / nH;
    if (parts > Bw) parts = Bw;
    if (parts < 1) parts = 1;
    return parts;
}

}  // namespace

#define STREAM(s_) hipStream_t stream = reinterpret_cast<hipStream_t>(s_)


int esvit_big_relpos_bias_bwd(const float* dbias_ws, int parts, const int64_t* index, int N, int nH, int table_rows, float* dtable,
                              int accumulate, hipStream_t stream);
int esvit_big_npb();

// fold `parts` frag-layout slabs ([parts][nH][nt * nt * 256]) and scatter the sums into dtable (window_attn_big.hip calls it with nt = NT)
int esvit_i_relpos_fold(const float* dbias_ws, int parts, int nt, const int64_t* index, int N, int nH, float* dtable, hipStream_t stream) {
    hipLaunchKernelGGL(relpos_bias_bwd_fold_kernel, dim3(nH * nt * nt * 256 / 32), dim3(256), 0, stream, dbias_ws, parts, nt, (const long*)index, N,
                       nH, dtable);
    ESVIT_CHECK_LAUNCH("relpos_bias_bwd");
    return ESVIT_OK;
}

extern "C" int esvit_relpos_bias_bwd(const float* dbias_ws, int parts, const int64_t* index, int N, int nH, int table_rows,
                                     float* dtable, int accumulate, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(dbias_ws && index && dtable && parts > 0 && N > 0 && N <= esvit_big_npb() && nH > 0 && table_rows > 0,
                    "esvit_relpos_bias_bwd: bad args");
#ifdef ESVIT_PROBE_SKIP_FINISH  // timing probe (WRONG gradients), see norm.hip
    return ESVIT_OK;
#endif
    if (N > NP) return esvit_big_relpos_bias_bwd(dbias_ws, parts, index, N, nH, table_rows, dtable, accumulate, stream);
    hipError_t e = accumulate ? hipSuccess : hipMemsetAsync(dtable, 0, (size_t)table_rows * nH * sizeof(float), stream);
    if (e != hipSuccess) {
        esvit_set_error("esvit_relpos_bias_bwd: memset failed: %s", hipGetErrorString(e));
        return ESVIT_ERR_HIP;
    }
    // fold the per-wave partial slabs and scatter in one launch (dbias_ws is left as it was)
    static_assert(FRAG_ELEMS == 4 * 4 * 256, "frag layout of the <= 64-token kernels");
    return esvit_i_relpos_fold(dbias_ws, parts, 4, index, N, nH, dtable, stream);
}

// 14x14-window kernels (window_attn_big.hip)
int esvit_big_frag_elems();
int esvit_big_npb();
int esvit_big_parts(int Bw, int nH);
int esvit_big_pad_rows(int Bw, int nH, int dtype);
int esvit_big_attn_fwd(int dtype, const void* qkv, const float* qkv_bias, const int32_t* win2tok, int L, const float* rel_table, int ws,
                       float* bias_frag_ws, const int32_t* region_ids, int nW, int nB, int N, int nH, int hd, float scale, void* out, float* lse,
                       float* attn_out, hipStream_t stream);
int esvit_big_attn_bwd(int dtype, const void* qkv, const float* qkv_bias, const int32_t* win2tok, int L, const void* dout,
                       const void* fout, const float* lse, const float* rel_table, int ws, float* bias_frag_ws, const int32_t* region_ids,
                       int nW, int nB, int N, int nH, int hd, float scale, void* dqkv, float* dbias_ws, float* dpad_ws, hipStream_t stream);
int esvit_big_relpos_bias_bwd(const float* dbias_ws, int parts, const int64_t* index, int N, int nH, int table_rows, float* dtable,
                              int accumulate, hipStream_t stream);

// answers of esvit_query (lib.cpp)
int esvit_i_attn_frag_elems(int N) { return N <= NP ? FRAG_ELEMS : (N <= esvit_big_npb() ? esvit_big_frag_elems() : -1); }
int esvit_i_attn_lse_elems(int N) { return N <= NP ? 0 : esvit_big_npb(); }
int esvit_i_attn_bwd_parts(int N, int Bw, int nH) { return N <= NP ? bwd_parts(Bw, nH) : esvit_big_parts(Bw, nH); }
int esvit_i_attn_bwd_pad_rows(int dtype, int N, int Bw, int nH) { return N <= NP ? bwd_parts(Bw, nH) : esvit_big_pad_rows(Bw, nH, dtype); }

static int fill_bias_frag(const float* rel_table, int ws, int N, int nH, float* bias_frag_ws, hipStream_t stream) {
    // closed-form index (no index tensor needed): same values as relative_position_index
    hipLaunchKernelGGL(relpos_bias_frag_from_table_kernel, dim3(ceil_div((long)nH * FRAG_ELEMS, 256)), dim3(256), 0, stream, rel_table, ws, N,
                       nH, bias_frag_ws);
    ESVIT_CHECK_LAUNCH("relpos_bias(frag)");
    return ESVIT_OK;
}

// (attn_branch.hip fills the same fragment-order bias)
int esvit_i_fill_bias_frag(const float* rel_table, int ws, int N, int nH, float* bias_frag_ws, hipStream_t stream) {
    return fill_bias_frag(rel_table, ws, N, nH, bias_frag_ws, stream);
}

extern "C" int esvit_window_attn_fwd(int dtype, const void* qkv, const float* qkv_bias, const int32_t* win2tok, int L,
                                     const float* rel_table, int ws, float* bias_frag_ws, const int32_t* region_ids, int nW, int nB, int N,
                                     int nH, int hd, float scale, void* out, float* lse, float* attn_out, esvit_stream_t s_) {
    STREAM(s_);
    // (N < ws * ws: a "window" of the first N positions of a ws x ws grid -- the 197 / 37 tokens of a ViT crop with a zero table)
    ESVIT_CHECK_ARG(qkv && qkv_bias && win2tok && out && nB > 0 && nW > 0 && nH > 0 && L > 0 && ws > 0 && N > 0 &&
                        (N == ws * ws || (N < ws * ws && N <= esvit_big_npb())),
                    "esvit_window_attn_fwd: bad args");
    ESVIT_CHECK_ARG(hd == HD || hd == 64, "esvit_window_attn_fwd: head_dim %d unsupported (32 or 64)", hd);
    ESVIT_CHECK_ARG(dtype == ESVIT_BF16 || dtype == ESVIT_F32, "esvit_window_attn_fwd: bad dtype");
    if (N > NP)
        return esvit_big_attn_fwd(dtype, qkv, qkv_bias, win2tok, L, rel_table, ws, bias_frag_ws, region_ids, nW, nB, N, nH, hd, scale, out, lse, attn_out,
                                  stream);
    ESVIT_CHECK_ARG(bias_frag_ws != nullptr, "esvit_window_attn_fwd: 7x7 windows need the bias_frag_ws scratch");
    if (rel_table) {  // NULL: bias_frag_ws still holds the fragment-order bias an earlier call of this step put there
        int rc = fill_bias_frag(rel_table, ws, N, nH, bias_frag_ws, stream);
        if (rc != ESVIT_OK) return rc;
    }
    const int Bw = nB * nW;
    ESVIT_CHECK_ARG((long)L * 3 * nH * hd * 4 < 0x7fff0000L, "esvit_window_attn_fwd: one image's qkv rows must fit a 2 GiB buffer descriptor");
    // persistent wave pairs, one head each, at most one window per pair; exactly as many as the chip keeps resident (163
    // registers -> three waves per SIMD -> six two-wave workgroups per CU x 256 CUs; 208 registers at head_dim 64 -> four): a
    // larger grid runs a second, partly empty round (2560 workgroups were 5-12 % slower, tools/bench_attn.py)
    int parts = (hd == HD ? 1536 : 1024) / nH;  // (rounded DOWN: one workgroup more than the chip holds costs a second round)
    if (parts < 1) parts = 1;
    if (parts > Bw) parts = Bw;
#define LAUNCH_FWD(TT, HH)                                                                                                      \
    {                                                                                                                           \
        const size_t lds = (size_t)3 * AttnCfgH<TT, HH>::QK_ELEMS * sizeof(TT);                                                 \
        auto kern = attn_out ? attn_fwd4_kernel<TT, true, HH> : attn_fwd4_kernel<TT, false, HH>;                                \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
        hipLaunchKernelGGL(kern, dim3(parts * nH), dim3(128), lds, stream, (const TT*)qkv, qkv_bias, win2tok, L,                \
                           (const float*)bias_frag_ws, region_ids, nW, Bw, N, nH, scale, parts, (TT*)out, attn_out);            \
    }
    if (dtype == ESVIT_BF16) {
        if (hd == HD) LAUNCH_FWD(bf16, 32) else LAUNCH_FWD(bf16, 64)
    } else {
        if (hd == HD) LAUNCH_FWD(float, 32) else LAUNCH_FWD(float, 64)
    }
#undef LAUNCH_FWD
    ESVIT_CHECK_LAUNCH("window_attn_fwd");
    return ESVIT_OK;
}

extern "C" int esvit_window_attn_bwd(int dtype, const void* qkv, const float* qkv_bias, const int32_t* win2tok, int L, const void* dout,
                                     const void* fwd_out, const float* lse, const float* rel_table, int ws, float* bias_frag_ws,
                                     const int32_t* region_ids, int nW, int nB, int N, int nH, int hd, float scale, void* dqkv,
                                     float* dbias_ws, float* dpad_ws, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(qkv && qkv_bias && win2tok && dout && dqkv && dbias_ws && dpad_ws && nB > 0 && nW > 0 && nH > 0 && L > 0 &&
                        ws > 0 && N > 0 && (N == ws * ws || (N < ws * ws && N <= esvit_big_npb())),
                    "esvit_window_attn_bwd: bad args");
    ESVIT_CHECK_ARG(hd == HD || hd == 64, "esvit_window_attn_bwd: head_dim %d unsupported (32 or 64)", hd);
    ESVIT_CHECK_ARG(dtype == ESVIT_BF16 || dtype == ESVIT_F32, "esvit_window_attn_bwd: bad dtype");
    if (N > NP)
        return esvit_big_attn_bwd(dtype, qkv, qkv_bias, win2tok, L, dout, fwd_out, lse, rel_table, ws, bias_frag_ws, region_ids, nW, nB,
                                  N, nH, hd, scale, dqkv, dbias_ws, dpad_ws, stream);
    ESVIT_CHECK_ARG(bias_frag_ws != nullptr, "esvit_window_attn_bwd: 7x7 windows need the bias_frag_ws scratch");
    if (rel_table) {  // NULL: bias_frag_ws still holds the fragment-order bias an earlier call of this step put there
        int rc = fill_bias_frag(rel_table, ws, N, nH, bias_frag_ws, stream);
        if (rc != ESVIT_OK) return rc;
    }
    const int Bw = nB * nW;
    const int parts = bwd_parts(Bw, nH);
    ESVIT_CHECK_ARG((long)L * 3 * nH * hd * 4 < 0x7fff0000L, "esvit_window_attn_bwd: one image's qkv rows must fit a 2 GiB buffer descriptor");
#define LAUNCH_BWD(TT, TR, HH)                                                                                                  \
    {                                                                                                                           \
        const size_t lds = (size_t)AttnCfgH<TT, HH>::BWD3_PER_PAIR * sizeof(TT);                                                \
        auto kern = attn_bwd3_kernel<TT, TR, HH>;                                                                               \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
        hipLaunchKernelGGL(kern, dim3(parts * nH), dim3(128), lds, stream, (const TT*)qkv, qkv_bias, win2tok, L,                \
                           (const TT*)dout, (const float*)bias_frag_ws, region_ids, nW, Bw, N, nH, scale, parts, (TT*)dqkv,     \
                           dbias_ws, dpad_ws);                                                                                  \
    }
    if (dtype == ESVIT_BF16) {
        if (hd == HD) LAUNCH_BWD(bf16, true, 32) else LAUNCH_BWD(bf16, true, 64)
    } else {
        if (hd == HD) LAUNCH_BWD(float, false, 32) else LAUNCH_BWD(float, false, 64)
    }
#undef LAUNCH_BWD
    ESVIT_CHECK_LAUNCH("window_attn_bwd");
    return ESVIT_OK;
}
