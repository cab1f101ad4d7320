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

// stage one [N][HD] matrix of the (window, head) slice into a [NP][LDQ] LDS image (rows >= N zero).
// Rows are window slots; slot t reads token row tok_base + w2t[t] of the token-ordered matrix `g`, or -- for a
// zero-pad slot (w2t[t] < 0, swin_transformer.py:286-290) -- the constant `pad` (the qkv bias of this head: LN'd
// zero rows give qkv = bias; nullptr = zeros).  Values are optionally scaled and re-rounded (q * head_dim^-0.5).
// `mytok` holds the window's slot->token map in registers (lane l: slot l; -1 for pad slots, slots >= N and idle
// waves), so the per-slot lookups are cross-lane reads (ds_bpermute) instead of dependent global loads.
template <typename T>
__device__ __forceinline__ void stage_rows(const T* __restrict__ g, long row_stride, int mytok, long tok_base, int N, bool active,
                                           float scale, const float* __restrict__ pad, T* lds, int lane) {
    constexpr int VEC = AttnCfg<T>::VEC, LDQ = AttnCfg<T>::LDQ, VPR = HD / VEC;
#pragma unroll
    for (int i = 0; i < NP * VPR / 64; ++i) {
        const int v = lane + 64 * i;
        const int t = v / VPR, dv = v % VPR;
        const int tok = __shfl(mytok, t, 64);
        Vec16<T> x = zero16<T>();
        if (active && t < N) {
            if (tok >= 0) {
                x = ld16<T>(g + (tok_base + tok) * row_stride + dv * VEC);
            } else if (pad) {
#pragma unroll
                for (int e = 0; e < Vec16<T>::N; ++e) x.set(e, pad[dv * VEC + e]);
            }
        }
        if (scale != 1.f) {
#pragma unroll
            for (int e = 0; e < Vec16<T>::N; ++e) x.set(e, x.get(e) * scale);
        }
        st16<T>(lds + t * LDQ + dv * VEC, x);
    }
}

// scores + softmax, shared by fwd and bwd: returns P^T fragments p[ki][qj] (fp32)
// Shift mask (swin_transformer.py:249-272): mask[q][key] = 0 if region(q) == region(key) else -100.  `myreg` holds the
// window's per-slot region ids in registers (lane l: slot l), or -1 in every lane when the block is unshifted, so the
// 49x49 mask is rebuilt from 20 cross-lane reads instead of a 16 KiB load per (window, head).
template <typename T>
__device__ __forceinline__ void scores_softmax(const T* Qs, const T* Ks, const float* __restrict__ bias_f, int myreg,
                                               bool masked, int lane, int c, int g, f32x4 (&p)[4][4]) {
    constexpr int LDQ = AttnCfg<T>::LDQ;
    Frag<T> kf[4], qf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        kf[i] = frag_kc<T>(Ks, LDQ, 16 * i, 0, c, g);
        qf[i] = frag_kc<T>(Qs, LDQ, 16 * i, 0, c, g);
    }
    int rq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rq[j] = __shfl(myreg, 16 * j + c, 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int rk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rk[r] = __shfl(myreg, 16 * i + 4 * g + r, 64);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 b = *reinterpret_cast<const f32x4*>(bias_f + ((i * 4 + j) * 64 + lane) * 4);
            if (masked) {
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] += (rk[r] != rq[j]) ? -100.f : 0.f;
            }
            p[i][j] = b;
            mma(kf[i], qf[j], p[i][j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float m = -3.0e38f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, p[i][j][r]);
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(p[i][j][r] - m);
                p[i][j][r] = e;
                s += e;
            }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float inv = 1.f / s;
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i][j] *= inv;
    }
}

// write a P^T-layout fragment set to the [q][key] LDS image
template <typename T>
__device__ __forceinline__ void store_pt(T* Ps, const f32x4 (&p)[4][4], int c, int g) {
    constexpr int LDP = AttnCfg<T>::LDP;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) store_frag4<T>(Ps + (16 * j + c) * LDP + 16 * i + 4 * g, p[i][j]);
}

// -------------------------------------------------------------------------------------------------
// Forward.  One wave per (window, head); qkv / out are TOKEN-ordered: pad -> roll -> window_partition and its
// inverse (swin_transformer.py:286-325) are the slot->token map `win2tok`, applied on the fly.
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                       const int* __restrict__ win2tok, int L, const float* __restrict__ bias_frag,
                                                       const int* __restrict__ region_ids, int nW, int Bw, int N, int nH,
                                                       float scale, T* __restrict__ out, float* __restrict__ attn_out) {
    using Cfg = AttnCfg<T>;
    constexpr int LDQ = Cfg::LDQ, LDP = Cfg::LDP, VEC = Cfg::VEC;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    T* base = reinterpret_cast<T*>(smem_raw) + wave * Cfg::FWD_PER_WAVE;
    T* Qs = base;
    T* Ks = base + Cfg::QK_ELEMS;
    T* Ps = base;  // overlays Q,K once the scores are in registers
    T* Vt = base + Cfg::R1;

    const long unit = (long)blockIdx.x * 4 + wave;
    const bool active = unit < (long)Bw * nH;
    const int bw = active ? (int)(unit / nH) : 0;
    const int h = active ? (int)(unit % nH) : 0;
    const int C = nH * HD;
    const int mytok = (active && lane < N) ? win2tok[(long)(bw % nW) * N + lane] : -1;
    const long tok_base = (long)(bw / nW) * L;
    const T* src = qkv + h * HD;

    stage_rows<T>(src, 3L * C, mytok, tok_base, N, active, scale, qkv_bias + h * HD, Qs, lane);
    stage_rows<T>(src + C, 3L * C, mytok, tok_base, N, active, 1.f, qkv_bias + C + h * HD, Ks, lane);
    {  // V transposed: Vt[d][key]
        constexpr int VPR = HD / VEC;
        const float* padv = qkv_bias + 2 * C + h * HD;
#pragma unroll
        for (int i = 0; i < NP * VPR / 64; ++i) {
            const int v = lane + 64 * i;
            const int t = v / VPR, dv = v % VPR;
            const int tok = __shfl(mytok, t, 64);
            Vec16<T> x = zero16<T>();
            if (active && t < N) {
                if (tok >= 0) {
                    x = ld16<T>(src + 2 * C + (tok_base + tok) * 3L * C + dv * VEC);
                } else {
#pragma unroll
                    for (int e = 0; e < Vec16<T>::N; ++e) x.set(e, padv[dv * VEC + e]);
                }
            }
#pragma unroll
            for (int e = 0; e < Vec16<T>::N; ++e) Vt[(dv * VEC + e) * LDP + t] = from_f32<T>(x.get(e));
        }
    }
    __builtin_amdgcn_wave_barrier();  // each wave owns its LDS slab: LDS instructions of one wave execute in order

    f32x4 p[4][4];
    const float* bias_f = bias_frag + (long)h * FRAG_ELEMS;
    const bool masked = region_ids != nullptr;
    const int myreg = (masked && active && lane < N) ? region_ids[(long)(bw % nW) * N + lane] : -1;
    scores_softmax<T>(Qs, Ks, bias_f, myreg, masked, lane, c, g, p);

    if (attn_out && active) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = 16 * j + c, key = 16 * i + 4 * g + r;
                    if (q < N && key < N) attn_out[((unit * N) + q) * N + key] = p[i][j][r];
                }
    }
    __builtin_amdgcn_wave_barrier();  // the fragment reads of Q,K precede the P writes that overlay them
    store_pt<T>(Ps, p, c, g);
    __builtin_amdgcn_wave_barrier();

    f32x4 o[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        o[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        Frag<T> vf[2];
        vf[0] = frag_kc<T>(Vt, LDP, 0, 32 * ks, c, g);
        vf[1] = frag_kc<T>(Vt, LDP, 16, 32 * ks, c, g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const Frag<T> pf = frag_kc<T>(Ps, LDP, 16 * i, 32 * ks, c, g);
            mma(pf, vf[0], o[i][0]);
            mma(pf, vf[1], o[i][1]);
        }
    }
    // output rows leave through an LDS transpose as 16-byte vectors (the P image is dead once the MFMAs above have
    // their operands): 4 stores per lane instead of 32 two-byte ones
    __builtin_amdgcn_wave_barrier();
    T* Og = base;  // [NP][LDQ], overlays P
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Og[(16 * i + 4 * g + r) * LDQ + c] = from_f32<T>(o[i][0][r]);
            Og[(16 * i + 4 * g + r) * LDQ + 16 + c] = from_f32<T>(o[i][1][r]);
        }
    __builtin_amdgcn_wave_barrier();
    {
        constexpr int VPR = HD / VEC;
        T* dst = out + h * HD;
#pragma unroll
        for (int i = 0; i < NP * VPR / 64; ++i) {
            const int v = lane + 64 * i;
            const int t = v / VPR, dv = v % VPR;
            const int tok = __shfl(mytok, t, 64);  // -1 for pad slots, slots >= N and idle waves
            if (tok >= 0) st16<T>(dst + (tok_base + tok) * (long)C + dv * VEC, ld16<T>(Og + t * LDQ + dv * VEC));
        }
    }
}

// store a [slot][d] result tile (D layout: row = 16i+4g+r, cols c / 16+c) to the token-ordered matrix; rows of
// zero-pad slots are summed into `pad` (they are gradients of the qkv bias)
template <typename T>
__device__ __forceinline__ void store_tok_rows(const f32x4 (&acc)[4][2], float mul, T* __restrict__ dst, long row_stride,
                                               int mytok, long tok_base, int N, bool active, f32x2* pad, int c, int g) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = 16 * i + 4 * g + r;
            const int tok = __shfl(mytok, t, 64);
            if (!active || t >= N) continue;
            const float v0 = acc[i][0][r] * mul, v1 = acc[i][1][r] * mul;
            if (tok >= 0) {
                T* rowp = dst + (tok_base + tok) * row_stride;
                rowp[c] = from_f32<T>(v0);
                rowp[16 + c] = from_f32<T>(v1);
            } else if (pad) {
                (*pad)[0] += v0;
                (*pad)[1] += v1;
            }
        }
}

// -------------------------------------------------------------------------------------------------
// Backward.  Block = 4 waves; wave `wv` (global) owns head h = wv % nH and the windows bw = wv / nH + k * parts.
// LDS per wave: two [64][32] operand images (Q,K then V,dO then Q,K again) + the [64][64] P / dS image = 19 KiB,
// so 8 waves fit a CU.  Partials written once per wave: bias gradient (frag layout) and the dK/dV sums of zero-pad
// slots.
template <typename T, bool USE_TR, int MINW>
__global__ __launch_bounds__(256, MINW) void attn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                          const int* __restrict__ win2tok, int L, const T* __restrict__ dout,
                                                          const float* __restrict__ bias_frag, const int* __restrict__ region_ids,
                                                          int nW, int Bw, int N, int nH, float scale, int parts,
                                                          T* __restrict__ dqkv, float* __restrict__ dbias_ws,
                                                          float* __restrict__ dpad_ws) {
    using Cfg = AttnCfg<T>;
    constexpr int LDQ = Cfg::LDQ, LDP = Cfg::LDP;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    T* base = reinterpret_cast<T*>(smem_raw) + wave * Cfg::BWD_PER_WAVE;
    T* bufA = base;                      // scale*Q, then V, then scale*Q
    T* bufB = base + Cfg::QK_ELEMS;      // K, then dO, then K
    T* Ps = base + 2 * Cfg::QK_ELEMS;    // P, then dS  ([q][key])

    const long wv = (long)blockIdx.x * 4 + wave;
    const bool wave_ok = wv < (long)parts * nH;
    const int h = (int)(wv % nH);
    const int part = (int)(wv / nH);
    const int C = nH * HD;
    const float* bias_f = bias_frag + (long)h * FRAG_ELEMS;
    const T* src = qkv + h * HD;
    T* dst = dqkv + h * HD;

    f32x4 db[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) db[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 padk = {0.f, 0.f}, padv = {0.f, 0.f};

    const int iters = (Bw + parts - 1) / parts;
    for (int it = 0; it < iters; ++it) {
        const int bw = part + it * parts;
        const bool active = wave_ok && bw < Bw;
        const int bwc = active ? bw : 0;
        const int mytok = (active && lane < N) ? win2tok[(long)(bwc % nW) * N + lane] : -1;
        const long tok_base = (long)(bwc / nW) * L;
        const bool masked = region_ids != nullptr;
        const int myreg = (masked && active && lane < N) ? region_ids[(long)(bwc % nW) * N + lane] : -1;

        // ---- phase 1: P = softmax(scale q k^T + bias + mask) ----
        __syncthreads();  // previous iteration's reads of bufA/bufB/Ps are complete
        stage_rows<T>(src, 3L * C, mytok, tok_base, N, active, scale, qkv_bias + h * HD, bufA, lane);
        stage_rows<T>(src + C, 3L * C, mytok, tok_base, N, active, 1.f, qkv_bias + C + h * HD, bufB, lane);
        __syncthreads();
        {
            f32x4 p[4][4];  // dies here: the dS step re-reads P from its LDS image, which keeps the kernel at 2 waves/SIMD
            scores_softmax<T>(bufA, bufB, bias_f, myreg, masked, lane, c, g, p);
            store_pt<T>(Ps, p, c, g);
        }
        __syncthreads();  // score reads of bufA/bufB done; Ps visible

        // ---- phase 2: dV = P^T dO;  dP^T = V dO^T;  dS = P o (dP - delta) ----
        stage_rows<T>(src + 2 * C, 3L * C, mytok, tok_base, N, active, 1.f, qkv_bias + 2 * C + h * HD, bufA, lane);
        stage_rows<T>(dout + h * HD, (long)C, mytok, tok_base, N, active, 1.f, nullptr, bufB, lane);
        __syncthreads();
        {
            f32x4 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const Frag<T> b0 = frag_ks<T, USE_TR>(bufB, LDQ, 0, 32 * ks, c, g);
                const Frag<T> b1 = frag_ks<T, USE_TR>(bufB, LDQ, 16, 32 * ks, c, g);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const Frag<T> a = frag_ks<T, USE_TR>(Ps, LDP, 16 * i, 32 * ks, c, g);
                    mma(a, b0, acc[i][0]);
                    mma(a, b1, acc[i][1]);
                }
            }
            store_tok_rows<T>(acc, 1.f, dst + 2 * C, 3L * C, mytok, tok_base, N, active, &padv, c, g);
        }
        __syncthreads();  // dV's reads of Ps (= P) are complete: its rows may now be overwritten with dS
        // dP^T = V dO^T and dS = P o (dP - delta), one query tile (16 columns) at a time to keep only 16 dP
        // registers live; dS goes straight into the [q][key] image
        {
            Frag<T> vf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) vf[i] = frag_kc<T>(bufA, LDQ, 16 * i, 0, c, g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const Frag<T> of = frag_kc<T>(bufB, LDQ, 16 * j, 0, c, g);
                f32x4 dpj[4], pj[4];
                float d = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dpj[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    mma(vf[i], of, dpj[i]);
                    pj[i] = load_frag4<T>(Ps + (16 * j + c) * LDP + 16 * i + 4 * g);  // this lane's own P elements
#pragma unroll
                    for (int r = 0; r < 4; ++r) d += pj[i][r] * dpj[i][r];
                }
                d += __shfl_xor(d, 16, 64);
                d += __shfl_xor(d, 32, 64);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 ds = pj[i] * (dpj[i] - d);
                    if (active) db[i][j] += ds;
                    store_frag4<T>(Ps + (16 * j + c) * LDP + 16 * i + 4 * g, ds);
                }
            }
        }
        __syncthreads();  // dS image complete; reads of bufA (= V), bufB (= dO) are complete

        // ---- phase 3: dQ = scale * dS K;  dK = dS^T (scale q) ----
        stage_rows<T>(src, 3L * C, mytok, tok_base, N, active, scale, qkv_bias + h * HD, bufA, lane);
        stage_rows<T>(src + C, 3L * C, mytok, tok_base, N, active, 1.f, qkv_bias + C + h * HD, bufB, lane);
        __syncthreads();
        {
            f32x4 aq[4][2], ak[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                aq[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                aq[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                ak[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                ak[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const Frag<T> kb0 = frag_ks<T, USE_TR>(bufB, LDQ, 0, 32 * ks, c, g);
                const Frag<T> kb1 = frag_ks<T, USE_TR>(bufB, LDQ, 16, 32 * ks, c, g);
                const Frag<T> qb0 = frag_ks<T, USE_TR>(bufA, LDQ, 0, 32 * ks, c, g);
                const Frag<T> qb1 = frag_ks<T, USE_TR>(bufA, LDQ, 16, 32 * ks, c, g);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const Frag<T> a = frag_kc<T>(Ps, LDP, 16 * i, 32 * ks, c, g);
                    mma(a, kb0, aq[i][0]);
                    mma(a, kb1, aq[i][1]);
                    const Frag<T> at = frag_ks<T, USE_TR>(Ps, LDP, 16 * i, 32 * ks, c, g);
                    mma(at, qb0, ak[i][0]);
                    mma(at, qb1, ak[i][1]);
                }
            }
            // dQ of a zero-pad slot is exactly 0 (its dO row is 0), so only dK needs the pad accumulator
            store_tok_rows<T>(aq, scale, dst, 3L * C, mytok, tok_base, N, active, nullptr, c, g);
            store_tok_rows<T>(ak, 1.f, dst + C, 3L * C, mytok, tok_base, N, active, &padk, c, g);
        }
    }
    // column c (and 16+c) sums over this lane's rows -> reduce the 4 row groups g
    padk[0] += __shfl_xor(padk[0], 16, 64); padk[0] += __shfl_xor(padk[0], 32, 64);
    padk[1] += __shfl_xor(padk[1], 16, 64); padk[1] += __shfl_xor(padk[1], 32, 64);
    padv[0] += __shfl_xor(padv[0], 16, 64); padv[0] += __shfl_xor(padv[0], 32, 64);
    padv[1] += __shfl_xor(padv[1], 16, 64); padv[1] += __shfl_xor(padv[1], 32, 64);
    if (wave_ok) {
        float* ws = dbias_ws + ((long)part * nH + h) * FRAG_ELEMS;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(ws + ((i * 4 + j) * 64 + lane) * 4) = db[i][j];
        if (g == 0) {
            float* pw = dpad_ws + (long)part * 2 * C + h * HD;  // [k | v][nH][hd]
            pw[c] = padk[0];
            pw[16 + c] = padk[1];
            pw[C + c] = padv[0];
            pw[C + 16 + c] = padv[1];
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Forward, second generation (the default): the first-generation kernel above spent 64% of its wave cycles parked
// on a serial chain (slot map -> row loads -> LDS -> compute -> store) with one window per wave.  Here a wave owns one
// head and a strided set of windows, requests the q, k, v rows of window it+1 (and the slot map of window it+2) while
// it computes window `it`, synchronises only with itself, and stores 16-byte rows.  Same math, same LDS images.
typedef unsigned int u32x4_f __attribute__((ext_vector_type(4)));

template <typename T, bool WANT_ATTN>
__global__ __launch_bounds__(256, 2) void attn_fwd2_kernel(const T* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                          const int* __restrict__ win2tok, int L, const float* __restrict__ bias_frag,
                                                          const int* __restrict__ region_ids, int nW, int Bw, int N, int nH, float scale,
                                                          int parts, T* __restrict__ out, float* __restrict__ attn_out) {
    using Cfg = AttnCfg<T>;
    constexpr int LDQ = Cfg::LDQ, LDP = Cfg::LDP, VEC = Cfg::VEC, ES = sizeof(T);
    constexpr int VPR = HD / VEC, NV = NP * VPR / 64, RSTEP = 64 / VPR;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    T* base = reinterpret_cast<T*>(smem_raw) + wave * Cfg::FWD_PER_WAVE;
    T* Qs = base;
    T* Ks = base + Cfg::QK_ELEMS;
    T* Ps = base;  // overlays Q,K once the scores are in registers; later the output staging image
    T* Vt = base + Cfg::R1;

    const long wv = (long)blockIdx.x * 4 + wave;
    const bool wave_ok = wv < (long)parts * nH;
    const int h = (int)(wv % nH);
    const int part = (int)(wv / nH);
    const int C = nH * HD;
    const bool masked = region_ids != nullptr;
    const int row0 = lane / VPR, dv = lane % VPR;
    const float* bias_f = bias_frag + (long)h * FRAG_ELEMS;

    Vec16<T> padq, padk_v, padv_v;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        padq.set(e, qkv_bias[h * HD + dv * VEC + e]);
        padq.set(e, padq.get(e) * scale);
        padk_v.set(e, qkv_bias[C + h * HD + dv * VEC + e]);
        padv_v.set(e, qkv_bias[2 * C + h * HD + dv * VEC + e]);
    }

    const int iters = (Bw + parts - 1) / parts;
    auto win_of = [&](int it, bool& act) -> int {
        const int bw = part + it * parts;
        act = wave_ok && it < iters && bw < Bw;
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
        int rowoff[NV];
        int myreg, bw;
        long tok_base;
        bool active;
    };
    auto issue_rows = [&](int it, int mytok, int myreg, Win& w) {
        w.bw = win_of(it, w.active);
        w.myreg = myreg;
        w.tok_base = (long)(w.bw / nW) * L;
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(qkv + w.tok_base * 3L * C), 0, (int)(L * 3L * C * ES), 0x00020000);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int tok = __shfl(mytok, row0 + RSTEP * i, 64);
            w.rowoff[i] = tok;
            const int vq = tok >= 0 ? tok * 3 * C * ES + dv * 16 : OOB;
            w.q[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, vq, h * HD * ES, 0);
            w.k[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, vq, (C + h * HD) * ES, 0);
            w.v[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, vq, (2 * C + h * HD) * ES, 0);
        }
    };

    Win cur, nxt;
    int tok1, reg1, tok2 = -1, reg2 = -1;
    load_map(0, tok1, reg1);
    issue_rows(0, tok1, reg1, cur);
    load_map(1, tok1, reg1);

    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int t = row0 + RSTEP * i;
            const bool padslot = cur.active && t < N && cur.rowoff[i] < 0;
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
#pragma unroll
            for (int e = 0; e < VEC; ++e) Vt[(dv * VEC + e) * LDP + t] = from_f32<T>(xv.get(e));  // V transposed: Vt[d][key]
        }
        __builtin_amdgcn_wave_barrier();
        const bool active = cur.active;
        const int myreg = cur.myreg;
        const long tok_base = cur.tok_base;
        const long unit = (long)cur.bw * nH + h;
        int rowoff[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) rowoff[i] = cur.rowoff[i];
        issue_rows(it + 1, tok1, reg1, nxt);
        load_map(it + 2, tok2, reg2);

        f32x4 p[4][4];
        scores_softmax<T>(Qs, Ks, bias_f, myreg, masked, lane, c, g, p);
        if constexpr (WANT_ATTN) {
            if (attn_out && active) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int q = 16 * j + c, key = 16 * i + 4 * g + r;
                            if (q < N && key < N) attn_out[((unit * N) + q) * N + key] = p[i][j][r];
                        }
            }
        }
        __builtin_amdgcn_wave_barrier();  // the fragment reads of Q,K precede the P writes that overlay them
        store_pt<T>(Ps, p, c, g);
        __builtin_amdgcn_wave_barrier();

        f32x4 o[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            o[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Frag<T> vf[2];
            vf[0] = frag_kc<T>(Vt, LDP, 0, 32 * ks, c, g);
            vf[1] = frag_kc<T>(Vt, LDP, 16, 32 * ks, c, g);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const Frag<T> pf = frag_kc<T>(Ps, LDP, 16 * i, 32 * ks, c, g);
                mma(pf, vf[0], o[i][0]);
                mma(pf, vf[1], o[i][1]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        T* Og = base;  // [NP][LDQ], overlays P
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Og[(16 * i + 4 * g + r) * LDQ + c] = from_f32<T>(o[i][0][r]);
                Og[(16 * i + 4 * g + r) * LDQ + 16 + c] = from_f32<T>(o[i][1][r]);
            }
        __builtin_amdgcn_wave_barrier();
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + tok_base * (long)C, 0, (int)(L * (long)C * ES), 0x00020000);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int t = row0 + RSTEP * i;
            const Vec16<T> x = ld16<T>(Og + t * LDQ + dv * VEC);
            const int vo = (active && rowoff[i] >= 0) ? rowoff[i] * C * ES + dv * 16 : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_f, x.v), ro, vo, h * HD * ES, 0);
        }
        cur = nxt;
        tok1 = tok2;
        reg1 = reg2;
    }
}

// -------------------------------------------------------------------------------------------------
// Forward, third generation (the default): attn_fwd2_kernel with each (window, head) shared by a pair of waves (one
// 128-thread workgroup): wave w owns query tiles {2w, 2w+1} -- their score columns, softmax, P rows and output rows.
// Half the registers per wave (three or more waves per SIMD instead of two) and half the dependent chain per window;
// three two-wave barriers per window.
template <typename T, bool WANT_ATTN, int HDIM>
__global__ __launch_bounds__(128, (HDIM == 32 ? 3 : 2)) void attn_fwd3_kernel(const T* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                          const int* __restrict__ win2tok, int L, const float* __restrict__ bias_frag,
                                                          const int* __restrict__ region_ids, int nW, int Bw, int N, int nH, float scale,
                                                          int parts, T* __restrict__ out, float* __restrict__ attn_out) {
    using Cfg = AttnCfgH<T, HDIM>;
    constexpr int KS = HDIM / 32, DT = HDIM / 16;  // k-steps over the head dim, 16-wide output column tiles
    constexpr int LDQ = Cfg::LDQ, LDP = Cfg::LDP, VEC = Cfg::VEC, ES = sizeof(T);
    constexpr int VPR = HDIM / VEC, NV = NP * VPR / 128, LSTEP = 128 / VPR;
    constexpr int NS = 32 * VPR / 64, SSTEP = 64 / VPR;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    T* base = reinterpret_cast<T*>(smem_raw);
    T* Qs = base;
    T* Ks = base + Cfg::QK_ELEMS;
    T* Ps = base;  // overlays Q,K once both waves have their scores
    T* Vt = base + Cfg::R1;

    const long unit = blockIdx.x;
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
#pragma unroll
            for (int e = 0; e < VEC; ++e) Vt[(dv * VEC + e) * LDP + t] = from_f32<T>(xv.get(e));
        }
        const bool active = cur.active;
        const int mytok = cur.mytok, myreg = cur.myreg;
        const long tok_base = cur.tok_base;
        const long unit_wh = (long)cur.bw * nH + h;
        int stok[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) stok[i] = __shfl(mytok, srow0 + SSTEP * i, 64);
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
        __syncthreads();  // both waves hold their scores: Q,K may be overlaid by P
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jl = 0; jl < 2; ++jl) store_frag4<T>(Ps + (16 * (2 * w + jl) + c) * LDP + 16 * i + 4 * g, p[i][jl]);
        __builtin_amdgcn_wave_barrier();  // P rows of this wave's queries are read back by this wave only

        f32x4 o[2][DT];
#pragma unroll
        for (int il = 0; il < 2; ++il)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[il][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {  // 64 keys = two 32-deep steps
            Frag<T> vf[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) vf[dt] = frag_kc<T>(Vt, LDP, 16 * dt, 32 * ks, c, g);
#pragma unroll
            for (int il = 0; il < 2; ++il) {
                const Frag<T> pf = frag_kc<T>(Ps, LDP, 16 * (2 * w + il), 32 * ks, c, g);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) mma(pf, vf[dt], o[il][dt]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        T* Og = Ps + 32 * w * LDP;  // [32][LDQ] over this wave's own (now dead) P rows
#pragma unroll
        for (int il = 0; il < 2; ++il)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) Og[(16 * il + 4 * g + r) * LDQ + 16 * dt + c] = from_f32<T>(o[il][dt][r]);
        __builtin_amdgcn_wave_barrier();
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + tok_base * (long)C, 0, (int)(L * (long)C * ES), 0x00020000);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int tl = lane / VPR + SSTEP * i;
            const Vec16<T> x = ld16<T>(Og + tl * LDQ + dv * VEC);
            const int vo = (active && stok[i] >= 0) ? stok[i] * C * ES + dv * 16 : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_f, x.v), ro, vo, h * HDIM * ES, 0);
        }
        cur = nxt;
        tok1 = tok2;
        reg1 = reg2;
    }
}

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

// Fourth-generation forward (default): attn_fwd3_kernel with P kept in registers and V staged in its natural [key][d]
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

    const long unit = blockIdx.x;
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
        int stok[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) stok[i] = __shfl(mytok, srow0 + SSTEP * i, 64);
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
                for (int dt = 0; dt < DT; ++dt) mma(pf, vf[dt], o[il][dt]);
            }
        }
        T* Og = Qs + 32 * w * LDQ;  // [32][LDQ]: this wave's own query rows (only it read them, and they are in registers now)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int il = 0; il < 2; ++il)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) Og[(16 * il + 4 * g + r) * LDQ + 16 * dt + c] = from_f32<T>(o[il][dt][r]);
        __builtin_amdgcn_wave_barrier();
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + tok_base * (long)C, 0, (int)(L * (long)C * ES), 0x00020000);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int tl = lane / VPR + SSTEP * i;
            const Vec16<T> x = ld16<T>(Og + tl * LDQ + dv * VEC);
            const int vo = (active && stok[i] >= 0) ? stok[i] * C * ES + dv * 16 : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_f, x.v), ro, vo, h * HDIM * ES, 0);
        }
        cur = nxt;
        tok1 = tok2;
        reg1 = reg2;
    }
}

// -------------------------------------------------------------------------------------------------
// Backward, second generation (the default).  Same math and the same work split as attn_bwd_kernel (wave = one head x a
// strided set of windows), rebuilt around what the profile showed the first one to be bound by -- exposed memory
// latency, block barriers and 2-byte scattered stores, not MFMA:
//   * q, k, v, dO of window it+1 are requested into registers (buffer loads; pad / inactive slots read as zeros through
//     an out-of-range offset, so there is no branch around a load) while window `it` is being computed, and the
//     slot->token map is requested one window further ahead;
//   * all four operand images stay in LDS for the whole window (no re-staging of q, k for the last phase), the
//     relative-position bias of the wave's head lives in registers across windows;
//   * every synchronisation is wave-local (each wave owns its LDS slab; LDS instructions of a wave execute in order);
//   * dQ, dK, dV leave through an LDS transpose as 16-byte row vectors (buffer stores, dropped for pad slots, whose
//     dK / dV rows are summed for the qkv-bias gradient instead).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct Bwd2Cfg {
    using Cfg = AttnCfg<T>;
    static constexpr int PER_WAVE = 5 * Cfg::QK_ELEMS + Cfg::P_ELEMS;  // Q, K, V, dO, out-staging, P/dS
    static constexpr int WAVES = sizeof(T) == 2 ? 4 : 2;                // 138 KiB / 127 KiB per workgroup
};

template <typename T, bool USE_TR, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void attn_bwd2_kernel(const T* __restrict__ qkv, const float* __restrict__ qkv_bias,
                                                                 const int* __restrict__ win2tok, int L, const T* __restrict__ dout,
                                                                 const float* __restrict__ bias_frag, const int* __restrict__ region_ids,
                                                                 int nW, int Bw, int N, int nH, float scale, int parts,
                                                                 T* __restrict__ dqkv, float* __restrict__ dbias_ws,
                                                                 float* __restrict__ dpad_ws) {
    using Cfg = AttnCfg<T>;
    constexpr int LDQ = Cfg::LDQ, LDP = Cfg::LDP, VEC = Cfg::VEC, ES = sizeof(T);
    constexpr int VPR = HD / VEC;        // 16-byte vectors per [slot][HD] row
    constexpr int NV = NP * VPR / 64;    // row vectors per lane per matrix
    constexpr int RSTEP = 64 / VPR;      // slot stride between a lane's vectors
    constexpr int OOB = 0x7ffffff0;      // beyond any descriptor's num_records: loads return 0, stores are dropped
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    T* base = reinterpret_cast<T*>(smem_raw) + wave * Bwd2Cfg<T>::PER_WAVE;
    T* Qs = base;
    T* Ks = base + Cfg::QK_ELEMS;
    T* Vs = base + 2 * Cfg::QK_ELEMS;
    T* Os = base + 3 * Cfg::QK_ELEMS;
    T* Sg = base + 4 * Cfg::QK_ELEMS;
    T* Ps = base + 5 * Cfg::QK_ELEMS;

    const long wv = (long)blockIdx.x * WAVES + wave;
    const bool wave_ok = wv < (long)parts * nH;
    const int h = (int)(wv % nH);
    const int part = (int)(wv / nH);
    const int C = nH * HD;
    const bool masked = region_ids != nullptr;
    const int row0 = lane / VPR, dv = lane % VPR;  // this lane's row vectors: slots row0 + RSTEP*i, columns dv*VEC..

    // relative-position bias of this head, frag layout, for the whole kernel
    f32x4 bias_r[4][4];
    {
        const float* bias_f = bias_frag + (long)h * FRAG_ELEMS;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) bias_r[i][j] = *reinterpret_cast<const f32x4*>(bias_f + ((i * 4 + j) * 64 + lane) * 4);
    }
    // rows of zero-pad slots: LN'd zero rows give qkv = bias (q additionally scaled and re-rounded like real rows)
    Vec16<T> padq, padk_v, padv_v;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        padq.set(e, qkv_bias[h * HD + dv * VEC + e]);
        padq.set(e, padq.get(e) * scale);
        padk_v.set(e, qkv_bias[C + h * HD + dv * VEC + e]);
        padv_v.set(e, qkv_bias[2 * C + h * HD + dv * VEC + e]);
    }

    f32x4 db[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) db[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float padk[VEC], padv[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) padk[e] = padv[e] = 0.f;

    const int iters = (Bw + parts - 1) / parts;
    auto win_of = [&](int it, bool& act) -> int {
        const int bw = part + it * parts;
        act = wave_ok && it < iters && bw < Bw;
        return act ? bw : 0;
    };
    auto load_map = [&](int it, int& tok, int& reg) {
        bool act;
        const int bw = win_of(it, act);
        tok = (act && lane < N) ? win2tok[(long)(bw % nW) * N + lane] : -1;
        reg = (masked && act && lane < N) ? region_ids[(long)(bw % nW) * N + lane] : -1;
    };
    // per-window state that travels with the prefetched rows
    struct Win {
        u32x4 q[NV], k[NV], v[NV], o[NV];
        int rowoff[NV];  // slot's token row (tok >= 0) or -1
        int mytok, myreg;
        long tok_base;
        bool active;
    };
    auto issue_rows = [&](int it, int mytok, int myreg, Win& w) {
        const int bw = win_of(it, w.active);
        w.mytok = mytok;
        w.myreg = myreg;
        w.tok_base = (long)(bw / nW) * L;
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(qkv + w.tok_base * 3L * C), 0, (int)(L * 3L * C * ES), 0x00020000);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dout + w.tok_base * (long)C), 0, (int)(L * (long)C * ES), 0x00020000);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int tok = __shfl(mytok, row0 + RSTEP * i, 64);
            w.rowoff[i] = tok;
            const int vq = tok >= 0 ? tok * 3 * C * ES + dv * 16 : OOB;
            const int vo = tok >= 0 ? tok * C * ES + dv * 16 : OOB;
            w.q[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, vq, h * HD * ES, 0);
            w.k[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, vq, (C + h * HD) * ES, 0);
            w.v[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, vq, (2 * C + h * HD) * ES, 0);
            w.o[i] = __builtin_amdgcn_raw_buffer_load_b128(ro, vo, h * HD * ES, 0);
        }
    };

    Win cur, nxt;
    int tok1, reg1, tok2 = -1, reg2 = -1;
    load_map(0, tok1, reg1);
    issue_rows(0, tok1, reg1, cur);
    load_map(1, tok1, reg1);

    for (int it = 0; it < iters; ++it) {
        // ---- operands of window `it`: registers -> LDS images (pad slots get the bias constants) ----
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int t = row0 + RSTEP * i;
            const bool padslot = cur.active && t < N && cur.rowoff[i] < 0;
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
        __builtin_amdgcn_wave_barrier();
        const int myreg = cur.myreg;
        const bool active = cur.active;
        const long tok_base = cur.tok_base;
        int rowoff[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) rowoff[i] = cur.rowoff[i];
        // ---- request window it+1 (its map arrived during the previous window) and the map of window it+2 ----
        issue_rows(it + 1, tok1, reg1, nxt);
        load_map(it + 2, tok2, reg2);

        // ---- phase 1: P = softmax(scale q k^T + bias + mask), written as the [q][key] image ----
        {
            f32x4 p[4][4];
            Frag<T> kf[4], qf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kf[i] = frag_kc<T>(Ks, LDQ, 16 * i, 0, c, g);
                qf[i] = frag_kc<T>(Qs, LDQ, 16 * i, 0, c, g);
            }
            int rq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) rq[j] = __shfl(myreg, 16 * j + c, 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int rk[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) rk[r] = __shfl(myreg, 16 * i + 4 * g + r, 64);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 b = bias_r[i][j];
                    if (masked) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) b[r] += (rk[r] != rq[j]) ? -100.f : 0.f;
                    }
                    p[i][j] = b;
                    mma(kf[i], qf[j], p[i][j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float m = -3.0e38f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) m = fmaxf(m, p[i][j][r]);
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __expf(p[i][j][r] - m);
                        p[i][j][r] = e;
                        s += e;
                    }
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                const float inv = 1.f / s;
#pragma unroll
                for (int i = 0; i < 4; ++i) p[i][j] *= inv;
            }
            store_pt<T>(Ps, p, c, g);
        }
        __builtin_amdgcn_wave_barrier();

        // LDS transpose of a [slot][d] result tile, then 16-byte row stores; pad-slot rows go to `padacc`
        auto emit = [&](const f32x4 (&acc)[4][2], float mul, int col0, float* padacc) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Sg[(16 * i + 4 * g + r) * LDQ + c] = from_f32<T>(acc[i][0][r] * mul);
                    Sg[(16 * i + 4 * g + r) * LDQ + 16 + c] = from_f32<T>(acc[i][1][r] * mul);
                }
            __builtin_amdgcn_wave_barrier();
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dqkv + tok_base * 3L * C, 0, (int)(L * 3L * C * ES), 0x00020000);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int t = row0 + RSTEP * i;
                const Vec16<T> x = ld16<T>(Sg + t * LDQ + dv * VEC);
                const int vo = (active && rowoff[i] >= 0) ? rowoff[i] * 3 * C * ES + dv * 16 : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x.v), rd, vo, (col0 + h * HD) * ES, 0);
                if (padacc && active && t < N && rowoff[i] < 0) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) padacc[e] += x.get(e);
                }
            }
        };

        // ---- phase 2: dV = P^T dO ----
        {
            f32x4 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const Frag<T> b0 = frag_ks<T, USE_TR>(Os, LDQ, 0, 32 * ks, c, g);
                const Frag<T> b1 = frag_ks<T, USE_TR>(Os, LDQ, 16, 32 * ks, c, g);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const Frag<T> a = frag_ks<T, USE_TR>(Ps, LDP, 16 * i, 32 * ks, c, g);
                    mma(a, b0, acc[i][0]);
                    mma(a, b1, acc[i][1]);
                }
            }
            emit(acc, 1.f, 2 * C, padv);
        }
        __builtin_amdgcn_wave_barrier();  // dV's reads of Ps (= P) precede the dS writes below (same wave: in order)
        // ---- dP^T = V dO^T and dS = P o (dP - delta), one query tile at a time; dS overwrites P in place ----
        {
            Frag<T> vf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) vf[i] = frag_kc<T>(Vs, LDQ, 16 * i, 0, c, g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const Frag<T> of = frag_kc<T>(Os, LDQ, 16 * j, 0, c, g);
                f32x4 dpj[4], pj[4];
                float d = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dpj[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                    mma(vf[i], of, dpj[i]);
                    pj[i] = load_frag4<T>(Ps + (16 * j + c) * LDP + 16 * i + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) d += pj[i][r] * dpj[i][r];
                }
                d += __shfl_xor(d, 16, 64);
                d += __shfl_xor(d, 32, 64);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 ds = pj[i] * (dpj[i] - d);
                    if (active) db[i][j] += ds;
                    store_frag4<T>(Ps + (16 * j + c) * LDP + 16 * i + 4 * g, ds);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- phase 3: dQ = scale * dS K;  dK = dS^T (scale q) ----
        {
            f32x4 aq[4][2], ak[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                aq[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                aq[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                ak[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                ak[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const Frag<T> kb0 = frag_ks<T, USE_TR>(Ks, LDQ, 0, 32 * ks, c, g);
                const Frag<T> kb1 = frag_ks<T, USE_TR>(Ks, LDQ, 16, 32 * ks, c, g);
                const Frag<T> qb0 = frag_ks<T, USE_TR>(Qs, LDQ, 0, 32 * ks, c, g);
                const Frag<T> qb1 = frag_ks<T, USE_TR>(Qs, LDQ, 16, 32 * ks, c, g);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const Frag<T> a = frag_kc<T>(Ps, LDP, 16 * i, 32 * ks, c, g);
                    mma(a, kb0, aq[i][0]);
                    mma(a, kb1, aq[i][1]);
                    const Frag<T> at = frag_ks<T, USE_TR>(Ps, LDP, 16 * i, 32 * ks, c, g);
                    mma(at, qb0, ak[i][0]);
                    mma(at, qb1, ak[i][1]);
                }
            }
            // dQ of a zero-pad slot is exactly 0 (its dO row is 0), so only dK and dV feed the bias gradient
            emit(aq, scale, 0, nullptr);
            emit(ak, 1.f, C, padk);
        }
        cur = nxt;
        tok1 = tok2;
        reg1 = reg2;
    }

    if (wave_ok) {
        float* ws = dbias_ws + ((long)part * nH + h) * FRAG_ELEMS;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(ws + ((i * 4 + j) * 64 + lane) * 4) = db[i][j];
    }
    // pad-row sums: lanes with equal dv hold partial sums of the same VEC columns -> reduce over the row index bits
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
#pragma unroll
        for (int o = VPR; o < 64; o <<= 1) {
            padk[e] += __shfl_xor(padk[e], o, 64);
            padv[e] += __shfl_xor(padv[e], o, 64);
        }
    }
    if (wave_ok && lane < VPR) {
        float* pw = dpad_ws + (long)part * 2 * C + h * HD + dv * VEC;  // [k | v][nH][hd]
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            pw[e] = padk[e];
            pw[C + e] = padv[e];
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Backward, third generation (the default): attn_bwd2_kernel with each (window, head) shared by a PAIR of waves (one
// 128-thread workgroup).  The LDS images of a window-head (34.6 KiB in bf16) limit a CU to four of them, so one wave per
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

    const long unit = blockIdx.x;  // one (head, part) per workgroup
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
    float padk[VEC], padv[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) padk[e] = padv[e] = 0.f;

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
        int stok[NS];  // token rows of the slots this lane stores
#pragma unroll
        for (int i = 0; i < NS; ++i) stok[i] = __shfl(mytok, srow0 + SSTEP * i, 64);
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

        // 32 result rows (slots 32w .. 32w+31) x HDIM: LDS transpose, 16-byte row stores, pad-slot rows into `padacc`
        auto emit = [&](const f32x4 (&acc)[2][DT], float mul, int col0, float* padacc) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int il = 0; il < 2; ++il)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) Sg[(16 * il + 4 * g + r) * LDQ + 16 * dt + c] = from_f32<T>(acc[il][dt][r] * mul);
            __builtin_amdgcn_wave_barrier();
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dqkv + tok_base * 3L * C, 0, (int)(L * 3L * C * ES), 0x00020000);
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int tl = lane / VPR + SSTEP * i;  // row inside the wave's 32-row tile
                const int t = 32 * w + tl;
                const Vec16<T> x = ld16<T>(Sg + tl * LDQ + dv * VEC);
                const int vo = (active && stok[i] >= 0) ? stok[i] * 3 * C * ES + dv * 16 : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, x.v), rd, vo, (col0 + h * HDIM) * ES, 0);
                if (padacc && active && t < N && stok[i] < 0) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) padacc[e] += x.get(e);
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
                    for (int dt = 0; dt < DT; ++dt) mma(a, bo[dt], acc[il][dt]);
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
                        mma(a, kb[dt], aq[il][dt]);
                        mma(at, qb[dt], ak[il][dt]);
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
    // pad-row sums: reduce over the row bits inside the wave, then over the two waves through LDS
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
#pragma unroll
        for (int o = VPR; o < 64; o <<= 1) {
            padk[e] += __shfl_xor(padk[e], o, 64);
            padv[e] += __shfl_xor(padv[e], o, 64);
        }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem_raw);  // [2 waves][k|v][HDIM]
    if (lane < VPR) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            red[(w * 2 + 0) * HDIM + dv * VEC + e] = padk[e];
            red[(w * 2 + 1) * HDIM + dv * VEC + e] = padv[e];
        }
    }
    __syncthreads();
    if (unit_ok && tid < 2 * HDIM) {
        const int kv = tid / HDIM, d = tid % HDIM;
        dpad_ws[(long)part * 2 * C + kv * C + h * HDIM + d] = red[(0 * 2 + kv) * HDIM + d] + red[(1 * 2 + kv) * HDIM + d];
    }
}

// bias_frag[h][frag] from the (2ws-1)^2 x nH table (swin_transformer.py:133-135)
__global__ void relpos_bias_fwd_kernel(const float* __restrict__ table, const long* __restrict__ index, int N, int nH,
                                       float* __restrict__ bias_frag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nH * FRAG_ELEMS) return;
    const int h = i / FRAG_ELEMS, e = i % FRAG_ELEMS;
    const int r = e & 3, lane = (e >> 2) & 63, f = e >> 8;
    const int c = lane & 15, g = lane >> 4;
    const int q = 16 * (f & 3) + c, key = 16 * (f >> 2) + 4 * g + r;
    float v = 0.f;
    if (key >= N) v = -1.0e30f;
    else if (q < N) v = table[index[(long)q * N + key] * nH + h];
    bias_frag[i] = v;
}

// same, with the index computed in closed form (a(q) - a(key) + (ws-1) 2ws) instead of read from the index buffer
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

// dense [nW][N][N] -> frag layout (padding 0)
__global__ void dense_to_frag_kernel(const float* __restrict__ dense, int nM, int N, float* __restrict__ frag) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)nM * FRAG_ELEMS) return;
    const int w = (int)(i / FRAG_ELEMS), e = (int)(i % FRAG_ELEMS);
    const int r = e & 3, lane = (e >> 2) & 63, f = e >> 8;
    const int c = lane & 15, g = lane >> 4;
    const int q = 16 * (f & 3) + c, key = 16 * (f >> 2) + 4 * g + r;
    frag[i] = (q < N && key < N) ? dense[((long)w * N + q) * N + key] : 0.f;
}

// dtable[index[q,key]][h] += total[h][frag(q,key)]   (total = partials already summed)
__global__ void relpos_bias_bwd_kernel(const float* __restrict__ ws, int parts, const long* __restrict__ index, int N, int nH,
                                       float* __restrict__ dtable) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nH * N * N) return;
    const int h = i / (N * N), qk = i % (N * N);
    const int q = qk / N, key = qk % N;
    const int f = (key >> 4) * 4 + (q >> 4);
    const int lane = ((key & 15) >> 2) * 16 + (q & 15), r = key & 3;
    const int e = (f * 64 + lane) * 4 + r;
    float s = 0.f;
    for (int p = 0; p < parts; ++p) s += ws[((long)p * nH + h) * FRAG_ELEMS + e];
    atomicAdd(dtable + index[qk] * nH + h, s);
}

static int g_attn_fwd_impl = 4;  // 4: attn_fwd4_kernel (P in registers); 3: attn_fwd3_kernel (a window-head per wave pair); 2: attn_fwd2_kernel (persistent waves, prefetch); 1: attn_fwd_kernel
static int g_attn_bwd_impl = 3;  // 3: attn_bwd3_kernel (bwd2 with a window-head shared by two waves); 2: attn_bwd2_kernel; 1: attn_bwd_kernel

inline int bwd_parts(int Bw, int nH) {
    // enough waves to fill the chip: 256 CUs x the resident backward waves (one 4-wave workgroup per CU for the
    // second-generation kernel, two for the first), at most one window per wave
    int parts = ((g_attn_bwd_impl == 2 ? 1024 : (g_attn_bwd_impl == 3 ? 1024 : 2048)) + nH - 1) / nH;  // v3: 4 two-wave workgroups per CU
    if (parts > Bw) parts = Bw;
    if (parts < 1) parts = 1;
    return parts;
}

}  // namespace

#define STREAM(s_) hipStream_t stream = reinterpret_cast<hipStream_t>(s_)


extern "C" int esvit_relpos_bias_fwd(const float* table, const int64_t* index, int N, int nH, float* bias_frag, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(table && index && bias_frag && N > 0 && N <= NP && nH > 0, "esvit_relpos_bias_fwd: bad args (N=%d)", N);
    hipLaunchKernelGGL(relpos_bias_fwd_kernel, dim3(ceil_div((long)nH * FRAG_ELEMS, 256)), dim3(256), 0, stream, table,
                       (const long*)index, N, nH, bias_frag);
    ESVIT_CHECK_LAUNCH("relpos_bias_fwd");
    return ESVIT_OK;
}

extern "C" int esvit_dense_to_frag(const float* dense, int n_mats, int N, float* frag, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(dense && frag && n_mats > 0 && N > 0 && N <= NP, "esvit_dense_to_frag: bad args");
    hipLaunchKernelGGL(dense_to_frag_kernel, dim3(ceil_div((long)n_mats * FRAG_ELEMS, 256)), dim3(256), 0, stream, dense, n_mats,
                       N, frag);
    ESVIT_CHECK_LAUNCH("dense_to_frag");
    return ESVIT_OK;
}

int esvit_big_relpos_bias_bwd(const float* dbias_ws, int parts, const int64_t* index, int N, int nH, int table_rows, float* dtable,
                              hipStream_t stream);
int esvit_big_npb();

extern "C" int esvit_relpos_bias_bwd(const float* dbias_ws, int parts, const int64_t* index, int N, int nH, int table_rows,
                                     float* dtable, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(dbias_ws && index && dtable && parts > 0 && N > 0 && N <= esvit_big_npb() && nH > 0 && table_rows > 0,
                    "esvit_relpos_bias_bwd: bad args");
    if (N > NP) return esvit_big_relpos_bias_bwd(dbias_ws, parts, index, N, nH, table_rows, dtable, stream);
    hipError_t e = hipMemsetAsync(dtable, 0, (size_t)table_rows * nH * sizeof(float), stream);
    if (e != hipSuccess) {
        esvit_set_error("esvit_relpos_bias_bwd: memset failed: %s", hipGetErrorString(e));
        return ESVIT_ERR_HIP;
    }
    // sum the per-wave partial slabs in place into slab 0 (out may alias row 0: each column is read then written by one thread)
    if (parts > 1) {
        int rc = esvit_partial_reduce(dbias_ws, parts, nH * FRAG_ELEMS, (long)nH * FRAG_ELEMS, const_cast<float*>(dbias_ws), 0, stream);
        if (rc != ESVIT_OK) return rc;
    }
    hipLaunchKernelGGL(relpos_bias_bwd_kernel, dim3(ceil_div((long)nH * N * N, 256)), dim3(256), 0, stream, dbias_ws, 1,
                       (const long*)index, N, nH, dtable);
    ESVIT_CHECK_LAUNCH("relpos_bias_bwd");
    return ESVIT_OK;
}

static int g_attn_use_tr = 1;
static int g_attn_minw = 1;  // waves per SIMD the backward kernel is compiled for (2: 256 registers + some scratch, 1: 512 registers)
extern "C" void esvit_debug_set_attn_tr_read(int on) { g_attn_use_tr = on; }
extern "C" void esvit_debug_set_attn_bwd_waves(int w) { g_attn_minw = w; }
extern "C" void esvit_debug_set_attn_bwd_impl(int v) { g_attn_bwd_impl = v; }
extern "C" void esvit_debug_set_attn_fwd_impl(int v) { g_attn_fwd_impl = v; }

// 14x14-window kernels (window_attn_big.hip)
int esvit_big_frag_elems();
int esvit_big_npb();
int esvit_big_parts(int Bw, int nH);
int esvit_big_pad_rows(int Bw, int nH, int dtype);
int esvit_big_attn_fwd(int dtype, const void* qkv, const float* qkv_bias, const int32_t* win2tok, int L, const float* rel_table, int ws,
                       float* bias_frag_ws, const int32_t* region_ids, int nW, int nB, int N, int nH, float scale, void* out, float* lse,
                       float* attn_out, hipStream_t stream);
int esvit_big_attn_bwd(int dtype, int use_tr, const void* qkv, const float* qkv_bias, const int32_t* win2tok, int L, const void* dout,
                       const void* fout, const float* lse, const float* rel_table, int ws, float* bias_frag_ws, const int32_t* region_ids,
                       int nW, int nB, int N, int nH, float scale, void* dqkv, float* dbias_ws, float* dpad_ws, hipStream_t stream);
int esvit_big_relpos_bias_bwd(const float* dbias_ws, int parts, const int64_t* index, int N, int nH, int table_rows, float* dtable,
                              hipStream_t stream);

extern "C" int esvit_attn_frag_elems(int N) { return N <= NP ? FRAG_ELEMS : (N <= esvit_big_npb() ? esvit_big_frag_elems() : -1); }
extern "C" int esvit_window_attn_lse_elems(int N) { return N <= NP ? 0 : esvit_big_npb(); }
extern "C" int esvit_window_attn_bwd_parts(int N, int Bw, int nH) { return N <= NP ? bwd_parts(Bw, nH) : esvit_big_parts(Bw, nH); }
extern "C" int esvit_window_attn_bwd_pad_rows(int dtype, int N, int Bw, int nH) {
    return N <= NP ? bwd_parts(Bw, nH) : esvit_big_pad_rows(Bw, nH, dtype);
}

static int fill_bias_frag(const float* rel_table, int ws, int N, int nH, float* bias_frag_ws, hipStream_t stream) {
    // closed-form index (no index tensor needed): same values as relative_position_index
    hipLaunchKernelGGL(relpos_bias_frag_from_table_kernel, dim3(ceil_div((long)nH * FRAG_ELEMS, 256)), dim3(256), 0, stream, rel_table, ws, N,
                       nH, bias_frag_ws);
    ESVIT_CHECK_LAUNCH("relpos_bias(frag)");
    return ESVIT_OK;
}

extern "C" int esvit_window_attn_fwd(int dtype, const void* qkv, const float* qkv_bias, const int32_t* win2tok, int L,
                                     const float* rel_table, int ws, float* bias_frag_ws, const int32_t* region_ids, int nW, int nB, int N,
                                     int nH, int hd, float scale, void* out, float* lse, float* attn_out, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(qkv && qkv_bias && win2tok && rel_table && out && nB > 0 && nW > 0 && nH > 0 && L > 0 && ws > 0 && N == ws * ws,
                    "esvit_window_attn_fwd: bad args");
    ESVIT_CHECK_ARG(hd == HD || (hd == 64 && N <= NP), "esvit_window_attn_fwd: head_dim %d unsupported (32, or 64 with N <= 64)", hd);
    ESVIT_CHECK_ARG(dtype == ESVIT_BF16 || dtype == ESVIT_F32, "esvit_window_attn_fwd: bad dtype");
    if (N > NP)
        return esvit_big_attn_fwd(dtype, qkv, qkv_bias, win2tok, L, rel_table, ws, bias_frag_ws, region_ids, nW, nB, N, nH, scale, out, lse, attn_out,
                                  stream);
    ESVIT_CHECK_ARG(bias_frag_ws != nullptr, "esvit_window_attn_fwd: 7x7 windows need the bias_frag_ws scratch");
    {
        int rc = fill_bias_frag(rel_table, ws, N, nH, bias_frag_ws, stream);
        if (rc != ESVIT_OK) return rc;
    }
    const int Bw = nB * nW;
    ESVIT_CHECK_ARG((long)L * 3 * nH * hd * 4 < 0x7fff0000L || hd == HD, "esvit_window_attn_fwd: image too large for head_dim 64");
    if ((g_attn_fwd_impl >= 3 || hd != HD) && (long)L * 3 * nH * hd * 4 < 0x7fff0000L) {
        // persistent wave pairs: 256 CUs x 10 resident two-wave workgroups, one head each, at most one window per pair
        int parts = ((hd == HD ? 2560 : 1024) + nH - 1) / nH;
        if (parts > Bw) parts = Bw;
#define LAUNCH_FWD3(TT, HH)                                                                                                     \
    {                                                                                                                           \
        const size_t lds3 = (size_t)AttnCfgH<TT, HH>::FWD_PER_WAVE * sizeof(TT), lds4 = (size_t)3 * AttnCfgH<TT, HH>::QK_ELEMS * sizeof(TT); \
        const size_t lds = lds3 > lds4 ? lds3 : lds4;                                                                           \
        auto kern = g_attn_fwd_impl == 4 ? (attn_out ? attn_fwd4_kernel<TT, true, HH> : attn_fwd4_kernel<TT, false, HH>)       \
                                         : (attn_out ? attn_fwd3_kernel<TT, true, HH> : attn_fwd3_kernel<TT, false, HH>);      \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
        hipLaunchKernelGGL(kern, dim3(parts * nH), dim3(128), lds, stream, (const TT*)qkv, qkv_bias, win2tok, L,                \
                           (const float*)bias_frag_ws, region_ids, nW, Bw, N, nH, scale, parts, (TT*)out, attn_out);            \
    }
        if (dtype == ESVIT_BF16) {
            if (hd == HD) LAUNCH_FWD3(bf16, 32) else LAUNCH_FWD3(bf16, 64)
        } else {
            if (hd == HD) LAUNCH_FWD3(float, 32) else LAUNCH_FWD3(float, 64)
        }
#undef LAUNCH_FWD3
        ESVIT_CHECK_LAUNCH("window_attn_fwd(v3)");
        return ESVIT_OK;
    }
    if (g_attn_fwd_impl == 2 && (long)L * 3 * nH * HD * 4 < 0x7fff0000L) {
        // persistent waves: 256 CUs x 8 resident waves, one head each, at most one window per wave
        int parts = (2048 + nH - 1) / nH;
        if (parts > Bw) parts = Bw;
        const int grid2 = ceil_div((long)parts * nH, 4);
        if (dtype == ESVIT_BF16) {
            const size_t lds = 4 * (size_t)AttnCfg<bf16>::FWD_PER_WAVE * sizeof(bf16);
            auto kern = attn_out ? attn_fwd2_kernel<bf16, true> : attn_fwd2_kernel<bf16, false>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(grid2), dim3(256), lds, stream, (const bf16*)qkv, qkv_bias, win2tok, L,
                               (const float*)bias_frag_ws, region_ids, nW, Bw, N, nH, scale, parts, (bf16*)out, attn_out);
        } else {
            const size_t lds = 4 * (size_t)AttnCfg<float>::FWD_PER_WAVE * sizeof(float);
            auto kern = attn_out ? attn_fwd2_kernel<float, true> : attn_fwd2_kernel<float, false>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(grid2), dim3(256), lds, stream, (const float*)qkv, qkv_bias, win2tok, L,
                               (const float*)bias_frag_ws, region_ids, nW, Bw, N, nH, scale, parts, (float*)out, attn_out);
        }
        ESVIT_CHECK_LAUNCH("window_attn_fwd(v2)");
        return ESVIT_OK;
    }
    const int grid = ceil_div((long)Bw * nH, 4);
    if (dtype == ESVIT_BF16) {
        const size_t lds = 4 * (size_t)AttnCfg<bf16>::FWD_PER_WAVE * sizeof(bf16);
        auto kern = attn_fwd_kernel<bf16>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, (const bf16*)qkv, qkv_bias, win2tok, L,
                           (const float*)bias_frag_ws, region_ids, nW, Bw, N, nH, scale, (bf16*)out, attn_out);
    } else {
        const size_t lds = 4 * (size_t)AttnCfg<float>::FWD_PER_WAVE * sizeof(float);
        auto kern = attn_fwd_kernel<float>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, (const float*)qkv, qkv_bias, win2tok, L,
                           (const float*)bias_frag_ws, region_ids, nW, Bw, N, nH, scale, (float*)out, attn_out);
    }
    ESVIT_CHECK_LAUNCH("window_attn_fwd");
    return ESVIT_OK;
}

extern "C" int esvit_window_attn_bwd(int dtype, const void* qkv, const float* qkv_bias, const int32_t* win2tok, int L, const void* dout,
                                     const void* fwd_out, const float* lse, const float* rel_table, int ws, float* bias_frag_ws,
                                     const int32_t* region_ids, int nW, int nB, int N, int nH, int hd, float scale, void* dqkv,
                                     float* dbias_ws, float* dpad_ws, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(qkv && qkv_bias && win2tok && dout && rel_table && dqkv && dbias_ws && dpad_ws && nB > 0 && nW > 0 && nH > 0 && L > 0 &&
                        ws > 0 && N == ws * ws,
                    "esvit_window_attn_bwd: bad args");
    ESVIT_CHECK_ARG(hd == HD || (hd == 64 && N <= NP), "esvit_window_attn_bwd: head_dim %d unsupported (32, or 64 with N <= 64)", hd);
    ESVIT_CHECK_ARG(dtype == ESVIT_BF16 || dtype == ESVIT_F32, "esvit_window_attn_bwd: bad dtype");
    if (N > NP)
        return esvit_big_attn_bwd(dtype, g_attn_use_tr, qkv, qkv_bias, win2tok, L, dout, fwd_out, lse, rel_table, ws, bias_frag_ws, region_ids, nW, nB,
                                  N, nH, scale, dqkv, dbias_ws, dpad_ws, stream);
    ESVIT_CHECK_ARG(bias_frag_ws != nullptr, "esvit_window_attn_bwd: 7x7 windows need the bias_frag_ws scratch");
    {
        int rc = fill_bias_frag(rel_table, ws, N, nH, bias_frag_ws, stream);
        if (rc != ESVIT_OK) return rc;
    }
    const int Bw = nB * nW;
    const int parts = bwd_parts(Bw, nH);
    ESVIT_CHECK_ARG((long)L * 3 * nH * hd * 4 < 0x7fff0000L || hd == HD, "esvit_window_attn_bwd: image too large for head_dim 64");
    if ((g_attn_bwd_impl == 3 || hd != HD) && (long)L * 3 * nH * hd * 4 < 0x7fff0000L) {
#define LAUNCH_BWD3(TT, TR, HH)                                                                                                     \
    {                                                                                                                           \
        const size_t lds = (size_t)AttnCfgH<TT, HH>::BWD3_PER_PAIR * sizeof(TT);                                                 \
        auto kern = attn_bwd3_kernel<TT, TR, HH>;                                                                                   \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
        hipLaunchKernelGGL(kern, dim3(parts * nH), dim3(128), lds, stream, (const TT*)qkv, qkv_bias, win2tok, L,                \
                           (const TT*)dout, (const float*)bias_frag_ws, region_ids, nW, Bw, N, nH, scale, parts, (TT*)dqkv,     \
                           dbias_ws, dpad_ws);                                                                                  \
    }
        if (dtype == ESVIT_BF16) {
            if (hd == HD) {
                if (g_attn_use_tr) LAUNCH_BWD3(bf16, true, 32) else LAUNCH_BWD3(bf16, false, 32)
            } else {
                LAUNCH_BWD3(bf16, true, 64)
            }
        } else {
            if (hd == HD) LAUNCH_BWD3(float, false, 32) else LAUNCH_BWD3(float, false, 64)
        }
#undef LAUNCH_BWD3
        ESVIT_CHECK_LAUNCH("window_attn_bwd(v3)");
        return ESVIT_OK;
    }
    if (g_attn_bwd_impl == 2 && (long)L * 3 * nH * HD * 4 < 0x7fff0000L) {
#define LAUNCH_BWD2(TT, TR)                                                                                                     \
    {                                                                                                                           \
        constexpr int WV = Bwd2Cfg<TT>::WAVES;                                                                                  \
        const size_t lds = WV * (size_t)Bwd2Cfg<TT>::PER_WAVE * sizeof(TT);                                                     \
        auto kern = attn_bwd2_kernel<TT, TR, WV>;                                                                               \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
        hipLaunchKernelGGL(kern, dim3(ceil_div((long)parts * nH, WV)), dim3(WV * 64), lds, stream, (const TT*)qkv, qkv_bias,    \
                           win2tok, L, (const TT*)dout, (const float*)bias_frag_ws, region_ids, nW, Bw, N, nH, scale, parts,    \
                           (TT*)dqkv, dbias_ws, dpad_ws);                                                                       \
    }
        if (dtype == ESVIT_BF16) {
            if (g_attn_use_tr) LAUNCH_BWD2(bf16, true) else LAUNCH_BWD2(bf16, false)
        } else {
            LAUNCH_BWD2(float, false)
        }
#undef LAUNCH_BWD2
        ESVIT_CHECK_LAUNCH("window_attn_bwd(v2)");
        return ESVIT_OK;
    }
    const int grid = ceil_div((long)parts * nH, 4);
#define LAUNCH_BWD(TT, TR)                                                                                                      \
    {                                                                                                                           \
        const size_t lds = 4 * (size_t)AttnCfg<TT>::BWD_PER_WAVE * sizeof(TT);                                                  \
        auto kern = g_attn_minw == 2 ? attn_bwd_kernel<TT, TR, 2> : attn_bwd_kernel<TT, TR, 1>;                                \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, (const TT*)qkv, qkv_bias, win2tok, L, (const TT*)dout,      \
                           (const float*)bias_frag_ws, region_ids, nW, Bw, N, nH, scale, parts,                                 \
                           (TT*)dqkv, dbias_ws, dpad_ws);                                                                       \
    }
    if (dtype == ESVIT_BF16) {
        if (g_attn_use_tr) LAUNCH_BWD(bf16, true) else LAUNCH_BWD(bf16, false)
    } else {
        LAUNCH_BWD(float, false)
    }
#undef LAUNCH_BWD
    ESVIT_CHECK_LAUNCH("window_attn_bwd");
    return ESVIT_OK;
}
