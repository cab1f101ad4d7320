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

template <typename T>
struct AttnCfg {
    static constexpr int VEC = ElemTraits<T>::VEC;
    static constexpr int LDQ = HD + VEC;  // [NP][LDQ] images of q, k, v, dO
    static constexpr int LDP = NP + VEC;  // [NP][LDP] image of P / dS, [HD][LDP] image of V^T
    static constexpr int QK_ELEMS = NP * LDQ;
    static constexpr int P_ELEMS = NP * LDP;
    static constexpr int VT_ELEMS = HD * LDP;
    static constexpr int R1 = (2 * QK_ELEMS > P_ELEMS) ? 2 * QK_ELEMS : P_ELEMS;  // Q,K overlaid by P
    static constexpr int FWD_PER_WAVE = R1 + VT_ELEMS;
    static constexpr int BWD_PER_WAVE = 4 * QK_ELEMS + P_ELEMS;
};

template <typename T>
__device__ __forceinline__ void store_frag4(T* p, f32x4 v) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<f32x4*>(p) = v;
    } else {
        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(p) = o;
    }
}

// stage one [N][HD] matrix of the (window, head) slice into a [NP][LDQ] LDS image (rows >= N zero)
template <typename T>
__device__ __forceinline__ void stage_rows(const T* __restrict__ g, long row_stride, int N, bool active, float scale,
                                           T* lds, int lane) {
    constexpr int VEC = AttnCfg<T>::VEC, LDQ = AttnCfg<T>::LDQ, VPR = HD / VEC;
#pragma unroll
    for (int i = 0; i < NP * VPR / 64; ++i) {
        const int v = lane + 64 * i;
        const int t = v / VPR, dv = v % VPR;
        Vec16<T> x = zero16<T>();
        if (active && t < N) x = ld16<T>(g + (long)t * row_stride + dv * VEC);
        if (scale != 1.f) {
#pragma unroll
            for (int e = 0; e < Vec16<T>::N; ++e) x.set(e, x.get(e) * scale);
        }
        st16<T>(lds + t * LDQ + dv * VEC, x);
    }
}

// scores + softmax, shared by fwd and bwd: returns P^T fragments p[ki][qj] (fp32)
template <typename T>
__device__ __forceinline__ void scores_softmax(const T* Qs, const T* Ks, const float* __restrict__ bias_f,
                                               const float* __restrict__ mask_f, int lane, int c, int g,
                                               f32x4 (&p)[4][4]) {
    constexpr int LDQ = AttnCfg<T>::LDQ;
    Frag<T> kf[4], qf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        kf[i] = frag_kc<T>(Ks, LDQ, 16 * i, 0, c, g);
        qf[i] = frag_kc<T>(Qs, LDQ, 16 * i, 0, c, g);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 b = *reinterpret_cast<const f32x4*>(bias_f + ((i * 4 + j) * 64 + lane) * 4);
            if (mask_f) b += *reinterpret_cast<const f32x4*>(mask_f + ((i * 4 + j) * 64 + lane) * 4);
            p[i][j] = b;
            mma(kf[i], qf[j], p[i][j]);
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
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_frag,
                                                       const float* __restrict__ mask_frag, int nW, int Bw, int N, int nH,
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
    const T* src = qkv + (long)bw * N * 3 * C + h * HD;

    stage_rows<T>(src, 3L * C, N, active, scale, Qs, lane);
    stage_rows<T>(src + C, 3L * C, N, active, 1.f, Ks, lane);
    {  // V transposed: Vt[d][key]
        constexpr int VPR = HD / VEC;
#pragma unroll
        for (int i = 0; i < NP * VPR / 64; ++i) {
            const int v = lane + 64 * i;
            const int t = v / VPR, dv = v % VPR;
            Vec16<T> x = zero16<T>();
            if (active && t < N) x = ld16<T>(src + 2 * C + (long)t * 3 * C + dv * VEC);
#pragma unroll
            for (int e = 0; e < Vec16<T>::N; ++e) Vt[(dv * VEC + e) * LDP + t] = from_f32<T>(x.get(e));
        }
    }
    __syncthreads();

    f32x4 p[4][4];
    const float* bias_f = bias_frag + (long)h * FRAG_ELEMS;
    const float* mask_f = mask_frag ? mask_frag + (long)(bw % nW) * FRAG_ELEMS : nullptr;
    scores_softmax<T>(Qs, Ks, bias_f, mask_f, lane, c, g, p);

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
    __syncthreads();  // all fragment reads of Q,K are done before P overwrites them
    store_pt<T>(Ps, p, c, g);
    __syncthreads();

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
    if (active) {
        T* dst = out + (long)bw * N * C + h * HD;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 16 * i + 4 * g + r;
                if (q < N) {
                    dst[(long)q * C + c] = from_f32<T>(o[i][0][r]);
                    dst[(long)q * C + 16 + c] = from_f32<T>(o[i][1][r]);
                }
            }
    }
}

// -------------------------------------------------------------------------------------------------
// Backward.  Block = 2 waves; wave `wv` (global) owns head h = wv % nH and the windows
// bw = wv / nH + k * parts, k = 0,1,...; its bias-gradient partial goes to dbias_ws[wv / nH][h].
template <typename T, bool USE_TR>
__global__ __launch_bounds__(128) void attn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                       const float* __restrict__ bias_frag,
                                                       const float* __restrict__ mask_frag, int nW, int Bw, int N, int nH,
                                                       float scale, int parts, T* __restrict__ dqkv,
                                                       float* __restrict__ dbias_ws) {
    using Cfg = AttnCfg<T>;
    constexpr int LDQ = Cfg::LDQ, LDP = Cfg::LDP;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    T* base = reinterpret_cast<T*>(smem_raw) + wave * Cfg::BWD_PER_WAVE;
    T* Qs = base;
    T* Ks = base + Cfg::QK_ELEMS;
    T* Vs = base + 2 * Cfg::QK_ELEMS;
    T* Os = base + 3 * Cfg::QK_ELEMS;  // dO
    T* Ps = base + 4 * Cfg::QK_ELEMS;  // P, then dS  ([q][key])

    const long wv = (long)blockIdx.x * 2 + wave;
    const bool wave_ok = wv < (long)parts * nH;
    const int h = (int)(wv % nH);
    const int part = (int)(wv / nH);
    const int C = nH * HD;
    const float* bias_f = bias_frag + (long)h * FRAG_ELEMS;

    f32x4 db[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) db[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int iters = (Bw + parts - 1) / parts;
    for (int it = 0; it < iters; ++it) {
        const int bw = part + it * parts;
        const bool active = wave_ok && bw < Bw;
        const int bwc = active ? bw : 0;
        const T* src = qkv + (long)bwc * N * 3 * C + h * HD;
        __syncthreads();  // previous iteration's LDS reads are complete
        stage_rows<T>(src, 3L * C, N, active, scale, Qs, lane);
        stage_rows<T>(src + C, 3L * C, N, active, 1.f, Ks, lane);
        stage_rows<T>(src + 2 * C, 3L * C, N, active, 1.f, Vs, lane);
        stage_rows<T>(dout + (long)bwc * N * C + h * HD, (long)C, N, active, 1.f, Os, lane);
        __syncthreads();

        f32x4 p[4][4];
        const float* mask_f = mask_frag ? mask_frag + (long)(bwc % nW) * FRAG_ELEMS : nullptr;
        scores_softmax<T>(Qs, Ks, bias_f, mask_f, lane, c, g, p);
        store_pt<T>(Ps, p, c, g);

        // dP^T = V dO^T  (same fragment positions as p)
        f32x4 dp[4][4];
        {
            Frag<T> vf[4], of[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                vf[i] = frag_kc<T>(Vs, LDQ, 16 * i, 0, c, g);
                of[i] = frag_kc<T>(Os, LDQ, 16 * i, 0, c, g);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dp[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    mma(vf[i], of[j], dp[i][j]);
                }
        }
        // dS = P o (dP - delta_q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) d += p[i][j][r] * dp[i][j][r];
            d += __shfl_xor(d, 16, 64);
            d += __shfl_xor(d, 32, 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dp[i][j] = p[i][j] * (dp[i][j] - d);
                if (active) db[i][j] += dp[i][j];
            }
        }
        __syncthreads();  // Ps (= P) written by all lanes

        // dV[key][d] = sum_q P[q][key] dO[q][d]: A = P^T via k-strided read of Ps, B = dO k-strided
        T* dst = dqkv + (long)bwc * N * 3 * C + h * HD;
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
            if (active) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = 16 * i + 4 * g + r;
                        if (key < N) {
                            dst[(long)key * 3 * C + 2 * C + c] = from_f32<T>(acc[i][0][r]);
                            dst[(long)key * 3 * C + 2 * C + 16 + c] = from_f32<T>(acc[i][1][r]);
                        }
                    }
            }
        }
        __syncthreads();  // reads of P complete
        store_pt<T>(Ps, dp, c, g);  // dS as [q][key]
        __syncthreads();

        // dQ[q][d] = scale * sum_key dS[q][key] K[key][d]: A = dS k-contiguous, B = K k-strided
        // dK[key][d] = sum_q dS[q][key] Qs[q][d] (Qs already holds scale*q): A = dS^T k-strided, B = Qs k-strided
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
            if (active) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int t = 16 * i + 4 * g + r;
                        if (t < N) {
                            dst[(long)t * 3 * C + c] = from_f32<T>(aq[i][0][r] * scale);
                            dst[(long)t * 3 * C + 16 + c] = from_f32<T>(aq[i][1][r] * scale);
                            dst[(long)t * 3 * C + C + c] = from_f32<T>(ak[i][0][r]);
                            dst[(long)t * 3 * C + C + 16 + c] = from_f32<T>(ak[i][1][r]);
                        }
                    }
            }
        }
    }
    if (wave_ok) {
        float* ws = dbias_ws + ((long)part * nH + h) * FRAG_ELEMS;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(ws + ((i * 4 + j) * 64 + lane) * 4) = db[i][j];
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

inline int bwd_parts(int Bw, int nH) {
    // enough waves to fill the chip (256 CUs x ~4 resident bwd waves), at most one window per wave
    int parts = (2048 + nH - 1) / nH;
    if (parts > Bw) parts = Bw;
    if (parts < 1) parts = 1;
    return parts;
}

}  // namespace

#define STREAM(s_) hipStream_t stream = reinterpret_cast<hipStream_t>(s_)

extern "C" int esvit_attn_frag_elems(int N) { return N <= NP ? FRAG_ELEMS : -1; }

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

extern "C" int esvit_relpos_bias_bwd(const float* dbias_ws, int parts, const int64_t* index, int N, int nH, int table_rows,
                                     float* dtable, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(dbias_ws && index && dtable && parts > 0 && N > 0 && N <= NP && nH > 0 && table_rows > 0,
                    "esvit_relpos_bias_bwd: bad args");
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
extern "C" void esvit_debug_set_attn_tr_read(int on) { g_attn_use_tr = on; }

extern "C" int esvit_window_attn_fwd(int dtype, const void* qkv, const float* bias_frag, const float* mask_frag, int nW, int Bw,
                                     int N, int nH, int hd, float scale, void* out, float* attn_out, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(qkv && bias_frag && out && Bw > 0 && nH > 0, "esvit_window_attn_fwd: bad args");
    ESVIT_CHECK_ARG(hd == HD, "esvit_window_attn_fwd: head_dim %d unsupported (32 only)", hd);
    if (N > NP) {
        esvit_set_error("esvit_window_attn_fwd: N=%d > %d (14x14 windows) not built yet", N, NP);
        return ESVIT_ERR_UNSUPPORTED;
    }
    if (mask_frag) ESVIT_CHECK_ARG(nW > 0 && Bw % nW == 0, "esvit_window_attn_fwd: Bw must be a multiple of nW");
    const int grid = ceil_div((long)Bw * nH, 4);
    if (dtype == ESVIT_BF16) {
        const size_t lds = 4 * (size_t)AttnCfg<bf16>::FWD_PER_WAVE * sizeof(bf16);
        auto kern = attn_fwd_kernel<bf16>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, (const bf16*)qkv, bias_frag, mask_frag, nW, Bw, N, nH, scale,
                           (bf16*)out, attn_out);
    } else if (dtype == ESVIT_F32) {
        const size_t lds = 4 * (size_t)AttnCfg<float>::FWD_PER_WAVE * sizeof(float);
        auto kern = attn_fwd_kernel<float>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, (const float*)qkv, bias_frag, mask_frag, nW, Bw, N, nH, scale,
                           (float*)out, attn_out);
    } else {
        esvit_set_error("esvit_window_attn_fwd: bad dtype");
        return ESVIT_ERR_ARG;
    }
    ESVIT_CHECK_LAUNCH("window_attn_fwd");
    return ESVIT_OK;
}

extern "C" int esvit_window_attn_bwd_parts(int Bw, int nH) { return bwd_parts(Bw, nH); }

extern "C" int esvit_window_attn_bwd(int dtype, const void* qkv, const void* dout, const float* bias_frag, const float* mask_frag,
                                     int nW, int Bw, int N, int nH, int hd, float scale, void* dqkv, float* dbias_ws,
                                     esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(qkv && dout && bias_frag && dqkv && dbias_ws && Bw > 0 && nH > 0, "esvit_window_attn_bwd: bad args");
    ESVIT_CHECK_ARG(hd == HD, "esvit_window_attn_bwd: head_dim %d unsupported (32 only)", hd);
    if (N > NP) {
        esvit_set_error("esvit_window_attn_bwd: N=%d > %d (14x14 windows) not built yet", N, NP);
        return ESVIT_ERR_UNSUPPORTED;
    }
    if (mask_frag) ESVIT_CHECK_ARG(nW > 0 && Bw % nW == 0, "esvit_window_attn_bwd: Bw must be a multiple of nW");
    const int parts = bwd_parts(Bw, nH);
    const int grid = ceil_div((long)parts * nH, 2);
    if (dtype == ESVIT_BF16) {
        const size_t lds = 2 * (size_t)AttnCfg<bf16>::BWD_PER_WAVE * sizeof(bf16);
        if (g_attn_use_tr) {
            auto kern = attn_bwd_kernel<bf16, true>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(128), lds, stream, (const bf16*)qkv, (const bf16*)dout, bias_frag, mask_frag, nW,
                               Bw, N, nH, scale, parts, (bf16*)dqkv, dbias_ws);
        } else {
            auto kern = attn_bwd_kernel<bf16, false>;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(128), lds, stream, (const bf16*)qkv, (const bf16*)dout, bias_frag, mask_frag, nW,
                               Bw, N, nH, scale, parts, (bf16*)dqkv, dbias_ws);
        }
    } else if (dtype == ESVIT_F32) {
        const size_t lds = 2 * (size_t)AttnCfg<float>::BWD_PER_WAVE * sizeof(float);
        auto kern = attn_bwd_kernel<float, false>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(128), lds, stream, (const float*)qkv, (const float*)dout, bias_frag, mask_frag, nW,
                           Bw, N, nH, scale, parts, (float*)dqkv, dbias_ws);
    } else {
        esvit_set_error("esvit_window_attn_bwd: bad dtype");
        return ESVIT_ERR_ARG;
    }
    ESVIT_CHECK_LAUNCH("window_attn_bwd");
    return ESVIT_OK;
}
