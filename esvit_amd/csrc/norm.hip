// Row-normalisation kernels: LayerNorm fwd/bwd (with the token<->window row maps of the Swin
// block folded into the store / load), PatchMerging gather+LayerNorm, row L2-normalise and the
// legacy weight_norm of DINOHead.last_layer.  All are HBM-bound: one pass over the row held in
// registers, 16-byte accesses, wave-shuffle reductions (no LDS except for the dgamma/dbeta
// block reduction).
#include "common.h"
#include "../../include/esvit_hip.h"

namespace {

template <typename T>
__device__ __forceinline__ f32x4 load4(const T* p);
template <>
__device__ __forceinline__ f32x4 load4<float>(const float* p) {
    return *reinterpret_cast<const f32x4*>(p);
}
template <>
__device__ __forceinline__ f32x4 load4<bf16>(const bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T>
__device__ __forceinline__ void store4(T* p, f32x4 v);
template <>
__device__ __forceinline__ void store4<float>(float* p, f32x4 v) {
    *reinterpret_cast<f32x4*>(p) = v;
}
template <>
__device__ __forceinline__ void store4<bf16>(bf16* p, f32x4 v) {
    bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = o;
}

__device__ __forceinline__ float hsum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

constexpr int LN_THREADS = 256;

// Source addressing for a LayerNorm row: plain rows, or the 2x2 PatchMerging gather
// (swin_transformer.py:410-414) where the 4C vector is [x(2i,2j), x(2i+1,2j), x(2i,2j+1), x(2i+1,2j+1)].
struct RowSrc {
    const float* x;
    int C;         // channels of the normalised row (4*Cin for merge)
    int merge;     // 0/1
    int H, W, Cin; // merge geometry (input grid)
    __device__ __forceinline__ const float* ptr(long r, int c4) const {
        if (!merge) return x + r * (long)C + c4 * 4;
        const int Ho = H / 2, Wo = W / 2;
        const long b = r / (Ho * Wo);
        const int rem = (int)(r % (Ho * Wo));
        const int i = rem / Wo, j = rem % Wo;
        const int c = c4 * 4;
        const int blk = c / Cin, cc = c % Cin;  // blk: 0 (0,0) 1 (1,0) 2 (0,1) 3 (1,1)
        const int di = blk & 1, dj = blk >> 1;
        return x + ((b * H + (2 * i + di)) * W + (2 * j + dj)) * (long)Cin + cc;
    }
    __device__ __forceinline__ float* dptr(float* dx, long r, int c4) const {
        return dx + (ptr(r, c4) - x);
    }
    // element offset of (row r, vector c4) in the un-gathered [.., C or Cin] layout, and the token row it belongs to
    __device__ __forceinline__ long off(long r, int c4) const { return ptr(r, c4) - x; }
    __device__ __forceinline__ long dst_row(long r, int c4) const {  // (the same sub-expressions as ptr(): folded by the compiler)
        if (!merge) return r;
        const int Ho = H / 2, Wo = W / 2;
        const long b = r / (Ho * Wo);
        const int rem = (int)(r % (Ho * Wo));
        const int i = rem / Wo, j = rem % Wo;
        const int blk = (c4 * 4) / Cin;
        return (b * H + (2 * i + (blk & 1))) * W + (2 * j + (blk >> 1));
    }
};

template <typename T, int G, int ITERS>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(RowSrc src, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, long rows,
                                                             T* __restrict__ y, float* __restrict__ y_f32,
                                                             float* __restrict__ mean, float* __restrict__ rstd,
                                                             const int* __restrict__ rowmap, int tokens, int period_out) {
    constexpr int RPB = LN_THREADS / G;
    const int C = src.C, C4 = C / 4;
    const int gl = threadIdx.x % G;
    const long r = (long)blockIdx.x * RPB + threadIdx.x / G;
    if (r >= rows) return;  // whole lane group exits together (G divides 64)
    f32x4 v[ITERS];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c4 = gl + it * G;
        v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c4 < C4) v[it] = *reinterpret_cast<const f32x4*>(src.ptr(r, c4));
        s += hsum4(v[it]);
    }
    s = group_sum<G>(s);
    const float mu = s / C;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c4 = gl + it * G;
        if (c4 < C4) {
            const f32x4 d = v[it] - mu;
            q += hsum4(d * d);
        }
    }
    q = group_sum<G>(q);
    const float rs = rsqrtf(q / C + eps);
    if (gl == 0) {
        mean[r] = mu;
        rstd[r] = rs;
    }
    long ro = r;
    if (rowmap) ro = (r / tokens) * (long)period_out + rowmap[r % tokens];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c4 = gl + it * G;
        if (c4 < C4) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c4 * 4);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c4 * 4);
            const f32x4 o = (v[it] - mu) * rs * gm + bt;
            store4<T>(y + ro * C + c4 * 4, o);
            if (y_f32) *reinterpret_cast<f32x4*>(y_f32 + r * C + c4 * 4) = o;
        }
    }
}

// backward: dx = rstd * (gdy - mean(gdy) - xhat * mean(gdy*xhat)), gdy = gamma*dy
template <typename T, int G, int ITERS, typename TA = T>  // TA: type of dx_act (bf16 from an fp32 upstream gradient: the patch-embedding norm)
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(RowSrc src, const T* __restrict__ dy,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ g_in, long rows,
                                                             float* __restrict__ dx, float* __restrict__ ws,
                                                             const int* __restrict__ rowmap, int tokens, int period_in,
                                                             TA* __restrict__ dx_act, const float* __restrict__ rowscale,
                                                             int rows_per_sample) {
    constexpr int RPB = LN_THREADS / G;
    const int C = src.C, C4 = C / 4;
    const int gl = threadIdx.x % G, grp = threadIdx.x / G;
    f32x4 dgam[ITERS], dbet[ITERS], gm[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        dgam[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        dbet[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c4 = gl + it * G;
        gm[it] = (c4 < C4) ? *reinterpret_cast<const f32x4*>(gamma + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (long r = (long)blockIdx.x * RPB + grp; r < rows; r += (long)gridDim.x * RPB) {
        const float mu = mean[r], rs = rstd[r];
        long ri = r;
        if (rowmap) ri = (r / tokens) * (long)period_in + rowmap[r % tokens];
        f32x4 xh[ITERS], gd[ITERS];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c4 = gl + it * G;
            xh[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            gd[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c4 < C4) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(src.ptr(r, c4));
                const f32x4 d = load4<T>(dy + ri * C + c4 * 4);
                xh[it] = (xv - mu) * rs;
                gd[it] = d * gm[it];
                dgam[it] += d * xh[it];
                dbet[it] += d;
                s1 += hsum4(gd[it]);
                s2 += hsum4(gd[it] * xh[it]);
            }
        }
        s1 = group_sum<G>(s1) / C;
        s2 = group_sum<G>(s2) / C;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c4 = gl + it * G;
            if (c4 < C4) {
                f32x4 o = (gd[it] - s1 - xh[it] * s2) * rs;
                const long off = src.off(r, c4);  // (the 2x2 merge scatters a row over four token rows)
                if (g_in) o += *reinterpret_cast<const f32x4*>(g_in + off);
                if (dx) *reinterpret_cast<f32x4*>(dx + off) = o;  // (NULL: only the activation-dtype copy is wanted)
                if (dx_act) {  // the activation-dtype, DropPath-scaled copy the next GEMMs of the backward read (saves a cast pass)
                    const float sc = rowscale ? rowscale[src.dst_row(r, c4) / rows_per_sample] : 1.f;
                    store4<TA>(dx_act + off, o * sc);
                }
            }
        }
    }
    // block reduction of dgamma/dbeta over the RPB row groups -> ws[blk][2][C]
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* sm = reinterpret_cast<f32x4*>(smem_raw);  // [RPB][2][C4]
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c4 = gl + it * G;
        if (c4 < C4) {
            sm[(grp * 2 + 0) * C4 + c4] = dgam[it];
            sm[(grp * 2 + 1) * C4 + c4] = dbet[it];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C4; i += LN_THREADS) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int g2 = 0; g2 < RPB; ++g2) s += sm[g2 * 2 * C4 + i];
        *reinterpret_cast<f32x4*>(ws + ((long)blockIdx.x * 2 * C4 + i) * 4) = s;
    }
}

struct LnCfg {
    int G, ITERS;
};
inline bool ln_cfg(int C, LnCfg* cfg) {
    if (C % 4 != 0 || C <= 0) return false;
    const int C4 = C / 4;
    // Swin widths are 96 * 2^s (and 4x that in PatchMerging): C4 = 3 * 2^k.  A lane group of C4/3 lanes with three
    // 16-byte vectors per lane keeps every lane busy (a power-of-two group would idle a quarter of them) and puts
    // three loads per lane in flight -- these kernels are latency-bound, not bandwidth-bound, at one vector per lane.
    if (C4 % 3 == 0) {
        const int G = C4 / 3;
        if (G == 8 || G == 16 || G == 32 || G == 64) {
            cfg->G = G;
            cfg->ITERS = 3;
            return true;
        }
    }
    if (C4 == 64) {  // bottleneck width 256
        cfg->G = 16;
        cfg->ITERS = 4;
        return true;
    }
    const int G = C4 <= 16 ? 16 : (C4 <= 32 ? 32 : 64);
    const int it = (C4 + G - 1) / G;
    const int allowed[] = {1, 2, 3, 4, 6, 8};
    for (int a : allowed)
        if (it <= a) {
            cfg->G = G;
            cfg->ITERS = a;
            return true;
        }
    return false;
}

template <int G_, int IT_>
struct LnShape {
    static constexpr int G = G_;
    static constexpr int ITERS = IT_;
};
template <typename F>
inline void ln_dispatch(const LnCfg& cfg, F&& f) {
    if (cfg.G == 8) f(LnShape<8, 3>{});
    else if (cfg.G == 16 && cfg.ITERS == 3) f(LnShape<16, 3>{});
    else if (cfg.G == 16 && cfg.ITERS == 4) f(LnShape<16, 4>{});
    else if (cfg.G == 16) f(LnShape<16, 1>{});
    else if (cfg.G == 32 && cfg.ITERS == 3) f(LnShape<32, 3>{});
    else if (cfg.G == 32) f(LnShape<32, 1>{});
    else if (cfg.ITERS == 1) f(LnShape<64, 1>{});
    else if (cfg.ITERS == 2) f(LnShape<64, 2>{});
    else if (cfg.ITERS == 3) f(LnShape<64, 3>{});
    else if (cfg.ITERS == 4) f(LnShape<64, 4>{});
    else if (cfg.ITERS == 6) f(LnShape<64, 6>{});
    else f(LnShape<64, 8>{});
}

inline int ln_bwd_nblk(long rows, int C) {
    LnCfg cfg;
    if (!ln_cfg(C, &cfg)) return 0;
    const int rpb = LN_THREADS / cfg.G;
    long nb = (rows + rpb - 1) / rpb;
    if (nb > 1024) nb = 1024;  // 8 resident 256-thread blocks per CU: one row per lane group in flight each
    if (nb < 1) nb = 1;
    return (int)nb;
}

template <typename T>
int ln_fwd_launch(RowSrc src, const float* gamma, const float* beta, float eps, long rows, void* y, float* y_f32,
                  float* mean, float* rstd, const int* rowmap, int tokens, int period_out, hipStream_t stream) {
    LnCfg cfg;
    ESVIT_CHECK_ARG(ln_cfg(src.C, &cfg), "layernorm: unsupported channel count %d", src.C);
    const int rpb = LN_THREADS / cfg.G;
    const int grid = ceil_div(rows, rpb);
    ln_dispatch(cfg, [&](auto shp) {
        using S = decltype(shp);
        hipLaunchKernelGGL((ln_fwd_kernel<T, S::G, S::ITERS>), dim3(grid), dim3(LN_THREADS), 0, stream, src, gamma, beta, eps,
                           rows, reinterpret_cast<T*>(y), y_f32, mean, rstd, rowmap, tokens, period_out);
    });
    ESVIT_CHECK_LAUNCH("layernorm_fwd");
    return ESVIT_OK;
}

template <typename T, typename TA = T>
int ln_bwd_launch(RowSrc src, const void* dy, const float* mean, const float* rstd, const float* gamma,
                  const float* g_in, long rows, float* dx, float* dgamma, float* dbeta, float* ws, const int* rowmap,
                  int tokens, int period_in, hipStream_t stream, void* dx_act = nullptr, const float* rowscale = nullptr,
                  int rows_per_sample = 1, int accumulate = 0) {
    LnCfg cfg;
    ESVIT_CHECK_ARG(ln_cfg(src.C, &cfg), "layernorm: unsupported channel count %d", src.C);
    const int nblk = ln_bwd_nblk(rows, src.C);
    const int rpb = LN_THREADS / cfg.G;
    const size_t lds = (size_t)rpb * 2 * (src.C / 4) * sizeof(f32x4);
    ESVIT_CHECK_ARG(lds <= 160 * 1024, "layernorm_bwd: C=%d too large", src.C);
    ln_dispatch(cfg, [&](auto shp) {
        using S = decltype(shp);
        auto kern = ln_bwd_kernel<T, S::G, S::ITERS, TA>;
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds);
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(LN_THREADS), lds, stream, src, reinterpret_cast<const T*>(dy), mean,
                           rstd, gamma, g_in, rows, dx, ws, rowmap, tokens, period_in, reinterpret_cast<TA*>(dx_act), rowscale,
                           rows_per_sample);
    });
    ESVIT_CHECK_LAUNCH("layernorm_bwd");
    // ws rows are [dgamma(C) | dbeta(C)]; reduce both halves (dgamma and dbeta may be separate allocations)
    const int C = src.C;
#ifdef ESVIT_PROBE_SKIP_FINISH  // timing probe (WRONG gradients): what the finishing launches cost on the main stream (profiles/r06_finish_offchain_ab.txt)
    return ESVIT_OK;
#endif
    if (dbeta == dgamma + C) return esvit_partial_reduce(ws, nblk, 2 * C, 2L * C, dgamma, accumulate, stream);
    int rc = esvit_partial_reduce(ws, nblk, C, 2L * C, dgamma, accumulate, stream);
    if (rc != ESVIT_OK) return rc;
    return esvit_partial_reduce(ws + C, nblk, C, 2L * C, dbeta, accumulate, stream);
}

// ---------------------------------------------------------------------------------------------
// row L2 normalise (vision_transformer.py:416: F.normalize(x, dim=-1, p=2), eps 1e-12)
// ---------------------------------------------------------------------------------------------
template <typename T, int ITERS>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ x, long R, int D, T* __restrict__ z,
                                                         float* __restrict__ inv_norm) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int D4 = D / 4;
    f32x4 v[ITERS];
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c4 = lane + it * 64;
        v[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c4 < D4) v[it] = load4<T>(x + r * D + c4 * 4);
        q += hsum4(v[it] * v[it]);
    }
    q = wave_sum(q);
    const float inv = 1.f / fmaxf(sqrtf(q), 1e-12f);
    if (lane == 0) inv_norm[r] = inv;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c4 = lane + it * 64;
        if (c4 < D4) store4<T>(z + r * D + c4 * 4, v[it] * inv);
    }
}

template <typename T, int ITERS>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const T* __restrict__ dz, const T* __restrict__ z,
                                                         const float* __restrict__ inv_norm, long R, int D,
                                                         T* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int D4 = D / 4;
    f32x4 g[ITERS], zz[ITERS];
    float dot = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c4 = lane + it * 64;
        g[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        zz[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c4 < D4) {
            g[it] = load4<T>(dz + r * D + c4 * 4);
            zz[it] = load4<T>(z + r * D + c4 * 4);
        }
        dot += hsum4(g[it] * zz[it]);
    }
    dot = wave_sum(dot);
    const float inv = inv_norm[r];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c4 = lane + it * 64;
        if (c4 < D4) store4<T>(dx + r * D + c4 * 4, (g[it] - zz[it] * dot) * inv);
    }
}

// ---------------------------------------------------------------------------------------------
// weight_norm (dim=0) of DINOHead.last_layer (vision_transformer.py:403): one wave per output row
// ---------------------------------------------------------------------------------------------
template <typename T, int ITERS>
__global__ __launch_bounds__(256) void weightnorm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                             int K, int D, T* __restrict__ w, T* __restrict__ wT,
                                                             float* __restrict__ inv_norm) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= K) return;
    const int D4 = D / 4;
    f32x4 x[ITERS];
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c4 = lane + it * 64;
        x[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c4 < D4) x[it] = *reinterpret_cast<const f32x4*>(v + (long)k * D + c4 * 4);
        q += hsum4(x[it] * x[it]);
    }
    q = wave_sum(q);
    const float inv = 1.f / sqrtf(q);
    if (lane == 0) inv_norm[k] = inv;
    const float sc = g[k] * inv;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c4 = lane + it * 64;
        if (c4 < D4) {
            const f32x4 o = x[it] * sc;
            if (w) store4<T>(w + (long)k * D + c4 * 4, o);
            if (wT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) wT[(long)(c4 * 4 + e) * K + k] = from_f32<T>(o[e]);
            }
        }
    }
}

template <int ITERS>
__global__ __launch_bounds__(256) void weightnorm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v,
                                                             const float* __restrict__ g,
                                                             const float* __restrict__ inv_norm, int K, int D,
                                                             float* __restrict__ dv, float* __restrict__ dg) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= K) return;
    const int D4 = D / 4;
    const float inv = inv_norm[k];
    f32x4 gw[ITERS], vh[ITERS];
    float dot = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c4 = lane + it * 64;
        gw[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        vh[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c4 < D4) {
            gw[it] = *reinterpret_cast<const f32x4*>(dw + (long)k * D + c4 * 4);
            vh[it] = *reinterpret_cast<const f32x4*>(v + (long)k * D + c4 * 4) * inv;
        }
        dot += hsum4(gw[it] * vh[it]);
    }
    dot = wave_sum(dot);
    if (dg && lane == 0) dg[k] = dot;
    const float sc = g[k] * inv;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c4 = lane + it * 64;
        if (c4 < D4) *reinterpret_cast<f32x4*>(dv + (long)k * D + c4 * 4) = (gw[it] - vh[it] * dot) * sc;
    }
}

template <int N_>
struct IntC {
    static constexpr int value = N_;
};
template <typename F>
inline void row_iters_dispatch(int d4, F&& f) {
    if (d4 <= 64) f(IntC<1>{});
    else if (d4 <= 128) f(IntC<2>{});
    else if (d4 <= 256) f(IntC<4>{});
    else f(IntC<8>{});
}

}  // namespace

extern "C" int esvit_layernorm_fwd(int dtype, const float* x, const float* gamma, const float* beta, float eps,
                                   int64_t rows, int C, void* y, float* y_f32, float* mean, float* rstd,
                                   const int32_t* rowmap, int tokens, int period_out, esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(x && gamma && beta && y && mean && rstd && rows > 0, "esvit_layernorm_fwd: bad args");
    if (rowmap) ESVIT_CHECK_ARG(tokens > 0 && period_out > 0, "esvit_layernorm_fwd: bad rowmap geometry");
    RowSrc src{x, C, 0, 0, 0, 0};
    if (dtype == ESVIT_BF16)
        return ln_fwd_launch<bf16>(src, gamma, beta, eps, rows, y, y_f32, mean, rstd, rowmap, tokens, period_out, stream);
    if (dtype == ESVIT_F32)
        return ln_fwd_launch<float>(src, gamma, beta, eps, rows, y, y_f32, mean, rstd, rowmap, tokens, period_out, stream);
    esvit_set_error("esvit_layernorm_fwd: bad dtype");
    return ESVIT_ERR_ARG;
}

int esvit_i_ln_bwd_blocks(long rows, int C) { return ln_bwd_nblk(rows, C); }  // esvit_query

// dx_act (optional) = cast(rowscale[row / rows_per_sample] * dx): the DropPath-scaled, activation-dtype gradient the following
// dgrad / wgrad GEMMs read (otherwise a separate esvit_gather_cast pass over dx)
extern "C" int esvit_layernorm_bwd(int dtype, const void* dy, const float* x, const float* mean, const float* rstd,
                                   const float* gamma, const float* g_in, int64_t rows, int C, float* dx,
                                   float* dgamma, float* dbeta, float* ws, const int32_t* rowmap, int tokens,
                                   int period_in, void* dx_act, int act_dtype, const float* rowscale, int rows_per_sample,
                                   esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(dy && x && mean && rstd && gamma && (dx || dx_act) && dgamma && dbeta && ws && rows > 0,
                    "esvit_layernorm_bwd: bad args");
    if (rowmap) ESVIT_CHECK_ARG(tokens > 0 && period_in > 0, "esvit_layernorm_bwd: bad rowmap geometry");
    if (dx_act) ESVIT_CHECK_ARG(!rowmap, "esvit_layernorm_bwd: dx_act is written at un-mapped rows only");
    if (dx_act) ESVIT_CHECK_ARG(act_dtype == dtype || (dtype == ESVIT_F32 && act_dtype == ESVIT_BF16), "esvit_layernorm_bwd: dx_act has dy's dtype, or bf16 from fp32");
    if (!dx) ESVIT_CHECK_ARG(!g_in, "esvit_layernorm_bwd: g_in is added to dx");
    if (rowscale) ESVIT_CHECK_ARG(dx_act && rows_per_sample > 0, "esvit_layernorm_bwd: rowscale scales dx_act and needs rows_per_sample");
    const int rps = rows_per_sample > 0 ? rows_per_sample : 1;
    RowSrc src{x, C, 0, 0, 0, 0};
    if (dtype == ESVIT_BF16)
        return ln_bwd_launch<bf16>(src, dy, mean, rstd, gamma, g_in, rows, dx, dgamma, dbeta, ws, rowmap, tokens, period_in, stream, dx_act,
                                   rowscale, rps);
    if (dtype == ESVIT_F32 && dx_act && act_dtype == ESVIT_BF16)
        return ln_bwd_launch<float, bf16>(src, dy, mean, rstd, gamma, g_in, rows, dx, dgamma, dbeta, ws, rowmap, tokens, period_in, stream,
                                          dx_act, rowscale, rps);
    if (dtype == ESVIT_F32)
        return ln_bwd_launch<float>(src, dy, mean, rstd, gamma, g_in, rows, dx, dgamma, dbeta, ws, rowmap, tokens, period_in, stream, dx_act,
                                    rowscale, rps);
    esvit_set_error("esvit_layernorm_bwd: bad dtype");
    return ESVIT_ERR_ARG;
}

extern "C" int esvit_merge_ln_fwd(int dtype, const float* x, const float* gamma, const float* beta, float eps, int nB,
                                  int H, int W, int C, void* y, float* mean, float* rstd, esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(x && gamma && beta && y && mean && rstd, "esvit_merge_ln_fwd: null pointer");
    ESVIT_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && C % 4 == 0 && nB > 0, "esvit_merge_ln_fwd: H,W must be even (H=%d W=%d)", H, W);
    RowSrc src{x, 4 * C, 1, H, W, C};
    const long rows = (long)nB * (H / 2) * (W / 2);
    if (dtype == ESVIT_BF16) return ln_fwd_launch<bf16>(src, gamma, beta, eps, rows, y, nullptr, mean, rstd, nullptr, 0, 0, stream);
    if (dtype == ESVIT_F32) return ln_fwd_launch<float>(src, gamma, beta, eps, rows, y, nullptr, mean, rstd, nullptr, 0, 0, stream);
    esvit_set_error("esvit_merge_ln_fwd: bad dtype");
    return ESVIT_ERR_ARG;
}

extern "C" int esvit_merge_ln_bwd(int dtype, const void* dy, const float* x, const float* mean, const float* rstd,
                                  const float* gamma, int nB, int H, int W, int C, float* dx, float* dgamma,
                                  float* dbeta, float* ws, void* dx_act, const float* rowscale, int rows_per_sample, int accumulate,
                                  esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && ws, "esvit_merge_ln_bwd: null pointer");
    ESVIT_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && C % 4 == 0 && nB > 0, "esvit_merge_ln_bwd: bad geometry");
    if (rowscale) ESVIT_CHECK_ARG(dx_act && rows_per_sample > 0, "esvit_merge_ln_bwd: rowscale scales dx_act and needs rows_per_sample");
    const int rps = rows_per_sample > 0 ? rows_per_sample : 1;
    RowSrc src{x, 4 * C, 1, H, W, C};
    const long rows = (long)nB * (H / 2) * (W / 2);
    if (dtype == ESVIT_BF16)
        return ln_bwd_launch<bf16>(src, dy, mean, rstd, gamma, nullptr, rows, dx, dgamma, dbeta, ws, nullptr, 0, 0, stream, dx_act, rowscale, rps,
                                   accumulate);
    if (dtype == ESVIT_F32)
        return ln_bwd_launch<float>(src, dy, mean, rstd, gamma, nullptr, rows, dx, dgamma, dbeta, ws, nullptr, 0, 0, stream, dx_act, rowscale, rps,
                                    accumulate);
    esvit_set_error("esvit_merge_ln_bwd: bad dtype");
    return ESVIT_ERR_ARG;
}

extern "C" int esvit_l2norm_fwd(int dtype, const void* x, int64_t R, int D, void* z, float* inv_norm, esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(x && z && inv_norm && R > 0 && D > 0 && D % 4 == 0 && D <= 2048, "esvit_l2norm_fwd: bad args (D=%d)", D);
    const int grid = ceil_div(R, 4);
    if (dtype == ESVIT_BF16) {
        row_iters_dispatch(D / 4, [&](auto it_) { hipLaunchKernelGGL((l2norm_fwd_kernel<bf16, decltype(it_)::value>), dim3(grid), dim3(256), 0, stream,
                                                       (const bf16*)x, (long)R, D, (bf16*)z, inv_norm); });
    } else {
        row_iters_dispatch(D / 4, [&](auto it_) { hipLaunchKernelGGL((l2norm_fwd_kernel<float, decltype(it_)::value>), dim3(grid), dim3(256), 0, stream,
                                                       (const float*)x, (long)R, D, (float*)z, inv_norm); });
    }
    ESVIT_CHECK_LAUNCH("l2norm_fwd");
    return ESVIT_OK;
}

extern "C" int esvit_l2norm_bwd(int dtype, const void* dz, const void* z, const float* inv_norm, int64_t R, int D,
                                void* dx, esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(dz && z && inv_norm && dx && R > 0 && D > 0 && D % 4 == 0 && D <= 2048, "esvit_l2norm_bwd: bad args");
    const int grid = ceil_div(R, 4);
    if (dtype == ESVIT_BF16) {
        row_iters_dispatch(D / 4, [&](auto it_) { hipLaunchKernelGGL((l2norm_bwd_kernel<bf16, decltype(it_)::value>), dim3(grid), dim3(256), 0, stream,
                                                       (const bf16*)dz, (const bf16*)z, inv_norm, (long)R, D, (bf16*)dx); });
    } else {
        row_iters_dispatch(D / 4, [&](auto it_) { hipLaunchKernelGGL((l2norm_bwd_kernel<float, decltype(it_)::value>), dim3(grid), dim3(256), 0, stream,
                                                       (const float*)dz, (const float*)z, inv_norm, (long)R, D, (float*)dx); });
    }
    ESVIT_CHECK_LAUNCH("l2norm_bwd");
    return ESVIT_OK;
}

extern "C" int esvit_weightnorm_fwd(int dtype, const float* v, const float* g, int K, int D, void* w, void* wT,
                                    float* inv_norm, esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(v && g && inv_norm && (w || wT) && K > 0 && D > 0 && D % 4 == 0 && D <= 2048, "esvit_weightnorm_fwd: bad args");
    const int grid = ceil_div(K, 4);
    if (dtype == ESVIT_BF16) {
        row_iters_dispatch(D / 4, [&](auto it_) { hipLaunchKernelGGL((weightnorm_fwd_kernel<bf16, decltype(it_)::value>), dim3(grid), dim3(256), 0, stream,
                                                       v, g, K, D, (bf16*)w, (bf16*)wT, inv_norm); });
    } else {
        row_iters_dispatch(D / 4, [&](auto it_) { hipLaunchKernelGGL((weightnorm_fwd_kernel<float, decltype(it_)::value>), dim3(grid), dim3(256), 0, stream,
                                                       v, g, K, D, (float*)w, (float*)wT, inv_norm); });
    }
    ESVIT_CHECK_LAUNCH("weightnorm_fwd");
    return ESVIT_OK;
}

extern "C" int esvit_weightnorm_bwd(const float* dw, const float* v, const float* g, const float* inv_norm, int K,
                                    int D, float* dv, float* dg, esvit_stream_t s_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(s_);
    ESVIT_CHECK_ARG(dw && v && g && inv_norm && dv && K > 0 && D > 0 && D % 4 == 0 && D <= 2048, "esvit_weightnorm_bwd: bad args");
    const int grid = ceil_div(K, 4);
    row_iters_dispatch(D / 4, [&](auto it_) { hipLaunchKernelGGL((weightnorm_bwd_kernel<decltype(it_)::value>), dim3(grid), dim3(256), 0, stream, dw, v, g,
                                                   inv_norm, K, D, dv, dg); });
    ESVIT_CHECK_LAUNCH("weightnorm_bwd");
    return ESVIT_OK;
}
