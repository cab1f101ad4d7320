// MFMA GEMM family for the EsViT hot path (gfx950).
//
//   C[M,N] = alpha * op(A)[M,K] * op(B)[K,N]  (+ fused epilogue)
//
// One kernel template covers the three shapes autograd needs:
//   forward  Y  = X  W^T   : A k-contiguous (M x K),  B k-contiguous (N x K)
//   dgrad    dX = dY W     : A k-contiguous,          B k-strided    (K x N)   [or cached W^T]
//   wgrad    dW = dY^T X   : A k-strided (K x M),     B k-strided    (K x N), split-K over rows
//
// Tiling: 256 threads = 4 waves (2 x 2), block tile BM x BN x 32, wave tile (BM/2) x (BN/2)
// built from 16x16 MFMA fragments (v_mfma_f32_16x16x32_bf16, or 8 x v_mfma_f32_16x16x4_f32 for
// the exact-fp32 parity mode).  Operands are register-staged global -> LDS (double-buffered,
// one barrier per k-tile).  K-contiguous tiles are read as one ds_read_b128 per fragment;
// K-strided tiles are stored as they lie in HBM and read with the gfx950 transpose read
// (ds_read_b64_tr_b16), so no operand is ever transposed through HBM.
//
// The lane->k assignment inside a fragment is "lane group g holds k = 8g..8g+7" for both A and
// B; a dot product is invariant to a permutation applied to both operands, so the fp32 path
// simply feeds element j of that 8-vector to the j-th 16x16x4 MFMA.
//
// Epilogue (all optional, fused on the fp32 accumulators): bias, GELU (+ pre-activation side
// output), GELU', per-sample DropPath scale, window->token row scatter (window_reverse + roll +
// crop of swin_transformer.py:315-325), residual add, bf16/fp32 store, split-K partial store.
#pragma once
#include "common.h"
#include "mfma.h"
#include "../../include/esvit_hip.h"

// the 256 x 256 eight-phase bf16 main loop (gemm_p8.hip); esvit_gemm's dispatcher (gemm.hip) checks what it requires
int esvit_gemm_p8_launch(const esvit_gemm_desc& d, hipStream_t stream);

namespace {

constexpr int BK = 32;          // k-tile of the register-staged loop
constexpr int NTHREADS = 256;   // workgroup of the register-staged loop (2 x 2 waves)

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>) -- every index is a constant by
// construction, so per-iteration register arrays never end up dynamically indexed (i.e. in scratch)
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ Frag<bf16> ones_frag(bf16) {
    Frag<bf16> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = (bf16)1.0f;
    return f;
}
__device__ __forceinline__ Frag<float> ones_frag(float) {
    Frag<float> f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f.v[j] = 1.0f;
    return f;
}

// row sums of op(A) (accumulated with an all-ones B fragment): lane c == 0 of each 16-lane group owns rows 4g+r
template <int FM>
__device__ __forceinline__ void store_colsum(const esvit_gemm_desc& p, const f32x4 (&accb)[FM], int m0, int wm_rows0, int z, int c, int g) {
    if (c != 0) return;
    float* dst = p.splitk > 1 ? p.colsum_partial + (long)z * p.M : p.colsum;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm_rows0 + i * 16 + 4 * g + r;
            if (m < p.M) dst[m] = accb[i][r] * p.alpha;
        }
}

// One operand tile: ROWS (BM or BN) x BK, in LDS either as [ROWS][BK+pad] (k contiguous) or as
// [BK][ROWS+pad] (k strided, i.e. exactly the HBM orientation).
template <typename T, bool KS, int ROWS, bool USE_TR>
struct Tile {
    static constexpr int VEC = ElemTraits<T>::VEC;
    static constexpr int LD = KS ? (ROWS + VEC) : (BK + VEC);
    static constexpr int ELEMS = KS ? (BK * LD) : (ROWS * LD);
    static constexpr int NVEC = ROWS * BK / VEC;
    static constexpr int VPT = (NVEC + NTHREADS - 1) / NTHREADS;

    Vec16<T> regs[VPT];

    // global -> registers.  base: operand pointer; ld: leading dim (elements); row0: first tile
    // row along the non-K dim; k0: first k; nrows/K: extents for zero-fill guards.
    __device__ __forceinline__ void load(const T* __restrict__ base, long ld, int row0, int k0, int nrows, int K) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = threadIdx.x + i * NTHREADS;
            bool ok = v < NVEC;
            long off = 0;
            if (KS) {
                const int kr = v / (ROWS / VEC), rv = v % (ROWS / VEC);
                ok = ok && (k0 + kr < K) && (row0 + rv * VEC < nrows);
                off = (long)(k0 + kr) * ld + row0 + rv * VEC;
            } else {
                const int r = v / (BK / VEC), kv = v % (BK / VEC);
                ok = ok && (row0 + r < nrows) && (k0 + kv * VEC < K);
                off = (long)(row0 + r) * ld + k0 + kv * VEC;
            }
            regs[i] = ok ? ld16<T>(base + off) : zero16<T>();
        }
    }
    // registers -> LDS
    __device__ __forceinline__ void store(T* lds) const {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int v = threadIdx.x + i * NTHREADS;
            if (v < NVEC) {
                int off;
                if (KS) {
                    const int kr = v / (ROWS / VEC), rv = v % (ROWS / VEC);
                    off = kr * LD + rv * VEC;
                } else {
                    const int r = v / (BK / VEC), kv = v % (BK / VEC);
                    off = r * LD + kv * VEC;
                }
                st16<T>(lds + off, regs[i]);
            }
        }
    }
    // LDS -> MFMA fragment for the 16 tile rows starting at r0: lane (c = l&15, g = l>>4) gets
    // element (row r0+c, k = 8g+j), j = 0..7.
    __device__ __forceinline__ static Frag<T> frag(const T* lds, int r0, int c, int g) {
        if constexpr (KS) return frag_ks<T, USE_TR>(lds, LD, r0, 0, c, g);
        else return frag_kc<T>(lds, LD, r0, 0, c, g);
    }
};

// ---- epilogue (shared by both main-loop variants) ----
// SR: rows staged per pass.  The wave's staging region is private (callers barrier once before the epilogue when the
// region aliases operand buffers), so passes need no workgroup barrier -- LDS instructions of one wave execute in
// order.  LOCAL: tight LDS budget (persistent kernel) -> no row pad, XOR swizzle instead.
//
// Memory-op ordering matters more than anything else here: vmcnt retires in order, so a wave that waits for a load
// issued AFTER its stores sits out the full store latency.  Every tensor the epilogue reads is therefore requested
// ahead of the stores it would otherwise queue behind: bias and the row map once per tile, and the per-element
// inputs of pass ps+1 (residual or GELU pre-activation, DropPath scale) before the stores of pass ps.
// TR: the accumulators hold the TRANSPOSED fragment layout of the LDS-DMA kernels (acc[r] = C[row c][column 4g + r], see
// epilogue_direct) -- only the staging write differs.
template <typename T, int BM, int BN, int WM, int WN, int SR = 16, bool TR = false>
__device__ __forceinline__ void gemm_epilogue(const esvit_gemm_desc& p, f32x4 (&acc)[BM / (16 * WM)][BN / (16 * WN)], char* smem_raw, int m0, int n0,
                                              int z) {
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr bool LOCAL = false;
    constexpr int FM = WTM / 16, FN = WTN / 16;
    const int M = p.M, N = p.N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    // Accumulator fragments hold a 4x1 column strip per lane (stride-16 columns), which would mean 2-byte
    // scattered stores.  Each wave therefore stages SR rows of its tile at a time through its own LDS region
    // and re-reads them as row-contiguous groups of 8 columns, so every global access of the epilogue (bias,
    // aux, residual, C) is a 16/32-byte vector.  Bank spread of the four lane groups (rows 4g+r): a 4-float
    // row pad, or (64-wide wave tiles, no room for a pad) an XOR of the 16-column block with g.
    constexpr bool SWZ = LOCAL && FN == 4;
    constexpr int LDE = SWZ ? WTN : WTN + 4;    // floats per staged row
    constexpr int CG = WTN / 8;                 // 8-column groups per row
    constexpr int ITEMS = (SR * CG + 63) / 64;  // groups per lane per pass
    constexpr int FPP = SR / 16;                // fragment rows per pass
    constexpr int NP = FM / FPP;                // passes
    constexpr bool CG_FIXED = (64 % CG) == 0;   // a lane keeps its column group across items -> bias loaded once
    static_assert(FM % FPP == 0 && (SR == 16 || SR == 32), "tile shape");
    float* stage = reinterpret_cast<float*>(smem_raw) + wave * (SR * LDE);
    const float alpha = p.alpha;

    auto stage_pass = [&](int ps) {  // accumulator rows of pass ps -> this wave's private LDS region
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int il = 0; il < FPP; ++il)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if constexpr (TR) {
                    static_assert(!TR || !SWZ, "transposed staging uses the padded rows");
                    *reinterpret_cast<f32x4*>(stage + (il * 16 + c) * LDE + j * 16 + 4 * g) = acc[FPP * ps + il][j] * alpha;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        stage[(il * 16 + 4 * g + r) * LDE + ((j * 16 + c) ^ (SWZ ? (g << 4) : 0))] = acc[FPP * ps + il][j][r] * alpha;
                }
            }
        __builtin_amdgcn_wave_barrier();
    };
    auto read_item = [&](int row_l, int cg, float (&v)[8]) {
        const int scol = (cg * 8) ^ (SWZ ? (((row_l >> 2) & 3) << 4) : 0);
        const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row_l * LDE + scol);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + row_l * LDE + scol + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = lo[e];
            v[4 + e] = hi[e];
        }
    };
    // item t of pass ps: 8 columns starting at n of row m (false: nothing to do)
    auto item_geom = [&](int ps, int t, int& row_l, int& cg, int& m, int& n) -> bool {
        const int id = lane + 64 * t;
        row_l = id / CG;
        cg = id % CG;
        m = m0 + wm * WTM + ps * SR + row_l;
        n = n0 + wn * WTN + cg * 8;
        return ((SR * CG) % 64 == 0 || id < SR * CG) && m < M && n < N;
    };

    if (p.splitk > 1) {  // fp32 partial of this split-K slice: no inputs, no conversions
        float* part = p.partial + (long)z * M * N;
        const bool pvec = (N % 4) == 0;
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            stage_pass(ps);
#pragma unroll
            for (int t = 0; t < ITEMS; ++t) {
                int row_l, cg, m, n;
                if (!item_geom(ps, t, row_l, cg, m, n)) continue;
                float v[8];
                read_item(row_l, cg, v);
                const int ne = min(8, N - n);
                float* dst = part + (long)m * N + n;
                if (ne == 8 && pvec) {
                    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
                } else {
                    for (int e = 0; e < ne; ++e) dst[e] = v[e];
                }
            }
        }
        return;
    }

    char* Cb = reinterpret_cast<char*>(p.C);
    const long c_batch = (long)z * p.strideC;
    T* auxp = reinterpret_cast<T*>(p.aux);
    const bool c_vec = (p.ldc % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) && ((c_batch % 8) == 0);
    const bool aux_vec = auxp && (p.ldaux % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.aux) & 15) == 0);
    const bool res_vec = p.residual && (p.ldr % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.residual) & 15) == 0);
    const bool bias_vec = p.bias && ((reinterpret_cast<uintptr_t>(p.bias) & 15) == 0);
    const int mode = p.epilogue;

    auto load_bias8 = [&](int n, float (&b)[8]) {
        const int ne = min(8, N - n);
        if (ne == 8 && bias_vec) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                b[e] = b0[e];
                b[4 + e] = b1[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) b[e] = e < ne ? p.bias[n + e] : 0.f;
        }
    };

    // once per tile: bias of this lane's column group(s) -- item t of every pass covers the same 8 columns
    constexpr int NBH = CG_FIXED ? 1 : ITEMS;
    float bias_h[NBH][8];
    static_for<NBH>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
#pragma unroll
        for (int e = 0; e < 8; ++e) bias_h[t][e] = 0.f;
        const int n = n0 + wn * WTN + ((lane + 64 * t) % CG) * 8;
        if (p.bias && n < N) load_bias8(n, bias_h[t]);
    });

    // general path (ragged edge tiles, row maps, unaligned operands, fp32 parity mode): inputs are read where they
    // are used -- the lean epilogue below covers the tiles that matter for speed
    auto dest_row = [&](int m, int tk) -> long { return p.rowmap ? (long)(m / p.rowmap_period) * p.rowmap_tokens + tk : (long)m; };
    static_for<NP>([&](auto psc) {
        constexpr int ps = decltype(psc)::value;
        stage_pass(ps);
        static_for<ITEMS>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            int row_l, cg, m, n;
            if (!item_geom(ps, t, row_l, cg, m, n)) return;
            int tk = 0;
            if (p.rowmap) {
                tk = p.rowmap[m % p.rowmap_period];
                if (tk < 0) return;
            }
            const int ne = min(8, N - n);
            const long drow = dest_row(m, tk);
            float v[8];
            read_item(row_l, cg, v);
            if (p.bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bias_h[CG_FIXED ? 0 : t][e];
            }
            if (mode == ESVIT_EPI_GELU || mode == ESVIT_EPI_QGELU) {
                if (auxp) {
                    T* ap = auxp + (long)m * p.ldaux + n;
                    if (ne == 8 && aux_vec) {
                        if constexpr (sizeof(T) == 2) {
                            bf16x8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
                            *reinterpret_cast<bf16x8*>(ap) = o;
                        } else {
                            *reinterpret_cast<f32x4*>(ap) = f32x4{v[0], v[1], v[2], v[3]};
                            *reinterpret_cast<f32x4*>(ap + 4) = f32x4{v[4], v[5], v[6], v[7]};
                        }
                    } else {
                        for (int e = 0; e < ne; ++e) ap[e] = from_f32<T>(v[e]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = mode == ESVIT_EPI_GELU ? gelu_f(v[e]) : qgelu_f(v[e]);
            } else if (mode == ESVIT_EPI_GELU_BWD || mode == ESVIT_EPI_QGELU_BWD) {
                const T* ap = auxp + (long)m * p.ldaux + n;
                float a[8];
                if (ne == 8 && aux_vec) {
                    if constexpr (sizeof(T) == 2) {
                        const bf16x8 x = *reinterpret_cast<const bf16x8*>(ap);
#pragma unroll
                        for (int e = 0; e < 8; ++e) a[e] = (float)x[e];
                    } else {
                        const f32x4 x0 = *reinterpret_cast<const f32x4*>(ap), x1 = *reinterpret_cast<const f32x4*>(ap + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            a[e] = x0[e];
                            a[4 + e] = x1[e];
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] = e < ne ? to_f32(ap[e]) : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= mode == ESVIT_EPI_GELU_BWD ? gelu_grad_f(a[e]) : qgelu_grad_f(a[e]);
            }
            if (p.rowscale) {
                const float rs = p.rowscale[drow / p.rows_per_sample];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= rs;
            }
            if (p.residual) {
                const float* rp = p.residual + drow * p.ldr + n;
                if (ne == 8 && res_vec) {
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] += r0[e];
                        v[4 + e] += r1[e];
                    }
                } else {
                    for (int e = 0; e < ne; ++e) v[e] += rp[e];
                }
            }
            const long o = c_batch + drow * p.ldc + n;
            if (p.out_f32) {
                float* cp = reinterpret_cast<float*>(Cb) + o;
                if (ne == 8 && c_vec) {
                    *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
                } else {
                    for (int e = 0; e < ne; ++e) cp[e] = v[e];
                }
            } else {
                T* cp = reinterpret_cast<T*>(Cb) + o;
                if (ne == 8 && c_vec) {
                    if constexpr (sizeof(T) == 2) {
                        bf16x8 ov;
#pragma unroll
                        for (int e = 0; e < 8; ++e) ov[e] = (bf16)v[e];
                        *reinterpret_cast<bf16x8*>(cp) = ov;
                    } else {
                        *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    }
                } else {
                    for (int e = 0; e < ne; ++e) cp[e] = from_f32<T>(v[e]);
                }
            }
        });
    });
}

// ---- direct epilogue of the LDS-DMA kernels: no LDS staging ----
// Those kernels issue every MFMA with the operands SWAPPED (the weight-side fragment in the A slot, the activation-side one
// in the B slot), which costs nothing in the main loop -- both fragments are "row c, k = 8g .. 8g+7" reads -- and leaves
// the TRANSPOSE of the usual accumulator layout behind:
//     acc[i][j][r] = C[row 16 i + c][column 16 j + 4 g + r]          (lane = 16 g + c)
// i.e. four CONSECUTIVE columns of one row per lane.  fp32 outputs, the fp32 residual and the bias are therefore plain 16-byte
// vectors straight from / to the accumulators (four lane groups = 64 contiguous bytes of a row); bf16 outputs pack two column
// blocks to 2 x 2 dwords and exchange one pair between the lane groups g and g ^ 1 (v_permlane16_swap), after which every lane
// holds eight consecutive columns = one 16-byte store.  The LDS round trip of the staged epilogue (64 ds_write_b32 + 16
// ds_read_b128 per 64 x 64 wave tile, plus its address arithmetic) is gone -- the GEMMs of this path are short-K and were
// bound by exactly that (profiles/r01_gemm_sq_counters.txt: ~9 VALU instructions per MFMA, most of them epilogue).
// Straight-line code per kind for a FULL interior tile with vector-aligned operands; everything else takes gemm_epilogue<TR>.
// Inputs of row block i+1 (residual / GELU pre-activation) are requested before the stores of row block i: vmcnt retires in order.
enum { EK_PLAIN = 0, EK_GELU = 1, EK_RES = 2, EK_GELU_BWD = 3 };

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t v = {(bf16)a, (bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// four columns (block j) | four columns (block j + 1) of one row per lane -> eight consecutive columns per lane:
// even g: block j, columns 4g .. 4g+7;  odd g: block j + 1, columns 4(g-1) .. 4(g-1)+7
__device__ __forceinline__ u32x4_t pair_rows(const f32x4& x, const f32x4& y) {
    const unsigned x0 = pack_bf16x2(x[0], x[1]), x1 = pack_bf16x2(x[2], x[3]);
    const unsigned y0 = pack_bf16x2(y[0], y[1]), y1 = pack_bf16x2(y[2], y[3]);
    const u32x2_t s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
    const u32x2_t s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
    return u32x4_t{s0[0], s1[0], s0[1], s1[1]};
}

// Output stores of the direct epilogues.  Non-temporal stores were measured in round 4 (profiles/r04_p8_timeline.txt,
// r04_gemm_instep_ab.txt, r04_pmc_traffic_nt.json): on the 256 x 128 two-set loop they stop the write stream from evicting the operand
// panels (187 -> 163 us on 87040 x 1536 x 384); on the 128-row kernels of the step they change nothing in time (four-way same-box A/B)
// and RAISE the traffic below the L2 (GEMM writes 30.4 -> 38.5 GB per step, reads 67.9 -> 61.4): plain stores here.
template <typename V>
__device__ __forceinline__ void store_stream(V* dst, const V& v) {
#ifdef ESVIT_NT_STORES  // tools/ab_routing.sh only
    __builtin_nontemporal_store(v, dst);
#else
    *dst = v;
#endif
}

// bf16 row block: v[j] = this lane's four columns of block j -> dst (the lane's row, at the wave tile's first column)
template <int FN>
__device__ __forceinline__ void store_row_bf16(bf16* dst, const f32x4 (&v)[FN], int g) {
    const int pc = 16 * (g & 1) + 4 * (g & ~1);  // column of the lane's 8-vector inside a block pair
#pragma unroll
    for (int j = 0; j + 1 < FN; j += 2) store_stream(reinterpret_cast<u32x4_t*>(dst + 16 * j + pc), pair_rows(v[j], v[j + 1]));
    if constexpr (FN & 1)
        store_stream(reinterpret_cast<u32x2_t*>(dst + 16 * (FN - 1) + 4 * g),
                     u32x2_t{pack_bf16x2(v[FN - 1][0], v[FN - 1][1]), pack_bf16x2(v[FN - 1][2], v[FN - 1][3])});
}

// core: one FM x FN block of transposed 16 x 16 accumulator fragments whose first row / first column are wrow0 / wcol0
template <int FM, int FN, int KIND, bool OUTF32, bool RS = false>
__device__ __forceinline__ void epilogue_direct_at(const esvit_gemm_desc& p, f32x4 (&acc)[FM][FN], long wrow0, int wcol0, void* Cbase, long ldc,
                                                   long c_first, const float* bias) {
    static_assert(!RS || (FN == 4 && !OUTF32 && KIND == EK_PLAIN), "row statistics: 64-column wave tiles, plain bf16 epilogue");
    const int lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const long row = wrow0 + c;      // this lane's row of row block 0
    const int col = wcol0 + 4 * g;  // this lane's first column of column block 0
    const int n0_stat = wcol0;

    f32x4 bias_h[FN];
    f32x4 cen_h[RS ? FN : 1];  // softmax statistics: centre * scale of the lane's columns
#pragma unroll
    for (int j = 0; j < FN; ++j) bias_h[j] = bias ? *reinterpret_cast<const f32x4*>(bias + col + 16 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (RS) {
#pragma unroll
        for (int j = 0; j < FN; ++j)
            cen_h[j] = p.rowstat_center ? *reinterpret_cast<const f32x4*>(p.rowstat_center + col + 16 * j) * p.rowstat_scale : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (p.alpha != 1.f) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] *= p.alpha;
    }
    const bool quick = p.epilogue == ESVIT_EPI_QGELU || p.epilogue == ESVIT_EPI_QGELU_BWD;  // QuickGELU instead of erf-GELU
    const long c_step = 16 * ldc;
    const long x_ld = KIND == EK_RES ? p.ldr : p.ldaux;
    const long c_off = c_first + row * ldc + (col - 4 * g);  // the wave tile's first column in this lane's row
    const long x_off = row * x_ld + col;
    bf16* auxp = reinterpret_cast<bf16*>(p.aux);

    // inputs of the next row block: residual (4 fp32 per block) + DropPath scale, or the GELU pre-activation (4 bf16 per block)
    f32x4 in_r[2][KIND == EK_RES ? FN : 1];
    u32x2_t in_a[2][KIND == EK_GELU_BWD ? FN : 1];
    float rs[2] = {1.f, 1.f};
    auto load_in = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (KIND == EK_RES) {
            const float* rp = p.residual + x_off + 16 * i * x_ld;
#pragma unroll
            for (int j = 0; j < FN; ++j) in_r[i & 1][j] = *reinterpret_cast<const f32x4*>(rp + 16 * j);
            if (p.rowscale) rs[i & 1] = p.rowscale[(row + 16 * i) / p.rows_per_sample];
        } else if constexpr (KIND == EK_GELU_BWD) {
            const bf16* ap = auxp + x_off + 16 * i * x_ld;
#pragma unroll
            for (int j = 0; j < FN; ++j) in_a[i & 1][j] = *reinterpret_cast<const u32x2_t*>(ap + 16 * j);
        }
    };
    if constexpr (KIND == EK_RES || KIND == EK_GELU_BWD) load_in(std::integral_constant<int, 0>{});

    static_for<FM>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr ((KIND == EK_RES || KIND == EK_GELU_BWD) && i + 1 < FM) load_in(std::integral_constant<int, i + 1>{});
        f32x4 v[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) v[j] = acc[i][j] + bias_h[j];
        if constexpr (KIND == EK_GELU) {
            if (auxp) store_row_bf16<FN>(auxp + x_off - 4 * g + 16 * i * x_ld, v, g);
            if (quick) {
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[j][e] = qgelu_f(v[j][e]);
            } else {
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[j][e] = gelu_f(v[j][e]);
            }
        } else if constexpr (KIND == EK_GELU_BWD) {
            f32x4 x[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const u32x2_t a = in_a[i & 1][j];
                x[j] = f32x4{__builtin_bit_cast(float, a[0] << 16), __builtin_bit_cast(float, a[0] & 0xffff0000u),
                             __builtin_bit_cast(float, a[1] << 16), __builtin_bit_cast(float, a[1] & 0xffff0000u)};
            }
            if (quick) {
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[j][e] *= qgelu_grad_f(x[j][e]);
            } else {
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[j][e] *= gelu_grad_f(x[j][e]);
            }
        } else if constexpr (KIND == EK_RES) {
#pragma unroll
            for (int j = 0; j < FN; ++j) v[j] = v[j] * rs[i & 1] + in_r[i & 1][j];
        }
        if constexpr (OUTF32) {
            float* cp = reinterpret_cast<float*>(Cbase) + c_off + 4 * g + i * c_step;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if (p.splitk > 1) *reinterpret_cast<f32x4*>(cp + 16 * j) = v[j];
                else store_stream(reinterpret_cast<f32x4*>(cp + 16 * j), v[j]);
            }
        } else {
            store_row_bf16<FN>(reinterpret_cast<bf16*>(Cbase) + c_off + i * c_step, v, g);
            if constexpr (RS) {
                // (max, sum 2^(z - max)) over this row's 64 columns of z = (stored logit - centre) * scale: 16 columns here,
                // then the four lane groups that share the row
                float z[FN][4];
                float m = -3.0e38f;
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        z[j][e] = fmaf((float)(bf16)v[j][e], p.rowstat_scale, -cen_h[j][e]);
                        m = fmaxf(m, z[j][e]);
                    }
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sum += __builtin_amdgcn_exp2f(z[j][e] - m);
                sum += __shfl_xor(sum, 16, 64);
                sum += __shfl_xor(sum, 32, 64);
                if (g == 0) {
                    float* sp = p.rowstat + ((row + 16 * i) * (p.N >> 6) + (n0_stat >> 6)) * 2;
                    *reinterpret_cast<f32x2*>(sp) = f32x2{m, sum};
                }
            }
        }
    });
    if constexpr (RS) {
        // esvit_gemm_desc::colstat: the stored logits (alpha = 1, no bias on this path: the accumulators rounded to bf16) summed over the
        // wave tile's 16 * FM rows per column.  The 16 rows of a block sit in lanes c = 0 .. 15: fold the row blocks in registers, then the
        // lanes; lane c == 0 of every group writes its 4 x FN columns.  (Kept out of the row loop above: one column block at a time.)
        // (the pointer is fetched from the kernel-argument segment HERE: as one more descriptor field live across the main loop it
        // pushed two scalars of every instance of the kernel into scratch memory, whose reloads cost this epilogue ~1 us per tile)
        // (valid because every kernel that instantiates this epilogue takes the descriptor BY VALUE AS ITS FIRST ARGUMENT: the
        // kernel-argument segment then starts with it -- gemm_dma_kernel / gemm_kernel, checked where they are declared)
        float* colstat;
        asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                     : "=s"(colstat)
                     : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(offsetof(esvit_gemm_desc, colstat))
                     : "memory");
        if (colstat) {
            float* cp = colstat + (wrow0 / (16 * FM)) * (long)p.N + col;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] += (float)(bf16)acc[i][j][e];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = row16_sum(t[e]);
                if (c == 0) *reinterpret_cast<f32x4*>(cp + 16 * j) = t;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int KIND, bool OUTF32, bool RS = false>
__device__ __forceinline__ void epilogue_direct(const esvit_gemm_desc& p, f32x4 (&acc)[BM / (16 * WM)][BN / (16 * WN)], int m0, int n0, void* Cbase,
                                                long ldc, long c_first, const float* bias) {
    constexpr int WTM = BM / WM, WTN = BN / WN;
    const int wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    epilogue_direct_at<WTM / 16, WTN / 16, KIND, OUTF32, RS>(p, acc, (long)m0 + wm * WTM, n0 + wn * WTN, Cbase, ldc, c_first, bias);
}

// LDS-DMA kernels: the direct epilogue when the tile and the operands allow it, else the general (staged) one
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue_bf16(const esvit_gemm_desc& p, f32x4 (&acc)[BM / (16 * WM)][BN / (16 * WN)], char* smem_raw, int m0, int n0,
                                                   int z) {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (p.splitk > 1 && m0 + BM <= p.M && n0 + BN <= p.N && (p.N % 4 == 0) && al16(p.partial)) {
        // split-K partial of a full tile: a plain fp32 store into this slice's [M, N] plane
        epilogue_direct<BM, BN, WM, WN, EK_PLAIN, true>(p, acc, m0, n0, p.partial + (long)z * p.M * p.N, p.N, 0, nullptr);
        return;
    }
    bool fast = p.splitk <= 1 && !p.rowmap && m0 + BM <= p.M && n0 + BN <= p.N && (p.ldc % 8 == 0) && al16(p.C) &&
                ((p.strideC * (long)z) % 8 == 0) && (!p.bias || al16(p.bias));
    int kind = EK_PLAIN;
    if (p.epilogue == ESVIT_EPI_GELU || p.epilogue == ESVIT_EPI_QGELU) {
        kind = EK_GELU;
        fast = fast && !p.residual && !p.rowscale && !p.out_f32 && (!p.aux || ((p.ldaux % 8 == 0) && al16(p.aux)));
    } else if (p.epilogue == ESVIT_EPI_GELU_BWD || p.epilogue == ESVIT_EPI_QGELU_BWD) {
        kind = EK_GELU_BWD;
        fast = fast && !p.residual && !p.rowscale && (p.ldaux % 8 == 0) && al16(p.aux);
    } else if (p.residual) {
        kind = EK_RES;
        fast = fast && (p.ldr % 4 == 0) && al16(p.residual);
    } else {
        fast = fast && !p.rowscale;
    }
    if (!fast) {
        __syncthreads();  // the staging regions alias the operand buffers: every wave is done reading them
        gemm_epilogue<bf16, BM, BN, WM, WN, 16, true>(p, acc, smem_raw, m0, n0, z);
        return;
    }
    if (kind == EK_GELU) epilogue_direct<BM, BN, WM, WN, EK_GELU, false>(p, acc, m0, n0, p.C, p.ldc, (long)z * p.strideC, p.bias);
    else if (kind == EK_GELU_BWD) {
        if (p.out_f32) epilogue_direct<BM, BN, WM, WN, EK_GELU_BWD, true>(p, acc, m0, n0, p.C, p.ldc, (long)z * p.strideC, p.bias);
        else epilogue_direct<BM, BN, WM, WN, EK_GELU_BWD, false>(p, acc, m0, n0, p.C, p.ldc, (long)z * p.strideC, p.bias);
    } else if (kind == EK_RES) {
        if (p.out_f32) epilogue_direct<BM, BN, WM, WN, EK_RES, true>(p, acc, m0, n0, p.C, p.ldc, (long)z * p.strideC, p.bias);
        else epilogue_direct<BM, BN, WM, WN, EK_RES, false>(p, acc, m0, n0, p.C, p.ldc, (long)z * p.strideC, p.bias);
    } else {
        if (p.out_f32) epilogue_direct<BM, BN, WM, WN, EK_PLAIN, true>(p, acc, m0, n0, p.C, p.ldc, (long)z * p.strideC, p.bias);
        else if constexpr (BN / WN == 64) {
            if (p.rowstat) epilogue_direct<BM, BN, WM, WN, EK_PLAIN, false, true>(p, acc, m0, n0, p.C, p.ldc, (long)z * p.strideC, p.bias);
            else epilogue_direct<BM, BN, WM, WN, EK_PLAIN, false>(p, acc, m0, n0, p.C, p.ldc, (long)z * p.strideC, p.bias);
        } else {
            epilogue_direct<BM, BN, WM, WN, EK_PLAIN, false>(p, acc, m0, n0, p.C, p.ldc, (long)z * p.strideC, p.bias);
        }
    }
}

// XCD-aware work order: tiles on grid.x, split-K slice / batch item on grid.y.  The dispatcher places block b on XCD
// b % 8; every XCD gets a contiguous range of tile ids (bijective remap), so the tiles of one output row panel share
// an L2.  Used for nz == 1 and batched problems; split-K uses the z-major list below.  (Round 1 measured z-major 8 %
// slower with the register-staged loop, profiles/r01_gemm_xcdmap_ab.txt; with the LDS-DMA loop the fabric traffic is
// what limits the weight gradients and z-major wins on every shape, profiles/r02_gemm_workorder_ab.txt.)
__device__ __forceinline__ void xcd_tile_map(int ntiles, int& tile, int& z) {
    const int b = blockIdx.x;
    const int q = ntiles / 8, r = ntiles % 8;
    const int xcd = b % 8, idx = b / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    z = blockIdx.y;
}
// z-major variant: the whole (z, tile) grid is one work list, tile fastest; every XCD gets a contiguous range, i.e. all
// tiles of ~nz/8 split-K slices, so each K slice of A and B is fetched by one XCD only.
__device__ __forceinline__ void xcd_work_map_zmajor(int ntiles, int& tile, int& z) {
    const int total = ntiles * gridDim.y;
    const int L = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = total / 8, r = total % 8;
    const int xcd = L % 8, idx = L / 8;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    z = w / ntiles;
    tile = w - z * ntiles;
}

// Tile id -> (row block, column block).  group_m <= 1: column-fastest.  Otherwise the ids walk down group_m row blocks
// before moving to the next column block: the ~64 tiles one XCD has in flight then cover group_m row panels x
// 64/group_m column panels instead of 1 x 64, so a wide B (N/BN >> 8, e.g. the 65536-wide last layer) is not
// re-streamed from the Infinity Cache once per row block.
__device__ __forceinline__ void tile_coords(int pid, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
    if (group_m <= 1) {
        tm = pid / tiles_n;
        tn = pid - tm * tiles_n;
        return;
    }
    const int gsize = group_m * tiles_n;
    const int gid = pid / gsize;
    const int first = gid * group_m;
    const int rows = min(tiles_m - first, group_m);
    const int w = pid - gid * gsize;
    tn = w / rows;
    tm = first + (w - tn * rows);
}

template <typename T, bool AKS, bool BKS, int BM, int BN, bool USE_TR>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const esvit_gemm_desc p) {
    // (the descriptor stays the FIRST argument, by value: the statistics epilogue reads esvit_gemm_desc::colstat from the
    // kernel-argument segment at offsetof(esvit_gemm_desc, colstat), epilogue_direct_at<RS>)
    using TA = Tile<T, AKS, BM, USE_TR>;
    using TB = Tile<T, BKS, BN, USE_TR>;
    constexpr int WTM = BM / 2, WTN = BN / 2;  // wave tile
    constexpr int FM = WTM / 16, FN = WTN / 16;

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* smem = reinterpret_cast<T*>(smem_raw);
    T* sA = smem;                   // 2 buffers
    T* sB = smem + 2 * TA::ELEMS;   // 2 buffers

    const int M = p.M, N = p.N, K = p.K;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    // XCD-aware tile order: block b runs on XCD b%8; give every XCD a contiguous range of
    // tile ids so the tiles that share an A panel hit the same L2 (bijective remap).
    const int ntiles = tiles_m * tiles_n;
    int pid, z;
    xcd_tile_map(ntiles, pid, z);
    const int tm = pid / tiles_n, tn = pid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const T* A = reinterpret_cast<const T*>(p.A);
    const T* B = reinterpret_cast<const T*>(p.B);
    int kbeg = 0, kend = K;
    if (p.splitk > 1) {
        const int nkt = (K + BK - 1) / BK;
        const int per = (nkt + p.splitk - 1) / p.splitk;
        kbeg = z * per * BK;
        kend = min(K, (z + 1) * per * BK);
    } else {
        A += (long)z * p.strideA;
        B += (long)z * p.strideB;
    }

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool do_colsum = p.colsum && tn == 0 && wn == 0;
    f32x4 accb[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const Frag<T> ones = ones_frag(T());

    TA ta;
    TB tb;
    const int nk = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
    if (nk > 0) {
        ta.load(A, p.lda, m0, kbeg, M, kend);
        tb.load(B, p.ldb, n0, kbeg, N, kend);
        ta.store(sA);
        tb.store(sB);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            ta.load(A, p.lda, m0, kbeg + (kt + 1) * BK, M, kend);
            tb.load(B, p.ldb, n0, kbeg + (kt + 1) * BK, N, kend);
        }
        const T* a_lds = sA + cur * TA::ELEMS;
        const T* b_lds = sB + cur * TB::ELEMS;
        Frag<T> af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = TA::frag(a_lds, wm * WTM + i * 16, c, g);
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[j] = TB::frag(b_lds, wn * WTN + j * 16, c, g);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) mma(af[i], bfr[j], acc[i][j]);
        if (do_colsum) {
#pragma unroll
            for (int i = 0; i < FM; ++i) mma(af[i], ones, accb[i]);
        }
        if (kt + 1 < nk) {
            ta.store(sA + (cur ^ 1) * TA::ELEMS);
            tb.store(sB + (cur ^ 1) * TB::ELEMS);
        }
        __syncthreads();
    }

    if (do_colsum) store_colsum<FM>(p, accb, m0, wm * WTM, z, c, g);
    gemm_epilogue<T, BM, BN, 2, 2>(p, acc, smem_raw, m0, n0, z);
}

// =================================================================================================
// bf16 fast path: operands travel HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), BK = 64.
// No staging registers, no ds_write pass; the LDS image is lane-linear (1 KiB per wave instruction), so
// bank conflicts are removed by an XOR swizzle applied to the per-lane SOURCE address and to the
// fragment read address (guide rule 21).  k-contiguous tiles: [ROWS][64] with chunk ^= row & 7
// (conflict-free ds_read_b128); k-strided tiles: [64][ROWS] with a per-k chunk XOR that spreads the
// 8 k-rows touched by one ds_read_b64_tr_b16 over distinct banks.  NBUF LDS buffers; the barrier at the
// end of a k-tile drains the DMA of the next one while this tile's MFMAs run.
//
// The workgroup is WM x WN waves (NT = 64 WM WN threads).  Two shapes are shipped:
//   2 x 2 waves, 128 x {64, 96, 128} tiles, two workgroups per CU  -- small / short-K problems, whose heavy
//                 epilogues then overlap the other workgroup's main loop;
//   8 waves,     256 x {192, 256} (and 192 x 192 for the weight gradients) tiles, one workgroup per CU -- half the
//                 operand bytes per FLOP of the 128-wide tiles (the 128 x 128 loop needs 63 B / clk / CU from the L2
//                 at the MFMA rate, more than the L2 delivers; profiles/r01_gemm_mainloop_probe.txt).
// =================================================================================================
template <bool KS, int ROWS, int BKD, int NT>
struct DmaTile {
    static constexpr int ELEMS = ROWS * BKD;
    static constexpr int CHUNKS = ELEMS / 8;
    static constexpr int INSTR_PER_WAVE = CHUNKS / NT;
    static constexpr int CPR = KS ? ROWS / 8 : BKD / 8;  // 16-byte chunks per LDS row
    static_assert(CHUNKS % NT == 0, "tile must be a whole number of instructions per wave");

    // k-strided image [BKD][ROWS]: one ds_read_b64_tr_b16 of a half wave touches k-rows {k0..k0+3, k0+8..k0+11}, 32
    // bytes each at the same column offset; banks repeat every 256 bytes, so the chunk index is XOR-ed per k-row such
    // that the eight 32-byte pieces fall into eight different 32-byte segments of the 256-byte bank row.  Row pitch
    // 2*ROWS bytes: 256 / 512 (ROWS 128 / 256): every k-row starts on the same bank -> move by (k & 3) * 32 B and
    // bit 3 of k * 128 B; 384 (ROWS 192): k-rows alternate between offsets 0 and 128 -> bit 1 of k * 32 B, bit 3 * 64 B;
    // 128 (ROWS 64): offsets 0 / 128 alternate and the row has only 8 chunks; 192 (ROWS 96): four 64-byte phases.
    __device__ __forceinline__ static int sw_ks(int k) {
        if constexpr (ROWS == 128 || ROWS == 256) return ((k & 3) << 1) | (((k >> 3) & 1) << 3);
        else if constexpr (ROWS == 192) return (((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2);
        else if constexpr (ROWS == 64) return (((k >> 1) & 1) << 1) | (((k >> 3) & 1) << 2);
        else return ((k >> 3) & 1) << 1;
    }
    __device__ __forceinline__ static int sw_kc(int r) {
        if constexpr (BKD == 64) return r & 7;               // 128-byte rows: 8 chunks
        else return (0x78 >> (2 * ((r >> 2) & 3))) & 3;      // 64-byte rows: 4 chunks, f(r>>2) = {0,2,3,1}
    }

    // one DMA instruction: 64 lanes x 16 bytes -> LDS bytes [slot * 1024, slot * 1024 + 1024) of the tile
    __device__ __forceinline__ static void issue_slot(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, long ld, int rows_left, int k0, int K,
                                                      int slot, int lane) {
        typedef __attribute__((address_space(3))) void lds_void;
        const int p = slot * 64 + lane;
        long off;
        bool ok;
        if constexpr (KS) {
            const int kr = p / CPR, cp = p % CPR;
            const int col = (cp ^ sw_ks(kr)) * 8;
            ok = (k0 + kr < K) && (col < rows_left);
            off = (long)(k0 + kr) * ld + col;
        } else {
            const int r = p / CPR, cp = p % CPR;
            const int k = k0 + ((cp ^ sw_kc(r)) * 8);
            ok = (r < rows_left) && (k < K);
            off = (long)r * ld + k;
        }
        const int voff = ok ? (int)(off * 2) : (int)0xfffffff8u;  // beyond num_records -> hardware returns 0
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds_tile + slot * 1024), 16, voff, 0, 0, 0);
    }

    // base: operand pointer already advanced to the tile's first row (k-contiguous) / column (k-strided).
    // The tile's instructions are spread over the workgroup's waves.
    __device__ __forceinline__ static void issue(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, long ld, int rows_left, int k0, int K,
                                                 int wave, int lane) {
#pragma unroll
        for (int i = 0; i < INSTR_PER_WAVE; ++i) issue_slot(rsrc, lds_tile, ld, rows_left, k0, K, wave * INSTR_PER_WAVE + i, lane);
    }

    // Fast path of issue() for a FULL k-tile: the per-lane byte offsets of the wave's instructions relative to (tile base,
    // k0 = 0) depend only on ld, so they are computed once per kernel and every k-tile is INSTR_PER_WAVE bare DMA
    // instructions with k0 folded into the scalar offset.  Rows past the end of a k-contiguous operand fall outside the
    // descriptor's num_records (checked on the vector offset) and read as 0; a k-strided operand's partial row tile and
    // any partial k-tile need per-lane predicates -> issue().
    __device__ __forceinline__ static void wave_offsets(long ld, int wave, int lane, int (&voff)[INSTR_PER_WAVE]) {
        static_for<INSTR_PER_WAVE>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int p = (wave * INSTR_PER_WAVE + i) * 64 + lane;
            const int r = p / CPR, cp = p % CPR;
            const long off = KS ? (long)r * ld + (cp ^ sw_ks(r)) * 8 : (long)r * ld + (cp ^ sw_kc(r)) * 8;
            voff[i] = (int)(off * 2);
        });
    }
    __device__ __forceinline__ static void issue_fast(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, long ld, int k0, int wave,
                                                      const int (&voff)[INSTR_PER_WAVE]) {
        typedef __attribute__((address_space(3))) void lds_void;
        const int soff = (int)(KS ? (long)k0 * ld * 2 : (long)k0 * 2);
        static_for<INSTR_PER_WAVE>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds_tile + (wave * INSTR_PER_WAVE + i) * 1024), 16, voff[i], soff, 0, 0);
        });
    }

    // fragment for the 16 tile rows at r0, k-step kk (32 deep)
    __device__ __forceinline__ static Frag<bf16> frag(const bf16* lds, int r0, int kk, int c, int g) {
        Frag<bf16> f;
        if constexpr (!KS) {
            const int row = r0 + c;
            const int pos = row * CPR + ((kk * 4 + g) ^ sw_kc(row));
            f.v = *reinterpret_cast<const bf16x8*>(lds + pos * 8);
        } else {
            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
            const int k_lo = kk * 32 + 8 * g + (c >> 2);
            const int chunk = (r0 >> 3) + ((c & 3) >> 1);
            const bf16* p0 = lds + (k_lo * CPR + (chunk ^ sw_ks(k_lo))) * 8 + (c & 1) * 4;
            const int k_hi = k_lo + 4;
            const bf16* p1 = lds + (k_hi * CPR + (chunk ^ sw_ks(k_hi))) * 8 + (c & 1) * 4;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
            const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            f.v = __builtin_bit_cast(bf16x8, both);
        }
        return f;
    }
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, long bytes_left) {
    const long capped = bytes_left > 0xfffffff0L ? 0xfffffff0L : (bytes_left < 0 ? 0 : bytes_left);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)capped, 0x00020000);
}

// counted wait: at most N of this wave's LDS-DMA loads still in flight (loads retire in order)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NBUF-deep ring of LDS tiles: NBUF-1 tiles are requested ahead; per k-tile ONE counted wait (only for the tile
// about to be consumed -- later tiles stay in flight across the barrier) and ONE barrier.
#ifdef ESVIT_PROBE_TIMELINE
// tools/probe only: per-workgroup phase timestamps (s_memrealtime, 100 MHz) and placement, to see how the co-resident
// workgroups of a CU interleave their main loops and epilogues
__device__ long* g_probe_timeline = nullptr;
#define ESVIT_TL(slot) do { if (g_probe_timeline && threadIdx.x == 0) g_probe_timeline[((long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (slot)] = (long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ESVIT_TL(slot) do { } while (0)
#endif

template <bool AKS, bool BKS, int BM, int BN, int BKD, int NBUF, int WM, int WN, int MINB = 2, bool EARLY = false>
__global__ __launch_bounds__(64 * WM * WN, MINB) void gemm_dma_kernel(const esvit_gemm_desc p, const int group_m) {
    // (the descriptor stays the FIRST argument, by value: the statistics epilogue reads esvit_gemm_desc::colstat from the
    // kernel-argument segment at offsetof(esvit_gemm_desc, colstat), epilogue_direct_at<RS>)
    constexpr int NT = 64 * WM * WN;
    ESVIT_TL(0);
#ifdef ESVIT_PROBE_TIMELINE
    if (g_probe_timeline && threadIdx.x == 0) {
        long* e_ = g_probe_timeline + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        e_[3] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
        e_[4] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    }
#endif
    using TA = DmaTile<AKS, BM, BKD, NT>;
    using TB = DmaTile<BKS, BN, BKD, NT>;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int FM = WTM / 16, FN = WTN / 16;
    static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile");
    constexpr int L = TA::INSTR_PER_WAVE + TB::INSTR_PER_WAVE;  // DMA instructions per wave per tile
    static_assert(NBUF >= 2 && NBUF <= 4, "ring depth");
    static_assert((NBUF - 2) * L < 64, "vmcnt is 6 bits");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int A_BYTES = TA::ELEMS * 2, B_BYTES = TB::ELEMS * 2;
    char* sA = smem_raw;                   // NBUF buffers
    char* sB = smem_raw + NBUF * A_BYTES;  // NBUF buffers

    const int M = p.M, N = p.N, K = p.K;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int ntiles = tiles_m * tiles_n;
    int pid, z;
    if (group_m < 0) xcd_work_map_zmajor(ntiles, pid, z);
    else xcd_tile_map(ntiles, pid, z);
    int tm, tn;
    tile_coords(pid, tiles_m, tiles_n, group_m, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const bf16* A = reinterpret_cast<const bf16*>(p.A);
    const bf16* B = reinterpret_cast<const bf16*>(p.B);
    int kbeg = 0, kend = K;
    if (p.splitk > 1) {
        const int nkt = (K + BKD - 1) / BKD;
        const int per = (nkt + p.splitk - 1) / p.splitk;
        kbeg = z * per * BKD;
        kend = min(K, (z + 1) * per * BKD);
    } else {
        A += (long)z * p.strideA;
        B += (long)z * p.strideB;
    }
    // per-block descriptors: base at the tile's first row / column, so every byte offset fits 32 bits
    const long a_rows_total = AKS ? (long)K : (long)M;  // rows of the stored matrix
    const long b_rows_total = BKS ? (long)K : (long)N;
#ifdef ESVIT_PROBE_A_RESIDENT  // tools/probe only: every row block reads the rows of row block 0 (an always-L2-resident A operand)
    const bf16* a_base = A;
#else
    const bf16* a_base = AKS ? A + m0 : A + (long)m0 * p.lda;
#endif
    const bf16* b_base = BKS ? B + n0 : B + (long)n0 * p.ldb;
    const long a_left = ((AKS ? a_rows_total : a_rows_total - m0) * p.lda - (AKS ? m0 : 0)) * 2;
    const long b_left = ((BKS ? b_rows_total : b_rows_total - n0) * p.ldb - (BKS ? n0 : 0)) * 2;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(a_base, a_left);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(b_base, b_left);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool do_colsum = p.colsum && tn == 0 && wn == 0;
    f32x4 accb[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const Frag<bf16> ones = ones_frag(bf16());

    const int nk = (kend > kbeg) ? (kend - kbeg + BKD - 1) / BKD : 0;
    // full k-tiles of row-complete operand tiles take the bare-DMA path (offsets precomputed, k0 in the scalar offset)
    int voffA[TA::INSTR_PER_WAVE], voffB[TB::INSTR_PER_WAVE];
    TA::wave_offsets(p.lda, wave, lane, voffA);
    TB::wave_offsets(p.ldb, wave, lane, voffB);
    const bool rows_ok_a = !AKS || (M - m0 >= BM), rows_ok_b = !BKS || (N - n0 >= BN);
    auto issue_tile = [&](int t, int slot) {
        const int k0 = kbeg + t * BKD;
        const bool fullk = k0 + BKD <= kend;
        if (fullk && rows_ok_a) TA::issue_fast(ra, sA + slot * A_BYTES, p.lda, k0, wave, voffA);
        else TA::issue(ra, sA + slot * A_BYTES, p.lda, M - m0, k0, kend, wave, lane);
        if (fullk && rows_ok_b) TB::issue_fast(rb, sB + slot * B_BYTES, p.ldb, k0, wave, voffB);
        else TB::issue(rb, sB + slot * B_BYTES, p.ldb, N - n0, k0, kend, wave, lane);
    };
    if constexpr (EARLY) {
        // "early release": a wave copies ALL fragments of k-tile kt into registers first, so the LDS buffer is free again
        // before the MFMAs start and the DMA of k-tile kt+2 goes into it -- two k-tiles in flight with two buffers.
        static_assert(NBUF == 2, "early release works on two buffers");
        constexpr int KK = BKD / 32;
        if (nk > 0) issue_tile(0, 0);
        if (nk > 1) issue_tile(1, 1);
        for (int kt = 0; kt < nk; ++kt) {
            const int b = kt & 1;
            if (kt + 1 < nk) wait_vmcnt<L>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();  // k-tile kt landed for every wave
            asm volatile("" ::: "memory");
            const bf16* a_lds = reinterpret_cast<const bf16*>(sA + b * A_BYTES);
            const bf16* b_lds = reinterpret_cast<const bf16*>(sB + b * B_BYTES);
            Frag<bf16> af[KK][FM], bfr[KK][FN];
#ifdef ESVIT_PROBE_SKIP_FRAG  // tools/probe only: the loop without its LDS reads (fragments of k-tile 0 only) -- what the DMA path alone sustains
            if (kt == 0)
#endif
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                for (int i = 0; i < FM; ++i) af[kk][i] = TA::frag(a_lds, wm * WTM + i * 16, kk, c, g);
#pragma unroll
                for (int j = 0; j < FN; ++j) bfr[kk][j] = TB::frag(b_lds, wn * WTN + j * 16, kk, c, g);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave holds its fragments: buffer b is free
            asm volatile("" ::: "memory");
            if (kt + 2 < nk) issue_tile(kt + 2, b);
#ifdef ESVIT_PROBE_SKIP_MFMA  // tools/probe only: one MFMA column per k-tile instead of all
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int i = 0; i < FM; ++i) mma(bfr[kk][i % FN], af[kk][i], acc[i][i % FN]);
            if (false)
#endif
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) mma(bfr[kk][j], af[kk][i], acc[i][j]);  // operands swapped: see epilogue_direct
                if constexpr (AKS) {
                    if (do_colsum) {
#pragma unroll
                        for (int i = 0; i < FM; ++i) mma(af[kk][i], ones, accb[i]);
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < NBUF - 1; ++t) {
            if (t < nk) issue_tile(t, t);
        }
        int buf = 0;  // ring slot of tile kt
        for (int kt = 0; kt < nk; ++kt) {
            const int ahead = min(nk - 1 - kt, NBUF - 2);  // tiles requested after kt that may stay in flight
            if (NBUF >= 4 && ahead >= 2) wait_vmcnt<(NBUF >= 4 ? 2 : 0) * L>();
            else if (NBUF >= 3 && ahead >= 1) wait_vmcnt<(NBUF >= 3 ? 1 : 0) * L>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();  // tile kt landed for every wave; every wave is done with tile kt-1
            asm volatile("" ::: "memory");
            const int nt = kt + NBUF - 1;
            if (nt < nk) {
                const int nb = (buf == 0) ? NBUF - 1 : buf - 1;  // the slot tile kt-1 just vacated
                issue_tile(nt, nb);
            }
            const bf16* a_lds = reinterpret_cast<const bf16*>(sA + buf * A_BYTES);
            const bf16* b_lds = reinterpret_cast<const bf16*>(sB + buf * B_BYTES);
    #pragma unroll
            for (int kk = 0; kk < BKD / 32; ++kk) {
                Frag<bf16> af[FM], bfr[FN];
    #pragma unroll
                for (int i = 0; i < FM; ++i) af[i] = TA::frag(a_lds, wm * WTM + i * 16, kk, c, g);
    #pragma unroll
                for (int j = 0; j < FN; ++j) bfr[j] = TB::frag(b_lds, wn * WTN + j * 16, kk, c, g);
    #pragma unroll
                for (int i = 0; i < FM; ++i)
    #pragma unroll
                    for (int j = 0; j < FN; ++j) mma(bfr[j], af[i], acc[i][j]);  // operands swapped: see epilogue_direct
                if constexpr (AKS) {  // the fused bias gradient exists for wgrad only: no branch in the fwd / dgrad loops
                    if (do_colsum) {
    #pragma unroll
                        for (int i = 0; i < FM; ++i) mma(af[i], ones, accb[i]);
                    }
                }
            }
            buf = (buf + 1 == NBUF) ? 0 : buf + 1;
        }
}
    ESVIT_TL(1);
    if constexpr (AKS) {
        if (do_colsum) store_colsum<FM>(p, accb, m0, wm * WTM, z, c, g);
    }
    gemm_epilogue_bf16<BM, BN, WM, WN>(p, acc, smem_raw, m0, n0, z);  // (the staged fallback barriers before it reuses the LDS)
    ESVIT_TL(2);
}

// =================================================================================================
// A-stationary short-K kernel (round 6).  C[M,N] = A[M,K] B^T (+ bias, GELU), A row-major, B = weight [N, K] row-major, bf16 out.
//
// The short-K forwards of the step (K = 256 / 384: stage-2 fc1 / qkv) spend their time delivering operands: a 128 x 128 tile of the loop
// above fetches 2 x 128 x K x 2 bytes for 128 x 128 x K MACs -- 1.6 GB through the L2 for fc1's 0.6 GB of algorithmic traffic -- and its
// four to six k-tiles are all pipeline fill and drain.  Here a workgroup owns 128 rows and WALKS over column tiles: every wave loads the
// fragments of ITS 32 rows (4 x 1 waves: whole K, K / 8 registers per lane) straight from global memory once, and only the weight streams
// through the LDS-DMA ring -- continuously, across the column tiles, so the ring never drains: the epilogue of tile t (stores only; the bias
// of the walked columns sits in LDS) runs while the first k-tiles of tile t + 1 are in flight.  Half the operand bytes per tile, no per-tile
// prologue.  A PROBE (-DESVIT_ASTAT, profiles/r06_gemm_astat_probe.txt): -8 % on fc1 with its two outputs, +3..+39 % elsewhere, not routed.
// Work item = (row block, column chunk) = (blockIdx / chunks, blockIdx % chunks): with chunks dividing 8 or a multiple of 8 the workgroups of
// an XCD walk the same weight columns (its L2 holds them).
// =================================================================================================
template <int KSTEPS, bool GELU, bool PREACT, int NBUF = 4>
__global__ __launch_bounds__(256, 2) void gemm_astat_kernel(const esvit_gemm_desc p, const int tiles_per_wg, const int chunks) {
    constexpr int BN = 128, BKD = 64, NT = 256;
    constexpr int K = KSTEPS * 32, NKT = K / BKD, FM = 2, FN = 8;
    static_assert(K % BKD == 0, "whole k-tiles");
    using TB = DmaTile<false, BN, BKD, NT>;
    constexpr int L = TB::INSTR_PER_WAVE;
    constexpr int B_BYTES = TB::ELEMS * 2;
    static_assert((NBUF - 2) * L < 64, "vmcnt is 6 bits");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* sB = smem_raw;                                                   // NBUF buffers
    float* sBias = reinterpret_cast<float*>(smem_raw + NBUF * B_BYTES);    // tiles_per_wg * BN floats

    const int tiles_n = p.N / BN;
    const int rb = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
    const int tile0 = chunk * tiles_per_wg;
    const int ntile = min(tiles_per_wg, tiles_n - tile0);
    if (ntile <= 0) return;
    const int m0 = rb * 128;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const bf16* A = reinterpret_cast<const bf16*>(p.A);
    const bf16* B = reinterpret_cast<const bf16*>(p.B);
    const bf16* b_base = B + (long)tile0 * BN * p.ldb;
    const __amdgpu_buffer_rsrc_t rbuf = make_rsrc(b_base, ((long)(p.N - tile0 * BN) * p.ldb) * 2);
    int voffB[L];
    TB::wave_offsets(p.ldb, wave, lane, voffB);
    const int total = ntile * NKT;
    auto issue = [&](int s_, int slot) {  // k-tile s_ of the walk: column tile s_ / NKT, k-tile s_ % NKT
        typedef __attribute__((address_space(3))) void lds_void;
        const int t = s_ / NKT, kt = s_ - t * NKT;
        const int soff = (int)(((long)t * BN * p.ldb + kt * BKD) * 2);
        static_for<L>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rbuf, (lds_void*)(sB + slot * B_BYTES + (wave * L + i) * 1024), 16, voffB[i], soff, 0, 0);
        });
    };
    // this wave's rows of A, whole K: fragment (row block i, k-step ks) = 8 consecutive k of row 16 i + c at k = 32 ks + 8 g
    Frag<bf16> af[FM][KSTEPS];
    {
        const bf16* ap = A + (long)(m0 + wave * 32 + c) * p.lda + 8 * g;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) af[i][ks].v = *reinterpret_cast<const bf16x8*>(ap + (long)16 * i * p.lda + 32 * ks);
    }
    if (p.bias) {
        for (int e = threadIdx.x; e < ntile * BN; e += NT) sBias[e] = p.bias[tile0 * BN + e];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // A fragments and the bias have arrived: from here on only ring DMAs and stores are counted
    __syncthreads();
#pragma unroll
    for (int s_ = 0; s_ < NBUF - 1; ++s_)
        if (s_ < total) issue(s_, s_);

    bf16* Cb = reinterpret_cast<bf16*>(p.C);
    bf16* Xb = reinterpret_cast<bf16*>(p.aux);
    const long row = m0 + wave * 32 + c;
    int buf = 0, s = 0;
    for (int t = 0; t < ntile; ++t) {
        f32x4 acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        static_for<NKT>([&](auto ktc) {
            constexpr int kt = decltype(ktc)::value;
            // k-tile s must have landed.  Requested after it: the k-tiles s + 1 .. s + NBUF - 2 (when they exist).  The previous tile's stores are
            // NOT added to the allowance although they were issued after k-tile s: loads retire in order among themselves, but stores retire out of
            // order with respect to loads, so "at most 2 L + EPI_OPS outstanding" can hold with k-tile s still in flight (measured: wrong tiles,
            // profiles/r06_gemm_astat_probe.txt).  The price: a wave waits for its stores at the first k-tiles of the next column tile.
            const int ahead = min(total - 1 - s, NBUF - 2);
            if (ahead >= 2) wait_vmcnt<2 * L>();
            else if (ahead == 1) wait_vmcnt<L>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();  // k-tile s landed for every wave; every wave is done with k-tile s - 1
            asm volatile("" ::: "memory");
            if (s + NBUF - 1 < total) issue(s + NBUF - 1, (buf == 0) ? NBUF - 1 : buf - 1);
            const bf16* b_lds = reinterpret_cast<const bf16*>(sB + buf * B_BYTES);
#pragma unroll
            for (int kk = 0; kk < BKD / 32; ++kk) {
                Frag<bf16> bfr[FN];
#pragma unroll
                for (int j = 0; j < FN; ++j) bfr[j] = TB::frag(b_lds, j * 16, kk, c, g);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) mma(bfr[j], af[i][2 * kt + kk], acc[i][j]);  // operands swapped: see epilogue_direct
            }
            buf = (buf + 1 == NBUF) ? 0 : buf + 1;
            ++s;
        });
        // epilogue: acc[i][j][r] = C[row 16 i + c][column 16 j + 4 g + r]; stores only (EPI_OPS of them per lane)
        const int n0 = (tile0 + t) * BN;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            f32x4 v[FN];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                v[j] = acc[i][j];
                if (p.bias) v[j] += *reinterpret_cast<const f32x4*>(sBias + t * BN + 16 * j + 4 * g);
            }
            if constexpr (GELU) {
                if constexpr (PREACT) store_row_bf16<FN>(Xb + (row + 16 * i) * p.ldaux + n0, v, g);
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[j][e] = gelu_f(v[j][e]);
            }
            store_row_bf16<FN>(Cb + (row + 16 * i) * p.ldc + n0, v, g);
        }
    }
}

template <int KSTEPS, bool GELU, bool PREACT>
int launch_gemm_astat(const esvit_gemm_desc& d, hipStream_t stream) {
    constexpr int NBUF = 4;
    using TB = DmaTile<false, 128, 64, 256>;
    const int tiles_m = d.M / 128, tiles_n = d.N / 128;
    // column chunks: enough workgroups for >= 4 rounds of the chip's 512 slots when the problem allows, chunks | 8 or 8 | chunks, and at
    // most 16 column tiles per workgroup (their bias slice sits in LDS)
    int chunks = 1;
    while ((long)tiles_m * chunks < 2048 && chunks < tiles_n) chunks *= 2;
    while (ceil_div(tiles_n, chunks) > 16) chunks *= 2;
    if (chunks > tiles_n) chunks = tiles_n;
    const int tpw = ceil_div(tiles_n, chunks);
    chunks = ceil_div(tiles_n, tpw);
    const size_t lds = (size_t)NBUF * TB::ELEMS * 2 + (size_t)tpw * 128 * sizeof(float);
    auto kern = gemm_astat_kernel<KSTEPS, GELU, PREACT, NBUF>;
    static unsigned long long lds_set = 0;
    esvit_raise_lds(kern, (int)lds, lds_set);
    hipLaunchKernelGGL(kern, dim3(tiles_m * chunks), dim3(256), lds, stream, d, tpw, chunks);
    ESVIT_CHECK_LAUNCH("esvit_gemm(a-stationary)");
    return ESVIT_OK;
}

// sum split-K partials: out[i] (+)= sum_z part[z*n + i]   (TO = float or the activation dtype).
// 256 threads = 64 element quads x 4 split slices (slice sl sums z = sl, sl+4, ...), four loads in flight per thread,
// slices combined through LDS: the 512-way reductions of the 96x96 stage-0 weights were latency-bound at one load in
// flight and 9 workgroups.
constexpr int SKR_QUADS = 64, SKR_SLICES = 4;
template <typename TO>
__device__ __forceinline__ void splitk_reduce_block(const float* __restrict__ part, int splits, long n, TO* __restrict__ out, int accumulate,
                                                    long block, f32x4 (&sm)[SKR_SLICES][SKR_QUADS]) {
    const int q = threadIdx.x & (SKR_QUADS - 1), sl = threadIdx.x / SKR_QUADS;
    const long i4 = (block * SKR_QUADS + q) * 4;
    const int cnt = i4 < n ? (int)min(4L, n - i4) : 0;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (cnt == 4) {
        const float* p = part + i4;
        int z = sl;
        for (; z + 3 * SKR_SLICES < splits; z += 4 * SKR_SLICES) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p + (long)z * n);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + (long)(z + SKR_SLICES) * n);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + (long)(z + 2 * SKR_SLICES) * n);
            const f32x4 v3 = *reinterpret_cast<const f32x4*>(p + (long)(z + 3 * SKR_SLICES) * n);
            s += (v0 + v1) + (v2 + v3);
        }
        for (; z < splits; z += SKR_SLICES) s += *reinterpret_cast<const f32x4*>(p + (long)z * n);
    } else if (cnt > 0) {
        for (int z = sl; z < splits; z += SKR_SLICES)
            for (int e = 0; e < cnt; ++e) s[e] += part[(long)z * n + i4 + e];
    }
    sm[sl][q] = s;
    __syncthreads();
    if (sl != 0 || cnt == 0) return;
    s = (sm[0][q] + sm[1][q]) + (sm[2][q] + sm[3][q]);
    for (int e = 0; e < cnt; ++e) {
        float v = s[e];
        if (accumulate) v += to_f32(out[i4 + e]);
        out[i4 + e] = from_f32<TO>(v);
    }
}

// blocks [0, blocks1): the split-K partial slabs of C;  blocks [blocks1, ..): the fused bias-gradient partials (part2, n2 -> out2)
template <typename TO>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, long n, TO* __restrict__ out,
                                                            int accumulate, int blocks1, const float* __restrict__ part2, long n2,
                                                            float* __restrict__ out2) {
    __shared__ f32x4 sm[SKR_SLICES][SKR_QUADS];
    if ((int)blockIdx.x < blocks1) splitk_reduce_block<TO>(part, splits, n, out, accumulate, blockIdx.x, sm);
    else splitk_reduce_block<float>(part2, splits, n2, out2, 0, (long)blockIdx.x - blocks1, sm);
}


// ---- launchers ----
inline int launch_splitk_reduce(const esvit_gemm_desc& d, bool act_out_bf16, hipStream_t stream) {
    const long n = (long)d.M * d.N;
    const int blocks = ceil_div(ceil_div(n, 4), SKR_QUADS);
    const int blocks2 = d.colsum ? ceil_div(ceil_div((long)d.M, 4), SKR_QUADS) : 0;  // bias-gradient partials ride along
    if (d.out_f32 || !act_out_bf16)
        hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(blocks + blocks2), dim3(256), 0, stream, d.partial, d.splitk, n,
                           reinterpret_cast<float*>(d.C), d.accumulate, blocks, d.colsum_partial, (long)d.M, d.colsum);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<bf16>, dim3(blocks + blocks2), dim3(256), 0, stream, d.partial, d.splitk, n,
                           reinterpret_cast<bf16*>(d.C), d.accumulate, blocks, d.colsum_partial, (long)d.M, d.colsum);
    ESVIT_CHECK_LAUNCH("esvit_gemm(splitk_reduce)");
    return ESVIT_OK;
}

template <typename T, bool AKS, bool BKS, int BM, int BN, bool USE_TR>
int launch_gemm(const esvit_gemm_desc& d, hipStream_t stream) {
    using TA = Tile<T, AKS, BM, USE_TR>;
    using TB = Tile<T, BKS, BN, USE_TR>;
    size_t lds = 2 * (size_t)(TA::ELEMS + TB::ELEMS) * sizeof(T);
    const size_t stage_bytes = 4 * 32 * (size_t)(BN / 2 + 4) * sizeof(float);
    if (lds < stage_bytes) lds = stage_bytes;
    auto kern = gemm_kernel<T, AKS, BKS, BM, BN, USE_TR>;
    static unsigned long long lds_set = 0;  // one-time raise of the dynamic LDS cap (per instantiation and device; idempotent)
    esvit_raise_lds(kern, (int)lds, lds_set);
    const int tiles = ceil_div(d.M, BM) * ceil_div(d.N, BN);
    const int nz = d.splitk > 1 ? d.splitk : d.batch;
    hipLaunchKernelGGL(kern, dim3(tiles, nz), dim3(NTHREADS), lds, stream, d);
    ESVIT_CHECK_LAUNCH("esvit_gemm");
    if (d.splitk > 1) return launch_splitk_reduce(d, sizeof(T) == 2, stream);
    return ESVIT_OK;
}

template <bool AKS, bool BKS, int BM, int BN, int BKD, int NBUF, int WM, int WN, int MINB = 2, bool EARLY = false>
int launch_gemm_dma(const esvit_gemm_desc& d, hipStream_t stream) {
    constexpr int NT = 64 * WM * WN;
    using TA = DmaTile<AKS, BM, BKD, NT>;
    using TB = DmaTile<BKS, BN, BKD, NT>;
    size_t lds = (size_t)NBUF * (TA::ELEMS + TB::ELEMS) * 2;
    const size_t stage_bytes = (size_t)WM * WN * 16 * (size_t)(BN / WN + 4) * sizeof(float);  // epilogue staging, one region per wave
    if (lds < stage_bytes) lds = stage_bytes;
    auto kern = gemm_dma_kernel<AKS, BKS, BM, BN, BKD, NBUF, WM, WN, MINB, EARLY>;
    static unsigned long long lds_set = 0;
    esvit_raise_lds(kern, (int)lds, lds_set);
    const int tiles = ceil_div(d.M, BM) * ceil_div(d.N, BN);
    const int nz = d.splitk > 1 ? d.splitk : d.batch;
    // Work order (measured on the step's shapes, profiles/r02_gemm_workorder_ab.txt):
    //  * split-K: z-major work list -- each K slice of A and B is fetched by one XCD (weight gradients 8.3 -> 7.3 ms/step)
    //  * 12..64 column tiles: walk 16 row blocks per column block (+5..13 %, profiles/r01_gemm_group_m_ab.txt)
    //  * > 64 column tiles (the 65536-wide last layer): pairs of row blocks share each B column tile (+14 %)
    //  * 2..11 column tiles on very tall problems (>= 2048 row blocks, stages 0/1): 16 / 32 row blocks (+2..22 %)
    const int tm_ = ceil_div(d.M, BM), tn_ = ceil_div(d.N, BN);
    int group_m = 1;
    if (d.splitk > 1) group_m = -1;
    else if (nz == 1) {
        if (tn_ > 64) group_m = 2;
        else if (tn_ >= 12) group_m = 16;
        else if (tn_ >= 2 && tm_ >= 2048) group_m = tn_ <= 3 ? 16 : 32;
    }
    hipLaunchKernelGGL(kern, dim3(tiles, nz), dim3(NT), lds, stream, d, group_m);
    ESVIT_CHECK_LAUNCH("esvit_gemm(dma)");
    if (d.splitk > 1) return launch_splitk_reduce(d, true, stream);
    return ESVIT_OK;
}

}  // namespace
