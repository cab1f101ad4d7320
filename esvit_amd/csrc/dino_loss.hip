// Fused DINOLoss / DDINOLoss kernels (main_esvit.py:603-770).
//
// The reference materialises softmax(teacher), log_softmax(student) and the gathered teacher
// rows for each of its 18 (teacher view, student crop) pairs.  Here every student logit row is
// visited by ONE workgroup that (pass 1) computes its log-sum-exp and (pass 2) streams the row
// again together with the <= 2 teacher rows it is scored against, producing the row's loss and
// its complete gradient in one go:
//     loss_r = w_r * sum_terms ( lse(z) - sum_k p_term[k] z[k] ),          z = s / tau_s
//     ds[k]  = w_r / tau_s * ( n_terms * softmax(z)[k] - sum_terms p_term[k] )
// Teacher probabilities are rebuilt on the fly from per-row (max, lse) statistics, so the
// sharpened/centred teacher distribution is never written to memory.  HBM-bound: algorithmic
// bytes = read s twice (second pass from L2), read the matched teacher rows, write ds.
#include "common.h"
#include "../../include/esvit_hip.h"

namespace {

constexpr int NT = 256;

// online (max, sum exp) accumulation
struct MS {
    float m, s;
};
__device__ __forceinline__ MS ms_merge(MS a, MS b) {
    const float m = fmaxf(a.m, b.m);
    MS r;
    r.m = m;
    r.s = a.s * __expf(a.m - m) + b.s * __expf(b.m - m);
    return r;
}
__device__ __forceinline__ MS block_ms(MS a, float* sm /* >= 2*NT/64 floats */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        MS b;
        b.m = __shfl_xor(a.m, o, 64);
        b.s = __shfl_xor(a.s, o, 64);
        a = ms_merge(a, b);
    }
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) {
        sm[2 * w] = a.m;
        sm[2 * w + 1] = a.s;
    }
    __syncthreads();
    MS r{sm[0], sm[1]};
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) r = ms_merge(r, MS{sm[2 * i], sm[2 * i + 1]});
    return r;
}

// row stats of (t - c) * inv_temp
template <typename T>
__global__ __launch_bounds__(NT) void teacher_stats_kernel(const T* __restrict__ t, const float* __restrict__ center,
                                                           float inv_temp, int K, float* __restrict__ row_max,
                                                           float* __restrict__ row_lse) {
    __shared__ float sm[2 * NT / 64];
    constexpr int V = Vec16<T>::N;
    const long r = blockIdx.x;
    const T* row = t + r * K;
    MS a{-3.0e38f, 0.f};
    for (int k = threadIdx.x * V; k < K; k += NT * V) {
        const Vec16<T> x = ld16<T>(row + k);
        float vals[V];
        float mx = -3.0e38f;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            vals[e] = (x.get(e) - center[k + e]) * inv_temp;
            mx = fmaxf(mx, vals[e]);
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < V; ++e) s += __expf(vals[e] - mx);
        a = ms_merge(a, MS{mx, s});
    }
    a = block_ms(a, sm);
    if (threadIdx.x == 0) {
        row_max[r] = a.m;
        row_lse[r] = __logf(a.s);
    }
}

// Fold the per-64-column (max, sum 2^(z - max)) pairs the last-layer GEMM left behind (esvit_gemm_desc::rowstat, base 2) into the
// natural-log statistics of teacher_stats_kernel: one wave per row, row_max = M / log2(e), row_lse = ln(sum).
__global__ __launch_bounds__(NT) void rowstat_combine_kernel(const float* __restrict__ st, long R, int nb, float* __restrict__ row_max,
                                                             float* __restrict__ row_lse) {
    const long r = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (r >= R) return;
    const int l = threadIdx.x & 63;
    const f32x2* p = reinterpret_cast<const f32x2*>(st) + r * nb;
    float m = -3.0e38f, sum = 0.f;
    for (int j = l; j < nb; j += 64) {
        const f32x2 v = p[j];
        const float mm = fmaxf(m, v[0]);
        sum = sum * __builtin_amdgcn_exp2f(m - mm) + v[1] * __builtin_amdgcn_exp2f(v[0] - mm);
        m = mm;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(sum, o, 64);
        const float mm = fmaxf(m, m2);
        sum = sum * __builtin_amdgcn_exp2f(m - mm) + s2 * __builtin_amdgcn_exp2f(m2 - mm);
        m = mm;
    }
    if (l == 0) {
        row_max[r] = m * 0.6931471805599453f;
        row_lse[r] = __logf(sum);
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void dino_ce_kernel(const T* __restrict__ s, const T* __restrict__ t,
                                                     const float* __restrict__ center, const float* __restrict__ t_row_max,
                                                     const float* __restrict__ t_row_lse, const int* __restrict__ tmatch,
                                                     const float* __restrict__ row_w, float inv_st, float inv_tt, int K,
                                                     float* __restrict__ row_loss, T* __restrict__ ds, const int* __restrict__ row_order,
                                                     const float* __restrict__ s_row_max, const float* __restrict__ s_row_lse) {
    __shared__ float sm[2 * NT / 64];
    __shared__ float sm2[NT / 64];
    constexpr int V = Vec16<T>::N;
    // work order: consecutive workgroups of an XCD take consecutive entries of row_order (image-major for the region loss), so
    // the teacher rows an image's student rows are scored against are re-read from that XCD's L2 / the Infinity Cache
    const int u = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const long r = row_order ? row_order[u] : u;
    const T* srow = s + r * K;
    T* drow = ds + r * K;
    const int t0 = tmatch[2 * r], t1 = tmatch[2 * r + 1];
    const float w = row_w[r];
    const int nterms = (t0 >= 0) + (t1 >= 0);

    // The kernel is bound by VALU issue (four exponentials per logit at quarter rate), so every exponential is ONE fused
    // multiply-add feeding v_exp_f32 (base 2): exp(x * inv_st - m) = exp2(x * A - m * A), A = inv_st * log2(e), and the constants of
    // the teacher terms are folded per column / per row.
    constexpr float LOG2E = 1.4426950408889634f;
    const float A = inv_st * LOG2E;

    // pass 1: log-sum-exp of z = s * inv_st (the maximum is taken on the raw logits: inv_st > 0) -- unless the GEMM that wrote the
    // logits already produced it (s_row_max + s_row_lse, esvit_rowstat_combine)
    float lse;
    if (s_row_max) {
        lse = s_row_max[r] + s_row_lse[r];
    } else {
        MS a{-3.0e38f, 0.f};
        for (int k = threadIdx.x * V; k < K; k += NT * V) {
            const Vec16<T> x = ld16<T>(srow + k);
            float mr = -3.0e38f;
#pragma unroll
            for (int e = 0; e < V; ++e) mr = fmaxf(mr, x.get(e));
            const float nb = -mr * A;
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < V; ++e) sum += __builtin_amdgcn_exp2f(fmaf(x.get(e), A, nb));
            a = ms_merge(a, MS{mr * inv_st, sum});
        }
        a = block_ms(a, sm);
        lse = a.m + __logf(a.s);
    }

    // pass 2: gradient + sum_k p_t[k] z[k].  An unused term gets offset +inf: exp2(-inf) = 0, no select per element.
    const T* trow0 = t + (long)(t0 >= 0 ? t0 : 0) * K;
    const T* trow1 = t + (long)(t1 >= 0 ? t1 : 0) * K;
    const float At = inv_tt * LOG2E;
    const float off0 = t0 >= 0 ? (t_row_max[t0] + t_row_lse[t0]) * LOG2E : INFINITY;
    const float off1 = t1 >= 0 ? (t_row_max[t1] + t_row_lse[t1]) * LOG2E : INFINITY;
    const float C = -lse * LOG2E;
    const float gs = w * inv_st, gsn = gs * nterms;
    float dot = 0.f;  // sum_k p_t[k] * s[k] (raw logits; scaled by inv_st once at the end)
    if (nterms == 2) {
        for (int k = threadIdx.x * V; k < K; k += NT * V) {
            const Vec16<T> x = ld16<T>(srow + k);
            const Vec16<T> y0 = ld16<T>(trow0 + k), y1 = ld16<T>(trow1 + k);
            f32x4 cv[V / 4];
#pragma unroll
            for (int q = 0; q < V / 4; ++q) cv[q] = *reinterpret_cast<const f32x4*>(center + k + 4 * q);
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float xe = x.get(e);
                const float ps = __builtin_amdgcn_exp2f(fmaf(xe, A, C));
                const float ck = cv[e / 4][e & 3];
                const float nbk = -ck * At;  // (y - c) * inv_tt * log2(e) = y * At + nbk
                const float pt = __builtin_amdgcn_exp2f(fmaf(y0.get(e), At, nbk - off0)) + __builtin_amdgcn_exp2f(fmaf(y1.get(e), At, nbk - off1));
                dot = fmaf(pt, xe, dot);
                o.set(e, fmaf(ps, gsn, -gs * pt));
            }
            st16<T>(drow + k, o);
        }
    } else {
        // one teacher term (the rows of a global crop are not scored against their own view, main_esvit.py:636-638, 719-721: 58 % of the
        // region rows): one exponential and one teacher row less per logit; no term at all: the gradient is zero
        const T* trow = t0 >= 0 ? trow0 : trow1;
        const float off = t0 >= 0 ? off0 : off1;
        for (int k = threadIdx.x * V; k < K; k += NT * V) {
            const Vec16<T> x = ld16<T>(srow + k);
            Vec16<T> o;
            if (nterms == 1) {
                const Vec16<T> y = ld16<T>(trow + k);
                f32x4 cv[V / 4];
#pragma unroll
                for (int q = 0; q < V / 4; ++q) cv[q] = *reinterpret_cast<const f32x4*>(center + k + 4 * q);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float xe = x.get(e);
                    const float ps = __builtin_amdgcn_exp2f(fmaf(xe, A, C));
                    const float pt = __builtin_amdgcn_exp2f(fmaf(y.get(e), At, -cv[e / 4][e & 3] * At - off));
                    dot = fmaf(pt, xe, dot);
                    o.set(e, fmaf(ps, gsn, -gs * pt));
                }
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) o.set(e, 0.f);
            }
            st16<T>(drow + k, o);
        }
    }
    dot = block_sum<NT>(dot, sm2) * inv_st;
    if (threadIdx.x == 0) row_loss[r] = w * (nterms * lse - dot);
}

// General form for DINOLoss with mixup targets (main_esvit.py:639-641): student row r is scored against up to NTERM
// teacher rows tmatch[r*NTERM + j] (-1 = unused) with individual weights term_w[r*NTERM + j]:
//   row_loss = sum_j w_j (lse - q_j . z),   ds = inv_st ((sum_j w_j) p_s - sum_j w_j q_j)
template <typename T, int NTERM>
__global__ __launch_bounds__(NT) void dino_ce_terms_kernel(const T* __restrict__ s, const T* __restrict__ t,
                                                           const float* __restrict__ center, const float* __restrict__ t_row_max,
                                                           const float* __restrict__ t_row_lse, const int* __restrict__ tmatch,
                                                           const float* __restrict__ term_w, float inv_st, float inv_tt, int K,
                                                           float* __restrict__ row_loss, T* __restrict__ ds, const int* __restrict__ row_order) {
    __shared__ float sm[2 * NT / 64];
    __shared__ float sm2[NT / 64];
    constexpr int V = Vec16<T>::N;
    const int u = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const long r = row_order ? row_order[u] : u;
    const T* srow = s + r * K;
    T* drow = ds + r * K;
    int tj[NTERM];
    float wj[NTERM], offj[NTERM];
    const T* trow[NTERM];
    float wsum = 0.f;
#pragma unroll
    for (int j = 0; j < NTERM; ++j) {
        tj[j] = tmatch[r * NTERM + j];
        wj[j] = tj[j] >= 0 ? term_w[r * NTERM + j] : 0.f;
        wsum += wj[j];
        trow[j] = t + (long)(tj[j] >= 0 ? tj[j] : 0) * K;
        offj[j] = tj[j] >= 0 ? t_row_max[tj[j]] + t_row_lse[tj[j]] : 0.f;
    }
    MS a{-3.0e38f, 0.f};
    for (int k = threadIdx.x * V; k < K; k += NT * V) {
        const Vec16<T> x = ld16<T>(srow + k);
        float mx = -3.0e38f;
#pragma unroll
        for (int e = 0; e < V; ++e) mx = fmaxf(mx, x.get(e) * inv_st);
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < V; ++e) sum += __expf(x.get(e) * inv_st - mx);
        a = ms_merge(a, MS{mx, sum});
    }
    a = block_ms(a, sm);
    const float lse = a.m + __logf(a.s);
    float dot = 0.f;
    for (int k = threadIdx.x * V; k < K; k += NT * V) {
        const Vec16<T> x = ld16<T>(srow + k);
        Vec16<T> y[NTERM];
#pragma unroll
        for (int j = 0; j < NTERM; ++j) y[j] = tj[j] >= 0 ? ld16<T>(trow[j] + k) : zero16<T>();
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float z = x.get(e) * inv_st;
            const float ps = __expf(z - lse);
            const float ck = center[k + e];
            float pt = 0.f;
#pragma unroll
            for (int j = 0; j < NTERM; ++j)
                if (tj[j] >= 0) pt += wj[j] * __expf((y[j].get(e) - ck) * inv_tt - offj[j]);
            dot += pt * z;
            o.set(e, inv_st * (wsum * ps - pt));
        }
        st16<T>(drow + k, o);
    }
    dot = block_sum<NT>(dot, sm2);
    if (threadIdx.x == 0) row_loss[r] = wsum * lse - dot;
}

// region matching (main_esvit.py:735-738): argmax over the Tt teacher tokens of view iq + row assembly
__global__ void region_match_kernel(const float* __restrict__ sim, int B, int S, int Tt, int ld, const int* __restrict__ crop_id,
                                    const int* __restrict__ cm_row, int* __restrict__ tmatch) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * S * 2) return;
    const int iq = (int)(i & 1);
    const long bs = i >> 1;
    const int s = (int)(bs % S);
    const long b = bs / S;
    int out = -1;
    if (crop_id[s] != iq) {
        const float* p = sim + bs * ld + iq * Tt;
        float best = p[0];
        int bj = 0;
        for (int j = 1; j < Tt; ++j) {
            const float v = p[j];
            if (v > best) {
                best = v;
                bj = j;
            }
        }
        out = iq * B * Tt + (int)b * Tt + bj;
    }
    tmatch[(long)cm_row[bs] * 2 + iq] = out;
}

}  // namespace

#define STREAM(s_) hipStream_t stream = reinterpret_cast<hipStream_t>(s_)

extern "C" int esvit_teacher_row_stats(int dtype, const void* t, const float* center, float inv_temp, int64_t R, int K,
                                       float* row_max, float* row_lse, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(t && center && row_max && row_lse && R > 0 && K > 0 && K % 8 == 0, "esvit_teacher_row_stats: bad args (K=%d)", K);
    if (dtype == ESVIT_BF16)
        hipLaunchKernelGGL(teacher_stats_kernel<bf16>, dim3((unsigned)R), dim3(NT), 0, stream, (const bf16*)t, center, inv_temp, K,
                           row_max, row_lse);
    else if (dtype == ESVIT_F32)
        hipLaunchKernelGGL(teacher_stats_kernel<float>, dim3((unsigned)R), dim3(NT), 0, stream, (const float*)t, center, inv_temp, K,
                           row_max, row_lse);
    else {
        esvit_set_error("esvit_teacher_row_stats: bad dtype");
        return ESVIT_ERR_ARG;
    }
    ESVIT_CHECK_LAUNCH("teacher_row_stats");
    return ESVIT_OK;
}

extern "C" int esvit_rowstat_combine(const float* rowstat, int64_t R, int nblocks, float* row_max, float* row_lse, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(rowstat && row_max && row_lse && R > 0 && nblocks > 0, "esvit_rowstat_combine: bad args");
    hipLaunchKernelGGL(rowstat_combine_kernel, dim3((unsigned)ceil_div(R, (long)(NT / 64))), dim3(NT), 0, stream, rowstat, (long)R, nblocks,
                       row_max, row_lse);
    ESVIT_CHECK_LAUNCH("rowstat_combine");
    return ESVIT_OK;
}

extern "C" int esvit_dino_ce_fwd_bwd(int dtype, const void* s, const void* t, const float* center, const float* t_row_max,
                                     const float* t_row_lse, const int32_t* tmatch, const float* row_w, int terms, const float* term_w,
                                     float inv_student_temp, float inv_teacher_temp, int64_t Rs, int K, float* row_loss, void* ds,
                                     const int32_t* row_order, const float* s_row_max, const float* s_row_lse, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG((s_row_max == nullptr) == (s_row_lse == nullptr) && !(s_row_max && term_w),
                    "esvit_dino_ce_fwd_bwd: student row statistics come as a (max, lse) pair, for the two-term form");
    ESVIT_CHECK_ARG(s && t && center && t_row_max && t_row_lse && tmatch && row_loss && ds && Rs > 0 && K > 0 && K % 8 == 0,
                    "esvit_dino_ce_fwd_bwd: bad args (K=%d)", K);
    ESVIT_CHECK_ARG(dtype == ESVIT_BF16 || dtype == ESVIT_F32, "esvit_dino_ce_fwd_bwd: bad dtype");
    if (term_w) {  // individually weighted terms (mixup targets)
        ESVIT_CHECK_ARG(terms == 4, "esvit_dino_ce_fwd_bwd: weighted terms come four per row (got %d)", terms);
        if (dtype == ESVIT_BF16)
            hipLaunchKernelGGL((dino_ce_terms_kernel<bf16, 4>), dim3((unsigned)Rs), dim3(NT), 0, stream, (const bf16*)s, (const bf16*)t, center,
                               t_row_max, t_row_lse, tmatch, term_w, inv_student_temp, inv_teacher_temp, K, row_loss, (bf16*)ds, row_order);
        else
            hipLaunchKernelGGL((dino_ce_terms_kernel<float, 4>), dim3((unsigned)Rs), dim3(NT), 0, stream, (const float*)s, (const float*)t,
                               center, t_row_max, t_row_lse, tmatch, term_w, inv_student_temp, inv_teacher_temp, K, row_loss, (float*)ds, row_order);
        ESVIT_CHECK_LAUNCH("dino_ce_fwd_bwd(terms)");
        return ESVIT_OK;
    }
    ESVIT_CHECK_ARG(row_w && terms == 2, "esvit_dino_ce_fwd_bwd: two equally weighted terms per row need row_w");
    if (dtype == ESVIT_BF16)
        hipLaunchKernelGGL(dino_ce_kernel<bf16>, dim3((unsigned)Rs), dim3(NT), 0, stream, (const bf16*)s, (const bf16*)t, center,
                           t_row_max, t_row_lse, tmatch, row_w, inv_student_temp, inv_teacher_temp, K, row_loss, (bf16*)ds, row_order, s_row_max,
                           s_row_lse);
    else
        hipLaunchKernelGGL(dino_ce_kernel<float>, dim3((unsigned)Rs), dim3(NT), 0, stream, (const float*)s, (const float*)t, center,
                           t_row_max, t_row_lse, tmatch, row_w, inv_student_temp, inv_teacher_temp, K, row_loss, (float*)ds, row_order, s_row_max,
                           s_row_lse);
    ESVIT_CHECK_LAUNCH("dino_ce_fwd_bwd");
    return ESVIT_OK;
}

extern "C" int esvit_region_match(const float* sim, int B, int S, int Tt, int ld, const int32_t* crop_id, const int32_t* cm_row,
                                  int32_t* tmatch, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(sim && crop_id && cm_row && tmatch && B > 0 && S > 0 && Tt > 0 && ld >= 2 * Tt, "esvit_region_match: bad args");
    hipLaunchKernelGGL(region_match_kernel, dim3(ceil_div((long)B * S * 2, 128)), dim3(128), 0, stream, sim, B, S, Tt, ld, crop_id,
                       cm_row, tmatch);
    ESVIT_CHECK_LAUNCH("region_match");
    return ESVIT_OK;
}
