// Global (all-to-all) attention of the monolithic ViT backbones (models/vision_transformer.py:67-94, `Attention`; deit_tiny /
// deit_small / vit_base, 197 tokens per 224^2 crop and 37 per 96^2 crop with the class token, head_dim 64).
//
// Unlike the windowed kernels of window_attn*.hip this path keeps the score matrix in HBM and runs on the GEMM family:
//   qkv [T, 3C] --heads_split--> q | k | v  [B, nH, Np, hd]  (Np = tokens padded to a multiple of 16, pad rows zero)
//   S = q k^T                     batched esvit_gemm  [B nH, Np, Np]
//   P = softmax(scale * S)        softmax_rows_fwd (in place; pad rows / columns zero)
//   O = P v                       batched esvit_gemm  [B nH, Np, hd]  --heads_merge--> [T, C]
// and the mirror image backward (dV = P^T dO, dP = dO v^T, dS = softmax', dq = dS k, dk = dS^T q).  The kernels here are the
// layout changes and the row softmax; they are plain streaming kernels bound by HBM.
#include "common.h"
#include "esvit_hip.h"

namespace {

// token-major [B * N, P * nH * hd] -> P tensors [B, nH, Np, hd] (pad rows zeroed); one 16-byte vector per thread
template <typename T>
__global__ __launch_bounds__(256) void heads_split_kernel(const T* __restrict__ x, int B, int N, int Np, int nH, int hd, int P, T* __restrict__ y) {
    constexpr int V = ElemTraits<T>::VEC;
    const int vpr = hd / V;  // vectors per (token, head)
    const long total = (long)P * B * nH * Np * vpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int v = (int)(i % vpr);
        long r = i / vpr;
        const int t = (int)(r % Np);
        r /= Np;
        const int h = (int)(r % nH);
        r /= nH;
        const int b = (int)(r % B);
        const int p = (int)(r / B);
        Vec16<T> val = zero16<T>();
        if (t < N) val = ld16<T>(x + ((long)b * N + t) * ((long)P * nH * hd) + ((long)p * nH + h) * hd + v * V);
        st16<T>(y + i * V, val);
    }
}

// P tensors [B, nH, Np, hd] -> token-major [B * N, P * nH * hd]
template <typename T>
__global__ __launch_bounds__(256) void heads_merge_kernel(const T* __restrict__ y, int B, int N, int Np, int nH, int hd, int P, T* __restrict__ x) {
    constexpr int V = ElemTraits<T>::VEC;
    const int vpr = hd / V;
    const long total = (long)B * N * P * nH * vpr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int v = (int)(i % vpr);
        long r = i / vpr;
        const int h = (int)(r % nH);
        r /= nH;
        const int p = (int)(r % P);
        r /= P;
        const int t = (int)(r % N);
        const int b = (int)(r / N);
        st16<T>(x + i * V, ld16<T>(y + ((((long)p * B + b) * nH + h) * Np + t) * hd + v * V));
    }
}

constexpr int SM_MAX = 4;  // elements per lane: rows of up to 256 padded tokens

// Vision Longformer's sliding-chunk neighbourhood (layers/slidingchunk_2d.py:268-287, exact = 0; layers/longformer2d.py:163-262) as a
// predicate on token pairs: chunk[t] = -1 for a global token, else (chunk row << 16) | chunk column of the token's w x w chunk.  A
// query sees every global token, and the local tokens of its own and the eight adjacent chunks; a global query sees everything.
__device__ __forceinline__ bool chunk_allows(int ci, int cj) {
    if ((ci | cj) < 0) return true;
    const int dx = (ci >> 16) - (cj >> 16), dy = (ci & 0xffff) - (cj & 0xffff);
    return dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1;
}

// one wave per row of the [rows = batch * Np, Np] score matrices, in place: P = softmax(scale * S[:, :N]), zero elsewhere
// (chunk != nullptr: only over the keys chunk_allows admits)
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(T* __restrict__ s, long rows, int N, int Np, float scale, const int* __restrict__ chunk) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    T* p = s + row * Np;
    const int qi = (int)(row % Np);
    const bool live = qi < N;
    const int ci = (chunk && live) ? chunk[qi] : -1;
    float v[SM_MAX];
    bool ok[SM_MAX];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < SM_MAX; ++k) {
        const int j = lane + 64 * k;
        ok[k] = live && j < N && (!chunk || chunk_allows(ci, chunk[j]));
        v[k] = ok[k] ? scale * to_f32(p[j]) : -INFINITY;
        m = fmaxf(m, v[k]);
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < SM_MAX; ++k) {
        v[k] = ok[k] ? __expf(v[k] - m) : 0.f;
        sum += v[k];
    }
    sum = wave_sum(sum);
    const float inv = live ? 1.f / sum : 0.f;
#pragma unroll
    for (int k = 0; k < SM_MAX; ++k) {
        const int j = lane + 64 * k;
        if (j < Np) p[j] = from_f32<T>(v[k] * inv);
    }
}

// dS = scale * P o (dP - sum_j P_j dP_j), in place over dP; zero on pad rows / columns
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const T* __restrict__ prob, T* __restrict__ dp, long rows, int N, int Np, float scale) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const T* p = prob + row * Np;
    T* d = dp + row * Np;
    const bool live = (int)(row % Np) < N;
    float pv[SM_MAX], dv[SM_MAX];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < SM_MAX; ++k) {
        const int j = lane + 64 * k;
        const bool in = live && j < N;
        pv[k] = in ? to_f32(p[j]) : 0.f;
        dv[k] = in ? to_f32(d[j]) : 0.f;
        dot += pv[k] * dv[k];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int k = 0; k < SM_MAX; ++k) {
        const int j = lane + 64 * k;
        if (j < Np) d[j] = from_f32<T>(scale * pv[k] * (dv[k] - dot));
    }
}

// rows longer than 256 padded tokens (patch_size 8: 785 tokens at 224^2; evaluation on larger images; the 3137-token first stage of
// Vision Longformer): the same arithmetic in passes over the row instead of registers, 16-byte vectors per lane, max and sum in ONE
// pass (online rescaling).  With a chunk table whose local tokens are ordered chunk row by chunk row (rowtok = tokens per chunk row,
// nglo global tokens in front) a local query only scans [0, nglo) and the three chunk rows around its own -- everything else is
// written as zero without being read.
struct RowSpan {
    int a_end, b_lo, b_hi;  // element ranges [0, a_end) and [b_lo, b_hi), vector-aligned, inside [0, Np)
};
template <int V>
__device__ __forceinline__ RowSpan row_span(int N, int Np, int ci, int nglo, int rowtok) {
    RowSpan r{0, 0, Np};
    if (ci >= 0 && rowtok > 0) {
        const int cx = ci >> 16;
        int lo = nglo + max(cx - 1, 0) * rowtok, hi = min(N, nglo + (cx + 2) * rowtok);
        lo = (lo / V) * V;
        hi = min(Np, ((hi + V - 1) / V) * V);
        r.a_end = min(((nglo + V - 1) / V) * V, lo);
        r.b_lo = lo;
        r.b_hi = hi;
    }
    return r;
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_fwd_long_kernel(T* __restrict__ s, long rows, int N, int Np, float scale, const int* __restrict__ chunk,
                                                                    int nglo, int rowtok) {
    constexpr int V = Vec16<T>::N;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    T* p = s + row * Np;
    const int qi = (int)(row % Np);
    const bool live = qi < N;
    const int ci = (chunk && live) ? chunk[qi] : -1;
    const RowSpan sp = row_span<V>(N, Np, chunk ? ci : -1, nglo, rowtok);
    auto okv = [&](int j0, bool (&ok)[V]) {
#pragma unroll
        for (int e = 0; e < V; ++e) ok[e] = j0 + e < N && (!chunk || chunk_allows(ci, chunk[j0 + e]));
    };
    float m = -INFINITY, sum = 0.f;
    auto scan = [&](int lo, int hi) {
        for (int j0 = lo + lane * V; j0 < hi; j0 += 64 * V) {
            const Vec16<T> x = ld16<T>(p + j0);
            bool ok[V];
            okv(j0, ok);
            float mv = -INFINITY;
#pragma unroll
            for (int e = 0; e < V; ++e) mv = fmaxf(mv, ok[e] ? scale * x.get(e) : -INFINITY);
            if (mv > -INFINITY) {
                const float mn = fmaxf(m, mv);
                float sv = 0.f;
#pragma unroll
                for (int e = 0; e < V; ++e) sv += ok[e] ? __expf(scale * x.get(e) - mn) : 0.f;
                sum = sum * __expf(m - mn) + sv;
                m = mn;
            }
        }
    };
    if (live) {
        scan(0, sp.a_end);
        scan(sp.b_lo, sp.b_hi);
    }
    const float mw = wave_max(m);
    sum = wave_sum(m > -INFINITY ? sum * __expf(m - mw) : 0.f);
    const float inv = live ? 1.f / sum : 0.f;
    for (int j0 = lane * V; j0 < Np; j0 += 64 * V) {
        Vec16<T> o = zero16<T>();
        if (live && (j0 < sp.a_end || (j0 >= sp.b_lo && j0 < sp.b_hi))) {
            const Vec16<T> x = ld16<T>(p + j0);
            bool ok[V];
            okv(j0, ok);
#pragma unroll
            for (int e = 0; e < V; ++e) o.set(e, ok[e] ? __expf(scale * x.get(e) - mw) * inv : 0.f);
        }
        st16<T>(p + j0, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_bwd_long_kernel(const T* __restrict__ prob, T* __restrict__ dp, long rows, int N, int Np, float scale,
                                                                    const int* __restrict__ chunk, int nglo, int rowtok) {
    constexpr int V = Vec16<T>::N;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const T* p = prob + row * Np;
    T* d = dp + row * Np;
    const int qi = (int)(row % Np);
    const bool live = qi < N;
    const int ci = (chunk && live) ? chunk[qi] : -1;
    const RowSpan sp = row_span<V>(N, Np, chunk ? ci : -1, nglo, rowtok);  // (P is zero outside the span: nothing to read there)
    float dot = 0.f;
    auto scan = [&](int lo, int hi) {
        for (int j0 = lo + lane * V; j0 < hi; j0 += 64 * V) {
            const Vec16<T> x = ld16<T>(p + j0), y = ld16<T>(d + j0);
#pragma unroll
            for (int e = 0; e < V; ++e) dot += (j0 + e < N) ? x.get(e) * y.get(e) : 0.f;
        }
    };
    if (live) {
        scan(0, sp.a_end);
        scan(sp.b_lo, sp.b_hi);
    }
    dot = wave_sum(dot);
    for (int j0 = lane * V; j0 < Np; j0 += 64 * V) {
        Vec16<T> o = zero16<T>();
        if (live && (j0 < sp.a_end || (j0 >= sp.b_lo && j0 < sp.b_hi))) {
            const Vec16<T> x = ld16<T>(p + j0), y = ld16<T>(d + j0);
#pragma unroll
            for (int e = 0; e < V; ++e) o.set(e, (j0 + e < N) ? scale * x.get(e) * (y.get(e) - dot) : 0.f);
        }
        st16<T>(d + j0, o);
    }
}

int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

#define STREAM(s_) hipStream_t stream = reinterpret_cast<hipStream_t>(s_)

extern "C" int esvit_heads_split(int dtype, const void* x, int B, int N, int Np, int nH, int hd, int parts, void* y, esvit_stream_t stream_) {
    STREAM(stream_);
    ESVIT_CHECK_ARG(x && y && B > 0 && N > 0 && Np >= N && nH > 0 && parts > 0, "esvit_heads_split: bad arguments");
    ESVIT_CHECK_ARG(hd % (dtype == ESVIT_BF16 ? 8 : 4) == 0, "esvit_heads_split: head_dim %d is not a whole number of 16-byte vectors", hd);
    const long total = (long)parts * B * nH * Np * (hd / (dtype == ESVIT_BF16 ? 8 : 4));
    if (dtype == ESVIT_BF16) hipLaunchKernelGGL(heads_split_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16*)x, B, N, Np, nH, hd, parts, (bf16*)y);
    else if (dtype == ESVIT_F32) hipLaunchKernelGGL(heads_split_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, (const float*)x, B, N, Np, nH, hd, parts, (float*)y);
    else ESVIT_CHECK_ARG(false, "esvit_heads_split: bad dtype %d", dtype);
    ESVIT_CHECK_LAUNCH("heads_split");
    return ESVIT_OK;
}

extern "C" int esvit_heads_merge(int dtype, const void* y, int B, int N, int Np, int nH, int hd, int parts, void* x, esvit_stream_t stream_) {
    STREAM(stream_);
    ESVIT_CHECK_ARG(x && y && B > 0 && N > 0 && Np >= N && nH > 0 && parts > 0, "esvit_heads_merge: bad arguments");
    ESVIT_CHECK_ARG(hd % (dtype == ESVIT_BF16 ? 8 : 4) == 0, "esvit_heads_merge: head_dim %d is not a whole number of 16-byte vectors", hd);
    const long total = (long)parts * B * nH * N * (hd / (dtype == ESVIT_BF16 ? 8 : 4));
    if (dtype == ESVIT_BF16) hipLaunchKernelGGL(heads_merge_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16*)y, B, N, Np, nH, hd, parts, (bf16*)x);
    else if (dtype == ESVIT_F32) hipLaunchKernelGGL(heads_merge_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, (const float*)y, B, N, Np, nH, hd, parts, (float*)x);
    else ESVIT_CHECK_ARG(false, "esvit_heads_merge: bad dtype %d", dtype);
    ESVIT_CHECK_LAUNCH("heads_merge");
    return ESVIT_OK;
}

static int softmax_rows_fwd_impl(int dtype, void* s, int64_t batch, int N, int Np, float scale, const int32_t* chunk, int nglo, int rowtok,
                                 hipStream_t stream) {
    ESVIT_CHECK_ARG(s && batch > 0 && N > 0 && Np >= N, "esvit_softmax_rows_fwd: bad arguments");
    ESVIT_CHECK_ARG(Np <= 64 * SM_MAX || Np % 8 == 0, "esvit_softmax_rows_fwd: long rows are read as 16-byte vectors (Np = %d)", Np);
    const long rows = batch * Np;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (Np > 64 * SM_MAX) {
        if (dtype == ESVIT_BF16) hipLaunchKernelGGL(softmax_rows_fwd_long_kernel<bf16>, grid, dim3(256), 0, stream, (bf16*)s, rows, N, Np, scale, chunk, nglo, rowtok);
        else if (dtype == ESVIT_F32) hipLaunchKernelGGL(softmax_rows_fwd_long_kernel<float>, grid, dim3(256), 0, stream, (float*)s, rows, N, Np, scale, chunk, nglo, rowtok);
        else ESVIT_CHECK_ARG(false, "esvit_softmax_rows_fwd: bad dtype %d", dtype);
        ESVIT_CHECK_LAUNCH("softmax_rows_fwd");
        return ESVIT_OK;
    }
    if (dtype == ESVIT_BF16) hipLaunchKernelGGL(softmax_rows_fwd_kernel<bf16>, grid, dim3(256), 0, stream, (bf16*)s, rows, N, Np, scale, chunk);
    else if (dtype == ESVIT_F32) hipLaunchKernelGGL(softmax_rows_fwd_kernel<float>, grid, dim3(256), 0, stream, (float*)s, rows, N, Np, scale, chunk);
    else ESVIT_CHECK_ARG(false, "esvit_softmax_rows_fwd: bad dtype %d", dtype);
    ESVIT_CHECK_LAUNCH("softmax_rows_fwd");
    return ESVIT_OK;
}

extern "C" int esvit_softmax_rows_fwd(int dtype, void* s, int64_t batch, int N, int Np, float scale, esvit_stream_t stream_) {
    STREAM(stream_);
    return softmax_rows_fwd_impl(dtype, s, batch, N, Np, scale, nullptr, 0, 0, stream);
}

extern "C" int esvit_softmax_rows_chunked_fwd(int dtype, void* s, int64_t batch, int N, int Np, float scale, const int32_t* chunk, int nglo,
                                              int chunk_row_tokens, esvit_stream_t stream_) {
    STREAM(stream_);
    ESVIT_CHECK_ARG(chunk != nullptr && nglo >= 0 && chunk_row_tokens >= 0, "esvit_softmax_rows_chunked_fwd: bad chunk arguments");
    return softmax_rows_fwd_impl(dtype, s, batch, N, Np, scale, chunk, nglo, chunk_row_tokens, stream);
}

static int softmax_rows_bwd_impl(int dtype, const void* p, void* dp, int64_t batch, int N, int Np, float scale, const int32_t* chunk, int nglo,
                                 int rowtok, hipStream_t stream) {
    ESVIT_CHECK_ARG(p && dp && batch > 0 && N > 0 && Np >= N, "esvit_softmax_rows_bwd: bad arguments");
    ESVIT_CHECK_ARG(Np <= 64 * SM_MAX || Np % 8 == 0, "esvit_softmax_rows_bwd: long rows are read as 16-byte vectors (Np = %d)", Np);
    const long rows = batch * Np;
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (Np > 64 * SM_MAX) {
        if (dtype == ESVIT_BF16)
            hipLaunchKernelGGL(softmax_rows_bwd_long_kernel<bf16>, grid, dim3(256), 0, stream, (const bf16*)p, (bf16*)dp, rows, N, Np, scale, chunk, nglo, rowtok);
        else if (dtype == ESVIT_F32)
            hipLaunchKernelGGL(softmax_rows_bwd_long_kernel<float>, grid, dim3(256), 0, stream, (const float*)p, (float*)dp, rows, N, Np, scale, chunk, nglo, rowtok);
        else ESVIT_CHECK_ARG(false, "esvit_softmax_rows_bwd: bad dtype %d", dtype);
        ESVIT_CHECK_LAUNCH("softmax_rows_bwd");
        return ESVIT_OK;
    }
    if (dtype == ESVIT_BF16) hipLaunchKernelGGL(softmax_rows_bwd_kernel<bf16>, grid, dim3(256), 0, stream, (const bf16*)p, (bf16*)dp, rows, N, Np, scale);
    else if (dtype == ESVIT_F32) hipLaunchKernelGGL(softmax_rows_bwd_kernel<float>, grid, dim3(256), 0, stream, (const float*)p, (float*)dp, rows, N, Np, scale);
    else ESVIT_CHECK_ARG(false, "esvit_softmax_rows_bwd: bad dtype %d", dtype);
    ESVIT_CHECK_LAUNCH("softmax_rows_bwd");
    return ESVIT_OK;
}

extern "C" int esvit_softmax_rows_bwd(int dtype, const void* p, void* dp, int64_t batch, int N, int Np, float scale, esvit_stream_t stream_) {
    STREAM(stream_);
    return softmax_rows_bwd_impl(dtype, p, dp, batch, N, Np, scale, nullptr, 0, 0, stream);
}

extern "C" int esvit_softmax_rows_chunked_bwd(int dtype, const void* p, void* dp, int64_t batch, int N, int Np, float scale, const int32_t* chunk,
                                              int nglo, int chunk_row_tokens, esvit_stream_t stream_) {
    STREAM(stream_);
    ESVIT_CHECK_ARG(chunk != nullptr && nglo >= 0 && chunk_row_tokens >= 0, "esvit_softmax_rows_chunked_bwd: bad chunk arguments");
    return softmax_rows_bwd_impl(dtype, p, dp, batch, N, Np, scale, chunk, nglo, chunk_row_tokens, stream);
}
