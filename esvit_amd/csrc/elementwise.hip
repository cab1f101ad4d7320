// Data-movement and small reduction kernels of the hot path (all HBM-bound, 16-byte accesses).
#include "common.h"
#include "../../include/esvit_hip.h"

namespace {

// ---- gather + scale + cast rows (fp32 -> activation dtype) ------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gather_cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long rows,
                                                          int C, const int* __restrict__ rowmap, int period, int tokens,
                                                          const float* __restrict__ rowscale, int rows_per_sample) {
    const int C4 = C / 4;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C4) return;
    const long r = i / C4;
    const int c4 = (int)(i % C4);
    long sr = r;
    bool valid = true;
    if (rowmap) {
        const int t = rowmap[r % period];
        valid = t >= 0;
        sr = (r / period) * (long)tokens + t;
    }
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
        v = *reinterpret_cast<const f32x4*>(src + sr * C + c4 * 4);
        if (rowscale) v *= rowscale[sr / rows_per_sample];
    }
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<f32x4*>(dst + r * C + c4 * 4) = v;
    } else {
        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(dst + r * C + c4 * 4) = o;
    }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 4 <= n) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[i + e] = from_f32<TD>(to_f32(src[i + e]));
    } else {
        for (long j = i; j < n; ++j) dst[j] = from_f32<TD>(to_f32(src[j]));
    }
}

// ---- column sums (bias gradients, centre partials) ---------------------------------------------
// stage 1: block = 32 column-vectors (8 columns each, one 16-byte load) x 8 row lanes over COLSUM_ROWS_PER_BLOCK rows
constexpr int COLSUM_ROWS_PER_BLOCK = 256;
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, long rows, int N, long ld, float* __restrict__ ws, long rows_per_block,
                                                     int direct_accumulate) {
    // direct_accumulate >= 0: ONE row block (gridDim.y == 1) whose sums are the result: ws is the output, += when the flag is 1
    constexpr int V = Vec16<T>::N;
    __shared__ float sm[8][32 * V + 1];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n0 = (blockIdx.x * 32 + tx) * V;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    if (n0 < N) {
        if (n0 + V <= N && (ld % V) == 0) {
            for (long r = r0 + ty; r < r1; r += 8) {
                const Vec16<T> v = ld16<T>(x + r * ld + n0);
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += v.get(e);
            }
        } else {
            for (long r = r0 + ty; r < r1; r += 8)
                for (int e = 0; e < V && n0 + e < N; ++e) acc[e] += to_f32(x[r * ld + n0 + e]);
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) sm[ty][tx * V + e] = acc[e];
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * V; i += 256) {
        const int n = blockIdx.x * 32 * V + i;
        if (n < N) {
            float s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s2 += sm[k][i];
            if (direct_accumulate == 1) s2 += ws[n];
            ws[(long)blockIdx.y * N + n] = s2;
        }
    }
}

// stage 2 (shared): block = 32 columns x SLICES row slices.  The grid is a few dozen workgroups (ncols / 32), so the kernel is bound by the
// latency of its dependent loads: 32 slices (1024 threads) where there are hundreds of partial rows -- the folds of the LayerNorm backward sit
// on the backward chain, where every microsecond of a small launch is step time (profiles/r06_finish_offchain_ab.txt) --, 8 otherwise
template <int SLICES>
__global__ __launch_bounds__(32 * SLICES) void partial_reduce_kernel(const float* __restrict__ ws, int nblk, int ncols, long ld,
                                                                      float* __restrict__ out, int accumulate) {
    __shared__ float sm[SLICES][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    float s = 0.f;
    if (c < ncols) {
        // 4 independent partial sums: keep several loads in flight
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int b = ty;
        for (; b + 3 * SLICES < nblk; b += 4 * SLICES) {
            s0 += ws[(long)b * ld + c];
            s1 += ws[(long)(b + SLICES) * ld + c];
            s2 += ws[(long)(b + 2 * SLICES) * ld + c];
            s3 += ws[(long)(b + 3 * SLICES) * ld + c];
        }
        for (; b < nblk; b += SLICES) s0 += ws[(long)b * ld + c];
        s = (s0 + s1) + (s2 + s3);
    }
    sm[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < ncols) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < SLICES; ++k) t += sm[k][tx];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// ---- PatchEmbed im2col -------------------------------------------------------------------------
// cols[(b*G + i)*G + j][c*P*P + ph*P + pw] = img[b][c][i*P+ph][j*P+pw]  (weight.view(E, 3*P*P) order)
template <typename T>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, T* __restrict__ cols, int nB, int S, int P,
                                                     int Kpad) {
    const int G = S / P;
    const long total = (long)nB * G * G * (Kpad / 4);
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int k4 = (int)(i % (Kpad / 4));
    const long row = i / (Kpad / 4);
    const int j = (int)(row % G), ii = (int)((row / G) % G);
    const long b = row / ((long)G * G);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int k = k4 * 4;
    if (k < 3 * P * P) {
        const int c = k / (P * P), ph = (k / P) % P, pw = k % P;  // P % 4 == 0 so 4 consecutive pw stay in one row
        v = *reinterpret_cast<const f32x4*>(img + ((b * 3 + c) * S + (ii * P + ph)) * (long)S + j * P + pw);
    }
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<f32x4*>(cols + row * Kpad + k) = v;
    } else {
        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(cols + row * Kpad + k) = o;
    }
}

// ---- token mean (AdaptiveAvgPool1d(1), swin_transformer.py:688) -------------------------------
template <typename T>
__global__ __launch_bounds__(256) void token_mean_fwd_kernel(const float* __restrict__ x, int nB, int Tk, int C,
                                                             float* __restrict__ out, T* __restrict__ out_act) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)nB * C) return;
    const long b = i / C;
    const int c = (int)(i % C);
    float s = 0.f;
    for (int t = 0; t < Tk; ++t) s += x[(b * Tk + t) * C + c];
    s /= Tk;
    out[i] = s;
    if (out_act) out_act[i] = from_f32<T>(s);
}
__global__ __launch_bounds__(256) void token_mean_bwd_kernel(const float* __restrict__ g_mean, const float* __restrict__ g_tok,
                                                             int nB, int Tk, int C, float* __restrict__ dx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)nB * Tk * C) return;
    const int c = (int)(i % C);
    const long b = i / ((long)Tk * C);
    float v = g_mean[b * C + c] / Tk;
    if (g_tok) v += g_tok[i];
    dx[i] = v;
}

// ---- misc --------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sum_f32_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    __shared__ float scratch[16];
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) s += x[i];
    s = block_sum<1024>(s, scratch);
    if (threadIdx.x == 0) out[0] = s;
}

template <typename T>
__global__ __launch_bounds__(256) void scale_inplace_kernel(T* __restrict__ x, long n, const float* __restrict__ scale) {
    const float s = scale[0];
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * Vec16<T>::N;
    if (i + Vec16<T>::N <= n) {
        Vec16<T> v = ld16<T>(x + i);
#pragma unroll
        for (int e = 0; e < Vec16<T>::N; ++e) v.set(e, v.get(e) * s);
        st16<T>(x + i, v);
    } else {
        for (long j = i; j < n; ++j) x[j] = from_f32<T>(to_f32(x[j]) * s);
    }
}

__global__ void center_ema_kernel(float* __restrict__ center, const float* __restrict__ colsum, float m, float inv_denom, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) center[k] = center[k] * m + colsum[k] * inv_denom * (1.f - m);
}

}  // namespace

#define STREAM(s_) hipStream_t stream = reinterpret_cast<hipStream_t>(s_)
#define BAD_DTYPE(name)                        \
    esvit_set_error(name ": bad dtype");       \
    return ESVIT_ERR_ARG

extern "C" int esvit_gather_cast(int dtype, const float* src, void* dst, int64_t rows, int C, const int32_t* rowmap,
                                 int period, int tokens, const float* rowscale, int rows_per_sample, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(src && dst && rows > 0 && C > 0 && C % 4 == 0, "esvit_gather_cast: bad args");
    if (rowmap) ESVIT_CHECK_ARG(period > 0 && tokens > 0, "esvit_gather_cast: bad rowmap geometry");
    if (rowscale) ESVIT_CHECK_ARG(rows_per_sample > 0, "esvit_gather_cast: rowscale needs rows_per_sample");
    const int grid = ceil_div(rows * (C / 4), 256);
    if (dtype == ESVIT_BF16)
        hipLaunchKernelGGL(gather_cast_kernel<bf16>, dim3(grid), dim3(256), 0, stream, src, (bf16*)dst, (long)rows, C, rowmap,
                           period, tokens, rowscale, rows_per_sample);
    else if (dtype == ESVIT_F32)
        hipLaunchKernelGGL(gather_cast_kernel<float>, dim3(grid), dim3(256), 0, stream, src, (float*)dst, (long)rows, C, rowmap,
                           period, tokens, rowscale, rows_per_sample);
    else {
        BAD_DTYPE("esvit_gather_cast");
    }
    ESVIT_CHECK_LAUNCH("gather_cast");
    return ESVIT_OK;
}

extern "C" int esvit_cast_f32_to(int dtype, const float* src, void* dst, int64_t n, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(src && dst && n > 0, "esvit_cast_f32_to: bad args");
    const int grid = ceil_div(ceil_div(n, 4), 256);
    if (dtype == ESVIT_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16>), dim3(grid), dim3(256), 0, stream, src, (bf16*)dst, (long)n);
    else if (dtype == ESVIT_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, stream, src, (float*)dst, (long)n);
    else {
        BAD_DTYPE("esvit_cast_f32_to");
    }
    ESVIT_CHECK_LAUNCH("cast_f32_to");
    return ESVIT_OK;
}

int esvit_partial_reduce(const float* ws, int nblk, int ncols, long ld, float* out, int accumulate, hipStream_t stream) {
    if (nblk >= 256)
        hipLaunchKernelGGL(partial_reduce_kernel<32>, dim3(ceil_div(ncols, 32)), dim3(1024), 0, stream, ws, nblk, ncols, ld, out, accumulate);
    else
        hipLaunchKernelGGL(partial_reduce_kernel<8>, dim3(ceil_div(ncols, 32)), dim3(256), 0, stream, ws, nblk, ncols, ld, out, accumulate);
    ESVIT_CHECK_LAUNCH("partial_reduce");
    return ESVIT_OK;
}

int esvit_i_colsum_blocks(long rows) { return ceil_div(rows, COLSUM_ROWS_PER_BLOCK); }  // esvit_query

extern "C" int esvit_colsum(int dtype, const void* x, int64_t rows, int N, int64_t ld, float* out, float* ws, int accumulate,
                            esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(x && out && ws && rows > 0 && N > 0 && ld >= N, "esvit_colsum: bad args");
    ESVIT_CHECK_ARG(((uintptr_t)x % 16) == 0, "esvit_colsum: x must be 16-byte aligned");
    const int nblk = ceil_div(rows, COLSUM_ROWS_PER_BLOCK);
    // few rows (the per-wave partial rows of the attention backward's pad-slot gradients, <= 4 row blocks): one pass straight into `out`
    const bool direct = nblk <= 4;
    const long rpb = direct ? rows : COLSUM_ROWS_PER_BLOCK;
    float* dst = direct ? out : ws;
    const int dflag = direct ? (accumulate ? 1 : 0) : -1;
    if (dtype == ESVIT_BF16) {
        dim3 grid(ceil_div(N, 32 * 8), direct ? 1 : nblk);
        hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(256), 0, stream, (const bf16*)x, (long)rows, N, (long)ld, dst, rpb, dflag);
    } else if (dtype == ESVIT_F32) {
        dim3 grid(ceil_div(N, 32 * 4), direct ? 1 : nblk);
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, (long)rows, N, (long)ld, dst, rpb, dflag);
    } else {
        BAD_DTYPE("esvit_colsum");
    }
    if (direct) {
        ESVIT_CHECK_LAUNCH("colsum");
        return ESVIT_OK;
    }
    ESVIT_CHECK_LAUNCH("colsum");
    return esvit_partial_reduce(ws, nblk, N, N, out, accumulate, stream);
}

extern "C" int esvit_patch_im2col(int dtype, const float* img, void* cols, int nB, int S, int P, int Kpad, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(img && cols && nB > 0 && S > 0 && P > 0 && P % 4 == 0 && S % P == 0 && Kpad % 8 == 0 && Kpad >= 3 * P * P,
                    "esvit_patch_im2col: bad args (S=%d P=%d Kpad=%d)", S, P, Kpad);
    const int G = S / P;
    const long total = (long)nB * G * G * (Kpad / 4);
    const int grid = ceil_div(total, 256);
    if (dtype == ESVIT_BF16) hipLaunchKernelGGL(im2col_kernel<bf16>, dim3(grid), dim3(256), 0, stream, img, (bf16*)cols, nB, S, P, Kpad);
    else if (dtype == ESVIT_F32) hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid), dim3(256), 0, stream, img, (float*)cols, nB, S, P, Kpad);
    else {
        BAD_DTYPE("esvit_patch_im2col");
    }
    ESVIT_CHECK_LAUNCH("patch_im2col");
    return ESVIT_OK;
}

extern "C" int esvit_token_mean_fwd(int dtype, const float* x, int nB, int T, int C, float* out, void* out_act, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(x && out && nB > 0 && T > 0 && C > 0, "esvit_token_mean_fwd: bad args");
    const int grid = ceil_div((long)nB * C, 256);
    if (dtype == ESVIT_BF16) hipLaunchKernelGGL(token_mean_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, stream, x, nB, T, C, out, (bf16*)out_act);
    else if (dtype == ESVIT_F32) hipLaunchKernelGGL(token_mean_fwd_kernel<float>, dim3(grid), dim3(256), 0, stream, x, nB, T, C, out, (float*)out_act);
    else {
        BAD_DTYPE("esvit_token_mean_fwd");
    }
    ESVIT_CHECK_LAUNCH("token_mean_fwd");
    return ESVIT_OK;
}

extern "C" int esvit_token_mean_bwd(const float* g_mean, const float* g_tok, int nB, int T, int C, float* dx, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(g_mean && dx && nB > 0 && T > 0 && C > 0, "esvit_token_mean_bwd: bad args");
    const int grid = ceil_div((long)nB * T * C, 256);
    hipLaunchKernelGGL(token_mean_bwd_kernel, dim3(grid), dim3(256), 0, stream, g_mean, g_tok, nB, T, C, dx);
    ESVIT_CHECK_LAUNCH("token_mean_bwd");
    return ESVIT_OK;
}

extern "C" int esvit_sum_f32(const float* x, int64_t n, float* out, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(x && out && n > 0, "esvit_sum_f32: bad args");
    hipLaunchKernelGGL(sum_f32_kernel, dim3(1), dim3(1024), 0, stream, x, (long)n, out);
    ESVIT_CHECK_LAUNCH("sum_f32");
    return ESVIT_OK;
}

extern "C" int esvit_scale_inplace(int dtype, void* x, int64_t n, const float* scale, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(x && scale && n > 0, "esvit_scale_inplace: bad args");
    if (dtype == ESVIT_BF16)
        hipLaunchKernelGGL(scale_inplace_kernel<bf16>, dim3(ceil_div(ceil_div(n, 8), 256)), dim3(256), 0, stream, (bf16*)x, (long)n, scale);
    else if (dtype == ESVIT_F32)
        hipLaunchKernelGGL(scale_inplace_kernel<float>, dim3(ceil_div(ceil_div(n, 4), 256)), dim3(256), 0, stream, (float*)x, (long)n, scale);
    else {
        BAD_DTYPE("esvit_scale_inplace");
    }
    ESVIT_CHECK_LAUNCH("scale_inplace");
    return ESVIT_OK;
}

extern "C" int esvit_center_ema(float* center, const float* colsum, float momentum, float denom, int K, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(center && colsum && K > 0 && denom > 0.f, "esvit_center_ema: bad args");
    hipLaunchKernelGGL(center_ema_kernel, dim3(ceil_div(K, 256)), dim3(256), 0, stream, center, colsum, momentum, 1.f / denom, K);
    ESVIT_CHECK_LAUNCH("center_ema");
    return ESVIT_OK;
}

