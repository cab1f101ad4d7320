// Convolutional pieces of the CvT backbone (BASELINE config 5; cvt_v4_transformer.py): ConvEmbed as im2col + the MFMA
// GEMM, the depthwise 3x3 convolution and the train-mode BatchNorm of the qkv projection.  Activations are token-major
// NHWC ([nB, H, W, C] == [nB*H*W, C]) like everywhere else in this library, so the reference's NCHW <-> NHWC rearranges
// around every LayerNorm (cvt_v4_transformer.py:55-58) never happen.  All kernels here are HBM-bound streaming kernels.
//
//   esvit_conv_im2col / esvit_conv_col2im   ConvEmbed.proj (cvt:363-368) forward gather and its adjoint
//   esvit_dwconv3x3 / esvit_dwconv3x3_wgrad DepthWiseConv2d.dw (cvt:87-94), data gradient = same kernel with flipped taps
//   esvit_col_sums2                         per-channel  sum(a), sum(a*b)  over rows: BatchNorm batch statistics (b = a) and
//                                           the two reductions of its backward (a = dy, b = pre-norm activations)
//   esvit_col_affine2                       y = a1[c]*x1 + a2[c]*x2 + a3[c]: BatchNorm apply and BatchNorm backward apply
#include "common.h"
#include "../../include/esvit_hip.h"

namespace {

// General form: one element per thread (any Cin; the fp32 parity mode of the NCHW stem; odd channel counts).
template <typename T>
__global__ void im2col_kernel(const void* __restrict__ src_, int nchw, int nB, int H, int W, int Cin, int k, int stride, int pad,
                              int Ho, int Wo, int Kpad, T* __restrict__ cols) {
    const long total = (long)nB * Ho * Wo * Kpad;
    const int KK = k * k * Cin;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i % Kpad);
        const long r = i / Kpad;
        float v = 0.f;
        if (j < KK) {
            const int c = j % Cin, kk = j / Cin;
            const int ky = kk / k, kx = kk % k;
            const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho);
            const long b = r / ((long)Wo * Ho);
            const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                if (nchw) v = reinterpret_cast<const float*>(src_)[((b * Cin + c) * H + iy) * W + ix];
                else v = to_f32(reinterpret_cast<const T*>(src_)[((b * H + iy) * W + ix) * Cin + c]);
            }
        }
        cols[i] = from_f32<T>(v);
    }
}

// Token sources (NHWC) whose channel count is a whole number of 16-byte vectors: a thread copies one vector of one tap -- the
// columns of a tap are Cin contiguous channels of one source token, so loads and stores are both 16-byte and coalesced over c.
// (The element-wise form above spends two 64-bit divisions and a 2-byte store per element: 0.3 ms per launch on CvT's stages.)
template <typename T>
__global__ __launch_bounds__(256) void im2col_nhwc_vec_kernel(const T* __restrict__ src, int nB, int H, int W, int Cin, int k, int stride, int pad,
                                                              int Ho, int Wo, int Kpad, T* __restrict__ cols) {
    constexpr int V = Vec16<T>::N;
    const int cv = Cin / V;               // vectors per tap
    const int per_row = Kpad / V;         // vectors per output row (the zero tail included)
    const int taps_v = k * k * cv;
    const long total = (long)nB * Ho * Wo * per_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int jv = (int)(i % per_row);
        const long r = i / per_row;
        Vec16<T> v = zero16<T>();
        if (jv < taps_v) {
            const int kk = jv / cv, c0 = (jv - kk * cv) * V;
            const int ky = kk / k, kx = kk - ky * k;
            const int ox = (int)(r % Wo);
            const long t = r / Wo;
            const int oy = (int)(t % Ho);
            const long b = t / Ho;
            const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = ld16<T>(src + ((b * H + iy) * W + ix) * Cin + c0);
        }
        st16<T>(cols + r * Kpad + (long)jv * V, v);
    }
}

// fp32 NCHW images -> bf16 columns (ky, kx, c): a thread produces eight consecutive columns of one output position (eight cached
// 4-byte reads from the image planes -- every pixel is read k^2 / stride^2 times over the launch, from the L1 / L2 --, ONE 16-byte
// store); the lanes of a wave cover consecutive column vectors, so the stores are contiguous.
__global__ __launch_bounds__(256) void im2col_nchw_vec_kernel(const float* __restrict__ src, int nB, int H, int W, int Cin, int k, int stride, int pad,
                                                              int Ho, int Wo, int Kpad, bf16* __restrict__ cols) {
    const int per_row = Kpad / 8;
    const int KK = k * k * Cin;
    const long total = (long)nB * Ho * Wo * per_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int jv = (int)(i % per_row);
        const long r = i / per_row;
        const int ox = (int)(r % Wo);
        const long t = r / Wo;
        const int oy = (int)(t % Ho);
        const long b = t / Ho;
        const float* img = src + b * Cin * (long)H * W;
        Vec16<bf16> v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = jv * 8 + e;
            float x = 0.f;
            if (j < KK) {
                const int kk = j / Cin, c = j - kk * Cin;
                const int ky = kk / k, kx = kk - ky * k;
                const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) x = img[((long)c * H + iy) * W + ix];
            }
            v.set(e, x);
        }
        st16<bf16>(cols + r * Kpad + (long)jv * 8, v);
    }
}

// dsrc[b,iy,ix,c] = sum over the (ky,kx) taps whose output position exists
template <typename T>
__global__ void col2im_kernel(const T* __restrict__ dcols, int nB, int H, int W, int Cin, int k, int stride, int pad, int Ho, int Wo,
                              int Kpad, float* __restrict__ dsrc) {
    const long total = (long)nB * H * W * Cin;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cin);
        const long p = i / Cin;
        const int ix = (int)(p % W), iy = (int)((p / W) % H);
        const long b = p / ((long)W * H);
        float s = 0.f;
        for (int ky = 0; ky < k; ++ky) {
            const int ty = iy + pad - ky;
            if (ty < 0 || ty % stride != 0) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int tx = ix + pad - kx;
                if (tx < 0 || tx % stride != 0) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                s += to_f32(dcols[((b * Ho + oy) * Wo + ox) * Kpad + (ky * k + kx) * Cin + c]);
            }
        }
        dsrc[i] = s;
    }
}

// the same gather, eight channels (one 16-byte vector of bf16 columns, two of fp32 output) per thread
__global__ __launch_bounds__(256) void col2im_vec_kernel(const bf16* __restrict__ dcols, int nB, int H, int W, int Cin, int k, int stride, int pad,
                                                         int Ho, int Wo, int Kpad, float* __restrict__ dsrc) {
    const int cv = Cin / 8;
    const long total = (long)nB * H * W * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cv) * 8;
        const long p = i / cv;
        const int ix = (int)(p % W);
        const long t = p / W;
        const int iy = (int)(t % H);
        const long b = t / H;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < k; ++ky) {
            const int ty = iy + pad - ky;
            if (ty < 0 || ty % stride != 0) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int tx = ix + pad - kx;
                if (tx < 0 || tx % stride != 0) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                const Vec16<bf16> v = ld16<bf16>(dcols + ((b * Ho + oy) * Wo + ox) * Kpad + (ky * k + kx) * Cin + c0);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += v.get(e);
            }
        }
        float* d = dsrc + p * Cin + c0;
        *reinterpret_cast<f32x4*>(d) = f32x4{s[0], s[1], s[2], s[3]};
        *reinterpret_cast<f32x4*>(d + 4) = f32x4{s[4], s[5], s[6], s[7]};
    }
}

// Thread layout shared by the three per-channel kernels below: a workgroup is C4 = C/4 channel lanes x PY position
// lanes (C4 * PY <= 256).  A thread keeps ONE group of 4 channels for its whole life -- its 36 taps (or its accumulators)
// sit in registers -- and strides over positions p = blockIdx.x * PY + ty, += gridDim.x * PY.
__device__ __forceinline__ f32x4 load4c(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4c(const bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void store4c(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4c(bf16* p, f32x4 v) {
    *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const T* __restrict__ x, const float* __restrict__ w, int flip, int nB, int H, int W, int C,
                                                        T* __restrict__ y) {
    const int c = threadIdx.x * 4, PY = blockDim.y;
    f32x4 wt[9];  // wt[t][e] = tap t of channel c+e
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ts = flip ? 8 - t : t;
#pragma unroll
        for (int e = 0; e < 4; ++e) wt[t][e] = w[(c + e) * 9 + ts];
    }
    const long P = (long)nB * H * W;
    for (long p = (long)blockIdx.x * PY + threadIdx.y; p < P; p += (long)gridDim.x * PY) {
        const int ix = (int)(p % W), iy = (int)((p / W) % H);
        const long b = p / ((long)W * H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sy = iy + ky - 1;
            if (sy < 0 || sy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int sx = ix + kx - 1;
                if (sx < 0 || sx >= W) continue;
                acc += load4c(x + ((b * H + sy) * W + sx) * C + c) * wt[ky * 3 + kx];
            }
        }
        store4c(y + p * C + c, acc);
    }
}

// ws[blk][(c+e)*9 + t] = sum over the block's positions of x(shifted by tap t) * dy
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, int nB, int H, int W, int C,
                                                              float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* sm = reinterpret_cast<f32x4*>(smem_raw);  // [PY][C4][9]
    const int c4 = threadIdx.x, c = c4 * 4, PY = blockDim.y, C4 = blockDim.x;
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long P = (long)nB * H * W;
    for (long p = (long)blockIdx.x * PY + threadIdx.y; p < P; p += (long)gridDim.x * PY) {
        const int ix = (int)(p % W), iy = (int)((p / W) % H);
        const long b = p / ((long)W * H);
        const f32x4 g = load4c(dy + p * C + c);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sy = iy + ky - 1;
            if (sy < 0 || sy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int sx = ix + kx - 1;
                if (sx < 0 || sx >= W) continue;
                acc[ky * 3 + kx] += g * load4c(x + ((b * H + sy) * W + sx) * C + c);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) sm[(threadIdx.y * C4 + c4) * 9 + t] = acc[t];
    __syncthreads();
    if (threadIdx.y == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            f32x4 s = sm[c4 * 9 + t];
            for (int yy = 1; yy < PY; ++yy) s += sm[(yy * C4 + c4) * 9 + t];
#pragma unroll
            for (int e = 0; e < 4; ++e) ws[((long)blockIdx.x * C + c + e) * 9 + t] = s[e];
        }
    }
}

// ---- strip versions (bf16, C % 8 == 0): a thread owns eight channels of FOUR consecutive positions of a row.  The three source rows
// of a strip are loaded once as six 16-byte vectors each (18 loads for 4 outputs instead of 36 eight-byte loads), the index
// arithmetic is paid once per strip, and the 3 x 3 taps slide over registers.  (The per-position kernels above are issue-bound:
// 83 us / 200 us per launch on CvT-13's stages against ~20 us of traffic.)
constexpr int DW_SW = 4;

__device__ __forceinline__ void dw_load_row(const bf16* __restrict__ row, int ix0, int W, int C, Vec16<bf16> (&v)[DW_SW + 2]) {
#pragma unroll
    for (int q = 0; q < DW_SW + 2; ++q) {
        const int sx = ix0 - 1 + q;
        v[q] = (sx >= 0 && sx < W) ? ld16<bf16>(row + (long)sx * C) : zero16<bf16>();
    }
}

__global__ __launch_bounds__(256) void dwconv3x3_strip_kernel(const bf16* __restrict__ x, const float* __restrict__ w, int flip, int nB, int H, int W,
                                                              int C, bf16* __restrict__ y) {
    const int c = threadIdx.x * 8, PY = blockDim.y;
    // the 72 taps of this thread's eight channels are contiguous in w [C][9]: eighteen 16-byte loads
    float wraw[72];
#pragma unroll
    for (int q = 0; q < 18; ++q) {
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(w + c * 9 + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) wraw[4 * q + e] = v4[e];
    }
    float wt[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) wt[t][e] = flip ? wraw[e * 9 + 8 - t] : wraw[e * 9 + t];
    const int strips_w = (W + DW_SW - 1) / DW_SW;
    const long S = (long)nB * H * strips_w;
    for (long s = (long)blockIdx.x * PY + threadIdx.y; s < S; s += (long)gridDim.x * PY) {
        const int ix0 = (int)(s % strips_w) * DW_SW;
        const long t = s / strips_w;
        const int iy = (int)(t % H);
        const long b = t / H;
        float acc[DW_SW][8];
#pragma unroll
        for (int o = 0; o < DW_SW; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o][e] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sy = iy + ky - 1;
            if (sy < 0 || sy >= H) continue;
            Vec16<bf16> v[DW_SW + 2];
            dw_load_row(x + ((b * H + sy) * (long)W) * C + c, ix0, W, C, v);
#pragma unroll
            for (int o = 0; o < DW_SW; ++o)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(v[o + kx].get(e), wt[ky * 3 + kx][e], acc[o][e]);
        }
#pragma unroll
        for (int o = 0; o < DW_SW; ++o) {
            if (ix0 + o >= W) break;
            Vec16<bf16> ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov.set(e, acc[o][e]);
            st16<bf16>(y + ((b * H + iy) * (long)W + ix0 + o) * C + c, ov);
        }
    }
}

// ws[blk][(c+e)*9 + t]: the same partial layout as dwconv3x3_wgrad_kernel
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_strip_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, int nB, int H, int W, int C,
                                                                    float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);  // [PY][CV][72]
    const int cv = threadIdx.x, c = cv * 8, PY = blockDim.y, CV = blockDim.x;
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    const int strips_w = (W + DW_SW - 1) / DW_SW;
    const long S = (long)nB * H * strips_w;
    for (long s = (long)blockIdx.x * PY + threadIdx.y; s < S; s += (long)gridDim.x * PY) {
        const int ix0 = (int)(s % strips_w) * DW_SW;
        const long t = s / strips_w;
        const int iy = (int)(t % H);
        const long b = t / H;
        Vec16<bf16> g[DW_SW];
#pragma unroll
        for (int o = 0; o < DW_SW; ++o) g[o] = (ix0 + o < W) ? ld16<bf16>(dy + ((b * H + iy) * (long)W + ix0 + o) * C + c) : zero16<bf16>();
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sy = iy + ky - 1;
            if (sy < 0 || sy >= H) continue;
            Vec16<bf16> v[DW_SW + 2];
            dw_load_row(x + ((b * H + sy) * (long)W) * C + c, ix0, W, C, v);
#pragma unroll
            for (int o = 0; o < DW_SW; ++o)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[ky * 3 + kx][e] = fmaf(g[o].get(e), v[o + kx].get(e), acc[ky * 3 + kx][e]);
        }
    }
    float* mine = sm + ((long)threadIdx.y * CV + cv) * 72;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) mine[t * 8 + e] = acc[t][e];
    __syncthreads();
    // 72 CV sums over the PY position lanes, spread over the block's threads
    const int tid = threadIdx.y * CV + cv, nthr = PY * CV;
    for (int i = tid; i < CV * 72; i += nthr) {
        const int cvi = i / 72, te = i % 72;
        float sum = 0.f;
        for (int yy = 0; yy < PY; ++yy) sum += sm[((long)yy * CV + cvi) * 72 + te];
        ws[((long)blockIdx.x * C + cvi * 8 + (te & 7)) * 9 + (te >> 3)] = sum;
    }
}

// ws[blk][0..C) = sum_r a[r][c],  ws[blk][C..2C) = sum_r a[r][c] * b[r][c]
template <typename T>
__global__ __launch_bounds__(256) void col_sums2_kernel(const T* __restrict__ a, const T* __restrict__ b, long rows, int C,
                                                        float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* sm = reinterpret_cast<f32x4*>(smem_raw);  // [PY][C4][2]
    const int c4 = threadIdx.x, c = c4 * 4, PY = blockDim.y, C4 = blockDim.x;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    for (long r = (long)blockIdx.x * PY + threadIdx.y; r < rows; r += (long)gridDim.x * PY) {
        const f32x4 av = load4c(a + r * C + c);
        s1 += av;
        s2 += av * load4c(b + r * C + c);
    }
    sm[(threadIdx.y * C4 + c4) * 2 + 0] = s1;
    sm[(threadIdx.y * C4 + c4) * 2 + 1] = s2;
    __syncthreads();
    if (threadIdx.y == 0) {
        for (int yy = 1; yy < PY; ++yy) {
            s1 += sm[(yy * C4 + c4) * 2 + 0];
            s2 += sm[(yy * C4 + c4) * 2 + 1];
        }
        store4c(ws + (long)blockIdx.x * 2 * C + c, s1);
        store4c(ws + (long)blockIdx.x * 2 * C + C + c, s2);
    }
}

// ACT 0: y = a1 x1 + a2 x2 + a3;  1: y = GELU(a1 x1 + a3);  2: y = x2 * GELU'(a1 x1 + a3);  3: y = max(a1 x1 + a3, 0);
// 4: y = x2 where a1 x1 + a3 > 0, else 0  (BatchNorm + ReLU of the residual stem and its backward)
template <typename T, int ACT>
__global__ void col_affine2_kernel(const T* __restrict__ x1, const T* __restrict__ x2, long n, int C, const float* __restrict__ a1,
                                   const float* __restrict__ a2, const float* __restrict__ a3, T* __restrict__ y) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        float v = a1[c] * to_f32(x1[i]) + a3[c];
        if constexpr (ACT == 0) {
            if (x2) v += a2[c] * to_f32(x2[i]);
        } else if constexpr (ACT == 1) {
            v = gelu_f(v);
        } else if constexpr (ACT == 2) {
            v = to_f32(x2[i]) * gelu_grad_f(v);
        } else if constexpr (ACT == 3) {
            v = fmaxf(v, 0.f);
        } else {
            v = v > 0.f ? to_f32(x2[i]) : 0.f;
        }
        y[i] = from_f32<T>(v);
    }
}

// BatchNorm2d coefficient arithmetic on [C]-sized vectors (one launch instead of ~15 tiny elementwise launches per layer).
// forward: sums = [sum d | sum d^2] over n positions -> coef = [a | shift | mean | rstd] with y = a*d + shift; updates the
// running statistics (momentum, unbiased variance) when they are given.
__global__ void bn_fwd_coeffs_kernel(const float* __restrict__ sums, float n, const float* __restrict__ gamma, const float* __restrict__ beta,
                                     float eps, float momentum, float* __restrict__ running_mean, float* __restrict__ running_var, int C,
                                     float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = sums[c] / n;
    const float var = fmaxf(sums[C + c] / n - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    const float a = gamma[c] * rstd;
    coef[c] = a;
    coef[C + c] = beta[c] - mean * a;
    coef[2 * C + c] = mean;
    coef[3 * C + c] = rstd;
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (n / (n - 1.f));
    }
}
// eval mode: coefficients from the running statistics
__global__ void bn_eval_coeffs_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, float eps, int C, float* __restrict__ coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rstd = rsqrtf(rvar[c] + eps);
    const float a = gamma[c] * rstd;
    coef[c] = a;
    coef[C + c] = beta[c] - rmean[c] * a;
    coef[2 * C + c] = rmean[c];
    coef[3 * C + c] = rstd;
}
// backward, local part: sums = [sum dy | sum dy*d] -> red = [sum dy | sum dy*xhat]  (= d beta | d gamma of this rank)
__global__ void bn_bwd_local_kernel(const float* __restrict__ sums, const float* __restrict__ coef, int C, float* __restrict__ red) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = coef[2 * C + c], rstd = coef[3 * C + c];
    red[c] = sums[c];
    red[C + c] = rstd * (sums[C + c] - mean * sums[c]);
}
// backward, after the cross-rank sum of red: d(d) = A*dy + B*d + Cc;  red == nullptr: fixed (eval) statistics, B = Cc = 0
__global__ void bn_bwd_coeffs_kernel(const float* __restrict__ red, float n, const float* __restrict__ gamma, const float* __restrict__ coef,
                                     int C, float* __restrict__ abc) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = coef[2 * C + c], rstd = coef[3 * C + c], g = gamma[c];
    const float m1 = red ? red[c] / n : 0.f, m2 = red ? red[C + c] / n : 0.f;
    const float A = g * rstd, B = -g * rstd * rstd * m2;
    abc[c] = A;
    abc[C + c] = B;
    abc[2 * C + c] = -g * rstd * m1 - B * mean;
}

// dst[b, y, x, :] = src[b, y, x, :] if (y < Hs && x < Ws) else 0, for y < Hd, x < Wd: zero-pads (Hd > Hs) or crops (Hd < Hs) a token grid
template <typename T>
__global__ void pad_crop_kernel(const T* __restrict__ src, int nB, int Hs, int Ws, int Hd, int Wd, int C, T* __restrict__ dst) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int CV = C / VEC;
    const long total = (long)nB * Hd * Wd * CV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        const long p = i / CV;
        const int x = (int)(p % Wd), y = (int)((p / Wd) % Hd);
        const long b = p / ((long)Wd * Hd);
        Vec16<T> v = zero16<T>();
        if (y < Hs && x < Ws) v = ld16<T>(src + ((b * Hs + y) * Ws + x) * C + cv * VEC);
        st16<T>(dst + p * C + cv * VEC, v);
    }
}

inline int grid_for(long n, int threads = 256, int cap = 8192) {
    long g = (n + threads - 1) / threads;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

inline dim3 chan_block(int C) {  // C/4 channel lanes x as many position lanes as fit 256 threads
    const int c4 = C / 4;
    int py = 256 / c4;
    if (py < 1) py = 1;
    return dim3(c4, py);
}

inline int reduce_blocks(long rows) {
    long nb = rows;
    if (nb > 512) nb = 512;
    if (nb < 1) nb = 1;
    return (int)nb;
}

}  // namespace

#define STREAM(s_) hipStream_t stream = reinterpret_cast<hipStream_t>(s_)

extern "C" int esvit_conv_im2col(int dtype, const void* src, int nchw, int nB, int H, int W, int Cin, int k, int stride, int pad, int Ho,
                                 int Wo, int Kpad, void* cols, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(src && cols && nB > 0 && H > 0 && W > 0 && Cin > 0 && k > 0 && stride > 0 && pad >= 0 && Kpad >= k * k * Cin,
                    "esvit_conv_im2col: bad args");
    ESVIT_CHECK_ARG(Ho == (H + 2 * pad - k) / stride + 1 && Wo == (W + 2 * pad - k) / stride + 1, "esvit_conv_im2col: bad output size");
    ESVIT_CHECK_ARG(dtype == ESVIT_BF16 || dtype == ESVIT_F32, "esvit_conv_im2col: bad dtype");
    const long total = (long)nB * Ho * Wo * Kpad;
    if (dtype == ESVIT_BF16 && !nchw && Cin % 8 == 0 && Kpad % 8 == 0 && ((uintptr_t)src % 16 == 0) && ((uintptr_t)cols % 16 == 0)) {
        hipLaunchKernelGGL(im2col_nhwc_vec_kernel<bf16>, dim3(grid_for(total / 8)), dim3(256), 0, stream, reinterpret_cast<const bf16*>(src), nB, H, W, Cin,
                           k, stride, pad, Ho, Wo, Kpad, reinterpret_cast<bf16*>(cols));
        ESVIT_CHECK_LAUNCH("conv_im2col(nhwc)");
        return ESVIT_OK;
    }
    if (dtype == ESVIT_BF16 && nchw && Kpad % 8 == 0 && ((uintptr_t)cols % 16 == 0)) {
        hipLaunchKernelGGL(im2col_nchw_vec_kernel, dim3(grid_for(total / 8)), dim3(256), 0, stream, reinterpret_cast<const float*>(src), nB, H, W, Cin, k,
                           stride, pad, Ho, Wo, Kpad, reinterpret_cast<bf16*>(cols));
        ESVIT_CHECK_LAUNCH("conv_im2col(nchw)");
        return ESVIT_OK;
    }
    if (dtype == ESVIT_BF16)
        hipLaunchKernelGGL(im2col_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, stream, src, nchw, nB, H, W, Cin, k, stride, pad, Ho, Wo,
                           Kpad, reinterpret_cast<bf16*>(cols));
    else
        hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, src, nchw, nB, H, W, Cin, k, stride, pad, Ho, Wo,
                           Kpad, reinterpret_cast<float*>(cols));
    ESVIT_CHECK_LAUNCH("conv_im2col");
    return ESVIT_OK;
}

extern "C" int esvit_conv_col2im(int dtype, const void* dcols, int nB, int H, int W, int Cin, int k, int stride, int pad, int Ho, int Wo,
                                 int Kpad, float* dsrc, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(dcols && dsrc && nB > 0 && H > 0 && W > 0 && Cin > 0 && k > 0 && stride > 0 && Kpad >= k * k * Cin,
                    "esvit_conv_col2im: bad args");
    ESVIT_CHECK_ARG(dtype == ESVIT_BF16 || dtype == ESVIT_F32, "esvit_conv_col2im: bad dtype");
    const long total = (long)nB * H * W * Cin;
    if (dtype == ESVIT_BF16 && Cin % 8 == 0 && Kpad % 8 == 0 && ((uintptr_t)dcols % 16 == 0) && ((uintptr_t)dsrc % 16 == 0)) {
        hipLaunchKernelGGL(col2im_vec_kernel, dim3(grid_for(total / 8)), dim3(256), 0, stream, reinterpret_cast<const bf16*>(dcols), nB, H, W, Cin, k,
                           stride, pad, Ho, Wo, Kpad, dsrc);
        ESVIT_CHECK_LAUNCH("conv_col2im(vec)");
        return ESVIT_OK;
    }
    if (dtype == ESVIT_BF16)
        hipLaunchKernelGGL(col2im_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, stream, reinterpret_cast<const bf16*>(dcols), nB, H, W,
                           Cin, k, stride, pad, Ho, Wo, Kpad, dsrc);
    else
        hipLaunchKernelGGL(col2im_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, reinterpret_cast<const float*>(dcols), nB, H, W,
                           Cin, k, stride, pad, Ho, Wo, Kpad, dsrc);
    ESVIT_CHECK_LAUNCH("conv_col2im");
    return ESVIT_OK;
}

extern "C" int esvit_dwconv3x3(int dtype, const void* x, const float* w, int flip, int nB, int H, int W, int C, void* y, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(x && w && y && nB > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "esvit_dwconv3x3: bad args (C=%d)", C);
    ESVIT_CHECK_ARG(dtype == ESVIT_BF16 || dtype == ESVIT_F32, "esvit_dwconv3x3: bad dtype");
    ESVIT_CHECK_ARG(C / 4 <= 256, "esvit_dwconv3x3: C=%d too wide", C);
    if (dtype == ESVIT_BF16 && C % 8 == 0 && C / 8 <= 256 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)w % 16 == 0)) {
        const int cvn = C / 8;
        const dim3 blk(cvn, 256 / cvn > 0 ? 256 / cvn : 1);
        const long strips = (long)nB * H * ((W + DW_SW - 1) / DW_SW);
        hipLaunchKernelGGL(dwconv3x3_strip_kernel, dim3(grid_for(strips, (int)blk.y, 2048)), blk, 0, stream, reinterpret_cast<const bf16*>(x), w, flip, nB, H,
                           W, C, reinterpret_cast<bf16*>(y));
        ESVIT_CHECK_LAUNCH("dwconv3x3(strip)");
        return ESVIT_OK;
    }
    const dim3 block = chan_block(C);
    const int grid = grid_for((long)nB * H * W, (int)block.y, 4096);
    if (dtype == ESVIT_BF16)
        hipLaunchKernelGGL(dwconv3x3_kernel<bf16>, dim3(grid), block, 0, stream, reinterpret_cast<const bf16*>(x), w, flip, nB, H, W, C,
                           reinterpret_cast<bf16*>(y));
    else
        hipLaunchKernelGGL(dwconv3x3_kernel<float>, dim3(grid), block, 0, stream, reinterpret_cast<const float*>(x), w, flip, nB, H, W, C,
                           reinterpret_cast<float*>(y));
    ESVIT_CHECK_LAUNCH("dwconv3x3");
    return ESVIT_OK;
}

int esvit_i_col_reduce_blocks(long rows) { return reduce_blocks(rows); }  // esvit_query

extern "C" int esvit_dwconv3x3_wgrad(int dtype, const void* x, const void* dy, int nB, int H, int W, int C, float* dw, float* ws,
                                     esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(x && dy && dw && ws && nB > 0 && H > 0 && W > 0 && C > 0, "esvit_dwconv3x3_wgrad: bad args");
    ESVIT_CHECK_ARG(dtype == ESVIT_BF16 || dtype == ESVIT_F32, "esvit_dwconv3x3_wgrad: bad dtype");
    ESVIT_CHECK_ARG(C % 4 == 0 && C / 4 <= 256, "esvit_dwconv3x3_wgrad: bad C=%d", C);
    const int nblk = reduce_blocks((long)nB * H * W);
    if (dtype == ESVIT_BF16 && C % 8 == 0 && C / 8 <= 256 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)dy % 16 == 0)) {
        const int cvn = C / 8;
        const dim3 blk(cvn, 256 / cvn > 0 ? 256 / cvn : 1);
        const size_t lds_s = (size_t)blk.x * blk.y * 72 * sizeof(float);
        static unsigned long long lds_set = 0;
        esvit_raise_lds(dwconv3x3_wgrad_strip_kernel, 73728, lds_set);
        hipLaunchKernelGGL(dwconv3x3_wgrad_strip_kernel, dim3(nblk), blk, lds_s, stream, reinterpret_cast<const bf16*>(x),
                           reinterpret_cast<const bf16*>(dy), nB, H, W, C, ws);
        ESVIT_CHECK_LAUNCH("dwconv3x3_wgrad(strip)");
        return esvit_partial_reduce(ws, nblk, 9 * C, 9L * C, dw, 0, stream);
    }
    const dim3 block = chan_block(C);
    const size_t lds = (size_t)block.x * block.y * 9 * sizeof(f32x4);
    if (dtype == ESVIT_BF16)
        hipLaunchKernelGGL(dwconv3x3_wgrad_kernel<bf16>, dim3(nblk), block, lds, stream, reinterpret_cast<const bf16*>(x),
                           reinterpret_cast<const bf16*>(dy), nB, H, W, C, ws);
    else
        hipLaunchKernelGGL(dwconv3x3_wgrad_kernel<float>, dim3(nblk), block, lds, stream, reinterpret_cast<const float*>(x),
                           reinterpret_cast<const float*>(dy), nB, H, W, C, ws);
    ESVIT_CHECK_LAUNCH("dwconv3x3_wgrad");
    return esvit_partial_reduce(ws, nblk, 9 * C, 9L * C, dw, 0, stream);
}

extern "C" int esvit_col_sums2(int dtype, const void* a, const void* b, int64_t rows, int C, float* out, float* ws, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(a && b && out && ws && rows > 0 && C > 0, "esvit_col_sums2: bad args");
    ESVIT_CHECK_ARG(dtype == ESVIT_BF16 || dtype == ESVIT_F32, "esvit_col_sums2: bad dtype");
    ESVIT_CHECK_ARG(C % 4 == 0 && C / 4 <= 256, "esvit_col_sums2: bad C=%d", C);
    const int nblk = reduce_blocks(rows);
    const dim3 block = chan_block(C);
    const size_t lds = (size_t)block.x * block.y * 2 * sizeof(f32x4);
    if (dtype == ESVIT_BF16)
        hipLaunchKernelGGL(col_sums2_kernel<bf16>, dim3(nblk), block, lds, stream, reinterpret_cast<const bf16*>(a),
                           reinterpret_cast<const bf16*>(b), (long)rows, C, ws);
    else
        hipLaunchKernelGGL(col_sums2_kernel<float>, dim3(nblk), block, lds, stream, reinterpret_cast<const float*>(a),
                           reinterpret_cast<const float*>(b), (long)rows, C, ws);
    ESVIT_CHECK_LAUNCH("col_sums2");
    return esvit_partial_reduce(ws, nblk, 2 * C, 2L * C, out, 0, stream);
}

template <typename T>
static void launch_col_affine2(int act, const void* x1, const void* x2, long n, int C, const float* a1, const float* a2, const float* a3,
                               void* y, hipStream_t stream) {
    const T* p1 = reinterpret_cast<const T*>(x1);
    const T* p2 = reinterpret_cast<const T*>(x2);
    T* py = reinterpret_cast<T*>(y);
    if (act == 1) hipLaunchKernelGGL((col_affine2_kernel<T, 1>), dim3(grid_for(n)), dim3(256), 0, stream, p1, p2, n, C, a1, a2, a3, py);
    else if (act == 2) hipLaunchKernelGGL((col_affine2_kernel<T, 2>), dim3(grid_for(n)), dim3(256), 0, stream, p1, p2, n, C, a1, a2, a3, py);
    else if (act == 3) hipLaunchKernelGGL((col_affine2_kernel<T, 3>), dim3(grid_for(n)), dim3(256), 0, stream, p1, p2, n, C, a1, a2, a3, py);
    else if (act == 4) hipLaunchKernelGGL((col_affine2_kernel<T, 4>), dim3(grid_for(n)), dim3(256), 0, stream, p1, p2, n, C, a1, a2, a3, py);
    else hipLaunchKernelGGL((col_affine2_kernel<T, 0>), dim3(grid_for(n)), dim3(256), 0, stream, p1, p2, n, C, a1, a2, a3, py);
}

extern "C" int esvit_col_affine2(int dtype, const void* x1, const void* x2, int64_t rows, int C, const float* a1, const float* a2,
                                 const float* a3, int act, void* y, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(x1 && a1 && a3 && y && rows > 0 && C > 0 && act >= 0 && act <= 4, "esvit_col_affine2: bad args");
    ESVIT_CHECK_ARG(act == 0 ? (!x2 || a2) : ((act == 1 || act == 3) ? !x2 : x2 != nullptr), "esvit_col_affine2: x2 / a2 do not fit act=%d", act);
    ESVIT_CHECK_ARG(dtype == ESVIT_BF16 || dtype == ESVIT_F32, "esvit_col_affine2: bad dtype");
    const long n = (long)rows * C;
    if (dtype == ESVIT_BF16) launch_col_affine2<bf16>(act, x1, x2, n, C, a1, a2, a3, y, stream);
    else launch_col_affine2<float>(act, x1, x2, n, C, a1, a2, a3, y, stream);
    ESVIT_CHECK_LAUNCH("col_affine2");
    return ESVIT_OK;
}

extern "C" int esvit_bn_fwd_coeffs(const float* sums, float n, const float* gamma, const float* beta, float eps, float momentum,
                                   float* running_mean, float* running_var, int C, float* coef, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(sums && gamma && beta && coef && C > 0 && n > 1.f && (!running_mean == !running_var), "esvit_bn_fwd_coeffs: bad args");
    hipLaunchKernelGGL(bn_fwd_coeffs_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, stream, sums, n, gamma, beta, eps, momentum, running_mean,
                       running_var, C, coef);
    ESVIT_CHECK_LAUNCH("bn_fwd_coeffs");
    return ESVIT_OK;
}

extern "C" int esvit_bn_eval_coeffs(const float* running_mean, const float* running_var, const float* gamma, const float* beta, float eps,
                                    int C, float* coef, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(running_mean && running_var && gamma && beta && coef && C > 0, "esvit_bn_eval_coeffs: bad args");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, stream, running_mean, running_var, gamma, beta, eps, C, coef);
    ESVIT_CHECK_LAUNCH("bn_eval_coeffs");
    return ESVIT_OK;
}

extern "C" int esvit_bn_bwd_local(const float* sums, const float* coef, int C, float* red, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(sums && coef && red && C > 0, "esvit_bn_bwd_local: bad args");
    hipLaunchKernelGGL(bn_bwd_local_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, stream, sums, coef, C, red);
    ESVIT_CHECK_LAUNCH("bn_bwd_local");
    return ESVIT_OK;
}

extern "C" int esvit_bn_bwd_coeffs(const float* red, float n, const float* gamma, const float* coef, int C, float* abc, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(gamma && coef && abc && C > 0 && n > 0.f, "esvit_bn_bwd_coeffs: bad args");
    hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, stream, red, n, gamma, coef, C, abc);
    ESVIT_CHECK_LAUNCH("bn_bwd_coeffs");
    return ESVIT_OK;
}

extern "C" int esvit_pad_crop_tokens(int dtype, const void* src, int nB, int Hs, int Ws, int Hd, int Wd, int C, void* dst, esvit_stream_t s_) {
    STREAM(s_);
    ESVIT_CHECK_ARG(src && dst && nB > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C > 0, "esvit_pad_crop_tokens: bad args");
    ESVIT_CHECK_ARG(dtype == ESVIT_BF16 || dtype == ESVIT_F32, "esvit_pad_crop_tokens: bad dtype");
    ESVIT_CHECK_ARG(C % (dtype == ESVIT_BF16 ? 8 : 4) == 0, "esvit_pad_crop_tokens: C=%d must be a multiple of the 16-byte vector", C);
    if (dtype == ESVIT_BF16)
        hipLaunchKernelGGL(pad_crop_kernel<bf16>, dim3(grid_for((long)nB * Hd * Wd * (C / 8))), dim3(256), 0, stream,
                           reinterpret_cast<const bf16*>(src), nB, Hs, Ws, Hd, Wd, C, reinterpret_cast<bf16*>(dst));
    else
        hipLaunchKernelGGL(pad_crop_kernel<float>, dim3(grid_for((long)nB * Hd * Wd * (C / 4))), dim3(256), 0, stream,
                           reinterpret_cast<const float*>(src), nB, Hs, Ws, Hd, Wd, C, reinterpret_cast<float*>(dst));
    ESVIT_CHECK_LAUNCH("pad_crop_tokens");
    return ESVIT_OK;
}
