// Crop producer: DataAugmentationDINO (datasets/build.py:203-261, utils.py:43-75) on decoded RGB images resident in HBM.
//
// The reference runs, per image and per crop, a chain of Pillow calls in 10 CPU workers (main_esvit.py:198):
//   img.crop(box).resize((S, S), BICUBIC) -> FLIP_LEFT_RIGHT -> ImageEnhance Brightness / Contrast / Color + HSV hue rotation in
//   a random order -> convert("L") -> ImageFilter.GaussianBlur -> ImageOps.solarize -> ToTensor -> Normalize.
// Here the random draws arrive as one int32 row per crop (esvit_amd/data.py samples them) and three kernels produce the crops:
//   aug_resize_kernel   one workgroup per TS x TS output tile: Resample.c's two passes (horizontal into LDS, vertical out of it),
//                       the 22-bit fixed-point taps of both axes computed in double by the workgroup itself; writes uint8 planes
//                       [n, 3, S, S] with the flip applied
//   aug_color_kernel    one workgroup per crop, the crop's three planes in LDS: the jitter operations in their drawn order
//                       (Contrast needs mean(L) of the image as it is at that point: a workgroup reduction), grayscale
//   aug_finish_kernel   one workgroup per (crop, channel): the plane in LDS, 3 + 3 box passes of BoxBlur.c ping-ponged between
//                       two LDS planes, solarize, ToTensor + Normalize, fp32 [n, 3, S, S] out
// All of it is byte / integer work bound by HBM (source box in, fp32 crop out); the uint8 planes between the kernels are
// 150 KB per 224^2 crop and live in L2 / MALL.  Arithmetic: augment_math.h, bit-exact against Pillow.
#include "common.h"
#include "esvit_hip.h"
#include "augment_math.h"

#pragma clang fp contract(off)

namespace {

constexpr int NP = ESVIT_AUG_PARAM_INTS;
// columns of a parameter row (include/esvit_hip.h)
enum { P_SRC = 0, P_TOP, P_LEFT, P_H, P_W, P_FLIP, P_OP0, P_OP1, P_OP2, P_OP3, P_BRIGHT, P_CONTRAST, P_SAT, P_HUE, P_GRAY, P_BLUR_R1, P_BLUR_WW,
       P_BLUR_FW, P_SOLARIZE };

// ---------------------------------------------------------------------------------------------------------------------
// resize: LDS = kx[TS][KX] | ky[TS][KY] | bounds[2][TS][2] | tmp[RMAX][TS] (packed r | g << 8 | b << 16)
// ---------------------------------------------------------------------------------------------------------------------
template <int TS>
__global__ __launch_bounds__(256) void aug_resize_kernel(const uint8_t* __restrict__ src, const int64_t* __restrict__ images,
                                                          const int32_t* __restrict__ params, int S, int KX, int KY, int RMAX,
                                                          uint8_t* __restrict__ planes) {
    extern __shared__ __align__(16) unsigned char smem[];
    int32_t* kx = reinterpret_cast<int32_t*>(smem);
    int32_t* ky = kx + TS * KX;
    int32_t* bnd = ky + TS * KY;  // [axis][TS][2]
    uint32_t* tmp = reinterpret_cast<uint32_t*>(bnd + 2 * TS * 2);

    const int crop = blockIdx.y;
    const int tiles = (S + TS - 1) / TS;
    const int ty = blockIdx.x / tiles, tx = blockIdx.x % tiles;
    const int32_t* p = params + (long)crop * NP;
    const int top = p[P_TOP], left = p[P_LEFT], h = p[P_H], w = p[P_W], flip = p[P_FLIP];
    const int64_t* im = images + (long)p[P_SRC] * 3;
    const long pitch = im[2] * 3;
    const uint8_t* base = src + im[0] + (long)top * pitch + (long)left * 3;
    const int tid = threadIdx.x;

    // the taps of this tile's TS columns and TS rows (Resample.c precompute_coeffs, one output position per thread)
    if (tid < 2 * TS) {
        const int axis = tid / TS, t = tid % TS;
        const int pos = (axis ? ty : tx) * TS + t;
        int first = 0, count = 0;
        if (pos < S) aug::resample_row(axis ? h : w, S, pos, axis ? KY : KX, &first, &count, (axis ? ky : kx) + t * (axis ? KY : KX));
        bnd[(axis * TS + t) * 2] = first;
        bnd[(axis * TS + t) * 2 + 1] = count;
    }
    __syncthreads();

    // source rows this tile's vertical pass reads: [y0, y1)
    const int ny = min(TS, S - ty * TS), nx = min(TS, S - tx * TS);
    const int y0 = bnd[(TS + 0) * 2];
    int R = bnd[(TS + ny - 1) * 2] + bnd[(TS + ny - 1) * 2 + 1] - y0;
    if (R > RMAX) R = RMAX;  // cannot happen when the host passed the true largest box

    // horizontal pass: tmp[r][xx] = clip8(sum_k src[y0 + r][first + k] * kx[xx][k]), uint8 per channel as in Resample.c
    for (int item = tid; item < R * TS; item += 256) {
        const int r = item / TS, xx = item % TS;
        if (xx >= nx) continue;
        const int first = bnd[xx * 2], count = bnd[xx * 2 + 1];
        const uint8_t* row = base + (long)(y0 + r) * pitch + (long)first * 3;
        const int32_t* k = kx + xx * KX;
        int32_t s0 = 1 << (aug::PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int i = 0; i < count; ++i) {
            const int32_t c = k[i];
            s0 += (int32_t)row[3 * i] * c;
            s1 += (int32_t)row[3 * i + 1] * c;
            s2 += (int32_t)row[3 * i + 2] * c;
        }
        tmp[r * TS + xx] = (uint32_t)aug::clip8(s0) | ((uint32_t)aug::clip8(s1) << 8) | ((uint32_t)aug::clip8(s2) << 16);
    }
    __syncthreads();

    // vertical pass out of LDS; RandomHorizontalFlip mirrors the column on the way out
    uint8_t* out = planes + (long)crop * 3 * S * S;
    for (int item = tid; item < TS * TS; item += 256) {
        const int yy = item / TS, xx = item % TS;
        if (yy >= ny || xx >= nx) continue;
        const int first = bnd[(TS + yy) * 2] - y0, count = bnd[(TS + yy) * 2 + 1];
        const int32_t* k = ky + yy * KY;
        int32_t s0 = 1 << (aug::PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int i = 0; i < count; ++i) {
            int r = first + i;
            if (r >= R) break;
            const uint32_t v = tmp[r * TS + xx];
            const int32_t c = k[i];
            s0 += (int32_t)(v & 255) * c;
            s1 += (int32_t)((v >> 8) & 255) * c;
            s2 += (int32_t)((v >> 16) & 255) * c;
        }
        const int Y = ty * TS + yy, X0 = tx * TS + xx;
        const int X = flip ? S - 1 - X0 : X0;
        const long o = (long)Y * S + X;
        out[o] = aug::clip8(s0);
        out[(long)S * S + o] = aug::clip8(s1);
        out[2L * S * S + o] = aug::clip8(s2);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// colour: the three planes of one crop in LDS
// ---------------------------------------------------------------------------------------------------------------------
constexpr int COLOR_THREADS = 1024;

__device__ __forceinline__ int block_sum(int v, int* scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();  // scratch may still be read from a previous reduction
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    int t = 0;
    for (int i = 0; i < COLOR_THREADS / 64; ++i) t += scratch[i];
    return t;
}

__global__ __launch_bounds__(COLOR_THREADS) void aug_color_kernel(const int32_t* __restrict__ params, int S, uint8_t* __restrict__ planes) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int scratch[COLOR_THREADS / 64];
    const int32_t* p = params + (long)blockIdx.x * NP;
    const int gray = p[P_GRAY];
    if (p[P_OP0] < 0 && p[P_OP1] < 0 && p[P_OP2] < 0 && p[P_OP3] < 0 && !gray) return;  // the jitter was not applied to this crop
    const int n = S * S, tid = threadIdx.x;
    uint8_t* g = planes + (long)blockIdx.x * 3 * n;
    uint8_t *R = smem, *G = smem + n, *B = smem + 2 * n;
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(g) & 3) == 0) {
        for (int i = tid; i < 3 * n / 4; i += COLOR_THREADS) reinterpret_cast<uint32_t*>(smem)[i] = reinterpret_cast<const uint32_t*>(g)[i];
    } else {
        for (int i = tid; i < 3 * n; i += COLOR_THREADS) smem[i] = g[i];
    }
    __syncthreads();

    for (int slot = 0; slot < 4; ++slot) {
        const int op = p[P_OP0 + slot];
        if (op < 0) continue;
        if (op == 0) {  // ImageEnhance.Brightness: blend with black
            const float f = __int_as_float(p[P_BRIGHT]);
            for (int i = tid; i < 3 * n; i += COLOR_THREADS) smem[i] = (uint8_t)aug::blend(0, smem[i], f);
        } else if (op == 1) {  // ImageEnhance.Contrast: blend with the grey of int(mean(L) + 0.5)
            const float f = __int_as_float(p[P_CONTRAST]);
            int part = 0;
            for (int i = tid; i < n; i += COLOR_THREADS) part += aug::rgb_to_l(R[i], G[i], B[i]);
            const int total = block_sum(part, scratch);
            const int mean = (int)((double)total / (double)n + 0.5);
            for (int i = tid; i < 3 * n; i += COLOR_THREADS) smem[i] = (uint8_t)aug::blend(mean, smem[i], f);
        } else if (op == 2) {  // ImageEnhance.Color: blend with the pixel's own L
            const float f = __int_as_float(p[P_SAT]);
            for (int i = tid; i < n; i += COLOR_THREADS) {
                const int r = R[i], gg = G[i], b = B[i], l = aug::rgb_to_l(r, gg, b);
                R[i] = (uint8_t)aug::blend(l, r, f);
                G[i] = (uint8_t)aug::blend(l, gg, f);
                B[i] = (uint8_t)aug::blend(l, b, f);
            }
        } else {  // hue: RGB -> HSV, h += delta (mod 256), HSV -> RGB
            const int delta = p[P_HUE];
            for (int i = tid; i < n; i += COLOR_THREADS) {
                int hh, ss, vv, r, gg, b;
                aug::rgb_to_hsv(R[i], G[i], B[i], &hh, &ss, &vv);
                aug::hsv_to_rgb((hh + delta) & 255, ss, vv, &r, &gg, &b);
                R[i] = (uint8_t)r;
                G[i] = (uint8_t)gg;
                B[i] = (uint8_t)b;
            }
        }
        __syncthreads();
    }
    if (gray) {  // RandomGrayscale: L replicated
        for (int i = tid; i < n; i += COLOR_THREADS) {
            const uint8_t l = (uint8_t)aug::rgb_to_l(R[i], G[i], B[i]);
            R[i] = l;
            G[i] = l;
            B[i] = l;
        }
        __syncthreads();
    }
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(g) & 3) == 0) {
        for (int i = tid; i < 3 * n / 4; i += COLOR_THREADS) reinterpret_cast<uint32_t*>(g)[i] = reinterpret_cast<const uint32_t*>(smem)[i];
    } else {
        for (int i = tid; i < 3 * n; i += COLOR_THREADS) g[i] = smem[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// finish: blur (two LDS planes), solarize, ToTensor + Normalize
// ---------------------------------------------------------------------------------------------------------------------
constexpr int FINISH_THREADS = 512;

__global__ __launch_bounds__(FINISH_THREADS) void aug_finish_kernel(const int32_t* __restrict__ params, int S, const uint8_t* __restrict__ planes,
                                                                     float* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int crop = blockIdx.x / 3, ch = blockIdx.x % 3;
    const int32_t* p = params + (long)crop * NP;
    const int n = S * S, tid = threadIdx.x;
    const uint8_t* g = planes + ((long)crop * 3 + ch) * n;
    float* o = out + ((long)crop * 3 + ch) * n;
    const int sol = p[P_SOLARIZE];
    const float mean = ch == 0 ? 0.485f : (ch == 1 ? 0.456f : 0.406f);
    const float stdv = ch == 0 ? 0.229f : (ch == 1 ? 0.224f : 0.225f);
    const int r1 = p[P_BLUR_R1];
    if (r1 <= 0) {  // no blur: stream the plane through
        for (int i = tid; i < n; i += FINISH_THREADS) {
            int v = g[i];
            if (sol && v >= 128) v = 255 - v;  // ImageOps.solarize, threshold 128
            o[i] = aug::normalize(v, mean, stdv);
        }
        return;
    }
    const int r = r1 - 1;
    const uint32_t ww = (uint32_t)p[P_BLUR_WW], fw = (uint32_t)p[P_BLUR_FW];
    uint8_t* a = smem;
    uint8_t* b = smem + ((n + 15) & ~15);
    for (int i = tid; i < n; i += FINISH_THREADS) a[i] = g[i];
    __syncthreads();
    for (int pass = 0; pass < 6; ++pass) {  // BoxBlur.c: three passes along x, then three along y, uint8 after each
        const bool vertical = pass >= 3;
        for (int i = tid; i < n; i += FINISH_THREADS) {
            const int y = i / S, x = i % S;
            b[i] = vertical ? aug::box_tap(a + x, S, S, y, r, ww, fw) : aug::box_tap(a + y * S, 1, S, x, r, ww, fw);
        }
        __syncthreads();
        uint8_t* t = a;
        a = b;
        b = t;
    }
    for (int i = tid; i < n; i += FINISH_THREADS) {
        int v = a[i];
        if (sol && v >= 128) v = 255 - v;
        o[i] = aug::normalize(v, mean, stdv);
    }
}

constexpr size_t LDS_MAX = 160 * 1024;

size_t resize_lds(int TS, int KX, int KY, int RMAX) { return ((size_t)TS * KX + (size_t)TS * KY + 4 * TS + (size_t)RMAX * TS) * 4; }

// rows of the horizontal pass one tile of TS output rows can need when the axis is resized in_size -> S
int tile_rows(int TS, int in_size, int S) {
    const double scale = (double)in_size / S, support = 2.0 * (scale < 1.0 ? 1.0 : scale);
    return (int)((TS - 1) * scale + 2 * support) + 3;
}

template <int TS>
int launch_resize(const uint8_t* src, const int64_t* images, const int32_t* params, int n, int S, int KX, int KY, int RMAX, uint8_t* planes,
                  hipStream_t stream) {
    auto kern = aug_resize_kernel<TS>;
    const size_t lds = resize_lds(TS, KX, KY, RMAX);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int tiles = (S + TS - 1) / TS;
    hipLaunchKernelGGL(kern, dim3(tiles * tiles, n), dim3(256), lds, stream, src, images, params, S, KX, KY, RMAX, planes);
    ESVIT_CHECK_LAUNCH("aug_crops(resize)");
    return ESVIT_OK;
}

}  // namespace

// largest crop box side esvit_aug_crops accepts for output size S (the smallest tile must fit its LDS)
int64_t esvit_i_aug_max_box(int S) {
    int64_t lo = S, hi = 1 << 20;
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) / 2;
        const int K = aug::resample_ksize((int)mid, S);
        if (resize_lds(8, K, K, tile_rows(8, (int)mid, S)) <= LDS_MAX) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

extern "C" int esvit_aug_crops(const uint8_t* src, const int64_t* images, const int32_t* params, int n, int S, int max_h, int max_w,
                               uint8_t* planes, float* out, esvit_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ESVIT_CHECK_ARG(src && images && params && planes && out, "esvit_aug_crops: null pointer");
    ESVIT_CHECK_ARG(n >= 0 && n <= 65535 && S > 0 && max_h > 0 && max_w > 0, "esvit_aug_crops: bad sizes n=%d S=%d box %dx%d", n, S, max_h, max_w);
    if (n == 0) return ESVIT_OK;
    const size_t plane_lds = (((size_t)S * S + 15) & ~(size_t)15);
    if (3 * plane_lds > LDS_MAX - 256) {
        esvit_set_error("esvit_aug_crops: S=%d: the three planes of a crop do not fit the LDS of a CU", S);
        return ESVIT_ERR_UNSUPPORTED;
    }
    const int KX = aug::resample_ksize(max_w, S), KY = aug::resample_ksize(max_h, S);
    int rc = ESVIT_ERR_UNSUPPORTED;
    if (resize_lds(32, KX, KY, tile_rows(32, max_h, S)) <= LDS_MAX / 2) rc = launch_resize<32>(src, images, params, n, S, KX, KY, tile_rows(32, max_h, S), planes, stream);
    else if (resize_lds(16, KX, KY, tile_rows(16, max_h, S)) <= LDS_MAX) rc = launch_resize<16>(src, images, params, n, S, KX, KY, tile_rows(16, max_h, S), planes, stream);
    else if (resize_lds(8, KX, KY, tile_rows(8, max_h, S)) <= LDS_MAX) rc = launch_resize<8>(src, images, params, n, S, KX, KY, tile_rows(8, max_h, S), planes, stream);
    else esvit_set_error("esvit_aug_crops: crop box %dx%d -> %d is beyond esvit_query(ESVIT_Q_AUG_MAX_BOX)", max_h, max_w, S);
    if (rc != ESVIT_OK) return rc;
    {
        auto kern = aug_color_kernel;
        const size_t lds = 3 * (size_t)S * S;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(n), dim3(COLOR_THREADS), lds, stream, params, S, planes);
        ESVIT_CHECK_LAUNCH("aug_crops(colour)");
    }
    {
        auto kern = aug_finish_kernel;
        const size_t lds = 2 * plane_lds;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(3 * n), dim3(FINISH_THREADS), lds, stream, params, S, planes, out);
        ESVIT_CHECK_LAUNCH("aug_crops(finish)");
    }
    return ESVIT_OK;
}
