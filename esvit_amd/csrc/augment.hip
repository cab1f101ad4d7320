// Crop producer: DataAugmentationDINO (datasets/build.py:203-261, utils.py:43-75) on decoded RGB images resident in HBM.
//
// The reference runs, per image and per crop, a chain of Pillow calls in 10 CPU workers (main_esvit.py:198):
//   img.crop(box).resize((S, S), BICUBIC) -> FLIP_LEFT_RIGHT -> ImageEnhance Brightness / Contrast / Color + HSV hue rotation in
//   a random order -> convert("L") -> ImageFilter.GaussianBlur -> ImageOps.solarize -> ToTensor -> Normalize.
// Here the random draws arrive as one int32 row per crop (esvit_amd/data.py samples them) and three kernels produce the crops,
// every one of them parallel over the PIXELS of all crops (thousands of workgroups, no per-image serial section):
//   aug_resize_kernel   one workgroup per TS x TS output tile: the source rows the tile needs are staged into LDS with aligned
//                       dword loads, then Resample.c's two passes run out of LDS (horizontal -> uint8 -> vertical), the 22-bit
//                       fixed-point taps of both axes computed in double by the workgroup itself; writes uint8 planes [n, 3, S, S]
//                       with the flip applied
//   aug_mean_kernel     ImageEnhance.Contrast blends with the grey of mean(L) of the image AS IT IS when its turn comes: crops
//                       that drew the jitter get the operations preceding Contrast applied in registers and L summed (one
//                       integer atomic per wave) -- the only global dependency of the whole chain
//   aug_finish_kernel   one workgroup per 64 x 64 tile: all jitter operations + grayscale in registers (4 pixels per lane), then,
//                       for blurred crops, BoxBlur.c's 3 + 3 box passes on the tile + halo in LDS (dword-wide, 4 outputs per
//                       lane), solarize, ToTensor + Normalize, float4 stores of the fp32 crop [n, 3, S, S]
//                       (aug_finish_plane_kernel: whole-plane fallback for box radii beyond the tile's halo budget)
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
// resize: LDS = kx[TS][KX] | ky[TS][KY] | bounds[2][TS][2] | tmp[RMAX][TS] (packed r | g << 8 | b << 16) | stage[RMAX][SWD] dwords
// ---------------------------------------------------------------------------------------------------------------------
template <int TS, bool STAGE>
__global__ __launch_bounds__(256) void aug_resize_kernel(const uint8_t* __restrict__ src, const int64_t* __restrict__ images,
                                                          const int32_t* __restrict__ params, int S, int KX, int KY, int RMAX, int SWD,
                                                          uint8_t* __restrict__ planes, int* __restrict__ sums) {
    extern __shared__ __align__(16) unsigned char smem[];
    int32_t* kx = reinterpret_cast<int32_t*>(smem);
    int32_t* ky = kx + TS * KX;
    int32_t* bnd = ky + TS * KY;  // [axis][TS][2]
    uint32_t* tmp = reinterpret_cast<uint32_t*>(bnd + 2 * TS * 2);
    uint32_t* stage = tmp + RMAX * TS;

    const int crop = blockIdx.y;
    const int tiles = (S + TS - 1) / TS;
    const int ty = blockIdx.x / tiles, tx = blockIdx.x % tiles;
    const int32_t* p = params + (long)crop * NP;
    const int top = p[P_TOP], left = p[P_LEFT], h = p[P_H], w = p[P_W], flip = p[P_FLIP];
    const int64_t* im = images + (long)p[P_SRC] * 3;
    const long pitch = im[2] * 3;
    const uint8_t* base = src + im[0] + (long)top * pitch + (long)left * 3;
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) sums[crop] = 0;  // the accumulator of aug_mean_kernel

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

    // source rows [y0, y0 + R) and columns [cx0, cx1) this tile reads
    const int ny = min(TS, S - ty * TS), nx = min(TS, S - tx * TS);
    const int y0 = bnd[(TS + 0) * 2];
    int R = bnd[(TS + ny - 1) * 2] + bnd[(TS + ny - 1) * 2 + 1] - y0;
    if (R > RMAX) R = RMAX;  // cannot happen when the host passed the true largest box
    const int cx0 = bnd[0], cx1 = bnd[(nx - 1) * 2] + bnd[(nx - 1) * 2 + 1];

    if constexpr (STAGE) {  // aligned dword loads of the row segments (pixels are 3 bytes: a segment starts anywhere in a dword);
                            // a wave takes four rows at a time, eight independent loads in flight per lane
        const int wave = tid >> 6, lane = tid & 63;
        const int wbytes = (cx1 - cx0) * 3;
        const uintptr_t a00 = reinterpret_cast<uintptr_t>(base + (long)y0 * pitch + (long)cx0 * 3);
        for (int r0 = wave; r0 < R; r0 += 16) {
            for (int d0 = lane; d0 < SWD; d0 += 128) {
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = r0 + 4 * (u >> 1), d = d0 + 64 * (u & 1);
                    const uintptr_t a = a00 + (uintptr_t)((long)r * pitch);
                    const int nd = (int)((((a + wbytes + 3) & ~(uintptr_t)3) - (a & ~(uintptr_t)3)) >> 2);
                    v[u] = (r < R && d < nd) ? reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3)[d] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = r0 + 4 * (u >> 1), d = d0 + 64 * (u & 1);
                    if (r < R && d < SWD) stage[r * SWD + d] = v[u];
                }
            }
        }
        __syncthreads();
    }

    // horizontal pass: tmp[r][xx] = clip8(sum_k src[y0 + r][first + k] * kx[xx][k]), uint8 per channel as in Resample.c
    for (int item = tid; item < R * TS; item += 256) {
        const int r = item / TS, xx = item % TS;
        if (xx >= nx) continue;
        const int first = bnd[xx * 2], count = bnd[xx * 2 + 1];
        const uint8_t* grow = base + (long)(y0 + r) * pitch;
        const uint8_t* row;
        if constexpr (STAGE) row = reinterpret_cast<const uint8_t*>(stage + r * SWD) + (reinterpret_cast<uintptr_t>(grow + (long)cx0 * 3) & 3) + (first - cx0) * 3;
        else row = grow + (long)first * 3;
        const int32_t* k = kx + xx * KX;
        int32_t s0 = 1 << (aug::PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int i = 0; i < count; ++i) {
            const int32_t c = k[i];
            s0 += AUG_MUL24(row[3 * i], c);  // |c| < 2^23: the 22-bit fixed-point taps of a normalised bicubic kernel
            s1 += AUG_MUL24(row[3 * i + 1], c);
            s2 += AUG_MUL24(row[3 * i + 2], c);
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
            s0 += AUG_MUL24(v & 255, c);
            s1 += AUG_MUL24((v >> 8) & 255, c);
            s2 += AUG_MUL24((v >> 16) & 255, c);
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
// the jitter of 4 pixels in registers
// ---------------------------------------------------------------------------------------------------------------------
struct Px4 {
    int r[4], g[4], b[4];
};

__device__ __forceinline__ void unpack4(uint32_t v, int (&c)[4]) {
    c[0] = v & 255;
    c[1] = (v >> 8) & 255;
    c[2] = (v >> 16) & 255;
    c[3] = v >> 24;
}
__device__ __forceinline__ uint32_t pack4(const int (&c)[4]) { return (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24); }

// hsv2rgb's per-byte quantities (augment_math.h hsv_sector / hsv_saturation), one table per workgroup in LDS
struct HsvTables {
    float f[256], fs[256];
    int i[256];
};
__device__ __forceinline__ void fill_hsv_tables(HsvTables& t) {  // blockDim.x >= 256; the caller synchronises
    if (threadIdx.x < 256) {
        aug::hsv_sector(threadIdx.x, &t.i[threadIdx.x], &t.f[threadIdx.x]);
        t.fs[threadIdx.x] = aug::hsv_saturation(threadIdx.x);
    }
}

// PREFIX: only the operations drawn BEFORE Contrast (what the image looks like when ImageEnhance.Contrast takes its mean);
// otherwise the whole chain (Contrast blending with `grey`) and RandomGrayscale
template <bool PREFIX>
__device__ __forceinline__ void colour_ops(Px4& px, const int32_t* __restrict__ p, int grey, const HsvTables& tab) {
    for (int slot = 0; slot < 4; ++slot) {
        const int op = p[P_OP0 + slot];
        if (op < 0) continue;
        if (op == 0) {  // ImageEnhance.Brightness: blend with black
            const float f = __int_as_float(p[P_BRIGHT]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                px.r[i] = aug::blend(0, px.r[i], f);
                px.g[i] = aug::blend(0, px.g[i], f);
                px.b[i] = aug::blend(0, px.b[i], f);
            }
        } else if (op == 1) {  // ImageEnhance.Contrast: blend with the grey of int(mean(L) + 0.5)
            if (PREFIX) return;
            const float f = __int_as_float(p[P_CONTRAST]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                px.r[i] = aug::blend(grey, px.r[i], f);
                px.g[i] = aug::blend(grey, px.g[i], f);
                px.b[i] = aug::blend(grey, px.b[i], f);
            }
        } else if (op == 2) {  // ImageEnhance.Color: blend with the pixel's own L
            const float f = __int_as_float(p[P_SAT]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int l = aug::rgb_to_l(px.r[i], px.g[i], px.b[i]);
                px.r[i] = aug::blend(l, px.r[i], f);
                px.g[i] = aug::blend(l, px.g[i], f);
                px.b[i] = aug::blend(l, px.b[i], f);
            }
        } else {  // hue: RGB -> HSV, h += delta (mod 256), HSV -> RGB
            const int delta = p[P_HUE];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int hh, ss, vv;
                aug::rgb_to_hsv(px.r[i], px.g[i], px.b[i], &hh, &ss, &vv);
                hh = (hh + delta) & 255;
                aug::hsv_to_rgb_t(ss, vv, tab.i[hh], tab.f[hh], tab.fs[ss], &px.r[i], &px.g[i], &px.b[i]);
            }
        }
    }
    if (!PREFIX && p[P_GRAY]) {  // RandomGrayscale: L replicated
#pragma unroll
        for (int i = 0; i < 4; ++i) px.r[i] = px.g[i] = px.b[i] = aug::rgb_to_l(px.r[i], px.g[i], px.b[i]);
    }
}

__device__ __forceinline__ bool has_contrast(const int32_t* __restrict__ p) { return p[P_OP0] == 1 || p[P_OP1] == 1 || p[P_OP2] == 1 || p[P_OP3] == 1; }
__device__ __forceinline__ bool has_colour(const int32_t* __restrict__ p) { return p[P_OP0] >= 0 || p[P_OP1] >= 0 || p[P_OP2] >= 0 || p[P_OP3] >= 0 || p[P_GRAY]; }

__device__ __forceinline__ Px4 load_px4(const uint8_t* __restrict__ g, long n, long i4) {  // dword i4 of the three planes of a crop
    Px4 px;
    unpack4(reinterpret_cast<const uint32_t*>(g)[i4], px.r);
    unpack4(reinterpret_cast<const uint32_t*>(g + n)[i4], px.g);
    unpack4(reinterpret_cast<const uint32_t*>(g + 2 * n)[i4], px.b);
    return px;
}

constexpr int MEAN_PX = 4096;  // pixels per workgroup of aug_mean_kernel

__global__ __launch_bounds__(256) void aug_mean_kernel(const int32_t* __restrict__ params, int S, const uint8_t* __restrict__ planes, int* __restrict__ sums) {
    __shared__ HsvTables tab;
    const int crop = blockIdx.y;
    const int32_t* p = params + (long)crop * NP;
    if (!has_contrast(p)) return;
    fill_hsv_tables(tab);
    __syncthreads();
    const long n = (long)S * S;
    const uint8_t* g = planes + (long)crop * 3 * n;
    int part = 0;
    for (int k = 0; k < MEAN_PX / 4 / 256; ++k) {
        const long i4 = (long)blockIdx.x * (MEAN_PX / 4) + k * 256 + threadIdx.x;
        if (i4 * 4 >= n) break;
        Px4 px = load_px4(g, n, i4);
        colour_ops<true>(px, p, 0, tab);
#pragma unroll
        for (int i = 0; i < 4; ++i) part += aug::rgb_to_l(px.r[i], px.g[i], px.b[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(sums + crop, part);
}

// ImageEnhance.Contrast: int(ImageStat.Stat(L).mean[0] + 0.5), the mean a double quotient
__device__ __forceinline__ int contrast_grey(int total, int S) { return (int)((double)total / (double)((long)S * S) + 0.5); }

__device__ __forceinline__ void store_normalized(float* __restrict__ o, uint32_t v, int sol, float mean, float stdv) {
    int c[4];
    unpack4(v, c);
    f32x4 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int t = c[i];
        if (sol && t >= 128) t = 255 - t;  // ImageOps.solarize, threshold 128
        f[i] = aug::normalize(t, mean, stdv);
    }
    *reinterpret_cast<f32x4*>(o) = f;
}

// ---------------------------------------------------------------------------------------------------------------------
// finish, tiled: jitter in registers; BoxBlur.c on tile + halo in LDS
// ---------------------------------------------------------------------------------------------------------------------
// item -> (item / d, item % d) for item < 2520, d <= 26: one multiply by ceil(2^16 / d) (exact in that range)
__device__ __forceinline__ void divmod_small(int item, int d, int magic, int& q, int& r) {
    q = (int)(AUG_UMUL24(item, magic) >> 16);
    r = item - AUG_MUL24(q, d);
}

constexpr int FT = 64;                 // tile side
constexpr int BLUR_RMAX = 2;           // largest box radius blurred in the tile (halo 3 * (r + 1) <= 9; the reference draws r <= 1)
constexpr int FHALO = 3 * (BLUR_RMAX + 1);
constexpr int EWS = 88;                // LDS row stride in bytes: 64 + 2 * 9 rounded out to dwords on both sides
constexpr int EHMAX = FT + 2 * FHALO;  // 82 rows
constexpr int FPLANE = EHMAX * EWS;

// one horizontal box pass over a region of eh rows x ew4 dwords: 4 outputs per lane from the dwords around them; taps beyond the
// region clamp to its edge byte (the image edge where the region ends at the image, discarded halo otherwise)
template <int R>
__device__ __forceinline__ void box_pass_x(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int eh, int ew4, uint32_t ww, uint32_t fw) {
    constexpr int DL = (R + 1 + 3) / 4, ND = 2 * DL + 1;
    const int magic = (65536 + ew4 - 1) / ew4;
    for (int item = threadIdx.x; item < eh * ew4; item += 256) {
        int y, cg;
        divmod_small(item, ew4, magic, y, cg);
        const uint32_t* row = reinterpret_cast<const uint32_t*>(in + y * EWS);
        int w[4 * ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int q = cg + d - DL;
            uint32_t v;
            if (q < 0) v = (row[0] & 255) * 0x01010101u;
            else if (q >= ew4) v = (row[ew4 - 1] >> 24) * 0x01010101u;
            else v = row[q];
            w[4 * d] = v & 255;
            w[4 * d + 1] = (v >> 8) & 255;
            w[4 * d + 2] = (v >> 16) & 255;
            w[4 * d + 3] = v >> 24;
        }
        int o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 4 * DL + i;
            uint32_t acc = 0;
#pragma unroll
            for (int d = -R; d <= R; ++d) acc += (uint32_t)w[c + d];
            const uint32_t bulk = AUG_UMUL24(acc, ww) + AUG_UMUL24(w[c - R - 1] + w[c + R + 1], fw);
            o[i] = (int)((bulk + (1u << 23)) >> 24);
        }
        reinterpret_cast<uint32_t*>(out + y * EWS)[cg] = pack4(o);
    }
}

template <int R>
__device__ __forceinline__ void box_pass_y(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int eh, int ew4, uint32_t ww, uint32_t fw) {
    const int magic = (65536 + ew4 - 1) / ew4;
    for (int item = threadIdx.x; item < eh * ew4; item += 256) {
        int y, cg;
        divmod_small(item, ew4, magic, y, cg);
        uint32_t acc[4] = {0, 0, 0, 0}, far[4] = {0, 0, 0, 0};
#pragma unroll
        for (int d = -R - 1; d <= R + 1; ++d) {
            int q = y + d;
            q = q < 0 ? 0 : (q > eh - 1 ? eh - 1 : q);
            const uint32_t v = reinterpret_cast<const uint32_t*>(in + q * EWS)[cg];
            if (d == -R - 1 || d == R + 1) {
                far[0] += v & 255, far[1] += (v >> 8) & 255, far[2] += (v >> 16) & 255, far[3] += v >> 24;
            } else {
                acc[0] += v & 255, acc[1] += (v >> 8) & 255, acc[2] += (v >> 16) & 255, acc[3] += v >> 24;
            }
        }
        int o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (int)((AUG_UMUL24(acc[i], ww) + AUG_UMUL24(far[i], fw) + (1u << 23)) >> 24);
        reinterpret_cast<uint32_t*>(out + y * EWS)[cg] = pack4(o);
    }
}

template <int R>
__device__ __forceinline__ void box_blur_tile(uint8_t* a, uint8_t* b, int eh, int ew4, uint32_t ww, uint32_t fw) {
    for (int pass = 0; pass < 6; ++pass) {  // BoxBlur.c: three passes along x, then three along y, uint8 after each; ends in `a`
        if (pass < 3) box_pass_x<R>(a, b, eh, ew4, ww, fw);
        else box_pass_y<R>(a, b, eh, ew4, ww, fw);
        __syncthreads();
        uint8_t* t = a;
        a = b;
        b = t;
    }
}

// BLUR = false renders the crops that drew no blur (LDS: the hue tables only, so many workgroups stay resident), BLUR = true the
// blurred ones; each instance leaves the other's crops alone
template <bool BLUR>
__global__ __launch_bounds__(256) void aug_finish_kernel(const int32_t* __restrict__ params, int S, const uint8_t* __restrict__ planes,
                                                          const int* __restrict__ sums, float* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int crop = blockIdx.y;
    const int32_t* p = params + (long)crop * NP;
    const int r1 = p[P_BLUR_R1];
    if (BLUR ? (r1 <= 0 || r1 > BLUR_RMAX + 1) : r1 > 0) return;  // (box radii beyond the halo budget: aug_finish_plane_kernel)
    const int tiles = (S + FT - 1) / FT;
    const int ty = blockIdx.x / tiles, tx = blockIdx.x % tiles;
    const int x0 = tx * FT, y0 = ty * FT, x1 = min(S, x0 + FT), y1 = min(S, y0 + FT);
    const long n = (long)S * S;
    const uint8_t* g = planes + (long)crop * 3 * n;
    float* o = out + (long)crop * 3 * n;
    const int sol = p[P_SOLARIZE];
    const bool colour = has_colour(p);
    const int grey = has_contrast(p) ? contrast_grey(sums[crop], S) : 0;
    const int tid = threadIdx.x;
    HsvTables& tab = *reinterpret_cast<HsvTables*>(smem + (BLUR ? 4 * FPLANE : 0));
    if (colour) {
        fill_hsv_tables(tab);
        __syncthreads();
    }
    const int w4 = (x1 - x0) >> 2, th = y1 - y0;

    if constexpr (!BLUR) {  // no blur: straight through, 4 pixels per lane
        const int magic = (65536 + w4 - 1) / w4;
        for (int item = tid; item < th * w4; item += 256) {
            int y, cg;
            divmod_small(item, w4, magic, y, cg);
            const long i4 = ((long)(y0 + y) * S + x0) / 4 + cg;
            Px4 px = load_px4(g, n, i4);
            if (colour) colour_ops<false>(px, p, grey, tab);
            store_normalized(o + i4 * 4, pack4(px.r), sol, 0.485f, 0.229f);
            store_normalized(o + n + i4 * 4, pack4(px.g), sol, 0.456f, 0.224f);
            store_normalized(o + 2 * n + i4 * 4, pack4(px.b), sol, 0.406f, 0.225f);
        }
        return;
    }
    // blurred crop: tile + halo of 3 passes * (box radius + 1) through the jitter into LDS
    const int halo = 3 * r1;
    const int ex0 = max(0, x0 - halo) & ~3, ex1 = min(S, (x1 + halo + 3) & ~3);
    const int ey0 = max(0, y0 - halo), ey1 = min(S, y1 + halo);
    const int ew4 = (ex1 - ex0) >> 2, eh = ey1 - ey0;
    uint8_t* B = smem + 3 * FPLANE;  // planes 0..2: the channels, plane 3: the other side of the ping-pong
    const int emagic = (65536 + ew4 - 1) / ew4;
    for (int item = tid; item < eh * ew4; item += 256) {
        int y, cg;
        divmod_small(item, ew4, emagic, y, cg);
        const long i4 = ((long)(ey0 + y) * S + ex0) / 4 + cg;
        Px4 px = load_px4(g, n, i4);
        if (colour) colour_ops<false>(px, p, grey, tab);
        reinterpret_cast<uint32_t*>(smem + y * EWS)[cg] = pack4(px.r);
        reinterpret_cast<uint32_t*>(smem + FPLANE + y * EWS)[cg] = pack4(px.g);
        reinterpret_cast<uint32_t*>(smem + 2 * FPLANE + y * EWS)[cg] = pack4(px.b);
    }
    __syncthreads();
    const uint32_t ww = (uint32_t)p[P_BLUR_WW], fw = (uint32_t)p[P_BLUR_FW];
    for (int c = 0; c < 3; ++c) {
        uint8_t* a = smem + c * FPLANE;
        switch (r1 - 1) {
            case 0: box_blur_tile<0>(a, B, eh, ew4, ww, fw); break;
            case 1: box_blur_tile<1>(a, B, eh, ew4, ww, fw); break;
            default: box_blur_tile<2>(a, B, eh, ew4, ww, fw); break;
        }
    }
    const int xo4 = (x0 - ex0) >> 2;
    const int magic = (65536 + w4 - 1) / w4;
    for (int item = tid; item < th * w4; item += 256) {
        int y, cg;
        divmod_small(item, w4, magic, y, cg);
        const long i4 = ((long)(y0 + y) * S + x0) / 4 + cg;
        const int l = (y0 + y - ey0) * EWS;
        store_normalized(o + i4 * 4, reinterpret_cast<const uint32_t*>(smem + l)[xo4 + cg], sol, 0.485f, 0.229f);
        store_normalized(o + n + i4 * 4, reinterpret_cast<const uint32_t*>(smem + FPLANE + l)[xo4 + cg], sol, 0.456f, 0.224f);
        store_normalized(o + 2 * n + i4 * 4, reinterpret_cast<const uint32_t*>(smem + 2 * FPLANE + l)[xo4 + cg], sol, 0.406f, 0.225f);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// finish, whole plane in LDS: any box radius (one workgroup per crop and channel)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PLANE_THREADS = 512;

__global__ __launch_bounds__(PLANE_THREADS) void aug_finish_plane_kernel(const int32_t* __restrict__ params, int S, const uint8_t* __restrict__ planes,
                                                                          const int* __restrict__ sums, float* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int crop = blockIdx.x / 3, ch = blockIdx.x % 3;
    const int32_t* p = params + (long)crop * NP;
    const int r1 = p[P_BLUR_R1];
    if (r1 <= BLUR_RMAX + 1) return;  // rendered by aug_finish_kernel
    const int n = S * S, tid = threadIdx.x;
    const uint8_t* g = planes + (long)crop * 3 * n;
    float* o = out + ((long)crop * 3 + ch) * n;
    const int sol = p[P_SOLARIZE];
    const float mean = ch == 0 ? 0.485f : (ch == 1 ? 0.456f : 0.406f);
    const float stdv = ch == 0 ? 0.229f : (ch == 1 ? 0.224f : 0.225f);
    const bool colour = has_colour(p);
    const int grey = has_contrast(p) ? contrast_grey(sums[crop], S) : 0;
    const int r = r1 - 1;
    const uint32_t ww = (uint32_t)p[P_BLUR_WW], fw = (uint32_t)p[P_BLUR_FW];
    uint8_t* a = smem;
    uint8_t* b = smem + ((n + 15) & ~15);
    HsvTables& tab = *reinterpret_cast<HsvTables*>(smem + 2 * ((n + 15) & ~15));
    fill_hsv_tables(tab);
    __syncthreads();
    for (int i4 = tid; i4 < n / 4; i4 += PLANE_THREADS) {
        Px4 px = load_px4(g, n, i4);
        if (colour) colour_ops<false>(px, p, grey, tab);
        reinterpret_cast<uint32_t*>(a)[i4] = pack4(ch == 0 ? px.r : (ch == 1 ? px.g : px.b));
    }
    __syncthreads();
    for (int pass = 0; pass < 6; ++pass) {
        const bool vertical = pass >= 3;
        for (int i = tid; i < n; i += PLANE_THREADS) {
            const int y = i / S, x = i % S;
            b[i] = vertical ? aug::box_tap(a + x, S, S, y, r, ww, fw) : aug::box_tap(a + y * S, 1, S, x, r, ww, fw);
        }
        __syncthreads();
        uint8_t* t = a;
        a = b;
        b = t;
    }
    for (int i4 = tid; i4 < n / 4; i4 += PLANE_THREADS) store_normalized(o + i4 * 4, reinterpret_cast<const uint32_t*>(a)[i4], sol, mean, stdv);
}

constexpr size_t LDS_MAX = 160 * 1024;

// rows (columns) of the source one tile of TS outputs can need when the axis is resized in_size -> S
int tile_rows(int TS, int in_size, int S) {
    const double scale = (double)in_size / S, support = 2.0 * (scale < 1.0 ? 1.0 : scale);
    return (int)((TS - 1) * scale + 2 * support) + 3;
}
int stage_dwords(int TS, int max_w, int S) { return (tile_rows(TS, max_w, S) * 3 + 3) / 4 + 2; }

size_t resize_lds(int TS, bool staged, int max_h, int max_w, int S) {
    const int KX = aug::resample_ksize(max_w, S), KY = aug::resample_ksize(max_h, S), RMAX = tile_rows(TS, max_h, S);
    return ((size_t)TS * KX + (size_t)TS * KY + 4 * TS + (size_t)RMAX * TS + (staged ? (size_t)RMAX * stage_dwords(TS, max_w, S) : 0)) * 4;
}

template <int TS, bool STAGE>
int launch_resize(const uint8_t* src, const int64_t* images, const int32_t* params, int n, int S, int max_h, int max_w, uint8_t* planes, int* sums,
                  hipStream_t stream) {
    auto kern = aug_resize_kernel<TS, STAGE>;
    const size_t lds = resize_lds(TS, STAGE, max_h, max_w, S);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int tiles = (S + TS - 1) / TS;
    hipLaunchKernelGGL(kern, dim3(tiles * tiles, n), dim3(256), lds, stream, src, images, params, S, aug::resample_ksize(max_w, S),
                       aug::resample_ksize(max_h, S), tile_rows(TS, max_h, S), stage_dwords(TS, max_w, S), planes, sums);
    ESVIT_CHECK_LAUNCH("aug_crops(resize)");
    return ESVIT_OK;
}

}  // namespace

// largest crop box side esvit_aug_crops accepts for output size S (the smallest, unstaged tile must fit its LDS)
int64_t esvit_i_aug_max_box(int S) {
    int64_t lo = S, hi = 1 << 20;
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) / 2;
        if (resize_lds(8, false, (int)mid, (int)mid, S) <= LDS_MAX) lo = mid;
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
    if (S % 4 != 0 || 2 * plane_lds + sizeof(HsvTables) > LDS_MAX) {
        esvit_set_error("esvit_aug_crops: S=%d: the output size must be a multiple of 4 and at most 280", S);
        return ESVIT_ERR_UNSUPPORTED;
    }
    int* sums = reinterpret_cast<int*>(planes + (size_t)n * 3 * S * S);
    int rc = ESVIT_ERR_UNSUPPORTED;
    // 32 x 32 tiles while four workgroups fit a CU's LDS (the phases of a workgroup are serial: residency hides them), else 16 x 16
    if (resize_lds(32, true, max_h, max_w, S) <= LDS_MAX / 4) rc = launch_resize<32, true>(src, images, params, n, S, max_h, max_w, planes, sums, stream);
    else if (resize_lds(16, true, max_h, max_w, S) <= LDS_MAX / 2) rc = launch_resize<16, true>(src, images, params, n, S, max_h, max_w, planes, sums, stream);
    else if (resize_lds(8, true, max_h, max_w, S) <= LDS_MAX) rc = launch_resize<8, true>(src, images, params, n, S, max_h, max_w, planes, sums, stream);
    else if (resize_lds(8, false, max_h, max_w, S) <= LDS_MAX) rc = launch_resize<8, false>(src, images, params, n, S, max_h, max_w, planes, sums, stream);
    else esvit_set_error("esvit_aug_crops: crop box %dx%d -> %d is beyond esvit_query(ESVIT_Q_AUG_MAX_BOX)", max_h, max_w, S);
    if (rc != ESVIT_OK) return rc;
    hipLaunchKernelGGL(aug_mean_kernel, dim3((S * S + MEAN_PX - 1) / MEAN_PX, n), dim3(256), 0, stream, params, S, planes, sums);
    ESVIT_CHECK_LAUNCH("aug_crops(mean)");
    {
        const int tiles = (S + FT - 1) / FT;
        hipLaunchKernelGGL(aug_finish_kernel<false>, dim3(tiles * tiles, n), dim3(256), sizeof(HsvTables), stream, params, S, planes, sums, out);
        ESVIT_CHECK_LAUNCH("aug_crops(finish)");
        hipLaunchKernelGGL(aug_finish_kernel<true>, dim3(tiles * tiles, n), dim3(256), 4 * FPLANE + sizeof(HsvTables), stream, params, S, planes, sums, out);
        ESVIT_CHECK_LAUNCH("aug_crops(finish, blurred)");
    }
    {
        auto kern = aug_finish_plane_kernel;
        const size_t lds = 2 * plane_lds + sizeof(HsvTables);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(3 * n), dim3(PLANE_THREADS), lds, stream, params, S, planes, sums, out);
        ESVIT_CHECK_LAUNCH("aug_crops(finish, whole plane)");
    }
    return ESVIT_OK;
}
