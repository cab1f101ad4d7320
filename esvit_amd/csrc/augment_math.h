// Scalar arithmetic of the crop producer (datasets/build.py:203-261 through Pillow's libImaging; see augment.hip).
// Every function reproduces one Pillow routine BIT FOR BIT, which pins the evaluation order, the float / double width of each
// intermediate and the absence of fused multiply-adds -- hence the pragma.  The functions are plain C++ (AUG_HD expands to
// __host__ __device__ under hipcc and to nothing under a host compiler), so tests/test_augment_cpu.py can compile this very
// header with g++ and check it against the oracle and against Pillow before any kernel runs.
#pragma once
#include <math.h>
#include <stdint.h>

#ifndef AUG_HD
#define AUG_HD __host__ __device__ __forceinline__
#endif

#pragma clang fp contract(off)

// products of values known to fit 24 bits: one full-rate instruction on the GPU (a 32-bit integer multiply is quarter rate)
#ifdef __HIP_DEVICE_COMPILE__
#define AUG_MUL24(a, b) __mul24((int)(a), (int)(b))
#define AUG_UMUL24(a, b) __umul24((unsigned)(a), (unsigned)(b))
#else
#define AUG_MUL24(a, b) ((int)(a) * (int)(b))
#define AUG_UMUL24(a, b) ((unsigned)(a) * (unsigned)(b))
#endif

namespace aug {

constexpr int PRECISION_BITS = 32 - 8 - 2;  // Resample.c: 8 bpc coefficients are 22-bit fixed point

// Resample.c bicubic_filter, a = -0.5
AUG_HD double bicubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// number of taps a row of coefficients can hold: Resample.c precompute_coeffs, ksize
AUG_HD int resample_ksize(int in_size, int out_size) {
    double filterscale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int)ceil(2.0 * filterscale) * 2 + 1;
}

// Resample.c precompute_coeffs + normalize_coeffs_8bpc for output position xx of an axis resized in_size -> out_size (the box
// is the whole axis): first tap, number of taps (<= kmax, the caller's row length) and the fixed-point taps
AUG_HD void resample_row(int in_size, int out_size, int xx, int kmax, int* first, int* count, int32_t* k) {
    const double scale = (double)in_size / out_size;
    double filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const double center = (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > kmax) xmax = kmax;  // cannot happen when kmax >= resample_ksize(in_size, out_size)
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += bicubic((x + xmin - center + 0.5) * ss);
    for (int x = 0; x < xmax; ++x) {
        double w = bicubic((x + xmin - center + 0.5) * ss);
        if (ww != 0.0) w /= ww;
        k[x] = w < 0 ? (int32_t)(-0.5 + w * (1 << PRECISION_BITS)) : (int32_t)(0.5 + w * (1 << PRECISION_BITS));
    }
    *first = xmin;
    *count = xmax;
}

AUG_HD uint8_t clip8(int32_t ss) {  // Resample.c clip8: arithmetic shift, clamp
    const int32_t v = ss >> PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// Convert.c rgb2l
AUG_HD int rgb_to_l(int r, int g, int b) { return (AUG_MUL24(r, 19595) + AUG_MUL24(g, 38470) + AUG_MUL24(b, 7471) + 0x8000) >> 16; }

// Blend.c ImagingBlend, one band: float product, float sum, truncation; clipped only when extrapolating
AUG_HD int blend(int deg, int img, float alpha) {
    const float prod = alpha * (float)(img - deg);
    const float t = (float)deg + prod;
    if (alpha >= 0.f && alpha <= 1.0f) return (int)t & 255;
    if (t <= 0.0f) return 0;
    if (t >= 255.0f) return 255;
    return (int)t;
}

// Convert.c rgb2hsv_row (float quotients, the "2.0 + rc - bc" forms are double expressions stored to a float).  Of rc, gc, bc only
// the two the branch uses are divided.
AUG_HD void rgb_to_hsv(int r, int g, int b, int* ph, int* ps, int* pv) {
    const int maxc = r > g ? (r > b ? r : b) : (g > b ? g : b);
    const int minc = r < g ? (r < b ? r : b) : (g < b ? g : b);
    *pv = maxc;
    if (minc == maxc) {
        *ph = 0;
        *ps = 0;
        return;
    }
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    // h = bc - gc | 2.0 + rc - bc | 4.0 + gc - rc  ==  base + plus - minus
    const int sel = r == maxc ? 0 : (g == maxc ? 1 : 2);
    const int nplus = sel == 0 ? maxc - b : (sel == 1 ? maxc - r : maxc - g);
    const int nminus = sel == 0 ? maxc - g : (sel == 1 ? maxc - b : maxc - r);
    const float plus = (float)nplus / cr, minus = (float)nminus / cr;
    float h;
    if (sel == 0) h = plus - minus;
    else h = (float)((sel == 1 ? 2.0 : 4.0) + (double)plus - (double)minus);
    const double t = (double)h / 6.0 + 1.0;  // in [5/6, 11/6]: fmod(t, 1.0) = t - floor(t), exact
    h = (float)(t - floor(t));
    int uh = (int)((double)h * 255.0), us = (int)((double)s * 255.0);
    *ph = uh < 0 ? 0 : (uh > 255 ? 255 : uh);
    *ps = us < 0 ? 0 : (us > 255 ? 255 : us);
}

// Convert.c hsv2rgb.  Its sector index i = floor(h * 6 / 255), fraction f and saturation fs = s / 255 depend on one byte each:
// hsv_sector / hsv_saturation are what a 256-entry table holds (the kernels keep them in LDS), hsv_to_rgb_t is the rest.
AUG_HD void hsv_sector(int h, int* i, float* f) {
    const double hf = (double)(float)h * 6.0 / 255.0;
    *i = (int)floor(hf);
    *f = (float)(hf - (double)(float)*i);
}
AUG_HD float hsv_saturation(int s) { return (float)((double)(float)s / 255.0); }

AUG_HD void hsv_to_rgb_t(int s, int v, int i, float f, float fs, int* pr, int* pg, int* pb) {
    if (s == 0) {
        *pr = *pg = *pb = v;
        return;
    }
    const double vf = (double)(float)v, f64 = (double)f, fs64 = (double)fs;
    int p = (int)floor(vf * (1.0 - fs64) + 0.5);  // C round() of a non-negative value
    int q = (int)floor(vf * (1.0 - fs64 * f64) + 0.5);
    int t = (int)floor(vf * (1.0 - fs64 * (1.0 - f64)) + 0.5);
    p = p < 0 ? 0 : (p > 255 ? 255 : p);
    q = q < 0 ? 0 : (q > 255 ? 255 : q);
    t = t < 0 ? 0 : (t > 255 ? 255 : t);
    switch (i % 6) {
        case 0: *pr = v; *pg = t; *pb = p; break;
        case 1: *pr = q; *pg = v; *pb = p; break;
        case 2: *pr = p; *pg = v; *pb = t; break;
        case 3: *pr = p; *pg = q; *pb = v; break;
        case 4: *pr = t; *pg = p; *pb = v; break;
        default: *pr = v; *pg = p; *pb = q; break;
    }
}

AUG_HD void hsv_to_rgb(int h, int s, int v, int* pr, int* pg, int* pb) {
    int i;
    float f;
    hsv_sector(h, &i, &f);
    hsv_to_rgb_t(s, v, i, f, hsv_saturation(s), pr, pg, pb);
}

// one output of BoxBlur.c ImagingLineBoxBlur8 at position x of a line of n values `stride` bytes apart: 2r + 1 full taps of weight
// ww and the two far taps of weight fw (24-bit fixed point; the host derives r, ww, fw from the Gaussian radius), indices clamped
AUG_HD uint8_t box_tap(const uint8_t* line, int stride, int n, int x, int r, uint32_t ww, uint32_t fw) {
    uint32_t acc = 0;
    for (int d = -r; d <= r; ++d) {
        int i = x + d;
        i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
        acc += line[i * stride];
    }
    int lo = x - r - 1, hi = x + r + 1;
    lo = lo < 0 ? 0 : lo;
    hi = hi > n - 1 ? n - 1 : hi;
    const uint32_t bulk = AUG_UMUL24(acc, ww) + AUG_UMUL24((uint32_t)line[lo * stride] + (uint32_t)line[hi * stride], fw);  // ww, fw < 2^24
    return (uint8_t)((bulk + (1u << 23)) >> 24);
}

// ToTensor + Normalize (datasets/build.py:213-216): u8 / 255 in float, (x - mean) / std in float
AUG_HD float normalize(int v, float mean, float stdv) { return ((float)v / 255.0f - mean) / stdv; }

}  // namespace aug
