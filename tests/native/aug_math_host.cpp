// Test harness: compiles the product's scalar arithmetic (esvit_amd/csrc/augment_math.h) for the HOST so that
// tests/test_augment_cpu.py can check it against the oracle / Pillow without a GPU.  Not part of the library.
#define AUG_HD static inline
#include "augment_math.h"

extern "C" {
int aug_t_ksize(int in_size, int out_size) { return aug::resample_ksize(in_size, out_size); }
void aug_t_coeffs(int in_size, int out_size, int kmax, int32_t* bounds, int32_t* kk) {
    for (int xx = 0; xx < out_size; ++xx) {
        for (int i = 0; i < kmax; ++i) kk[xx * kmax + i] = 0;
        aug::resample_row(in_size, out_size, xx, kmax, bounds + 2 * xx, bounds + 2 * xx + 1, kk + xx * kmax);
    }
}
void aug_t_blend(const uint8_t* deg, const uint8_t* img, long n, float alpha, uint8_t* out) {
    for (long i = 0; i < n; ++i) out[i] = (uint8_t)aug::blend(deg[i], img[i], alpha);
}
void aug_t_rgb_to_hsv(const uint8_t* in, long n, uint8_t* out) {
    for (long i = 0; i < n; ++i) {
        int h, s, v;
        aug::rgb_to_hsv(in[3 * i], in[3 * i + 1], in[3 * i + 2], &h, &s, &v);
        out[3 * i] = (uint8_t)h, out[3 * i + 1] = (uint8_t)s, out[3 * i + 2] = (uint8_t)v;
    }
}
void aug_t_hsv_to_rgb(const uint8_t* in, long n, uint8_t* out) {
    for (long i = 0; i < n; ++i) {
        int r, g, b;
        aug::hsv_to_rgb(in[3 * i], in[3 * i + 1], in[3 * i + 2], &r, &g, &b);
        out[3 * i] = (uint8_t)r, out[3 * i + 1] = (uint8_t)g, out[3 * i + 2] = (uint8_t)b;
    }
}
void aug_t_to_l(const uint8_t* in, long n, uint8_t* out) {
    for (long i = 0; i < n; ++i) out[i] = (uint8_t)aug::rgb_to_l(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
}
void aug_t_box_line(const uint8_t* line, int n, int r, uint32_t ww, uint32_t fw, uint8_t* out) {
    for (int x = 0; x < n; ++x) out[x] = aug::box_tap(line, 1, n, x, r, ww, fw);
}
void aug_t_normalize(const uint8_t* in, long n, float mean, float stdv, float* out) {
    for (long i = 0; i < n; ++i) out[i] = aug::normalize(in[i], mean, stdv);
}
}
