// Library-level plumbing: version, error string, and the host-side integer index maps
// (relative-position index, pad->roll->partition window maps, shift mask).  The index maps are
// pure integer arithmetic restated from SURVEY.md Appendix A1 and are checked bit-exactly
// against the reference tensors in tests/ (they need no GPU).
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/esvit_hip.h"

static thread_local char g_err[512] = "";

void esvit_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int esvit_version(void) { return 200; }
extern "C" const char* esvit_last_error(void) { return g_err; }

// scratch sizes / capabilities: answered by the translation unit that owns the kernel
int esvit_i_attn_frag_elems(int N);
int esvit_i_attn_lse_elems(int N);
int esvit_i_attn_bwd_parts(int N, int Bw, int nH);
int esvit_i_attn_bwd_pad_rows(int dtype, int N, int Bw, int nH);
int esvit_i_ln_bwd_blocks(long rows, int C);
int esvit_i_colsum_blocks(long rows);
int esvit_i_col_reduce_blocks(long rows);
int esvit_i_update_chunk_elems();
int esvit_i_mlp_fused_supported(int dtype, int C);
int64_t esvit_i_aug_max_box(int S);

extern "C" int64_t esvit_query(int what, int64_t a, int64_t b, int64_t c) {
    switch (what) {
        case ESVIT_Q_ATTN_FRAG_ELEMS: return esvit_i_attn_frag_elems((int)a);
        case ESVIT_Q_ATTN_LSE_ELEMS: return esvit_i_attn_lse_elems((int)a);
        case ESVIT_Q_ATTN_BWD_PARTS: return esvit_i_attn_bwd_parts((int)a, (int)b, (int)c);
        case ESVIT_Q_ATTN_BWD_PAD_ROWS: return esvit_i_attn_bwd_pad_rows((int)(c >> 32), (int)a, (int)b, (int)(c & 0xffffffff));
        case ESVIT_Q_LN_BWD_BLOCKS: return esvit_i_ln_bwd_blocks((long)a, (int)b);
        case ESVIT_Q_COLSUM_BLOCKS: return esvit_i_colsum_blocks((long)a);
        case ESVIT_Q_COL_REDUCE_BLOCKS: return esvit_i_col_reduce_blocks((long)a);
        case ESVIT_Q_UPDATE_CHUNK_ELEMS: return esvit_i_update_chunk_elems();
        case ESVIT_Q_MLP_FUSED: return esvit_i_mlp_fused_supported((int)a, (int)b);
        case ESVIT_Q_AUG_MAX_BOX: return a > 0 ? esvit_i_aug_max_box((int)a) : 0;
    }
    esvit_set_error("esvit_query: unknown question %d", what);
    return ESVIT_ERR_ARG;
}

// swin_transformer.py:100-109: idx[p,q] = (ph-qh+ws-1)*(2ws-1) + (pw-qw+ws-1)
extern "C" int esvit_relative_position_index(int ws, int64_t* out) {
    if (ws <= 0 || !out) {
        esvit_set_error("esvit_relative_position_index: bad args");
        return ESVIT_ERR_ARG;
    }
    const int N = ws * ws;
    for (int p = 0; p < N; ++p)
        for (int q = 0; q < N; ++q) {
            const int ph = p / ws, pw = p % ws, qh = q / ws, qw = q % ws;
            out[(int64_t)p * N + q] = (int64_t)(ph - qh + ws - 1) * (2 * ws - 1) + (pw - qw + ws - 1);
        }
    return ESVIT_OK;
}

// swin_transformer.py:286-325.  Padded grid Hp x Wp (zero rows/cols appended bottom/right),
// rolled by -shift: rolled(i,j) = padded((i+s)%Hp, (j+s)%Wp); window id (i/ws)*(Wp/ws)+(j/ws),
// slot (i%ws)*ws + (j%ws).
extern "C" int esvit_window_maps(int H, int W, int ws, int shift, int32_t* win2tok, int32_t* tok2win, int32_t* region_ids) {
    if (H <= 0 || W <= 0 || ws <= 0 || shift < 0 || shift >= ws) {
        esvit_set_error("esvit_window_maps: bad geometry H=%d W=%d ws=%d shift=%d", H, W, ws, shift);
        return ESVIT_ERR_ARG;
    }
    if (region_ids && shift == 0) {
        esvit_set_error("esvit_window_maps: region ids exist for shifted windows only");
        return ESVIT_ERR_ARG;
    }
    const int Hp = (H + ws - 1) / ws * ws, Wp = (W + ws - 1) / ws * ws;
    const int nWw = Wp / ws, N = ws * ws;
    // swin_transformer.py:249-272: region ids on the padded grid in rolled coordinates via the three python slices
    // (0,-ws), (-ws,-shift), (-shift,None) applied in order; id = 3*band(i)+band(j).  The shift mask is 0 where two
    // slots of a window share an id.
    auto band = [&](int t, int L) { return t < L - ws ? 0 : (t < L - shift ? 1 : 2); };
    for (int i = 0; i < Hp; ++i)
        for (int j = 0; j < Wp; ++j) {
            const int si = (i + shift) % Hp, sj = (j + shift) % Wp;  // source coordinate in the padded grid
            const int slot = ((i / ws) * nWw + (j / ws)) * N + (i % ws) * ws + (j % ws);
            const bool real = si < H && sj < W;
            if (win2tok) win2tok[slot] = real ? si * W + sj : -1;
            if (tok2win && real) tok2win[si * W + sj] = slot;
            if (region_ids) region_ids[slot] = 3 * band(i, Hp) + band(j, Wp);
        }
    return ESVIT_OK;
}

// swin_transformer.py:249-272
extern "C" int esvit_shift_mask(int H, int W, int ws, int shift, float* mask, int* n_windows) {
    if (H <= 0 || W <= 0 || ws <= 0 || shift <= 0 || shift >= ws || !mask) {
        esvit_set_error("esvit_shift_mask: bad geometry H=%d W=%d ws=%d shift=%d", H, W, ws, shift);
        return ESVIT_ERR_ARG;
    }
    const int Hp = (H + ws - 1) / ws * ws, Wp = (W + ws - 1) / ws * ws;
    const int nWw = Wp / ws, nW = (Hp / ws) * nWw, N = ws * ws;
    // python slice(a, b) on a length-L axis with negative a/b -> [L+a, L+b); an empty slice writes nothing
    auto band = [&](int t, int L) {
        int r = -1;
        if (t >= 0 && t < L - ws) r = 0;            // slice(0, -ws)
        if (t >= L - ws && t < L - shift) r = 1;    // slice(-ws, -shift)
        if (t >= L - shift) r = 2;                  // slice(-shift, None)
        return r;
    };
    // region id per window slot
    int* ids = new int[(size_t)nW * N];
    for (int i = 0; i < Hp; ++i)
        for (int j = 0; j < Wp; ++j) {
            const int rh = band(i, Hp), rw = band(j, Wp);
            // cnt increments over (h, w) in order 0..8; untouched cells keep 0 (cannot happen: bands cover the axis)
            const int id = 3 * rh + rw;
            ids[((i / ws) * nWw + (j / ws)) * N + (i % ws) * ws + (j % ws)] = id;
        }
    for (int w = 0; w < nW; ++w)
        for (int p = 0; p < N; ++p)
            for (int q = 0; q < N; ++q)
                mask[((size_t)w * N + p) * N + q] = (ids[w * N + p] == ids[w * N + q]) ? 0.f : -100.f;
    delete[] ids;
    if (n_windows) *n_windows = nW;
    return ESVIT_OK;
}
