// esvit_gemm: shape -> kernel dispatch of the MFMA GEMM family (kernels: gemm_kernels.h).
//
// Main loops (one fragment / epilogue convention; esvit_gemm_desc.kernel forces one, 0 = the rules below):
//   ESVIT_GEMM_REGSTAGE  register-staged, 128-row tiles, 4 waves: the exact-fp32 parity mode (v_mfma_f32_16x16x4_f32) only;
//   ESVIT_GEMM_DMA4      LDS-DMA, 128 x {64, 96, 128} tiles as 2 x 2 waves, two workgroups per CU: the default;
//   ESVIT_GEMM_DMA4W     the same loop with whole-width wave rows -- 128 x 192 as 2 x 2 waves of 64 x 96, 128 x 96 as 4 x 1
//                        waves of 32 x 96 -- for the 96 * 2^s wide backbone: 192-byte instead of 96-byte output row pieces
//                        and fewer operand bytes per FLOP; chosen per epilogue kind from profiles/r02_gemm_kernels_b128.jsonl;
//   ESVIT_GEMM_DMA8      LDS-DMA, 256 x 256 tiles, 8 waves, one workgroup per CU: very long reductions only;
//   ESVIT_GEMM_P8        the 256 x 256 eight-phase loop (gemm_p8.hip): counted DMA waits that never drain the queue, staggered wave
//                        halves, accumulators in AccVGPRs -- K % 64 == 0, no row map; row statistics on whole tiles.
// The choice is a pure function of the descriptor (esvit_gemm_select).
#include "gemm_kernels.h"

namespace {

struct GemmChoice {
    int kernel, bm, bn;
};

// tile of the 8-wave kernel
inline void dma8_tile(const esvit_gemm_desc&, int& bm, int& bn) {
    bm = 256;
    bn = 256;
}

// ESVIT_GEMM_DMA4W: 128 x 192 where N is a multiple of 192, 128 x 96 (4 x 1 waves) for the other multiples of 96
inline void dma4w_tile(const esvit_gemm_desc& d, int& bm, int& bn) {
    bm = 128;
    bn = (d.N % 192 == 0) ? 192 : 96;
}

inline void dma4_tile(const esvit_gemm_desc& d, int& bm, int& bn) {
    bm = 128;
    bn = ((d.N % 96 == 0) && (d.N % 128 != 0)) ? 96 : (d.N <= 64 ? 64 : 128);
}

// Every workgroup of a launch runs about equally long, so a launch takes ceil(workgroups / resident slots) rounds.
inline double round_efficiency(long wgs, long slots) {
    const long rounds = (wgs + slots - 1) / slots;
    return (double)wgs / (double)(rounds * slots);
}

// what the eight-phase loop can run: whole 64-deep k-tiles (its DMA has no per-lane k predicate), 32-bit DMA offsets, and none of
// the row map / row statistics extras that live in the 128-row kernels
bool p8_supports(int dtype, const esvit_gemm_desc& d) {
    if (dtype != ESVIT_BF16 || d.K % 64 != 0 || d.rowmap) return false;
    if (d.rowstat && (d.M % 256 != 0 || d.N % 256 != 0 || d.colstat)) return false;  // row statistics: whole 256 x 256 tiles only (32-column blocks), no column sums
    if (d.a_kstrided && !d.b_kstrided) return false;
    const long a_bytes = (d.a_kstrided ? (long)d.K : (long)d.M) * d.lda * 2, b_bytes = (d.b_kstrided ? (long)d.K : (long)d.N) * d.ldb * 2;
    return a_bytes < 0xfff00000L && b_bytes < 0xfff00000L;
}

GemmChoice choose(int dtype, const esvit_gemm_desc& d) {
    GemmChoice c{ESVIT_GEMM_REGSTAGE, 128, 128};
    if (dtype != ESVIT_BF16) {
        dma4_tile(d, c.bm, c.bn);
        return c;
    }
    int want = d.kernel;
    if (d.rowstat) {  // the statistics epilogue: the 128 x 128 tile (2 x 2 waves of 64 x 64, 64-column blocks); on request the 256 x 256
                      // eight-phase loop (32-column blocks, whole tiles only)
#ifdef ESVIT_P8_ROWSTAT_AUTO  // tools/ab_routing.sh only
        const bool p8 = (want == ESVIT_GEMM_P8 || (want == ESVIT_GEMM_AUTO && (long)(d.M / 256) * (d.N / 256) >= 1024)) && p8_supports(dtype, d);
#else
        const bool p8 = want == ESVIT_GEMM_P8 && p8_supports(dtype, d);
#endif
        c.kernel = p8 ? ESVIT_GEMM_P8 : ESVIT_GEMM_DMA4;
        c.bm = c.bn = p8 ? 256 : 128;
        return c;
    }
    if (want == ESVIT_GEMM_AUTO) {
        want = ESVIT_GEMM_DMA4;
        const int nz = d.splitk > 1 ? d.splitk : (d.batch > 1 ? d.batch : 1);
        const long t8 = (long)ceil_div(d.M, 256) * ceil_div(d.N, 256);
        const bool gelu_bwd_ = d.epilogue == ESVIT_EPI_GELU_BWD || d.epilogue == ESVIT_EPI_QGELU_BWD;
#ifdef ESVIT_NO_P8_ROUTING  // tools/ab_routing.sh only: the round-3 choice, for a same-box A/B of the rules below
        if (false) {
#else
        // The eight-phase loop (staggered schedule) is taken where the PER-LAUNCH IN-STEP trace shows a gain (tools/trace_ab.sh,
        // profiles/r04_gemm_instep_ab.txt, section 5): long reductions over whole 256-column tiles -- stage-3 / head data gradients
        // -12..-16 %, stage-3 fc2 forward -25 %, head fc2 forward -6..-9 %, the split-K data gradient of the last layer -6 %, its
        // weight gradient -11 %.  NOT taken: 384-wide data gradients on the 256 x 128 variant (+11 % in the step), and anything
        // with a short reduction (the workgroup has no partner to hide its epilogue behind).
#ifdef ESVIT_P8_K768  // tools/trace_two_libs.sh: the K = 768 forwards of stage 3 and of the head's first layer (N >= 2048) too
        const bool k_ok = d.K >= 2048 || (d.K >= 768 && d.N >= 2048 && !d.b_kstrided);
#else
        const bool k_ok = d.K >= 2048;
#endif
        if (p8_supports(dtype, d) && d.batch <= 1 && !d.a_kstrided && k_ok && d.N % 256 == 0 && d.M >= 4096 && !gelu_bwd_ &&
            (d.N >= 768 || d.splitk > 1)) {
            want = ESVIT_GEMM_P8;
        } else if (p8_supports(dtype, d) && d.batch <= 1 && d.a_kstrided && d.M % 256 == 0 && d.N % 256 == 0 && (long)d.M * d.N >= 2000000L
#ifndef ESVIT_P8_WGRAD_COLSUM  // tools/ab_routing.sh: weight gradients with the fused bias gradient too
                   && !d.colsum
#endif
        ) {
            want = ESVIT_GEMM_P8;
#ifdef ESVIT_P8_WGRAD_STAGE2  // round-6 A/B: the stage-2 weight gradients (384 / 1152 / 1536-wide, K = 87040) on ragged 256 x 256 tiles, bias gradient fused
        } else if (p8_supports(dtype, d) && d.batch <= 1 && d.a_kstrided && (long)d.M * d.N >= 400000L && d.K >= 16384 && d.M % 128 == 0 && d.N % 128 == 0) {
            want = ESVIT_GEMM_P8;
#endif
#endif
        } else if (!d.a_kstrided && !d.rowmap && d.K >= 4096 && d.N >= 192 && t8 * nz >= 128) {
            // One 256 x 256 tile per CU has no second workgroup to hide its prologue / epilogue behind: measured
            // (profiles/r02_gemm_kernels_b128_first.jsonl) it wins only where the main loop is very long -- the dgrad of
            // the 65536-wide last layer (K = out_dim, split-K): 1266 -> 865 us.  Everything else, the weight gradients
            // included, is faster with two 4-wave workgroups per CU.
            want = ESVIT_GEMM_DMA8;
        } else if (d.N % 96 == 0 && !d.rowmap && d.batch <= 1) {
            // whole-width wave rows (profiles/r02_gemm_kernels_b128.jsonl, per epilogue kind):
            int bm4, bn4, bmw, bnw;
            dma4_tile(d, bm4, bn4);
            dma4w_tile(d, bmw, bnw);
            const double e4 = round_efficiency((long)ceil_div(d.M, bm4) * ceil_div(d.N, bn4) * nz, 512);
            const double ew = round_efficiency((long)ceil_div(d.M, bmw) * ceil_div(d.N, bnw) * nz, 512);
            const bool gelu = d.epilogue == ESVIT_EPI_GELU || d.epilogue == ESVIT_EPI_QGELU;
            const bool gelu_bwd = d.epilogue == ESVIT_EPI_GELU_BWD || d.epilogue == ESVIT_EPI_QGELU_BWD;
            // (rules re-measured with the early-release DMA4 loop: profiles/r02_gemm_kernels_b128_b.jsonl)
            if (d.a_kstrided)  // weight gradients: the 4 x 1 layout for 96-wide outputs, 128 x 192 where it halves the column tiles
                want = (bnw == 96 || d.N == 192 || (d.N >= 768 && (long)d.M * d.N >= 1000000L)) ? ESVIT_GEMM_DMA4W : ESVIT_GEMM_DMA4;
            else if (gelu_bwd) want = ESVIT_GEMM_DMA4;                                  // -6..-18 %
            else if (d.residual) want = bnw == 96 ? ESVIT_GEMM_DMA4W : ESVIT_GEMM_DMA4;  // 4 x 1 waves +4..7 %, 128 x 192 -8..-30 %
            else if (gelu) want = (bnw == 192 && d.N <= 768) ? ESVIT_GEMM_DMA4W : ESVIT_GEMM_DMA4;
            else if (d.b_kstrided) want = d.N <= 192 ? ESVIT_GEMM_DMA4W : ESVIT_GEMM_DMA4;  // plain dgrads: early release wins from N = 384 up (+10..35 %)
            else want = ew >= 0.9 * e4 ? ESVIT_GEMM_DMA4W : ESVIT_GEMM_DMA4;             // plain / bias forward: +5..25 % unless the wider tile quantises worse
        }
    }
    if (want == ESVIT_GEMM_DMA4W && !(d.N % 96 == 0)) want = ESVIT_GEMM_DMA4;
#ifdef ESVIT_NO_TILE192  // tools/trace_two_libs.sh: what the launches on the (spilling) 128 x 192 instance cost on the 128 x 128 / 128 x 96 ones
    if (want == ESVIT_GEMM_DMA4W && d.N % 192 == 0 && d.kernel == ESVIT_GEMM_AUTO) want = ESVIT_GEMM_DMA4;
#endif
    c.kernel = want;
    if (want == ESVIT_GEMM_P8) c.bm = c.bn = 256;
    else if (want == ESVIT_GEMM_DMA8) dma8_tile(d, c.bm, c.bn);
    else if (want == ESVIT_GEMM_DMA4W) dma4w_tile(d, c.bm, c.bn);
    else dma4_tile(d, c.bm, c.bn);
    return c;
}

template <bool AKS, bool BKS>
int run_dma4(const esvit_gemm_desc& d, int bn, hipStream_t stream) {
    // early buffer release (gemm_kernels.h): +6 % over the plain two-buffer loop summed over the step's shapes, up to +20 % on
    // the dgrad / wgrad layouts (profiles/r02_gemm_early_release.jsonl)
    if (bn == 96) return launch_gemm_dma<AKS, BKS, 128, 96, 64, 2, 2, 2, 2, true>(d, stream);
    if (bn == 64) return launch_gemm_dma<AKS, BKS, 128, 64, 64, 2, 2, 2, 2, true>(d, stream);
    return launch_gemm_dma<AKS, BKS, 128, 128, 64, 2, 2, 2, 2, true>(d, stream);
}

// 8 waves: 256 x 256 as 4 x 2 waves of 64 x 128 (measured faster than 2 x 4 waves of 128 x 64 on every layout)
template <bool AKS, bool BKS>
int run_dma8(const esvit_gemm_desc& d, hipStream_t stream) {
    return launch_gemm_dma<AKS, BKS, 256, 256, 64, 2, 4, 2>(d, stream);
}

template <bool AKS, bool BKS>
int run_dma4w(const esvit_gemm_desc& d, int bn, hipStream_t stream) {
    if (bn == 192) return launch_gemm_dma<AKS, BKS, 128, 192, 64, 2, 2, 2>(d, stream);
    return launch_gemm_dma<AKS, BKS, 128, 96, 64, 2, 4, 1>(d, stream);
}

template <typename T, bool AKS, bool BKS>
int run_regstage(const esvit_gemm_desc& d, int bn, hipStream_t stream) {
    if (bn == 96) return launch_gemm<T, AKS, BKS, 128, 96, true>(d, stream);
    if (bn == 64) return launch_gemm<T, AKS, BKS, 128, 64, true>(d, stream);
    return launch_gemm<T, AKS, BKS, 128, 128, true>(d, stream);
}

template <bool AKS, bool BKS>
int run_layout(int dtype, const esvit_gemm_desc& d, const GemmChoice& c, hipStream_t stream) {
    if (dtype == ESVIT_BF16) {
        if (c.kernel == ESVIT_GEMM_P8) return esvit_gemm_p8_launch(d, stream);
        if constexpr (!AKS) {  // the 8-wave tile is not instantiated for the weight-gradient layout (measured slower there)
            if (c.kernel == ESVIT_GEMM_DMA8) return run_dma8<AKS, BKS>(d, stream);
        }
        if (c.kernel == ESVIT_GEMM_DMA4W) return run_dma4w<AKS, BKS>(d, c.bn, stream);
        return run_dma4<AKS, BKS>(d, c.bn, stream);
    }
    return run_regstage<float, AKS, BKS>(d, c.bn, stream);
}

// which forced main loops exist for which problem
int check_selector(int dtype, const esvit_gemm_desc& d) {
    ESVIT_CHECK_ARG(d.kernel >= ESVIT_GEMM_AUTO && d.kernel <= ESVIT_GEMM_P8, "esvit_gemm: bad kernel selector %d", d.kernel);
    ESVIT_CHECK_ARG(!(d.kernel == ESVIT_GEMM_P8 && !p8_supports(dtype, d)),
                    "esvit_gemm: the eight-phase loop needs bf16, K %% 64 == 0, no row map, and row statistics only over whole 256 x 256 tiles without column sums");
    if (dtype != ESVIT_BF16)
        ESVIT_CHECK_ARG(d.kernel == ESVIT_GEMM_AUTO || d.kernel == ESVIT_GEMM_REGSTAGE, "esvit_gemm: fp32 runs on the register-staged loop only");
    else
        ESVIT_CHECK_ARG(d.kernel != ESVIT_GEMM_REGSTAGE, "esvit_gemm: the register-staged loop is the fp32 parity mode; bf16 runs on the LDS-DMA loops");
    ESVIT_CHECK_ARG(!(d.kernel == ESVIT_GEMM_DMA8 && d.a_kstrided), "esvit_gemm: the 8-wave tile does not exist for the weight-gradient layout");
    return ESVIT_OK;
}

int validate(int dtype, esvit_gemm_desc& d) {
    ESVIT_CHECK_ARG(d.A && d.B && d.C, "esvit_gemm: null operand");
    ESVIT_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0, "esvit_gemm: bad shape M=%d N=%d K=%d", d.M, d.N, d.K);
    const int vec = dtype == ESVIT_BF16 ? 8 : 4;
    ESVIT_CHECK_ARG(dtype == ESVIT_BF16 || dtype == ESVIT_F32, "esvit_gemm: bad dtype %d", dtype);
    ESVIT_CHECK_ARG(d.lda % vec == 0 && d.ldb % vec == 0, "esvit_gemm: lda/ldb must be multiples of %d", vec);
    if (!d.a_kstrided) ESVIT_CHECK_ARG(d.K % vec == 0, "esvit_gemm: K=%d must be a multiple of %d", d.K, vec);
    if (d.a_kstrided) ESVIT_CHECK_ARG(d.M % vec == 0, "esvit_gemm: M=%d must be a multiple of %d (k-strided A)", d.M, vec);
    if (d.b_kstrided) ESVIT_CHECK_ARG(d.N % vec == 0, "esvit_gemm: N=%d must be a multiple of %d (k-strided B)", d.N, vec);
    ESVIT_CHECK_ARG(((uintptr_t)d.A % 16 == 0) && ((uintptr_t)d.B % 16 == 0), "esvit_gemm: operands must be 16-byte aligned");
    ESVIT_CHECK_ARG(!(d.a_kstrided && !d.b_kstrided), "esvit_gemm: layout a_kstrided=1,b_kstrided=0 is not used on the path");
    if (d.batch < 1) d.batch = 1;
    if (d.splitk > 1) {
        ESVIT_CHECK_ARG(d.batch == 1 && d.partial && d.ldc == d.N, "esvit_gemm: split-K needs batch=1, a workspace and a dense C");
        ESVIT_CHECK_ARG(!d.bias && !d.residual && !d.rowmap && d.epilogue == 0, "esvit_gemm: split-K has no fused epilogue");
    } else {
        d.splitk = 1;
    }
    if (d.rowmap) ESVIT_CHECK_ARG(d.rowmap_period > 0 && d.rowmap_tokens > 0, "esvit_gemm: bad rowmap geometry");
    if (d.rowscale) ESVIT_CHECK_ARG(d.rows_per_sample > 0, "esvit_gemm: rowscale needs rows_per_sample");
    if (d.epilogue == ESVIT_EPI_GELU_BWD || d.epilogue == ESVIT_EPI_QGELU_BWD) ESVIT_CHECK_ARG(d.aux != nullptr, "esvit_gemm: GELU' needs aux");
    ESVIT_CHECK_ARG(d.epilogue >= 0 && d.epilogue <= ESVIT_EPI_QGELU_BWD, "esvit_gemm: bad epilogue %d", d.epilogue);
    if (d.colsum && d.splitk > 1) ESVIT_CHECK_ARG(d.colsum_partial != nullptr, "esvit_gemm: colsum with split-K needs colsum_partial");
    if (d.colsum) ESVIT_CHECK_ARG(d.batch == 1, "esvit_gemm: colsum is not batched");
    if (d.colstat) ESVIT_CHECK_ARG(d.rowstat != nullptr, "esvit_gemm: colstat comes with rowstat");
    if (d.rowstat) {
        ESVIT_CHECK_ARG(dtype == ESVIT_BF16 && !d.a_kstrided && !d.b_kstrided && d.batch == 1 && d.splitk <= 1 && !d.out_f32 && d.epilogue == 0 &&
                            !d.residual && !d.rowmap && !d.rowscale && !d.bias && d.alpha == 1.f,
                        "esvit_gemm: row statistics come with the plain bf16 forward epilogue only");
        ESVIT_CHECK_ARG(d.M % 128 == 0 && d.N % 128 == 0 && d.ldc % 8 == 0 && ((uintptr_t)d.C % 16 == 0) &&
                            (!d.rowstat_center || (uintptr_t)d.rowstat_center % 16 == 0),
                        "esvit_gemm: row statistics need M and N in whole 128 x 128 tiles (M=%d N=%d)", d.M, d.N);
        if (d.colstat) ESVIT_CHECK_ARG(d.kernel != ESVIT_GEMM_P8 && (uintptr_t)d.colstat % 16 == 0, "esvit_gemm: colstat lives in the 128 x 128 statistics epilogue");
        ESVIT_CHECK_ARG(d.kernel == ESVIT_GEMM_AUTO || d.kernel == ESVIT_GEMM_DMA4 || d.kernel == ESVIT_GEMM_P8,
                        "esvit_gemm: row statistics exist in the 128 x 128 tile of the default main loop and in the 256 x 256 eight-phase loop");
    }
    return check_selector(dtype, d);
}

#ifdef ESVIT_ASTAT  // probe build only (tools/probe/astat_check.py, profiles/r06_gemm_astat_probe.txt)
// the A-stationary short-K kernel (gemm_kernels.h): plain / bias / GELU(+pre-activation) bf16 forwards over whole 128 x 128 tiles, K = 256 / 384
bool astat_supports(int dtype, const esvit_gemm_desc& d) {
    if (dtype != ESVIT_BF16 || d.kernel != ESVIT_GEMM_AUTO || d.a_kstrided || d.b_kstrided || d.batch > 1 || d.splitk > 1) return false;
    if (d.rowmap || d.rowstat || d.colstat || d.colsum || d.residual || d.rowscale || d.out_f32 || d.alpha != 1.f) return false;
    if (!(d.epilogue == 0 || d.epilogue == ESVIT_EPI_GELU)) return false;
    if (d.M % 128 != 0 || d.N % 128 != 0 || !(d.K == 256 || d.K == 384)) return false;
    if (d.lda % 8 != 0 || d.ldc % 8 != 0 || ((uintptr_t)d.C % 16) != 0) return false;
    if (d.epilogue == ESVIT_EPI_GELU && d.aux && (d.ldaux % 8 != 0 || ((uintptr_t)d.aux % 16) != 0)) return false;
    if ((long)d.N * d.ldb * 2 >= 0x7ff00000L) return false;  // 32-bit DMA offsets inside a column chunk
    return (long)d.M * d.N >= 8000000L;                       // (small problems stay where they were measured)
}

int run_astat(const esvit_gemm_desc& d, hipStream_t stream) {
    const bool gelu = d.epilogue == ESVIT_EPI_GELU, pre = gelu && d.aux;
    if (d.K == 256) {
        if (!gelu) return launch_gemm_astat<8, false, false>(d, stream);
        return pre ? launch_gemm_astat<8, true, true>(d, stream) : launch_gemm_astat<8, true, false>(d, stream);
    }
    if (!gelu) return launch_gemm_astat<12, false, false>(d, stream);
    return pre ? launch_gemm_astat<12, true, true>(d, stream) : launch_gemm_astat<12, true, false>(d, stream);
}
#endif

}  // namespace

extern "C" int esvit_gemm_select(int dtype, const esvit_gemm_desc* dp, int* tile_m, int* tile_n, int* resident_slots) {
    ESVIT_CHECK_ARG(dp != nullptr, "esvit_gemm_select: null descriptor");
    esvit_gemm_desc d = *dp;
    if (d.batch < 1) d.batch = 1;
    ESVIT_CHECK_ARG(d.M > 0 && d.N > 0 && d.K > 0, "esvit_gemm_select: bad shape M=%d N=%d K=%d", d.M, d.N, d.K);
    {
        const int rc = check_selector(dtype, d);
        if (rc != ESVIT_OK) return rc;
    }
    const GemmChoice c = choose(dtype, d);
    if (tile_m) *tile_m = c.bm;
    if (tile_n) *tile_n = c.bn;
    if (resident_slots) *resident_slots = (c.kernel == ESVIT_GEMM_DMA8 || c.kernel == ESVIT_GEMM_P8) ? 256 : 512;  // workgroups the chip holds at once (256 CUs)
    return c.kernel;
}

extern "C" int esvit_gemm(int dtype, const esvit_gemm_desc* dp, esvit_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    ESVIT_CHECK_ARG(dp != nullptr, "esvit_gemm: null descriptor");
    esvit_gemm_desc d = *dp;
    const int rc = validate(dtype, d);
    if (rc != ESVIT_OK) return rc;
#ifdef ESVIT_ASTAT
    if (astat_supports(dtype, d)) return run_astat(d, stream);
#endif
    const GemmChoice c = choose(dtype, d);
    if (!d.a_kstrided && !d.b_kstrided) return run_layout<false, false>(dtype, d, c, stream);
    if (!d.a_kstrided && d.b_kstrided) return run_layout<false, true>(dtype, d, c, stream);
    return run_layout<true, true>(dtype, d, c, stream);
}
