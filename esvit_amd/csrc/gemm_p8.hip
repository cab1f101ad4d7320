// ESVIT_GEMM_P8: the 256 x 256 x 64 bf16 main loop -- 8 waves, one workgroup per CU, 128 KiB of LDS, eight phases per pair of
// k-tiles (cdna_hip_programming.md, "The 256^2 8-phase template"; written from that description, not from its listing).
//
// Why a second large-tile loop next to ESVIT_GEMM_DMA8: a 128 x 128 tile needs 64 operand bytes per 4096 FLOP, and the LDS-DMA
// path delivers 27-42 B/clk/CU -- the 4-wave loop is bound by operand delivery at ~1.0 PFLOP/s (profiles/r03_gemm_mainloop_ablation.txt).
// A 256 x 256 tile halves the operand bytes per FLOP; what it must not do is drain its DMA queue at a barrier every k-tile.
//
// Geometry.  The k-tile (256 x 64 of A, 256 x 64 of B) lives in LDS as four 16 KiB half-tiles B0 A0 B1 A1 (128 rows each), two
// buffers (even / odd k-tile) = 128 KiB.  Wave (wr, wc), wr in {0,1}, wc in {0..3}, owns the four 64 x 32 output quadrants
//     rows  qm * 128 + wr * 64 + [0, 64)      columns  qn * 128 + wc * 32 + [0, 32)        (qm, qn in {0, 1})
// so that quadrant (qm, qn) reads only half-tiles A_qm and B_qn: the half-tiles of a k-tile are consumed one after the other and
// can be re-filled one after the other.  A k-tile is four phases of 16 MFMAs (v_mfma_f32_16x16x32_bf16) per wave:
//     phase 0: read B0 (2 x 2 fragments), A0 (4 x 2)  -> quadrant (0,0)       phase 1: read B1 -> quadrant (0,1)
//     phase 2: read A1                                 -> quadrant (1,1)       phase 3: (nothing to read) -> quadrant (1,0)
// Each phase also requests ONE half-tile (two LDS-DMA instructions per wave), seven half-tiles ahead of the one it consumes:
//     phase 0 of k-tile t -> A1 of t+1,   phases 1 / 2 / 3 -> B0 / A0 / B1 of t+2 (the half-tiles t itself has just finished with).
// One counted wait per k-tile (phase 3: vmcnt(6) = everything but the three youngest half-tiles has landed, i.e. k-tile t+1 is
// complete) -- the DMA queue is never drained inside the loop.  Every phase is  {reads + DMA issue | barrier | MFMAs | barrier};
// the wr = 1 waves run one barrier behind the wr = 0 waves, so on every SIMD (waves w and w + 4 share one) one wave is in its MFMA
// block while the other reads and issues: the matrix pipe never waits for an LDS read burst.
//
// Hazards (intervals between consecutive workgroup barriers; the wr = 0 half reads in interval 2 phi and computes in 2 phi + 1,
// the wr = 1 half one interval later):
//   RAW  a half-tile of k-tile t+1 is read from interval 2 (4t + 4) on; every wave's own share of it has landed before that wave's
//        phase-3 barrier of k-tile t (intervals 2 (4t + 3) / + 1): a full barrier lies between the last wait and the first read.
//   WAR  half-tile X of k-tile t is re-filled (for t+2) in the phase after the one that read it; the late half's reads of that
//        phase are retired before ITS first barrier of the phase (B0: the counted lgkmcnt below, issued first; A0, B1, A1: re-filled
//        two or more intervals after their reads were waited for by the MFMAs that consumed them).
//
// (A second schedule -- LDS reads of the next half-phase slipped between the MFMAs, one barrier per phase, no stagger -- was built in
// round 4 as well: equal on the nt probes, 25-30 % SLOWER on the k-strided layouts (two ds_read_b64_tr_b16 per fragment crowd the MFMA
// issue slots): tools/probe/gemm_p8_pipelined.hip keeps it.)
//
// The epilogue works straight from the (transposed) accumulators, one 64 x 32 quadrant at a time: epilogue_direct_at (plain / bias,
// GELU + pre-activation, GELU', residual + DropPath scale, fp32 / bf16, split-K partial); ragged edge tiles take a masked path.
#include "gemm_kernels.h"

#ifdef ESVIT_P8_TIMELINE
// tools/probe only: per-item phase stamps (s_memrealtime, 100 MHz) of the persistent walk: [start, after k-tile 0, after k-tile 1,
// end of the k-loop, end of the epilogue (stores issued), k-tiles, workgroup, -]
__device__ long* g_p8_timeline = nullptr;
#define P8_TL(item, slot, val) do { if (g_p8_timeline && threadIdx.x == 0) g_p8_timeline[(long)(item) * 8 + (slot)] = (long)(val); } while (0)
#define P8_NOW() __builtin_amdgcn_s_memrealtime()
#else
#define P8_TL(item, slot, val) do { } while (0)
#define P8_NOW() 0
#endif

namespace {

constexpr int P8_NT = 512;
constexpr int P8_HALF = 128 * 64 * 2;  // bytes of one half-tile
constexpr int P8_BUF = 4 * P8_HALF;    // one k-tile
enum { H_B0 = 0, H_A0 = 1, H_B1 = 2, H_A1 = 3 };

// MFMA with the accumulator pinned to the AccVGPR half of the register file and updated in place: 128 accumulator registers
// + 64 fragment registers leave the allocator no room for the copies it makes of loop-carried MFMA results otherwise (it then
// spills address registers INSIDE the k-loop, and every reload carries a vmcnt(0) that drains the DMA queue).
__device__ __forceinline__ void mma_acc(const Frag<bf16>& a, const Frag<bf16>& b, f32x4& c) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a.v), "v"(b.v));
}

template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// masked epilogue of one quadrant (ragged edge tiles, unaligned operands): every option of the descriptor, four columns at a time
__device__ __forceinline__ void p8_epilogue_masked(const esvit_gemm_desc& p, f32x4 (&acc)[4][2], long wrow0, int wcol0, int z) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));  // (the lane-dependent address pieces are recomputed here: kept live across the k-loop they get spilled)
    const int c = lane & 15, g = lane >> 4;
    const int M = p.M, N = p.N;
    const int mode = p.epilogue;
    bf16* auxp = reinterpret_cast<bf16*>(p.aux);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long m = wrow0 + 16 * i + c;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = wcol0 + 16 * j + 4 * g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e >= N) continue;
                float v = acc[i][j][e] * p.alpha;
                if (p.splitk > 1) {
                    p.partial[(long)z * M * N + m * N + n + e] = v;
                    continue;
                }
                if (p.bias) v += p.bias[n + e];
                if (mode == ESVIT_EPI_GELU || mode == ESVIT_EPI_QGELU) {
                    if (auxp) auxp[m * p.ldaux + n + e] = (bf16)v;
                    v = mode == ESVIT_EPI_GELU ? gelu_f(v) : qgelu_f(v);
                } else if (mode == ESVIT_EPI_GELU_BWD || mode == ESVIT_EPI_QGELU_BWD) {
                    const float a = (float)auxp[m * p.ldaux + n + e];
                    v *= mode == ESVIT_EPI_GELU_BWD ? gelu_grad_f(a) : qgelu_grad_f(a);
                }
                if (p.rowscale) v *= p.rowscale[m / p.rows_per_sample];
                if (p.residual) v += p.residual[m * p.ldr + n + e];
                const long o = (long)z * p.strideC + m * p.ldc + n + e;
                if (p.out_f32) reinterpret_cast<float*>(p.C)[o] = v;
                else reinterpret_cast<bf16*>(p.C)[o] = (bf16)v;
            }
        }
    }
}

// Fast epilogue kinds (template parameter of the kernel: one straight-line path per instantiation keeps the epilogue inside the
// 128 ordinary registers the AccVGPR split leaves; everything else -- ragged edge tiles, unaligned operands, rarer option
// combinations -- takes p8_epilogue_masked):
enum { P8_BF16 = 0,      // C (bf16) = alpha acc + bias
       P8_F32 = 1,       // C (fp32) = alpha acc + bias; also the split-K partial slab
       P8_GELU = 2,      // aux (bf16) = v = alpha acc + bias; C (bf16) = gelu(v)   (erf-GELU or QuickGELU)
       P8_RES = 3,       // C (fp32) = (alpha acc + bias) * rowscale[row / rows_per_sample] + residual
       P8_GELU_BWD = 4,  // C (bf16) = alpha acc * gelu'(aux)
       P8_BF16_RS = 5    // P8_BF16 without bias + the softmax statistics of the stored row pieces (esvit_gemm_desc::rowstat, 32-column blocks)
};

// Interior quadrant (64 rows x 32 columns per wave), straight from the transposed accumulators: lane (c, g) holds, for row block i
// and column block j, the four consecutive columns 16 j + 4 g .. + 3 of row 16 i + c.  bf16 rows leave as one 16-byte store per
// lane (two column blocks exchanged between lane groups g and g ^ 1: store_row_bf16), fp32 rows as two.  The inputs of a quadrant
// (residual / GELU pre-activation) are all requested before its first store: vmcnt retires in order.
template <int EPI>
__device__ __forceinline__ void p8_epilogue_fast(const esvit_gemm_desc& p, f32x4 (&acc)[4][2], long wrow0, int wcol0, int z) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));  // (the lane-dependent address pieces are recomputed here: kept live across the k-loop they get spilled)
    const int c = lane & 15, g = lane >> 4;
    const long row = wrow0 + c;
    const int col = wcol0 + 4 * g;
    const float alpha = p.alpha;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI != P8_GELU_BWD && EPI != P8_BF16_RS) {
        if (p.bias && p.splitk <= 1) {
            b0 = *reinterpret_cast<const f32x4*>(p.bias + col);
            b1 = *reinterpret_cast<const f32x4*>(p.bias + col + 16);
        }
    }
    if constexpr (EPI == P8_BF16) {
        bf16* cp = reinterpret_cast<bf16*>(p.C) + (long)z * p.strideC + row * p.ldc + wcol0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v[2] = {acc[i][0] * alpha + b0, acc[i][1] * alpha + b1};
            store_row_bf16<2>(cp + 16 * i * p.ldc, v, g);
        }
    } else if constexpr (EPI == P8_BF16_RS) {
        // logits + (max, sum 2^(z - max)) of z = (stored logit - centre) * scale over this wave's 32 columns of every row: 8 values
        // per lane, then the four lane groups that share the row (esvit_rowstat_combine folds the N / 32 blocks of a row)
        bf16* cp = reinterpret_cast<bf16*>(p.C) + (long)z * p.strideC + row * p.ldc + wcol0;
        f32x4 cen[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if (p.rowstat_center) {
            cen[0] = *reinterpret_cast<const f32x4*>(p.rowstat_center + col) * p.rowstat_scale;
            cen[1] = *reinterpret_cast<const f32x4*>(p.rowstat_center + col + 16) * p.rowstat_scale;
        }
        const long nb = p.N >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v[2] = {acc[i][0] * alpha, acc[i][1] * alpha};
            store_row_bf16<2>(cp + 16 * i * p.ldc, v, g);
            float zz[2][4];
            float m = -3.0e38f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    zz[j][e] = fmaf((float)(bf16)v[j][e], p.rowstat_scale, -cen[j][e]);
                    m = fmaxf(m, zz[j][e]);
                }
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) sum += __builtin_amdgcn_exp2f(zz[j][e] - m);
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            if (g == 0) *reinterpret_cast<f32x2*>(p.rowstat + ((row + 16 * i) * nb + (wcol0 >> 5)) * 2) = f32x2{m, sum};
        }
    } else if constexpr (EPI == P8_F32) {
        float* cp;
        long ld;
        if (p.splitk > 1) {
            ld = p.N;
            cp = p.partial + (long)z * p.M * p.N + row * ld + col;
        } else {
            ld = p.ldc;
            cp = reinterpret_cast<float*>(p.C) + (long)z * p.strideC + row * ld + col;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (p.splitk > 1) {
                *reinterpret_cast<f32x4*>(cp + 16 * i * ld) = acc[i][0] * alpha + b0;
                *reinterpret_cast<f32x4*>(cp + 16 * i * ld + 16) = acc[i][1] * alpha + b1;
            } else {
                store_stream(reinterpret_cast<f32x4*>(cp + 16 * i * ld), acc[i][0] * alpha + b0);
                store_stream(reinterpret_cast<f32x4*>(cp + 16 * i * ld + 16), acc[i][1] * alpha + b1);
            }
        }
    } else if constexpr (EPI == P8_GELU) {
        bf16* cp = reinterpret_cast<bf16*>(p.C) + (long)z * p.strideC + row * p.ldc + wcol0;
        bf16* ap = p.aux ? reinterpret_cast<bf16*>(p.aux) + row * p.ldaux + wcol0 : nullptr;
        const bool quick = p.epilogue == ESVIT_EPI_QGELU;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v[2] = {acc[i][0] * alpha + b0, acc[i][1] * alpha + b1};
            if (ap) store_row_bf16<2>(ap + 16 * i * p.ldaux, v, g);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[j][e] = quick ? qgelu_f(v[j][e]) : gelu_f(v[j][e]);
            store_row_bf16<2>(cp + 16 * i * p.ldc, v, g);
        }
    } else if constexpr (EPI == P8_RES) {
        const float* rp = p.residual + row * p.ldr + col;
        f32x4 r[4][2];
        float rs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r[i][0] = *reinterpret_cast<const f32x4*>(rp + 16 * i * p.ldr);
            r[i][1] = *reinterpret_cast<const f32x4*>(rp + 16 * i * p.ldr + 16);
            rs[i] = p.rowscale ? p.rowscale[(row + 16 * i) / p.rows_per_sample] : 1.f;
        }
        float* cp = reinterpret_cast<float*>(p.C) + (long)z * p.strideC + row * p.ldc + col;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            store_stream(reinterpret_cast<f32x4*>(cp + 16 * i * p.ldc), (acc[i][0] * alpha + b0) * rs[i] + r[i][0]);
            store_stream(reinterpret_cast<f32x4*>(cp + 16 * i * p.ldc + 16), (acc[i][1] * alpha + b1) * rs[i] + r[i][1]);
        }
    } else {  // P8_GELU_BWD
        const bf16* ap = reinterpret_cast<const bf16*>(p.aux) + row * p.ldaux + col;
        u32x2_t a[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i][0] = *reinterpret_cast<const u32x2_t*>(ap + 16 * i * p.ldaux);
            a[i][1] = *reinterpret_cast<const u32x2_t*>(ap + 16 * i * p.ldaux + 16);
        }
        bf16* cp = reinterpret_cast<bf16*>(p.C) + (long)z * p.strideC + row * p.ldc + wcol0;
        const bool quick = p.epilogue == ESVIT_EPI_QGELU_BWD;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 v[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 x = {__builtin_bit_cast(float, a[i][j][0] << 16), __builtin_bit_cast(float, a[i][j][0] & 0xffff0000u),
                                 __builtin_bit_cast(float, a[i][j][1] << 16), __builtin_bit_cast(float, a[i][j][1] & 0xffff0000u)};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[j][e] = acc[i][j][e] * alpha * (quick ? qgelu_grad_f(x[e]) : gelu_grad_f(x[e]));
            }
            store_row_bf16<2>(cp + 16 * i * p.ldc, v, g);
        }
    }
}

// does the descriptor's epilogue equal fast kind EPI on interior tiles (alignment included)?  Evaluated per launch on the host.
inline bool p8_epi_matches(const esvit_gemm_desc& d, int epi) {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (d.splitk > 1) return epi == P8_F32 && (d.N % 4 == 0) && al16(d.partial);
    if (!((d.ldc % 8 == 0) && al16(d.C) && (d.strideC % 8 == 0) && (!d.bias || al16(d.bias)))) return false;
    const bool gelu = d.epilogue == ESVIT_EPI_GELU || d.epilogue == ESVIT_EPI_QGELU;
    const bool gelu_bwd = d.epilogue == ESVIT_EPI_GELU_BWD || d.epilogue == ESVIT_EPI_QGELU_BWD;
    switch (epi) {
        case P8_BF16: return !gelu && !gelu_bwd && !d.residual && !d.rowscale && !d.out_f32 && !d.rowstat;
        case P8_BF16_RS: return d.rowstat && !gelu && !gelu_bwd && !d.residual && !d.rowscale && !d.out_f32 && !d.bias;
        case P8_F32: return !gelu && !gelu_bwd && !d.residual && !d.rowscale && d.out_f32;
        case P8_GELU: return gelu && !d.residual && !d.rowscale && !d.out_f32 && (!d.aux || ((d.ldaux % 8 == 0) && al16(d.aux)));
        case P8_RES: return !gelu && !gelu_bwd && d.residual && d.out_f32 && (d.ldr % 4 == 0) && al16(d.residual);
        case P8_GELU_BWD: return gelu_bwd && !d.bias && !d.residual && !d.rowscale && !d.out_f32 && (d.ldaux % 8 == 0) && al16(d.aux);
    }
    return false;
}

// one unit of work: an output tile and its split-K slice / batch item
struct P8Item {
    int kt0;  // first k-tile (split-K slice)
    int nk;   // k-tiles
    int m0, n0, z, tn;
};

template <bool AKS, bool BKS, int EPI>
__global__ __launch_bounds__(P8_NT, 1) void gemm_p8_kernel(const esvit_gemm_desc p, const int group_m, const int epi_fast) {
    using HA = DmaTile<AKS, 128, 64, P8_NT>;
    using HB = DmaTile<BKS, 128, 64, P8_NT>;
    static_assert(HA::INSTR_PER_WAVE == 2 && HB::INSTR_PER_WAVE == 2, "two LDS-DMA instructions per wave and half-tile");
    constexpr int NRA = AKS ? 16 : 8;  // LDS read instructions of one A sub-tile (4 x 2 fragments)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;

    const int M = p.M, N = p.N, K = p.K;
    const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
    const int ntiles = tiles_m * tiles_n;
    const int nz = p.splitk > 1 ? p.splitk : p.batch;
    const int total = ntiles * nz;
    const long a_rows = AKS ? (long)K : (long)M, b_rows = BKS ? (long)K : (long)N;

    // Work list: item w = (z = w / ntiles, tile = w % ntiles), tiles in tile_coords order.  The dispatcher places workgroup b on XCD
    // b % 8: every XCD owns a contiguous range of the list and its gridDim.x / 8 workgroups walk it with that stride, so the tiles
    // in flight on one XCD at any time are neighbours (shared A row panels / B column panels in one L2), and with split-K every K
    // slice of A and B is fetched by one XCD only.
    const int per_xcd = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, widx = blockIdx.x >> 3;
    const int wq = total / 8, wrem = total % 8;
    const int w_begin = xcd < wrem ? xcd * (wq + 1) : wrem * (wq + 1) + (xcd - wrem) * wq;
    const int w_end = w_begin + wq + (xcd < wrem ? 1 : 0);
    if (w_begin + widx >= w_end) return;

    auto make_item = [&](int w) __attribute__((always_inline)) {
        P8Item it;
        const int z = w / ntiles, pid = w - z * ntiles;
        int tm, tn;
        tile_coords(pid, tiles_m, tiles_n, group_m, tm, tn);
        it.m0 = tm * 256;
        it.n0 = tn * 256;
        it.z = z;
        it.tn = tn;
        it.kt0 = 0;
        it.nk = K / 64;  // K % 64 == 0 (dispatcher)
        if (p.splitk > 1) {
            const int per = (it.nk + p.splitk - 1) / p.splitk;
            it.kt0 = min(it.nk, z * per);
            it.nk = min(it.nk, (z + 1) * per) - it.kt0;
        }
        return it;
    };

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;

    int voffA[2], voffB[2];
    HA::wave_offsets(p.lda, wave, lane, voffA);
    HB::wave_offsets(p.ldb, wave, lane, voffB);
    const long a_kstep = AKS ? p.lda * 128 : 128, b_kstep = BKS ? p.ldb * 128 : 128;  // bytes per k-tile in the scalar offset

    // ---- the request stream ----
    // Half-tiles are requested in ONE sequence B0 A0 B1 A1 | B0 A0 B1 A1 | ... that runs through the k-tiles of an item and on
    // into the next item of this workgroup, seven half-tiles ahead of the k-tile being multiplied: the DMA queue does not drain
    // at an output-tile boundary either, the first k-tiles of the next tile land while this tile's epilogue stores.  Past the end of
    // the work the requests go through a descriptor of zero records: no memory access, and the wave's vmcnt bookkeeping is the
    // same in every k-tile.  State: the item being requested (sw), its four half-tile descriptors, the k-tile (absolute index skt,
    // sleft left in the item).
    // One buffer descriptor per half-tile, based at the half's first row (k-contiguous operand) / first column (k-strided):
    // per-lane offsets stay small, rows past the end of a k-contiguous operand fall outside num_records and read as 0.
    __amdgpu_buffer_rsrc_t sra0, sra1, srb0, srb1;
    int sw = w_begin + widx - per_xcd, skt = 0, sleft = 0;
    auto half_rsrc = [&](const bf16* base, bool ks, long rows_total, long ld, int first) __attribute__((always_inline)) {
        const bf16* b = ks ? base + first : base + (long)first * ld;
        const long left = ks ? (rows_total * ld - first) * 2 : (rows_total - first) * ld * 2;
        return make_rsrc(b, left);
    };
    auto stream_next_item = [&]() __attribute__((always_inline)) {
        sleft = 0;
        while (sleft == 0) {  // (an empty split-K slice has nothing to request)
            sw += per_xcd;
            if (sw >= w_end) {
                sra0 = sra1 = srb0 = srb1 = make_rsrc(p.A, 0);
                skt = 0;
                sleft = 0x7fffffff;
                return;
            }
            const P8Item it = make_item(sw);
            const bf16* A = reinterpret_cast<const bf16*>(p.A) + (p.splitk > 1 ? 0 : (long)it.z * p.strideA);
            const bf16* B = reinterpret_cast<const bf16*>(p.B) + (p.splitk > 1 ? 0 : (long)it.z * p.strideB);
            sra0 = half_rsrc(A, AKS, a_rows, p.lda, it.m0);
            sra1 = half_rsrc(A, AKS, a_rows, p.lda, it.m0 + 128);
            srb0 = half_rsrc(B, BKS, b_rows, p.ldb, it.n0);
            srb1 = half_rsrc(B, BKS, b_rows, p.ldb, it.n0 + 128);
            skt = it.kt0;
            sleft = it.nk;
        }
    };
    // request the stream's next half-tile (H, compile time: the call sites follow the sequence) into the buffer at byte offset boff
    auto stage = [&](auto hc, int boff) __attribute__((always_inline)) {
        constexpr int H = decltype(hc)::value;
        char* dst = smem + boff + H * P8_HALF + wave * 2048;
        if constexpr (H == H_A0 || H == H_A1) {
            const int soff = (int)(skt * a_kstep);
            const __amdgpu_buffer_rsrc_t r = H == H_A0 ? sra0 : sra1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst), 16, voffA[0], soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + 1024), 16, voffA[1], soff, 0, 0);
        } else {
            const int soff = (int)(skt * b_kstep);
            const __amdgpu_buffer_rsrc_t r = H == H_B0 ? srb0 : srb1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst), 16, voffB[0], soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + 1024), 16, voffB[1], soff, 0, 0);
        }
        if constexpr (H == H_A1) {  // the k-tile is complete: on to the next one
            ++skt;
            if (--sleft == 0) stream_next_item();
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    f32x4 acc[2][2][4][2];
    float csum[2][4];        // fused bias gradient (weight-gradient layout only): this lane's share of the row sums of op(A)
    Frag<bf16> fa[2][4];     // [k-step][row fragment] of the current A sub-tile
    Frag<bf16> fb[2][2][2];  // [qn][k-step][column fragment]: both B sub-tiles stay in registers (quadrant (1,0) re-uses B0)

    auto read_a = [&](int boff, auto qmc) __attribute__((always_inline)) {
        constexpr int QM = decltype(qmc)::value;
        const bf16* lds = reinterpret_cast<const bf16*>(smem + boff + (QM ? H_A1 : H_A0) * P8_HALF);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[kk][i] = HA::frag(lds, wr * 64 + i * 16, kk, c, g);
    };
    auto read_b = [&](int boff, auto qnc) __attribute__((always_inline)) {
        constexpr int QN = decltype(qnc)::value;
        const bf16* lds = reinterpret_cast<const bf16*>(smem + boff + (QN ? H_B1 : H_B0) * P8_HALF);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[QN][kk][j] = HB::frag(lds, wc * 32 + j * 16, kk, c, g);
    };
    auto mfma_quadrant = [&](auto qmc, auto qnc) __attribute__((always_inline)) {
        constexpr int QM = decltype(qmc)::value, QN = decltype(qnc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma_acc(fb[QN][kk][j], fa[kk][i], acc[QM][QN][i][j]);  // operands swapped: see epilogue_direct
        __builtin_amdgcn_s_setprio(0);
    };
    // bias gradient: row c of row fragment i, this lane's eight k of every k-step: v_dot2c_f32_bf16 against (1, 1) -- eight VALU
    // instructions per k-tile and quadrant row in the two wc = 0 waves, issued in their read interval (the matrix pipe is busy with
    // the other half's MFMAs then); an all-ones MFMA fragment would need 32 more accumulator registers than the file has
    auto colsum_a = [&](auto qmc) __attribute__((always_inline)) {
        constexpr int QM = decltype(qmc)::value;
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        const bf16x2_t one2 = {(bf16)1.0f, (bf16)1.0f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    csum[QM][i] = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{fa[kk][i].v[2 * e], fa[kk][i].v[2 * e + 1]}, one2, csum[QM][i], false);
    };
    auto bar = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // prologue: the first k-tile of the stream and the first three half-tiles of the second
    stream_next_item();
    stage(I0{}, 0);
    stage(I1{}, 0);
    stage(I2{}, 0);
    stage(I3{}, 0);
    stage(I0{}, P8_BUF);
    stage(I1{}, P8_BUF);
    stage(I2{}, P8_BUF);
    wait_vmcnt<6>();
    bar();
    int boff = 0;  // byte offset of the buffer that holds the k-tile being multiplied

    for (int w = w_begin + widx; w < w_end; w += per_xcd) {
        const P8Item cur = make_item(w);
        static_for<16>([&](auto ic) {
            constexpr int q = decltype(ic)::value >> 2, i = decltype(ic)::value & 3;
            acc[q >> 1][q & 1][i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[q >> 1][q & 1][i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        });
        const bool do_colsum = AKS && p.colsum && cur.tn == 0 && wc == 0;
        if constexpr (AKS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) csum[0][i] = csum[1][i] = 0.f;
        }
        P8_TL(w, 0, P8_NOW());
        P8_TL(w, 5, cur.nk);
        P8_TL(w, 6, blockIdx.x);
        if (wr == 1) bar();  // the wr = 1 half runs one barrier behind
        for (int t = 0; t < cur.nk; ++t) {
            const int other = boff ^ P8_BUF;
            // phase 0
            read_b(boff, I0{});
            __builtin_amdgcn_sched_barrier(0);
            read_a(boff, I0{});
            stage(I3{}, other);                     // A1 of the next k-tile
            wait_lgkmcnt<(NRA < 15 ? NRA : 15)>();  // the B0 reads (issued first) are retired: B0 may be re-filled from the next phase on
            bar();
            mfma_quadrant(I0{}, I0{});
            bar();
            // phase 1
            read_b(boff, I1{});
            stage(I0{}, boff);  // B0 of the k-tile after the next: into the half-tile phase 0 has finished with
            if constexpr (AKS) {
                if (do_colsum) colsum_a(I0{});  // (the A0 fragments of phase 0 are still in registers)
            }
            bar();
            mfma_quadrant(I0{}, I1{});
            bar();
            // phase 2
            read_a(boff, I1{});
            stage(I1{}, boff);  // A0
            bar();
            mfma_quadrant(I1{}, I1{});
            bar();
            // phase 3
            stage(I2{}, boff);  // B1
            if constexpr (AKS) {
                if (do_colsum) colsum_a(I1{});
            }
            wait_vmcnt<6>();    // all of the next k-tile has landed; the three half-tiles just requested stay in flight
            bar();
            mfma_quadrant(I1{}, I0{});
            bar();
            boff = other;
            if (t < 2) P8_TL(w, 1 + t, P8_NOW());
        }
        if (wr == 0) bar();
        P8_TL(w, 3, P8_NOW());

        // ---- epilogue of the item: straight from the accumulators; nothing waits for its stores, the next item's first k-tiles
        // (requested during the last two k-tiles above) land meanwhile ----
        if constexpr (AKS) {
            if (do_colsum) {  // the four lane groups hold four k-slices of the same rows
                float* dst = p.splitk > 1 ? p.colsum_partial + (long)cur.z * M : p.colsum;
#pragma unroll
                for (int qm = 0; qm < 2; ++qm)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = csum[qm][i];
                        v += __shfl_xor(v, 16, 64);
                        v += __shfl_xor(v, 32, 64);
                        const int m = cur.m0 + qm * 128 + wr * 64 + i * 16 + c;
                        if (g == 0 && m < M) dst[m] = v * p.alpha;
                    }
            }
        }
        const bool fast = epi_fast && cur.m0 + 256 <= M && cur.n0 + 256 <= N;
        static_for<4>([&](auto qc) {
            constexpr int qm = decltype(qc)::value >> 1, qn = decltype(qc)::value & 1;
            const long wrow0 = (long)cur.m0 + qm * 128 + wr * 64;
            const int wcol0 = cur.n0 + qn * 128 + wc * 32;
            if (fast) p8_epilogue_fast<EPI>(p, acc[qm][qn], wrow0, wcol0, cur.z);
            else p8_epilogue_masked(p, acc[qm][qn], wrow0, wcol0, cur.z);
            __builtin_amdgcn_sched_barrier(0);
        });
        P8_TL(w, 4, P8_NOW());
    }
}

template <bool AKS, bool BKS, int EPI>
int launch_p8(const esvit_gemm_desc& d, bool epi_fast, hipStream_t stream) {
    auto kern = gemm_p8_kernel<AKS, BKS, EPI>;
    constexpr int lds = 2 * P8_BUF;
    static unsigned long long lds_set = 0;
    esvit_raise_lds(kern, lds, lds_set);
    const int tm_ = ceil_div(d.M, 256), tn_ = ceil_div(d.N, 256);
    const int nz = d.splitk > 1 ? d.splitk : d.batch;
    int group_m = 1;
    if (nz == 1) {
        if (tn_ > 32) group_m = 2;
        else if (tn_ >= 6) group_m = 8;
        else if (tn_ >= 2 && tm_ >= 1024) group_m = 16;
    }
    // one workgroup per CU walks its share of the work list as ONE k-tile stream (the next item's first k-tiles are requested
    // before the current item's epilogue)
    const long total = (long)tm_ * tn_ * nz;
    const int grid = total > 256 ? 256 : (int)((total + 7) / 8 * 8);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(P8_NT), lds, stream, d, group_m, (int)epi_fast);
    ESVIT_CHECK_LAUNCH("esvit_gemm(p8)");
    if (d.splitk > 1) return launch_splitk_reduce(d, true, stream);
    return ESVIT_OK;
}

// the instantiation whose fast epilogue is the descriptor's (else the plain one with every tile on the masked path)
template <bool AKS, bool BKS, int... EPIS>
int dispatch_p8(const esvit_gemm_desc& d, hipStream_t stream) {
    int rc = ESVIT_ERR_UNSUPPORTED;
    bool done = false;
    auto try_one = [&](auto ec) {
        constexpr int E = decltype(ec)::value;
        if (!done && p8_epi_matches(d, E)) {
            rc = launch_p8<AKS, BKS, E>(d, true, stream);
            done = true;
        }
    };
    (try_one(std::integral_constant<int, EPIS>{}), ...);
    if (!done) rc = launch_p8<AKS, BKS, P8_F32>(d, false, stream);
    return rc;
}

}  // namespace

// bf16 only; K % 64 == 0, no row map, operands below 4 GiB (32-bit DMA offsets): checked by esvit_gemm's dispatcher (gemm.hip)
int esvit_gemm_p8_launch(const esvit_gemm_desc& d, hipStream_t stream) {
    if (!d.a_kstrided && !d.b_kstrided) return dispatch_p8<false, false, P8_BF16, P8_BF16_RS, P8_GELU, P8_RES, P8_F32>(d, stream);  // forward
    if (!d.a_kstrided && d.b_kstrided) return dispatch_p8<false, true, P8_BF16, P8_GELU_BWD, P8_F32>(d, stream);       // dgrad
    return dispatch_p8<true, true, P8_F32, P8_RES>(d, stream);                                                          // wgrad
}

#ifdef ESVIT_P8_TIMELINE
extern "C" __attribute__((visibility("default"))) int p8_probe_set_timeline(long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_p8_timeline), &buf, sizeof(buf));
}
extern "C" __attribute__((visibility("default"))) int p8_probe_gemm(const esvit_gemm_desc* d, void* stream) {
    esvit_gemm_desc dd = *d;
    if (dd.batch < 1) dd.batch = 1;
    if (dd.splitk < 1) dd.splitk = 1;
    return esvit_gemm_p8_launch(dd, reinterpret_cast<hipStream_t>(stream));
}
void esvit_set_error(const char*, ...) {}
#endif
