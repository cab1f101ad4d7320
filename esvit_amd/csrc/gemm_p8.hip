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
// The epilogue works straight from the (transposed) accumulators, one 64 x 32 quadrant at a time: epilogue_direct_at (plain / bias,
// GELU + pre-activation, GELU', residual + DropPath scale, fp32 / bf16, split-K partial); ragged edge tiles take a masked path.
#include "gemm_kernels.h"

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
    const int lane = threadIdx.x & 63;
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

// one quadrant: the straight-line epilogue when the whole 256 x 256 tile is interior and the operands are vector-aligned
__device__ __forceinline__ void p8_epilogue_quadrant(const esvit_gemm_desc& p, f32x4 (&acc)[4][2], long wrow0, int wcol0, int z, bool fast, int kind) {
    if (!fast) {
        p8_epilogue_masked(p, acc, wrow0, wcol0, z);
        return;
    }
    if (p.splitk > 1) {
        epilogue_direct_at<4, 2, EK_PLAIN, true>(p, acc, wrow0, wcol0, p.partial + (long)z * p.M * p.N, p.N, 0, nullptr);
        return;
    }
    const long cf = (long)z * p.strideC;
    if (kind == EK_GELU) epilogue_direct_at<4, 2, EK_GELU, false>(p, acc, wrow0, wcol0, p.C, p.ldc, cf, p.bias);
    else if (kind == EK_GELU_BWD) {
        if (p.out_f32) epilogue_direct_at<4, 2, EK_GELU_BWD, true>(p, acc, wrow0, wcol0, p.C, p.ldc, cf, p.bias);
        else epilogue_direct_at<4, 2, EK_GELU_BWD, false>(p, acc, wrow0, wcol0, p.C, p.ldc, cf, p.bias);
    } else if (kind == EK_RES) {
        if (p.out_f32) epilogue_direct_at<4, 2, EK_RES, true>(p, acc, wrow0, wcol0, p.C, p.ldc, cf, p.bias);
        else epilogue_direct_at<4, 2, EK_RES, false>(p, acc, wrow0, wcol0, p.C, p.ldc, cf, p.bias);
    } else {
        if (p.out_f32) epilogue_direct_at<4, 2, EK_PLAIN, true>(p, acc, wrow0, wcol0, p.C, p.ldc, cf, p.bias);
        else epilogue_direct_at<4, 2, EK_PLAIN, false>(p, acc, wrow0, wcol0, p.C, p.ldc, cf, p.bias);
    }
}

template <bool AKS, bool BKS>
__global__ __launch_bounds__(P8_NT, 1) void gemm_p8_kernel(const esvit_gemm_desc p, const int group_m) {
    using HA = DmaTile<AKS, 128, 64, P8_NT>;
    using HB = DmaTile<BKS, 128, 64, P8_NT>;
    static_assert(HA::INSTR_PER_WAVE == 2 && HB::INSTR_PER_WAVE == 2, "two LDS-DMA instructions per wave and half-tile");
    constexpr int NRA = AKS ? 16 : 8;  // LDS read instructions of one A sub-tile (4 x 2 fragments)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    typedef __attribute__((address_space(3))) void lds_void;

    const int M = p.M, N = p.N, K = p.K;
    const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
    const int ntiles = tiles_m * tiles_n;
    int pid, z;
    if (group_m < 0) xcd_work_map_zmajor(ntiles, pid, z);
    else xcd_tile_map(ntiles, pid, z);
    int tm, tn;
    tile_coords(pid, tiles_m, tiles_n, group_m, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;
    const bf16* A = reinterpret_cast<const bf16*>(p.A);
    const bf16* B = reinterpret_cast<const bf16*>(p.B);
    int kbeg = 0, kend = K;
    if (p.splitk > 1) {
        const int nkt = K / 64;
        const int per = (nkt + p.splitk - 1) / p.splitk;
        kbeg = min(K, z * per * 64);
        kend = min(K, (z + 1) * per * 64);
    } else {
        A += (long)z * p.strideA;
        B += (long)z * p.strideB;
    }
    const int nk = (kend - kbeg) / 64;  // K % 64 == 0 (launcher)

    // one buffer descriptor per half-tile, based at the half's first row (k-contiguous operand) / first column (k-strided):
    // per-lane offsets stay small, rows past the end of a k-contiguous operand fall outside num_records and read as 0
    const long a_rows = AKS ? (long)K : (long)M, b_rows = BKS ? (long)K : (long)N;
    auto half_rsrc = [&](const bf16* base, bool ks, long rows_total, long ld, int first) __attribute__((always_inline)) {
        const bf16* b = ks ? base + first : base + (long)first * ld;
        const long left = ks ? (rows_total * ld - first) * 2 : (rows_total - first) * ld * 2;
        return make_rsrc(b, left);
    };
    const __amdgpu_buffer_rsrc_t ra0 = half_rsrc(A, AKS, a_rows, p.lda, m0), ra1 = half_rsrc(A, AKS, a_rows, p.lda, m0 + 128);
    const __amdgpu_buffer_rsrc_t rb0 = half_rsrc(B, BKS, b_rows, p.ldb, n0), rb1 = half_rsrc(B, BKS, b_rows, p.ldb, n0 + 128);
    const __amdgpu_buffer_rsrc_t rnull = make_rsrc(A, 0);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int wr = wave >> 2, wc = wave & 3;

    int voffA[2], voffB[2];
    HA::wave_offsets(p.lda, wave, lane, voffA);
    HB::wave_offsets(p.ldb, wave, lane, voffB);
    const long a_kstep = AKS ? p.lda * 128 : 128, b_kstep = BKS ? p.ldb * 128 : 128;  // bytes per k-tile in the scalar offset

    // request half-tile H of k-tile `t` into the buffer at byte offset `boff`; a k-tile past the end is requested through a
    // descriptor of zero records: no memory access, and the wave's vmcnt bookkeeping stays the same in every k-tile
    auto stage = [&](auto hc, int boff, int t) __attribute__((always_inline)) {
        constexpr int H = decltype(hc)::value;
        char* dst = smem + boff + H * P8_HALF + wave * 2048;
        const long kt = kbeg / 64 + t;
        const bool live = t < nk;
        if constexpr (H == H_A0 || H == H_A1) {
            const int soff = (int)(kt * a_kstep);
            __amdgpu_buffer_rsrc_t r = H == H_A0 ? ra0 : ra1;
            if (!live) r = rnull;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst), 16, voffA[0], soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + 1024), 16, voffA[1], soff, 0, 0);
        } else {
            const int soff = (int)(kt * b_kstep);
            __amdgpu_buffer_rsrc_t r = H == H_B0 ? rb0 : rb1;
            if (!live) r = rnull;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst), 16, voffB[0], soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(dst + 1024), 16, voffB[1], soff, 0, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;

    f32x4 acc[2][2][4][2];
    static_for<16>([&](auto ic) {
        constexpr int q = decltype(ic)::value >> 2, i = decltype(ic)::value & 3;
        acc[q >> 1][q & 1][i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[q >> 1][q & 1][i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    });

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
    auto bar = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    if (nk > 0) {
        // prologue: k-tile 0 and the first three half-tiles of k-tile 1
        stage(I0{}, 0, 0);
        stage(I1{}, 0, 0);
        stage(I2{}, 0, 0);
        stage(I3{}, 0, 0);
        stage(I0{}, P8_BUF, 1);
        stage(I1{}, P8_BUF, 1);
        stage(I2{}, P8_BUF, 1);
        wait_vmcnt<6>();
        bar();
        if (wr == 1) bar();  // the wr = 1 half runs one barrier behind
        int boff = 0;        // byte offset of k-tile t's buffer
        for (int t = 0; t < nk; ++t) {
            const int other = boff ^ P8_BUF;
            // phase 0
            read_b(boff, I0{});
            __builtin_amdgcn_sched_barrier(0);
            read_a(boff, I0{});
            stage(I3{}, other, t + 1);
            wait_lgkmcnt<(NRA < 15 ? NRA : 15)>();  // the B0 reads (issued first) are retired: B0 may be re-filled from the next phase on
            bar();
            mfma_quadrant(I0{}, I0{});
            bar();
            // phase 1
            read_b(boff, I1{});
            stage(I0{}, boff, t + 2);
            bar();
            mfma_quadrant(I0{}, I1{});
            bar();
            // phase 2
            read_a(boff, I1{});
            stage(I1{}, boff, t + 2);
            bar();
            mfma_quadrant(I1{}, I1{});
            bar();
            // phase 3
            stage(I2{}, boff, t + 2);
            wait_vmcnt<6>();  // all of k-tile t+1 has landed; B0 A0 B1 of t+2 stay in flight
            bar();
            mfma_quadrant(I1{}, I0{});
            bar();
            boff = other;
        }
        if (wr == 0) bar();
    }

    // ---- epilogue ----
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    bool fast = m0 + 256 <= M && n0 + 256 <= N;
    int kind = EK_PLAIN;
    if (p.splitk > 1) {
        fast = fast && (N % 4 == 0) && al16(p.partial);
    } else {
        fast = fast && (p.ldc % 8 == 0) && al16(p.C) && ((p.strideC * (long)z) % 8 == 0) && (!p.bias || al16(p.bias));
        if (p.epilogue == ESVIT_EPI_GELU || p.epilogue == ESVIT_EPI_QGELU) {
            kind = EK_GELU;
            fast = fast && !p.residual && !p.rowscale && !p.out_f32 && (!p.aux || ((p.ldaux % 8 == 0) && al16(p.aux)));
        } else if (p.epilogue == ESVIT_EPI_GELU_BWD || p.epilogue == ESVIT_EPI_QGELU_BWD) {
            kind = EK_GELU_BWD;
            fast = fast && !p.residual && !p.rowscale && (p.ldaux % 8 == 0) && al16(p.aux);
        } else if (p.residual) {
            kind = EK_RES;
            fast = fast && (p.ldr % 4 == 0) && al16(p.residual);
        } else {
            fast = fast && !p.rowscale;
        }
    }
    static_for<4>([&](auto qc) {
        constexpr int qm = decltype(qc)::value >> 1, qn = decltype(qc)::value & 1;
        p8_epilogue_quadrant(p, acc[qm][qn], (long)m0 + qm * 128 + wr * 64, n0 + qn * 128 + wc * 32, z, fast, kind);
    });
}

template <bool AKS, bool BKS>
int launch_p8(const esvit_gemm_desc& d, hipStream_t stream) {
    auto kern = gemm_p8_kernel<AKS, BKS>;
    constexpr int lds = 2 * P8_BUF;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    const int tm_ = ceil_div(d.M, 256), tn_ = ceil_div(d.N, 256);
    const int nz = d.splitk > 1 ? d.splitk : d.batch;
    int group_m = 1;
    if (d.splitk > 1) group_m = -1;
    else if (nz == 1) {
        if (tn_ > 32) group_m = 2;
        else if (tn_ >= 6) group_m = 8;
        else if (tn_ >= 2 && tm_ >= 1024) group_m = 16;
    }
    hipLaunchKernelGGL(kern, dim3(tm_ * tn_, nz), dim3(P8_NT), lds, stream, d, group_m);
    ESVIT_CHECK_LAUNCH("esvit_gemm(p8)");
    if (d.splitk > 1) return launch_splitk_reduce(d, true, stream);
    return ESVIT_OK;
}

}  // namespace

// bf16 only; K % 64 == 0, no row map, operands below 4 GiB (32-bit DMA offsets): checked by esvit_gemm's dispatcher (gemm.hip)
int esvit_gemm_p8_launch(const esvit_gemm_desc& d, hipStream_t stream) {
    if (!d.a_kstrided && !d.b_kstrided) return launch_p8<false, false>(d, stream);
    if (!d.a_kstrided && d.b_kstrided) return launch_p8<false, true>(d, stream);
    return launch_p8<true, true>(d, stream);
}
