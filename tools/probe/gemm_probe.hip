// Experimental main-loop shapes of the LDS-DMA GEMM (NOT part of libesvit_hip.so): instantiations of the product's kernel
// template (esvit_amd/csrc/gemm_kernels.h) with other ring depths / k-tiles, for A/B timing on the GPU
// (tools/bench_gemm.py).  Built by tools/probe/build.sh into tools/probe/libgemm_probe.so.
#define ESVIT_PROBE_TIMELINE 1
#include "gemm_kernels.h"

// timeline buffer: 8 longs per workgroup [t_start, t_mainloop_end, t_end, hw_id, xcc_id, -, -, -]
extern "C" int probe_set_timeline(long* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_probe_timeline), &buf, sizeof(buf));
}

extern "C" int probe_gemm(int variant, const esvit_gemm_desc* dp, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    esvit_gemm_desc d = *dp;
    if (d.batch < 1) d.batch = 1;
    if (d.splitk < 1) d.splitk = 1;
    const bool nt = !d.a_kstrided && !d.b_kstrided, nn = !d.a_kstrided && d.b_kstrided, tn = d.a_kstrided && d.b_kstrided;
    switch (variant) {
#ifndef ESVIT_PROBE_ONLY7
    case 1:  // the product's 8-wave shape: 256 x 256, BK 64, 2 buffers
        if (nt) return launch_gemm_dma<false, false, 256, 256, 64, 2, 2, 4>(d, stream);
        if (nn) return launch_gemm_dma<false, true, 256, 256, 64, 2, 2, 4>(d, stream);
        if (tn) return launch_gemm_dma<true, true, 256, 256, 64, 2, 2, 4>(d, stream);
        break;
    case 2:  // BK 32, 4-deep ring (two k-tiles in flight across each barrier)
        if (nt) return launch_gemm_dma<false, false, 256, 256, 32, 4, 2, 4>(d, stream);
        if (nn) return launch_gemm_dma<false, true, 256, 256, 32, 4, 2, 4>(d, stream);
        if (tn) return launch_gemm_dma<true, true, 256, 256, 32, 4, 2, 4>(d, stream);
        break;
    case 3:  // BK 32, 3-deep ring
        if (nt) return launch_gemm_dma<false, false, 256, 256, 32, 3, 2, 4>(d, stream);
        if (nn) return launch_gemm_dma<false, true, 256, 256, 32, 3, 2, 4>(d, stream);
        if (tn) return launch_gemm_dma<true, true, 256, 256, 32, 3, 2, 4>(d, stream);
        break;
    case 4:  // 256 x 128, BK 64, 3-deep ring (144 KiB)
        if (nt) return launch_gemm_dma<false, false, 256, 128, 64, 3, 4, 2>(d, stream);
        if (nn) return launch_gemm_dma<false, true, 256, 128, 64, 3, 4, 2>(d, stream);
        if (tn) return launch_gemm_dma<true, true, 256, 128, 64, 3, 4, 2>(d, stream);
        break;
    case 5:  // 256 x 256 as 4 x 2 waves of 64 x 128
        if (nt) return launch_gemm_dma<false, false, 256, 256, 64, 2, 4, 2>(d, stream);
        if (nn) return launch_gemm_dma<false, true, 256, 256, 64, 2, 4, 2>(d, stream);
        if (tn) return launch_gemm_dma<true, true, 256, 256, 64, 2, 4, 2>(d, stream);
        break;
    case 6:  // 4 waves, 128 x 128 (the product's small shape) for reference
        if (nt) return launch_gemm_dma<false, false, 128, 128, 64, 2, 2, 2>(d, stream);
        if (nn) return launch_gemm_dma<false, true, 128, 128, 64, 2, 2, 2>(d, stream);
        if (tn) return launch_gemm_dma<true, true, 128, 128, 64, 2, 2, 2>(d, stream);
        break;
#endif
    case 7:  // the product's default: 128 x 128, early buffer release
        if (nt) return launch_gemm_dma<false, false, 128, 128, 64, 2, 2, 2, 2, true>(d, stream);
        if (nn) return launch_gemm_dma<false, true, 128, 128, 64, 2, 2, 2, 2, true>(d, stream);
        if (tn) return launch_gemm_dma<true, true, 128, 128, 64, 2, 2, 2, 2, true>(d, stream);
        break;
    default:
        break;
    }
    return ESVIT_ERR_UNSUPPORTED;
}

void esvit_set_error(const char*, ...) {}

// ---- L2 -> CU delivery microbenchmark: how many bytes per clock and CU the LDS-DMA path (buffer_load ... lds) and the
// register path (global_load_dwordx4) sustain when the data are L2-resident (each workgroup re-reads a small window).
// mode 0: LDS-DMA, 1: loads to registers.  bytes_per_wg = window each workgroup cycles through.
namespace {
template <int MODE>
__global__ __launch_bounds__(256, 2) void l2_stream_kernel(const char* __restrict__ src, long window, int iters, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef __attribute__((address_space(3))) void lds_void;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = src + ((long)blockIdx.x % 64) * window;  // 64 windows: L2-resident working set
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)window, 0x00020000);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // each iteration: the workgroup moves 32 KB (8 instructions of 1 KB per wave), as one k-tile of the 128 x 128 GEMM does
    for (int it = 0; it < iters; ++it) {
        const int off0 = (int)(((long)it * 32768) % window);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int off = off0 + (wave * 8 + i) * 1024 + lane * 16;
            if constexpr (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(lds + ((it & 1) * 32768) + (wave * 8 + i) * 1024), 16, off, 0, 0, 0);
            } else {
                const f32x4 v = *reinterpret_cast<const f32x4*>(base + off);
                acc += v;
            }
        }
        if constexpr (MODE == 0) {
            if (it & 1) {  // keep two "k-tiles" in flight, then drain: same depth as the GEMM ring
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (MODE == 0) acc[0] = reinterpret_cast<float*>(lds)[threadIdx.x];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}
}  // namespace

// ---- how the delivery rate depends on the bytes in flight and on where the data lands (the question behind the GEMM main loop:
// tools/gemm_timeline.py shows a k-tile taking ~1 us with or without its MFMAs and fragment reads).
// A workgroup of 4 waves streams "operand tiles" of 128 rows x 64 bf16 (16 KB; row pitch = pitch bytes, GEMM-like: 8 lanes per
// 128-byte row piece) -- DEPTH of them in flight -- either by LDS-DMA into a ring (MODE 0) or into registers (MODE 1: a wave takes its
// 32 rows as MFMA-fragment-shaped 16-byte loads, row c / k-chunk g, two k-halves, two row blocks = 4 instructions).  `rows_per_wg`
// rows are walked k-tile by k-tile (ktiles per row block), then the next 128 rows; wg_stride_rows = distance between the row
// ranges of consecutive workgroups (0: all workgroups read the same rows = cache resident).
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256, 2) void inflight_kernel(const char* __restrict__ src, long pitch, int ktiles, int row_blocks, long wg_stride_rows,
                                                          float* __restrict__ sink, int order) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef __attribute__((address_space(3))) void lds_void;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const char* base = src + (long)blockIdx.x * wg_stride_rows * pitch;
    const long span = (long)row_blocks * 128 * pitch;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)(span > 0x7fffffffL ? 0x7fffffffL : span), 0x00020000);
    const int total = ktiles * row_blocks;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto tile_off = [&](int t) -> int { return (int)((long)(t / ktiles) * 128 * pitch + (long)(t % ktiles) * 128); };
    if constexpr (MODE == 0) {
        // per wave: 4 instructions per tile; instruction q covers rows (wave * 4 + q) * 8 .. +7, lane: row l / 8, 16-byte piece l % 8
        int voff[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = (wave * 4 + q) * 8 + lane / 8;
            int ch = lane % 8;  // 16-byte piece of the row's 128-byte line this lane fetches
            if (order == 1) ch ^= row & 7;              // the GEMM's XOR swizzle (sw_kc)
            else if (order == 2) ch = (ch + row) & 7;   // rotation: ascending with one wrap
            else if (order == 3) ch ^= (row >> 1) & 3;  // XOR confined to 64-byte halves... (pairs of lanes stay adjacent)
            voff[q] = (int)(row * pitch + ch * 16);
        }
        auto issue = [&](int t) {
            const int so = tile_off(t);
            char* dst = lds + (t % DEPTH) * 16384;
#pragma unroll
            for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + (wave * 4 + q) * 1024), 16, voff[q], so, 0, 0);
        };
#pragma unroll
        for (int t = 0; t < DEPTH - 1; ++t)
            if (t < total) issue(t);
        for (int t = 0; t < total; ++t) {
            if (t + DEPTH - 1 < total) {
                issue(t + DEPTH - 1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * 4) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (DEPTH <= 2) __builtin_amdgcn_s_barrier();  // (a ring this shallow needs the consumer's barrier; deeper ones are left free-running)
        }
        __syncthreads();
        acc[0] = reinterpret_cast<float*>(lds)[threadIdx.x];
    } else {
        int voff[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) voff[q] = (int)((wave * 32 + (q >> 1) * 16 + c) * pitch + ((q & 1) * 4 + g) * 16);
        f32x4 ring[DEPTH][4];
        auto issue = [&](auto slot, int t) {
            constexpr int sl = decltype(slot)::value;
            const int so = tile_off(t);
#pragma unroll
            for (int q = 0; q < 4; ++q) ring[sl][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[q], so, 0));
        };
        static_for<DEPTH - 1>([&](auto sc) {
            constexpr int t = decltype(sc)::value;
            if (t < total) issue(sc, t);
        });
        for (int t0 = 0; t0 < total; t0 += DEPTH) {
            static_for<DEPTH>([&](auto sc) {
                constexpr int sl = decltype(sc)::value;
                const int t = t0 + sl;
                constexpr int nsl = (sl + DEPTH - 1) % DEPTH;
                if (t + DEPTH - 1 < total) issue(std::integral_constant<int, nsl>{}, t + DEPTH - 1);
                if (t < total) {
                    // consume slot sl (the compiler places the counted wait)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc += ring[sl][q];
                }
            });
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

template <int MODE, int DEPTH>
static int launch_inflight(const void* src, long pitch, int ktiles, int row_blocks, long wg_stride_rows, int wgs, float* sink, hipStream_t stream, int order) {
    auto kern = inflight_kernel<MODE, DEPTH>;
    const int lds = MODE == 0 ? DEPTH * 16384 : 0;
    if (lds) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, stream, (const char*)src, pitch, ktiles, row_blocks, wg_stride_rows, sink, order);
    return (int)hipGetLastError();
}

// order (LDS-DMA mode): which 16-byte piece of its row's line a lane fetches -- 0 ascending, 1 the GEMM's XOR swizzle, 2 rotation, 3 XOR of piece pairs
extern "C" int probe_inflight(int mode, int depth, const void* src, long pitch, int ktiles, int row_blocks, long wg_stride_rows, int wgs, float* sink,
                              void* stream_, int order) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
#define CASE(M_, D_) if (mode == M_ && depth == D_) return launch_inflight<M_, D_>(src, pitch, ktiles, row_blocks, wg_stride_rows, wgs, sink, stream, order);
    CASE(0, 1) CASE(0, 2) CASE(0, 3) CASE(0, 4) CASE(0, 5)
    CASE(1, 1) CASE(1, 2) CASE(1, 4) CASE(1, 6) CASE(1, 8) CASE(1, 12)
#undef CASE
    return -1;
}

extern "C" int probe_l2_stream(int mode, const void* src, long window, int iters, int wgs, float* sink, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (mode == 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(l2_stream_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipLaunchKernelGGL(l2_stream_kernel<0>, dim3(wgs), dim3(256), 65536, stream, (const char*)src, window, iters, sink);
    } else {
        hipLaunchKernelGGL(l2_stream_kernel<1>, dim3(wgs), dim3(256), 0, stream, (const char*)src, window, iters, sink);
    }
    return (int)hipGetLastError();
}
