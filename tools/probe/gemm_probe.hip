// Experimental main-loop shapes of the LDS-DMA GEMM (NOT part of libesvit_hip.so): instantiations of the product's kernel
// template (esvit_amd/csrc/gemm_kernels.h) with other ring depths / k-tiles, for A/B timing on the GPU
// (tools/bench_gemm.py).  Built by tools/probe/build.sh into tools/probe/libgemm_probe.so.
#include "gemm_kernels.h"

extern "C" int probe_gemm(int variant, const esvit_gemm_desc* dp, void* stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    esvit_gemm_desc d = *dp;
    if (d.batch < 1) d.batch = 1;
    if (d.splitk < 1) d.splitk = 1;
    const bool nt = !d.a_kstrided && !d.b_kstrided, nn = !d.a_kstrided && d.b_kstrided, tn = d.a_kstrided && d.b_kstrided;
    switch (variant) {
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
    default:
        break;
    }
    return ESVIT_ERR_UNSUPPORTED;
}

void esvit_set_error(const char*, ...) {}
