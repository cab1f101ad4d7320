"""L2 -> CU delivery rate of the LDS-DMA path vs loads to registers (tools/probe/libgemm_probe.so: probe_l2_stream)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libgemm_probe.so"))
lib.probe_l2_stream.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
sink = torch.zeros(4, device=dev)
for window_kb in (64, 256, 1024, 8192):
    window = window_kb * 1024
    src = torch.randn(64 * window // 4, device=dev)
    for mode, name in ((0, "lds-dma"), (1, "registers")):
        for wgs in (256, 512):
            iters = 2048
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(2):
                lib.probe_l2_stream(mode, src.data_ptr(), window, iters, wgs, sink.data_ptr(), st)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            lib.probe_l2_stream(mode, src.data_ptr(), window, iters, wgs, sink.data_ptr(), st)
            e.record()
            torch.cuda.synchronize()
            sec = s.elapsed_time(e) * 1e-3
            total = wgs * iters * 32768
            print(json.dumps({"window_kb_per_wg": window_kb, "working_set_mb": 64 * window_kb / 1024, "path": name, "workgroups": wgs,
                              "TB_per_s": round(total / sec / 1e12, 2), "B_per_clk_per_cu_at_2.4GHz": round(total / sec / 256 / 2.4e9, 1)}), flush=True)
    del src
