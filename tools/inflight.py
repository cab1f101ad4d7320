"""Delivery rate into a CU as a function of the bytes in flight and of the landing zone (LDS-DMA ring vs registers), on GEMM-shaped
operand tiles (tools/probe/libgemm_probe.so: probe_inflight; tools/probe/build.sh first).  Prints one JSON line per case."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libgemm_probe.so"))
lib.probe_inflight.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
sink = torch.zeros(4, device=dev)
K = 1536
pitch = K * 2
ktiles = K // 64
wgs = 512
for source, row_blocks, stride in (("hbm-stream", 4, 4 * 128), ("l2-resident", 4, 0)):
    rows = (wgs * stride if stride else 0) + row_blocks * 128
    src = (torch.randn(rows * K // 2 + 1024, device=dev)).view(-1)
    cases = [(0, "lds-dma", d, 0) for d in (1, 2, 3, 4, 5)] + [(1, "registers", d, 0) for d in (1, 2, 4, 6, 8, 12)]
    cases += [(0, "lds-dma", 2, o) for o in (1, 2, 3)]   # lane order inside a 128-byte line: XOR swizzle / rotation / pair XOR
    if True:
        for mode, name, depth, order in cases:
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            reps = 1 if stride else 8
            def run():
                for _ in range(reps):
                    rc = lib.probe_inflight(mode, depth, src.data_ptr(), pitch, ktiles, row_blocks, stride, wgs, sink.data_ptr(), st, order)
                    assert rc == 0, rc
            run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            run()
            e.record()
            torch.cuda.synchronize()
            sec = s.elapsed_time(e) * 1e-3
            total = reps * wgs * row_blocks * ktiles * 16384
            print(json.dumps({"source": source, "path": name, "lane_order": ["ascending", "xor", "rotate", "pair-xor"][order], "tiles_in_flight_per_wg": depth, "kb_in_flight_per_wg": depth * 16,
                              "TB_per_s": round(total / sec / 1e12, 2), "B_per_clk_per_cu_at_2.4GHz": round(total / sec / 256 / 2.4e9, 1),
                              "us_per_16KB_tile_per_wg": round(sec / reps / (row_blocks * ktiles) * 1e6, 3)}), flush=True)
    del src
