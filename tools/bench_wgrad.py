"""Micro-benchmark of the weight-gradient GEMM (both operands k-strided, split-K) on the step's shapes.
usage: python tools/bench_wgrad.py [batch] [pipes...]   -- one JSON line per shape with the time under each LDS-DMA pipeline."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esvit_amd import ops
from tools.bench_kernels import timeit

dev = torch.device("cuda:0")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    pipes = [int(x) for x in sys.argv[2:]] or [1]
    shapes = []
    for C, L, Ll in [(96, 3136, 576), (192, 784, 144), (384, 196, 36), (768, 49, 9)]:
        for tok in (2 * B * L, 8 * B * Ll):
            shapes += [(3 * C, C, tok), (C, C, tok), (4 * C, C, tok), (C, 4 * C, tok)]
    tot = {p: 0.0 for p in pipes}
    for Nout, Kin, tok in shapes:
        dy = torch.randn(tok, Nout, device=dev).to(torch.bfloat16)
        x = torch.randn(tok, Kin, device=dev).to(torch.bfloat16)
        row = dict(Nout=Nout, Kin=Kin, tokens=tok)
        for p in pipes:
            ops.lib.esvit_debug_set_gemm_pipe(p)
            t = timeit(lambda: ops.linear_wgrad(dy, x, want_bias=True))
            row["us_pipe%d" % p] = round(t * 1e6, 1)
            tot[p] += t * 1e6
        ops.lib.esvit_debug_set_gemm_pipe(1)
        print(json.dumps(row))
        del dy, x
    print(json.dumps({"total_us": {p: round(v) for p, v in tot.items()}}))


if __name__ == "__main__":
    main()
