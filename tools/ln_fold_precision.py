"""What folding LayerNorm into the following GEMM would cost in precision (CPU, needs /root/reference).

    python tools/ln_fold_precision.py

For every Swin-T block (the reference's own module, deterministic fixture fill, 2 x 224^2 random crops) the input x of norm1 and the
qkv weight are taken and three versions of  y = LayerNorm(x) W^T  are compared with the fp32 result:
  shipped:  bf16(LayerNorm(x)) . bf16(W)^T                                    (what ln_fwd + the MFMA GEMM compute; fp32 accumulation)
  folded:   rstd * (bf16(x) . bf16(gamma o W)^T) - rstd * mu * colsum(bf16(gamma o W)) + beta W^T     (GEMM on the raw bf16 residual)
  folded, centred rows: the same with bf16(x - rowmean(x))  (would need the mean before the GEMM: not a fusion any more)
Relative error = max |y - y_fp32| / max |y_fp32| over the block's rows.  The last tables add outlier channels (a per-channel offset:
it is part of the row's variance and survives LayerNorm) and a row mean (a per-row offset: LayerNorm removes it exactly, the bf16
rounding of the raw x does not) of k sigma to the activations of one block."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_loader as RL  # noqa: E402
from tests import golden_utils as GU  # noqa: E402


def bf(t):
    return t.to(torch.bfloat16).float()


def versions(x, ln, W):
    g, b, eps = ln.weight, ln.bias, ln.eps
    mu = x.mean(-1, keepdim=True)
    var = x.var(-1, unbiased=False, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    xn = (x - mu) * rstd * g + b
    ref = xn @ W.t()
    shipped = bf(xn) @ bf(W).t()
    gw = bf(g * W)
    folded = rstd * (bf(x) @ gw.t()) - rstd * mu * gw.sum(1) + b @ W.t()
    centred = rstd * (bf(x - mu) @ gw.t()) + b @ W.t()
    s = ref.abs().max()
    return [((v - ref).abs().max() / s).item() for v in (shipped, folded, centred)]


def main():
    torch.manual_seed(0)
    ns = RL.load()
    cfg = RL.swin_config()
    m = ns.models.build_model(cfg, is_teacher=True, use_dense_prediction=True)
    GU.fill_state_dict(m.state_dict(), 31)
    m.eval()
    grabbed = []
    for li, layer in enumerate(m.layers):
        for bi, blk in enumerate(layer.blocks):
            blk.norm1.register_forward_hook(lambda mod, inp, out, tag=(li, bi), blk=blk: grabbed.append((tag, inp[0].detach().reshape(-1, inp[0].shape[-1]), mod, blk.attn.qkv.weight.detach())))
    with torch.no_grad():
        m.forward_features(GU.make_crops(2, n_local=0)[0])
    print("block     rows     C   |mean|/sigma of x   shipped     folded      folded, centred rows")
    for (li, bi), x, ln, W in grabbed:
        ratio = (x.mean(0).abs() / x.std(0)).mean().item()
        e = versions(x, ln, W)
        print("s%d b%d  %7d  %4d   %8.2f          %.2e   %.2e   %.2e" % (li, bi, x.shape[0], x.shape[1], ratio, *e))
    (li, bi), x, ln, W = [g for g in grabbed if g[0] == (2, 5)][0]
    print("\nstage 2 block 5 with a per-CHANNEL offset of k sigma added to x (outlier channels: part of the row's variance):")
    for k in (0, 4, 16, 64):
        off = k * x.std(0) * torch.sign(torch.randn(x.shape[1]))
        e = versions(x + off, ln, W)
        print("k = %3d   shipped %.2e   folded %.2e   folded, centred rows %.2e" % (k, *e))
    print("\n... and with a per-ROW offset of k row-sigma (a row mean LayerNorm removes exactly; the bf16 rounding of the raw x does not):")
    for k in (0, 1, 4, 16, 64):
        off = k * x.std(1, keepdim=True)
        e = versions(x + off, ln, W)
        print("k = %3d   shipped %.2e   folded %.2e   folded, centred rows %.2e" % (k, *e))
    rm = (x.mean(1).abs() / x.std(1))
    print("\nrow |mean| / row sigma of the fixture's activations: median %.3f, max %.3f" % (rm.median().item(), rm.max().item()))

if __name__ == "__main__":
    main()
