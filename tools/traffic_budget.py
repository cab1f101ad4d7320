"""Activation traffic of the Swin-T W7 step per stage and branch under the CURRENT dataflow (bytes each kernel reads / writes per
token-channel, listed below) against a block-fused dataflow (forward: read x, write y = 8 B; backward: read x, dy, write dx = 12 B,
weight gradients accumulated on the chip).  Algorithmic bytes of activations only (weights, index maps, split-K partials and cache
effects are in the PMC numbers of profiles/r05_pmc_traffic.json, not here).  python tools/traffic_budget.py [batch]"""
import sys

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
C = [96, 192, 384, 768]
depth = [2, 2, 6, 2]
tok_s = [B * (2 * 3136 + 8 * 576) // 4 ** s for s in range(4)]   # student rows per stage (2 x 224^2 + 8 x 96^2 crops)
tok_t = [B * 2 * 3136 // 4 ** s for s in range(4)]               # teacher rows (2 x 224^2)
# bytes per token-channel, current dataflow (DESIGN.md 12 lists the tensors behind every number)
attn_fwd_s = [18, 28, 32, 32]     # stage 0: fused kernel with side outputs; stage 1: unfused, LayerNorm output from the previous fused MLP; 2-3: unfused
attn_fwd_t = [8, 8, 32, 32]       # stages 0-1: fused kernel
attn_bwd_s = [54, 54, 54, 54]
mlp_fwd_s = [8, 10, 40, 40]       # stages 0-1: fused; 2-3: LayerNorm + fc1 (GELU output and pre-activation) + fc2
mlp_fwd_t = [8, 8, 32, 32]
mlp_bwd_s = [52, 52, 64, 64]      # stages 0-1: fused data-gradient kernel + the hidden-sized operands of the two weight-gradient GEMMs
ideal_f, ideal_b = 8, 12
rows = []
tot_c = tot_i = 0.0
for s in range(4):
    ns, nt = tok_s[s] * C[s] * depth[s], tok_t[s] * C[s] * depth[s]
    for name, cur, ideal in (("attention fwd (student)", attn_fwd_s[s] * ns, ideal_f * ns), ("attention fwd (teacher)", attn_fwd_t[s] * nt, ideal_f * nt),
                             ("attention bwd", attn_bwd_s[s] * ns, ideal_b * ns), ("MLP fwd (student)", mlp_fwd_s[s] * ns, ideal_f * ns),
                             ("MLP fwd (teacher)", mlp_fwd_t[s] * nt, ideal_f * nt), ("MLP bwd", mlp_bwd_s[s] * ns, ideal_b * ns)):
        rows.append((s, name, cur / 1e9, ideal / 1e9))
        tot_c += cur / 1e9
        tot_i += ideal / 1e9
print("| stage (C, blocks) | branch | current GB | block-fused GB |")
print("|---|---|---|---|")
for s, name, cur, ideal in rows:
    print("| %d (%d, %d) | %s | %.2f | %.2f |" % (s, C[s], depth[s], name, cur, ideal))
print("| all | blocks of the backbone | %.1f | %.1f |" % (tot_c, tot_i))
