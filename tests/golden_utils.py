"""Shared by oracle/gen_golden.py (which runs the reference) and the tests (which run the oracle and the HIP
path): deterministic parameter fill and synthetic crops, so goldens need to store outputs only."""
import os
import zlib

import torch

NANO = dict(embed_dim=32, depths=(2, 2, 2, 2), heads=(1, 2, 4, 8), window=7, img=224)
NANO_HEAD = dict(out_dim=4096, hidden_dim=256, bottleneck_dim=64)
NANO14 = dict(embed_dim=32, depths=(2, 2, 2, 2), heads=(1, 2, 4, 8), window=14, img=224)
# CvT (BASELINE config 5) in miniature: head_dim 64 everywhere, crops 112 / 48 -> grids 28,14,7 / 12 (padded to 14),6,3
NANO_CVT = dict(dims=(64, 128, 192), heads=(1, 2, 3), depths=(1, 2, 1), sizes=(112, 48))
# the other cvt_v4 yaml files (s1_rpe.yaml, s1_shift.yaml, s1_rpe_shift.yaml, res_stem/*): every stage's map must hold a whole window
# (the reference's bias / mask shapes) and, with SHIFT, be a multiple of it (the reference's masked path loses the map height when it
# has to crop a padded map): two stages; 112 / 56 crops -> grids 28, 14 / 14, 7; 112 / 64 crops -> 28, 14 / 16 (padded to 21), 8 (to 14)
NANO_CVT_VARIANTS = {
    "rpe_shift": dict(cfg=dict(dims=(64, 128), heads=(1, 2), depths=(2, 2), rel_pos_embed=True, shift=True), sizes=(112, 56), n_local=2),
    "shift": dict(cfg=dict(dims=(64, 128), heads=(1, 2), depths=(1, 2), shift=True), sizes=(112, 56), n_local=2),
    "rpe": dict(cfg=dict(dims=(64, 128), heads=(1, 2), depths=(1, 1), rel_pos_embed=True), sizes=(112, 64), n_local=2),
    # res_stem/s3_w14.yaml in miniature: 14x14 windows at head_dim 32 on the 28x28 maps, crops 112 / 48 -> 28, 14 / 12, 6
    "res_stem": dict(cfg=dict(dims=(64, 128), heads=(2, 4), depths=(1, 1), res_stem=True, windows=(14, 7)), sizes=(112, 48), n_local=2),
}
NANO_VIT = dict(embed_dim=64, depth=2, heads=2, patch=16, sizes=(64, 32))  # 16 + 1 and 4 + 1 tokens: the position embedding is interpolated
SWIN_T = dict(embed_dim=96, depths=(2, 2, 6, 2), heads=(3, 6, 12, 24), window=7, img=224)


def fill_state_dict(sd, seed=0):
    """In-place deterministic fill keyed by parameter name (independent of construction order / RNG stream)."""
    for name, t in sd.items():
        if not t.is_floating_point():
            continue
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + seed) % (2 ** 31))
        r = torch.randn(t.shape, generator=g)
        if name.endswith("weight_g"):
            v = 1.0 + 0.1 * r
        elif "norm" in name and name.endswith("weight"):
            v = 1.0 + 0.1 * r
        elif name.endswith("bias"):
            v = 0.02 * r
        elif "relative_position_bias_table" in name:
            v = 0.2 * r
        else:
            v = 0.05 * r
        t.copy_(v.to(t.dtype))
    return sd


def make_crops(B, n_local=8, seed=1234, sizes=(224, 96)):
    """torch.randn crops, list order [g, g, l x n_local]: the package's generator, so that bench.py needs nothing from tests/."""
    from esvit_amd.data import synthetic_crops
    return synthetic_crops(B, n_local=n_local, seed=seed, sizes=sizes)


def probe(t, n=16):
    """small fingerprint of a tensor: (shape, sum, abs-sum, first n and strided n values)"""
    f = t.detach().float().reshape(-1)
    stride = max(1, f.numel() // n)
    return dict(shape=tuple(t.shape), sum=f.double().sum().item(), asum=f.double().abs().sum().item(),
                head=f[:n].clone(), strided=f[::stride][:n].clone())


def make_knn_set(seed, n_train=1500, n_test=400, dim=48, classes=10, noise=1.0):
    """synthetic L2-normalised features: class centres + gaussian noise (deterministic in `seed`)"""
    g = torch.Generator().manual_seed(seed)
    centres = torch.randn(classes, dim, generator=g)
    ytr = torch.randint(0, classes, (n_train,), generator=g)
    yte = torch.randint(0, classes, (n_test,), generator=g)
    xtr = torch.nn.functional.normalize(centres[ytr] + noise * torch.randn(n_train, dim, generator=g), dim=1)
    xte = torch.nn.functional.normalize(centres[yte] + noise * torch.randn(n_test, dim, generator=g), dim=1)
    return xtr, ytr, xte, yte


# a training checkpoint written by the reference (utils.save_on_master of a DistributedDataParallel student and a plain teacher, three-stage
# nano Swin + V / R heads) and what the reference's eval_knn.py path makes of it (tests/golden/ref_checkpoint.{pth,pt})
REF_CKPT = dict(embed_dim=32, depths=(1, 1, 1), heads=(1, 2, 4), window=7, head=dict(out_dim=256, hidden_dim=64, bottleneck_dim=32),
                n_train=120, n_test=100, classes=6, size=64, k=40, T=0.07)


def ref_ckpt_data():
    """class-dependent synthetic images: a per-class pattern plus noise, so that the k-NN vote is not a coin toss"""
    c = REF_CKPT
    g = torch.Generator().manual_seed(4242)
    protos = torch.randn(c["classes"], 3, c["size"], c["size"], generator=g)
    ytr = torch.arange(c["n_train"]) % c["classes"]
    yte = torch.arange(c["n_test"]) % c["classes"]
    xtr = protos[ytr] + 6.0 * torch.randn(c["n_train"], 3, c["size"], c["size"], generator=g)
    xte = protos[yte] + 6.0 * torch.randn(c["n_test"], 3, c["size"], c["size"], generator=g)
    return xtr, ytr, xte, yte


KNN_CASES = [dict(seed=5, k=10, T=0.07, noise=2.0), dict(seed=6, k=20, T=0.07, noise=3.0), dict(seed=7, k=20, T=0.5, noise=4.0)]


# ---- DINOHead(use_bn=True) fixture (tests/golden/variants.pt; oracle/gen_golden.py:gen_variants) -------------------------
BN_HEAD = dict(in_dim=48, out_dim=96, hidden_dim=64, bottleneck_dim=32, rows=40)


def fill_bn_head(sd, seed):
    """fill_state_dict + sane BatchNorm entries (gamma around 1, positive running variance)"""
    fill_state_dict(sd, seed)
    for name, t in sd.items():
        if name.endswith("running_var"):
            t.copy_(t.abs() * 4 + 0.5)
        elif name.startswith(("mlp.1.", "mlp.4.")) and name.endswith("weight"):
            t.copy_(1.0 + 2.0 * t)
    return sd


def bn_head_inputs():
    g = torch.Generator().manual_seed(4242)
    x = torch.randn(BN_HEAD["rows"], BN_HEAD["in_dim"], generator=g)
    probe = torch.randn(BN_HEAD["rows"], BN_HEAD["out_dim"], generator=g)  # loss = sum(logits * probe)
    return x, probe


# ---- LARS / SGD fixture: a tiny parameter set, three seeded steps -------------------------------------------------------
LARS_SHAPES = {"w1": (33, 20), "b1": (33,), "w2": (7, 33), "g": (33,), "w3": (5, 3, 3, 3)}
LARS_SCHED = [(0.3, 1e-4), (0.2, 2e-4), (0.1, 3e-4)]  # (lr, weight decay of the regularised group) per step


def lars_case():
    """-> (params dict, [grads dict per step]); the first tensor's gradient is large enough to be clipped at 3.0"""
    g = torch.Generator().manual_seed(777)
    params = {n: torch.randn(s, generator=g) * 0.3 for n, s in LARS_SHAPES.items()}
    grads = [{n: torch.randn(s, generator=g) * (2.0 if n == "w1" else 0.05) for n, s in LARS_SHAPES.items()} for _ in LARS_SCHED]
    return params, grads


# ---- DINOLoss with mixup targets (main_esvit.py:518-534, 639-641) ---------------------------------------------------------
MIXUP = dict(B=6, ncrops=4, K=64, lams=(0.7, 0.35))  # the first two crops are mixed (num_mixup_views = 2), the others are not


MIXUP_SMOOTHING = 0.1  # the second fixture (tests/golden/mixup_smoothing.pt): --smoothing 0.1, and per-sample mixing ratios for crop 1


def mixup_case(smoothing=0.0):
    """-> (student logits [ncrops*B, K], teacher logits [2B, K], centre [1, K], [T_v [B, B] per crop]).  T_v of a mixed crop is
    what timm's Mixup returns for targets = arange(B): lam * onehot(a) + (1 - lam) * onehot(B - 1 - a) (batch mode pairs a sample
    with the flipped batch); un-mixed crops get the identity, as train_one_epoch builds it.  With label smoothing the one-hot rows
    are timm's `one_hot(x, num_classes, on_value = 1 - smoothing + off, off_value = off = smoothing / num_classes)` (num_classes is
    the batch size, main_esvit.py:230), and the second mixed crop uses per-sample ratios (mixup_mode 'elem')."""
    c = MIXUP
    g = torch.Generator().manual_seed(9090)
    s = torch.randn(c["ncrops"] * c["B"], c["K"], generator=g)
    t = torch.randn(2 * c["B"], c["K"], generator=g)
    center = torch.randn(1, c["K"], generator=g) * 0.1
    eye = torch.eye(c["B"])
    if smoothing == 0.0:
        T = [lam * eye + (1 - lam) * eye.flip(0) for lam in c["lams"]]
    else:
        B = c["B"]
        off = smoothing / B
        on = 1.0 - smoothing + off
        idx = torch.arange(B).view(-1, 1)
        y1 = torch.full((B, B), off).scatter_(1, idx, on)
        y2 = torch.full((B, B), off).scatter_(1, idx.flip(0), on)
        lam_elem = torch.rand(B, 1, generator=g) * 0.6 + 0.2
        T = [y1 * c["lams"][0] + y2 * (1.0 - c["lams"][0]), y1 * lam_elem + y2 * (1.0 - lam_elem)]
    T += [eye.clone() for _ in range(c["ncrops"] - len(c["lams"]))]
    return s, t, center, T


# ---- USE_APE: three-stage nano Swin at 112^2 -------------------------------------------------------------------------------------------
def patch_norm_inputs(num_features, B=2):
    """PATCH_NORM False fixture: one 112^2 batch for forward_features, a (112^2, 64^2) crop pair for the multi-crop forward"""
    g = torch.Generator().manual_seed(2929)
    return (torch.randn(B, 3, 112, 112, generator=g), torch.randn(B, 3, 64, 64, generator=g), torch.randn(B, num_features, generator=g),
            torch.randn(2 * B, num_features, generator=g))


def ape_inputs(num_features, B=3):
    g = torch.Generator().manual_seed(1717)
    return torch.randn(B, 3, 112, 112, generator=g), torch.randn(B, num_features, generator=g)


# ---- full-width fixtures (tests/golden/full_width.pt; oracle/gen_golden.py:gen_full) -----------------------------------------
FULL_CASES = {
    # name: (dense prediction, out_dim, batch, student seed, teacher seed, crop seed, n crops used, head_dense norm_last_layer)
    "swin_t_k65536_b8": dict(dense=True, K=65536, B=8, s_seed=31, t_seed=32, crop_seed=55, ncrops=10),
    "swin_t_k8192_b2": dict(dense=True, K=8192, B=2, s_seed=3, t_seed=4, crop_seed=99, ncrops=10),
    "config1_bs4": dict(dense=False, K=65536, B=4, s_seed=11, t_seed=12, crop_seed=21, ncrops=2),
}
FULL_SAMPLE = 32768  # elements kept per sampled gradient tensor

# ---- full-width fixtures of BASELINE configs 3-5 (tests/golden/full_configs.pt; oracle/gen_golden.py:gen_full_configs) ---------
# arch: the name esvit_amd.config.model_config builds; yaml: the reference experiment file the fixture was generated from
FULL_CFG_CASES = {
    "swin_t_w14_k8192_b2": dict(arch="swin_tiny_w14", yaml="experiments/imagenet/swin/swin_tiny_patch4_window14_224.yaml",
                                K=8192, B=2, s_seed=41, t_seed=42, crop_seed=77),
    "swin_b_w14_k8192_b2": dict(arch="swin_base_w14", yaml="experiments/imagenet/swin/swin_base_patch4_window14_224.yaml",
                                K=8192, B=2, s_seed=43, t_seed=44, crop_seed=78),
    "cvt13_s1_k8192_b2": dict(arch="cvt_s1", yaml="experiments/imagenet/cvt_v4/s1.yaml",
                              K=8192, B=2, s_seed=45, t_seed=46, crop_seed=79),
}
# Vision Longformer at full width: experiments/imagenet/vil/vil_tiny/base.yaml
FULL_VIL_CASES = {
    "vil_tiny_k8192_b2": dict(arch="vil_tiny", yaml="experiments/imagenet/vil/vil_tiny/base.yaml", K=8192, B=2, s_seed=47, t_seed=48, crop_seed=80),
    # the other yaml of the directory (heads 3-3-6-12, widths 96..768): asked for by the round-3 verdict
    "vil_small_k8192_b2": dict(arch="vil_small", yaml="experiments/imagenet/vil/vil_small/base.yaml", K=8192, B=2, s_seed=49, t_seed=50, crop_seed=81),
}
FULL_CFG_SAMPLE = 8192


def fill_full_cfg_pair(student, teacher, c):
    """the deterministic parameters / buffers of a FULL_CFG_CASES entry (reference modules in gen_golden, ours in the tests)"""
    fill_state_dict(student.state_dict(), c["s_seed"])
    fill_state_dict(teacher.state_dict(), c["t_seed"])
    for m in (student, teacher):
        for k, v in m.state_dict().items():
            if k.endswith("running_var"):  # BatchNorm (CvT): a valid variance
                v.abs_().add_(0.5)
    student.head.last_layer.weight_g.data.fill_(1)
    student.head_dense.last_layer.weight_g.data.fill_(1)
    for p in teacher.parameters():
        p.requires_grad = False


def strided(t, n=FULL_SAMPLE):
    f = t.detach().reshape(-1)
    return f[:: max(1, f.numel() // n)][:n].clone()


def full_sampled_names(names):
    return names[:: max(1, len(names) // 10)][:10] + [n for n in ("head.last_layer.weight_v", "head_dense.last_layer.weight_v") if n in names]


# ---- observed parity deltas (GPU runs) ----------------------------------------------------------------------------------
PARITY_OBSERVED = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_observed.jsonl")


def record_parity(**kw):
    """observed deltas of a -m gpu parity test: printed (pytest -s / -rP) and appended to gpurun_out/parity_observed.jsonl when
    that directory exists; the bf16 bounds asserted in the tests are <= 3x the values committed in profiles/r03_parity_observed.jsonl"""
    import json
    line = json.dumps(kw)
    print("PARITY", line)
    try:
        if os.path.isdir(os.path.dirname(PARITY_OBSERVED)):
            with open(PARITY_OBSERVED, "a") as fh:
                fh.write(line + "\n")
    except OSError:
        pass


# ---- linear probe (eval_linear.py) fixture: tests/golden/linear_probe.pt --------------------------------------------------------
LINEAR_PROBE = dict(n_last_blocks=2, avgpool=False, num_labels=12, batches=3, batch=4, lr=0.05, crop=224, seed=2024)


def linear_probe_data():
    """-> (train batches, val batches) of (images, labels) for the linear-probe fixture"""
    c = LINEAR_PROBE
    g = torch.Generator().manual_seed(c["seed"])
    mk = lambda: (torch.randn(c["batch"], 3, c["crop"], c["crop"], generator=g), torch.randint(0, c["num_labels"], (c["batch"],), generator=g))  # noqa: E731
    return [mk() for _ in range(c["batches"])], [mk() for _ in range(2)]


def linear_probe_init(clf, seed=5):
    g = torch.Generator().manual_seed(seed)
    clf.linear.weight.data.copy_(0.01 * torch.randn(clf.linear.weight.shape, generator=g))
    clf.linear.bias.data.zero_()


# ---- DINOHead(nlayers != 3) (vision_transformer.py:388-402) -------------------------------------------------------------
HEAD_NLAYERS = dict(in_dim=48, out_dim=512, hidden_dim=64, bottleneck_dim=32, rows=24, cases=(1, 2, 4))


HEAD_BN_NLAYERS = (1, 2, 4)  # DINOHead(use_bn=True) beside the default nlayers = 3 (tests/golden/head_bn_nlayers.pt)


def fill_bn_head_n(sd, seed):
    """fill_state_dict + sane BatchNorm entries for any depth (BatchNorm1d weights are the 1-d `mlp.N.weight` tensors)"""
    fill_state_dict(sd, seed)
    for name, t in sd.items():
        if name.endswith("running_var"):
            t.copy_(t.abs() * 4 + 0.5)
        elif name.startswith("mlp.") and name.endswith("weight") and t.dim() == 1:
            t.copy_(1.0 + 2.0 * t)
    return sd


def head_nlayers_inputs():
    g = torch.Generator().manual_seed(515)
    c = HEAD_NLAYERS
    return torch.randn(c["rows"], c["in_dim"], generator=g), torch.randn(c["rows"], c["out_dim"], generator=g)
