"""Generate tests/golden/*.pt from the REFERENCE's own modules (run in the build container only):

    python -m oracle.gen_golden

Fixtures are small (fingerprints + a few full tensors); inputs and weights are regenerated in the
tests from tests/golden_utils.py, so nothing large is committed."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader as RL  # noqa: E402
from tests import golden_utils as GU  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def gen_index_maps(ns):
    swin = ns.models.swin_transformer
    out = {}
    for ws in (7, 14):
        attn = swin.WindowAttention(32, (ws, ws), 1)
        out["rpi_%d" % ws] = attn.relative_position_index.numpy().astype(np.int64)
        for H in (56, 28, 14, 7, 24, 12, 6, 3):
            for shift in (0, ws // 2):
                blk = swin.SwinTransformerBlock(32, (1024, 1024), 1, window_size=ws, shift_size=shift)
                blk.norm1 = torch.nn.Identity()
                cap = {}

                def hook(mod, args, cap=cap):
                    cap["xw"], cap["mask"] = args[0].detach().clone(), (None if args[1] is None else args[1].detach().clone())

                h = blk.attn.register_forward_pre_hook(hook)
                x = torch.zeros(1, H * H, 32)
                x[0, :, 0] = torch.arange(H * H, dtype=torch.float32) + 1.0
                with torch.no_grad():
                    blk(x)
                h.remove()
                ids = cap["xw"][..., 0].round().to(torch.int64).reshape(-1) - 1
                out["win2tok_%d_%d_%d" % (ws, H, shift)] = ids.numpy().astype(np.int32)
                if shift > 0:
                    out["mask_%d_%d" % (ws, H)] = cap["mask"].numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "index_maps.npz"), **out)
    print("index_maps.npz:", len(out), "arrays")


def build_nano(ns, teacher=False, window=None):
    cfg = RL.swin_config(embed_dim=GU.NANO["embed_dim"], depths=GU.NANO["depths"], heads=GU.NANO["heads"],
                         window=window or GU.NANO["window"])
    m = ns.models.build_model(cfg, is_teacher=teacher, use_dense_prediction=True)
    hk = dict(hidden_dim=GU.NANO_HEAD["hidden_dim"], bottleneck_dim=GU.NANO_HEAD["bottleneck_dim"])
    m.head = ns.DINOHead(m.num_features, GU.NANO_HEAD["out_dim"], norm_last_layer=True, **hk)
    m.head_dense = ns.DINOHead(m.num_features, GU.NANO_HEAD["out_dim"], norm_last_layer=False, **hk)
    return m


def gen_nano(ns):
    RL.ensure_single_process_group()
    torch.manual_seed(0)
    student, teacher = build_nano(ns), build_nano(ns, teacher=True)
    GU.fill_state_dict(student.state_dict(), seed=0)
    GU.fill_state_dict(teacher.state_dict(), seed=7)  # a different teacher so that the EMA is visible
    student.head.last_layer.weight_g.data.fill_(1)    # norm_last_layer=True keeps g == 1 (vision_transformer.py:404)
    for p in teacher.parameters():
        p.requires_grad = False
    B = 2
    crops = GU.make_crops(B)
    K = GU.NANO_HEAD["out_dim"]
    g = {"keys": [(k, tuple(v.shape), str(v.dtype)) for k, v in student.state_dict().items()],
         "param_names": [n for n, _ in student.named_parameters()],
         "trainable": [n for n, p in student.named_parameters() if p.requires_grad]}

    loss_fn = ns.DDINOLoss(K, 10, 0.04, 0.07, 5, 10)
    loss_fn.center.copy_(0.01 * torch.randn(1, K, generator=torch.Generator().manual_seed(5)))
    loss_fn.center_grid.copy_(0.01 * torch.randn(1, K, generator=torch.Generator().manual_seed(6)))
    g["center0"], g["center_grid0"] = loss_fn.center.clone(), loss_fn.center_grid.clone()
    epoch = 2
    t_out = teacher(crops[:2])
    s_out = student(crops)
    g["s_cls"], g["s_reg"], g["s_fea"] = GU.probe(s_out[0]), GU.probe(s_out[1]), GU.probe(s_out[2])
    g["t_cls"], g["t_reg"], g["t_fea"] = GU.probe(t_out[0]), GU.probe(t_out[1]), GU.probe(t_out[2])
    g["s_cls_full"] = s_out[0].detach().clone()          # [20, K] fp32 = 320 KB
    g["s_reg_rows"] = s_out[1][::17].detach().clone()    # 20 rows
    g["npatch"] = (list(s_out[3]), list(t_out[3]))
    loss = loss_fn(s_out, t_out, epoch, None)
    g["ddino_loss"] = loss.item()
    g["center1"], g["center_grid1"] = loss_fn.center.clone(), loss_fn.center_grid.clone()
    student.zero_grad()
    loss.backward()
    g["grads"] = {n: GU.probe(p.grad) for n, p in student.named_parameters() if p.grad is not None}
    g["grad_norms"] = {n: p.grad.norm().item() for n, p in student.named_parameters() if p.grad is not None}
    g["no_grad"] = [n for n, p in student.named_parameters() if p.grad is None]
    # second loss call with the updated centres (same activations)
    with torch.no_grad():
        g["ddino_loss_2"] = loss_fn([t.detach() if torch.is_tensor(t) else t for t in s_out], t_out, epoch, None).item()
    # view-level DINOLoss on the cls logits of the two global crops only (config 1 shape)
    vl = ns.DINOLoss(K, 2, 0.04, 0.07, 5, 10)
    vl.center.copy_(g["center0"])
    s2 = student.head(student.forward_features(torch.cat(crops[:2]))[0])
    l2 = vl(s2, t_out[0], epoch, None)
    g["dino_loss_2crops"] = l2.item()
    g["dino_center1"] = vl.center.clone()
    # epoch < freeze_last_layer: clip -> cancel_gradients_last_layer -> AdamW -> EMA on copies (main_esvit.py:569-574, utils.py:118-123)
    s_c, t_c = build_nano(ns), build_nano(ns, teacher=True)  # (weight_norm modules do not deepcopy)
    s_c.load_state_dict(student.state_dict())
    t_c.load_state_dict(teacher.state_dict())
    for p in t_c.parameters():
        p.requires_grad = False
    for (n, pc), p0 in zip(s_c.named_parameters(), student.parameters()):
        pc.requires_grad = p0.requires_grad
        pc.grad = None if p0.grad is None else p0.grad.clone()
    opt_c = torch.optim.AdamW(ns.utils.get_params_groups(s_c))
    for i, pg in enumerate(opt_c.param_groups):
        pg["lr"] = 5e-4
        if i == 0:
            pg["weight_decay"] = 0.04
    ns.utils.clip_gradients(s_c, 3.0)
    ns.utils.cancel_gradients_last_layer(0, s_c, 1)
    g["cancelled"] = [n for n, p in s_c.named_parameters() if p.grad is None and n not in g["no_grad"]]
    opt_c.step()
    with torch.no_grad():
        for pq, pk in zip(s_c.parameters(), t_c.parameters()):
            pk.data.mul_(0.996).add_((1 - 0.996) * pq.detach().data)
    g["student_after_cancel"] = {n: GU.probe(p) for n, p in s_c.named_parameters()}
    g["teacher_after_cancel"] = {n: GU.probe(p) for n, p in t_c.named_parameters()}
    # one update step: clip 3.0 -> AdamW -> EMA (utils.py:106-115, main_esvit.py:574, 587-590)
    opt = torch.optim.AdamW(ns.utils.get_params_groups(student))
    lr, wd, m = 5e-4, 0.04, 0.996
    for i, pg in enumerate(opt.param_groups):
        pg["lr"] = lr
        if i == 0:
            pg["weight_decay"] = wd
    g["clip_norms"] = ns.utils.clip_gradients(student, 3.0)
    opt.step()
    with torch.no_grad():
        for pq, pk in zip(student.parameters(), teacher.parameters()):
            pk.data.mul_(m).add_((1 - m) * pq.detach().data)
    g["student_after"] = {n: GU.probe(p) for n, p in student.named_parameters()}
    g["teacher_after"] = {n: GU.probe(p) for n, p in teacher.named_parameters()}
    g["group_sizes"] = [len(pg["params"]) for pg in opt.param_groups]
    # attention probabilities of the last block for one global crop (forward_selfattention, swin_transformer.py:766-787)
    with torch.no_grad():
        GU.fill_state_dict(student.state_dict(), seed=0)
        student.head.last_layer.weight_g.data.fill_(1)
        g["last_attn"] = GU.probe(student.forward_selfattention(crops[0]))
        # eval_linear.py's feature hook (swin_transformer.py:799-837): n = 3 starts inside stage 2, n = 1 is the normed last block
        g["last_blocks_n3"] = student.forward_return_n_last_blocks(crops[0], n=3, depth=list(GU.NANO["depths"])).clone()
        g["last_blocks_n1_local"] = student.forward_return_n_last_blocks(crops[2], n=1, depth=list(GU.NANO["depths"])).clone()
    torch.save(g, os.path.join(OUT, "nano_step.pt"))
    print("nano_step.pt: loss", g["ddino_loss"], g["ddino_loss_2"], g["dino_loss_2crops"], "no_grad", g["no_grad"])


def gen_nano14(ns):
    """W=14 variant (BASELINE.json configs 3/4 use 14x14 windows): 196-token windows, 27x27 bias tables, stage 3 falls back to
    7x7.  One image, 2 global + 2 local crops, to keep the fixture and the CPU time small."""
    RL.ensure_single_process_group()
    student, teacher = build_nano(ns, window=14), build_nano(ns, teacher=True, window=14)
    GU.fill_state_dict(student.state_dict(), seed=0)
    GU.fill_state_dict(teacher.state_dict(), seed=7)
    student.head.last_layer.weight_g.data.fill_(1)
    for p in teacher.parameters():
        p.requires_grad = False
    crops = GU.make_crops(1, n_local=2)
    K = GU.NANO_HEAD["out_dim"]
    g = {"keys": [(k, tuple(v.shape), str(v.dtype)) for k, v in student.state_dict().items()]}
    loss_fn = ns.DDINOLoss(K, 4, 0.04, 0.07, 5, 10)
    t_out = teacher(crops[:2])
    s_out = student(crops)
    for nm, t in (("s_cls", s_out[0]), ("s_reg", s_out[1]), ("s_fea", s_out[2]), ("t_cls", t_out[0]), ("t_reg", t_out[1])):
        g[nm] = GU.probe(t)
    loss = loss_fn(s_out, t_out, 2, None)
    g["ddino_loss"] = loss.item()
    student.zero_grad()
    loss.backward()
    g["grad_norms"] = {n: p.grad.norm().item() for n, p in student.named_parameters() if p.grad is not None}
    g["grads"] = {n: GU.probe(p.grad) for n, p in student.named_parameters() if p.grad is not None and ("attn" in n or "patch_embed" in n)}
    torch.save(g, os.path.join(OUT, "nano14_step.pt"))
    print("nano14_step.pt: loss", g["ddino_loss"])


def build_nano_cvt(ns, teacher=False):
    cfg = RL.cvt_config(dims=GU.NANO_CVT["dims"], heads=GU.NANO_CVT["heads"], depths=GU.NANO_CVT["depths"])
    m = ns.models.build_model(cfg, is_teacher=teacher, use_dense_prediction=True)
    hk = dict(hidden_dim=GU.NANO_HEAD["hidden_dim"], bottleneck_dim=GU.NANO_HEAD["bottleneck_dim"])
    fea = GU.NANO_CVT["dims"][-1]
    m.head = ns.DINOHead(fea, GU.NANO_HEAD["out_dim"], norm_last_layer=True, **hk)
    m.head_dense = ns.DINOHead(fea, GU.NANO_HEAD["out_dim"], norm_last_layer=False, **hk)
    return m


def gen_nano_cvt(ns):
    """CvT (cvt_v4_transformer, BASELINE config 5) in miniature: ConvEmbed 7/4/2 + 3/2/1, depthwise-conv + BatchNorm qkv, windowed
    attention at head_dim 64 with 7x7 / 6x6 / 3x3 windows and a padded 12 -> 14 grid, QuickGELU FFN; train-mode BatchNorm."""
    RL.ensure_single_process_group()
    student, teacher = build_nano_cvt(ns), build_nano_cvt(ns, teacher=True)
    GU.fill_state_dict(student.state_dict(), seed=0)
    GU.fill_state_dict(teacher.state_dict(), seed=7)
    for m in (student, teacher):
        for k, v in m.state_dict().items():
            if k.endswith("running_var"):
                v.abs_().add_(0.5)
    student.head.last_layer.weight_g.data.fill_(1)
    for p in teacher.parameters():
        p.requires_grad = False
    crops = GU.make_crops(2, n_local=3, sizes=GU.NANO_CVT["sizes"])
    K = GU.NANO_HEAD["out_dim"]
    g = {"keys": [(k, tuple(v.shape), str(v.dtype)) for k, v in student.state_dict().items()],
         "param_names": [n for n, _ in student.named_parameters()]}
    loss_fn = ns.DDINOLoss(K, 5, 0.04, 0.07, 5, 10)
    t_out = teacher(crops[:2])
    s_out = student(crops)
    for nm, t in (("s_cls", s_out[0]), ("s_reg", s_out[1]), ("s_fea", s_out[2]), ("t_cls", t_out[0]), ("t_reg", t_out[1]), ("t_fea", t_out[2])):
        g[nm] = GU.probe(t)
    g["s_fea_full"] = s_out[2].detach().clone()
    g["npatch"] = (list(s_out[3]), list(t_out[3]))
    loss = loss_fn(s_out, t_out, 2, None)
    g["ddino_loss"] = loss.item()
    student.zero_grad()
    loss.backward()
    g["grads"] = {n: GU.probe(p.grad) for n, p in student.named_parameters() if p.grad is not None}
    g["grad_norms"] = {n: p.grad.norm().item() for n, p in student.named_parameters() if p.grad is not None}
    g["no_grad"] = [n for n, p in student.named_parameters() if p.grad is None]
    g["bn_buffers"] = {k: v.detach().clone() for k, v in student.state_dict().items() if "running_" in k or "num_batches" in k}
    # inference consumers (eval_linear.py / eval_knn.py): eval-mode BatchNorm (running statistics) on a fresh parameter fill
    GU.fill_state_dict(student.state_dict(), seed=0)
    for k, v in student.state_dict().items():
        if k.endswith("running_var"):
            v.abs_().add_(0.5)
    student.eval()
    with torch.no_grad():
        cls, region = student.forward_features(crops[0])
        g["eval_cls"], g["eval_region"] = GU.probe(cls), GU.probe(region)
        g["eval_last_blocks"] = student.forward_return_n_last_blocks(crops[2], n=2, depth=list(GU.NANO_CVT["depths"])).clone()
    student.train()
    torch.save(g, os.path.join(OUT, "nano_cvt_step.pt"))
    print("nano_cvt_step.pt: loss", g["ddino_loss"], "npatch", g["npatch"], "params", len(g["param_names"]), "no_grad", g["no_grad"])


def build_cvt_variant(ns, case, teacher=False):
    m = ns.models.build_model(RL.cvt_config(**case["cfg"]), is_teacher=teacher, use_dense_prediction=True)
    hk = dict(hidden_dim=GU.NANO_HEAD["hidden_dim"], bottleneck_dim=GU.NANO_HEAD["bottleneck_dim"])
    fea = case["cfg"]["dims"][-1]
    m.head = ns.DINOHead(fea, GU.NANO_HEAD["out_dim"], norm_last_layer=True, **hk)
    m.head_dense = ns.DINOHead(fea, GU.NANO_HEAD["out_dim"], norm_last_layer=False, **hk)
    return m


def gen_cvt_variants(ns):
    """the CvT variants of the other cvt_v4 yaml files (REL_POS_EMBED, SHIFT, RES_STEM) in miniature, one training step each through
    the reference's own modules"""
    RL.ensure_single_process_group()
    out = {}
    for name, case in GU.NANO_CVT_VARIANTS.items():
        student, teacher = build_cvt_variant(ns, case), build_cvt_variant(ns, case, teacher=True)
        GU.fill_state_dict(student.state_dict(), seed=0)
        GU.fill_state_dict(teacher.state_dict(), seed=7)
        for m in (student, teacher):
            for k, v in m.state_dict().items():
                if k.endswith("running_var"):
                    v.abs_().add_(0.5)
        student.head.last_layer.weight_g.data.fill_(1)
        for p in teacher.parameters():
            p.requires_grad = False
        crops = GU.make_crops(2, n_local=case["n_local"], sizes=case["sizes"])
        g = {"keys": [(k, tuple(v.shape), str(v.dtype)) for k, v in student.state_dict().items()],
             "param_names": [n for n, _ in student.named_parameters()]}
        loss_fn = ns.DDINOLoss(GU.NANO_HEAD["out_dim"], 2 + case["n_local"], 0.04, 0.07, 5, 10)
        t_out = teacher(crops[:2])
        s_out = student(crops)
        for nm, t in (("s_cls", s_out[0]), ("s_reg", s_out[1]), ("s_fea", s_out[2]), ("t_cls", t_out[0]), ("t_reg", t_out[1]), ("t_fea", t_out[2])):
            g[nm] = GU.probe(t)
        g["npatch"] = (list(s_out[3]), list(t_out[3]))
        loss = loss_fn(s_out, t_out, 2, None)
        g["ddino_loss"] = loss.item()
        student.zero_grad()
        loss.backward()
        g["grads"] = {n: GU.probe(p.grad) for n, p in student.named_parameters() if p.grad is not None}
        g["grad_norms"] = {n: p.grad.norm().item() for n, p in student.named_parameters() if p.grad is not None}
        g["bn_buffers"] = {k: v.detach().clone() for k, v in student.state_dict().items() if "running_" in k or "num_batches" in k}
        out[name] = g
        print("nano_cvt_variants.pt:", name, "loss", g["ddino_loss"], "npatch", g["npatch"], "params", len(g["param_names"]))
    torch.save(out, os.path.join(OUT, "nano_cvt_variants.pt"))


def build_nano_vit(ns, teacher=False):
    import importlib
    from functools import partial
    vits = importlib.import_module("models.vision_transformer")  # the reference's module (main_esvit.py:36, `vits`)
    v = GU.NANO_VIT
    m = vits.VisionTransformer(img_size=[v["sizes"][0]], patch_size=v["patch"], embed_dim=v["embed_dim"], depth=v["depth"], num_heads=v["heads"],
                               mlp_ratio=4, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), drop_path_rate=0.0,
                               use_dense_prediction=True)
    hk = dict(hidden_dim=GU.NANO_HEAD["hidden_dim"], bottleneck_dim=GU.NANO_HEAD["bottleneck_dim"])
    m.head = ns.DINOHead(v["embed_dim"], GU.NANO_HEAD["out_dim"], norm_last_layer=True, **hk)
    m.head_dense = ns.DINOHead(v["embed_dim"], GU.NANO_HEAD["out_dim"], norm_last_layer=False, **hk)
    return m


def gen_nano_vit(ns):
    """the monolithic ViT (models/vision_transformer.py: deit_tiny / deit_small / vit_base, the DEFAULT --arch) in miniature:
    class token, interpolated position embedding for the local crops, global attention, dense prediction heads"""
    RL.ensure_single_process_group()
    student, teacher = build_nano_vit(ns), build_nano_vit(ns, teacher=True)
    GU.fill_state_dict(student.state_dict(), seed=0)
    GU.fill_state_dict(teacher.state_dict(), seed=7)
    student.head.last_layer.weight_g.data.fill_(1)
    for p in teacher.parameters():
        p.requires_grad = False
    crops = GU.make_crops(2, n_local=3, sizes=GU.NANO_VIT["sizes"])
    K = GU.NANO_HEAD["out_dim"]
    g = {"keys": [(k, tuple(v.shape), str(v.dtype)) for k, v in student.state_dict().items()],
         "param_names": [n for n, _ in student.named_parameters()]}
    loss_fn = ns.DDINOLoss(K, 5, 0.04, 0.07, 5, 10)
    t_out = teacher(crops[:2])
    s_out = student(crops)
    for nm, t in (("s_cls", s_out[0]), ("s_reg", s_out[1]), ("s_fea", s_out[2]), ("t_cls", t_out[0]), ("t_reg", t_out[1]), ("t_fea", t_out[2])):
        g[nm] = GU.probe(t)
    g["s_fea_full"] = s_out[2].detach().clone()
    g["npatch"] = (list(s_out[3]), list(t_out[3]))
    loss = loss_fn(s_out, t_out, 2, None)
    g["ddino_loss"] = loss.item()
    student.zero_grad()
    loss.backward()
    g["grads"] = {n: GU.probe(p.grad) for n, p in student.named_parameters() if p.grad is not None}
    g["grad_norms"] = {n: p.grad.norm().item() for n, p in student.named_parameters() if p.grad is not None}
    g["no_grad"] = [n for n, p in student.named_parameters() if p.grad is None]
    # view-level branch (use_dense_prediction False: forward returns head(cls) only) and the evaluation hooks
    with torch.no_grad():
        student.use_dense_prediction = False
        g["view_only"] = GU.probe(student(crops))
        student.use_dense_prediction = True
        g["last_blocks"] = student.forward_return_n_last_blocks(crops[2], n=2, return_patch_avgpool=True).clone()
        g["last_attn"] = student.forward_selfattention(crops[0]).clone()
    torch.save(g, os.path.join(OUT, "nano_vit_step.pt"))
    print("nano_vit_step.pt: loss", g["ddino_loss"], "npatch", g["npatch"], "params", len(g["param_names"]), "no_grad", g["no_grad"])


def gen_full_vit(ns):
    """deit_small at FULL width from the reference's own modules (197 / 37 tokens, 6 heads of 64, 12 blocks), batch 2, out_dim 4096:
    the step the GPU parity test of the monolithic ViT compares with at real geometry -- loss, every gradient norm, strided samples
    of a dozen gradient tensors, probes of the outputs"""
    import importlib
    RL.ensure_single_process_group()
    vits = importlib.import_module("models.vision_transformer")
    K = 4096

    def make(seed):
        m = vits.deit_small(patch_size=16, drop_path_rate=0.0, use_dense_prediction=True)
        m.head, m.head_dense = ns.DINOHead(384, K, norm_last_layer=True), ns.DINOHead(384, K, norm_last_layer=False)
        GU.fill_state_dict(m.state_dict(), seed)
        return m
    student, teacher = make(41), make(42)
    student.head.last_layer.weight_g.data.fill_(1)
    for p in teacher.parameters():
        p.requires_grad = False
    crops = GU.make_crops(2, seed=77)
    loss_fn = ns.DDINOLoss(K, 10, 0.04, 0.07, 5, 10)
    t_out = teacher(crops[:2])
    s_out = student(crops)
    loss = loss_fn(s_out, t_out, 2, None)
    student.zero_grad()
    loss.backward()
    names = [n for n, p in student.named_parameters() if p.grad is not None]
    prm = dict(student.named_parameters())
    g = {"K": K, "loss": loss.item(), "npatch": (list(s_out[3]), list(t_out[3])),
         "s_cls": GU.probe(s_out[0]), "s_reg": GU.probe(s_out[1]), "s_fea": GU.probe(s_out[2]), "t_cls": GU.probe(t_out[0]),
         "grad_norms": {n: prm[n].grad.norm().item() for n in names},
         "grad_samples": {n: GU.strided(prm[n].grad, 4096) for n in GU.full_sampled_names(names)},
         "center_after": loss_fn.center.clone(), "center_grid_after": loss_fn.center_grid.clone()}
    torch.save(g, os.path.join(OUT, "full_vit.pt"))
    print("full_vit.pt: loss", g["loss"], "npatch", g["npatch"], "grads", len(names))


def gen_knn(ns):
    """top-1 / top-5 of the reference's knn_classifier on synthetic feature sets (tests/golden_utils.make_knn_set)"""
    ref_knn = RL.load_knn_classifier()
    saved = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self  # the reference calls .cuda() on its scratch tensor (eval_knn.py:199)
    try:
        out = []
        for c in GU.KNN_CASES:
            xtr, ytr, xte, yte = GU.make_knn_set(c["seed"], noise=c["noise"])
            out.append(tuple(ref_knn(xtr, ytr, xte, yte, c["k"], c["T"], num_classes=10)))
    finally:
        torch.Tensor.cuda = saved
    torch.save({"cases": GU.KNN_CASES, "top": out}, os.path.join(OUT, "knn.pt"))
    print("knn golden:", out)


def gen_linear_probe(ns):
    """eval_linear.py's linear probe from the reference's own train / validate_network / LinearClassifier on its own nano Swin
    backbone (forward_return_n_last_blocks): per-epoch train stats, validation stats and the classifier after two epochs"""
    import torch.nn as nn
    train, validate, LinearClassifier = RL.load_eval_linear()
    c = GU.LINEAR_PROBE
    model = build_nano(ns)
    GU.fill_state_dict(model.state_dict(), 0)
    model.eval()
    depths = list(GU.NANO["depths"])
    dims = [GU.NANO["embed_dim"] * 2 ** i for i in range(4) for _ in range(depths[i])]
    clf = LinearClassifier(sum(dims[-c["n_last_blocks"]:]), c["num_labels"])
    GU.linear_probe_init(clf)
    opt = torch.optim.SGD(clf.parameters(), c["lr"], momentum=0.9, weight_decay=0)
    tr, va = GU.linear_probe_data()
    saved = (torch.Tensor.cuda, nn.Module.cuda, torch.cuda.synchronize, ns.utils.is_dist_avail_and_initialized)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None
    ns.utils.is_dist_avail_and_initialized = lambda: False
    try:
        stats = [train(model, clf, opt, tr, ep, c["n_last_blocks"], c["avgpool"], depths) for ep in range(2)]
        val = validate(va, model, clf, c["n_last_blocks"], c["avgpool"], depths)
    finally:
        torch.Tensor.cuda, nn.Module.cuda, torch.cuda.synchronize, ns.utils.is_dist_avail_and_initialized = saved
    out = {"train": stats, "val": val, "weight": clf.linear.weight.detach().clone(), "bias": clf.linear.bias.detach().clone(),
           "keys": list(clf.state_dict().keys()), "dim": clf.linear.weight.shape[1]}
    torch.save(out, os.path.join(OUT, "linear_probe.pt"))
    print("linear_probe.pt:", stats, val)


def gen_variants(ns):
    """SURVEY.md 8f-3 step variants, each from the reference's own code:
    bn_head -- DINOHead(use_bn=True) (vision_transformer.py:384-418) in train mode (batch statistics, running-stat update)
               and in eval mode (running statistics): logits, every gradient of sum(logits * probe), the buffers after."""
    RL.ensure_single_process_group()
    g = {}
    c = GU.BN_HEAD
    head = ns.DINOHead(c["in_dim"], c["out_dim"], use_bn=True, hidden_dim=c["hidden_dim"], bottleneck_dim=c["bottleneck_dim"])
    GU.fill_bn_head(head.state_dict(), 31)
    x, probe = GU.bn_head_inputs()
    x = x.clone().requires_grad_(True)
    head.train()
    out = head(x)
    (out * probe).sum().backward()
    g["bn_head"] = {
        "logits": out.detach().clone(),
        "dx": x.grad.clone(),
        "grads": {n: p.grad.clone() for n, p in head.named_parameters() if p.grad is not None},
        "buffers_after": {n: b.clone() for n, b in head.named_buffers()},
    }
    head.eval()
    with torch.no_grad():
        g["bn_head"]["logits_eval"] = head(x.detach()).clone()
    # lars -- the reference's utils.LARS (utils.py:519-557) over the two groups of utils.get_params_groups, three steps with
    #         utils.clip_gradients' per-tensor rule (utils.py:106-115) applied first, as train_one_epoch does
    p0, grads = GU.lars_case()

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            for n, t in p0.items():
                self.register_parameter(n, torch.nn.Parameter(t.clone()))
    net = Holder()
    opt = ns.utils.LARS(ns.utils.get_params_groups(net))
    for (lr, wd), gs in zip(GU.LARS_SCHED, grads):
        for i, pg in enumerate(opt.param_groups):
            pg["lr"] = lr
            if i == 0:
                pg["weight_decay"] = wd
        for n, prm in net.named_parameters():
            prm.grad = gs[n].clone()
        ns.utils.clip_gradients(net, 3.0)
        opt.step()
    g["lars"] = {"params": {n: prm.detach().clone() for n, prm in net.named_parameters()},
                 "mu": {n: opt.state[prm]["mu"].clone() for n, prm in net.named_parameters()}}
    # ape -- USE_APE (swin_transformer.py:623-627, 680-681): three-stage nano Swin at 112^2 (28 x 28 token grid), features of a batch
    #        and the gradients of sum(cls * probe) wrt the embedding and two other parameters
    cfg = RL.swin_config(embed_dim=GU.NANO["embed_dim"], depths=(2, 2, 2), heads=(1, 2, 4), window=GU.NANO["window"], img=112)
    cfg.MODEL["SPEC"]["USE_APE"] = True
    m = ns.models.build_model(cfg, is_teacher=False, use_dense_prediction=False)
    GU.fill_state_dict(m.state_dict(), 17)
    xa, pr = GU.ape_inputs(m.num_features)
    cls = m.forward_features(xa)
    (cls * pr).sum().backward()
    names = ["absolute_pos_embed", "patch_embed.proj.weight", "layers.0.blocks.0.attn.qkv.weight", "norm.weight"]
    prm = dict(m.named_parameters())
    g["ape"] = {"keys": [(k, tuple(v.shape)) for k, v in m.state_dict().items()], "cls": cls.detach().clone(),
                "grads": {n: prm[n].grad.clone() for n in names}}
    # oddmerge -- PatchMerging on an odd feature map (swin_transformer.py:406-408): four-stage nano Swin at 112^2 (28, 14, 7 -> 4)
    cfg = RL.swin_config(embed_dim=GU.NANO["embed_dim"], depths=GU.NANO["depths"], heads=GU.NANO["heads"], window=GU.NANO["window"], img=112)
    m = ns.models.build_model(cfg, is_teacher=False, use_dense_prediction=True)
    GU.fill_state_dict(m.state_dict(), 23)
    xa, pr = GU.ape_inputs(m.num_features)
    cls, region = m.forward_features(xa)
    ((cls * pr).sum() + region.sum() * 0.01).backward()
    names = ["patch_embed.proj.weight", "layers.2.downsample.norm.weight", "layers.2.downsample.reduction.weight", "layers.3.blocks.1.mlp.fc2.bias"]
    prm = dict(m.named_parameters())
    g["oddmerge"] = {"cls": cls.detach().clone(), "region": region.detach().clone(), "grads": {n: prm[n].grad.clone() for n in names}}
    # mixup -- the reference's DINOLoss.forward with targets_mixup (main_esvit.py:639-641)
    mc = GU.MIXUP
    s_l, t_l, c0, T = GU.mixup_case()
    lf = ns.DINOLoss(mc["K"], mc["ncrops"], 0.04, 0.07, 5, 10)
    lf.center.copy_(c0)
    s_l = s_l.clone().requires_grad_(True)
    lm = lf(s_l, t_l, 2, T)
    lm.backward()
    g["mixup"] = {"loss": lm.item(), "ds": s_l.grad.clone(), "center_after": lf.center.clone()}
    torch.save(g, os.path.join(OUT, "variants.pt"))
    print("variants.pt: mixup loss", g["mixup"]["loss"])
    print("variants.pt: bn_head logits", tuple(g["bn_head"]["logits"].shape), "grads", len(g["bn_head"]["grads"]))


def gen_full(ns):
    """Full-width steps from the reference's own modules (Swin-T W=7 widths, real out_dim): what the GPU parity tests at full
    size compare with.  Per case: loss, centres after the loss call, every gradient norm, strided samples of twelve gradient
    tensors and of the student outputs.  tests/golden_utils.FULL_CASES names weights / crops by seed so that the tests rebuild
    exactly the same inputs."""
    RL.ensure_single_process_group()
    out = {}
    for name, c in GU.FULL_CASES.items():
        torch.manual_seed(0)
        cfg = RL.swin_config(embed_dim=GU.SWIN_T["embed_dim"], depths=GU.SWIN_T["depths"], heads=GU.SWIN_T["heads"], window=GU.SWIN_T["window"])
        K, B = c["K"], c["B"]
        student = ns.models.build_model(cfg, is_teacher=False, use_dense_prediction=c["dense"])
        teacher = ns.models.build_model(cfg, is_teacher=True, use_dense_prediction=c["dense"])
        student.head, teacher.head = ns.DINOHead(student.num_features, K), ns.DINOHead(teacher.num_features, K)
        if c["dense"]:
            nl = name != "swin_t_k8192_b2"  # (that test builds the dense student head with norm_last_layer=False)
            student.head_dense = ns.DINOHead(student.num_features, K, norm_last_layer=nl)
            teacher.head_dense = ns.DINOHead(teacher.num_features, K)
        GU.fill_state_dict(student.state_dict(), c["s_seed"])
        GU.fill_state_dict(teacher.state_dict(), c["t_seed"])
        student.head.last_layer.weight_g.data.fill_(1)
        if c["dense"] and name == "swin_t_k65536_b8":
            student.head_dense.last_layer.weight_g.data.fill_(1)
        for p in teacher.parameters():
            p.requires_grad = False
        crops = GU.make_crops(B, seed=c["crop_seed"])[:c["ncrops"]]
        if c["dense"]:
            loss_fn = ns.DDINOLoss(K, 10, 0.04, 0.04, 0, 1)
        else:
            loss_fn = ns.DINOLoss(K, 2, 0.04, 0.04, 0, 1)
        with torch.no_grad():
            t_out = teacher(crops[:2])
        s_out = student(crops)
        loss = loss_fn(s_out, t_out, 0, None)
        loss.backward()
        names = [n for n, p in student.named_parameters() if p.requires_grad]
        prm = dict(student.named_parameters())
        g = {"loss": loss.item(), "grad_norm": {n: prm[n].grad.norm().item() for n in names if prm[n].grad is not None},
             "sampled": {n: GU.strided(prm[n].grad) for n in GU.full_sampled_names(names)},
             "center": loss_fn.center.clone()}
        if c["dense"]:
            g["center_grid"] = loss_fn.center_grid.clone()
            g["s_out"] = [GU.strided(s_out[0]), GU.strided(s_out[1]), GU.strided(s_out[2])]
            g["s_out_absmax"] = [s_out[i].detach().abs().max().item() for i in range(3)]
            g["npatch"] = list(s_out[3])
        else:
            g["s_out"] = [GU.strided(s_out)]
            g["s_out_absmax"] = [s_out.detach().abs().max().item()]
            g["t_out"] = GU.strided(t_out)
        out[name] = g
        print("full_width:", name, "loss", g["loss"], "params with grad", len(g["grad_norm"]))
        del student, teacher, s_out, t_out, loss
    torch.save(out, os.path.join(OUT, "full_width.pt"))


def _yaml_config(path):
    """the reference's experiment yaml as the config object its build_model reads (stochastic depth off: both sides deterministic)"""
    import yaml
    with open(os.path.join(RL.REF_ROOT, path)) as fh:
        y = yaml.safe_load(fh)
    model = dict(NUM_CLASSES=0, INIT_WEIGHTS=False, PRETRAINED="", PRETRAINED_LAYERS=["*"])
    model.update(y["MODEL"])
    model["SPEC"] = dict(model["SPEC"], DROP_PATH_RATE=0.0)
    return RL.AttrDict(MODEL=model, TRAIN=dict(IMAGE_SIZE=[224, 224]), FINETUNE=dict(FINETUNE=False, FROZEN_LAYERS=[]), VERBOSE=False)


def _full_cfg_case(ns, name, c):
    torch.manual_seed(0)
    cfg = _yaml_config(c["yaml"])
    if c["arch"].startswith("vil_"):
        # vil_tiny/base.yaml names the model 'cls_vil' (the registry key is the module name) and leaves out the optional MSVIT keys
        # get_cls_model reads (vision_longformer.py:755-786); DROP_PATH is this family's stochastic-depth key
        cfg["MODEL"]["NAME"] = "vision_longformer"   # (item access: attribute access hands out copies)
        spec = cfg["MODEL"]["SPEC"]
        spec["DROP_PATH"] = 0.0
        spec["MSVIT"] = dict(dict(POOL_METHOD=None, WITH_SE=None, SE_MLP_RATIO=0.625, SE_MLP_BALANCE=False), **spec["MSVIT"])
        for k in ("POOL_METHOD", "WITH_SE"):
            if spec["MSVIT"][k] == "None":
                spec["MSVIT"][k] = None
    K, B = c["K"], c["B"]
    student = ns.models.build_model(cfg, is_teacher=False, use_dense_prediction=True)
    teacher = ns.models.build_model(cfg, is_teacher=True, use_dense_prediction=True)
    fea = student.num_features if hasattr(student, "num_features") else (student.out_planes if hasattr(student, "out_planes") else cfg.MODEL.SPEC.DIM_EMBED[-1])
    student.head, teacher.head = ns.DINOHead(fea, K), ns.DINOHead(fea, K)
    student.head_dense, teacher.head_dense = ns.DINOHead(fea, K), ns.DINOHead(fea, K)
    GU.fill_full_cfg_pair(student, teacher, c)
    crops = GU.make_crops(B, seed=c["crop_seed"])
    loss_fn = ns.DDINOLoss(K, 10, 0.04, 0.04, 0, 1)
    with torch.no_grad():
        t_out = teacher(crops[:2])
    s_out = student(crops)
    loss = loss_fn(s_out, t_out, 0, None)
    loss.backward()
    names = [n for n, p in student.named_parameters() if p.requires_grad]
    prm = dict(student.named_parameters())
    n = GU.FULL_CFG_SAMPLE
    g = {"loss": loss.item(), "grad_norm": {k: prm[k].grad.norm().item() for k in names if prm[k].grad is not None},
         "sampled": {k: GU.strided(prm[k].grad, n) for k in GU.full_sampled_names(names)},
         "center": loss_fn.center.clone(), "center_grid": loss_fn.center_grid.clone(),
         "s_out": [GU.strided(s_out[i], n) for i in range(3)], "s_out_absmax": [s_out[i].detach().abs().max().item() for i in range(3)],
         "npatch": list(s_out[3]), "param_names": [k for k, _ in student.named_parameters()],
         "keys": [(k, tuple(v.shape)) for k, v in student.state_dict().items()]}
    if c["arch"].startswith("vil_"):  # the evaluation hook of eval_linear.py / eval_knn.py on the same weights
        student.eval()
        with torch.no_grad():
            depth = [cf['n'] for cf in student.layer_cfgs]
            g["last_blocks"] = student.forward_return_n_last_blocks(crops[2][:1], n=4, depth=depth).clone()
        student.train()
    print("full fixture:", name, "loss", g["loss"], "params with grad", len(g["grad_norm"]), "npatch", g["npatch"])
    return g


def gen_full_configs(ns):
    """BASELINE.json configs 3-5 at FULL width from the reference's own modules and experiment yamls: Swin-T W=14, Swin-B W=14
    (widths 128..1024: a GEMM tile family no W=7 fixture touches) and CvT-13 (cvt_v4 s1.yaml), 2x224^2 + 8x96^2 crops, V+R heads,
    DDINOLoss, B = 2, out_dim 8192.  Same contents per case as gen_full."""
    RL.ensure_single_process_group()
    torch.save({name: _full_cfg_case(ns, name, c) for name, c in GU.FULL_CFG_CASES.items()}, os.path.join(OUT, "full_configs.pt"))


def gen_full_vil(ns):
    """Vision Longformer: the reference's own MsViT from experiments/imagenet/vil/vil_tiny/base.yaml (sliding-chunk
    attention of layers/longformer2d.py + layers/slidingchunk_2d.py in stages 1-2, head_dim 48 in stage 1), same step and contents as
    gen_full_configs, plus forward_return_n_last_blocks"""
    RL.ensure_single_process_group()
    torch.save({name: _full_cfg_case(ns, name, c) for name, c in GU.FULL_VIL_CASES.items()}, os.path.join(OUT, "full_vil.pt"))


def gen_head_nlayers(ns):
    """the reference's DINOHead with nlayers = 1, 2, 4 (vision_transformer.py:388-402): logits, input gradient, parameter gradients"""
    c = GU.HEAD_NLAYERS
    out = {}
    x, probe = GU.head_nlayers_inputs()
    for n in c["cases"]:
        head = ns.DINOHead(c["in_dim"], c["out_dim"], nlayers=n, hidden_dim=c["hidden_dim"], bottleneck_dim=c["bottleneck_dim"], norm_last_layer=False)
        GU.fill_state_dict(head.state_dict(), 60 + n)
        xr = x.clone().requires_grad_(True)
        y = head(xr)
        (y * probe).sum().backward()
        out[n] = {"keys": [(k, tuple(v.shape)) for k, v in head.state_dict().items()], "logits": y.detach().clone(), "dx": xr.grad.clone(),
                  "grads": {k: p.grad.clone() for k, p in head.named_parameters()}}
    torch.save(out, os.path.join(OUT, "head_nlayers.pt"))
    print("head_nlayers.pt:", {n: len(v["grads"]) for n, v in out.items()})


def gen_head_bn_nlayers(ns):
    """the reference's DINOHead(use_bn=True) with nlayers = 1, 2, 4 (vision_transformer.py:388-402) in train mode: logits, input and
    parameter gradients of sum(logits * probe), the BatchNorm buffers after the pass"""
    c = GU.HEAD_NLAYERS
    out = {}
    x, probe = GU.head_nlayers_inputs()
    for n in GU.HEAD_BN_NLAYERS:
        head = ns.DINOHead(c["in_dim"], c["out_dim"], use_bn=True, nlayers=n, hidden_dim=c["hidden_dim"], bottleneck_dim=c["bottleneck_dim"],
                           norm_last_layer=False)
        GU.fill_bn_head_n(head.state_dict(), 80 + n)
        head.train()
        xr = x.clone().requires_grad_(True)
        y = head(xr)
        (y * probe).sum().backward()
        out[n] = {"keys": [(k, tuple(v.shape)) for k, v in head.state_dict().items()], "logits": y.detach().clone(), "dx": xr.grad.clone(),
                  "grads": {k: p.grad.clone() for k, p in head.named_parameters()},
                  "buffers": {k: v.detach().clone() for k, v in head.state_dict().items() if "running_" in k or "num_batches" in k}}
    torch.save(out, os.path.join(OUT, "head_bn_nlayers.pt"))
    print("head_bn_nlayers.pt:", {n: len(v["grads"]) for n, v in out.items()})


def gen_ref_checkpoint(ns):
    """a training checkpoint as main_esvit.py writes it (utils.save_on_master: DistributedDataParallel student, plain teacher, loss
    state) and the eval_knn.py path over it -- build_model(is_teacher=True) with NUM_CLASSES 0, utils.load_pretrained_weights,
    the backbone's features of a synthetic set, knn_classifier -- all with the reference's own functions"""
    import torch.nn as nn
    RL.ensure_single_process_group()
    c = GU.REF_CKPT
    cfg = RL.swin_config(embed_dim=c["embed_dim"], depths=c["depths"], heads=c["heads"], window=c["window"])

    def with_heads(teacher):
        m = ns.models.build_model(cfg, is_teacher=teacher, use_dense_prediction=True)
        fea = m.num_features
        m.head = ns.DINOHead(fea, c["head"]["out_dim"], norm_last_layer=True, hidden_dim=c["head"]["hidden_dim"], bottleneck_dim=c["head"]["bottleneck_dim"])
        m.head_dense = ns.DINOHead(fea, c["head"]["out_dim"], norm_last_layer=False, hidden_dim=c["head"]["hidden_dim"],
                                   bottleneck_dim=c["head"]["bottleneck_dim"])
        return m
    student, teacher = with_heads(False), with_heads(True)
    GU.fill_state_dict(student.state_dict(), 41)
    GU.fill_state_dict(teacher.state_dict(), 42)
    ddp = nn.parallel.DistributedDataParallel(student)  # (CPU, one gloo rank: the state_dict keys carry DDP's "module." prefix)
    loss = ns.DDINOLoss(c["head"]["out_dim"], 10, 0.04, 0.07, 5, 10)
    path = os.path.join(OUT, "ref_checkpoint.pth")
    ns.utils.save_on_master({"student": ddp.state_dict(), "teacher": teacher.state_dict(), "epoch": 3, "dino_loss": loss.state_dict()}, path)
    xtr, ytr, xte, yte = GU.ref_ckpt_data()
    ref_knn = RL.load_knn_classifier()
    out = {"student_keys": list(ddp.state_dict().keys())[:4]}
    saved = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for key in ("teacher", "student"):
            model = ns.models.build_model(cfg, is_teacher=True)
            ns.utils.load_pretrained_weights(model, path, key, "swin_nano", 4)
            model.eval()
            with torch.no_grad():
                ftr, fte = model(xtr), model(xte)
            ntr, nte = nn.functional.normalize(ftr, dim=1, p=2), nn.functional.normalize(fte, dim=1, p=2)
            out[key] = {"train": ftr.clone(), "test": fte.clone(), "top": tuple(ref_knn(ntr, ytr, nte, yte, c["k"], c["T"], num_classes=c["classes"]))}
    finally:
        torch.Tensor.cuda = saved
    torch.save(out, os.path.join(OUT, "ref_checkpoint.pt"))
    print("ref_checkpoint.pth:", os.path.getsize(path), "bytes; knn", {k: out[k]["top"] for k in ("teacher", "student")})


def gen_mixup_smoothing(ns):
    """the reference's DINOLoss.forward with label-smoothed mixup targets (main_esvit.py:230, 639-641): dense [B, B] matrices"""
    RL.ensure_single_process_group()
    mc = GU.MIXUP
    s_l, t_l, c0, T = GU.mixup_case(GU.MIXUP_SMOOTHING)
    lf = ns.DINOLoss(mc["K"], mc["ncrops"], 0.04, 0.07, 5, 10)
    lf.center.copy_(c0)
    s_l = s_l.clone().requires_grad_(True)
    lm = lf(s_l, t_l, 2, T)
    lm.backward()
    torch.save({"loss": lm.item(), "ds": s_l.grad.clone(), "center_after": lf.center.clone()}, os.path.join(OUT, "mixup_smoothing.pt"))
    print("mixup_smoothing.pt: loss", lm.item())


def gen_patch_norm(ns):
    """PATCH_NORM False (swin_transformer.py:532-535, 545-546: PatchEmbed without its LayerNorm): three-stage nano Swin, features of a
    112^2 batch and the multi-crop forward over a (112^2, 64^2) pair, with the gradients of a probe-weighted sum"""
    names = ["patch_embed.proj.weight", "patch_embed.proj.bias", "layers.0.blocks.0.norm1.weight", "layers.1.downsample.reduction.weight", "norm.bias"]
    out = {}
    for mode in ("features", "multi_crop"):
        cfg = RL.swin_config(embed_dim=GU.NANO["embed_dim"], depths=(2, 2, 2), heads=(1, 2, 4), window=GU.NANO["window"], img=112)
        cfg.MODEL["SPEC"]["PATCH_NORM"] = False
        m = ns.models.build_model(cfg, is_teacher=False, use_dense_prediction=False)
        GU.fill_state_dict(m.state_dict(), 29)
        xa, xb, pa, pab = GU.patch_norm_inputs(m.num_features)
        if mode == "features":
            y = m.forward_features(xa)
            (y * pa).sum().backward()
        else:
            y = m([xa, xb])
            (y * pab).sum().backward()
        prm = dict(m.named_parameters())
        out[mode] = {"keys": [(k, tuple(v.shape)) for k, v in m.state_dict().items()], "y": y.detach().clone(),
                     "grads": {n: prm[n].grad.clone() for n in names}}
    torch.save(out, os.path.join(OUT, "patch_norm.pt"))
    print("patch_norm.pt:", {k: tuple(v["y"].shape) for k, v in out.items()})


def main():
    os.makedirs(OUT, exist_ok=True)
    ns = RL.load()
    only = sys.argv[1:]  # e.g. `python oracle/gen_golden.py cvt` regenerates one fixture
    if not only or "maps" in only:
        gen_index_maps(ns)
    if not only or "nano" in only:
        gen_nano(ns)
    if not only or "nano14" in only:
        gen_nano14(ns)
    if not only or "cvt" in only:
        gen_nano_cvt(ns)
    if not only or "cvt_variants" in only:
        gen_cvt_variants(ns)
    if not only or "vit" in only:
        gen_nano_vit(ns)
    if not only or "knn" in only:
        gen_knn(ns)
    if not only or "variants" in only:
        gen_variants(ns)
    if not only or "linear" in only:
        gen_linear_probe(ns)
    if not only or "head_nlayers" in only:
        gen_head_nlayers(ns)
    if not only or "head_bn_nlayers" in only:
        gen_head_bn_nlayers(ns)
    if not only or "patch_norm" in only:
        gen_patch_norm(ns)
    if not only or "mixup_smoothing" in only:
        gen_mixup_smoothing(ns)
    if not only or "ref_checkpoint" in only:
        gen_ref_checkpoint(ns)
    if "full" in only:  # minutes of CPU time: regenerated on request only
        gen_full(ns)
    if "full_vit" in only:
        gen_full_vit(ns)
    if "full_configs" in only:
        gen_full_configs(ns)
    if "full_vil" in only:
        gen_full_vil(ns)


if __name__ == "__main__":
    main()
