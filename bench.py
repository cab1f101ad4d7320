#!/usr/bin/env python
"""EsViT pre-training step benchmark (BASELINE.json metric: images/sec, Swin-T W=7, 2x224^2 + 8x96^2 crops, view+region
loss, bf16, per-parameter clip + AdamW + teacher EMA, DP over RCCL when --gpus > 1).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`python bench.py --gpus N` without a torchrun environment re-executes itself under torch.distributed.run with N ranks.

Prints ONE JSON line on rank 0.  A "step" = teacher fwd (2 global crops) + student fwd (10 crops) + DDINOLoss + backward +
gradient all-reduce (N>1) + fused clip/AdamW/EMA on a fixed synthetic batch resident in HBM.  `value` = images/s over all
ranks; the timed region carries no instrumentation.  `roofline` describes the dominant kernel family (the MFMA GEMM: 99% of
the step's FLOPs; SURVEY.md 8d bounds it by the bf16 MFMA peak) from HIP events recorded around every GEMM launch of a few
extra steps AFTER the timed region (those steps run on ONE stream -- the teacher pass and the weight-gradient GEMMs, which the timed steps
overlap on side streams, are back in line -- so that a launch's duration is its own); `cpu_baseline` times the reference's PyTorch path on the host cores on a bounded sample
of the same workload -- the reference's own modules when /root/reference is present (kind "reference"), otherwise the
oracle's restatement of them (kind "port").
"""
import argparse
import json
import math
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL across processes needs it)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")       # kernel arguments in device memory: shorter launch gaps (~1500 launches / step, +0.6 %)

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMG = 154.5          # Swin-T W7 V+R, teacher fwd + student fwd + 2x student bwd + loss
GFLOP_PER_IMG_BY_ARCH = {"swin_tiny_w7": 154.5, "swin_tiny_w14": 197.0, "swin_base_w14": 628.5, "cvt_s1": 136.7}


def _head_flops(rows, C, K=65536):
    """DINOHead (vision_transformer.py:391-418): C -> 2048 -> 2048 -> 256 -> K per row"""
    return 2.0 * rows * (C * 2048 + 2048 * 2048 + 2048 * 256 + 256 * K)


def _blocks_flops(N, C, depth, keys=None):
    """depth transformer blocks on N tokens of width C: qkv + proj + 4C MLP, and attention over `keys` keys per query (default: all)"""
    keys = N if keys is None else min(keys, N)
    return depth * (2.0 * N * (3 * C * C + C * C + 8 * C * C) + 4.0 * N * keys * C)


def _vit_crop_flops(S, C, depth, patch=16):
    N = (S // patch) ** 2 + 1
    return 2.0 * (N - 1) * 3 * patch * patch * C + _blocks_flops(N, C, depth) + _head_flops(1, C) + _head_flops(N - 1, C)


def _vil_crop_flops(S, arch):
    """Vision Longformer: ALGORITHMIC flops -- a local query of a sliding-chunk stage sees its global tokens and nine w x w chunks
    (layers/longformer2d.py:140-152), whatever the implementation multiplies"""
    fl, side, cin = 0.0, S, 3
    for layer in arch.split('_'):
        c = {'h': 3, 'd': 192, 'n': 1, 's': 1, 'g': 1, 'p': 2, 'f': 7}
        c.update({a[0]: int(a[1:]) for a in layer.split(',')})
        side //= c['p']
        N = side * side + c['g']
        fl += 2.0 * side * side * cin * c['p'] ** 2 * c['d'] + _blocks_flops(N, c['d'], c['n'], keys=(c['g'] + 9 * c['f'] ** 2) if c['s'] else None)
        cin = c['d']
    return fl + _head_flops(1, cin) + _head_flops(side * side, cin)


def _step_gflop_per_img(crop_flops):
    """teacher forward on the two 224^2 crops + student forward and backward (2x forward) on all ten crops"""
    g, l = crop_flops(224), crop_flops(96)
    return (2 * g + 3 * (2 * g + 8 * l)) / 1e9


def _analytic_gflops():
    from esvit_amd.models.vision_longformer import VIL_SPECS
    out = {name: _step_gflop_per_img(lambda S, a=arch: _vil_crop_flops(S, a)) for name, arch in VIL_SPECS.items()}
    for name, (C, depth) in {"deit_tiny": (192, 12), "deit_small": (384, 12), "vit_base": (768, 12)}.items():
        out[name] = _step_gflop_per_img(lambda S, C=C, depth=depth: _vit_crop_flops(S, C, depth))
    return out


BF16_PEAK_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E (MI355X_MICROARCH.md; ~6.3 TB/s is what a streaming kernel reaches)
OUT_DIM = 65536
PMC_TRAFFIC_FILES = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f)
                     for f in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")]


def pmc_mfma_busy(arch, batch):
    """Counter-derived MFMA-busy of the whole step and of the GEMM family from the committed rocprofv3 PMC pass of this command
    (tools/pmc_mfma_busy.sh: sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE per XCD instance)) -- BASELINE.json's metric asks
    for "MFMA util %".  Not measured in this process.  -> dict or None"""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06_step_mfma_busy.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
    except (OSError, ValueError):
        return None
    if d.get("arch") != arch or d.get("batch") != batch:
        return None
    fam = d.get("families", {})
    return {"step": d.get("mfma_busy"), "step_vs_kernel_time_at_2p4GHz": d.get("mfma_busy_vs_kernel_time_at_2p4GHz"),
            "gemm_dma_kernel": (fam.get("gemm_dma_kernel") or {}).get("mfma_busy"), "gemm_p8_kernel": (fam.get("gemm_p8_kernel") or {}).get("mfma_busy"),
            "source": "rocprofv3 PMC pass of this command, committed as profiles/r06_step_mfma_busy.json (not read in this run)"}


def pmc_gemm_traffic_per_launch(arch, batch, launches_per_step):
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate
    runs of this same command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; produced by
    tools/pmc_traffic.py).  -> (bytes per launch, source file) or (None, None) when no PMC run exists for this arch / batch.
    The number is NOT measured in this process (a PMC pass needs rocprofv3 around the whole run): the bench line says so in
    roofline.traffic_source."""
    for path in PMC_TRAFFIC_FILES:  # the newest committed PMC run wins
        try:
            with open(path) as fh:
                runs = json.load(fh)["runs"]
        except (OSError, ValueError, KeyError):
            continue
        for r in runs:
            if r["arch"] == arch and r["batch"] == batch:
                return r["gemm_bytes_per_step"] / launches_per_step, "profiles/" + os.path.basename(path)
    return None, None


def build(dev, drop_path, arch="swin_tiny_w7"):
    import esvit_amd
    from esvit_amd import config as CFG
    if arch in ("deit_tiny", "deit_small", "vit_base"):  # built by name, main_esvit.py:305-327
        from esvit_amd.models import vision_transformer as vits
        student = vits.__dict__[arch](patch_size=16, drop_path_rate=drop_path, use_dense_prediction=True)
        teacher = vits.__dict__[arch](patch_size=16, use_dense_prediction=True)
        for m in (student, teacher):
            m.head, m.head_dense = esvit_amd.DINOHead(m.embed_dim, OUT_DIM), esvit_amd.DINOHead(m.embed_dim, OUT_DIM)
        student, teacher = student.to(dev), teacher.to(dev)
        teacher.load_state_dict(student.state_dict())
        for p in teacher.parameters():
            p.requires_grad = False
        return student, teacher, esvit_amd.DDINOLoss(OUT_DIM, 10, 0.04, 0.04, 0, 100).to(dev)
    cfg = CFG.model_config(arch, DROP_PATH=drop_path) if arch.startswith("vil_") else CFG.model_config(arch, DROP_PATH_RATE=drop_path)
    student = esvit_amd.build_model(cfg, use_dense_prediction=True)
    student.head = esvit_amd.DINOHead(student.num_features, OUT_DIM)
    student.head_dense = esvit_amd.DINOHead(student.num_features, OUT_DIM)
    teacher = esvit_amd.build_model(cfg, is_teacher=True, use_dense_prediction=True)
    teacher.head = esvit_amd.DINOHead(teacher.num_features, OUT_DIM)
    teacher.head_dense = esvit_amd.DINOHead(teacher.num_features, OUT_DIM)
    student, teacher = student.to(dev), teacher.to(dev)
    teacher.load_state_dict(student.state_dict())
    for p in teacher.parameters():
        p.requires_grad = False
    loss = esvit_amd.DDINOLoss(OUT_DIM, 10, 0.04, 0.04, 0, 100).to(dev)
    return student, teacher, loss


def synthetic_crops(*a, **k):
    from esvit_amd.data import synthetic_crops as f
    return f(*a, **k)


def _cpu_threads():
    # the small per-op tensors of this workload scale poorly past a few dozen threads (128 threads measured 5x slower than
    # 8), so the baseline uses at most 32 host threads and reports the count it used as `cores`
    n = min(32, os.cpu_count() or 1)
    torch.set_num_threads(n)
    return n


def _port_baseline(dense, bs, steps):
    """oracle/esvit_oracle.py (restatement of the reference's PyTorch path): one full step = teacher fwd, student fwd, loss,
    backward, clip + AdamW + EMA, centre update; fp32"""
    from oracle import esvit_oracle as O
    import esvit_amd
    from esvit_amd import config as CFG
    torch.manual_seed(0)
    cfg = CFG.swin_config("swin_tiny_w7", DROP_PATH_RATE=0.0)
    m = esvit_amd.build_model(cfg, use_dense_prediction=dense)
    m.head = esvit_amd.DINOHead(m.num_features, OUT_DIM)
    if dense:
        m.head_dense = esvit_amd.DINOHead(m.num_features, OUT_DIM)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    all_names = [n for n, _ in m.named_parameters()]
    params = {n: sd[n] for n in all_names}
    teacher = {n: sd[n].clone() for n in all_names}
    reg = {n for n in names if not (n.endswith(".bias") or sd[n].ndim == 1)}
    crops = synthetic_crops(bs) if dense else synthetic_crops(bs)[:2]
    ncrops = len(crops)
    c0, cg0 = torch.zeros(1, OUT_DIM), torch.zeros(1, OUT_DIM)
    state = {}
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        leaf = {n: params[n].detach().requires_grad_(True) for n in names}
        full = dict(sd)
        full.update(params)
        full.update(leaf)
        tfull = dict(sd)
        tfull.update(teacher)
        with torch.no_grad():
            t_out = O.swin_multicrop(tfull, crops[:2], O.SWIN_T, dense=dense)
        s_out = O.swin_multicrop(full, crops, O.SWIN_T, dense=dense)
        if dense:
            loss, bc, bg = O.ddino_loss(s_out, t_out, c0, cg0, 0.04, ncrops)
        else:
            loss, bc = O.dino_loss(s_out, t_out, c0, 0.04, ncrops)
        loss.backward()
        with torch.no_grad():
            c0 = O.center_update(c0, bc, 2 * bs)
            if dense:
                cg0 = O.center_update(cg0, bg, 98 * bs)
            grads = {n: leaf[n].grad for n in names}
            O.clip_adamw_ema(params, grads, state, teacher, reg, 5e-4, 0.04, 0.996, clip=3.0)
        times.append(time.perf_counter() - t0)
    times = sorted(times[1:])
    return times[len(times) // 2]


def _reference_baseline(dense, bs, steps):
    """the reference's OWN modules (models.build_model, DINOHead, DINOLoss / DDINOLoss, utils.clip_gradients,
    torch.optim.AdamW over utils.get_params_groups, the EMA loop of main_esvit.py:587-590) imported from /root/reference under
    the shims of SURVEY.md 8c; fp32 (the CPU path of main_esvit.py:541-574 with fp16_scaler None)"""
    from oracle import ref_loader as RL
    ns = RL.load()
    RL.ensure_single_process_group()
    torch.manual_seed(0)
    cfg = RL.swin_config()

    def make(teacher):
        m = ns.models.build_model(cfg, is_teacher=teacher, use_dense_prediction=dense)
        m.head = ns.DINOHead(m.num_features, OUT_DIM)
        if dense:
            m.head_dense = ns.DINOHead(m.num_features, OUT_DIM)
        return m
    student, teacher = make(False), make(True)
    teacher.load_state_dict(student.state_dict())
    for p in teacher.parameters():
        p.requires_grad = False
    crops = synthetic_crops(bs) if dense else synthetic_crops(bs)[:2]
    loss_fn = (ns.DDINOLoss if dense else ns.DINOLoss)(OUT_DIM, len(crops), 0.04, 0.04, 0, 100)
    opt = torch.optim.AdamW(ns.utils.get_params_groups(student))
    for i, pg in enumerate(opt.param_groups):
        pg["lr"] = 5e-4
        if i == 0:
            pg["weight_decay"] = 0.04
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        teacher_out = teacher(crops[:2])
        student_out = student(crops)
        loss = loss_fn(student_out, teacher_out, 1, None)
        opt.zero_grad()
        loss.backward()
        ns.utils.clip_gradients(student, 3.0)
        opt.step()
        with torch.no_grad():
            for pq, pk in zip(student.parameters(), teacher.parameters()):
                pk.data.mul_(0.996).add_((1 - 0.996) * pq.detach().data)
        times.append(time.perf_counter() - t0)
    times = sorted(times[1:])
    return times[len(times) // 2]


def cpu_baseline():
    """both CPU legs SURVEY.md 8(d) asks for, bounded to ~10-30 s each: BASELINE config 2's CPU twin (2x224 + 8x96 crops, V+R,
    bs 2) as the main entry, BASELINE config 1 (2x224 crops, view-level loss, bs 4) under `config1`"""
    from oracle import ref_loader as RL
    cores = _cpu_threads()
    use_ref = RL.available()
    run = _reference_baseline if use_ref else _port_baseline
    kind = "reference" if use_ref else "port"
    what = ("the reference's own modules (/root/reference via oracle/ref_loader.py)" if use_ref else "oracle/esvit_oracle.py (restatement of the reference path)")
    med2 = run(True, 2, 4)
    med1 = run(False, 4, 4)
    return {"value": 2 / med2, "unit": "images/s", "cores": cores, "kind": kind,
            "sample": "%s, fp32, Swin-T W7 2x224+8x96 V+R out_dim 65536, bs=2, median of 4 full steps (%.2f s/step), %d host threads" % (what, med2, cores),
            "config1": {"value": 4 / med1, "unit": "images/s", "cores": cores, "kind": kind,
                        "sample": "%s, fp32, Swin-T W7 2x224 crops only, view-level DINOLoss, bs=4, median of 4 full steps (%.2f s/step)" % (what, med1)}}


def torch_eager_gpu_baseline(dev, bs, steps=5):
    """The same CPU-oracle code (a port of the reference's PyTorch path) run on the GPU under torch.autocast(bf16): what
    stock PyTorch-ROCm (hipBLASLt + eager aten kernels) delivers for this step on the same MI355X.  A second BASELINE leg
    like cpu_baseline (the oracle is the thing compared against, never part of the measured product path); off unless
    --torch-eager is given, never part of `value`."""
    from oracle import esvit_oracle as O
    import esvit_amd
    from esvit_amd import config as CFG
    torch.manual_seed(0)
    cfg = CFG.swin_config("swin_tiny_w7", DROP_PATH_RATE=0.0)
    m = esvit_amd.build_model(cfg, use_dense_prediction=True)
    m.head = esvit_amd.DINOHead(m.num_features, OUT_DIM)
    m.head_dense = esvit_amd.DINOHead(m.num_features, OUT_DIM)
    sd = {k: v.clone().to(dev) for k, v in m.state_dict().items()}
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    all_names = [n for n, _ in m.named_parameters()]
    params = {n: sd[n] for n in all_names}
    teacher = {n: sd[n].clone() for n in all_names}
    reg = {n for n in names if not (n.endswith(".bias") or sd[n].ndim == 1)}
    crops = [c.to(dev) for c in synthetic_crops(bs)]
    c0, cg0 = torch.zeros(1, OUT_DIM, device=dev), torch.zeros(1, OUT_DIM, device=dev)
    state = {}

    def step():
        nonlocal c0, cg0
        leaf = {n: params[n].detach().requires_grad_(True) for n in names}
        full = dict(sd)
        full.update(params)
        full.update(leaf)
        tfull = dict(sd)
        tfull.update(teacher)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            with torch.no_grad():
                t_out = O.swin_multicrop(tfull, crops[:2], O.SWIN_T)
            s_out = O.swin_multicrop(full, crops, O.SWIN_T)
            loss, bc, bg = O.ddino_loss(s_out, t_out, c0, cg0, 0.04, 10)
        loss.backward()
        with torch.no_grad():
            c0, cg0 = O.center_update(c0, bc.float(), 2 * bs), O.center_update(cg0, bg.float(), 98 * bs)
            O.clip_adamw_ema(params, {n: leaf[n].grad for n in names}, state, teacher, reg, 5e-4, 0.04, 0.996, clip=3.0)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": bs / dt, "unit": "images/s", "batch": bs, "ms_per_step": dt * 1e3,
            "what": "oracle/esvit_oracle.py (port of the reference path) on the same GPU, torch eager + autocast(bf16)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="images per GPU (BASELINE.json configs 3/4: 1024 over 8 GPUs)")
    ap.add_argument("--drop-path", type=float, default=0.1)
    ap.add_argument("--arch", default="swin_tiny_w7", choices=["swin_tiny_w7", "swin_tiny_w14", "swin_base_w14", "swin_small_w7", "swin_base_w7", "cvt_s1", "deit_tiny", "deit_small", "vit_base", "vil_tiny", "vil_small"],
                    help="BASELINE.json's metric is quoted on swin_tiny_w7 (default); configs 3/4 are swin_tiny_w14 / swin_base_w14")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--single-stream", action="store_true", help="teacher forward and weight-gradient GEMMs on the main stream too (per-kernel profiles: no overlapped durations)")
    ap.add_argument("--augment", action="store_true", help="produce the crops INSIDE the timed step with the GPU crop producer (esvit_amd.data: "
                    "DataAugmentationDINO on decoded uint8 images resident in HBM) instead of feeding fixed crop tensors")
    ap.add_argument("--grad-payload", default="fp32", choices=["fp32", "bf16"], help="wire format of the data-parallel gradient all-reduce "
                    "(bf16: half the bytes per xGMI link; moments and parameters stay fp32)")
    ap.add_argument("--per-group", action="store_true", help="one backbone pass per resolution group (the reference's schedule) instead of the ragged multi-crop route")
    ap.add_argument("--gemm-table", default=None, help="write a per-shape GEMM time table to this file")
    ap.add_argument("--torch-eager", type=int, default=0, metavar="BATCH",
                    help="also time the reference-path port under torch eager + autocast(bf16) on this GPU at the given batch")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become N ranks (one process per GPU) under torch.distributed.run
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1 or os.environ.get("ESVIT_FORCE_REDUCER") == "1":
        if world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import esvit_amd
    from esvit_amd import functional as F
    from esvit_amd import ops
    from esvit_amd.engine import EsvitTrainer
    esvit_amd.set_precision("bf16")
    torch.manual_seed(0)  # identical replicas ...
    student, teacher, loss_fn = build(dev, args.drop_path, args.arch)
    if args.per_group:
        student.ragged_multi_crop = teacher.ragged_multi_crop = False
    torch.manual_seed(1000 + rank)  # ... but every rank draws its own stochastic-depth masks (and has its own crops)
    if args.single_stream:
        F.WGRAD_STREAM = False
    trainer = EsvitTrainer(student, teacher, loss_fn, clip_grad=3.0, freeze_last_layer=1, teacher_stream=not args.single_stream,
                           grad_payload=args.grad_payload)
    B = args.batch
    crops = [c.to(dev) for c in synthetic_crops(B, seed=1234 + rank)]
    # constants from the first post-warm-up iteration of the reference schedules
    lr, wd, mom, epoch = 5e-4 * B * world / 256.0, 0.04, 0.996, 1

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    next_crops = lambda: crops  # noqa: E731
    if args.augment:  # decoded images (ImageNet-like sizes) resident in HBM -> the 10-crop list, drawn anew every step
        import numpy as np
        from esvit_amd import data as D
        rs = np.random.default_rng(99 + rank)
        decoded = D.PackedImages([torch.from_numpy(rs.integers(0, 256, (int(h), int(w), 3), dtype=np.uint8)).to(dev)
                                  for h, w in zip(rs.integers(300, 520, B), rs.integers(300, 520, B))])
        producer = D.DataAugmentationDINO((0.4, 1.0), (0.05, 0.4), (8,), (96,), seed=7 + rank)
        # the random draws need the image sizes only: a DataLoader worker makes them (DataAugmentationDINO.collate); here one
        # background thread draws the next step's rows.  As data.GpuAugmentedLoader does, the crops of step n + 1 are rendered on a
        # second HIP stream while step n runs; the step's stream only waits for an event.
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(1)
        aug_stream = torch.cuda.Stream()
        state = {"draws": pool.submit(producer.draw, decoded)}

        def render():
            draws = state["draws"].result()
            state["draws"] = pool.submit(producer.draw, decoded)
            aug_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(aug_stream):
                out = producer(decoded, draws=draws)
                ev = torch.cuda.Event()
                ev.record(aug_stream)
            return out, ev
        state["ready"] = render()

        def next_crops():
            out, ev = state["ready"]
            state["ready"] = render()
            main = torch.cuda.current_stream()
            main.wait_event(ev)
            for c in out:
                c.record_stream(main)
            return out
    for _ in range(args.warmup):
        trainer.step(next_crops(), lr, wd, mom, epoch)
    sync()
    t0 = time.perf_counter()
    loss, losses = None, []
    for i in range(args.steps):
        loss = trainer.step(next_crops(), lr, wd, mom, epoch)
        losses.append(loss)  # (device scalars: read after the timed region)
    sync()
    dt = time.perf_counter() - t0
    # roofline leg, OUTSIDE the timed region: HIP events around every GEMM launch of PROF_STEPS extra steps (two event records
    # per launch cost ~2.7 us of stream time each, ~1.3 ms per instrumented step)
    PROF_STEPS = 4
    prof, prof_steps = None, 0
    if not args.no_roofline:
        ops._EVENT_POOL.extend(torch.cuda.Event(enable_timing=True) for _ in range(2400 * PROF_STEPS))
        prof = []
        ops.GEMM_PROFILE = prof
        side, trainer._side = trainer._side, None  # one stream while instrumented: a launch's events then bracket that launch alone
        wg, F.WGRAD_STREAM = F.WGRAD_STREAM, False  # (the weight-gradient GEMMs back in line too: functional._side_run)
        for _ in range(PROF_STEPS):  # every rank runs them (the collectives need all ranks); rank 0 reports
            trainer.step(crops, lr, wd, mom, epoch)
            prof_steps += 1
        sync()
        ops.GEMM_PROFILE = None
        trainer._side, F.WGRAD_STREAM = side, wg
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = t.item()
    loss_v = loss.item()
    # a step whose loss is not finite had its update refused by the non-finite guard: such a line is not a measurement (rounds
    # 2-4 reported the ViT / CvT / ViL configurations that way without noticing: DESIGN.md 6, the head_dim-64 attention store)
    losses_finite = bool(torch.isfinite(torch.stack(losses).float()).all())
    refused = int(trainer.updater.take_skipped()) if hasattr(trainer.updater, "take_skipped") else 0

    if rank == 0:
        ips = args.steps * B * world / dt
        out = {"metric": "images/sec (global+local crops) Swin-T W=7 V+R" if args.arch == "swin_tiny_w7" else "images/sec (global+local crops) %s V+R" % args.arch, "value": ips, "unit": "images/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
               "data": "synthetic decoded uint8 images -> GPU crop producer (in the timed step)" if args.augment else "synthetic",
               "config": {"workload": "%s, 2x224^2+8x96^2 crops, DDINOLoss (view+region), out_dim 65536, per-param clip 3.0 + "
                                      "AdamW + teacher EMA, drop_path %.2f" % ({"swin_tiny_w7": "Swin-T W=7"}.get(args.arch, args.arch), args.drop_path),
                          "global_batch": B * world, "per_gpu_batch": B, "parallelism": "dp%d" % world},
               "final_loss": loss_v, "losses_finite": losses_finite, "refused_updates": refused,
               # (sanity reference, not a parity claim: with random-init heads both cross-entropies start at ln(out_dim) = 11.09; parity
               # of the loss against the reference's own module is asserted in tests/test_parity_gpu.py / test_step_gpu.py)
               "ln_out_dim": math.log(65536.0),
               "step_mfma_frac": None}
        gf = GFLOP_PER_IMG_BY_ARCH.get(args.arch) or _analytic_gflops().get(args.arch)  # SURVEY.md 8(d), or counted here (ViT / ViL)
        if gf:
            out["gflop_per_image"] = gf
            out["step_mfma_frac"] = ips / world * gf / 1e3 / BF16_PEAK_TFLOPS
        if prof:
            tot_fl = sum(r[0] for r in prof)
            tot_by = sum(r[4] for r in prof)
            tot_ms = sum(r[1].elapsed_time(r[2]) for r in prof)
            if args.gemm_table:
                agg = {}
                for f, a, b, key, by in prof:
                    e = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
                    e[0] += 1
                    e[1] += a.elapsed_time(b)
                    e[2] += f
                    e[3] += by
                with open(args.gemm_table, "w") as fh:
                    for key, (n, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                        fh.write("M=%7d N=%6d K=%7d aks=%d bks=%d splitk=%3d calls/step %5.1f ms/step %7.3f TF %7.1f GB/s %6.0f MB/call %7.1f\n" % (
                            key + (n / prof_steps, ms / prof_steps, fl / ms / 1e9, by / ms / 1e6, by / n / 1e6)))
            # The GEMM family (all fwd / dgrad / wgrad launches of the step) is the dominant kernel; SURVEY.md 8(d) bounds the
            # dense contractions by the bf16 MFMA peak: achieved = sum(2MNK) / sum(launch time).  The HBM view of the same
            # launches (algorithmic bytes = A + B + C (+ side tensors) once per launch) is reported beside it: on this
            # workload's unfused shapes the family moves ~156 FLOP per algorithmic byte, half the 312 FLOP/B ridge.
            gbs = tot_by / (tot_ms * 1e-3) / 1e9
            ach_tf = tot_fl / (tot_ms * 1e-3) / 1e12
            traffic, traffic_file = pmc_gemm_traffic_per_launch(args.arch, B, len(prof) / prof_steps)
            out["roofline"] = {"bound": "mfma", "kernel": "gemm_p8_kernel / gemm_dma_kernel (all fwd/dgrad/wgrad GEMM launches of the step)",
                               "achieved": ach_tf, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / BF16_PEAK_TFLOPS,
                               "traffic": traffic,
                               "traffic_source": None if traffic is None else "rocprofv3 PMC passes of this command, committed as %s (not read in this run)" % traffic_file,
                               "launches_per_step": len(prof) / prof_steps, "instrumented_steps": prof_steps,
                               "flops_per_launch": tot_fl / len(prof), "algorithmic_bytes_per_launch": tot_by / len(prof),
                               "avg_launch_us": tot_ms * 1e3 / len(prof), "gemm_ms_per_step": tot_ms / prof_steps,
                               "flop_per_byte": tot_fl / tot_by, "mfma_busy": pmc_mfma_busy(args.arch, B),
                               "hbm_view": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        if args.torch_eager and world == 1:
            del trainer, student, teacher, loss_fn, crops
            torch.cuda.empty_cache()
            out["torch_eager_gpu"] = torch_eager_gpu_baseline(dev, args.torch_eager)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()
    if not losses_finite:
        print("bench.py: a timed step produced a non-finite loss (its update was refused) -- this line is not a measurement", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
