"""The EsViT pre-training step as one call (integration level L2 of SURVEY.md 8b) and the data-parallel gradient
exchange.  ``train_one_epoch`` has the reference's signature (main_esvit.py:499-501) so it can be installed with
``main_esvit.train_one_epoch = esvit_amd.engine.train_one_epoch``; it replaces lines 565-590 (zero_grad, backward,
185 x .item() clipping, AdamW, EMA loop) by backward + bucketed RCCL all-reduce + the fused update kernels.
"""
import math
import os
import sys

import torch
import torch.distributed as dist

from . import params as P
from .update import FusedClipAdamWEMA


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class GradBucketReducer:
    """Data-parallel gradient averaging over RCCL/xGMI, overlapped with backward.

    Parameters are packed into flat fp32 buckets in *reverse registration order* (the order autograd produces
    their gradients).  A post-accumulate hook copies each finished gradient into its bucket slot and, when a
    bucket is full, launches ``all_reduce`` asynchronously (RCCL runs it on its own stream, concurrently with the
    rest of backward).  ``finish()`` waits for the collectives and re-points ``p.grad`` at the averaged bucket
    views.  One process per GPU; world_size 1 short-circuits everything.

    Bucket size: xGMI is point-to-point (7 links x ~153 GB/s), a ring all-reduce of S bytes moves 2*(7/8)*S per
    link, so 64 MiB buckets (~0.8 ms each on one ring) amortise launch latency while leaving 4-5 buckets for
    Swin-T's 295 MB of fp32 gradients to pipeline against backward.
    """

    def __init__(self, module, bucket_mb=64, process_group=None):
        self.group = process_group
        self.world = _world()
        self.params = [p for p in module.parameters() if p.requires_grad]
        # ESVIT_FORCE_REDUCER=1 exercises the hook/bucket/all-reduce machinery on a single rank (used to validate the
        # RCCL code path on a 1-GPU box)
        self.enabled = self.world > 1 or (os.environ.get("ESVIT_FORCE_REDUCER") == "1" and dist.is_initialized())
        self.buckets, self.slot = [], {}
        if not self.enabled:
            return
        cap = int(bucket_mb * 1024 * 1024 // 4)
        cur, cur_n = [], 0
        for p in reversed(self.params):
            if cur and cur_n + p.numel() > cap:
                self.buckets.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(cur)
        dev = self.params[0].device
        self.flat, self.pending, self.handles = [], [], []
        for bi, plist in enumerate(self.buckets):
            n = sum(p.numel() for p in plist)
            self.flat.append(torch.zeros(n, dtype=torch.float32, device=dev))
            off = 0
            for p in plist:
                self.slot[id(p)] = (bi, off, p.numel())
                off += p.numel()
            self.pending.append(len(plist))
        self._inv_world = torch.full((), 1.0 / self.world, dtype=torch.float32, device=dev)
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._hook)
        self._armed = False

    def begin(self):
        """call before backward"""
        if not self.enabled:
            return
        self.handles = []
        self.pending = [len(b) for b in self.buckets]
        self._armed = True

    def _hook(self, p):
        if not self._armed or p.grad is None:
            return
        bi, off, n = self.slot[id(p)]
        view = self.flat[bi][off:off + n].view_as(p)
        view.copy_(p.grad)
        p.grad = view
        self.pending[bi] -= 1
        if self.pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi):
        h = dist.all_reduce(self.flat[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.handles.append((bi, h))

    def finish(self):
        """call after backward: flush buckets whose parameters received no gradient, wait for the collectives"""
        if not self.enabled:
            return
        self._armed = False
        for bi, left in enumerate(self.pending):
            if left > 0:  # e.g. last_layer frozen in epoch 0: its slots keep zeros
                self._launch(bi)
        for bi, h in self.handles:
            h.wait()
            if self.world > 1:
                if self.flat[bi].is_cuda:
                    from . import ops
                    ops.scale_inplace(self.flat[bi], self._inv_world)  # SUM -> mean
                else:
                    self.flat[bi].mul_(self._inv_world)
        self.handles = []


class EsvitTrainer:
    """teacher fwd -> student fwd -> loss -> backward (+ overlapped grad all-reduce) -> fused clip/AdamW/EMA."""

    def __init__(self, student, teacher, loss_fn, clip_grad=3.0, freeze_last_layer=1, bucket_mb=64):
        self.student, self.teacher, self.loss_fn = student, teacher, loss_fn
        self.clip_grad, self.freeze_last_layer = clip_grad, freeze_last_layer
        self.updater = FusedClipAdamWEMA(student, teacher)
        self.reducer = GradBucketReducer(student, bucket_mb)
        loss_fn.assume_unit_grad = True  # loss.backward() below always uses grad_output == 1

    def step(self, images, lr, wd, momentum, epoch):
        with torch.no_grad():
            teacher_out = self.teacher(images[:2])
        student_out = self.student(images)
        loss = self.loss_fn(student_out, teacher_out, epoch, None)
        self.reducer.begin()
        loss.backward()
        self.reducer.finish()
        self.updater.step(lr, wd, momentum, clip_grad=self.clip_grad, skip_last_layer=epoch < self.freeze_last_layer)
        self.updater.zero_grad(set_to_none=True)  # (the updater invalidated / refreshed the cached weight casts itself)
        return loss.detach()


_TRAINERS = {}


def train_one_epoch(student, teacher, teacher_without_ddp, dino_loss, data_loader, optimizer, lr_schedule, wd_schedule,
                    momentum_schedule, epoch, mixup_fn, fp16_scaler, args):
    """Drop-in for main_esvit.train_one_epoch (same signature).  `optimizer` may be a FusedClipAdamWEMA (its state is
    then used and checkpointed by the caller) or a torch optimizer (ignored in favour of a trainer-owned fused one);
    `student` may be DDP-wrapped (its .module is trained, gradients are reduced by GradBucketReducer instead)."""
    if mixup_fn is not None:
        raise NotImplementedError("mixup (main_esvit.py:518-534) is out of scope (SURVEY.md 8f-3)")
    net = student.module if hasattr(student, "module") else student
    key = (id(net), id(teacher_without_ddp), id(dino_loss))
    tr = _TRAINERS.get(key)
    if tr is None:
        tr = EsvitTrainer(net, teacher_without_ddp, dino_loss, clip_grad=args.clip_grad, freeze_last_layer=args.freeze_last_layer)
        if isinstance(optimizer, FusedClipAdamWEMA):
            tr.updater = optimizer
        _TRAINERS[key] = tr
    n_it, total, last = len(data_loader), 0.0, None
    for it, (images, _) in enumerate(data_loader):
        git = n_it * epoch + it
        images = [im.cuda(non_blocking=True) for im in images]
        last = tr.step(images, lr_schedule[git], wd_schedule[git], momentum_schedule[git], epoch)
        if it % 10 == 0 or it == n_it - 1:  # the reference syncs every iteration (main_esvit.py:546,593); 1-in-10 keeps the NaN guard
            v = last.item()
            if not math.isfinite(v):
                print("Loss is {}, stopping training".format(v))
                sys.exit(1)
            total = v
    return {"loss": total, "lr": float(lr_schedule[n_it * epoch + n_it - 1]), "wd": float(wd_schedule[n_it * epoch + n_it - 1])}
