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

from . import functional as F
from . import params as P
from .update import FusedClipAdamWEMA, bind_torch_optimizer, optimizer_rule


def _world(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class GradBucketReducer:
    """Data-parallel gradient averaging over RCCL/xGMI, overlapped with backward.

    Parameters are packed into flat fp32 buckets in *reverse registration order* (the order autograd produces their
    gradients); every slot starts on a 16-byte boundary (the fused update reads gradients with 16-byte vector loads).
    The weight-gradient GEMMs write straight into their bucket slot (esvit_amd.params.grad_out: the autograd
    functions pass the slot as the GEMM's output and hand autograd a fresh alias of it, which AccumulateGrad adopts
    without a copy), so the large tensors are never copied; the small ones (biases, LayerNorm, bias tables) are copied
    by the post-accumulate hook.  When the last gradient of a bucket has arrived its
    ``all_reduce`` is launched asynchronously -- RCCL runs it on its own stream, concurrently with the rest of
    backward -- as an AVG reduction (no separate scale pass; gloo, used by the CPU tests, sums and scales).
    ``finish()`` waits for the collectives.  One process per GPU; world_size 1 short-circuits everything.

    Bucket size: xGMI is point-to-point (7 links x ~153 GB/s), a ring all-reduce of S bytes moves 2*(7/8)*S per
    link, so a 32 MiB bucket is ~0.4 ms on one ring -- long enough to amortise launch latency, and nine to ten of them
    pipeline Swin-T's 295 MB of fp32 gradients against a ~35 ms backward.  What stays exposed is the LAST bucket (the
    earliest layers, whose gradients complete when backward ends): halving the bucket halves that tail.
    """

    ALIGN = 4  # floats

    def __init__(self, module, bucket_mb=32, process_group=None, overlap=True, payload="fp32"):
        # A bucket is launched when the post-accumulate hook of its last parameter has fired.  AccumulateGrad runs once
        # per parameter and backward -- after autograd has summed every contribution of the graph -- so this is correct
        # for any schedule: the ragged route (one contribution per parameter), one backbone pass per resolution group,
        # CvT.  A bucket slot is handed to a producer at most once per backward (params.grad_out); a sum that did not end
        # up in its slot is packed by the hook.  overlap=False (debug): nothing is launched before backward has ended and
        # no producer writes into a bucket.
        self.overlap = overlap
        # payload "bf16": a finished bucket is rounded to bf16 into a wire buffer, THAT is all-reduced (half the bytes over the
        # seven xGMI links: 147 instead of 295 MB per step for Swin-T), and the mean is widened back into the fp32 bucket the fused
        # update reads -- the moments and the parameters are updated in fp32 either way.  Default "fp32": bit-for-bit the
        # reference's DDP arithmetic.
        assert payload in ("fp32", "bf16"), payload
        self.payload = payload
        self.group = process_group
        self.world = _world(process_group)
        self.params = [p for p in module.parameters() if p.requires_grad]
        # ESVIT_FORCE_REDUCER=1 exercises the hook/bucket/all-reduce machinery on a single rank (used to validate the
        # RCCL code path on a 1-GPU box)
        self.enabled = self.world > 1 or (os.environ.get("ESVIT_FORCE_REDUCER") == "1" and dist.is_initialized())
        self.buckets, self.slot = [], {}
        self._hooks = []
        if not self.enabled:
            return
        cap = int(bucket_mb * 1024 * 1024 // 4)
        al = self.ALIGN
        cur, cur_n = [], 0
        for p in reversed(self.params):
            n = -(-p.numel() // al) * al
            if cur and cur_n + n > cap:
                self.buckets.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += n
        if cur:
            self.buckets.append(cur)
        dev = self.params[0].device
        self.flat, self.pending, self.handles = [], [], []
        for bi, plist in enumerate(self.buckets):
            off = 0
            for p in plist:
                self.slot[id(p)] = (bi, off, p.numel())
                off += -(-p.numel() // al) * al
            self.flat.append(torch.zeros(off, dtype=torch.float32, device=dev))
            self.pending.append(len(plist))
        assert all(f.data_ptr() % 16 == 0 for f in self.flat)
        self.wire = [torch.empty(f.numel(), dtype=torch.bfloat16, device=dev) for f in self.flat] if payload == "bf16" else None
        backend = dist.get_backend(self.group) if dist.is_initialized() else ""
        self._avg = backend == "nccl"  # RCCL reduces with AVG in one pass; gloo has SUM only
        self._inv_world = torch.full((), 1.0 / self.world, dtype=torch.float32, device=dev)
        self.views = {}
        for p in self.params:
            bi, off, n = self.slot[id(p)]
            self.views[id(p)] = self.flat[bi][off:off + n].view_as(p)
            self._hooks.append(p.register_post_accumulate_grad_hook(self._arrived))
        self._armed = False

    def begin(self):
        """call before backward (with every p.grad None: autograd then adopts the bucket views as the gradients)"""
        if not self.enabled:
            return
        self.handles = []
        self.pending = [len(b) for b in self.buckets]
        self._armed = True
        P.set_grad_sink(self.views if self.overlap else None)

    def _arrived(self, p):
        if not self._armed or p.grad is None:
            return
        view = self.views[id(p)]
        if p.grad.data_ptr() != view.data_ptr():  # not produced in place: pack it (the small tensors)
            view.copy_(p.grad)
            p.grad = view
        if not self.overlap:
            return  # every bucket is launched by finish()
        bi = self.slot[id(p)][0]
        self.pending[bi] -= 1
        if self.pending[bi] == 0:
            self._launch(bi)

    def _launch(self, bi):
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        buf = self.flat[bi]
        if self.wire is not None:
            buf = self.wire[bi]
            if buf.is_cuda:
                from . import ops
                ops.cast_to_act(self.flat[bi], dtype=torch.bfloat16, out=buf)
            else:
                buf.copy_(self.flat[bi])
        h = dist.all_reduce(buf, op=op, group=self.group, async_op=True)
        self.handles.append((bi, h))

    def finish(self):
        """call after backward: flush buckets whose parameters received no gradient, wait for the collectives"""
        if not self.enabled:
            return
        self._armed = False
        P.set_grad_sink(None)
        for bi, left in enumerate(self.pending):
            if left > 0:  # e.g. the weight-normed last layer's frozen g: its slot keeps zeros
                self._launch(bi)
        for bi, h in self.handles:
            h.wait()
            if self.wire is not None:
                self.flat[bi].copy_(self.wire[bi])  # bf16 mean -> the fp32 bucket
            if not self._avg and self.world > 1:
                if self.flat[bi].is_cuda:
                    from . import ops
                    ops.scale_inplace(self.flat[bi], self._inv_world)  # SUM -> mean
                else:
                    self.flat[bi].mul_(self._inv_world)
        self.handles = []

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


class _ScalerOptimizerView:
    """What ``torch.cuda.amp.GradScaler.unscale_`` / ``.step`` need from an optimizer: ``param_groups`` (whose gradients
    they check for inf / nan and unscale in place) and ``step()`` (called only when every gradient is finite)."""

    def __init__(self, param_groups):
        self.param_groups = param_groups
        self.stepped = False
        self._fn = None

    def step(self):
        self.stepped = True
        self._fn()


class EsvitTrainer:
    """teacher fwd -> student fwd -> loss -> backward (+ overlapped grad all-reduce) -> fused clip/AdamW/EMA."""

    def __init__(self, student, teacher, loss_fn, clip_grad=3.0, freeze_last_layer=1, bucket_mb=32, updater=None, teacher_stream=True,
                 grad_payload="fp32"):
        self.student, self.teacher, self.loss_fn = student, teacher, loss_fn
        # the teacher forward (no autograd, its own scratch) runs on a second HIP stream beside the student forward: the two
        # streams fill each other's tails and launch gaps (+0.7 % at B = 128; same loss to 1e-5)
        self._side = torch.cuda.Stream() if (teacher_stream and next(student.parameters()).is_cuda) else None
        self.clip_grad, self.freeze_last_layer = clip_grad, freeze_last_layer
        self.updater = updater if updater is not None else FusedClipAdamWEMA(student, teacher)
        self.reducer = GradBucketReducer(student, bucket_mb, payload=grad_payload)

    def step(self, images, lr, wd, momentum, epoch, scaler=None, teacher_images=None, targets_mixup=None):
        """scaler: a ``torch.cuda.amp.GradScaler`` (the reference's --use_fp16 mode, main_esvit.py:417-419, 576-584) or None.
        teacher_images / targets_mixup: the un-mixed global views and the per-crop target matrices of the mixup mode
        (main_esvit.py:515-538); `images` are then the mixed student inputs."""
        t_in = images[:2] if teacher_images is None else teacher_images
        arm = getattr(self.loss_fn, "arm_logit_stats", None)
        if arm is not None:  # the heads' last-layer GEMMs also emit the softmax statistics of this step's loss (loss.py)
            arm(self.student, self.teacher, epoch)
        try:
            student_out, teacher_out = self._forwards(images, t_in)
        finally:
            if arm is not None:
                self.loss_fn.disarm_logit_stats(self.student, self.teacher)
        if scaler is not None:
            return self._scaled_update(scaler, student_out, teacher_out, lr, wd, momentum, epoch, targets_mixup)
        # loss.backward() below always uses grad_output == 1: the loss skips its rescale pass for this call only
        prev = getattr(self.loss_fn, "assume_unit_grad", False)
        self.loss_fn.assume_unit_grad = True
        try:
            loss = self.loss_fn(student_out, teacher_out, epoch, targets_mixup)
        finally:
            self.loss_fn.assume_unit_grad = prev
        self.reducer.begin()
        loss.backward()
        F._side_join(final=True)
        self.reducer.finish()
        self.updater.step(lr, wd, momentum, clip_grad=self.clip_grad, skip_last_layer=epoch < self.freeze_last_layer)
        self.updater.zero_grad(set_to_none=True)  # (the updater invalidated / refreshed the cached weight casts itself)
        return loss.detach()

    def _forwards(self, images, t_in):
        # (the first step stays on one stream: it fills the per-geometry index tables and weight casts both networks share)
        warm, self._warm = getattr(self, "_warm", False), True
        if self._side is not None and warm:
            side, main = self._side, torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side), torch.no_grad():
                teacher_out = self.teacher(t_in)
            student_out = self.student(images)
            main.wait_stream(side)
            for t in (teacher_out if isinstance(teacher_out, (tuple, list)) else [teacher_out]):
                if torch.is_tensor(t):
                    t.record_stream(main)
                    for st in getattr(t, "esvit_row_stats", (None,))[1:]:  # (row max, row lse, (rows, column sums) or None)
                        for u in (st if isinstance(st, tuple) else (st,)):
                            if torch.is_tensor(u):
                                u.record_stream(main)
        else:
            with torch.no_grad():
                teacher_out = self.teacher(t_in)
            student_out = self.student(images)
        return student_out, teacher_out

    def _scaled_update(self, scaler, student_out, teacher_out, lr, wd, momentum, epoch, targets_mixup=None):
        """main_esvit.py:576-584 with the fused update as the optimizer: scale(loss).backward() -> unscale_ -> (clip +
        AdamW + EMA if every gradient is finite) -> update().  The activations stay bf16 (the modules keep their own
        precision policy; fp16 is not an activation dtype of the kernels), so the scale factor only ever matters through
        GradScaler's own protocol: power-of-two scales are exact in bf16 / fp32 and the step equals the unscaled one."""
        loss = self.loss_fn(student_out, teacher_out, epoch, targets_mixup)  # grad_output = scale: the loss rescales its gradient
        self.reducer.begin()
        scaler.scale(loss).backward()
        self.reducer.finish()
        view = getattr(self, "_scaler_view", None)
        if view is None:
            view = self._scaler_view = _ScalerOptimizerView(self.updater.param_groups)
        view.stepped = False
        skip = epoch < self.freeze_last_layer
        view._fn = lambda: self.updater.step(lr, wd, momentum, clip_grad=self.clip_grad, skip_last_layer=skip)
        if self.clip_grad:
            scaler.unscale_(view)  # the clip compares true gradient norms (main_esvit.py:579-580)
        scaler.step(view)          # runs view.step() unless a gradient is inf / nan
        scaler.update()
        self.updater.zero_grad(set_to_none=True)
        if not view.stepped:
            # optimizer step skipped: the reference's EMA loop (main_esvit.py:587-590) still runs.  With no gradients the
            # fused kernel leaves the student and the moments untouched and only applies the EMA.
            self.updater.step(lr, wd, momentum, clip_grad=self.clip_grad, skip_last_layer=skip)
        return loss.detach()


_TRAINERS = {}


def train_one_epoch(student, teacher, teacher_without_ddp, dino_loss, data_loader, optimizer, lr_schedule, wd_schedule,
                    momentum_schedule, epoch, mixup_fn, fp16_scaler, args):
    """Drop-in for main_esvit.train_one_epoch (same signature).

    `optimizer`: a FusedClipAdamWEMA is used as is.  What the unmodified train_esvit builds (main_esvit.py:408-415:
    ``torch.optim.AdamW``, ``torch.optim.SGD(lr=0, momentum=0.9)``, ``utils.LARS``) is BOUND to the fused updater: its
    ``state`` entries (step, exp_avg, exp_avg_sq; momentum_buffer; mu) are the very tensors the fused kernel updates and its
    param_groups receive the schedule values, so the caller's ``optimizer.state_dict()`` / ``load_state_dict()``
    (main_esvit.py:444-452, 476-488) checkpoint and restore the real state.  Any other optimizer is refused.  `student` may be DDP-wrapped (its .module is trained, gradients are
    reduced by GradBucketReducer instead).  `fp16_scaler`: the reference's GradScaler protocol is followed with the fused
    update in the optimizer's place (EsvitTrainer._scaled_update); the caller keeps checkpointing ``fp16_scaler.state_dict()``.
    `mixup_fn` (timm's Mixup or any callable (samples, targets) -> (samples, target matrix)): the crops are mixed and the
    target matrices reach the loss exactly as in main_esvit.py:515-544 (DINOLoss uses them, DDINOLoss ignores them).
    Returns the rank-averaged epoch means the reference logs (main_esvit.py:593-600)."""
    net = student.module if hasattr(student, "module") else student
    if optimizer_rule(optimizer) is None:
        raise TypeError("esvit_amd.engine.train_one_epoch drives the optimizers of main_esvit.py:408-415 -- torch.optim.AdamW, "
                        "torch.optim.SGD(momentum), utils.LARS -- or an esvit_amd.update.FusedClipAdamWEMA; got %s" % type(optimizer).__name__)
    key = (id(net), id(teacher_without_ddp), id(dino_loss))
    tr = _TRAINERS.get(key)
    stale = (tr is None or tr.student is not net or tr.loss_fn is not dino_loss or
             (tr.updater is not optimizer and getattr(tr.updater, "bound", None) is not optimizer))
    if stale:
        if tr is not None:
            tr.reducer.close()
        updater = optimizer if isinstance(optimizer, FusedClipAdamWEMA) else bind_torch_optimizer(optimizer, net, teacher_without_ddp)
        tr = EsvitTrainer(net, teacher_without_ddp, dino_loss, clip_grad=args.clip_grad, freeze_last_layer=args.freeze_last_layer,
                          updater=updater)
        _TRAINERS[key] = tr
    n_it = len(data_loader)
    dev = next(net.parameters()).device
    loss_sum = torch.zeros((), dtype=torch.float32, device=dev)
    last = None
    last_look = -1  # iteration of the previous look at the loss / the refused-update counter
    win_refused = win_seen = 0  # consecutive looks in which every update was refused: refusals / updates in them
    for it, (images, _) in enumerate(data_loader):
        git = n_it * epoch + it
        images = [im.cuda(non_blocking=True) for im in images]
        teacher_images, targets_mixup = None, None
        if mixup_fn is not None:  # main_esvit.py:515-538: the first num_mixup_views crops are mixed, the teacher sees the originals
            teacher_images, student_input, targets_mixup, n_mix = images[:2], [], [], 0
            bs = args.batch_size_per_gpu
            for samples in images:
                if n_mix < args.num_mixup_views:
                    samples, targets = mixup_fn(samples, torch.arange(0, bs, dtype=torch.long, device=samples.device))
                    n_mix += 1
                else:
                    targets = torch.eye(bs, device=samples.device)
                student_input.append(samples)
                targets_mixup.append(targets)
            images = student_input
        last = tr.step(images, lr_schedule[git], wd_schedule[git], momentum_schedule[git], epoch, scaler=fp16_scaler,
                       teacher_images=teacher_images, targets_mixup=targets_mixup)
        loss_sum += last
        # the reference syncs on loss.item() every iteration and exits BEFORE the update (main_esvit.py:546-551).  Here the fused
        # update refuses to touch student, teacher and optimizer state when any gradient is non-finite (device-side flag,
        # update.hip), so a NaN step leaves the state as it was; the host looks at the loss one iteration in ten
        if it % 10 == 0 or it == n_it - 1:
            v = loss_sum.item()  # (a NaN / inf of any earlier iteration stays in the running sum)
            if not math.isfinite(v):
                print("Loss is {}, stopping training".format(v))
                sys.exit(1)
            # a gradient overflow with a FINITE loss (bf16 backward) is refused by the update kernel too: say so, take the refused
            # updates back out of the step counts, and stop when nothing but refusals happened since the last look
            refused = tr.updater.take_skipped()
            since = it - last_look  # updates launched since the previous look (the first look of an epoch follows one update)
            last_look = it
            # stop only on a full window of refusals: looks can be 1-9 iterations apart (the end-of-epoch look), and one transient
            # bf16 overflow in such a short window is not "nothing but refusals" -- refusals accumulate across looks until at
            # least 10 updates have been seen, a look with an accepted update in it clears the window
            if refused < since:
                win_refused = win_seen = 0
            else:
                win_refused, win_seen = win_refused + refused, win_seen + since
            if refused:
                print("WARNING: %d of the last %d updates were skipped (non-finite gradients, finite loss)" % (refused, since))
                if win_seen >= 10 and win_refused >= win_seen:
                    print("every one of the last %d updates was skipped, stopping training" % win_seen)
                    sys.exit(1)
    mean = loss_sum / max(n_it, 1)
    if _world() > 1:  # metric_logger.synchronize_between_processes (main_esvit.py:597)
        dist.all_reduce(mean)
        mean = mean / _world()
    # the reference returns every meter's epoch global_avg (main_esvit.py:598-600): the means of the schedule values of this epoch
    its = range(n_it * epoch, n_it * epoch + n_it)
    lr_avg = sum(float(lr_schedule[i]) for i in its) / max(n_it, 1)
    wd_avg = sum(float(wd_schedule[i]) for i in its) / max(n_it, 1)
    return {"loss": mean.item(), "lr": lr_avg, "wd": wd_avg}
