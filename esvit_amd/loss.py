"""DINOLoss / DDINOLoss behind the reference signatures (main_esvit.py:603-770).

forward(student_output, teacher_output, epoch, targets_mixup) -> 0-d fp32 loss whose backward feeds the
student logits; ``state_dict()`` = {center[, center_grid]} as in the reference.  The work is done by
four fused HIP kernels (teacher row statistics, cosine-similarity GEMM + region matching, the student
log-softmax-CE + gradient kernel, centre column sums + EMA); the teacher softmax, the 18 per-pair
log-softmax tensors and the gathered teacher rows of the reference are never materialised.
``targets_mixup`` is accepted and ignored by DDINOLoss exactly as in the reference; the DINOLoss mixup branch
(main_esvit.py:639-641) is a 'next' row.
"""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops


def _ops():
    return ops


class _LossFn(torch.autograd.Function):
    """loss value with precomputed d loss / d logits (the CE kernel produces both in one pass)"""

    @staticmethod
    def forward(ctx, loss, unit_grad, *pairs):
        n = len(pairs) // 2
        ctx.unit_grad = unit_grad
        ctx.save_for_backward(*pairs[n:])
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        grads = list(ctx.saved_tensors)
        if not ctx.unit_grad:
            g = g.contiguous().float()
            grads = [_ops().scale_inplace(d, g) for d in grads]
        return (None, None) + tuple(grads) + (None,) * len(grads)


def _teacher_temp_schedule(warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs, nepochs):
    return np.concatenate((np.linspace(warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs),
                           np.ones(nepochs - warmup_teacher_temp_epochs) * teacher_temp))


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


LOGIT_STATS = os.environ.get("ESVIT_LOGIT_STATS", "1") != "0"


def _heads_of(model):
    """the DINOHead modules of a (possibly wrapped) backbone: (view-level head, region-level head), either may be None"""
    from .head import DINOHead
    m = getattr(model, "module", model)
    hv, hd = getattr(m, "head", None), getattr(m, "head_dense", None)
    return (hv if isinstance(hv, DINOHead) else None), (hd if isinstance(hd, DINOHead) else None)


def _taken_stats(x, token):
    """(row_max, row_lse) a head attached to its logits for exactly this request, else None"""
    st = getattr(x, "esvit_row_stats", None)
    if st is not None and st[0] == token:
        return st[1], st[2]
    return None


def _taken_col_sums(x, token):
    """batch sums per column a teacher head's last-layer GEMM attached to its logits for exactly this request (fp32 [K] over the
    x.shape[0] rows of x), else None"""
    st = getattr(x, "esvit_row_stats", None)
    if st is None or len(st) <= 3 or st[3] is None or st[0] != token:
        return None
    rows, sums = st[3]
    return sums if rows == x.shape[0] and sums.numel() == x.shape[-1] else None


class _DeferredCenter:
    """The centre update of step n is only needed by the loss of step n+1 (main_esvit.py:748, 752-770): the all-reduce of
    the batch sums is launched asynchronously right after they are computed -- RCCL runs it on its own stream under the
    student's backward -- and the EMA is applied when the centres are next read (`synchronize()`: next forward,
    state_dict(), or explicitly).  With one process there is nothing to wait for and the EMA is applied at once."""
    _pending = None

    def _reduce_and_apply(self, buf, apply):
        if _world() > 1:
            self.synchronize()
            h = dist.all_reduce(buf, async_op=True)
            self._pending = (h, buf, apply)
        else:
            apply(buf, 1)

    def synchronize(self):
        if self._pending is not None:
            h, buf, apply = self._pending
            self._pending = None
            h.wait()
            apply(buf, _world())

    # ---- softmax statistics from the heads' last-layer GEMM (esvit_gemm_desc::rowstat) ----
    # The train step announces the coming loss call: the heads then emit, next to the logits, the row statistics this loss would
    # otherwise compute in a pass of its own (teacher: esvit_teacher_row_stats; student: the first pass of the CE kernel).  A request
    # is (inverse temperature, centre, token); the token ties the statistics to the centre values they were computed with -- any
    # centre update in between changes `_center_version` and the loss falls back to its own passes.
    _center_version = 0
    wants_col_sums = False  # a loss whose update_center takes the teacher logits' column sums from the heads sets this

    def _token(self, who, inv_temp):
        return (id(self), who, self._center_version, float(inv_temp))

    def _centers(self):
        return (self.center, getattr(self, "center_grid", None))

    def arm_logit_stats(self, student, teacher, epoch):
        """call before the teacher / student forwards of one step (engine.EsvitTrainer.step does)"""
        if not LOGIT_STATS:
            return
        self.synchronize()  # the centres the teacher statistics use must be final
        inv_tt, inv_st = 1.0 / float(self.teacher_temp_schedule[epoch]), 1.0 / self.student_temp
        for who, model, inv_t, cens in (("t", teacher, inv_tt, self._centers()), ("s", student, inv_st, (None, None))):
            for lvl, (head, cen) in enumerate(zip(_heads_of(model), cens)):
                if head is not None:
                    # (the teacher heads also hand over the column sums of their logits: the centre update's input)
                    head.logit_stats = (inv_t, None if cen is None else cen.view(-1), self._token(who + str(lvl), inv_t), who == "t" and self.wants_col_sums)

    @staticmethod
    def disarm_logit_stats(student, teacher):
        for model in (student, teacher):
            for head in _heads_of(model):
                if head is not None:
                    head.logit_stats = None

    def state_dict(self, *args, **kwargs):
        self.synchronize()
        return super().state_dict(*args, **kwargs)


class DINOLoss(_DeferredCenter, nn.Module):
    def __init__(self, out_dim, ncrops, warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs, nepochs,
                 student_temp=0.1, center_momentum=0.9):
        super().__init__()
        self.student_temp, self.center_momentum, self.ncrops = student_temp, center_momentum, ncrops
        self.register_buffer("center", torch.zeros(1, out_dim))
        self.teacher_temp_schedule = _teacher_temp_schedule(warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs, nepochs)
        self.assume_unit_grad = False  # set by the fused train step: loss.backward() with grad 1 needs no rescale pass
        self._tables = {}

    def _static(self, B, device):
        key = (B, str(device))
        t = self._tables.get(key)
        if t is None:
            n_terms = 2 * self.ncrops - 2
            tm = np.full((self.ncrops * B, 2), -1, dtype=np.int32)
            for v in range(self.ncrops):
                for iq in range(2):
                    if v != iq:
                        tm[v * B:(v + 1) * B, iq] = iq * B + np.arange(B)
            w = np.full((self.ncrops * B,), 1.0 / (n_terms * B), dtype=np.float32)
            t = (torch.from_numpy(tm).to(device), torch.from_numpy(w).to(device))
            self._tables[key] = t
        return t

    def _mixup_terms(self, targets_mixup, B, device):
        """main_esvit.py:639-641: for teacher view iq and student crop v the loss is mean_a -sum_b T_v[a, b] q_a . logp_b, i.e.
        student row (v, b) is scored against the teacher rows a with T_v[a, b] != 0.  Mixup / cutmix targets (and the identity
        of the un-mixed crops) have at most two non-zeros per column, which gives at most four weighted terms per student row.
        Label smoothing (--smoothing > 0, main_esvit.py:230) adds the same constant off_v = smoothing / B to every entry of a mixed
        crop's matrix: T_v = S_v + off_v 1 1^T with S_v sparse as before, and the constant part scores every student row against the
        MEAN teacher distribution of the view (`_smoothing_term`).  -> (tmatch [ncrops*B, 4], weights [ncrops*B, 4], off [ncrops])"""
        T = torch.stack([t.to(device=device, dtype=torch.float32) for t in targets_mixup])       # [ncrops, a, b]
        assert T.shape == (self.ncrops, B, B), "targets_mixup: one [B, B] matrix per crop"
        off = T.amin(dim=(1, 2))                                                                  # 0 without smoothing
        S = T - off.view(-1, 1, 1)
        if not self.__dict__.get("_mixup_checked"):
            # structure check on the FIRST call only (it reads values back from the device: not something for every step); mixup /
            # cutmix / smoothing settings keep this structure for the whole run
            tol = 1e-6 * float(T.abs().max())  # (per-sample mixing ratios leave the constant part exact to an ulp only)
            if int((S.abs() > tol).sum(1).max()) > 2:
                raise NotImplementedError("DINOLoss mixup targets that are not (at most two entries per column) + (one constant per crop): "
                                          "the four-term cross-entropy kernel and the mean-teacher term cover mixup / cutmix / label smoothing")
            self.__dict__["_mixup_checked"] = True
        w2, a2 = torch.topk(S.abs(), 2, dim=1)                                                   # [ncrops, 2, b]
        w2 = torch.gather(S, 1, a2)
        n_terms = 2 * self.ncrops - 2
        tm = torch.full((self.ncrops, B, 4), -1, dtype=torch.int32, device=device)
        tw = torch.zeros((self.ncrops, B, 4), dtype=torch.float32, device=device)
        for iq in range(2):
            for j in range(2):
                tm[:, :, 2 * iq + j] = (iq * B + a2[:, j, :]).to(torch.int32)
                tw[:, :, 2 * iq + j] = w2[:, j, :] / (n_terms * B)
            tm[iq, :, 2 * iq:2 * iq + 2] = -1   # student and teacher on the same view: skipped (main_esvit.py:636-638)
        tm[tw == 0] = -1
        return tm.view(-1, 4).contiguous(), tw.view(-1, 4).contiguous(), off

    def _smoothing_term(self, o, s, t, mx, lse, off, B, inv_st, inv_tt):
        """the constant part of smoothed mixup targets: (1 / B) sum_a sum_b off_v q_a . logp_b = off_v sum_b qbar . logp_b with
        qbar the mean teacher distribution of the view.  qbar enters the same cross-entropy kernel as a teacher row of its own:
        the logit row  c + temp * log(qbar)  has exactly qbar as its centred, sharpened softmax.  (A handful of small torch
        launches on [2B, K]; only runs with --use_mixup and --smoothing > 0.)"""
        K = t.shape[1]
        q = torch.exp((t.float() - self.center) * inv_tt - (mx + lse)[:, None])
        qbar = q.view(2, B, K).mean(1)
        tbar = (torch.log(qbar).clamp_min(-80.0) / inv_tt + self.center).to(t.dtype).contiguous()
        mx2, lse2 = o.teacher_row_stats(tbar, self.center, inv_tt)
        n_terms = 2 * self.ncrops - 2
        tm = torch.full((self.ncrops, B, 4), -1, dtype=torch.int32, device=s.device)
        tw = torch.zeros((self.ncrops, B, 4), dtype=torch.float32, device=s.device)
        for iq in range(2):
            tm[:, :, iq] = iq
            tw[:, :, iq] = (off / n_terms).view(-1, 1)
            tm[iq, :, iq] = -1
            tw[iq, :, iq] = 0.0
        tm[tw == 0] = -1
        return o.dino_ce(s.detach(), tbar, self.center, mx2, lse2, tm.view(-1, 4).contiguous(), None, inv_st, inv_tt,
                         term_w=tw.view(-1, 4).contiguous())

    def forward(self, student_output, teacher_output, epoch, targets_mixup=None):
        o = _ops()
        self.synchronize()
        s, t = student_output.contiguous(), teacher_output.detach().contiguous()
        B = t.shape[0] // 2
        inv_tt = 1.0 / float(self.teacher_temp_schedule[epoch])
        inv_st = 1.0 / self.student_temp
        t_st, s_st = _taken_stats(teacher_output, self._token("t0", inv_tt)), _taken_stats(student_output, self._token("s0", inv_st))
        mx, lse = t_st if t_st is not None else o.teacher_row_stats(t, self.center, inv_tt)
        if targets_mixup:
            tmatch, tw, off = self._mixup_terms(targets_mixup, B, s.device)
            row_loss, ds = o.dino_ce(s.detach(), t, self.center, mx, lse, tmatch, None, inv_st, inv_tt, term_w=tw)
            # The constant part of the targets (label smoothing) is scored on EVERY call unless the caller declared it absent
            # (mixup_smoothing = 0.0): whether a batch carries it is not latched from the first batch -- with off = 0 the
            # term is an exact no-op that costs a handful of small launches on the mixup path only.
            if getattr(self, "mixup_smoothing", None) is None or self.mixup_smoothing > 0:
                row2, ds2 = self._smoothing_term(o, s, t, mx, lse, off, B, inv_st, inv_tt)
                row_loss, ds = row_loss + row2, ds + ds2
        else:
            tmatch, w = self._static(B, s.device)
            row_loss, ds = o.dino_ce(s.detach(), t, self.center, mx, lse, tmatch, w, inv_st, inv_tt, s_stats=s_st)
        loss = o.sum_f32(row_loss)
        self.update_center(t)
        return _LossFn.apply(loss, self.assume_unit_grad, student_output, ds)

    @torch.no_grad()
    def update_center(self, teacher_output):
        o = _ops()
        rows = teacher_output.shape[0]

        def apply(cs, w):
            self._center_version += 1
            o.center_ema(self.center, cs, self.center_momentum, rows * w)
        self._reduce_and_apply(o.colsum(teacher_output), apply)


class DDINOLoss(_DeferredCenter, nn.Module):
    wants_col_sums = True  # update_center below consumes them

    def __init__(self, out_dim, ncrops, warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs, nepochs,
                 student_temp=0.1, center_momentum=0.9):
        super().__init__()
        self.student_temp, self.center_momentum, self.ncrops = student_temp, center_momentum, ncrops
        self.register_buffer("center", torch.zeros(1, out_dim))
        self.register_buffer("center_grid", torch.zeros(1, out_dim))
        self.teacher_temp_schedule = _teacher_temp_schedule(warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs, nepochs)
        self.assume_unit_grad = False
        self._tables = {}

    def _static(self, B, s_npatch, Tt, device):
        """index tables that depend only on (B, crop layout): crop-major <-> image-major permutations, the cls
        teacher rows, and the per-row loss weights 0.5/(n_terms*B) and 0.5/(n_terms*B*Ts) (Appendix A5)."""
        key = (B, tuple(s_npatch), Tt, str(device))
        t = self._tables.get(key)
        if t is not None:
            return t
        nc = self.ncrops
        n_terms = 2 * nc - 2
        sizes = [s_npatch[0]] * 2 + [s_npatch[1]] * (nc - 2)  # main_esvit.py:710
        S = int(sum(sizes))
        soff = np.concatenate(([0], np.cumsum(sizes)))
        crop_id = np.concatenate([np.full(sz, v, dtype=np.int32) for v, sz in enumerate(sizes)])
        cm_row = np.empty((B, S), dtype=np.int32)
        w_reg = np.empty((B * S,), dtype=np.float32)
        for v, sz in enumerate(sizes):
            rows = B * soff[v] + np.arange(B)[:, None] * sz + np.arange(sz)[None, :]   # [B, sz]
            cm_row[:, soff[v]:soff[v + 1]] = rows
            w_reg[rows.reshape(-1)] = 0.5 / (n_terms * B * sz)
        t_perm = (np.arange(2)[None, :, None] * B * Tt + np.arange(B)[:, None, None] * Tt + np.arange(Tt)[None, None, :]).reshape(-1)
        tm_cls = np.full((nc * B, 2), -1, dtype=np.int32)
        for v in range(nc):
            for iq in range(2):
                if v != iq:
                    tm_cls[v * B:(v + 1) * B, iq] = iq * B + np.arange(B)
        w_cls = np.full((nc * B,), 0.5 / (n_terms * B), dtype=np.float32)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        t = dict(S=S, crop_id=dev(crop_id), cm_row=dev(cm_row.reshape(-1)), t_perm=dev(t_perm.astype(np.int32)), tm_cls=dev(tm_cls),
                 w_cls=dev(w_cls), w_reg=dev(w_reg))
        self._tables[key] = t
        return t

    def forward(self, student_output, teacher_output, epoch, targets_mixup=None):
        o = _ops()
        self.synchronize()
        s_cls, s_reg, s_fea, s_np = student_output
        t_cls, t_reg, t_fea, t_np = teacher_output
        inv_tt = 1.0 / float(self.teacher_temp_schedule[epoch])
        inv_st = 1.0 / self.student_temp
        st_tc, st_tg = _taken_stats(t_cls, self._token("t0", inv_tt)), _taken_stats(t_reg, self._token("t1", inv_tt))
        st_sc, st_sg = _taken_stats(s_cls, self._token("s0", inv_st)), _taken_stats(s_reg, self._token("s1", inv_st))
        cs_tc, cs_tg = _taken_col_sums(t_cls, self._token("t0", inv_tt)), _taken_col_sums(t_reg, self._token("t1", inv_tt))
        s_cls_c, s_reg_c = s_cls.contiguous(), s_reg.contiguous()
        t_cls, t_reg = t_cls.detach().contiguous(), t_reg.detach().contiguous()
        Tt = int(t_np[0])
        B = t_reg.shape[0] // (2 * Tt)
        tb = self._static(B, [int(n) for n in s_np], Tt, s_cls.device)
        S = tb["S"]
        # region matching on fp32 backbone features (main_esvit.py:735-736)
        sf = o.gather_cast(s_fea.detach().float().contiguous(), B * S, rowmap=tb["cm_row"], tokens=1, dtype=torch.float32)
        tf = o.gather_cast(t_fea.detach().float().contiguous(), B * 2 * Tt, rowmap=tb["t_perm"], tokens=1, dtype=torch.float32)
        sfn, _ = o.l2norm_fwd(sf)
        tfn, _ = o.l2norm_fwd(tf)
        D = sfn.shape[1]
        ld = -(-2 * Tt // 8) * 8
        sim = o.batched_nt(sfn.view(B, S, D), tfn.view(B, 2 * Tt, D), ld)
        tm_reg = torch.empty((B * S, 2), dtype=torch.int32, device=s_cls.device)
        o.region_match(sim, Tt, tb["crop_id"], tb["cm_row"], tm_reg)
        # teacher statistics, student CE + gradient
        mx_c, lse_c = st_tc if st_tc is not None else o.teacher_row_stats(t_cls, self.center, inv_tt)
        mx_g, lse_g = st_tg if st_tg is not None else o.teacher_row_stats(t_reg, self.center_grid, inv_tt)
        n_cls = s_cls_c.shape[0]
        row_loss = torch.empty((n_cls + s_reg_c.shape[0],), dtype=torch.float32, device=s_cls.device)
        _, ds_cls = o.dino_ce(s_cls_c.detach(), t_cls, self.center, mx_c, lse_c, tb["tm_cls"], tb["w_cls"], inv_st, inv_tt,
                              row_loss=row_loss[:n_cls], s_stats=st_sc)
        _, ds_reg = o.dino_ce(s_reg_c.detach(), t_reg, self.center_grid, mx_g, lse_g, tm_reg, tb["w_reg"], inv_st, inv_tt,
                              row_loss=row_loss[n_cls:], row_order=tb["cm_row"], s_stats=st_sg)  # image-major work order: teacher rows stay cached
        loss = o.sum_f32(row_loss)
        self.update_center(t_cls, t_reg, col_sums=(cs_tc, cs_tg))
        return _LossFn.apply(loss, self.assume_unit_grad, s_cls, s_reg, ds_cls, ds_reg)

    @torch.no_grad()
    def update_center(self, teacher_output, teacher_grid_output, col_sums=(None, None)):
        """main_esvit.py:752-770; the two (1,K) partial sums travel in ONE all-reduce.  col_sums: the batch sums the heads' last-layer
        GEMMs already produced (a 1.6 GB read of the region logits less per step at B = 128), else they are computed here."""
        o = _ops()
        K = self.center.shape[1]
        buf = torch.empty((2, K), dtype=torch.float32, device=self.center.device)
        for row, (t, cs) in enumerate(zip((teacher_output, teacher_grid_output), col_sums)):
            if cs is not None and cs.numel() == K:
                buf[row].copy_(cs.view(-1))
            else:
                o.colsum(t, out=buf[row])
        r_cls, r_reg = teacher_output.shape[0], teacher_grid_output.shape[0]

        def apply(b, w):
            self._center_version += 1
            o.center_ema(self.center, b[0], self.center_momentum, r_cls * w)
            o.center_ema(self.center_grid, b[1], self.center_momentum, r_reg * w)
        self._reduce_and_apply(buf, apply)
