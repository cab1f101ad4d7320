"""Fused student update + teacher EMA (reference: utils.py:106-123, torch.optim.AdamW as driven by
main_esvit.py:506-510/574, EMA loop main_esvit.py:587-590) -- two multi-tensor HIP launches, no
host synchronisation.

``FusedClipAdamWEMA`` keeps the optimizer state in the same layout as ``torch.optim.AdamW``
(``state_dict()`` / ``load_state_dict()`` produce / accept the reference checkpoint's
``optimizer`` entry: per-parameter ``step``, ``exp_avg``, ``exp_avg_sq``; two param groups from
``get_params_groups``), so checkpoints stay interchangeable with the unmodified caller.
"""
import numpy as np
import torch

from . import ops
from . import params as P

TFIELDS = 12


def get_params_groups(model):
    """utils.py:672-683: biases and 1-D parameters are not regularised; frozen parameters are skipped."""
    regularized, not_regularized = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if name.endswith(".bias") or len(param.shape) == 1:
            not_regularized.append(param)
        else:
            regularized.append(param)
    return [{"params": regularized}, {"params": not_regularized, "weight_decay": 0.0}]


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0):
    """utils.py:161-173."""
    warmup_iters = warmup_epochs * niter_per_ep
    warm = np.linspace(start_warmup_value, base_value, warmup_iters) if warmup_epochs > 0 else np.array([])
    n = epochs * niter_per_ep - warmup_iters
    it = np.arange(n)
    sched = final_value + 0.5 * (base_value - final_value) * (1 + np.cos(np.pi * it / n))
    sched = np.concatenate((warm, sched))
    assert len(sched) == epochs * niter_per_ep
    return sched


class FusedClipAdamWEMA:
    """clip -> optimizer rule -> EMA for (student, teacher) in one pass.

    student / teacher: modules with positionally matching ``.parameters()`` (main_esvit.py:589).
    Only parameters with ``requires_grad`` are optimised; *every* parameter is EMA-ed, exactly as the
    reference loop does.  rule: "adamw" (torch.optim.AdamW; the class name dates from when it was the only one),
    "sgd" (torch.optim.SGD(lr=0, momentum=0.9), main_esvit.py:413) or "lars" (utils.LARS, utils.py:519-557);
    the latter two keep their single state tensor (momentum_buffer / mu) where AdamW keeps exp_avg.
    """

    STATE_KEY = {"adamw": "exp_avg", "sgd": "momentum_buffer", "lars": "mu"}

    def __init__(self, student, teacher, betas=(0.9, 0.999), eps=1e-8, rule="adamw", momentum=0.9, eta=0.001):
        assert rule in self.STATE_KEY, rule
        self.rule, self.momentum, self.eta = rule, float(momentum), float(eta)
        self.betas, self.eps = betas, eps
        self.params = list(student.parameters())
        self.names = [n for n, _ in student.named_parameters()]
        self.teacher_params = list(teacher.parameters()) if teacher is not None else [None] * len(self.params)
        assert len(self.teacher_params) == len(self.params)
        groups = get_params_groups(student)
        self.param_groups = [dict(params=groups[0]["params"], lr=0.0, weight_decay=0.0, betas=betas, eps=eps),
                             dict(params=groups[1]["params"], lr=0.0, weight_decay=0.0, betas=betas, eps=eps)]
        gid = {}
        for gi, g in enumerate(groups):
            for p in g["params"]:
                gid[id(p)] = gi
        self.group_of = [gid.get(id(p), 1) for p in self.params]
        self.trainable = [p.requires_grad for p in self.params]
        dev = self.params[0].device
        self.device = dev
        self.exp_avg = [torch.zeros_like(p) if t else None for p, t in zip(self.params, self.trainable)]
        self.exp_avg_sq = [torch.zeros_like(p) if (t and rule == "adamw") else None for p, t in zip(self.params, self.trainable)]
        self.steps = [0] * len(self.params)
        chunk = ops.update_chunk_elems()
        chunks = []
        for ti, p in enumerate(self.params):
            for ci in range(-(-p.numel() // chunk)):
                chunks.append((ti, ci))
        self.nchunks = len(chunks)
        self.chunks = torch.tensor(chunks, dtype=torch.int32).to(dev)
        self.sqnorms = torch.zeros(len(self.params) * (3 if rule == "lars" else 1), dtype=torch.float32, device=dev)
        # updates the kernel refused (a non-finite gradient statistic): counted on the device, collected by take_skipped()
        self.skipped = torch.zeros(1, dtype=torch.int32, device=dev)
        self._skipped_seen = 0
        self._stepped_log = []  # per update launch since the last take_skipped(): the parameter indices whose step count it advanced
        # ring of pinned staging tables: a buffer is rewritten only after the async H2D copy that read it completed
        self._ring = [torch.zeros((len(self.params), TFIELDS), dtype=torch.int64).pin_memory() for _ in range(4)]
        self._ring_np = [t.numpy() for t in self._ring]
        for tab in self._ring_np:  # static columns
            for i, p in enumerate(self.params):
                tp = self.teacher_params[i]
                tab[i, 0] = p.data_ptr()
                tab[i, 2] = self.exp_avg[i].data_ptr() if self.trainable[i] else 0
                tab[i, 3] = self.exp_avg_sq[i].data_ptr() if self.exp_avg_sq[i] is not None else 0
                tab[i, 4] = tp.data_ptr() if tp is not None else 0
                tab[i, 5] = p.numel()
                tab[i, 6] = self.group_of[i]
        for tp in self.teacher_params:
            if tp is not None and not tp.requires_grad:
                P.manage(tp)  # the teacher is written by this updater only: its bf16 copies are refreshed in the same kernel
        self.bound = None  # a torch.optim.AdamW whose state mirrors this updater (bind_torch_optimizer)
        self._bc_cache = {}
        self._ring_ev = [None] * len(self._ring)
        self._ring_pos = 0
        self._table_dev = torch.zeros((len(self.params), TFIELDS), dtype=torch.int64, device=dev)

    @staticmethod
    def _bits(x):
        return int(np.float32(x).view(np.uint32))

    def step(self, lr, weight_decay, ema_momentum, clip_grad=3.0, skip_last_layer=False):
        """One update.  lr / weight_decay / ema_momentum are the per-iteration schedule values
        (main_esvit.py:506-510, 588); skip_last_layer mirrors cancel_gradients_last_layer (utils.py:118-123)."""
        b1, b2 = self.betas
        if self.bound is not None:
            self._adopt_bound_state()
        slot = self._ring_pos
        self._ring_pos = (slot + 1) % len(self._ring)
        if self._ring_ev[slot] is not None:
            self._ring_ev[slot].synchronize()
        tab = self._ring_np[slot]
        gptr, flags, bcs = [0] * len(self.params), [0] * len(self.params), [0] * len(self.params)
        stepped = []
        self._stepped_log.append(stepped)
        if len(self._stepped_log) > 4096:  # (nobody collects: keep the log bounded)
            del self._stepped_log[:2048]
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None or not self.trainable[i] or (skip_last_layer and "last_layer" in self.names[i]):
                continue
            assert g.is_contiguous() and g.dtype == torch.float32
            self.steps[i] += 1
            stepped.append(i)
            t = self.steps[i]
            bc = self._bc_cache.get(t)
            if bc is None:
                bc = self._bits(1.0 - b1 ** t) | (self._bits(1.0 - b2 ** t) << 32)
                if bc >= 1 << 63:
                    bc -= 1 << 64
                if len(self._bc_cache) > 64:
                    self._bc_cache.clear()
                self._bc_cache[t] = bc
            gptr[i], flags[i], bcs[i] = g.data_ptr(), 1, bc
        tab[:, 1], tab[:, 7], tab[:, 8] = gptr, flags, bcs
        # activation-dtype copies the forward GEMMs read (esvit_amd.params): refreshed by the update kernel itself
        fresh = []
        sb, tb = [0] * len(self.params), [0] * len(self.params)
        for i, p in enumerate(self.params):
            sb[i] = P.cast_buffer_ptr(p, fresh)
            tp = self.teacher_params[i]
            tb[i] = P.cast_buffer_ptr(tp, fresh) if tp is not None else 0
        tab[:, 10], tab[:, 11] = sb, tb
        tab = self._ring[slot]
        self._table_dev.copy_(tab, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._ring_ev[slot] = ev
        n = len(self.params)
        ops.grad_sqnorm(self._table_dev, n, self.chunks, self.nchunks, self.sqnorms, stats=3 if self.rule == "lars" else 1)
        if self.rule == "adamw":
            rule, h1, h2 = ops.RULE_ADAMW, b1, b2
        else:
            rule, h1, h2 = (ops.RULE_SGD if self.rule == "sgd" else ops.RULE_LARS), self.momentum, self.eta
        ops.fused_clip_update_ema(rule, self._table_dev, n, self.chunks, self.nchunks, self.sqnorms, float(clip_grad or 0.0), float(lr),
                                  float(weight_decay), h1, h2, self.eps, float(ema_momentum), skipped=self.skipped)
        P.invalidate()          # parameters changed behind autograd's back ...
        P.mark_fresh(fresh)     # ... but these cached casts were rewritten by the kernel
        for g in self.param_groups:
            g["lr"] = float(lr)
        self.param_groups[0]["weight_decay"] = float(weight_decay)
        if self.bound is not None:
            self._publish_bound_state(lr, weight_decay)

    def _settle_skipped(self):
        """look at the kernel's refused-update counter (SYNCHRONISES with the device) and take the refused launches back out of the
        step counts (the bias correction of AdamW, the 'step' entries of the optimizer state): a refused update advanced nothing.
        Which launches were refused is not recorded, only how many; the launches between two looks step the same parameters (a look
        happens at every epoch end, where freeze_last_layer may change the set), so the most recent ones are rolled back.  A bound
        torch optimizer's published 'step' entries are corrected at once, so a checkpoint written right after a look is exact."""
        total = int(self.skipped.item())
        new = total - self._skipped_seen
        self._skipped_seen = total
        if new > 0:
            for stepped in self._stepped_log[len(self._stepped_log) - new:]:
                for i in stepped:
                    self.steps[i] -= 1
            if self.bound is not None:
                for i, p in enumerate(self.params):
                    e = self.bound.state.get(p)
                    if e and "step" in e:
                        e["step"].fill_(float(self.steps[i]))
            self._unreported = getattr(self, "_unreported", 0) + new
        self._stepped_log = []

    def take_skipped(self):
        """How many update launches since the last call were refused by the kernel's non-finite guard (synchronises; see
        _settle_skipped for what happens to their step counts)."""
        self._settle_skipped()
        n = getattr(self, "_unreported", 0)
        self._unreported = 0
        return n

    # ---- a torch.optim.AdamW bound to this updater (integration level L2 with the unmodified train_esvit) -------------
    def _refresh_static_rows(self, i):
        for tab in self._ring_np:
            tab[i, 2] = self.exp_avg[i].data_ptr()
            tab[i, 3] = self.exp_avg_sq[i].data_ptr() if self.exp_avg_sq[i] is not None else 0

    def _adopt_bound_state(self):
        """optimizer.load_state_dict() (resume, utils.py:126-158) replaces the state tensors: adopt them"""
        st = self.bound.state
        for i, p in enumerate(self.params):
            e = st.get(p)
            if not e or not self.trainable[i]:
                continue
            k = self.STATE_KEY[self.rule]
            if e.get(k) is None:  # (torch.optim.SGD stores momentum_buffer = None before its first step)
                continue
            if e[k] is not self.exp_avg[i] or (self.rule == "adamw" and e["exp_avg_sq"] is not self.exp_avg_sq[i]):
                assert e[k].dtype == torch.float32 and e[k].is_contiguous() and e[k].device == p.device
                self.exp_avg[i] = e[k]
                if self.rule == "adamw":
                    self.exp_avg_sq[i] = e["exp_avg_sq"]
                self._refresh_static_rows(i)
            self.steps[i] = int(float(e["step"])) if "step" in e else max(self.steps[i], 1)

    def _publish_bound_state(self, lr, weight_decay):
        st = self.bound.state
        for i, p in enumerate(self.params):
            if self.steps[i] <= 0:
                continue
            e = st.get(p)
            if self.rule != "adamw":
                if not e or e.get(self.STATE_KEY[self.rule]) is not self.exp_avg[i]:
                    st[p] = {self.STATE_KEY[self.rule]: self.exp_avg[i]}
            elif not e:
                st[p] = {"step": torch.tensor(float(self.steps[i])), "exp_avg": self.exp_avg[i], "exp_avg_sq": self.exp_avg_sq[i]}
            else:
                e["step"].fill_(float(self.steps[i]))
        for g in self.bound.param_groups:
            g["lr"] = float(lr)
        self.bound.param_groups[0]["weight_decay"] = float(weight_decay)

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    # ---- torch.optim.AdamW-compatible checkpoint layout -----------------------------------------
    def _ordered(self):
        order = []
        for g in self.param_groups:
            order.extend(g["params"])
        pos = {id(p): i for i, p in enumerate(self.params)}
        return [pos[id(p)] for p in order]

    def state_dict(self):
        self._settle_skipped()  # (a checkpoint never carries the step of a refused update)
        order = self._ordered()
        state = {}
        for k, i in enumerate(order):
            if self.steps[i] > 0:
                if self.rule == "adamw":
                    state[k] = {"step": torch.tensor(float(self.steps[i])), "exp_avg": self.exp_avg[i], "exp_avg_sq": self.exp_avg_sq[i]}
                else:
                    state[k] = {self.STATE_KEY[self.rule]: self.exp_avg[i]}
        groups, k = [], 0
        for g in self.param_groups:
            n = len(g["params"])
            if self.rule == "adamw":
                groups.append({"lr": g["lr"], "betas": self.betas, "eps": self.eps, "weight_decay": g["weight_decay"],
                               "amsgrad": False, "params": list(range(k, k + n))})
            else:
                groups.append({"lr": g["lr"], "momentum": self.momentum, "weight_decay": g["weight_decay"], "params": list(range(k, k + n)),
                               **({"eta": self.eta} if self.rule == "lars" else {"dampening": 0, "nesterov": False})})
            k += n
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        order = self._ordered()
        for k, st in sd["state"].items():
            i = order[int(k)]
            self.steps[i] = int(float(st["step"])) if "step" in st else 1
            self.exp_avg[i].copy_(st[self.STATE_KEY[self.rule]])
            if self.rule == "adamw":
                self.exp_avg_sq[i].copy_(st["exp_avg_sq"])
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            g["lr"], g["weight_decay"] = sg["lr"], sg["weight_decay"]


def optimizer_rule(optimizer):
    """-> "adamw" / "sgd" / "lars" for the optimizers train_esvit can build (main_esvit.py:408-415), else None.
    LARS is recognised structurally (the reference's class lives in its utils.py): a momentum optimizer with an ``eta`` default."""
    if isinstance(optimizer, FusedClipAdamWEMA):
        return optimizer.rule
    if isinstance(optimizer, torch.optim.AdamW):
        return "adamw"
    if isinstance(optimizer, torch.optim.SGD):
        d = optimizer.defaults
        if d.get("nesterov") or d.get("dampening", 0) != 0 or d.get("maximize"):
            return None
        return "sgd"
    d = getattr(optimizer, "defaults", {})
    if type(optimizer).__name__ == "LARS" and "eta" in d and "momentum" in d:
        if d.get("weight_decay_filter") is not None or d.get("lars_adaptation_filter") is not None:
            return None  # (utils.LARS.step never consults them; refuse rather than guess)
        return "lars"
    return None


def bind_torch_optimizer(optimizer, student, teacher):
    """Fused updater whose state tensors ARE the `state` of the caller's optimizer (torch.optim.AdamW / torch.optim.SGD /
    utils.LARS built over get_params_groups(student), main_esvit.py:408-415): the fused kernel updates exp_avg / exp_avg_sq
    (momentum_buffer; mu) in place, `step` and the param_groups' lr / weight_decay are mirrored after every update, and
    tensors swapped in by optimizer.load_state_dict() are adopted before the next one.  optimizer.step() itself is never called."""
    rule = optimizer_rule(optimizer)
    assert rule is not None, type(optimizer)
    pg = optimizer.param_groups
    assert len(pg) == 2 and pg[1]["weight_decay"] == 0.0, "expected the two groups of utils.get_params_groups (utils.py:672-683)"
    if rule == "adamw":
        upd = FusedClipAdamWEMA(student, teacher, betas=tuple(pg[0]["betas"]), eps=pg[0]["eps"])
    else:
        upd = FusedClipAdamWEMA(student, teacher, rule=rule, momentum=pg[0]["momentum"], eta=pg[0].get("eta", 0.001))
    ours = [[id(p) for p in g["params"]] for g in upd.param_groups]
    theirs = [[id(p) for p in g["params"]] for g in pg]
    assert ours == theirs, "the optimizer's parameter groups do not match get_params_groups(student)"
    upd.bound = optimizer
    upd._adopt_bound_state()
    return upd
