"""DINOHead behind the reference constructor (vision_transformer.py:384-418): parameters are registered under
the reference's names -- ``mlp.{0,2,4}.{weight,bias}`` (``mlp.{0,3,6}`` + BatchNorm1d ``mlp.{1,4}`` with use_bn) and the legacy weight-norm pair
``last_layer.weight_g`` [K,1] / ``last_layer.weight_v`` [K,256] -- and the forward is one fused autograd node."""
import torch
import torch.nn as nn

from . import functional as Fn


class _WeightNormLinear(nn.Module):
    """parameter holder with the state_dict layout of ``nn.utils.weight_norm(nn.Linear(in, out, bias=False))``"""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        w = torch.empty(out_features, in_features)
        nn.init.kaiming_uniform_(w, a=5 ** 0.5)  # nn.Linear default init, as in the reference before weight_norm
        self.weight_g = nn.Parameter(w.norm(dim=1, keepdim=True))
        self.weight_v = nn.Parameter(w)


class DINOHead(nn.Module):
    def __init__(self, in_dim, out_dim, use_bn=False, norm_last_layer=True, nlayers=3, hidden_dim=2048, bottleneck_dim=256):
        super().__init__()
        nlayers = max(nlayers, 1)
        self.nlayers = nlayers
        self.use_bn = bool(use_bn)
        self.sync_bn_group = None  # process group of the batch statistics (None = default group; False = this rank only)
        # (inv_temp, centre | None, token) set by the loss for one step (loss.arm_logit_stats): the last-layer GEMM then also emits
        # the softmax row statistics the loss needs, handed over as the attribute `esvit_row_stats` of the returned logits
        self.logit_stats = None
        if nlayers == 1:  # vision_transformer.py:388-389 (no BatchNorm either way)
            self.mlp = nn.Linear(in_dim, bottleneck_dim)
        elif use_bn:  # --use_bn_in_head (vision_transformer.py:391-402): Linear, BatchNorm1d, GELU per hidden layer (nlayers = 3: mlp.{0,3,6}, mlp.{1,4})
            layers = [nn.Linear(in_dim, hidden_dim), nn.BatchNorm1d(hidden_dim), nn.GELU()]
            for _ in range(nlayers - 2):
                layers += [nn.Linear(hidden_dim, hidden_dim), nn.BatchNorm1d(hidden_dim), nn.GELU()]
            layers.append(nn.Linear(hidden_dim, bottleneck_dim))
            self.mlp = nn.Sequential(*layers)
        else:               # Linear + GELU (nlayers - 1 times), then the Linear into the bottleneck (vision_transformer.py:391-402)
            layers = [nn.Linear(in_dim, hidden_dim), nn.GELU()]
            for _ in range(nlayers - 2):
                layers += [nn.Linear(hidden_dim, hidden_dim), nn.GELU()]
            layers.append(nn.Linear(hidden_dim, bottleneck_dim))
            self.mlp = nn.Sequential(*layers)
        for m in (self.mlp.modules() if isinstance(self.mlp, nn.Sequential) else [self.mlp]):
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02, a=-2.0, b=2.0)
                nn.init.constant_(m.bias, 0)
        self.last_layer = _WeightNormLinear(bottleneck_dim, out_dim)
        self.last_layer.weight_g.data.fill_(1)
        if norm_last_layer:
            self.last_layer.weight_g.requires_grad = False

    def _bn_state(self, bn):
        st = {"group": self.sync_bn_group}
        if self.training:
            if bn.track_running_stats:
                st.update(running_mean=bn.running_mean, running_var=bn.running_var, num_batches_tracked=bn.num_batches_tracked)
        else:  # nn.BatchNorm1d.eval(): the running statistics
            st.update(eval=True, eval_mean=bn.running_mean, eval_var=bn.running_var)
        return st

    def forward(self, x):
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        if self.use_bn and self.nlayers not in (1, 3):
            mods = list(self.mlp)
            hidden = [(mods[i].weight, mods[i].bias, mods[i + 1].weight, mods[i + 1].bias) for i in range(0, len(mods) - 1, 3)]
            return self._finish(Fn.dino_head_bn_n(x2, [self._bn_state(mods[i + 1]) for i in range(0, len(mods) - 1, 3)], hidden,
                                                  (mods[-1].weight, mods[-1].bias), self.last_layer.weight_v, self.last_layer.weight_g,
                                                  self._stats_request()), lead)
        if self.use_bn and self.nlayers == 3:
            m = self.mlp
            prm = [m[0].weight, m[0].bias, m[1].weight, m[1].bias, m[3].weight, m[3].bias, m[4].weight, m[4].bias, m[6].weight, m[6].bias,
                   self.last_layer.weight_v, self.last_layer.weight_g]
            return self._finish(Fn.dino_head_bn(x2, self._bn_state(m[1]), self._bn_state(m[4]), prm, self._stats_request()), lead)
        if self.nlayers != 3:
            lins = [self.mlp] if isinstance(self.mlp, nn.Linear) else [m for m in self.mlp if isinstance(m, nn.Linear)]
            return self._finish(Fn.dino_head_n(x2, [(m.weight, m.bias) for m in lins], self.last_layer.weight_v, self.last_layer.weight_g,
                                               self._stats_request()), lead)
        prm = [self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias, self.mlp[4].weight, self.mlp[4].bias,
               self.last_layer.weight_v, self.last_layer.weight_g]
        return self._finish(Fn.dino_head(x2, prm, self._stats_request()), lead)

    def _stats_request(self):
        if self.logit_stats is None:
            return None
        return self.logit_stats[:2] + (bool(self.logit_stats[3]) if len(self.logit_stats) > 3 else False,)

    def _finish(self, out, lead):
        y, mx, lse = out
        y = y.view(*lead, y.shape[-1])
        if mx is not None:
            # (token, row max, row log-sum-exp, batch sums per column or None: the last one does not depend on centre / temperature)
            cs = getattr(mx, "esvit_col_sums", None)
            y.esvit_row_stats = (self.logit_stats[2], mx, lse, None if cs is None else (int(mx.shape[0]), cs))
        return y
