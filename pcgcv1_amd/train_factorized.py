"""One optimisation step of the reference's train_factorized.py (150-190) on MI355X: the autoencoder
(models.model_voxception or models.model_simple, chosen like `--model`, train_factorized.py:66) with the factorized
prior on its latents.

    trainer = Trainer(weights, model="model_simple", alpha=2.0, beta=3.0, lr=1e-4)
    terms = trainer.step(x)

forward   y = A(x); y~ = y + U(-.5,.5); p = EntropyBottleneck(y~); x~ = S(y~)                train_factorized.py:160-163
loss      alpha * (beta * BCE_empty + BCE_full) + bpp,  bpp = sum(log p) / (-ln2 * num_points)  train_factorized.py:165-170
update    tf.train.AdamOptimizer over analysis + synthesis + estimator variables            train_factorized.py:175-186
Same machinery as train_hyper.Trainer (explicit reverse pass over the layer tables, flat parameter / gradient buffers,
one all_reduce per step under torch.distributed, TF-format checkpoints); only the graph differs.
"""
import numpy as np
import torch

from . import _lib
from .models import spec
from .train_hyper import LN2, Trainer as _HyperTrainer

_TABLES = {
    "model_voxception": {k: spec.NETS[k] for k in ("analysis_transform", "synthesis_transform")},
    "model_simple": spec.SIMPLE_NETS,
}


class Trainer(_HyperTrainer):
    def __init__(self, weights, model="model_voxception", alpha=2.0, beta=3.0, lr=1e-4, group=None):
        name = model if isinstance(model, str) else getattr(model, "__name__", "")
        name = name.split(".")[-1]
        if name not in _TABLES:
            raise ValueError("unknown model %r (model_voxception, model_simple)" % (model,))
        super().__init__(weights, alpha=alpha, beta=beta, gamma=0.0, delta=1.0, lr=lr, group=group, nets=_TABLES[name])

    def forward_backward(self, x, noise_y=None, noise_z=None, grad_scale=1.0, with_iou=False, _before_readback=None):
        lib = _lib.hip()
        x = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x, np.float32))
        x = x.to(self.dev, torch.float32).contiguous()
        self._join_dw_stream()
        self.flat_g.zero_()
        self._held.clear()
        self._prepare()
        y, ca = self._run_net("analysis_transform", x)
        assert int(y.shape[-1]) == self.eb_C, "estimator has %d channels, the latents %d" % (self.eb_C, int(y.shape[-1]))
        ny = (torch.rand_like(y) - 0.5) if noise_y is None else torch.as_tensor(noise_y, dtype=torch.float32).to(self.dev).contiguous()
        eb_params = self.flat_p[self.eb_off:]
        y_t, lik = torch.empty_like(y), torch.empty_like(y)
        _lib.check(lib.pcgc_factorized_likelihood(_lib.dptr(y), _lib.dptr(eb_params), _lib.dptr(ny), _lib.dptr(y_t), _lib.dptr(lik),
                                                  y.numel(), self.eb_C, 1e-9, _lib.stream()))
        x_t, cs = self._run_net("synthesis_transform", y_t)
        sums = torch.empty(4, dtype=torch.float64, device=self.dev)
        ws = torch.empty(int(lib.pcgc_bce_workspace_bytes(x_t.numel())), dtype=torch.uint8, device=self.dev)
        _lib.check(lib.pcgc_bce_sums(_lib.dptr(x_t), _lib.dptr(x), x_t.numel(), _lib.dptr(sums), _lib.dptr(ws), ws.numel(), _lib.stream()))
        logs = torch.empty(1, dtype=torch.float64, device=self.dev)
        ws2 = torch.empty(int(lib.pcgc_sum_log_workspace_bytes()), dtype=torch.uint8, device=self.dev)
        _lib.check(lib.pcgc_sum_log(_lib.dptr(lik), lik.numel(), _lib.dptr(logs), _lib.dptr(ws2), ws2.numel(), _lib.stream()))
        s0, n0, s1, n1 = (float(v) for v in sums.cpu().numpy())
        num_points = n1
        empty, full = s0 / n0, s1 / n1
        bpp = float(logs.cpu().numpy()[0]) / (-LN2 * num_points)
        loss = self.alpha * (self.beta * empty + full) + bpp
        gs = float(grad_scale)
        dx_t = torch.empty_like(x_t)
        _lib.check(lib.pcgc_bce_bwd(_lib.dptr(x_t), _lib.dptr(x), gs * self.alpha * self.beta / n0, gs * self.alpha / n1,
                                    _lib.dptr(dx_t), x_t.numel(), _lib.stream()))
        dy_t = self._run_net_bwd(cs, dx_t)
        dy_l = torch.empty_like(y)
        wsf = torch.empty(int(lib.pcgc_factorized_bwd_workspace_bytes(self.eb_C)), dtype=torch.uint8, device=self.dev)
        _lib.check(lib.pcgc_factorized_likelihood_bwd(_lib.dptr(y_t), _lib.dptr(eb_params), gs / (-LN2 * num_points), 1e-9,
                                                      _lib.dptr(dy_l), _lib.dptr(self.flat_g[self.eb_off:]), y.numel(), self.eb_C,
                                                      _lib.dptr(wsf), wsf.numel(), _lib.stream()))
        self._add(dy_t, dy_l)
        self._run_net_bwd(ca, dy_t, need_dx=False)
        self._finish_weights()
        if _before_readback is not None:             # Trainer.step: the optimiser update (this pass read its sums mid-way already)
            _before_readback(sums)
        terms = dict(loss=loss, bpp=bpp, empty=empty, full=full, num_points=num_points)
        if with_iou:
            terms["IoU"] = self.iou(x_t, x)          # train_factorized.py:196-205
        return terms
