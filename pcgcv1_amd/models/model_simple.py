"""Operator surface of the reference's models/model_simple.py (the ablation autoencoder of the factorized mode:
`--modelname=models.model_simple`, test.py:35, train_factorized.py:66) on MI355X.

AnalysisTransform (model_simple.py:12-49): Conv3D 9^3 s2 (1 -> 32, ReLU), 5^3 s2 (32 -> 32, ReLU), 5^3 s2 (32 -> 32,
linear, no bias): x [B,N,N,N,1] -> y [B,N/8,N/8,N/8,32].  SynthesisTransform (52-95): Conv3DTranspose 5^3 s2 (ReLU),
5^3 s2 (ReLU), 9^3 s2 (32 -> 1, linear).  Same zero-argument constructors and call signatures as the reference;
every layer is one pcgc_conv3d_fwd call (the shape-generic direct kernel: these kernel sizes have no tile kernel —
the model is 1.35 GMAC per cube per transform against 5.2 for model_voxception and is not on the benchmark path).
Checkpoint keys: "analysis_transform/conv_1/kernel", ..., "synthesis_transform/deconv_3/bias" (TF layouts).
"""
import numpy as np
import torch

from .. import _lib
from . import spec
from .model_voxception import conv3d


class _Seq(object):
    net_name = None

    def __init__(self):
        self._layers = None

    def load_weights(self, weights, prefix=None):
        dev = _lib.require_gpu()
        prefix = (prefix if prefix is not None else self.net_name) + "/"
        bound = []
        for l in spec.SIMPLE_NETS[self.net_name]():
            k = weights.get(prefix + l.name + "/kernel", weights.get(l.name + "/kernel"))
            if k is None:
                raise KeyError("missing weight %s%s/kernel" % (prefix, l.name))
            k = np.ascontiguousarray(k, np.float32)
            if k.shape != tuple(spec.kernel_shape(l)):
                raise ValueError("%s/kernel: expected shape %r, got %r" % (l.name, tuple(spec.kernel_shape(l)), k.shape))
            b = None
            if l.bias:
                b = weights.get(prefix + l.name + "/bias", weights.get(l.name + "/bias"))
                if b is None:
                    raise KeyError("missing weight %s%s/bias" % (prefix, l.name))
                b = torch.from_numpy(np.ascontiguousarray(b, np.float32)).to(dev)
            bound.append((l, torch.from_numpy(k).to(dev), b))
        self._layers = bound
        return self

    def __call__(self, x):
        if self._layers is None:
            raise _lib.PcgcError("%s: no weights bound (call load_weights)" % type(self).__name__)
        dev = _lib.require_gpu()
        if not torch.is_tensor(x):
            x = torch.from_numpy(np.ascontiguousarray(x, np.float32))
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        for l, k, b in self._layers:
            x = conv3d(x, k, b, stride=l.stride, transposed=(l.kind == "tconv"), relu=l.relu)
        return x


class AnalysisTransform(_Seq):
    """model_simple.py:12-49."""
    net_name = "analysis_transform"


class SynthesisTransform(_Seq):
    """model_simple.py:52-95."""
    net_name = "synthesis_transform"
