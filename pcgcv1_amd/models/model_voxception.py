"""Operator surface of the reference's models/model_voxception.py on MI355X.

Same class names, zero-argument constructors and call signatures as the
reference (AnalysisTransform 71-144, SynthesisTransform 147-214, HyperEncoder
217-252, HyperDecoder 255-308): `net(x)` takes / returns NDHWC float32 tensors
(torch, on the GPU; numpy is accepted and uploaded), HyperDecoder returns
`(loc, |scale|)`.  The convolution stacks run in libpcgc_hip.so
(`pcgc_net_forward`, include/pcgc.h) — batched, not one cube per call.

Weights: the reference restores lazily from a tf.train.Checkpoint
(transform.py:107-112).  Here `load_weights(dict)` binds numpy arrays keyed by
the same Keras attribute paths ("vrn1_1/conv1_1/kernel", "down_1/kernel", ...),
in TensorFlow layouts; see pcgcv1_amd/checkpoint.py.
"""
import ctypes

import numpy as np
import torch

from .. import _lib
from . import spec

_KIND = {"analysis_transform": 0, "synthesis_transform": 1, "hyper_encoder": 2, "hyper_decoder": 3}


class _Net(object):
    net_name = None

    def __init__(self):
        self._handle = None
        self._ws = {}            # scratch per stream: the same net may run on several streams at once
        self._params = None
        self.algo = 0

    # -- weights ---------------------------------------------------------
    def load_weights(self, weights, prefix=None):
        """weights: dict name -> numpy array (TF layouts). Keys may carry the checkpoint prefix
        ("analysis_transform/conv_in/kernel") or not ("conv_in/kernel")."""
        dev = _lib.require_gpu()
        prefix = (prefix if prefix is not None else self.net_name) + "/"
        tensors = []
        for l in spec.NETS[self.net_name]():
            for suffix, shape in (("kernel", spec.kernel_shape(l)),) + ((("bias", (l.cout,)),) if l.bias else ()):
                key = "%s/%s" % (l.name, suffix)
                arr = weights.get(prefix + key, weights.get(key))
                if arr is None:
                    raise KeyError("missing weight %s%s" % (prefix, key))
                arr = np.ascontiguousarray(arr, np.float32)
                if arr.shape != tuple(shape):
                    raise ValueError("%s: expected shape %r, got %r" % (key, tuple(shape), arr.shape))
                tensors.append(torch.from_numpy(arr).to(dev))
        lib = _lib.hip()
        assert len(tensors) == lib.pcgc_net_param_count(_KIND[self.net_name])
        ptrs = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        handle = ctypes.c_void_p()
        _lib.check(lib.pcgc_net_create(_KIND[self.net_name], ptrs, len(tensors), _lib.stream(), ctypes.byref(handle)),
                   "pcgc_net_create")
        torch.cuda.current_stream().synchronize()       # the library has copied + repacked; tensors may go
        self.close()
        self._handle = handle
        return self

    def close(self):
        if self._handle is not None:
            _lib.hip().pcgc_net_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_algo(self, algo):
        """0 = MFMA kernels where available (default), 1 = direct kernels only."""
        self.algo = int(algo)
        if self._handle is not None:
            _lib.check(_lib.hip().pcgc_net_set_algo(self._handle, self.algo))
        return self

    def set_skip_counter(self, counter):
        """Test aid (AnalysisTransform at 64^3): an int32 device tensor of one element that receives +1 per wave tile the
        empty-space skipping copies instead of computing; None switches it off."""
        self._skip_counter = counter                          # keeps the tensor alive while the library holds its address
        _lib.check(_lib.hip().pcgc_net_set_skip_counter(self._handle, _lib.dptr(counter) if counter is not None else None))
        return self

    def set_profiling(self, on):
        _lib.check(_lib.hip().pcgc_net_set_profiling(self._handle, int(bool(on))))
        return self

    def profile_report(self):
        """Drain per-launch timings: list of dicts(layer, name, kernel, cin, cout, k, mode, B, Din, ms)."""
        lib = _lib.hip()
        need = ctypes.c_size_t(0)
        # first call with a generous buffer; the records are consumed by the call that copies them
        buf = ctypes.create_string_buffer(1 << 20)
        _lib.check(lib.pcgc_net_profile_report(self._handle, buf, len(buf), ctypes.byref(need)))
        rows = []
        for line in buf.value.decode().splitlines():
            f = line.split()
            rows.append(dict(layer=int(f[0]), name=f[1], kernel=f[2], cin=int(f[3]), cout=int(f[4]), k=int(f[5]),
                             mode=int(f[6]), B=int(f[7]), Din=int(f[8]), ms=float(f[9])))
        return rows

    # -- forward ---------------------------------------------------------
    def _forward(self, x, n_out, lower_bound=0.0, out=None):
        if self._handle is None:
            raise _lib.PcgcError("%s: no weights bound (call load_weights / checkpoint.restore)" % type(self).__name__)
        dev = _lib.require_gpu()
        if not torch.is_tensor(x):
            x = torch.from_numpy(np.ascontiguousarray(x, np.float32))
        x = x.to(device=dev, dtype=torch.float32).contiguous()
        assert x.dim() == 5 and x.shape[1] == x.shape[2] == x.shape[3], "expected [B, D, D, D, C]"
        B, D = int(x.shape[0]), int(x.shape[1])
        cin, cout, dout = self._geometry(D)
        assert x.shape[4] == cin, "expected %d input channels, got %d" % (cin, x.shape[4])
        lib = _lib.hip()
        _lib.check(lib.pcgc_net_set_algo(self._handle, self.algo))
        need = lib.pcgc_net_workspace_bytes(self._handle, B, D)
        skey = int(torch.cuda.current_stream().cuda_stream)
        ws = self._ws.get(skey)
        if ws is None or ws.numel() < need:
            ws = self._ws[skey] = torch.empty(int(need), dtype=torch.uint8, device=dev)
        if out is not None:          # the caller's buffer (a slice of a larger batch): no copy afterwards
            assert n_out == 1 and out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (B, dout, dout, dout, cout)
            outs = [out]
        else:
            outs = [torch.empty((B, dout, dout, dout, cout), dtype=torch.float32, device=dev) for _ in range(n_out)]
        _lib.check(lib.pcgc_net_forward(self._handle, _lib.dptr(x), _lib.dptr(outs[0]),
                                        _lib.dptr(outs[1]) if n_out > 1 else None, B, D, float(lower_bound),
                                        _lib.dptr(ws), ws.numel(), _lib.stream()), "pcgc_net_forward")
        return outs

    def __call__(self, x, out=None):
        return self._forward(x, 1, out=out)[0]


class AnalysisTransform(_Net):
    """model_voxception.py:71-144. x [B,N,N,N,1] -> y [B,N/4,N/4,N/4,16]."""
    net_name = "analysis_transform"

    def _geometry(self, D):
        return 1, 16, D // 4


class SynthesisTransform(_Net):
    """model_voxception.py:147-214. y [B,n,n,n,16] -> occupancy logits [B,4n,4n,4n,1]."""
    net_name = "synthesis_transform"

    def _geometry(self, D):
        return 16, 1, D * 4


class HyperEncoder(_Net):
    """model_voxception.py:217-252. y [B,n,n,n,16] -> z [B,n/2,n/2,n/2,8]."""
    net_name = "hyper_encoder"

    def _geometry(self, D):
        return 16, 8, D // 2


class HyperDecoder(_Net):
    """model_voxception.py:255-308. z [B,m,m,m,8] -> (loc, |scale|), each [B,2m,2m,2m,16].
    `lower_bound` folds the caller's tf.maximum(scale, lower_bound) (transform.py:145-146, 232-233)
    into the epilogue; the default 0 reproduces the bare operator."""
    net_name = "hyper_decoder"

    def _geometry(self, D):
        return 8, 16, D * 2

    def __call__(self, x, lower_bound=0.0):
        loc, scale = self._forward(x, 2, lower_bound)
        return loc, scale


def conv3d(x, kernel, bias=None, stride=1, transposed=False, relu=False, algo=0):
    """One Keras Conv3D/Conv3DTranspose(padding='same') through pcgc_conv3d_fwd (layer-level entry used by
    the parity tests).  x torch cuda [B,D,D,D,Cin]; kernel/bias torch cuda in TF layouts."""
    dev = _lib.require_gpu()
    x = x.to(dev, torch.float32).contiguous()
    kernel = kernel.to(dev, torch.float32).contiguous()
    bias = None if bias is None else bias.to(dev, torch.float32).contiguous()
    B, D, cin = int(x.shape[0]), int(x.shape[1]), int(x.shape[4])
    k = int(kernel.shape[0])
    cout = int(kernel.shape[3] if transposed else kernel.shape[4])
    dout = 2 * D if transposed else D // stride
    y = torch.empty((B, dout, dout, dout, cout), dtype=torch.float32, device=dev)
    _lib.check(_lib.hip().pcgc_conv3d_fwd(_lib.dptr(x), _lib.dptr(kernel), _lib.dptr(bias), _lib.dptr(y), B, D, cin, cout,
                                          k, 2 if transposed else stride, int(transposed), int(relu), algo, _lib.stream()),
               "pcgc_conv3d_fwd")
    return y


def vrn_block(x, params):
    """One _VoxceptionResNet block (model_voxception.py:56-68) through pcgc_vrn_fwd.  x torch cuda [B,D,D,D,C];
    params = the block's ten tensors {conv1_1, conv1_2, conv2_1, conv2_2, conv2_3} x {kernel, bias} in TF layouts."""
    import ctypes
    dev = _lib.require_gpu()
    x = x.to(dev, torch.float32).contiguous()
    ps = [p.to(dev, torch.float32).contiguous() for p in params]
    assert len(ps) == 10
    B, D, C = int(x.shape[0]), int(x.shape[1]), int(x.shape[4])
    arr = (ctypes.c_void_p * 10)(*[p.data_ptr() for p in ps])
    nbytes = int(_lib.hip().pcgc_vrn_workspace_bytes(B, D, C))
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    _lib.check(_lib.hip().pcgc_vrn_fwd(_lib.dptr(x), arr, _lib.dptr(y), B, D, C, _lib.dptr(ws), nbytes, _lib.stream()),
               "pcgc_vrn_fwd")
    return y

