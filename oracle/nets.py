"""ORACLE — TEST INFRASTRUCTURE ONLY (not the product path).

CPU restatement (torch-CPU fp32, plus a slow pure-numpy cross-check) of the
reference's learned operators:

  models/model_voxception.py:11-68    _VoxceptionResNet
  models/model_voxception.py:71-144   AnalysisTransform
  models/model_voxception.py:147-214  SynthesisTransform
  models/model_voxception.py:217-252  HyperEncoder
  models/model_voxception.py:255-308  HyperDecoder

The arithmetic itself (tf.keras.layers.Conv3D / Conv3DTranspose with
padding='same', channels-last) lives in tensorflow-gpu==1.13.1 (README.md:19),
which is NOT under /root/reference and cannot be installed here; the reference
holds no test vectors for it.           *** PARITY UNPINNED ***
The semantics restated:
  stride 1 : y[o,co] = b + sum_{k in [0,3)^3, ci} x[o+k-1,ci] W[k,ci,co], zero outside
  stride 2 : y[o]    = sum x[2o+k] W[k]            (pad 0 before / 1 after per axis, N even)
  T-conv s2: y[o,co] = b + sum_{2i+k=o} x[i,ci] W[k,co,ci],  o in [0,2N)
Kernel layouts are TensorFlow's: conv [kd,kh,kw,Cin,Cout]; transpose conv
[kd,kh,kw,Cout,Cin].  All tensors are NDHWC float32 numpy arrays at the API.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# layer primitives (torch CPU)
# --------------------------------------------------------------------------
def _to_ncdhw(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).permute(0, 4, 1, 2, 3)


def _to_ndhwc(t):
    return t.permute(0, 2, 3, 4, 1).contiguous().numpy()


def conv3d_same(x, kernel, bias=None, stride=1, relu=False):
    """Keras Conv3D(padding='same'). x NDHWC, kernel [kd,kh,kw,Cin,Cout]."""
    t = _to_ncdhw(x)
    w = torch.from_numpy(np.ascontiguousarray(kernel, dtype=np.float32)).permute(4, 3, 0, 1, 2).contiguous()
    b = None if bias is None else torch.from_numpy(np.ascontiguousarray(bias, dtype=np.float32))
    k = kernel.shape[0]
    if stride == 1:
        y = F.conv3d(t, w, b, stride=1, padding=(k - 1) // 2)
    else:
        # TF 'SAME' with stride 2 on an even size: pad_total = k - 2, floor(total / 2) before, the rest after
        # (k = 3: (0, 1); k = 5: (1, 2); k = 9: (3, 4) — model_simple.py:20-41)
        assert stride == 2 and k % 2 == 1 and x.shape[1] % 2 == 0
        pb = max(k - 2, 0) // 2
        pa = max(k - 2, 0) - pb
        t = F.pad(t, (pb, pa, pb, pa, pb, pa))
        y = F.conv3d(t, w, b, stride=2, padding=0)
    if relu:
        y = torch.relu(y)
    return _to_ndhwc(y)


def conv3d_transpose_same(x, kernel, bias=None, relu=False):
    """Keras Conv3DTranspose(k, strides 2, padding='same'): the adjoint of the stride-2 'SAME' conv above, i.e.
    y[o] = sum_{2i + k - pb = o} x[i] W[k], pb = (k - 2) // 2, o in [0, 2N).  kernel [kd,kh,kw,Cout,Cin]."""
    t = _to_ncdhw(x)
    w = torch.from_numpy(np.ascontiguousarray(kernel, dtype=np.float32)).permute(4, 3, 0, 1, 2).contiguous()
    b = None if bias is None else torch.from_numpy(np.ascontiguousarray(bias, dtype=np.float32))
    n = x.shape[1]
    pb = max(kernel.shape[0] - 2, 0) // 2
    y = F.conv_transpose3d(t, w, None, stride=2, padding=0)[:, :, pb:pb + 2 * n, pb:pb + 2 * n, pb:pb + 2 * n]
    if b is not None:
        y = y + b.reshape(1, -1, 1, 1, 1)
    if relu:
        y = torch.relu(y)
    return _to_ndhwc(y)


# --------------------------------------------------------------------------
# slow pure-numpy restatement of the same three formulas (small cases only);
# it shares no code with torch and is used to cross-check the torch calls.
# --------------------------------------------------------------------------
def conv3d_same_naive(x, kernel, bias=None, stride=1, relu=False):
    B, D, H, W, Ci = x.shape
    k = kernel.shape[0]
    Co = kernel.shape[4]
    x = x.astype(np.float64)
    kernel = kernel.astype(np.float64)
    if stride == 1:
        p = (k - 1) // 2
        xp = np.pad(x, ((0, 0), (p, p), (p, p), (p, p), (0, 0)))
        y = np.zeros((B, D, H, W, Co))
        for a in range(k):
            for b_ in range(k):
                for c in range(k):
                    y += np.einsum("bdhwi,io->bdhwo", xp[:, a:a + D, b_:b_ + H, c:c + W, :], kernel[a, b_, c])
    else:
        pb = max(k - 2, 0) // 2
        pa = max(k - 2, 0) - pb
        xp = np.pad(x, ((0, 0), (pb, pa), (pb, pa), (pb, pa), (0, 0)))
        Do, Ho, Wo = D // 2, H // 2, W // 2
        y = np.zeros((B, Do, Ho, Wo, Co))
        for a in range(k):
            for b_ in range(k):
                for c in range(k):
                    y += np.einsum("bdhwi,io->bdhwo",
                                   xp[:, a:a + 2 * Do:2, b_:b_ + 2 * Ho:2, c:c + 2 * Wo:2, :], kernel[a, b_, c])
    if bias is not None:
        y += bias.astype(np.float64)
    if relu:
        y = np.maximum(y, 0)
    return y.astype(np.float32)


def conv3d_transpose_same_naive(x, kernel, bias=None, relu=False):
    B, D, H, W, Ci = x.shape
    Co = kernel.shape[3]
    k = kernel.shape[0]
    pb = max(k - 2, 0) // 2
    y = np.zeros((B, 2 * D + k, 2 * H + k, 2 * W + k, Co))
    xd = x.astype(np.float64)
    kd = kernel.astype(np.float64)
    for a in range(k):
        for b_ in range(k):
            for c in range(k):
                y[:, a:a + 2 * D:2, b_:b_ + 2 * H:2, c:c + 2 * W:2, :] += np.einsum(
                    "bdhwi,oi->bdhwo", xd, kd[a, b_, c])
    y = y[:, pb:pb + 2 * D, pb:pb + 2 * H, pb:pb + 2 * W, :]
    if bias is not None:
        y = y + bias.astype(np.float64)
    if relu:
        y = np.maximum(y, 0)
    return y.astype(np.float32)


# --------------------------------------------------------------------------
# networks.  `w` maps "<layer>/kernel" and "<layer>/bias" to numpy arrays with
# the layer names of the reference's Keras attributes (= checkpoint keys).
# --------------------------------------------------------------------------
def _conv(w, name, x, stride=1, relu=False):
    return conv3d_same(x, w[name + "/kernel"], w.get(name + "/bias"), stride=stride, relu=relu)


def vrn_block(w, name, x):
    """model_voxception.py:56-68."""
    t11 = _conv(w, name + "/conv1_1", x, relu=True)
    t12 = _conv(w, name + "/conv1_2", t11, relu=True)
    t21 = _conv(w, name + "/conv2_1", x, relu=True)
    t22 = _conv(w, name + "/conv2_2", t21, relu=True)
    t23 = _conv(w, name + "/conv2_3", t22, relu=True)
    residual = np.concatenate([t12, t23], axis=-1)
    return np.maximum(x + residual, 0).astype(np.float32)


def analysis_transform(w, x):
    """model_voxception.py:125-144.  x [B,N,N,N,1] -> y [B,N/4,N/4,N/4,16]."""
    f = _conv(w, "conv_in", x, relu=True)
    for i in (1, 2, 3):
        f = vrn_block(w, "vrn1_%d" % i, f)
    f = _conv(w, "down_1", f, stride=2, relu=True)      # use_bias=False (model_voxception.py:99)
    for i in (1, 2, 3):
        f = vrn_block(w, "vrn2_%d" % i, f)
    f = _conv(w, "down_2", f, stride=2, relu=True)      # use_bias=False (model_voxception.py:111)
    for i in (1, 2, 3):
        f = vrn_block(w, "vrn3_%d" % i, f)
    return _conv(w, "conv_out", f, relu=False)


def synthesis_transform(w, y):
    """model_voxception.py:195-214.  y [B,n,n,n,16] -> logits [B,4n,4n,4n,1]."""
    f = _conv(w, "deconv_in", y, relu=True)
    for i in (1, 2, 3):
        f = vrn_block(w, "vrn1_%d" % i, f)
    f = conv3d_transpose_same(f, w["up_1/kernel"], w["up_1/bias"], relu=True)
    for i in (1, 2, 3):
        f = vrn_block(w, "vrn2_%d" % i, f)
    f = conv3d_transpose_same(f, w["up_2/kernel"], w["up_2/bias"], relu=True)
    for i in (1, 2, 3):
        f = vrn_block(w, "vrn3_%d" % i, f)
    return _conv(w, "deconv_out", f, relu=False)


def hyper_encoder(w, y):
    """model_voxception.py:246-252."""
    f = _conv(w, "conv1", y, relu=True)
    f = _conv(w, "conv2", f, stride=2, relu=True)
    return _conv(w, "conv3", f, relu=False)


def hyper_decoder(w, z):
    """model_voxception.py:299-308.  Returns (loc, |scale|)."""
    f = _conv(w, "conv1", z, relu=True)
    f = conv3d_transpose_same(f, w["conv2/kernel"], w["conv2/bias"], relu=True)
    f = _conv(w, "conv3", f, relu=True)
    loc = _conv(w, "conv4_1", f, relu=False)
    scale = _conv(w, "conv4_2", f, relu=False)
    return loc, np.abs(scale)


def simple_analysis_transform(w, x):
    """models/model_simple.py:12-49: 9^3 s2 ReLU, 5^3 s2 ReLU, 5^3 s2 linear (no bias)."""
    f = _conv(w, "conv_1", x, stride=2, relu=True)
    f = _conv(w, "conv_2", f, stride=2, relu=True)
    return _conv(w, "conv_3", f, stride=2)


def simple_synthesis_transform(w, y):
    """models/model_simple.py:52-95: transposed 5^3 ReLU, 5^3 ReLU, 9^3 linear."""
    f = conv3d_transpose_same(y, w["deconv_1/kernel"], w.get("deconv_1/bias"), relu=True)
    f = conv3d_transpose_same(f, w["deconv_2/kernel"], w.get("deconv_2/bias"), relu=True)
    return conv3d_transpose_same(f, w["deconv_3/kernel"], w.get("deconv_3/bias"))


def sub(weights, prefix):
    """Slice a flat checkpoint-style dict ('analysis_transform/conv_in/kernel')."""
    p = prefix + "/"
    return {k[len(p):]: v for k, v in weights.items() if k.startswith(p)}
