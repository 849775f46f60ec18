"""ORACLE — TEST INFRASTRUCTURE ONLY (not the product path).

numpy float32 restatement of the reference's two entropy models:

  models/entropy_model.py:25-70     EntropyBottleneck.build  (parameter shapes / init)
  models/entropy_model.py:72-98     _logits_cumulative
  models/entropy_model.py:114-151   _likelihood
  models/entropy_model.py:153-181   call
  models/entropy_model.py:183-221   _get_cdf
  models/entropy_model.py:223-306   compress / decompress
  models/conditional_entropy_model.py:21-32    _standardized_cumulative (Laplace)
  models/conditional_entropy_model.py:34-56    _likelihood
  models/conditional_entropy_model.py:71-93    call
  models/conditional_entropy_model.py:95-124   _get_cdf
  models/conditional_entropy_model.py:126-201  compress / decompress

The integer side (pmf_to_quantized_cdf, range_encode, range_decode) is
oracle/coder.c (TF 1.13 contrib/coder restated; parity unpinned, see there).
No reference test pins these functions: *** PARITY UNPINNED *** — the float
formulas are restated line by line from the files above, evaluated in float32
with numpy (tf.math.round = round-half-to-even = np.rint).
"""
import numpy as np

from . import coder

F32 = np.float32


# ----------------------------------------------------------------------------
# EntropyBottleneck (factorized prior)
# ----------------------------------------------------------------------------
def eb_init_params(channels, init_scale=8.0, filters=(3, 3, 3), rng=None):
    """entropy_model.py:41-68: matrix_i const init, bais_i U(-.5,.5), factor_i zeros."""
    rng = rng or np.random.default_rng(0)
    f = (1,) + tuple(filters) + (1,)
    scale = init_scale ** (1.0 / (len(filters) + 1))
    p = {}
    for i in range(len(filters) + 1):
        init = np.log(np.expm1(1.0 / scale / f[i + 1]))
        p["matrix_%d" % i] = np.full((channels, f[i + 1], f[i]), init, F32)
        p["bais_%d" % i] = rng.uniform(-0.5, 0.5, (channels, f[i + 1], 1)).astype(F32)   # (sic) entropy_model.py:58
        p["factor_%d" % i] = np.zeros((channels, f[i + 1], 1), F32)
    return p


def _softplus(x):
    # tf.nn.softplus: log(exp(x) + 1)
    return np.logaddexp(x, F32(0)).astype(F32)


def eb_logits_cumulative(p, inputs):
    """entropy_model.py:72-98. inputs (C,1,n) float32 -> (C,1,n)."""
    logits = inputs.astype(F32)
    n_layers = len([k for k in p if k.startswith("matrix_")])
    for i in range(n_layers):
        matrix = _softplus(p["matrix_%d" % i])
        logits = np.matmul(matrix, logits).astype(F32)
        logits = (logits + p["bais_%d" % i]).astype(F32)
        factor = np.tanh(p["factor_%d" % i]).astype(F32)
        logits = (logits + factor * np.tanh(logits).astype(F32)).astype(F32)
    return logits


def _sigmoid(x):
    x = x.astype(F32)
    return (F32(1) / (F32(1) + np.exp(-x).astype(F32))).astype(F32)


def eb_likelihood_c1n(p, values_c1n):
    """entropy_model.py:137-143 on an already (C,1,n)-shaped tensor."""
    half = F32(0.5)
    lower = eb_logits_cumulative(p, values_c1n - half)
    upper = eb_logits_cumulative(p, values_c1n + half)
    sign = -np.sign(lower + upper).astype(F32)
    return np.abs(_sigmoid(sign * upper) - _sigmoid(sign * lower)).astype(F32)


def eb_call(p, inputs, training=False, noise=None, likelihood_bound=1e-9):
    """entropy_model.py:153-181. inputs [..., C]. Returns (values, likelihood)."""
    x = np.asarray(inputs, F32)
    if training:
        outputs = (x + noise.astype(F32)).astype(F32)
    else:
        outputs = np.rint(x).astype(F32)
    C = x.shape[-1]
    flat = np.moveaxis(outputs, -1, 0).reshape(C, 1, -1)
    lik = eb_likelihood_c1n(p, flat)
    lik = np.moveaxis(lik.reshape((C,) + outputs.shape[:-1]), 0, -1)
    lik = np.maximum(lik, F32(likelihood_bound))
    return outputs, lik.astype(F32)


def eb_pmf(p, min_v, max_v, likelihood_bound=1e-9):
    """entropy_model.py:199-214: pmf [C, N] over the integers min_v..max_v."""
    C = p["matrix_0"].shape[0]
    a = np.arange(min_v, max_v + 1, dtype=F32).reshape(1, 1, -1)
    a = np.tile(a, (C, 1, 1))
    lik = eb_likelihood_c1n(p, a)
    return np.maximum(lik, F32(likelihood_bound)).reshape(C, -1).astype(F32)


def eb_get_cdf(p, min_v, max_v, precision=16):
    """entropy_model.py:183-221 -> int32 [1, C, N+1]."""
    pmf = eb_pmf(p, min_v, max_v)
    cdf = coder.pmf_to_quantized_cdf(pmf, precision)
    return cdf.reshape(1, cdf.shape[0], -1)


def eb_compress(p, inputs, precision=16):
    """entropy_model.py:223-261. inputs [B,...,C] -> (bytes, min_v, max_v)."""
    x = np.asarray(inputs, F32)
    C = x.shape[-1]
    values = np.rint(x)
    min_v = int(np.floor(values.min()))
    max_v = int(np.ceil(values.max()))
    cdf = eb_get_cdf(p, min_v, max_v, precision)
    sym = (values.reshape(-1, C).astype(np.int32) - min_v).astype(np.int16)
    return coder.range_encode(sym, cdf, precision), min_v, max_v


def eb_decompress(p, string, min_v, max_v, shape, precision=16):
    """entropy_model.py:263-306."""
    shape = tuple(int(s) for s in shape)
    C = shape[-1]
    cdf = eb_get_cdf(p, int(min_v), int(max_v), precision)
    rows = int(np.prod(shape)) // C
    sym = coder.range_decode(string, (rows, C), cdf, precision)
    return (sym.astype(np.int32) + int(min_v)).reshape(shape).astype(F32)


# ----------------------------------------------------------------------------
# SymmetricConditional (Laplace prior conditioned on loc, scale)
# ----------------------------------------------------------------------------
def sc_standardized_cumulative(inputs, loc, scale):
    """conditional_entropy_model.py:21-32."""
    inputs = inputs.astype(F32)
    mask_r = (inputs > loc).astype(F32)
    mask_l = (inputs <= loc).astype(F32)
    e = np.exp((-np.abs(inputs - loc).astype(F32) / scale).astype(F32)).astype(F32)
    c_l = (F32(0.5) * e).astype(F32)
    c_r = (F32(1.0) - F32(0.5) * e).astype(F32)
    return (c_l * mask_l + c_r * mask_r).astype(F32)


def sc_likelihood(inputs, loc, scale):
    """conditional_entropy_model.py:34-56 (including the sign(2q-loc) quirk)."""
    inputs = inputs.astype(F32)
    upper = (inputs + F32(0.5)).astype(F32)
    lower = (inputs - F32(0.5)).astype(F32)
    sign = np.sign(((upper + lower).astype(F32) - loc).astype(F32)).astype(F32)
    upper = (-sign * (upper - loc).astype(F32) + loc).astype(F32)
    lower = (-sign * (lower - loc).astype(F32) + loc).astype(F32)
    cu = sc_standardized_cumulative(upper, loc, scale)
    cl = sc_standardized_cumulative(lower, loc, scale)
    return np.abs(cu - cl).astype(F32)


def sc_call(inputs, loc, scale, training=False, noise=None, likelihood_bound=1e-9):
    """conditional_entropy_model.py:71-93."""
    x = np.asarray(inputs, F32)
    loc = np.asarray(loc, F32)
    scale = np.asarray(scale, F32)
    outputs = (x + noise.astype(F32)).astype(F32) if training else np.rint(x).astype(F32)
    lik = np.maximum(sc_likelihood(outputs, loc, scale), F32(likelihood_bound))
    return outputs, lik.astype(F32)


def sc_pmf(loc, scale, min_v, max_v, likelihood_bound=1e-9):
    """conditional_entropy_model.py:104-120: pmf [rows, C, N]."""
    a = np.arange(min_v, max_v + 1, dtype=F32).reshape(1, 1, -1)
    lik = sc_likelihood(a, np.asarray(loc, F32)[..., None], np.asarray(scale, F32)[..., None])
    return np.maximum(lik, F32(likelihood_bound)).astype(F32)


def sc_get_cdf(loc, scale, min_v, max_v, precision=16):
    """conditional_entropy_model.py:95-124 -> int32 [rows, C, N+1]."""
    pmf = sc_pmf(loc, scale, min_v, max_v)
    r, c, n = pmf.shape
    return coder.pmf_to_quantized_cdf(pmf.reshape(r * c, n), precision).reshape(r, c, n + 1)


def sc_compress(inputs, loc, scale, precision=16):
    """conditional_entropy_model.py:126-163 (one call = one string)."""
    x = np.asarray(inputs, F32)
    C = x.shape[-1]
    loc = np.asarray(loc, F32).reshape(-1, C)
    scale = np.asarray(scale, F32).reshape(-1, C)
    values = np.rint(x.reshape(-1, C))
    min_v = int(np.floor(values.min()))
    max_v = int(np.ceil(values.max()))
    cdf = sc_get_cdf(loc, scale, min_v, max_v, precision)
    sym = (values.astype(np.int32) - min_v).astype(np.int16)
    return coder.range_encode(sym, cdf, precision), min_v, max_v


def sc_decompress(string, loc, scale, min_v, max_v, datashape, precision=16):
    """conditional_entropy_model.py:165-201."""
    datashape = tuple(int(s) for s in datashape)
    C = datashape[-1]
    loc = np.asarray(loc, F32).reshape(-1, C)
    scale = np.asarray(scale, F32).reshape(-1, C)
    cdf = sc_get_cdf(loc, scale, int(min_v), int(max_v), precision)
    rows = int(np.prod(datashape)) // C
    sym = coder.range_decode(string, (rows, C), cdf, precision)
    return (sym.astype(np.int32) + int(min_v)).reshape(datashape).astype(F32)
