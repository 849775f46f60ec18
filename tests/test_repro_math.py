"""Reproducible elementary functions (pcgcv1_amd/csrc/repro_math.h): the host library, the HIP kernels and the
oracle's independent numpy restatement must agree BIT FOR BIT, and each must stay within a few ulp of libm."""
import ctypes

import numpy as np
import pytest

from oracle import entropy as oent
from pcgcv1_amd import _lib

FUNCS = [(0, "exp", oent.r_exp), (1, "log", oent.r_log), (2, "tanh", oent.r_tanh), (3, "sigmoid", oent.r_sigmoid),
         (4, "softplus", oent.r_softplus)]


def _inputs(fn):
    rng = np.random.default_rng(100 + fn)
    if fn == 1:      # log: normal positive numbers; (1, 2] is what softplus feeds it
        x = np.concatenate([rng.uniform(1, 2, 200000), np.exp(rng.uniform(-80, 80, 200000)), [1.0, 2.0, 0.5, 1e-30, 3e38]])
    elif fn == 2:
        x = np.concatenate([rng.uniform(-12, 12, 300000), rng.uniform(-0.7, 0.7, 100000), [0.0, -0.0, 0.625, -0.625, 50, -50]])
    else:
        x = np.concatenate([rng.uniform(-90, 90, 300000), rng.uniform(-2, 2, 100000), -np.exp(rng.uniform(-20, 5, 100000)),
                            [0.0, -0.0, -87.0, -88.0, 88.0, 89.0, -1e10, 1e10, np.nan, -np.inf]])
    return np.ascontiguousarray(x, np.float32)


def _libm(fn, x):
    x = x.astype(np.float64)
    with np.errstate(all="ignore"):
        return [np.exp(x), np.log(x), np.tanh(x), 1.0 / (1.0 + np.exp(-x)), np.logaddexp(x, 0.0)][fn]


@pytest.mark.parametrize("fn,name,oracle", FUNCS)
def test_host_matches_oracle_bitwise_and_libm_within_ulps(fn, name, oracle):
    x = _inputs(fn)
    y = np.empty_like(x)
    _lib.check_host(_lib.host().pcgc_host_repro_eval(fn, x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), x.size))
    with np.errstate(all="ignore"):
        ref = oracle(x)
    assert np.array_equal(y.view(np.uint32), ref.view(np.uint32)), name
    ok = np.isfinite(x) & (np.abs(x) < 86)            # inside the clamp of exp
    exact = _libm(fn, x[ok])
    ulp = np.abs(y[ok].astype(np.float64) - exact) / np.spacing(np.abs(exact).astype(np.float32)).astype(np.float64)
    assert ulp.max() <= 3.0, (name, ulp.max())


def test_exp_clamp_and_monotone():
    x = np.array([-1e10, -88.0, -87.0, np.nan], np.float32)
    assert np.unique(oent.r_exp(x)).size == 1                       # everything at or below -87 (and NaN) is exp(-87)
    g = np.linspace(-30, 0, 200001).astype(np.float32)
    assert np.all(np.diff(oent.r_exp(g).astype(np.float64)) >= 0)  # the Laplace tail must not wiggle


@pytest.mark.gpu
@pytest.mark.parametrize("fn,name,oracle", FUNCS)
def test_device_matches_oracle_bitwise(fn, name, oracle):
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    x = _inputs(fn)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    _lib.check(_lib.hip().pcgc_repro_eval(fn, _lib.dptr(xd), _lib.dptr(yd), x.size, _lib.stream()))
    with np.errstate(all="ignore"):
        ref = oracle(x)
    assert np.array_equal(yd.cpu().numpy().view(np.uint32), ref.view(np.uint32)), name
