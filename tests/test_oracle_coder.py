"""oracle/coder.c (TF 1.13 contrib/coder restated; parity unpinned) — self-consistency:
C vs an independent pure-Python restatement, round trips, structural properties."""
import numpy as np
import pytest

from oracle import coder


def _rand_pmf(rng, n, peaky):
    p = rng.random(n).astype(np.float64) ** peaky
    p /= p.sum()
    return np.maximum(p, 1e-9).astype(np.float32)


@pytest.mark.parametrize("seed", range(6))
def test_cdf_c_vs_python(seed):
    rng = np.random.default_rng(seed)
    for n in (2, 3, 5, 9, 17, 31):
        for peaky in (1, 4, 12):
            pmf = _rand_pmf(rng, n, peaky)
            if seed % 2:
                pmf = (pmf * rng.uniform(0.3, 1.0)).astype(np.float32)      # truncated tails: sum < 1 -> gain branch
            c = coder.pmf_to_quantized_cdf(pmf[None])[0]
            assert c.tolist() == coder.py_pmf_to_quantized_cdf_row(pmf)
            assert c[0] == 0 and c[-1] == 65536 and np.all(np.diff(c) >= 1)


def test_cdf_ties_and_bounds():
    pmf = np.full((1, 8), 1e-9, np.float32)                                  # everything at the likelihood bound
    c = coder.pmf_to_quantized_cdf(pmf)[0]
    assert c[-1] == 65536 and np.all(np.diff(c) >= 1)
    assert c.tolist() == coder.py_pmf_to_quantized_cdf_row(pmf[0])
    pmf = np.array([[0.5, 0.5]], np.float32)
    assert coder.pmf_to_quantized_cdf(pmf)[0].tolist() == [0, 32768, 65536]
    pmf = np.array([[0.7, 0.7, 0.7]], np.float32)                            # sum > 1 -> penalty branch, equal keys
    c = coder.pmf_to_quantized_cdf(pmf)[0]
    assert c[-1] == 65536 and c.tolist() == coder.py_pmf_to_quantized_cdf_row(pmf[0])


@pytest.mark.parametrize("seed", range(5))
def test_range_coder_c_vs_python_and_roundtrip(seed):
    rng = np.random.default_rng(100 + seed)
    rows, cols, n = 40, 4, int(rng.integers(2, 12))
    pmf = np.stack([_rand_pmf(rng, n, rng.choice([1, 6, 20])) for _ in range(rows * cols)])
    cdf = coder.pmf_to_quantized_cdf(pmf).reshape(rows, cols, n + 1)
    # draw symbols from the model so streams are realistic, plus some adversarial rare symbols
    u = rng.integers(0, 65536, (rows, cols))
    sym = np.array([[np.searchsorted(cdf[r, c], u[r, c], side="right") - 1 for c in range(cols)]
                    for r in range(rows)], np.int16)
    sym[::7, 0] = n - 1
    s_c = coder.range_encode(sym, cdf)
    s_py = coder.py_range_encode(sym.reshape(-1).tolist(), [cdf[r, c].tolist() for r in range(rows) for c in range(cols)])
    assert s_c == s_py
    assert np.array_equal(coder.range_decode(s_c, (rows, cols), cdf), sym)
    assert coder.py_range_decode(s_c, [cdf[r, c].tolist() for r in range(rows) for c in range(cols)]) == sym.reshape(-1).tolist()
    assert len(s_c) == 0 or s_c[-1] != 0                                     # Finalize never writes a trailing zero


def test_broadcast_cdf_and_carry_stress():
    rng = np.random.default_rng(7)
    C, n = 8, 5
    pmf = np.stack([_rand_pmf(rng, n, 8) for _ in range(C)])
    cdf = coder.pmf_to_quantized_cdf(pmf).reshape(1, C, n + 1)               # entropy_model.py:219 shape
    sym = rng.integers(0, n, (4096, C)).astype(np.int16)
    s = coder.range_encode(sym, cdf)
    assert np.array_equal(coder.range_decode(s, sym.shape, cdf), sym)
    # long runs of the most probable symbol with p ~ 1 drive the carry/delay path
    pmf = np.array([[1 - 3e-5, 1e-5, 2e-5]], np.float32)
    cdf = coder.pmf_to_quantized_cdf(pmf).reshape(1, 1, 4)
    sym = np.zeros((20000, 1), np.int16)
    sym[rng.integers(0, 20000, 40), 0] = rng.integers(1, 3, 40)
    s = coder.range_encode(sym, cdf)
    assert np.array_equal(coder.range_decode(s, sym.shape, cdf), sym)
    assert s == coder.py_range_encode(sym[:, 0].tolist(), [cdf[0, 0].tolist()] * len(sym))
    # all-most-probable stream: a handful of bytes; empty input: empty string
    assert len(coder.range_encode(np.zeros((65536, 1), np.int16), cdf)) < 64
    assert coder.range_encode(np.zeros((0, 1), np.int16), cdf) == b""


def test_symbol_out_of_range_is_an_error():
    cdf = coder.pmf_to_quantized_cdf(np.array([[0.5, 0.5]], np.float32)).reshape(1, 1, 3)
    with pytest.raises(ValueError):
        coder.range_encode(np.array([[2]], np.int16), cdf)
