"""oracle/bitstream.py against bytes written by the reference's writer (tests/golden/bitstream_hyper.npz)."""
import numpy as np

from oracle import bitstream as ob


def test_writer_bytes(golden):
    g = golden("bitstream_hyper.npz")
    y_strings = ob.unpack_strings(g["y_concat"].tobytes(), g["y_lens"])
    assert ob.pack_strings_head(y_strings, g["y_min_vs"], g["y_max_vs"], g["y_shape"]) == g["strings_head"].tobytes()
    assert ob.pack_strings(y_strings) == g["strings"].tobytes()
    assert ob.pack_strings_hyper(g["z_string"].tobytes(), int(g["z_min_v"]), int(g["z_max_v"]),
                                 g["z_shape"]) == g["strings_hyper"].tobytes()
    assert ob.pack_pointnums(g["points_numbers"]) == g["pointnums"].tobytes()
    # structural pins recorded in demo.ipynb:700-705 (416 = 2 + 202 + 202 + 10; 404 = 2*202; z header 12 B)
    assert len(g["strings_head"]) == 2 + 6 + (4 * 1 + 2 * 3) + 10
    assert len(g["strings_hyper"]) == 12 + len(g["z_string"])


def test_reader(golden):
    g = golden("bitstream_hyper.npz")
    mn, mx, lens, y_shape = ob.unpack_strings_head(g["strings_head"].tobytes())
    assert np.array_equal(mn, g["y_min_vs"]) and np.array_equal(mx, g["y_max_vs"])
    assert list(lens) == list(g["y_lens"]) and np.array_equal(y_shape, g["y_shape"])
    z, zmin, zmax, z_shape = ob.unpack_strings_hyper(g["strings_hyper"].tobytes())
    assert z == g["z_string"].tobytes() and (zmin, zmax) == (-6, 5) and np.array_equal(z_shape, g["z_shape"])
    assert np.array_equal(ob.unpack_pointnums(g["pointnums"].tobytes()), g["points_numbers"])
    # the case the reference's own reader parses: same answers
    mn, mx, lens, y_shape = ob.unpack_strings_head(g["h_strings_head"].tobytes())
    assert np.array_equal(mn, g["h_read_y_min_vs"]) and np.array_equal(mx, g["h_read_y_max_vs"])
    assert list(lens) == list(g["h_lens"]) and np.array_equal(y_shape, g["h_read_y_shape"])
