"""Pins the CPU oracle (oracle/softras_oracle.c) to the golden vectors recorded from the
reference's own kernel bodies (tests/golden/make_golden.py).

Bar: forward bit-exact on soft_colors / aggrs_info / faces_info except where libm's expf differs
in the last ulp between the two builds (same glibc here, so exact); backward sums equal to fp32
round-off of the accumulated magnitude."""
import numpy as np
import pytest

import golden_io
from oracle import softras as oracle


@pytest.mark.parametrize("case", golden_io.softras_cases())
def test_oracle_matches_reference_golden(case):
    d = golden_io.load(case)
    out = oracle.render(d["face_vertices"], d["face_textures"], grad_soft_colors=d["grad_soft_colors"],
                        **golden_io.softras_kwargs(d))
    np.testing.assert_array_equal(out["faces_info"], d["faces_info"])
    np.testing.assert_array_equal(out["soft_colors"], d["soft_colors"])
    np.testing.assert_array_equal(out["aggrs_info"], d["aggrs_info"])
    # same summation order as the generator's sequential atomicAdd replay -> also exact
    np.testing.assert_array_equal(out["grad_faces"], d["grad_faces"])
    np.testing.assert_array_equal(out["grad_textures"], d["grad_textures"].reshape(out["grad_textures"].shape))


def test_golden_is_not_trivial():
    """the fixtures really exercise coverage, softmax and gradients"""
    d = golden_io.load("softras_bottle128_depth")
    a = d["soft_colors"][:, 3]
    assert 0.05 < (a > 0.5).mean() < 0.9
    assert np.abs(d["grad_faces"]).max() > 1e-3
    assert np.isfinite(d["grad_faces"]).all() and np.isfinite(d["soft_colors"]).all()
