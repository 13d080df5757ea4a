"""GPU: the reference's own unit tests for the hot path (tests/known_answers.py) run
against the HIP blocks through the C ABI."""
import pytest

import known_answers as KA

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("check", KA.ALL_CHECKS, ids=lambda f: f.__name__)
def test_reference_known_answers(check):
    import rustradio_amd as rr
    check(rr)


def test_quad_known_fast_mode():
    import rustradio_amd as rr
    KA.check_quad_known(rr, rr.ATAN2_FAST)


def test_multiband_matches_oracle():
    """fir::multiband (setup time, host code of the product) == the oracle restatement, bit for bit"""
    import numpy as np
    import rustradio_amd as rr
    from oracle import pyoracle as orc
    for ntaps, bands in ((101, [(0.1, 0.3)]), (64, [(0.0, 0.2), (0.5, 0.75)]), (1001, [(0.25, 0.26)])):
        w = np.blackman(ntaps).astype(np.float32)
        a, b = rr.multiband(bands, w), orc.multiband(bands, w)
        assert a is not None and np.array_equal(a, b)
    assert rr.multiband([(0.5, 0.2)], np.ones(16, np.float32)) is None
    assert rr.multiband([(0.0, 2.5)], np.ones(16, np.float32)) is None
