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
