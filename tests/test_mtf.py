"""CPU: the peak-valley MTF mirror (pylinac_amd/mtf.py) against the reference's KATs
(tests_basic/core/test_mtf.py:11-30) and, in the build container, the reference class itself."""
import warnings

import numpy as np
import pytest

from oracle import ref_loader
from pylinac_amd import mtf as m2


def test_reference_known_answers():
    m = m2.MTF((0.1, 0.2, 0.3), (500, 300, 100), (25, 50, 75))
    assert m.relative_resolution(50) == pytest.approx(0.24, abs=0.03)
    assert m.relative_resolution(90) == pytest.approx(0.15, abs=0.03)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert m.relative_resolution(10) == pytest.approx(0.3, abs=0.03)
        m2.MTF((0.1, 0.2, 0.3, 0.4), (500, 300, 500, 100), (25, 50, 25, 75))  # non-monotonic: warns, no raise
    with pytest.raises(ValueError):
        m2.MTF((0.1,), (500,), (25,))
    assert m2.michelson(np.array([3.0, 1.0])) == 0.5


@pytest.mark.skipif(not ref_loader.reference_available(), reason="/root/reference not present")
def test_bit_equal_to_reference_class():
    ref = ref_loader.ref("core.mtf")
    rng = np.random.default_rng(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for t in range(200):
            k = int(rng.integers(2, 9))
            sp = np.sort(rng.uniform(0.1, 2, k))
            mx = np.sort(rng.uniform(200, 900, k))[::-1]
            mn = np.sort(rng.uniform(10, 150, k))
            if t % 5 == 0:
                mx = rng.permutation(mx)
            a = ref.MTF(list(sp), list(mx), list(mn))
            b = m2.MTF(list(sp), list(mx), list(mn))
            assert a.norm_mtfs == b.norm_mtfs
            for x in (10, 50, 80, 90, 99):
                assert a.relative_resolution(x) == b.relative_resolution(x)
