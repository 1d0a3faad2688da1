"""CPU: host-side logic of the mirror API (argument parsing, weights, sharding, error behaviour)."""
import numpy as np
import pytest
from scipy.ndimage import _filters

from oracle import pylinac_oracle as o
from pylinac_amd import array_utils as au
from pylinac_amd import dist as pdist
from pylinac_amd import ops


def test_gaussian_weights_bit_equal_scipy():
    for sigma in (1, 2, 3, 5, 7, 27):
        w, lw = ops.gaussian_weights(sigma)
        ref = _filters._gaussian_kernel1d(float(sigma), 0, int(4.0 * float(sigma) + 0.5))[::-1]
        assert lw == int(4.0 * sigma + 0.5) and np.array_equal(w, ref)
        assert np.array_equal(w, w[::-1])  # exactly symmetric -> scipy's symmetric branch


def test_resolve_filter_size_like_reference():
    assert au.resolve_filter_size(1000, 0.05) == 50
    assert au.resolve_filter_size(7, 0.1) == 1
    assert au.resolve_filter_size(96, 0.05) == o.resolve_filter_size(np.zeros((96, 4)), 0.05) == 5
    assert au.resolve_filter_size(10, 3) == 3
    for bad in (2.3, 1.0, 0.0, -0.5):
        with pytest.raises(ValueError, match="not between"):
            au.resolve_filter_size(10, bad)


def test_empty_and_bad_inputs_raise_like_reference():
    with pytest.raises(ValueError, match="must not be empty"):
        au.filter(np.array([]), 3)
    with pytest.raises(ValueError, match="must not be empty"):
        au.ground(np.array([]))
    with pytest.raises(ValueError, match="Max must be larger"):
        au.stretch(np.arange(5.0), min=1, max=1)
    with pytest.raises(ValueError):
        au.geometric_center_idx(np.zeros((2, 2)))
    assert au.geometric_center_idx(np.arange(8)) == 3.5
    assert au.geometric_center_value(np.array([1.0, 3.0, 5.0, 7.0])) == 4.0


@pytest.mark.parametrize("length", [9, 200, 1024])
def test_peak_params_match_parse_peak_args(length):
    values = np.linspace(-1, 3, length)
    cases = [dict(), dict(threshold=0.3, peak_separation=0.05), dict(threshold=15, peak_separation=7),
             dict(search_region=(0.2, 0.8)), dict(search_region=(3, length - 2)), dict(threshold=1, peak_separation=1),
             dict(threshold=0, peak_separation=0.001), dict(search_region=(0.9, 0.1))]
    for kw in cases:
        sep, shift, thr, trimmed = o.parse_peak_args(kw.get("peak_separation", 0), kw.get("search_region", (0.0, 1.0)),
                                                     kw.get("threshold", -np.inf), values)
        p = ops.make_peak_params(length, **kw)
        assert p.distance == max(int(np.ceil(sep)), 1)
        assert p.region_hi - p.region_lo == len(trimmed)
        if len(trimmed):
            assert p.region_lo == shift
        if p.threshold_is_ratio:
            assert values.min() + p.threshold * (values.max() - values.min()) == thr
        else:
            assert p.threshold == thr


def test_peak_params_defaults_and_errors():
    p = ops.make_peak_params(100, fwxm_height=0.3, max_number=1)
    assert p.rel_height == 1 - 0.3 and p.max_number == 1 and p.sort_key == 0 and p.has_prominence == 0
    assert ops.make_peak_params(100).max_number == -1
    with pytest.raises(KeyError):
        ops.make_peak_params(100, peak_sort="nonsense")


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 256, 10000):
        for world in (1, 2, 3, 4, 8):
            spans = [pdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_synthetic_frames_are_deterministic_and_u16():
    import torch

    from pylinac_amd.synthetic import epid_open_field_frames

    a = epid_open_field_frames(2, 64, 80, seed0=5)
    b = epid_open_field_frames(2, 64, 80, seed0=5)
    assert a.dtype == torch.uint16 and a.shape == (2, 64, 80)
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    x = a.numpy()
    assert x.max() > 30000 and x.min() < 5000


def test_hann_window_restatement_is_scipys():
    """mtf.hann restates scipy.signal.windows.hann (the reference's default ESF window) bit for bit."""
    from scipy.signal import windows

    from pylinac_amd import mtf

    for m in (1, 2, 3, 6, 8, 97, 256, 1001):
        assert np.array_equal(mtf.hann(m), windows.hann(m)), m
    assert np.array_equal(mtf.boxcar(5), windows.boxcar(5))


def test_regionprops_formulas_vs_skimage_golden(golden):
    """regionprops.py (exact integer central moments + scikit-image 0.18.3's expressions) against scikit-image itself"""
    import next_row_checks as checks

    checks.check_regionprops_formulas(golden("regionprops"), checks.raw_moments_numpy)


def test_batched_phantom_axis_fits_are_polyfit_per_volume():
    """ct.find_phantom_axes_batch (stacked gelsd through the gufunc behind np.linalg.lstsq) == np.polyfit volume by volume,
    bit for bit, with outlier slices, unequal kept-slice counts and volumes that send it to the per-volume loop."""
    import numpy as np

    from pylinac_amd import ct

    rng = np.random.default_rng(0)
    for trial in range(12):
        nv, spv = 7, 40
        roi = np.zeros((nv * spv, 8))
        roi[:, 3] = 255 + rng.normal(0, 0.3, nv * spv) + np.tile(np.arange(spv) * 0.03, nv)
        roi[:, 4] = 256 + rng.normal(0, 0.3, nv * spv)
        k = rng.integers(0, nv * spv, 15)
        roi[k, 3] += rng.choice([-9, 9], 15)
        if trial % 3 == 0:
            roi[rng.integers(0, nv * spv, 3), 0] = 3          # a slice without the phantom: per-volume path
        fzx, fzy = ct.find_phantom_axes_batch(roi, nv)
        for v in range(nv):
            zx, zy, _ = ct.find_phantom_axis_volume(None, 0.5, roi=roi[v * spv:(v + 1) * spv])
            assert np.array_equal(fzx[v], zx) and np.array_equal(fzy[v], zy)


def test_fma_quotient_identity_exhaustive(tmp_path):
    """The three-operation quotient of pl_quot (csrc/pl_common.h) == float64 division for all 65536 x 65535 integer pairs
    the picket-fence kernels can meet (tests/fma_quotient_check.c, gcc; a few seconds on 8 cores)."""
    import subprocess
    from pathlib import Path

    src = Path(__file__).resolve().parent / "fma_quotient_check.c"
    exe = tmp_path / "fma_quotient_check"
    subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "mismatches: 0" in r.stdout, r.stdout + r.stderr


def test_detection_conditions_are_accepted_by_identity_only():
    """a user callable that shares a default predicate's NAME must not be silently replaced by the built-in."""
    import functools

    from pylinac_amd import metrics as m

    disk = [m.is_right_size_bb, m.is_round, m.is_right_circumference, m.is_symmetric, m.is_solid]
    m.SizedDiskLocator((0, 0), (10, 10), 2, 1, detection_conditions=disk[::-1])            # order is immaterial

    def is_round(region, *a, **k):                                                            # a stricter user version
        return False

    for bad in ([is_round] + disk[:1] + disk[2:], disk[:-1] + [lambda r: True], disk[:-1] + [functools.partial(disk[-1])],
                disk[:-1]):
        with pytest.raises(NotImplementedError):
            m.SizedDiskLocator((0, 0), (10, 10), 2, 1, detection_conditions=bad)
    m.GlobalSizedFieldLocator(10, 10, 1, detection_conditions=[m.is_right_square_perimeter, m.is_right_area_square])
    with pytest.raises(NotImplementedError):
        m.GlobalSizedFieldLocator(10, 10, 1, detection_conditions=disk)


def test_profile_cache_follows_x_values_rebinding():
    """what the edge search derived from the old coordinates must not answer after ``x_values`` is rebound
    (PhysicalProfileMixin.gamma does that to deep copies, pylinac/core/profile.py:861-866)."""
    import copy

    from pylinac_amd import profile as p

    prof = p.ProfileBase(np.array([0, 1, 2, 3, 2, 1, 0.0]))
    prof._cache["probe"] = 1
    clone = copy.deepcopy(prof)
    assert clone._cache == {"probe": 1}
    clone.x_values = clone.x_values - 3
    assert clone._cache == {} and prof._cache == {"probe": 1}
    assert np.array_equal(clone.x_values, np.arange(7) - 3)
    prof.values = prof.values * 2
    assert prof._cache == {}
