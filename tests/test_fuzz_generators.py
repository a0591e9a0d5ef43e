"""The case generators and oracle workers of the round-6 differential campaigns (profiles/r06/fuzz_*.py) run here, without a GPU,
on a few seeds each: the campaigns themselves need the MI355X (their records sit next to the scripts), but their inputs and their
checker must not rot."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06")


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    import sys

    sys.path.insert(0, HERE)
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.path.remove(HERE)
    return mod


def test_engine_campaign_cases_and_oracle():
    F = _load("fuzz_engines")
    kinds = set()
    for seed in (0, 3, 23, 24):
        c = F.make_case(seed)
        assert c["masks"].shape == (c["B"], c["H"], c["W"]) and c["depth"].shape == (c["P"], c["H"], c["W"])
        assert (c["mb"] != 0).sum() == c["masks"].sum()
        s, rec, st, nv, kap = F.oracle_case(seed)
        assert s == seed and rec.shape == (c["B"], 39) and len(st) == len(nv) == len(kap) == c["B"]
        assert np.isnan(rec[st != 0]).all() and np.isfinite(rec[st == 0][:, :15]).all()
        kinds.update(c["mkind"])
    assert 14 in kinds   # (seeds 23 / 24 are polygon cases: the polygon generator and oracle/poly_oracle.py ran)


def test_point_cloud_and_annotation_campaign_cases_and_oracle():
    P = _load("fuzz_points")
    for seed in (1, 5):
        c = P.make_case(seed)
        s, out = P.oracle_case(seed)
        for method in ("pca", "convex_hull"):
            rec, st, nv, kap = out[method]
            assert rec.shape == (c["B"], 39) and set(np.unique(st)) <= {0, 1, 2, 3, 4}
    A = _load("fuzz_annotations")
    for seed in (1, 4):
        c = A.make_case(seed)
        s, (rec, st, keep, keep_default, nv, kap, gap) = A.oracle_case(seed)
        assert len(c["anns"]) == c["n"] == len(st)
        skipped = np.array([m is None for m in c["masks"]])
        assert (st[skipped] == 6).all() and not keep[skipped].any()
        assert (gap[(st == 0)] >= 0).all()


def test_reference_axis_noise_rule():
    from tests.test_gpu_parity import reference_axis_noise

    n = reference_axis_noise([1e9, 1e9, 50.0, np.nan], [10, 100, 100, 100], [1.0, 0.5, 1.0, 1.0])
    assert n[0] == 0.0 and n[3] == 0.0                 # below 20 points the reference is exact; no kappa, no slack
    assert n[1] == pytest.approx(8 * 2.0 ** -52 * 1e9 / 0.5) and n[2] < 1e-13
