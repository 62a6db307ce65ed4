"""the mapper's coefficient search.  CPU part: the oracle (numpy restatement) against the 439 calls
recorded from the reference itself (tests/golden/_mapper_search.json.gz), and the library's candidate enumeration."""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import mapper_search_oracle as mso

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "_mapper_search.json.gz")


def cases():
    with gzip.open(GOLDEN, "rb") as f:
        return json.loads(f.read().decode())["cases"]


def test_fixture_file_is_what_the_capture_script_describes():
    cs = cases()
    assert len(cs) == 439 and sum(c["ab"] is None for c in cs) == 218
    assert {len(c["tt"]) for c in cs} >= {4, 64, 1024, 4096}
    assert {(c["fbs_size"], c["max_fbs_size"]) for c in cs} >= {(15, 30), (15, 15), (31, 62), (7, 14), (2, 4)}


def test_oracle_equals_the_reference_on_every_recorded_call():
    for c in cases():
        ab, mvt = mso.find_lincomb_coefs_search(c["x"], c["y"], c["tt"], c["fbs_size"], c["max_fbs_size"])
        if c["ab"] is None:
            assert ab is None and mvt is None
        else:
            assert list(ab) == c["ab"] and mvt.tolist() == c["mvt"]


def test_candidate_order_matches_the_reference_grouping():
    groups = mso.candidates_by_size(3, 2)
    assert [k for k, _ in groups] == sorted({abs(a) * 2 + abs(b) * 1 for a in range(-2, 3) for b in range(0, 4)})
    assert groups[0] == (0, [(0, 0)])
    assert groups[1][1] == [(0, 1)] and groups[2][1] == [(1, 0), (0, 2), (-1, 0)]      # pairs of one size, descending
    n = sum(len(p) for _, p in groups)
    assert n == 5 * 4


@pytest.mark.gpu
def test_kernel_equals_the_reference_on_every_recorded_call():
    from tfhe_fbs_map_amd.mapper_search import find_lincomb_coefs_search
    for c in cases():
        ab, mvt = find_lincomb_coefs_search(np.column_stack([c["x"], c["y"]]), c["tt"], c["fbs_size"], c["max_fbs_size"])
        if c["ab"] is None:
            assert ab is None and mvt is None
        else:
            assert list(ab) == c["ab"] and np.asarray(mvt).tolist() == c["mvt"]


@pytest.mark.gpu
def test_kernel_equals_the_oracle_on_random_cones():
    """Sizes up to the reference's limit of 16 support variables (65 536 rows), value ranges up to 32, random tables."""
    from tfhe_fbs_map_amd.mapper_search import find_lincomb_coefs_search
    rng = np.random.default_rng(3)
    found = 0
    for trial in range(40):
        R = int(2 ** rng.integers(2, 17 if trial % 8 == 0 else 11))
        sx, sy = int(rng.integers(1, 16)), int(rng.integers(1, 16))
        x = rng.integers(0, sx + 1, R) + int(rng.integers(-3, 4))
        y = rng.integers(0, sy + 1, R) + int(rng.integers(-3, 4))
        p = int(rng.choice([3, 7, 15, 31]))
        maxp = p if trial % 3 == 0 else 2 * p
        a0, b0 = int(rng.integers(0, 3)), int(rng.integers(-2, 3))
        v = a0 * x + b0 * y
        tt = ((v - v.min()) % 2 if trial % 2 else rng.integers(0, 2, R)).astype(np.int64)     # half of them have a solution
        want_ab, want_mvt = mso.find_lincomb_coefs_search(x, y, tt, p, maxp)
        ab, mvt = find_lincomb_coefs_search(np.column_stack([x, y]), tt, p, maxp)
        assert (ab is None) == (want_ab is None), trial
        if ab is not None:
            found += 1
            assert tuple(ab) == tuple(want_ab) and np.array_equal(mvt, want_mvt), trial
    assert found >= 5
