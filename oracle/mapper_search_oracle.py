"""TEST INFRASTRUCTURE, NOT PRODUCT CODE.  CPU restatement (numpy) of the reference mapper's coefficient search,
`MapToFBSHeur._find_lincomb_coefs_search` (/root/reference/fbs_mapper/map_to_fbs.py:363-392) with the helpers it rests
on: `_mvt_size` :70-71, `_comp_boot_test_vector` :73-76, `_is_mvt_valid` :78-79, `_is_test_vector_valid` :81-98,
`_is_lut_valid` :100-113, `_generate_coefs_grouped_by_fbs_size` :344-361.

PINNED: equals the reference's own results on the 439 recorded calls of tests/golden/_mapper_search.json.gz (captured by
importing the reference, tests/golden/capture_mapper_search.py) -- tests/test_mapper_search.py.  Only tests/ and
tools/mapper_search_bench.py (the CPU baseline of that kernel) may import this module.

Given the multi-value columns x, y of two cones over the rows of their joint truth table and the merged gate's output
bit per row, find integers (a, b) such that v = a x + b y is a legal bootstrap input: rows with different output bits
never share a value, and the table over [min v, max v] is evaluable at `fbs_size` (fits, or is at most `max_fbs_size`
long and negacyclic-compatible, don't-care slots filled with 0 or with 1).  Candidates are tried by increasing
|a| (size_x - 1) + |b| (size_y - 1); within the first size that has a legal candidate the smallest sum of squares wins,
the first in the reference's order (pairs sorted descending) on ties."""
import itertools

import numpy as np


def mvt_size(mvt):                                   # :70-71
    return int(np.max(mvt) - np.min(mvt) + 1)


def comp_boot_test_vector(tt, mvt, missing_val):     # :73-76
    lo, hi = int(mvt.min()), int(mvt.max())
    tv = [missing_val] * (hi - lo + 1)
    for v, t in zip(mvt, tt):
        tv[int(v) - lo] = int(t)
    return tv


def is_mvt_valid(tt, mvt):                           # :78-79
    return len(set(mvt[tt == 0].tolist()) & set(mvt[tt == 1].tolist())) == 0


def is_test_vector_valid(tv, fbs_size, max_fbs_size):   # :81-98
    if len(tv) <= fbs_size:
        return True
    if len(tv) <= max_fbs_size:
        tv = np.array(tv)
        start, end = tv[0:len(tv) - fbs_size], tv[fbs_size:]
        mode1 = bool(np.all(start != end))
        mode2 = bool(np.all(start == end) and np.all(0 == start))
        mode3 = bool(np.all(start == end) and np.all(1 == start))
        return mode1 or mode2 or mode3
    return False


def is_lut_valid(tt, mvt, fbs_size, max_fbs_size):   # :100-113
    if not is_mvt_valid(tt, mvt):
        return False
    if mvt_size(mvt) <= fbs_size:
        return True
    return (is_test_vector_valid(comp_boot_test_vector(tt, mvt, 0), fbs_size, max_fbs_size)
            or is_test_vector_valid(comp_boot_test_vector(tt, mvt, 1), fbs_size, max_fbs_size))


def candidates_by_size(size1, size2):                # :344-361
    """[(key, [(a, b), ...])] in the order the reference walks them: keys ascending, pairs descending."""
    if size1 < size2:
        pairs = itertools.product(range(size2 + 1), range(-size1, size1 + 1))
    else:
        pairs = itertools.product(range(-size2, size2 + 1), range(size1 + 1))
    groups = {}
    for a, b in pairs:
        groups.setdefault(abs(a) * (size1 - 1) + abs(b) * (size2 - 1), []).append((a, b))
    return [(k, sorted(groups[k], reverse=True)) for k in sorted(groups)]


def find_lincomb_coefs_search(x, y, r_tt, fbs_size, max_fbs_size):   # :363-392
    x, y, r_tt = np.asarray(x, np.int64), np.asarray(y, np.int64), np.asarray(r_tt, np.int64)
    best_ab, best_mvt, best_norm2 = None, None, None
    for _, pairs in candidates_by_size(mvt_size(x), mvt_size(y)):
        for a, b in pairs:
            mvt = a * x + b * y
            norm2 = int(np.square(mvt).sum())
            if best_ab is None or norm2 < best_norm2:
                if is_lut_valid(r_tt, mvt, fbs_size, max_fbs_size):
                    best_ab, best_mvt, best_norm2 = (a, b), mvt, norm2
        if best_ab is not None:
            break
    return best_ab, best_mvt
