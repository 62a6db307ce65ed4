"""The reference mapper's coefficient search on the GPU.

Mirror of `MapToFBSHeur._find_lincomb_coefs_search(xy_mvt, r_tt)` (/root/reference/fbs_mapper/map_to_fbs.py:363-392):
same arguments, same return value -- `((a, b), mvt)` with `mvt = a * xy_mvt[:, 0] + b * xy_mvt[:, 1]`, or `(None, None)`
-- computed by `fbs_search_lincomb_coefs` of libfbsexec.so (every candidate of the reference's grid in one kernel launch;
csrc/fbs_mapper_search.hip).  `install(MapToFBSHeur)` replaces the reference's method by this one; nothing else of the
mapper is rebuilt here (the heuristic mappers stay the reference's).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as nat


class CoefSearcher:
    """One per device; not thread-safe."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        rc = nat.lib.fbs_searcher_create(device, C.byref(self._h))
        if rc != 0:
            self._h = None
            raise nat.FbsError(rc, nat.lib.fbs_searcher_last_error(None).decode())

    def close(self):
        if getattr(self, "_h", None) and nat.lib is not None:
            nat.lib.fbs_searcher_destroy(self._h)
        self._h = None

    __del__ = close

    @property
    def last_kernel_ms(self):
        return float(nat.lib.fbs_searcher_last_kernel_ms(self._h))

    def search(self, xy_mvt, r_tt, fbs_size, max_fbs_size):
        xy = np.asarray(xy_mvt)
        x = np.ascontiguousarray(xy[:, 0], np.int32)
        y = np.ascontiguousarray(xy[:, 1], np.int32)
        assert np.array_equal(x, xy[:, 0]) and np.array_equal(y, xy[:, 1]), "cone values beyond 32 bits"
        tt = np.ascontiguousarray(np.asarray(r_tt) != 0, np.uint8)
        ab = (C.c_int32 * 2)()
        found = C.c_int(0)
        mvt = np.empty(len(x), np.int64)
        rc = nat.lib.fbs_search_lincomb_coefs(self._h, x.ctypes.data, y.ctypes.data, tt.ctypes.data, len(x), int(fbs_size),
                                              int(max_fbs_size), C.cast(ab, C.c_void_p), mvt.ctypes.data, C.byref(found))
        if rc != 0:
            raise nat.FbsError(rc, nat.lib.fbs_searcher_last_error(self._h).decode())
        if not found.value:
            return None, None
        return (int(ab[0]), int(ab[1])), mvt


_default: dict[int, CoefSearcher] = {}


def find_lincomb_coefs_search(xy_mvt, r_tt, fbs_size, max_fbs_size, device: int = 0):
    """Functional form of the reference method (which reads fbs_size / max_fbs_size from its mapper object)."""
    s = _default.get(device)
    if s is None:
        s = _default[device] = CoefSearcher(device)
    return s.search(xy_mvt, r_tt, fbs_size, max_fbs_size)


def install(mapper_class, device: int = 0):
    """Make the reference's `MapToFBSHeur` search on the GPU: `install(map_to_fbs.MapToFBSHeur)`.  Instances created with
    cone_merger="search" bind `self._find_lincomb_coefs = self._find_lincomb_coefs_search` in __init__
    (map_to_fbs.py:62-65), so patching the class attribute before constructing the mapper is enough."""
    def _find_lincomb_coefs_search(self, xy_mvt, r_tt):
        return find_lincomb_coefs_search(xy_mvt, r_tt, self.fbs_size, self.max_fbs_size, device)

    mapper_class._find_lincomb_coefs_search = _find_lincomb_coefs_search
    return mapper_class
