#!/usr/bin/env python3
"""Golden vectors for the mapper's coefficient search, captured by IMPORTING the reference.

Runs only in the build container.  `MapToFBSHeur._find_lincomb_coefs_search`
(/root/reference/fbs_mapper/map_to_fbs.py:363-392) is wrapped so that every call made while the reference's `search`
mapper maps a set of circuits is recorded: inputs (the two cones' multi-value columns `xy_mvt`, the merged truth table
`r_tt`, `fbs_size`, `max_fbs_size`) and its result ((a, b) and the merged multi-value table, or None).  Only data is
written (tests/golden/_mapper_search.json.gz); no reference source.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/capture_mapper_search.py
"""
import gzip
import json
import logging
import os
import sys

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "fbs_mapper"))
sys.path.insert(0, os.path.join(REF, "experiments"))

import numpy as np  # noqa: E402

_argv = sys.argv
sys.argv = ["capture"]
import bit_exec_env  # noqa: E402
import map_to_fbs  # noqa: E402
import generate_benchmarks as gb  # noqa: E402
sys.argv = _argv

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from capture_reference import array_multiplier, gen_from_reference, ripple_adder  # noqa: E402

logging.disable(logging.CRITICAL)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_mapper_search.json.gz")

records, seen = [], set()
orig = map_to_fbs.MapToFBSHeur._find_lincomb_coefs_search


def spy(self, xy_mvt, r_tt):
    ab, mvt = orig(self, xy_mvt, r_tt)
    key = (self.fbs_size, self.max_fbs_size, np.asarray(xy_mvt).tobytes(), np.asarray(r_tt).tobytes())
    if key not in seen:
        seen.add(key)
        records.append(dict(fbs_size=int(self.fbs_size), max_fbs_size=int(self.max_fbs_size),
                            x=[int(v) for v in np.asarray(xy_mvt)[:, 0]], y=[int(v) for v in np.asarray(xy_mvt)[:, 1]],
                            tt=[int(v) for v in np.asarray(r_tt)],
                            ab=None if ab is None else [int(ab[0]), int(ab[1])],
                            mvt=None if mvt is None else [int(v) for v in np.asarray(mvt)]))
    return ab, mvt


map_to_fbs.MapToFBSHeur._find_lincomb_coefs_search = spy

circuits = dict(aes_sbox=gen_from_reference(gb.aes_sbox), ascon_lut=gen_from_reference(gb.ascon_lut),
                simon_iter=gen_from_reference(gb.simon_iter), full_adder=gen_from_reference(gb.full_adder_bench),
                two_input_gates=gen_from_reference(gb._2_input_gates), aoi21=gen_from_reference(gb.aoi21_bench),
                trivium_iter_v2=gen_from_reference(gb.TriviumIter.trivium_iter_v2),
                kreyvium_iter_v1=gen_from_reference(gb.KreyviumIter.kreyvium_iter_v1),
                adder8=ripple_adder(8), mul4=array_multiplier(4))
for name, build in circuits.items():
    for p in (2, 3, 7, 15, 31):
        for strict in (False, True):
            env = build()
            mapper = map_to_fbs.MapToFBSHeur(fbs_size=p, max_fbs_size=p if strict else 2 * p, max_truth_table_size=16,
                                             cone_merger="search")
            before = len(records)
            try:
                mapper.map(env)
                note = ""
            except AssertionError:          # the reference itself gives up on some (circuit, p, strict) combinations
                note = "  (reference asserted)"
            print("%-18s p=%2d strict=%d  +%d cases%s" % (name, p, strict, len(records) - before, note), flush=True)

# keep the file small: all failing searches and a spread of table sizes, at most ~600 cases
rng = np.random.default_rng(1)
order = rng.permutation(len(records))
keep, per_size = [], {}
for i in order:
    r = records[i]
    bucket = (len(r["tt"]), r["fbs_size"], r["ab"] is None)
    if per_size.get(bucket, 0) < 12:
        per_size[bucket] = per_size.get(bucket, 0) + 1
        keep.append(r)
keep.sort(key=lambda r: (r["fbs_size"], r["max_fbs_size"], len(r["tt"])))
with gzip.open(OUT, "wb") as f:
    f.write(json.dumps(dict(source="MapToFBSHeur._find_lincomb_coefs_search, map_to_fbs.py:363-392", cases=keep)).encode())
print("recorded %d distinct calls, kept %d -> %s (%d bytes)" % (len(records), len(keep), OUT, os.path.getsize(OUT)))
