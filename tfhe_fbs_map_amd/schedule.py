"""Scheduling arithmetic of FBS programs that needs neither a GPU nor torch: the level plan of a lowered program (the rule of
fbs_program_load, csrc/fbs_capi.cpp), how a range is cut over ranks, the measured launch-time staircases of the blind-rotation
kernels, and `choose_sharding`, which lays ranks over a program's two independent axes (fbs_mapper/fbs_exec_env.py:211-223: gates
within a bootstrap level, and the sample axis).  `distributed.py` (the only module that imports torch) re-exports these names;
`fbs_exec_env.ExecConfig.choose` uses them to decide a program's parameter set."""
from __future__ import annotations

import numpy as np


# --------------------------------------------------------------------------------------------
# schedule (same rule as fbs_program_load in csrc/fbs_capi.cpp); used by tests and CPU backends
# --------------------------------------------------------------------------------------------
def plan_levels(low):
    """`low` = LutExecEnv.lower().  Returns dict(depth, n_wires, lin=[[stage,...] per level], boot=[stage per level])
    where a lincomb stage is dict(dst, term_off, srcs, coefs, consts) and a boot stage dict(src, dst, table),
    its gates sorted by source wire (gates that share a source share its key switch)."""
    n_in = len(low["input_names"])
    n_wires = n_in + len(low["kind"])
    level = [0] * n_wires
    sub = [0] * n_wires
    is_lin = [False] * n_wires
    for i, kind in enumerate(low["kind"]):
        w = n_in + i
        if kind == 0:
            is_lin[w] = True
            srcs = low["term_src"][low["arg0"][i]:low["arg0"][i] + low["arg1"][i]]
            level[w] = max((level[s] for s in srcs), default=0)
            sub[w] = max((sub[s] + 1 for s in srcs if is_lin[s] and level[s] == level[w]), default=0)
        else:
            level[w] = level[low["arg0"][i]] + 1
    depth = max((level[n_in + i] for i, k in enumerate(low["kind"]) if k == 1), default=0)
    lin = [dict() for _ in range(depth + 1)]
    gates = [[] for _ in range(depth)]
    for i, kind in enumerate(low["kind"]):
        w = n_in + i
        if kind == 0:
            st = lin[level[w]].setdefault(sub[w], dict(dst=[], term_off=[0], srcs=[], coefs=[], consts=[]))
            a, c = low["arg0"][i], low["arg1"][i]
            st["dst"].append(w)
            st["srcs"] += low["term_src"][a:a + c]
            st["coefs"] += low["term_coef"][a:a + c]
            st["term_off"].append(len(st["srcs"]))
            st["consts"].append(low["const_coef"][i])
        else:
            gates[level[w] - 1].append((low["arg0"][i], w, low["arg1"][i]))
    boot = []
    for g in gates:
        g.sort(key=lambda x: x[0])              # stable: program order within one source
        boot.append(dict(src=[x[0] for x in g], dst=[x[1] for x in g], table=[x[2] for x in g]))
    return dict(depth=depth, n_wires=n_wires, n_inputs=n_in,
                lin=[[d[k] for k in sorted(d)] for d in lin], boot=boot)


def split_range(total, parts, r):
    """Contiguous slice r of `parts` near-equal slices of range(total)."""
    chunk = -(-total // parts)
    return min(total, r * chunk), min(total, (r + 1) * chunk), chunk


# --------------------------------------------------------------------------------------------
# how to cut a program over G ranks
# --------------------------------------------------------------------------------------------
# One key switch + blind rotation launch of `count` bootstraps on one MI355X at the benchmark shape P1024, in ms
# (profiles/r03/batch_sweep.txt, the buffer-load kernels).  Up to one bootstrap per CU a launch costs the latency of one bootstrap on a whole CU
# (k_blind_rotate_cu), up to two that of two workgroups sharing a CU, then the small workgroups, then whole rounds of four per
# CU; beyond a round, rounds + remainder.
LAUNCH_MS_P1024 = ((1, 2.91), (128, 2.92), (256, 3.14), (257, 5.46), (512, 5.53), (513, 8.21), (768, 8.21), (769, 9.29), (1024, 9.29))
ROUND_MS_P1024 = 9.20      # per further round of 1024 in a long launch (8192 bootstraps: 73.8 ms)


def launch_ms(count, cost=1.0):
    """Modelled time of one bootstrap launch of `count` ciphertexts; `cost` = params.bootstrap_cost of the parameter set."""
    if count <= 0:
        return 0.0
    rounds, rest = divmod(int(count), 1024)
    if rounds and rest >= 896:
        rounds, rest = rounds + 1, 0
    ms = rounds * ROUND_MS_P1024 + (0.2 if rounds == 1 and not rest else 0.0)
    if rest:
        xs, ys = zip(*LAUNCH_MS_P1024)
        ms += float(np.interp(rest, xs, ys))
    return ms * cost


def allgather_ms(rows_per_rank, ranks, ct_bytes=8200, link_GBps=153.0, efficiency=0.8, latency_us=30.0):
    """One all-gather of `rows_per_rank` ciphertexts from each of `ranks` GPUs over xGMI: every GPU receives ranks - 1 slices,
    each over its own point-to-point link (at most 7 per GPU), so the time is one slice over one link, plus a fixed latency.
    A model with its assumptions in the signature -- nothing here has been timed on more than one GPU."""
    if ranks <= 1:
        return 0.0
    per_link = rows_per_rank * ct_bytes * -(-(ranks - 1) // min(ranks - 1, 7))
    return latency_us * 1e-3 + per_link / (link_GBps * 1e9 * efficiency) * 1e3


def choose_sharding(level_width, T, world, cost=1.0, ct_bytes=8200):
    """How to lay `world` ranks over a program's two independent axes (fbs_mapper/fbs_exec_env.py:211-223): `sample_groups`
    groups that each take a slice of the T samples through the whole program (no communication), times `gate_groups` ranks
    per group that cut every level's (gate, sample) batch among themselves (one all-gather per level).

    Per level every rank ends up with about width * T / world bootstraps whichever way the cut goes, so what decides is
    (i) whether there are samples enough to cut (T < world forces gate groups), (ii) the collectives gate groups pay, and
    (iii) how the slices fall on the launch-time staircase (`launch_ms`: a slice of 257 bootstraps costs two rounds of the
    one-bootstrap-per-CU kernel, 256 cost one).  All divisor pairs of `world` are priced; ties go to fewer collectives.
    -> dict(sample_groups, gate_groups, predicted_ms, single_gpu_ms, candidates)."""
    level_width = [int(w) for w in level_width]
    cands = []
    for gs in range(1, world + 1):
        if world % gs or gs > max(1, T):
            continue
        gg = world // gs
        samples = -(-T // gs)
        compute = sum(launch_ms(-(-w * samples // gg), cost) for w in level_width)
        comm = sum(allgather_ms(-(-w * samples // gg), gg, ct_bytes) for w in level_width) if gg > 1 else 0.0
        cands.append(dict(sample_groups=gs, gate_groups=gg, compute_ms=compute, allgather_ms=comm, predicted_ms=compute + comm))
    best = min(cands, key=lambda c: (round(c["predicted_ms"], 6), c["gate_groups"]))
    single = sum(launch_ms(w * T, cost) for w in level_width)
    return dict(sample_groups=best["sample_groups"], gate_groups=best["gate_groups"], predicted_ms=best["predicted_ms"],
                single_gpu_ms=single, predicted_speedup=single / best["predicted_ms"] if best["predicted_ms"] else 1.0, candidates=cands)


