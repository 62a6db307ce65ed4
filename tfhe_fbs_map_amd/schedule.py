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
# One key switch + blind rotation launch of `count` bootstraps on one MI355X, in ms, per kernel FAMILY: which kernels a parameter
# set runs on decides the shape of its staircase, the number of blind-rotation steps scales it.  Up to one bootstrap per CU a
# launch costs the latency of one bootstrap on a whole CU, then come the shapes for two and three per CU, then whole rounds of
# four per CU; beyond a round, rounds + remainder.  Measured (profiles/r04/launch_staircases.txt; r03/batch_sweep.txt for P1024):
#   "p1024"     k = 1, N = 1024, one key bit per step, three levels, n = 630 (BASELINE's benchmark shape): k_blind_rotate_cu /
#               its lean variant / small workgroups / k_blind_rotate<10,6,3,4>
#   "k2"        k = 2, N = 1024, two key bits per step, n = 734 (the default 128-bit sets for p <= 15 at small norms):
#               k_blind_rotate_cu_k2 for one, two, three rounds of one bootstrap per CU, k_blind_rotate_pairs_k2<10,4> beyond
#   "k3"        k = 3, N = 512, two key bits per step, n = 614 (the default 128-bit sets for p <= 8): k_blind_rotate_glwe<9,4,2,1 / 2 / 3>
#   "n2048"     k = 1, N = 2048, two key bits per step, one level, n = 714 (p = 15 at heavy norms; shared rotations):
#               k_blind_rotate_cu_pairs<11,1> up to one per CU, k_blind_rotate_pairs<11,7,4> beyond
#   "n2048_l2"  the same with two levels, n = 766 (p = 31, BASELINE configs[4]): k_blind_rotate_cu_pairs<11,2>, round after round
LAUNCH_MS_P1024 = ((1, 2.83), (128, 2.89), (256, 3.09), (257, 5.43), (512, 5.52), (513, 8.0), (768, 8.17), (769, 8.31), (896, 8.96), (1024, 9.30))
ROUND_MS_P1024 = 9.20      # per further round of 1024 in a long launch (8192 bootstraps: 73.8 ms)
LAUNCH_FAMILIES = {
    "p1024": dict(steps=630, stairs=LAUNCH_MS_P1024, round_ms=ROUND_MS_P1024, full_from=896),
    "k2": dict(steps=367, stairs=((1, 2.08), (128, 2.21), (256, 2.46), (257, 4.54), (512, 4.55), (513, 6.6), (768, 6.5), (769, 6.87), (896, 7.04),
                                  (1024, 7.39)), round_ms=7.3, full_from=769),
    # k = 3 at N = 512, two key bits per step, one level, n = 614 (the default 128-bit sets for p <= 8 at ordinary norms): k_blind_rotate_glwe with
    # one bootstrap per workgroup up to one per CU, two up to two, three beyond -- a ROUND is 768 bootstraps; longer launches are cut
    "k3": dict(steps=307, stairs=((1, 1.62), (64, 1.55), (128, 1.60), (256, 1.80), (257, 2.52), (512, 2.85), (513, 3.62), (768, 4.00)), round_ms=3.72, full_from=513,
               round=768),
    "n2048": dict(steps=357, stairs=((1, 2.52), (128, 2.59), (256, 2.86), (257, 4.69), (512, 4.86), (513, 7.8), (768, 7.4), (769, 9.66), (1024, 9.34)),
                  round_ms=9.0, full_from=769),
    "n2048_l2": dict(steps=383, stairs=((1, 4.10), (128, 4.06), (256, 4.41), (257, 9.0), (512, 8.23), (513, 13.6), (768, 12.2), (769, 18.2), (1024, 16.34)),
                     round_ms=16.2, full_from=769),
}


def launch_family(params):
    """-> (family name, scale): the staircase a parameter set's launches follow and the factor on its times (blind-rotation steps
    over the steps of the set the staircase was measured at).  Shapes without a measured staircase of their own take P1024's,
    scaled by the modelled cost of a bootstrap (params.bootstrap_cost)."""
    steps = params.n // 2 if params.bsk_group == 2 else params.n
    if params.k == 2 and params.N == 1024 and params.bsk_group == 2 and params.l_bsk == 1:
        name = "k2"
    elif params.k == 3 and params.N == 512 and params.bsk_group == 2 and params.l_bsk == 1:
        name = "k3"
    elif params.N == 2048 and params.bsk_group == 2 and params.l_bsk <= 2:
        name = "n2048" if params.l_bsk == 1 else "n2048_l2"
    elif params.N == 1024 and params.bsk_group != 2 and params.l_bsk == 3:
        name = "p1024"
    else:
        from .params import bootstrap_cost
        return "p1024", bootstrap_cost(params)
    return name, steps / LAUNCH_FAMILIES[name]["steps"]


def launch_ms(count, cost=1.0, params=None):
    """Modelled time of one bootstrap launch of `count` ciphertexts.  `params`: the parameter set (its own kernel family's
    staircase, `launch_family`); without it the P1024 staircase times `cost` = params.bootstrap_cost of the set."""
    if count <= 0:
        return 0.0
    name, scale = launch_family(params) if params is not None else ("p1024", cost)
    fam = LAUNCH_FAMILIES[name]
    rounds, rest = divmod(int(count), fam.get("round", 1024))     # (a round: the bootstraps the throughput shape holds on the whole chip)
    if rounds and rest >= fam["full_from"]:
        rounds, rest = rounds + 1, 0
    ms = rounds * fam["round_ms"] + (0.1 if rounds == 1 and not rest else 0.0)   # (the first round of a launch: fill and drain)
    if rest:
        xs, ys = zip(*fam["stairs"])
        ms += float(np.interp(rest, xs, ys))
    return ms * scale


def allgather_ms(rows_per_rank, ranks, ct_bytes=8200, link_GBps=153.0, efficiency=0.8, latency_us=30.0):
    """One all-gather of `rows_per_rank` ciphertexts from each of `ranks` GPUs over xGMI: every GPU receives ranks - 1 slices,
    each over its own point-to-point link (at most 7 per GPU), so the time is one slice over one link, plus a fixed latency.
    A model with its assumptions in the signature -- nothing here has been timed on more than one GPU."""
    if ranks <= 1:
        return 0.0
    per_link = rows_per_rank * ct_bytes * -(-(ranks - 1) // min(ranks - 1, 7))
    return latency_us * 1e-3 + per_link / (link_GBps * 1e9 * efficiency) * 1e3


def choose_sharding(level_width, T, world, cost=1.0, ct_bytes=None, params=None):
    """How to lay `world` ranks over a program's two independent axes (fbs_mapper/fbs_exec_env.py:211-223): `sample_groups`
    groups that each take a slice of the T samples through the whole program (no communication), times `gate_groups` ranks
    per group that cut every level's (gate, sample) batch among themselves (one all-gather per level).

    Per level every rank ends up with about width * T / world bootstraps whichever way the cut goes, so what decides is
    (i) whether there are samples enough to cut (T < world forces gate groups), (ii) the collectives gate groups pay, and
    (iii) how the slices fall on the launch-time staircase (`launch_ms`: a slice of 257 bootstraps costs two rounds of the
    one-bootstrap-per-CU kernel, 256 cost one) -- the staircase of the parameter set actually loaded when `params` is given
    (its kernel family's: `launch_family`), P1024's times `cost` otherwise.  All divisor pairs of `world` are priced; ties go
    to fewer collectives.
    -> dict(sample_groups, gate_groups, predicted_ms, single_gpu_ms, candidates)."""
    level_width = [int(w) for w in level_width]
    if ct_bytes is None:
        ct_bytes = 8200 if params is None else 8 * (params.k * params.N + 1)
    cands = []
    for gs in range(1, world + 1):
        if world % gs or gs > max(1, T):
            continue
        gg = world // gs
        samples = -(-T // gs)
        compute = sum(launch_ms(-(-w * samples // gg), cost, params) for w in level_width)
        comm = sum(allgather_ms(-(-w * samples // gg), gg, ct_bytes) for w in level_width) if gg > 1 else 0.0
        cands.append(dict(sample_groups=gs, gate_groups=gg, compute_ms=compute, allgather_ms=comm, predicted_ms=compute + comm))
    best = min(cands, key=lambda c: (round(c["predicted_ms"], 6), c["gate_groups"]))
    single = sum(launch_ms(w * T, cost, params) for w in level_width)
    return dict(sample_groups=best["sample_groups"], gate_groups=best["gate_groups"], predicted_ms=best["predicted_ms"],
                single_gpu_ms=single, predicted_speedup=single / best["predicted_ms"] if best["predicted_ms"] else 1.0, candidates=cands)


