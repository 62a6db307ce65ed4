"""What would 2, 4 or 8 ranks each have to do?  Measured on ONE GPU, level by level.

This is NOT a scaling curve: no second GPU is involved and no collective runs.  It answers the question that decides one --
how long does rank r's share of every level take on the kernels as they are? -- by running those shares, one after the other,
on the one GPU this build has, through the product's own backend (GpuBackend / fbs_level_bootstrap_dev, device-resident
wires, send buffers, scatter).  Per level and layout (`sample_groups` x `gate_groups`, distributed.ShardedRunner):

    level time = max over the measured ranks of (linear combinations + the rank's slice of the level's bootstraps)
                 + scatter of the gathered rows                     (measured)
                 + all-gather of the slices over xGMI               (MODELLED: distributed.allgather_ms, assumptions printed)

and the projected speed-up is the single-GPU time of the same program (measured in the same process, same inputs) over
the sum of the level times.  Ranks measured: the first and the last of every group (slices differ by at most one chunk
boundary; `--all-ranks` measures every one).  Layouts: gate (1 x G), sample (G x 1) and whatever `choose_sharding` picks.

    python3 tools/scale_emulation.py [--circuits a,b] [--samples 1000,64] [--ranks 2,4,8] [--out profiles/r03/scale_emulation.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

from tests.helpers import load_fixture                                       # noqa: E402
from tfhe_fbs_map_amd import Context, Program, params_for, parse_fbs           # noqa: E402
from tfhe_fbs_map_amd.distributed import GpuBackend, allgather_ms, choose_sharding, split_range   # noqa: E402

CONFIGS = {"adder128__search_p15": "BASELINE configs[0]/[1] stand-in (EPFL adder.blif is fetched from the network by the reference)",
           "mul16__search_p15": "BASELINE configs[2] stand-in (ISCAS85 c6288 = 16x16 multiplier)",
           "trivium_stream_v2__search_p15": "BASELINE configs[3] stand-in (EPFL log2.blif is not available offline)",
           "basic_adder128": "a 128-bit ripple-carry adder lowered gate by gate (p = 3): a small plaintext modulus on the default k = 3 sets (--secure k3)",
           "adder128__search_p31": "BASELINE configs[4] (fbs_size = 31): the 128-bit adder mapped @31; run with --secure (p = 31 at N = 2048, l = 2)"}


class Timer:
    def __init__(self):
        self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        self.a.record()
        return self

    def __exit__(self, *exc):
        self.b.record()
        self.b.synchronize()
        self.ms = self.a.elapsed_time(self.b)


def measure(name, T, ranks_list, all_ranks, secure):
    if name.startswith("basic_adder"):
        # a ripple-carry adder built gate by gate and lowered like the reference's MapToFBSBasic (one linear combination + one table per
        # two-input gate: p = 3) -- no fixture of that size exists; what a small plaintext modulus looks like on several ranks
        from tfhe_fbs_map_amd.fbs_exec_env import min_fbs_size
        from tfhe_fbs_map_amd.netlist import map_basic
        from tools.fusion_bench import ripple_adder
        env = map_basic(ripple_adder(int(name[len("basic_adder"):])))
        low = env.lower()
        p = min_fbs_size(low["tables"])
    else:
        rec = load_fixture(name)
        env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
        low = env.lower()
        p = int(name.rsplit("_p", 1)[-1])
    if secure:
        from tfhe_fbs_map_amd import choose_params
        from tfhe_fbs_map_amd.params import DEFAULT_GLWE_DIMS
        # (--secure k2: GLWE dimension 2 admitted; --secure k3: 2 and 3, what ExecConfig admits for every program -- each ahead of the
        # alternatives at every launch size, so one GPU and every rank run the SAME set; --secure k1: the k = 1 sets only)
        prm = choose_params(p, env.stats()["norm2_linprod"], glwe_dims={"k2": (1, 2), "k3": DEFAULT_GLWE_DIMS}.get(secure, (1,)))
    else:
        prm = params_for(p)
    ctx = Context(prm, seed=1)
    prog = Program(ctx, ctx.tvset(low["tables"]), len(low["input_names"]), low["kind"], low["arg0"], low["arg1"],
                   low["const_coef"], low["term_coef"], low["term_src"], low["out_wire"])
    be = GpuBackend(prog)
    widths = list(prog.level_width)
    ctw = prm.ct_words
    rng = np.random.default_rng(42)
    bits = rng.integers(0, 2, (prog.n_inputs, T))
    cts = torch.from_numpy(ctx.encrypt(bits, nonce0=0).view(np.int64)).cuda()
    wires = be.new_wires(T)
    chunk_max = max(widths) * T
    rows = be.new_rows(chunk_max)
    ctx.reserve(max_keyswitches=chunk_max)
    stream = torch.cuda.current_stream().cuda_stream

    layouts, chosen = {}, {}
    for G in ranks_list:
        pick = choose_sharding(widths, T, G, params=prm)     # priced on the staircase of the set actually loaded
        for label, gs in (("gate", 1), ("sample", G if T >= G else None), ("chosen", pick["sample_groups"])):
            if gs is not None and (label != "chosen" or (gs, G // gs) not in [v for (g, _), v in layouts.items() if g == G]):
                layouts[(G, label)] = (gs, G // gs)
        chosen[G] = (pick["sample_groups"], pick["gate_groups"], pick["predicted_speedup"])
    per_level = {k: [] for k in layouts}
    single = []

    def boot(L, s0, sc, f0, f1, into_rows):
        prog.level_bootstrap_dev(L, wires.data_ptr(), T, s0, sc, f0, f1, d_rows=rows.data_ptr() if into_rows else 0, stream=stream)

    for warm in (True, False):                                 # first pass: warm-up of every launch shape, untimed results dropped
        be.load_inputs(wires, T, cts, T)
        for L in range(prog.depth + 1):
            with Timer() as t_lin:
                be.lincomb_level(wires, T, L, T)
            if L == prog.depth:
                break
            total = widths[L] * T
            with Timer() as t_full:
                boot(L, 0, T, 0, total, False)
            if L % 8 == 0:      # (a long run must keep talking: a silent command is taken for hung after a few minutes)
                print("  %s T=%d %s pass, level %d / %d" % (name, T, "warm-up" if warm else "timed", L, prog.depth), file=sys.stderr, flush=True)
            if not warm:
                single.append(dict(level=L, width=widths[L], bootstraps=total, lincomb_ms=t_lin.ms, bootstrap_ms=t_full.ms))
            for (G, label), (gs, gg) in layouts.items():
                worst, sizes = 0.0, []
                for sg in sorted({0, gs - 1} if not all_ranks else set(range(gs))):
                    s0, s1, _ = split_range(T, gs, sg)
                    sc = s1 - s0
                    for pos in sorted({0, gg - 1} if not all_ranks else set(range(gg))):
                        f0, f1, _ = split_range(widths[L] * sc, gg, pos)
                        sizes.append(f1 - f0)
                        if f1 > f0:
                            with Timer() as t:
                                boot(L, s0, sc, f0, f1, gg > 1)
                            worst = max(worst, t.ms)
                scatter = 0.0
                if gg > 1:
                    s0, s1, _ = split_range(T, gs, 0)
                    n_rows = widths[L] * (s1 - s0)
                    with Timer() as t:
                        prog.level_scatter_dev(L, wires.data_ptr(), T, s0, s1 - s0, rows.data_ptr(), 0, n_rows, stream=stream)
                    scatter = t.ms
                    with Timer():                                 # put the level's real results back (the rows held slices only)
                        boot(L, 0, T, 0, total, False)
                if not warm:
                    slice_rows = max(sizes)
                    per_level[(G, label)].append(dict(level=L, slice=slice_rows, compute_ms=worst + t_lin.ms * (1.0 / gs), scatter_ms=scatter,
                                                      allgather_model_ms=allgather_ms(slice_rows, gg, ctw * 8)))
    from bench import cleartext
    out = ctx.decrypt(be.read_outputs(wires, T, T).cpu().numpy().view(np.uint64))
    clear = cleartext(low, bits)
    ok = all(bool(np.array_equal(out[k], clear[k])) for k, w in enumerate(low["out_wire"]) if w >= 0)
    single_ms = sum(x["lincomb_ms"] + x["bootstrap_ms"] for x in single)
    result = dict(circuit=name, stands_for=CONFIGS.get(name, ""), samples=T, depth=prog.depth, bootstraps=prog.n_bootstrap * T,
                  level_widths=widths, params=dict(n=prm.n, N=prm.N, k=prm.k, l=prm.l_bsk, beta=prm.beta_bsk, key_bits_per_step=prm.bsk_group),
                  outputs_equal_cleartext=ok, choose_sharding={str(g): dict(sample_groups=v[0], gate_groups=v[1], predicted_speedup=v[2])
                                                                for g, v in chosen.items()},
                  single_gpu_ms=single_ms, single_gpu_fbs_per_s=prog.n_bootstrap * T / single_ms * 1e3,
                  depth_times_latency_floor_ms=prog.depth * min(x["bootstrap_ms"] for x in single) if single else 0.0, layouts=[])
    for (G, label), (gs, gg) in layouts.items():
        lv = per_level[(G, label)]
        total = sum(x["compute_ms"] + x["scatter_ms"] + x["allgather_model_ms"] for x in lv)
        hist = {}
        for x in lv:
            b = "<=256" if x["slice"] <= 256 else "<=512" if x["slice"] <= 512 else "<=1024" if x["slice"] <= 1024 else ">1024"
            hist[b] = hist.get(b, 0) + 1
        result["layouts"].append(dict(ranks=G, layout=label, sample_groups=gs, gate_groups=gg, projected_ms=total,
                                      projected_speedup=single_ms / total if total else None,
                                      compute_ms=sum(x["compute_ms"] for x in lv), scatter_ms=sum(x["scatter_ms"] for x in lv),
                                      allgather_model_ms=sum(x["allgather_model_ms"] for x in lv), collectives=len(lv) if gg > 1 else 0,
                                      slice_size_histogram=hist, levels=lv))
    ctx.close()
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--circuits", default=",".join(CONFIGS))
    ap.add_argument("--samples", default="1000,64")
    ap.add_argument("--ranks", default="2,4,8")
    ap.add_argument("--all-ranks", action="store_true")
    ap.add_argument("--secure", nargs="?", const="k1", default=None, choices=["k1", "k2", "k3"],
                    help="the 128-bit set chosen for each program instead of the reduced-noise benchmark set (k2: GLWE dimension 2 admitted)")
    ap.add_argument("--skip", default="trivium_stream_v2__search_p15:1000", help="circuit:samples pairs to leave out (minutes of GPU each)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "scale_emulation.json"))
    args = ap.parse_args()
    skip = {tuple(x.split(":")) for x in args.skip.split(",") if x}
    runs = []
    t0 = time.time()
    for name in args.circuits.split(","):
        for T in (int(v) for v in args.samples.split(",")):
            if (name, str(T)) in skip:
                continue
            r = measure(name, T, [int(v) for v in args.ranks.split(",")], args.all_ranks, args.secure)
            runs.append(r)
            best = {}
            for lay in r["layouts"]:
                best.setdefault(lay["ranks"], []).append("%s %.2fx" % (lay["layout"], lay["projected_speedup"]))
            print("%-32s T=%-5d 1 GPU %9.1f ms (%6.0f FBS/s)  " % (name, T, r["single_gpu_ms"], r["single_gpu_fbs_per_s"]) +
                  "  ".join("G=%d: %s" % (g, ", ".join(v)) for g, v in sorted(best.items())) + "  [%.0f s]" % (time.time() - t0), flush=True)
    doc = dict(what="per-rank work of 2/4/8-way sharding measured on ONE MI355X, level by level (tools/scale_emulation.py); "
                    "a 1-GPU measurement of per-rank compute plus a MODELLED all-gather -- not a scaling curve",
               allgather_model="distributed.allgather_ms: one slice per point-to-point xGMI link (153 GB/s x 0.8), 30 us per collective",
               device=torch.cuda.get_device_name(0), runs=runs)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
