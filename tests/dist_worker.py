"""Worker for tests/test_distributed_cpu.py: one gloo rank.  The level arithmetic is done by the CPU oracle
(test infrastructure) so that the partitioning / all-gather logic of tfhe_fbs_map_amd.distributed can be
checked without GPUs; on GPUs the same runners drive `GpuBackend`."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import lut_oracle, tfhe_oracle as orc      # noqa: E402
from tests.helpers import load_fixture, oracle_eval_program, subsample      # noqa: E402
from tfhe_fbs_map_amd.distributed import plan_levels      # noqa: E402


class OracleBackend:
    """The backend interface of tfhe_fbs_map_amd.distributed on host memory: wires are [n_wires * T, ctw] (slot = wire
    id, no reuse), the arithmetic is the CPU oracle's."""

    def __init__(self, o, low):
        self.o, self.tables, self.ctw = o, low["tables"], o.ctw
        self.plan = plan_levels(low)
        self.low = low
        self.depth = self.plan["depth"]
        self.level_width = [len(b["src"]) for b in self.plan["boot"]]
        self.n_inputs, self.n_outputs = self.plan["n_inputs"], len(low["out_wire"])
        self.calls = []

    def new_wires(self, T):
        return torch.zeros((self.plan["n_wires"] * max(1, T), self.ctw), dtype=torch.int64)

    def new_rows(self, rows):
        return torch.zeros((max(1, rows), self.ctw), dtype=torch.int64)

    def load_inputs(self, wires, T, in_cts, s_count):
        src = np.ascontiguousarray(in_cts, np.uint64).reshape(self.n_inputs, s_count, self.ctw)
        w = wires.numpy().view(np.uint64)
        for i in range(self.n_inputs):
            w[i * T:i * T + s_count] = src[i]

    def lincomb_level(self, wires, T, L, s_count):
        w = wires.numpy().view(np.uint64)
        for st in self.plan["lin"][L]:
            for g, dst in enumerate(st["dst"]):
                terms = range(st["term_off"][g], st["term_off"][g + 1])
                for s in range(s_count):
                    w[dst * T + s] = self.o.lincomb([w[st["srcs"][t] * T + s] for t in terms],
                                                    [st["coefs"][t] for t in terms], st["consts"][g])

    def bootstrap_level(self, wires, T, L, s_count, f0, f1, rows=None):
        w = wires.numpy().view(np.uint64)
        b = self.plan["boot"][L]
        self.calls.append(f1 - f0)
        r = None if rows is None else rows.numpy().view(np.uint64)
        for f in range(f0, f1):
            g, s = divmod(f, s_count)
            out, _ = self.o.bootstrap_batch(w[b["src"][g] * T + s][None], [self.tables[b["table"][g]]], None, threads=1)
            if r is None:
                w[b["dst"][g] * T + s] = out[0]
            else:
                r[f - f0] = out[0]

    def scatter_level(self, wires, T, L, s_count, rows, f0, f1):
        w = wires.numpy().view(np.uint64)
        r = rows.numpy().view(np.uint64)
        b = self.plan["boot"][L]
        for f in range(f0, f1):
            g, s = divmod(f, s_count)
            w[b["dst"][g] * T + s] = r[f - f0]

    def read_outputs(self, wires, T, s_count):
        out = torch.zeros((self.n_outputs, s_count, self.ctw), dtype=torch.int64)
        for k, wire in enumerate(self.low["out_wire"]):
            if wire >= 0:
                out[k] = wires[wire * T:wire * T + s_count]
        return out


def main():
    name, T, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from tfhe_fbs_map_amd import Params, parse_fbs
    from tfhe_fbs_map_amd.distributed import GateShardedRunner, SampleShardedRunner, ShardedRunner
    rec = load_fixture(name)
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
    low = env.lower()
    p = max(7, max(len(t) for t in low["tables"]))
    prm = Params(n=8, log_n_poly=8, p_msg=p, sigma_lwe=1 << 8, sigma_glwe=1 << 8)
    o = orc.Oracle(prm, seed=21)
    ins, expect = subsample(rec, T)
    cts = np.stack([o.encrypt(ins[n], nonce0=100 * i) for i, n in enumerate(low["input_names"])])
    res = {}
    modes = [("gate", GateShardedRunner), ("sample", SampleShardedRunner)]
    if world == 4:                                        # two sample groups of two gate-sharding ranks each
        modes.append(("grid", lambda be: ShardedRunner(be, sample_groups=2)))
    for mode, cls in modes:
        be = OracleBackend(o, low)
        runner = cls(be)
        out = runner.run(cts, T)
        res[mode] = out
        res[mode + "_collectives"] = runner.collectives
        res[mode + "_fbs_done"] = sum(be.calls)
    if rank == 0:
        # single-process answer by the plain instruction-by-instruction evaluation
        ops, outs = lut_oracle.read_fbs(rec["fbs"])
        wires = oracle_eval_program(o, ops, outs, {n: cts[i] for i, n in enumerate(low["input_names"])})
        ref = np.stack([wires[src] if src not in ("0", "1") else np.zeros((T, o.ctw), np.uint64) for _, src in outs])
        dec = np.stack([o.decrypt(res["gate"][k]) for k in range(len(outs))])
        np.savez(out_path, gate=res["gate"], sample=res["sample"], ref=ref, dec=dec,
                 gate_collectives=res["gate_collectives"], sample_collectives=res["sample_collectives"],
                 gate_fbs=res["gate_fbs_done"], sample_fbs=res["sample_fbs_done"], world=world,
                 **({"grid": res["grid"], "grid_collectives": res["grid_collectives"], "grid_fbs": res["grid_fbs_done"]} if world == 4 else {}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
