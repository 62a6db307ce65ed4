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


class OracleBackend:
    def __init__(self, o, tables):
        self.o, self.tables, self.ctw = o, tables, o.ctw
        self.calls = []

    def new_wires(self, n_wires, T):
        return torch.zeros((n_wires * T, self.ctw), dtype=torch.int64)

    def upload(self, wires, row0, cts):
        flat = torch.from_numpy(np.ascontiguousarray(cts, np.uint64).reshape(-1, self.ctw).view(np.int64))
        wires[row0:row0 + flat.shape[0]].copy_(flat)

    def download(self, wires, rows):
        return wires[rows].numpy().view(np.uint64)

    def lincomb(self, wires, T, st):
        w = wires.numpy().view(np.uint64)
        for g, dst in enumerate(st["dst"]):
            terms = range(st["term_off"][g], st["term_off"][g + 1])
            for s in range(T):
                w[dst * T + s] = self.o.lincomb([w[st["srcs"][t] * T + s] for t in terms],
                                                [st["coefs"][t] for t in terms], st["consts"][g])

    def bootstrap(self, wires, T, src, dst, table, s_begin, s_end):
        w = wires.numpy().view(np.uint64)
        self.calls.append((len(src), s_begin, s_end))
        for g in range(len(src)):
            rows = slice(src[g] * T + s_begin, src[g] * T + s_end)
            out, _ = self.o.bootstrap_batch(w[rows], [self.tables[table[g]]], None, threads=1)
            w[dst[g] * T + s_begin:dst[g] * T + s_end] = out


def main():
    name, T, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from tfhe_fbs_map_amd import Params, parse_fbs
    from tfhe_fbs_map_amd.distributed import GateShardedRunner, SampleShardedRunner
    rec = load_fixture(name)
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
    low = env.lower()
    p = max(7, max(len(t) for t in low["tables"]))
    prm = Params(n=8, log_n_poly=8, p_msg=p, sigma_lwe=1 << 8, sigma_glwe=1 << 8)
    o = orc.Oracle(prm, seed=21)
    ins, expect = subsample(rec, T)
    cts = np.stack([o.encrypt(ins[n], nonce0=100 * i) for i, n in enumerate(low["input_names"])])
    res = {}
    for mode, cls in (("gate", GateShardedRunner), ("sample", SampleShardedRunner)):
        be = OracleBackend(o, low["tables"])
        runner = cls(low, be)
        out = runner.run(cts, T)
        res[mode] = out
        res[mode + "_collectives"] = runner.collectives
        res[mode + "_fbs_done"] = sum(g * (b - a) for g, a, b in be.calls)
    if rank == 0:
        # single-process answer by the plain instruction-by-instruction evaluation
        ops, outs = lut_oracle.read_fbs(rec["fbs"])
        wires = oracle_eval_program(o, ops, outs, {n: cts[i] for i, n in enumerate(low["input_names"])})
        ref = np.stack([wires[src] if src not in ("0", "1") else np.zeros((T, o.ctw), np.uint64) for _, src in outs])
        dec = np.stack([o.decrypt(res["gate"][k]) for k in range(len(outs))])
        np.savez(out_path, gate=res["gate"], sample=res["sample"], ref=ref, dec=dec,
                 gate_collectives=res["gate_collectives"], sample_collectives=res["sample_collectives"],
                 gate_fbs=res["gate_fbs_done"], sample_fbs=res["sample_fbs_done"], world=world)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
