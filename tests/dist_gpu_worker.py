"""Worker for tests/test_gpu_distributed.py: one rank of a 2-rank job whose ranks SHARE the box's single GPU.  The product
data path is complete -- libfbsexec kernels on device-resident wires through GpuBackend, slices of every level bootstrapped
into send buffers, gathered, scattered -- only the transport is gloo instead of RCCL (which refuses two ranks on one
device)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.helpers import load_fixture, subsample      # noqa: E402


def main():
    name, T, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    fused = len(sys.argv) > 4 and sys.argv[4] == "fused"
    k2 = len(sys.argv) > 4 and sys.argv[4] in ("k2", "fused_k2")   # GLWE dimension 2 at N = 1024: ciphertexts and rows of 2 N + 1 words
    fused = fused or (len(sys.argv) > 4 and sys.argv[4] in ("fused_k2", "fused_k3"))   # ... with shared rotations: rows of (k + 1) N words
    k3 = len(sys.argv) > 4 and sys.argv[4] in ("k3", "fused_k3")   # GLWE dimension 3 at N = 512 (k_blind_rotate_glwe): 3 N + 1 words, rows of 4 N
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from tfhe_fbs_map_amd import Context, FbsError, Params, Program, parse_fbs
    from tfhe_fbs_map_amd.distributed import GateShardedRunner, GpuBackend, SampleShardedRunner
    rec = load_fixture(name)
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
    low = env.lower()
    p = max(7, max(len(t) for t in low["tables"]))
    prm = Params(n=12, log_n_poly=10, p_msg=p, sigma_lwe=1 << 8, sigma_glwe=1 << 8)
    if k2:
        from tests.helpers import toy_k2
        prm = toy_k2(p)
    if k3:
        prm = Params(n=12, log_n_poly=9, k=3, l_bsk=1, beta_bsk=18, t_ksk=8, gamma_ksk=2, p_msg=p, sigma_lwe=1 << 8, sigma_glwe=4, bsk_group=2)
    ctx = Context(prm, seed=21)                       # keys replicated: every rank derives them from the seed
    prog = Program(ctx, ctx.tvset(low["tables"]), len(low["input_names"]), low["kind"], low["arg0"], low["arg1"],
                   low["const_coef"], low["term_coef"], low["term_src"], low["out_wire"], fuse_tables=fused)
    ins, expect = subsample(rec, T)
    cts = ctx.encrypt(np.stack([ins[n] for n in low["input_names"]]), nonce0=7)
    res = {}
    for mode, cls in (("gate", GateShardedRunner), ("sample", SampleShardedRunner)):
        runner = cls(GpuBackend(prog))
        if fused:
            assert prog.n_rotations < prog.n_bootstrap and prog.row_words == (prm.k + 1) * prm.N
        res[mode] = runner.run(cts, T)
        res[mode + "_collectives"] = runner.collectives
        res[mode + "_fbs"] = runner.bootstraps_done
    if rank == 0:
        ref = prog.eval(cts, T)                       # the single-process answer
        np.savez(out_path, gate=res["gate"], sample=res["sample"], ref=ref, dec=ctx.decrypt(res["gate"]) if res["gate"].size else np.zeros(0),
                 dec_sample=ctx.decrypt(res["sample"]), world=world,
                 gate_collectives=res["gate_collectives"], sample_collectives=res["sample_collectives"],
                 gate_fbs=res["gate_fbs"], sample_fbs=res["sample_fbs"], depth=prog.depth, n_bootstrap=prog.n_bootstrap)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
