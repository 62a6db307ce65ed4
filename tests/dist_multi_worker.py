"""Worker for tests/test_gpu_multi.py: one rank of an N-rank job with ONE GPU PER RANK on the nccl backend (= RCCL over xGMI on an
MI355X node) -- the deployment shape.  Every layout of distributed.ShardedRunner on a program's device-resident wires; rank 0
also evaluates the program alone (`prog.eval`) and writes both for the parent to compare."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests.helpers import load_fixture, subsample, toy_k2      # noqa: E402


def main():
    name, T, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    flavour = sys.argv[4] if len(sys.argv) > 4 else ""
    local = int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    rank, world = dist.get_rank(), dist.get_world_size()
    from tfhe_fbs_map_amd import Context, Params, Program, parse_fbs
    from tfhe_fbs_map_amd.distributed import GateShardedRunner, GpuBackend, SampleShardedRunner, ShardedRunner
    rec = load_fixture(name)
    low = parse_fbs(rec["fbs"], inputs=rec["program_inputs"]).lower()
    p = max(7, max(len(t) for t in low["tables"]))
    prm = toy_k2(p) if flavour == "k2" else Params(n=12, log_n_poly=10, p_msg=p, sigma_lwe=1 << 8, sigma_glwe=1 << 8)
    if flavour == "k3":                               # GLWE dimension 3 at N = 512: the default shape for p <= 8 (k_blind_rotate_glwe)
        prm = Params(n=12, log_n_poly=9, k=3, l_bsk=1, beta_bsk=18, t_ksk=8, gamma_ksk=2, p_msg=p, sigma_lwe=1 << 8, sigma_glwe=4, bsk_group=2)
    ctx = Context(prm, seed=21, device=local)         # keys replicated: every rank derives them from the seed
    prog = Program(ctx, ctx.tvset(low["tables"]), len(low["input_names"]), low["kind"], low["arg0"], low["arg1"],
                   low["const_coef"], low["term_coef"], low["term_src"], low["out_wire"], fuse_tables=flavour == "fused")
    ins, _ = subsample(rec, T)
    cts = ctx.encrypt(np.stack([ins[n] for n in low["input_names"]]), nonce0=7)
    res = {}
    layouts = [("gate", lambda: GateShardedRunner(GpuBackend(prog))), ("sample", lambda: SampleShardedRunner(GpuBackend(prog)))]
    if world >= 4 and world % 2 == 0:
        layouts.append(("grid", lambda: ShardedRunner(GpuBackend(prog), sample_groups=2)))
    for label, make in layouts:
        runner = make()
        res[label] = runner.run(cts, T)
        res[label + "_collectives"] = runner.collectives
        res[label + "_fbs"] = runner.bootstraps_done
    if rank == 0:
        ref = prog.eval(cts, T)                       # the single-process answer
        np.savez(out_path, ref=ref, dec=ctx.decrypt(ref), world=world, backend=dist.get_backend(), depth=prog.depth,
                 n_bootstrap=prog.n_bootstrap, n_rotations=prog.n_rotations, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
