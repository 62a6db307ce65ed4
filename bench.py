#!/usr/bin/env python3
"""Headline benchmark: functional bootstraps per second (batched), N=1024 -- BASELINE.json configs[1].

A "step" is one pass of the hot path (key switch + modulus switch + blind rotation + sample extraction)
over one synthetic batch of `--batch` independent ciphertexts per GPU, inputs and outputs resident in
HBM.  N > 1: one process per GPU (torch.distributed.run), the batch is per rank (weak scaling), keys are
replicated, no data-path collective; time is max over ranks between barriers.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel: blind rotation,
timed live with HIP events on its own stream through libfbsexec's profile hooks) and `cpu_baseline`
(the CPU oracle on a bounded sample of the same batch, all host cores; N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8.0e12      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="independent FBS per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="FBS timed on the host CPU (-1: 64 per thread, 0: skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from tfhe_fbs_map_amd import Context, Params
    prm = Params()                                   # P1024, p = 15
    ctx = Context(prm, seed=1, device=local)         # keys replicated: every rank derives them from the seed
    B = args.batch
    rng = np.random.default_rng(42)
    tables = [[0] + [int(v) for v in rng.integers(0, 2, 14)] for _ in range(16)]
    tv = ctx.tvset(tables)
    rng = np.random.default_rng(42 + rank)
    msgs = rng.integers(0, 15, B)
    ids = (np.arange(B) % 16).astype(np.uint32)
    cts = ctx.encrypt(msgs, nonce0=rank * B)
    d_in = torch.from_numpy(cts.view(np.int64)).cuda()
    d_ids = torch.from_numpy(ids.view(np.int32)).cuda()
    d_out = torch.empty_like(d_in)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr(), stream)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.profile(True)
    ctx.profile_read(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_read(reset=True)
    ctx.profile(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = d_out.cpu().numpy().view(np.uint64)
    ok = bool(np.array_equal(ctx.decrypt(out), [tables[i][m] for i, m in zip(ids, msgs)]))
    if dist is not None:
        flag = torch.tensor([1 if ok else 0], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())

    if rank == 0:
        value = world * B * args.steps / elapsed
        br_ms = prof["blind_rotate"]["ms"] / max(1, prof["blind_rotate"]["launches"])
        ks_ms = prof["keyswitch"]["ms"] / max(1, prof["keyswitch"]["launches"])
        N, n, k = prm.N, prm.n, prm.k
        # algorithmic bytes of ONE blind-rotation launch: per FBS every bootstrapping-key row once, the test
        # vector, the mod-switched input and the extracted output (DESIGN.md "Bytes"); SURVEY 8(d)'s whole-FBS
        # figure (103 309 328 B) additionally holds the key-switching key, which is the other kernel's.
        br_bytes_per_fbs = n * (k + 1) * prm.l_bsk * (k + 1) * N * 8 + N * 8 + (n + 1) * 4 + (k * N + 1) * 8
        achieved = br_bytes_per_fbs * B / (br_ms * 1e-3)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_blind_rotate.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        roofline = dict(bound="hbm", kernel="k_blind_rotate<10,6,true,1>", achieved=achieved / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s",
                        frac=achieved / HBM_PEAK, traffic=traffic, avg_launch_ms=br_ms, bytes_per_unit=br_bytes_per_fbs,
                        units_per_launch=B, keyswitch_avg_launch_ms=ks_ms,
                        whole_path_bytes_per_fbs=prm.bytes_per_fbs(),
                        whole_path_frac=value / world * prm.bytes_per_fbs() / HBM_PEAK,
                        note="achieved/frac are ALGORITHMIC bytes over time; the key stream is served from L2/MALL after first "
                             "touch (traffic = PMC-measured fabric bytes per launch) and the kernel is bound by FP64 issue: "
                             "4.8e9 VALU wave-instructions per launch (PMC) against a measured ceiling of one v_fma_f64 per SIMD per "
                             "4.8-5.0 nominal cycles at two waves per SIMD (profiles/r01/fp64_issue_rate.txt) = 9.7 ms; see DESIGN.md")
        result = dict(metric="functional bootstraps/sec (batched), N=1024", value=value, unit="FBS/s", n_gpus=world,
                      steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3,
                      higher_is_better=True, scaling="weak", vs_baseline=None,
                      dtype="f64 (exact integer arithmetic mod a 46-bit prime via FMA; residues in 64-bit words)",
                      data="synthetic",
                      config=dict(workload="BASELINE configs[1]: %d independent FBS per GPU per step, P1024 "
                                           "(n=630 N=1024 k=1 l=3 beta=7 t=8 gamma=2), p=15, 16 random tables" % B,
                                  batch_per_gpu=B, parallelism="replicas of the batch per GPU, keys replicated, no collective",
                                  device=ctx.device_info),
                      decrypt_ok=ok, roofline=roofline)
        if world == 1 and args.cpu_sample != 0:
            result["cpu_baseline"] = cpu_baseline(prm, tables, cts, ids, out, args.cpu_sample)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(prm, tables, cts, ids, gpu_out, sample):
    """The oracle (a plain-C port, OpenMP over independent FBS) on the first `sample` ciphertexts of the
    same batch, same keys; also checks those ciphertexts bit-for-bit against the GPU's."""
    from oracle import tfhe_oracle
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)          # a one-GPU box is a 16-core share of the host; oversubscribing it only thrashes
    if sample < 0:
        sample = min(len(cts), 64 * cores)
    sample = max(1, min(sample, len(cts)))
    orc = tfhe_oracle.Oracle(prm, seed=1)
    t0 = time.perf_counter()
    ref, used = orc.bootstrap_batch(cts[:sample], tables, ids[:sample], threads=cores)
    dt = time.perf_counter() - t0
    return dict(value=sample / dt, unit="FBS/s", cores=used, kind="port",
                sample="first %d ciphertexts of the timed batch, %.1f s" % (sample, dt),
                bit_exact_vs_gpu=bool(np.array_equal(ref, gpu_out[:sample])))


if __name__ == "__main__":
    main()
