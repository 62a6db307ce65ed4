#!/usr/bin/env python3
"""Headline benchmark: functional bootstraps per second (batched), N=1024 -- BASELINE.json configs[1].

    python bench.py [--gpus N] [--steps K] [--warmup W]                      # the headline (workload "batch")
    python bench.py --workload circuit --mode gate|sample [--gpus N] ...     # BASELINE configs[3] stand-in

A "step" is one pass of the hot path (key switch + modulus switch + blind rotation + sample extraction)
  batch:   over one synthetic batch of `--batch` independent ciphertexts per GPU (weak scaling: keys replicated,
           no data-path collective);
  circuit: over one whole mapped program (default: the reference's trivium_stream_v2 mapped @15 by its search
           mapper, tests/golden fixture) on `--samples` samples, level by level --
           mode gate:   every level's (gate, sample) batch cut across the ranks, one RCCL all-gather per level
                        (north_star's shape; strong scaling: the program and T are fixed);
           mode sample: every rank evaluates the whole program on its own `--samples` samples (weak scaling).
Inputs and outputs are resident in HBM.  Time is the max over ranks between barrier + synchronize on both sides.

--gpus N with no WORLD_SIZE in the environment: this process only LAUNCHES -- it starts N ranks with
`python -m torch.distributed.run` as a child process before anything here touches the GPU, relays rank 0's JSON line
and exits with the child's code.  With WORLD_SIZE set (the driver's own torchrun) it is a rank.

Prints ONE JSON line (rank 0): the contract fields; `roofline` for the dominant kernel (blind rotation, timed live with HIP
events on its own stream through libfbsexec's profile hooks: `frac` = ALGORITHMIC work over time over the FP64 issue peak,
`valu_frac` = executed instructions from the offline PMC record of profiles/r03/, refused when taken on other kernel sources);
at N = 1 `cpu_baseline` (the scalar CPU oracle on a bounded sample of the same batch, and `tuned`: the AVX-512 IFMA baseline,
both checked bit for bit against the GPU's ciphertexts), `secure` (the same batch at the 128-bit parameter sets) and
`shared_rotations`; at N > 1 `sharded`: one mapped circuit cut over the ranks three ways (gate groups with one RCCL all-gather
per level, sample groups, and the layout distributed.choose_sharding picks), strong scaling, next to the weak-scaling headline.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # MI355X_MICROARCH.md: 8.0 TB/s spec
CLOCK_HZ = 2.4e9           # nominal engine clock
SIMDS = 256 * 4            # 256 CUs x 4 SIMDs, one FP64 VALU wave-instruction per SIMD per 4 cycles


def _latest_pmc_file():
    """Counters of the timed kernels, collected offline (tools/profile_round.sh + collect_profiles.sh): the newest round's record."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", "pmc_blind_rotate.json")))
    return found[-1] if found else os.path.join(ROOT, "profiles", "r03", "pmc_blind_rotate.json")


PMC_FILE = _latest_pmc_file()
BUTTERFLY_INSTR = 8        # exact 46-bit modular butterfly on the FP64 pipe: 6-instruction product, one add, one subtract
MAC_INSTR = 7              # exact product + lazy accumulate


def pmc_key(kernel, n, units):
    """Key of a kernel's PMC record: the instantiation, the LWE dimension of the parameter set and the bootstraps per launch."""
    return "%s@n=%d@units=%d" % (kernel, n, units)


def kernel_sources_sha256():
    """What the offline PMC record must have been collected on: the kernel sources as they are now."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "tfhe_fbs_map_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".hpp", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(src, name), "rb").read())
    return h.hexdigest()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["batch", "circuit"], default="batch")
    ap.add_argument("--batch", type=int, default=1024, help="batch: independent FBS per GPU per step")
    ap.add_argument("--mode", choices=["gate", "sample", "auto"], default="gate",
                    help="circuit: what is cut across the ranks (auto: distributed.choose_sharding lays them out)")
    ap.add_argument("--no-sharded-legs", action="store_true", help="batch, --gpus > 1: skip the gate- and sample-sharded circuit legs")
    ap.add_argument("--sharded-circuit", default="trivium_stream_v2__search_p15", help="batch, --gpus > 1: the circuit of those legs")
    ap.add_argument("--sharded-samples", type=int, default=64, help="... and its samples")
    ap.add_argument("--sharded-reduced-noise", action="store_true", help="... on the reduced-noise benchmark set instead of the 128-bit set chosen for the circuit")
    ap.add_argument("--circuit", default="trivium_stream_v2__search_p15", help="circuit: fixture under tests/golden")
    ap.add_argument("--samples", type=int, default=64, help="circuit: samples per input (per rank in mode sample)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="FBS timed on the host CPU (-1: 64 per thread, 0: skip)")
    ap.add_argument("--no-secure", action="store_true", help="skip the secure-parameter-set leg")
    ap.add_argument("--oracle-sample", type=int, default=32,
                    help="ciphertexts of every secure leg's timed batch checked word for word against the scalar oracle (0: skip)")
    ap.add_argument("--secure", action="store_true",
                    help="circuit: the 128-bit parameter set choose_params returns for the program's (p, norm2) instead of the "
                         "reduced-noise benchmark set")
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = 20 if args.workload == "batch" else 2
    if args.warmup is None:
        args.warmup = 3 if args.workload == "batch" else 1
    return args


# ---------------------------------------------------------------------------------------------------------------------
def launch(args, argv):
    """Parent of an N-rank run: never touches the GPU (no torch.cuda call, no HIP call)."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    line = None
    for out in child.stdout.splitlines():
        if out.startswith("{") and '"metric"' in out:
            line = out
    if child.returncode != 0 or line is None:
        sys.stderr.write(child.stderr[-4000:])
        errs = [o for o in child.stdout.splitlines() if o.startswith("{") and '"error"' in o]
        sys.stderr.write("\n".join(errs[-args.gpus:]) + "\nbench.py: the %d-rank run failed (exit code %d)\n" % (args.gpus, child.returncode))
        return child.returncode or 1
    print(line, flush=True)
    return 0


# ---------------------------------------------------------------------------------------------------------------------
def worker(args):
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    have_gpu = torch.cuda.is_available()
    dist = None
    # test hook (tests/test_gpu_bench.py): FBS_BENCH_SHARE_GPU=1 lets the ranks of a multi-rank run share the box's GPUs and meet
    # over gloo -- RCCL refuses two ranks on one device -- so that the N > 1 code path of this file runs on a one-GPU box
    share = have_gpu and os.environ.get("FBS_BENCH_SHARE_GPU") == "1"
    if share:
        local = local % torch.cuda.device_count()
    if world > 1:
        import torch.distributed as dist
        if have_gpu and not share:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            if have_gpu:
                torch.cuda.set_device(local)
            dist.init_process_group("gloo")         # (or a GPU-less host: the ranks still meet, then fail at the context)
    elif have_gpu:
        torch.cuda.set_device(local)

    from tfhe_fbs_map_amd import FbsError
    try:
        result = run_batch(args, rank, world, local, dist) if args.workload == "batch" else run_circuit(args, rank, world, local, dist)
    except FbsError as e:
        # no CPU path: a rank without a usable gfx950 device says so and fails
        print(json.dumps(dict(error=str(e), code=e.code, rank=rank, name="FBS_E_DEVICE" if e.code == -2 else "FBS_E_%d" % -e.code)), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return 3
    if args.workload == "batch" and world > 1 and not args.no_sharded_legs:
        sharded = sharded_legs(rank, world, local, dist, args.sharded_circuit, args.sharded_samples, args.sharded_reduced_noise)   # every rank takes part; rank 0 holds the record
        if rank == 0:
            result["sharded"] = sharded
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def fence(dist):
    import torch
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(value, dist):
    import torch
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ok(ok, dist):
    import torch
    if dist is None:
        return ok
    flag = torch.tensor([1 if ok else 0], device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def params_record(prm):
    """What the numbers were taken at: shape, both noises and the security they amount to."""
    import math
    from tfhe_fbs_map_amd import MODULUS
    from tfhe_fbs_map_amd.params import margin_sigmas, security_bits
    return dict(n=prm.n, N=prm.N, k=prm.k, l=prm.l_bsk, beta=prm.beta_bsk, t=prm.t_ksk, gamma=prm.gamma_ksk, p=prm.p_msg,
                key_bits_per_step=prm.bsk_group,
                sigma_lwe=prm.sigma_lwe, sigma_glwe=prm.sigma_glwe,
                log2_sigma_lwe_over_q=round(math.log2(prm.sigma_lwe / MODULUS), 2),
                log2_sigma_glwe_over_q=round(math.log2(prm.sigma_glwe / MODULUS), 2),
                security_bits_estimate=round(security_bits(prm), 1),
                margin_sigmas_at_norm2_1=round(margin_sigmas(prm, 1), 2),
                randomness="test-grade (ChaCha20 streams from a 64-bit benchmark seed, Irwin-Hall(12) noise): the security estimate "
                           "is of the noise LEVELS; see tfhe_fbs_map_amd._native.RANDOMNESS_GRADE")


def timed_batch(ctx, prm, B, rank, steps, warmup, dist, n_tables=16):  # noqa: C901
    """`steps` passes over one resident batch of B ciphertexts; returns (seconds, profile, tables, cts, ids, msgs, out, ok)."""
    import numpy as np
    import torch
    p = prm.p_msg
    rng = np.random.default_rng(42)
    tables = [[0] + [int(v) for v in rng.integers(0, 2, p - 1)] for _ in range(n_tables)]
    tv = ctx.tvset(tables)
    rng = np.random.default_rng(42 + rank)
    msgs = rng.integers(0, p, B)
    ids = (np.arange(B) % n_tables).astype(np.uint32)
    cts = ctx.encrypt(msgs, nonce0=rank * B)
    d_in = torch.from_numpy(cts.view(np.int64)).cuda()
    d_ids = torch.from_numpy(ids.view(np.int32)).cuda()
    d_out = torch.empty_like(d_in)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ctx.bootstrap_batch_dev(tv, d_in.data_ptr(), d_ids.data_ptr(), B, d_out.data_ptr(), stream)

    for _ in range(warmup):
        step()
    ctx.profile(True)
    ctx.profile_read(reset=True)
    fence(dist)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence(dist)
    elapsed = time.perf_counter() - t0
    kernels = ctx.profile_kernels()
    prof = ctx.profile_read(reset=True)
    prof["kernels"] = kernels
    ctx.profile(False)
    out = d_out.cpu().numpy().view(np.uint64)
    ok = bool(np.array_equal(ctx.decrypt(out), [tables[i][m] for i, m in zip(ids, msgs)]))
    return elapsed, prof, tables, cts, ids, msgs, out, ok


def algorithmic_instr_per_bootstrap(prm):
    """per CMUX step (k+1) l forward and k+1 inverse transforms of N/2 log2 N butterflies and (k+1)^2 l N
    pointwise multiply-accumulates (55.3 k modular products at P1024), priced at what this arithmetic costs on the FP64 pipe
    (8 instructions per butterfly, 7 per MAC), in wave-instructions (64 lanes).  Two key bits per step: half the steps, and
    the 3 (k+1)^2 l N bundle products on top."""
    N, k, l = prm.N, prm.k, prm.l_bsk
    butterflies = ((k + 1) * l + (k + 1)) * (N // 2) * prm.log_n_poly
    macs = (k + 1) ** 2 * l * N
    steps = prm.n
    if prm.bsk_group == 2:
        steps = prm.n // 2
        macs += 3 * (k + 1) ** 2 * l * N
    return steps * (butterflies * BUTTERFLY_INSTR + macs * MAC_INSTR) / 64.0


def roofline_record(prm, prof, kernels, B, steps, cus=256):
    """The dominant kernel is the blind rotation.  It is bound by FP64 VALU issue, not by HBM: every workgroup walks the
    bootstrapping key in step, so after the first touch the key comes out of L2 / Infinity Cache (PMC: a few % of the
    algorithmic bytes reach the fabric).
      achieved / frac (= algorithmic_frac): the ALGORITHMIC work (`algorithmic_instr_per_bootstrap`, SURVEY 8(d)(ii) x the
          instruction price of this arithmetic) per launch / the launch time measured live here with HIP events on the
          launch stream / the chip's FP64 issue rate (1024 SIMDs x 2.4 GHz / 4 cycles).
      valu_frac: issue-slot OCCUPANCY -- VALU wave-instructions the kernel actually executed (PMC SQ_INSTS_VALU, collected
          offline with rocprofv3 on this command: profiles/r03/) x 4 cycles over the same time; issue_cycle_frac charges the
          32-bit ones 2 cycles (MI355X_MICROARCH.md: v_fma_f32 2 cycles with two waves on a SIMD) where the record has the mix.
          Both null when the record was collected on other kernel sources than the ones timed (csrc hash).
      key_stream_vs_hbm_peak: SURVEY 8(d)'s contract bytes (every key row once per bootstrap) over the time over 8 TB/s -- NOT
          a utilisation (the key stream does not come from HBM; the figure passes 1 on the whole path); fabric_GBps is what
          reached the fabric (PMC FETCH_SIZE x2 + WRITE_SIZE)."""
    ks = prof["keyswitch"]
    ks_ms = ks["ms"] / max(1, ks["launches"])
    br_all = {k: v for k, v in kernels.items() if v["kind"] == "blind_rotate"}
    if not br_all:
        return None
    name = max(br_all, key=lambda k: br_all[k]["ms"])
    br = br_all[name]
    br_ms = br["ms"] / max(1, br["launches"])
    N, n, k = prm.N, prm.n, prm.k
    # a launch the launcher cuts (whole rounds as four-bootstrap workgroups + the leftovers in another shape) shows as two
    # kernels: each is reported with its own time, and the dominant one is priced on the bootstraps IT ran
    per_round = 4 * cus                    # a round of the whole-CU shape: four bootstraps per CU (cus = fbs_ctx_stat cu_count)
    if len(br_all) == 1:
        units = B
    elif name == "k_blind_rotate<10,6,3,4>":
        units = B - B % per_round
    else:
        units = B % per_round
    peak = SIMDS * CLOCK_HZ / 4.0
    algo = algorithmic_instr_per_bootstrap(prm) * units
    ggsw = (prm.n // 2 * 3 if prm.bsk_group == 2 else n) * (k + 1) * prm.l_bsk * (k + 1) * N * 8
    br_bytes_per_fbs = ggsw + N * 8 + (n + 1) * 4 + (k * N + 1) * 8
    rec = dict(bound="fp64_valu", kernel=name, unit="VALU wave-instr/s", avg_launch_ms=br_ms, units_per_launch=units, peak=peak,
               achieved=algo / (br_ms * 1e-3), frac=algo / (br_ms * 1e-3) / peak, algorithmic_frac=algo / (br_ms * 1e-3) / peak,
               algorithmic_instr_per_bootstrap_step=algorithmic_instr_per_bootstrap(prm) / (n if prm.bsk_group != 2 else n // 2),
               algorithmic_formula="steps x ([(k+1) l + (k+1)] N/2 log2 N butterflies x %d + (k+1)^2 l N MACs x %d) / 64 lanes" % (BUTTERFLY_INSTR, MAC_INSTR),
               valu_frac=None, issue_cycle_frac=None, traffic=None, fabric_GBps=None,
               launches=[dict(kernel=kn, launches_per_step=v["launches"] / max(1, steps), avg_launch_ms=v["ms"] / max(1, v["launches"]))
                         for kn, v in sorted(br_all.items(), key=lambda kv: -kv[1]["ms"])],
               key_stream_vs_hbm_peak=br_bytes_per_fbs * units / (br_ms * 1e-3) / HBM_PEAK,
               key_stream_bytes_per_fbs=br_bytes_per_fbs,
               keyswitch_kernel=ks["kernel"], keyswitch_avg_launch_ms=ks_ms,
               measured_issue_ceiling=dict(
                   what="dependent v_fma_f64 chains (ILP 4-8) on every SIMD, tools/fp64_ilp.hip: one instruction per SIMD per 4.8-5.0 "
                        "nominal cycles with two waves per SIMD (what 256-register kernels get), 4.7 with four -- not the nominal 4",
                   frac_of_peak_two_waves_per_simd=[round(4 / 5.0, 2), round(4 / 4.8, 2)], source="profiles/r01/fp64_issue_rate.txt"),
               note="bound by FP64 VALU issue (one wave-instruction per SIMD per 4 cycles at the nominal 2.4 GHz); frac = algorithmic "
                    "work over time over that peak, valu_frac = executed instructions over the same (occupancy); "
                    "key_stream_vs_hbm_peak counts every key row once per bootstrap as SURVEY 8(d) prescribes and is not a utilisation: "
                    "the key is served from L2/Infinity Cache after first touch, `traffic` is what reached the fabric")
    pmc = None
    if os.path.exists(PMC_FILE):
        records = json.load(open(PMC_FILE))
        # (records of round 3 were keyed by the kernel name, "@units" at most: one kernel at two parameter sets could not both be held)
        pmc = records.get(pmc_key(name, n, units)) or records.get("%s@%d" % (name, units)) or records.get(name)
    if pmc is None:
        rec["pmc"] = "no record for %s in %s" % (name, os.path.relpath(PMC_FILE, ROOT))
    elif pmc.get("csrc_sha256") != kernel_sources_sha256():
        rec["pmc"] = "stale: %s was collected on other kernel sources (csrc sha256 %s..., now %s...); re-run tools/profile_round.sh" % (
            os.path.relpath(PMC_FILE, ROOT), str(pmc.get("csrc_sha256"))[:12], kernel_sources_sha256()[:12])
    elif pmc.get("units_per_launch") != units or pmc.get("n") != n:
        rec["pmc"] = "record is for %s bootstraps per launch at n = %s" % (pmc.get("units_per_launch"), pmc.get("n"))
    else:
        insts = pmc["SQ_INSTS_VALU_per_launch"]
        rec["valu_frac"] = insts / (br_ms * 1e-3) / peak
        rec["valu_insts_per_launch"] = insts
        rec["valu_insts_per_wave_per_step"] = pmc.get("valu_per_wave_per_step")
        if pmc.get("valu_64bit_per_launch") is not None:
            cycles = 4.0 * pmc["valu_64bit_per_launch"] + 2.0 * (insts - pmc["valu_64bit_per_launch"])
            rec["issue_cycle_frac"] = cycles / (br_ms * 1e-3) / (SIMDS * CLOCK_HZ)
            rec["valu_64bit_share"] = pmc["valu_64bit_per_launch"] / insts
        rec["traffic"] = pmc.get("hbm_bytes_per_launch")          # bytes per launch, FETCH_SIZE x2 + WRITE_SIZE
        if rec["traffic"]:
            rec["fabric_GBps"] = rec["traffic"] / (br_ms * 1e-3) / 1e9
        # the committed rocprofv3 --kernel-trace --stats summary of this command (profiles/r03/kernel_stats_*.csv): this kernel's
        # duration there, beside the live avg_launch_ms (under the tracer a launch runs 1-4 % longer; its minimum is the live time)
        if pmc.get("rocprof_avg_launch_ms") is not None:
            rec["rocprof_kernel_trace_ms"] = dict(avg=pmc["rocprof_avg_launch_ms"], min=pmc.get("rocprof_min_launch_ms"),
                                                  live_over_avg=br_ms / pmc["rocprof_avg_launch_ms"])
        rec["pmc"] = "%s; collected offline in separate rocprofv3 --pmc passes of this command on these kernel sources, not in this run" % pmc.get("source")
        rec["traffic_source"] = rec["pmc"]
    return rec


def run_batch(args, rank, world, local, dist):
    import numpy as np
    from tfhe_fbs_map_amd import P1024, Context
    prm = P1024                                       # reduced-noise benchmark set, p = 15
    ctx = Context(prm, seed=1, device=local)          # keys replicated: every rank derives them from the seed
    B = args.batch
    elapsed, prof, tables, cts, ids, msgs, out, ok = timed_batch(ctx, prm, B, rank, args.steps, args.warmup, dist)
    elapsed = max_over_ranks(elapsed, dist)
    ok = all_ok(ok, dist)
    if rank != 0:
        return None
    value = world * B * args.steps / elapsed
    result = dict(metric="functional bootstraps/sec (batched), N=1024", value=value, unit="FBS/s", n_gpus=world,
                  steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3,
                  higher_is_better=True, scaling="weak", vs_baseline=None,
                  dtype="f64 (exact integer arithmetic mod a 46-bit prime via FMA; residues in 64-bit words)",
                  data="synthetic",
                  config=dict(workload="BASELINE configs[1]: %d independent FBS per GPU per step, P1024 "
                                       "(n=630 N=1024 k=1 l=3 beta=7 t=8 gamma=2), p=15, 16 random tables; REDUCED NOISE "
                                       "(a kernel benchmark shape, not a secure configuration: see `params` and `secure`)" % B,
                              batch_per_gpu=B, parallelism="replicas of the batch per GPU, keys replicated, no collective",
                              device=ctx.device_info, params=params_record(prm)),
                  decrypt_ok=ok, roofline=roofline_record(prm, prof, prof["kernels"], B, args.steps, ctx.stat("cu_count")))
    if world == 1 and args.cpu_sample != 0:
        result["cpu_baseline"] = cpu_baseline(prm, tables, cts, ids, out, args.cpu_sample)
    ctx.close()
    if world == 1 and not args.no_secure:
        sec = result["secure"] = secure_leg(B, local, max(3, args.steps // 2), args.oracle_sample)
        # the deployable number beside the benchmark shape: the same batch at the 128-bit set for p = 15, norm2 70 (N = 1024, k = 2).
        # `value` stays on BASELINE's P1024 configuration.
        result["value_secure"] = dict(value=sec["value"], unit="FBS/s", decrypt_ok=sec["decrypt_ok"], bit_exact_vs_oracle=sec["bit_exact_vs_oracle"],
                                      what="the same batch at the 128-bit parameter set choose_params(15, 70, glwe_dims=(1, 2)) returns: LutExecEnv.eval's default",
                                      params={k: sec["params"][k] for k in ("n", "N", "k", "l", "beta", "t", "gamma", "p", "key_bits_per_step",
                                                                            "security_bits_estimate")},
                                      margin_sigmas_at_norm2_70=sec["margin_sigmas_at_norm2_70"], kernel=sec["blind_rotate_kernel"])
        result["shared_rotations"] = shared_rotations_leg(local)
        # (the headline shape is BASELINE's benchmark set at reduced noise; what a deployment runs, in one line of `config`)
        p4 = sec["n1024_p4"].get("k2", sec["n1024_p4"])
        result["config"]["secure_summary"] = (
            "128-bit parameter sets, same batch: p=15 %.0f FBS/s (n=%d N=%d k=%d l=%d, %d key bits per step%s); p=4 at N=1024 %.0f FBS/s "
            "(n=%d k=%d l=%d); p=31 %.0f FBS/s (n=%d N=%d l=%d, %d key bits per step)"
            % (sec["value"], sec["params"]["n"], sec["params"]["N"], sec["params"]["k"], sec["params"]["l"], sec["params"]["key_bits_per_step"],
               "; k=1 N=%d: %.0f" % (sec["k1"]["params"]["N"], sec["k1"]["value"]) if "k1" in sec else "",
               p4["value"], p4["params"]["n"], p4["params"]["k"], p4["params"]["l"], sec["p31"]["value"], sec["p31"]["params"]["n"],
               sec["p31"]["params"]["N"], sec["p31"]["params"]["l"], sec["p31"]["params"]["key_bits_per_step"]))
        k3 = sec["n1024_p4"].get("k3_n512")
        if k3:                                                               # the default set for p <= 8: GLWE dimension 3 at N = 512
            result["config"]["secure_summary"] += "; p=4 at the default set (n=%d N=%d k=%d, two full rounds of %d) %.0f FBS/s" % (
                k3["params"]["n"], k3["params"]["N"], k3["params"]["k"], k3["batch"] // 2, k3["value"])
    return result


def sharded_legs(rank, world, local, dist, circuit="trivium_stream_v2__search_p15", T=64, reduced_noise=False):
    """--gpus N > 1: north_star's multi-GPU shape next to the weak-scaling headline, in the same line -- one whole mapped program
    (BASELINE configs[3] stand-in) on T samples, STRONG scaling, cut three ways by distributed.ShardedRunner: gate-sharded (every
    level's (gate, sample) batch over all ranks, one RCCL all-gather per level), sample-sharded (no data-path collective) and
    the layout distributed.choose_sharding picks.  Every rank calls this (collectives inside); rank 0 returns the record."""
    import gzip
    import numpy as np
    import torch
    from tfhe_fbs_map_amd import Context, Program, choose_params, params_for, parse_fbs
    from tfhe_fbs_map_amd.distributed import GpuBackend, ShardedRunner, choose_sharding
    with gzip.open(os.path.join(ROOT, "tests", "golden", circuit + ".json.gz"), "rb") as f:
        rec = json.loads(f.read().decode())
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
    low = env.lower()
    # the 128-bit set LutExecEnv.eval would choose for this program (its own p and norm2; GLWE dimension 2 admitted): what a
    # deployment's ranks run, every rank the same set.  --sharded-reduced-noise: the benchmark set instead.
    p_msg = int(circuit.rsplit("_p", 1)[-1])
    prm = params_for(p_msg) if reduced_noise else choose_params(p_msg, env.stats()["norm2_linprod"], glwe_dims=(1, 2))
    ctx = Context(prm, seed=1, device=local)
    prog = Program(ctx, ctx.tvset(low["tables"]), len(low["input_names"]), low["kind"], low["arg0"], low["arg1"],
                   low["const_coef"], low["term_coef"], low["term_src"], low["out_wire"])
    bits = np.random.default_rng(42).integers(0, 2, (prog.n_inputs, T))
    clear = cleartext(low, bits)
    d_in = torch.from_numpy(ctx.encrypt(bits, nonce0=0).view(np.int64)).cuda()
    ctx.reserve(max_keyswitches=prog.max_width * T)
    pick = choose_sharding(list(prog.level_width), T, world, params=prm)     # priced on the staircase of the set actually loaded
    legs = {}
    for label, gs in (("gate", 1), ("sample", world if T >= world else None), ("chosen", pick["sample_groups"])):
        if gs is None or (label == "chosen" and any(v["sample_groups"] == gs for v in legs.values())):
            continue
        runner = ShardedRunner(GpuBackend(prog), sample_groups=gs)
        runner.time_collectives = True
        runner.run_local(d_in, T)                                     # warm-up (buffers, every launch shape)
        runner.collective_ms()
        before = runner.collectives
        ctx.profile(True)
        ctx.profile_read(reset=True)
        fence(dist)
        t0 = time.perf_counter()
        out, s0, s1 = runner.run_local(d_in, T)
        fence(dist)
        elapsed = max_over_ranks(time.perf_counter() - t0, dist)
        kernels = ctx.profile_kernels()
        prof = ctx.profile_read(reset=True)
        ctx.profile(False)
        got = ctx.decrypt(out.cpu().numpy().view(np.uint64))
        ok = all(np.array_equal(got[k][:s1 - s0], clear[k][s0:s1]) for k, w in enumerate(low["out_wire"]) if w >= 0)
        legs[label] = dict(sample_groups=gs, gate_groups=world // gs, rccl_ranks=world, seconds=elapsed,
                           value=prog.n_bootstrap * T / elapsed, unit="FBS/s", scaling="strong", backend=dist.get_backend(),
                           collectives_per_step=runner.collectives - before, allgather_ms_rank0=runner.collective_ms(),
                           kernels_ms_rank0={k: v["ms"] for k, v in prof.items()},
                           kernel_instantiations_rank0={k: dict(launches=v["launches"], ms=round(v["ms"], 3)) for k, v in kernels.items()},
                           decrypt_ok=all_ok(ok, dist))
    ctx.close()
    if rank != 0:
        return None
    return dict(workload="%s (reference mapper output, %d bootstraps, depth %d, widest level %d) on %d samples, %s, "
                         "one evaluation per layout" % (circuit, prog.n_bootstrap, prog.depth, prog.max_width, T,
                                                        "reduced-noise benchmark set" if reduced_noise else "the 128-bit set chosen for its (p, norm2)"),
                params=params_record(prm),
                choose_sharding=dict(sample_groups=pick["sample_groups"], gate_groups=pick["gate_groups"],
                                     predicted_speedup_over_one_gpu=round(pick["predicted_speedup"], 2)), legs=legs)


def oracle_sample_check(prm, tables, cts, ids, gpu_out, sample):
    """A bounded sample of a leg's timed batch -- both ends of the launch (every sub-slot of the first and of the last workgroup)
    and a spread in between -- bootstrapped by the scalar oracle on the same keys and compared word for word with what the GPU
    left.  The oracle is the checker here, after the timed region; it is never what is measured."""
    import numpy as np
    if sample <= 0:
        return dict(bit_exact_vs_oracle=None, oracle_sample="skipped (--oracle-sample 0)")
    from oracle import tfhe_oracle
    B = len(cts)
    edge = max(1, min(8, sample // 4))
    spread = np.random.default_rng(B).integers(0, B, max(0, sample - 2 * edge))
    pick = np.unique(np.concatenate([np.arange(min(edge, B)), np.arange(max(0, B - edge), B), spread]))
    ref, _ = tfhe_oracle.Oracle(prm, seed=1).bootstrap_batch(cts[pick], tables, ids[pick])
    return dict(bit_exact_vs_oracle=bool(np.array_equal(ref, gpu_out[pick])),
                oracle_sample="%d ciphertexts of the timed batch (both ends and a spread), scalar oracle on the same keys" % len(pick))


def secure_leg(B, local, steps, oracle_sample=32):
    """The same batch at the parameter set `choose_params` returns for p = 15 at norm2 = 70 (the 16x16 multiplier's and
    the adder's linear combinations), 128-bit noise, 6 sigma: what a deployment would run.  The selector takes two key bits
    per blind-rotation step there (bsk_group = 2); the one-bit-per-step choice is timed beside it."""
    from tfhe_fbs_map_amd import Context, choose_params
    from tfhe_fbs_map_amd.params import DEFAULT_GLWE_DIMS, bootstrap_cost, margin_sigmas

    def one(prm, B=B):
        ctx = Context(prm, seed=1, device=local)
        elapsed, prof, tables, cts, ids, _, out, ok = timed_batch(ctx, prm, B, 0, steps, 2, None)
        br, ks = prof["blind_rotate"], prof["keyswitch"]
        roof = roofline_record(prm, prof, prof["kernels"], B, steps, ctx.stat("cu_count"))
        rec = dict(value=B * steps / elapsed, unit="FBS/s", steps=steps, batch=B, decrypt_ok=ok,
                   **oracle_sample_check(prm, tables, cts, ids, out, oracle_sample), params=params_record(prm),
                   roofline={k: roof[k] for k in ("kernel", "avg_launch_ms", "frac", "algorithmic_frac", "valu_frac", "issue_cycle_frac",
                                                  "fabric_GBps", "pmc")} if roof else None,
                   margin_sigmas_at_norm2_70=round(margin_sigmas(prm, 70), 2), modelled_cost_vs_p1024=round(bootstrap_cost(prm), 3),
                   blind_rotate_kernel=br["kernel"], blind_rotate_avg_launch_ms=br["ms"] / max(1, br["launches"]),
                   keyswitch_avg_launch_ms=ks["ms"] / max(1, ks["launches"]))
        ctx.close()
        return rec

    # GLWE dimension 2 at N = 1024 admitted (`glwe_dims`), as ExecConfig does for every program; the k = 1 choice beside it
    rec = one(choose_params(15, 70, glwe_dims=(1, 2)))
    if rec["params"]["k"] != 1:
        rec["k1"] = dict(one(choose_params(15, 70)), note="the k = 1 choice for the same (p, norm2): N = 2048 (ExecConfig(glwe_dims=(1,)); shared rotations)")
    if rec["params"]["key_bits_per_step"] != 1:
        rec["one_key_bit_per_step"] = one(choose_params(15, 70, groups=(1,)))
    # the metric names N = 1024: at 128-bit noise that polynomial size carries small plaintext moduli only -- p = 4 is the
    # reference's own Trivium / Kreyvium comparison point (experiments/analyse_results.py:317)
    small = choose_params(4, 2)
    rec["n1024_p4"] = dict(one(small), note="128-bit set for p = 4 at norm2 = 2: the N = 1024 kernels at a secure parameter set")
    if choose_params(4, 2, glwe_dims=(1, 2)).k == 2:
        rec["n1024_p4"]["k2"] = one(choose_params(4, 2, glwe_dims=(1, 2)))
    # ... and what ExecConfig() takes for p <= 8 at ordinary norms: GLWE dimension 3 at N = 512 (k N = 1536, between the two noise floors
    # k = 1 offers) on k_blind_rotate_glwe, four waves per bootstrap, three bootstraps per workgroup -- a round of the chip is 3 x CUs
    # bootstraps, so this leg is timed on two full rounds (a batch of 1 024 is cut into a round and a launch of its own for the rest)
    k3 = choose_params(4, 2, glwe_dims=DEFAULT_GLWE_DIMS)
    if k3.k == 3 and B >= 1024:
        rec["n1024_p4"]["k3_n512"] = dict(one(k3, B=1536), note="the default 128-bit set for p = 4 (ExecConfig: glwe_dims = (1, 2, 3)): k = 3, N = 512, "
                                                             "two key bits per step; two full rounds of 768 bootstraps")
    # BASELINE configs[4] (fbs_size = 31; no non-power-of-two N here: the 128-bit set is N = 2048 with two gadget levels)
    big = choose_params(31, 325)
    rec["p31"] = dict(one(big), note="128-bit set for p = 31 at norm2 = 325 (BASELINE configs[4]: fbs_size = 31)")
    if rec["p31"]["params"]["key_bits_per_step"] != 1:
        rec["p31"]["one_key_bit_per_step"] = one(choose_params(31, 325, groups=(1,)))
    return rec


def shared_rotations_leg(local, bits=32, T=512):
    """Several tables on ONE blind rotation (include/fbs_exec.h FBS_LOAD_FUSE_TABLES): a ripple-carry adder built
    gate by gate and lowered like the reference's MapToFBSBasic -- XOR and AND of the same pair of wires are two tables on
    one linear combination (fbs_mapper/map_to_fbs.py:41-45) -- evaluated with and without shared rotations, each at the
    128-bit parameter set chosen for its own noise statistic; every sum is decrypted and checked."""
    import numpy as np
    from tfhe_fbs_map_amd import ExecConfig
    from tfhe_fbs_map_amd.fbs_exec_env import min_fbs_size
    from tfhe_fbs_map_amd.netlist import BitExecEnv, map_basic
    from tfhe_fbs_map_amd.params import margin_sigmas
    env = BitExecEnv()
    a = [env.input("a%d" % i) for i in range(bits)]
    b = [env.input("b%d" % i) for i in range(bits)]
    carry = None
    for i in range(bits):
        x, g = env.op_xor(a[i], b[i]), env.op_and(a[i], b[i])
        s_i, carry = (x, g) if carry is None else (env.op_xor(x, carry), env.op_or(g, env.op_and(x, carry)))
        env.output("s%d" % i, s_i)
    env.output("cout", carry)
    lut = map_basic(env)
    low = lut.lower()
    p = min_fbs_size(low["tables"])
    rng = np.random.default_rng(3)
    x, y = rng.integers(0, 2, (bits, T)), rng.integers(0, 2, (bits, T))
    ins = {("a%d" % i): x[i] for i in range(bits)} | {("b%d" % i): y[i] for i in range(bits)}
    want = sum((x[i].astype(object) + y[i].astype(object)) << i for i in range(bits))
    rec = dict(circuit="%d-bit ripple-carry adder, one table per two-input gate" % bits, samples=T, p=p,
               norm2=lut.stats()["norm2_linprod"], norm2_with_shared_rotations=round(lut.fusion_stats(p)["norm2_linprod"], 2))
    for fuse in (False, True):
        cfg = ExecConfig(seed=9, fuse_tables=fuse, device=local)
        ctx, fused = cfg.choose(lut, p)
        prog = cfg.program_for(ctx, low, fused)
        cts = ctx.encrypt(np.stack([ins[n] for n in low["input_names"]]).astype(np.int64), nonce0=1)
        prog.eval(cts[:, :8].copy(), 8)
        t0 = time.perf_counter()
        res = prog.eval(cts, T)
        dt = time.perf_counter() - t0
        dec = ctx.decrypt(res)
        got = sum(dec[low["out_names"].index("s%d" % i)].astype(object) << i for i in range(bits))
        got = got + (dec[low["out_names"].index("cout")].astype(object) << bits)
        norm2 = rec["norm2_with_shared_rotations"] if fused else rec["norm2"]
        rec["fused" if fuse else "plain"] = dict(
            seconds=round(dt, 4), tables=prog.n_bootstrap, blind_rotations=prog.n_rotations,
            tables_per_s=round(prog.n_bootstrap * T / dt), params=params_record(ctx.params),
            margin_sigmas_at_its_norm2=round(margin_sigmas(ctx.params, norm2), 2), all_sums_correct=bool(all(g == w for g, w in zip(got, want))))
        ctx.close()
    rec["speedup"] = round(rec["plain"]["seconds"] / rec["fused"]["seconds"], 3)
    return rec


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(prm, tables, cts, ids, gpu_out, sample):
    """The oracle (a plain-C port, OpenMP over independent FBS) on the first `sample` ciphertexts of the same batch,
    same keys, on all the cores this process may use -- and on ONE thread; also checks those
    ciphertexts bit-for-bit against the GPU's."""
    import numpy as np
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
    one = max(1, min(sample, 24))
    t0 = time.perf_counter()
    orc.bootstrap_batch(cts[:one], tables, ids[:one], threads=1)
    dt1 = time.perf_counter() - t0
    rec = dict(value=sample / dt, unit="FBS/s", cores=used, kind="port", cpu_model=cpu_model(),
               sample="first %d ciphertexts of the timed batch, %.1f s" % (sample, dt),
               one_thread=dict(value=one / dt1, unit="FBS/s", sample="first %d ciphertexts, %.1f s" % (one, dt1)),
               bit_exact_vs_gpu=bool(np.array_equal(ref, gpu_out[:sample])),
               note="the scalar checker (plain C, 128-bit products, one bootstrap per thread): the parity witness, not a tuned library; "
                    "`tuned` beside it is what the same scheme does on this CPU when written for it")
    rec["tuned"] = tuned_cpu_baseline(orc, tables, cts, ids, gpu_out, cores)
    return rec


def tuned_cpu_baseline(orc, tables, cts, ids, gpu_out, cores):
    """oracle/tfhe_tuned.c: AVX-512 IFMA (the 46-bit modulus fits the 52-bit multiplier), eight bootstraps per vector, OpenMP over
    groups; held to the scalar oracle word for word by tests/test_oracle_tfhe.py and checked against the GPU's ciphertexts here.
    `cores` threads: a one-GPU box is a 16-core share of its host, and the record says how many were used."""
    import numpy as np
    from oracle import tfhe_tuned
    if not tfhe_tuned.supported():
        return dict(kind="tuned", value=None, note="this host CPU has no AVX-512 IFMA")
    try:
        t = tfhe_tuned.Tuned(orc)
    except ValueError as e:
        return dict(kind="tuned", value=None, note=str(e))
    sample = min(len(cts), 128 * cores)
    t.bootstrap_batch(cts[:8], tables, ids[:8], threads=1)          # touch the key once
    passes = 0
    t0 = time.perf_counter()
    while passes < 3 or (time.perf_counter() - t0 < 1.0 and passes < 50):   # a bounded sample: 10-30 core-seconds of work
        out, used = t.bootstrap_batch(cts[:sample], tables, ids[:sample], threads=cores)
        passes += 1
    dt = (time.perf_counter() - t0) / passes
    one = min(sample, 64)
    t0 = time.perf_counter()
    t.bootstrap_batch(cts[:one], tables, ids[:one], threads=1)
    dt1 = time.perf_counter() - t0
    return dict(kind="tuned", value=sample / dt, unit="FBS/s", cores=used, cpu_model=cpu_model(),
                sample="first %d ciphertexts of the timed batch, %d passes of %.2f s" % (sample, passes, dt),
                one_thread=dict(value=one / dt1, unit="FBS/s", sample="first %d ciphertexts, %.2f s" % (one, dt1)),
                bit_exact_vs_gpu=bool(np.array_equal(out, gpu_out[:sample])),
                how="AVX-512 IFMA Shoup products, 8 bootstraps per vector, OpenMP over groups of 8 (oracle/tfhe_tuned.c)")


# ---------------------------------------------------------------------------------------------------------------------
def run_circuit(args, rank, world, local, dist):
    import gzip
    import numpy as np
    import torch
    from tfhe_fbs_map_amd import Context, Program, min_fbs_size, params_for, parse_fbs
    from tfhe_fbs_map_amd.distributed import GateShardedRunner, GpuBackend

    with gzip.open(os.path.join(ROOT, "tests", "golden", args.circuit + ".json.gz"), "rb") as f:
        rec = json.loads(f.read().decode())
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"])
    low = env.lower()
    stats = env.stats()
    tail = args.circuit.rsplit("_p", 1)[-1].split("_")[0]            # fixtures are named <circuit>__<mapper>_p<fbs_size>
    p = int(tail) if tail.isdigit() else min_fbs_size(low["tables"])
    if args.secure:
        from tfhe_fbs_map_amd import choose_params
        from tfhe_fbs_map_amd.params import REFERENCE_MARGIN
        prm = choose_params(p, stats["norm2_linprod"], floor_margin=REFERENCE_MARGIN)
    else:
        prm = params_for(p)                                           # the reduced-noise benchmark set for p
    ctx = Context(prm, seed=1, device=local)
    prog = Program(ctx, ctx.tvset(low["tables"]), len(low["input_names"]), low["kind"], low["arg0"], low["arg1"],
                   low["const_coef"], low["term_coef"], low["term_src"], low["out_wire"])
    T = args.samples
    # gate mode: every rank holds the SAME T samples; sample mode: rank r its own T samples
    rng = np.random.default_rng(42 + (rank if args.mode == "sample" else 0))
    bits = rng.integers(0, 2, (prog.n_inputs, T))
    clear = cleartext(low, bits)
    d_in = torch.from_numpy(ctx.encrypt(bits, nonce0=0).view(np.int64)).cuda()
    collectives = lambda: 0                                                          # noqa: E731
    window = (0, T)
    layout = dict(sample_groups=1, gate_groups=world) if args.mode == "gate" else dict(sample_groups=world, gate_groups=1)
    if args.mode == "gate":
        runner = GateShardedRunner(GpuBackend(prog))
        step = lambda: runner.run_device(d_in, T)                                   # noqa: E731
        collectives = lambda: runner.collectives                                    # noqa: E731
    elif args.mode == "auto":
        # the layout choose_sharding prices cheapest for this program, T and world: sample groups x gate groups (strong scaling)
        from tfhe_fbs_map_amd.distributed import ShardedRunner, choose_sharding
        pick = choose_sharding(list(prog.level_width), T, world)
        layout = dict(sample_groups=pick["sample_groups"], gate_groups=pick["gate_groups"],
                      predicted_speedup_over_one_gpu=round(pick["predicted_speedup"], 2))
        runner = ShardedRunner(GpuBackend(prog), sample_groups=pick["sample_groups"])
        window = split_window(T, pick["sample_groups"], rank // pick["gate_groups"])
        step = lambda: runner.run_local(d_in, T)[0]                                 # noqa: E731
        collectives = lambda: runner.collectives                                    # noqa: E731
    else:
        # sample-sharded: the rank's share IS the whole program on its own samples -- one device-side call, no exchange
        d_out = torch.empty((prog.n_outputs, T, prm.ct_words), dtype=torch.int64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream

        def step():
            prog.eval_dev(d_in.data_ptr(), T, d_out.data_ptr(), stream)
            return d_out
    for _ in range(args.warmup):
        out = step()
    ctx.profile(True)
    ctx.profile_read(reset=True)
    fence(dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence(dist)
    elapsed = max_over_ranks(time.perf_counter() - t0, dist)
    prof = ctx.profile_read(reset=True)
    got = ctx.decrypt(out.cpu().numpy().view(np.uint64))
    ok = all(np.array_equal(got[k][:window[1] - window[0]], clear[k][window[0]:window[1]]) for k, w in enumerate(low["out_wire"]) if w >= 0)
    ok = all_ok(ok, dist)
    if rank != 0:
        return None
    fbs_per_eval = prog.n_bootstrap * T * (world if args.mode == "sample" else 1)
    value = fbs_per_eval * args.steps / elapsed
    return dict(metric="functional bootstraps/sec (batched), N=1024", value=value, unit="FBS/s", n_gpus=world, steps=args.steps,
                warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3, higher_is_better=True,
                scaling="weak" if args.mode == "sample" else "strong", vs_baseline=None,
                dtype="f64 (exact integer arithmetic mod a 46-bit prime via FMA; residues in 64-bit words)", data="synthetic",
                config=dict(workload="BASELINE configs[3] stand-in: %s (reference mapper output, %d bootstraps, depth %d, widest level %d) "
                                     "on %d samples%s, %s" % (args.circuit, prog.n_bootstrap, prog.depth, prog.max_width, T,
                                                                              " per rank" if args.mode == "sample" else "",
                                                                              "128-bit set chosen for its (p, norm2)" if args.secure else "reduced-noise benchmark set"),
                            mode=args.mode, layout=layout, rccl_ranks=world, collectives_per_step=collectives() // max(1, args.steps + args.warmup),
                            parallelism=("levels cut across ranks, one all-gather per level" if args.mode == "gate" else
                                         "samples cut across ranks, no data-path collective" if args.mode == "sample" else
                                         "sample groups x gate groups as distributed.choose_sharding lays them out"),
                            wire_slots=prog.n_slots, wires=prog.n_inputs + len(low["kind"]), key_switches=prog.n_keyswitch,
                            norm2_linprod=stats["norm2_linprod"], device=ctx.device_info, params=params_record(prm)),
                decrypt_ok=ok,
                kernels_ms_per_step={k: v["ms"] / args.steps for k, v in prof.items()})


def split_window(T, parts, r):
    chunk = -(-T // parts)
    return min(T, r * chunk), min(T, (r + 1) * chunk)


def cleartext(low, bits):
    """What the program computes in the clear (the reference's LutExecEnv.eval loop, fbs_exec_env.py:208-229, on the
    lowered arrays) -- the check of the bench's own outputs, not a timed path."""
    import numpy as np
    n_in = len(low["input_names"])
    wires = [bits[i].astype(np.int64) for i in range(n_in)]
    for i, kind in enumerate(low["kind"]):
        if kind == 0:
            a, c = low["arg0"][i], low["arg1"][i]
            v = np.full(bits.shape[1], low["const_coef"][i], np.int64)
            for t in range(a, a + c):
                v = v + low["term_coef"][t] * wires[low["term_src"][t]]
            wires.append(v)
        else:
            table = np.asarray(low["tables"][low["arg1"][i]], np.int64)
            wires.append(table[wires[low["arg0"][i]]])
    return [wires[w] if w >= 0 else np.full(bits.shape[1], -1 - w, np.int64) for w in low["out_wire"]]


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch(args, argv)
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
