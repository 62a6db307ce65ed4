"""Multi-GPU evaluation of FBS programs: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on MI355X; "gloo" in the CPU tests).  No counterpart in the reference, which is single-process
(SURVEY 8e); the two modes are the two independent axes of its `eval` loop
(fbs_mapper/fbs_exec_env.py:211-223): gates within a bootstrap level, and the sample axis.

* `SampleShardedRunner` -- rank r evaluates the WHOLE program on its slice of the samples.  No data-path
  collective; outputs are all-gathered once at the end.
* `ShardedRunner`       -- both at once: sample groups x gate groups, laid out by `choose_sharding` from the
  program's level widths, T and the measured launch-time staircase.
* `GateShardedRunner`   -- the wire buffer is replicated; at every level the flattened (gate, sample)
  batch is cut into G contiguous slices, rank r bootstraps slice r into a contiguous send buffer and ONE
  all-gather per level publishes the new ciphertexts, which a copy kernel then files into their wire slots
  (xGMI is point-to-point; the per-level payload is W*T/G ciphertexts of 8.2 KB per rank -- e.g. 0.3 GB per
  rank and level for the 300-gate levels of trivium_stream_v2 at T = 1000, a few ms on seven links against
  0.4 s of bootstraps -- so one direct all-gather per level is the right shape, not a ring of many small ones).

What crosses the links is the bootstrap OUTPUT, a ciphertext under the big key (8 200 B): it has to feed the
next level's linear combinations, which the single-use modulus-switched form (2 524 B) cannot.

Keys are replicated: every rank derives the same keys from the same seed.  Arithmetic is exact, so both
modes return, bit for bit, what one GPU returns.

The level arithmetic is behind a small backend object (new_wires / load_inputs / lincomb_level /
bootstrap_level / scatter_level / read_outputs) so that the partitioning and collective logic can be exercised
on CPU ranks (tests inject an oracle-backed backend); `GpuBackend` is the product: every call is a few
kernel launches of libfbsexec on torch's current stream, with the index arrays uploaded once by
`fbs_program_load` -- no host synchronisation inside a level.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


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
# product backend: libfbsexec on device-resident wires
# --------------------------------------------------------------------------------------------
class GpuBackend:
    """Steps one loaded `Program` on torch-owned device memory.  `ctw` words per ciphertext; a wire buffer is
    [n_slots * T, ctw] int64 (the library's slot layout, include/fbs_exec.h "one level at a time")."""

    def __init__(self, program):
        self.prog = program
        self.ctx = program.ctx
        self.ctw = self.ctx.params.ct_words
        self.depth = program.depth
        self.level_width = list(program.level_width)
        self.n_inputs, self.n_outputs = program.n_inputs, program.n_outputs
        self.device = torch.device("cuda", torch.cuda.current_device())

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def new_wires(self, T):
        return torch.empty((self.prog.n_slots * max(1, T), self.ctw), dtype=torch.int64, device=self.device)

    def new_rows(self, rows):
        """rows of the level calls' send / gather buffers: one ciphertext each, or 2N words for a fused program (a shared
        rotation's accumulator travels in its row: include/fbs_exec.h FBS_LOAD_FUSE_TABLES)"""
        return torch.empty((max(1, rows), self.prog.row_words), dtype=torch.int64, device=self.device)

    def load_inputs(self, wires, T, in_cts, s_count):
        """in_cts: [n_inputs][s_count][ctw] host array (or device tensor) -> the inputs' wire slots"""
        if s_count == 0:
            return
        src = in_cts if torch.is_tensor(in_cts) else \
            torch.from_numpy(np.ascontiguousarray(in_cts, np.uint64).view(np.int64))
        src = src.reshape(self.n_inputs, s_count, self.ctw).to(self.device, non_blocking=True)
        view = wires.view(self.prog.n_slots, -1, self.ctw)
        view[torch.from_numpy(self.prog.in_slot.astype(np.int64)).to(self.device), :s_count] = src

    def lincomb_level(self, wires, T, L, s_count):
        if s_count:
            self.prog.level_lincomb_dev(L, wires.data_ptr(), T, 0, s_count, stream=self._stream())

    def bootstrap_level(self, wires, T, L, s_count, f0, f1, rows=None):
        if f1 > f0:
            self.prog.level_bootstrap_dev(L, wires.data_ptr(), T, 0, s_count, f0, f1,
                                          d_rows=0 if rows is None else rows.data_ptr(), stream=self._stream())

    def scatter_level(self, wires, T, L, s_count, rows, f0, f1):
        if f1 > f0:
            self.prog.level_scatter_dev(L, wires.data_ptr(), T, 0, s_count, rows.data_ptr(), f0, f1, stream=self._stream())

    def read_outputs(self, wires, T, s_count):
        """-> device tensor [n_outputs, s_count, ctw]; constant outputs are left zero (the caller knows them)"""
        view = wires.view(self.prog.n_slots, -1, self.ctw)
        slots = torch.from_numpy(np.maximum(self.prog.out_slot, 0)).to(self.device)
        out = view[slots, :s_count].contiguous()
        if (self.prog.out_slot < 0).any():
            out[torch.from_numpy(self.prog.out_slot < 0).to(self.device)] = 0
        return out


# --------------------------------------------------------------------------------------------
def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _all_gather_rows(out, send, world, group):
    """all-gather of equal-sized [rows, ctw] tensors into out = [world*rows, ctw]"""
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, send, group=group)
    else:
        dist.all_gather(list(out.chunk(world)), send, group=group)


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


# --------------------------------------------------------------------------------------------
class ShardedRunner:
    """`sample_groups` x `gate_groups` ranks (rank = sample group * gate_groups + position in its gate group).  A sample group
    evaluates the whole program on its slice of the samples; inside it every level's flattened (gate, sample) batch is cut
    into gate_groups contiguous slices, rank r bootstraps slice r into a contiguous send buffer, ONE all-gather per level
    (within the group) publishes the new ciphertexts and a copy kernel files them into their wire slots.  sample_groups =
    world is the sample-sharded mode (no data-path collective), sample_groups = 1 the gate-sharded one (north_star's shape);
    `choose_sharding` picks.  Every rank calls `run` with the SAME input ciphertexts."""

    def __init__(self, backend, group=None, sample_groups=None, always_gather=False):
        """always_gather: take the send-buffer / all-gather / scatter path even with a single rank per gate group (tests)."""
        self.be, self.group, self.always_gather = backend, group, always_gather
        self.rank, self.world = _world(group)
        self.sample_groups = self.world if sample_groups is None else int(sample_groups)
        if self.sample_groups < 1 or self.world % self.sample_groups:
            raise ValueError("sample_groups must divide the number of ranks")
        self.gate_groups = self.world // self.sample_groups
        self.sg, self.gg = divmod(self.rank, self.gate_groups)
        self.gate_group = group
        if 1 < self.gate_groups < self.world:
            # every rank creates every subgroup, in the same order (torch.distributed's rule); it keeps its own
            for g in range(self.sample_groups):
                ranks = list(range(g * self.gate_groups, (g + 1) * self.gate_groups))
                sub = dist.new_group(ranks)
                if g == self.sg:
                    self.gate_group = sub
        self.collectives = 0
        self.bootstraps_done = 0
        self._bufs = None
        self.time_collectives = False      # bench: bracket every all-gather with events on the current stream
        self._events = []

    def collective_ms(self):
        """Device time of the all-gathers since the last call (time_collectives = True), synchronising on them."""
        ms = 0.0
        for a, b in self._events:
            b.synchronize()
            ms += a.elapsed_time(b)
        self._events = []
        return ms

    def _gather(self, out, send, ranks, group):
        if self.time_collectives and send.is_cuda:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _all_gather_rows(out, send, ranks, group)
            b.record()
            self._events.append((a, b))
        else:
            _all_gather_rows(out, send, ranks, group)
        self.collectives += 1

    def _buffers(self, Tr):
        be = self.be
        chunk = max((-(-w * Tr // self.gate_groups) for w in be.level_width), default=1)
        key = (Tr, self.gate_groups)
        if self._bufs is None or self._bufs[0] != key:
            gather = self.gate_groups > 1 or self.always_gather
            self._bufs = (key, be.new_wires(Tr), be.new_rows(chunk), be.new_rows(chunk * self.gate_groups) if gather else None)
        return self._bufs[1:]

    def run_local(self, in_cts, T):
        """This rank's share: -> (backend tensor [n_outputs, chunk, ctw] of which the first s1 - s0 samples are valid, s0, s1);
        complete on every rank of a sample group."""
        be = self.be
        s0, s1, chunk = split_range(T, self.sample_groups, self.sg)
        Tr, cnt = max(1, chunk), s1 - s0
        wires, send, gathered = self._buffers(Tr)
        if cnt:
            local = in_cts.reshape(be.n_inputs, T, be.ctw)[:, s0:s1]
            be.load_inputs(wires, Tr, local if torch.is_tensor(local) else np.ascontiguousarray(local), cnt)
        for L in range(be.depth + 1):
            be.lincomb_level(wires, Tr, L, cnt)              # cheap, done redundantly on every rank of the group
            if L == be.depth:
                break
            total = be.level_width[L] * cnt
            f0, f1, rows = split_range(total, self.gate_groups, self.gg)
            self.bootstraps_done += f1 - f0
            if self.gate_groups == 1 and not self.always_gather:
                be.bootstrap_level(wires, Tr, L, cnt, f0, f1)    # straight into the wire slots
                continue
            if total == 0:
                continue
            be.bootstrap_level(wires, Tr, L, cnt, f0, f1, send)
            self._gather(gathered[:self.gate_groups * rows], send[:rows], self.gate_groups, self.gate_group)
            be.scatter_level(wires, Tr, L, cnt, gathered, 0, total)   # slices are contiguous: rank r's rows start at r * rows
        return be.read_outputs(wires, Tr, Tr), s0, s1

    def run_device(self, in_cts, T):
        """Gate-sharded use (sample_groups = 1): the backend's output tensor [n_outputs, T, ctw] (constant outputs zero)."""
        if self.sample_groups != 1:
            raise ValueError("run_device returns whole outputs: it is for sample_groups = 1 (use run or run_local)")
        return self.run_local(in_cts, T)[0]

    def run(self, in_cts, T):
        """-> host array [n_outputs, T, ctw], the same on every rank."""
        be = self.be
        in_cts = in_cts if torch.is_tensor(in_cts) else np.asarray(in_cts)
        send, s0, s1 = self.run_local(in_cts, T)
        if self.sample_groups == 1:
            return send.cpu().numpy().view(np.uint64)[:, :T]
        n_out, Tr = send.shape[0], send.shape[1]
        send = send.reshape(n_out * Tr, be.ctw).contiguous()
        gathered = torch.empty((self.world * n_out * Tr, be.ctw), dtype=send.dtype, device=send.device)
        self._gather(gathered, send, self.world, self.group)          # one gather of the outputs at the very end
        g = gathered.cpu().numpy().view(np.uint64).reshape(self.world, n_out, Tr, be.ctw)
        out = np.zeros((n_out, T, be.ctw), np.uint64)
        for sg in range(self.sample_groups):
            a, b_, _ = split_range(T, self.sample_groups, sg)
            out[:, a:b_] = g[sg * self.gate_groups, :, :b_ - a]      # (every rank of a sample group holds the group's outputs)
        return out


class GateShardedRunner(ShardedRunner):
    """Per-level all-gather over all ranks (sample_groups = 1)."""

    def __init__(self, backend, group=None, always_gather=False):
        super().__init__(backend, group, sample_groups=1, always_gather=always_gather)


class SampleShardedRunner(ShardedRunner):
    """Whole program per rank on a slice of the samples; one all-gather of the outputs at the end."""

    def __init__(self, backend, group=None):
        super().__init__(backend, group, sample_groups=None)
