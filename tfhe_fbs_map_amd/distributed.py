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


from .schedule import (LAUNCH_FAMILIES, LAUNCH_MS_P1024, ROUND_MS_P1024, allgather_ms, choose_sharding, launch_family,  # noqa: F401  (re-exported)
                       launch_ms, plan_levels, split_range)


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
        """rows of the level calls' send / gather buffers: one ciphertext each, or (k + 1) N words for a fused program (a shared
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
