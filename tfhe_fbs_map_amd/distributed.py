"""Multi-GPU evaluation of FBS programs: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on MI355X; "gloo" in the CPU tests).  No counterpart in the reference, which is single-process
(SURVEY 8e); the two modes are the two independent axes of its `eval` loop
(fbs_mapper/fbs_exec_env.py:211-223): gates within a bootstrap level, and the sample axis.

* `SampleShardedRunner` -- rank r evaluates the WHOLE program on its slice of the samples.  No data-path
  collective; outputs are all-gathered once at the end.
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
        return torch.empty((max(1, rows), self.ctw), dtype=torch.int64, device=self.device)

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


class GateShardedRunner:
    """Per-level all-gather.  Every rank calls `run` with the SAME input ciphertexts."""

    def __init__(self, backend, group=None, always_gather=False):
        """always_gather: take the send-buffer / all-gather / scatter path even with a single rank (tests)."""
        self.be, self.group, self.always_gather = backend, group, always_gather
        self.collectives = 0
        self.bootstraps_done = 0
        self._bufs = None

    def _buffers(self, T, world):
        be = self.be
        chunk = max((-(-w * T // world) for w in be.level_width), default=1)
        key = (T, world)
        if self._bufs is None or self._bufs[0] != key:
            gather = world > 1 or self.always_gather
            self._bufs = (key, be.new_wires(T), be.new_rows(chunk), be.new_rows(chunk * world) if gather else None)
        return self._bufs[1:]

    def run_device(self, in_cts, T):
        """Evaluate; returns the backend's output tensor [n_outputs, T, ctw] (constant outputs zero)."""
        be = self.be
        rank, world = _world(self.group)
        wires, send, gathered = self._buffers(T, world)
        be.load_inputs(wires, T, in_cts, T)
        for L in range(be.depth + 1):
            be.lincomb_level(wires, T, L, T)                  # cheap, done redundantly on every rank
            if L == be.depth:
                break
            total = be.level_width[L] * T
            f0, f1, chunk = split_range(total, world, rank)
            self.bootstraps_done += f1 - f0
            if world == 1 and not self.always_gather:
                be.bootstrap_level(wires, T, L, T, f0, f1)    # straight into the wire slots
                continue
            be.bootstrap_level(wires, T, L, T, f0, f1, send)
            _all_gather_rows(gathered[:world * chunk], send[:chunk], world, self.group)
            self.collectives += 1
            be.scatter_level(wires, T, L, T, gathered, 0, total)   # slices are contiguous: rank r's rows start at r*chunk
        return be.read_outputs(wires, T, T)

    def run(self, in_cts, T):
        out = self.run_device(in_cts, T)
        return out.cpu().numpy().view(np.uint64)


class SampleShardedRunner:
    """Whole program per rank on a slice of the samples; one all-gather of the outputs at the end."""

    def __init__(self, backend, group=None):
        self.be, self.group = backend, group
        self.collectives = 0
        self.bootstraps_done = 0

    def run_local(self, in_cts, T):
        """This rank's share: -> (device tensor [n_outputs, chunk, ctw] of which the first s1-s0 samples are valid, s0, s1)"""
        be = self.be
        rank, world = _world(self.group)
        s0, s1, chunk = split_range(T, world, rank)
        Tr, cnt = max(1, chunk), s1 - s0
        wires = be.new_wires(Tr)
        if cnt:
            local = in_cts.reshape(be.n_inputs, T, be.ctw)[:, s0:s1]
            be.load_inputs(wires, Tr, local if torch.is_tensor(local) else np.ascontiguousarray(local), cnt)
        for L in range(be.depth + 1):
            be.lincomb_level(wires, Tr, L, cnt)
            if L < be.depth:
                be.bootstrap_level(wires, Tr, L, cnt, 0, be.level_width[L] * cnt)
                self.bootstraps_done += be.level_width[L] * cnt
        out = be.read_outputs(wires, Tr, Tr)
        return out, s0, s1

    def run(self, in_cts, T):
        be = self.be
        rank, world = _world(self.group)
        in_cts = in_cts if torch.is_tensor(in_cts) else np.asarray(in_cts)
        send, s0, s1 = self.run_local(in_cts, T)
        n_out, Tr = send.shape[0], send.shape[1]
        send = send.reshape(n_out * Tr, be.ctw)
        if world > 1:
            gathered = torch.empty((world * n_out * Tr, be.ctw), dtype=send.dtype, device=send.device)
            _all_gather_rows(gathered, send.contiguous(), world, self.group)
            self.collectives += 1
        else:
            gathered = send
        g = gathered.cpu().numpy().view(np.uint64).reshape(world, n_out, Tr, be.ctw)
        out = np.zeros((n_out, T, be.ctw), np.uint64)
        for r in range(world):
            a, b_, _ = split_range(T, world, r)
            out[:, a:b_] = g[r, :, :b_ - a]
        return out
