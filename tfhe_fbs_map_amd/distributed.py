"""Multi-GPU evaluation of FBS programs: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI on MI355X; "gloo" in the CPU tests).  No counterpart in the reference, which is single-process
(SURVEY 8e); the two modes are the two independent axes of its `eval` loop
(fbs_mapper/fbs_exec_env.py:211-223): gates within a bootstrap level, and the sample axis.

* `SampleShardedRunner` -- rank r evaluates the WHOLE program on its slice of the samples.  No data-path
  collective; outputs are all-gathered once at the end.
* `GateShardedRunner`   -- the wire buffer is replicated; at every level the flattened (gate, sample)
  batch is cut into G contiguous slices, rank r bootstraps slice r and ONE all-gather per level publishes
  the new ciphertexts (xGMI is point-to-point; the per-level payload is small -- W*T/G ciphertexts of
  8.2 KB -- so a single direct all-gather per level is the right shape, not a ring of many small ones).

Keys are replicated: every rank derives the same keys from the same seed.  Arithmetic is exact, so both
modes return, bit for bit, what one GPU returns.

The level arithmetic itself is behind a small backend object so that the partitioning and collective
logic can be exercised on CPU ranks (tests inject an oracle-backed backend); `GpuBackend` is the product.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


# --------------------------------------------------------------------------------------------
# schedule (same rule as fbs_program_load in csrc/fbs_capi.cpp)
# --------------------------------------------------------------------------------------------
def plan_levels(low):
    """`low` = LutExecEnv.lower().  Returns dict(depth, n_wires, lin=[[stage,...] per level], boot=[stage per level])
    where a lincomb stage is dict(dst, term_off, srcs, coefs, consts) and a boot stage dict(src, dst, table)."""
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
    boot = [dict(src=[], dst=[], table=[]) for _ in range(depth)]
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
            b = boot[level[w] - 1]
            b["src"].append(low["arg0"][i]); b["dst"].append(w); b["table"].append(low["arg1"][i])
    return dict(depth=depth, n_wires=n_wires, n_inputs=n_in,
                lin=[[d[k] for k in sorted(d)] for d in lin], boot=boot)


def split_range(total, parts, r):
    """Contiguous slice r of `parts` near-equal slices of range(total)."""
    chunk = -(-total // parts)
    return min(total, r * chunk), min(total, (r + 1) * chunk), chunk


def rectangles(f0, f1, T):
    """Cover flattened indices [f0, f1) of a [gates][T] grid by (gate_begin, gate_end, s_begin, s_end) boxes."""
    out = []
    while f0 < f1:
        g, s = divmod(f0, T)
        if s == 0 and f1 - f0 >= T:
            g1 = g + (f1 - f0) // T
            out.append((g, g1, 0, T))
            f0 = g1 * T
        else:
            s1 = min(T, s + (f1 - f0))
            out.append((g, g + 1, s, s1))
            f0 += s1 - s
    return out


# --------------------------------------------------------------------------------------------
# product backend: libfbsexec on device-resident wires
# --------------------------------------------------------------------------------------------
class GpuBackend:
    def __init__(self, ctx, tvset):
        self.ctx, self.tv = ctx, tvset
        self.ctw = ctx.params.ct_words
        self.device = torch.device("cuda", torch.cuda.current_device())

    def new_wires(self, n_wires, T):
        return torch.zeros((n_wires * T, self.ctw), dtype=torch.int64, device=self.device)

    def upload(self, wires, row0, cts):
        flat = torch.from_numpy(np.ascontiguousarray(cts, np.uint64).reshape(-1, self.ctw).view(np.int64))
        wires[row0:row0 + flat.shape[0]].copy_(flat)

    def download(self, wires, rows):
        return wires[rows].cpu().numpy().view(np.uint64)

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def lincomb(self, wires, T, st):
        self.ctx.lincomb_dev(wires.data_ptr(), T, st["dst"], st["term_off"], st["srcs"], st["coefs"], st["consts"],
                             stream=self._stream())

    def bootstrap(self, wires, T, src, dst, table, s_begin, s_end):
        self.ctx.bootstrap_wires_dev(self.tv, wires.data_ptr(), T, src, dst, table, s_begin, s_end, stream=self._stream())


# --------------------------------------------------------------------------------------------
def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _all_gather_rows(send, world, group):
    """all-gather of equal-sized [rows, ctw] tensors -> [world*rows, ctw]"""
    if world == 1:
        return send
    out = torch.empty((world * send.shape[0], send.shape[1]), dtype=send.dtype, device=send.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, send.contiguous(), group=group)
    else:
        dist.all_gather(list(out.chunk(world)), send.contiguous(), group=group)
    return out


class GateShardedRunner:
    """Per-level all-gather.  Every rank calls `run` with the SAME input ciphertexts."""

    def __init__(self, low, backend, group=None):
        self.low, self.be, self.group = low, backend, group
        self.plan = plan_levels(low)
        self.collectives = 0

    def run(self, in_cts, T):
        be, plan = self.be, self.plan
        rank, world = _world(self.group)
        wires = be.new_wires(plan["n_wires"], T)
        be.upload(wires, 0, np.asarray(in_cts).reshape(plan["n_inputs"] * T, be.ctw))
        for L in range(plan["depth"] + 1):
            for st in plan["lin"][L]:
                be.lincomb(wires, T, st)                      # cheap, done redundantly on every rank
            if L == plan["depth"]:
                break
            b = plan["boot"][L]
            total = len(b["src"]) * T
            f0, f1, chunk = split_range(total, world, rank)
            for g0, g1, s0, s1 in rectangles(f0, f1, T):
                be.bootstrap(wires, T, b["src"][g0:g1], b["dst"][g0:g1], b["table"][g0:g1], s0, s1)
            if world > 1:
                # row of wire buffer for flattened index f of this level: dst[f // T] * T + f % T
                f = np.arange(world * chunk)
                valid = f < total
                rows_all = np.where(valid, np.asarray(b["dst"], np.int64)[np.minimum(f // T, len(b["dst"]) - 1)] * T + f % T, 0)
                mine = torch.from_numpy(rows_all[rank * chunk:(rank + 1) * chunk]).to(wires.device)
                gathered = _all_gather_rows(wires.index_select(0, mine), world, self.group)
                self.collectives += 1
                keep = torch.from_numpy(np.nonzero(valid)[0]).to(wires.device)
                wires.index_copy_(0, torch.from_numpy(rows_all[valid]).to(wires.device), gathered.index_select(0, keep))
        return self._outputs(wires, T)

    def _outputs(self, wires, T):
        be = self.be
        out = np.zeros((len(self.low["out_wire"]), T, be.ctw), np.uint64)
        for k, w in enumerate(self.low["out_wire"]):
            if w >= 0:
                out[k] = be.download(wires, slice(w * T, (w + 1) * T))
        return out


class SampleShardedRunner:
    """Whole program per rank on a slice of the samples; one all-gather of the outputs at the end."""

    def __init__(self, low, backend, group=None):
        self.low, self.be, self.group = low, backend, group
        self.plan = plan_levels(low)
        self.collectives = 0

    def run(self, in_cts, T):
        be, plan = self.be, self.plan
        rank, world = _world(self.group)
        s0, s1, chunk = split_range(T, world, rank)
        Tr = max(1, chunk)
        n_in, n_out = plan["n_inputs"], len(self.low["out_wire"])
        wires = be.new_wires(plan["n_wires"], Tr)
        local = np.zeros((n_in, Tr, be.ctw), np.uint64)
        local[:, :s1 - s0] = np.asarray(in_cts).reshape(n_in, T, be.ctw)[:, s0:s1]
        be.upload(wires, 0, local.reshape(n_in * Tr, be.ctw))
        for L in range(plan["depth"] + 1):
            for st in plan["lin"][L]:
                be.lincomb(wires, Tr, st)
            if L < plan["depth"]:
                b = plan["boot"][L]
                be.bootstrap(wires, Tr, b["src"], b["dst"], b["table"], 0, max(1, s1 - s0))
        rows = np.concatenate([np.arange(max(w, 0) * Tr, max(w, 0) * Tr + Tr) for w in self.low["out_wire"]]) \
            if n_out else np.zeros(0, np.int64)
        send = wires.index_select(0, torch.from_numpy(rows).to(wires.device))
        gathered = _all_gather_rows(send, world, self.group)
        self.collectives += 1 if world > 1 else 0
        g = gathered.cpu().numpy().view(np.uint64).reshape(world, n_out, Tr, be.ctw)
        out = np.zeros((n_out, T, be.ctw), np.uint64)
        for r in range(world):
            a, b_, _ = split_range(T, world, r)
            out[:, a:b_] = g[r, :, :b_ - a]
        for k, w in enumerate(self.low["out_wire"]):
            if w < 0:
                out[k] = 0
        return out
