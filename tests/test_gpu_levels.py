"""The loaded program, one level at a time (include/fbs_exec.h "a loaded program, one level at a time"): what the
multi-GPU runners are built from.  Shared key switches, wire-slot reuse, slices of a level into contiguous rows and
back, the device-buffer whole-program call, and the two runners on the nccl backend (RCCL) with one rank -- all
against `fbs_eval` and the CPU oracle, word for word."""
import os
import socket

import numpy as np
import pytest

from oracle import lut_oracle, tfhe_oracle as orc
from tests.helpers import load_fixture, oracle_eval_program, subsample, toy_glwe

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from tfhe_fbs_map_amd import _native
    return _native


def load(nat, toy_params, name, T, seed=6, merge=True, k=1):
    from tfhe_fbs_map_amd import parse_fbs
    rec = load_fixture(name)
    ops, outs = lut_oracle.read_fbs(rec["fbs"])
    tables = [op[3] for op in ops if op[0] == "boot"]
    p = max(7, max(len(t) for t in tables))
    prm = toy_params.replace(p_msg=p) if k == 1 else toy_glwe(k, p)
    ctx = nat.Context(prm, seed=seed)
    env = parse_fbs(rec["fbs"], inputs=rec["program_inputs"], merge_linear_prods=merge)
    low = env.lower()
    ins, expect = subsample(rec, T)
    cts = ctx.encrypt(np.stack([ins[n] for n in low["input_names"]]), nonce0=9)
    prog = nat.Program(ctx, ctx.tvset(low["tables"]), len(low["input_names"]), low["kind"], low["arg0"], low["arg1"],
                       low["const_coef"], low["term_coef"], low["term_src"], low["out_wire"])
    return rec, ctx, prm, low, cts, prog, expect, (ops, outs)


@pytest.mark.parametrize("name", ["adder8__basic_p2", "2_input_gates__basic_p2", "half_adder__basic_p2"])
def test_gates_that_share_a_source_share_its_key_switch(nat, toy_params, name):
    """The reference's one-gate-one-bootstrap lowering puts several tables on one linear combination
    (map_to_fbs.py:41-45; its CSE only merges identical tables, fbs_exec_env.py:93-100): the loaded program runs ONE
    key switch + modulus switch per distinct source of a level.  Exact: the ciphertexts are those of the oracle, which
    key-switches per bootstrap."""
    T = 3
    rec, ctx, prm, low, cts, prog, expect, (ops, outs) = load(nat, toy_params, name, T)
    assert prog.n_keyswitch < prog.n_bootstrap, (prog.n_keyswitch, prog.n_bootstrap)
    assert sum(prog.level_sources) == prog.n_keyswitch and sum(prog.level_width) == prog.n_bootstrap
    distinct = len({(op[2]) for op in ops if op[0] == "boot"})
    assert prog.n_keyswitch == distinct                     # every source sits in one level only
    got = prog.eval(cts, T)
    o = orc.Oracle(prm, seed=6)
    wires = oracle_eval_program(o, ops, outs, {n: cts[i] for i, n in enumerate(low["input_names"])})
    for k, (out_name, src) in enumerate(outs):
        if src not in ("0", "1"):
            assert np.array_equal(got[k], wires[src]), out_name
            assert np.array_equal(ctx.decrypt(got[k]), expect[out_name])


def test_key_switch_count_is_the_number_of_distinct_sources(nat, toy_params):
    """The heuristic mappers rarely put two tables on one linear combination: whatever they did, the
    loaded program switches each distinct source once."""
    for name in ("aes_sbox__search_p7", "mul4__naive_p7"):
        _, _, _, _, _, prog, _, (ops, _) = load(nat, toy_params, name, 1)
        boots = [op for op in ops if op[0] == "boot"]
        assert prog.n_bootstrap == len(boots) and prog.n_keyswitch == len({op[2] for op in boots})


@pytest.mark.parametrize("name", ["adder8__search_p7", "mul4__naive_p7", "adder8__basic_p2"])
def test_wire_slots_are_reused(nat, toy_params, name):
    _, _, _, low, _, prog, _, _ = load(nat, toy_params, name, 1)
    n_wires = len(low["input_names"]) + len(low["kind"])
    assert prog.n_slots < n_wires
    assert len(set(prog.in_slot.tolist())) == len(low["input_names"])          # inputs never share a slot
    live_outputs = [s for s in prog.out_slot.tolist() if s >= 0]
    assert len(set(live_outputs)) == len(set(w for w in low["out_wire"] if w >= 0))


@pytest.mark.parametrize("name,T,k", [("adder8__basic_p2", 5, 1), ("adder8__search_p7", 4, 1), ("edge_outputs", 3, 1), ("aes_sbox__basic_p2", 2, 1),
                                      ("adder8__search_p7", 4, 2), ("adder8__basic_p2", 5, 2), ("edge_outputs", 3, 2),
                                      ("adder8__search_p7", 4, 3), ("adder8__basic_p2", 5, 3)])
def test_levels_in_slices_through_rows_and_back(nat, toy_params, name, T, k):
    """Every level cut into three ragged slices (cuts inside a gate's samples and between gates that share a source),
    each slice bootstrapped into a contiguous row buffer and scattered back -- the gate-sharded data path on one GPU
    -- equals fbs_eval word for word.  The wire buffer has a larger sample stride than the samples in use.
    k = 2: the same with GLWE dimension 2 (ciphertexts and rows of 2 N + 1 words), fbs_eval itself held to the oracle."""
    import torch
    rec, ctx, prm, low, cts, prog, expect, (ops, outs) = load(nat, toy_params, name, T, k=k)
    ref = prog.eval(cts, T)
    if k == 2:
        assert prm.k == 2 and prm.ct_words == 2 * prm.N + 1
        wires_o = oracle_eval_program(orc.Oracle(prm, seed=6), ops, outs, {n: cts[i] for i, n in enumerate(low["input_names"])})
        for j, (out_name, src) in enumerate(outs):
            if src not in ("0", "1"):
                assert np.array_equal(ref[j], wires_o[src]), out_name
                assert np.array_equal(ctx.decrypt(ref[j]), expect[out_name])
    ctw = prm.ct_words
    stride = T + 2
    wires = torch.zeros((prog.n_slots, stride, ctw), dtype=torch.int64, device="cuda")
    d_in = torch.from_numpy(cts.view(np.int64)).cuda()
    wires[torch.from_numpy(prog.in_slot.astype(np.int64)).cuda(), :T] = d_in
    for L in range(prog.depth + 1):
        prog.level_lincomb_dev(L, wires.data_ptr(), stride, 0, T)
        if L == prog.depth:
            break
        total = prog.level_width[L] * T
        cuts = sorted({0, total, min(total, max(0, total // 3 + 1)), min(total, (2 * total) // 3 + (1 if T > 1 else 0))})
        for f0, f1 in zip(cuts, cuts[1:]):
            rows = torch.full((f1 - f0, ctw), -1, dtype=torch.int64, device="cuda")
            prog.level_bootstrap_dev(L, wires.data_ptr(), stride, 0, T, f0, f1, d_rows=rows.data_ptr())
            prog.level_scatter_dev(L, wires.data_ptr(), stride, 0, T, rows.data_ptr(), f0, f1)
    ctx.sync()
    got = wires.cpu().numpy().view(np.uint64)
    for k, slot in enumerate(prog.out_slot.tolist()):
        if slot >= 0:
            assert np.array_equal(got[slot, :T], ref[k]), low["out_names"][k]


def test_eval_on_device_buffers_equals_eval_on_host_buffers(nat, toy_params):
    import torch
    T = 6
    rec, ctx, prm, low, cts, prog, expect, _ = load(nat, toy_params, "edge_outputs", T)
    ref = prog.eval(cts, T)
    d_in = torch.from_numpy(cts.view(np.int64)).cuda()
    d_out = torch.empty((prog.n_outputs, T, prm.ct_words), dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()                       # a second stream: the scratch hand-over is ordered on the device
    prog.eval_dev(d_in.data_ptr(), T, d_out.data_ptr())
    with torch.cuda.stream(side):
        d_out2 = torch.empty_like(d_out)
        prog.eval_dev(d_in.data_ptr(), T, d_out2.data_ptr(), stream=side.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), ref)
    assert np.array_equal(d_out2.cpu().numpy().view(np.uint64), ref)


@pytest.fixture(scope="module")
def one_rank_nccl():
    """torch.distributed on the nccl backend (= RCCL) with world size 1: the collective code path of the runners on a
    real GPU.  More ranks need more GPUs than a test box has; the partition logic is covered on gloo (CPU)."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists")
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("name,T,k", [("adder8__search_p7", 5, 1), ("adder8__basic_p2", 4, 1), ("edge_outputs", 3, 1), ("adder8__search_p7", 5, 2), ("adder8__search_p7", 5, 3)])
def test_runners_on_rccl_equal_program_eval(nat, toy_params, one_rank_nccl, name, T, k):
    """GateShardedRunner (send rows -> all_gather_into_tensor -> scatter, forced even with one rank) and
    SampleShardedRunner on the nccl backend == fbs_eval, word for word (k = 2: rows of 2 N + 1 words)."""
    from tfhe_fbs_map_amd.distributed import GateShardedRunner, GpuBackend, SampleShardedRunner
    rec, ctx, prm, low, cts, prog, expect, _ = load(nat, toy_params, name, T, k=k)
    ref = prog.eval(cts, T)
    const = np.array([w < 0 for w in low["out_wire"]])
    gate = GateShardedRunner(GpuBackend(prog), always_gather=True)
    got = gate.run(cts, T)
    assert gate.collectives == prog.depth
    assert np.array_equal(got[~const], ref[~const])
    sample = SampleShardedRunner(GpuBackend(prog))
    assert np.array_equal(sample.run(cts, T)[~const], ref[~const])
    plain = GateShardedRunner(GpuBackend(prog))                 # one rank, no collective: straight into the slots
    assert np.array_equal(plain.run(cts, T)[~const], ref[~const]) and plain.collectives == 0
    for k, name_ in enumerate(low["out_names"]):
        if not const[k]:
            assert np.array_equal(ctx.decrypt(got[k]), expect[name_])


def test_samples_are_chunked_when_the_wire_slots_do_not_fit(nat, toy_params, monkeypatch):
    """fbs_eval / fbs_eval_dev evaluate the samples in chunks when the wire buffer would not fit in HBM.  With the budget
    capped (test hook FBS_WIRE_BUDGET_MB) 37 samples run as several ragged chunks and must equal the one-chunk result."""
    import torch
    T = 37
    rec, ctx, prm, low, cts, prog, expect, _ = load(nat, toy_params, "adder8__search_p7", T)
    ref = prog.eval(cts, T)
    monkeypatch.setenv("FBS_WIRE_BUDGET_MB", "2")               # 2 MB * 0.6 / (slots * 8200 B): a handful of samples
    per_sample = prog.n_slots * prm.ct_words * 8
    assert 2 * 2**20 * 0.6 / per_sample < T / 3
    ctx2 = nat.Context(prm, seed=6)                              # a fresh context: its wire buffer has not grown yet
    prog2 = nat.Program(ctx2, ctx2.tvset(low["tables"]), len(low["input_names"]), low["kind"], low["arg0"], low["arg1"],
                        low["const_coef"], low["term_coef"], low["term_src"], low["out_wire"])
    assert np.array_equal(prog2.eval(cts, T), ref)
    d_in = torch.from_numpy(cts.view(np.int64)).cuda()
    d_out = torch.empty((prog2.n_outputs, T, prm.ct_words), dtype=torch.int64, device="cuda")
    prog2.eval_dev(d_in.data_ptr(), T, d_out.data_ptr())
    ctx2.sync()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), ref)
    for k, name in enumerate(low["out_names"]):
        assert np.array_equal(ctx2.decrypt(ref[k]), expect[name])


def test_long_level_into_rows_runs_as_whole_rounds_plus_a_remainder(nat):
    """A level of the benchmark shape longer than a round is launched as whole rounds in whole-CU workgroups and the rest as a
    launch of its own (csrc/fbs_blind_rotate.hip, whole_cu_share) -- also when the results go to a contiguous row buffer
    (the send buffer of the gate-sharded mode) and when the range starts in the middle of the level."""
    import torch
    from tfhe_fbs_map_amd import P1024, parse_fbs
    prm = P1024.replace(n=16, p_msg=7)                     # l = 3, beta = 7 at N = 1024: the benchmark kernels, short rotation
    text = "m1 = 1 * a + 2 * b\nm2 = Bootstrap(m1, [0, 1, 1, 0])\nm3 = Bootstrap(m2, [1, 0])\nOutput x = m2\nOutput y = m3\n"
    env = parse_fbs(text, inputs=["a", "b"])
    low = env.lower()
    ctx = nat.Context(prm, seed=2)
    prog = nat.Program(ctx, ctx.tvset(low["tables"]), 2, low["kind"], low["arg0"], low["arg1"], low["const_coef"],
                       low["term_coef"], low["term_src"], low["out_wire"])
    T = 1700                                               # one round (1024) + 676
    rng = np.random.default_rng(3)
    bits = rng.integers(0, 2, (2, T))
    cts = ctx.encrypt(bits, nonce0=1)
    ref = prog.eval(cts, T)
    assert np.array_equal(ctx.decrypt(ref[0]), np.array([0, 1, 1, 0])[bits[0] + 2 * bits[1]])
    ctw = prm.ct_words
    wires = torch.zeros((prog.n_slots, T, ctw), dtype=torch.int64, device="cuda")
    wires[torch.from_numpy(prog.in_slot.astype(np.int64)).cuda()] = torch.from_numpy(cts.view(np.int64)).cuda()
    for L in range(prog.depth + 1):
        prog.level_lincomb_dev(L, wires.data_ptr(), T, 0, T)
        if L == prog.depth:
            break
        total = prog.level_width[L] * T
        for f0, f1 in ((0, 1300), (1300, total)):          # 1300 = a round and 276; the second range starts mid-level
            rows = torch.full((f1 - f0, ctw), -1, dtype=torch.int64, device="cuda")
            prog.level_bootstrap_dev(L, wires.data_ptr(), T, 0, T, f0, f1, d_rows=rows.data_ptr())
            prog.level_scatter_dev(L, wires.data_ptr(), T, 0, T, rows.data_ptr(), f0, f1)
    ctx.sync()
    got = wires.cpu().numpy().view(np.uint64)
    for k, slot in enumerate(prog.out_slot.tolist()):
        assert np.array_equal(got[slot], ref[k]), low["out_names"][k]
