"""`python bench.py --gpus N` as the driver starts it (no WORLD_SIZE in the environment): the parent must start the
ranks itself, as child processes, and relay their outcome.  On this GPU-less host the ranks meet (gloo) and then each
fails cleanly at `fbs_ctx_create` with FBS_E_DEVICE -- there is no CPU path to fall back to."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_2_launches_its_own_ranks_and_reports_their_failure():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the launcher is exercised by the real multi-GPU bench")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                         "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert rc.returncode != 0
    assert rc.stdout.strip() == ""                                 # no JSON line is invented
    assert "FBS_E_DEVICE" in rc.stderr and "the 2-rank run failed" in rc.stderr
    assert rc.stderr.count('"rank": ') == 2                         # both ranks got as far as the context


def test_rank_mode_refuses_a_mismatched_world():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                        timeout=300)
    assert rc.returncode != 0 and "WORLD_SIZE=4 but --gpus 2" in rc.stderr
