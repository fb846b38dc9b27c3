"""CPU: bench.py's own main() driven through the --gpus 2 control flow (tests/bench_rehearsal.py)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(nproc, extra_env=None, gpus=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "tests" / "bench_rehearsal.py"), "--gpus", str(gpus or nproc),
           "--steps", "2", "--warmup", "1", "--rows", "3000", "--dim", "64", "--nq", "9", "--k", "5",
           "--no-extra", "--no-encoder", "--no-cpu-baseline"]
    return subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)


def test_two_rank_control_flow_emits_one_json_line_from_rank_0():
    r = _run(2)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["unit"] == "queries/s" and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert out["value"] > 0 and abs(out["value"] - 9 * 2 / (out["ms_per_step"] * 2 / 1e3)) < 1e-6 * out["value"]
    cfg = out["config"]
    assert cfg["rows_total"] == 6000 and cfg["rows_per_gpu"] == 3000 and "workload" in cfg   # weak scaling: rows PER rank
    strong = cfg["strong_10M"]
    assert strong["rows_total"] == 3000 and strong["rows_per_gpu"] == 1500 and strong["scaling"] == "strong"
    assert "NOT a measurement" in out["data"]              # a rehearsal can never be mistaken for a bench line


def test_a_failing_rank_fails_the_job_and_names_itself():
    r = _run(2, {"CSS_REHEARSAL_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert "rank 1 failed" in r.stderr and "injected failure" in r.stderr


def test_gpus_flag_must_match_the_launch():
    r = _run(1, gpus=2)                                     # --gpus 2 under a 1-rank launch
    assert r.returncode != 0 and "needs torch.distributed.run with 2 ranks" in r.stderr


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` as the driver types it for N = 1 (no torch.distributed.run, WORLD_SIZE unset): the
    parent starts the two ranks as child processes before touching any device, relays rank 0's one JSON line and the
    job's exit status."""
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "3000", "--dim", "64", "--nq", "9", "--k", "5",
            "--no-extra", "--no-encoder", "--no-cpu-baseline"]
    cmd = [sys.executable, str(ROOT / "tests" / "bench_rehearsal.py")] + args
    r = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["rows_total"] == 6000 and "starting 2 ranks" in r.stderr
    # ... and a failing rank still fails the whole command
    r = subprocess.run(cmd, cwd=str(ROOT), env=dict(env, CSS_REHEARSAL_FAIL_RANK="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "rank 1 failed" in r.stderr


def test_debug_switches_refuse_the_headline():
    env = dict(os.environ, CSS_KNN_DBG="1", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, str(ROOT / "tests" / "bench_rehearsal.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--rows", "2000",
           "--dim", "64", "--nq", "5", "--k", "5", "--no-extra", "--no-encoder", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "CSS_KNN_DBG=1" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    env.pop("CSS_KNN_DBG")
    env["CSS_KNN_GROWTH"] = "4"
    r = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert "CSS_KNN_GROWTH=4" in out["env_overrides"] and "invalid" not in out
