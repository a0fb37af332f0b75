"""`python bench.py --gpus N` must work when run directly (the driver does exactly that): the parent becomes a launcher
before anything touches torch / HIP, starts one worker per GPU with the torchrun environment, relays rank 0's ONE JSON
line and fails when any worker fails.  Exercised here with the `--stub-cpu` workload (no GPU in this tier)."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(cmd, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.lstrip().startswith("{")]


@pytest.mark.parametrize("n", [1, 2, 3])
def test_direct_launch(n):
    r = _run([sys.executable, BENCH, "--gpus", str(n), "--steps", "2", "--warmup", "1", "--stub-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out["n_gpus"] == n and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["frames_total"] == n * 2 * 20          # SUM over ranks
    assert out["value"] == pytest.approx(out["config"]["frames_total"] / (out["ms_per_step"] * 2e-3), rel=1e-6)   # / MAX time
    if n > 1:
        assert out["config"]["master"].startswith("127.0.0.1:") and out["config"]["local_rank"] == 0


def test_failing_replica_fails_the_job():
    t0 = time.time()
    r = _run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--stub-cpu"], env={"BC_STUB_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert _json_lines(r.stdout) == []
    assert "rank 1 exit code 3" in r.stderr
    assert time.time() - t0 < 120, "the surviving replica must be stopped, not waited for"


def test_under_torchrun():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--stub-cpu"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["config"]["frames_total"] == 40


def test_world_size_mismatch_is_an_error():
    r = _run([sys.executable, BENCH, "--gpus", "1", "--stub-cpu"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_launcher_parent_never_imports_torch():
    """The launcher must stay GPU-free: it may not even import torch (checked in-process with an import blocker)."""
    code = (
        "import sys, builtins, runpy\n"
        "real = builtins.__import__\n"
        "def guard(name, *a, **k):\n"
        "    assert name.split('.')[0] != 'torch', 'launcher imported torch'\n"
        "    return real(name, *a, **k)\n"
        "builtins.__import__ = guard\n"
        f"sys.argv = [{BENCH!r}, '--gpus', '2', '--steps', '1', '--warmup', '0', '--stub-cpu']\n"
        f"runpy.run_path({BENCH!r}, run_name='__main__')\n")
    r = _run([sys.executable, "-c", code])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(_json_lines(r.stdout)) == 1


def test_eight_replicas_get_disjoint_cores_and_private_miopen_dirs(tmp_path):
    """The 8-GPU shape of the launcher (stub workload): every rank pins itself to its own slice of the host cores and gets its
    own MIOpen user-db / cache directory BEFORE torch is imported; rank 0's line carries per-rank fps with min / max."""
    r = _run([sys.executable, BENCH, "--gpus", "8", "--steps", "1", "--warmup", "0", "--stub-cpu"], env={"BC_BENCH_MIOPEN_DIR": str(tmp_path)}, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_lines(r.stdout)[0]
    assert out["n_gpus"] == 8 and out["config"]["frames_total"] == 8 * 20
    pr = out["per_rank_fps"]
    assert len(pr["all"]) == 8 and pr["min"] <= pr["max"] and pr["min"] > 0
    firsts = out["rank_env"]["first_core_of_each_rank"]
    n_cores = len(os.sched_getaffinity(0))
    if n_cores >= 8:
        assert len(set(firsts)) == 8, firsts          # disjoint slices
    assert out["rank_env"]["MIOPEN_USER_DB_PATH"].startswith(str(tmp_path)) and out["rank_env"]["MIOPEN_USER_DB_PATH"].endswith("rank0/db")
    for rank in range(8):
        assert os.path.isdir(tmp_path / f"rank{rank}" / "db") and os.path.isdir(tmp_path / f"rank{rank}" / "cache")


def test_sigterm_to_the_launcher_stops_the_workers():
    """A driver timeout ends the launcher with SIGTERM: the workers (own sessions) must not be orphaned holding GPUs."""
    import signal

    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    p = subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--steps", "100000", "--warmup", "0", "--stub-cpu"], env=e, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)

    def workers():
        out = subprocess.run(["ps", "-eo", "pid,ppid,args"], capture_output=True, text=True).stdout.splitlines()
        return [int(l.split()[0]) for l in out if len(l.split()) > 2 and l.split()[1] == str(p.pid) and "bench.py" in l]

    deadline = time.time() + 60
    while time.time() < deadline and len(workers()) < 2:
        time.sleep(0.2)
    kids = workers()
    assert len(kids) == 2, kids
    time.sleep(1.0)
    p.send_signal(signal.SIGTERM)
    p.wait(timeout=60)
    assert p.returncode != 0
    deadline = time.time() + 30
    alive = kids
    while time.time() < deadline and alive:
        alive = [k for k in alive if os.path.exists(f"/proc/{k}") and "zombie" not in open(f"/proc/{k}/status").read().lower()]
        time.sleep(0.2)
    assert not alive, f"workers survived the launcher: {alive}"
