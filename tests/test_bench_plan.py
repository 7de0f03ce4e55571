"""bench.py --dry-run (no GPU): the phases an invocation will go through, each with its watchdog budget -- what the first 8-GPU
run of the N > 1 path will execute, readable before it is made."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plan(args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"] + args, capture_output=True, text=True, timeout=120, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["dry_run"] is True
    return d


def test_default_line_plan():
    d = _plan([])
    names = [p["phase"] for p in d["phases"]]
    assert d["n_gpus"] == 1 and names[0] == "main" and names[-1] == "cpu_baseline"
    assert {"step_autoreset", "classic_path", "secondary:xworld7_ego3", "secondary:simple_game", "secondary:simple_race", "parity"} <= set(names)
    assert all(p["budget_s"] is None for p in d["phases"])               # N = 1: nothing can hang on a collective
    assert "secondary:simple_game" not in [p["phase"] for p in _plan(["--no-secondary"])["phases"]]
    assert not any(n.startswith("secondary:") for n in [p["phase"] for p in _plan(["--envs-per-gpu", "2048"])["phases"]])


def test_eight_gpu_plan_with_and_without_a_launcher():
    for d in (_plan(["--gpus", "8"]), _plan(["--gpus", "8"], {"WORLD_SIZE": "8", "RANK": "0"})):
        names = [p["phase"] for p in d["phases"]]
        assert d["n_gpus"] == 8
        assert names == ["main", "step_autoreset", "screens_gather:screens", "screens_gather:grids", "screens_gather:grids_nodraw", "c5", "parity",
                         "teardown"]
        budgets = {p["phase"]: p["budget_s"] for p in d["phases"]}
        assert budgets["main"] == 720.0 and budgets["c5"] == 480.0 and budgets["screens_gather:grids"] == 240.0 and budgets["parity"] is None
    two = [p["phase"] for p in _plan(["--gpus", "2", "--gather", "screens", "--phase-timeout", "100"])["phases"]]
    assert "c5" not in two and "screens_gather:grids" not in two and "screens_gather:screens" in two
    # rank 1 of a launcher prints nothing
    e = dict(os.environ, WORLD_SIZE="2", RANK="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--gpus", "2"], capture_output=True, text=True, timeout=120, env=e)
    assert out.returncode == 0 and out.stdout.strip() == ""
