"""bench.py --gpus N is honoured or refused, never silently reduced (BASELINE configs[4]).

The reference has replicas only (rl_training/rl_games/runner.py:260-265): the sharded job is this build's own entry point,
so its launcher is tested here -- the refusals on CPU, the self-spawn path on a GPU box."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(argv, env_extra=None, timeout=900):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def json_lines(stdout):
    out = []
    for line in stdout.splitlines():
        line = line.strip()
        if line.startswith("{"):
            try:
                out.append(json.loads(line))
            except ValueError:
                pass
    return out


def test_more_gpus_than_devices_is_refused_loudly():
    """no device (this container) or one device (the GPU box): --gpus 2 exits non-zero naming both numbers, no JSON line"""
    import torch

    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    want = visible + 1 if visible else 2
    r = run_bench(["--gpus", str(want), "--steps", "5", "--warmup", "1"], timeout=300)
    assert r.returncode != 0
    assert f"--gpus {want}" in r.stderr and f"{visible} HIP device(s) visible" in r.stderr, r.stderr[-2000:]
    assert json_lines(r.stdout) == []


def test_world_size_mismatch_is_refused_loudly():
    r = run_bench(["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"}, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "WORLD_SIZE=4" in r.stderr
    assert json_lines(r.stdout) == []
    r = run_bench(["--gpus", "8"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and json_lines(r.stdout) == []


@pytest.mark.gpu
def test_self_spawn_path_prints_the_same_line():
    """AGX_BENCH_SPAWN=1 sends --gpus 1 through the launcher (torch.distributed.run, one rank): ONE JSON line on stdout with the
    keys of the direct run, n_gpus 1, and the rank's device diagnostics on stderr"""
    argv = ["--gpus", "1", "--steps", "50", "--warmup", "10", "--no-cpu-baseline", "--no-depth", "--no-lidar", "--no-strict"]
    direct = run_bench(argv)
    assert direct.returncode == 0, direct.stderr[-3000:]
    spawned = run_bench(argv, {"AGX_BENCH_SPAWN": "1"})
    assert spawned.returncode == 0, spawned.stderr[-3000:]
    a, b = json_lines(direct.stdout), json_lines(spawned.stdout)
    assert len(a) == 1 and len(b) == 1
    a, b = a[0], b[0]
    assert a["n_gpus"] == b["n_gpus"] == 1 and a["metric"] == b["metric"] and a["config"] == b["config"]
    assert set(a) == set(b)
    assert "torch.distributed.run" in spawned.stderr and "[bench rank 0/1]" in spawned.stderr and '"can_access_peer"' in spawned.stderr
    assert 0.2 < b["value"] / a["value"] < 5.0


@pytest.mark.gpu
def test_exchange_selftest_only_prints_one_verdict_per_backend():
    """`bench.py --gpus N --exchange-selftest-only` builds every exchange backend, runs checksum-verified exchange
    steps and prints ONE JSON verdict per backend, no timing.  Here: a world of one through the real code path (RCCL communicator of
    one rank, the rank's own IPC-free receive buffer), resets inside the steps."""
    r = run_bench(["--gpus", "1", "--exchange-selftest-only", "--selftest-steps", "300", "--num-envs", "2048"], timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = json_lines(r.stdout)
    assert [v["backend"] for v in lines] == ["process_group", "peer_push", "rccl_thread"], r.stdout[-2000:]
    for v in lines:
        assert v["selftest"] == "exchange" and v["ok"] is True and v["world"] == 1 and v["steps"] == 300, v
        (pr,) = v["per_rank"]
        assert pr["own_slice_ok"] and pr["first_bad_step_by_sender"] == {} and pr["resets"] > 300 * 2  # ~4 truncations per step at 2048 envs


@pytest.mark.gpu
def test_sharded_run_starts_with_the_exchange_preflight():
    """whatever command a multi-GPU lease runs first, the exchange verdicts come first.  A timed run with more
    than one rank (here: the sharded code path in a world of one, AGX_BENCH_FORCE_DIST + AGX_BENCH_PREFLIGHT) runs
    `--preflight-steps` checksum-verified exchange steps per backend before anything is timed: verdicts on stderr, their summary in
    the ONE stdout line (`exchange.preflight`), the timed legs after it."""
    r = run_bench(["--gpus", "1", "--steps", "50", "--warmup", "10", "--no-cpu-baseline", "--no-depth", "--no-lidar", "--no-strict",
                   "--preflight-steps", "60"], {"AGX_BENCH_FORCE_DIST": "1", "AGX_BENCH_PREFLIGHT": "1"}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    (line,) = json_lines(r.stdout)
    pre = line["exchange"]["preflight"]
    assert [p["backend"] for p in pre] == ["process_group", "peer_push", "rccl_thread"]
    assert all(p["ok"] is True and p["steps"] == 60 and p["first_failure"] is None for p in pre), pre
    assert r.stderr.count("[bench preflight]") == 3
    assert line["value"] > 1e6 and len(r.stdout) < 8192


def _bench_module():
    sys.path.insert(0, ROOT)
    import bench

    return bench


@pytest.mark.parametrize("name", ["r05_bench_driver_style", "r05_bench_default", "r05_bench_forced_dist_world1", "r05_bench_lidar"])
def test_stdout_line_stays_parseable_from_the_drivers_tail(name):
    """the driver keeps ~9 KB of stdout and parses the ONE JSON line out of it; round 5's 23 KB line came
    back as "parsed": null.  The line built from round 5's full records is < 8 KB, parses from its own last 8 KB, and still
    carries the contract's keys and the two objects the contract adds."""
    bench = _bench_module()
    full = json.load(open(os.path.join(ROOT, "profiles", name + ".json")))
    assert len(json.dumps(full)) > 4000  # the canned record is the big one
    line = json.dumps(bench.compact_line(full, "bench_detail.json")) + "\n"
    assert len(line) < bench.LINE_LIMIT < 8192
    d = json.loads(line[-8192:])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["value"] == pytest.approx(full["value"], rel=1e-5)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4)
    assert all(not isinstance(v, (dict, list)) for v in r.values())  # nothing nested
    if "cpu_baseline" in full:
        assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert d["detail"] == "bench_detail.json"


def test_stdout_line_guard_drops_secondary_objects_never_the_contract():
    bench = _bench_module()
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_driver_style.json")))
    full["config"]["num_envs_per_gpu"] = "x" * 1500  # something a later edit let grow
    full["plus_depth"]["exchange"] = "y" * 2500
    full["plus_lidar"]["exchange"] = "z" * 2500
    line = bench.compact_line(full, "bench_detail.json")
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    assert "dropped_for_size" in line and "roofline" in line and "cpu_baseline" in line and "value" in line


def test_a_failed_preflight_still_prints_one_parseable_line():
    """the line bench.py prints when the exchange preflight of an N > 1 run hangs or fails (no timing happened): small, parseable,
    names the failing leg"""
    bench = _bench_module()
    pre = [{"backend": "process_group", "ok": True, "steps": 200, "seconds": 1.2, "first_failure": None},
           {"backend": "peer_push", "ok": False, "steps": 200, "seconds": 60.0,
            "first_failure": {"rank": 3, "error": "RuntimeError: " + "x" * 900, "failed_in": "run", "first_bad_step_by_sender": {"5": 17}}}]
    out = {"metric": "env-steps/sec (not measured: the exchange preflight failed)", "value": None, "n_gpus": 8,
           "error": "exchange preflight: backend peer_push failed", "exchange": {"preflight": pre}}
    line = json.dumps(bench.compact_line(out, None))
    d = json.loads(line)
    assert len(line) < bench.LINE_LIMIT and d["value"] is None and d["n_gpus"] == 8 and "peer_push" in d["error"]
    assert [p["backend"] for p in d["exchange"]["preflight"]] == ["process_group", "peer_push"]
    assert d["exchange"]["preflight"][1]["ok"] is False and "rank" in d["exchange"]["preflight"][1]["first_failure"]
