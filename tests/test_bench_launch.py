"""bench.py's launch logic on CPU: `python bench.py --gpus 2` with no torchrun environment must start two ranks
itself and print ONE line with n_gpus = 2; it must refuse — not shrink — when the devices are not there; a native-communicator
bootstrap that fails, hangs or fails its probe on one rank must end in the torch.distributed carrier on every rank.
The engine is tests/bench_stub.py (HIPADJ_BENCH_STUB=1, gloo): no kernel runs and the line says so."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def _lines(stdout):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.startswith("{")]


def _expected_dp(ntraj):
    sys.path.insert(0, ROOT)
    import bench
    u0, _ = bench.inputs(ntraj)
    return u0.sum(axis=0)


OVERLAPPED = "rccl on the handle's second stream, overlapped with the next pass (hipadj_comm_overlap)"


def test_native_carrier_can_stay_in_stream():
    """HIPADJ_COMM_OVERLAP=0: the collective stays on the handle's stream (the round-3 behaviour); the default moves it to the second stream (next test)."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--ntraj", "101", "--no-cpu-baseline", "--no-extras"],
             {"HIPADJ_BENCH_STUB": "1", "HIPADJ_BENCH_STUB_COMM": "ok", "HIPADJ_COMM_TIMEOUT": "5", "HIPADJ_COMM_OVERLAP": "0"})
    assert r.returncode == 0, r.stderr[-2000:]
    ln = _lines(r.stdout)[0]
    assert ln["config"]["dp_allreduce"] == "rccl in-stream (hipadj_comm)"
    np.testing.assert_allclose(ln["stub_dp"], _expected_dp(101), rtol=1e-13)


@pytest.mark.parametrize("comm,carrier,ranks", [("ok", OVERLAPPED, 2), ("fail", "torch.distributed nccl, async", 2),
                                                ("hang", "torch.distributed nccl, async", 2), ("badcheck", "torch.distributed nccl, async", 2)])
def test_gpus_2_without_torchrun_starts_two_ranks_and_prints_one_line(comm, carrier, ranks):
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--ntraj", "101", "--no-cpu-baseline"],
             {"HIPADJ_BENCH_STUB": "1", "HIPADJ_BENCH_STUB_COMM": comm, "HIPADJ_COMM_TIMEOUT": "5"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _lines(r.stdout)
    assert len(lines) == 1, r.stdout
    ln = lines[0]
    assert ln["n_gpus"] == 2 and ln["metric"] == "STUB_no_kernel_ran" and "STUB" in ln["data"]
    assert ln["config"]["dp_allreduce"] == carrier and ln["config"]["rccl_ranks"] == ranks
    assert (ln["config"]["native_allreduce_fallback"] is None) == (comm == "ok")
    assert ln["config"]["ntraj_per_gpu"] == 51 and ln["config"]["ntraj_total"] == 101          # strong scaling: the ensemble is sharded
    np.testing.assert_allclose(ln["stub_dp"], _expected_dp(101), rtol=1e-13)                   # the all-reduce spanned both shards
    assert ln["weak_scaling"]["ntraj_total"] == 202


def test_the_stdout_line_is_compact_strict_json_and_the_tables_go_to_the_extras_file(tmp_path):
    """round 5's 20 KB line was not parsed by the driver.  The line is < 4096 bytes of strict JSON (no NaN / Infinity literals) with the contract's keys;
    a full-sized result (the shape of a real N = 1 run, with loss_paths / other_configs / prose notes) still compacts below the limit, and the extras file holds everything."""
    extras = str(tmp_path / "extras.json")
    r = _run(["--steps", "3", "--warmup", "1", "--ntraj", "101", "--no-cpu-baseline", "--no-extras"], {"HIPADJ_BENCH_STUB": "1", "HIPADJ_BENCH_EXTRAS": extras})
    assert r.returncode == 0, r.stderr[-2000:]
    out = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(out) == 1 and len(out[0]) < 4096

    def strict(x):
        raise ValueError(f"non-strict JSON constant {x}")
    ln = json.loads(out[0], parse_constant=strict)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in ln, k
    assert "workload" in ln["config"] and {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(ln["roofline"])
    full = json.load(open(extras))
    assert full["roofline"]["kernel_ms_note"] and ln["extras"] == extras
    sys.path.insert(0, ROOT)
    import bench
    big = dict(full)
    big["cpu_baseline"] = dict(value=149066.1234567, unit="trajectories/s", cores=16, kind="port", sample="x" * 900, sample_short="10000 trajectories x 30 repeats", ns_per_vjp_step=1.68,
                               single_thread_value=9360.0, repeats=30, spread_rel=0.05, cores_note="y" * 500)
    big["cold_burst"] = dict(ms_per_step=0.13, value=7.6e7, unit="trajectories/s", whole_pass_frac=0.46)
    big["shard_sizes"] = [dict(ntraj=n, gpus_of_layout=10000 // n, ms_per_step=0.03, trajectories_per_s=1e7, kernel_ms=0.03, time_segments=20, launches_per_pass=1,
                               implied_speedup_if_allreduce_hidden=4.0) for n in (1250, 2500, 5000)]
    big["other_configs"] = [{"config": "z" * 300, "note": "w" * 500} for _ in range(30)]
    big["loss_paths"] = {"note": "v" * 5000}
    big["roofline"]["traffic"] = 548.9e6
    big["roofline"]["traffic_source"] = "t" * 600
    line = bench.compact(big)
    assert len(line) < 4096
    back = json.loads(line, parse_constant=strict)
    assert back["cpu_baseline"]["cores"] == 16 and back["cpu_baseline"]["kind"] == "port" and back["roofline"]["traffic"] == 548.9e6
    assert "other_configs" not in back and "loss_paths" not in back and len(back["shard_sizes"]) == 3


def test_gpus_n_refuses_instead_of_running_fewer():
    # no stub: the real path on a machine without (enough) HIP devices — must exit non-zero before any line is printed
    r = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"], {})
    assert r.returncode != 0 and not _lines(r.stdout)
    assert "HIP device" in r.stderr


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"HIPADJ_BENCH_STUB": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not _lines(r.stdout)
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "1"], {"HIPADJ_BENCH_STUB": "1", "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and not _lines(r.stdout)


def test_a_worker_that_dies_after_the_headline_still_yields_the_line():
    """bench.py runs the measurement in a child of a GPU-free supervisor (bench.supervise): the child checkpoints its result once the headline is complete; when it is killed
    later — a GPU fault in a secondary figure makes ROCr abort the process, which no `except` sees — the supervisor prints the checkpoint and says which part is missing."""
    common = ["--steps", "3", "--warmup", "1", "--ntraj", "101", "--no-cpu-baseline", "--no-extras"]
    r = _run(common, {"HIPADJ_BENCH_STUB": "1", "HIPADJ_BENCH_TEST_DIE": "after_headline"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["steps"] == 3
    assert lines[0]["secondary_figures_incomplete"]["worker_exit_code"] != 0
    ok = _run(common, {"HIPADJ_BENCH_STUB": "1"})                                   # the undisturbed run: the same line without the note
    assert ok.returncode == 0 and "secondary_figures_incomplete" not in _lines(ok.stdout)[0]
    assert {k for k in _lines(ok.stdout)[0]} == {k for k in lines[0]} - {"secondary_figures_incomplete"}
    direct = _run(common, {"HIPADJ_BENCH_STUB": "1", "HIPADJ_BENCH_SUPERVISE": "0", "HIPADJ_BENCH_TEST_DIE": "after_headline"})   # without the supervisor the death is what the caller sees
    assert direct.returncode != 0 and not _lines(direct.stdout)


def test_ending_the_supervisor_ends_the_worker():
    """A launcher that terminates a rank (torchrun on another rank's failure, the driver's timeout) signals the SUPERVISOR: the request is passed on, and a worker whose supervisor
    is killed outright is killed with it (PR_SET_PDEATHSIG) — no orphan keeps a GPU."""
    import signal
    import time
    import psutil
    env = dict(os.environ, HIPADJ_BENCH_STUB="1", HIPADJ_BENCH_TEST_DIE="sleep")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for how in (signal.SIGTERM, signal.SIGKILL):
        sup = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--ntraj", "101", "--no-cpu-baseline", "--no-extras"],
                               env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        worker = None
        for _ in range(300):                       # the worker appears, then reaches its sleep (the stub's headline takes a second or two)
            kids = psutil.Process(sup.pid).children()
            if kids:
                worker = kids[0]
                break
            time.sleep(0.1)
        assert worker is not None
        time.sleep(3.0)
        sup.send_signal(how)
        sup.wait(timeout=20)
        gone, alive = psutil.wait_procs([worker], timeout=20)
        assert not alive, f"the worker outlived its supervisor ({how!r})"
