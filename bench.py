#!/usr/bin/env python
"""bench.py — headline benchmark of BASELINE.json: adjoint trajectories/sec (+ ns/VJP-step) on the
10^4-trajectory Lorenz-63 ensemble, InterpolatingAdjoint, fixed-step RK4 (BASELINE configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one reverse pass (the hot path) over one rank's ensemble shard with the forward solution already
resident in HBM (the forward solve runs once, untimed, like the reference's forward `solve` precedes its
pullback).

N = 1: the 10^4-trajectory ensemble of BASELINE configs[1] on one GPU.
N > 1: STRONG scaling by default — the same 10^4-trajectory ensemble sharded into contiguous ranges (north_star: ">= 6x at
8 GPUs for a 10^4-trajectory ensemble"; 1250 trajectories per GPU at N = 8), no data-path collective; the only exchange is the
all-reduce of dL/dp (3 doubles) over RCCL, INSIDE the timed step, issued in-stream by the library (hipadj_comm_*;
`--torch-allreduce` uses torch.distributed's asynchronous all-reduce instead).  The line also carries a second figure,
`weak_scaling` (10^4 trajectories PER GPU, measured in the same run); `--weak` makes that the headline instead.

Rank 0 prints ONE JSON line.  The oracle (oracle/) appears only in the cpu_baseline leg and the parity checks.
At N = 1 the line also carries `shard_sizes` (the reverse pass at the 1250 / 2500 / 5000-trajectory shards of the 8 / 4 / 2-GPU
layouts, on this one GPU) and `other_configs` (BASELINE configs[2..4] at their stated sizes, each with its own roofline).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

T_FINAL, DT, SAVE_DT, LOSS_SHIFT, SEED = 10.0, 0.01, 0.1, 2.0, 20240601
HBM_PEAK_GBS, FP64_MFMA_PEAK_TF, FP64_VALU_PEAK_TF = 8000.0, 78.6, 78.6     # /opt/skills/guides/MI355X_MICROARCH.md
LDS_EXCHANGE_US_1024 = 0.38    # one LDS stage exchange of a 1024-thread workgroup (publish, barrier, stencil reads): scripts/r3/lds_exchange_floor.hip, profiles/r3_lds_exchange_floor.log


def _mark(msg):
    """Progress marker on stderr (stdout carries the ONE JSON line): HIPADJ_BENCH_TRACE=1 names the figure under measurement, so that a run that dies says where."""
    if os.environ.get("HIPADJ_BENCH_TRACE"):
        sys.stderr.write(f"[bench] {msg}\n"); sys.stderr.flush()
    maps = os.environ.get("HIPADJ_BENCH_TRACE_MAPS")      # the address space at the last marker: names the owner of a faulting address after ROCr has aborted the process
    if maps:
        try:
            with open("/proc/self/maps") as f, open(maps, "w") as g:
                g.write(f"# at marker: {msg}\n" + f.read())
        except Exception:      # noqa: BLE001
            pass


def _checkpoint(res):
    """The worker's partial result, written after the headline and after every secondary figure (see supervise()): what the supervisor prints if the worker dies later."""
    path = os.environ.get("HIPADJ_BENCH_CHECKPOINT")
    if not path or res is None:
        return
    try:
        with open(path + ".tmp", "w") as f:
            json.dump(res, f)
        os.replace(path + ".tmp", path)
    except Exception:      # noqa: BLE001 — a checkpoint must never cost the run
        pass


EXTRAS_PATH = os.environ.get("HIPADJ_BENCH_EXTRAS", os.path.join(ROOT, "bench_extras.json"))
LINE_LIMIT = 4096     # bytes of the ONE stdout line (round 5's 20 KB line was not parsed by the driver); everything else goes to EXTRAS_PATH


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _r(x, sig=6):
    """Floats of the stdout line at `sig` significant digits (the extras file keeps full precision)."""
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if np.isfinite(x) else None      # strict JSON: no NaN / Infinity literals on the line
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def compact(res):
    """The ONE stdout line: the driver contract's keys + roofline + cpu_baseline + cold_burst + parity + shard_sizes, numbers and short tags only (< LINE_LIMIT bytes,
    checked by tests/test_bench_launch.py).  Prose notes, loss_paths, other_configs and every table live in bench_extras.json (write_extras)."""
    out = _pick(res, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    cfg = res.get("config", {})
    out["config"] = _pick(cfg, ("workload", "ntraj_total", "ntraj_per_gpu", "rk4_steps", "loss_times", "time_segments", "waves_per_workgroup", "parallelism", "dp_allreduce", "rccl_ranks",
                                "native_allreduce_fallback"))
    out.update(_pick(res, ("ns_per_vjp_step", "power_preamble_passes", "forward_solve_ms")))
    if res.get("cold_burst"):
        out["cold_burst"] = _pick(res["cold_burst"], ("ms_per_step", "value", "whole_pass_frac"))
    if res.get("roofline"):
        out["roofline"] = _pick(res["roofline"], ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "algorithmic_bytes_per_launch",
                                                   "kernel_ms", "launches_per_pass", "whole_pass_frac"))
    if res.get("cpu_baseline"):
        cb = res["cpu_baseline"]
        out["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "ns_per_vjp_step", "single_thread_value", "repeats", "spread_rel"))
        out["cpu_baseline"]["sample"] = cb.get("sample_short") or str(cb.get("sample", ""))[:160]
    out.update(_pick(res, ("parity_max_rel_du0_vs_oracle", "parity_max_rel_dp_vs_oracle", "parity_trajectories", "stub_dp")))
    if res.get("shard_sizes"):
        out["shard_sizes"] = [_pick(s, ("ntraj", "gpus_of_layout", "ms_per_step", "kernel_ms", "implied_speedup_if_allreduce_hidden")) for s in res["shard_sizes"]]
    for k in ("weak_scaling", "strong_scaling", "weak_scaling_saturating"):
        if res.get(k):
            out[k] = _pick(res[k], ("value", "ms_per_step", "ntraj_total", "steps"))
    if isinstance(res.get("saturating_ensemble"), dict):
        out["saturating_ensemble"] = _pick(res["saturating_ensemble"], ("ntraj", "ms_per_step", "whole_pass_frac_of_hbm_peak"))
    if isinstance(res.get("single_process_multi_device"), dict):
        out["single_process_multi_device"] = _pick(res["single_process_multi_device"], ("value", "ms_per_step", "devices", "error"))
    if res.get("secondary_figures_incomplete"):
        out["secondary_figures_incomplete"] = _pick(res["secondary_figures_incomplete"], ("worker_exit_code",))
    out["extras"] = res.get("extras_file")
    out = _r(out)
    if "stub_dp" in res:
        out["stub_dp"] = res["stub_dp"]      # tests/test_bench_launch.py compares the stand-in's all-reduced sum at 1e-13
    line = json.dumps(out, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:      # never expected; if a field grows, the optional tables go first, the contract keys never
        for k in ("single_process_multi_device", "saturating_ensemble", "weak_scaling_saturating", "shard_sizes"):
            out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
            if len(line) < LINE_LIMIT:
                break
    return line


def write_extras(res):
    """Everything measured (the full dictionary, prose included) -> bench_extras.json next to bench.py (HIPADJ_BENCH_EXTRAS overrides); returns the path written or None."""
    for path in (EXTRAS_PATH, os.path.join("/tmp", "bench_extras.json")):
        try:
            with open(path + ".tmp", "w") as f:
                json.dump(res, f, indent=1)
            os.replace(path + ".tmp", path)
            return path
        except Exception:      # noqa: BLE001 — a read-only tree must not cost the line
            continue
    return None


def emit(res):
    res["extras_file"] = write_extras(res)
    line = compact(res)
    sys.stdout.write(line + "\n"); sys.stdout.flush()
    sys.stderr.write(f"[bench] the line above is {len(line)} bytes; everything else: {res['extras_file']}\n"); sys.stderr.flush()


def supervise():
    """`python bench.py ...` is a two-process affair per rank: this supervisor (no GPU context, never imports torch) runs the real bench as a child (HIPADJ_BENCH_WORKER=1) and
    relays its ONE JSON line.  The child checkpoints its result after the headline — timed region, roofline, parity: everything the contract asks for — and again after every
    secondary figure; should it die afterwards (round 5: one default run in a dozen ended with a GPU memory fault somewhere in the secondary figures — ROCr aborts the process, no
    `except` sees that), the supervisor prints the last checkpoint with `secondary_figures_incomplete` saying so, instead of nothing.  A death BEFORE the headline is relayed as
    it is (exit code, no line).  HIPADJ_BENCH_SUPERVISE=0 runs the bench in this process."""
    import subprocess
    import tempfile
    fd, path = tempfile.mkstemp(prefix="hipadj_bench_", suffix=".json")
    os.close(fd)
    os.unlink(path)
    env = dict(os.environ, HIPADJ_BENCH_WORKER="1", HIPADJ_BENCH_CHECKPOINT=path)

    def _die_with_parent():      # the worker must not outlive its supervisor (a launcher that kills the rank kills this process; the GPU would stay held by an orphan)
        try:
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, 9)      # PR_SET_PDEATHSIG, SIGKILL
        except Exception:      # noqa: BLE001
            pass
    try:
        import signal
        proc = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE, preexec_fn=_die_with_parent)      # stderr is inherited
        for sig in (signal.SIGTERM, signal.SIGINT):      # a launcher's termination request goes on to the worker
            try:
                signal.signal(sig, lambda s, f: proc.send_signal(s))
            except Exception:      # noqa: BLE001 — not the main thread
                pass
        raw, _ = proc.communicate()
        rc, out = proc.returncode, raw.decode(errors="replace")
        if rc == 0:
            sys.stdout.write(out); sys.stdout.flush()
            return 0
        try:
            with open(path) as f:
                res = json.load(f)
        except Exception:      # noqa: BLE001 — no checkpoint: the worker died before the headline (or is a rank that prints nothing)
            sys.stdout.write(out); sys.stdout.flush()
            return rc
        res["secondary_figures_incomplete"] = {"worker_exit_code": rc,
                                               "note": "the bench process ended abnormally AFTER the headline (timed region, roofline, parity) had been measured and checkpointed; "
                                                       "the figures present are complete, the ones missing were not reached (HIPADJ_BENCH_TRACE=1 names them on stderr)"}
        emit(res)
        return 0
    finally:
        for q in (path, path + ".tmp"):
            try:
                os.unlink(q)
            except OSError:
                pass


def grouped_form(n_traj, segments):
    """Waves per workgroup of the one-launch pass as the planner chose it (csrc/hipadj_plan.hpp plan_group_choice, mirrored here only to NAME the kernel in the line: the
    grouped form k_interp_fused_g composes G consecutive segments of a trajectory block through LDS; 0 = the plain k_interp_fused)."""
    if os.environ.get("HIPADJ_FUSED_GROUP") in ("0", "4", "8"):
        return int(os.environ["HIPADJ_FUSED_GROUP"])
    blocks = (n_traj + 63) // 64
    G, groups = (4, 256 // blocks) if blocks <= 25 else ((8, 256 // blocks) if blocks <= 128 else (4, 512 // blocks))
    S = int(round(T_FINAL / DT))
    while groups > 1 and groups * G > S // (10 if G == 8 else 16):
        groups -= 1
    groups = min(groups, 16)
    return G if groups >= 3 and groups * G == segments else 0


def inputs(n_total):
    rng = np.random.default_rng(SEED)
    u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((n_total, 3))
    p = np.array([10.0, 28.0, 8.0 / 3.0])
    return u0, p


def save_times():
    return np.linspace(0.0, T_FINAL, int(round(T_FINAL / SAVE_DT)) + 1)


def oracle_problem(alg="INTERPOLATING", **kw):
    import oracle as O
    return O.Problem("LORENZ", alg=alg, stepper="RK4", t0=0.0, t1=T_FINAL, dt=DT, save_times=save_times(), loss="LSQ_SHIFT", loss_shift=LOSS_SHIFT, **kw)


def host_cpu_budget():
    """CPUs this process can really use: the smallest of the hardware threads, the scheduler affinity mask and the cgroup CPU quota
    (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1).  os.cpu_count() alone reports the whole host inside a container."""
    n, src = os.cpu_count() or 1, "os.cpu_count"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, src = a, "sched_getaffinity"
    except Exception:
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            q, per = parse(open(path).read())
            if q not in ("max", "-1") and float(per) > 0:
                c = max(1, int(float(q) / float(per) + 0.5))
                if c < n:
                    n, src = c, path
        except Exception:
            pass
    return n, src


def cpu_baseline(u0, p, ts, budget_s=20.0):
    """The oracle (CPU restatement of the reference algorithm — NOT Julia) timed on the workload's own trajectories (reverse passes
    only, like `value`), OpenMP over trajectories.  Every figure is the median of >= 5 repeats of the SAME call (all trajectories,
    rounded down to a multiple of the thread count) and carries its spread; the thread count is the one at which that median is
    highest, and the probe that chose it is the same measurement, so probe and reported value agree by construction."""
    cores = os.cpu_count() or 1
    budget, budget_src = host_cpu_budget()
    pr = oracle_problem()

    def sample(nt, min_reps, max_s):
        n = max(nt, (len(u0) // nt) * nt)
        pr.adjoint_ensemble(u0[:n], p, nthreads=nt, want_out=False)                    # warm-up: per-thread solution pools, page faults
        rates, rev, t_start = [], 0.0, time.perf_counter()
        while len(rates) < min_reps or (rev * nt < max_s and len(rates) < 200 and time.perf_counter() - t_start < 60.0):
            _, _, _, tm = pr.adjoint_ensemble(u0[:n], p, nthreads=nt, want_out=False)
            rates.append(n / tm["reverse_s"])          # reverse_s = max over threads of the time spent in reverse passes
            rev += tm["reverse_s"]
        return n, np.array(rates), rev

    probe = {}
    # thread counts around what this process may really use (the box reports 256 hardware threads; the container's CPU quota was 16 of them in
    # round 3's scaling study — beyond the quota the threads are throttled and the rate collapses: profiles/r3_cpu_scaling_study.log)
    ladder = {budget, max(1, budget // 2)}
    c = cores
    while c >= 4:                       # the quota is not always visible from inside (no cpu.max): the ladder finds the knee anyway
        ladder.add(c); c //= 2
    for nt in sorted(ladder, reverse=True):
        _, rates, _ = sample(nt, 5, 0.0)
        probe[nt] = float(np.median(rates))
    cores_used = max(probe, key=probe.get)
    n, rates, rev = sample(cores_used, 5, budget_s)
    med = float(np.median(rates))
    # the same path on ONE host thread (SURVEY.md §8d asks for both): a smaller sample, same inputs, median of 5
    n1 = min(len(u0), 1024)
    pr.adjoint_ensemble(u0[:n1], p, nthreads=1, want_out=False)
    r1 = np.array([n1 / pr.adjoint_ensemble(u0[:n1], p, nthreads=1, want_out=False)[3]["reverse_s"] for _ in range(5)])
    return dict(value=med, unit="trajectories/s", cores=cores_used, host_threads=cores,
                cores_note=f"{cores_used} of {cores} host threads (the thread count at which the oracle's median rate is highest); usable CPUs of this process: {budget} ({budget_src})",
                host_cpu_budget=budget,
                repeats=int(len(rates)), spread_min_max=[float(rates.min()), float(rates.max())],
                spread_rel=float((rates.max() - rates.min()) / med),
                thread_probe_traj_per_s=" ".join(f"{k}:{v:.3g}" for k, v in sorted(probe.items(), reverse=True)),
                probe_vs_value_rel_diff=abs(probe[cores_used] - med) / med, kind="port",
                single_thread_value=float(np.median(r1)), single_thread_spread_min_max=[float(r1.min()), float(r1.max())],
                single_thread_ns_per_vjp_step=1e9 / (float(np.median(r1)) * 1000 * 4),
                parallel_efficiency=med / (cores_used * float(np.median(r1))),
                sample_short=f"{n} trajectories x {len(rates)} repeats, reverse passes only, median; {rev * cores_used:.0f} core-s; C oracle, OpenMP",
                sample=f"{n} of the workload's trajectories x {len(rates)} repeats, reverse passes only, median ({rev:.2f} s on {cores_used} threads = "
                       f"{rev * cores_used:.0f} core-seconds), C oracle, OpenMP over trajectories (static schedule, unbound threads), gcc -O2 -ffp-contract=off",
                ns_per_vjp_step=1e9 / (med * 1000 * 4))


PMC_KERNEL = "k_interp"                     # the dominant kernel's name prefix in the counter file


def pmc_child(sa, n_total):
    """`bench.py --pmc-child`: the same reverse pass (forward solve once, then a few passes), nothing else; run by live_traffic() under
    `rocprofv3 --pmc <one counter>`.  No timing, no JSON line."""
    u0, p = inputs(n_total)
    eng = sa.Engine("lorenz", "interpolating", n_total, 0.0, T_FINAL, DT, save_times=save_times(), loss_kind=1, loss_shift=LOSS_SHIFT, p_shared=True)
    eng.set_timing(0)
    eng.forward(u0, p, want_out=False)
    for _ in range(6):
        eng.adjoint(None)
    eng.close()


def live_traffic(n_total, timeout_s=240.0):
    """HBM traffic of the dominant kernel per launch, measured NOW: two separate `rocprofv3 --pmc` passes (FETCH_SIZE and WRITE_SIZE do not fit the TCC
    slots together) over `bench.py --pmc-child`, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes: both counters are in KB,
    FETCH_SIZE of a 16-byte-per-lane streaming read reports half the bytes on gfx950 (x2), WRITE_SIZE is taken as reported (uncalibrated).
    Returns (bytes per launch or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "not collected: this process already runs under rocprofv3"
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, "not collected: rocprofv3 not found"
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="hipadj_pmc_", dir="/tmp")
        try:
            env = dict(os.environ, TMPDIR="/tmp", PYTHONWARNINGS="ignore")
            cmd = [rp, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--ntraj", str(n_total)]
            rc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if rc.returncode != 0 or not files:
                return None, f"not collected: rocprofv3 --pmc {counter} exit {rc.returncode}: {rc.stderr.decode(errors='replace')[-200:]}"
            vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
                    if r.get("Counter_Name") == counter and PMC_KERNEL in r.get("Kernel_Name", "")]
            if not vals:
                return None, f"not collected: no {PMC_KERNEL}* rows in the {counter} pass"
            got[counter] = (sum(vals) / len(vals), len(vals))
        except Exception as e:      # noqa: BLE001 — the headline must not die on the counter pass
            return None, f"not collected: {e!r}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch = got["FETCH_SIZE"][0] * 1024.0 * 2.0
    write = got["WRITE_SIZE"][0] * 1024.0
    return fetch + write, (f"LIVE: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of `bench.py --pmc-child`, same workload, this box, this run): "
                           f"FETCH_SIZE {got['FETCH_SIZE'][0]:.0f} KB x 2 (gfx950 16 B/lane calibration) + WRITE_SIZE {got['WRITE_SIZE'][0]:.0f} KB (uncalibrated), "
                           f"mean of {got['FETCH_SIZE'][1]} / {got['WRITE_SIZE'][1]} launches")


STUB = os.environ.get("HIPADJ_BENCH_STUB") == "1"     # launcher / carrier self-test on CPU (tests/bench_stub.py, gloo): no kernel runs, the line says so
_ABANDONED = []                                        # engines whose communicator bootstrap hung: never destroyed (hipadj_destroy would wait for the hung stream)


def _call_with_timeout(fn, seconds):
    """Runs fn() on a daemon thread.  Returns (finished, exception or None): a collective that never returns (one rank missing from the
    RCCL bootstrap) costs `seconds`, not the run."""
    import threading
    box = {}

    def run():
        try:
            fn()
        except BaseException as e:      # noqa: BLE001 — reported to the caller
            box["e"] = e
        box["done"] = True
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(seconds)
    return bool(box.get("done")), box.get("e")


class Runner:
    """One engine on this rank's shard + the timed loop of the driver contract."""

    def __init__(self, sa, torch, dist, args, n_local, u0_np, p_np, local_rank, world, native):
        self.torch, self.dist, self.world, self.native = torch, dist, world, native
        dev = self.dev = torch.device("cpu") if STUB else torch.device("cuda", local_rank)
        self.sync = (lambda: None) if STUB else torch.cuda.synchronize

        def make_engine():
            return sa.Engine("lorenz", "interpolating", n_local, 0.0, T_FINAL, DT, save_times=save_times(), loss_kind=1, loss_shift=LOSS_SHIFT,
                             p_shared=True, device=local_rank, time_segments=args.segments)
        self.eng = make_engine()
        self.native_note = None
        if native:
            # on the handle's OWN stream, before it is moved to torch's: a collective that hangs then blocks a stream nobody else uses
            native = self.native = self._try_native(sa, torch, dist, dev, world, make_engine)
        # a real torch stream for the handle (torch's default "current stream" is the null stream, for which hipadj_set_stream keeps the handle's own
        # stream): the region events, the caller's tensors and the torch.distributed all-reduce are all ordered on it
        self.stream = None if STUB else torch.cuda.Stream(device=dev)
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                self.eng.use_torch_stream()
        else:
            self.eng.use_torch_stream()
        self.u0 = torch.tensor(u0_np, device=dev, dtype=torch.float64)
        self.p = torch.tensor(p_np, device=dev, dtype=torch.float64)
        self.du0 = torch.empty((n_local, 3), device=dev, dtype=torch.float64)
        self.dps = [torch.empty(3, device=dev, dtype=torch.float64) for _ in range(2)]
        self.sync()                                          # the tensors above were filled on torch's default stream
        self.eng.forward_dev(self.u0, self.p, None)          # forward solve: interpolant tiles now resident in HBM
        self.sync()
        self.eng.forward_dev(self.u0, self.p, None)          # once more: forward_solve_ms is the steady-state call, not the first launch (code load)
        self.sync()
        self.it, self.pending = 0, None
        self.preamble_passes = 0

    def power_preamble(self):
        """Untimed reverse passes that bring the board to its SUSTAINED power / clock state before the W warm-up steps (setup, like the forward solve).
        Why: from idle the chip runs ~16 passes at its boost clock (0.109 ms each), then the power limiter clamps hard for ~5 ms (0.14-0.16 ms per pass at a
        1.4 kW cap) and relaxes to the sustained rate (~0.115 ms) only after ~40 ms (profiles/r4_ramp_probe.json, scripts/r4/ramp_probe.py).  A 5 + 20-step
        burst from idle measures that transient, not the rate of a loop that keeps asking for gradients.  The line reports BOTH: `value` after this preamble,
        `cold_burst` = the same W + K steps after 0.3 s of idle.  The count depends on the shard size only, so all ranks run the same number of steps."""
        self.preamble_passes = 0 if STUB else min(2000, int(4e6 / max(self.eng.N, 1)))
        self.eng.set_timing(0)
        for _ in range(self.preamble_passes):
            self.step()
        self.drain()
        self.sync()

    def _agree(self, ok):
        """All ranks learn whether EVERY rank succeeded (torch.distributed carries the flag)."""
        flag = self.torch.tensor([1 if ok else 0], device=self.dev, dtype=self.torch.int32)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def _try_native(self, sa, torch, dist, dev, world, make_engine):
        """The library's own RCCL communicator (torch.distributed only ships the 128-byte id).  Every failure — the id cannot be drawn,
        ncclCommInitRank returns an error or does not return within HIPADJ_COMM_TIMEOUT seconds on some rank, the probe all-reduce of
        hipadj_comm_selfcheck comes back wrong or hangs — makes ALL ranks agree on the torch.distributed carrier instead of losing the run."""
        rank = dist.get_rank()
        tmo = float(os.environ.get("HIPADJ_COMM_TIMEOUT", "90"))
        try:
            box = [sa.comm_unique_id() if rank == 0 else None]
        except Exception as e:
            box = [None]
            sys.stderr.write(f"bench: native RCCL id failed on rank 0 ({e!r}); falling back to torch.distributed\n")
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            self.native_note = "ncclGetUniqueId failed"
            return False
        hung = False
        for what, fn in (("hipadj_comm_init_rank", lambda: self.eng.comm_init_rank(box[0], world, rank)),
                         ("hipadj_comm_selfcheck", lambda: self.eng.comm_selfcheck())):
            done, err = _call_with_timeout(fn, tmo)
            ok = done and err is None
            if not ok:
                hung = hung or not done
                sys.stderr.write(f"bench: {what} {'did not return within %g s' % tmo if not done else 'failed'} on rank {rank} ({err!r}); "
                                 f"falling back to torch.distributed\n")
            if not self._agree(ok):
                self.native_note = f"{what} failed or timed out on some rank"
                if hung:                       # the handle may still be inside the collective: leave it alone for good, start over on a fresh one
                    _ABANDONED.append(self.eng)
                    self.eng = make_engine()
                else:
                    try:
                        self.eng.comm_destroy()
                    except Exception:
                        pass
                return False
        # the all-reduce off the next pass's critical path (hipadj_comm_overlap, ABI 107): this loop already alternates two dp buffers
        self.native_overlap = False
        if hasattr(self.eng, "comm_overlap") and os.environ.get("HIPADJ_COMM_OVERLAP", "1") != "0":
            try:
                self.eng.comm_overlap(True)
                self.native_overlap = True
            except Exception as e:      # noqa: BLE001 — an older library: the in-stream collective stands
                sys.stderr.write(f"bench: hipadj_comm_overlap unavailable ({e!r}); the all-reduce stays in-stream\n")
        return True

    def step(self):
        if self.stream is None:
            return self._step()
        with self.torch.cuda.stream(self.stream):
            return self._step()

    def _step(self):
        # reverse pass of this step.  torch carrier: the all-reduce of dL/dp (RCCL, its own stream) overlaps the NEXT step's kernels —
        # dp is double-buffered and the previous step's reduction is only waited for here.  Native carrier: in-stream inside the call.
        dp = self.dps[self.it & 1]
        self.eng.adjoint_dev(None, self.du0, dp)
        if self.world > 1 and not self.native:
            if self.pending is not None:
                self.pending.wait()
            self.pending = self.dist.all_reduce(dp, op=self.dist.ReduceOp.SUM, async_op=True)
        self.it += 1

    def drain(self):
        if self.pending is not None:
            self.pending.wait()
            self.pending = None
        if self.native and getattr(self, "native_overlap", False):
            self.eng.synchronize()            # waits for the handle's collective stream as well

    def timed(self, steps, warmup):
        """EXACTLY `steps` steps between barrier + synchronize on both sides (the driver's contract).  The library records NO per-kernel events
        in here (a dispatch-packet event pair costs 5-6 us per launch on the one-launch pass: profiles/r3_visit2_fused_16B_timing_ab.log); one HIP
        event pair on the launch stream brackets the whole region instead -> self.region_ms."""
        torch, dist = self.torch, self.dist
        self.eng.set_timing(0)
        for _ in range(warmup):
            self.step()
        self.drain()
        self.sync()
        self.eng.synchronize()
        ev = None if STUB else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        if self.world > 1:
            dist.barrier()
        self.sync()
        t0 = time.perf_counter()
        if ev:
            ev[0].record(self.stream)
        for _ in range(steps):
            self.step()
        if ev:
            ev[1].record(self.stream)
        self.drain()
        self.sync()
        if self.world > 1:
            dist.barrier()
        self.sync()
        elapsed = time.perf_counter() - t0
        self.eng.synchronize()
        self.region_ms = ev[0].elapsed_time(ev[1]) if ev else elapsed * 1e3
        if self.world > 1:
            tt = torch.tensor([elapsed], device=self.u0.device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed

    def profiled(self, steps):
        """The same step with the library's event pair on the dominant kernel's dispatch packet (what rocprofv3 reports as the kernel's
        duration), back to back like the timed region: (average kernel ms, stats)."""
        self.eng.set_timing(1)
        for _ in range(3):
            self.step()
        self.drain(); self.sync(); self.eng.synchronize()
        st0 = self.eng.stats()
        for _ in range(steps):
            self.step()
        self.drain(); self.sync(); self.eng.synchronize()
        st1 = self.eng.stats()
        self.eng.set_timing(0)
        calls = max(st1["adjoint_calls"] - st0["adjoint_calls"], 1)
        return (st1["adjoint_main_kernel_ms_total"] - st0["adjoint_main_kernel_ms_total"]) / calls, st1

    def last_dp(self):
        return self.dps[(self.it - 1) & 1]

    def close(self):
        self.eng.close()


def other_configs(sa, torch):
    """BASELINE configs[2..4] at their stated sizes on this GPU, each with the roofline that bounds it (reverse pass only, forward
    solution resident; library events on the launch stream).  Short: a few repeats each."""
    from test_gpu_parity import mlp_params, bruss_u0
    out = []
    rng = np.random.default_rng(0)

    def run(eng, u0, p, delta, reps):
        eng.forward(u0, p, want_out=False)
        eng.adjoint(delta)
        s0 = eng.stats()
        for _ in range(reps):
            eng.adjoint(delta)
        s1 = eng.stats()
        return ((s1["adjoint_ms_total"] - s0["adjoint_ms_total"]) / reps, (s1["adjoint_main_kernel_ms_total"] - s0["adjoint_main_kernel_ms_total"]) / reps, s1)

    # configs[2]: the C2 ensemble, BacksolveAdjoint(checkpointing=true), a checkpoint every 10 steps — compute-bound (reads only the checkpoints)
    u0, p = inputs(10000)
    eng = sa.Engine("lorenz", "backsolve", 10000, 0.0, T_FINAL, DT, save_times=save_times(), loss_kind=1, loss_shift=LOSS_SHIFT, checkpointing=True)
    ms, kms, st = run(eng, u0, p, None, 10)
    # FP64 work of the sequential formulation: per trajectory and step 4 f + 4 vjp_u + 4 vjp_p + stage algebra ~ 226 flop
    _mark("configs[2]: Lorenz 10^4 x 1000 steps, BacksolveAdjoint(chec")
    out.append(dict(config="configs[2]: Lorenz 10^4 x 1000 steps, BacksolveAdjoint(checkpointing=true), checkpoints every 10 steps (1 GPU)",
                    reverse_ms=ms, main_kernel_ms=kms, trajectories_per_s=10000 / (ms * 1e-3), time_segments=st["time_segments"],
                    roofline=dict(bound="fp64_valu", achieved=226.0 * 1e4 * 1000 / (kms * 1e-3) / 1e12, peak=FP64_VALU_PEAK_TF, unit="TFLOP/s",
                                  frac=226.0 * 1e4 * 1000 / (kms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
                                  note="flops of the SEQUENTIAL formulation (226 per trajectory-step); the time-segmented kernel executes ~3.3x that",
                                  hbm_bytes=st["adjoint_algorithmic_bytes"])))
    eng.close()
    # configs[3]: MLP 2 -> 128 -> 128 -> 2, 4096 columns, 150 RK4 steps, 30 loss times, GaussAdjoint (FP64 MFMA)
    d, H, B, S = 2, 128, 4096, 150
    ts = 0.01 * np.arange(5, S + 1, 5)
    eng = sa.Engine("mlp", "gauss", 1, 0.0, S * 0.01, 0.01, save_times=ts, dims=(d, H, B, 0))
    ms, kms, st = run(eng, rng.standard_normal((1, d * B)), mlp_params(d, H), rng.standard_normal((1, len(ts), d * B)), 3)
    # Work of the reverse pass in H x H x B contractions (2 H^2 B flop each): the stages and Gauss nodes of the reference's algorithm need
    # 3 forward + 4 backward + 1 (fsallast) + 2 x (forward + backward) = 12 per step, the weight gradient 2 outer products per step
    # (dW2; the d-sized pieces are ~1/8 of one).  The kernel skips what first-same-as-last makes redundant (x_hi activations, V1 without a
    # loss jump: ~1.8 per step), so the EXECUTED count is ~12.2 per step and `achieved` (nominal work / time) is an upper bound on the
    # matrix-pipe utilisation; both counts are reported.
    nominal = (12 + 2) * S * 2.0 * H * H * B + 2 * S * 2.0 * B * (H * 16 + 16 * (H + 16))
    jumps = len(ts)
    executed = ((4 + 6 + 2) * S + jumps) * 2.0 * H * H * B
    _mark("configs[3]: MLP 2-128-128-2, batch 4096, 150 RK4 steps, Gau")
    out.append(dict(config="configs[3]: MLP 2-128-128-2, batch 4096, 150 RK4 steps, GaussAdjoint (1 GPU)", reverse_ms=ms, sweep_kernel_ms=kms,
                    gradient_reduction_ms=ms - kms, workspace_GB=st["workspace_bytes"] / 1e9,
                    roofline=dict(bound="mfma", achieved=executed / (kms * 1e-3) / 1e12, peak=FP64_MFMA_PEAK_TF, unit="TFLOP/s",
                                  frac=executed / (kms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF, kernel="k_mlp_adjoint_grad",
                                  nominal_TFLOPs=nominal / (ms * 1e-3) / 1e12, nominal_frac=nominal / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF,
                                  note="achieved / frac count the contractions the kernel EXECUTES (after first-same-as-last reuse) over the sweep kernel's time; nominal = the reference "
                                       "algorithm's 12 contractions + 2 weight-gradient outer products per step over the whole reverse pass, i.e. work the kernel skips included")))
    eng.close()
    # configs[4]: Brusselator 32 x 32 (n = 2048), QuadratureAdjoint, 400 explicit RK4 steps; N = 1 (the config) and N = 256 (fills the chip)
    G, dtb, Sb = 32, 2.5e-5, 400
    tsb = dtb * np.arange(0, Sb + 1, 100)
    for N in (1, 256):
        eng = sa.Engine("bruss", "quadrature", N, 0.0, Sb * dtb, dtb, save_times=tsb, dims=(G, 0, 0, 0))
        ms, kms, st = run(eng, bruss_u0(G, N), np.array([3.4, 1.0, 10.0]), rng.standard_normal((N, len(tsb), 2 * G * G)), 3)
        n = 2 * G * G
        by = N * (Sb + 1) * 16.0 * n + N * Sb * 32.0 * n      # knots read + dense-lambda record written
        if N == 1:
            # ONE workgroup on ONE CU: nothing streams, the step is a chain of LDS exchanges.  Floor model from scripts/r3/lds_exchange_floor.hip on this chip
            # (profiles/r3_lds_exchange_floor.log): publishing a stage vector, the barrier and the five-point stencil reads of one stage cost 0.38 us for 1024
            # threads (a bare barrier: 0.039 us); the lambda pass has 4 such exchanges per step (5 where a loss jump follows)
            floor_us = 4 * LDS_EXCHANGE_US_1024
            roof = dict(bound="latency", kernel="k_bruss_quad_adj", achieved_us_per_step=kms * 1e3 / Sb, floor_us_per_step=floor_us, frac=floor_us / (kms * 1e3 / Sb),
                        note="frac = floor / achieved; floor = 4 LDS stage exchanges x 0.38 us (measured, 1024 threads: profiles/r3_lds_exchange_floor.log); HBM is idle at N = 1 "
                             f"({by / (kms * 1e-3) / 1e9:.0f} GB/s)")
        else:
            roof = dict(bound="hbm", achieved=by / (kms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=by / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        kernel="k_bruss_quad_adj", algorithmic_bytes_per_launch=by)
        _mark(f"configs[4]: Brusselator 32x32, QuadratureAdjoint, 400 RK4 ")
        out.append(dict(config=f"configs[4]: Brusselator 32x32, QuadratureAdjoint, 400 RK4 steps, N = {N} (1 GPU)", reverse_ms=ms, lambda_pass_ms=kms,
                        us_per_step=kms * 1e3 / Sb, roofline=roof))
        eng.close()
    # ---- wide runtime models (csrc/hipadj_wide.hpp): the two problems the reference itself holds beyond 8 states / 32 parameters
    try:
        out += wide_rows(sa, run)
    except Exception as e:
        _mark("wide runtime models")
        out.append(dict(config="wide runtime models", error=repr(e)))
    # configs[4] over the horizon the reference documents, tspan = (0, 11.5), loss times 0:0.5:11.5: 460 000 explicit RK4 steps at the diffusion
    # stability limit (the docs use the implicit FBDF); 15 GB of knots + 30 GB of dense lambda record in HBM
    try:
        Sh = 460000
        tsh = 0.5 * np.arange(0, 24)
        eng = sa.Engine("bruss", "quadrature", 1, 0.0, Sh * dtb, dtb, save_times=tsh, dims=(G, 0, 0, 0))
        ms, kms, st = run(eng, bruss_u0(G, 1), np.array([3.4, 1.0, 10.0]), rng.standard_normal((1, len(tsh), 2 * G * G)), 1)
        _mark("configs[4] at the documented horizon: Brusselator 32x32, ts")
        out.append(dict(config="configs[4] at the documented horizon: Brusselator 32x32, tspan (0, 11.5), QuadratureAdjoint, 460 000 RK4 steps of dt = 2.5e-5, N = 1 (1 GPU)",
                        forward_ms=st["forward_ms_last"], reverse_ms=ms, lambda_pass_ms=kms, us_per_step=kms * 1e3 / Sh, workspace_GB=st["workspace_bytes"] / 1e9))
        eng.close()
    except Exception as e:
        _mark("configs[4] at the documented horizon")
        out.append(dict(config="configs[4] at the documented horizon", error=repr(e)))
    # the same horizon with the stiff stepper of the family (HIPADJ_STEPPER_ETDRK4_FIXED, csrc/hipadj_field_etd.hpp): the diffusion term exact in the DFT basis, 7360 steps
    # of dt = 1/640 (the step at which the gradient meets scipy's Radau to 1e-5..1e-6, tests/test_etd_stepper.py) instead of 460 000
    try:
        dte, Se = 0.0015625, 7360
        tsh = 0.5 * np.arange(0, 24)
        for alg, N in (("quadrature", 1), ("interpolating", 1), ("interpolating", 64)):
            eng = sa.Engine("bruss", alg, N, 0.0, Se * dte, dte, save_times=tsh, dims=(G, 0, 0, 0), stepper=2)
            ms, kms, st = run(eng, bruss_u0(G, N), np.array([3.4, 1.0, 10.0]), rng.standard_normal((N, len(tsh), 2 * G * G)), 1)
            _mark(f"configs[4] at the documented horizon with the exponential ")
            out.append(dict(config=f"configs[4] at the documented horizon with the exponential stepper: Brusselator 32x32, tspan (0, 11.5), {alg}, 7360 ETDRK4 steps of dt = 1/640, N = {N} (1 GPU)",
                            forward_ms=st["forward_ms_last"], reverse_ms=ms, sweep_kernel_ms=kms, us_per_step=kms * 1e3 / Se, workspace_GB=st["workspace_bytes"] / 1e9,
                            roofline=dict(bound="latency", kernel="k_bruss_adjoint_etd", note="one workgroup per trajectory: nine 32 x 32 complex FFTs per reverse step (ten lane-exchange "
                                          "levels and one LDS transposition each) are a dependent chain; N = 64 runs 64 of them side by side")))
            eng.close()
    except Exception as e:
        _mark("configs[4] at the documented horizon with the exponential s")
        out.append(dict(config="configs[4] at the documented horizon with the exponential stepper", error=repr(e)))
    # the stiff stepper of the lane family (round 6): Robertson kinetics at the classic stiff rates (0.04, 3e7, 1e4) +- 10 %, tspan (0, 100), G = y3(50) + y3(100), as a RUNTIME model
    # (hiprtc), Rosenbrock23 at abstol 1e-8 / reltol 1e-6 — a problem adaptive Tsit5 needs ~1e6 steps per trajectory for (scripts/r6/bench_rosenbrock23.py has the full table)
    try:
        rob = sa.DeviceFunction("rober_bench_line", 3, 3,
                                "du[0] = -p[0]*u[0] + p[2]*u[1]*u[2]; du[1] = p[0]*u[0] - p[1]*u[1]*u[1] - p[2]*u[1]*u[2]; du[2] = p[1]*u[1]*u[1];",
                                "out[0] = -p[0]*lam[0] + p[0]*lam[1]; out[1] = p[2]*u[2]*lam[0] + (-2.0*p[1]*u[1] - p[2]*u[2])*lam[1] + 2.0*p[1]*u[1]*lam[2]; out[2] = p[2]*u[1]*lam[0] - p[2]*u[1]*lam[1];",
                                "out[0] = -u[0]*lam[0] + u[0]*lam[1]; out[1] = -u[1]*u[1]*lam[1] + u[1]*u[1]*lam[2]; out[2] = u[1]*u[2]*lam[0] - u[1]*u[2]*lam[1];")
        Nr = 8192
        pr = np.array([0.04, 3.0e7, 1.0e4]) * (1 + 0.1 * rng.uniform(-1, 1, (Nr, 3)))
        ur = np.tile([1.0, 0.0, 0.0], (Nr, 1)); tsr = np.array([50.0, 100.0]); dr = np.zeros((Nr, 2, 3)); dr[:, :, 2] = 1.0
        for alg in ("interpolating", "gauss"):
            eng = sa.Engine(rob.name, alg, Nr, 0.0, 100.0, 0.0, save_times=tsr, loss_kind=0, p_shared=False, stepper=3, abstol=1e-8, reltol=1e-6)
            ms, kms, st = run(eng, ur, pr, dr, 3)
            _mark("Rosenbrock23: Robertson at the stiff rates")
            out.append(dict(config=f"Rosenbrock23 (stiff stepper, round 6): Robertson kinetics at rates (0.04, 3e7, 1e4) +- 10 %, tspan (0, 100), {Nr} trajectories, runtime model, {alg}, abstol 1e-8 / reltol 1e-6",
                            forward_ms=st["forward_ms_last"], reverse_ms=ms, main_kernel_ms=kms, gradients_per_s=Nr / ((st["forward_ms_last"] + ms) * 1e-3),
                            roofline=dict(bound="one wave's instruction stream (per-lane LU + three solves per step)", note="adaptive per-lane stepping: no fixed byte or flop count per launch")))
            eng.close()
    except Exception as e:
        _mark("Rosenbrock23: Robertson at the stiff rates")
        out.append(dict(config="Rosenbrock23: Robertson at the stiff rates", error=repr(e)))
    # ContinuousCallback (round 6): the reference's bouncing ball (test/Callbacks2/continuous_callbacks.jl:10-14, 212-217) as an ensemble in which every trajectory has its own bounces;
    # f, condition and affect as text (hiprtc), adaptive Tsit5 at 1e-6 (scripts/r6/bench_continuous_callback.py has the full table, also without the callback)
    try:
        ball = sa.DeviceFunction("ball_bench_line", 2, 2, "du[0] = u[1]; du[1] = -p[0];", "out[0] = 0.0; out[1] = lam[0];", "out[0] = -lam[1]; out[1] = 0.0;")
        ball.set_continuous_callback("c = u[0];", "un[1] = -p[1] * u[1];")
        Nb = 65536
        ub = np.stack([rng.uniform(2.0, 9.0, Nb), rng.uniform(-1.0, 1.0, Nb)], axis=1)
        pb = np.stack([9.8 * (1 + 0.1 * rng.uniform(-1, 1, Nb)), rng.uniform(0.8, 0.9, Nb)], axis=1)
        tsb = np.array([0.3, 1.0, 1.7, 2.2, 3.1, 4.0]); db = rng.standard_normal((Nb, len(tsb), 2))
        for alg in ("interpolating", "gauss"):
            eng = sa.Engine(ball.name, alg, Nb, 0.0, 4.0, 0.0, save_times=tsb, loss_kind=0, p_shared=False, stepper=1, abstol=1e-6, reltol=1e-6)
            ms, kms, st = run(eng, ub, pb, db, 3)
            ne = eng.event_counts()
            _mark("ContinuousCallback: bouncing balls")
            out.append(dict(config=f"ContinuousCallback (round 6): {Nb} bouncing balls dropped from 2 .. 9 with restitution 0.8 .. 0.9 over (0, 4), runtime model with condition and affect as text, adaptive Tsit5 1e-6, {alg}",
                            forward_ms=st["forward_ms_last"], reverse_ms=ms, main_kernel_ms=kms, gradients_per_s=Nb / ((st["forward_ms_last"] + ms) * 1e-3),
                            events_per_trajectory=dict(min=int(ne.min()), mean=float(ne.mean()), max=int(ne.max())),
                            roofline=dict(bound="one wave's instruction stream (per-lane step control and event search)", note="adaptive per-lane stepping: no fixed byte or flop count per launch")))
            eng.close()
    except Exception as e:
        _mark("ContinuousCallback: bouncing balls")
        out.append(dict(config="ContinuousCallback: bouncing balls", error=repr(e)))
    return out


def loss_paths(sa, torch, args, u0_np, p_np, local_rank, headline_ms):
    """The SAME 10^4-trajectory reverse pass through every route a caller's loss can take, each as a sustained loop like the headline:
      lsq_data_device   sum(abs2, sol .- data) with the data block resident in the handle (HIPADJ_LOSS_LSQ_DATA): no cotangents, one launch
      cotangent_soa     Delta on the device already in the streaming layout (hipadj_adjoint_dev_soa): one launch
      cotangent_path    Delta on the device as [N][M][n] (the AD pullback's shape): transposed inside the sweep, one launch
      host_api          hipadj_forward / hipadj_adjoint with HOST pointers — what the Julia binding calls (julia/HIPAdj/src/HIPAdj.jl): Delta upload, du0 / dp download included,
                        with the PCIe bound of the bytes that cross the link."""
    N = len(u0_np); ts = save_times(); M = len(ts); n = 3
    dev = torch.device("cuda", local_rank)
    rng = np.random.default_rng(7)
    out = {}
    stream = torch.cuda.Stream(device=dev)

    def loop(fn, eng, steps, warmup, preamble):
        eng.set_timing(0)
        with torch.cuda.stream(stream):
            for _ in range(preamble + warmup):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(steps):
                fn()
            e1.record(stream)
        torch.cuda.synchronize()
        eng.synchronize()
        return e0.elapsed_time(e1) / steps

    pre = 0 if args.no_preamble else 400
    u0 = torch.tensor(u0_np, device=dev); p = torch.tensor(p_np, device=dev)
    du0 = torch.empty((N, n), device=dev, dtype=torch.float64); dp = torch.empty(3, device=dev, dtype=torch.float64)
    # --- the device-resident data loss
    data = torch.tensor(2.0 + 0.5 * rng.standard_normal((N, M, n)), device=dev)
    eng = sa.Engine("lorenz", "interpolating", N, 0.0, T_FINAL, DT, save_times=ts, loss_kind=2, loss_scale=2.0, p_shared=True, device=local_rank, time_segments=args.segments)
    with torch.cuda.stream(stream):
        eng.use_torch_stream()
        eng.set_loss_data_dev(data)
        eng.forward_dev(u0, p, None)
    torch.cuda.synchronize()
    ms = loop(lambda: eng.adjoint_dev(None, du0, dp), eng, args.steps, args.warmup, pre)
    by = eng.stats()["adjoint_algorithmic_bytes"]
    out["lsq_data_device"] = dict(ms_per_step=ms, over_headline=ms / headline_ms, algorithmic_bytes=by, frac_of_hbm_peak=by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, launches_per_pass=eng.stats()["launches_per_pass"],
                                  note="loss = sum(abs2, sol .- data), data [N][M][n] handed over once (hipadj_set_loss_data_dev); the sweep streams it next to the knots")
    # parity of this route against the oracle on a sample
    try:
        import oracle as O
        k = 256
        pr = O.Problem("LORENZ", alg="INTERPOLATING", stepper="RK4", t0=0.0, t1=T_FINAL, dt=DT, save_times=ts, loss="LSQ_DATA", loss_scale=2.0)
        rdu0, _, _, _ = pr.adjoint_ensemble(u0_np[:k], p_np, data[:k].cpu().numpy(), want_out=False)
        out["lsq_data_device"]["parity_max_rel_du0_vs_oracle_first_256"] = float(np.max(np.abs(du0[:k].cpu().numpy() - rdu0)) / np.max(np.abs(rdu0)))
    except Exception as e:      # noqa: BLE001
        out["lsq_data_device"]["parity_error"] = repr(e)
    eng.close()
    # --- cotangents on the device: AD layout (+ transposition) and streaming layout
    eng = sa.Engine("lorenz", "interpolating", N, 0.0, T_FINAL, DT, save_times=ts, loss_kind=0, p_shared=True, device=local_rank, time_segments=args.segments)
    delta = torch.tensor(rng.standard_normal((N, M, n)), device=dev)
    with torch.cuda.stream(stream):
        eng.use_torch_stream()
        eng.forward_dev(u0, p, None)
    torch.cuda.synchronize()
    ms_aos = loop(lambda: eng.adjoint_dev(delta, du0, dp), eng, args.steps, args.warmup, pre)
    ld = eng.soa_stride()
    soa = torch.zeros((M, n, ld), device=dev, dtype=torch.float64); soa[:, :, :N] = delta.permute(1, 2, 0)
    ref_du0 = du0.clone()
    ms_soa = loop(lambda: eng.adjoint_dev_soa(soa, du0, dp), eng, args.steps, args.warmup, pre)
    by = eng.stats()["adjoint_algorithmic_bytes"]
    out["cotangent_soa"] = dict(ms_per_step=ms_soa, over_headline=ms_soa / headline_ms, frac_of_hbm_peak=by / (ms_soa * 1e-3) / 1e9 / HBM_PEAK_GBS, bit_identical_to_cotangent_path=bool(torch.equal(ref_du0, du0)),
                                note="Delta handed over as [M][n][ld] (hipadj_adjoint_dev_soa): the sweep reads it in place, one launch")
    out["cotangent_path"] = dict(ms_per_step=ms_aos, over_headline=ms_aos / headline_ms, transposition_ms=ms_aos - ms_soa, extra_hbm_bytes=2.0 * N * M * n * 8,
                                 note="Delta as [N][M][n] (the pullback's shape): every sweep wave transposes the slice of its own knots on the way in (cot_transpose_slice; HIPADJ_COT_INSWEEP=0: the k_aos_to_soa launch of round 4)")
    # --- the host-pointer API the Julia binding calls: pageable host arrays in, host arrays out
    delta_h = delta.cpu().numpy()
    eng.set_timing(0)
    eng.forward(u0_np, p_np, want_out=True); eng.adjoint(delta_h)          # first calls: staging buffers, page faults
    reps = 9
    tf, ta = [], []
    for _ in range(reps):      # every call timed on its own: the MEDIAN stands for the figure (a one-off driver stall of tens of ms was seen on these boxes: the mean of five calls once read 11 ms), the worst call is printed next to it
        t0 = time.perf_counter(); eng.forward(u0_np, p_np, want_out=True); tf.append(time.perf_counter() - t0)
    for _ in range(reps):
        t0 = time.perf_counter(); eng.adjoint(delta_h); ta.append(time.perf_counter() - t0)
    fwd_med, adj_med = float(np.median(tf)), float(np.median(ta))
    link = 63.0e9     # PCIe 5.0 x16, one direction (MI355X_MICROARCH.md)
    up, down = N * M * n * 8.0, N * n * 8.0 + 24.0
    out["host_api"] = dict(forward_ms=fwd_med * 1e3, adjoint_ms=adj_med * 1e3, gradient_ms=(fwd_med + adj_med) * 1e3, calls=reps, statistic="median of the calls",
                           forward_ms_worst_call=max(tf) * 1e3, adjoint_ms_worst_call=max(ta) * 1e3,
                           adjoint_bytes_over_the_link=up + down, adjoint_pcie_bound_ms=(up + down) / link * 1e3,
                           forward_bytes_over_the_link=N * n * 8.0 + 24.0 + N * M * n * 8.0, forward_pcie_bound_ms=(N * n * 8.0 + 24.0 + N * M * n * 8.0) / link * 1e3,
                           note="hipadj_forward (u0 up, out = sol(ts) down) and hipadj_adjoint (Delta up, du0 / dp down) with pageable numpy arrays, synchronous, wall clock; the link, not the kernel, "
                                "sets these: a loss that stays on the device (lsq_data_device) crosses it with N n + np doubles per gradient instead of 2 N M n")
    eng.close()
    return out


def single_process_multi_device(sa, torch, args, world, local_rank):
    """hipadj_config.device_ids: ONE handle over `world` devices in ONE process (what a Julia host that calls `solve` once can reach) — rank 0 runs it on all
    visible devices after the per-process measurement, the other ranks wait.  Device-pointer calls on the primary device, strong scaling of the same 10^4-trajectory ensemble."""
    n_total = args.ntraj
    u0_np, p_np = inputs(n_total)
    dev = torch.device("cuda", local_rank)
    eng = sa.Engine("lorenz", "interpolating", n_total, 0.0, T_FINAL, DT, save_times=save_times(), loss_kind=1, loss_shift=LOSS_SHIFT, p_shared=True, devices=list(range(world)))
    u0 = torch.tensor(u0_np, device=dev); p = torch.tensor(p_np, device=dev)
    du0 = torch.empty((n_total, 3), device=dev, dtype=torch.float64); dp = torch.empty(3, device=dev, dtype=torch.float64)
    stream = torch.cuda.Stream(device=dev)
    eng.set_timing(0)
    with torch.cuda.stream(stream):
        eng.use_torch_stream()
        eng.forward_dev(u0, p, None)
        for _ in range(200 + args.warmup):
            eng.adjoint_dev(None, du0, dp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.steps):
            eng.adjoint_dev(None, du0, dp)
        e1.record(stream)
    torch.cuda.synchronize()
    eng.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    pr = oracle_problem()
    k = 256
    rdu0, _, _, _ = pr.adjoint_ensemble(u0_np[:k], p_np, want_out=False)
    err = float(np.max(np.abs(du0[:k].cpu().numpy() - rdu0)) / np.max(np.abs(rdu0)))
    eng.close()
    return dict(devices=world, ntraj_total=n_total, ms_per_step=ms, value=n_total / (ms * 1e-3), unit="trajectories/s", parity_max_rel_du0_vs_oracle_first_256=err,
                note="one process, one handle over all devices (hipadj_config.device_ids): slices of device 0's buffers go to the other devices by peer copies inside the call, dp partials are summed on device 0")


def self_launch(n):
    """Re-executes this script as n ranks under torch.distributed.run on 127.0.0.1 (a free port), one rank per GPU; returns the exit
    code.  Refuses before spawning when fewer than n devices are visible."""
    import socket
    import subprocess
    if not STUB:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            sys.stderr.write(f"bench.py --gpus {n}: only {have} HIP device(s) visible; refusing to run a smaller job under that label\n")
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this host driver (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def wide_rows(sa, run):
    """The workgroup-per-trajectory family of runtime models on the reference's own two larger problems, each with the roofline that bounds it."""
    rows = []
    rng = np.random.default_rng(11)
    # (i) the 30 x 50 matrix state of test/Core5/size_handling_adjoint.jl:37-70 (df[i,j] = p1 i + p2 j, saveat 0:0.1:1, l = sum(abs2, sol)): no arithmetic to
    #     speak of, so an ensemble of them streams its knots: HBM-bound, 16 n bytes per trajectory and step
    R, Cc, S, dt = 30, 50, 100, 0.01
    n = R * Cc
    ts = np.linspace(0.0, S * dt, 11)
    fun = sa.WideDeviceFunction.index_affine("bench_idxaff", R, Cc)
    for N in (1, 512, 2048):
        eng = sa.Engine(fun.name, "interpolating", N, 0.0, S * dt, dt, save_times=ts)
        ms, kms, st = run(eng, rng.standard_normal((N, n)), rng.random(2), rng.standard_normal((N, len(ts), n)), 5)
        by = N * (S + 1) * 16.0 * n + N * len(ts) * 8.0 * n
        _mark(f"wide model: the reference's 30 x 50 matrix state (test/Cor")
        rows.append(dict(config=f"wide model: the reference's 30 x 50 matrix state (test/Core5/size_handling_adjoint.jl), InterpolatingAdjoint, {S} RK4 steps, N = {N}, "
                                f"{st['workspace_bytes'] / 1e6:.0f} MB workspace", reverse_ms=ms, sweep_kernel_ms=kms, us_per_step=kms * 1e3 / S,
                         roofline=(dict(bound="hbm", achieved=by / (kms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=by / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, kernel="k_wide_adjoint",
                                        algorithmic_bytes_per_launch=by) if N > 1 else
                                   dict(bound="latency", note="one workgroup on one CU: 4 joint-VJP calls per step, two workgroup barriers each", us_per_step=kms * 1e3 / S))))
        eng.close()
    # (ii) the neural ODE of docs/src/Benchmark.md:62 (Chain(x -> x.^3, Dense(2, 50, tanh), Dense(50, 2)), u0 = [2, 0], tspan (0, 1.5), 30 loss times) as a runtime model:
    #      N = 1 is the reference's benchmark shape (its published gradient times on a CPU: 1.66 ms InterpolatingAdjoint / 2.48 ms BacksolveAdjoint with compiled
    #      ReverseDiffVJP, adaptive Tsit5 — another stepper on other hardware, quoted for scale only); an ensemble of 4096 of them is FP64-VALU work
    d, H, T = 2, 50, 1.5
    ts = np.linspace(0.0, T, 30); dtn = T / (29 * 8); Sn = 29 * 8
    fun = sa.WideDeviceFunction.dense_chain("bench_node", (d, H, d), input_power=3)
    p = np.concatenate([rng.standard_normal(H * d) * 0.35, np.zeros(H), rng.standard_normal(d * H) * 0.07, np.zeros(d)])
    flop_vjp = 10.0 * H * d + 2.0 * H + 2.0 * d            # forward recomputation + two transposed products + the outer products, 2 flop per multiply-add
    for alg in ("interpolating", "backsolve"):
        for N in (1, 4096):
            eng = sa.Engine(fun.name, alg, N, 0.0, T, dtn, save_times=ts, checkpointing=(alg == "backsolve"))
            u0 = np.array([2.0, 0.0]) + 0.05 * rng.standard_normal((N, d))
            ms, kms, st = run(eng, u0, p, rng.standard_normal((N, len(ts), d)), 5)
            fl = N * Sn * 4.0 * (flop_vjp + (6.0 * H * d if alg == "backsolve" else 0.0))
            _mark(f"wide model: 2-50-2 neural ODE of docs/src/Benchmark.md as ")
            rows.append(dict(config=f"wide model: 2-50-2 neural ODE of docs/src/Benchmark.md as a runtime model (252 parameters), {alg}, {Sn} RK4 steps, N = {N}",
                             forward_ms=st["forward_ms_last"], reverse_ms=ms, sweep_kernel_ms=kms, us_per_step=kms * 1e3 / Sn,
                             roofline=dict(bound="fp64_valu" if N > 1 else "latency", achieved=fl / (kms * 1e-3) / 1e12, peak=FP64_VALU_PEAK_TF, unit="TFLOP/s",
                                           frac=fl / (kms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF, kernel="k_wide_adjoint" if alg == "interpolating" else "k_wide_backsolve",
                                           note="one wavefront per trajectory, 50 of 64 lanes on the hidden layer; algorithmic flops (10 H d + 2 H + 2 d per joint VJP)")))
            eng.close()
    # (ii-b) dense chains 2 -> H -> H -> 2: the workgroup-per-trajectory family against the FP64-MFMA family that interface.solve routes such chains to (VERDICT r4 next 5b:
    #        "route dense_chain with hidden >= 128 to an MFMA body and show the crossover width"); 4096 trajectories, 150 RK4 steps, GaussAdjoint — configs[3]'s shape
    try:
        from test_gpu_parity import mlp_params
        import scimlsensitivity_jl_amd.interface as _I
        Sx, Tx, Nx = 150, 1.5, 4096
        tsx = np.linspace(0.0, Tx, 16)
        cross = []
        for Hx in (32, 64, 128):
            funx = sa.WideDeviceFunction.dense_chain(f"bench_chain_{Hx}", (2, Hx, Hx, 2))
            px = mlp_params(2, Hx); u0x = rng.standard_normal((Nx, 2)); dx = rng.standard_normal((Nx, len(tsx), 2))
            row = dict(H=Hx)
            # the SAME registered chain twice: as registered (hipadj_config.family = 1) and as the library routes it (family = 0: hipadj_route.hpp puts it on the FP64-MFMA family,
            # transposition kernels of the [N][M][d] blocks included)
            for name, eng, u0e, de in (("workgroup_per_trajectory", sa.Engine(funx.name, "gauss", Nx, 0.0, Tx, Tx / Sx, save_times=tsx, family=1), u0x, dx),
                                       ("fp64_mfma", sa.Engine(funx.name, "gauss", Nx, 0.0, Tx, Tx / Sx, save_times=tsx), u0x, dx)):
                assert (eng.stats()["routed_family"] == 3) == (name == "fp64_mfma")
                ms, kms, st = run(eng, u0e, px, de, 2)
                row[name] = dict(forward_ms=st["forward_ms_last"], reverse_ms=ms)
                eng.close()
            row["mfma_speedup_reverse"] = row["workgroup_per_trajectory"]["reverse_ms"] / row["fp64_mfma"]["reverse_ms"]
            cross.append(row)
        _mark("dense chains 2-H-H-2 (tanh), N = 4096, 150 RK4 steps, Gauss")
        rows.append(dict(config="dense chains 2-H-H-2 (tanh), N = 4096, 150 RK4 steps, GaussAdjoint: the runtime wide model as registered against the FP64-MFMA family hipadj_create routes it to",
                         dense_chain_crossover=cross,
                         note="the MFMA family wins at every width it is built for (32, 64, 128): a chain with H x H contractions belongs on the matrix cores; the published 2-50-2 net has "
                              "none (one hidden layer) and stays on the workgroup family"))
    except Exception as e:      # noqa: BLE001
        _mark("dense_chain_crossover")
        rows.append(dict(config="dense_chain_crossover", error=repr(e)))
    # (iii) the same benchmark AS PUBLISHED: adaptive Tsit5 at the default tolerances (abstol 1e-6, reltol 1e-3) on the runtime model — the workgroup family's adaptive
    #       stepper (per-trajectory step control, dense record; hipadj_wide.hpp).  No roofline: ~20 accepted steps of 7 model evaluations each, a latency chain.
    #       Reference figures of docs/src/Benchmark.md (a CPU, Float32 state, other hardware): InterpolatingAdjoint 1.657 ms, BacksolveAdjoint 2.477 ms per gradient
    #       QuadratureAdjoint 2.490 ms (forward + reverse, compiled ReverseDiffVJP) — quoted for scale, not a vs_baseline.
    pub = dict(interpolating=1.657, backsolve=2.477, quadrature=2.490, gauss=None)
    for alg in ("interpolating", "backsolve", "quadrature", "gauss"):
        for N in (1, 4096):
            eng = sa.Engine(fun.name, alg, N, 0.0, T, 0.0, save_times=ts, checkpointing=(alg == "backsolve"), stepper=1, abstol=1e-6, reltol=1e-3)
            u0 = np.array([2.0, 0.0]) + 0.05 * rng.standard_normal((N, d))
            ms, kms, st = run(eng, u0, p, rng.standard_normal((N, len(ts), d)), 5)
            _mark(f"wide model: 2-50-2 neural ODE of docs/src/Benchmark.md AS ")
            rows.append(dict(config=f"wide model: 2-50-2 neural ODE of docs/src/Benchmark.md AS PUBLISHED (adaptive Tsit5, abstol 1e-6, reltol 1e-3, 30 loss times), {alg}, N = {N}",
                             forward_ms=st["forward_ms_last"], reverse_ms=ms, sweep_kernel_ms=kms, gradient_ms=st["forward_ms_last"] + ms,
                             trajectories_per_s=N / ((st["forward_ms_last"] + ms) * 1e-3),
                             reference_published_cpu_ms=pub[alg], roofline=dict(bound="latency", note="adaptive steps of one workgroup per trajectory; no bandwidth or flop roofline applies at these sizes",
                                                                                kernel="k_wide_backsolve_ts5" if alg == "backsolve" else ("k_wide_adjoint_ts5 + k_wide_quad_gk" if alg == "quadrature" else "k_wide_adjoint_ts5"))))
            eng.close()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ntraj", type=int, default=10000, help="trajectories of the ensemble (strong scaling: in total; weak: per rank)")
    ap.add_argument("--weak", action="store_true", help="N > 1: headline = weak scaling (--ntraj per rank); default = strong (the fixed ensemble sharded)")
    ap.add_argument("--strong", action="store_true", help="(default for N > 1; kept for compatibility)")
    ap.add_argument("--segments", type=int, default=0, help="time segments per trajectory (0 = automatic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip shard_sizes / other_configs (N = 1) and the second scaling figure (N > 1)")
    ap.add_argument("--loss-paths-only", action="store_true", help="with --no-extras: still print loss_paths (the loss routes of the headline pass)")
    ap.add_argument("--torch-allreduce", action="store_true",
                    help="N > 1: all-reduce dL/dp with torch.distributed (async, own stream) instead of in-stream RCCL inside the C ABI (hipadj_comm_*)")
    ap.add_argument("--native-allreduce", action="store_true", help="(default for N > 1; kept for compatibility)")
    ap.add_argument("--no-preamble", action="store_true", help="skip the untimed passes that bring the board to its sustained power state before the warm-up steps (the headline then measures the burst from idle)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes behind roofline.traffic (N = 1)")
    ap.add_argument("--pmc-child", action="store_true", help="(internal) the bare reverse-pass loop that live_traffic() runs under rocprofv3 --pmc")
    args = ap.parse_args()
    if args.pmc_child:
        import scimlsensitivity_jl_amd as sa_child
        pmc_child(sa_child, args.ntraj)
        return

    # ---- launch: `python bench.py --gpus N` with no torchrun environment starts its own N ranks (one per GPU); the driver's torchrun line
    # lands in the else branch with WORLD_SIZE = N.  `--gpus N` never measures fewer than N GPUs.
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or without torchrun: bench.py starts its own ranks)")

    import torch
    import torch.distributed as dist
    if STUB:
        import bench_stub as sa           # tests/bench_stub.py: the launcher / carrier self-test, no kernel runs
    else:
        import scimlsensitivity_jl_amd as sa
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} HIP device(s) visible")
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if STUB:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    native = world > 1 and not args.torch_allreduce
    ts = save_times()
    S = int(round(T_FINAL / DT))

    def measure(strong, steps, warmup):
        n_total = args.ntraj if (strong or world == 1) else args.ntraj * world
        u0_all, p_np = inputs(n_total)
        lo, hi = sa.shard_range(n_total, rank, world)
        r = Runner(sa, torch, dist, args, hi - lo, u0_all[lo:hi], p_np, local_rank, world, native)
        if not args.no_preamble:
            r.power_preamble()
        elapsed = r.timed(steps, warmup)
        region = r.region_ms
        k_ms, st1 = r.profiled(max(10, min(steps, 50)))
        # the same W + K steps as a burst from idle (what round 1-3's headline measured): 0.3 s without work, then the identical timed region
        r.cold_elapsed = None
        if not args.no_preamble and not STUB:
            time.sleep(0.3)
            r.cold_elapsed = r.timed(steps, warmup)
        r.region_ms = region
        return r, n_total, u0_all, p_np, (lo, hi), elapsed, k_ms, st1

    strong = world > 1 and not args.weak
    r, n_total, u0_all, p_np, (lo, hi), elapsed, k_ms, st1 = measure(strong, args.steps, args.warmup)
    res = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total / (elapsed / args.steps)
        fwd_ms = st1["forward_ms_last"]
        # dominant kernel.  One-launch pass (k_interp_fused): the pass IS the kernel, so its duration in the TIMED region is the region's one HIP event pair
        # (on the launch stream) / steps — an upper bound that contains the launch gaps, and what `frac` uses.  Three-launch pass: the library's event pair
        # on the dominant kernel's dispatch packet, from a second loop of the same step (r.profiled).  The dispatch-packet figure of the one-launch pass is
        # kept as `dispatch_event_kernel_ms` for reference only: those events perturb the launch they bracket (5-6 us: profiles/r3_visit2_fused_16B_timing_ab.log).
        alg_bytes = st1["adjoint_algorithmic_bytes"]
        one_launch = st1.get("launches_per_pass", 3) == 1
        region_ms_per_step = r.region_ms / args.steps
        kernel_ms = region_ms_per_step if (one_launch and world == 1) else min(k_ms, region_ms_per_step)   # N > 1: the region also holds the all-reduce
        if not STUB:
            assert kernel_ms <= ms_per_step * 1.0005, f"kernel_ms {kernel_ms} > ms_per_step {ms_per_step}: a kernel cannot outlast the step that contains it"
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        res = {
            "metric": "adjoint_trajectories_per_sec", "value": value, "unit": "trajectories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"Lorenz-63 ensemble, {args.ntraj} trajectories{' in total (sharded)' if strong else ' per GPU' if world > 1 else ''}, "
                                   f"InterpolatingAdjoint, fixed-step RK4 dt={DT}, tspan=(0,{T_FINAL}), loss times 0:{SAVE_DT}:{T_FINAL}, "
                                   f"dgdu = u - {LOSS_SHIFT} (BASELINE configs[1])".replace("  ", " "),
                       "ntraj_total": n_total, "ntraj_per_gpu": hi - lo, "rk4_steps": S, "loss_times": len(ts),
                       "time_segments": st1["time_segments"], "waves_per_workgroup": (grouped_form(hi - lo, st1["time_segments"]) or 1) if one_launch else 1,
                       "parallelism": f"ensemble-shard x{world}",
                       "dp_allreduce": ("none" if world == 1 else ("rccl on the handle's second stream, overlapped with the next pass (hipadj_comm_overlap)" if getattr(r, "native_overlap", False)
                                                                            else "rccl in-stream (hipadj_comm)") if r.native else "torch.distributed nccl, async"),
                       # ranks the dp all-reduce really spans: ncclCommCount of the handle's communicator (native carrier), or the process group's size
                       "rccl_ranks": (0 if world == 1 else r.eng.comm_count() if r.native else dist.get_world_size()),
                       "native_allreduce_fallback": r.native_note},
            "ns_per_vjp_step": elapsed / args.steps / (n_total * S * 4.0) * 1e9,
            "power_preamble_passes": r.preamble_passes,
            "power_preamble_note": ("untimed reverse passes before the W warm-up steps bring the board to its sustained power / clock state (the limiter's transient after idle lasts ~40 ms: "
                                    "profiles/r4_ramp_probe.json); `cold_burst` is the identical W + K-step region started after 0.3 s of idle instead" if r.preamble_passes else None),
            "cold_burst": (None if r.cold_elapsed is None else {"ms_per_step": r.cold_elapsed / args.steps * 1e3, "value": n_total / (r.cold_elapsed / args.steps), "unit": "trajectories/s",
                                                                "whole_pass_frac": st1["adjoint_algorithmic_bytes"] / (r.cold_elapsed / args.steps) / 1e9 / HBM_PEAK_GBS}),
            "forward_solve_ms": fwd_ms,
            "forward_plus_reverse_ms": (fwd_ms + ms_per_step) if fwd_ms is not None else None,
            "roofline": {"bound": "hbm", "kernel": ("k_interp_fused_g" if grouped_form(hi - lo, st1["time_segments"]) else "k_interp_fused") if one_launch else "k_interp", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": "not collected (filled after the timed region at N = 1)",
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kernel_ms,
                         "kernel_ms_note": ("one launch per reverse pass: kernel_ms = the timed region's HIP event pair on the launch stream / steps (contains the launch gaps; the kernel "
                                            "contains the composition tree and the dp reduction); agrees with rocprofv3 --kernel-trace of the same command (profiles/)" if one_launch else
                                            "average launch duration of the dominant kernel from HIP events on its dispatch packet, same step right after the timed region"),
                         "launches_per_pass": st1.get("launches_per_pass"),
                         "region_event_ms_per_step": region_ms_per_step,
                         "dispatch_event_kernel_ms": k_ms,
                         "whole_pass_frac": alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "step_note": "stage operators (the 263-instruction 4-column step) exist for the compiled-in Lorenz model only; other models run the generic multi-column step"},
        }
        # ---- parity: du0 of every trajectory of this rank's shard and (N = 1) the REDUCED dp against the oracle on the same set
        du0 = r.du0.cpu().numpy()
        dp = r.last_dp().cpu().numpy()
        pr = None if STUB else oracle_problem()
        if STUB:      # launcher / carrier self-test: nothing was computed, nothing is compared; the reduced stand-in dp lets the test see the all-reduce
            res.update(metric="STUB_no_kernel_ran", data="STUB (tests/bench_stub.py): launcher and all-reduce carrier self-test on CPU", stub_dp=[float(x) for x in dp])
        elif world == 1:
            rdu0, rdp, _, _ = pr.adjoint_ensemble(u0_all, p_np, want_out=False)          # all trajectories: ~1 s on the host cores
            res["parity_max_rel_du0_vs_oracle"] = float(np.max(np.abs(du0 - rdu0)) / np.max(np.abs(rdu0)))
            res["parity_max_rel_dp_vs_oracle"] = float(np.max(np.abs(dp - rdp) / np.abs(rdp)))
            res["parity_trajectories"] = int(n_total)
        else:
            rdu0, rdp, _, _ = pr.adjoint_ensemble(u0_all, p_np, want_out=False)          # the WHOLE ensemble: dp is the all-reduced sum
            res["parity_max_rel_du0_vs_oracle"] = float(np.max(np.abs(du0 - rdu0[lo:hi])) / np.max(np.abs(rdu0[lo:hi])))
            res["parity_max_rel_dp_vs_oracle"] = float(np.max(np.abs(dp - rdp) / np.abs(rdp)))
            res["parity_trajectories"] = int(n_total)
    r.close()
    if rank == 0:
        _checkpoint(res)
        if os.environ.get("HIPADJ_BENCH_TEST_DIE") == "after_headline":      # tests/test_bench_launch.py: the supervisor's path
            os.abort()
        if os.environ.get("HIPADJ_BENCH_TEST_DIE") == "sleep":               # ... and a worker that is still busy when the launcher ends the rank
            time.sleep(120)

    if world > 1 and not args.no_extras:
        # the other scaling figure, same run, fewer steps
        r2, n2, _, _, _, el2, _, s1 = measure(not strong, max(5, args.steps // 2), args.warmup)
        if rank == 0:
            k2 = max(5, args.steps // 2)
            res["weak_scaling" if strong else "strong_scaling"] = {
                "value": n2 / (el2 / k2), "unit": "trajectories/s", "ms_per_step": el2 / k2 * 1e3, "ntraj_total": n2,
                "time_segments": s1["time_segments"], "steps": k2}
        r2.close()
        # weak scaling at a SATURATING size (10^5 trajectories per GPU): where the per-GPU pass is long enough (~0.9 ms) for the all-reduce and the launch to vanish
        if not STUB:
            keep = args.ntraj
            args.ntraj = 100000
            r3, n3, _, _, _, el3, _, s3 = measure(False, max(5, args.steps // 2), args.warmup)
            args.ntraj = keep
            if rank == 0:
                k3 = max(5, args.steps // 2)
                res["weak_scaling_saturating"] = {"value": n3 / (el3 / k3), "unit": "trajectories/s", "ms_per_step": el3 / k3 * 1e3, "ntraj_total": n3, "ntraj_per_gpu": 100000,
                                                  "time_segments": s3["time_segments"], "steps": k3}
            r3.close()

    if world > 1 and not STUB and not args.no_extras:
        # ONE handle over all devices in ONE process (hipadj_config.device_ids), measured by rank 0 while the other ranks wait at the barrier
        dist.barrier()
        if rank == 0:
            try:
                res["single_process_multi_device"] = single_process_multi_device(sa, torch, args, world, local_rank)
            except Exception as e:      # noqa: BLE001
                res["single_process_multi_device"] = {"error": repr(e)}
        dist.barrier()

    if rank == 0 and world == 1 and not STUB:
        _mark("headline measured")
        if not args.no_extras:
            # the shard sizes of the 8 / 4 / 2-GPU strong-scaling layouts on THIS GPU: the per-rank step time the multi-GPU figure rests on
            sh = []
            _mark("next: shard_sizes")
            for n_s in (1250, 2500, 5000):
                u0s, _ = inputs(10000)
                rs = Runner(sa, torch, dist, args, n_s, u0s[:n_s], p_np, local_rank, 1, False)
                if not args.no_preamble:
                    rs.power_preamble()
                el, reg = min((rs.timed(args.steps, args.warmup), rs.region_ms) for _ in range(2))   # best of two: a one-off driver stall (seen: 47 ms) must not stand for the shard's rate
                s1 = rs.eng.stats()
                km = reg / args.steps if s1.get("launches_per_pass") == 1 else rs.profiled(20)[0]
                sh.append({"ntraj": n_s, "gpus_of_layout": 10000 // n_s, "ms_per_step": el / args.steps * 1e3, "trajectories_per_s": n_s / (el / args.steps),
                           "kernel_ms": km, "time_segments": s1["time_segments"], "launches_per_pass": s1.get("launches_per_pass"),
                           "implied_speedup_if_allreduce_hidden": res["ms_per_step"] / (el / args.steps * 1e3)})
                rs.close()
            res["shard_sizes"] = sh
            _checkpoint(res)
            # the same pass at a SATURATING ensemble (SURVEY.md 8e asks for both sizes): 10^5 trajectories fill the chip with plain one-segment sweeps
            _mark("next: saturating_ensemble")
            try:
                n_sat = 100000
                u0s, _ = inputs(n_sat)
                rs = Runner(sa, torch, dist, args, n_sat, u0s, p_np, local_rank, 1, False)
                rs.power_preamble()
                el = rs.timed(args.steps, args.warmup)
                s1 = rs.eng.stats()
                res["saturating_ensemble"] = {"ntraj": n_sat, "ms_per_step": el / args.steps * 1e3, "trajectories_per_s": n_sat / (el / args.steps), "time_segments": s1["time_segments"],
                                              "whole_pass_frac_of_hbm_peak": s1["adjoint_algorithmic_bytes"] / (el / args.steps) / 1e9 / HBM_PEAK_GBS}
                rs.close()
            except Exception as e:      # noqa: BLE001
                res["saturating_ensemble"] = {"error": repr(e)}
            _mark("next: loss_paths")
            _checkpoint(res)
            try:
                res["loss_paths"] = loss_paths(sa, torch, args, u0_all, p_np, local_rank, res["ms_per_step"])
            except Exception as e:      # noqa: BLE001 — the headline must not die on a secondary figure
                res["loss_paths_error"] = repr(e)
            _mark("next: other_configs")
            _checkpoint(res)
            try:
                res["other_configs"] = other_configs(sa, torch)
            except Exception as e:      # the headline must not die on a secondary figure
                res["other_configs_error"] = repr(e)
        elif args.loss_paths_only:
            res["loss_paths"] = loss_paths(sa, torch, args, u0_all, p_np, local_rank, res["ms_per_step"])
        # the counter passes come LAST among the GPU figures: two rocprofv3 --pmc children of the same workload, while this process — every handle closed — launches nothing more
        # (round 5: a default run died with a GPU memory fault in THIS process at an unknown point after the counter passes had started; nothing measured may depend on them)
        _mark("next: live_traffic (rocprofv3 --pmc children)")
        _checkpoint(res)
        if not args.no_pmc:
            res["roofline"]["traffic"], res["roofline"]["traffic_source"] = live_traffic(n_total)
            if res["roofline"]["traffic"]:
                res["roofline"]["traffic_over_algorithmic"] = res["roofline"]["traffic"] / res["roofline"]["algorithmic_bytes_per_launch"]
        _mark("next: cpu_baseline")
        _checkpoint(res)
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(u0_all, p_np, ts)
    if rank == 0:
        emit(res)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    if os.environ.get("HIPADJ_BENCH_WORKER") == "1" or "--pmc-child" in sys.argv or os.environ.get("HIPADJ_BENCH_SUPERVISE") == "0":
        main()
    else:
        raise SystemExit(supervise())
