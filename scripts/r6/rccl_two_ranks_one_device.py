#!/usr/bin/env python
"""What does a TWO-rank RCCL communicator on ONE device say?  (every multi-rank number of this repository is "unmeasured" because the pool has 1-GPU boxes;
this records, once, what the library's own communicator path — hipadj_comm_unique_id / hipadj_comm_init_rank / hipadj_comm_selfcheck / an all-reduced reverse pass — does when
both ranks sit on device 0.)  Two processes, 60 s limit each; prints one JSON line with each rank's outcome.   python scripts/r6/rccl_two_ranks_one_device.py"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def rank_main(rank, idfile, outfile):
    import numpy as np
    import scimlsensitivity_jl_amd as sa
    res = dict(rank=rank)
    try:
        if rank == 0:
            uid = sa.comm_unique_id()
            with open(idfile + ".tmp", "wb") as f:
                f.write(bytes(uid))
            os.replace(idfile + ".tmp", idfile)
        else:
            t0 = time.time()
            while not os.path.exists(idfile) and time.time() - t0 < 30:
                time.sleep(0.05)
            uid = open(idfile, "rb").read()
        N = 1250
        ts = np.linspace(0.0, 10.0, 101)
        eng = sa.Engine("lorenz", "interpolating", N, 0.0, 10.0, 0.01, save_times=ts, loss_kind=1, loss_shift=2.0)
        rng = np.random.default_rng(20240601 + rank)
        u0 = np.array([1.0, 0.0, 0.0]) + 0.1 * rng.standard_normal((N, 3)); p = np.array([10.0, 28.0, 8.0 / 3.0])
        eng.forward(u0, p, want_out=False)
        _, dp_local = eng.adjoint(None)
        res["dp_local"] = [float(x) for x in dp_local]
        t0 = time.time()
        eng.comm_init_rank(uid, 2, rank)
        res["comm_init_rank_s"] = round(time.time() - t0, 2)
        eng.comm_selfcheck()
        res["selfcheck"] = "ok"
        res["comm_count"] = eng.comm_count()
        _, dp = eng.adjoint(None)
        res["dp_allreduced"] = [float(x) for x in dp]
        eng.close()
        res["ok"] = True
    except Exception as e:      # noqa: BLE001
        res["ok"] = False; res["error"] = repr(e)[:600]
    with open(outfile, "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "_rank":
        rank_main(int(sys.argv[2]), sys.argv[3], sys.argv[4])
        sys.exit(0)
    d = tempfile.mkdtemp(prefix="hipadj_rccl2_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "_rank", str(r), os.path.join(d, "id"), os.path.join(d, f"out{r}.json")], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in (0, 1)]
    out = {"what": "two ranks of the library's RCCL communicator on ONE device (device 0), one process each", "ranks": []}
    for r, pr in enumerate(procs):
        try:
            _, err = pr.communicate(timeout=90)
            row = json.load(open(os.path.join(d, f"out{r}.json"))) if os.path.exists(os.path.join(d, f"out{r}.json")) else dict(rank=r, ok=False, error="no result file", exit_code=pr.returncode)
            row["stderr_tail"] = err.decode(errors="replace")[-400:]
        except subprocess.TimeoutExpired:
            pr.kill(); _, err = pr.communicate()
            row = dict(rank=r, ok=False, error="did not finish within 90 s (communicator bootstrap or collective hung)", stderr_tail=err.decode(errors="replace")[-400:])
        out["ranks"].append(row)
    if all(x.get("ok") for x in out["ranks"]):
        a, b = out["ranks"]
        s = [x + y for x, y in zip(a["dp_local"], b["dp_local"])]
        out["allreduce_is_the_sum_of_the_shards"] = all(abs(v - w) <= 1e-12 * abs(w) for v, w in zip(a["dp_allreduced"], s)) and a["dp_allreduced"] == b["dp_allreduced"]
    print(json.dumps(out))
