"""GPU: bench.py's other entry paths run and print the contract's line -- the multi-target workload (config 5, what every rank of an
N > 1 launch also runs as `batch_mode`) and the launch the driver uses for N > 1 (`python -m torch.distributed.run ... bench.py
--gpus N`), rehearsed with two ranks on the one GPU of the box (TRX2_BENCH_FORCE_DEVICE=0: gloo instead of RCCL, which refuses two
ranks on one device).  Control flow only: no performance is read off these runs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(out):
    return json.loads([l for l in out.strip().splitlines() if l.startswith("{")][-1])


def test_bench_multi_target_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "5", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d["metric"] == "decoys/sec" and d["n_gpus"] == 1 and d["scaling"] == "strong" and d["value"] > 0 and d["all_decoys_converged"]
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1 and len(d["per_rank"]) == 1


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, TRX2_BENCH_FORCE_DEVICE="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--nmax", "12"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    # N > 1: `value` is the metric's job on one target PER RANK (weak scaling); --nmax 12 shortens it for this control-flow test and says so
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["metric"] == "decoys/sec"
    assert "init_num=10" in d["config"]["workload"] and "NOT the CLI default" in d["config"]["workload"] and len(d["per_rank_seconds"]) == 2
    assert d["config"]["decoys_per_job"] == 2 * (10 + 12) and abs(d["value"] * d["ms_per_step"] * 2e-3 - 2 * 44) < 1e-6
    assert 0 < d["roofline"]["frac"] < 1 and 0 < d["roofline_step"]["frac"] < 1
    s = d["sub_records"]
    w = s["config2"]                                   # the weak-scaling calls of config 2 stay on the line as a sub-record
    assert w["scaling"] == "weak" and w["n_gpus"] == 2 and w["value"] > 0 and w["all_decoys_converged"] and len(w["per_rank_seconds"]) == 2
    bm = s["config5_batch_mode"]
    assert bm["scaling"] == "strong" and len(bm["per_rank"]) == 2 and bm["all_decoys_converged"] and bm["value"] > 0
    bq = s["batch_mode_queue"]                         # 16 targets per GPU pulled from the shared queue: the work grows with N
    assert bq["scaling"] == "weak" and bq["all_targets_folded"] and "32 independent targets" in bq["workload"]
    assert bq["decoys_per_step"] == 32 * (2 * 10 + 2 * 10) and sum(bq["per_rank_decoys_last_step"]) == 1280 and min(bq["per_rank_decoys_last_step"]) > 0


def test_bench_on_rccl_with_one_rank():
    """The N > 1 path on RCCL itself, as far as one GPU allows: torchrun with ONE rank and TRX2_BENCH_REHEARSE_NCCL=1 takes every branch a
    multi-GPU run takes in bench.py -- process group on nccl (= RCCL) bound to the device, barriers around the timed job, the object broadcast
    of the work directory, the CUDA all-gather / all-reduce of the per-rank seconds, teardown -- in the same process as libtrx2fold's streams and
    engine threads.  World size 1: it shows that RCCL initialises and the collectives run with this code ('nobody has seen
    RCCL initialise with this code'), nothing about scaling; the queue's store and the gloo summary group short-circuit at one rank and are
    covered by the two-rank tests (tests/test_sharding.py, test_bench_two_ranks_on_one_gpu)."""
    env = dict(os.environ, TRX2_BENCH_REHEARSE_NCCL="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TRX2_BENCH_FORCE_DEVICE", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29519", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--nmax", "8"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["decoys_per_job"] == 2 * (10 + 8) and len(d["per_rank_seconds"]) == 1
    bq = d["sub_records"]["batch_mode_queue"]
    assert bq["all_targets_folded"] and "16 independent targets" in bq["workload"] and bq["decoys_per_step"] == 16 * (2 * 10 + 2 * 10)
    w = d["sub_records"]["config2"]
    assert w["scaling"] == "weak" and w["value"] > 0 and w["all_decoys_converged"]
