"""bench.py's N>1 path without GPUs: `python bench.py --gpus 2 --dry-run` must re-execute itself under
torch.distributed.run with 2 ranks (free port, 127.0.0.1), form the process group, run the step's collective sequence
(all-gather p, reduce-scatter, bucketed gradient all-reduce through comm.GradReducer), take the max-over-ranks time and
print ONE JSON line from rank 0 with n_gpus = 2 and the rccl block.  The reference launches the same way:
run_inbatch.sh:50-53 (python -m torch.distributed.run --nproc_per_node=$NPROC train.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "2"
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True,
                       timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def test_gpus_flag_spawns_that_many_ranks_and_reports_them():
    p, lines = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--pairs", "8"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout          # exactly one JSON line, from rank 0
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["config"]["global_batch"] == 16 and r["config"]["pairs_per_gpu"] == 8 and r["config"]["parallelism"] == "dp2"
    assert r["config"]["final_loss"] == 0.0            # the stand-in step checks every collective's result (nan otherwise)
    assert r["value"] > 0 and abs(r["value"] - 16 * 3 / (r["ms_per_step"] * 3e-3)) < 1e-2 * r["value"]
    rc = r["rccl"]
    assert rc["ranks_seen"] == 2 and rc["backend"] == "gloo"
    assert rc["grad_allreduce"]["overlapped_with_backward"] and rc["grad_allreduce"]["collectives_per_step"] >= 3
    for k in ("all_gather_p_ms", "reduce_scatter_dp_ms", "all_reduce_grads_ms"):
        assert rc[k] > 0


def test_single_rank_dry_run_needs_no_launcher():
    p, lines = _run(["--gpus", "1", "--dry-run", "--steps", "2", "--warmup", "0", "--pairs", "4"])
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and "rccl" not in r and r["config"]["global_batch"] == 4


def test_world_size_mismatch_is_an_error():
    p, _ = _run(["--gpus", "4", "--dry-run"], {"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                             "MASTER_PORT": "29999"})
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)


import pytest


@pytest.mark.gpu
def test_bench_line_schema_on_the_device():
    """the real (non dry-run) bench path at a small configuration: one JSON line with the contract's keys, a live GEMM sample
    from the library-side timing hook, the measured-traffic record only for the configuration it was measured on"""
    p, lines = _run(["--steps", "2", "--warmup", "1", "--pairs", "16", "--model", "ViT-B/32", "--no-cpu-baseline", "--no-secondary",
                     "--no-retrieval"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["dtype"] == "bf16" and r["data"] == "synthetic" and r["vs_baseline"] is None
    assert "workload" in r["config"] and r["config"]["global_batch"] == 16
    rf = r["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert rf["launches_timed"] > 0 and rf["achieved"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] is None            # profiles/pmc_gemm_traffic.json is the ViT-L/14, 512-pair measurement
    assert abs(r["value"] - 16 * 2 / (r["ms_per_step"] * 2e-3)) < 1e-2 * r["value"]


@pytest.mark.gpu
def test_bench_two_ranks_sharing_the_gpu():
    """the REAL multi-rank bench path (self-launch, replica sync at the DDP point, overlapped bucketed all-reduce, the rccl block,
    per-rank checksums, and the N > 1 retrieval / embedding blocks) with two ranks on the one GPU a test box has, over gloo (UNIIR_BENCH_SHARED_GPU=1; RCCL cannot put two
    ranks on one device): ranks seeded differently must end the timed steps with identical parameters"""
    p, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "16", "--model", "ViT-B/32", "--no-cpu-baseline",
                     "--shard-rows", "40030", "--shard-queries", "64,300", "--embed-items", "64", "--check-sharded"],
                    {"UNIIR_BENCH_SHARED_GPU": "1"})
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, (lines, p.stderr[-2000:])
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 32 and r["config"]["parallelism"] == "dp2"
    rc = r["rccl"]
    assert rc["ranks_seen"] == 2 and rc["replicas_identical"] is True and len(rc["replica_checksums"]) == 2
    assert rc["grad_allreduce"]["collectives_per_step"] >= 1
    assert abs(r["value"] - 32 * 2 / (r["ms_per_step"] * 2e-3)) < 1e-2 * r["value"]
    # the other two BASELINE metrics at N > 1: sharded retrieval (configs[3]) and embedding extraction (configs[2])
    rt = r["retrieval"]
    assert "error" not in rt, rt
    assert rt["ranks_seen"] == 2 and rt["pool_rows"] == 2 * 40030 and rt["rows_per_rank"] == 40030
    for key, nq in (("q64", 64), ("q300", 300)):
        q = rt[key]
        assert q["equals_single_shard_search"] is True         # == one search over the concatenated pool, bit for bit
        assert q["queries_per_rank"] == nq // 2 and q["ms"] > 0 and q["M_candidates_per_s"] > 0
        for piece in ("all_gather_queries_ms", "gather_topk_ms", "merge_ms"):
            assert q[piece] > 0
    em = r["embed"]
    assert "error" not in em, em
    assert em["items_per_batch_per_rank"] == 64 and em["out_shape"] == [64, 512] and em["value"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_over_rccl_when_two_gpus_are_visible():
    """Round 5: the first box that shows two GPUs runs the multi-rank bench over RCCL (backend "nccl", one rank per device) UNDER A
    CHECKER before any 8-GPU measurement does: ranks seen, replicas identical after the timed steps, the sharded search equal to the
    single-shard search, every collective timed.  Skipped on the one-GPU boxes of this environment (the shared-GPU gloo test above
    covers the same code path there)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (RCCL puts one rank on each)")
    p, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "16", "--model", "ViT-B/32", "--no-cpu-baseline",
                     "--shard-rows", "40030", "--shard-queries", "64,300", "--embed-items", "64", "--check-sharded"],
                    {"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert p.returncode == 0, p.stderr[-3000:]
    assert len(lines) == 1, (lines, p.stderr[-2000:])
    r = json.loads(lines[0])
    rc = r["rccl"]
    assert r["n_gpus"] == 2 and rc["backend"] == "nccl" and rc["ranks_seen"] == 2 and rc["replicas_identical"] is True
    for k in ("all_gather_p_ms", "reduce_scatter_dp_ms", "all_reduce_grads_ms"):
        assert rc[k] > 0
    rt = r["retrieval"]
    assert "error" not in rt and rt["ranks_seen"] == 2
    assert all(rt[key]["equals_single_shard_search"] is True for key in ("q64", "q300"))
    assert "error" not in r["embed"] and r["embed"]["value"] > 0
