"""bench.py's job-size handling and its SHARDED path, in the GPU-less build container.

The driver runs `python bench.py --gpus N` (N = 1) and, for N > 1, `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N`.  A line must only ever be printed for the job size asked for:
  * `--gpus N` without a launcher starts its own N ranks (and refuses when the box shows fewer GPUs);
  * a launcher that started another number of ranks than --gpus is refused (exit 2), never answered with an N = 1 line;
  * the line echoes the rank count the ENGINE's communicator reports (gk_comm_info: ncclCommCount), `rccl_ranks`.
The sharded path itself -- objects block-sharded over the ranks, one all-gather of [bitmaps | counts | tail] per sweep issued by
the engine (the reference's loop is serial, pkg/audit/manager.go:591-642) -- runs here at world size 2 on the
TEST-ONLY CPU emulation (`--test-hostemu`: tests/native/libgkgpu_hostemu.so, collectives through gloo) and must report the same
global totals as one process over the same objects."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GK_FORCE_DIST"):
        e.pop(k, None)
    e.update(env or {})
    e["GK_HOST_THREADS"] = "2"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=timeout, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_gpus_flag_without_enough_gpus_fails_loudly():
    """`python bench.py --gpus 2` on a box that shows fewer than two GPUs: no ranks are started, no line is printed"""
    rc, line, err = _run(["--gpus", "2", "--lean", "--steps", "1", "--warmup", "0"])
    assert rc == 2 and line is None and "only" in err and "GPU" in err


def test_world_size_that_differs_from_gpus_is_refused():
    rc, line, err = _run(["--gpus", "2", "--test-hostemu", "--lean", "--reviews", "256", "--steps", "1", "--warmup", "0"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc == 2 and line is None and "--gpus 2" in err
    rc, line, err = _run(["--gpus", "1", "--test-hostemu", "--lean", "--reviews", "256", "--steps", "1", "--warmup", "0"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc == 2 and line is None


def test_sharded_bench_at_world_size_two_equals_one_process():
    """weak scaling: two ranks x 1 344 objects = objects [0, 2 688) of the stream = one process over 2 688 objects"""
    common = ["--test-hostemu", "--lean", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    rc, two, err = _run(["--gpus", "2", "--reviews", "1344"] + common)       # (no launcher: bench.py starts its own two ranks)
    assert rc == 0 and two is not None, err[-2000:]
    rc, one, err = _run(["--gpus", "1", "--reviews", "2688"] + common)
    assert rc == 0 and one is not None, err[-2000:]
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and two["scaling"] == "weak" and two["emulated"] and two["steps"] == 2 and two["warmup"] == 1
    assert one["n_gpus"] == 1 and one["rccl_ranks"] is None
    assert two["config"]["reviews_total"] == one["config"]["reviews_total"] == 2688 and two["config"]["reviews_rank0"] == 1344
    assert two["config"]["global_violating_pairs"] == one["config"]["global_violating_pairs"] > 1000
    assert two["config"]["violating_pairs_rank0"] < two["config"]["global_violating_pairs"]
    assert "CPU EMULATION" in two["data"]
    # the exchange step is readable from the line: bytes received per rank = one other rank's slot, the all-gather's own duration, the local
    # sweep beside it, whether the overlapped pipeline is on (never in the emulation), and what bounds a step; absent at N = 1
    x = two["exchange"]
    assert one["exchange"] is None
    assert set(x) >= {"exchange_bytes_per_rank", "exchange_ms", "sweep_ms_local", "overlap_enabled", "hidden_by_overlap", "bound", "slot_bytes"}
    assert x["exchange_bytes_per_rank"] == x["slot_bytes"] * (two["n_gpus"] - 1) > 50 * (1344 // 64) * 8
    assert x["exchange_ms"] > 0 and x["sweep_ms_local"] > 0 and x["overlap_enabled"] is False and x["bound"] in ("exchange", "sweep")
    # strong scaling: 2 688 objects split over two ranks
    rc, strong, err = _run(["--gpus", "2", "--reviews", "2688", "--scaling", "strong"] + common)
    assert rc == 0 and strong is not None, err[-2000:]
    assert strong["scaling"] == "strong" and strong["config"]["reviews_total"] == 2688 and strong["config"]["reviews_rank0"] == 1344
    assert strong["config"]["global_violating_pairs"] == one["config"]["global_violating_pairs"]
