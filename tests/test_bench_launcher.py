"""bench.py's own multi-rank launch (`--gpus N` with no WORLD_SIZE in the environment) and its
strong-scaling mode (BASELINE.json configs 3 and 5: one batch of 64 sharded over the GPUs)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, extra_env, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env)
    p = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


def test_gpus_flag_starts_that_many_ranks_and_prints_one_line():
    """CPU box: the parent starts 3 ranks before anything touches a GPU, they rendezvous over gloo, cut
    config 3's batch of 64 into contiguous chunks and rank 0 prints the one line (dry run: no kernels)."""
    p, lines = _run(["--gpus", "3", "--config", "3", "--steps", "1", "--warmup", "0"],
                    {"YP_BENCH_DRY": "1", "YP_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["scaling"] == "strong" and d["dry_run"] is True
    assert [tuple(c) for c in d["config"]["chunks"]] == [(0, 22), (22, 43), (43, 64)]
    assert d["config"]["images_per_step"] == 64


def test_weak_scaling_is_the_default_for_the_headline_config():
    p, lines = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"YP_BENCH_DRY": "1", "YP_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert [tuple(c) for c in d["config"]["chunks"]] == [(0, 32), (32, 64)] and d["config"]["images_per_step"] == 64


def test_world_size_that_contradicts_gpus_is_refused():
    p, lines = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"],
                    {"YP_BENCH_DRY": "1", "YP_BENCH_BACKEND": "gloo", "WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0 and not lines
    assert "WORLD_SIZE" in p.stderr


@pytest.mark.gpu
def test_two_ranks_on_the_one_gpu_box_shard_one_batch():
    """The real kernels under the self-launch: 2 ranks (both on device 0, gloo for the barrier) split a
    batch of 8 config-3 images; one line, n_gpus 2, a measured value."""
    p, lines = _run(["--gpus", "2", "--config", "3", "--batch", "8", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                    {"YP_BENCH_ONE_DEVICE": "1", "YP_BENCH_BACKEND": "gloo"}, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["batch_per_gpu"] == 4 and d["config"]["images_per_step"] == 8
    assert "sharded over 2" in d["config"]["workload"]


def test_eight_ranks_rendezvous_and_cut_the_batches_of_configs_3_and_5():
    """The 8-GPU shape of BASELINE configs 3 and 5 rehearsed on the CPU (gloo, no kernels): eight ranks, one batch of
    64, eight contiguous chunks of 8 in rank order."""
    for cfg in ("3", "5"):
        p, lines = _run(["--gpus", "8", "--config", cfg, "--steps", "1", "--warmup", "0"],
                        {"YP_BENCH_DRY": "1", "YP_BENCH_BACKEND": "gloo"}, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 8 and d["scaling"] == "strong"
        assert [tuple(c) for c in d["config"]["chunks"]] == [(8 * r, 8 * r + 8) for r in range(8)]


def test_launcher_stops_the_other_ranks_when_one_dies():
    """A rank that fails before the rendezvous leaves the others waiting for it: the launcher must notice, stop
    them and exit non-zero instead of hanging."""
    import time
    t0 = time.monotonic()
    p, lines = _run(["--gpus", "3", "--steps", "1", "--warmup", "0"],
                    {"YP_BENCH_DRY": "1", "YP_BENCH_BACKEND": "gloo", "YP_BENCH_DIE_RANK": "1"}, timeout=240)
    assert p.returncode != 0 and not lines
    assert "rank 1 exited with status 3" in p.stderr
    assert time.monotonic() - t0 < 120
