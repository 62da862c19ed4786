"""N>1 path on CPU: two gloo ranks each take their contiguous chunk of the batch, run the per-image
path (the oracle stands in for the GPU kernels here - there is no CPU product path) and rank 0
gathers.  No data-path collective is involved; the result must equal the unsharded one."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from yolact_amd import synth
from yolact_amd.sharding import concat_results, gather_results, shard_bounds, slice_net_outs


def test_shard_bounds_cover_the_batch():
    for B in (0, 1, 2, 5, 32, 64, 65):
        for G in (1, 2, 3, 4, 8):
            spans = [shard_bounds(B, G, r) for r in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(G - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q):
    from oracle import c_oracle as co
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net, shp = synth.make_toy(21, batch=B)
    lo, hi = shard_bounds(B, world, rank)
    local = None
    if hi > lo:
        local = co.detect(slice_net_outs(net, lo, hi), shp, top_k=50, nms_mode="per_class", max_dets=20)
    dist.barrier()
    full = gather_results(local, dst=0)
    if rank == 0:
        ref = co.detect(net, shp, top_k=50, nms_mode="per_class", max_dets=20)
        ok = all(np.array_equal(full[k], ref[k]) for k in ref)
        q.put(bool(ok) and full["num_detections"].shape[0] == B)
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 1])
def test_two_rank_gloo_sharding_matches_unsharded(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def test_concat_results_torch_and_numpy():
    a = {"num_detections": np.array([1, 2]), "x": np.zeros((2, 3))}
    b = {"num_detections": np.array([3]), "x": np.ones((1, 3))}
    out = concat_results([a, None, b])
    assert out["x"].shape == (3, 3) and list(out["num_detections"]) == [1, 2, 3]
    ta = {k: torch.from_numpy(v) for k, v in a.items()}
    tb = {k: torch.from_numpy(v) for k, v in b.items()}
    out = concat_results([ta, tb])
    assert tuple(out["x"].shape) == (3, 3)
