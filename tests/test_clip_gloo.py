"""Multi-rank host logic on CPU (gloo, world_size 2 and 3): window sharding has no data-path
collective and the final gather reassembles the frames in window order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from coupe.dvsg_amd.clip import shard_range


def test_shard_range_partitions_contiguously():
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(512, 8, 3) == (192, 256)          # configs[3]: 64 windows per GPU
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, result_dir):
    import torch.distributed as dist
    from coupe.dvsg_amd.clip import stabilize_windows_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H, W = 6, 8
        g = torch.Generator().manual_seed(5)
        patches = torch.rand((n, H, W, 21), generator=g)
        u = patches[..., 18:].contiguous()
        calls = []

        def run_fn(p, uu):           # stand-in hot path: tags every frame with its own content
            calls.append(p.shape[0])
            return uu * 2.0 + p[..., :3]

        out = stabilize_windows_sharded(run_fn, patches, u, batch=3, dst=0)
        lo, hi = shard_range(n, world, rank)
        assert sum(calls) == hi - lo and all(c <= 3 for c in calls)
        if rank == 0:
            assert out.shape == (n, H, W, 3)
            assert torch.equal(out, u * 2.0 + patches[..., :3])
            np.save(os.path.join(result_dir, "ok.npy"), np.array([n, world]))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 10), (2, 7), (3, 8), (3, 2)])
def test_sharded_windows_gather_in_order(tmp_path, world, n):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    assert np.array_equal(np.load(tmp_path / "ok.npy"), [n, world])
