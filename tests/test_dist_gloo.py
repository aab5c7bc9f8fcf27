"""The N>1 path on CPU: world_size 2, gloo.  Rays shard by root-index range with disjoint RNG counter ranges per
rank and the only exchange is ONE sum-reduce of the accumulator (+ landed scalar) at the drain point.  The
tracer behind each rank here is the CPU oracle (the HIP library needs a GPU); the sharding and reduction code
under test — ice_halo_sim_amd.dist.shard_range / reduce_accumulators — is exactly what bench.py runs over RCCL."""
import os
import socket

import numpy as np
import pytest

from ice_halo_sim_amd import scenes
from ice_halo_sim_amd.dist import shard_range

W, H, N = 96, 48, 30_000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _trace_shard(rank, world, n_total):
    from tests._oracle_backend import OracleBackend, run_session
    sc, rd = scenes.config2_scene(), scenes.config2_render(W, H)
    _, count = shard_range(n_total, rank, world)
    ob = OracleBackend(seed=42, rank=rank)
    run_session(ob, sc, rd, scenes.wl_discrete(550.0), count)
    img, landed = ob.ReadbackXyzAccum()
    ob.close()
    return img, landed


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from ice_halo_sim_amd.dist import reduce_accumulators, reduce_image, reduce_scalar
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    img, landed = _trace_shard(rank, world, N)
    acc = torch.zeros(W * H * 3 + 4, dtype=torch.float32)
    acc[: W * H * 3] = torch.from_numpy(img.ravel())
    # the two-step form bench.py uses (image every step, landed scalar once at readback) must give the same result
    acc2 = reduce_image(acc.clone())
    landed2 = reduce_scalar(landed, acc.device)
    acc, landed = reduce_accumulators(acc, landed)
    assert torch.equal(acc, acc2) and landed == landed2
    np.save(os.path.join(out_dir, "acc%d.npy" % rank), acc.numpy())
    np.save(os.path.join(out_dir, "landed%d.npy" % rank), np.array([landed]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 8, 1000, 50_000_001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == total
            pos = 0
            for s, c in spans:
                assert s == pos
                pos += c
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_world2_gloo_reduce_matches_sum_of_shards(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    acc0 = np.load(tmp_path / "acc0.npy")
    acc1 = np.load(tmp_path / "acc1.npy")
    landed0 = float(np.load(tmp_path / "landed0.npy")[0])
    landed1 = float(np.load(tmp_path / "landed1.npy")[0])
    # reference: the same two shards traced in this process and summed
    (i0, l0), (i1, l1) = _trace_shard(0, 2, N), _trace_shard(1, 2, N)
    assert np.allclose(acc0[: W * H * 3], (i0 + i1).ravel(), rtol=1e-6, atol=1e-6)
    assert landed0 == pytest.approx(l0 + l1, rel=1e-12)
    assert not acc1.any() and landed1 == 0.0           # non-root ranks are drained by the reduce
    # the two shards are different rays (disjoint counter ranges), not replicas
    assert not np.allclose(i0, i1)
    assert l0 != l1
