"""The N > 1 path on the GPU box's ONE device: two ranks share it over gloo and run exactly what bench.py runs per rank —
ice_halo_sim_amd.dist.ShardedTracer (the HIP backend accumulating straight into a torch tensor, rank-offset RNG counters) and the
drain-point reduce — so the reduced image can be checked against the ranks' own images.  (The driver's 8-GPU run uses RCCL on the
same code; gloo reduces CUDA tensors as all_reduce, which dist.reduce_image knows.)"""
import os
import socket

import numpy as np
import pytest

from ice_halo_sim_amd import abi, scenes

pytestmark = pytest.mark.gpu

W, H, N = 480, 270, 3 << 20   # per rank: above the hit log's 2 Mi-ray threshold, so the production route runs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from ice_halo_sim_amd.dist import ShardedTracer
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, rd = scenes.config2_scene(), scenes.config2_render(W, H)
    tr = ShardedTracer(sc, rd, seed=42, device=0, rank=rank, world=world, **{"async": 1})
    for wl in (scenes.wl_discrete(530.0), scenes.wl_discrete(610.0)):
        tr.trace_session_layers(wl, N)
    torch.cuda.synchronize()
    own = tr.acc[: W * H * 3].cpu().numpy().copy()          # this rank's image before the reduce
    own_landed = tr.backend.take_landed()
    tr.landed = own_landed                                     # (readback adds the device tally: put it back)
    route = tr.backend.last_route()
    tr.reduce_to_root()
    img, landed = tr.readback()
    np.save(os.path.join(out_dir, "own%d.npy" % rank), own)
    np.save(os.path.join(out_dir, "red%d.npy" % rank), img.ravel())
    np.save(os.path.join(out_dir, "meta%d.npy" % rank), np.array([own_landed, landed, route.mode_mask, route.accum_mask], np.float64))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reduce_to_the_sum_of_their_images(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    own = [np.load(tmp_path / ("own%d.npy" % r)) for r in range(2)]
    red = [np.load(tmp_path / ("red%d.npy" % r)) for r in range(2)]
    meta = [np.load(tmp_path / ("meta%d.npy" % r)) for r in range(2)]
    assert all(int(m[2]) == abi.MODE_PLAIN and int(m[3]) == abi.ACCUM_LOG for m in meta)      # the production route, on both ranks
    # rank 0 holds the sum of the two images (one float add per element: exact), rank 1 is drained
    assert np.array_equal(red[0], own[0] + own[1])
    assert not red[1].any()
    assert meta[0][1] == pytest.approx(meta[0][0] + meta[1][0], rel=1e-12) and meta[1][1] == 0.0
    # the two shards are different rays (rank-offset counters), equally bright
    assert not np.allclose(own[0], own[1])
    assert meta[0][0] == pytest.approx(meta[1][0], rel=5e-3)
    assert own[0].sum(dtype=np.float64) == pytest.approx(own[1].sum(dtype=np.float64), rel=5e-3)
