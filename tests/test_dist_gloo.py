"""The N>1 path on CPU: world_size 2, gloo.  Rays shard by root-index range with disjoint RNG counter ranges per
rank and the only exchange is ONE sum-reduce of the accumulator (+ landed scalar) at the drain point.  The
tracer behind each rank here is the CPU oracle (the HIP library needs a GPU); the sharding and reduction code
under test — ice_halo_sim_amd.dist.shard_range / reduce_accumulators — is exactly what bench.py runs over RCCL."""
import os
import socket

import numpy as np
import pytest

from ice_halo_sim_amd import abi, scenes
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


# ---- world 8: the two BASELINE readings of "rays sharded across 8" (configs[3]: 400 M rays single-scatter column, configs[4]: 200 M rays of
# the stochastic-geometry scene with a D65 pool), scaled down 10^4-fold, as ONE strong-scaling job each ---------------------------------------
W8 = 8


def _job(kind):
    from ice_halo_sim_amd import abi
    if kind == "configs3":       # BASELINE configs[3]: single-scatter hex column, 400 M rays over 8 GPUs
        return scenes.config2_scene(), scenes.config2_render(W, H), scenes.wl_discrete(550.0), 40_000
    # BASELINE configs[4]: examples/bench_config_stoch.json's shape, 200 M rays over 8 GPUs
    sc = scenes.scene([(0.0, [scenes.stochastic_prism_entry()])], max_hits=8)
    return sc, scenes.render(abi.LENS_RECTANGULAR, W, H, el=0.0, visible=abi.VISIBLE_FULL), scenes.wl_illuminant("D65", 31), 20_000


def _trace_job_shard(kind, rank, world):
    from tests._oracle_backend import OracleBackend, run_session
    sc, rd, wl, total = _job(kind)
    start, count = shard_range(total, rank, world)
    ob = OracleBackend(seed=42, rank=rank, threads=1)
    st = run_session(ob, sc, rd, wl, count)
    img, landed = ob.ReadbackXyzAccum()
    ob.close()
    return img, landed, int(st[0].root_count), (start, count)


def _worker8(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from ice_halo_sim_amd.dist import reduce_image, reduce_scalar
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert dist.get_world_size() == world
    for kind in ("configs3", "configs4"):
        img, landed, roots, span = _trace_job_shard(kind, rank, world)
        acc = torch.zeros(W * H * 3 + 4, dtype=torch.float32)
        acc[: W * H * 3] = torch.from_numpy(img.ravel())
        own_y = float(img[..., 1].sum(dtype=np.float64))
        acc = reduce_image(acc)                       # the drain-point collective
        total_landed = reduce_scalar(landed, acc.device)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "roots": roots, "span": span, "sum_y": own_y, "landed": landed})
        np.save(os.path.join(out_dir, "%s_acc%d.npy" % (kind, rank)), acc.numpy())
        if rank == 0:
            np.save(os.path.join(out_dir, "%s_meta.npy" % kind), np.array([total_landed] + [g["sum_y"] for g in per_rank] + [g["landed"] for g in per_rank] + [g["roots"] for g in per_rank]))
    dist.barrier()
    dist.destroy_process_group()


def test_world8_gloo_strong_scaling_jobs(tmp_path):
    """Eight ranks (gloo, CPU; the oracle as each rank's tracer) run configs[3]'s and configs[4]'s "N rays sharded across 8" as one job each:
    shard_range cuts the job's roots into eight contiguous spans, rank r traces its span on the counter range r << 40, ONE reduce sums
    the images on rank 0 and drains the others.  The reduced image is the sum of the eight shards, the shards are eight DIFFERENT sets of
    rays of equal brightness, and the job's landed weight is that of the same job on one rank up to Monte-Carlo scatter."""
    import torch.multiprocessing as mp
    from tests._oracle_backend import OracleBackend, run_session
    port = _free_port()
    mp.spawn(_worker8, args=(W8, port, str(tmp_path)), nprocs=W8, join=True)
    for kind in ("configs3", "configs4"):
        sc, rd, wl, total = _job(kind)
        meta = np.load(tmp_path / ("%s_meta.npy" % kind))
        total_landed, ys, landeds, roots = meta[0], meta[1:1 + W8], meta[1 + W8:1 + 2 * W8], meta[1 + 2 * W8:]
        assert roots.sum() == total and roots.max() - roots.min() <= 1          # the job's rays, dealt out evenly
        accs = [np.load(tmp_path / ("%s_acc%d.npy" % (kind, r))) for r in range(W8)]
        assert all(not a.any() for a in accs[1:])                                # non-root ranks are drained
        shards = [_trace_job_shard(kind, r, W8) for r in range(W8)]             # the same shards again, in this process
        want = sum(s[0].astype(np.float64) for s in shards)
        assert np.allclose(accs[0][: W * H * 3], want.ravel(), rtol=2e-6, atol=1e-6 * want.max())
        assert total_landed == pytest.approx(sum(s[1] for s in shards), rel=1e-12)
        assert np.allclose(ys, [s[0][..., 1].sum(dtype=np.float64) for s in shards], rtol=1e-9)
        assert len({round(float(y), 6) for y in ys}) == W8                       # eight different images ...
        assert np.all(np.abs(landeds / landeds.mean() - 1.0) <= 0.05)            # ... of the same brightness (a few thousand rays each)
        # the whole job on ONE rank: other rays (counter range 0 only), the same statistics
        ob = OracleBackend(seed=42, threads=4)
        run_session(ob, sc, rd, wl, total)
        _, landed1 = ob.ReadbackXyzAccum()
        ob.close()
        assert total_landed == pytest.approx(landed1, rel=0.02)


# ---- raypath-colour jobs: the class lanes reduce like the image, and the composite happens ONCE, on the root ---------------------------
def _lane_shard(rank, world, n_total):
    """one rank's class lanes of a two-crystal raypath-colour scene (the oracle tracer), and its landed weight"""
    from tests._oracle_backend import OracleBackend, run_session
    T = scenes.filter_term
    ob = OracleBackend(seed=42, rank=rank)
    ee = lambda lo=1, hi=None: T("entry_exit", min_len=lo, max_len=hi)
    ob.set_color([scenes.color_set([(ee(1), "", 0)]), scenes.color_set([(ee(2, 2), "", 1), (ee(3), "", 2)])],
                 [scenes.color_class([0]), scenes.color_class([1]), scenes.color_class([2])])
    full = {"type": "uniform", "mean": 0.0, "std": 360.0}
    ax = scenes.axis(zenith={"type": "gauss", "mean": 90.0, "std": 20.0}, azimuth=full, roll=full)
    sc = scenes.scene([(0.0, [scenes.entry(scenes.prism_crystal(1.3), ax, 0.5, 1, color_id=1), scenes.entry(scenes.prism_crystal(0.4), ax, 0.5, 2, color_id=2)])],
                      max_hits=6, sun_altitude=25.0)
    rd = scenes.render(abi.LENS_DUAL_FISHEYE_EQUAL_AREA, 64, 32, visible=abi.VISIBLE_FULL)
    _, count = shard_range(n_total, rank, world)
    run_session(ob, sc, rd, scenes.wl_discrete(550.0), count)
    _, landed = ob.ReadbackXyzAccum()
    lanes = ob.ReadbackClassLanes()
    ob.close()
    return lanes, landed


def _lane_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from ice_halo_sim_amd.dist import reduce_class_lanes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lanes, landed = _lane_shard(rank, world, N)
    summed, tot = reduce_class_lanes(lanes, landed)
    assert (summed is None) == (rank != 0)
    if rank == 0:
        np.save(os.path.join(out_dir, "lanes.npy"), summed)
        np.save(os.path.join(out_dir, "tot.npy"), np.array([tot]))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_class_lanes_reduce_and_one_composite_on_the_root(tmp_path):
    import torch.multiprocessing as mp
    from tests.test_compositor import oracle_composite
    port = _free_port()
    mp.spawn(_lane_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    summed, tot = np.load(tmp_path / "lanes.npy"), float(np.load(tmp_path / "tot.npy")[0])
    (l0, w0), (l1, w1) = _lane_shard(0, 2, N), _lane_shard(1, 2, N)
    assert np.array_equal(summed, l0 + l1) and tot == pytest.approx(w0 + w1, rel=1e-12)
    assert all(float(l0[c].max()) > 0 and float(l1[c].max()) > 0 for c in range(3))
    cls = [{"color": c, "bits": 1 << i} for i, c in enumerate(([1, 0, 0], [0, 1, 0], [0, 0, 1]))]
    ok, whole, _, p99 = oracle_composite(summed, tot, cls, "painter", 1.0, 1.0)
    assert ok and p99 > 0
    # why the composite is not done per rank: each shard's own P99 anchors its own exposure, and the parts do not add up to the whole
    parts = [oracle_composite(l, w, cls, "painter", 1.0, 1.0) for l, w in ((l0, w0), (l1, w1))]
    assert all(p[0] for p in parts) and min(parts[0][3], parts[1][3]) < p99
    assert np.abs(parts[0][1] + parts[1][1] - whole).max() > 0.05


def test_bench_self_launch_starts_ranks_and_relays_their_exit_code():
    """`python bench.py --gpus 2` with WORLD_SIZE unset becomes the launcher (bench.self_launch).  Without a GPU the ranks refuse to run —
    the product has no CPU path — and the launcher must hand that failure back; under RCCL it must refuse before starting anything."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import torch
    if torch.cuda.is_available():
        pytest.skip("the GPU half is tests/test_gpu_dist.py::test_bench_launches_its_own_ranks_when_started_plainly")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=dict(env, HALO_BENCH_BACKEND="nccl"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "only 0 HIP device(s) visible" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=dict(env, HALO_BENCH_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stderr.count("bench.py needs an MI355X") >= 2, r.stderr[-1500:]      # both ranks started, both said why
