"""Multi-GPU sharding of the trace path: one process per GPU, rays split by root-index range, ONE sum-reduce of the
XYZ accumulator (+ the landed-weight scalar) at the drain point.  torch is plumbing here: it owns the accumulator
tensor (so torch.distributed — RCCL on ROCm, gloo in CPU tests — can reduce it in place) and the stream.

Reference: there is no multi-GPU code in Lumice; the only reduction is N worker threads → one consumer
(src/server/server.cpp:1189).  The drain point we reduce at is Simulator::DrainDeviceXyz (simulator.cpp:1409-1477).
"""
import numpy as np


def shard_range(total, rank, world):
    """Contiguous split of `total` root rays: rank r gets [start, start+count); counts differ by at most one."""
    base, rem = divmod(int(total), int(world))
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def reduce_accumulators(acc, landed, group=None, dst=0):
    """Sum-reduce every rank's accumulator tensor and landed scalar onto `dst`; non-root ranks are drained (zeroed).
    Works on any torch.distributed backend (nccl == RCCL on ROCm; gloo on CPU)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return acc, landed
    rank = dist.get_rank(group)
    lt = torch.tensor([landed], dtype=torch.float64, device=acc.device)
    if acc.is_cuda and dist.get_backend(group) == "gloo":   # gloo reduces CUDA tensors only as all_reduce
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(lt, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM, group=group)
        dist.reduce(lt, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if rank != dst:
        acc.zero_()
        return acc, 0.0
    return acc, float(lt.item())


def reduce_image(acc, group=None, dst=0):
    """Sum-reduce the accumulator tensors onto `dst` and drain the other ranks.  Stream-ordered: nothing waits on the host
    (the backend launches on torch's current stream, the collective is enqueued behind it)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return acc
    if acc.is_cuda and dist.get_backend(group) == "gloo":   # gloo reduces CUDA tensors only as all_reduce
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if dist.get_rank(group) != dst:
        acc.zero_()
    return acc


def reduce_scalar(value, device, group=None, dst=0):
    """Sum of one float64 per rank, valid on `dst` (0.0 elsewhere)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if t.is_cuda and dist.get_backend(group) == "gloo":
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()) if dist.get_rank(group) == dst else 0.0


def reduce_class_lanes(lanes, total_intensity, group=None, dst=0, device=None):
    """Raypath-colour jobs: sum-reduce every rank's class lanes (numpy (classes, H, W) float32, as ReadbackClassLanes hands them out)
    and total intensity onto `dst` — ONE more reduce of classes * W * H floats at the drain point, next to the image's.  The root gets
    (summed lanes, summed intensity) to load into its consumer (`HipTraceBackend.LoadClassLanes`) and composite once
    (`CompositeColorClasses`: the participating P99 is a statistic of the WHOLE image, so compositing per rank and adding would be
    wrong); other ranks get (None, 0.0).  `device`: where the transfer tensor lives (a cuda device under RCCL, None = CPU for gloo)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return lanes, total_intensity
    t = torch.from_numpy(np.ascontiguousarray(lanes, np.float32).reshape(-1))
    if device is not None:
        t = t.to(device)
    t = reduce_image(t, group, dst)
    tot = reduce_scalar(float(total_intensity), t.device, group, dst)
    if dist.get_rank(group) != dst:
        return None, 0.0
    return t.cpu().numpy().reshape(np.shape(lanes)), tot


class ShardedTracer:
    """One rank's slice of a trace job on one MI355X."""

    def __init__(self, scene, render, seed=42, device=0, rank=0, world=1, **options):
        import torch
        from .backend import HipTraceBackend
        self.torch = torch
        self.scene, self.render = scene, render
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", device)
        self.backend = HipTraceBackend(device=device, seed=seed, **options)
        self.backend.set_option("rank", rank)      # disjoint 64-bit RNG counter range per shard
        self.n_floats = render.width * render.height * 3 + 4
        self.acc = torch.zeros(self.n_floats, dtype=torch.float32, device=self.device)
        self.backend.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.backend.bind_accumulator(self.acc.data_ptr(), self.n_floats)
        # the accumulator is this object's tensor: it reads it only at the drain points below, so a session's closing fold may stay pending at
        # EndSession and run under the next session's trace kernels (flush() brings the tensor up to date before anything here touches it)
        self.backend.set_option("defer_fold", 1)
        self.landed = 0.0

    def trace_session_layers(self, wl, n_rays, shuffle=True):
        """BeginSession → layers → EndSession for this rank's `n_rays` roots. Returns every layer's stats (with option async=1
        the final layer's tallies are deferred to collect_stats; the layers before it are always synchronous — the host needs
        their continuation counts)."""
        b = self.backend
        b.BeginSession(self.scene, self.render, wl, n_rays)
        sts = []
        for li in range(self.scene.layer_count):
            sts.append(b.TraceLayer(n_rays if li == 0 else 0))
            if li + 1 < self.scene.layer_count:
                b.Recombine(shuffle)
        b.EndSession()
        return sts

    def trace_session(self, wl, n_rays, shuffle=True):
        """Same; returns the last layer's stats."""
        return self.trace_session_layers(wl, n_rays, shuffle)[-1]

    def reduce_to_root(self):
        """The drain-point collective: ONE sum-reduce of the image accumulator, enqueued on the stream behind this rank's
        trace kernels; non-root ranks are drained.  The landed-weight scalars stay on their devices until readback()."""
        self.backend.flush()
        self.acc = reduce_image(self.acc)

    def zero(self):
        self.backend.take_landed()
        self.acc.zero_()
        self.landed = 0.0
        self.torch.cuda.synchronize(self.device)

    def readback(self):
        """COLLECTIVE (every rank calls it): the root gets the reduced image and the summed landed weight; all ranks are
        left drained."""
        self.landed = reduce_scalar(self.landed + self.backend.take_landed(), self.device)
        w, h = self.render.width, self.render.height
        img = self.acc[: w * h * 3].cpu().numpy().reshape(h, w, 3).copy()
        landed = self.landed
        self.zero()
        return img, landed

    def composite(self, classes, mode="painter", display_exposure_scale=1.0, intensity_factor=1.0, total_intensity=None):
        """COLLECTIVE, raypath-colour jobs: every rank hands over its class lanes, the root composites the sum on its device.
        `total_intensity`: this rank's landed weight so far (default: what the backend has tallied, read without draining the image).
        Returns (produced, linear_rgb, srgb, p99) on the root, None elsewhere."""
        lanes = self.backend.ReadbackClassLanes()
        mine = self.landed + self.backend.take_landed() if total_intensity is None else total_intensity
        if total_intensity is None:
            self.landed = mine      # the scalar was taken off the device: keep it for readback()
        lanes, tot = reduce_class_lanes(lanes, mine, device=self.device)
        if lanes is None:
            return None
        self.backend.LoadClassLanes(lanes, tot)
        return self.backend.CompositeColorClasses(classes, mode, display_exposure_scale, intensity_factor)
