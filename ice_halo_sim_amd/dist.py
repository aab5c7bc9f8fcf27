"""Multi-GPU sharding of the trace path: one process per GPU, rays split by root-index range, ONE sum-reduce of the
XYZ accumulator (+ the landed-weight scalar) at the drain point.  torch is plumbing here: it owns the accumulator
tensor (so torch.distributed — RCCL on ROCm, gloo in CPU tests — can reduce it in place) and the stream.

Reference: there is no multi-GPU code in Lumice; the only reduction is N worker threads → one consumer
(src/server/server.cpp:1189).  The drain point we reduce at is Simulator::DrainDeviceXyz (simulator.cpp:1409-1477).
"""
import time

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
    """One rank's slice of a trace job on one MI355X.

    The drain-point collective is OFF the critical path (round 6): the rank owns TWO accumulator tensors.  reduce_to_root() queues the reduce
    of the tensor just traced into on a side stream (behind everything the trace stream holds for it) and binds the other tensor, so step
    k + 1's sessions trace while step k's image crosses xGMI; a tensor is traced into again only behind the reduce that last read it (an
    event wait on the stream, never the host).  On the root the two tensors hold the running totals of the even and the odd drains: the image
    is their sum (total(), readback()).  With one rank there is no collective and one tensor."""

    def __init__(self, scene, render, seed=42, device=0, rank=0, world=1, overlap_reduce=True, **options):
        import torch
        from .backend import HipTraceBackend
        self.torch = torch
        self.scene, self.render = scene, render
        self.rank, self.world = rank, world
        self.device = torch.device("cuda", device)
        self.backend = HipTraceBackend(device=device, seed=seed, **options)
        self.backend.set_option("rank", rank)      # disjoint 64-bit RNG counter range per shard
        self.n_floats = render.width * render.height * 3 + 4
        self.two = bool(overlap_reduce) and world > 1
        self.accs = [torch.zeros(self.n_floats, dtype=torch.float32, device=self.device) for _ in range(2 if self.two else 1)]
        self.cur = 0
        self.side = torch.cuda.Stream(device=self.device) if world > 1 else None
        self.reduced = [None] * len(self.accs)      # event behind the last reduce (and drain) of each tensor, on the side stream
        self.pending = [None] * len(self.accs)      # (work, start event, host time) of a queued reduce whose completion has not been ordered yet
        self.reduce_spans = []                      # (start event, end event, host t_call, host t_completed) of every reduce, events ON THE SIDE STREAM
        self.reduce_log = []                        # host times of every queued reduce: {"t_call", "t_done"}
        self.backend.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.backend.bind_accumulator(self.acc.data_ptr(), self.n_floats)
        # the accumulator is this object's tensor: it reads it only at the drain points below, so a session's closing fold may stay pending at
        # EndSession and run under the next session's trace kernels (flush() brings the tensor up to date before anything here touches it)
        self.backend.set_option("defer_fold", 1)
        self.landed = 0.0

    @property
    def acc(self):
        """the tensor the backend is tracing into"""
        return self.accs[self.cur]

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
        """The drain-point collective: ONE sum-reduce of the image accumulator, queued behind this rank's trace kernels; non-root ranks
        are drained.  With two tensors it runs on the side stream while the backend goes on tracing into the other one.  The landed-weight
        scalars stay on their devices until readback()."""
        torch = self.torch
        self.backend.flush()                          # the pending planes belong to the tensor that is about to leave
        if self.world == 1:
            return
        import torch.distributed as dist
        main = torch.cuda.current_stream(self.device)
        k = self.cur
        buf = self.accs[k]
        ready = torch.cuda.Event()
        ready.record(main)
        self.side.wait_event(ready)
        gloo = dist.get_backend() == "gloo"
        with torch.cuda.stream(self.side):
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(self.side)
            # async_op: under gloo (the one-GPU rehearsal: device -> host -> TCP -> device on gloo's own threads) a blocking call would hold
            # THIS thread until the image is back, and nothing of the next step could be queued; under RCCL the call never blocks the host
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True) if gloo else dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM, async_op=True)
        self.pending[k] = (work, e0, time.perf_counter())
        # when the collective really ended, whenever this thread next looks (gloo completes lazily, below): its future's callback stamps the host time
        idx = len(self.reduce_log)
        self.reduce_log.append({"t_call": time.perf_counter(), "t_done": None})
        try:
            work.get_future().add_done_callback(lambda f, i=idx: self.reduce_log[i].__setitem__("t_done", time.perf_counter()))
        except Exception:   # a backend without futures: the spans' events still say what happened
            pass
        if not self.two:
            # serialised drain (overlap_reduce=False, bench.py's A/B): the same queue, but the trace stream waits for the collective before anything else
            self._complete(k)
            main.wait_event(self.reduced[k])
            return
        if not gloo:
            self._complete(k)                         # RCCL: wait() only orders streams — the side stream behind the collective, the drain behind that
        self.cur ^= 1
        self._complete(self.cur)                      # the tensor about to be traced into: its last reduce (two drains ago) must be through
        if self.reduced[self.cur] is not None:
            main.wait_event(self.reduced[self.cur])
        self.backend.bind_accumulator(self.acc.data_ptr(), self.n_floats)

    def _complete(self, k):
        """finish the queued reduce of tensor k: the side stream waits for the collective (gloo: so does this thread), the other ranks' copy
        is drained, and the event later users of the tensor wait for is recorded"""
        if self.pending[k] is None:
            return
        work, e0, t_call = self.pending[k]
        self.pending[k] = None
        torch = self.torch
        import torch.distributed as dist
        with torch.cuda.stream(self.side):
            work.wait()
            if dist.get_rank() != 0:
                self.accs[k].zero_()
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(self.side)
        self.reduced[k] = e1
        self.reduce_spans.append((e0, e1, t_call, time.perf_counter()))

    def _join(self):
        """the current stream waits for every queued reduce"""
        main = self.torch.cuda.current_stream(self.device)
        for k in range(len(self.accs)):
            self._complete(k)
            if self.reduced[k] is not None:
                main.wait_event(self.reduced[k])

    def set_overlap(self, on):
        """switch the drain between the side-stream reduce (two tensors) and the in-line one (what round 5 shipped) — bench.py times both"""
        if len(self.accs) < 2:
            return
        self._join()
        if self.cur != 0:                             # in-line mode keeps everything in tensor 0
            self.backend.flush()
            self.accs[0] += self.accs[1]
            self.accs[1].zero_()
            self.cur = 0
            self.backend.bind_accumulator(self.acc.data_ptr(), self.n_floats)
        self.two = bool(on)

    def total(self):
        """The image this rank holds (root: the reduced running total; others: what they have traced since their last drain), as a
        tensor of n_floats — behind the pending folds and every queued reduce."""
        self.backend.flush()
        self._join()
        return self.accs[0].clone() if len(self.accs) == 1 else self.accs[0] + self.accs[1]

    def zero(self):
        self.backend.take_landed()
        self.backend.flush()
        self._join()
        for a in self.accs:
            a.zero_()
        self.landed = 0.0
        self.torch.cuda.synchronize(self.device)

    def readback(self):
        """COLLECTIVE (every rank calls it): the root gets the reduced image and the summed landed weight; all ranks are
        left drained."""
        self.landed = reduce_scalar(self.landed + self.backend.take_landed(), self.device)
        w, h = self.render.width, self.render.height
        img = self.total()[: w * h * 3].cpu().numpy().reshape(h, w, 3).copy()
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
