"""The CPU oracle behind the same seam-shaped interface as ice_halo_sim_amd.backend.HipTraceBackend.
Checker only: imported by tests, smoke() and bench.py's cpu_baseline leg."""
import ctypes as C

import numpy as np

from ice_halo_sim_amd import abi
from ice_halo_sim_amd.backend import EXIT_DTYPE
from tests import _libs


class OracleBackend:
    def __init__(self, seed=42, fma=False, **options):
        self._L = _libs.oracle(fma=fma)   # fma: the contracted build (tests/_libs.py), the yardstick for ill-conditioned exits
        self._h = self._L.ho_create(int(seed) & 0xFFFFFFFF)
        self._render = None
        self._scene = None
        self._pending_roots = 0
        for k, v in options.items():
            self.set_option(k, v)

    def close(self):
        if self._h:
            self._L.ho_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_option(self, key, value):
        rc = self._L.ho_set_option(self._h, key.encode(), int(value))
        assert rc == 0, key

    def set_filters(self, filters):
        arr = (abi.HaloFilter * max(1, len(filters)))(*filters)
        assert self._L.ho_set_filters(self._h, arr, len(filters)) == 0

    def set_color(self, sets, classes):
        sa = (abi.HaloColorSet * max(1, len(sets)))(*sets)
        ca = (abi.HaloColorClass * max(1, len(classes)))(*classes)
        assert self._L.ho_set_color(self._h, sa, len(sets), ca, len(classes)) == 0
        self._n_classes = len(classes)

    def ReadbackClassLanes(self):
        w, h, n = self._render.width, self._render.height, getattr(self, "_n_classes", 0)
        out = np.zeros((n, h, w), np.float32)
        if n:
            assert self._L.ho_readback_class_lanes(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), w, h, n) == 0
        return out

    def BeginSession(self, scene, render, wl, ray_num=0):
        self._render, self._scene = render, scene
        assert self._L.ho_begin(self._h, C.byref(scene), C.byref(render), C.byref(wl), int(ray_num)) == 0

    def TraceLayer(self, count=0, host_rays=None):
        stats = abi.HaloLayerStats()
        if host_rays is None:
            assert self._L.ho_trace_layer(self._h, int(count), None, C.byref(stats)) == 0
        else:
            d, p, w, tf = (np.ascontiguousarray(host_rays[0], np.float32), np.ascontiguousarray(host_rays[1], np.float32),
                           np.ascontiguousarray(host_rays[2], np.float32), np.ascontiguousarray(host_rays[3], np.uint32))
            crystal = host_rays[4] if len(host_rays) > 4 else None
            hr = abi.HaloHostRays(d.ctypes.data_as(C.POINTER(C.c_float)), p.ctypes.data_as(C.POINTER(C.c_float)),
                                  w.ctypes.data_as(C.POINTER(C.c_float)), tf.ctypes.data_as(C.POINTER(C.c_uint32)),
                                  C.cast(C.pointer(crystal), C.c_void_p) if crystal is not None else None)
            assert self._L.ho_trace_layer(self._h, w.shape[0], C.byref(hr), C.byref(stats)) == 0
        self._pending_roots += int(stats.root_count)
        return stats

    def Recombine(self, shuffle=True):
        n = C.c_uint64()
        assert self._L.ho_recombine(self._h, 1 if shuffle else 0, C.byref(n)) == 0
        return n.value

    def continuation(self):
        n = self._L.ho_continuation_dump(self._h, None, 0)
        out = np.zeros((n, 5), np.float32)
        if n:
            self._L.ho_continuation_dump(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), n)
        return out

    def DrainExits(self, max_records=None):
        if max_records is None:
            max_records = max(1, self._pending_roots * (self._scene.max_hits + 1))
        buf = (abi.HaloExitRecord * int(max_records))()
        n = C.c_uint64()
        assert self._L.ho_drain_exits(self._h, buf, int(max_records), C.byref(n)) == 0
        self._pending_roots = 0
        return np.frombuffer(buf, dtype=EXIT_DTYPE, count=min(n.value, int(max_records))).copy()

    def ReadbackXyzAccum(self):
        w, h = self._render.width, self._render.height
        img = np.empty((h, w, 3), np.float32)
        landed = C.c_double()
        assert self._L.ho_readback_xyz64(self._h, img.ctypes.data_as(C.POINTER(C.c_float)), w, h, C.byref(landed)) == 0
        return img, landed.value

    def EndSession(self):
        assert self._L.ho_end(self._h) == 0

    def ConsumeDeviceFused(self):
        assert self._L.ho_consumer_fold(self._h) == 0

    def Consume(self, xyz, landed, lanes=None):
        xyz = np.ascontiguousarray(xyz, np.float32)
        h, w = xyz.shape[0], xyz.shape[1]
        lp, nc = None, 0
        if lanes is not None:
            lanes = np.ascontiguousarray(lanes, np.float32)
            lp, nc = lanes.ctypes.data_as(C.POINTER(C.c_float)), lanes.shape[0]
        assert self._L.ho_consumer_consume(self._h, xyz.ctypes.data_as(C.POINTER(C.c_float)), w, h, float(landed), lp, nc) == 0
        self._cons_size = (w, h)

    def Snapshot(self, intensity_factor=1.0, ray_color=(-1.0, -1.0, -1.0), background=(0.0, 0.0, 0.0), want_xyz=True):
        w, h = getattr(self, "_cons_size", None) or (self._render.width, self._render.height)
        d = abi.HaloDisplay(float(intensity_factor), (C.c_float * 3)(*ray_color), (C.c_float * 3)(*background))
        rgb = np.empty((h, w, 3), np.uint8)
        xyz = np.empty((h, w, 3), np.float32) if want_xyz else None
        tot = C.c_double()
        assert self._L.ho_consumer_snapshot(self._h, C.byref(d), rgb.ctypes.data_as(C.POINTER(C.c_uint8)),
                                            xyz.ctypes.data_as(C.POINTER(C.c_float)) if want_xyz else None, C.byref(tot)) == 0
        return rgb, xyz, tot.value


def run_session(backend, scene, render, wl, n_rays, shuffle=True):
    """BeginSession → layers (TraceLayer → Recombine) → EndSession, like
    Simulator::SimulateOneWavelengthWithBackend (reference simulator.cpp:1498-1560). Returns per-layer stats."""
    backend.BeginSession(scene, render, wl, n_rays)
    stats = []
    for li in range(scene.layer_count):
        stats.append(backend.TraceLayer(n_rays if li == 0 else 0))
        if li + 1 < scene.layer_count:
            backend.Recombine(shuffle)
    backend.EndSession()
    return stats
