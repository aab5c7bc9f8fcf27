"""Host-side mirror of the reference's trace-backend seam over the C ABI (libhalo_hip.so).

`HipTraceBackend` follows `lumice::TraceBackend` (reference src/core/backend/trace_backend.hpp:367-641)
method for method: BeginSession / TraceLayer / Recombine / DrainExits / ReadbackXyzAccum / EndSession,
same state machine, `BackendUnavailableError` as the one recoverable error.  It is a thin ctypes layer:
all computing happens in the HIP library, and importing this module without the built library (or
without a gfx950 device at create time) fails loudly — there is no CPU fallback here.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# HALO_LIB selects another build of the same library (tools/phase_probe.py loads the -DHALO_PROBE build); default = the product
LIB_PATH = os.environ.get("HALO_LIB") or os.path.join(_HERE, "libhalo_hip.so")
_lib = None


class BackendUnavailableError(RuntimeError):
    """Reference: BackendUnavailableError, trace_backend.hpp:140-158 (the only recoverable error)."""


class BackendError(RuntimeError):
    pass


def load_library():
    """dlopen the in-tree HIP library and declare every symbol of include/halo_trace.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libhalo_hip.so is not built — run `python -m ice_halo_sim_amd.build` (needs hipcc)")
    L = C.CDLL(LIB_PATH)
    H = C.c_void_p
    f32p = C.POINTER(C.c_float)
    sig = {
        "halo_abi_version": (C.c_int, []),
        "halo_abi_sizeof": (C.c_uint64, [C.c_int]),
        "halo_device_count": (C.c_int, []),
        "halo_create": (C.c_int, [C.c_int, C.c_uint32, C.POINTER(H)]),
        "halo_destroy": (C.c_int, [H]),
        "halo_last_error": (C.c_char_p, [H]),
        "halo_set_option": (C.c_int, [H, C.c_char_p, C.c_int64]),
        "halo_set_stream": (C.c_int, [H, C.c_void_p]),
        "halo_bind_accumulator": (C.c_int, [H, C.c_void_p, C.c_uint64]),
        "halo_set_filters": (C.c_int, [H, C.POINTER(abi.HaloFilter), C.c_int32]),
        "halo_begin": (C.c_int, [H, C.POINTER(abi.HaloScene), C.POINTER(abi.HaloRender), C.POINTER(abi.HaloWl), C.c_uint64]),
        "halo_trace_layer": (C.c_int, [H, C.c_uint64, C.POINTER(abi.HaloHostRays), C.POINTER(abi.HaloLayerStats)]),
        "halo_recombine": (C.c_int, [H, C.c_int, C.POINTER(C.c_uint64)]),
        "halo_drain_exits": (C.c_int, [H, C.POINTER(abi.HaloExitRecord), C.c_uint64, C.POINTER(C.c_uint64)]),
        "halo_end": (C.c_int, [H]),
        "halo_readback_xyz": (C.c_int, [H, f32p, C.c_int, C.c_int, f32p]),
        "halo_readback_xyz64": (C.c_int, [H, f32p, C.c_int, C.c_int, C.POINTER(C.c_double)]),
        "halo_sync": (C.c_int, [H]),
        "halo_last_sample_counts": (C.c_int, [H, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "halo_last_route": (C.c_int, [H, C.POINTER(abi.HaloRouteInfo)]),
        "halo_set_color": (C.c_int, [H, C.POINTER(abi.HaloColorSet), C.c_int, C.POINTER(abi.HaloColorClass), C.c_int]),
        "halo_readback_class_lanes": (C.c_int, [H, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int]),
        "halo_generate_shapes": (C.c_int, [H, C.POINTER(abi.HaloCrystal), C.c_uint64, C.c_uint32, C.c_int, C.POINTER(abi.HaloGeomTables)]),
        "halo_collect_stats": (C.c_int, [H, C.POINTER(abi.HaloLayerStats)]),
        "halo_take_landed": (C.c_int, [H, C.POINTER(C.c_double)]),
        "halo_flush": (C.c_int, [H]),
        "halo_collect_timing": (C.c_int, [H, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
        "halo_consumer_fold": (C.c_int, [H]),
        "halo_consumer_consume": (C.c_int, [H, f32p, C.c_int, C.c_int, C.c_float, f32p, C.c_int]),
        "halo_consumer_snapshot": (C.c_int, [H, C.POINTER(abi.HaloDisplay), C.POINTER(C.c_uint8), f32p, C.POINTER(C.c_double)]),
        "halo_consumer_reset": (C.c_int, [H]),
        "halo_consumer_composite": (C.c_int, [H, C.POINTER(abi.HaloComposite), f32p, C.POINTER(C.c_uint8), f32p, C.POINTER(C.c_int32)]),
        "halo_consumer_load_lanes": (C.c_int, [H, f32p, C.c_int, C.c_int, C.c_int, C.c_double]),
        "halo_host_parse_composite_mode": (C.c_int, [C.c_char_p]),
        "halo_host_prism_geometry": (C.c_int, [C.c_float, f32p, C.POINTER(abi.HaloGeomTables)]),
        "halo_host_pyramid_geometry": (C.c_int, [C.c_float] * 5 + [f32p, C.POINTER(abi.HaloGeomTables)]),
        "halo_host_shape_scalars": (C.c_int, [C.POINTER(abi.HaloCrystal), C.c_uint32, C.c_uint64, C.c_int, f32p]),
        "halo_host_build_lat_lut": (C.c_int, [C.POINTER(abi.HaloDist), f32p, f32p, f32p]),
        "halo_host_build_proj_params": (C.c_int, [C.POINTER(abi.HaloRender), C.c_void_p]),
        "halo_host_partition": (C.c_int, [f32p, C.c_int, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
        "halo_host_refractive_index": (C.c_double, [C.c_double]),
        "halo_reduce_accumulator": (C.c_int, [H, C.c_void_p, C.c_int, C.c_int]),
        "halo_host_illuminant_spd": (C.c_float, [C.c_int, C.c_float]),
        "halo_host_wl_pool": (C.c_int, [C.POINTER(abi.HaloWl), f32p, C.c_int]),
        "halo_host_reduce_raypath": (C.c_int, [C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]),
        "halo_host_filter_fast_check": (C.c_int, [C.POINTER(abi.HaloFilter), C.POINTER(abi.HaloAxis), C.POINTER(C.c_uint8), C.c_int32, C.POINTER(C.c_float), C.c_int32,
                                                  C.POINTER(C.c_int32)]),
        "halo_host_color_fast_mask": (C.c_int, [C.POINTER(abi.HaloColorSet), C.POINTER(abi.HaloAxis), C.POINTER(C.c_uint8), C.c_int32, C.POINTER(C.c_float), C.c_int32,
                                                C.c_uint64, C.POINTER(C.c_uint64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "halo_abi_version", "halo_abi_sizeof", "halo_device_count", "halo_create", "halo_destroy", "halo_last_error",
    "halo_set_option", "halo_set_stream", "halo_bind_accumulator", "halo_set_filters", "halo_begin", "halo_trace_layer", "halo_recombine",
    "halo_drain_exits", "halo_end", "halo_readback_xyz", "halo_readback_xyz64", "halo_sync", "halo_flush", "halo_collect_timing", "halo_last_sample_counts", "halo_last_route", "halo_set_color", "halo_readback_class_lanes", "halo_generate_shapes", "halo_collect_stats", "halo_take_landed", "halo_consumer_fold", "halo_consumer_consume", "halo_consumer_snapshot", "halo_consumer_reset", "halo_consumer_composite", "halo_consumer_load_lanes", "halo_host_parse_composite_mode", "halo_host_prism_geometry",
    "halo_host_pyramid_geometry", "halo_host_shape_scalars", "halo_host_build_lat_lut", "halo_host_build_proj_params", "halo_host_partition",
    "halo_host_refractive_index", "halo_host_reduce_raypath", "halo_host_filter_fast_check", "halo_host_color_fast_mask", "halo_host_illuminant_spd", "halo_host_wl_pool", "halo_reduce_accumulator",
]


class HipTraceBackend:
    """One backend instance = one `Simulator::Run()` thread's backend (single-threaded use)."""

    def __init__(self, device=0, seed=42, **options):
        self._L = load_library()
        self._h = C.c_void_p()
        rc = self._L.halo_create(int(device), int(seed) & 0xFFFFFFFF, C.byref(self._h))
        if rc == abi.HALO_UNAVAILABLE:
            raise BackendUnavailableError("no gfx950 device %d (halo_create)" % device)
        if rc != abi.HALO_OK:
            raise BackendError("halo_create failed")
        self._render = None
        self._scene = None
        self._pending_roots = 0
        for k, v in options.items():
            self.set_option(k, v)

    # --- plumbing ---
    def _check(self, rc):
        if rc == abi.HALO_OK:
            return
        msg = self._L.halo_last_error(self._h)
        msg = msg.decode() if msg else "?"
        if rc == abi.HALO_UNAVAILABLE:
            raise BackendUnavailableError(msg)
        raise BackendError(msg)

    def close(self):
        if self._h:
            self._L.halo_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        self._check(self._L.halo_set_option(self._h, key.encode(), int(value)))

    def set_stream(self, hip_stream_ptr):
        self._check(self._L.halo_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def set_filters(self, filters):
        """Filter table referenced by HaloEntry.filter_id (1-based; 0 = none)."""
        arr = (abi.HaloFilter * max(1, len(filters)))(*filters)
        self._check(self._L.halo_set_filters(self._h, arr, len(filters)))

    def set_color(self, sets, classes):
        """Raypath-colour tables: `sets` referenced by HaloEntry.color_id (1-based), `classes` define the Y lanes."""
        sa = (abi.HaloColorSet * max(1, len(sets)))(*sets)
        ca = (abi.HaloColorClass * max(1, len(classes)))(*classes)
        self._check(self._L.halo_set_color(self._h, sa, len(sets), ca, len(classes)))
        self._n_classes = len(classes)

    def ReadbackClassLanes(self):
        """TraceBackend::ReadbackClassLanes: (class_count, H, W) float32 Y lanes; zeroes the device lanes."""
        w, h, n = self._render.width, self._render.height, getattr(self, "_n_classes", 0)
        out = np.zeros((n, h, w), np.float32)
        if n:
            self._check(self._L.halo_readback_class_lanes(self._h, out.ctypes.data_as(C.POINTER(C.c_float)), w, h, n))
        return out

    def bind_accumulator(self, device_ptr, n_floats):
        self._check(self._L.halo_bind_accumulator(self._h, C.c_void_p(device_ptr), int(n_floats)))

    def sync(self):
        self._check(self._L.halo_sync(self._h))

    def collect_timing(self):
        """(trace_ms, post_ms, launches) of the sessions' LAST layers since the previous call: the trace kernels' own spans (HIP events on the
        launch's stream) and those of their accumulation passes."""
        t, p, n = C.c_double(), C.c_double(), C.c_uint64()
        self._check(self._L.halo_collect_timing(self._h, C.byref(t), C.byref(p), C.byref(n)))
        return t.value, p.value, n.value

    def flush(self):
        """Queue the pending closing folds and make the backend's stream wait for them (no host wait): what is queued on that stream afterwards
        sees the finished image — the call a holder of a BOUND accumulator makes before it touches its memory when option defer_fold is on."""
        self._check(self._L.halo_flush(self._h))

    def generate_shapes(self, crystal, first_index, n, on_device=True):
        """`n` sampled instances of `crystal` as HaloGeomTables (device generator or the host builder)."""
        out = (abi.HaloGeomTables * n)()
        self._check(self._L.halo_generate_shapes(self._h, C.byref(crystal), int(first_index), int(n), int(on_device), out))   # 0 host | 1 general records | 2 prism records
        return out

    def last_sample_counts(self):
        """(stochastic crystal instances sampled, rays whose orientation was drawn) in the session traced last."""
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self._L.halo_last_sample_counts(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_route(self):
        """HaloRouteInfo of the session traced last: which kernel instantiations and accumulation routes really ran."""
        r = abi.HaloRouteInfo()
        self._check(self._L.halo_last_route(self._h, C.byref(r)))
        return r

    def collect_stats(self):
        """Summed tallies of every layer traced since the previous call (waits for the stream); the only source of
        exit / pixel-hit / kernel-time numbers for layers traced with option async=1."""
        st = abi.HaloLayerStats()
        self._check(self._L.halo_collect_stats(self._h, C.byref(st)))
        return st

    def take_landed(self):
        v = C.c_double()
        self._check(self._L.halo_take_landed(self._h, C.byref(v)))
        return v.value

    # --- the seam (trace_backend.hpp:374-632) ---
    def SupportsDeviceXyzAccum(self):
        return True

    def SupportsThirdClockDrain(self):
        return True

    def BeginSession(self, scene, render, wl, ray_num=0):
        self._render = render
        self._scene = scene
        self._lanes_shape = None   # the session's lanes take the render's size
        self._check(self._L.halo_begin(self._h, C.byref(scene), C.byref(render), C.byref(wl), int(ray_num)))

    def TraceLayer(self, count=0, host_rays=None):
        """First layer: `count` self-generated roots, or injected crystal-local rays (d, p, w, tf arrays).
        Later layers: consumes the continuation from Recombine.  Returns HaloLayerStats."""
        stats = abi.HaloLayerStats()
        if host_rays is None:
            self._check(self._L.halo_trace_layer(self._h, int(count), None, C.byref(stats)))
            self._pending_roots += int(stats.root_count)
            return stats
        d, p, w, tf = (np.ascontiguousarray(host_rays[0], np.float32), np.ascontiguousarray(host_rays[1], np.float32),
                       np.ascontiguousarray(host_rays[2], np.float32), np.ascontiguousarray(host_rays[3], np.uint32))
        n = w.shape[0]
        crystal = host_rays[4] if len(host_rays) > 4 else None      # HostRayBatch::crystal: an abi.HaloGeomTables, or None = the entry's own
        hr = abi.HaloHostRays(d.ctypes.data_as(C.POINTER(C.c_float)), p.ctypes.data_as(C.POINTER(C.c_float)),
                              w.ctypes.data_as(C.POINTER(C.c_float)), tf.ctypes.data_as(C.POINTER(C.c_uint32)),
                              C.cast(C.pointer(crystal), C.c_void_p) if crystal is not None else None)
        self._check(self._L.halo_trace_layer(self._h, n, C.byref(hr), C.byref(stats)))
        self._pending_roots += int(stats.root_count)
        return stats

    def Recombine(self, shuffle=True):
        n = C.c_uint64()
        self._check(self._L.halo_recombine(self._h, 1 if shuffle else 0, C.byref(n)))
        return n.value

    def DrainExits(self, max_records=None):
        """Captured exit records since the last drain (only with option capture_exits=1); destructive.  With max_records the
        drain is piecewise: at most that many records now, the rest stays pending (halo_drain_exits)."""
        n = C.c_uint64()
        self._check(self._L.halo_drain_exits(self._h, None, 0, C.byref(n)))   # pending count, nothing consumed
        take = n.value if max_records is None else min(n.value, int(max_records))
        buf = (abi.HaloExitRecord * max(1, int(take)))()
        if take:
            self._check(self._L.halo_drain_exits(self._h, buf, int(take), C.byref(n)))
            take = n.value
        self._pending_roots = 0
        return np.frombuffer(buf, dtype=EXIT_DTYPE, count=int(take)).copy()

    def ReadbackXyzAccum(self, width=None, height=None):
        """Returns (xyz[H,W,3] float32, landed_weight float64) and zeroes the device accumulator."""
        w, h = (width or self._render.width), (height or self._render.height)
        img = np.empty((h, w, 3), np.float32)
        landed = C.c_double()
        self._check(self._L.halo_readback_xyz64(self._h, img.ctypes.data_as(C.POINTER(C.c_float)), w, h, C.byref(landed)))
        return img, landed.value

    def EndSession(self):
        self._check(self._L.halo_end(self._h))

    # --- consumer on device (RenderConsumer, reference src/server/render.cpp) ---
    def ConsumeDeviceFused(self):
        """Fold the device accumulator into the Neumaier running image and zero it (render.cpp:138-201)."""
        self._check(self._L.halo_consumer_fold(self._h))

    def Consume(self, xyz, landed, lanes=None):
        """RenderConsumer::ConsumeDeviceFused(const SimData&) for a drained image the caller holds: xyz float32[H,W,3] is Neumaier-folded
        into the running image, `landed` added to the total intensity, lanes float32[classes,H,W] (optional) added to the class lanes."""
        xyz = np.ascontiguousarray(xyz, np.float32)
        h, w = xyz.shape[0], xyz.shape[1]
        lp, nc = None, 0
        if lanes is not None:
            lanes = np.ascontiguousarray(lanes, np.float32)
            lp, nc = lanes.ctypes.data_as(C.POINTER(C.c_float)), lanes.shape[0]
        self._check(self._L.halo_consumer_consume(self._h, xyz.ctypes.data_as(C.POINTER(C.c_float)), w, h, float(landed), lp, nc))
        self._cons_size = (w, h)

    def Snapshot(self, intensity_factor=1.0, ray_color=(-1.0, -1.0, -1.0), background=(0.0, 0.0, 0.0), want_xyz=True):
        """PrepareSnapshot + PostSnapshot: returns (rgb uint8[H,W,3], xyz float32[H,W,3] | None, total_intensity)."""
        w, h = getattr(self, "_cons_size", None) or (self._render.width, self._render.height)
        d = abi.HaloDisplay(float(intensity_factor), (C.c_float * 3)(*ray_color), (C.c_float * 3)(*background))
        rgb = np.empty((h, w, 3), np.uint8)
        xyz = np.empty((h, w, 3), np.float32) if want_xyz else None
        tot = C.c_double()
        self._check(self._L.halo_consumer_snapshot(self._h, C.byref(d), rgb.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                   xyz.ctypes.data_as(C.POINTER(C.c_float)) if want_xyz else None, C.byref(tot)))
        return rgb, xyz, tot.value

    def ResetConsumer(self):
        self._check(self._L.halo_consumer_reset(self._h))

    # --- display-side composite of the class lanes (reference src/server/component_compositor.cpp) ---
    def LoadClassLanes(self, lanes, total_intensity=-1.0):
        """Replace the device lanes with (class_count, H, W) float32 host values (lanes summed over ranks, a saved consumer);
        total_intensity >= 0 replaces the consumer's total intensity too."""
        lanes = np.ascontiguousarray(lanes, np.float32)
        n, h, w = lanes.shape
        self._check(self._L.halo_consumer_load_lanes(self._h, lanes.ctypes.data_as(C.POINTER(C.c_float)), w, h, n, float(total_intensity)))
        self._lanes_shape = (n, h, w)

    def CompositeColorClasses(self, classes, mode="painter", display_exposure_scale=1.0, intensity_factor=1.0, want_srgb=True):
        """CompositeColorClassesLinear + LinearRgbToSrgbU8 on the device lanes (which stay).  Returns (produced, linear_rgb[H,W,3]
        float32, srgb[H,W,3] uint8 | None, participating_p99_y)."""
        n, h, w = getattr(self, "_lanes_shape", None) or (getattr(self, "_n_classes", 0), self._render.height, self._render.width)
        spec = abi.composite(classes, mode, display_exposure_scale, intensity_factor)
        lin = np.zeros((h, w, 3), np.float32)
        srgb = np.zeros((h, w, 3), np.uint8) if want_srgb else None
        p99, produced = C.c_float(-1.0), C.c_int32(0)
        self._check(self._L.halo_consumer_composite(self._h, C.byref(spec), lin.ctypes.data_as(C.POINTER(C.c_float)),
                                                    srgb.ctypes.data_as(C.POINTER(C.c_uint8)) if want_srgb else None, C.byref(p99), C.byref(produced)))
        return bool(produced.value), lin, srgb, p99.value


EXIT_DTYPE = np.dtype([("dir", np.float32, 3), ("weight", np.float32), ("root", np.uint32), ("seq", np.uint16),
                       ("layer", np.uint8), ("path_len", np.uint8), ("path", np.uint8, abi.PATH_CAP),
                       ("pixel", np.int32), ("crystal_id", np.uint16), ("wl_idx", np.uint16), ("color_mask", np.uint64)])
assert EXIT_DTYPE.itemsize == C.sizeof(abi.HaloExitRecord)
