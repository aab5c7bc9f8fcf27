"""One seed of tests/test_gpu_fuzz.py in detail: what the scene is and how the unmatched exits differ (missing on one side, other direction, other weight)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ice_halo_sim_amd import abi
from tests._oracle_backend import OracleBackend, run_session
from tests.test_gpu_fuzz import make_case
from tests.test_gpu_parity import hip_backend

DN = ["none", "uniform", "gauss", "zigzag", "laplacian", "gauss_legacy"]
def d(x): return "%s(%.3g,%.3g)" % (DN[x.type], x.center, x.spread)
for seed in [int(a) for a in sys.argv[1:]]:
    sc, rd, wl, filters, clock = make_case(seed)
    print("=== seed", seed, "max_hits", sc.max_hits, "sun", sc.sun_altitude, sc.sun_azimuth, sc.sun_diameter, "lens", rd.lens_type, rd.width, rd.height, "fov", rd.fov, "vis", rd.visible, "wl", wl.wavelength, wl.illuminant, wl.pool_size, "clock", clock)
    L = sc.layers[0]
    for i in range(L.entry_count):
        e = L.entries[i]; c = e.crystal
        print("  entry", i, "kind", c.kind, "h", [d(c.height[k]) for k in range(3)], "fd", [d(c.face_dist[k]) for k in range(6)], "wedge", c.wedge_upper_deg, c.wedge_lower_deg, "sync", list(c.sync_group))
        print("     axis lat", d(e.axis.latitude), "az", d(e.axis.azimuth), "roll", d(e.axis.roll), "prop", e.proportion, "filter", e.filter_id)
    for k, f in enumerate(filters):
        print("  filter", k + 1, "action", f.action, "sym", f.symmetry, "complex", f.is_complex, "or", f.or_count, [f.terms[j].type for j in range(4)])
    hb = hip_backend(seed=seed, capture_exits=1, geom_clock=clock); ob = OracleBackend(seed=seed, capture_exits=1, threads=8, geom_clock=clock)
    for b in (hb, ob): b.set_filters(filters)
    run_session(hb, sc, rd, wl, 60_000); run_session(ob, sc, rd, wl, 60_000)
    eh, eo = hb.DrainExits(), ob.DrainExits(); hb.close(); ob.close()
    kh = (eh["root"].astype(np.int64) << 8) | eh["seq"].astype(np.int64); ko = (eo["root"].astype(np.int64) << 8) | eo["seq"].astype(np.int64)
    ih, io = np.argsort(kh), np.argsort(ko)
    common, ch, co = np.intersect1d(kh[ih], ko[io], return_indices=True)
    a, b = eh[ih][ch], eo[io][co]
    dd = np.abs(a["dir"] - b["dir"]).max(axis=1); dw = np.abs(a["weight"] - b["weight"]) / np.maximum(np.abs(b["weight"]), 1e-12)
    bad = (dd > 2e-5) | ((dw > 2e-4) & (np.abs(a["weight"] - b["weight"]) >= 1e-9))
    only_h = np.setdiff1d(kh, ko); only_o = np.setdiff1d(ko, kh)
    print("  exits hip %d oracle %d | only hip %d (roots %d) only oracle %d (roots %d) | common differing %d (roots %d): dir>2e-5 %d, weight %d" % (
        len(eh), len(eo), len(only_h), len(np.unique(only_h >> 8)), len(only_o), len(np.unique(only_o >> 8)), bad.sum(), len(np.unique(a["root"][bad])), (dd > 2e-5).sum(), ((dw > 2e-4)).sum()))
    if bad.any():
        print("  differing pairs: |dw|/w percentiles 50/90/99/max %s | dir diff 50/99/max %s | their weights median %.3g" % (
            np.round(np.percentile(dw[bad], [50, 90, 99, 100]), 6), np.round(np.percentile(dd[bad], [50, 99, 100]), 7), np.median(b["weight"][bad])))
    # exits that pair up but land elsewhere: another pixel, or inside the frame on one side only (what the landed weights may differ by)
    ph, po = a["pixel"].astype(np.int64), b["pixel"].astype(np.int64)
    moved = ph != po
    one_side = moved & ((ph < 0) | (po < 0))
    print("  paired exits in another pixel: %d, of them in the frame on one side only: %d (weights %s; direction differences %s)" % (
        moved.sum(), one_side.sum(), np.round(b["weight"][one_side][:6], 5), np.abs(a["dir"] - b["dir"]).max(axis=1)[one_side][:6]))
    for k in np.nonzero(moved)[0][:8]:
        print("     moved: root %d seq %d len %d pixel hip %d (row %d col %d) oracle %d (row %d col %d) dir hip %s oracle %s w %.5g" % (
            a["root"][k], a["seq"][k], a["path_len"][k], ph[k], ph[k] // rd.width, ph[k] % rd.width, po[k], po[k] // rd.width, po[k] % rd.width,
            np.array2string(a["dir"][k], precision=8), np.array2string(b["dir"][k], precision=8), b["weight"][k]))
    bad_roots = np.unique(np.concatenate([only_h >> 8, only_o >> 8, a["root"][bad].astype(np.int64)]))
    print("  rays with any difference: %d of %d (%.3f %%)" % (len(bad_roots), 60000, 100.0 * len(bad_roots) / 60000))
    # which crystal entry the differing rays belong to (rays are dealt out to the entries in order, PartitionCrystalRayNum)
    import ctypes as C
    from ice_halo_sim_amd import backend
    from tests._libs import fptr
    props = np.array([L.entries[i].proportion for i in range(L.entry_count)], np.float32)
    carry = np.zeros(L.entry_count); cnt = (C.c_uint64 * L.entry_count)()
    backend.load_library().halo_host_partition(fptr(props), L.entry_count, 60000, carry.ctypes.data_as(C.POINTER(C.c_double)), cnt)
    edges = np.concatenate([[0], np.cumsum(list(cnt))])
    print("  entry ranges", list(edges), "differing rays per entry", [int(((bad_roots >= edges[i]) & (bad_roots < edges[i + 1])).sum()) for i in range(L.entry_count)],
          "| only-hip roots per entry", [int((((only_h >> 8) >= edges[i]) & ((only_h >> 8) < edges[i + 1])).sum()) for i in range(L.entry_count)],
          "| only-oracle", [int((((only_o >> 8) >= edges[i]) & ((only_o >> 8) < edges[i + 1])).sum()) for i in range(L.entry_count)])
    show = list(bad_roots[:2]) + list(np.unique(only_h >> 8)[:2]) + list(np.unique(only_o >> 8)[:2])
    for r in show:
        print("   root", r)
        for tag, e in (("hip", eh), ("ora", eo)):
            m = e[e["root"] == r]
            for x in m[np.argsort(m["seq"])][:8]:
                print("      %s seq %2d len %d path %s dir (%.5f %.5f %.5f) w %.5g" % (tag, x["seq"], x["path_len"], list(x["path"][:x["path_len"]]), x["dir"][0], x["dir"][1], x["dir"][2], x["weight"]))
