"""Which generated crystal differs between the device team generator, the device serial builder and the host builder, and where."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ice_halo_sim_amd import scenes, abi
from ice_halo_sim_amd.backend import HipTraceBackend

u = lambda m, s: {"type": "uniform", "mean": m, "std": s}
cr = scenes.pyramid_crystal(1.0, u(1.0, 0.8), 1.0, face_distance=[u(1.0, 0.1)] * 6)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
hb = HipTraceBackend(seed=1234)
dev = hb.generate_shapes(cr, 10_000_000_000, n, on_device=True)
host = hb.generate_shapes(cr, 10_000_000_000, n, on_device=False)
hb.set_option("gen_serial", 1)
ser = hb.generate_shapes(cr, 10_000_000_000, n, on_device=True)
fields = [("face_n", 3, "face_cnt"), ("face_d", 1, "face_cnt"), ("face_number", 1, "face_cnt"), ("tri_v", 9, "tri_cnt"), ("tri_n", 3, "tri_cnt"), ("tri_area", 1, "tri_cnt"), ("tri_face", 1, "tri_cnt")]
for k in range(n):
    for tag, a, b in (("team-vs-host", dev[k], host[k]), ("serial-vs-host", ser[k], host[k])):
        diffs = []
        if (a.face_cnt, a.tri_cnt) != (b.face_cnt, b.tri_cnt):
            diffs.append("counts %s vs %s" % ((a.face_cnt, a.tri_cnt), (b.face_cnt, b.tri_cnt)))
        else:
            for name, w, cntf in fields:
                if not hasattr(a, name):
                    continue
                c = getattr(a, cntf) * w
                x = np.array(getattr(a, name)[:c]); y = np.array(getattr(b, name)[:c])
                if x.tobytes() != y.tobytes():
                    idx = np.nonzero(x != y)[0]
                    diffs.append("%s: %d entries differ, first %d: %r vs %r (max abs %g)" % (name, len(idx), idx[0], x[idx[0]], y[idx[0]], float(np.max(np.abs(x.astype(np.float64) - y.astype(np.float64))))))
        if diffs:
            print(k, tag, "; ".join(diffs))
hb.close()
