#!/usr/bin/env python3
"""Generate tests/golden/ref_shared_fixture.npz from the REFERENCE itself (build container only).

Inputs are seeded random draws plus the reference's own fixed sample pool
(test/golden-analytic/core/closed_form_samples_generated.hpp: kPrismWellConditionedSamples — data, parsed here);
expected outputs come from oracle/_ref/libref_shared.so, i.e. the reference's src/core/shared/*.h,
src/core/color_util.hpp and test/support/exact_prism_oracle.hpp compiled where they lie.  Only inputs and outputs
are stored.  The fixture lets the oracle be re-pinned on machines without /root/reference (the GPU box).
"""
import ctypes as C
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import _libs  # noqa: E402
from tests._libs import fptr, i32ptr, u32ptr  # noqa: E402

REF_SAMPLES = "/root/reference/test/golden-analytic/core/closed_form_samples_generated.hpp"


def parse_prism_pool():
    txt = open(REF_SAMPLES).read()
    m = re.search(r"kPrismWellConditionedSamples\[\]\s*=\s*\{(.*?)\n\};", txt, re.S)
    rows = re.findall(r"\{([^{}]*)\}", m.group(1))
    out = []
    for r in rows:
        vals = [float.fromhex(t.strip()) for t in r.split(",") if t.strip()]
        if len(vals) == 6:
            out.append(vals)
    return np.asarray(out, np.float32)


def main():
    R = _libs.ref()
    rng = np.random.default_rng(415)
    fx = {}
    # --- pcg hash / uniform streams ---
    x = rng.integers(0, 2**32, 512, dtype=np.uint32)
    fx["hash_in"] = x
    fx["hash_out"] = np.array([R.ref_pcg_hash(int(v)) for v in x], np.uint32)
    seeds = rng.integers(0, 2**32, (128, 2), dtype=np.uint32)
    uni = np.zeros((128, 8), np.float32)
    gau = np.zeros(128, np.float32)
    for i, (s, g) in enumerate(seeds):
        st = np.array([s, g, 0], np.uint32)
        for k in range(8):
            uni[i, k] = R.ref_pcg_uniform(u32ptr(st))
        gau[i] = R.ref_pcg_gaussian(u32ptr(st))
    fx["stream_seeds"], fx["stream_uniform8"], fx["stream_gauss_after8"] = seeds, uni, gau
    hi = rng.integers(0, 2**32, (256, 4), dtype=np.uint32)
    hi[::3, 1] = 0
    hi[::5, 0] = 0xFFFFFFFF - rng.integers(0, 100, len(hi[::5]))
    hi[::5, 2] = rng.integers(0, 300, len(hi[::5]))
    fx["hi_in"] = hi
    fx["hi_adv"] = np.array([R.ref_pcg_advance_hi(int(a), int(b), int(c)) for a, b, c, _ in hi], np.uint32)
    fx["hi_seed"] = np.array([R.ref_pcg_seed_with_high(int(d), int(b)) for _, b, _, d in hi], np.uint32)
    # --- get_dist over all six types ---
    gd_in = np.zeros((600, 5), np.float32)
    gd_out = np.zeros(600, np.float32)
    gd_seed = rng.integers(0, 2**32, (600, 2), dtype=np.uint32)
    for i in range(600):
        dtype, mean, std = i % 6, float(rng.normal()), float(abs(rng.normal()) + 0.01)
        st = np.array([gd_seed[i, 0], gd_seed[i, 1], 0], np.uint32)
        gd_out[i] = R.ref_pcg_get_dist(u32ptr(st), dtype, mean, std)
        gd_in[i] = [dtype, mean, std, st[2], 0]
    fx["getdist_seed"], fx["getdist_in"], fx["getdist_out"] = gd_seed, gd_in, gd_out
    # --- rotations ---
    ang = rng.uniform(-7, 7, (256, 3)).astype(np.float32)
    mats = np.zeros((256, 9), np.float32)
    inv = np.zeros((256, 3), np.float32)
    vec = rng.normal(size=(256, 3)).astype(np.float32)
    for i in range(256):
        R.ref_build_crystal_rotation_9(float(ang[i, 0]), float(ang[i, 1]), float(ang[i, 2]), fptr(mats[i]))
        R.ref_apply_inverse_mat9(fptr(mats[i]), fptr(vec[i]), fptr(inv[i]))
    fx["rot_angles"], fx["rot_mat9"], fx["rot_vec"], fx["rot_inv"] = ang, mats, vec, inv
    # --- triangle / cap / categorical / feistel ---
    tri_v = rng.normal(size=(256, 9)).astype(np.float32)
    tri_p = np.zeros((256, 3), np.float32)
    cap_in = np.stack([rng.uniform(0, 6.3, 256), rng.uniform(-1.5, 1.5, 256), rng.uniform(0, 0.05, 256)], 1).astype(np.float32)
    cap_d = np.zeros((256, 3), np.float32)
    tc_seed = rng.integers(0, 2**32, (256, 2), dtype=np.uint32)
    for i in range(256):
        st = np.array([tc_seed[i, 0], tc_seed[i, 1], 0], np.uint32)
        R.ref_sample_triangle(u32ptr(st), fptr(tri_v[i]), fptr(tri_p[i]))
        R.ref_sample_sph_cap(u32ptr(st), float(cap_in[i, 0]), float(cap_in[i, 1]), float(cap_in[i, 2]), fptr(cap_d[i]))
    fx["tc_seed"], fx["tri_v"], fx["tri_p"], fx["cap_in"], fx["cap_d"] = tc_seed, tri_v, tri_p, cap_in, cap_d
    cat_w = rng.normal(size=(256, 20)).astype(np.float32)
    cat_w[::9] = -np.abs(cat_w[::9])
    cat_u = rng.random(256).astype(np.float32)
    fx["cat_w"], fx["cat_u"] = cat_w, cat_u
    fx["cat_out"] = np.array([R.ref_categorical_sample(fptr(cat_w[i]), 20, float(cat_u[i])) for i in range(256)], np.uint32)
    fe = []
    for n in (1, 2, 3, 5, 16, 17, 1000, 4097, 100003, 47_000_000):
        seed = int(rng.integers(0, 2**32, dtype=np.uint32))
        for i in rng.integers(0, n, 24):
            fe.append((int(i), n, seed, R.ref_feistel_bijection(int(i), n, seed)))
    fx["feistel"] = np.asarray(fe, np.uint32)
    # --- fresnel / slab ---
    fr_in = np.stack([rng.uniform(0, 3, 512), rng.choice([1.31, 1 / 1.31, 1.3, 0.75], 512)], 1).astype(np.float32)
    fx["fresnel_in"] = fr_in
    fx["fresnel_out"] = np.array([R.ref_reflect_ratio(float(a), float(b)) for a, b in fr_in], np.float32)
    sl = rng.normal(size=(512, 10)).astype(np.float32)
    fx["slab_in"] = sl
    fx["slab_out"] = np.array([R.ref_slab_face_t(fptr(np.ascontiguousarray(r[0:3])), fptr(np.ascontiguousarray(r[3:6])),
                                                 fptr(np.ascontiguousarray(r[6:9])), float(r[9])) for r in sl], np.float32)
    # --- LUT inversion on a synthetic monotone table ---
    th = np.linspace(0.3, 2.9, 257).astype(np.float32)
    cdf = np.sort(rng.random(257)).astype(np.float32)
    cdf[0], cdf[-1] = 0.0, 1.0
    xi = np.concatenate([rng.random(300), [0.0, 1.0]]).astype(np.float32)
    fx["lut_theta"], fx["lut_cdf"], fx["lut_xi"] = th, cdf, xi
    fx["lut_inv"] = np.array([R.ref_invert_lat_lut(float(v), fptr(th), fptr(cdf), 257) for v in xi], np.float32)
    fx["lut_bin"] = np.array([R.ref_lat_lut_bin(float(v), fptr(th), 257) for v in fx["lut_inv"]], np.uint32)
    nl = rng.uniform(-20, 20, 256).astype(np.float32)
    nlo = np.zeros((256, 2), np.float32)
    for i, v in enumerate(nl):
        a, f = C.c_float(), C.c_int()
        R.ref_normalize_latitude(float(v), C.byref(a), C.byref(f))
        nlo[i] = [a.value, f.value]
    fx["normlat_in"], fx["normlat_out"] = nl, nlo
    # --- projection: ProjParams (76-byte POD rows, produced by the oracle's BuildProjParams) x directions ---
    from ice_halo_sim_amd import abi, scenes
    O = _libs.oracle()
    dirs = rng.normal(size=(96, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    pps, outs = [], []
    for lens in range(11):
        for k in range(3):
            cfg = scenes.render(lens, [512, 1920, 2048][k], [256, 1080, 1024][k], fov=[60.0, 120.0, 180.0][k] if lens != 10 else 60.0,
                                az=float(rng.uniform(-180, 180)), el=float(rng.uniform(-60, 80)), ro=float(rng.uniform(-30, 30)),
                                visible=k, overlap=0.0872 if (lens in (4, 5, 6) and k == 2) else 0.0, lens_shift=(k * 7, -k * 5))
            pp = abi.ProjParams()
            O.ho_build_proj_params(C.byref(cfg), C.byref(pp))
            raw = np.frombuffer(bytes(pp), np.uint32).copy()
            res = np.zeros((len(dirs), 7), np.int32)
            for j, d in enumerate(dirs):
                R.ref_project_exit_to_pixel(raw.ctypes.data, float(d[0]), float(d[1]), float(d[2]), i32ptr(res[j]))
            pps.append(raw)
            outs.append(res)
    fx["proj_params"], fx["proj_dirs"], fx["proj_out"] = np.asarray(pps), dirs, np.asarray(outs)
    # --- CMF via SpectrumToXyz ---
    wls = np.arange(350, 841, 1.0, dtype=np.float32)
    cm = np.zeros((len(wls), 3), np.float32)
    for i, w in enumerate(wls):
        R.ref_spectrum_to_xyz(float(w), 1.0, fptr(cm[i]))
    fx["cmf_wl"], fx["cmf_xyz"] = wls, cm
    # --- prism pool from the reference's own generated samples + exact-integer verdicts ---
    pool = parse_prism_pool()
    verdict = np.zeros((len(pool), 8), np.int32)
    for i, d in enumerate(pool):
        R.ref_exact_prism(fptr(np.ascontiguousarray(d)), i32ptr(verdict[i]))
    fx["prism_pool_dist"], fx["prism_pool_exact"] = pool, verdict
    out = os.path.join(ROOT, "tests", "golden", "ref_shared_fixture.npz")
    np.savez_compressed(out, **fx)
    print("wrote", out, os.path.getsize(out), "bytes;", len(pool), "prism pool rows")


if __name__ == "__main__":
    main()
