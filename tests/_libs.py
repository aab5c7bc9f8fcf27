"""Loaders for the three native libraries the tests touch.

  oracle()  — oracle/liboracle.so      CPU restatement (the checker)
  ref()     — oracle/_ref/libref_shared.so   the reference's own shared-math headers (build container only)
  product() — ice_halo_sim_amd/libhalo_hip.so   the thing under test (through the C ABI)
"""
import ctypes as C
import os
import subprocess

import numpy as np

from ice_halo_sim_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
ORACLE_FMA_SO = os.path.join(ROOT, "oracle", "liboracle_fma.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_shared.so")

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)


class HoStream(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("global_idx", C.c_uint32), ("slot", C.c_uint32)]


class HoGenParams(C.Structure):
    _fields_ = [("lat_path", C.c_uint32), ("lat_mean_rad", C.c_float), ("lat_std_rad", C.c_float),
                ("lat_lut_n", C.c_uint32), ("az_type", C.c_uint32), ("az_mean_rad", C.c_float),
                ("az_std_rad", C.c_float), ("roll_type", C.c_uint32), ("roll_mean_rad", C.c_float),
                ("roll_std_rad", C.c_float)]


class HoPixelHit(C.Structure):
    _fields_ = [("px", C.c_int32), ("py", C.c_int32), ("bump_landed", C.c_int32)]


class HoProjResult(C.Structure):
    _fields_ = [("hits", HoPixelHit * 2), ("count", C.c_int32)]


_oracle = {}


def oracle(fma=False):
    """fma=True: the same source compiled with products contracted into FMAs (oracle/Makefile: liboracle_fma.so) — the second
    legitimate rounding of the reference's expressions, the tests' yardstick for ill-conditioned exits"""
    if fma in _oracle:
        return _oracle[fma]
    so = ORACLE_FMA_SO if fma else ORACLE_SO
    src = os.path.join(ROOT, "oracle", "halo_oracle.c")
    deps = [src, os.path.join(ROOT, "oracle", "halo_oracle.h"), os.path.join(ROOT, "include", "halo_trace.h")]
    if (not os.path.exists(so)) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps if os.path.exists(d)):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), so])
    L = C.CDLL(so)
    sp = C.POINTER(HoStream)
    L.ho_pcg_hash.restype = C.c_uint32; L.ho_pcg_hash.argtypes = [C.c_uint32]
    L.ho_u01_from_hash.restype = C.c_float; L.ho_u01_from_hash.argtypes = [C.c_uint32]
    L.ho_pcg_advance_hi.restype = C.c_uint32; L.ho_pcg_advance_hi.argtypes = [C.c_uint32] * 3
    L.ho_pcg_seed_with_high.restype = C.c_uint32; L.ho_pcg_seed_with_high.argtypes = [C.c_uint32] * 2
    L.ho_pcg_uniform.restype = C.c_float; L.ho_pcg_uniform.argtypes = [sp]
    L.ho_pcg_gaussian.restype = C.c_float; L.ho_pcg_gaussian.argtypes = [sp]
    L.ho_pcg_get_dist.restype = C.c_float; L.ho_pcg_get_dist.argtypes = [sp, C.c_uint32, C.c_float, C.c_float]
    L.ho_normalize_latitude.restype = None; L.ho_normalize_latitude.argtypes = [C.c_float, f32p, i32p]
    L.ho_invert_lat_lut.restype = C.c_float; L.ho_invert_lat_lut.argtypes = [C.c_float, f32p, f32p, C.c_uint32]
    L.ho_lat_lut_bin.restype = C.c_uint32; L.ho_lat_lut_bin.argtypes = [C.c_float, f32p, C.c_uint32]
    L.ho_sample_lat_lon_roll.restype = None
    L.ho_sample_lat_lon_roll.argtypes = [sp, C.POINTER(HoGenParams), f32p, f32p, f32p, f32p, f32p, f32p]
    L.ho_build_crystal_rotation_9.restype = None; L.ho_build_crystal_rotation_9.argtypes = [C.c_float] * 3 + [f32p]
    L.ho_apply_inverse_mat9.restype = None; L.ho_apply_inverse_mat9.argtypes = [f32p, f32p, f32p]
    L.ho_sample_triangle.restype = None; L.ho_sample_triangle.argtypes = [sp, f32p, f32p]
    L.ho_sample_sph_cap.restype = None; L.ho_sample_sph_cap.argtypes = [sp, C.c_float, C.c_float, C.c_float, f32p]
    L.ho_feistel_bijection.restype = C.c_uint32; L.ho_feistel_bijection.argtypes = [C.c_uint32] * 3
    L.ho_categorical_sample.restype = C.c_uint32; L.ho_categorical_sample.argtypes = [f32p, C.c_uint32, C.c_float]
    L.ho_reflect_ratio.restype = C.c_float; L.ho_reflect_ratio.argtypes = [C.c_float, C.c_float]
    L.ho_slab_face_t.restype = C.c_float; L.ho_slab_face_t.argtypes = [f32p, f32p, f32p, C.c_float]
    L.ho_ice_refractive_index.restype = C.c_double; L.ho_ice_refractive_index.argtypes = [C.c_double]
    L.ho_build_proj_params.restype = None; L.ho_build_proj_params.argtypes = [C.POINTER(abi.HaloRender), C.POINTER(abi.ProjParams)]
    L.ho_project_exit_to_pixel.restype = HoProjResult
    L.ho_project_exit_to_pixel.argtypes = [C.POINTER(abi.ProjParams), C.c_float, C.c_float, C.c_float]
    L.ho_build_lat_lut.restype = None; L.ho_build_lat_lut.argtypes = [C.POINTER(abi.HaloDist), f32p, f32p, f32p]
    L.ho_select_lat_path.restype = C.c_uint32; L.ho_select_lat_path.argtypes = [C.POINTER(abi.HaloAxis)]
    L.ho_prism_geometry.restype = None; L.ho_prism_geometry.argtypes = [C.c_float, f32p, C.POINTER(abi.HaloGeomTables)]
    L.ho_prism_corner_ring.restype = C.c_int; L.ho_prism_corner_ring.argtypes = [C.c_float, f32p, f32p, f32p, i32p]
    L.ho_partition.restype = None
    L.ho_partition.argtypes = [f32p, C.c_int, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.ho_illuminant_spd.restype = C.c_float; L.ho_illuminant_spd.argtypes = [C.c_int, C.c_float]
    L.ho_cmf.restype = None; L.ho_cmf.argtypes = [C.c_float, f32p, f32p, f32p]
    L.ho_create.restype = C.c_void_p; L.ho_create.argtypes = [C.c_uint32]
    L.ho_destroy.restype = None; L.ho_destroy.argtypes = [C.c_void_p]
    L.ho_set_option.restype = C.c_int; L.ho_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.ho_begin.restype = C.c_int
    L.ho_begin.argtypes = [C.c_void_p, C.POINTER(abi.HaloScene), C.POINTER(abi.HaloRender), C.POINTER(abi.HaloWl), C.c_uint64]
    L.ho_trace_layer.restype = C.c_int
    L.ho_trace_layer.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(abi.HaloHostRays), C.POINTER(abi.HaloLayerStats)]
    L.ho_recombine.restype = C.c_int; L.ho_recombine.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
    L.ho_drain_exits.restype = C.c_int
    L.ho_drain_exits.argtypes = [C.c_void_p, C.POINTER(abi.HaloExitRecord), C.c_uint64, C.POINTER(C.c_uint64)]
    L.ho_end.restype = C.c_int; L.ho_end.argtypes = [C.c_void_p]
    L.ho_readback_xyz64.restype = C.c_int
    L.ho_readback_xyz64.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.ho_continuation_dump.restype = C.c_uint64; L.ho_continuation_dump.argtypes = [C.c_void_p, f32p, C.c_uint64]
    L.ho_pyramid_geometry.restype = None; L.ho_pyramid_geometry.argtypes = [C.c_float] * 5 + [f32p, C.POINTER(abi.HaloGeomTables)]
    L.ho_pyramid_face_mask.restype = C.c_int; L.ho_pyramid_face_mask.argtypes = [C.c_float] * 5 + [f32p, i32p]
    u8p = C.POINTER(C.c_uint8)
    L.ho_reduce_raypath.restype = None; L.ho_reduce_raypath.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
    L.ho_compute_sigma_a.restype = C.c_int; L.ho_compute_sigma_a.argtypes = [C.c_float]
    L.ho_is_d_applicable.restype = C.c_int; L.ho_is_d_applicable.argtypes = [C.POINTER(abi.HaloAxis)]
    L.ho_filter_check.restype = C.c_int
    L.ho_filter_check.argtypes = [C.POINTER(abi.HaloFilter), C.POINTER(abi.HaloAxis), u8p, C.c_int, f32p, C.c_int]
    L.ho_color_mask.restype = C.c_uint64
    L.ho_color_mask.argtypes = [C.POINTER(abi.HaloColorSet), C.POINTER(abi.HaloAxis), u8p, C.c_int, f32p, C.c_int, C.c_uint64]
    L.ho_shape_scalars.restype = None; L.ho_shape_scalars.argtypes = [C.POINTER(abi.HaloCrystal), C.c_uint32, C.c_uint64, f32p]
    L.ho_set_filters.restype = C.c_int; L.ho_set_filters.argtypes = [C.c_void_p, C.POINTER(abi.HaloFilter), C.c_int32]
    L.ho_set_color.restype = C.c_int; L.ho_set_color.argtypes = [C.c_void_p, C.POINTER(abi.HaloColorSet), C.c_int32, C.POINTER(abi.HaloColorClass), C.c_int32]
    L.ho_readback_class_lanes.restype = C.c_int; L.ho_readback_class_lanes.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int]
    L.ho_neumaier_add.restype = None; L.ho_neumaier_add.argtypes = [f32p, f32p, C.c_float]
    L.ho_gamut_clip_xyz.restype = None; L.ho_gamut_clip_xyz.argtypes = [f32p, f32p]
    L.ho_xyz_to_linear_rgb.restype = None; L.ho_xyz_to_linear_rgb.argtypes = [f32p, f32p]
    L.ho_linear_to_srgb.restype = C.c_float; L.ho_linear_to_srgb.argtypes = [C.c_float]
    L.ho_consumer_fold.restype = C.c_int; L.ho_consumer_fold.argtypes = [C.c_void_p]
    L.ho_consumer_consume.restype = C.c_int; L.ho_consumer_consume.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int, C.c_float, f32p, C.c_int]
    L.ho_participating_exposure_scale.restype = C.c_float; L.ho_participating_exposure_scale.argtypes = [C.c_float] * 3
    L.ho_parse_composite_mode.restype = C.c_int; L.ho_parse_composite_mode.argtypes = [C.c_char_p]
    L.ho_composite.restype = C.c_int
    L.ho_composite.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_float, C.POINTER(abi.HaloComposite), f32p, C.POINTER(C.c_uint8), f32p,
                               C.POINTER(C.c_int32)]
    L.ho_consumer_snapshot.restype = C.c_int
    L.ho_consumer_snapshot.argtypes = [C.c_void_p, C.POINTER(abi.HaloDisplay), C.POINTER(C.c_uint8), f32p, C.POINTER(C.c_double)]
    _oracle[fma] = L
    return L


_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is not None:
        return _ref
    L = C.CDLL(REF_SO)
    L.ref_pcg_hash.restype = C.c_uint32; L.ref_pcg_hash.argtypes = [C.c_uint32]
    L.ref_u01_from_hash.restype = C.c_float; L.ref_u01_from_hash.argtypes = [C.c_uint32]
    L.ref_pcg_advance_hi.restype = C.c_uint32; L.ref_pcg_advance_hi.argtypes = [C.c_uint32] * 3
    L.ref_pcg_seed_with_high.restype = C.c_uint32; L.ref_pcg_seed_with_high.argtypes = [C.c_uint32] * 2
    L.ref_pcg_uniform.restype = C.c_float; L.ref_pcg_uniform.argtypes = [u32p]
    L.ref_pcg_gaussian.restype = C.c_float; L.ref_pcg_gaussian.argtypes = [u32p]
    L.ref_pcg_get_dist.restype = C.c_float; L.ref_pcg_get_dist.argtypes = [u32p, C.c_uint32, C.c_float, C.c_float]
    L.ref_normalize_latitude.restype = None; L.ref_normalize_latitude.argtypes = [C.c_float, f32p, i32p]
    L.ref_invert_lat_lut.restype = C.c_float; L.ref_invert_lat_lut.argtypes = [C.c_float, f32p, f32p, C.c_uint32]
    L.ref_lat_lut_bin.restype = C.c_uint32; L.ref_lat_lut_bin.argtypes = [C.c_float, f32p, C.c_uint32]
    L.ref_sample_lat_lon_roll.restype = None
    L.ref_sample_lat_lon_roll.argtypes = [u32p, C.c_void_p, f32p, f32p, f32p, f32p, f32p, f32p]
    L.ref_build_crystal_rotation_9.restype = None; L.ref_build_crystal_rotation_9.argtypes = [C.c_float] * 3 + [f32p]
    L.ref_apply_inverse_mat9.restype = None; L.ref_apply_inverse_mat9.argtypes = [f32p, f32p, f32p]
    L.ref_sample_triangle.restype = None; L.ref_sample_triangle.argtypes = [u32p, f32p, f32p]
    L.ref_sample_sph_cap.restype = None; L.ref_sample_sph_cap.argtypes = [u32p, C.c_float, C.c_float, C.c_float, f32p]
    L.ref_feistel_bijection.restype = C.c_uint32; L.ref_feistel_bijection.argtypes = [C.c_uint32] * 3
    L.ref_categorical_sample.restype = C.c_uint32; L.ref_categorical_sample.argtypes = [f32p, C.c_uint32, C.c_float]
    L.ref_wl_stream_seed.restype = C.c_uint32; L.ref_wl_stream_seed.argtypes = [C.c_uint32]
    L.ref_geom_shape_stream_seed.restype = C.c_uint32; L.ref_geom_shape_stream_seed.argtypes = [C.c_uint32]
    L.ref_reflect_ratio.restype = C.c_float; L.ref_reflect_ratio.argtypes = [C.c_float, C.c_float]
    L.ref_slab_face_t.restype = C.c_float; L.ref_slab_face_t.argtypes = [f32p, f32p, f32p, C.c_float]
    L.ref_proj_params_size.restype = C.c_int
    L.ref_project_exit_to_pixel.restype = None
    L.ref_project_exit_to_pixel.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, i32p]
    L.ref_spectrum_to_xyz.restype = None; L.ref_spectrum_to_xyz.argtypes = [C.c_float, C.c_float, f32p]
    L.ref_exact_prism.restype = None; L.ref_exact_prism.argtypes = [f32p, i32p]
    L.ref_gamut_clip_xyz.restype = None; L.ref_gamut_clip_xyz.argtypes = [f32p, f32p]
    L.ref_xyz_to_linear_rgb.restype = None; L.ref_xyz_to_linear_rgb.argtypes = [f32p, f32p]
    L.ref_linear_to_srgb.restype = C.c_float; L.ref_linear_to_srgb.argtypes = [C.c_float]
    L.ref_xyz_to_srgb_u8.restype = None; L.ref_xyz_to_srgb_u8.argtypes = [f32p, C.POINTER(C.c_uint8), C.c_int, C.c_float]
    L.ref_neumaier_add.restype = None; L.ref_neumaier_add.argtypes = [f32p, f32p, C.c_float]
    _ref = L
    return L


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(f32p)


def u32ptr(a):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


def i32ptr(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(i32p)


def bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)
