"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module,
and only to check (or time, as the CPU baseline) what the HIP path produced.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from ogl_beamforming_amd import params as P

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY_PATH = os.path.join(_HERE, "liboracle.so")
REF_LIBRARY_PATH = os.path.join(_HERE, "_ref", "libref_math.so")


class OracleParameterBlock(C.Structure):
    _fields_ = [
        ("parameters", P.Parameters),
        ("shaders", C.c_int32 * P.MAX_STAGES),
        ("filter_slots", C.c_uint8 * P.MAX_STAGES),
        ("shader_count", C.c_uint32),
        ("data_kind", C.c_int32),
        ("channel_mapping", C.c_int16 * P.MAX_CHANNELS),
        ("sparse_elements", C.c_int16 * P.MAX_CHANNELS),
        ("transmit_receive_orientations", C.c_uint8 * P.MAX_CHANNELS),
        ("focal_vectors", (C.c_float * 2) * P.MAX_CHANNELS),
        ("filters", P.FilterParameters * P.FILTER_SLOTS),
    ]


class OracleStage(C.Structure):
    _fields_ = [("kind", C.c_int), ("in_kind", C.c_int), ("out_kind", C.c_int),
                ("in_stride", C.c_int * 3), ("out_stride", C.c_int * 3),
                ("filter_slot", C.c_int), ("user_index", C.c_int)]


class OraclePlan(C.Structure):
    _fields_ = [
        ("stage_count", C.c_int), ("stages", OracleStage * P.MAX_STAGES),
        ("first_image_stage", C.c_int), ("iq_pipeline", C.c_int), ("chunk_channel_count", C.c_int),
        ("input_sample_count", C.c_int), ("das_sampling_frequency", C.c_float), ("das_time_offset", C.c_float),
        ("rf_size", C.c_uint32), ("pipeline_data_kind", C.c_int), ("output_points", C.c_int * 3),
        ("das_voxel_transform", C.c_float * 16), ("das_sparse", C.c_int),
    ]


def build(force=False):
    if force or not os.path.exists(LIBRARY_PATH):
        subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return LIBRARY_PATH


_lib = None


def library():
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(LIBRARY_PATH)
        fp = C.POINTER(C.c_float)
        lib.oracle_beamform.restype = C.c_int
        lib.oracle_beamform.argtypes = [C.POINTER(OracleParameterBlock), C.c_void_p, fp, C.POINTER(C.c_uint64), C.c_int]
        lib.oracle_beamform_subgrid.restype = C.c_int
        lib.oracle_beamform_subgrid.argtypes = [C.POINTER(OracleParameterBlock), C.c_void_p, fp, C.POINTER(C.c_uint64), C.c_int,
                                                C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
        lib.oracle_set_nearest_ambiguity_buffer.restype = None
        lib.oracle_set_nearest_ambiguity_buffer.argtypes = [C.c_void_p]
        lib.oracle_set_f64_frame.restype = None
        lib.oracle_set_f64_frame.argtypes = [C.c_void_p]
        lib.oracle_set_subgrid_stride.restype = None
        lib.oracle_set_subgrid_stride.argtypes = [C.c_uint32, C.c_uint32]
        lib.oracle_set_rows_outermost.restype = None
        lib.oracle_set_rows_outermost.argtypes = [C.c_int]
        lib.oracle_plan.restype = C.c_int
        lib.oracle_plan.argtypes = [C.POINTER(OracleParameterBlock), C.POINTER(OraclePlan)]
        lib.oracle_hadamard_transpose.argtypes = [C.c_int, fp]
        lib.oracle_kaiser_low_pass.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int, fp]
        lib.oracle_bessel_i0.restype = C.c_double
        lib.oracle_bessel_i0.argtypes = [C.c_double]
        lib.oracle_tukey_window.restype = C.c_float
        lib.oracle_tukey_window.argtypes = [C.c_float, C.c_float]
        lib.oracle_rf_chirp.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, fp]
        lib.oracle_baseband_chirp.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, fp]
        lib.oracle_real_filter_first_moment.restype = C.c_float
        lib.oracle_real_filter_first_moment.argtypes = [fp, C.c_int, C.c_float]
        lib.oracle_complex_filter_first_moment.restype = C.c_float
        lib.oracle_complex_filter_first_moment.argtypes = [fp, C.c_int, C.c_float]
        lib.oracle_m4_mul.argtypes = [fp, fp, fp]
        lib.oracle_das_transform.argtypes = [fp, fp, C.POINTER(C.c_int), fp]
        lib.oracle_das_transform_2d.argtypes = [C.c_int, fp, fp, C.c_float, fp]
        lib.oracle_das_transform_3d.argtypes = [fp, fp, fp]
        lib.oracle_enable_hilbert.argtypes = [C.c_int]
        lib.oracle_enable_hilbert.restype = None
        lib.oracle_hilbert_fir.argtypes = [fp]
        lib.oracle_hilbert_fir.restype = None
        lib.oracle_sum.argtypes = [fp, fp, C.c_float, C.c_uint64]
        lib.oracle_sum.restype = None
        lib.oracle_display.argtypes = [fp, C.c_uint64, C.c_int, C.c_float, C.c_float, C.c_float, fp]
        lib.oracle_display.restype = None
        lib.oracle_min_max.argtypes = [fp, C.c_uint64, C.c_int, fp]
        lib.oracle_min_max.restype = None
        lib.oracle_filter_create.restype = C.c_int
        lib.oracle_filter_create.argtypes = [C.POINTER(P.FilterParameters), fp, C.c_int, fp]
        _lib = lib
    return _lib


def parameter_block(bp, filters=()):
    """What the library holds for one block after beamformer_push_simple_parameters
    (lib/ogl_beamformer_lib.c:620-646) and beamformer_create_filter."""
    pb = OracleParameterBlock()
    C.memmove(C.byref(pb.parameters), C.byref(bp), C.sizeof(P.Parameters))
    for i in range(bp.compute_stages_count):
        pb.shaders[i] = bp.compute_stages[i]
        pb.filter_slots[i] = bp.compute_stage_parameters[i] & 0xFF
    pb.shader_count = bp.compute_stages_count
    pb.data_kind = bp.data_kind
    for i in range(P.MAX_CHANNELS):
        pb.channel_mapping[i] = bp.channel_mapping[i]
        pb.sparse_elements[i] = bp.sparse_elements[i]
        pb.transmit_receive_orientations[i] = bp.transmit_receive_orientations[i]
        pb.focal_vectors[i][0] = bp.steering_angles[i]
        pb.focal_vectors[i][1] = bp.focal_depths[i]
    for slot, f in enumerate(filters):
        if f is not None:
            pb.filters[slot] = f
    return pb


def plan(bp, filters=()):
    pb = parameter_block(bp, filters)
    out = OraclePlan()
    ok = library().oracle_plan(C.byref(pb), C.byref(out))
    return out if ok else None


def beamform(bp, rf, filters=(), threads=0, z=(0, 0), y=(0, 0), timing=None, stride=(1, 1), flags=None, truth=None):
    """Whole frame on the CPU, 16-channel chunks as the reference runs it.  z / y = (first,
    count) restrict the computed planes / rows (count 0 = whole axis); stride = (z, y) steps
    between the computed planes / rows.  Returns (frame (Z, Y, X) float32|complex64, pairs);
    timing (a dict) receives das_seconds; flags (a dict) receives, for nearest interpolation, "budget": per
    voxel the sum of |other sample - chosen sample| over the taps whose index sat within 2^-10 of a rounding
    boundary (what tap flips can move the coherent sum by), and "near_half" = budget > 0; truth (a dict) receives "frame": the
    same frame with every DAS stage run in double precision on the same float32 DAS input (oracle_set_f64_frame) -- complex128 / float64."""
    pb = parameter_block(bp, filters)
    p = plan(bp, filters)
    if p is None:
        raise RuntimeError("oracle could not plan this pipeline")
    pts = [max(1, v) for v in bp.output_points[:3]]
    ny, nz = (y[1] or pts[1]), (z[1] or pts[2])
    voxels = pts[0] * ny * nz
    n = 2 if p.iq_pipeline else 1
    out = np.zeros(voxels * n, np.float32)
    pairs = C.c_uint64(0)
    das_seconds = C.c_double(0)
    rf = np.ascontiguousarray(rf)
    library().oracle_set_subgrid_stride(stride[0], stride[1])
    budget = np.zeros(voxels, np.float32) if flags is not None else None
    library().oracle_set_nearest_ambiguity_buffer(budget.ctypes.data_as(C.c_void_p) if budget is not None else None)
    exact = np.zeros(voxels * n, np.float64) if truth is not None else None
    library().oracle_set_f64_frame(exact.ctypes.data_as(C.c_void_p) if exact is not None else None)
    try:
        ok = library().oracle_beamform_subgrid(C.byref(pb), rf.ctypes.data_as(C.c_void_p),
                                               out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(pairs), threads,
                                               z[0], z[1], y[0], y[1], C.byref(das_seconds))
    finally:
        library().oracle_set_subgrid_stride(1, 1)
        library().oracle_set_nearest_ambiguity_buffer(None)
        library().oracle_set_f64_frame(None)
    if truth is not None:
        truth["frame"] = (exact.view(np.complex128) if p.iq_pipeline else exact).reshape(nz, ny, pts[0])
    if flags is not None:
        flags["budget"] = budget.reshape(nz, ny, pts[0])
        flags["near_half"] = flags["budget"] > 0
    if not ok:
        raise RuntimeError("oracle_beamform failed")
    if timing is not None:
        timing["das_seconds"] = das_seconds.value
    frame = out.view(np.complex64) if p.iq_pipeline else out
    return frame.reshape(nz, ny, pts[0]), int(pairs.value)


def sum_frames(frames):
    """The reference's Sum stage over `frames` (oldest first): cleared image, then one
    sum.glsl pass per frame with prescale 1/len(frames) (beamformer_core.c:1417-1448)."""
    lib = library()
    first = np.ascontiguousarray(frames[0])
    out = np.zeros(first.shape, first.dtype)
    fp = C.POINTER(C.c_float)
    prescale = np.float32(1.0) / np.float32(len(frames))
    for f in frames:
        f = np.ascontiguousarray(f, dtype=first.dtype)
        lib.oracle_sum(out.ctypes.data_as(fp), f.ctypes.data_as(fp), C.c_float(float(prescale)), out.nbytes // 4)
    return out


def min_max(frame):
    lib = library()
    frame = np.ascontiguousarray(frame)
    out = np.zeros(2, np.float32)
    fp = C.POINTER(C.c_float)
    lib.oracle_min_max(frame.ctypes.data_as(fp), frame.size, int(np.iscomplexobj(frame)), out.ctypes.data_as(fp))
    return out


def display(frame, threshold_db=55.0, gamma=1.0, db_cutoff=0.0):
    """render_3d.frag.glsl:50-73 on a frame: float32 intensities in [0, 1]"""
    lib = library()
    frame = np.ascontiguousarray(frame)
    out = np.zeros(frame.shape, np.float32)
    fp = C.POINTER(C.c_float)
    lib.oracle_display(frame.ctypes.data_as(fp), frame.size, int(np.iscomplexobj(frame)), threshold_db, gamma, db_cutoff,
                       out.ctypes.data_as(fp))
    return out


def enable_hilbert(enable=True):
    """the build-defined Hilbert stage (oracle.h); off by default, as capabilities.hilbert is 0"""
    library().oracle_enable_hilbert(1 if enable else 0)
