"""ctypes binding of libogl_beamformer_lib.so.

This is the stub a maintainer of the reference would keep: the reference exposes its client
library to Python through cffi over a preprocessed header (build.c:4798-4800); the function
names, argument order and return conventions below are those of
lib/ogl_beamformer_lib_base.h.  The library is the MI355X build in this repository; there
is no CPU fallback -- loading fails loudly when the .so is missing, and compute calls fail
with LibError.SharedMemory when no HIP device is present.
"""
import ctypes as C
import os

import numpy as np

from . import params as P

_HERE = os.path.dirname(os.path.abspath(__file__))
# OGL_BEAMFORMER_LIB points at another build of the same library (kernel experiments)
LIBRARY_PATH = os.environ.get("OGL_BEAMFORMER_LIB") or os.path.join(_HERE, "libogl_beamformer_lib.so")


class BeamformerError(RuntimeError):
    def __init__(self, kind, message):
        super().__init__(f"{P.LibError(kind).name}: {message}")
        self.kind = P.LibError(kind)


def _share_torch_hip_runtime():
    """One HIP runtime per process.  The PyTorch-ROCm wheel ships its own libamdhip64.so.7
    beside libtorch; the dynamic loader keys on the SONAME, so whichever copy is mapped first
    serves both.  The system copy first and torch's HSA runtime second leaves torch without a
    device ("No HIP GPUs are available"), so when torch is installed its copy is mapped before
    the library, without importing torch.  C and MATLAB clients never see this: they link
    the ROCm installation's runtime (csrc/Makefile rpath)."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return                                  # already mapped by torch itself
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass                                # the ROCm installation's runtime serves alone


def _load():
    if not os.path.exists(LIBRARY_PATH):
        raise ImportError(
            f"{LIBRARY_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU implementation to fall back to.")
    _share_torch_hip_runtime()
    lib = C.CDLL(LIBRARY_PATH)
    u32, i32, u64, vp = C.c_uint32, C.c_int32, C.c_uint64, C.c_void_p
    sig = {
        "beamformer_get_api_version": (u32, []),
        "beamformer_get_last_error": (i32, []),
        "beamformer_get_last_error_string": (C.c_char_p, []),
        "beamformer_error_string": (C.c_char_p, [i32]),
        "beamformer_maximum_frames_for_parameters": (u64, [C.POINTER(P.Parameters)]),
        "beamformer_maximum_frames_for_simple_parameters": (u64, [C.POINTER(P.SimpleParameters)]),
        "beamformer_maximum_rf_data_size": (u64, []),
        "beamformer_beamform_data": (u32, [C.POINTER(P.SimpleParameters), vp, u32, vp, i32]),
        "beamformer_set_global_timeout": (None, [u32]),
        "beamformer_push_data_with_compute": (u32, [vp, u32, u32, u32]),
        "beamformer_get_last_frames": (u32, [vp, u64, u32]),
        "beamformer_reserve_parameter_blocks": (u32, [u32]),
        "beamformer_set_pipeline_stage_parameters": (u32, [u32, i32]),
        "beamformer_set_pipeline_stage_parameters_at": (u32, [u32, i32, u32]),
        "beamformer_push_pipeline": (u32, [C.POINTER(i32), u32, i32]),
        "beamformer_push_pipeline_at": (u32, [C.POINTER(i32), u32, i32, u32]),
        "beamformer_push_simple_parameters": (u32, [C.POINTER(P.SimpleParameters)]),
        "beamformer_push_simple_parameters_at": (u32, [C.POINTER(P.SimpleParameters), u32]),
        "beamformer_push_parameters": (u32, [C.POINTER(P.Parameters)]),
        "beamformer_push_parameters_at": (u32, [C.POINTER(P.Parameters), u32]),
        "beamformer_push_channel_mapping": (u32, [C.POINTER(C.c_int16), u32]),
        "beamformer_push_channel_mapping_at": (u32, [C.POINTER(C.c_int16), u32, u32]),
        "beamformer_push_sparse_elements": (u32, [C.POINTER(C.c_int16), u32]),
        "beamformer_push_sparse_elements_at": (u32, [C.POINTER(C.c_int16), u32, u32]),
        "beamformer_push_focal_vectors": (u32, [C.POINTER(C.c_float), u32]),
        "beamformer_push_focal_vectors_at": (u32, [C.POINTER(C.c_float), u32, u32]),
        "beamformer_push_transmit_receive_orientations": (u32, [C.POINTER(C.c_uint8), u32]),
        "beamformer_push_transmit_receive_orientations_at": (u32, [C.POINTER(C.c_uint8), u32, u32]),
        "beamformer_create_filter": (u32, [C.POINTER(P.FilterParameters), C.c_uint8, C.c_uint8]),
        "beamformer_live_parameters_get_dirty_flag": (i32, []),
        "beamformer_set_live_parameters": (u32, [C.POINTER(P.LiveImagingParameters)]),
        "beamformer_get_live_parameters": (C.POINTER(P.LiveImagingParameters), []),
        "beamformer_compute_timings": (u32, [C.POINTER(P.ComputeStatsTable), i32]),
        # MI355X extensions (include/ogl_beamformer_hip.h)
        "beamformer_hip_set_device": (u32, [i32]),
        "beamformer_hip_get_device": (i32, []),
        "beamformer_hip_set_devices": (u32, [C.POINTER(i32), u32]),
        "beamformer_hip_get_device_count": (u32, []),
        "beamformer_hip_get_device_frame_timings": (u32, [u32, C.POINTER(P.HipFrameTimings)]),
        "beamformer_hip_get_device_info": (u32, [u32, C.POINTER(P.HipDeviceInfo)]),
        "beamformer_hip_set_stream": (u32, [vp]),
        "beamformer_hip_set_output_shard": (u32, [u32, u32, u32]),
        "beamformer_hip_push_device_data_with_compute": (u32, [vp, u32, u32, u32]),
        "beamformer_hip_synchronize": (u32, []),
        "beamformer_hip_get_last_frame_info": (u32, [C.POINTER(P.HipFrameInfo)]),
        "beamformer_hip_get_last_frame_timings": (u32, [C.POINTER(P.HipFrameTimings)]),
        "beamformer_hip_enable_frame_graphs": (u32, [u32]),
        "beamformer_hip_frame_graph_counts": (u32, [C.POINTER(u64), C.POINTER(u64)]),
        "beamformer_hip_enable_pair_counting": (u32, [u32]),
        "beamformer_hip_frame_min_max": (u32, [C.POINTER(C.c_float)]),
        "beamformer_hip_sum_last_frames": (u32, [u32, vp, u64]),
        "beamformer_hip_display_last_frame": (u32, [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), u64]),
        "beamformer_hip_enable_hilbert": (u32, [u32]),
        "beamformer_hip_set_das_path": (u32, [u32]),
        "beamformer_hip_zbp_parameters": (u32, [vp, u64, C.POINTER(P.SimpleParameters), C.POINTER(P.HipZbpPayload)]),
        "beamformer_hip_zbp_load": (u32, [C.c_char_p, u32, C.POINTER(P.SimpleParameters), C.POINTER(vp), C.POINTER(u64)]),
        "beamformer_hip_zbp_free": (None, [vp]),
        "beamformer_hip_zbp_last_error": (C.c_char_p, []),
        "beamformer_hip_host_das_transform": (None, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(i32), C.POINTER(C.c_float)]),
        "beamformer_hip_host_hadamard": (u32, [u32, C.POINTER(C.c_float)]),
        "beamformer_hip_host_filter": (i32, [C.POINTER(P.FilterParameters), C.POINTER(C.c_float), u32,
                                        C.POINTER(C.c_float), C.POINTER(u32)]),
        "beamformer_hip_describe_plan": (u32, [u32, C.POINTER(P.HipPlan)]),
        "beamformer_hip_describe_das": (u32, [u32, C.POINTER(P.HipDasDescription)]),
        "beamformer_hip_set_hook": (u32, [C.c_char_p, C.c_char_p]),
        "beamformer_hip_shutdown": (None, []),
    }
    for name, (restype, argtypes) in sig.items():
        fn = getattr(lib, name)          # AttributeError: the library must export every symbol
        fn.restype = restype
        fn.argtypes = argtypes
    lib._signatures = sig
    return lib


_lib = None


def library():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def exported_symbols():
    """Names the headers under include/ declare (kept in step by tests/test_abi.py)."""
    return sorted(library()._signatures)


def last_error():
    lib = library()
    return P.LibError(lib.beamformer_get_last_error()), lib.beamformer_get_last_error_string().decode()


def _check(result):
    if not result:
        kind, message = last_error()
        raise BeamformerError(kind, message)
    return result


def frame_shape(bp):
    pts = [max(1, int(v)) for v in bp.output_points[:3]]
    return (pts[2], pts[1], pts[0])      # z slowest, x fastest (das.glsl:132-136)


def output_is_complex(bp):
    stages = list(bp.compute_stages[: bp.compute_stages_count])
    return int(P.ShaderKind.Demodulate) in stages or P.DATA_KIND_COMPLEX[int(bp.data_kind)]


def beamform(bp, rf, filters=(), timeout_ms=-1):
    """One frame through the C ABI exactly as tests/throughput.c drives the reference:
    create_filter* -> push_simple_parameters -> push_data_with_compute -> get_last_frames.
    `rf` is a C-contiguous numpy array holding raw_data_dimensions[1] rows.  Returns the frame
    as float32 or complex64 with shape (Z, Y, X)."""
    lib = library()
    for slot, fp in enumerate(filters):
        if fp is not None:
            _check(lib.beamformer_create_filter(C.byref(fp), slot, 0))
    _check(lib.beamformer_push_simple_parameters(C.byref(bp)))
    lib.beamformer_set_global_timeout(C.c_uint32(timeout_ms & 0xFFFFFFFF).value)
    rf = np.ascontiguousarray(rf)
    _check(lib.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 0))
    return get_last_frame(bp)


def get_last_frame(bp, shard_planes=None):
    lib = library()
    shape = list(frame_shape(bp))
    if shard_planes is not None:
        shape[0] = shard_planes
    info = P.HipFrameInfo()
    _check(lib.beamformer_hip_get_last_frame_info(C.byref(info)))
    complex_out = info.data_kind == int(P.DataKind.Float32Complex)
    voxels = int(np.prod(shape))
    # with several devices the info describes the ingest device's slab; the export is the whole frame
    whole = (voxels * (8 if complex_out else 4) + 63) // 64 * 64
    raw = np.empty((max(int(info.size_bytes), whole) + 3) // 4, dtype=np.float32)
    _check(lib.beamformer_get_last_frames(raw.ctypes.data_as(C.c_void_p), raw.nbytes, 1))
    if complex_out:
        return raw[: 2 * voxels].view(np.complex64).reshape(shape)
    return raw[:voxels].reshape(shape)


def load_zbp(path, frame_number=0):
    """(SimpleParameters, raw RF bytes as a numpy uint8 array) from a ZBP .bp file, through the
    library's loader (tests/throughput.c:150-374 of the reference)."""
    lib = library()
    bp = P.SimpleParameters()
    rf, size = C.c_void_p(), C.c_uint64()
    if not lib.beamformer_hip_zbp_load(os.fsencode(path), frame_number, C.byref(bp), C.byref(rf), C.byref(size)):
        raise ValueError(f"{path}: {lib.beamformer_hip_zbp_last_error().decode()}")
    try:
        data = np.ctypeslib.as_array(C.cast(rf, C.POINTER(C.c_uint8)), shape=(size.value,)).copy()
    finally:
        lib.beamformer_hip_zbp_free(rf)
    return bp, data


def set_hook(name, value=None):
    """beamformer_hip_set_hook: a test / measurement hook of the library (name without the BEAMFORMER_HIP_ prefix; None = off)"""
    ok = library().beamformer_hip_set_hook(name.encode(), None if value is None else str(value).encode())
    assert ok, f"unknown hook {name}"


def describe_das(bp, filters=(), slot=0):
    """The DAS kernel the library would run for these parameters under the current das path mode, and why the others
    were declined (beamformer_hip_describe_das): (path, kernel, name, {path number: reason}, description struct).  Needs no device."""
    L = library()
    for i, fp in enumerate(filters):
        assert L.beamformer_create_filter(C.byref(fp), i, slot), last_error()
    assert L.beamformer_push_simple_parameters_at(C.byref(bp), slot), last_error()
    d = P.HipDasDescription()
    assert L.beamformer_hip_describe_das(slot, C.byref(d)), last_error()
    reasons = {k: bytes(d.declined[k]).split(b"\0")[0].decode() for k in range(8)}
    return int(d.path), d.kernel.decode(), d.name.decode(), reasons, d
