"""the headless shared-memory server lets the UNMODIFIED reference
client library drive the MI355X backend.

The client here is oracle/_ref/libogl_beamformer_lib_ref.so: the reference's own
lib/ogl_beamformer_lib.c compiled in place by oracle/Makefile (test infrastructure; it
travels to the GPU box prebuilt).  The server is ogl_beamforming_amd/ogl_beamformer_server
(csrc/shm_server.cpp)."""
import ctypes as C
import os
import re
import signal
import subprocess
import time

import numpy as np
import pytest

from ogl_beamforming_amd import params as P
from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SERVER = os.path.join(ROOT, "ogl_beamforming_amd", "ogl_beamformer_server")
REF_CLIENT = os.path.join(ROOT, "oracle", "_ref", "libogl_beamformer_lib_ref.so")


def layout():
    out = {}
    for line in open(os.path.join(ROOT, "tests", "golden", "shm_layout.txt")):
        if not line.startswith("#") and line.strip():
            k, v = line.split()
            out[k] = int(v)
    return out


def test_server_layout_matches_compiled_reference():
    """the numbers the server static_asserts are the compiled reference's"""
    want = layout()
    src = open(os.path.join(ROOT, "ogl_beamforming_amd", "csrc", "shm_server.cpp")).read()
    for key, pattern in {
        "sizeof.BeamformerSharedMemory": r"sizeof\(ShmHeader\) == (\d+)",
        "sizeof.BeamformerParameterBlock": r"sizeof\(ShmBlock\) == (\d+)",
        "sizeof.BeamformWork": r"sizeof\(ShmWork\) == (\d+)",
        "sizeof.BeamformWorkQueue": r"sizeof\(ShmQueue\) == (\d+)",
        "sizeof.BeamformerComputePipeline": r"sizeof\(ShmPipeline\) == (\d+)",
        "sizeof.Arena": r"kArenaHeaderBytes = (\d+)",
        "shm.locks": r"offsetof\(ShmHeader, locks\) == (\d+)",
        "shm.reserved_parameter_blocks": r"offsetof\(ShmHeader, reserved_parameter_blocks\) == (\d+)",
        "shm.rf_block_rf_size": r"offsetof\(ShmHeader, rf_block_rf_size\) == (\d+)",
        "shm.beamformed_frame_buffer_size": r"offsetof\(ShmHeader, beamformed_frame_buffer_size\) == (\d+)",
        "shm.capabilities": r"offsetof\(ShmHeader, capabilities\) == (\d+)",
        "shm.live_imaging_parameters": r"offsetof\(ShmHeader, live_imaging_parameters\) == (\d+)",
        "shm.live_imaging_dirty_flags": r"offsetof\(ShmHeader, live_imaging_dirty_flags\) == (\d+)",
        "shm.external_work_queue": r"offsetof\(ShmHeader, external_work_queue\) == (\d+)",
        "queue.work_items": r"offsetof\(ShmQueue, items\) == (\d+)",
        "work.create_filter.filter_slot": r"offsetof\(ShmWork, create_filter.filter_slot\) == (\d+)",
        "work.export.size": r"offsetof\(ShmWork, export_.size\) == (\d+)",
        "block.region_update_flags": r"offsetof\(ShmBlock, region_update_flags\) == (\d+)",
        "block.pipeline": r"offsetof\(ShmBlock, pipeline\) == (\d+)",
        "block.channel_mapping": r"offsetof\(ShmBlock, channel_mapping\) == (\d+)",
        "block.sparse_elements": r"offsetof\(ShmBlock, sparse_elements\) == (\d+)",
        "block.transmit_receive_orientations": r"offsetof\(ShmBlock, transmit_receive_orientations\) == (\d+)",
        "block.focal_vectors": r"offsetof\(ShmBlock, focal_vectors\) == (\d+)",
    }.items():
        m = re.search(pattern, src)
        assert m, key
        assert int(m.group(1)) == want[key], key
    for key, value in {"enum.WorkKind_ComputeIndirect": 1, "enum.WorkKind_CreateFilter": 2, "enum.WorkKind_ExportBuffer": 3,
                       "enum.Lock_ScratchSpace": 0, "enum.Lock_UploadRF": 1, "enum.Lock_ExportSync": 2,
                       "enum.Lock_DispatchCompute": 3, "enum.Lock_Count": 4, "enum.Export_Stats": 1}.items():
        assert want[key] == value


class Server:
    def __init__(self):
        self.proc = subprocess.Popen([SERVER], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT)
        line = self.proc.stdout.readline()
        assert line.startswith("ready"), line

    def expect(self, prefix, timeout=60.0):
        deadline = time.time() + timeout
        while time.time() < deadline:
            line = self.proc.stdout.readline()
            if not line:
                break
            if line.startswith(prefix):
                return line.strip()
        raise AssertionError(f"server never said {prefix!r}")

    def stop(self):
        self.proc.send_signal(signal.SIGTERM)
        try:
            self.proc.wait(timeout=20)
        except subprocess.TimeoutExpired:
            self.proc.kill()


@pytest.fixture(scope="module")
def server():
    """one server for the whole module: the reference client keeps its mapping of the region in
    a process global (lib/ogl_beamformer_lib.c:29-34), so it cannot follow a restarted server"""
    if not os.path.exists(SERVER):
        pytest.fail(f"{SERVER} not built (python -c 'import __graft_entry__ as g; g.build()')")
    if not os.path.exists(REF_CLIENT):
        pytest.skip("reference client library not built (needs /root/reference: make -C oracle ref)")
    s = Server()
    yield s
    s.stop()


def reference_client():
    lib = C.CDLL(REF_CLIENT)
    lib.beamformer_get_last_error_string.restype = C.c_char_p
    lib.beamformer_get_last_frames.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
    lib.beamformer_push_data_with_compute.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.beamformer_maximum_rf_data_size.restype = C.c_uint64
    lib.beamformer_maximum_frames_for_simple_parameters.restype = C.c_uint64
    return lib


def test_protocol_handshake_without_a_device(server):
    """CPU: the reference client connects, pushes parameters and work; the server answers every
    item and releases every lock (no compute can happen here: no HIP device)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by the GPU test")
    ref = reference_client()
    acq = cases.make("config1_small")
    assert ref.beamformer_get_api_version() == 33
    assert ref.beamformer_create_filter(C.byref(acq.filters[0]), 0, 0), ref.beamformer_get_last_error_string()
    server.expect("create_filter block 0 slot 0 ok")
    assert ref.beamformer_push_simple_parameters(C.byref(acq.bp)), ref.beamformer_get_last_error_string()
    assert ref.beamformer_maximum_frames_for_simple_parameters(C.byref(acq.bp)) == (4 << 30) // (64 * 64 * 8)
    assert ref.beamformer_maximum_rf_data_size() > acq.rf.nbytes
    ref.beamformer_set_global_timeout(5000)
    rf = np.ascontiguousarray(acq.rf)
    assert ref.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 0), ref.beamformer_get_last_error_string()
    assert server.expect("upload").startswith(f"upload block 0 bytes {rf.nbytes}")
    assert "failed" in server.expect("compute block 0")
    out = np.zeros(64 * 64 * 2, np.float32)
    ref.beamformer_get_last_frames(out.ctypes.data_as(C.c_void_p), out.nbytes, 1)     # returns: nothing hangs
    server.expect("export kind 0 count 1")
    # every lock is free again
    assert ref.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 0), ref.beamformer_get_last_error_string()
    server.expect("compute block 0")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["config1_small", "config4_small", "hercules_demod_decode_cw", "rca_shuffled_padded", "rca_a1s2"])
def test_reference_client_beamforms_through_the_server(name, server, oracle):
    """GPU: parameters, filter, RF and the pulled image all travel through the reference's own
    client code and shared-memory protocol; the image equals the oracle's."""
    ref = reference_client()
    acq = cases.make(name)
    for slot, fp in enumerate(acq.filters):
        assert ref.beamformer_create_filter(C.byref(fp), slot, 0), ref.beamformer_get_last_error_string()
    assert ref.beamformer_push_simple_parameters(C.byref(acq.bp)), ref.beamformer_get_last_error_string()
    ref.beamformer_set_global_timeout(20000)
    rf = np.ascontiguousarray(acq.rf)
    assert ref.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 0), ref.beamformer_get_last_error_string()
    want, _ = oracle.beamform(acq.bp, acq.rf, acq.filters)
    out = np.zeros(want.size * (2 if want.dtype == np.complex64 else 1) + 16, np.float32)
    assert ref.beamformer_get_last_frames(out.ctypes.data_as(C.c_void_p), out.nbytes, 1), ref.beamformer_get_last_error_string()
    assert "ok" in server.expect("compute block 0")
    got = (out[: 2 * want.size].view(np.complex64) if want.dtype == np.complex64 else out[: want.size]).reshape(want.shape)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    err = np.abs(got[ok] - want[ok]).max() / np.abs(want[ok]).max()
    assert err <= (cases.tolerance(acq) if acq.bp.interpolation_mode else 5e-2), err
    # the stats table export (BeamformerExportKind_Stats) through the same protocol
    lib_timings = getattr(ref, "beamformer_compute_timings")
    table = P.ComputeStatsTable()
    assert lib_timings(C.byref(table), 20000), ref.beamformer_get_last_error_string()
    ids = [table.shader_ids[i] for i in range(table.shader_count)]
    assert int(P.ShaderKind.DAS) in ids


def test_parameter_validation_agrees_with_the_compiled_reference_client(server, bflib):
    """Differential test of the drop-in's validation: 400 random parameter sets -- many invalid:
    counts past the limits, stage lists that start wrong, unknown stages and kinds, demodulation
    of complex data -- pushed through the compiled reference client (talking to the server) and
    through this library in process.  Same verdict and same error code every time
    (validate_parameters / validate_pipeline order, lib/ogl_beamformer_lib.c:252-311)."""
    ref = reference_client()
    ours = bflib.library()
    rng = np.random.default_rng(41)
    base = cases.make("config1_small")
    S, D = P.ShaderKind, P.DataKind
    verdicts = {True: 0, False: 0}
    for it in range(400):
        bp = P.SimpleParameters.from_buffer_copy(bytes(base.bp))
        n = int(rng.integers(0, 6))
        pool = [int(S.Decode), int(S.Filter), int(S.Demodulate), int(S.DAS), int(S.Hilbert), int(S.Sum), 7, 11, -1]
        weights = np.array([6, 4, 5, 6, 1, 1, 1, 1, 1], float)
        stages = [int(rng.choice(pool, p=weights / weights.sum())) for _ in range(n)]
        bp.compute_stages_count = int(rng.choice([n, n, n, 17, 0])) if rng.integers(0, 8) == 0 else n
        for i, s in enumerate(stages[:16]):
            bp.compute_stages[i] = s
        bp.data_kind = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 9], p=[.2, .15, .15, .15, .15, .1, .05, .05]))
        bp.channel_count = int(rng.choice([1, 16, 64, 256, 257, 0, 1000]))
        bp.acquisition_count = int(rng.choice([1, 8, 128, 256, 257, 0]))
        bp.sample_count = int(rng.choice([256, 4096, 0, 1]))
        bp.decimation_rate = int(rng.integers(0, 5))
        bp.raw_data_dimensions[0] = bp.sample_count * bp.acquisition_count
        bp.raw_data_dimensions[1] = int(rng.choice([bp.channel_count, 256, 0]))
        got_ref = bool(ref.beamformer_push_simple_parameters(C.byref(bp)))
        err_ref = int(ref.beamformer_get_last_error())
        got = bool(ours.beamformer_push_simple_parameters(C.byref(bp)))
        err = int(ours.beamformer_get_last_error())
        assert got == got_ref, (it, stages, bp.data_kind, bp.channel_count, bp.acquisition_count, bp.sample_count, err, err_ref)
        if not got:
            assert err == err_ref, (it, stages, bp.data_kind, P.LibError(err).name, P.LibError(err_ref).name)
        verdicts[got] += 1
    assert verdicts[True] >= 40 and verdicts[False] >= 40, verdicts       # both outcomes were exercised


def test_array_pushes_and_data_size_checks_agree_with_the_reference_client(server, bflib):
    """the remaining client-side checks (lib .c:438-464 array pushes, :410-429 filters, :503-511 data
    sizes, :239-250 block reservation), same differential arrangement"""
    ref = reference_client()
    ours = bflib.library()
    ref.beamformer_push_focal_vectors.argtypes = [C.POINTER(C.c_float), C.c_uint32]
    rng = np.random.default_rng(43)

    def both(name, *args):
        a = bool(getattr(ref, name)(*args)); ea = int(ref.beamformer_get_last_error())
        b = bool(getattr(ours, name)(*args)); eb = int(ours.beamformer_get_last_error())
        assert a == b, (name, args[1:], a, b, P.LibError(ea).name, P.LibError(eb).name)
        if not a:
            assert ea == eb, (name, args[1:], P.LibError(ea).name, P.LibError(eb).name)
        return a

    i16 = (C.c_int16 * 512)(*range(512))
    f32 = (C.c_float * 1024)()
    u8 = (C.c_uint8 * 512)()
    for count in (0, 1, 255, 256, 257, 512):
        both("beamformer_push_channel_mapping", i16, count)
        both("beamformer_push_sparse_elements", i16, count)
        both("beamformer_push_focal_vectors", f32, count)
        both("beamformer_push_transmit_receive_orientations", u8, count)
    for kind in (0, 1, 2, 7, -1):
        fp = P.FilterParameters()
        fp.kind = kind
        fp.sampling_frequency = 12.5e6
        fp.kaiser.cutoff_frequency, fp.kaiser.beta, fp.kaiser.length = 2e6, 5.0, 16
        both("beamformer_create_filter", C.byref(fp), 0, 0)
    for stages, count, kind in (([0, 3], 2, 0), ([2, 3], 2, 1), ([3], 1, 0), ([0] * 17, 17, 0), ([2, 0, 3], 3, 4), ([0, 9], 2, 0), ([0, 3], 2, 6)):
        arr = (C.c_int32 * 32)(*stages)
        both("beamformer_push_pipeline", arr, count, kind)

    # data size validation against the parameters in force (no compute needed to be rejected)
    acq = cases.make("config1_small")
    for lib in (ref, ours):
        assert lib.beamformer_push_simple_parameters(C.byref(acq.bp))
    rf = np.ascontiguousarray(acq.rf)
    ptr = rf.ctypes.data_as(C.c_void_p)
    for size, tag, slot in ((rf.nbytes - 2, 0, 0), (rf.nbytes + 2, 0, 0), (0, 0, 0), (rf.nbytes, 9, 0), (rf.nbytes, 0, 5)):
        a = bool(ref.beamformer_push_data_with_compute(ptr, size, tag, slot)); ea = int(ref.beamformer_get_last_error())
        b = bool(ours.beamformer_push_data_with_compute(ptr, size, tag, slot)); eb = int(ours.beamformer_get_last_error())
        assert not a and not b, (size, tag, slot, a, b)
        import torch
        if torch.cuda.is_available() or eb != int(P.LibError.SharedMemory):     # without a device ours stops earlier: no backend
            assert ea == eb, (size, tag, slot, P.LibError(ea).name, P.LibError(eb).name)


@pytest.mark.gpu
def test_reference_client_one_shot_call_through_the_server(server, oracle):
    """beamformer_beamform_data of the reference client (lib .c:704-736): parameters, RF, compute and
    the pulled image in one call, through shared memory"""
    ref = reference_client()
    ref.beamformer_beamform_data.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int32]
    acq = cases.make("config2_small")
    assert ref.beamformer_create_filter(C.byref(acq.filters[0]), 0, 0), ref.beamformer_get_last_error_string()
    want, _ = oracle.beamform(acq.bp, acq.rf, acq.filters)
    out = np.zeros(want.shape, np.complex64)
    rf = np.ascontiguousarray(acq.rf)
    ref.beamformer_set_global_timeout(20000)
    assert ref.beamformer_beamform_data(C.byref(acq.bp), rf.ctypes.data_as(C.c_void_p), rf.nbytes,
                                        out.ctypes.data_as(C.c_void_p), 20000), ref.beamformer_get_last_error_string()
    assert np.abs(out - want).max() <= cases.tolerance(acq) * np.abs(want).max()
