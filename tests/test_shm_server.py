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
