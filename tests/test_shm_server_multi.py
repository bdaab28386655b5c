"""The headless server spreading every frame over several device contexts (`--devices 0,0`): an unmodified
reference client (compiled from /root/reference by `make -C oracle ref`) pushes through shared memory and
pulls a whole frame, unaware of the split.  The client lives in a child process: the reference client keeps
its mapping of the region in a process global, and tests/test_shm_server.py has used this process's."""
import os
import signal
import subprocess
import sys
import time

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SERVER = os.path.join(ROOT, "ogl_beamforming_amd", "ogl_beamformer_server")
REF_CLIENT = os.path.join(ROOT, "oracle", "_ref", "libogl_beamformer_lib_ref.so")

CLIENT = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, {root!r})
from tests import cases
acq = cases.make({name!r})
ref = C.CDLL({client!r})
ref.beamformer_get_last_error_string.restype = C.c_char_p
ref.beamformer_get_last_frames.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
ref.beamformer_push_data_with_compute.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
for slot, fp in enumerate(acq.filters):
    assert ref.beamformer_create_filter(C.byref(fp), slot, 0), ref.beamformer_get_last_error_string()
assert ref.beamformer_push_simple_parameters(C.byref(acq.bp)), ref.beamformer_get_last_error_string()
ref.beamformer_set_global_timeout(20000)
rf = np.ascontiguousarray(acq.rf)
pts = [max(1, v) for v in acq.bp.output_points[:3]]
out = np.zeros(pts[0] * pts[1] * pts[2] * 2 + 16, np.float32)
for _ in range(4):                                   # several frames through the three RF slots
    assert ref.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 0), ref.beamformer_get_last_error_string()
assert ref.beamformer_get_last_frames(out.ctypes.data_as(C.c_void_p), out.nbytes, 1), ref.beamformer_get_last_error_string()
np.save({out!r}, out)
"""


@pytest.mark.parametrize("name", ["config4_small", "hercules_demod_decode_cw"])
def test_reference_client_through_a_two_context_server(name, oracle, tmp_path):
    if not os.path.exists(REF_CLIENT):
        pytest.skip("reference client library not built (needs /root/reference: make -C oracle ref)")
    server = subprocess.Popen([SERVER, "--devices", "0,0"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT)
    try:
        assert server.stdout.readline().startswith("ready")
        out_path = str(tmp_path / "frame.npy")
        script = CLIENT.format(root=ROOT, name=name, client=REF_CLIENT, out=out_path)
        r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    finally:
        server.send_signal(signal.SIGTERM)
        try:
            server.wait(timeout=20)
        except subprocess.TimeoutExpired:
            server.kill()
    acq = cases.make(name)
    want, _ = oracle.beamform(acq.bp, acq.rf, acq.filters)
    raw = np.load(out_path)
    got = (raw[: 2 * want.size].view(np.complex64) if want.dtype == np.complex64 else raw[: want.size]).reshape(want.shape)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    err = np.abs(got[ok] - want[ok]).max() / np.abs(want[ok]).max()
    assert err <= (cases.tolerance(acq) if acq.bp.interpolation_mode else 5e-2), err
