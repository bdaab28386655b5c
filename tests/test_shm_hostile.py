"""A hostile shared-memory client: work items written straight into the region with indices and sizes the
reference's client library would never produce.  Every value that arrives through the region is client
writable, so the server must answer (or ignore) each of them without indexing outside its lock and block
arrays, and stay alive.  CPU only (no device is needed to refuse a request)."""
import mmap
import os
import signal
import struct
import subprocess
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SERVER = os.path.join(ROOT, "ogl_beamforming_amd", "ogl_beamformer_server")
NAME = "/ogl_beamformer_hostile_test"
QUEUE, ITEMS, ITEM = 344, 352, 40                     # beamformer_shared_memory.c:133-166 (tests/golden/shm_layout.txt)
RESERVED, RF_BLOCK_RF_SIZE, LOCKS = 88, 96, 8
LOCK_UPLOAD_RF = 1
KIND_COMPUTE, KIND_COMPUTE_INDIRECT, KIND_CREATE_FILTER, KIND_EXPORT = 0, 1, 2, 3


class Region:
    def __init__(self):
        fd = os.open("/dev/shm" + NAME, os.O_RDWR)
        self.m = mmap.mmap(fd, 1 << 20)
        os.close(fd)

    def push(self, raw):
        assert len(raw) == ITEM
        widx = struct.unpack_from("<Q", self.m, QUEUE)[0] & 0xFFFFFFFF
        self.m[ITEMS + (widx & 63) * ITEM: ITEMS + (widx & 63) * ITEM + ITEM] = raw
        q = struct.unpack_from("<Q", self.m, QUEUE)[0]
        struct.pack_into("<Q", self.m, QUEUE, (q & ~0xFFFFFFFF) | ((widx + 1) & 0xFFFFFFFF))


def work(kind, lock, payload=b""):
    return struct.pack("<ii", kind, lock) + payload.ljust(32, b"\0")


def test_server_survives_hostile_work_items():
    if not os.path.exists(SERVER):
        pytest.fail(f"{SERVER} not built")
    proc = subprocess.Popen([SERVER, "--name", NAME, "--size", str(1 << 24)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT)
    try:
        assert proc.stdout.readline().startswith("ready")
        r = Region()

        def expect(prefix, timeout=20.0):
            deadline = time.time() + timeout
            while time.time() < deadline:
                line = proc.stdout.readline()
                assert line, "server died"
                if line.startswith(prefix):
                    return line.strip()
            raise AssertionError(f"server never said {prefix!r}")

        # export with a negative lock index, then one past the lock array
        r.push(work(KIND_EXPORT, -5, struct.pack("<IIQ", 0, 1, 64)))
        assert "out of range" in expect("export")
        r.push(work(KIND_EXPORT, 4 + 16, struct.pack("<IIQ", 0, 1, 64)))
        assert "out of range" in expect("export")
        # compute on a parameter block far outside the reserved ones
        r.push(work(KIND_COMPUTE, 0, struct.pack("<II", 0, 9999)))
        assert "failed" in expect("compute block 9999")
        # a client that claims 4 billion reserved blocks, then names block 20 (past the protocol's 16)
        struct.pack_into("<I", r.m, RESERVED, 0xFFFFFFFF)
        r.push(work(KIND_COMPUTE, 0, struct.pack("<II", 0, 20)))
        assert "failed" in expect("compute block 20")
        struct.pack_into("<I", r.m, RESERVED, 1)
        # an export larger than the region
        r.push(work(KIND_EXPORT, 2, struct.pack("<IIQ", 0, 1, 1 << 40)))
        assert "failed" in expect("export kind 0")
        # an RF upload that claims more bytes than the scratch arena holds
        struct.pack_into("<i", r.m, LOCKS + 4 * LOCK_UPLOAD_RF, 1)
        struct.pack_into("<Q", r.m, RF_BLOCK_RF_SIZE, (0 << 32) | 0xFFFFFFF0)
        assert "failed" in expect("upload")
        # unknown work kind
        r.push(work(77, 0))
        expect("unknown work kind 77")
        assert proc.poll() is None
    finally:
        proc.send_signal(signal.SIGTERM)
        try:
            proc.wait(timeout=20)
        except subprocess.TimeoutExpired:
            proc.kill()
        try:
            os.unlink("/dev/shm" + NAME)
        except OSError:
            pass
