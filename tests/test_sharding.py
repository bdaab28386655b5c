"""Multi-GPU host logic on CPU: slab arithmetic, and a world_size-2 gloo run of the same
broadcast -> slab -> gather sequence bench.py uses over RCCL."""
import os
import socket
import subprocess
import sys

from ogl_beamforming_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slabs_cover_the_grid():
    for planes in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 4, 8):
            parts = sharding.slabs(world, planes)
            assert parts[0][0] == 0 and sum(n for _, n in parts) == planes
            for (a, n), (b, _) in zip(parts, parts[1:]):
                assert a + n == b
            counts = [n for _, n in parts]
            assert max(counts) - min(counts) <= 1
    assert sharding.slabs(8, 512) == [(64 * r, 64) for r in range(8)]


def test_two_rank_gloo_sharded_frame_is_bit_identical(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    result = tmp_path / "result.txt"
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(result)]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert result.read_text() == "OK"
