"""bench.py's contract on a GPU box at a reduced scale: the JSON line and its roofline /
cpu_baseline objects, and the multi-GPU schedule (pipelined RF broadcast, slab verification)
rehearsed with two ranks on one GPU over gloo."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def last_json(text):
    return json.loads([l for l in text.splitlines() if l.startswith("{")][-1])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scale", "0.125", "--steps", "3", "--warmup", "1",
                        "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["unit"] == "voxels/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 64 ** 3 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    roof = d["roofline"]
    # `bound` names the resource that limits the dominant kernel (this reduced frame has 5 transmits: the gather kernel, bound by the
    # texture-address path; the full-size frame: "valu-issue"); achieved / peak / frac keep the contract's HBM formula
    assert roof["bound"] in ("valu-issue", "texture-address path (per-lane gathers served by L1)") and roof["contract_note"]
    assert roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert d["config"]["das_plan"]["kernel"] == roof["kernel"]
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and roof["kernel_ms"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["unit"] == "voxels/s" and cpu["sample"]
    # the same frames with the RF pushed from host memory (what tests/throughput.c:535-557 times), beside `value`, never as it
    up = d["config"]["upload_inclusive"]
    assert up["ms_per_frame"] > 0 and up["rf_bytes_per_frame"] > 0 and "beamformer_push_data_with_compute" in up["what"]
    assert up["voxels_per_s"] == pytest.approx(64 ** 3 / (up["ms_per_frame"] * 1e-3), rel=1e-6)
    assert up["rf_GBps"] == pytest.approx(up["rf_bytes_per_frame"] / (up["ms_per_frame"] * 1e-3) / 1e9, rel=1e-6)
    assert up["voxels_per_s"] <= d["value"] * 1.5          # (an upload cannot make frames much faster; small frames jitter)


def test_gpus_argument_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts one rank per GPU itself (a child torch.distributed.run) and relays
    the line; here both ranks rehearse on the one GPU over gloo.  n_gpus in the line is what was asked for."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "0.125", "--steps", "3", "--warmup", "1",
                        "--rehearse-on-one-gpu", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["slab_verified"] is True


def test_config2_line_names_the_block_staged_kernel():
    """BASELINE config 2 at full size (2.7 ms a frame): the line's das_plan and roofline name das_tile.hip, `bound` is VALU issue,
    `binding` prices the launch against the kernel's own VALU stream (committed PMC pass) with the LDS read path beside it"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    roof = d["roofline"]
    assert d["config"]["das_plan"]["kernel"] == roof["kernel"] == "das_tile_kernel" and d["config"]["das_plan"]["tile_window_samples"] == 32
    assert roof["bound"] == "valu-issue"
    b = roof["binding"]
    assert 0.3 < b["frac"] < 1.0 and b["lds_read_path"]["cycles_per_term_per_cu"] == 16.0 and 0.2 < b["lds_read_path"]["frac"] < 1.0
    assert 1024 * 1024 / (d["ms_per_step"] * 1e-3) == pytest.approx(d["value"], rel=1e-6)


@pytest.mark.parametrize("extra", [[], ["--serial-broadcast"]])
def test_two_ranks_rehearsed_on_one_gpu(extra):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "0.125", "--steps", "4",
           "--warmup", "2", "--rehearse-on-one-gpu", "--no-cpu-baseline", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["slab_verified"] is True
    assert ("overlaps" in d["config"]["sharding"]) == (not extra)
    assert d["roofline"]["pairs_total"] >= d["roofline"]["pairs_per_launch"] > 0


def test_in_process_two_device_contexts():
    """bench.py --in-process: the library's own multi-device mode (beamformer_hip_set_devices), here with the
    same ordinal twice -- the JSON line, per-device DAS times, pairs summed over devices"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--in-process", "--devices", "0,0", "--scale", "0.125",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = last_json(r.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 2 and "beamformer_hip_set_devices" in d["config"]["sharding"]
    assert len(d["config"]["device_das_ms"]) == 2 and all(v > 0 for v in d["config"]["device_das_ms"])
    assert d["config"]["rf_checksum_equal_on_all_ranks"] is True and len(d["config"]["devices"]) == 2
    assert [dev["slab"] for dev in d["config"]["devices"]] == [[0, 32], [32, 32]]
    assert d["roofline"]["pairs_total"] > d["roofline"]["pairs_per_launch"] > 0
    assert abs(d["value"] - 64 ** 3 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
