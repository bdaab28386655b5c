#!/usr/bin/env python3
"""bench.py -- beamformed voxels/s of the DAS hot path on MI355X (BASELINE.json metric).

A step is one whole frame of BASELINE.json configs[3] through the C ABI: Int16 RF resident
in HBM -> Demodulate -> DAS with fused coherency weighting -> 512^3 complex voxels
(3-D row-column array, 256 channels x 75 plane waves).
With N > 1 (one process per GPU, launched by torch.distributed.run) the volume is the SAME
512^3 grid cut into N z-slabs (strong scaling): rank 0 holds the RF frame, every step
broadcasts it over RCCL/xGMI (the one collective of the path), every rank beamforms its
slab.  No reduction collective.

`--in-process` runs the same N-way slab split inside ONE process through the library's own
multi-device mode (beamformer_hip_set_devices: RF copied to the peers with hipMemcpyPeerAsync,
no torch.distributed) -- what a C / MATLAB client of the push-RF / pull-image API gets.

Prints ONE JSON line (rank 0).  `roofline` prices the DAS kernel against the HBM roofline with
the ALGORITHMIC gather bytes of BASELINE.md section 4 (the contract's `frac`: it exceeds 1
because those bytes are L1/L2 hits, not HBM reads), next to `binding` -- the resource that
actually limits the kernel, priced against the ceiling tools/microbench.hip measured
(profiles/r02_microbench.json) -- and the measured HBM traffic (`traffic`, `hbm_measured_GBps`).
`cpu_baseline` times the CPU oracle (a port of the reference shaders; the reference itself
cannot be built here) on evenly spaced planes of the same frame on the host cores.
"""
import argparse
import ctypes as C
import hashlib
import json
import re
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8.0e12   # B/s, MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)

PATH_NAMES = ["general kernel", "separable-delay gather kernel", "separable-delay LDS-staged kernel", "per-voxel factored kernel",
              "HERCULES aligned-grid kernel", "per-voxel factored kernel with block-wide LDS staging"]
KERNEL_NAMES = ["das_kernel", "das_rca_separable_kernel", "das_rca_staged_kernel", "das_factored_kernel", "das_hercules_kernel", "das_tile_kernel"]
# the sources whose hash ties a committed PMC figure to the code that produced it (tools/summarize_profiles.py): the
# kernel's own file plus the headers every DAS kernel includes
KERNEL_FILES = {"das_kernel": "das.hip", "das_rca_separable_kernel": "das_separable.hip", "das_rca_staged_kernel": "das_staged.hip",
                "das_rca_staged_real_kernel": "das_staged_real.hip", "das_rca_staged_cubic_kernel": "das_staged_cubic.hip", "das_factored_kernel": "das_factored.hip", "das_hercules_kernel": "das_hercules.hip", "das_tile_kernel": "das_tile.hip"}
COMMON_SOURCES = ["das_common.h", "das_exact.h", "bf_kernels.h", "das_select.cpp"]       # (the selection rules decide tile shapes, windows and walks: they change a kernel's traffic)

def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=str, default="4",
                    help="BASELINE.json configs index (1-based; 4 = the metric's workload), or harness:{tpw,tpw_swapped,vls,hercules,forces}[:yz] = "
                         "the frame the reference's own throughput harness beamforms (tests/throughput.c:443-491): then NOT the metric's workload")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink factor for rehearsal runs (metric is scale 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="CPU-baseline budget (DAS time summed over its three legs)")
    ap.add_argument("--das-path", type=int, default=0, help="0 auto, 1 general DAS kernel, 2 auto without the LDS-staged kernel (the gather kernel instead), 3 LDS-staged kernel wherever its window bound holds, 4 factored kernel wherever it applies, 6 HERCULES aligned-grid kernel also on narrow grids; flags to add: 16 no channel split, 256 / 512 the block-staged factored kernel (das_tile.hip) on / off")
    ap.add_argument("--serial-broadcast", action="store_true",
                    help="multi-GPU: broadcast and compute back to back on one stream instead of pipelined")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development aid: run the N-rank code path with every rank on GPU 0 and gloo for the "
                         "collectives (RCCL refuses two ranks per device); the value is NOT the metric")
    ap.add_argument("--in-process", action="store_true",
                    help="N devices inside this one process (beamformer_hip_set_devices) instead of one process per GPU")
    ap.add_argument("--devices", type=str, default="",
                    help="--in-process: comma separated HIP ordinals (default 0..N-1; an ordinal may repeat, e.g. 0,0 on a one-GPU box: then NOT the metric)")
    ap.add_argument("--frame-graph", action="store_true",
                    help="replay every frame from a captured hipGraph (beamformer_hip_enable_frame_graphs; BASELINE configs[4] names a "
                         "hipGraph-captured frame).  Off by default: never faster (profiles/r02_graph_probe.json), and a graph frame times as one segment")
    ap.add_argument("--interpolation", choices=["nearest", "linear", "cubic"], default=None,
                    help="override the configuration's interpolation mode (config 5 with cubic = the reference harness's own setting)")
    ap.add_argument("--planes", type=int, default=0, help="profiling aid: beamform only this many centre z-planes (value is then NOT the metric)")
    args = ap.parse_args()
    args.gpus_given = any(a == "--gpus" or a.startswith("--gpus=") for a in sys.argv[1:])
    return args


def resolve_gpus(args, environ, visible_devices):
    """What `--gpus N` means in this invocation -- decided BEFORE anything touches the GPU (a process that has initialised HIP must not be
    replaced, and a scaling harness that quietly ran on one GPU wastes the lease).  Returns (mode, n_gpus, ordinals):
      "single"      this process, one GPU;
      "in-process"  this process, N devices through beamformer_hip_set_devices (`--in-process`);
      "rank"        this process is one of N ranks started by torch.distributed.run (WORLD_SIZE must equal --gpus);
      "spawn"       `--gpus N > 1` without a launcher: start `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
                    and relay its JSON line and exit code.
    Raises SystemExit with a message instead of ever reporting fewer GPUs than asked for."""
    world = int(environ.get("WORLD_SIZE", "0") or 0)
    if args.gpus < 1:
        raise SystemExit(f"bench.py: --gpus {args.gpus}: at least one GPU")
    if world:
        if args.in_process:
            raise SystemExit("bench.py: --in-process inside a torch.distributed.run launch: one or the other")
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a line for the wrong GPU count")
        if not args.rehearse_on_one_gpu and visible_devices < world:
            raise SystemExit(f"bench.py: {world} ranks but only {visible_devices} GPU(s) visible (one process per GPU)")
        return ("rank" if world > 1 else "single"), world, [int(environ.get("LOCAL_RANK", "0"))]
    if args.in_process:
        ordinals = [int(v) for v in args.devices.split(",") if v != ""]
        if ordinals:
            if args.gpus_given and len(ordinals) != args.gpus:
                raise SystemExit(f"bench.py: --gpus {args.gpus} but --devices names {len(ordinals)} ordinal(s)")
        else:
            ordinals = list(range(args.gpus))
        if max(ordinals) >= visible_devices or min(ordinals) < 0:
            raise SystemExit(f"bench.py: --devices {ordinals}: only {visible_devices} GPU(s) visible")
        return "in-process", len(ordinals), ordinals
    if args.devices:
        raise SystemExit("bench.py: --devices goes with --in-process")
    if args.gpus == 1:
        return "single", 1, [0]
    if args.rehearse_on_one_gpu:
        if visible_devices < 1:
            raise SystemExit("bench.py: no GPU visible")
    elif visible_devices < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {visible_devices} GPU(s) visible: refusing to print a line that says n_gpus {args.gpus}")
    return "spawn", args.gpus, list(range(args.gpus))


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher around it: one process per GPU under torch.distributed.run, as a child of this
    (GPU-untouched) process; its stdout -- rank 0's one JSON line -- and its exit code pass through."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    # torch.cuda.device_count() reads the driver's device list without creating a HIP context (what makes the child launch below legal)
    mode, n_gpus, ordinals = resolve_gpus(args, os.environ, torch.cuda.device_count())
    if mode == "spawn":
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    world = n_gpus if mode == "rank" else 1
    rank = int(os.environ.get("RANK", "0")) if mode == "rank" else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if mode == "rank" else 0
    distributed = mode == "rank"
    rehearse = args.rehearse_on_one_gpu
    in_process = mode == "in-process"
    if distributed:
        import torch.distributed as dist
        if rehearse:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
    else:
        dist = None
        torch.cuda.set_device(ordinals[0])
    device = torch.device("cuda", local_rank if distributed else ordinals[0])

    from ogl_beamforming_amd import configs, lib, params as P, sharding
    L = lib.library()
    if in_process:
        arr = (C.c_int32 * len(ordinals))(*ordinals)
        assert L.beamformer_hip_set_devices(arr, len(ordinals)), lib.last_error()
    else:
        assert L.beamformer_hip_set_device(device.index), lib.last_error()
    L.beamformer_hip_set_das_path(args.das_path)
    if args.frame_graph:
        assert not in_process or n_gpus == 1, "frame graphs are a one-device feature"
        L.beamformer_hip_enable_frame_graphs(1)

    # every rank builds the parameter block (cheap, deterministic); only rank 0 owns the RF
    harness = args.config.startswith("harness:")
    if not harness:
        args.config = int(args.config)
    acq = configs.by_name(args.config, args.scale)
    bp = acq.bp
    if args.interpolation:
        bp.interpolation_mode = {"nearest": 0, "linear": 1, "cubic": 2}[args.interpolation]
    X, Y, Z = (max(1, v) for v in bp.output_points[:3])
    voxels_total = X * Y * Z
    for slot, fp in enumerate(acq.filters):
        assert L.beamformer_create_filter(C.byref(fp), slot, 0), lib.last_error()
    assert L.beamformer_push_simple_parameters(C.byref(bp)), lib.last_error()
    z_first, z_count = sharding.slab(rank, world, Z)
    if args.planes and not distributed:
        z_count = min(Z, args.planes)
        z_first = (Z - z_count) // 2
        voxels_total = X * Y * z_count
    if distributed or args.planes:
        assert L.beamformer_hip_set_output_shard(0, z_first, z_count), lib.last_error()
    planes_rank0 = sharding.slab(0, n_gpus, z_count)[1] if in_process else z_count   # planes of the launch `roofline` prices

    rf_host = torch.from_numpy(np.ascontiguousarray(acq.rf).view(np.uint8).reshape(-1))
    rf_dev = torch.empty(rf_host.numel(), dtype=torch.uint8, device=device)
    if rank == 0:
        rf_dev.copy_(rf_host)
    torch.cuda.synchronize(device)
    stream = None
    if not in_process:
        # One explicit stream carries the RCCL broadcast and every kernel of the library, so a frame's
        # DAS is ordered behind its broadcast without host synchronisation.  (torch's default stream
        # has handle 0, which the library reads as "use your own stream": never pass that.)
        stream = torch.cuda.Stream(device=device)
        torch.cuda.set_stream(stream)
        assert stream.cuda_stream != 0
        assert L.beamformer_hip_set_stream(C.c_void_p(stream.cuda_stream)), lib.last_error()

    # Multi-GPU schedule: the broadcast of frame n+1 runs on its own stream into the
    # other of two RF buffers while the kernels of frame n run; events order "broadcast landed ->
    # compute" and "frame that read this buffer finished -> next broadcast into it".  A step is
    # one broadcast issued plus one frame computed; the pipeline is primed before the timed region.
    # (--in-process: the library does the equivalent itself, three RF slots deep.)
    pipelined = distributed and not args.serial_broadcast
    rf_bounce = torch.empty(rf_host.numel(), dtype=torch.uint8) if (distributed and rehearse) else None
    if pipelined:
        comm = torch.cuda.Stream(device=device)
        bufs = [torch.empty_like(rf_dev), torch.empty_like(rf_dev)]
        landed = [torch.cuda.Event(), torch.cuda.Event()]
        released = [torch.cuda.Event(), torch.cuda.Event()]
        for e in released:
            e.record(stream)
        state = {"issued": 0, "computed": 0}

    def broadcast_into(buf):
        if rehearse:                               # gloo has no device tensors: bounce through the host
            if rank == 0:
                rf_bounce.copy_(rf_dev)
            sharding.broadcast_rf(rf_bounce, src=0)
            buf.copy_(rf_bounce)
        else:
            if rank == 0 and buf.data_ptr() != rf_dev.data_ptr():
                buf.copy_(rf_dev, non_blocking=True)
            sharding.broadcast_rf(buf, src=0)      # RCCL over xGMI

    def issue_broadcast():
        k = state["issued"] % 2
        with torch.cuda.stream(comm):
            comm.wait_event(released[k])
            broadcast_into(bufs[k])
            landed[k].record(comm)
        state["issued"] += 1

    min_max = (C.c_float * 2)()

    def push(buf):
        ok = L.beamformer_hip_push_device_data_with_compute(C.c_void_p(buf.data_ptr()), buf.numel(), 0, 0)
        assert ok, lib.last_error()
        if args.config == 5:                       # BASELINE configs[4] ends in min_max: part of the step
            assert L.beamformer_hip_frame_min_max(min_max), lib.last_error()

    def step():
        if pipelined:
            k = state["computed"] % 2
            stream.wait_event(landed[k])
            push(bufs[k])
            released[k].record(stream)
            state["computed"] += 1
            issue_broadcast()                      # next frame's RF, overlapping this frame's kernels
            return
        if distributed:
            broadcast_into(rf_dev)                 # same stream as the kernels
        push(rf_dev)

    if pipelined:
        issue_broadcast()                          # prime: frame 0's RF

    def fence():
        if in_process:
            assert L.beamformer_hip_synchronize(), lib.last_error()     # every device of the library
        torch.cuda.synchronize(device)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)

    # what the library will run for this block, in its own words (beamformer_hip_describe_das)
    dd = P.HipDasDescription()
    assert L.beamformer_hip_describe_das(0, C.byref(dd)), lib.last_error()
    das_plan = {"kernel": dd.kernel.decode(), "tile_shift": list(dd.tile_shift), "blocks": list(dd.blocks), "split_shift": int(dd.split_shift),
                "tile_walk": int(dd.tile_walk), "u_axis": int(dd.u_axis), "u_shift": int(dd.u_shift), "v_shift": int(dd.v_shift),
                "window_samples": int(dd.window_samples), "uniform_tables": int(dd.uniform_tables), "lds_bytes": int(dd.lds_bytes), "threads": int(dd.threads),
                "channel_chunk": int(dd.channel_chunk), "hercules_prepared_copy": int(dd.hercules_prepared_copy),
                "tile_window_samples": int(dd.tile_window_samples), "tile_spread_estimate": round(float(dd.tile_spread_estimate), 2),
                "declined": {str(k): bytes(dd.declined[k]).split(b"\0")[0].decode() for k in range(8) if bytes(dd.declined[k]).split(b"\0")[0]}}

    # one untimed frame with the geometry-only pair count: G of BASELINE.md section 4
    L.beamformer_hip_enable_pair_counting(1)
    step()
    fence()
    t = P.HipFrameTimings()
    assert L.beamformer_hip_get_last_frame_timings(C.byref(t)), lib.last_error()
    pairs_all_devices = int(t.das_pairs)           # in-process: summed over the library's devices
    if in_process:
        assert L.beamformer_hip_get_device_frame_timings(0, C.byref(t)), lib.last_error()
    pairs_local = int(t.das_pairs)
    taps, sample_bytes, das_path = int(t.das_taps), int(t.das_sample_bytes), int(t.das_path)
    L.beamformer_hip_enable_pair_counting(0)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0

    # Multi-GPU runs check what the schedule delivered, outside the timed region:
    #  (1) the slab of the last timed frame must be bit-identical to a recomputation from the same RF
    #      buffer after a full fence (a broadcast that had not landed, or a buffer overwritten early);
    #  (2) every rank's RF buffer must hold rank 0's bytes (a rank that consistently beamforms stale,
    #      zero or partially landed RF passes (1)): a 64-bit checksum is all-reduced (MIN == MAX).
    verified, rf_checksum_ok = None, None
    if distributed:
        last_buf = bufs[(state["computed"] - 1) % 2] if pipelined else rf_dev
        first = lib.get_last_frame(bp, shard_planes=z_count).copy()
        push(last_buf)
        fence()
        again = lib.get_last_frame(bp, shard_planes=z_count)
        verified = bool(np.array_equal(first.view(np.uint32), again.view(np.uint32)))
        assert verified, f"rank {rank}: the pipelined frame differs from its recomputation"
        words = last_buf[: last_buf.numel() // 8 * 8].view(torch.int64)
        weights = torch.arange(1, words.numel() + 1, device=words.device, dtype=torch.int64)
        check = (words * weights).sum().reshape(1)               # wraps mod 2^64: position sensitive
        check = check.cpu() if rehearse else check
        lo, hi = check.clone(), check.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        rf_checksum_ok = bool(int(lo) == int(hi))
        assert rf_checksum_ok, f"rank {rank}: RF buffer differs from rank 0's after the broadcast"

    # DAS kernel duration: HIP event pairs recorded by the library on the stream it launches
    # on (beamformer_compute_timings = the reference's per-stage stats table; in-process: the
    # slowest device per stage)
    stats = P.ComputeStatsTable()
    assert L.beamformer_compute_timings(C.byref(stats), -1), lib.last_error()
    n_stage = int(stats.shader_count)
    ids = [int(stats.shader_ids[i]) for i in range(n_stage)]
    das_col = ids.index(int(P.ShaderKind.DAS))
    info = P.HipFrameInfo()
    L.beamformer_hip_get_last_frame_info(C.byref(info))
    last_id = int(info.frame_id) - (1 if distributed else 0)      # skip the verification frame
    rows = [(last_id - k) % 32 for k in range(args.steps)]
    das_s = float(np.mean([stats.times[r][das_col] for r in rows]))
    stage_ms = {P.ShaderKind(ids[i]).name: float(np.mean([stats.times[r][i] for r in rows])) * 1e3 for i in range(n_stage)}
    device_das_ms, devices_info = None, None
    if in_process and n_gpus > 1:
        # what every device did with the newest frame: its slab, its DAS time, how long the RF copy into it took, whether that copy
        # crossed xGMI directly, and a checksum of the RF it read -- computed on the device, compared here
        device_das_ms, devices_info = [], []
        for i in range(n_gpus):
            di = P.HipDeviceInfo()
            assert L.beamformer_hip_get_device_info(i, C.byref(di)), lib.last_error()
            device_das_ms.append(float(di.das_ms))
            devices_info.append({"ordinal": int(di.ordinal), "slab": [int(di.slab_first), int(di.slab_count)], "das_ms": float(di.das_ms), "frame_ms": float(di.frame_ms),
                                 "peer_copy_ms": float(di.peer_copy_ms), "peer_access": {2: "ingest device", 1: "direct (xGMI peer access)", 0: "staged through host memory"}[int(di.peer_access)],
                                 "rf_checksum": f"{int(di.rf_checksum):016x}", "rf_bytes": int(di.rf_bytes)})
        das_s = device_das_ms[0] * 1e-3              # the launch `roofline` prices is device 0's
        rf_checksum_ok = len({d["rf_checksum"] for d in devices_info}) == 1 and all(d["rf_bytes"] > 0 for d in devices_info)
        assert rf_checksum_ok, f"the devices beamformed different RF: {devices_info}"

    # The same frames with the RF coming from HOST memory through beamformer_push_data_with_compute -- what the reference's own harness times
    # (tests/throughput.c:535-557: push -> frame, raw-RF GB/s) and what a C / MATLAB client of the unchanged API gets: the caller's bytes
    # are copied into one of three pinned slots, cross PCIe on the copy stream while the previous frame computes, and the frame follows.
    # Reported beside `value` (config.upload_inclusive), never as it.
    upload = None
    if not distributed and n_gpus == 1 and not args.planes:
        host_rf = np.ascontiguousarray(acq.rf)
        host_ptr, host_bytes = host_rf.ctypes.data_as(C.c_void_p), host_rf.nbytes

        def host_step():
            assert L.beamformer_push_data_with_compute(host_ptr, host_bytes, 0, 0), lib.last_error()
            if args.config == 5:
                assert L.beamformer_hip_frame_min_max(min_max), lib.last_error()

        for _ in range(max(1, args.warmup)):
            host_step()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            host_step()
        fence()
        per_frame = (time.perf_counter() - t1) / args.steps
        upload = {"ms_per_frame": per_frame * 1e3, "voxels_per_s": voxels_total / per_frame, "rf_GBps": host_bytes / per_frame / 1e9,
                  "rf_bytes_per_frame": host_bytes,
                  "what": "host RF -> beamformer_push_data_with_compute (pinned three-slot upload, H2D on the copy stream overlapping the previous frame) "
                          "-> frame, as tests/throughput.c:535-557 measures; `value` above starts from RF resident in HBM"}

    if distributed:
        agg = torch.tensor([elapsed, das_s, float(pairs_local)], dtype=torch.float64, device="cpu" if rehearse else device)
        mx = agg.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = agg.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, pairs_total = float(mx[0]), int(sm[2].item())
    else:
        pairs_total = pairs_all_devices

    # a measured copy figure next to the 8 TB/s nominal peak (1 GiB device-to-device
    # copies: bytes read + bytes written over the event-timed duration)
    hbm_copy = None
    if rank == 0 and not args.planes:
        a_buf = torch.empty(1 << 30, dtype=torch.uint8, device=device)
        b_buf = torch.empty_like(a_buf)
        b_buf.copy_(a_buf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b_buf.copy_(a_buf)
        e1.record()
        e1.synchronize()
        hbm_copy = 10 * 2 * a_buf.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a_buf, b_buf

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = voxels_total / (elapsed / args.steps)
        voxel_bytes = 8 if int(info.data_kind) == int(P.DataKind.Float32Complex) else 4
        cw = 4 if bp.coherency_weighting else 0
        # per launch = rank 0's / device 0's slab; the others are symmetric
        gather_bytes = pairs_local * taps * sample_bytes
        bytes_alg = gather_bytes + (X * Y * planes_rank0) * (voxel_bytes + cw)
        achieved = bytes_alg / das_s / 1e9
        whole_metric_frame = args.scale == 1.0 and not args.planes and n_gpus == 1
        traffic, traffic_source = measured_traffic(args.config, KERNEL_NAMES[das_path]) if whole_metric_frame else (None, "not the profiled launch (scale, planes or device count differ)")
        if distributed:
            how = (f"{world} z-slab(s), one process per GPU, RF broadcast via " + ("gloo through the host (one-GPU rehearsal)" if rehearse else "RCCL")
                   + (", broadcast of frame n+1 overlaps frame n" if pipelined else ", serial"))
        elif in_process and n_gpus > 1:
            how = (f"{n_gpus} z-slab(s) inside one process (beamformer_hip_set_devices, ordinals {ordinals}): RF copied to the peers with "
                   "hipMemcpyPeerAsync on per-destination streams, three RF slots deep")
        else:
            how = "none"
        out = {
            "metric": ("beamformed voxels/s of the reference harness's 512 x 1024 view plane (tests/throughput.c) -- a profile line, not BASELINE's metric" if harness
                       else "beamformed voxels/s (and %HBM-roofline), 256-ch 3D DAS 512^3, 1/2/4/8 GPUs"),
            "value": value, "unit": "voxels/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": (f"reference harness frame {args.config} (tests/throughput.c:443-491; NOT the metric's workload)" if harness else f"BASELINE configs[{args.config - 1}]") +
                            f" ({acq.name}): {P.AcquisitionKind(bp.acquisition_kind).name} "
                            f"{bp.channel_count} ch x {bp.acquisition_count} tx, {P.DataKind(bp.data_kind).name} RF {bp.sample_count} samples"
                            f" -> {' -> '.join(P.ShaderKind(v).name for v in bp.compute_stages[:bp.compute_stages_count])}"
                            f"{' + coherency weighting' if bp.coherency_weighting else ''} -> {X}x{Y}x{Z} "
                            f"{'complex' if voxel_bytes == 8 else 'real'} voxels",
                "notes": ((acq.notes + "; ") if acq.notes else "") +
                         ("every frame replayed from a captured hipGraph (--frame-graph): the whole frame times as one segment, reported as the DAS kernel"
                          if args.frame_graph else
                          "every stage is launched directly on a HIP stream, not replayed from a captured hipGraph (available: --frame-graph / "
                          "beamformer_hip_enable_frame_graphs): a frame is <= 6 launches and replay measured no faster (profiles/r02_graph_probe.json, "
                          "profiles/r02_other_configs.json, DESIGN.md section 8)"),
                "scale": args.scale, "interpolation": P.InterpolationMode(bp.interpolation_mode).name,
                "f_number": bp.f_number, "sharding": how,
                "upload_inclusive": upload,
                "das_path": PATH_NAMES[das_path], "das_plan": das_plan, "slab_verified": verified, "rf_checksum_equal_on_all_ranks": rf_checksum_ok,
                "stage_ms": stage_ms, **({"device_das_ms": device_das_ms, "devices": devices_info} if device_das_ms else {}),
            },
            "roofline": {
                # what limits the dominant kernel (`binding` prices it): VALU issue for the LDS-staged kernel, the per-CU texture-address path for
                # the gather kernels.  achieved / peak / frac keep the contract's HBM formula (see contract_note)
                "bound": {"das_rca_staged_kernel": "valu-issue", "das_hercules_kernel": "valu-issue", "das_tile_kernel": "valu-issue"}.get(KERNEL_NAMES[das_path], "texture-address path (per-lane gathers served by L1)"),
                "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved * 1e9 / HBM_PEAK,
                "contract_note": "achieved / peak / frac follow the contract's formula: ALGORITHMIC gather bytes / kernel time / 8 TB/s.  The quotient exceeds 1 "
                                 "because the taps are served by LDS / L1 / L2, not HBM (hbm_measured_GBps is what reaches the HBM side): HBM is not the binding "
                                 "resource -- `bound` names it and `binding` prices the kernel against it",
                "traffic": traffic, "traffic_source": traffic_source,
                "hbm_measured_GBps": (traffic / das_s / 1e9) if traffic else None,
                "hbm_copy_measured": hbm_copy,        # GB/s (read + write) of a 1 GiB device-to-device copy on this box
                "kernel": KERNEL_NAMES[das_path], "kernel_ms": das_s * 1e3,
                "algorithmic_bytes_per_launch": bytes_alg,
                "pairs_per_launch": pairs_local, "pairs_total": pairs_total,
                "model": "G*taps*sizeof(sample) + V*(sizeof(voxel)+4 with CW); logical gather bytes, "
                         "compulsory HBM traffic is ~1e4x smaller (BASELINE.md section 4)",
                "binding": binding_resource(args.config, KERNEL_NAMES[das_path], gather_bytes, taps * sample_bytes, das_s, pairs_local, das_plan, cw=bool(bp.coherency_weighting)),
            },
        }
        if not args.no_cpu_baseline and n_gpus == 1:
            out["cpu_baseline"] = cpu_baseline(acq, args.cpu_seconds)
        assert out["n_gpus"] == (args.gpus if (args.gpus_given or not in_process) else len(ordinals)), (out["n_gpus"], args.gpus)
        print(json.dumps(out), flush=True)

    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def kernel_source_hash(kernel):
    """sha256 over the sources of one DAS kernel (its file + the common headers) with comments and whitespace removed: a
    committed PMC figure stays valid through edits that cannot change that kernel's code object, and is refused after any
    that can."""
    h = hashlib.sha256()
    for name in [KERNEL_FILES.get(kernel, "das.hip")] + COMMON_SOURCES:
        p = os.path.join(ROOT, "ogl_beamforming_amd", "csrc", name)
        if os.path.exists(p):
            text = open(p, encoding="utf-8", errors="replace").read()
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
            text = re.sub(r"//[^\n]*", " ", text)
            h.update(name.encode())
            h.update(re.sub(r"\s+", " ", text).strip().encode())
    return h.hexdigest()[:16]


def measured_traffic(config, kernel):
    """HBM-side bytes per DAS launch of the whole metric frame from the committed rocprofv3 PMC passes
    (profiles/das_traffic.json, written by tools/summarize_profiles.py from tools/pmc_das.py runs):
    (bytes, source) -- bytes is None when the committed figure was taken from other kernel sources."""
    path = os.path.join(ROOT, "profiles", "das_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
        entry = table[f"config{config}" if isinstance(config, int) else config][kernel]
    except (OSError, KeyError, ValueError):
        return None, "no committed PMC pass for this configuration and kernel"
    if entry.get("kernel_source_sha16") != kernel_source_hash(kernel):
        return None, f"profiles/das_traffic.json is stale: taken at kernel sources {entry.get('kernel_source_sha16')}, now {kernel_source_hash(kernel)}"
    return entry["hbm_bytes_per_launch"], (f"profiles/das_traffic.json ({entry.get('round', '?')}): rocprofv3 --pmc FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE of "
                                          f"`{entry.get('command', 'bench.py')}`, kernel sources {entry.get('kernel_source_sha16')}")


def newest_profile(suffix):
    """the newest round's profiles/rNN_<suffix> (the round's profile pass writes them: tools/profile_round.sh)"""
    import glob
    names = sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    if not names:
        raise OSError(suffix)
    return names[-1]


def binding_resource(config, kernel, gather_bytes, bytes_per_gather, das_s, terms, das_plan=None, cw=True):
    """The physical resource that limits the DAS launch, priced against measured ceilings (tools/microbench.hip):
      * gather kernels -- the per-CU vector-memory (texture-address) path: a wave64 per-lane gather instruction of up
        to 16 bytes per lane takes 16 clocks per CU whatever its width (pattern "das_like", window resident in L1);
      * the LDS-staged kernel -- VALU issue: the inner loop's VALU stream (36 instructions per 4 (voxel, channel,
        transmit) terms), measured as a stream with no memory instruction in it ("valu_stream", shipping form), and
        next to it the same loop with its LDS reads ("inner_loop_with_lds": what the instruction mix allows);
    achieved is measured live in this run; the VALU / LDS busy fractions next to it come from the committed PMC pass."""
    out = {}
    try:
        micro_name = newest_profile("microbench.json")
        with open(os.path.join(ROOT, "profiles", micro_name)) as f:
            micro = json.load(f)
        cus = int(micro["compute_units"])
        if kernel == "das_hercules_kernel" and micro.get("hercules_stream") and bytes_per_gather >= 16:
            # VALU issue, priced against the kernel's own instruction stream run with no memory instruction (tools/microbench.hip
            # hercules_probe: the expressions of das_hercules.hip's unchecked IQ loop, 4 pairs per iteration)
            cubic = bytes_per_gather >= 32
            rows = [v for v in micro["hercules_stream"] if ("cubic" in v["stream"]) == cubic]
            # the stream that ran: prepared copy or raw taps (coarse grids), with or without the |s| sum of coherency weighting -- where the
            # microbenchmark file has those rows (round 4 on); else the lighter prepared stream, and `stream_note` below says so
            raw = bool(cubic and das_plan is not None and not das_plan.get("hercules_prepared_copy"))
            exact = [v for v in rows if v.get("taps") == ("raw" if raw else "prepared") and v.get("coherency_weighting") == bool(cw)]
            matched = bool(exact)
            rows = exact or [v for v in rows if v.get("coherency_weighting", True)] or rows
            best = min(rows, key=lambda v: v["cycles_per_pair_per_simd_wall"])
            per_pair = best["cycles_per_pair_per_simd_wall"]
            peak = cus * 4 * 64 * best["clock_ghz"] * 1e9 / per_pair
            out = {
                "resource": f"VALU issue: {per_pair:.1f} clk per wave64 (channel, transmit-element) pair per SIMD for the HERCULES inner loop's instruction stream, measured with no memory instruction in it",
                "achieved": terms / das_s / 1e12, "unit": "T pairs/s (pairs that pass the f-number test)", "peak": peak / 1e12,
                "peak_model": f"{cus} CUs x 4 SIMDs x 64 lanes x {best['clock_ghz']:.3f} GHz (clock sustained in that probe) / {per_pair:.2f} clk",
                "frac": terms / das_s / peak, "probe_clock_ghz": best["clock_ghz"], "peak_cycles_per_pair_per_simd": per_pair,
                "source": f"profiles/{micro_name} hercules_stream (tools/microbench.hip hercules_probe); achieved measured in this run",
            }
            out["stream"] = best["stream"]
            if matched:
                out["stream_note"] = "the probe's stream is this frame's (same interpolation form and coherency-weighting setting); pairs that fail the f-number test execute and are not counted in `achieved`"
            elif cubic and das_plan is not None and not das_plan.get("hercules_prepared_copy"):
                out["stream_note"] = ("the probe's cubic stream evaluates the PREPARED segment polynomial (three packed fmas per pair); this frame is on a coarse grid, gathers the "
                                      "four raw taps with two gather instructions per pair and builds the Catmull-Rom cubic per pair (nine more packed operations), and a share "
                                      "of the executed pairs fails the f-number test: `frac` is measured against a lighter stream than the one that ran and counts passing pairs only")
        elif kernel == "das_tile_kernel" and config == 2:
            # VALU issue.  NOT an independent ceiling: the denominator is the VALU-busy time per executed wave64 term that the committed PMC pass
            # counted for THIS kernel on THIS frame (SQ_ACTIVE_INST_VALU), so `frac` restates the kernel's VALU utilisation (passing terms
            # over executed ones included) -- labelled so (`frac_is`); other frames that take das_tile.hip are left unpriced.  Next to it the LDS
            # read path: two ds_read_b128 per term and lane = 2 KiB per wave64 term at 128 B/clk/CU
            clock = max(g["clock_ghz"] for g in micro["gather"])
            tile_name = newest_profile("pmc_tile_cfg2.json")
            with open(os.path.join(ROOT, "profiles", tile_name)) as f:
                pmc = json.load(f)
            per_term = pmc["valu_busy_cycles_per_wave_term_per_simd"]
            peak = cus * 4 * 64 * clock * 1e9 / per_term
            lds_clk = 16.0
            lds_peak = cus * 64 * clock * 1e9 / lds_clk
            out = {
                "resource": f"VALU issue: {per_term:.1f} clk of VALU-busy time per wave64 (voxel, channel, transmit) term per SIMD "
                            f"({pmc['valu_instructions_per_wave_term']:.1f} instructions: transmit delay and phase, window position, cubic Horner, rotate-accumulate, "
                            "plus the staging, reductions and fold around the loop)",
                "achieved": terms / das_s / 1e12, "unit": "T terms/s (terms that pass the f-number test)", "peak": peak / 1e12,
                "peak_model": f"{cus} CUs x 4 SIMDs x 64 lanes x {clock:.3f} GHz / {per_term:.1f} clk (every SIMD issuing nothing but this kernel's VALU stream)",
                "frac": terms / das_s / peak, "probe_clock_ghz": clock, "peak_cycles_per_term_per_simd": per_term,
                "frac_is": "VALU utilisation of this kernel on this frame restated (its own PMC-counted VALU-busy time per term is the denominator), not a fraction of an independent ceiling",
                "source": f"profiles/{tile_name} (tools/pmc_das.py, SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU over config 2's launch); achieved measured in this run",
                "lds_read_path": {"cycles_per_term_per_cu": lds_clk, "peak": lds_peak / 1e12, "frac": terms / das_s / lds_peak,
                                  "model": "two ds_read_b128 per term and lane = 2 KiB per wave64 term at 128 B/clk/CU: the floor under the gathers' 32.6 clk "
                                           "(two wave64 gather instructions through L1), which is what das_factored.hip waits for on the same frame"},
            }
        elif kernel == "das_tile_kernel":
            out = {"resource": "VALU issue (das_tile.hip); unpriced on this frame: the kernel has no independent stream probe, and its PMC-derived utilisation "
                               "figure is committed for BASELINE config 2 only"}
        elif kernel == "das_rca_staged_kernel":
            best = min((v for v in micro["valu_stream"] if v["waves_per_simd"] == 8 and v["stream"].startswith("das_staged term, shipping form")),
                       key=lambda v: v["cycles_per_term_per_simd_wall"])
            per_term = best["cycles_per_term_per_simd_wall"]
            peak = cus * 4 * 64 * best["clock_ghz"] * 1e9 / per_term          # terms per second with every SIMD issuing nothing else
            out = {
                "resource": f"VALU issue: {per_term:.1f} clk per wave64 term per SIMD for the inner loop's 36 VALU instructions per 4 terms, measured as a stream with no memory instruction in it",
                "achieved": terms / das_s / 1e12, "unit": "T terms/s (terms that pass the f-number test)",
                "peak": peak / 1e12,
                "peak_model": f"{cus} CUs x 4 SIMDs x 64 lanes x {best['clock_ghz']:.3f} GHz (clock sustained in that probe) / {per_term:.2f} clk",
                "frac": terms / das_s / peak, "probe_clock_ghz": best["clock_ghz"], "peak_cycles_per_term_per_simd": per_term,
                "source": f"profiles/{micro_name} valu_stream (tools/microbench.hip term_probe); achieved measured in this run",
            }
            # the datasheet figure next to the measured one: wave64 issue costs of MI355X_MICROARCH.md (plain f32 2 clk, packed f32 and
            # conversions 4, transcendentals 8) for the loop's 36 instructions per 4 terms: 4 packed adds + 12 packed fmas + 2 packed adds
            # (72) + 4 v_lshlrev_b16 (8) + 4 x (v_mul, v_fmac, v_sqrt) (48) = 128 clk per 4 terms
            issue_clk = 32.0
            issue_peak = cus * 4 * 64 * best["clock_ghz"] * 1e9 / issue_clk
            out["issue_bound"] = {"cycles_per_term_per_simd": issue_clk, "peak": issue_peak / 1e12, "frac": terms / das_s / issue_peak,
                                  "model": "datasheet issue rates (2 / 4 / 8 clk per wave64 instruction: plain, packed or converting, transcendental) x the inner loop's "
                                           "36 VALU instructions per 4 terms = 32 clk per term per SIMD, at the probe's clock: the ceiling if the hardware met its "
                                           "datasheet rates; `frac` above prices against what the same instruction stream MEASURED"}
            # the same loop WITH its LDS reads (delays, phasors, taps at addresses formed as the kernel forms them) and
            # nothing else of the kernel -- no staging, no barriers: what the instruction mix of the inner loop allows
            loops = [v for v in micro["valu_stream"] if v["stream"].startswith("das_staged inner loop with its LDS reads: address by v_lshlrev_b16")]
            if loops:
                lp = min(loops, key=lambda v: v["cycles_per_term_per_simd_wall"])
                lp_peak = cus * 4 * 64 * lp["clock_ghz"] * 1e9 / lp["cycles_per_term_per_simd_wall"]
                out["inner_loop_with_lds"] = {
                    "what": "tools/microbench.hip loop_probe: the inner loop's VALU stream and its LDS reads at config 4's shape, 8 waves per SIMD, no staging or barriers",
                    "cycles_per_term_per_simd": lp["cycles_per_term_per_simd_wall"], "probe_clock_ghz": lp["clock_ghz"],
                    "peak": lp_peak / 1e12, "frac": terms / das_s / lp_peak}
        else:
            inst = "global_load_dwordx4" if bytes_per_gather >= 16 else "global_load_dwordx2" if bytes_per_gather == 8 else "global_load_dword"
            rows = [g for g in micro["gather"] if g["inst"] == inst and g["level"] == "L1" and g["pattern"] == "das_like"]
            best = max(rows, key=lambda g: g["bytes_per_clk_per_cu_wall"])
            # a gather of 2 x 16 bytes per lane (cubic IQ) is two such instructions: same bytes per clock
            peak = best["bytes_per_clk_per_cu_wall"] * cus * best["clock_ghz"] * 1e9
            out = {
                "resource": f"per-CU vector-memory (texture-address) path: {inst} per-lane gathers, {64 * min(bytes_per_gather, 16) / best['bytes_per_clk_per_cu_wall']:.1f} clk per wave64 instruction",
                "achieved": gather_bytes / das_s / 1e9, "unit": "GB/s",
                "peak": peak / 1e9,
                "peak_model": f"{best['bytes_per_clk_per_cu_wall']:.1f} B/clk/CU (measured, L1-resident window, DAS-like addresses, {best['waves_per_simd']} waves/SIMD) x {cus} CUs x {best['clock_ghz']:.3f} GHz (clock sustained in that probe)",
                "frac": gather_bytes / das_s / peak,
                "peak_bytes_per_clk_per_cu": best["bytes_per_clk_per_cu_wall"], "probe_clock_ghz": best["clock_ghz"],
                "source": f"profiles/{micro_name} (tools/microbench.hip); achieved measured in this run",
            }
    except (OSError, KeyError, ValueError, StopIteration):
        out = {"resource": "unpriced: profiles/r0x_microbench.json missing or without the probe this kernel needs"}
    try:
        bound_name = newest_profile("das_bound.json")
        with open(os.path.join(ROOT, "profiles", bound_name)) as f:
            bound = json.load(f)
        entry = bound[f"config{config}" if isinstance(config, int) else config][kernel]
        out["valu_busy_frac"] = entry["valu_busy_frac"]
        out["ta_busy_frac"] = entry.get("ta_busy_frac")
        if entry.get("lds_idx_active_frac") is not None:
            out["lds_idx_active_frac"] = entry["lds_idx_active_frac"]
        # The chip lowers its clock under this kernel (the probe's loop sustains more): cycles the launch was
        # resident (GRBM_GUI_ACTIVE / 8 of the committed whole-frame PMC pass) over this run's kernel time
        # estimate the clock it actually held; against the ceiling at THAT clock the kernel sits higher.
        if entry.get("planes") == "whole frame" and "peak" in out and out.get("frac"):
            clock = entry["kernel_cycles_per_launch"] / das_s / 1e9
            out["sustained_clock_ghz_est"] = clock
            out["frac_at_sustained_clock"] = out["frac"] * out["probe_clock_ghz"] / clock
            if "inner_loop_with_lds" in out:
                out["inner_loop_with_lds"]["frac_at_sustained_clock"] = out["inner_loop_with_lds"]["frac"] * out["inner_loop_with_lds"]["probe_clock_ghz"] / clock
        out["valu_source"] = (f"profiles/{bound_name}: SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x SIMDs) of `{entry.get('command')}`, "
                              f"kernel sources {entry.get('kernel_source_sha16')}" + ("" if entry.get("kernel_source_sha16") == kernel_source_hash(kernel) else " (STALE: sources changed since)"))
        per_cu = (entry.get("per_gather_instruction") or {}).get("elapsed_cycles_per_cu")
        if per_cu and out.get("peak_bytes_per_clk_per_cu"):
            # `frac` counts the bytes of terms that PASS the f-number test; a wave issues the gather when any of its lanes passes.  By
            # executed wave64 gather instructions (the committed PMC pass of this frame: SQ_INSTS_VMEM_RD per CU over the launch's cycles)
            # the same path is this busy -- the difference is lanes the aperture test masks inside active waves
            ceiling = 64.0 * min(bytes_per_gather, 16) / out["peak_bytes_per_clk_per_cu"]
            out["executed_gather_instructions"] = {"elapsed_clk_per_instruction_per_cu": per_cu, "ceiling_clk_per_instruction": ceiling, "frac": ceiling / per_cu,
                                                   "lanes_passing_frac": out["frac"] * per_cu / ceiling if out.get("frac") else None}
        if entry.get("wave_cycles"):
            out["wave_cycles"] = entry["wave_cycles"]
        if kernel != "das_rca_staged_kernel" and entry["valu_busy_frac"] > out.get("frac", 0):
            out["resource_note"] = "VALU issue is the tighter bound for this kernel (valu_busy_frac)"
    except (OSError, KeyError, ValueError, StopIteration):
        pass
    return out


def cpu_quota():
    """CPUs' worth of time the cgroup grants this process (cgroup v2 cpu.max, v1 cfs quota), or None when unlimited / unknown.
    A one-GPU box of the pool shows 256 logical CPUs to nproc and sched_getaffinity but runs under `cpu.max = 1600000 100000`:
    16 CPUs.  Threads beyond the quota are throttled, not parallel -- round 2's 256-thread leg was 16 CPUs time-sliced 16 ways,
    which is why it lost to the 16-thread one."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0:
            return max(1, quota // period)
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline(acq, budget_s):
    """The CPU oracle (port of the reference shaders) on a bounded sample of the same frame: all x,
    a few rows, on EVENLY SPACED z-planes (the f-number culling -- and with it the work per voxel --
    varies with depth); voxels are independent, so voxels/s extrapolates linearly.  Legs as
    BASELINE.md section 3 asks: every CPU the process may USE (affinity capped by the cgroup's CPU quota), half of them,
    one thread -- and, when the quota is below the visible CPU count, one oversubscribed leg at twice the quota to show
    that more threads than the quota buy nothing.  The oracle's DAS hands ROWS of the image to the threads and walks
    the 16-channel chunks inside each (one parallel region; bit-identical to the reference's chunk-by-chunk
    order, tests/test_oracle.py)."""
    from oracle import binding as oracle
    bp = acq.bp
    X, Y, Z = (max(1, v) for v in bp.output_points[:3])
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    per_voxel_pairs = bp.channel_count * bp.acquisition_count

    def sample(threads, seconds):
        # ~1.2e7 pairs/s per thread for the scalar port (hyperthreads share cores: 0.6e7 assumed past 64 threads);
        # the sample is sized for the budget and so that every thread gets several rows (a row of X voxels over all
        # channels is the oracle's unit of parallel work)
        rate = 1.2e7 * min(threads, 64) + 0.6e7 * max(0, threads - 64)
        target_voxels = max(X, int(seconds * rate / per_voxel_pairs))
        planes = max(1, min(Z, 8, target_voxels // X))
        rows = max(1, min(Y, target_voxels // (X * planes)))
        if rows * planes < 3 * threads:                       # fewer rows than threads would idle most of them
            rows = max(1, min(Y, -(-3 * threads // planes)))
        z_stride = max(1, Z // planes)
        y_stride = max(1, Y // rows)
        timing = {}
        t0 = time.perf_counter()
        _, pairs = oracle.beamform(bp, acq.rf, acq.filters, threads=threads, z=(z_stride // 2, planes), y=(y_stride // 2, rows),
                                   stride=(z_stride, y_stride), timing=timing)
        wall = time.perf_counter() - t0
        das_s = timing["das_seconds"]
        return {"value": X * rows * planes / das_s, "unit": "voxels/s", "cores": threads,
                "sample": (f"oracle DAS over {X}x{rows}x{planes} voxels ({planes} evenly spaced z-planes from {z_stride // 2} step {z_stride}, {rows} evenly spaced rows each) "
                           f"of the {X}x{Y}x{Z} frame, {pairs} pairs in {das_s:.2f} s DAS time ({wall:.2f} s incl. the single-threaded pre-DAS stages over the whole RF)")}

    quota = cpu_quota()
    usable = max(1, min(affinity, quota or affinity, int(os.environ.get("BENCH_CPU_THREADS", str(affinity)))))
    counts = sorted({usable, max(1, usable // 2), 1}, reverse=True)
    oversubscribed = min(affinity, 2 * usable) if usable < affinity else None
    weight = {n: (0.4 if n == usable else 0.25) for n in counts}
    if oversubscribed:
        weight[oversubscribed] = 0.2
    total = sum(weight.values())
    legs = {n: sample(n, budget_s * weight[n] / total) for n in weight}
    best = max((legs[n] for n in counts), key=lambda leg: leg["value"])
    # the headline figure is the fastest leg within the quota; every leg is listed
    out = {
        "value": best["value"], "unit": "voxels/s", "cores": best["cores"], "kind": "port", "sample": best["sample"],
        "all_usable_cpus": legs[usable], "half_of_them": legs[max(1, usable // 2)], "one_thread": legs[1],
        "nproc": os.cpu_count(), "affinity": affinity, "cgroup_cpu_quota": quota,
    }
    if oversubscribed:
        out["twice_the_quota"] = legs[oversubscribed]
    return out


if __name__ == "__main__":
    main()
