#!/usr/bin/env python3
"""bench.py -- beamformed voxels/s of the DAS hot path on MI355X (BASELINE.json metric).

A step is one whole frame of BASELINE.json configs[3] through the C ABI: Int16 RF resident
in HBM -> Demodulate -> DAS with fused coherency weighting -> 512^3 complex voxels
(3-D row-column array, 256 channels x 75 plane waves).
With N > 1 (one process per GPU, launched by torch.distributed.run) the volume is the SAME
512^3 grid cut into N z-slabs (strong scaling): rank 0 holds the RF frame, every step
broadcasts it over RCCL/xGMI (the one collective of the path), every rank beamforms its
slab.  No reduction collective.

Prints ONE JSON line (rank 0).  `roofline` prices the DAS kernel against the HBM roofline
with the ALGORITHMIC gather bytes of BASELINE.md section 4; `cpu_baseline` times the CPU
oracle (a port of the reference shaders, the reference itself cannot be built here) on a
bounded sub-grid of the same frame on the host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8.0e12   # B/s, MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=4, help="BASELINE.json configs index (1-based)")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink factor for rehearsal runs (metric is scale 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU-baseline budget")
    ap.add_argument("--das-path", type=int, default=0, help="0 auto, 1 general DAS kernel, 3 prefer the LDS-staged kernel, 4 factored kernel wherever it applies, 5 LDS row-cache experiment")
    ap.add_argument("--serial-broadcast", action="store_true",
                    help="multi-GPU: broadcast and compute back to back on one stream instead of pipelined")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development aid: run the N-rank code path with every rank on GPU 0 and gloo for the "
                         "collectives (RCCL refuses two ranks per device); the value is NOT the metric")
    ap.add_argument("--planes", type=int, default=0, help="profiling aid: beamform only this many centre z-planes (value is then NOT the metric)")
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    rehearse = args.rehearse_on_one_gpu
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
        torch.cuda.set_device(0)
    device = torch.device("cuda", local_rank if distributed else 0)

    from ogl_beamforming_amd import configs, lib, params as P, sharding
    L = lib.library()
    assert L.beamformer_hip_set_device(device.index), lib.last_error()
    L.beamformer_hip_set_das_path(args.das_path)

    # every rank builds the parameter block (cheap, deterministic); only rank 0 owns the RF
    acq = configs.config(args.config, args.scale)
    bp = acq.bp
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

    rf_host = torch.from_numpy(np.ascontiguousarray(acq.rf).view(np.uint8).reshape(-1))
    rf_dev = torch.empty(rf_host.numel(), dtype=torch.uint8, device=device)
    if rank == 0:
        rf_dev.copy_(rf_host)
    torch.cuda.synchronize(device)
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
        torch.cuda.synchronize(device)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)

    # one untimed frame with the geometry-only pair count: G of BASELINE.md section 4
    L.beamformer_hip_enable_pair_counting(1)
    step()
    fence()
    t = P.HipFrameTimings()
    assert L.beamformer_hip_get_last_frame_timings(C.byref(t)), lib.last_error()
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

    # Multi-GPU runs check what the schedule delivered: the slab of the last timed frame must be
    # bit-identical to a recomputation from the same RF buffer after a full fence (a broadcast
    # that had not landed, or a buffer overwritten early, would show here).  Outside the timed region.
    verified = None
    if distributed:
        last_buf = bufs[(state["computed"] - 1) % 2] if pipelined else rf_dev
        first = lib.get_last_frame(bp, shard_planes=z_count).copy()
        push(last_buf)
        fence()
        again = lib.get_last_frame(bp, shard_planes=z_count)
        verified = bool(np.array_equal(first.view(np.uint32), again.view(np.uint32)))
        assert verified, f"rank {rank}: the pipelined frame differs from its recomputation"

    # DAS kernel duration: HIP event pairs recorded by the library on the stream it launches
    # on (beamformer_compute_timings = the reference's per-stage stats table)
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

    if distributed:
        agg = torch.tensor([elapsed, das_s, float(pairs_local)], dtype=torch.float64, device="cpu" if rehearse else device)
        mx = agg.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = agg.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, das_s_max, pairs_total = float(mx[0]), float(mx[1]), int(sm[2].item())
    else:
        das_s_max, pairs_total = das_s, pairs_local

    # a measured copy figure next to the 8 TB/s nominal peak (1 GiB device-to-device
    # copies: bytes read + bytes written over the event-timed duration)
    hbm_copy = None
    if rank == 0 and not args.planes:
        a_buf = torch.empty(1 << 30, dtype=torch.uint8, device=device)
        b_buf = torch.empty_like(a_buf)
        b_buf.copy_(a_buf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10):
            b_buf.copy_(a_buf)
        e1.record(stream)
        e1.synchronize()
        hbm_copy = 10 * 2 * a_buf.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a_buf, b_buf

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = voxels_total / (elapsed / args.steps)
        voxel_bytes = 8 if int(info.data_kind) == int(P.DataKind.Float32Complex) else 4
        cw = 4 if bp.coherency_weighting else 0
        # per launch = this rank's slab; ranks are symmetric, report rank 0's launch
        bytes_alg = pairs_local * taps * sample_bytes + (X * Y * z_count) * (voxel_bytes + cw)
        achieved = bytes_alg / das_s / 1e9
        out = {
            "metric": "beamformed voxels/s (and %HBM-roofline), 256-ch 3D DAS 512^3, 1/2/4/8 GPUs",
            "value": value, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[{args.config - 1}] ({acq.name}): {P.AcquisitionKind(bp.acquisition_kind).name} "
                            f"{bp.channel_count} ch x {bp.acquisition_count} tx, {P.DataKind(bp.data_kind).name} RF {bp.sample_count} samples"
                            f" -> {' -> '.join(P.ShaderKind(v).name for v in bp.compute_stages[:bp.compute_stages_count])}"
                            f"{' + coherency weighting' if bp.coherency_weighting else ''} -> {X}x{Y}x{Z} "
                            f"{'complex' if voxel_bytes == 8 else 'real'} voxels",
                **({"notes": acq.notes} if acq.notes else {}),
                "scale": args.scale, "interpolation": P.InterpolationMode(bp.interpolation_mode).name,
                "f_number": bp.f_number, "sharding": (f"{world} z-slab(s), RF broadcast via " + ("gloo through the host (one-GPU rehearsal)" if rehearse else "RCCL")
                             + (", broadcast of frame n+1 overlaps frame n" if pipelined else ", serial")) if distributed else "none",
                "das_path": PATH_NAMES[das_path], "slab_verified": verified,
                "stage_ms": stage_ms,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "hbm_copy_measured": hbm_copy,        # GB/s (read + write) of a 1 GiB device-to-device copy on this box
                "frac": achieved * 1e9 / HBM_PEAK, "traffic": measured_traffic(args, world, das_path),
                "kernel": KERNEL_NAMES[das_path], "kernel_ms": das_s * 1e3,
                "algorithmic_bytes_per_launch": bytes_alg,
                "pairs_per_launch": pairs_local, "pairs_total": pairs_total,
                "model": "G*taps*sizeof(sample) + V*(sizeof(voxel)+4 with CW); logical gather bytes, "
                         "compulsory HBM traffic is ~1e4x smaller (BASELINE.md section 4)",
                # what actually binds the kernel: every tap is a per-lane L1 gather and the
                # texture-address path retires 64 B per clock per CU
                "l1_gather": {"achieved": pairs_local * taps * sample_bytes / das_s / 1e9, "unit": "GB/s",
                              "peak": 64 * 256 * 2.4e9 / 1e9, "peak_model": "64 B/clk/CU x 256 CUs x 2.4 GHz",
                              "frac": pairs_local * taps * sample_bytes / das_s / (64 * 256 * 2.4e9)},
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(acq, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    if distributed:
        dist.barrier()
        dist.destroy_process_group()


PATH_NAMES = ["general kernel", "separable-delay gather kernel", "separable-delay LDS-staged kernel", "per-voxel factored kernel",
              "factored kernel with LDS row cache", "HERCULES aligned-grid kernel"]
KERNEL_NAMES = ["das_kernel", "das_rca_separable_kernel", "das_rca_staged_kernel", "das_factored_kernel", "das_rowcache_kernel", "das_hercules_kernel"]


def measured_traffic(args, world, das_path):
    """HBM bytes per DAS launch from the committed rocprofv3 PMC passes of this same command
    (profiles/das_traffic.json); None when this run's configuration was not profiled."""
    if world != 1 or args.scale != 1.0 or args.planes or args.config != 4:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "das_traffic.json")) as f:
            table = json.load(f)["config4_scale1_n1"]
        return table[KERNEL_NAMES[das_path]]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(acq, budget_s):
    """The CPU oracle (port of the reference shaders) on a bounded sub-grid of the same
    frame: all x, a few rows of the centre z-plane; voxels are independent, so voxels/s
    extrapolates linearly."""
    from oracle import binding as oracle
    bp = acq.bp
    X, Y, Z = (max(1, v) for v in bp.output_points[:3])
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    # a one-GPU box owns a 16-core share of the host (gpurun's guidance for worker pools)
    cores = min(cores, int(os.environ.get("BENCH_CPU_THREADS", "16")))
    per_voxel_pairs = bp.channel_count * bp.acquisition_count

    def sample(threads, seconds):
        # ~1.2e7 pairs/s/core for the scalar port; size the sample for the budget
        target_voxels = max(X, int(seconds * 1.2e7 * threads / per_voxel_pairs))
        rows = max(1, min(Y, target_voxels // X))
        timing = {}
        t0 = time.perf_counter()
        _, pairs = oracle.beamform(bp, acq.rf, acq.filters, threads=threads, z=(Z // 2, 1), y=((Y - rows) // 2, rows), timing=timing)
        wall = time.perf_counter() - t0
        das_s = timing["das_seconds"]
        return X * rows / das_s, (f"oracle DAS over {X}x{rows}x1 voxels (z plane {Z // 2}, {rows} centre rows) of the {X}x{Y}x{Z} frame, "
                                  f"{pairs} pairs in {das_s:.2f} s DAS time ({wall:.2f} s incl. the single-threaded pre-DAS stages over the whole RF)")

    value, text = sample(cores, 0.6 * budget_s)
    one, one_text = sample(1, 0.4 * budget_s) if cores > 1 else (value, text)
    return {
        "value": value, "unit": "voxels/s", "cores": cores, "kind": "port", "sample": text,
        "one_thread": {"value": one, "unit": "voxels/s", "sample": one_text},
        "nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
    }


if __name__ == "__main__":
    main()
