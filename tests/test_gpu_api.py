"""Behaviour of the C ABI on a device beyond single-frame parity: frame ring order and
64-byte rounding (beamformer_core.c:440-466, :1474-1494), several parameter blocks, the
one-shot call, data-size validation (lib/ogl_beamformer_lib.c:503-511), device-resident RF,
sharding, the per-stage stats table (beamformer_compute_stats.c:3-10)."""
import ctypes as C

import numpy as np
import pytest

from ogl_beamforming_amd import params as P
from tests import cases

pytestmark = pytest.mark.gpu
E = P.LibError


def push(L, bflib, acq, slot=0):
    for s, fp in enumerate(acq.filters):
        assert L.beamformer_create_filter(C.byref(fp), s, slot), bflib.last_error()
    assert L.beamformer_push_simple_parameters_at(C.byref(acq.bp), slot), bflib.last_error()
    rf = np.ascontiguousarray(acq.rf)
    assert L.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, slot), bflib.last_error()


def test_last_frames_are_returned_oldest_first_and_rounded_to_64_bytes(bflib, oracle):
    L = bflib.library()
    L.beamformer_set_global_timeout(0xFFFFFFFF)
    a = cases.make("rca_cubic_real")          # 20 x 20 x 1 float32 = 1600 B -> 1600 (multiple of 64)
    b = cases.make("hercules_real")           # 10 x 12 x 14 float32 = 6720 B -> 6720
    c = cases.make("forces")                  # 20 x 1 x 20 float32 = 1600 B
    refs = [oracle.beamform(x.bp, x.rf, x.filters)[0] for x in (a, b, c)]
    for x in (a, b, c):
        push(L, bflib, x)
    sizes = [(r.size * 4 + 63) // 64 * 64 for r in refs]
    out = np.zeros(sum(sizes) // 4, np.float32)
    assert L.beamformer_get_last_frames(out.ctypes.data_as(C.c_void_p), out.nbytes, 3)
    offset = 0
    for ref, size in zip(refs, sizes):
        got = out[offset // 4: offset // 4 + ref.size].reshape(ref.shape)
        assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()
        offset += size
    # a buffer with room for two frames gets the two OLDEST of the requested three (the most
    # recent are dropped, lib/ogl_beamformer_lib_base.h:100-102)
    out2 = np.zeros((sizes[0] + sizes[1]) // 4, np.float32)
    assert L.beamformer_get_last_frames(out2.ctypes.data_as(C.c_void_p), out2.nbytes, 3)
    assert np.abs(out2[: refs[0].size].reshape(refs[0].shape) - refs[0]).max() <= 1e-4 * np.abs(refs[0]).max()
    # null / empty requests fail without touching the error (lib .c:700)
    assert not L.beamformer_get_last_frames(None, 64, 1)
    assert not L.beamformer_get_last_frames(out.ctypes.data_as(C.c_void_p), out.nbytes, 0)


def test_two_parameter_blocks_keep_their_own_plans(bflib, oracle):
    L = bflib.library()
    L.beamformer_set_global_timeout(0xFFFFFFFF)
    assert L.beamformer_reserve_parameter_blocks(2)
    try:
        a, b = cases.make("config1_small"), cases.make("uforces_sparse")
        ra, rb = oracle.beamform(a.bp, a.rf, a.filters)[0], oracle.beamform(b.bp, b.rf, b.filters)[0]
        push(L, bflib, a, 0)
        push(L, bflib, b, 1)
        rf = np.ascontiguousarray(a.rf)      # block 0 again, parameters untouched
        assert L.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 0)
        got = bflib.get_last_frame(a.bp)
        assert np.abs(got - ra).max() <= 2e-3 * np.abs(ra).max()
        rf = np.ascontiguousarray(b.rf)
        assert L.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 1)
        got = bflib.get_last_frame(b.bp)
        ok = ~np.isnan(rb)
        assert np.array_equal(np.isnan(got), np.isnan(rb))
        assert np.abs(got[ok] - rb[ok]).max() <= 1e-4 * np.abs(rb[ok]).max()
        assert not L.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 2)
        assert bflib.last_error()[0] == E.ParameterBlockUnallocated
    finally:
        L.beamformer_reserve_parameter_blocks(1)


def test_one_shot_beamform_data(bflib, oracle):
    L = bflib.library()
    acq = cases.make("config2_small")
    assert L.beamformer_create_filter(C.byref(acq.filters[0]), 0, 0)
    ref, _ = oracle.beamform(acq.bp, acq.rf, acq.filters)
    out = np.zeros(ref.shape, np.complex64)
    rf = np.ascontiguousarray(acq.rf)
    assert L.beamformer_beamform_data(C.byref(acq.bp), rf.ctypes.data_as(C.c_void_p), rf.nbytes,
                                      out.ctypes.data_as(C.c_void_p), 60000), bflib.last_error()
    assert np.abs(out - ref).max() <= 2e-3 * np.abs(ref).max()


def test_push_data_validation(bflib):
    L = bflib.library()
    acq = cases.make("config1_small")
    push(L, bflib, acq)
    rf = np.ascontiguousarray(acq.rf)
    ptr = rf.ctypes.data_as(C.c_void_p)
    assert not L.beamformer_push_data_with_compute(ptr, rf.nbytes, 7, 0) and bflib.last_error()[0] == E.InvalidImagePlane
    assert not L.beamformer_push_data_with_compute(ptr, rf.nbytes - 2, 0, 0) and bflib.last_error()[0] == E.DataSizeMismatch
    assert not L.beamformer_push_data_with_compute(ptr, rf.nbytes + 2, 0, 0) and bflib.last_error()[0] == E.DataSizeMismatch
    # a channel mapping that names a raw row that does not exist would read out of bounds
    bad = (C.c_int16 * 256)(*([acq.bp.raw_data_dimensions[1]] * 256))
    assert L.beamformer_push_channel_mapping(bad, acq.bp.channel_count)
    assert not L.beamformer_push_data_with_compute(ptr, rf.nbytes, 0, 0) and bflib.last_error()[0] == E.DataSizeMismatch
    assert L.beamformer_maximum_rf_data_size() == (4 << 30) // 3


def test_device_resident_rf_shards_and_stats(bflib, oracle):
    import torch
    L = bflib.library()
    L.beamformer_set_global_timeout(0xFFFFFFFF)
    acq = cases.make("config4_small")
    ref, _ = oracle.beamform(acq.bp, acq.rf, acq.filters)
    push(L, bflib, acq)                                  # parameters + one host-pushed frame
    whole = bflib.get_last_frame(acq.bp)
    rf = torch.from_numpy(np.ascontiguousarray(acq.rf).view(np.uint8).reshape(-1)).cuda()
    torch.cuda.synchronize()
    Z = acq.bp.output_points[2]
    parts = []
    for first, count in ((0, Z // 3), (Z // 3, Z - Z // 3)):
        assert L.beamformer_hip_set_output_shard(0, first, count)
        assert L.beamformer_hip_push_device_data_with_compute(C.c_void_p(rf.data_ptr()), rf.numel(), 0, 0), bflib.last_error()
        parts.append(bflib.get_last_frame(acq.bp, shard_planes=count))
        info = P.HipFrameInfo()
        assert L.beamformer_hip_get_last_frame_info(C.byref(info)) and info.points[2] == count
    assert L.beamformer_hip_set_output_shard(0, 0, 0)
    assert not L.beamformer_hip_set_output_shard(0, Z - 1, 2) and bflib.last_error()[0] == E.FrameSizeOverflow
    stitched = np.concatenate(parts, axis=0)
    assert np.array_equal(stitched, whole, equal_nan=True)           # bit identical to the unsharded frame
    ok = ~np.isnan(ref)
    assert np.abs(whole[ok] - ref[ok]).max() <= 2e-3 * np.abs(ref[ok]).max()

    stats = P.ComputeStatsTable()
    assert L.beamformer_compute_timings(C.byref(stats), -1)
    ids = [stats.shader_ids[i] for i in range(stats.shader_count)]
    assert ids == [int(P.ShaderKind.Demodulate), int(P.ShaderKind.DAS), int(P.ShaderKind.CoherencyWeighting)]
    info = P.HipFrameInfo()
    L.beamformer_hip_get_last_frame_info(C.byref(info))
    row = stats.times[info.frame_id % 32]
    assert row[1] > 0 and row[0] > 0                                  # seconds per stage of the newest frame
    t = P.HipFrameTimings()
    assert L.beamformer_hip_get_last_frame_timings(C.byref(t)) and t.frame_ms > 0 and t.das_path == 1


def test_sum_of_last_frames_matches_the_sum_stage(bflib, oracle):
    L = bflib.library()
    L.beamformer_set_global_timeout(0xFFFFFFFF)
    acq = cases.make("config2_small")
    push(L, bflib, acq)
    frames = [bflib.get_last_frame(acq.bp).copy()]
    rng = np.random.default_rng(5)
    for _ in range(2):                                    # three different frames of one geometry
        rf = np.ascontiguousarray((acq.rf.astype(np.float32) * rng.uniform(0.5, 2.0)
                                   + rng.standard_normal(acq.rf.shape).astype(np.float32)).astype(acq.rf.dtype))
        assert L.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 0), bflib.last_error()
        frames.append(bflib.get_last_frame(acq.bp).copy())
    for count in (1, 2, 3):
        ref = oracle.sum_frames(frames[-count:])
        out = np.zeros((frames[0].nbytes + 63) // 64 * 16, np.float32)
        assert L.beamformer_hip_sum_last_frames(count, out.ctypes.data_as(C.c_void_p), out.nbytes), bflib.last_error()
        got = out[: ref.size * 2].view(np.complex64).reshape(ref.shape)
        assert np.array_equal(got, ref)                   # same additions in the same order: bit exact
    mm = (C.c_float * 2)()
    assert L.beamformer_hip_frame_min_max(mm)
    assert np.array_equal(np.array(mm[:], np.float32), oracle.min_max(frames[-1]))
    # frames of another size cannot be averaged with these; undersized output is refused
    other = cases.make("hercules_real")
    push(L, bflib, other)
    out = np.zeros(1 << 20, np.float32)
    assert not L.beamformer_hip_sum_last_frames(2, out.ctypes.data_as(C.c_void_p), out.nbytes)
    assert bflib.last_error()[0] == E.DataSizeMismatch
    assert not L.beamformer_hip_sum_last_frames(1, out.ctypes.data_as(C.c_void_p), 16)
    assert bflib.last_error()[0] == E.ExportSpaceOverflow
    assert not L.beamformer_hip_sum_last_frames(0, out.ctypes.data_as(C.c_void_p), out.nbytes)


def test_pipelined_host_pushes_keep_their_own_data(bflib):
    """Host pushes go through three pinned slots and a copy stream; nine back-to-back pushes
    of alternating RF (x1, x2: exact in every stage) must come out as F, 2F, F, ... with no
    slot read before its upload landed or overwritten before its frame finished."""
    L = bflib.library()
    L.beamformer_set_global_timeout(0xFFFFFFFF)
    for name in ("config1_small", "rca_shuffled_padded"):      # direct copy / ingest-kernel path
        acq = cases.make(name)
        push(L, bflib, acq)
        base = bflib.get_last_frame(acq.bp).copy()
        assert np.abs(acq.rf).max() < 16000
        rfs = [np.ascontiguousarray(acq.rf), np.ascontiguousarray(acq.rf * 2)]
        for n in range(9):
            rf = rfs[n & 1]
            assert L.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 0), bflib.last_error()
        size = (base.nbytes + 63) // 64 * 64
        out = np.zeros(4 * size // 4, np.float32)
        assert L.beamformer_get_last_frames(out.ctypes.data_as(C.c_void_p), out.nbytes, 4)
        for k in range(4):                                      # frames 5..8 -> x2, x1, x2, x1
            got = out[k * size // 4: k * size // 4 + base.size * (2 if np.iscomplexobj(base) else 1)]
            got = got.view(base.dtype).reshape(base.shape)
            scale = 2 if (5 + k) & 1 else 1
            assert np.array_equal(got, base * scale, equal_nan=True), (name, k)


def test_display_reduction_matches_the_render_shader_maths(bflib, oracle):
    L = bflib.library()
    L.beamformer_set_global_timeout(0xFFFFFFFF)
    for name in ("config2_small", "hercules_real"):
        acq = cases.make(name)
        push(L, bflib, acq)
        frame = bflib.get_last_frame(acq.bp)
        peak_db = 20 * np.log10(np.nanmax(np.abs(frame)))
        for threshold, gamma, cutoff in ((peak_db - 6, 1.0, 0.0), (peak_db, 0.5, 0.0), (peak_db - 3, 1.0, 50.0), (55.0, 1.0, 50.0)):
            ref = oracle.display(frame, threshold, gamma, cutoff)
            out = np.zeros(frame.size, np.float32)
            assert L.beamformer_hip_display_last_frame(threshold, gamma, cutoff, out.ctypes.data_as(C.POINTER(C.c_float)), out.size)
            got = out.reshape(frame.shape)
            ok = ~np.isnan(ref)
            assert np.array_equal(np.isnan(got), np.isnan(ref))
            assert got[ok].min() >= 0 and got[ok].max() <= 1
            assert np.abs(got[ok] - ref[ok]).max() <= 2e-6, (name, threshold, gamma, cutoff)   # pow/log differ by ulps
        assert not L.beamformer_hip_display_last_frame(55.0, 1.0, 0.0, out.ctypes.data_as(C.POINTER(C.c_float)), frame.size - 1)
        assert bflib.last_error()[0] == E.ExportSpaceOverflow


def test_decode_benchmark_harness_runs(tmp_path):
    """ogl_beamformer_decode_bench (csrc/decode_bench.c, C11): the reference's tests/decode.c study
    -- Decode-only pipelines for Hadamard orders 2..256 -- end to end through the C ABI."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ogl_beamforming_amd", "ogl_beamformer_decode_bench")
    r = subprocess.run([exe, "--warmup", "2", "--dump", str(tmp_path / "stats")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("decode")]
    assert [int(l.split()[1]) for l in lines] == [2, 4, 8, 12, 16, 20, 24, 32, 40, 48, 64, 80, 96, 128, 160, 192, 256]
    for l in lines:
        assert float(l.split("Decode kernel")[1].split()[0]) > 0            # the stage ran and was timed
    table = np.fromfile(tmp_path / "stats" / "decode_256.bin", dtype=np.uint8)
    assert table.size == 2248                                              # sizeof(BeamformerComputeStatsTable)


@pytest.mark.parametrize("name", ["config4_small", "config5_small", "config2_small", "config1_small", "forces"])
def test_frame_graphs_replay_bit_identical_frames(name, bflib):
    """beamformer_hip_enable_frame_graphs (BASELINE configs[4]: hipGraph-captured frame): eight pipelined host
    pushes with changing RF -- the frame-ring slot and the RF slot move every frame -- replayed from ONE
    instantiated graph updated in place, bit-identical to direct launches; a parameter change in between
    (replan) is picked up; the stats table still reports the frame."""
    L = bflib.library()
    L.beamformer_set_global_timeout(0xFFFFFFFF)          # (the library starts with the reference's 0 = try once: whichever test ran before)
    acq = cases.make(name)
    rng = np.random.default_rng(5)
    rfs = [np.ascontiguousarray(acq.rf)]
    for _ in range(3):
        noise = rng.normal(0, 1, acq.rf.shape)
        rfs.append(np.ascontiguousarray((acq.rf + (50 * noise if acq.rf.dtype == np.int16 else 0.05 * noise)).astype(acq.rf.dtype)))

    def run(graphs, f_number):
        bp = P.SimpleParameters.from_buffer_copy(bytes(acq.bp))
        bp.f_number = f_number
        L.beamformer_hip_enable_frame_graphs(1 if graphs else 0)
        try:
            for slot, fp in enumerate(acq.filters):
                assert L.beamformer_create_filter(C.byref(fp), slot, 0)
            assert L.beamformer_push_simple_parameters(C.byref(bp))
            out = []
            for k in range(8):
                rf = rfs[k % len(rfs)]
                assert L.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 0), bflib.last_error()
                if k in (2, 5, 7):
                    out.append(bflib.get_last_frame(bp).copy())
            stats = P.ComputeStatsTable()
            assert L.beamformer_compute_timings(C.byref(stats), -1)
            return out, stats
        finally:
            L.beamformer_hip_enable_frame_graphs(0)

    replayed0, built0 = C.c_uint64(), C.c_uint64()
    L.beamformer_hip_frame_graph_counts(C.byref(replayed0), C.byref(built0))
    for f_number in (acq.bp.f_number, acq.bp.f_number * 1.5):            # the second value forces a replan
        direct, _ = run(False, f_number)
        graphed, stats = run(True, f_number)
        for a, b in zip(direct, graphed):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        ids = [int(stats.shader_ids[i]) for i in range(int(stats.shader_count))]
        assert int(P.ShaderKind.DAS) in ids
    replayed, built = C.c_uint64(), C.c_uint64()
    L.beamformer_hip_frame_graph_counts(C.byref(replayed), C.byref(built))
    assert replayed.value - replayed0.value >= 2 * 6                    # all but each plan's first frame(s)
    assert 1 <= built.value - built0.value <= 4                         # one graph per plan, updated in place


def test_a_push_that_fails_leaves_no_stale_last_frame(bflib):
    """A push that fails after it has taken its frame id (here: the plan is refused at commit time) leaves a tombstone under that id.
    Every reader of "the newest frame" -- get_last_frames, the frame info, the timings, min/max, the rolling sum, the display reduction --
    then FAILS instead of serving the record that sat in the same ring slot BeamformerMaxBacklogFrames pushes ago (or returning
    success with the caller's buffer unwritten); the next good push is served normally and older good frames stay exportable."""
    from tests.test_hilbert import acquisitions
    L = bflib.library()
    L.beamformer_set_global_timeout(0xFFFFFFFF)
    good = cases.make("rca_cubic_real")
    first = bflib.beamform(good.bp, good.rf, good.filters).copy()
    # a pipeline the library accepts while the Hilbert stage is enabled and can no longer plan once it is not
    bad = acquisitions()["rca_i16"]
    assert L.beamformer_hip_enable_hilbert(1)
    try:
        assert L.beamformer_push_simple_parameters(C.byref(bad.bp)), bflib.last_error()
    finally:
        assert L.beamformer_hip_enable_hilbert(0)
    rf = np.ascontiguousarray(bad.rf)
    assert not L.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 0)
    assert bflib.last_error()[0] == E.InvalidComputeStage
    sentinel = np.full(first.size + 64, -7.0, np.float32)
    assert not L.beamformer_get_last_frames(sentinel.ctypes.data_as(C.c_void_p), sentinel.nbytes, 1)
    assert bflib.last_error()[0] == E.InvalidAccess and (sentinel == -7.0).all()
    assert not L.beamformer_hip_get_last_frame_info(C.byref(P.HipFrameInfo()))
    assert not L.beamformer_hip_get_last_frame_timings(C.byref(P.HipFrameTimings()))
    assert not L.beamformer_hip_frame_min_max((C.c_float * 2)())
    assert not L.beamformer_hip_sum_last_frames(1, sentinel.ctypes.data_as(C.c_void_p), sentinel.nbytes)
    assert not L.beamformer_hip_display_last_frame(55.0, 1.0, 50.0, sentinel.ctypes.data_as(C.POINTER(C.c_float)), sentinel.size)
    assert (sentinel == -7.0).all()
    # the last TWO frames: the older, good one is still exported; the call reports the missing newest one
    assert not L.beamformer_get_last_frames(sentinel.ctypes.data_as(C.c_void_p), sentinel.nbytes, 2)
    assert np.array_equal(sentinel[: first.size].reshape(first.shape), first)
    # and the library is not wedged: the next good push is the newest frame again, with its timings
    again = bflib.beamform(good.bp, good.rf, good.filters)
    assert np.array_equal(again.view(np.uint32), first.view(np.uint32))
    t = P.HipFrameTimings()
    assert L.beamformer_hip_get_last_frame_timings(C.byref(t)) and t.stage_count > 0
