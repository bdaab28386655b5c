"""Several devices behind the unchanged push-RF / pull-image calls (beamformer_hip_set_devices,
SURVEY section 8e; the reference's export semantics lib/ogl_beamformer_lib.c:656-702).  A one-GPU box
lists the same ordinal several times: each entry is a separate device context of the library (own
streams, RF ring, plan tables, frame ring) fed by hipMemcpyPeerAsync, so the orchestration is the real
one; only the link the copies travel over differs."""
import ctypes as C

import numpy as np
import pytest

from ogl_beamforming_amd import params as P
from tests import cases

pytestmark = pytest.mark.gpu


def use_devices(lib, ordinals):
    lib.beamformer_hip_shutdown()
    arr = (C.c_int32 * len(ordinals))(*ordinals)
    assert lib.beamformer_hip_set_devices(arr, len(ordinals))
    assert lib.beamformer_hip_get_device_count() == len(ordinals)


@pytest.fixture()
def devices(bflib):
    lib = bflib.library()
    yield lambda ordinals: use_devices(lib, ordinals)
    use_devices(lib, [0])                      # leave the process-wide library as the other tests expect it


def same_bits(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("name, count", [("config4_small", 2), ("config4_small", 3), ("config5_small", 4),
                                         ("rca_vls_cw", 5), ("config2_small", 3), ("hercules_wide_cw", 2),
                                         ("forces", 2)])
def test_slabs_stitch_to_the_one_device_frame(name, count, bflib, devices):
    """N device contexts, plain beamformer_beamform_data-style calls: the pulled frame is bit-identical
    to the one-device frame (also when planes do not divide evenly, and for 2-D images, where all but
    the first device hold an empty slab)."""
    acq = cases.make(name)
    devices([0])
    one = bflib.beamform(acq.bp, acq.rf, acq.filters).copy()
    lib = bflib.library()
    mm_one = (C.c_float * 2)()
    assert lib.beamformer_hip_frame_min_max(mm_one)
    devices([0] * count)
    many = bflib.beamform(acq.bp, acq.rf, acq.filters)
    assert same_bits(one, many)
    # per-device slabs: the cut of ogl_beamforming_amd/sharding.py (what one-process-per-GPU runs use)
    from ogl_beamforming_amd import sharding
    Z = one.shape[0]
    planes = []
    for i in range(count):
        t = P.HipFrameTimings()
        assert lib.beamformer_hip_get_device_frame_timings(i, C.byref(t))
        planes.append(int(t.das_voxels) // (one.shape[1] * one.shape[2]))
    assert planes == [c for _, c in sharding.slabs(count, Z)]
    mm = (C.c_float * 2)()
    assert lib.beamformer_hip_frame_min_max(mm)
    assert np.array_equal(np.array(mm[:]), np.array(mm_one[:]), equal_nan=True)


@pytest.mark.parametrize("name, count", [("rca_sep_ragged_cubic", 3), ("rca_staged_cubic", 2)])
def test_block_staged_kernel_slabs(name, count, bflib, devices):
    """das_tile.hip (das path 5, asked for with flag 0x100) on z-slabs of a volume: every device context runs it on its planes
    (z_first, z_count in the kernel's tiling) and the stitched frame is bit-identical to the one-device frame."""
    acq = cases.make(name)
    lib = bflib.library()
    try:
        lib.beamformer_hip_set_das_path(0x14 | 0x100)
        devices([0])
        one = bflib.beamform(acq.bp, acq.rf, acq.filters).copy()
        t = P.HipFrameTimings()
        assert lib.beamformer_hip_get_last_frame_timings(C.byref(t)) and int(t.das_path) == 5
        devices([0] * count)
        many = bflib.beamform(acq.bp, acq.rf, acq.filters)
        assert same_bits(one, many)
        # (a slab whose planes reach the end of an RF row is handed to the factored kernel by the row-end rule, das_select.h)
        paths = []
        for i in range(count):
            assert lib.beamformer_hip_get_device_frame_timings(i, C.byref(t)) and int(t.das_voxels) > 0
            paths.append(int(t.das_path))
        assert set(paths) <= {5, 3} and 5 in paths, paths
    finally:
        lib.beamformer_hip_set_das_path(0)


def test_pipelined_pushes_export_sum_and_display(bflib, devices):
    """Five frames pushed back to back through three RF slots on three device contexts, then the last
    two exported oldest-first, averaged and display-reduced: all equal to the one-device results."""
    acq = cases.make("config4_small")
    rng = np.random.default_rng(3)
    frames_rf = [np.ascontiguousarray(acq.rf if k == 0 else rng.integers(-2000, 2000, acq.rf.shape).astype(acq.rf.dtype))
                 for k in range(5)]
    lib = bflib.library()
    Zs, Ys, Xs = bflib.frame_shape(acq.bp)
    frame_bytes = (Zs * Ys * Xs * 8 + 63) // 64 * 64

    def run():
        for slot, fp in enumerate(acq.filters):
            assert lib.beamformer_create_filter(C.byref(fp), slot, 0)
        assert lib.beamformer_push_simple_parameters(C.byref(acq.bp))
        for rf in frames_rf:
            assert lib.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 0), bflib.last_error()
        out = np.zeros(2 * frame_bytes // 4, np.float32)
        assert lib.beamformer_get_last_frames(out.ctypes.data_as(C.c_void_p), out.nbytes, 2)
        avg = np.zeros(frame_bytes // 4, np.float32)
        assert lib.beamformer_hip_sum_last_frames(2, avg.ctypes.data_as(C.c_void_p), avg.nbytes), bflib.last_error()
        shown = np.zeros(Zs * Ys * Xs, np.float32)
        assert lib.beamformer_hip_display_last_frame(55.0, 1.0, 50.0, shown.ctypes.data_as(C.POINTER(C.c_float)), shown.size)
        return out, avg[: Zs * Ys * Xs * 2], shown

    devices([0])
    want = run()
    devices([0, 0, 0])
    got = run()
    for a, b in zip(want, got):
        assert same_bits(a, b)
    # the two exported frames differ (different RF) and the newest equals a fresh single push of frames_rf[-1]
    assert not same_bits(got[0][: frame_bytes // 4], got[0][frame_bytes // 4:])


def test_device_resident_rf_sharded_block_and_pair_count(bflib, devices, oracle):
    """RF that is already on the ingest device (read in place there, copied to the peers), a block with an
    output shard (the devices split the shard, not the grid), and the geometry-only pair count summed
    over devices."""
    import torch
    acq = cases.make("config4_small")
    lib = bflib.library()
    Zs, Ys, Xs = bflib.frame_shape(acq.bp)
    z0, zc = 5, Zs - 9
    ref, pairs = oracle.beamform(acq.bp, acq.rf, acq.filters, z=(z0, zc))
    devices([0, 0])
    for slot, fp in enumerate(acq.filters):
        assert lib.beamformer_create_filter(C.byref(fp), slot, 0)
    assert lib.beamformer_push_simple_parameters(C.byref(acq.bp))
    assert lib.beamformer_hip_set_output_shard(0, z0, zc)
    try:
        rf = torch.from_numpy(np.ascontiguousarray(acq.rf).view(np.uint8).reshape(-1)).to("cuda:0")
        torch.cuda.synchronize()
        lib.beamformer_hip_enable_pair_counting(1)
        assert lib.beamformer_hip_push_device_data_with_compute(C.c_void_p(rf.data_ptr()), rf.numel(), 0, 0), bflib.last_error()
        got = bflib.get_last_frame(acq.bp, shard_planes=zc)
        t = P.HipFrameTimings()
        assert lib.beamformer_hip_get_last_frame_timings(C.byref(t))
    finally:
        lib.beamformer_hip_enable_pair_counting(0)
        lib.beamformer_hip_set_output_shard(0, 0, 0)
    assert int(t.das_voxels) == zc * Ys * Xs
    assert abs(int(t.das_pairs) - pairs) <= max(4, 2e-4 * pairs)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.abs(got[ok] - ref[ok]).max() / np.abs(ref[ok]).max() <= cases.tolerance(acq)


def test_device_set_rules(bflib, devices):
    lib = bflib.library()
    devices([0, 0])
    acq = cases.make("config1_small")
    bflib.beamform(acq.bp, acq.rf, acq.filters)
    two = (C.c_int32 * 2)(0, 0)
    three = (C.c_int32 * 3)(0, 0, 0)
    assert lib.beamformer_hip_set_devices(two, 2)                       # the set in use
    assert not lib.beamformer_hip_set_devices(three, 3)                 # a different one: only after shutdown
    assert not lib.beamformer_hip_set_device(0)
    assert not lib.beamformer_hip_set_stream(C.c_void_p(1))             # a stream belongs to one device
    assert not lib.beamformer_hip_set_devices(two, 0) and not lib.beamformer_hip_set_devices(two, 9)


def test_device_info_reports_slabs_copies_and_one_rf_checksum(bflib, devices):
    """beamformer_hip_get_device_info: every device of the set names its slab, its DAS time, the time of the RF copy into it,
    how the copy travels (2 ingest device / same GPU, 1 direct peer access, 0 staged through the host) and a checksum of the RF
    it read, computed on the device: equal everywhere, and equal to numpy's on the pushed bytes"""
    acq = cases.make("config4_small")
    devices([0, 0, 0])
    frame = bflib.beamform(acq.bp, acq.rf, acq.filters)
    lib = bflib.library()
    infos = []
    for i in range(3):
        di = P.HipDeviceInfo()
        assert lib.beamformer_hip_get_device_info(i, C.byref(di))
        infos.append(di)
    assert [int(d.ordinal) for d in infos] == [0, 0, 0]
    assert all(int(d.peer_access) == 2 for d in infos)                 # one GPU listed three times: no link to enable
    assert sum(int(d.slab_count) for d in infos) == frame.shape[0]
    assert infos[0].peer_copy_ms == 0 and all(d.peer_copy_ms > 0 for d in infos[1:])
    assert all(d.das_ms > 0 and d.frame_ms >= d.das_ms for d in infos)
    sums = {int(d.rf_checksum) for d in infos}
    assert len(sums) == 1 and all(int(d.rf_bytes) == acq.rf.nbytes for d in infos)
    words = np.ascontiguousarray(acq.rf).view(np.uint8).reshape(-1)[: acq.rf.nbytes // 8 * 8].view(np.uint64)
    want = int((words * np.arange(1, words.size + 1, dtype=np.uint64)).sum(dtype=np.uint64))
    assert sums == {want}
    assert not lib.beamformer_hip_get_device_info(3, C.byref(P.HipDeviceInfo()))


def test_pipeline_without_das_on_several_devices_exports_one_zero_frame(bflib, devices):
    """a pipeline that never reaches DAS leaves a zero frame (beamformer_core.c:1573-1585): with several devices the ingest device
    holds it whole and the others an empty slab -- the stitched export has the one-device size, not N times it"""
    acq = cases.make("hercules_demod_decode_cw")
    acq.bp.compute_stages_count = 2                         # {Demodulate, Decode}
    devices([0])
    one = bflib.beamform(acq.bp, acq.rf, acq.filters).copy()
    assert not one.any()
    devices([0, 0])
    two = bflib.beamform(acq.bp, acq.rf, acq.filters)
    assert same_bits(one, two)


# ---- the same two properties across PHYSICAL devices: these run wherever two or more GPUs are visible (an 8-GPU node at round
# end) and skip on a one-GPU box, where every list above names ordinal 0 several times and nothing crosses a link

def _visible_devices():
    import torch
    return torch.cuda.device_count()           # reads the driver's list; creates no context


@pytest.mark.parametrize("name, count", [("config4_small", 2), ("config5_small", 4), ("rca_vls_cw", 8)])
def test_slabs_stitch_across_distinct_devices(name, count, bflib, devices):
    """ordinals 0..N-1: hipMemcpyPeerAsync between real devices, checked peer-access enable, cross-device stream waits -- the
    stitched frame must still be bit-identical to the one-device frame"""
    if _visible_devices() < count:
        pytest.skip(f"{count} distinct GPUs needed, {_visible_devices()} visible")
    acq = cases.make(name)
    devices([0])
    one = bflib.beamform(acq.bp, acq.rf, acq.filters).copy()
    devices(list(range(count)))
    many = bflib.beamform(acq.bp, acq.rf, acq.filters)
    assert same_bits(one, many)
    # and again in the other direction of the ring: the LAST device ingests, device 0 is a peer
    devices(list(reversed(range(count))))
    many = bflib.beamform(acq.bp, acq.rf, acq.filters)
    assert same_bits(one, many)


def test_device_info_across_distinct_devices(bflib, devices):
    """beamformer_hip_get_device_info on two physical GPUs: distinct ordinals, the peer's copy travels over a link (direct peer
    access or staged through the host -- never 'same GPU'), and both read the same RF bytes"""
    if _visible_devices() < 2:
        pytest.skip(f"2 distinct GPUs needed, {_visible_devices()} visible")
    acq = cases.make("config4_small")
    devices([0, 1])
    frame = bflib.beamform(acq.bp, acq.rf, acq.filters)
    lib = bflib.library()
    infos = []
    for i in range(2):
        di = P.HipDeviceInfo()
        assert lib.beamformer_hip_get_device_info(i, C.byref(di))
        infos.append(di)
    assert [int(d.ordinal) for d in infos] == [0, 1]
    assert int(infos[0].peer_access) == 2 and int(infos[1].peer_access) in (0, 1)
    assert sum(int(d.slab_count) for d in infos) == frame.shape[0]
    assert infos[1].peer_copy_ms > 0
    assert int(infos[0].rf_checksum) == int(infos[1].rf_checksum) and all(int(d.rf_bytes) == acq.rf.nbytes for d in infos)
