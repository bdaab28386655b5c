"""Parity at BASELINE.json's full sizes through size-independent properties (the CPU oracle
would need hours for a 512^3 frame):
  * the known scatterer of the synthetic acquisition peaks at its voxel,
  * the two independent DAS implementations (general kernel, separable fast path) agree,
  * z-slab shards are bit-identical to the same planes of the whole frame (the multi-GPU path),
  * the pipeline is linear in the RF (no coherency weighting),
  * the image extrema kernel agrees with numpy on the pulled frame.
The oracle still checks a few full-size planes directly (sub-grid restatement)."""
import ctypes as C

import numpy as np
import pytest

from ogl_beamforming_amd import configs as cfg
from ogl_beamforming_amd import params as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def config4():
    return cfg.config(4)          # 256 ch x 75 tx, 512^3: the configuration the metric is quoted on


def run(bflib, acq, shard=None, path=0):
    L = bflib.library()
    for slot, fp in enumerate(acq.filters):
        assert L.beamformer_create_filter(C.byref(fp), slot, 0)
    assert L.beamformer_push_simple_parameters(C.byref(acq.bp))
    L.beamformer_hip_set_das_path(path)
    if shard:
        assert L.beamformer_hip_set_output_shard(0, shard[0], shard[1])
    else:
        assert L.beamformer_hip_set_output_shard(0, 0, 0)
    L.beamformer_set_global_timeout(0xFFFFFFFF)
    rf = np.ascontiguousarray(acq.rf)
    try:
        assert L.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 0), bflib.last_error()
        return bflib.get_last_frame(acq.bp, shard_planes=shard[1] if shard else None)
    finally:
        L.beamformer_hip_set_output_shard(0, 0, 0)
        L.beamformer_hip_set_das_path(0)


def test_full_size_frame_properties(bflib, oracle, config4):
    acq = config4
    X, Y, Z = acq.bp.output_points[0], acq.bp.output_points[1], acq.bp.output_points[2]
    whole = run(bflib, acq)
    assert whole.shape == (Z, Y, X) and whole.dtype == np.complex64

    # (1) physics: the scatterer the RF was synthesised from
    mag = np.abs(np.nan_to_num(whole))
    z, y, x = np.unravel_index(np.argmax(mag), mag.shape)
    m = np.array(acq.bp.das_voxel_transform[:], np.float64).reshape(4, 4).T
    want = (np.array(acq.scatterers[0]) - m[:3, 3]) / np.diag(m[:3, :3]) * (np.array([X, Y, Z]) - 1)
    assert abs(x - want[0]) <= 2 and abs(y - want[1]) <= 2 and abs(z - want[2]) <= 2, ((x, y, z), want)

    # (2) image extrema kernel
    mm = (C.c_float * 2)()
    assert bflib.library().beamformer_hip_frame_min_max(mm)
    finite = np.abs(whole[~np.isnan(whole)])
    assert mm[1] == pytest.approx(float(finite.max()), rel=1e-6)
    assert mm[0] == pytest.approx(float(finite.min()), rel=1e-6, abs=1e-30)

    # (3) z-slab shards (what each GPU of a node computes) are bit-identical to the whole frame
    for first, count in ((0, 3), (255, 2), (509, 3)):
        slab = run(bflib, acq, shard=(first, count))
        assert np.array_equal(slab, whole[first:first + count], equal_nan=True)

    # (4) the oracle on a few full-size rows (same frame, sub-grid restatement)
    ref, _ = oracle.beamform(acq.bp, acq.rf, acq.filters, z=(int(round(want[2])), 1), y=(int(round(want[1])) - 1, 3), threads=16)
    got = whole[int(round(want[2])):int(round(want[2])) + 1, int(round(want[1])) - 1:int(round(want[1])) + 2]
    scale = np.abs(ref[~np.isnan(ref)]).max()
    assert np.array_equal(np.isnan(ref), np.isnan(got))
    ok = ~np.isnan(ref)
    assert np.abs(got[ok] - ref[ok]).max() / scale < 2e-3        # Int16 -> f16-staged Demodulate tolerance

    # (5) the general kernel (independent implementation) on a slab of the same frame
    general = run(bflib, acq, shard=(250, 4), path=1)
    fast = whole[250:254]
    ok = ~np.isnan(fast) & ~np.isnan(general)
    assert np.array_equal(np.isnan(fast), np.isnan(general))
    assert np.abs(fast[ok] - general[ok]).max() / np.abs(fast[ok]).max() < 1e-4

    # (6) the automatic choice is the LDS-staged kernel; the gather kernel it replaced agrees on the same slab
    t = P.HipFrameTimings()
    run(bflib, acq, shard=(250, 4))
    assert bflib.library().beamformer_hip_get_last_frame_timings(C.byref(t)) and int(t.das_path) == 2
    gathered = run(bflib, acq, shard=(250, 4), path=2)
    assert bflib.library().beamformer_hip_get_last_frame_timings(C.byref(t)) and int(t.das_path) == 1
    assert np.array_equal(np.isnan(fast), np.isnan(gathered))
    assert np.abs(fast[ok] - gathered[ok]).max() / np.abs(fast[ok]).max() < 1e-4

    # (7) the WHOLE frame with every term range-checked (STAGED_CHECKED): no term of the 2.5e12 leaves its staged window
    # (a violated host bound would read a neighbouring transmit's window -- wrong voxels, no fault), and the frame is the same
    bflib.set_hook("STAGED_CHECKED", "1")
    try:
        checked = run(bflib, acq)
        assert bflib.library().beamformer_hip_get_last_frame_timings(C.byref(t)) and int(t.das_path) == 2
        assert int(t.staged_window_violations) == 0
    finally:
        bflib.set_hook("STAGED_CHECKED", None)
    assert np.array_equal(checked.view(np.uint32), whole.view(np.uint32))


def test_full_size_linearity(bflib, config4):
    """B(a x + b y) = a B(x) + b B(y) for the whole pipeline without coherency weighting
    (Float32 data, so that scaling does not interact with Int16 / binary16 rounding)."""
    acq = config4
    bp = P.SimpleParameters.from_buffer_copy(bytes(acq.bp))
    bp.coherency_weighting = 0
    bp.data_kind = int(P.DataKind.Float32)
    rng = np.random.default_rng(11)
    x = rng.normal(0, 1, acq.rf.shape).astype(np.float32)
    y = acq.rf.astype(np.float32) / 4096
    lin = cfg.Acquisition("lin", bp, acq.filters, x)
    shard = (300, 2)
    bx = run(bflib, lin, shard=shard)
    lin.rf = y
    by = run(bflib, lin, shard=shard)
    lin.rf = (0.5 * x - 2.0 * y).astype(np.float32)
    bz = run(bflib, lin, shard=shard)
    want = 0.5 * bx - 2.0 * by
    assert np.abs(want).max() > 0
    assert np.abs(bz - want).max() / np.abs(want).max() < 2e-4


@pytest.mark.parametrize("n, path_name", [(2, "tile"), (3, "hercules"), (5, "hercules")])
def test_other_baseline_configs_at_full_size(n, path_name, bflib, oracle):
    """Configs 2, 3 and 5 at BASELINE sizes (config 5 on a 16-plane slab: 4.3 s per whole frame):
    the scatterer (where the acquisition has one) peaks where it was placed, the automatic DAS path is the expected one
    (factored kernel with block-wide LDS staging for config 2 -- every chunk of every block served from the staged windows --,
    HERCULES aligned-grid kernel for configs 3 and 5), it agrees with the general kernel on the
    same frame, and the oracle agrees on a few full-size rows around the scatterer."""
    acq = cfg.config(n)
    X, Y, Z = (max(1, v) for v in acq.bp.output_points[:3])
    m = np.array(acq.bp.das_voxel_transform[:], np.float64).reshape(4, 4).T
    want = None
    if acq.scatterers:
        sc = np.array(acq.scatterers[0], np.float64)
        if Z == 1:                                       # (X, Y, 1) image: voxel y runs along world z
            want = np.array([(sc[0] - m[0, 3]) / m[0, 0] * (X - 1), (sc[2] - m[2, 3]) / m[2, 1] * (Y - 1), 0.0])
        else:
            want = (sc - m[:3, 3]) / np.diag(m[:3, :3]) * (np.array([X, Y, Z]) - 1)
    wz, wy = (int(round(want[2])), int(round(want[1]))) if want is not None else (Z // 2, Y // 2)
    shard = None if n != 5 else (max(0, wz - 8), 16)
    frame = run(bflib, acq, shard=shard)
    t = P.HipFrameTimings()
    assert bflib.library().beamformer_hip_get_last_frame_timings(C.byref(t))
    assert int(t.das_path) == {"general": 0, "factored": 3, "hercules": 4, "tile": 5}[path_name]
    if path_name == "tile":
        # 1024 blocks x 32 chunks of four channels: (almost) all of them read from the staged windows
        assert int(t.tile_staged_chunks) + int(t.tile_gather_chunks) <= 1024 * 32
        assert int(t.tile_gather_chunks) <= 0.02 * int(t.tile_staged_chunks), (int(t.tile_staged_chunks), int(t.tile_gather_chunks))
    z0 = shard[0] if shard else 0
    if want is not None:
        mag = np.abs(np.nan_to_num(frame))
        z, y, x = np.unravel_index(np.argmax(mag), mag.shape)
        assert abs(x - want[0]) <= 2 and abs(y - want[1]) <= 2 and abs(z + z0 - want[2]) <= 2, ((x, y, z + z0), want)

    rows = (max(0, wy - 1), 3)
    ref, _ = oracle.beamform(acq.bp, acq.rf, acq.filters, z=(wz, 1), y=rows, threads=16)
    got = frame[wz - z0: wz - z0 + 1, rows[0]: rows[0] + 3]
    assert np.array_equal(np.isnan(ref), np.isnan(got))
    ok = ~np.isnan(ref)
    from tests import cases
    assert np.abs(got[ok] - ref[ok]).max() / np.abs(ref[ok]).max() <= cases.tolerance(acq)

    # the specialised kernel against the general kernel over the whole frame (config 5: the slab)
    general = run(bflib, acq, path=1, shard=shard)
    assert bflib.library().beamformer_hip_get_last_frame_timings(C.byref(t)) and int(t.das_path) == 0
    assert np.array_equal(np.isnan(general), np.isnan(frame))
    both = ~np.isnan(frame)
    # Config 2 carries a scatterer and band-limited (demodulated) data: 1e-4.  Configs 3 and 5 are white noise
    # straight out of the Hadamard decode: two kernels whose sample index differs by one ulp (1.2e-4 samples at
    # index 1500) interpolate sample-to-sample jumps of ~1.4 sigma differently, and the extreme over 1.7e7
    # incoherent sums reaches ~1.3e-4 of the frame's maximum; each kernel is held to the oracle above.
    assert np.abs(general[both] - frame[both]).max() / np.abs(frame[both]).max() < (1e-4 if n == 2 else 3e-4)


@pytest.mark.parametrize("interp, cw, points", [(P.InterpolationMode.Linear, True, (300, 201, 37)),
                                                (P.InterpolationMode.Cubic, False, (129, 333, 21)),
                                                (P.InterpolationMode.Nearest, True, (257, 96, 19))])
def test_odd_shapes_at_medium_size(interp, cw, points, bflib, oracle):
    """200 channels x 33 transmits into grids that are no multiple of any tile, on the automatic
    path (linear and cubic: the LDS-staged kernels), the gather kernel, the factored kernel and the general kernel: oracle
    rows of the middle plane."""
    path = 0.40 * 3072 / 25e6 * 1540.0
    acq = cfg.rca("odd", 200, 33, 3072, points, (-14e-3, -9e-3, 0.15 * path), (14e-3, 9e-3, 0.40 * path), seed=5,
                  orientation=0x12, interp=interp, cw=cw, f_number=0.7, angles=np.linspace(-14, 14, 33))
    z, y0 = points[2] // 2, points[1] // 3
    flags = {} if interp == P.InterpolationMode.Nearest else None
    ref, _ = oracle.beamform(acq.bp, acq.rf, acq.filters, z=(z, 1), y=(y0, 4), threads=16, flags=flags)
    ok = ~np.isnan(ref)
    seen = set()
    for mode in (0, 2, 4, 1):
        gpu = run(bflib, acq, path=mode)
        t = P.HipFrameTimings()
        assert bflib.library().beamformer_hip_get_last_frame_timings(C.byref(t))
        seen.add(int(t.das_path))
        got = gpu[z:z + 1, y0:y0 + 4]
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        err = np.abs(got[ok] - ref[ok]) / np.abs(ref[ok]).max()
        if interp == P.InterpolationMode.Nearest:
            # 6600 taps per voxel: nearly every voxel holds a tap within 2^-10 of a rounding boundary, so the
            # mismatch-fraction bar has nothing to bite on.  Voxels without such a tap must meet SURVEY 8c's
            # bar (< 1e-3 of them off by > 1e-3; tolerance 2e-3 here, f16-staged); the others are bounded
            # through the median and, with coherency weighting off, by the oracle's ambiguity budget.
            clean = ok & ~flags["near_half"]
            if clean.any():
                bad_clean = np.abs(got[clean] - ref[clean]) / np.abs(ref[ok]).max() > 2e-3
                assert np.mean(bad_clean) < 1e-3
            if not cw:
                slack = 2e-3 * np.abs(ref[ok]).max() + 1.01 * flags["budget"]
                assert (np.abs(got - ref)[ok] <= slack[ok]).all()
            assert np.median(err) < 1e-4
        else:
            assert err.max() <= 2e-3                                          # Int16 -> f16-staged Demodulate
    assert seen == ({2, 1, 3, 0} if interp == P.InterpolationMode.Linear else {2, 3, 0} if interp == P.InterpolationMode.Cubic else {3, 0})


def _f32c_rca(name, C, A, points, lo, hi, *, seed, interp, cw, pitch, f_number, orientation, angles, scatterer):
    """A BASELINE geometry fed Float32Complex RF straight into DAS (quirk Q6: decoding off, no Demodulate),
    so that nothing is staged through binary16 and the DAS tolerance (1e-4) applies at full size.  The RF is
    unit Gaussian IQ noise plus the baseband echo of one point scatterer ("one deterministic
    point-scatterer echo per transmit so images are checkable"): amplitude x Gaussian envelope x
    e^{-j 2 pi fd t0}, which the DAS rotation e^{+j 2 pi fd t} re-phases to a coherent peak."""
    fs, fd, S = 12.5e6, 6.25e6, 2048
    acq = cfg.rca(name, C, A, S, points, lo, hi, seed=seed, data_kind=P.DataKind.Float32Complex, demodulate=False,
                  interp=interp, cw=cw, pitch=pitch, f_number=f_number, orientation=orientation, angles=angles,
                  fs=fs, fd=fd)
    delays = cfg._rca_delays(acq.bp, scatterer, angles, np.full(A, np.inf), [orientation] * A)      # seconds, (C, A)
    n0 = (delays * fs).reshape(-1)
    iq = acq.rf.reshape(C * A, S, 2)
    width = 2.5 * fs / fd * 2                                   # samples (a 2.5-cycle burst at fd, seen at fs = 2 fd)
    offsets = np.arange(-24, 25)
    idx = np.floor(n0).astype(np.int64)[:, None] + offsets[None, :]
    env = 8.0 * np.exp(-((idx - n0[:, None]) / width) ** 2)
    phase = -2.0 * np.pi * fd * (n0 / fs)
    ok = (idx >= 0) & (idx < S)
    rows = np.broadcast_to(np.arange(C * A)[:, None], idx.shape)
    np.add.at(iq[:, :, 0], (rows[ok], idx[ok]), (env * np.cos(phase)[:, None])[ok].astype(np.float32))
    np.add.at(iq[:, :, 1], (rows[ok], idx[ok]), (env * np.sin(phase)[:, None])[ok].astype(np.float32))
    acq.scatterers = [scatterer]
    return acq


def _rows_against_oracle(frame, acq, oracle, selections, peak, local_tol=5e-4):
    """Each selection = (z or None, y0): three rows.  Two bars: the contract's -- error relative to the frame's
    peak <= 1e-4 -- and a local one relative to the compared rows' own maximum.  Away from the scatterer a
    voxel is an INCOHERENT sum of noise taps, so its relative error is the per-tap error, and two float
    evaluations of a 2000-sample index that differ by one ulp (2.4e-4 samples) already differ by
    2 pi x 0.5 x 2.4e-4 = 7.5e-4 rad in the demodulation phase: 5e-4 is what float arithmetic leaves there
    (coherency weighting squares the coherent sum and so doubles its relative error: 2e-3 there)."""
    worst = (0.0, 0.0)
    for z, y0 in selections:
        kw = dict(z=(z, 1)) if z is not None else {}
        ref, _ = oracle.beamform(acq.bp, acq.rf, acq.filters, y=(y0, 3), threads=16, **kw)
        got = frame[z:z + 1, y0:y0 + 3] if z is not None else frame[:, y0:y0 + 3]
        assert np.array_equal(np.isnan(ref), np.isnan(got)), (z, y0)
        ok = ~np.isnan(ref)
        assert ok.any() and np.abs(ref[ok]).max() > 0
        delta = np.abs(got[ok] - ref[ok]).max()
        assert delta / peak <= 1e-4, (z, y0, delta / peak)
        assert delta / np.abs(ref[ok]).max() <= local_tol, (z, y0, delta / np.abs(ref[ok]).max())
        worst = (max(worst[0], delta / peak), max(worst[1], delta / np.abs(ref[ok]).max()))
    return worst


@pytest.mark.parametrize("das_mode, kernel", [(0, 2), (2, 1)], ids=["staged", "gather"])
def test_config4_f32_complex_rows_at_the_edges_of_the_volume(das_mode, kernel, bflib, oracle):
    """Config 4's geometry (256 ch x 75 tx -> 512^3, coherency weighting; the headline kernel -- LDS-staged,
    the automatic choice -- and the gather kernel it replaced) on
    Float32Complex RF: three rows at each of z, y in {first, middle, last}, plus the scatterer's rows,
    against the float oracle -- first and last planes, tile corners, the range-checked inner loop (rows
    whose index leaves the RF at the volume's edges) and the tail of the XCD tile walk."""
    path = 4096 / 25e6 * cfg.SPEED_OF_SOUND
    z0, z1 = 0.12 * path, 0.30 * path
    half = 255 / 2 * 0.15e-3
    scatterer = (0.2 * half, -0.3 * half, z0 + 0.5 * (z1 - z0))
    acq = _f32c_rca("config4_f32c", 256, 75, (512, 512, 512), (-half, -half, z0), (half, half, z1), seed=4,
                    interp=P.InterpolationMode.Linear, cw=True, pitch=0.15e-3, f_number=0.5, orientation=0x12,
                    angles=np.linspace(-18.5, 18.5, 75), scatterer=scatterer)
    frame = run(bflib, acq, path=das_mode)
    t = P.HipFrameTimings()
    assert bflib.library().beamformer_hip_get_last_frame_timings(C.byref(t)) and int(t.das_path) == kernel
    mag = np.abs(np.nan_to_num(frame))
    pz, py, px = np.unravel_index(np.argmax(mag), mag.shape)
    want = (np.array(scatterer) - np.array([-half, -half, z0])) / np.array([2 * half, 2 * half, z1 - z0]) * 511
    # (the echo's Gaussian envelope spans ~7 planes either side: the axial maximum is flat)
    assert abs(px - want[0]) <= 2 and abs(py - want[1]) <= 2 and abs(pz - want[2]) <= 4, ((px, py, pz), want)
    selections = [(z, y0) for z in (0, 255, 511) for y0 in (0, 255, 509)] + [(int(pz), max(0, int(py) - 1))]
    worst = _rows_against_oracle(frame, acq, oracle, selections, float(mag.max()), local_tol=2e-3)
    print(f"config 4 geometry, f32 complex RF, {3 * len(selections)} rows: max error {worst[0]:.2e} of the peak, {worst[1]:.2e} of the rows' own maximum")


def test_config2_block_staged_frames_repeat_bit_for_bit(bflib):
    """Config 2 at full size on das_tile.hip, twelve frames of the same push: 1024 blocks on 256 CUs, four rounds of the chip, every
    block handing 32 x 4 x 2 LDS buffers from its staging threads to its consumers -- a missing barrier shows as a frame that differs."""
    acq = cfg.config(2)
    first = run(bflib, acq).copy()
    t = P.HipFrameTimings()
    assert bflib.library().beamformer_hip_get_last_frame_timings(C.byref(t)) and int(t.das_path) == 5 and int(t.tile_gather_chunks) == 0
    for k in range(11):
        again = run(bflib, acq)
        assert np.array_equal(first.view(np.uint32), again.view(np.uint32)), f"frame {k + 2} differs from the first"


def test_config2_f32_complex_first_and_last_rows(bflib, oracle):
    """Config 2's geometry (128 ch x 31 tx -> 1024^2, cubic, block-staged factored kernel) on Float32Complex RF: the
    first and the last three image rows (depth extremes) and the scatterer's rows against the float oracle."""
    path = 4096 / 25e6 * cfg.SPEED_OF_SOUND
    z0, z1 = 0.12 * path, 0.40 * path
    scatterer = (-2.0e-3, 0.0, z0 + 0.4 * (z1 - z0))
    acq = _f32c_rca("config2_f32c", 128, 31, (1024, 1024, 1), (-12.8e-3, 0, z0), (12.8e-3, 0, z1), seed=2,
                    interp=P.InterpolationMode.Cubic, cw=False, pitch=0.2e-3, f_number=1.0, orientation=0x22,
                    angles=np.linspace(-15.0, 15.0, 31), scatterer=scatterer)
    frame = run(bflib, acq)
    t = P.HipFrameTimings()
    assert bflib.library().beamformer_hip_get_last_frame_timings(C.byref(t)) and int(t.das_path) == 5
    mag = np.abs(frame)
    _, py, px = np.unravel_index(np.argmax(mag), mag.shape)
    assert abs(px - (scatterer[0] + 12.8e-3) / 25.6e-3 * 1023) <= 2 and abs(py - 0.4 * 1023) <= 2, (px, py)
    worst = _rows_against_oracle(frame, acq, oracle, [(None, 0), (None, 1021), (None, max(0, int(py) - 1))], float(mag.max()))
    print(f"config 2 geometry, f32 complex RF: max error {worst[0]:.2e} of the peak, {worst[1]:.2e} of the rows' own maximum")


@pytest.mark.parametrize("demodulate, interp", [(True, P.InterpolationMode.Linear), (False, P.InterpolationMode.Linear),
                                                (True, P.InterpolationMode.Cubic)], ids=["iq", "real", "iq_cubic"])
@pytest.mark.parametrize("transmits", [24, 48, 75, 100, 128])
def test_staged_kernel_against_the_gather_kernel_over_transmit_counts(transmits, demodulate, interp, bflib):
    """The staged kernels' staging width per thread is a template parameter (1 to 4 window elements per thread and
    channel for IQ samples, up to 8 for real ones, by the transmit count and the window size); the BASELINE
    configurations only reach two of them.  Config 4's geometry at 128 channels with 24 to 128 transmits, IQ
    (Demodulate; linear and cubic) and real (undecimated: 64-sample windows) samples, one slab: the staged kernel
    (automatic) against what path 2 runs -- the gather kernel, or the block-staged factored kernel for cubic --, which the
    oracle-checked tests hold to the oracle."""
    Cn, S = 128, 2048
    half = (Cn - 1) / 2 * 0.15e-3
    path = S / 25e6 * cfg.SPEED_OF_SOUND
    z0, z1 = 0.12 * path, 0.30 * path
    acq = cfg.rca(f"tx{transmits}", Cn, transmits, S, (256, 256, 256), (-half, -half, z0), (half, half, z1), seed=9, cw=True,
                  pitch=0.15e-3, orientation=0x12, f_number=0.5, angles=np.linspace(-18.5, 18.5, transmits), demodulate=demodulate, interp=interp)
    t = P.HipFrameTimings()
    staged = run(bflib, acq, shard=(120, 4))
    assert bflib.library().beamformer_hip_get_last_frame_timings(C.byref(t))
    if interp == P.InterpolationMode.Cubic and transmits == 128:
        # 128 transmits x 32-byte window elements do not fit the LDS: the staged cubic kernel declines, the factored kernel runs -- in its
        # block-staged form (das_tile.hip): the slab's grid is fine enough for 64-sample windows
        assert int(t.das_path) == 5
        return
    assert int(t.das_path) == 2
    bflib.set_hook("STAGED_CHECKED", "1")
    try:
        checked = run(bflib, acq, shard=(120, 4))
        assert bflib.library().beamformer_hip_get_last_frame_timings(C.byref(t)) and int(t.das_path) == 2 and int(t.staged_window_violations) == 0
    finally:
        bflib.set_hook("STAGED_CHECKED", None)
    assert np.array_equal(checked.view(np.uint32), staged.view(np.uint32))
    gathered = run(bflib, acq, shard=(120, 4), path=2)          # cubic: the factored kernel (block-staged form)
    assert bflib.library().beamformer_hip_get_last_frame_timings(C.byref(t)) and int(t.das_path) == (5 if interp == P.InterpolationMode.Cubic else 1)
    assert np.array_equal(np.isnan(staged), np.isnan(gathered))
    ok = ~np.isnan(gathered)
    assert ok.any() and np.abs(staged[ok] - gathered[ok]).max() / np.abs(gathered[ok]).max() < 1e-4



def test_config1_at_full_size_against_the_whole_oracle_frame(bflib, oracle):
    """BASELINE configs[0] at its own (small) size: 64 channels x 2048 samples x 1 plane wave -> 256 x 256, every voxel against
    the oracle on the automatic path (general kernel with the channel split) and without the split"""
    from tests import cases
    from tests.test_gpu_parity import compare
    acq = cfg.config(1)
    ref, _ = oracle.beamform(acq.bp, acq.rf, acq.filters, threads=16)
    for mode in (0, 0x11):
        gpu = run(bflib, acq, path=mode)
        compare(gpu, ref, acq)


def test_config5_rows_at_the_edges_of_the_volume(bflib, oracle):
    """Config 5 (HERCULES 256 ch x 128 tx -> 512^3, Demodulate -> Decode -> DAS, coherency weighting) as a WHOLE frame on the
    aligned-grid kernel: oracle rows at z in {0, 255, 510} x y in {0, 255, 510} (first and last planes and rows: the kernel's
    range-checked loop at full width, the tail of the tile walk) -- one strided oracle pass, 9 rows of 512 voxels."""
    from tests import cases
    acq = cfg.config(5)
    frame = run(bflib, acq)
    t = P.HipFrameTimings()
    assert bflib.library().beamformer_hip_get_last_frame_timings(C.byref(t)) and int(t.das_path) == 4
    ref, _ = oracle.beamform(acq.bp, acq.rf, acq.filters, threads=16, z=(0, 3), y=(0, 3), stride=(255, 255))
    got = frame[0:511:255, 0:511:255]
    assert got.shape == ref.shape == (3, 3, 512)
    assert np.array_equal(np.isnan(ref), np.isnan(got))
    ok = ~np.isnan(ref)
    assert ok.any() and np.abs(ref[ok]).max() > 0
    peak = np.abs(frame[~np.isnan(frame)]).max()
    # binary16 staging (fp16 RF through Demodulate and Decode): 2e-3 of the frame's peak, and of the compared rows' own maximum
    # with the factor incoherent noise sums carry (tests/test_gpu_full_size.py _rows_against_oracle)
    delta = np.abs(got[ok] - ref[ok]).max()
    assert delta / peak <= cases.tolerance(acq), delta / peak
    assert delta / np.abs(ref[ok]).max() <= 4 * cases.tolerance(acq), delta / np.abs(ref[ok]).max()


@pytest.mark.parametrize("kind, path", [("tpw", 3), ("forces", 3), ("hercules", 4), ("vls", 3)])
def test_reference_harness_frame_at_full_size(kind, path, bflib, oracle):
    """The frame the reference's own throughput harness beamforms (tests/throughput.c:20-23, :443-491): 256 channels x 128
    transmits x 4096 samples -> the 512 x 1024 XZ view plane, cubic, F# 0.5, {Demodulate, Decode, DAS}.  The automatic path
    (factored kernel, gather loop with all gathers of a transmit issued together; HERCULES: the aligned-grid kernel reading raw taps)
    against oracle rows at the first, a middle and the last depths."""
    from tests import cases
    acq = cfg.harness(kind)
    p, kernel, _, reasons, d = bflib.describe_das(acq.bp, acq.filters)
    assert p == path, (kernel, reasons)
    if kind == "hercules":
        assert int(d.hercules_prepared_copy) == 0
    frame = run(bflib, acq)
    t = P.HipFrameTimings()
    assert bflib.library().beamformer_hip_get_last_frame_timings(C.byref(t)) and int(t.das_path) == path
    assert frame.shape == (1, 1024, 512)
    ref, _ = oracle.beamform(acq.bp, acq.rf, acq.filters, threads=16, y=(0, 4), stride=(1, 341))
    got = frame[:, 0:1024:341]
    assert got.shape == ref.shape
    assert np.array_equal(np.isnan(ref), np.isnan(got))
    ok = ~np.isnan(ref)
    peak = np.abs(frame[~np.isnan(frame)]).max()
    delta = np.abs(got[ok] - ref[ok]).max()
    assert delta / peak <= cases.tolerance(acq), delta / peak
    assert delta / np.abs(ref[ok]).max() <= 4 * cases.tolerance(acq), delta / np.abs(ref[ok]).max()
