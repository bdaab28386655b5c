"""Known-answer tests that pin the CPU oracle itself (the reference ships no expected outputs
for this path): physics of a point scatterer, Hadamard encode/decode,
filter impulse responses, the float64 twin, chunk independence, sub-grid bit-equality."""
import ctypes as C

import numpy as np
import pytest

from ogl_beamforming_amd import configs as cfg
from ogl_beamforming_amd import params as P
from tests import cases

S = P.ShaderKind
D = P.DataKind
fp = C.POINTER(C.c_float)


def expected_voxel(acq):
    bp = acq.bp
    m = np.array(bp.das_voxel_transform[:], np.float64).reshape(4, 4).T
    pts = np.array([max(1, v) for v in bp.output_points[:3]])
    world = np.array(acq.scatterers[0])
    # solve voxel_transform * p = world for the axes that have extent
    p = np.linalg.lstsq(m[:3, :3], world - m[:3, 3], rcond=None)[0]
    return p * np.maximum(1, pts - 1)


@pytest.mark.parametrize("n,scale", [(1, 0.25), (2, 0.125), (4, 0.0625)])
def test_point_scatterer_peak(n, scale, oracle):
    acq = cfg.config(n, scale)
    frame, _ = oracle.beamform(acq.bp, acq.rf, acq.filters)
    mag = np.abs(np.nan_to_num(frame))
    z, y, x = np.unravel_index(np.argmax(mag), mag.shape)
    want = expected_voxel(acq)
    assert abs(x - want[0]) <= 2 and abs(y - want[1]) <= 2 and abs(z - want[2]) <= 2, ((x, y, z), want)
    assert mag.max() > 8 * np.median(mag[mag > 0])


def test_decode_inverts_hadamard_encoding(oracle):
    """x -> encode with Ht -> oracle pipeline {Decode} must give x back: build a frame whose
    decoded data is known and compare DAS of (encoded, decode on) with DAS of (plain, decode off)"""
    base = cfg.hercules("plain", 16, 8, 512, (8, 8, 8), cases.LO3, cases.HI3, seed=5, decode=0)
    plain = base.rf.reshape(16, 8, 512).astype(np.int32) // 8
    h = np.zeros(64, np.float32)
    oracle.library().oracle_hadamard_transpose(8, h.ctypes.data_as(fp))
    H = h.reshape(8, 8).astype(np.int32)
    encoded = np.einsum("jt,cts->cjs", H, plain)                  # in[j] = sum_t Ht[j][t] x[t]
    enc = cfg.hercules("encoded", 16, 8, 512, (8, 8, 8), cases.LO3, cases.HI3, seed=5, decode=1)
    a, _ = oracle.beamform(base.bp, plain.astype(np.int16).reshape(16, -1), base.filters)
    b, _ = oracle.beamform(enc.bp, encoded.astype(np.int16).reshape(16, -1), enc.filters)
    assert np.abs(a).max() > 0
    assert np.allclose(a, b, rtol=1e-5, atol=1e-3 * np.abs(a).max())


def test_filter_impulse_response_is_reversed_taps(oracle):
    """y[n] = sum_j h[j] x[n + j - (L-1)]: an impulse at n0 comes out as h reversed"""
    from oracle.binding import library
    L = library()

    class OracleFilter(C.Structure):
        _fields_ = [("filter_length", C.c_int), ("complex_filter", C.c_int), ("demodulate", C.c_int),
                    ("sampling_frequency", C.c_float), ("demodulation_frequency", C.c_float),
                    ("decimation_rate", C.c_int), ("sample_count", C.c_int), ("batch_sample_count", C.c_int),
                    ("in_stride", C.c_int * 3), ("out_stride", C.c_int * 3), ("in_kind", C.c_int), ("out_kind", C.c_int),
                    ("channels", C.c_int), ("transmits", C.c_int), ("in_elements", C.c_int64), ("workgroup", C.c_int),
                    ("coefficients", fp)]
    taps = np.array([1, 2, 3, 4, 5, 6, 7], np.float32)
    S_ = 256
    x = np.zeros(S_, np.float32)
    x[100] = 1.0
    y = np.zeros(S_, np.float32)
    f = OracleFilter()
    f.filter_length, f.decimation_rate, f.sample_count = 7, 1, S_
    f.in_stride[:] = [1, S_, S_]
    f.out_stride[:] = [1, S_, S_]
    f.in_kind = f.out_kind = int(D.Float32)
    f.channels = f.transmits = 1
    f.in_elements, f.workgroup = S_, 64
    f.coefficients = taps.ctypes.data_as(fp)
    L.oracle_filter.argtypes = [C.POINTER(OracleFilter), C.c_void_p, C.c_void_p, C.c_uint32]
    L.oracle_filter(C.byref(f), x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), 0)
    assert np.array_equal(y[100:107], taps[::-1])
    assert y[:100].sum() == 0 and y[107:].sum() == 0


def test_float64_twin_bounds_the_float_oracle(oracle):
    """the float restatement stays within 1e-3 of the float64 twin (tolerance budget of
    SURVEY section 8c)"""
    from oracle.binding import OracleParameterBlock, library
    acq = cases.make("rca_f32_complex_in")       # plan = [DAS]: RF feeds DAS directly
    L = library()
    f32, _ = oracle.beamform(acq.bp, acq.rf, acq.filters)

    class OracleDAS(C.Structure):
        _fields_ = [("acquisition_kind", C.c_uint32), ("sparse", C.c_int32), ("acquisition_count", C.c_int32),
                    ("channel_count", C.c_int32), ("chunk_channel_count", C.c_int32), ("sample_count", C.c_int32),
                    ("sampling_frequency", C.c_float), ("demodulation_frequency", C.c_float), ("speed_of_sound", C.c_float),
                    ("time_offset", C.c_float), ("interpolation_mode", C.c_uint32), ("f_number", C.c_float),
                    ("single_orientation", C.c_int32), ("transmit_receive_orientation", C.c_uint32),
                    ("single_focus", C.c_int32), ("focus_depth", C.c_float), ("transmit_angle", C.c_float),
                    ("output_size", C.c_uint32 * 3), ("readi_group_count", C.c_uint32), ("coherency_weighting", C.c_int32),
                    ("complex_data", C.c_int32), ("xdc_transform", C.c_float * 16), ("voxel_transform", C.c_float * 16),
                    ("xdc_element_pitch", C.c_float * 2), ("rf_element_offset", C.c_uint32), ("channel_offset", C.c_int32),
                    ("readi_group", C.c_uint32), ("focal_vectors", fp), ("sparse_elements", C.POINTER(C.c_int16)),
                    ("transmit_receive_orientations", C.POINTER(C.c_uint8)), ("readi_hadamard", fp),
                    ("z_first", C.c_uint32), ("z_count", C.c_uint32), ("y_first", C.c_uint32), ("y_count", C.c_uint32),
                    ("z_stride", C.c_uint32), ("y_stride", C.c_uint32), ("threads", C.c_int32),
                    ("row_first", C.c_int64), ("row_count", C.c_int64)]
    bp = acq.bp
    d = OracleDAS()
    d.acquisition_kind, d.acquisition_count, d.channel_count = bp.acquisition_kind, bp.acquisition_count, bp.channel_count
    d.chunk_channel_count, d.sample_count = bp.channel_count, bp.sample_count
    d.sampling_frequency, d.demodulation_frequency = bp.sampling_frequency, bp.demodulation_frequency
    d.speed_of_sound, d.time_offset, d.interpolation_mode, d.f_number = bp.speed_of_sound, bp.time_offset, bp.interpolation_mode, bp.f_number
    d.single_orientation, d.transmit_receive_orientation, d.single_focus = bp.single_orientation, bp.transmit_receive_orientation, bp.single_focus
    d.transmit_angle, d.focus_depth = bp.focal_vector[0], bp.focal_vector[1]
    d.output_size[:] = [max(1, v) for v in bp.output_points[:3]]
    d.complex_data = 1
    d.xdc_transform[:], d.voxel_transform[:] = bp.xdc_transform[:], bp.das_voxel_transform[:]
    d.xdc_element_pitch[:] = bp.xdc_element_pitch[:]
    fv = np.array([[bp.steering_angles[i], bp.focal_depths[i]] for i in range(256)], np.float32)
    orient = np.array(bp.transmit_receive_orientations[:], np.uint8)
    sparse = np.zeros(256, np.int16)
    d.focal_vectors = fv.ctypes.data_as(fp)
    d.transmit_receive_orientations = orient.ctypes.data_as(C.POINTER(C.c_uint8))
    d.sparse_elements = sparse.ctypes.data_as(C.POINTER(C.c_int16))
    voxels = int(np.prod(d.output_size[:]))
    out = np.zeros(2 * voxels, np.float64)
    inc = np.zeros(voxels, np.float64)
    rf = np.ascontiguousarray(acq.rf)
    L.oracle_das_f64.restype = C.c_uint64
    L.oracle_das_f64.argtypes = [C.POINTER(OracleDAS), fp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.oracle_das_f64(C.byref(d), rf.ctypes.data_as(fp), out.ctypes.data_as(C.POINTER(C.c_double)), inc.ctypes.data_as(C.POINTER(C.c_double)))
    f64 = (out[0::2] + 1j * out[1::2]).reshape(f32.shape)
    scale = np.abs(f64).max()
    assert scale > 0
    assert np.abs(f32 - f64).max() / scale < 1e-3


def test_chunking_does_not_change_the_result(oracle):
    """the reference beamforms 16 channels at a time into the same frame; 32 channels in two
    chunks must equal the sum of two 16-channel frames"""
    acq = cfg.rca("chunks", 32, 2, 512, (12, 12, 1), cases.LO3, cases.HI3, seed=9, f_number=0.0)
    whole, _ = oracle.beamform(acq.bp, acq.rf, acq.filters)
    parts = []
    for half in range(2):
        sub = cfg.rca("chunk", 16, 2, 512, (12, 12, 1), cases.LO3, cases.HI3, seed=9, f_number=0.0)
        # same geometry: the 16 elements of this half sit at their positions in the 32-element array
        sub.bp.xdc_transform[:] = acq.bp.xdc_transform[:]
        sub.bp.xdc_transform[12] = acq.bp.xdc_transform[12] - half * 16 * acq.bp.xdc_element_pitch[0]
        rf = acq.rf[half * 16:(half + 1) * 16]
        f, _ = oracle.beamform(sub.bp, rf, sub.filters)
        parts.append(f)
    assert np.abs(whole).max() > 0
    assert np.allclose(whole, parts[0] + parts[1], rtol=2e-5, atol=2e-5 * np.abs(whole).max())


def test_subgrid_is_bit_identical(oracle):
    acq = cases.make("config4_small")
    full, _ = oracle.beamform(acq.bp, acq.rf, acq.filters)
    sub, _ = oracle.beamform(acq.bp, acq.rf, acq.filters, z=(9, 5), y=(3, 11))
    assert np.array_equal(full[9:14, 3:14, :], sub, equal_nan=True)


def test_coherency_weighting_is_componentwise_and_nan_on_empty(oracle):
    """coherency_weighting.glsl:36: c *= c / incoherent per component; 0/0 = NaN where no
    (channel, transmit) term lands inside the RF"""
    acq = cases.make("hercules_demod_decode_cw")
    frame, _ = oracle.beamform(acq.bp, acq.rf, acq.filters)
    assert np.isnan(frame).any() and (~np.isnan(frame)).any()
    acq.bp.coherency_weighting = 0
    plain, _ = oracle.beamform(acq.bp, acq.rf, acq.filters)
    ok = ~np.isnan(frame)
    # weighted = (re^2, im^2)/inc >= 0 componentwise wherever it is defined
    assert (frame.real[ok] >= 0).all() and (frame.imag[ok] >= 0).all()
    assert np.all(plain[~ok] == 0)


def test_software_binary16_is_ieee(oracle):
    """Every f16-staged tolerance rests on oracle_f16.h: all 65536 bit patterns widen exactly as
    numpy's IEEE float16 does, and a million random floats (plus overflow, subnormal and tie
    cases) narrow to the same bits (round to nearest even)."""
    import ctypes as C
    L = oracle.library()
    L.oracle_f16_roundtrip.argtypes = [C.POINTER(C.c_uint16), C.POINTER(C.c_float), C.c_uint64]
    L.oracle_f16_bits_from_f32.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_uint16), C.c_uint64]
    bits = np.arange(65536, dtype=np.uint16)
    wide = np.zeros(65536, np.float32)
    L.oracle_f16_roundtrip(bits.ctypes.data_as(C.POINTER(C.c_uint16)), wide.ctypes.data_as(C.POINTER(C.c_float)), bits.size)
    want = bits.view(np.float16).astype(np.float32)
    assert np.array_equal(wide.view(np.uint32)[~np.isnan(want)], want.view(np.uint32)[~np.isnan(want)])
    assert np.all(np.isnan(wide[np.isnan(want)]))
    rng = np.random.default_rng(3)
    with np.errstate(invalid="ignore", over="ignore"):
      x = np.concatenate([
        rng.standard_normal(400000).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 6, 400000).astype(np.float32),
        rng.uniform(-70000, 70000, 300000).astype(np.float32),
        (rng.integers(0, 65536, 200000).astype(np.uint16).view(np.float16).astype(np.float32)
         * (1 + rng.choice([-2.0 ** -11, 0, 2.0 ** -11, 2.0 ** -12], 200000)).astype(np.float32)),     # around ties
        np.array([0.0, -0.0, 65504, 65519.99, 65520, 1e9, -1e9, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5, np.inf, -np.inf], np.float32)])
    x = x[~np.isnan(x)]
    got = np.zeros(x.size, np.uint16)
    L.oracle_f16_bits_from_f32(x.ctypes.data_as(C.POINTER(C.c_float)), got.ctypes.data_as(C.POINTER(C.c_uint16)), x.size)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(got, ref), np.flatnonzero(got != ref)[:5]


def test_oracle_output_is_frozen(oracle):
    """The oracle is the yardstick of every GPU parity test: its output for twelve named
    acquisitions is frozen in tests/golden/oracle_frames.npz (made by make_oracle_frames.py), so a
    change to oracle/*.c that moves any result shows up here first."""
    import os
    from tests import cases
    golden = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_frames.npz"))
    names = [k for k in golden.files if not k.endswith(".pairs")]
    assert len(names) == 12
    for name in names:
        acq = cases.make(name)
        frame, pairs = oracle.beamform(acq.bp, acq.rf, acq.filters)
        want = golden[name]
        if frame.shape[0] > 16:
            frame = frame[::8]                                             # as stored
        assert frame.shape == want.shape and frame.dtype == want.dtype
        assert np.array_equal(np.isnan(frame), np.isnan(want)), name
        ok = ~np.isnan(want)
        # same C, same flags (-ffp-contract=off): equal to the last bit here; a libm of another
        # vintage may move sin/cos by an ulp, hence the tiny allowance
        assert np.abs(frame[ok] - want[ok]).max() <= 1e-6 * np.abs(want[ok]).max(), name
        assert int(pairs) == int(golden[name + ".pairs"]), name


@pytest.mark.parametrize("name", ["config4_small", "hercules_demod_decode_cw", "uforces_sparse", "rca_nearest_real", "readi", "harness_tpw_small"])
def test_rows_outermost_schedule_is_bit_identical_to_the_chunk_by_chunk_one(name):
    """oracle_beamform's default schedule hands the rows of the image to the threads and walks the 16-channel chunks inside each
    (one parallel region: a 256-thread host scales -- bench.py's cpu_baseline); the reference's literal schedule is one pass over
    the image per chunk (beamformer_core.c:1604-1614).  Every voxel receives the same additions in the same order: same bits,
    same pair count, also on a strided sub-grid."""
    from oracle import binding as oracle
    from tests import cases
    acq = cases.make(name)
    L = oracle.library()
    try:
        L.oracle_set_rows_outermost(0)
        a, pa = oracle.beamform(acq.bp, acq.rf, acq.filters, threads=4)
        a2, pa2 = oracle.beamform(acq.bp, acq.rf, acq.filters, threads=3, y=(1, 3), stride=(1, 2))
        L.oracle_set_rows_outermost(1)
        b, pb = oracle.beamform(acq.bp, acq.rf, acq.filters, threads=4)
        b2, pb2 = oracle.beamform(acq.bp, acq.rf, acq.filters, threads=3, y=(1, 3), stride=(1, 2))
    finally:
        L.oracle_set_rows_outermost(1)
    assert pa == pb and pa2 == pb2
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(a2.view(np.uint32), b2.view(np.uint32))
