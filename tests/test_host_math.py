"""Host DSP math pinned against the COMPILED REFERENCE (tests/golden/host_math.npz, emitted by
tests/golden/make_golden.py from /root/reference/math.c via oracle/_ref/libref_math.so):
both the oracle's restatement (oracle/oracle_math.c) and the product's (csrc/host_math.cpp,
reached through the beamformer_hip_host_* hooks of the C ABI)."""
import ctypes as C
import os

import numpy as np
import pytest

from ogl_beamforming_amd import params as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "host_math.npz"))
fp = C.POINTER(C.c_float)


def ptr(a):
    return a.ctypes.data_as(fp)


def test_reference_builds_only_power_of_two_hadamard():
    """quirk Q1: math.c:96 returns NULL for every order that needs the 12/20 base"""
    for order, built in zip(G["hadamard_orders"], G["hadamard_built"]):
        assert bool(built) == (order & (order - 1) == 0), order


@pytest.mark.parametrize("order", [int(o) for o in G["hadamard_orders"]])
def test_hadamard(order, oracle, bflib):
    o = np.zeros(order * order, np.float32)
    p = np.zeros(order * order, np.float32)
    assert oracle.library().oracle_hadamard_transpose(order, ptr(o)) == 1
    assert bflib.library().beamformer_hip_host_hadamard(order, ptr(p)) == 1
    assert np.array_equal(o, p)
    H = o.reshape(order, order)
    assert set(np.unique(H)) == {-1.0, 1.0}
    assert np.array_equal(H @ H.T, order * np.eye(order))            # decode o encode = identity
    if f"hadamard_{order}" in G.files:
        assert np.array_equal(H, G[f"hadamard_{order}"].astype(np.float32))   # bit-exact vs reference


def test_hadamard_unsupported_orders(oracle, bflib):
    buf = np.zeros(64 * 64, np.float32)
    for order in (3, 6, 28, 36, 44):
        assert oracle.library().oracle_hadamard_transpose(order, ptr(buf)) == 0
        assert bflib.library().beamformer_hip_host_hadamard(order, ptr(buf)) == 0


def test_hadamard_base_rows_match_reference_tables(oracle):
    """first rows of the order-12 / order-20 bases as printed in math.c:38-76"""
    h12 = np.zeros(144, np.float32)
    oracle.library().oracle_hadamard_transpose(12, ptr(h12))
    assert h12.reshape(12, 12)[1].tolist() == [1, -1, -1, 1, -1, -1, -1, 1, 1, 1, -1, 1]
    assert h12.reshape(12, 12)[2].tolist() == [1, 1, -1, -1, 1, -1, -1, -1, 1, 1, 1, -1]
    h20 = np.zeros(400, np.float32)
    oracle.library().oracle_hadamard_transpose(20, ptr(h20))
    assert h20.reshape(20, 20)[1].tolist() == [1, -1, -1, 1, 1, -1, -1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1, -1, -1, 1]
    assert h20.reshape(20, 20)[2].tolist() == [1, -1, 1, 1, -1, -1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1, -1, -1, 1, -1]


def product_filter(bflib, fparams):
    taps = np.zeros(8192, np.float32)
    delay = C.c_float(0)
    cplx = C.c_uint32(0)
    n = bflib.library().beamformer_hip_host_filter(C.byref(fparams), ptr(taps), taps.size, C.byref(delay), C.byref(cplx))
    return n, taps[: n * (2 if cplx.value else 1)].copy(), delay.value, bool(cplx.value)


def test_kaiser_bit_exact(oracle, bflib):
    for i, (fc, fs, beta, n) in enumerate(G["kaiser_args"]):
        n = int(n)
        ref = G[f"kaiser_{i}"]
        o = np.zeros(n, np.float32)
        oracle.library().oracle_kaiser_low_pass(fc, fs, beta, n, ptr(o))
        assert np.array_equal(o, ref), i
        f = P.FilterParameters()
        f.kind, f.sampling_frequency = int(P.FilterKind.Kaiser), fs
        f.kaiser.cutoff_frequency, f.kaiser.beta, f.kaiser.length = fc, beta, n
        length, taps, delay, cplx = product_filter(bflib, f)
        assert length == n and not cplx
        assert np.array_equal(taps, ref), i
        assert delay == pytest.approx(n / 2 / np.float32(fs), rel=1e-6)          # beamformer_core.c:377


def test_bessel_i0(oracle):
    for x, ref in zip(G["i0_x"], G["i0"]):
        assert oracle.library().oracle_bessel_i0(float(x)) == pytest.approx(ref, rel=1e-14)


def test_chirps_and_moments(oracle, bflib):
    L = oracle.library()
    for i, (f0, f1, fs, n, rev) in enumerate(G["chirp_args"]):
        n, rev = int(n), int(rev)
        a = np.zeros(n, np.float32)
        L.oracle_rf_chirp(f0, f1, fs, n, rev, ptr(a))
        assert np.array_equal(a, G[f"rf_chirp_{i}"]), i
        b = np.zeros(2 * n, np.float32)
        L.oracle_baseband_chirp(f0, f1, fs, n, rev, 0.5, ptr(b))
        assert np.array_equal(b, G[f"baseband_chirp_{i}"]), i
        assert L.oracle_real_filter_first_moment(ptr(a), n, fs) == G["real_moments"][i]
        assert L.oracle_complex_filter_first_moment(ptr(b), n, fs) == G["complex_moments"][i]
        if rev:      # beamformer_filter_create builds reversed (matched) chirps (beamformer_core.c:385-389)
            for complex_taps, ref, moment in ((1, G[f"baseband_chirp_{i}"], G["complex_moments"][i]),
                                              (0, G[f"rf_chirp_{i}"], G["real_moments"][i])):
                f = P.FilterParameters()
                f.kind, f.sampling_frequency, f.complex = int(P.FilterKind.MatchedChirp), fs, complex_taps
                f.matched_chirp.duration = (n + 0.5) / fs
                f.matched_chirp.min_frequency, f.matched_chirp.max_frequency = f0, f1
                length, taps, delay, cplx = product_filter(bflib, f)
                assert length == n and cplx == bool(complex_taps)
                assert np.array_equal(taps, ref)
                assert delay == moment


def test_tukey(oracle):
    L = oracle.library()
    for key, r in (("tukey_02", 0.2), ("tukey_05", 0.5)):
        ours = np.array([L.oracle_tukey_window(float(t), r) for t in G["tukey_t"]], np.float32)
        assert np.array_equal(ours, G[key])


def test_das_transforms_and_m4(oracle):
    L = oracle.library()
    for i in range(len(G["das_transform"])):
        m = np.zeros(16, np.float32)
        pts = (C.c_int * 3)(*[int(v) for v in G["das_transform_points_in"][i]])
        L.oracle_das_transform(ptr(G["das_transform_lo"][i].copy()), ptr(G["das_transform_hi"][i].copy()), pts, ptr(m))
        assert list(pts) == G["das_transform_points_out"][i].tolist()
        assert np.array_equal(m, G["das_transform"][i]), i
    for plane in range(3):
        m = np.zeros(16, np.float32)
        L.oracle_das_transform_2d(plane, ptr(np.array([-7e-3, 4e-3], np.float32)), ptr(np.array([9e-3, 33e-3], np.float32)),
                                  2.5e-3, ptr(m))
        assert np.array_equal(m, G["das_transform_2d"][plane]), plane
    for i in range(4):
        out = np.zeros(16, np.float32)
        L.oracle_m4_mul(ptr(G["m4_a"][i].copy()), ptr(G["m4_b"][i].copy()), ptr(out))
        assert np.array_equal(out, G["m4_ab"][i])
    # the synthetic-config helpers use the same transforms
    from ogl_beamforming_amd import configs
    # (numpy computes the extents in float64 before the cast: equal to the last ulp only)
    assert np.allclose(configs.das_transform_2d_xz((-9.6e-3, 5e-3), (9.6e-3, 45e-3)), G["das_transform"][1], rtol=1e-6, atol=0)
    assert np.allclose(configs.das_transform_3d((-19e-3, -19e-3, 10e-3), (19e-3, 19e-3, 90e-3)), G["das_transform"][2], rtol=1e-6, atol=0)


def test_against_live_reference_when_present(oracle):
    """in the build container the compiled reference itself is available: spot-check beyond
    the committed fixture"""
    from oracle import binding
    if not os.path.exists(binding.REF_LIBRARY_PATH):
        pytest.skip("oracle/_ref/libref_math.so not built (no /root/reference here)")
    r = C.CDLL(binding.REF_LIBRARY_PATH)
    r.ref_kaiser_low_pass.argtypes = [C.c_float] * 3 + [C.c_int, fp]
    rng = np.random.default_rng(3)
    for _ in range(20):
        fs = float(rng.uniform(5e6, 60e6)); fc = float(rng.uniform(0.02, 0.45) * fs)
        beta = float(rng.uniform(0, 14)); n = int(rng.integers(3, 300))
        a, b = np.zeros(n, np.float32), np.zeros(n, np.float32)
        oracle.library().oracle_kaiser_low_pass(fc, fs, beta, n, ptr(a))
        r.ref_kaiser_low_pass(fc, fs, beta, n, ptr(b))
        assert np.array_equal(a, b)
