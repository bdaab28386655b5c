#!/usr/bin/env python3
"""Generates tests/golden/host_math.npz from the reference's own host math.

Runs only in the build container: it loads oracle/_ref/libref_math.so, which oracle/Makefile
compiles (ROCm clang) from the reference sources in place under /root/reference
(math.c, external/cephes.c, generated/beamformer.c via oracle/ref_harness.c).  The fixture is
DATA -- inputs and the reference's outputs -- and is committed; the reference never is.

    make -C oracle ref && python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "libref_math.so")


def main():
    r = C.CDLL(REF)
    fp = C.POINTER(C.c_float)
    f = lambda a: a.ctypes.data_as(fp)
    out = {}

    # Hadamard (math.c:35-134): every order the reference can build, and which it cannot
    orders = [2, 4, 8, 12, 16, 20, 24, 32, 40, 48, 64, 80, 96, 128, 160, 192, 256]   # tests/decode.c:17-19
    built = []
    for n in orders:
        a = np.zeros(n * n, np.float32)
        ok = r.ref_hadamard_transpose(n, 0, f(a))
        built.append(ok)
        if ok:
            out[f"hadamard_{n}"] = a.astype(np.int8).reshape(n, n)
    out["hadamard_orders"] = np.array(orders, np.int32)
    out["hadamard_built"] = np.array(built, np.int32)

    # Kaiser low pass (math.c:750-767)
    r.ref_kaiser_low_pass.argtypes = [C.c_float] * 3 + [C.c_int, fp]
    kaiser = [(2.5e6, 25e6, 5.65, 36), (3.125e6, 12.5e6, 5.65, 36), (1e6, 12.5e6, 3.0, 33), (3.9e6, 50e6, 8.6, 128),
              (0.1, 1.0, 0.0, 7), (5e6, 20e6, 12.5, 201), (4e6, 25e6, 4.0, 21)]
    out["kaiser_args"] = np.array(kaiser, np.float64)
    for i, (fc, fs, beta, n) in enumerate(kaiser):
        a = np.zeros(n, np.float32)
        r.ref_kaiser_low_pass(fc, fs, beta, n, f(a))
        out[f"kaiser_{i}"] = a

    # chirps (math.c:769-797)
    r.ref_rf_chirp.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, fp]
    r.ref_baseband_chirp.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, fp]
    chirps = [(2e6, 8e6, 25e6, 125, 1), (2e6, 8e6, 25e6, 125, 0), (-2e6, 2e6, 12.5e6, 50, 1), (1e6, 3e6, 10e6, 64, 1)]
    out["chirp_args"] = np.array(chirps, np.float64)
    for i, (f0, f1, fs, n, rev) in enumerate(chirps):
        a = np.zeros(n, np.float32)
        r.ref_rf_chirp(f0, f1, fs, n, rev, f(a))
        out[f"rf_chirp_{i}"] = a
        b = np.zeros(2 * n, np.float32)
        r.ref_baseband_chirp(f0, f1, fs, n, rev, 0.5, f(b))
        out[f"baseband_chirp_{i}"] = b

    # filter first moments (math.c:713-737) of the chirps above
    r.ref_real_filter_first_moment.restype = C.c_float
    r.ref_real_filter_first_moment.argtypes = [fp, C.c_int, C.c_float]
    r.ref_complex_filter_first_moment.restype = C.c_float
    r.ref_complex_filter_first_moment.argtypes = [fp, C.c_int, C.c_float]
    out["real_moments"] = np.array([r.ref_real_filter_first_moment(f(out[f"rf_chirp_{i}"]), chirps[i][3], chirps[i][2])
                                    for i in range(len(chirps))], np.float32)
    out["complex_moments"] = np.array([r.ref_complex_filter_first_moment(f(out[f"baseband_chirp_{i}"]), chirps[i][3], chirps[i][2])
                                       for i in range(len(chirps))], np.float32)

    # Tukey window (math.c:739-747)
    r.ref_tukey_window.restype = C.c_float
    r.ref_tukey_window.argtypes = [C.c_float, C.c_float]
    ts = np.linspace(0, 1, 101, dtype=np.float32)
    out["tukey_t"] = ts
    out["tukey_02"] = np.array([r.ref_tukey_window(float(t), 0.2) for t in ts], np.float32)
    out["tukey_05"] = np.array([r.ref_tukey_window(float(t), 0.5) for t in ts], np.float32)

    # modified Bessel I0 (external/cephes.c:24-103)
    r.ref_cephes_i0.restype = C.c_double
    r.ref_cephes_i0.argtypes = [C.c_double]
    xs = np.array([0, 0.1, 0.5, 1, 2, 3, 4, 5, 5.65, 6, 7, 8, 8.6, 9, 10, 12, 12.5, 15, 20, 30], np.float64)
    out["i0_x"] = xs
    out["i0"] = np.array([r.ref_cephes_i0(float(x)) for x in xs], np.float64)

    # DAS transforms (math.c:799-920) and m4_mul (math.c:448-458)
    cases = [((-60e-3, 10e-3, 0), (60e-3, 165e-3, 0), (512, 1, 1024)),      # tests/throughput.c:20-23
             ((-9.6e-3, 5e-3, 0), (9.6e-3, 45e-3, 0), (256, 256, 1)),
             ((-19e-3, -19e-3, 10e-3), (19e-3, 19e-3, 90e-3), (512, 512, 512)),
             ((1e-3, 2e-3, 3e-3), (4e-3, 6e-3, 9e-3), (1, 1, 64)),
             ((-5e-3, 0, 5e-3), (5e-3, 0, 30e-3), (1, 128, 96))]
    mats, pts = [], []
    for lo, hi, p in cases:
        m = np.zeros(16, np.float32)
        pp = (C.c_int * 3)(*p)
        r.ref_das_transform(f(np.array(lo, np.float32)), f(np.array(hi, np.float32)), pp, f(m))
        mats.append(m)
        pts.append(list(pp))
    out["das_transform_lo"] = np.array([c[0] for c in cases], np.float32)
    out["das_transform_hi"] = np.array([c[1] for c in cases], np.float32)
    out["das_transform_points_in"] = np.array([c[2] for c in cases], np.int32)
    out["das_transform_points_out"] = np.array(pts, np.int32)
    out["das_transform"] = np.array(mats, np.float32)
    r.ref_das_transform_2d.argtypes = [C.c_int, fp, fp, C.c_float, fp]
    planes = []
    for plane in range(3):
        m = np.zeros(16, np.float32)
        r.ref_das_transform_2d(plane, f(np.array([-7e-3, 4e-3], np.float32)), f(np.array([9e-3, 33e-3], np.float32)), 2.5e-3, f(m))
        planes.append(m)
    out["das_transform_2d"] = np.array(planes, np.float32)
    rng = np.random.default_rng(7)
    a = rng.normal(size=(4, 16)).astype(np.float32)
    b = rng.normal(size=(4, 16)).astype(np.float32)
    prod = np.zeros((4, 16), np.float32)
    for i in range(4):
        r.ref_m4_mul(f(a[i]), f(b[i]), f(prod[i]))
    out["m4_a"], out["m4_b"], out["m4_ab"] = a, b, prod

    # structure layout of the reference's generated/beamformer.c as compiled
    buf = C.create_string_buffer(8192)
    n = r.ref_describe_offsets(buf, 8192)
    out["struct_offsets"] = np.frombuffer(buf.raw[:n], np.uint8)
    out["struct_sizes"] = np.array([r.ref_sizeof(i) for i in range(6)], np.int32)

    # shared-memory protocol layout for the headless server (csrc/shm_server.cpp)
    buf = C.create_string_buffer(8192)
    n = r.ref_describe_shm_layout(buf, 8192)
    with open(os.path.join(HERE, "shm_layout.txt"), "w") as f:
        f.write("# shared-memory protocol v33 layout of the COMPILED reference (oracle/ref_harness.c: ref_describe_shm_layout)\n"
                "# regenerate: make -C oracle ref && python tests/golden/make_golden.py\n" + buf.raw[:n].decode())

    np.savez_compressed(os.path.join(HERE, "host_math.npz"), **out)
    print("wrote", os.path.join(HERE, "host_math.npz"), os.path.getsize(os.path.join(HERE, "host_math.npz")), "bytes")


if __name__ == "__main__":
    sys.exit(main())
