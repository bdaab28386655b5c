"""The build-defined Hilbert stage (beamformer_hip_enable_hilbert; the reference has no
implementation to be identical to: capabilities.hilbert = 0).  Off by default exactly like the
reference; when enabled, planner and kernel must agree with the oracle's statement of the same
definition, and the definition must do what its name says (analytic signal)."""
import ctypes as C

import numpy as np
import pytest

from ogl_beamforming_amd import configs as cfg, params as P
from tests.test_gpu_parity import compare, reference

S, D, I, K = P.ShaderKind, P.DataKind, P.InterpolationMode, P.AcquisitionKind
LO3, HI3 = (-2e-3, -2e-3, 3e-3), (2e-3, 2e-3, 9e-3)


def acquisitions():
    out = {}
    out["rca_i16"] = cfg.rca("rca_hilbert", 16, 3, 512, (16, 16, 1), (-2e-3, 0, 3e-3), (2e-3, 0, 9e-3), seed=71, demodulate=False)
    out["rca_f32_cubic_cw"] = cfg.rca("rca_hilbert_f32", 16, 4, 512, (12, 10, 3), LO3, HI3, seed=72, demodulate=False,
                                      data_kind=D.Float32, interp=I.Cubic, cw=True, orientation=0x12)
    out["hercules_decode"] = cfg.hercules("hercules_hilbert", 16, 8, 512, (8, 8, 6), LO3, HI3, seed=73, data_kind=D.Float16)
    out["forces_decode"] = cfg.forces("forces_hilbert", 16, 8, 512, (16, 1, 12), LO3, HI3, seed=74)
    for acq in out.values():                                  # {Decode, DAS} -> {Decode, Hilbert, DAS}
        bp = acq.bp
        assert list(bp.compute_stages[:2]) == [int(S.Decode), int(S.DAS)]
        bp.compute_stages[1], bp.compute_stages[2] = int(S.Hilbert), int(S.DAS)
        bp.compute_stages_count = 3
    return out


@pytest.fixture
def hilbert(bflib, oracle):
    assert bflib.library().beamformer_hip_enable_hilbert(1)
    oracle.enable_hilbert(True)
    yield
    bflib.library().beamformer_hip_enable_hilbert(0)
    oracle.enable_hilbert(False)


def test_stage_is_refused_by_default_as_in_the_reference(bflib, oracle):
    acq = acquisitions()["rca_i16"]
    L = bflib.library()
    assert not L.beamformer_push_simple_parameters(C.byref(acq.bp))
    assert bflib.last_error()[0] == P.LibError.InvalidComputeStage
    assert oracle.plan(acq.bp, acq.filters) is None


def test_planner_with_the_stage_enabled(bflib, oracle, hilbert):
    L = bflib.library()
    for name, acq in acquisitions().items():
        assert L.beamformer_push_simple_parameters(C.byref(acq.bp)), (name, bflib.last_error())
        ours = P.HipPlan()
        assert L.beamformer_hip_describe_plan(0, C.byref(ours)), name
        ref = oracle.plan(acq.bp, acq.filters)
        assert ref is not None and ours.stage_count == ref.stage_count, name
        kinds = [ours.stages[i].kind for i in range(ours.stage_count)]
        assert int(S.Hilbert) in kinds and ours.iq_pipeline == 1 == ref.iq_pipeline
        for i in range(ours.stage_count):
            a, b = ours.stages[i], ref.stages[i]
            assert (a.kind, a.in_kind, a.out_kind) == (b.kind, b.in_kind, b.out_kind), (name, i)
            assert list(a.in_stride) == list(b.in_stride) and list(a.out_stride) == list(b.out_stride), (name, i)
        h = ours.stages[kinds.index(int(S.Hilbert))]
        assert h.out_kind == int(D.Float32Complex) and h.in_kind in (int(D.Int16), int(D.Float16), int(D.Float32))
        assert ours.das_time_offset == pytest.approx(ref.das_time_offset, rel=1e-6)
        assert ours.das_time_offset == pytest.approx(acq.bp.time_offset + 31 / acq.bp.sampling_frequency, rel=1e-5)
    # demodulation switches the stage off (beamformer_core.c:567); complex data cannot feed it
    acq = acquisitions()["rca_i16"]
    acq.bp.compute_stages[0] = int(S.Demodulate)
    acq.filters = [cfg.kaiser_filter(acq.bp.sampling_frequency / 2, acq.bp.demodulation_frequency / 2)]
    assert L.beamformer_create_filter(C.byref(acq.filters[0]), 0, 0)
    assert L.beamformer_push_simple_parameters(C.byref(acq.bp))
    plan = P.HipPlan()
    assert L.beamformer_hip_describe_plan(0, C.byref(plan))
    assert int(S.Hilbert) not in [plan.stages[i].kind for i in range(plan.stage_count)]
    acq = acquisitions()["rca_i16"]
    acq.bp.data_kind = int(D.Float32Complex)
    assert L.beamformer_push_simple_parameters(C.byref(acq.bp))
    assert not L.beamformer_hip_describe_plan(0, C.byref(plan))
    assert oracle.plan(acq.bp, acq.filters) is None


def test_the_definition_is_an_analytic_signal(oracle):
    """x[n] = cos(2 pi f n)  ->  y[n] ~ exp(j 2 pi f (n - 31)) in the band the 63-tap transformer passes"""
    L = oracle.library()
    taps = np.zeros(126, np.float32)
    L.oracle_hilbert_fir(taps.ctypes.data_as(C.POINTER(C.c_float)))
    h = taps[0::2] + 1j * taps[1::2]
    assert h[31] == 1 and np.all(taps[0::2][np.arange(63) != 31] == 0)
    assert np.allclose(taps[1::2], -taps[1::2][::-1], atol=0)            # antisymmetric: type III
    n = np.arange(400)
    for f in (0.08, 0.2, 0.31, 0.42):
        x = np.cos(2 * np.pi * f * n)
        y = np.array([np.sum(h * x[k - 62: k + 1]) for k in range(62, 400)])
        want = np.exp(2j * np.pi * f * (n[62:] - 31))
        assert np.abs(y - want).max() < 0.02, (f, np.abs(y - want).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(acquisitions()))
def test_frame_parity_with_the_stage(name, bflib, oracle, hilbert):
    acq = acquisitions()[name]
    ref, _, flags = reference(oracle, acq)
    assert ref.dtype == np.complex64
    for path in (0, 1):
        bflib.library().beamformer_hip_set_das_path(path)
        try:
            gpu = bflib.beamform(acq.bp, acq.rf, acq.filters)
        finally:
            bflib.library().beamformer_hip_set_das_path(0)
        compare(gpu, ref, acq, flags)
