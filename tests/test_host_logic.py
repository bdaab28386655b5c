"""Host logic of the C ABI that needs no GPU: parameter validation in the reference's order
and with its error codes (lib/ogl_beamformer_lib.c:252-311, :438-456, :509-511), the planner
(beamformer_core.c:553-1013) against SURVEY's worked examples and the oracle's restatement."""
import ctypes as C

import numpy as np
import pytest

from ogl_beamforming_amd import configs as cfg
from ogl_beamforming_amd import params as P
from tests import cases

E = P.LibError
S = P.ShaderKind
D = P.DataKind


@pytest.fixture()
def L(bflib):
    lib = bflib.library()
    lib.beamformer_reserve_parameter_blocks(1)
    return lib


def err(bflib):
    return bflib.last_error()[0]


def base_parameters():
    return cfg.config(1, 0.25).bp


def test_validate_parameters_order(L, bflib):
    bp = base_parameters()
    assert L.beamformer_push_simple_parameters(C.byref(bp))
    p = P.Parameters.from_buffer_copy(bytes(bp)[:264])
    p.contrast_mode = 7
    assert not L.beamformer_push_parameters(C.byref(p)) and err(bflib) == E.InvalidContrastMode
    p.contrast_mode = 1                                   # A1S2 needs 3x the samples per row
    assert not L.beamformer_push_parameters(C.byref(p)) and err(bflib) == E.DataSizeMismatch
    p.contrast_mode = 0
    p.raw_data_dimensions[0] = p.sample_count * p.acquisition_count - 1
    assert not L.beamformer_push_parameters(C.byref(p)) and err(bflib) == E.DataSizeMismatch
    p.raw_data_dimensions[0] = p.sample_count * p.acquisition_count
    p.output_points[:] = [1024, 1024, 1024, 1]            # 8 GiB frame > 4 GiB ring
    assert not L.beamformer_push_parameters(C.byref(p)) and err(bflib) == E.FrameSizeOverflow
    assert L.beamformer_maximum_frames_for_parameters(C.byref(p)) == 2**64 - 1
    p.output_points[:] = [512, 512, 512, 1]
    p.coherency_weighting = 0
    assert L.beamformer_maximum_frames_for_parameters(C.byref(p)) == 4     # 4 GiB ring / 1 GiB
    p.coherency_weighting = 1
    assert L.beamformer_maximum_frames_for_parameters(C.byref(p)) == 3     # lib .c:262-273


def test_validate_pipeline_order(L, bflib):
    def push(stages, kind):
        arr = (C.c_int32 * len(stages))(*[int(s) for s in stages])
        return L.beamformer_push_pipeline(arr, len(stages), int(kind))
    assert not push([S.Demodulate, S.DAS], 9) and err(bflib) == E.InvalidDataKind
    assert not push([S.Demodulate] * 17, D.Int16) and err(bflib) == E.ComputeStageOverflow
    assert not push([S.Demodulate, S.Reshape], D.Int16) and err(bflib) == E.InvalidComputeStage
    assert not push([S.Decode, S.Hilbert, S.DAS], D.Int16) and err(bflib) == E.InvalidComputeStage   # capabilities.hilbert = 0
    assert not push([S.Demodulate, S.DAS], D.Int16Complex) and err(bflib) == E.InvalidDemodulationDataKind
    assert not push([S.DAS], D.Int16) and err(bflib) == E.InvalidStartShader
    assert not push([S.Filter, S.DAS], D.Int16) and err(bflib) == E.InvalidStartShader
    assert push([S.Demodulate, S.Decode, S.DAS], D.Int16)
    assert push([S.Decode, S.DAS], D.Float32Complex)


def test_block_and_array_limits(L, bflib):
    assert not L.beamformer_reserve_parameter_blocks(17) and err(bflib) == E.ParameterBlockOverflow
    assert L.beamformer_reserve_parameter_blocks(2)
    bp = base_parameters()
    assert L.beamformer_push_simple_parameters_at(C.byref(bp), 1)
    assert not L.beamformer_push_simple_parameters_at(C.byref(bp), 2) and err(bflib) == E.ParameterBlockUnallocated
    mapping = (C.c_int16 * 300)()
    assert not L.beamformer_push_channel_mapping(mapping, 257) and err(bflib) == E.BufferOverflow
    assert L.beamformer_push_channel_mapping(mapping, 256)
    fv = (C.c_float * 600)()
    assert not L.beamformer_push_focal_vectors(fv, 257) and err(bflib) == E.BufferOverflow
    assert L.beamformer_push_focal_vectors(fv, 256)
    f = P.FilterParameters()
    f.kind = 5
    assert not L.beamformer_create_filter(C.byref(f), 0, 0) and err(bflib) == E.InvalidFilterKind
    assert L.beamformer_reserve_parameter_blocks(1)


def test_live_parameters_roundtrip(L):
    lp = P.LiveImagingParameters()
    lp.active = 1
    lp.transmit_power = 0.5
    assert L.beamformer_set_live_parameters(C.byref(lp))
    back = L.beamformer_get_live_parameters().contents
    assert back.active == 1 and back.transmit_power == 0.5
    assert L.beamformer_live_parameters_get_dirty_flag() == -1


def test_compute_fails_loudly_without_a_device(L, bflib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    acq = cfg.config(1, 0.25)
    assert L.beamformer_push_simple_parameters(C.byref(acq.bp))
    rf = np.ascontiguousarray(acq.rf)
    assert not L.beamformer_push_data_with_compute(rf.ctypes.data_as(C.c_void_p), rf.nbytes, 0, 0)
    assert err(bflib) == E.SharedMemory                    # no CPU fallback
    out = np.zeros(16, np.float32)
    assert not L.beamformer_get_last_frames(out.ctypes.data_as(C.c_void_p), out.nbytes, 1)
    assert L.beamformer_maximum_rf_data_size() == 2**64 - 1


def describe(L, acq):
    for slot, fp in enumerate(acq.filters):
        assert L.beamformer_create_filter(C.byref(fp), slot, 0)
    assert L.beamformer_push_simple_parameters(C.byref(acq.bp))
    plan = P.HipPlan()
    assert L.beamformer_hip_describe_plan(0, C.byref(plan))
    return plan


def test_planner_worked_example_a(L):
    """SURVEY section 8a example A: Int16, {Demodulate, Decode, DAS}, Hadamard, C 256, A 128, S 4096"""
    acq = cfg.hercules("example_a", 256, 128, 4096, (8, 8, 8), (-1e-3, -1e-3, 5e-3), (1e-3, 1e-3, 9e-3), seed=1,
                       stages=(S.Demodulate, S.Decode, S.DAS))
    plan = describe(L, acq)
    kinds = [plan.stages[i].kind for i in range(plan.stage_count)]
    assert kinds == [S.Demodulate, S.Decode, S.DAS]
    demod, decode, das = plan.stages[0], plan.stages[1], plan.stages[2]
    assert (demod.in_kind, demod.out_kind) == (D.Int16Complex, D.Float16Complex)
    assert list(demod.in_stride) == [1, 4096 * 128, 4096]            # raw real-sample units
    assert list(demod.out_stride) == [256 * 128, 128, 1]             # [sample][channel][transmit], chunk = all channels
    assert (decode.in_kind, decode.out_kind) == (D.Float16Complex, D.Float32Complex)
    assert list(decode.out_stride) == [1, 2048 * 128, 2048]
    assert (das.in_kind, das.out_kind) == (D.Float32Complex, D.Float32Complex)
    assert plan.das_samples == 2048 and plan.iq_pipeline == 1
    assert plan.das_sampling_frequency == pytest.approx(12.5e6)
    assert plan.das_time_offset == pytest.approx(36 / 2 / 12.5e6)     # Kaiser length 36 at fs/2


def test_planner_worked_examples_b_c(L):
    # B: Int16, {Decode, DAS}, decode None -> Reshape(i16 -> f32) -> DAS real
    acq = cfg.rca("example_b", 32, 4, 512, (8, 8, 1), (-1e-3, 0, 5e-3), (1e-3, 0, 9e-3), seed=1, demodulate=False)
    plan = describe(L, acq)
    assert [plan.stages[i].kind for i in range(plan.stage_count)] == [S.Reshape, S.DAS]
    r = plan.stages[0]
    assert (r.in_kind, r.out_kind) == (D.Int16, D.Float32)
    assert list(r.in_stride) == list(r.out_stride) == [1, 512 * 4, 512]
    assert plan.iq_pipeline == 0
    # C: Float16, {Demodulate, DAS}: filter writes f32 complex straight in DAS layout
    acq = cfg.config(2, 0.0625)
    plan = describe(L, acq)
    assert [plan.stages[i].kind for i in range(plan.stage_count)] == [S.Demodulate, S.DAS]
    assert (plan.stages[0].in_kind, plan.stages[0].out_kind) == (D.Float16Complex, D.Float32Complex)
    # coherency weighting adds the implicit node after DAS (beamformer_core.c:676-678)
    plan = describe(L, cfg.config(4, 0.0625))
    assert [plan.stages[i].kind for i in range(plan.stage_count)] == [S.Demodulate, S.DAS, S.CoherencyWeighting]


@pytest.mark.parametrize("name", sorted(cases.CASES))
def test_planner_agrees_with_oracle(name, L, oracle):
    """stage order, element kinds and DAS parameters equal the oracle's restatement of
    plan_compute_pipeline; strides equal wherever the 16-channel chunk does not enter."""
    acq = cases.make(name)
    ours = describe(L, acq)
    ref = oracle.plan(acq.bp, acq.filters)
    assert ref is not None
    assert ours.stage_count == ref.stage_count
    C_ = acq.bp.channel_count
    chunk = min(C_, 16)
    for i in range(ours.stage_count):
        a, b = ours.stages[i], ref.stages[i]
        assert (a.kind, a.in_kind, a.out_kind) == (b.kind, b.in_kind, b.out_kind), (name, i)
        for mine, theirs in ((a.in_stride, b.in_stride), (a.out_stride, b.out_stride)):
            for k in range(3):
                if theirs[k] % chunk == 0 and mine[k] != theirs[k] and chunk != C_:
                    assert mine[k] == theirs[k] // chunk * C_          # Decode layout scales with the chunk
                else:
                    assert mine[k] == theirs[k], (name, i, list(mine), list(theirs))
    assert ours.das_samples == ref.input_sample_count
    assert ours.das_time_offset == pytest.approx(ref.das_time_offset, rel=1e-6, abs=1e-12)
    assert np.allclose(np.array(ours.das_voxel_transform[:]), np.array(ref.das_voxel_transform[:]), rtol=1e-6, atol=1e-12)


def test_planner_agrees_with_oracle_on_random_pipelines(L, oracle):
    """Differential test of the two independent restatements of plan_compute_pipeline
    (csrc/planner.cpp, oracle/oracle_plan.c) over 1500 random client-valid pipelines: any data
    kind, repeated and oddly ordered stages, decode on/off, decimation, sampling mode, coherency
    weighting.  16 channels = one reference chunk, so strides must agree exactly."""
    rng = np.random.default_rng(7)
    base = cfg.rca("fz", 16, 4, 256, (8, 8, 1), (-1e-3, 0, 5e-3), (1e-3, 0, 9e-3), seed=1)
    pool = [int(S.Decode), int(S.Filter), int(S.Demodulate), int(S.DAS)]
    filters = [base.filters[0], base.filters[0]]
    for slot, fp in enumerate(filters):
        assert L.beamformer_create_filter(C.byref(fp), slot, 0)
    for it in range(1500):
        bp = P.SimpleParameters.from_buffer_copy(bytes(base.bp))
        stages = [int(rng.choice([int(S.Decode), int(S.Demodulate)]))] + [int(rng.choice(pool)) for _ in range(int(rng.integers(0, 5)))]
        stages.append(int(S.DAS))
        bp.data_kind = int(rng.integers(0, 6))
        if bp.data_kind in (int(D.Int16Complex), int(D.Float32Complex), int(D.Float16Complex)):   # lib .c:285-296
            stages = [s if s != int(S.Demodulate) else int(S.Filter) for s in stages]
            stages[0] = int(S.Decode)
        bp.compute_stages_count = len(stages)
        for i, s in enumerate(stages):
            bp.compute_stages[i] = s
            bp.compute_stage_parameters[i] = int(rng.integers(0, 2))
        bp.decode_mode = int(rng.integers(0, 2))
        bp.decimation_rate = int(rng.integers(0, 4))
        bp.sampling_mode = int(rng.integers(0, 2))
        bp.coherency_weighting = int(rng.integers(0, 2))
        bp.acquisition_count = int(rng.choice([1, 2, 4, 8, 12]))
        bp.raw_data_dimensions[0] = bp.sample_count * bp.acquisition_count
        assert L.beamformer_push_simple_parameters(C.byref(bp)), (it, stages)
        ours = P.HipPlan()
        assert L.beamformer_hip_describe_plan(0, C.byref(ours)), (it, stages)
        ref = oracle.plan(bp, filters)
        assert ref is not None and ours.stage_count == ref.stage_count, (it, stages)
        for i in range(ours.stage_count):
            a, b = ours.stages[i], ref.stages[i]
            assert (a.kind, a.in_kind, a.out_kind) == (b.kind, b.in_kind, b.out_kind), (it, stages, i)
            assert list(a.in_stride) == list(b.in_stride) and list(a.out_stride) == list(b.out_stride), (it, stages, i)
        assert ours.das_samples == ref.input_sample_count and ours.iq_pipeline == ref.iq_pipeline
        assert ours.das_time_offset == pytest.approx(ref.das_time_offset, rel=1e-6, abs=1e-12)


def test_pipeline_without_das_is_planned(L, bflib):
    """tests/decode.c of the reference pushes {Decode} alone (:236-238): legal, the frame stays
    zero.  The last stage gets the layout and kind DAS would have asked for."""
    bp = P.Parameters()
    bp.decode_mode = 1                                                # BeamformerDecodeMode_Hadamard
    bp.sample_count, bp.channel_count, bp.acquisition_count = 4096, 12, 12
    bp.raw_data_dimensions[0], bp.raw_data_dimensions[1] = 4096 * 12, 256
    assert L.beamformer_push_parameters(C.byref(bp)), bflib.last_error()
    mapping = (C.c_int16 * 256)(*[(i * 167 + 13) & 255 for i in range(256)])
    assert L.beamformer_push_channel_mapping(mapping, 256)
    stage = (C.c_int32 * 1)(int(S.Decode))
    assert L.beamformer_push_pipeline(stage, 1, int(D.Int16)), bflib.last_error()
    plan = P.HipPlan()
    assert L.beamformer_hip_describe_plan(0, C.byref(plan))
    kinds = [(plan.stages[i].kind, plan.stages[i].in_kind, plan.stages[i].out_kind) for i in range(plan.stage_count)]
    assert kinds == [(S.Reshape, D.Int16, D.Int16), (S.Decode, D.Int16, D.Float32)]
    assert list(plan.stages[1].in_stride) == [12 * 12, 12, 1]                 # decode layout (beamformer_core.c:645-664)
    assert list(plan.stages[1].out_stride) == [1, 4096 * 12, 4096]            # what DAS would read
    # decode switched off: nothing to run at all
    bp.decode_mode = 0
    assert L.beamformer_push_parameters(C.byref(bp))
    assert L.beamformer_hip_describe_plan(0, C.byref(plan))
    assert [plan.stages[i].kind for i in range(plan.stage_count)] in ([], [S.Reshape])


@pytest.mark.parametrize("demodulate", [False, True])
def test_planner_refuses_rows_shorter_than_the_interpolation_support(demodulate, L):
    """A DAS input row shorter than the taps of its interpolation (1 nearest, 2 linear, 4 cubic;
    sample_rf, das.glsl:99-124) has no valid sample index, and the kernels' unsigned range tests
    against S-1 / S-3 would wrap: the plan is refused instead of launched."""
    I = P.InterpolationMode
    for interp, support in ((I.Nearest, 1), (I.Linear, 2), (I.Cubic, 4)):
        for das_samples in (1, 2, 3, 4):
            raw = das_samples * 2 if demodulate else das_samples
            acq = cfg.rca("short", 16, 2, 256, (8, 8, 1), (-1e-3, 0, 5e-3), (1e-3, 0, 9e-3), seed=1,
                          demodulate=demodulate, interp=interp)
            bp = acq.bp
            bp.sample_count = raw
            bp.raw_data_dimensions[0] = raw * bp.acquisition_count
            for slot, fp in enumerate(acq.filters):
                assert L.beamformer_create_filter(C.byref(fp), slot, 0)
            assert L.beamformer_push_simple_parameters(C.byref(bp))
            plan = P.HipPlan()
            ok = bool(L.beamformer_hip_describe_plan(0, C.byref(plan)))
            assert ok == (das_samples >= support), (interp, das_samples, demodulate)
            if ok:
                assert plan.das_samples == das_samples
    # decimation that leaves no sample at all
    acq = cfg.rca("short", 16, 2, 256, (8, 8, 1), (-1e-3, 0, 5e-3), (1e-3, 0, 9e-3), seed=1, interp=I.Nearest)
    acq.bp.sample_count = 4
    acq.bp.decimation_rate = 4
    acq.bp.raw_data_dimensions[0] = 4 * acq.bp.acquisition_count
    assert L.beamformer_create_filter(C.byref(acq.filters[0]), 0, 0)
    assert L.beamformer_push_simple_parameters(C.byref(acq.bp))
    assert not L.beamformer_hip_describe_plan(0, C.byref(P.HipPlan()))


def test_planner_refuses_wild_decimation_and_readi_groups(L):
    """found by tests/plan_fuzz.cpp: a decimation rate whose doubling wraps to 0 divided by zero, and an
    unchecked readi_group_count sized a 17 GB Hadamard matrix (and readi_group indexed past it on the device)"""
    acq = cfg.rca("wild", 16, 2, 256, (8, 8, 1), (-1e-3, 0, 5e-3), (1e-3, 0, 9e-3), seed=1)
    assert L.beamformer_create_filter(C.byref(acq.filters[0]), 0, 0)
    for rate, ok in ((0x80000000, False), (257, False), (4, True)):
        acq.bp.decimation_rate = rate
        assert L.beamformer_push_simple_parameters(C.byref(acq.bp))
        assert bool(L.beamformer_hip_describe_plan(0, C.byref(P.HipPlan()))) == ok, rate
    f = cfg.forces("wild_readi", 16, 4, 512, (8, 1, 8), (-1e-3, 0, 5e-3), (1e-3, 0, 9e-3), seed=2, decode=0, readi_groups=4, readi_group=2)
    for groups, group, ok in ((4, 2, True), (4, 4, False), (65536, 0, False), (0x80000000, 0, False), (300, 1, False)):
        f.bp.readi_group_count, f.bp.readi_group = groups, group
        assert L.beamformer_push_simple_parameters(C.byref(f.bp))
        assert bool(L.beamformer_hip_describe_plan(0, C.byref(P.HipPlan()))) == ok, (groups, group)


def test_planner_survives_fuzzing(tmp_path):
    """csrc/planner.cpp + csrc/host_math.cpp under AddressSanitizer + UBSan (CPU build; the GPU pool has no
    sanitizers): 300 000 parameter blocks with boundary and wild values in every field a client can write.
    Every block is planned or refused without a sanitizer report, and accepted plans keep the invariants the
    kernels rely on (tests/plan_fuzz.cpp)."""
    import os, shutil, subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    here = os.path.dirname(os.path.abspath(__file__))
    exe = tmp_path / "plan_fuzz"
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize=enum",
                            "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                            os.path.join(here, "plan_fuzz.cpp"), "-o", str(exe)], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-2000:]
    for seed in ("11", "12", "13"):
        run = subprocess.run([str(exe), "100000", seed], capture_output=True, text=True, timeout=600)
        assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
        accepted, refused = (int(v) for v in run.stdout.split()[1::2])
        assert accepted > 5000 and refused > 5000                        # both outcomes were exercised
