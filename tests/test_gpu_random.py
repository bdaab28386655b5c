"""Randomised parity: 48 seeded small acquisitions drawn over family x geometry x interpolation
x element kind x stages x f-number x coherency weighting x per-transmit orientations / focal
depths, each beamformed through the C ABI on the automatic DAS path and compared with the
oracle (same tolerances as tests/test_gpu_parity.py).  Complements the named cases: the
combinations here are not hand-picked."""
import numpy as np
import pytest

from ogl_beamforming_amd import configs as cfg, params as P
from tests import cases
from tests.test_gpu_parity import compare, last_das_path, last_timings, reference

pytestmark = pytest.mark.gpu
S, D, I, K = P.ShaderKind, P.DataKind, P.InterpolationMode, P.AcquisitionKind


def draw(seed):
    rng = np.random.default_rng(1000 + seed)
    pick = lambda *v: v[int(rng.integers(0, len(v)))]
    interp = pick(I.Nearest, I.Linear, I.Linear, I.Cubic)
    cw = bool(rng.integers(0, 2))
    f_number = pick(0.0, 0.5, 1.0, 2.0)
    C = int(pick(8, 12, 16, 24, 40))
    samples = int(pick(256, 384, 512))
    family = pick("rca2d", "rca3d", "rca3d", "vls", "hercules", "forces", "uforces")
    common = dict(seed=seed, interp=interp, cw=cw, f_number=f_number)
    path = 0.40 * samples / 25e6 * 1540.0
    z0, z1 = 0.15 * path, 0.40 * path
    if family in ("rca2d", "rca3d", "vls"):
        A = int(pick(1, 2, 3, 5, 9))
        kind = pick(D.Int16, D.Float16, D.Float32, D.Int16Complex, D.Float32Complex)
        demod = kind in (D.Int16, D.Float16, D.Float32) and bool(rng.integers(0, 2))
        if family == "rca2d":
            points, lo, hi, orientation = (int(pick(12, 20, 33)), int(pick(12, 17)), 1), (-2e-3, 0, z0), (2e-3, 0, z1), 0x22
        else:
            points = (int(pick(6, 9, 16)), int(pick(6, 10)), int(pick(3, 5)))
            lo, hi, orientation = (-2e-3, -2e-3, z0), (2e-3, 2e-3, z1), pick(0x12, 0x21)
        depths = None
        if family == "vls":
            depths = rng.uniform(1.5 * z1, 4.0 * z1, A) * rng.choice([-1.0, 1.0], A)
        acq = cfg.rca(f"random{seed}", C, A, samples, points, lo, hi, data_kind=kind, orientation=orientation,
                      demodulate=demod, depths=depths, angles=rng.uniform(-12, 12, A) if A > 1 else None,
                      kind=K.RCA_VLS if family == "vls" else K.RCA_TPW, **common)
        bp = acq.bp
        if family == "rca3d" and A > 1 and rng.integers(0, 3) == 0:
            # per-transmit TRANSMIT orientation varies (receive fixed): still factorises
            for a in range(A):
                tx = int(pick(1, 2, 0))
                bp.transmit_receive_orientations[a] = (tx << 4) | (orientation & 0xF)
        elif family == "rca3d" and A > 1 and rng.integers(0, 4) == 0:
            # receive orientation varies too: general kernel only
            for a in range(A):
                bp.transmit_receive_orientations[a] = int(pick(0x12, 0x21))
        return acq
    A = int(pick(4, 8, 12, 16))
    kind = pick(D.Int16, D.Float16, D.Float32)
    stages = pick((S.Decode, S.DAS), (S.Demodulate, S.Decode, S.DAS))
    if family == "hercules":
        return cfg.hercules(f"random{seed}", C, A, samples, (int(pick(6, 9)), int(pick(6, 8)), int(pick(4, 6))),
                            (-1.5e-3, -1.5e-3, z0), (1.5e-3, 1.5e-3, z1), data_kind=kind, stages=stages,
                            orientation=pick(0x12, 0x21), focal=pick((0.0, np.inf), (0.0, -4.0 * z1), (4.0, np.inf)), **common)
    sparse = None
    akind = K.FORCES
    if family == "uforces":
        akind = K.UFORCES
        sparse = np.sort(rng.choice(max(C, A), A - 1, replace=False))
    return cfg.forces(f"random{seed}", C, A, samples, (int(pick(12, 20, 31)), 1, int(pick(10, 16))), (-2e-3, 0, z0), (2e-3, 0, z1),
                      data_kind=kind, stages=stages, kind=akind, sparse=sparse, **common)


STAGED_DRAWS = []          # seeds whose forced-staged pass ran the LDS-staged kernel (reported by the last test of this module)


@pytest.mark.parametrize("seed", range(72))
def test_random_acquisition(seed, bflib, oracle, hooks):
    acq = draw(seed)
    ref, pairs, flags = reference(oracle, acq)
    ok = ~np.isnan(ref)
    if not ok.any() or np.max(np.abs(ref[ok])) == 0:
        pytest.skip("the draw produced an empty image (aperture closed everywhere)")
    bflib.library().beamformer_hip_set_das_path(0)
    gpu = bflib.beamform(acq.bp, acq.rf, acq.filters)
    compare(gpu, ref, acq, flags)
    # and the general kernel on the same input, whatever the automatic choice was
    first_path = last_das_path(bflib)
    if first_path != 0:
        bflib.library().beamformer_hip_set_das_path(1)
        try:
            gpu = bflib.beamform(acq.bp, acq.rf, acq.filters)
        finally:
            bflib.library().beamformer_hip_set_das_path(0)
        compare(gpu, ref, acq, flags)
    # row-column draws the gather kernel took also go through the LDS-staged kernel (path 3; it declines -- and the gather
    # kernel runs again -- when the interpolation is not linear, the data real or the delay spread too wide for a window)
    if first_path in (1, 2):
        bflib.library().beamformer_hip_set_das_path(3)
        hooks.set("STAGED_CHECKED")          # every term range-checked: a position outside its staged window is counted
        try:
            gpu = bflib.beamform(acq.bp, acq.rf, acq.filters)
            t = last_timings(bflib)
            assert int(t.das_path) in (1, 2)
            assert int(t.staged_window_violations) == 0, "a term left its staged window: plan_staged's bound is wrong"
            if int(t.das_path) == 2:
                STAGED_DRAWS.append(seed)
        finally:
            hooks.clear("STAGED_CHECKED")
            bflib.library().beamformer_hip_set_das_path(0)
        compare(gpu, ref, acq, flags)
    # HERCULES-family draws also go through the aligned-grid kernel (forced: these grids are narrower than
    # the automatic rule asks for), whichever loop order and sparsity the draw produced
    if P.AcquisitionKind(acq.bp.acquisition_kind) in (P.AcquisitionKind.HERCULES, P.AcquisitionKind.UHERCULES, P.AcquisitionKind.HERO_PA):
        bflib.library().beamformer_hip_set_das_path(6)
        try:
            gpu = bflib.beamform(acq.bp, acq.rf, acq.filters)
            assert last_das_path(bflib) == 5
        finally:
            bflib.library().beamformer_hip_set_das_path(0)
        compare(gpu, ref, acq, flags)



def draw_separable(seed):
    """a row-column acquisition the LDS-staged kernels can take: receive and transmit on different axes, 6-20 transmits, plane or
    focused / diverging waves, ragged grids, linear / cubic interpolation, IQ or real samples (the generator of the CPU property
    test tests/test_das_select.py, with RF)"""
    rng = np.random.default_rng(2000 + seed)
    C = int(rng.choice([16, 32, 48]))
    A = int(rng.integers(6, 20))
    focused = bool(rng.integers(0, 2))
    mode = int(rng.integers(0, 4))                       # 0, 1: IQ linear; 2: IQ cubic; 3: real linear
    pitch = float(rng.choice([0.15e-3, 0.2e-3, 0.3e-3]))
    half = (C - 1) / 2 * pitch * float(rng.uniform(0.6, 2.0))
    z0 = float(rng.uniform(3e-3, 10e-3))
    z1 = z0 + float(rng.uniform(2e-3, 8e-3))
    points = (int(rng.integers(20, 110)), int(rng.integers(20, 70)), int(rng.integers(1, 3)) + 1)
    depths = rng.choice([-30e-3, -12e-3, 25e-3, 60e-3, np.inf], A) if focused else None
    return cfg.rca(f"staged{seed}", C, A, int(rng.choice([512, 1024, 2048])), points, (-half, -half * float(rng.uniform(0.5, 1.2)), z0), (half, half, z1),
                   seed=seed, orientation=int(rng.choice([0x12, 0x21])), cw=bool(rng.integers(0, 2)), f_number=float(rng.uniform(0.3, 1.5)),
                   pitch=pitch, angles=np.linspace(-float(rng.uniform(2, 20)), float(rng.uniform(2, 20)), A), depths=depths,
                   kind=K.RCA_VLS if focused else K.RCA_TPW, interp=I.Cubic if mode == 2 else I.Linear,
                   demodulate=mode != 3, data_kind=P.DataKind.Int16)


@pytest.mark.parametrize("seed", range(32))
def test_random_separable_acquisition_on_the_staged_kernels(seed, bflib, oracle, hooks):
    """32 draws aimed at the LDS-staged kernels (the 72 general draws above reach them once): automatic path, every term
    range-checked with the window-violation count on, against the oracle"""
    acq = draw_separable(seed)
    ref, pairs, flags = reference(oracle, acq)
    ok = ~np.isnan(ref)
    if not ok.any() or np.max(np.abs(ref[ok])) == 0:
        pytest.skip("empty image")
    bflib.library().beamformer_hip_set_das_path(0)
    path = bflib.describe_das(acq.bp, acq.filters)[0]
    gpu = bflib.beamform(acq.bp, acq.rf, acq.filters)
    assert last_das_path(bflib) == path
    compare(gpu, ref, acq, flags)
    if path == 2:
        hooks.set("STAGED_CHECKED")
        checked = bflib.beamform(acq.bp, acq.rf, acq.filters)
        t = last_timings(bflib)
        assert int(t.das_path) == 2 and int(t.staged_window_violations) == 0, "a term left its staged window: plan_staged's bound is wrong"
        compare(checked, ref, acq, flags)
        STAGED_DRAWS.append(100 + seed)


def test_random_draws_reach_the_staged_kernel():
    """how many draws exercised the LDS-staged kernels with the window-violation count on (none may be zero by luck)"""
    print(f"staged draws: {len(STAGED_DRAWS)}: {STAGED_DRAWS}")
    assert len(STAGED_DRAWS) >= 12, STAGED_DRAWS
