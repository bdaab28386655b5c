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


# 1001: round 4's out-of-sample fuzz draw whose median error on the gather kernel (1.4e-5: the phase of 96 turns rounded once more than the
# shader rounds it) is the size of the float oracle's own distance from its double twin -- the draw behind compare()'s median rule
@pytest.mark.parametrize("seed", list(range(72)) + [1001])
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
            assert last_das_path(bflib) == 4
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


TILE_DRAWS = []            # (seed, staged chunks, gathered chunks) of the draws below that ran das_tile.hip


def draw_tile(seed):
    """a cubic IQ acquisition the block-staged factored kernel (das_tile.hip) can take: 2-D compounding or a view plane with tx and rx
    on one axis, a thin volume, or FORCES / UFORCES; fine to moderately coarse grids (so that blocks meet chunks that fit their window and
    chunks that do not), 4-24 plane, focused or diverging transmits, ragged tiles, short rows (terms off the end of a row: the checked
    loop), f-numbers from near field to narrow apertures, with and without coherency weighting"""
    rng = np.random.default_rng(3000 + seed)
    family = str(rng.choice(["tpw", "tpw", "vls", "volume", "forces", "uforces"]))
    C = int(rng.choice([12, 16, 24, 32]))
    pitch = float(rng.choice([0.15e-3, 0.2e-3, 0.3e-3]))
    # lateral half width: voxels of 25 um ... 250 um, and in a third of the draws around 1 mm (chunks that do not fit a window)
    coarse = rng.integers(0, 3) == 0
    half = (C - 1) / 2 * pitch * (float(rng.uniform(3.0, 6.0)) if coarse else float(rng.uniform(0.15, 1.3)))
    z0 = float(rng.uniform(2e-3, 9e-3))
    z1 = z0 + float(rng.uniform(0.4e-3, 6e-3))
    samples = int(rng.choice([384, 512, 768, 1024]))
    cw = bool(rng.integers(0, 2))
    f_number = float(rng.uniform(0.3, 1.6))
    nx, ny = int(rng.integers(40, 90 if coarse else 200)), int(rng.integers(18, 70))
    if family in ("forces", "uforces"):
        if family == "uforces":
            sparse = sorted(int(v) for v in rng.choice(np.arange(C), size=int(rng.integers(5, 9)), replace=False))
            return cfg.forces(f"tile{seed}", C, len(sparse) + 1, samples, (nx, 1, ny), (-half, 0, z0), (half, 0, z1), seed=seed, kind=K.UFORCES, sparse=sparse,
                              decode=0, interp=I.Cubic, cw=cw, f_number=f_number, pitch=pitch, stages=(S.Demodulate, S.DAS))
        return cfg.forces(f"tile{seed}", C, C, samples, (nx, 1, ny), (-half, 0, z0), (half, 0, z1), seed=seed, interp=I.Cubic, cw=cw, f_number=f_number,
                          pitch=pitch, stages=(S.Demodulate, S.Decode, S.DAS))
    A = int(rng.integers(4, 25))
    depths = rng.choice([-30e-3, -12e-3, 25e-3, 60e-3, np.inf], A) if family == "vls" else None
    if family == "volume":
        points, lo, hi = (nx, ny, int(rng.integers(2, 5))), (-half, -half * 0.3, z0), (half, half * 0.3, z1)
    else:
        points, lo, hi = (nx, ny, 1), (-half, 0, z0), (half, 0, z1)
    return cfg.rca(f"tile{seed}", C, A, samples, points, lo, hi, seed=seed, orientation=int(rng.choice([0x22, 0x22, 0x11])) if family != "volume" else 0x22,
                   cw=cw, f_number=f_number, pitch=pitch, angles=np.linspace(-float(rng.uniform(2, 18)), float(rng.uniform(2, 18)), A), depths=depths,
                   kind=K.RCA_VLS if family == "vls" else K.RCA_TPW, interp=I.Cubic, data_kind=P.DataKind.Int16)


@pytest.mark.parametrize("seed", range(32))
def test_random_acquisition_on_the_block_staged_kernel(seed, bflib, oracle):
    """32 draws aimed at das_tile.hip, asked for with flag 0x100 (no channel split): against the oracle, with the counts of chunks
    served from staged windows and of chunks sent through the kernel's gather loop"""
    acq = draw_tile(seed)
    ref, pairs, flags = reference(oracle, acq)
    ok = ~np.isnan(ref)
    if not ok.any() or np.max(np.abs(ref[ok])) == 0:
        pytest.skip("empty image")
    lib = bflib.library()
    try:
        lib.beamformer_hip_set_das_path(0x10 | 0x100)
        path = bflib.describe_das(acq.bp, acq.filters)[0]
        gpu = bflib.beamform(acq.bp, acq.rf, acq.filters)
        t = last_timings(bflib)
        assert int(t.das_path) == path
    finally:
        lib.beamformer_hip_set_das_path(0)
    compare(gpu, ref, acq, flags)            # against the oracle, nothing else (round 3 accepted voxels on which another kernel agreed)
    if int(t.das_path) == 5:
        TILE_DRAWS.append((seed, int(t.tile_staged_chunks), int(t.tile_gather_chunks)))


def test_random_draws_reach_the_block_staged_kernel():
    """most draws must have run das_tile.hip, and between them both kinds of chunk in quantity"""
    print(f"block-staged draws: {len(TILE_DRAWS)}: {TILE_DRAWS}")
    assert len(TILE_DRAWS) >= 20, TILE_DRAWS
    # (draws whose rows end inside the image -- most of the coarse ones -- go to the factored kernel by the row-end rule)
    assert sum(1 for _, s, g in TILE_DRAWS if s > 0) >= 12 and sum(1 for _, s, g in TILE_DRAWS if g > 0) >= 1, TILE_DRAWS


PLANE_DRAWS = []           # (seed, das path, row-end planes) of the draws below


def draw_plane(seed):
    """a VIEW PLANE through row-column / HERCULES / FORCES data, as the reference's harness beamforms one out of every dataset
    (tests/throughput.c:443-446; math.c:844-885): one voxel along z, depth on voxel y, 56-160 voxels wide so that the aligned-grid HERCULES
    kernel and the factored kernel's band walk take it; a record that ends inside the image in about half the draws (terms at the ends
    of the RF rows: the kernels' row-end instantiations), all three interpolations, with and without coherency weighting"""
    rng = np.random.default_rng(4000 + seed)
    kind = str(rng.choice(["tpw", "tpw_swapped", "vls", "hercules", "hercules", "forces"]))
    plane = "xz" if kind == "forces" else str(rng.choice(["xz", "xz", "yz"]))
    C = int(rng.choice([16, 32]))
    A = int(rng.choice([8, 16]))
    samples = int(rng.choice([512, 768, 1024]))
    k = samples / 4096.0
    nx, ny = int(rng.choice([56, 64, 96, 128, 160])), int(rng.integers(20, 72))
    reach = float(rng.uniform(0.75, 1.15))                      # > ~0.95: the deepest rows lie beyond the record
    width = float(rng.uniform(0.5, 1.1))
    lo = (-60e-3 * k * width, -60e-3 * k * width, 10e-3 * k)
    hi = (60e-3 * k * width, 60e-3 * k * width, 165e-3 * k * reach)
    fs, fd = 20e6, 5e6
    pitch = 0.25e-3 * max(k, 64.0 / C * k)
    interp = [I.Linear, I.Cubic, I.Cubic, I.Nearest][int(rng.integers(0, 4))]
    cw = bool(rng.integers(0, 2))
    f_number = float(rng.choice([0.5, 0.5, 1.0, 1.5]))
    canonical = (S.Demodulate, S.Decode, S.DAS)
    points = (nx, ny, 1)
    if kind in ("tpw", "tpw_swapped", "vls"):
        depths = np.full(A, -40e-3 * k * float(rng.uniform(0.5, 2.0))) if kind == "vls" else None
        return cfg.rca(f"plane{seed}", C, A, samples, points, lo, hi, seed=seed, interp=interp, cw=cw, f_number=f_number, pitch=pitch, fs=fs, fd=fd,
                       orientation=0x21 if kind == "tpw_swapped" else 0x12, angles=np.linspace(-float(rng.uniform(4, 18)), float(rng.uniform(4, 18)), A),
                       depths=depths, stages=canonical, plane=plane, kind=K.RCA_VLS if kind == "vls" else K.RCA_TPW)
    if kind == "hercules":
        return cfg.hercules(f"plane{seed}", C, A, samples, points, lo, hi, seed=seed, interp=interp, cw=cw, f_number=f_number, pitch=pitch, fs=fs, fd=fd,
                            stages=canonical, plane=plane)
    return cfg.forces(f"plane{seed}", C, A, samples, points, lo, hi, seed=seed, interp=interp, cw=cw, f_number=f_number, pitch=pitch, fs=fs, fd=fd,
                      stages=canonical)


@pytest.mark.parametrize("seed", range(40))
def test_random_view_plane(seed, bflib, oracle):
    """40 random view planes against the oracle: on the automatic path, on the kernel a full-size plane of the kind gets (no channel split;
    HERCULES: the aligned-grid kernel asked for), and on the general kernel"""
    acq = draw_plane(seed)
    ref, pairs, flags = reference(oracle, acq)
    ok = ~np.isnan(ref)
    if not ok.any() or np.max(np.abs(ref[ok])) == 0:
        pytest.skip("the draw produced an empty image")
    lib = bflib.library()
    # automatic; without the channel split of small frames (0x10: the kernel a full-size plane gets); HERCULES: the aligned-grid kernel asked for
    # (6: small frames go to the general kernel's channel split by themselves); the general kernel
    modes = [0, 0x10] + ([6] if int(acq.bp.acquisition_kind) in (int(K.HERCULES), int(K.UHERCULES)) else []) + [1]
    seen = set()
    for mode in modes:
        lib.beamformer_hip_set_das_path(mode)
        try:
            gpu = bflib.beamform(acq.bp, acq.rf, acq.filters)
        finally:
            lib.beamformer_hip_set_das_path(0)
        t = last_timings(bflib)
        if (int(t.das_path), mode & 0x10) in seen and mode != 6:
            continue
        seen.add((int(t.das_path), mode & 0x10))
        PLANE_DRAWS.append((seed, int(t.das_path), int(t.das_row_end_planes)))
        compare(gpu, ref, acq, flags)


def test_random_view_planes_reach_the_plane_kernels():
    """the draws above are worth their name only if they run the kernels they aim at"""
    if len(PLANE_DRAWS) < 30:
        pytest.skip("needs the draws of this module in the same session")
    paths = {p for _, p, _ in PLANE_DRAWS}
    assert int(P.DasPath.Hercules) in paths and int(P.DasPath.Factored) in paths, sorted(PLANE_DRAWS)


# ---- row ends (csrc/das_exact.h).  sample_rf's range test is a step: round 3's fast kernels formed the index as a rounded receive term plus
# a rounded transmit term and kept or dropped a term within an ulp of the end of an RF row differently from the oracle -- one whole tap of
# difference at a voxel.  These are the draws of round 3's out-of-sample fuzz (tools/auto_fuzz.py 72 600, tools/tile_fuzz.py) that failed
# for it, fixed here as cases: every kernel that can take the draw, against the oracle.
ROW_END_SEPARABLE = [96, 107, 112, 114, 120, 125, 130, 142, 160, 194, 201, 237, 241, 254, 259, 261, 268, 305, 318, 319, 343, 346, 361, 367, 385,
                     398, 434, 439, 480, 495, 547, 584, 593]


@pytest.mark.parametrize("seed", ROW_END_SEPARABLE)
def test_row_end_draws_of_the_separable_generator(seed, bflib, oracle, hooks):
    """automatic path, the general kernel, the gather kernel (never staged), the LDS-staged kernel wherever its window holds (every term
    range-checked) and the factored kernel (block staging on request): the oracle's frame from each"""
    acq = draw_separable(seed)
    ref, pairs, flags = reference(oracle, acq)
    lib = bflib.library()
    ran = {}
    for mode in (0, 1, 2, 3, 0x14, 0x114):
        if mode == 3:
            hooks.set("STAGED_CHECKED")
        lib.beamformer_hip_set_das_path(mode)
        try:
            gpu = bflib.beamform(acq.bp, acq.rf, acq.filters)
            t = last_timings(bflib)
            ran[mode] = (int(t.das_path), int(t.das_row_end_planes))
            assert int(t.staged_window_violations) == 0
        finally:
            lib.beamformer_hip_set_das_path(0)
            if mode == 3:
                hooks.clear("STAGED_CHECKED")
        compare(gpu, ref, acq, flags)
    assert ran[1][0] == 0, ran


@pytest.mark.parametrize("seed", [540, 11, 63, 131, 207])
def test_row_end_draws_of_the_tile_generator(seed, bflib, oracle):
    """540: the draw on which das_tile.hip's window-relative position arithmetic flipped ALONE in round 3; block staging asked for, the
    factored kernel and the general kernel: the oracle's frame from each"""
    acq = draw_tile(seed)
    ref, pairs, flags = reference(oracle, acq)
    ok = ~np.isnan(ref)
    if not ok.any() or np.max(np.abs(ref[ok])) == 0:
        pytest.skip("empty image")
    lib = bflib.library()
    for mode in (0x110, 0x210, 0x11):
        lib.beamformer_hip_set_das_path(mode)
        try:
            gpu = bflib.beamform(acq.bp, acq.rf, acq.filters)
        finally:
            lib.beamformer_hip_set_das_path(0)
        compare(gpu, ref, acq, flags)
