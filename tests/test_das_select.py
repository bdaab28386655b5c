"""The DAS kernel selection (csrc/das_select.cpp) on the CPU: beamformer_hip_describe_das needs no device.
(1) the named cases get the kernel written down for them (tests/cases.py EXPECTED_AUTOMATIC) and every kernel not taken says why;
(2) PROPERTY: whenever the LDS-staged kernel is chosen, the delay spread of EVERY tile fits the window the host chose -- brute force
    in float64 over random row-column geometries (plane and focused transmits, ragged grids), restating only the index formula of
    das.glsl:126-130, :187-231 and the kernel's window construction (das_staged.hip: element floor(min R) + floor(min T) + j)."""
import numpy as np
import pytest

from ogl_beamforming_amd import configs as cfg
from ogl_beamforming_amd import lib
from ogl_beamforming_amd import params as P
from tests import cases


@pytest.mark.parametrize("name", sorted(cases.EXPECTED_AUTOMATIC))
def test_automatic_selection_of_the_named_cases(name):
    acq = cases.make(name)
    lib.library().beamformer_hip_set_das_path(0)
    path, kernel, label, reasons, d = lib.describe_das(acq.bp, acq.filters)
    assert path == cases.EXPECTED_AUTOMATIC[name], (kernel, reasons)
    assert int(d.row_end_planes) == cases.EXPECTED_ROW_END_PLANES.get(name, 0)
    assert kernel.startswith("das_") and label
    assert not reasons[path]
    assert all(reasons[k] for k in (1, 2, 3, 4) if k != path), reasons


def test_forced_paths_and_reasons():
    L = lib.library()
    acq = cases.make("rca_staged_auto")
    try:
        L.beamformer_hip_set_das_path(1)
        path, _, _, reasons, _ = lib.describe_das(acq.bp, acq.filters)
        assert path == 0 and "general kernel was asked for" in reasons[2]
        L.beamformer_hip_set_das_path(2)
        path, _, _, reasons, _ = lib.describe_das(acq.bp, acq.filters)
        assert path == 1 and "no LDS staging" in reasons[2]
        L.beamformer_hip_set_das_path(4)
        assert lib.describe_das(acq.bp, acq.filters)[0] == 3
        L.beamformer_hip_set_das_path(0)
        # the uniform-table form falls back to the tables in LDS when the global table may not be allocated
        fine = cases.make("rca_staged_fine")
        _, _, _, _, d = lib.describe_das(fine.bp, fine.filters)
        assert d.uniform_tables == 1 and d.u_shift == 6 and d.v_shift == 4
        lib.set_hook("STAGED_TABLE_CAP", "1")
        path, _, _, _, d = lib.describe_das(fine.bp, fine.filters)
        assert path == 2 and d.uniform_tables == 0
    finally:
        lib.set_hook("STAGED_TABLE_CAP", None)
        L.beamformer_hip_set_das_path(0)


def test_block_staged_kernel_selection():
    """das_tile.hip (path 5): automatic for BASELINE config 2 at full size (fine grid, cubic IQ, tx and rx on one axis: 64 x 16 tiles,
    32-sample windows -- the derivative bound says 27.9 samples, the spread sampled on the image's extreme tiles 26 --, the banded plane walk), declined with its reason on small frames (channel split), under flag 0x200, for other
    sample kinds; flag 0x100 asks for it wherever the kernel is able to run, with the 32-sample window on very fine grids."""
    L = lib.library()
    try:
        L.beamformer_hip_set_das_path(0)
        full = cfg.config(2)
        path, kernel, _, reasons, d = lib.describe_das(full.bp, full.filters)
        assert (path, kernel) == (5, "das_tile_kernel") and "block-staged" in reasons[3]
        assert list(d.tile_shift) == [6, 4, 0] and list(d.blocks) == [16, 64, 1] and d.tile_window_samples == 32 and d.tile_walk == 3
        assert 20.0 <= d.tile_spread_estimate <= 26.0 and list(d.tile_estimate_shift) == [6, 4, 0]
        L.beamformer_hip_set_das_path(0x200)
        path, _, _, reasons, d = lib.describe_das(full.bp, full.filters)
        assert path == 3 and "0x200" in reasons[5] and d.tile_window_samples == 0
        # frames under the channel-split size: from 192 blocks the block-staged kernel runs instead of the split one (480^2: 8 x 30)
        L.beamformer_hip_set_das_path(0)
        for points, want in ((480, 5), (384, 3)):
            full.bp.output_points[0] = full.bp.output_points[1] = points
            path, _, _, _, d = lib.describe_das(full.bp, full.filters)
            assert path == want and (d.split_shift == 0) == (want == 5), (points, path, d.split_shift)
        L.beamformer_hip_set_das_path(0)
        small = cases.make("config2_small")
        path, _, _, reasons, _ = lib.describe_das(small.bp, small.filters)
        assert path == 3 and "channel split" in reasons[5]
        real = cases.make("forces")
        assert "cubic interpolation of IQ samples only" in lib.describe_das(real.bp, real.filters)[3][5]
        coarse = cfg.harness("tpw")
        L.beamformer_hip_set_das_path(0x10)
        path, _, _, reasons, _ = lib.describe_das(coarse.bp, coarse.filters)
        assert path == 3 and "coarse grid" in reasons[5]
        # asked for on the harness plane, the block-staged kernel is taken away again by the ROW-END rule: at F# 0.5 the outermost channels of
        # the deepest pixels echo from beyond the 2048 samples a row holds, and das_tile.hip carries no exact evaluation of such terms
        # (csrc/das_exact.h; the factored kernel behind it does)
        L.beamformer_hip_set_das_path(0x14 | 0x100)
        path, _, _, _, d = lib.describe_das(coarse.bp, coarse.filters)
        assert path == 3 and d.row_end_planes == 1
        fine = cases.make("tile_w32")
        path, _, _, _, d = lib.describe_das(fine.bp, fine.filters)
        assert path == 5 and d.tile_window_samples == 32
    finally:
        L.beamformer_hip_set_das_path(0)


def _matrix(m16):
    return np.array(m16[:], np.float64).reshape(4, 4).T          # column major


def _brute_force_spread(bp, d, das_fs):
    """max over tiles, channels, transmits and planes of (max R - floor(min R)) + (max T - floor(min T)): what the staged window must
    hold (das_staged.hip), from the geometry alone"""
    X, Y, Z = (max(1, v) for v in bp.output_points[:3])
    size = (X, Y, Z)
    vox, xdc = _matrix(bp.das_voxel_transform), _matrix(bp.xdc_transform)
    C, A = bp.channel_count, bp.acquisition_count
    c, fs = bp.speed_of_sound, das_fs
    orient = [bp.transmit_receive_orientation if bp.single_orientation else bp.transmit_receive_orientations[a] for a in range(A)]
    rx_rows = (orient[0] & 0xF) == 1
    u_axis, v_axis = int(d.u_axis), 1 - int(d.u_axis)
    U, V = 1 << d.u_shift, 1 << d.v_shift

    def world(ix, iy, iz):
        p = np.array([ix / max(1, X - 1), iy / max(1, Y - 1), iz / max(1, Z - 1), 1.0])
        return (vox @ p)[:3]

    worst = 0.0
    zs = sorted({0, Z // 2, Z - 1})
    for iz in zs:
        # receive index per (channel, voxel along u) and transmit index per (transmit, voxel along v)
        nu, nv = size[u_axis], size[v_axis]
        R = np.zeros((C, nu))
        for iu in range(nu):
            idx = [0, 0, iz]
            idx[u_axis] = iu
            w = world(*idx)
            t = (xdc @ np.append(w, 1.0))[:3]
            lateral = t[1] if rx_rows else t[0]
            pitch = bp.xdc_element_pitch[1] if rx_rows else bp.xdc_element_pitch[0]
            R[:, iu] = np.hypot(lateral - np.arange(C) * pitch, t[2]) / c * fs
        T = np.zeros((A, nv))
        for iv in range(nv):
            idx = [0, 0, iz]
            idx[v_axis] = iv
            w = world(*idx)
            for a in range(A):
                tx = (orient[a] >> 4) & 0xF
                ang = bp.focal_vector[0] if bp.single_focus else bp.steering_angles[a]
                dep = bp.focal_vector[1] if bp.single_focus else bp.focal_depths[a]
                rad = np.radians(np.float32(ang))
                if tx == 0:
                    dist = 0.0
                else:
                    px = w[1] if tx == 1 else w[0]
                    if np.isinf(dep):
                        dist = px * np.sin(rad) + w[2] * np.cos(rad)
                    else:
                        dist = np.hypot(px - dep * np.sin(rad), w[2] - dep * np.cos(rad))
                T[a, iv] = (dist / c + bp.time_offset) * fs
        for u0 in range(0, nu, U):
            r = R[:, u0:u0 + U]
            rs = r.max(axis=1) - np.floor(r.min(axis=1))
            for v0 in range(0, nv, V):
                t = T[:, v0:v0 + V]
                ts = t.max(axis=1) - np.floor(t.min(axis=1))
                worst = max(worst, rs.max() + ts.max())
    return worst


@pytest.mark.parametrize("seed", range(40))
def test_staged_window_holds_every_tile_brute_force(seed):
    """random separable row-column acquisitions: the window plan_staged chose holds (max R - floor min R) + (max T - floor min T)
    of every tile, with the taps and the rounding of the position (linear: round(p) <= W - 2, p = R' + T'' and T'' carries - 1/2;
    cubic: segment n <= W - 3)"""
    rng = np.random.default_rng(1000 + seed)
    C = int(rng.choice([16, 32, 48]))
    A = int(rng.integers(6, 20))
    focused = bool(rng.integers(0, 2))
    cubic = bool(rng.integers(0, 3) == 0)
    pitch = float(rng.choice([0.15e-3, 0.2e-3, 0.3e-3]))
    half = (C - 1) / 2 * pitch * float(rng.uniform(0.6, 2.5))
    z0 = float(rng.uniform(3e-3, 12e-3))
    z1 = z0 + float(rng.uniform(2e-3, 10e-3))
    points = (int(rng.integers(20, 140)), int(rng.integers(20, 90)), int(rng.integers(1, 4)) + 1)
    depths = None
    if focused:
        depths = rng.choice([-30e-3, -12e-3, 25e-3, 60e-3, np.inf], A)
    acq = cfg.rca(f"prop{seed}", C, A, 2048, points, (-half, -half * float(rng.uniform(0.5, 1.2)), z0), (half, half, z1), seed=seed,
                  orientation=int(rng.choice([0x12, 0x21])), cw=bool(rng.integers(0, 2)), f_number=float(rng.uniform(0.3, 1.5)), pitch=pitch,
                  angles=np.linspace(-float(rng.uniform(2, 20)), float(rng.uniform(2, 20)), A), depths=depths,
                  kind=P.AcquisitionKind.RCA_VLS if focused else P.AcquisitionKind.RCA_TPW,
                  interp=P.InterpolationMode.Cubic if cubic else P.InterpolationMode.Linear, noise=False)
    L = lib.library()
    L.beamformer_hip_set_das_path(3)
    try:
        path, kernel, _, reasons, d = lib.describe_das(acq.bp, acq.filters)
    finally:
        L.beamformer_hip_set_das_path(0)
    assert path in (1, 2), (kernel, reasons)
    if path != 2:
        pytest.skip("declined by plan_staged: " + reasons[2])
    plan = P.HipPlan()
    assert L.beamformer_hip_describe_plan(0, plan)
    worst = _brute_force_spread(acq.bp, d, plan.das_sampling_frequency)
    W = int(d.window_samples)
    # linear: p = R' + T'' < W - 1.5 with T'' = T - floor(min T) - 1/2  <=>  spread < W - 1;  cubic (window one sample early): spread < W - 3
    budget = W - 3 if cubic else W - 1
    assert worst < budget - 0.01, f"tile spread {worst:.2f} samples does not fit the {W}-sample window (U = {1 << d.u_shift}, V = {1 << d.v_shift})"


def test_full_size_configurations_stay_clear_of_row_ends_and_the_harness_planes_do_not():
    """the host bound per plane (das_select.cpp plane_index_bounds): BASELINE configs 2-5 at full size have no in-aperture term within
    reach of an end of its RF row -- their kernels run the instantiation without any row-end code, as in round 3 --, while on the
    reference harness's own F# 0.5 view plane the outermost channels of the deepest pixels echo from beyond the row's 2048 samples"""
    lib.library().beamformer_hip_set_das_path(0)
    for n in (2, 3, 4, 5):
        acq = cfg.config(n)
        d = lib.describe_das(acq.bp, acq.filters)[4]
        assert int(d.row_ends) == 0 and int(d.row_end_planes) == 0, n
    for kind in ("tpw", "forces", "hercules"):
        acq = cfg.harness(kind)
        d = lib.describe_das(acq.bp, acq.filters)[4]
        assert int(d.row_ends) == 1 and int(d.row_end_planes) == 0, kind


def test_random_view_planes_select_the_plane_kernels():
    """the view-plane generator of tests/test_gpu_random.py (the reference harness's shape at test size) is aimed at the HERCULES aligned-grid
    kernel and at the factored kernel: the selection rules say so without a device (small frames: with the channel split switched off,
    flag 0x10, as the GPU test and the fuzz ask for them)"""
    from tests import test_gpu_random as R
    L = lib.library()
    taken = {}
    L.beamformer_hip_set_das_path(0x10)
    try:
        for seed in range(24):
            acq = R.draw_plane(seed)
            path, kernel, _, _, d = lib.describe_das(acq.bp, acq.filters)
            hercules = int(acq.bp.acquisition_kind) in (int(P.AcquisitionKind.HERCULES), int(P.AcquisitionKind.UHERCULES))
            taken.setdefault(hercules, set()).add(path)
    finally:
        L.beamformer_hip_set_das_path(0)
    assert taken[True] == {int(P.DasPath.Hercules)}, taken
    assert int(P.DasPath.Factored) in taken[False] and int(P.DasPath.Hercules) not in taken[False], taken
