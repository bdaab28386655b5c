"""Synthetic acquisitions: parameter blocks + seeded RF for the five BASELINE.json configs
 and scaled-down variants of the same geometries for parity tests.

Host-side plumbing only (numpy): builds a BeamformerSimpleParameters exactly as a client
of the reference would (cf. tests/throughput.c:150-374, :432-491) and synthesises raw RF:
seeded Gaussian noise plus the echo of point scatterers, so that an image has a known
peak.  Nothing here computes a beamformed value.
"""
import math
from dataclasses import dataclass, field

import numpy as np

from . import params as P

SPEED_OF_SOUND = 1540.0


def das_transform_3d(lo, hi):
    """math.c:894-904 (column major 4x4 as 16 floats)."""
    m = np.zeros(16, np.float32)
    m[0], m[5], m[10] = hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]
    m[12], m[13], m[14], m[15] = lo[0], lo[1], lo[2], 1.0
    return m


def das_transform_2d_xz(lo, hi, y_off=0.0):
    """math.c:844-877: image x -> world x, image y -> world z."""
    m = np.zeros(16, np.float32)
    m[0] = hi[0] - lo[0]          # column 0: U * extent
    m[6] = hi[1] - lo[1]          # column 1: V * extent, V = (0,0,1)
    m[9] = 1.0                    # column 2: normal (0,1,0)
    m[12], m[13], m[14], m[15] = lo[0], y_off, lo[1], 1.0
    return m


def das_transform_2d_yz(lo, hi, x_off=0.0):
    """math.c:844-870, :879-885 (normal (-1, 0, 0)): image x -> world y, image y -> world z."""
    m = np.zeros(16, np.float32)
    m[1] = hi[0] - lo[0]          # column 0: U = (0,1,0) * extent
    m[6] = hi[1] - lo[1]          # column 1: V = cross(U, N) = (0,0,1) * extent
    m[8] = -1.0                   # column 2: the normal
    m[12], m[13], m[14], m[15] = -x_off, lo[0], lo[1], 1.0     # t = N * offset + min
    return m


def _voxel_transform(points, lo, hi, plane=None, plane_offset=0.0):
    """(X, Y, 1) images are view planes with depth on image y (das_transform of math.c:906-922 gives the XZ plane;
    plane="yz" asks for math.c:879-885's); anything with z planes is a volume."""
    dims = sum(1 for p in points if p > 1)
    if dims == 3 or points[2] > 1:
        return das_transform_3d(lo, hi)
    if plane == "yz":
        return das_transform_2d_yz((lo[1], lo[2]), (hi[1], hi[2]), plane_offset)
    return das_transform_2d_xz((lo[0], lo[2]), (hi[0], hi[2]), plane_offset)


def translation(x=0.0, y=0.0, z=0.0):
    m = np.zeros(16, np.float32)
    m[0] = m[5] = m[10] = m[15] = 1.0
    m[12], m[13], m[14] = x, y, z
    return m


@dataclass
class Acquisition:
    name: str
    bp: P.SimpleParameters
    filters: list
    rf: np.ndarray                       # raw data as pushed (raw_data_dimensions[1] rows)
    scatterers: list = field(default_factory=list)
    seed: int = 0
    notes: str = ""

    @property
    def voxels(self):
        p = self.bp.output_points
        return max(1, p[0]) * max(1, p[1]) * max(1, p[2])

    @property
    def pairs_upper_bound(self):
        return self.voxels * self.bp.channel_count * self.bp.acquisition_count


def kaiser_filter(fs, cutoff, length=36, beta=5.65):
    """tests/throughput.c:463-473"""
    fp = P.FilterParameters()
    fp.kind = int(P.FilterKind.Kaiser)
    fp.sampling_frequency = fs
    fp.complex = 0
    fp.kaiser.cutoff_frequency = cutoff
    fp.kaiser.beta = beta
    fp.kaiser.length = length
    return fp


def matched_chirp_filter(fs, duration, fmin, fmax, complex_taps=True):
    """tests/throughput.c:475-485"""
    fp = P.FilterParameters()
    fp.kind = int(P.FilterKind.MatchedChirp)
    fp.sampling_frequency = fs
    fp.complex = 1 if complex_taps else 0
    fp.matched_chirp.duration = duration
    fp.matched_chirp.min_frequency = fmin
    fp.matched_chirp.max_frequency = fmax
    return fp


def _base_parameters(C, A, S, points, kind, data_kind, stages, fs, fd, pitch, interp, f_number, cw,
                     voxel_transform, xdc_transform, time_offset=0.0, decode=0, raw_pad=0, contrast=0):
    bp = P.SimpleParameters()
    bp.das_voxel_transform[:] = [float(v) for v in voxel_transform]
    bp.xdc_transform[:] = [float(v) for v in xdc_transform]
    bp.xdc_element_pitch[:] = [float(pitch[0]), float(pitch[1])]
    samples_per_row = A * S * (3 if contrast else 1) + raw_pad
    bp.raw_data_dimensions[:] = [samples_per_row, C]
    bp.sample_count, bp.channel_count, bp.acquisition_count = S, C, A
    bp.acquisition_kind = int(kind)
    bp.decode_mode = decode
    bp.sampling_mode = 1
    bp.time_offset = time_offset
    bp.output_points[:] = [points[0], points[1], points[2], 1]
    bp.sampling_frequency, bp.demodulation_frequency = fs, fd
    bp.speed_of_sound = SPEED_OF_SOUND
    bp.f_number = f_number
    bp.interpolation_mode = int(interp)
    bp.coherency_weighting = 1 if cw else 0
    bp.decimation_rate = 1
    bp.contrast_mode = contrast
    bp.emission_parameters.kind = 0
    bp.emission_parameters.sine.cycles = 2
    bp.emission_parameters.sine.frequency = fd
    for i in range(C):
        bp.channel_mapping[i] = i
    for i, s in enumerate(stages):
        bp.compute_stages[i] = int(s)
        bp.compute_stage_parameters[i] = 0
    bp.compute_stages_count = len(stages)
    bp.data_kind = int(data_kind)
    return bp


def _noise(rng, shape, data_kind, sigma=None):
    base = P.DATA_KIND_NUMPY[int(data_kind)]
    if base == "int16":
        return np.clip(np.rint(rng.normal(0, 1000.0 if sigma is None else sigma, shape)), -32000, 32000).astype(np.int16)
    return rng.normal(0, 1.0 if sigma is None else sigma, shape).astype(base)


def _add_echo(rf, rows, delays_s, fs, fd, amplitude, cycles=2.5):
    """rf[rows] += windowed tone bursts arriving at delays_s (vectorised over rows).
    rf: float32 work array [n_rows, S] of real samples."""
    S = rf.shape[1]
    width = cycles * fs / fd                      # samples
    n0 = delays_s * fs
    half = int(math.ceil(3 * width))
    offsets = np.arange(-half, half + 1, dtype=np.float32)
    centre = np.floor(n0).astype(np.int64)
    idx = centre[:, None] + offsets[None, :].astype(np.int64)
    t = (idx - n0[:, None]) / fs
    pulse = amplitude * np.exp(-((idx - n0[:, None]) / width) ** 2) * np.cos(2 * np.pi * fd * t)
    ok = (idx >= 0) & (idx < S)
    r = np.broadcast_to(rows[:, None], idx.shape)
    np.add.at(rf, (r[ok], idx[ok]), pulse[ok].astype(np.float32))


def _finish_rf(work, data_kind):
    base = P.DATA_KIND_NUMPY[int(data_kind)]
    if base == "int16":
        return np.clip(np.rint(work), -32767, 32767).astype(np.int16)
    return work.astype(base)


def _rca_delays(bp, point, angles_deg, depths, orient):
    """Arrival time of a scatterer's echo for every (channel, transmit), following the
    geometry the DAS stage assumes (das.glsl:154-231)."""
    C, A = bp.channel_count, bp.acquisition_count
    world = np.array(point, np.float64)
    m = np.array(bp.xdc_transform[:], np.float64).reshape(4, 4).T     # column major -> matrix
    xdc = m[:3, :3] @ world + m[:3, 3]
    delays = np.zeros((C, A))
    for a in range(A):
        txrx = orient[a]
        tx, rx = (txrx >> 4) & 0xF, txrx & 0xF
        ang = math.radians(float(angles_deg[a]))
        if tx == 0:
            txd = 0.0
        else:
            px = world[1] if tx == 1 else world[0]
            if math.isinf(depths[a]):
                txd = px * math.sin(ang) + world[2] * math.cos(ang)
            else:
                fx, fz = depths[a] * math.sin(ang), depths[a] * math.cos(ang)
                txd = math.hypot(px - fx, world[2] - fz)
        lateral = xdc[1] if rx == 1 else xdc[0]
        pitch = bp.xdc_element_pitch[1] if rx == 1 else bp.xdc_element_pitch[0]
        ch = np.arange(C)
        delays[:, a] = (txd + np.hypot(lateral - ch * pitch, xdc[2])) / SPEED_OF_SOUND
    return delays


def rca(name, C, A, S, points, lo, hi, *, seed, data_kind=P.DataKind.Int16, interp=P.InterpolationMode.Linear,
        cw=False, f_number=1.0, pitch=0.3e-3, fs=25e6, fd=6.25e6, orientation=0x22, angles=None,
        depths=None, single=False, demodulate=True, kind=P.AcquisitionKind.RCA_TPW, scatterers=None,
        noise=True, channel_shuffle=False, raw_pad=0, contrast=False, stages=None, plane=None, plane_offset=0.0):
    """Row-column / linear array, plane or diverging waves (configs 1, 2, 4)."""
    rng = np.random.default_rng(seed)
    vt = _voxel_transform(points, lo, hi, plane, plane_offset)
    half = (C - 1) / 2 * pitch
    # world -> transducer: element `ch` sits at transducer x (and y) = ch * pitch
    xt = translation(half, half if (orientation & 0xF) == 1 or ((orientation >> 4) & 0xF) == 1 else 0.0, 0.0)
    if stages is None:
        stages = ([P.ShaderKind.Demodulate] if demodulate else [P.ShaderKind.Decode]) + [P.ShaderKind.DAS]
    demodulate = P.ShaderKind.Demodulate in stages
    bp = _base_parameters(C, A, S, points, kind, data_kind, list(stages), fs, fd, (pitch, pitch), interp, f_number, cw,
                          vt, xt, raw_pad=raw_pad, contrast=1 if contrast else 0)
    if angles is None:
        angles = np.zeros(A) if A == 1 else np.linspace(-15.0, 15.0, A)
    if depths is None:
        depths = np.full(A, np.inf)
    bp.single_focus = 1 if single else 0
    bp.single_orientation = 1 if single else 0
    bp.transmit_receive_orientation = orientation
    bp.focal_vector[:] = [float(angles[0]), float(depths[0])]
    for a in range(A):
        bp.steering_angles[a] = float(angles[a])
        bp.focal_depths[a] = float(depths[a])
        bp.transmit_receive_orientations[a] = orientation
    if channel_shuffle:
        perm = rng.permutation(C)
        for i in range(C):
            bp.channel_mapping[i] = int(perm[i])

    row_samples = bp.raw_data_dimensions[0]
    n_scalar = 2 if P.DATA_KIND_COMPLEX[int(data_kind)] else 1
    work = (rng.normal(0, 1000.0 if "int" in P.DATA_KIND_NUMPY[int(data_kind)] else 1.0, (C, row_samples * n_scalar))
            .astype(np.float32) if noise else np.zeros((C, row_samples * n_scalar), np.float32))
    scatterers = scatterers or []
    if scatterers and n_scalar == 1 and not contrast:
        amp = 8000.0 if "int" in P.DATA_KIND_NUMPY[int(data_kind)] else 8.0
        view = work.reshape(C, -1)
        for pt in scatterers:
            d = _rca_delays(bp, pt, angles, depths, [orientation] * A)
            rows_rf = np.zeros((C * A, S), np.float32)
            _add_echo(rows_rf, np.arange(C * A), d.reshape(-1), fs, fd, amp)
            view[:, : A * S] += rows_rf.reshape(C, A * S)
    rf_mapped = _finish_rf(work, data_kind)
    # raw row r holds the data of the channel whose mapping names it
    raw = np.empty_like(rf_mapped)
    for ch in range(C):
        raw[bp.channel_mapping[ch]] = rf_mapped[ch]
    filt = [kaiser_filter(fs / 2, fd / 2)] if demodulate else []
    return Acquisition(name, bp, filt, raw, scatterers, seed)


def hercules(name, C, A, S, points, lo, hi, *, seed, data_kind=P.DataKind.Int16, interp=P.InterpolationMode.Linear,
             cw=False, f_number=1.0, pitch=0.3e-3, fs=25e6, fd=6.25e6, orientation=0x12, focal=(0.0, np.inf),
             stages=(P.ShaderKind.Decode, P.ShaderKind.DAS), decode=1, kind=P.AcquisitionKind.HERCULES,
             sparse=None, filters=None, noise_sigma=None, plane=None, plane_offset=0.0):
    """2-D aperture: receive channel along one axis x decoded transmit element along the
    other (configs 3, 5; das.glsl:233-286)."""
    rng = np.random.default_rng(seed)
    vt = _voxel_transform(points, lo, hi, plane, plane_offset)
    xt = translation((C - 1) / 2 * pitch, (A - 1) / 2 * pitch, 0.0)
    bp = _base_parameters(C, A, S, points, kind, data_kind, list(stages), fs, fd, (pitch, pitch), interp, f_number, cw,
                          vt, xt, decode=decode)
    bp.single_focus = 1
    bp.single_orientation = 1
    bp.transmit_receive_orientation = orientation
    bp.focal_vector[:] = [float(focal[0]), float(focal[1])]
    for a in range(A):
        bp.steering_angles[a] = float(focal[0])
        bp.focal_depths[a] = float(focal[1])
        bp.transmit_receive_orientations[a] = orientation
    if sparse is not None:
        for i, e in enumerate(sparse):
            bp.sparse_elements[i] = int(e)
    n_scalar = 2 if P.DATA_KIND_COMPLEX[int(data_kind)] else 1
    raw = _noise(rng, (C, A * S * n_scalar), data_kind, noise_sigma)
    if filters is None:
        filters = [kaiser_filter(fs / 2, fd / 2)] if P.ShaderKind.Demodulate in stages else []
    return Acquisition(name, bp, filters, raw, [], seed)


def forces(name, C, A, S, points, lo, hi, *, seed, data_kind=P.DataKind.Int16, interp=P.InterpolationMode.Linear,
           cw=False, f_number=1.0, pitch=0.3e-3, fs=25e6, fd=6.25e6, kind=P.AcquisitionKind.FORCES,
           stages=(P.ShaderKind.Decode, P.ShaderKind.DAS), decode=1, sparse=None, readi_groups=0, readi_group=0):
    """FORCES / UFORCES / READI (das.glsl:288-366): imaging plane x-z."""
    rng = np.random.default_rng(seed)
    vt = _voxel_transform(points, lo, hi)
    xt = translation((C - 1) / 2 * pitch, (C - 1) / 2 * pitch, 0.0)
    bp = _base_parameters(C, A, S, points, kind, data_kind, list(stages), fs, fd, (pitch, pitch), interp, f_number, cw,
                          vt, xt, decode=decode)
    bp.single_focus = 1
    bp.single_orientation = 1
    bp.transmit_receive_orientation = 0x22
    bp.focal_vector[:] = [0.0, float("inf")]
    bp.readi_group_count = readi_groups
    bp.readi_group = readi_group
    if sparse is not None:
        for i, e in enumerate(sparse):
            bp.sparse_elements[i] = int(e)
    n_scalar = 2 if P.DATA_KIND_COMPLEX[int(data_kind)] else 1
    raw = _noise(rng, (C, A * S * n_scalar), data_kind)
    filters = [kaiser_filter(fs / 2, fd / 2)] if P.ShaderKind.Demodulate in stages else []
    return Acquisition(name, bp, filters, raw, [], seed)


# ----------------------------------------------------------------------------- BASELINE configs

def config(n, scale=1.0):
    """BASELINE.json configs[n-1].  scale < 1 shrinks channel,
    transmit, sample and voxel counts (same geometry family) for parity tests."""
    def s(v, lo=1, mult=1):
        r = max(lo, int(round(v * scale)))
        return max(mult, r // mult * mult)

    def depth_range(samples, fs=25e6, fraction=0.40):
        """Axial extent [z_lo, z_hi] whose two-way travel fits the recorded samples."""
        path = samples / fs * SPEED_OF_SOUND
        return 0.12 * path, fraction * path

    if n == 1:   # 2-D, 64-ch linear array, 1 plane wave -> 256 x 256
        C, S = s(64, 8), s(2048, 256, 128)
        z0, z1 = depth_range(S)
        return rca("config1", C, 1, S, (s(256, 16), s(256, 16), 1), (-9.6e-3, 0, z0), (9.6e-3, 0, z1),
                   seed=1, single=True, orientation=0x22, interp=P.InterpolationMode.Linear, f_number=1.0,
                   scatterers=[(1.5e-3, 0.0, z0 + 0.45 * (z1 - z0))])
    if n == 2:   # 2-D, 128-ch, 31 compounded plane waves, fp16 RF, cubic -> 1024 x 1024
        C, A, S = s(128, 8), s(31, 3), s(4096, 256, 128)
        z0, z1 = depth_range(S)
        return rca("config2", C, A, S, (s(1024, 16), s(1024, 16), 1), (-12.8e-3, 0, z0), (12.8e-3, 0, z1),
                   seed=2, data_kind=P.DataKind.Float16, interp=P.InterpolationMode.Cubic, pitch=0.2e-3,
                   orientation=0x22, f_number=1.0, scatterers=[(-2.0e-3, 0.0, z0 + 0.4 * (z1 - z0))])
    if n == 3:   # 3-D, 32 x 32 aperture restated as HERCULES 32 rx x 32 decoded transmits -> 256^3
        C, A, S = s(32, 8, 4), s(32, 8, 4), s(2048, 512, 128)
        A = 1 << int(math.log2(A))
        focus = -20e-3                                  # diverging wave: virtual source behind the array
        z0, z1 = depth_range(S, fraction=0.30)
        acq = hercules("config3", C, A, S, (s(256, 8), s(256, 8), s(256, 8)), (-4.8e-3, -4.8e-3, z0),
                       (4.8e-3, 4.8e-3, z1), seed=3, focal=(0.0, focus), f_number=1.0)
        acq.bp.time_offset = focus / SPEED_OF_SOUND     # the wave leaves the array at t = 0
        acq.notes = ("BASELINE's 32 x 32 matrix probe (1024 channels) exceeds the API's 256-channel limit; restated as a "
                     "HERCULES aperture of 32 receive channels x 32 decoded transmit elements = 1024 element pairs, one "
                     "diverging wave (virtual source 20 mm behind the array) -- SURVEY 8d config 3")
        return acq
    if n == 4:   # 3-D RCA, 256 ch x 75 plane waves, coherency weighting -> 512^3 (the headline metric)
        C, A, S = s(256, 16, 16), s(75, 3), s(4096, 512, 128)
        z0, z1 = depth_range(S, fraction=0.30)
        half = (C - 1) / 2 * 0.15e-3
        return rca("config4", C, A, S, (s(512, 8), s(512, 8), s(512, 8)), (-half, -half, z0), (half, half, z1),
                   seed=4, cw=True, pitch=0.15e-3, orientation=0x12, f_number=0.5,
                   angles=np.linspace(-18.5, 18.5, A) if A > 1 else np.zeros(1),
                   scatterers=[(0.2 * half, -0.3 * half, z0 + 0.5 * (z1 - z0))])
    if n == 5:   # full pipeline, fp16 Hadamard-encoded RF, 256 ch x 128 tx -> 512^3
        C, A, S = s(256, 16, 16), s(128, 8), s(2048, 512, 128)
        A = 1 << int(math.log2(A))
        z0, z1 = depth_range(S, fraction=0.35)
        acq = hercules("config5", C, A, S, (s(512, 8), s(512, 8), s(512, 8)),
                       (-(C - 1) / 2 * 0.15e-3, -(A - 1) / 2 * 0.15e-3, z0),
                       ((C - 1) / 2 * 0.15e-3, (A - 1) / 2 * 0.15e-3, z1), seed=5,
                       data_kind=P.DataKind.Float16, pitch=0.15e-3, cw=True, f_number=0.5, interp=P.InterpolationMode.Linear,
                       stages=(P.ShaderKind.Demodulate, P.ShaderKind.Decode, P.ShaderKind.DAS))
        acq.notes = ("the reference's canonical stage order {Demodulate, Decode, DAS} (tests/throughput.c:455-461) in place of "
                     "BASELINE's {Decode, Filter, DAS}: coherency weighting is implicit after DAS, min/max is the library's "
                     "beamformer_hip_frame_min_max -- SURVEY 8d config 5")
        return acq
    raise ValueError(n)


# ----------------------------------------------------------------------------- the reference harness's frame

HARNESS_KINDS = ("tpw", "tpw_swapped", "vls", "hercules", "forces")


def harness(kind, scale=1.0, plane="xz"):
    """The frame tests/throughput.c beamforms out of every dataset (`execute_study`, :443-491; globals :20-23): a
    512 x 1024 view plane (lateral -60..60 mm, axial 10..165 mm, `das_transform` of the harness's (512, 1, 1024) points
    = the XZ plane, math.c:906-922), cubic interpolation, F# 0.5, decimation 1, {Demodulate, Decode, DAS} with the
    Kaiser low-pass of :463-473 -- here on synthetic Int16 RF of the size of the lab's row-column datasets: 256
    channels x 128 transmits x 4096 samples.  20 MHz sampling, 5 MHz centre (4x sampling), 0.25 mm pitch: a
    64 mm aperture whose recorded range (158 mm at normal incidence) covers the harness's axial extent but for the
    deepest oblique paths.  `kind`: tpw (plane waves steered along y, received along x), tpw_swapped (steered along x,
    received along y), vls (diverging waves), hercules (Hadamard-encoded, 2-D aperture), forces.  plane="yz": the
    other view plane the reference offers (math.c:879-885).  scale < 1 shrinks every count for parity tests."""
    def s(v, lo=1, mult=1):
        r = max(lo, int(round(v * scale)))
        return max(mult, r // mult * mult)

    C, A, S = s(256, 16, 16), s(128, 8), s(4096, 512, 128)
    A = 1 << int(math.log2(A))
    points = (s(512, 16), s(1024, 16), 1)
    lo, hi = (-60e-3, -60e-3, 10e-3), (60e-3, 60e-3, 165e-3)
    fs, fd, pitch = 20e6, 5e6, 0.25e-3
    if scale < 1.0:
        # the same rays through a shorter record: shrink the geometry with the sample count
        k = S / 4096.0
        lo, hi = tuple(v * k for v in lo), tuple(v * k for v in hi)
        pitch *= max(k, 64.0 / C * k)
    canonical = (P.ShaderKind.Demodulate, P.ShaderKind.Decode, P.ShaderKind.DAS)
    name = f"harness_{kind}" + ("" if plane == "xz" else f"_{plane}")
    if kind in ("tpw", "tpw_swapped", "vls"):
        orientation = 0x21 if kind == "tpw_swapped" else 0x12
        angles = np.linspace(-18.0, 18.0, A)
        depths = np.full(A, -40e-3 * (S / 4096.0)) if kind == "vls" else None
        acq = rca(name, C, A, S, points, lo, hi, seed=71, interp=P.InterpolationMode.Cubic, f_number=0.5, pitch=pitch, fs=fs, fd=fd,
                  orientation=orientation, angles=angles, depths=depths, stages=canonical, plane=plane,
                  kind=P.AcquisitionKind.RCA_VLS if kind == "vls" else P.AcquisitionKind.RCA_TPW,
                  scatterers=[(0.1 * hi[0], 0.0, lo[2] + 0.45 * (hi[2] - lo[2]))] if plane == "xz" else
                             [(0.0, 0.1 * hi[1], lo[2] + 0.45 * (hi[2] - lo[2]))])
    elif kind == "hercules":
        acq = hercules(name, C, A, S, points, lo, hi, seed=72, interp=P.InterpolationMode.Cubic, f_number=0.5, pitch=pitch, fs=fs, fd=fd,
                       stages=canonical, plane=plane)
    elif kind == "forces":
        if plane != "xz":
            raise ValueError("FORCES images the x-z plane")
        acq = forces(name, C, A, S, points, lo, hi, seed=73, interp=P.InterpolationMode.Cubic, f_number=0.5, pitch=pitch, fs=fs, fd=fd,
                     stages=canonical)
    else:
        raise ValueError(kind)
    acq.notes = ("tests/throughput.c:443-491 on synthetic RF: the harness's 512 x 1024 view plane, cubic, F# 0.5, "
                 "{Demodulate, Decode, DAS}; Decode is dropped by the planner when the acquisition is not encoded (beamformer_core.c:627-629)")
    return acq


def by_name(spec, scale=1.0):
    """`4` / `"4"`: BASELINE configs (1-based); `"harness:<kind>[:yz]"`: the reference harness's frame."""
    if isinstance(spec, str) and spec.startswith("harness:"):
        parts = spec.split(":")
        return harness(parts[1], scale, parts[2] if len(parts) > 2 else "xz")
    return config(int(spec), scale)
