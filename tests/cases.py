"""Small acquisitions shared by the CPU and GPU parity tests: every DAS geometry family,
interpolation mode, data kind and pre-DAS stage combination the reference can run, at sizes
the CPU oracle finishes in well under a second each."""
import numpy as np

from ogl_beamforming_amd import configs as cfg
from ogl_beamforming_amd import params as P

I = P.InterpolationMode
K = P.AcquisitionKind
D = P.DataKind
S = P.ShaderKind

LO3, HI3 = (-3e-3, -3e-3, 6e-3), (3e-3, 3e-3, 18e-3)


def _cases():
    c = {}
    # the five BASELINE.json configs, shrunk
    c["config1_small"] = lambda: cfg.config(1, 0.25)
    c["config2_small"] = lambda: cfg.config(2, 0.0625)
    c["config3_small"] = lambda: cfg.config(3, 0.125)
    c["config4_small"] = lambda: cfg.config(4, 0.0625)
    c["config5_small"] = lambda: cfg.config(5, 0.0625)
    # RCA variants
    c["rca_nearest_real"] = lambda: cfg.rca("rca_nearest_real", 16, 3, 512, (24, 1, 40), LO3, HI3, seed=11,
                                            interp=I.Nearest, demodulate=False, data_kind=D.Int16,
                                            single=False, orientation=0x22)
    c["rca_cubic_real"] = lambda: cfg.rca("rca_cubic_real", 16, 2, 512, (20, 20, 1), LO3, HI3, seed=12,
                                          interp=I.Cubic, demodulate=False, data_kind=D.Int16)
    c["rca_vls_cw"] = lambda: cfg.rca("rca_vls_cw", 32, 4, 512, (12, 10, 14), LO3, HI3, seed=13, cw=True,
                                      kind=K.RCA_VLS, orientation=0x21, depths=np.array([-10e-3, 25e-3, -15e-3, 40e-3]),
                                      angles=np.array([-5.0, 0.0, 5.0, 10.0]), f_number=0.7)
    c["rca_sep_ragged_cubic"] = lambda: cfg.rca("rca_sep_ragged_cubic", 32, 5, 512, (21, 37, 5), LO3, HI3, seed=41,
                                                interp=I.Cubic, orientation=0x12, cw=True, f_number=1.5,
                                                angles=np.linspace(-8, 8, 5))
    c["rca_sep_real_nearest"] = lambda: cfg.rca("rca_sep_real_nearest", 16, 3, 512, (19, 18, 3), LO3, HI3, seed=42,
                                                interp=I.Nearest, orientation=0x21, demodulate=False,
                                                angles=np.linspace(-5, 5, 3))
    # LDS-staged kernel: a coarse grid whose delay spread needs the 64-sample window, one that
    # fits no window (falls back to the gather kernel), and ragged tile edges
    c["rca_staged_w64"] = lambda: cfg.rca("rca_staged_w64", 32, 4, 1024, (40, 40, 3), (-4e-3, -4e-3, 8e-3), (4e-3, 4e-3, 11e-3),
                                          seed=43, orientation=0x12, cw=True, f_number=0.3, angles=np.linspace(-10, 10, 4))
    c["rca_staged_too_wide"] = lambda: cfg.rca("rca_staged_too_wide", 16, 3, 1024, (24, 24, 2), (-14e-3, -14e-3, 6e-3),
                                               (14e-3, 14e-3, 9e-3), seed=44, orientation=0x12, f_number=0.2,
                                               angles=np.linspace(-10, 10, 3))
    c["rca_staged_ragged"] = lambda: cfg.rca("rca_staged_ragged", 48, 7, 512, (45, 70, 2), LO3, HI3, seed=45,
                                             orientation=0x21, cw=True, f_number=0.6, angles=np.linspace(-12, 12, 7))
    # 13 transmits: at or above executor.cpp's kStagedMinTransmits the staged kernel is the automatic choice (and 13 is not
    # a multiple of 4: the zero-padded transmit batch)
    c["rca_staged_auto"] = lambda: cfg.rca("rca_staged_auto", 32, 13, 512, (40, 36, 3), LO3, HI3, seed=46, orientation=0x12,
                                           cw=True, f_number=0.6, angles=np.linspace(-12, 12, 13))
    # focused / diverging transmits (distances instead of plane-wave projections in the transmit table) on the staged kernel
    c["rca_vls_staged"] = lambda: cfg.rca("rca_vls_staged", 32, 8, 2048, (40, 36, 3), LO3, HI3, seed=47, orientation=0x12, cw=True,
                                          kind=K.RCA_VLS, f_number=0.6, angles=np.linspace(-8, 8, 8),
                                          depths=np.array([-12e-3, 30e-3, -20e-3, 45e-3, 25e-3, -15e-3, 60e-3, -30e-3]))
    # the same focused transmits over RF rows too short for most of them: terms that fall off the end of a row in nearly every
    # wave (the unchecked / checked loop decision of the staged kernel; 256 samples after demodulation)
    c["rca_vls_staged_short_rows"] = lambda: cfg.rca("rca_vls_staged_short_rows", 32, 8, 512, (40, 36, 3), LO3, HI3, seed=48,
                                                     orientation=0x12, cw=True, kind=K.RCA_VLS, f_number=0.6, angles=np.linspace(-8, 8, 8),
                                                     depths=np.array([-12e-3, 30e-3, -20e-3, 45e-3, 25e-3, -15e-3, 60e-3, -30e-3]))
    # real samples (no Demodulate) on the staged kernel's real-data twin, with and without coherency weighting; 9 transmits
    # (padded batch), short rows in the second one (checked loop)
    c["rca_staged_real"] = lambda: cfg.rca("rca_staged_real", 32, 9, 1024, (40, 36, 3), LO3, HI3, seed=49, orientation=0x12, cw=True,
                                           demodulate=False, data_kind=D.Int16, f_number=0.6, angles=np.linspace(-12, 12, 9))
    c["rca_staged_real_short_rows"] = lambda: cfg.rca("rca_staged_real_short_rows", 24, 7, 256, (45, 70, 2), LO3, HI3, seed=50,
                                                      orientation=0x21, cw=False, demodulate=False, data_kind=D.Float32, f_number=0.6,
                                                      angles=np.linspace(-12, 12, 7))
    # cubic interpolation of IQ samples on the staged kernel's cubic twin (9 transmits: padded batch), and with rows short enough
    # that most waves run its checked loop
    c["rca_staged_cubic"] = lambda: cfg.rca("rca_staged_cubic", 32, 9, 1024, (40, 36, 3), LO3, HI3, seed=51, orientation=0x12, cw=True,
                                            interp=I.Cubic, f_number=0.6, angles=np.linspace(-12, 12, 9))
    c["rca_staged_cubic_short_rows"] = lambda: cfg.rca("rca_staged_cubic_short_rows", 24, 7, 384, (45, 70, 2), LO3, HI3, seed=52,
                                                       orientation=0x21, cw=False, interp=I.Cubic, f_number=0.6,
                                                       angles=np.linspace(-12, 12, 7))
    # a lateral grid fine enough for 64 x 16 tiles (64 voxels along the receive axis = x): the staged kernel's wave-uniform
    # transmit tables (global table + scalar loads), with coherency weighting and 13 transmits (padded batch), and with focused
    # transmits over short rows (its checked loop)
    c["rca_staged_fine"] = lambda: cfg.rca("rca_staged_fine", 32, 13, 512, (150, 36, 3), LO3, HI3, seed=53, orientation=0x12,
                                           cw=True, f_number=0.6, angles=np.linspace(-12, 12, 13))
    c["rca_staged_fine_vls_short_rows"] = lambda: cfg.rca("rca_staged_fine_vls_short_rows", 32, 8, 512, (150, 20, 2), LO3, HI3, seed=54,
                                                          orientation=0x12, cw=False, kind=K.RCA_VLS, f_number=0.6, angles=np.linspace(-8, 8, 8),
                                                          depths=np.array([-12e-3, 30e-3, -20e-3, 45e-3, 25e-3, -15e-3, 60e-3, -30e-3]))
    c["rca_f32_complex_in"] = lambda: cfg.rca("rca_f32_complex_in", 16, 2, 256, (16, 16, 1), LO3, HI3, seed=14,
                                              demodulate=False, data_kind=D.Float32Complex, interp=I.Cubic)
    c["rca_i16_complex_in"] = lambda: cfg.rca("rca_i16_complex_in", 16, 2, 256, (16, 16, 1), LO3, HI3, seed=15,
                                              demodulate=False, data_kind=D.Int16Complex, cw=True)
    c["rca_f32_demod"] = lambda: cfg.rca("rca_f32_demod", 16, 2, 512, (16, 1, 24), LO3, HI3, seed=16,
                                         data_kind=D.Float32, interp=I.Nearest)
    c["rca_shuffled_padded"] = lambda: cfg.rca("rca_shuffled_padded", 24, 2, 384, (16, 16, 1), LO3, HI3, seed=17,
                                               channel_shuffle=True, raw_pad=37)
    c["rca_a1s2"] = lambda: cfg.rca("rca_a1s2", 16, 2, 256, (16, 16, 1), LO3, HI3, seed=18, contrast=True)
    c["rca_flash_none_tx"] = lambda: cfg.rca("rca_flash_none_tx", 16, 1, 512, (16, 16, 1), LO3, HI3, seed=19,
                                             kind=K.Flash, orientation=0x02, single=True)
    # HERCULES family
    c["hercules_real"] = lambda: cfg.hercules("hercules_real", 16, 8, 512, (10, 12, 14), LO3, HI3, seed=21,
                                              interp=I.Cubic, f_number=0.8)
    c["uhercules_sparse"] = lambda: cfg.hercules("uhercules_sparse", 16, 8, 512, (10, 10, 10), LO3, HI3, seed=22,
                                                 kind=K.UHERCULES, sparse=[0, 3, 5, 9, 12, 14, 15], decode=0,
                                                 orientation=0x21)
    c["hercules_demod_decode_cw"] = lambda: cfg.hercules(
        "hercules_demod_decode_cw", 16, 16, 512, (10, 10, 12), LO3, HI3, seed=23, cw=True,
        stages=(S.Demodulate, S.Decode, S.DAS), interp=I.Nearest)
    c["hercules_order12"] = lambda: cfg.hercules("hercules_order12", 16, 12, 256, (8, 8, 8), LO3, HI3, seed=24)
    c["hercules_chirp"] = lambda: cfg.hercules(
        "hercules_chirp", 16, 4, 1024, (8, 8, 8), LO3, HI3, seed=25, decode=0, data_kind=D.Float32,
        stages=(S.Demodulate, S.DAS),
        filters=[cfg.matched_chirp_filter(12.5e6, 4e-6, -2e6, 2e6, complex_taps=True)])
    # HERCULES volumes whose x-y extent fills whole 256-voxel tiles (the shape full-size frames run in)
    c["hercules_table_cw"] = lambda: cfg.hercules(
        "hercules_table_cw", 16, 16, 512, (16, 16, 3), LO3, HI3, seed=27, cw=True, f_number=0.7,
        stages=(S.Demodulate, S.Decode, S.DAS))
    c["hercules_table_real_cubic"] = lambda: cfg.hercules("hercules_table_real_cubic", 24, 8, 512, (32, 16, 2), LO3, HI3, seed=28,
                                                          interp=I.Cubic, f_number=1.0, focal=(0.0, -12e-3))
    c["uhercules_table_sparse"] = lambda: cfg.hercules("uhercules_table_sparse", 16, 8, 512, (16, 16, 2), LO3, HI3, seed=29,
                                                       kind=K.UHERCULES, sparse=[0, 3, 5, 9, 12, 14, 15], decode=0,
                                                       orientation=0x21, data_kind=D.Float32Complex, cw=True, f_number=0.5)
    # HERCULES grids wide enough (>= 32 voxels along x, <= 25 % idle lanes) for the aligned-grid kernel
    # (das_hercules.hip) to be picked automatically: both orientations (inner loop = transmits / = channels),
    # ragged x edge, transmit count not a multiple of the batch, f-number culling that is partial per wave.
    # (Frames this small normally get the general kernel's channel split; fewer than 8 channels rules that out,
    # so the first two exercise the automatic selection.)
    c["hercules_wide_cw"] = lambda: cfg.hercules(
        "hercules_wide_cw", 7, 16, 512, (64, 6, 3), LO3, HI3, seed=51, cw=True, f_number=0.7,
        stages=(S.Demodulate, S.Decode, S.DAS))
    c["hercules_wide_real_swapped"] = lambda: cfg.hercules(
        "hercules_wide_real_swapped", 6, 8, 512, (56, 5, 2), (-4e-3, -2e-3, 3e-3), (4e-3, 2e-3, 9e-3), seed=52,
        orientation=0x21, f_number=1.2, focal=(0.0, -12e-3))
    c["uhercules_wide_sparse_cubic"] = lambda: cfg.hercules(
        "uhercules_wide_sparse_cubic", 16, 8, 512, (48, 4, 2), LO3, HI3, seed=53, kind=K.UHERCULES,
        sparse=[0, 3, 5, 9, 12, 14, 15], decode=0, data_kind=D.Float32Complex, cw=True, f_number=0.9, interp=I.Cubic)
    # the reference harness's own setting: the canonical pipeline with cubic interpolation (the aligned-grid kernel then gathers the
    # per-sample segment polynomials its pre-pass wrote)
    c["hercules_wide_cubic_cw"] = lambda: cfg.hercules(
        "hercules_wide_cubic_cw", 7, 16, 512, (64, 6, 3), LO3, HI3, seed=55, cw=True, f_number=0.7, interp=I.Cubic,
        stages=(S.Demodulate, S.Decode, S.DAS))
    c["hercules_wide_nearest"] = lambda: cfg.hercules(
        "hercules_wide_nearest", 8, 7, 512, (40, 3, 3), LO3, HI3, seed=54, decode=0, interp=I.Nearest,
        data_kind=D.Float32Complex, f_number=0.6)
    # BASELINE config 5 in its literal stage order {Decode, Filter, DAS} on fp16 RF: Decode's output stays
    # binary16 (accumulated with per-operation rounding), Filter stages through binary16, real-valued DAS + CW
    c["config5_literal_order"] = lambda: cfg.hercules(
        "config5_literal_order", 16, 16, 512, (10, 10, 12), LO3, HI3, seed=26, cw=True, data_kind=D.Float16,
        stages=(S.Decode, S.Filter, S.DAS), filters=[cfg.kaiser_filter(25e6, 5e6, length=24, beta=5.0)])
    # FORCES family
    c["forces"] = lambda: cfg.forces("forces", 16, 16, 512, (20, 1, 20), LO3, HI3, seed=31)
    # das_tile.hip (block-wide LDS staging of cubic polynomial windows; automatic on fine grids with tx and rx on one axis -- BASELINE
    # config 2's class): a 25 um x 69 um view-plane grid, ragged in both directions (160 = 2.5 tiles of 64, 48 = 3 of 16) ...
    c["tile_tpw"] = lambda: cfg.rca("tile_tpw", 24, 9, 768, (160, 48, 1), (-2.0e-3, 0, 6.0e-3), (2.0e-3, 0, 9.3e-3), seed=71,
                                    interp=I.Cubic, orientation=0x22, f_number=1.0, pitch=0.2e-3, angles=np.linspace(-10, 10, 9))
    # ... the same over twice the depth range: a 64 x 16 tile's delays spread over about 50 samples (64-sample windows, 8 transmits a group;
    # tile_tpw's 25 fit the 32-sample window) ...
    c["tile_tpw_w64"] = lambda: cfg.rca("tile_tpw_w64", 24, 9, 768, (160, 48, 1), (-2.0e-3, 0, 6.0e-3), (2.0e-3, 0, 13.0e-3), seed=79,
                                        interp=I.Cubic, orientation=0x22, f_number=1.0, pitch=0.2e-3, angles=np.linspace(-10, 10, 9))
    # ... coherency weighting over rows that end inside the image (index 396 at the deepest voxels, 384 samples: the range-checked
    # loop, zeros past the end), 7 transmits (a ragged group of 8), 18 channels (a ragged chunk of 4) ...
    c["tile_tpw_cw_short"] = lambda: cfg.rca("tile_tpw_cw_short", 18, 7, 384, (96, 40, 1), (-1.2e-3, 0, 9.5e-3), (1.2e-3, 0, 12.2e-3),
                                             seed=72, interp=I.Cubic, orientation=0x22, cw=True, f_number=0.8, pitch=0.2e-3,
                                             angles=np.linspace(-8, 8, 7))
    # ... a grid fine enough for the 32-sample window (16 transmits per staged group) ...
    c["tile_w32"] = lambda: cfg.rca("tile_w32", 16, 20, 768, (128, 32, 1), (-0.5e-3, 0, 8.0e-3), (0.5e-3, 0, 8.3e-3), seed=77,
                                    interp=I.Cubic, orientation=0x22, f_number=1.2, pitch=0.2e-3, angles=np.linspace(-9, 9, 20))
    # ... a thin volume whose tile takes voxels of two z planes (64 x 8 x 2) ...
    c["tile_thin_volume"] = lambda: cfg.rca("tile_thin_volume", 16, 8, 768, (96, 6, 6), (-1.4e-3, -0.3e-3, 7.0e-3), (1.4e-3, 0.3e-3, 7.5e-3), seed=78,
                                            interp=I.Cubic, orientation=0x22, cw=True, f_number=1.0, pitch=0.2e-3, angles=np.linspace(-8, 8, 8))
    # ... diverging and focused transmits (the square root per transmit), steered along the receive axis ...
    c["tile_vls"] = lambda: cfg.rca("tile_vls", 16, 6, 768, (128, 32, 1), (-1.6e-3, 0, 7.0e-3), (1.6e-3, 0, 9.2e-3), seed=73,
                                    interp=I.Cubic, orientation=0x22, kind=K.RCA_VLS, f_number=0.9, pitch=0.2e-3,
                                    angles=np.linspace(-6, 6, 6), depths=np.array([-12e-3, 30e-3, -20e-3, 45e-3, 25e-3, -15e-3]))
    # ... the near field under a wide aperture: the receive spread of most chunks does not fit a window there and the block runs its
    # gather loop for them (both kinds of chunk in one frame) ...
    c["tile_near_field"] = lambda: cfg.rca("tile_near_field", 96, 5, 512, (128, 32, 1), (-10.0e-3, 0, 3.0e-3), (10.0e-3, 0, 8.0e-3), seed=74,
                                           interp=I.Cubic, orientation=0x22, f_number=0.3, pitch=0.2e-3, angles=np.linspace(-10, 10, 5))
    # ... and FORCES / UFORCES (transmit element = a channel; sparse: the first acquisition is skipped, das.glsl:297-299) on IQ samples
    c["tile_forces"] = lambda: cfg.forces("tile_forces", 16, 16, 512, (96, 1, 40), (-1.5e-3, 0, 8.0e-3), (1.5e-3, 0, 10.0e-3), seed=75,
                                          interp=I.Cubic, f_number=0.9, pitch=0.2e-3,
                                          stages=(P.ShaderKind.Demodulate, P.ShaderKind.Decode, P.ShaderKind.DAS))
    c["tile_uforces_cw"] = lambda: cfg.forces("tile_uforces_cw", 16, 8, 512, (64, 1, 48), (-1.2e-3, 0, 8.0e-3), (1.2e-3, 0, 10.4e-3), seed=76,
                                              kind=K.UFORCES, sparse=[1, 4, 6, 8, 11, 13, 15], decode=0, interp=I.Cubic, cw=True,
                                              f_number=0.9, pitch=0.2e-3, stages=(P.ShaderKind.Demodulate, P.ShaderKind.DAS))
    c["uforces_sparse"] = lambda: cfg.forces("uforces_sparse", 16, 8, 512, (16, 1, 16), LO3, HI3, seed=32,
                                             kind=K.UFORCES, sparse=[1, 4, 6, 8, 11, 13, 15], decode=0, interp=I.Cubic,
                                             cw=True)
    c["readi"] = lambda: cfg.forces("readi", 16, 4, 512, (16, 1, 16), LO3, HI3, seed=33, decode=0,
                                    readi_groups=4, readi_group=2)
    c["forces_filter_f32"] = lambda: cfg.forces("forces_filter_f32", 16, 4, 512, (12, 12, 1), LO3, HI3, seed=34,
                                                decode=0, data_kind=D.Float32, stages=(S.Decode, S.Filter, S.DAS))
    return c


CASES = _cases()


def make(name):
    acq = CASES[name]()
    if name == "forces_filter_f32":
        acq.filters = [cfg.kaiser_filter(25e6, 4e6, length=21, beta=4.0)]
    return acq


# tolerance on max|gpu - oracle| / max|oracle| (stated per SURVEY section 8c):
#   f32 staging, linear/cubic 1e-4; stages the reference runs through binary16 2e-3;
#   nearest interpolation is judged by a mismatch fraction instead (a tap can flip)
def tolerance(acq):
    base = P.DATA_KIND_NUMPY[int(acq.bp.data_kind)]
    stages = list(acq.bp.compute_stages[: acq.bp.compute_stages_count])
    f16_staged = base in ("int16", "float16") and (int(S.Demodulate) in stages or int(S.Filter) in stages
                                                   or (int(S.Decode) in stages and acq.bp.decode_mode))
    return 2e-3 if f16_staged else 1e-4


# ---- view planes (round 3): the reference's everyday frame is not a volume but a plane through 3-D data -- its throughput
# harness beamforms a 512 x 1024 XZ plane out of every dataset (tests/throughput.c:20-23, :443-446; das_transform_2d_xz /
# _yz, math.c:844-885: image y is world depth, image z is the plane's normal, one voxel thick)
def _plane_cases():
    c = {}
    for kind in cfg.HARNESS_KINDS:
        c[f"harness_{kind}_small"] = (lambda kind=kind: cfg.harness(kind, 0.0625))
    c["harness_tpw_yz_small"] = lambda: cfg.harness("tpw", 0.0625, "yz")
    c["harness_hercules_yz_small"] = lambda: cfg.harness("hercules", 0.0625, "yz")
    # wide enough along image x (>= 32 voxels, no channel split: 7 channels) for the aligned-grid HERCULES kernel to be the
    # automatic choice on a view plane; the YZ plane puts world y along the lanes: the loop roles swap
    # (1024 samples: the deepest pixels' echoes arrive at sample 344 of the 512 that demodulation leaves.  With 512-sample rows -- the
    # "_short_rows" twins -- rows END inside the image: the kernel's checked loop and its row-end pass (csrc/das_exact.h) run)
    c["hercules_plane_xz"] = lambda: cfg.hercules(
        "hercules_plane_xz", 7, 16, 1024, (64, 48, 1), LO3, HI3, seed=81, cw=True, f_number=0.7, interp=I.Cubic,
        stages=(S.Demodulate, S.Decode, S.DAS), plane="xz", plane_offset=0.4e-3)
    c["hercules_plane_yz"] = lambda: cfg.hercules(
        "hercules_plane_yz", 7, 16, 1024, (64, 40, 1), LO3, HI3, seed=82, cw=True, f_number=0.7,
        stages=(S.Demodulate, S.Decode, S.DAS), plane="yz", plane_offset=-0.3e-3)
    c["hercules_plane_xz_short_rows"] = lambda: cfg.hercules(
        "hercules_plane_xz_short_rows", 7, 16, 512, (64, 48, 1), LO3, HI3, seed=81, cw=True, f_number=0.7, interp=I.Cubic,
        stages=(S.Demodulate, S.Decode, S.DAS), plane="xz", plane_offset=0.4e-3)
    c["hercules_plane_yz_short_rows"] = lambda: cfg.hercules(
        "hercules_plane_yz_short_rows", 7, 16, 512, (64, 40, 1), LO3, HI3, seed=82, cw=True, f_number=0.7,
        stages=(S.Demodulate, S.Decode, S.DAS), plane="yz", plane_offset=-0.3e-3)
    c["uhercules_plane_xz_sparse"] = lambda: cfg.hercules(
        "uhercules_plane_xz_sparse", 6, 8, 512, (56, 33, 1), LO3, HI3, seed=83, kind=K.UHERCULES,
        sparse=[0, 3, 5, 9, 12, 14, 15], decode=0, data_kind=D.Float32Complex, f_number=0.9, orientation=0x21, plane="xz")
    return c


CASES.update(_plane_cases())


# What the automatic selection is EXPECTED to pick for the named cases -- written down, not derived: the rules live in ONE place
# (csrc/das_select.cpp; beamformer_hip_describe_das reports its decision and the reason every other kernel declined), and the
# test below checks (1) that the kernel that ran is the one described and (2) these expectations.
# 0 general, 1 separable-delay gather, 2 LDS-staged, 3 per-voxel factored, 4 HERCULES aligned-grid (5: block-staged factored)
EXPECTED_AUTOMATIC = {
    "config1_small": 0, "config2_small": 3, "config3_small": 0, "config4_small": 1, "config5_small": 0,
    "rca_staged_auto": 2, "rca_staged_w64": 1, "rca_staged_too_wide": 1, "rca_staged_ragged": 2, "rca_vls_staged": 2,
    "rca_staged_real": 2, "rca_staged_cubic": 2, "rca_staged_fine": 2, "rca_sep_ragged_cubic": 3, "rca_sep_real_nearest": 3,
    "hercules_wide_cw": 4, "hercules_wide_real_swapped": 4, "hercules_wide_cubic_cw": 4, "hercules_plane_xz": 4, "hercules_plane_yz": 4,
    "forces": 3, "uforces_sparse": 3, "readi": 0, "harness_tpw_small": 3, "harness_forces_small": 3, "harness_hercules_small": 0,
    "hercules_plane_xz_short_rows": 4, "hercules_plane_yz_short_rows": 4,
}
# ... and how many z-planes of the case the row-end rule hands to the kernel BEHIND that choice (cases not listed: none)
EXPECTED_ROW_END_PLANES = {
    "rca_staged_auto": 1, "rca_staged_fine": 1, "rca_staged_ragged": 1,
    "rca_staged_cubic_short_rows": 1, "rca_staged_real_short_rows": 1, "rca_staged_fine_vls_short_rows": 2, "rca_vls_staged_short_rows": 3,
}
# ... and the cases of which NO plane keeps the staged kernel (their das path is the fallback's)
ROW_END_EVERY_PLANE = {"rca_staged_fine_vls_short_rows", "rca_vls_staged_short_rows"}
