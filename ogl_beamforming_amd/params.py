"""ctypes mirrors of the parameter structures of include/ogl_beamformer_lib.h.

The reference ships the same structures to Python through a cffi-ready preprocessed
header (build.c:4798-4800) and to MATLAB through generated classes; field names, order
and sizes follow generated/beamformer.c:304-461 of the reference.
"""
import ctypes as C
import enum

MAX_CHANNELS = 256
MAX_EMISSIONS = 256
MAX_STAGES = 16
MAX_PARAMETER_BLOCKS = 16
FILTER_SLOTS = 4


class DataKind(enum.IntEnum):
    Int16 = 0
    Int16Complex = 1
    Float32 = 2
    Float32Complex = 3
    Float16 = 4
    Float16Complex = 5


DATA_KIND_BYTES = {0: 2, 1: 4, 2: 4, 3: 8, 4: 2, 5: 4}
DATA_KIND_NUMPY = {0: "int16", 1: "int16", 2: "float32", 3: "float32", 4: "float16", 5: "float16"}
DATA_KIND_COMPLEX = {0: False, 1: True, 2: False, 3: True, 4: False, 5: True}


class ShaderKind(enum.IntEnum):
    Decode = 0
    Filter = 1
    Demodulate = 2
    DAS = 3
    Hilbert = 4
    CoherencyWeighting = 5
    Reshape = 6
    MinMax = 7
    Sum = 8


class AcquisitionKind(enum.IntEnum):
    FORCES = 0
    UFORCES = 1
    HERCULES = 2
    RCA_VLS = 3
    RCA_TPW = 4
    UHERCULES = 5
    RACES = 6
    EPIC_FORCES = 7
    EPIC_UFORCES = 8
    EPIC_UHERCULES = 9
    Flash = 10
    HERO_PA = 11
    ULM = 12


class InterpolationMode(enum.IntEnum):
    Nearest = 0
    Linear = 1
    Cubic = 2


class RCAOrientation(enum.IntEnum):
    None_ = 0
    Rows = 1
    Columns = 2


class FilterKind(enum.IntEnum):
    Kaiser = 0
    MatchedChirp = 1


class LibError(enum.IntEnum):
    None_ = 0
    VersionMismatch = 1
    InvalidAccess = 2
    ParameterBlockOverflow = 3
    ParameterBlockUnallocated = 4
    ComputeStageOverflow = 5
    InvalidComputeStage = 6
    InvalidStartShader = 7
    InvalidDemodulationDataKind = 8
    InvalidImagePlane = 9
    InvalidFilterKind = 10
    InvalidDataKind = 11
    InvalidContrastMode = 12
    BufferOverflow = 13
    DataSizeMismatch = 14
    WorkQueueFull = 15
    ExportSpaceOverflow = 16
    SharedMemory = 17
    SyncVariable = 18
    FrameSizeOverflow = 19
    RFDataSizeOverflow = 20


class SineParameters(C.Structure):
    _fields_ = [("cycles", C.c_float), ("frequency", C.c_float)]


class ChirpParameters(C.Structure):
    _fields_ = [("duration", C.c_float), ("min_frequency", C.c_float), ("max_frequency", C.c_float)]


class _EmissionUnion(C.Union):
    _fields_ = [("sine", SineParameters), ("chirp", ChirpParameters)]


class EmissionParameters(C.Structure):
    _anonymous_ = ("u",)
    _fields_ = [("kind", C.c_int32), ("u", _EmissionUnion)]


class KaiserFilterParameters(C.Structure):
    _fields_ = [("cutoff_frequency", C.c_float), ("beta", C.c_float), ("length", C.c_uint32)]


class MatchedChirpFilterParameters(C.Structure):
    _fields_ = [("duration", C.c_float), ("min_frequency", C.c_float), ("max_frequency", C.c_float)]


class _FilterUnion(C.Union):
    _fields_ = [("kaiser", KaiserFilterParameters), ("matched_chirp", MatchedChirpFilterParameters)]


class FilterParameters(C.Structure):
    _anonymous_ = ("u",)
    _fields_ = [("kind", C.c_int32), ("sampling_frequency", C.c_float), ("complex", C.c_uint32),
                ("u", _FilterUnion)]


_PARAMETER_FIELDS = [
    ("das_voxel_transform", C.c_float * 16),
    ("xdc_transform", C.c_float * 16),
    ("xdc_element_pitch", C.c_float * 2),
    ("raw_data_dimensions", C.c_uint32 * 2),
    ("focal_vector", C.c_float * 2),
    ("transmit_receive_orientation", C.c_uint32),
    ("sample_count", C.c_uint32),
    ("channel_count", C.c_uint32),
    ("acquisition_count", C.c_uint32),
    ("acquisition_kind", C.c_int32),
    ("decode_mode", C.c_int32),
    ("sampling_mode", C.c_int32),
    ("time_offset", C.c_float),
    ("single_focus", C.c_uint32),
    ("single_orientation", C.c_uint32),
    ("output_points", C.c_int32 * 4),
    ("sampling_frequency", C.c_float),
    ("demodulation_frequency", C.c_float),
    ("speed_of_sound", C.c_float),
    ("f_number", C.c_float),
    ("interpolation_mode", C.c_int32),
    ("coherency_weighting", C.c_uint32),
    ("decimation_rate", C.c_uint32),
    ("contrast_mode", C.c_int32),
    ("emission_parameters", EmissionParameters),
    ("readi_group_count", C.c_uint32),
    ("readi_group", C.c_uint32),
]


class Parameters(C.Structure):
    _fields_ = list(_PARAMETER_FIELDS)


class SimpleParameters(C.Structure):
    _fields_ = list(_PARAMETER_FIELDS) + [
        ("channel_mapping", C.c_int16 * MAX_CHANNELS),
        ("sparse_elements", C.c_int16 * MAX_EMISSIONS),
        ("transmit_receive_orientations", C.c_uint8 * MAX_EMISSIONS),
        ("steering_angles", C.c_float * MAX_EMISSIONS),
        ("focal_depths", C.c_float * MAX_EMISSIONS),
        ("compute_stages", C.c_int32 * MAX_STAGES),
        ("compute_stage_parameters", C.c_int32 * MAX_STAGES),
        ("compute_stages_count", C.c_uint32),
        ("data_kind", C.c_int32),
    ]


class LiveImagingParameters(C.Structure):
    _fields_ = [
        ("active", C.c_uint32), ("save_enabled", C.c_uint32), ("save_active", C.c_uint32),
        ("acquisition_kind", C.c_uint32), ("acquisition_kind_enabled_flags", C.c_uint64),
        ("transmit_power", C.c_float), ("image_plane_offsets", C.c_float * 4),
        ("tgc_control_points", C.c_float * 8), ("save_name_tag_length", C.c_int32),
        ("save_name_tag", C.c_uint8 * 128),
    ]


class ComputeStatsTable(C.Structure):
    _fields_ = [
        ("shader_count", C.c_uint64), ("shader_ids", C.c_uint32 * MAX_STAGES),
        ("times", (C.c_float * MAX_STAGES) * 32), ("rf_time_deltas", C.c_float * 32),
    ]


HIP_MAX_TIMED_STAGES = 24


class HipFrameInfo(C.Structure):
    _fields_ = [("device_pointer", C.c_void_p), ("size_bytes", C.c_uint64), ("points", C.c_uint32 * 3),
                ("data_kind", C.c_uint32), ("frame_id", C.c_uint32), ("parameter_block", C.c_uint32)]


class HipFrameTimings(C.Structure):
    _fields_ = [
        ("stage_count", C.c_uint32), ("stage_kind", C.c_uint32 * HIP_MAX_TIMED_STAGES),
        ("stage_ms", C.c_float * HIP_MAX_TIMED_STAGES), ("frame_ms", C.c_float),
        ("das_pairs", C.c_uint64), ("das_voxels", C.c_uint64), ("das_taps", C.c_uint32),
        ("das_sample_bytes", C.c_uint32), ("das_path", C.c_uint32), ("staged_window_violations", C.c_uint32),
        ("tile_staged_chunks", C.c_uint32), ("tile_gather_chunks", C.c_uint32), ("das_row_end_planes", C.c_uint32),
    ]


class DasPath(enum.IntEnum):
    """BeamformerHipFrameTimings::das_path / BeamformerHipDasDescription::path (csrc/das_select.h)"""
    General = 0
    Gather = 1
    Staged = 2
    Factored = 3
    Hercules = 4
    Tile = 5


class HipDeviceInfo(C.Structure):
    _fields_ = [("ordinal", C.c_int32), ("peer_access", C.c_int32), ("slab_first", C.c_uint32), ("slab_count", C.c_uint32),
                ("peer_copy_ms", C.c_float), ("das_ms", C.c_float), ("frame_ms", C.c_float),
                ("rf_checksum", C.c_uint64), ("rf_bytes", C.c_uint64)]


class HipDasDescription(C.Structure):
    _fields_ = [("path", C.c_int32), ("kernel", C.c_char * 48), ("name", C.c_char * 64), ("declined", (C.c_char * 160) * 8),
                ("tile_shift", C.c_uint32 * 3), ("blocks", C.c_uint32 * 3), ("split_shift", C.c_uint32), ("tile_walk", C.c_uint32),
                ("row_end_planes", C.c_uint32), ("tile_window_samples", C.c_uint32), ("u_axis", C.c_uint32), ("u_shift", C.c_uint32), ("v_shift", C.c_uint32), ("window_samples", C.c_uint32),
                ("uniform_tables", C.c_uint32), ("lds_bytes", C.c_uint32), ("threads", C.c_uint32), ("channel_chunk", C.c_uint32),
                ("hercules_prepared_copy", C.c_uint32), ("tile_spread_estimate", C.c_float), ("tile_estimate_shift", C.c_uint32 * 3), ("row_ends", C.c_uint32)]


class HipPlanStage(C.Structure):
    _fields_ = [("kind", C.c_int32), ("in_kind", C.c_int32), ("out_kind", C.c_int32),
                ("in_stride", C.c_int64 * 3), ("out_stride", C.c_int64 * 3)]


class HipPlan(C.Structure):
    _fields_ = [("stage_count", C.c_uint32), ("stages", HipPlanStage * MAX_STAGES),
                ("das_samples", C.c_uint32), ("iq_pipeline", C.c_uint32),
                ("das_sampling_frequency", C.c_float), ("das_time_offset", C.c_float),
                ("das_voxel_transform", C.c_float * 16)]


class HipZbpPayload(C.Structure):
    _fields_ = [("major", C.c_uint32), ("data_kind", C.c_uint32), ("compression_kind", C.c_uint32),
                ("reserved", C.c_uint32), ("offset", C.c_uint64), ("size", C.c_uint64)]


assert C.sizeof(Parameters) == 264
assert C.sizeof(SimpleParameters) == 3728
assert C.sizeof(FilterParameters) == 24
assert C.sizeof(LiveImagingParameters) == 208
assert C.sizeof(ComputeStatsTable) == 2248
