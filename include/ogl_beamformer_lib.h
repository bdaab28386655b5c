/* ogl_beamformer_lib.h -- C ABI of the MI355X-native beamformer core.
 *
 * This is the drop-in boundary for the hot path
 *     RF upload -> [Reshape] -> Decode -> Filter/Demodulate -> DAS -> CoherencyWeighting
 * of rnpnr/ogl_beamforming.  Every type below has the byte layout of the type of
 * the same name in the reference (generated/beamformer.c) and every function has the
 * signature, argument meaning, return convention and error codes of the function of the
 * same name in the reference client library (lib/ogl_beamformer_lib_base.h:37-173,
 * lib/ogl_beamformer_lib.c).  A program written against the reference's generated
 * out/ogl_beamformer_lib.h (build.c:4694-4805: plain C base types, arrays for
 * vectors/matrices) recompiles against this header unchanged and links
 * libogl_beamformer_lib.so from this repository instead of the reference's.
 *
 * What differs is what sits behind the symbols: the reference library writes into a
 * POSIX shared-memory region served by a separate Vulkan process
 * (beamformer_shared_memory.c); this library owns HIP device buffers and launches
 * hand-written gfx950 kernels in-process.  There is no CPU fallback: without a HIP
 * device every call that needs the beamformer fails with
 * BeamformerLibErrorKind_SharedMemory (the reference's "server not reachable" error).
 *
 * Matrices are column-major 4x4 (base_types.h:116-120; math.c:448-458).
 */
#ifndef OGL_BEAMFORMER_LIB_H
#define OGL_BEAMFORMER_LIB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef BEAMFORMER_LIB_EXPORT
  #define BEAMFORMER_LIB_EXPORT __attribute__((visibility("default")))
#endif

/* ---- compile-time limits (generated/beamformer.c:5-13) ---- */
#define BeamformerChunkChannelCount        (16)
#define BeamformerFilterSlots              (4)
#define BeamformerMaxBacklogFrames         (4096)
#define BeamformerMaxChannelCount          (256)
#define BeamformerMaxEmissionsCount        (256)
#define BeamformerMaxComputeShaderStages   (16)
#define BeamformerMaxParameterBlocks       (16)
#define BeamformerMaxRawDataFramesInFlight (3)

/* protocol version reported by beamformer_get_api_version()
 * (beamformer_shared_memory.c:2) */
#define BEAMFORMER_SHARED_MEMORY_VERSION   (33UL)

/* ---- enumerations (generated/beamformer.c:26-174) ---- */
typedef enum {
	BeamformerDecodeMode_None     = 0,
	BeamformerDecodeMode_Hadamard = 1,
	BeamformerDecodeMode_Count,
} BeamformerDecodeMode;

typedef enum {
	BeamformerRCAOrientation_None    = 0,
	BeamformerRCAOrientation_Rows    = 1,
	BeamformerRCAOrientation_Columns = 2,
	BeamformerRCAOrientation_Count,
} BeamformerRCAOrientation;

typedef enum {
	BeamformerSamplingMode_2X = 0,
	BeamformerSamplingMode_4X = 1,
	BeamformerSamplingMode_Count,
} BeamformerSamplingMode;

/* element kind of the RF the client pushes; byte sizes 2,4,4,8,2,4
 * (generated/beamformer.c:507-514) */
typedef enum {
	BeamformerDataKind_Int16          = 0,
	BeamformerDataKind_Int16Complex   = 1,
	BeamformerDataKind_Float32        = 2,
	BeamformerDataKind_Float32Complex = 3,
	BeamformerDataKind_Float16        = 4,
	BeamformerDataKind_Float16Complex = 5,
	BeamformerDataKind_Count,
} BeamformerDataKind;

typedef enum {
	BeamformerContrastMode_None = 0,
	BeamformerContrastMode_A1S2 = 1,
	BeamformerContrastMode_Count,
} BeamformerContrastMode;

typedef enum {
	BeamformerEmissionKind_Sine  = 0,
	BeamformerEmissionKind_Chirp = 1,
	BeamformerEmissionKind_Count,
} BeamformerEmissionKind;

typedef enum {
	BeamformerInterpolationMode_Nearest = 0,
	BeamformerInterpolationMode_Linear  = 1,
	BeamformerInterpolationMode_Cubic   = 2,
	BeamformerInterpolationMode_Count,
} BeamformerInterpolationMode;

typedef enum {
	BeamformerViewPlaneTag_XZ        = 0,
	BeamformerViewPlaneTag_YZ        = 1,
	BeamformerViewPlaneTag_XY        = 2,
	BeamformerViewPlaneTag_Arbitrary = 3,
	BeamformerViewPlaneTag_Count,
} BeamformerViewPlaneTag;

typedef enum {
	BeamformerAcquisitionKind_FORCES         = 0,
	BeamformerAcquisitionKind_UFORCES        = 1,
	BeamformerAcquisitionKind_HERCULES       = 2,
	BeamformerAcquisitionKind_RCA_VLS        = 3,
	BeamformerAcquisitionKind_RCA_TPW        = 4,
	BeamformerAcquisitionKind_UHERCULES      = 5,
	BeamformerAcquisitionKind_RACES          = 6,
	BeamformerAcquisitionKind_EPIC_FORCES    = 7,
	BeamformerAcquisitionKind_EPIC_UFORCES   = 8,
	BeamformerAcquisitionKind_EPIC_UHERCULES = 9,
	BeamformerAcquisitionKind_Flash          = 10,
	BeamformerAcquisitionKind_HERO_PA        = 11,
	BeamformerAcquisitionKind_ULM            = 12,
	BeamformerAcquisitionKind_Count,
} BeamformerAcquisitionKind;

typedef enum {
	BeamformerFilterKind_Kaiser       = 0,
	BeamformerFilterKind_MatchedChirp = 1,
	BeamformerFilterKind_Count,
} BeamformerFilterKind;

typedef enum {
	BeamformerLiveFeedbackFlags_ImagePlaneOffsets = 0,
	BeamformerLiveFeedbackFlags_TransmitPower     = 1,
	BeamformerLiveFeedbackFlags_TGCControlPoints  = 2,
	BeamformerLiveFeedbackFlags_SaveData          = 3,
	BeamformerLiveFeedbackFlags_SaveNameTag       = 4,
	BeamformerLiveFeedbackFlags_StopImaging       = 5,
	BeamformerLiveFeedbackFlags_AcquisitionKind   = 6,
	BeamformerLiveFeedbackFlags_Count,
} BeamformerLiveFeedbackFlags;

/* stage identifiers a client may place in a pipeline are Decode..Hilbert
 * (lib/ogl_beamformer_lib.c:289-292); the rest are inserted by the planner */
typedef enum {
	BeamformerShaderKind_Decode             = 0,
	BeamformerShaderKind_Filter             = 1,
	BeamformerShaderKind_Demodulate         = 2,
	BeamformerShaderKind_DAS                = 3,
	BeamformerShaderKind_Hilbert            = 4,
	BeamformerShaderKind_CoherencyWeighting = 5,
	BeamformerShaderKind_Reshape            = 6,
	BeamformerShaderKind_MinMax             = 7,
	BeamformerShaderKind_Sum                = 8,
	BeamformerShaderKind_RenderBeamformed   = 9,
	BeamformerShaderKind_Count,

	BeamformerShaderKind_ComputeFirst = BeamformerShaderKind_Decode,
	BeamformerShaderKind_ComputeLast  = BeamformerShaderKind_Hilbert,
	BeamformerShaderKind_ComputeCount = 5,
} BeamformerShaderKind;

/* ---- parameter structures ---- */

/* generated/beamformer.c:304-321 */
typedef struct { float cycles; float frequency; } BeamformerSineParameters;
typedef struct { float duration; float min_frequency; float max_frequency; } BeamformerChirpParameters;
typedef struct {
	BeamformerEmissionKind kind;
	union {
		BeamformerSineParameters  sine;
		BeamformerChirpParameters chirp;
	};
} BeamformerEmissionParameters;

/* generated/beamformer.c:323-343 */
typedef struct { float cutoff_frequency; float beta; uint32_t length; } BeamformerKaiserFilterParameters;
typedef struct { float duration; float min_frequency; float max_frequency; } BeamformerMatchedChirpFilterParameters;
typedef struct {
	BeamformerFilterKind kind;
	float                sampling_frequency;
	uint32_t             complex;
	union {
		BeamformerKaiserFilterParameters       kaiser;
		BeamformerMatchedChirpFilterParameters matched_chirp;
	};
} BeamformerFilterParameters;

/* The fields every frame is computed from.  Layout: generated/beamformer.c:381-409
 * (IDL beamformer.meta:171-217); 264 bytes, 4-byte aligned. */
#define BEAMFORMER_PARAMETERS_FIELDS \
	float    das_voxel_transform[16];       /* unit cube -> world [m], column major     */ \
	float    xdc_transform[16];             /* world -> transducer, column major        */ \
	float    xdc_element_pitch[2];          /* [m]                                      */ \
	uint32_t raw_data_dimensions[2];        /* elements per raw row, raw rows           */ \
	float    focal_vector[2];               /* angle [deg], focal depth [m] (inf: plane)*/ \
	uint32_t transmit_receive_orientation;  /* rx = bits 0-3, tx = bits 4-7             */ \
	uint32_t sample_count;                                                                 \
	uint32_t channel_count;                                                                \
	uint32_t acquisition_count;                                                            \
	BeamformerAcquisitionKind acquisition_kind;                                            \
	BeamformerDecodeMode      decode_mode;                                                 \
	BeamformerSamplingMode    sampling_mode;                                               \
	float    time_offset;                   /* [s]                                      */ \
	uint32_t single_focus;                                                                 \
	uint32_t single_orientation;                                                           \
	int32_t  output_points[4];              /* x, y, z voxels; w = frames to average    */ \
	float    sampling_frequency;            /* [Hz]                                     */ \
	float    demodulation_frequency;        /* [Hz]                                     */ \
	float    speed_of_sound;                /* [m/s]                                    */ \
	float    f_number;                                                                     \
	BeamformerInterpolationMode interpolation_mode;                                        \
	uint32_t coherency_weighting;                                                          \
	uint32_t decimation_rate;                                                              \
	BeamformerContrastMode       contrast_mode;                                            \
	BeamformerEmissionParameters emission_parameters;                                      \
	uint32_t readi_group_count;                                                            \
	uint32_t readi_group;

typedef struct { BEAMFORMER_PARAMETERS_FIELDS } BeamformerParameters;

/* generated/beamformer.c:411-448; 3728 bytes */
typedef struct {
	BEAMFORMER_PARAMETERS_FIELDS
	int16_t  channel_mapping[BeamformerMaxChannelCount];
	int16_t  sparse_elements[BeamformerMaxEmissionsCount];
	uint8_t  transmit_receive_orientations[BeamformerMaxEmissionsCount];
	float    steering_angles[BeamformerMaxEmissionsCount];
	float    focal_depths[BeamformerMaxEmissionsCount];
	int32_t  compute_stages[BeamformerMaxComputeShaderStages];
	int32_t  compute_stage_parameters[BeamformerMaxComputeShaderStages];
	uint32_t compute_stages_count;
	BeamformerDataKind data_kind;
} BeamformerSimpleParameters;

/* generated/beamformer.c:450-461; 208 bytes */
typedef struct {
	uint32_t active;
	uint32_t save_enabled;
	uint32_t save_active;
	uint32_t acquisition_kind;
	uint64_t acquisition_kind_enabled_flags;
	float    transmit_power;
	float    image_plane_offsets[BeamformerViewPlaneTag_Count];
	float    tgc_control_points[8];
	int32_t  save_name_tag_length;
	uint8_t  save_name_tag[128];
} BeamformerLiveImagingParameters;

/* per-stage seconds of the last 32 frames (beamformer_compute_stats.c:3-10); 2248 bytes */
typedef struct {
	uint64_t shader_count;
	uint32_t shader_ids[BeamformerMaxComputeShaderStages];
	float    times[32][BeamformerMaxComputeShaderStages];
	float    rf_time_deltas[32];
} BeamformerComputeStatsTable;

/* ---- errors (lib/ogl_beamformer_lib_base.h:10-35) ---- */
typedef enum {
	BeamformerLibErrorKind_None                        =  0,
	BeamformerLibErrorKind_VersionMismatch             =  1,
	BeamformerLibErrorKind_InvalidAccess               =  2,
	BeamformerLibErrorKind_ParameterBlockOverflow      =  3,
	BeamformerLibErrorKind_ParameterBlockUnallocated   =  4,
	BeamformerLibErrorKind_ComputeStageOverflow        =  5,
	BeamformerLibErrorKind_InvalidComputeStage         =  6,
	BeamformerLibErrorKind_InvalidStartShader          =  7,
	BeamformerLibErrorKind_InvalidDemodulationDataKind =  8,
	BeamformerLibErrorKind_InvalidImagePlane           =  9,
	BeamformerLibErrorKind_InvalidFilterKind           = 10,
	BeamformerLibErrorKind_InvalidDataKind             = 11,
	BeamformerLibErrorKind_InvalidContrastMode         = 12,
	BeamformerLibErrorKind_BufferOverflow              = 13,
	BeamformerLibErrorKind_DataSizeMismatch            = 14,
	BeamformerLibErrorKind_WorkQueueFull               = 15,
	BeamformerLibErrorKind_ExportSpaceOverflow         = 16,
	BeamformerLibErrorKind_SharedMemory                = 17,
	BeamformerLibErrorKind_SyncVariable                = 18,
	BeamformerLibErrorKind_FrameSizeOverflow           = 19,
	BeamformerLibErrorKind_RFDataSizeOverflow          = 20,
} BeamformerLibErrorKind;

/* ---- functions ----
 * Unless noted: returns 1 on success, 0 on failure with the reason left in
 * beamformer_get_last_error() (sticky, like the reference's global, lib .c:29-34).
 * The caller owns every pointer; data is copied in or out before the call returns. */

/* lib .c:206-210 */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_get_api_version(void);

/* lib .c:212-231 */
BEAMFORMER_LIB_EXPORT BeamformerLibErrorKind beamformer_get_last_error(void);
BEAMFORMER_LIB_EXPORT const char *beamformer_get_last_error_string(void);
BEAMFORMER_LIB_EXPORT const char *beamformer_error_string(BeamformerLibErrorKind kind);

/* how many frames fit the frame ring before the oldest is overwritten; UINT64_MAX on
 * error (lib .c:325-347) */
BEAMFORMER_LIB_EXPORT uint64_t beamformer_maximum_frames_for_parameters(BeamformerParameters *);
BEAMFORMER_LIB_EXPORT uint64_t beamformer_maximum_frames_for_simple_parameters(BeamformerSimpleParameters *);

/* largest single RF data set that can be pushed; UINT64_MAX on error (lib .c:313-323) */
BEAMFORMER_LIB_EXPORT uint64_t beamformer_maximum_rf_data_size(void);

/* one-shot: push parameters, push data, compute, optionally pull the image
 * (lib .c:704-736).  out_data (may be 0) receives X*Y*Z float32 values, x2 if any
 * stage is Demodulate or Hilbert. */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_beamform_data(BeamformerSimpleParameters *bp, void *data,
                                                        uint32_t data_size, void *out_data,
                                                        int32_t timeout_ms);

/* timeout for calls without a timeout argument; default 0, (uint32_t)-1 blocks forever
 * (lib .c:233-237) */
BEAMFORMER_LIB_EXPORT void beamformer_set_global_timeout(uint32_t timeout_ms);

/* copy the channel-mapped RF to the device and queue one frame of compute on the
 * parameter block; returns before the frame is finished (lib .c:572-594) */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_push_data_with_compute(void *data, uint32_t size,
                                                                 uint32_t image_plane_tag,
                                                                 uint32_t parameter_slot);

/* the last `count` frames, oldest first, each rounded up to 64 bytes; waits for
 * outstanding compute (lib .c:693-702, beamformer_core.c:1474-1494) */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_get_last_frames(void *out_data, uint64_t out_data_size, uint32_t count);

/* lib .c:239-250 */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_reserve_parameter_blocks(uint32_t count);

/* lib .c:364-408 */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_set_pipeline_stage_parameters(uint32_t stage_index, int32_t parameter);
BEAMFORMER_LIB_EXPORT uint32_t beamformer_set_pipeline_stage_parameters_at(uint32_t stage_index, int32_t parameter,
                                                                           uint32_t parameter_slot);
BEAMFORMER_LIB_EXPORT uint32_t beamformer_push_pipeline(int32_t *shaders, uint32_t shader_count,
                                                        BeamformerDataKind data_kind);
BEAMFORMER_LIB_EXPORT uint32_t beamformer_push_pipeline_at(int32_t *shaders, uint32_t shader_count,
                                                           BeamformerDataKind data_kind, uint32_t parameter_slot);

/* lib .c:596-653 */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_push_simple_parameters(BeamformerSimpleParameters *bp);
BEAMFORMER_LIB_EXPORT uint32_t beamformer_push_simple_parameters_at(BeamformerSimpleParameters *bp, uint32_t parameter_slot);
BEAMFORMER_LIB_EXPORT uint32_t beamformer_push_parameters(BeamformerParameters *);
BEAMFORMER_LIB_EXPORT uint32_t beamformer_push_parameters_at(BeamformerParameters *, uint32_t parameter_slot);

/* lib .c:438-464 */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_push_channel_mapping(int16_t *mapping, uint32_t count);
BEAMFORMER_LIB_EXPORT uint32_t beamformer_push_channel_mapping_at(int16_t *mapping, uint32_t count, uint32_t parameter_slot);
BEAMFORMER_LIB_EXPORT uint32_t beamformer_push_sparse_elements(int16_t *elements, uint32_t count);
BEAMFORMER_LIB_EXPORT uint32_t beamformer_push_sparse_elements_at(int16_t *elements, uint32_t count, uint32_t parameter_slot);
/* count (angle [deg], depth [m]) pairs */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_push_focal_vectors(float *vectors, uint32_t count);
BEAMFORMER_LIB_EXPORT uint32_t beamformer_push_focal_vectors_at(float *vectors, uint32_t count, uint32_t parameter_slot);
BEAMFORMER_LIB_EXPORT uint32_t beamformer_push_transmit_receive_orientations(uint8_t *values, uint32_t count);
BEAMFORMER_LIB_EXPORT uint32_t beamformer_push_transmit_receive_orientations_at(uint8_t *values, uint32_t count,
                                                                                uint32_t parameter_slot);

/* lib .c:410-429; coefficients are generated on the host exactly as
 * beamformer_core.c:366-398 does */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_create_filter(BeamformerFilterParameters *filter,
                                                        uint8_t filter_slot, uint8_t parameter_block);

/* lib .c:756-788 */
BEAMFORMER_LIB_EXPORT int32_t  beamformer_live_parameters_get_dirty_flag(void);
BEAMFORMER_LIB_EXPORT uint32_t beamformer_set_live_parameters(BeamformerLiveImagingParameters *);
BEAMFORMER_LIB_EXPORT BeamformerLiveImagingParameters *beamformer_get_live_parameters(void);

/* lib .c:738-754 (exported but absent from the reference's base header) */
BEAMFORMER_LIB_EXPORT uint32_t beamformer_compute_timings(BeamformerComputeStatsTable *output, int32_t timeout_ms);

#ifdef __cplusplus
}

static_assert(sizeof(BeamformerParameters)            ==  264, "layout: generated/beamformer.c:381-409");
static_assert(sizeof(BeamformerSimpleParameters)      == 3728, "layout: generated/beamformer.c:411-448");
static_assert(sizeof(BeamformerFilterParameters)      ==   24, "layout: generated/beamformer.c:335-343");
static_assert(sizeof(BeamformerEmissionParameters)    ==   16, "layout: generated/beamformer.c:315-321");
static_assert(sizeof(BeamformerLiveImagingParameters) ==  208, "layout: generated/beamformer.c:450-461");
static_assert(sizeof(BeamformerComputeStatsTable)     == 2248, "layout: beamformer_compute_stats.c:3-10");
#endif

#endif /* OGL_BEAMFORMER_LIB_H */
