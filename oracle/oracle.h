/* oracle.h -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A literal, single-precision CPU restatement of the reference's hot path
 *   RF ingest -> [Reshape] -> Decode -> Filter/Demodulate -> DAS -> CoherencyWeighting
 * written by reading shaders/{reshape,decode,filter,das,coherency_weighting}.glsl,
 * beamformer_core.c:553-1013 (planner), :1289-1400 + :1519-1626 (executor), math.c and
 * lib/ogl_beamformer_lib.c:491-570 of rnpnr/ogl_beamforming.  Each function cites the
 * lines it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call
 * this code, and only as the checker.  The product (libogl_beamformer_lib.so) never
 * links it and has no CPU fallback.
 *
 * PARITY STATUS: the host math (Hadamard, Kaiser, chirps, windows, filter moments,
 * DAS transforms) is pinned against the compiled reference (oracle/_ref/libref_math.so,
 * fixtures under tests/golden/).  The shader restatements are PARITY UNPINNED by the
 * reference: its tests/ hold no expected outputs for this path and its GLSL cannot be
 * built or run here (no Vulkan loader/ICD, empty glslang submodule).  They are pinned
 * only by physics known-answer tests and a float64 twin (oracle_das_f64).
 *
 * Documented departures from the reference's text (each has a test):
 *   Q1  Hadamard orders 12*2^k and 20*2^k are produced (math.c:96 returns NULL for them).
 *   Q2  Float16 data is IEEE half in Reshape (reshape.glsl:7-10 reads it as int16).
 *   Q3  IQ rotation arguments are range-reduced (turns = frac(f*t)) before sin/cos;
 *       the reference feeds ~1e3..1e4 rad to the driver's sin/cos.
 *   Q4  Decode bounds the sample loop by the sample count (decode.glsl:46,:125 use
 *       OutputTransmitStride, equal to it only for DAS-layout output).
 *   Q5  The last 16-channel chunk is clamped to the channels that exist
 *       (beamformer_core.c:595,:1604-1606 run a full chunk past the end of the data).
 *   Q6  A plan that starts with DAS (Float32/Float32Complex data, decode off) beamforms
 *       the ingested RF directly (the reference reads an unwritten slot,
 *       beamformer_core.c:1353-1362).
 *   Q7  A Kaiser filter always has real taps (beamformer_core.c:372-377); the ComplexFilter
 *       flag is honoured only for matched chirps (the reference would read real taps as
 *       pairs, beamformer_core.c:833,:840).
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>
#include "../include/ogl_beamformer_lib.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- host math (oracle_math.c) ---------------- */
/* math.c:35-134; out[dim*dim] of +-1 as float.  Returns 0 if no construction exists. */
int    oracle_hadamard_transpose(int dim, float *out);
double oracle_bessel_i0(double x);                                   /* external/cephes.c:24-103 */
void   oracle_kaiser_low_pass(float cutoff, float fs, float beta, int length, float *out); /* math.c:750-767 */
float  oracle_tukey_window(float t, float tapering);                 /* math.c:739-747 */
void   oracle_rf_chirp(float fmin, float fmax, float fs, int length, int reverse, float *out);      /* math.c:769-781 */
void   oracle_baseband_chirp(float fmin, float fmax, float fs, int length, int reverse,
                             float scale, float *out /* 2*length */);                             /* math.c:783-797 */
float  oracle_real_filter_first_moment(const float *h, int length, float fs);                      /* math.c:726-737 */
float  oracle_complex_filter_first_moment(const float *h, int length, float fs);                   /* math.c:713-724 */
void   oracle_m4_mul(const float *a, const float *b, float *out);                                  /* math.c:448-458 */
void   oracle_das_transform(const float *min3, const float *max3, int *points3, float *out16);     /* math.c:799-920 */
void   oracle_das_transform_2d(int plane, const float *min2, const float *max2, float offset, float *out16);
void   oracle_das_transform_3d(const float *min3, const float *max3, float *out16);

/* beamformer_core.c:366-398.  coefficients: length (real) or 2*length (complex) floats.
 * Returns length, writes time_delay. */
int    oracle_filter_create(const BeamformerFilterParameters *fp, float *coefficients, int cap,
                            float *time_delay);

/* ---------------- stages (oracle_stages.c) ---------------- */
/* lib/ogl_beamformer_lib.c:491-570: dst[ch][a*S+s] = raw[channel_mapping[ch]][...],
 * A1S2 contrast reduction a-b-c per sample.  dst holds C*A*S elements of data_kind. */
void oracle_channel_map(const void *raw, void *dst, const BeamformerParameters *bp, int data_kind,
                        const int16_t *channel_mapping);

typedef struct {
	int   size[3];            /* x = samples, y = channels, z = transmits (reshape.glsl:61-65) */
	int   in_stride[3], out_stride[3];
	int   in_kind, out_kind;  /* BeamformerDataKind */
	int   interleave;         /* reshape.glsl:71-75 */
} OracleReshape;
void oracle_reshape(const OracleReshape *r, const void *left, const void *right, void *out);

typedef struct {
	int   transmit_count, chunk_channel_count, sample_count; /* sample_count = dispatch extent */
	int   active_channels;    /* channels of the chunk that exist (0: all chunk_channel_count) */
	int   out_stride[3];      /* sample, channel, transmit */
	int   in_kind, out_kind;
	const float *hadamard;    /* transmit_count^2, as uploaded (f16-exact +-1) */
} OracleDecode;
/* decode.glsl:24-73 / :119-150 (both produce the same sums; order j ascending) */
void oracle_decode(const OracleDecode *d, const void *in, void *out);

typedef struct {
	int   filter_length, complex_filter, demodulate;
	float sampling_frequency, demodulation_frequency;  /* only read when demodulate */
	int   decimation_rate;
	int   sample_count;       /* FilterBake.SampleCount (already /2D when demodulating) */
	int   batch_sample_count; /* != 0: deinterleave */
	int   in_stride[3], out_stride[3];
	int   in_kind, out_kind;
	int   channels, transmits;
	int64_t in_elements;      /* elements of in_kind readable from `in` (reads beyond give 0) */
	int   workgroup;          /* gl_WorkGroupSize.x (64): fixes the LDS-local phase index */
	const float *coefficients;
} OracleFilter;
/* filter.glsl:68-135 */
void oracle_filter(const OracleFilter *f, const void *in, void *out, uint32_t output_element_offset);

/* coherency_weighting.glsl:28-37 */
void oracle_coherency_weighting(float *coherent, const float *incoherent, uint32_t voxels,
                                int complex_data, float scale);

/* BUILD-DEFINED Hilbert stage (PARITY UNPINNED).  The reference's Hilbert stage is an out-of-tree
 * CUDA routine that its snapshot cannot load (capabilities.hilbert = 0, beamformer.c:262-263), so
 * there is nothing to restate.  This build defines it as the analytic signal x + j H{x} along
 * samples with a 63-tap type-III FIR Hilbert transformer, Hamming window:
 *   y[n] = sum_{j<63} h[j] x[n - 62 + j],  h[31] = 1,  h[31 + m] = -j (2 / (pi m)) w[31 + m] for odd m,
 * i.e. real part = the input delayed by 31 samples, imaginary part = its Hilbert transform with
 * the same delay, which the planner adds to the DAS time offset (31 / fs).  Off unless
 * oracle_enable_hilbert(1) (the product: beamformer_hip_enable_hilbert). */
#define ORACLE_HILBERT_LENGTH 63
void oracle_enable_hilbert(int enable);
void oracle_hilbert_fir(float *taps_re_im /* 2 * ORACLE_HILBERT_LENGTH */);
void oracle_hilbert(const OracleFilter *f, const void *in, void *out);

/* sum.glsl:7-12, one pass: out += prescale * in over `floats` components */
void oracle_sum(float *out, const float *in, float prescale, uint64_t floats);

/* sample_value of render_3d.frag.glsl:50-73 on one frame: out[i] in [0,1] */
void oracle_display(const float *frame, uint64_t voxels, int complex_data, float threshold_db, float gamma,
                    float db_cutoff, float *out);

/* build-defined reduction (shaders/min_max.glsl is dead code in the reference):
 * min and max over voxels of |v| (complex) or v (real).  PARITY UNPINNED. */
void oracle_min_max(const float *frame, uint64_t voxels, int complex_data, float *out2);

/* ---------------- DAS (oracle_das.c) ---------------- */
typedef struct {
	/* BeamformerDASBakeParameters (generated/beamformer.c:206-231) */
	uint32_t acquisition_kind;
	int32_t  sparse;
	int32_t  acquisition_count, channel_count, chunk_channel_count, sample_count;
	float    sampling_frequency, demodulation_frequency, speed_of_sound, time_offset;
	uint32_t interpolation_mode;
	float    f_number;
	int32_t  single_orientation;
	uint32_t transmit_receive_orientation;
	int32_t  single_focus;
	float    focus_depth, transmit_angle;
	uint32_t output_size[3];
	uint32_t readi_group_count;
	int32_t  coherency_weighting;
	int32_t  complex_data;        /* InputDataKind == Float32Complex */
	/* BeamformerDASPushConstants (generated/beamformer.c:261-269) */
	float    xdc_transform[16], voxel_transform[16], xdc_element_pitch[2];
	uint32_t rf_element_offset;
	int32_t  channel_offset;
	uint32_t readi_group;
	/* BeamformerComputeArrayParameters (generated/beamformer.c:463-467) */
	const float   *focal_vectors;                /* [256][2] */
	const int16_t *sparse_elements;              /* [256] */
	const uint8_t *transmit_receive_orientations;/* [256] */
	const float   *readi_hadamard;               /* readi_group_count^2 or NULL */
	/* build extension: z-slab of the output grid (whole grid when z_count == 0) */
	uint32_t z_first, z_count;
	uint32_t y_first, y_count;    /* likewise rows (whole when y_count == 0); output holds only the sub-grid */
	uint32_t z_stride, y_stride;  /* sub-grid sampling: plane z_first + k z_stride, row y_first + k y_stride (0 = 1) */
	int32_t  threads;             /* OpenMP threads, 0 = default */
	/* build extension: only rows [row_first, row_first + row_count) of the sub-grid's z x y rows (all when row_count == 0):
	 * oracle_beamform's rows-outermost schedule hands every thread its own rows and walks the channel chunks inside */
	int64_t  row_first, row_count;
} OracleDAS;
/* das.glsl:368-407: output[...] += sum, incoherent[...] += |.| sums.
 * rf: elements (float or float pair); output: float or float pair per voxel.
 * Returns the number of (voxel, channel, transmit) triples that passed the
 * apodization test (G of BASELINE.md section 4). */
uint64_t oracle_das(const OracleDAS *p, const float *rf, float *output, float *incoherent);
/* Schedule of oracle_beamform's DAS (test infrastructure; the frames are bit-identical either way, tests/test_oracle.py):
 * 0 = the reference's literal order -- per 16-channel chunk one pass over the image (beamformer_core.c:1604-1614), the rows of a
 * pass shared out to the threads, a fork / join per chunk; 1 (default) = ONE parallel region: every chunk's DAS input is kept,
 * each thread owns a fixed set of rows of the image for the whole frame and walks the chunks in the reference's order over them --
 * the same additions per voxel in the same order, no barrier between chunks (bench.py's cpu_baseline on a many-core host) */
void oracle_set_rows_outermost(int enable);
/* float64 twin of the same loops (truth for tolerance budgeting); outputs double */
uint64_t oracle_das_f64(const OracleDAS *p, const float *rf, double *output, double *incoherent);

/* ---------------- planner + executor (oracle_plan.c) ---------------- */
typedef struct {
	int kind;                       /* BeamformerShaderKind */
	int in_kind, out_kind;
	int in_stride[3], out_stride[3];
	int filter_slot;
	int user_index;
} OracleStage;

typedef struct {
	int          stage_count;
	OracleStage  stages[BeamformerMaxComputeShaderStages];
	int          first_image_stage;
	int          iq_pipeline;
	int          chunk_channel_count;
	int          input_sample_count;     /* samples entering DAS */
	float        das_sampling_frequency;
	float        das_time_offset;
	uint32_t     rf_size;                /* bytes of one ping-pong slot */
	int          pipeline_data_kind;
	int          output_points[3];
	float        das_voxel_transform[16];
	int          das_sparse;
} OraclePlan;

typedef struct {
	BeamformerParameters        parameters;
	int32_t                     shaders[BeamformerMaxComputeShaderStages];
	uint8_t                     filter_slots[BeamformerMaxComputeShaderStages];
	uint32_t                    shader_count;
	int32_t                     data_kind;
	int16_t                     channel_mapping[BeamformerMaxChannelCount];
	int16_t                     sparse_elements[BeamformerMaxChannelCount];
	uint8_t                     transmit_receive_orientations[BeamformerMaxChannelCount];
	float                       focal_vectors[BeamformerMaxChannelCount][2];
	BeamformerFilterParameters  filters[BeamformerFilterSlots];
} OracleParameterBlock;

/* beamformer_core.c:553-1013 with cooperative_matrix = 0, subgroup 64 */
int  oracle_plan(const OracleParameterBlock *pb, OraclePlan *plan);
/* beamformer_core.c:1519-1626 (+ lib .c:491-570 ingest): one whole frame, 16-channel
 * chunks, ping-pong slots.  out: X*Y*Z float or float pair.  Returns 1 on success. */
int  oracle_beamform(const OracleParameterBlock *pb, const void *raw, float *out,
                     uint64_t *pairs_out, int threads);
/* the same frame restricted to z planes [z_first, z_first+z_count) and rows
 * [y_first, y_first+y_count) (0 counts = whole axis); out holds X * rows * planes voxels.
 * das_seconds (optional) receives the wall time spent inside oracle_das. */
/* Nearest-interpolation parity aid: while a buffer is set (one float per voxel of the computed (sub-)grid,
 * zeroed by the caller), oracle_das adds to each voxel the ambiguity budget of its sum: for every tap whose
 * sample index lies within 2^-10 of a rounding boundary (k + 1/2, or an end of the valid range), where an
 * implementation that differs by float rounding legitimately picks the other sample, |other - chosen|.
 * A voxel with a zero entry has no such tap.  NULL switches it off. */
void oracle_set_nearest_ambiguity_buffer(float *budget);
/* the double-precision twin of every DAS stage of later oracle_beamform* calls into `frame` (voxels x 1 or 2 doubles), or NULL: off */
void oracle_set_f64_frame(double *frame);
/* Sampling strides of the sub-grid for later oracle_beamform_subgrid calls (1, 1 = contiguous): plane
 * z_first + k z_stride, row y_first + k y_stride.  bench.py's CPU baseline times evenly spaced planes. */
void oracle_set_subgrid_stride(uint32_t z_stride, uint32_t y_stride);
int  oracle_beamform_subgrid(const OracleParameterBlock *pb, const void *raw, float *out, uint64_t *pairs_out,
                             int threads, uint32_t z_first, uint32_t z_count, uint32_t y_first, uint32_t y_count,
                             double *das_seconds);

#ifdef __cplusplus
}
#endif
#endif
