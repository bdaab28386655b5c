/* oracle_plan.c -- CPU ORACLE (test infrastructure): the reference's pipeline planner
 * (beamformer_core.c:553-1013, cooperative matrices absent, subgroup size 64) and its
 * per-frame executor (beamformer_core.c:1289-1400, :1519-1626): 16-channel chunks through
 * the pre-image stages, ping-pong slots 0/1 with slot 2 feeding DAS, DAS accumulating into
 * a cleared frame, then the post-image stages.  PARITY UNPINNED by the reference. */
#include "oracle.h"
#ifdef _OPENMP
#include <omp.h>
#else
static int omp_get_max_threads(void) { return 1; }
static int omp_get_num_threads(void) { return 1; }
static int omp_get_thread_num(void) { return 0; }
#endif
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double oracle_now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static const int kind_byte_size[6]     = {2, 4, 4, 8, 2, 4};
static const int kind_element_size[6]  = {2, 2, 4, 4, 2, 2};
static const int kind_complex[6]       = {0, 1, 0, 1, 0, 1};
#define KIND_DONT_CARE BeamformerDataKind_Count

typedef struct Node {
	int kind, user_index;
	int in_kind, out_kind;
	int in_stride[3], out_stride[3];
} Node;

static int stride_dont_care(const int *s) { return s[0] == 0 || s[1] == 0 || s[2] == 0; }
static int stride_equal(const int *a, const int *b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; }

static int g_hilbert_enabled;
void oracle_enable_hilbert(int enable) { g_hilbert_enabled = enable != 0; }

int oracle_plan(const OracleParameterBlock *pb, OraclePlan *plan)
{
	const BeamformerParameters *bp = &pb->parameters;
	memset(plan, 0, sizeof(*plan));

	int run_hilbert = 0, demodulate = 0;                                   /* :556-567 */
	for (uint32_t i = 0; i < pb->shader_count; i++) {
		if (pb->shaders[i] == BeamformerShaderKind_Hilbert)    run_hilbert = 1;
		if (pb->shaders[i] == BeamformerShaderKind_Demodulate) demodulate  = 1;
	}
	if (demodulate) run_hilbert = 0;
	if (run_hilbert && !g_hilbert_enabled) return 0;                       /* capabilities.hilbert = 0: the client refuses the stage */

	float fs = bp->sampling_frequency;
	int   input_sample_count = (int)bp->sample_count;
	int   A = (int)bp->acquisition_count;
	int   D = bp->decimation_rate > 1 ? (int)bp->decimation_rate : 1;

	int input_kind = pb->data_kind;                                        /* :577-587 */
	if (demodulate) {
		if (input_kind == BeamformerDataKind_Int16)   input_kind = BeamformerDataKind_Int16Complex;
		if (input_kind == BeamformerDataKind_Float16) input_kind = BeamformerDataKind_Float16Complex;
		if (input_kind == BeamformerDataKind_Float32) input_kind = BeamformerDataKind_Float32Complex;
		input_sample_count /= (2 * D);
		fs                 /= (float)(2 * D);
	}
	plan->iq_pipeline = kind_complex[input_kind] || run_hilbert;           /* :589 */
	int das_kind = plan->iq_pipeline ? BeamformerDataKind_Float32Complex : BeamformerDataKind_Float32;

	int C  = (int)bp->channel_count;
	int Cc = C < BeamformerChunkChannelCount ? C : BeamformerChunkChannelCount;   /* :595 */
	plan->chunk_channel_count = Cc;
	plan->rf_size = (uint32_t)input_sample_count * (uint32_t)A * (uint32_t)Cc * (uint32_t)kind_byte_size[das_kind];
	for (int i = 0; i < 3; i++) plan->output_points[i] = bp->output_points[i] > 1 ? bp->output_points[i] : 1;

	/* first pass (:609-683) */
	Node nodes[2 * BeamformerMaxComputeShaderStages + 4];
	int  count = 0;
	memset(nodes, 0, sizeof(nodes));
	Node *root = &nodes[count++];
	root->kind = -1; root->user_index = -1;
	root->in_kind = root->out_kind = input_kind;
	root->in_stride[0] = root->out_stride[0] = 1;
	root->in_stride[1] = root->out_stride[1] = (int)bp->sample_count * A;
	root->in_stride[2] = root->out_stride[2] = (int)bp->sample_count;

	for (uint32_t it = 0; it < pb->shader_count; it++) {
		int shader = pb->shaders[it];
		if (shader == BeamformerShaderKind_Hilbert && !run_hilbert) continue;
		if (shader == BeamformerShaderKind_Decode && bp->decode_mode == BeamformerDecodeMode_None) continue;
		if (shader == BeamformerShaderKind_Sum || shader == BeamformerShaderKind_MinMax) continue;

		Node *node = &nodes[count++];
		node->kind = shader; node->user_index = (int)it;
		node->in_kind = node->out_kind = KIND_DONT_CARE;
		switch (shader) {
		case BeamformerShaderKind_Decode:{                                 /* :645-664 */
			int low_precision = kind_element_size[input_kind] < 4;
			if (low_precision && kind_complex[input_kind]) node->in_kind = BeamformerDataKind_Float16Complex;
			node->in_stride[0] = Cc * A;
			node->in_stride[1] = A;
			node->in_stride[2] = 1;
		}break;
		case BeamformerShaderKind_DAS:{                                    /* :666-679 */
			node->in_kind = node->out_kind = das_kind;
			node->in_stride[0]  = 1;
			node->in_stride[1]  = input_sample_count * A;
			node->in_stride[2]  = input_sample_count;
			node->out_stride[0] = 1;
			node->out_stride[1] = plan->output_points[0];
			node->out_stride[2] = plan->output_points[0] * plan->output_points[1];
			if (bp->coherency_weighting) {
				Node *cw = &nodes[count++];
				cw->kind = BeamformerShaderKind_CoherencyWeighting; cw->user_index = -1;
				cw->in_kind = cw->out_kind = KIND_DONT_CARE;
			}
		}break;
		default: break;
		}
	}

	/* second pass (:685-735); Reshape nodes are spliced in front of the node they feed */
	Node resolved[2 * BeamformerMaxComputeShaderStages + 4];
	int  rcount = 0;
	resolved[rcount++] = nodes[0];
	for (int i = 1; i < count; i++) {
		Node *node = &nodes[i];
		Node *prev = &resolved[rcount - 1];
		int needs_reshape = 0;
		{
			int in_dc = stride_dont_care(node->in_stride), prev_dc = stride_dont_care(prev->out_stride);
			if (prev_dc && !in_dc) memcpy(prev->out_stride, node->in_stride, sizeof(node->in_stride));
			if (!prev_dc && in_dc) memcpy(node->in_stride, prev->out_stride, sizeof(node->in_stride));
			if (prev_dc && in_dc) {
				memcpy(prev->out_stride, prev->in_stride, sizeof(node->in_stride));
				memcpy(node->in_stride,  prev->in_stride, sizeof(node->in_stride));
			}
			needs_reshape |= !stride_equal(node->in_stride, prev->out_stride);
		}
		{
			int in_dc = node->in_kind == KIND_DONT_CARE, prev_dc = prev->out_kind == KIND_DONT_CARE;
			if (prev_dc && !in_dc) prev->out_kind = node->in_kind;
			if (!prev_dc && in_dc) node->in_kind  = prev->out_kind;
			if (prev_dc && in_dc)  node->in_kind  = prev->out_kind = prev->in_kind;
			needs_reshape |= node->in_kind != prev->out_kind;
		}
		if (needs_reshape) {
			Node r; memset(&r, 0, sizeof(r));
			r.kind = BeamformerShaderKind_Reshape; r.user_index = -1;
			r.in_kind = prev->out_kind;  memcpy(r.in_stride,  prev->out_stride, sizeof(r.in_stride));
			r.out_kind = node->in_kind;  memcpy(r.out_stride, node->in_stride,  sizeof(r.out_stride));
			resolved[rcount++] = r;
		}
		resolved[rcount++] = *node;
	}
	if (resolved[rcount - 1].out_kind == KIND_DONT_CARE)                   /* :737-739 */
		resolved[rcount - 1].out_kind = resolved[rcount - 1].in_kind;

	float time_offset = bp->time_offset;
	plan->first_image_stage = 0;
	for (int i = 1; i < rcount; i++) {
		if (plan->stage_count >= BeamformerMaxComputeShaderStages) break;    /* :523 */
		OracleStage *st = &plan->stages[plan->stage_count++];
		Node *n = &resolved[i];
		st->kind = n->kind; st->in_kind = n->in_kind; st->out_kind = n->out_kind;
		memcpy(st->in_stride,  n->in_stride,  sizeof(st->in_stride));
		memcpy(st->out_stride, n->out_stride, sizeof(st->out_stride));
		st->user_index  = n->user_index;
		st->filter_slot = n->user_index >= 0 ? pb->filter_slots[n->user_index] : 0;
		switch (n->kind) {
		case BeamformerShaderKind_Demodulate:
		case BeamformerShaderKind_Filter:{                                 /* :830-835 */
			float coeffs[8192], delay = 0;
			const BeamformerFilterParameters *fp = &pb->filters[st->filter_slot % BeamformerFilterSlots];
			if (oracle_filter_create(fp, coeffs, 8192, &delay) < 0) return 0;
			time_offset += delay;
		}break;
		case BeamformerShaderKind_Hilbert:{                                /* build-defined, oracle.h */
			if (kind_complex[st->in_kind]) return 0;                         /* needs real input */
			time_offset += (float)((ORACLE_HILBERT_LENGTH - 1) / 2) / fs;
		}break;
		case BeamformerShaderKind_DAS:{                                    /* :882, :906-917 */
			plan->first_image_stage = plan->stage_count - 1 + 1;
			memcpy(plan->das_voxel_transform, bp->das_voxel_transform, sizeof(plan->das_voxel_transform));
			if (bp->acquisition_kind == BeamformerAcquisitionKind_UFORCES ||
			    bp->acquisition_kind == BeamformerAcquisitionKind_FORCES)
				oracle_m4_mul(bp->xdc_transform, bp->das_voxel_transform, plan->das_voxel_transform);
			plan->das_sparse = bp->acquisition_kind == BeamformerAcquisitionKind_UFORCES ||
			                   bp->acquisition_kind == BeamformerAcquisitionKind_UHERCULES;
			plan->das_time_offset = time_offset;
		}break;
		default: break;
		}
	}
	/* the reference stores first_image_shader_index = index of the DAS stage + ... : it is
	 * set to shader_count at the time DAS is pushed (:882), i.e. the DAS stage's index + 1;
	 * stages [0, first_image) run per chunk -- DAS included (:1610-1613). */
	if (plan->first_image_stage == 0) plan->first_image_stage = plan->stage_count;   /* :1009-1010 */

	plan->input_sample_count     = input_sample_count;
	plan->das_sampling_frequency = fs;
	plan->pipeline_data_kind     = input_kind;
	return 1;
}

/* one stage on one chunk: beamformer_core.c:1289-1400 */
typedef struct {
	const OracleParameterBlock *pb;
	const OraclePlan *plan;
	uint8_t *ping_pong;       /* 3 slots of slot_bytes */
	size_t   slot_bytes;
	int      input_index;     /* cc->ping_pong_input_index */
	float   *frame, *incoherent;
	double  *frame64, *incoherent64;     /* oracle_set_f64_frame: the double twin's sums, or NULL */
	float   *hadamard, *readi_hadamard;
	uint64_t pairs;
	int      threads;
	int      chunk_channels;  /* channels in the chunk being run (Q5: the last chunk is clamped) */
	uint32_t z_first, z_count, y_first, y_count;   /* sub-grid (build extension), 0 counts = whole */
	double   das_seconds;
	/* rows-outermost schedule: the DAS stage of every chunk is recorded (parameters + a copy of its input) instead of run */
	int        defer_das;
	OracleDAS *deferred;          /* one per chunk */
	uint8_t   *deferred_inputs;   /* chunk k's DAS input at k * slot_bytes */
	int        deferred_count;
} Exec;

static int rows_outermost = 1;
void oracle_set_rows_outermost(int enable) { rows_outermost = enable != 0; }

/* Tolerance truth (test infrastructure): when set, every DAS stage of the next oracle_beamform* call ALSO runs in double precision
 * (oracle_das_f64: the same loops on the same float32 DAS input) into this buffer -- voxels x (1 or 2) doubles, cleared by the call --
 * and coherency weighting is applied to it in double.  What a parity test compares a GPU voxel with when the float oracle and the GPU
 * disagree by more than the bar: no float evaluation is asked to be closer to the truth than the oracle's own. */
static double *f64_frame;
void oracle_set_f64_frame(double *frame) { f64_frame = frame; }

static uint32_t subgrid_z_stride = 1, subgrid_y_stride = 1;   /* oracle_set_subgrid_stride */

static void run_stage(Exec *e, int slot, int channel_offset, const uint8_t *rf_pointer, int64_t rf_elements_left)
{
	const OraclePlan  *plan = e->plan;
	const OracleStage *st   = &plan->stages[slot];
	const BeamformerParameters *bp = &e->pb->parameters;
	/* Cc: the planned chunk (layout strides); Cn: channels actually present in this chunk */
	int A = (int)bp->acquisition_count, Cc = plan->chunk_channel_count, Cn = e->chunk_channels;
	int das_index = plan->first_image_stage - 1;                /* :1304 */
	int output_index = !e->input_index, das_output_index = 2;
	uint8_t *pp_in  = e->ping_pong + e->slot_bytes * (size_t)e->input_index;
	uint8_t *pp_out = e->ping_pong + e->slot_bytes * (size_t)output_index;
	uint8_t *pp_das = e->ping_pong + e->slot_bytes * (size_t)das_output_index;

	switch (st->kind) {
	case BeamformerShaderKind_Decode:{
		OracleDecode d = {0};
		d.transmit_count = A; d.chunk_channel_count = Cc; d.active_channels = Cn; d.sample_count = plan->input_sample_count;
		memcpy(d.out_stride, st->out_stride, sizeof(d.out_stride));
		d.in_kind = st->in_kind; d.out_kind = st->out_kind; d.hadamard = e->hadamard;
		oracle_decode(&d, pp_in, (slot + 1) == das_index ? pp_das : pp_out);
		e->input_index = !e->input_index;
	}break;
	case BeamformerShaderKind_Filter:
	case BeamformerShaderKind_Demodulate:{
		int demod = st->kind == BeamformerShaderKind_Demodulate;
		float coeffs[8192], delay;
		const BeamformerFilterParameters *fp = &e->pb->filters[st->filter_slot % BeamformerFilterSlots];
		OracleFilter f = {0};
		f.filter_length  = oracle_filter_create(fp, coeffs, 8192, &delay);
		f.complex_filter = fp->complex != 0 && fp->kind == BeamformerFilterKind_MatchedChirp;   /* Q7 */
		f.demodulate     = demod;
		f.coefficients   = coeffs;
		f.sample_count   = plan->input_sample_count;                                   /* :845 */
		f.decimation_rate = demod ? (bp->decimation_rate > 1 ? (int)bp->decimation_rate : 1) : 1;
		int deinterleave = kind_complex[st->in_kind] && !kind_complex[st->out_kind];  /* :848-851 */
		if (deinterleave) f.batch_sample_count = Cc * plan->input_sample_count * A;
		memcpy(f.in_stride,  st->in_stride,  sizeof(f.in_stride));
		memcpy(f.out_stride, st->out_stride, sizeof(f.out_stride));
		f.in_kind = st->in_kind; f.out_kind = st->out_kind;
		f.channels = Cn; f.transmits = A; f.workgroup = 64;
		if (demod) {                                                                    /* :870-873 */
			f.demodulation_frequency = bp->demodulation_frequency;
			f.sampling_frequency     = bp->sampling_frequency / 2;
		}
		const void *in = slot == 0 ? (const void *)rf_pointer : (const void *)pp_in;    /* :1337 */
		f.in_elements  = slot == 0 ? rf_elements_left
		                           : (int64_t)(e->slot_bytes / (size_t)kind_byte_size[st->in_kind]);
		/* output_element_offset selects the slot inside the one bound buffer (:1338-1342) */
		uint8_t *out = (slot + 1) == das_index ? pp_das : pp_out;
		oracle_filter(&f, in, out, 0);
		e->input_index = !e->input_index;
	}break;
	case BeamformerShaderKind_Hilbert:{
		float taps[2 * ORACLE_HILBERT_LENGTH];
		oracle_hilbert_fir(taps);
		OracleFilter f = {0};
		f.filter_length = ORACLE_HILBERT_LENGTH; f.complex_filter = 1; f.coefficients = taps;
		f.sample_count = plan->input_sample_count;
		memcpy(f.in_stride,  st->in_stride,  sizeof(f.in_stride));
		memcpy(f.out_stride, st->out_stride, sizeof(f.out_stride));
		f.in_kind = st->in_kind; f.out_kind = st->out_kind;
		f.channels = Cn; f.transmits = A;
		const void *in = slot == 0 ? (const void *)rf_pointer : (const void *)pp_in;
		oracle_hilbert(&f, in, (slot + 1) == das_index ? pp_das : pp_out);
		e->input_index = !e->input_index;
	}break;
	case BeamformerShaderKind_Reshape:{
		OracleReshape r = {0};
		r.size[0] = plan->input_sample_count; r.size[1] = Cn; r.size[2] = A;              /* :975-977 */
		memcpy(r.in_stride,  st->in_stride,  sizeof(r.in_stride));
		memcpy(r.out_stride, st->out_stride, sizeof(r.out_stride));
		r.in_kind = st->in_kind; r.out_kind = st->out_kind;
		r.interleave = !kind_complex[st->in_kind] && kind_complex[st->out_kind];        /* :961-965 */
		const uint8_t *left  = slot == 0 ? rf_pointer : pp_in;                          /* :1381-1386 */
		const uint8_t *right = left + (size_t)r.size[0] * Cc * r.size[2] * (size_t)kind_byte_size[st->in_kind];
		oracle_reshape(&r, left, right, (slot + 1) == das_index ? pp_das : pp_out);
		e->input_index = !e->input_index;
	}break;
	case BeamformerShaderKind_DAS:{
		OracleDAS d; memset(&d, 0, sizeof(d));
		d.acquisition_kind = bp->acquisition_kind; d.sparse = plan->das_sparse;
		d.acquisition_count = A; d.channel_count = (int)bp->channel_count; d.chunk_channel_count = Cn;
		d.sample_count = plan->input_sample_count;
		d.sampling_frequency = plan->das_sampling_frequency;
		d.demodulation_frequency = bp->demodulation_frequency;
		d.speed_of_sound = bp->speed_of_sound; d.time_offset = plan->das_time_offset;
		d.interpolation_mode = bp->interpolation_mode; d.f_number = bp->f_number;
		d.single_orientation = (int)bp->single_orientation;
		d.transmit_receive_orientation = bp->transmit_receive_orientation;
		d.single_focus = (int)bp->single_focus;
		d.transmit_angle = bp->focal_vector[0]; d.focus_depth = bp->focal_vector[1];
		for (int i = 0; i < 3; i++) d.output_size[i] = (uint32_t)plan->output_points[i];
		d.readi_group_count = bp->readi_group_count; d.readi_group = bp->readi_group;
		d.coherency_weighting = bp->coherency_weighting != 0;
		d.complex_data = st->in_kind == BeamformerDataKind_Float32Complex;
		memcpy(d.xdc_transform,   bp->xdc_transform,         sizeof(d.xdc_transform));
		memcpy(d.voxel_transform, plan->das_voxel_transform, sizeof(d.voxel_transform));
		memcpy(d.xdc_element_pitch, bp->xdc_element_pitch,   sizeof(d.xdc_element_pitch));
		d.rf_element_offset = 0;            /* pp_das is passed as the base instead of an offset */
		d.channel_offset = channel_offset;
		d.focal_vectors = &e->pb->focal_vectors[0][0];
		d.sparse_elements = e->pb->sparse_elements;
		d.transmit_receive_orientations = e->pb->transmit_receive_orientations;
		d.readi_hadamard = e->readi_hadamard;
		d.threads = e->threads;
		d.z_first = e->z_first; d.z_count = e->z_count; d.y_first = e->y_first; d.y_count = e->y_count;
		d.z_stride = subgrid_z_stride; d.y_stride = subgrid_y_stride;
		if (e->defer_das) {
			memcpy(e->deferred_inputs + (size_t)e->deferred_count * e->slot_bytes, pp_das, e->slot_bytes);
			e->deferred[e->deferred_count++] = d;
			break;
		}
		double t0 = oracle_now();
		e->pairs += oracle_das(&d, (const float *)pp_das, e->frame, e->incoherent);
		e->das_seconds += oracle_now() - t0;
		if (e->frame64) oracle_das_f64(&d, (const float *)pp_das, e->frame64, e->incoherent64);
	}break;
	case BeamformerShaderKind_CoherencyWeighting:{
		uint32_t voxels = (uint32_t)plan->output_points[0] * (e->y_count ? e->y_count : (uint32_t)plan->output_points[1])
		                  * (e->z_count ? e->z_count : (uint32_t)plan->output_points[2]);
		oracle_coherency_weighting(e->frame, e->incoherent, voxels, plan->iq_pipeline, 1.0f);  /* :949 */
		if (e->frame64) {
			int n = plan->iq_pipeline ? 2 : 1;
			for (uint32_t i = 0; i < voxels; i++)
				for (int c = 0; c < n; c++) {
					double v = e->frame64[n * (uint64_t)i + c];
					e->frame64[n * (uint64_t)i + c] = v * (v / e->incoherent64[i]);
				}
		}
	}break;
	default: break;
	}
}

int oracle_beamform(const OracleParameterBlock *pb, const void *raw, float *out, uint64_t *pairs_out, int threads)
{
	return oracle_beamform_subgrid(pb, raw, out, pairs_out, threads, 0, 0, 0, 0, 0);
}

void oracle_set_subgrid_stride(uint32_t z_stride, uint32_t y_stride)
{
	subgrid_z_stride = z_stride ? z_stride : 1;
	subgrid_y_stride = y_stride ? y_stride : 1;
}

int oracle_beamform_subgrid(const OracleParameterBlock *pb, const void *raw, float *out, uint64_t *pairs_out,
                            int threads, uint32_t z_first, uint32_t z_count, uint32_t y_first, uint32_t y_count,
                            double *das_seconds)
{
	const BeamformerParameters *bp = &pb->parameters;
	OraclePlan plan;
	if (!oracle_plan(pb, &plan)) return 0;
	int das_stage = -1;
	for (int i = 0; i < plan.stage_count; i++) if (plan.stages[i].kind == BeamformerShaderKind_DAS) das_stage = i;
	if (das_stage < 0) return 0;
	/* Q6 (beamformer_core.c:1353-1362 vs :1337): a plan whose first stage is DAS reads a
	 * ping-pong slot nothing wrote, so the reference cannot run Float32(Complex) data with
	 * decoding off.  Build-defined: DAS consumes the ingested RF directly. */
	size_t bytes     = (size_t)kind_byte_size[pb->data_kind];
	size_t rf_bytes  = bytes * bp->sample_count * bp->acquisition_count * bp->channel_count;
	uint8_t *mapped  = (uint8_t *)calloc(1, rf_bytes + 64);
	oracle_channel_map(raw, mapped, bp, pb->data_kind, pb->channel_mapping);

	uint64_t voxels = (uint64_t)plan.output_points[0] * (y_count ? y_count : (uint32_t)plan.output_points[1])
	                  * (z_count ? z_count : (uint32_t)plan.output_points[2]);
	int elements = plan.iq_pipeline ? 2 : 1;

	Exec e; memset(&e, 0, sizeof(e));
	e.pb = pb; e.plan = &plan; e.threads = threads;
	e.z_first = z_first; e.z_count = z_count; e.y_first = y_first; e.y_count = y_count;
	/* ping-pong: 3 x round_up(rf_size, 64) (:1229); intermediate kinds never exceed the DAS kind */
	e.slot_bytes = ((size_t)plan.rf_size + 63) & ~(size_t)63;
	{	/* head-room so that a wider-than-DAS intermediate cannot overrun in the checker */
		size_t wide = (size_t)bp->sample_count * bp->acquisition_count * (size_t)plan.chunk_channel_count * 8;
		if (wide > e.slot_bytes) e.slot_bytes = wide;
	}
	e.ping_pong  = (uint8_t *)calloc(3, e.slot_bytes + 64);
	e.frame      = out;
	memset(out, 0, sizeof(float) * elements * voxels);                         /* :1573-1578 */
	e.incoherent = (float *)calloc(voxels + 16, sizeof(float));                /* :1580-1585 */
	if (f64_frame) {
		e.frame64 = f64_frame;
		memset(e.frame64, 0, sizeof(double) * elements * voxels);
		e.incoherent64 = (double *)calloc(voxels + 16, sizeof(double));
	}
	int A = (int)bp->acquisition_count;
	e.hadamard = (float *)calloc((size_t)A * A + 1, sizeof(float));
	if (bp->decode_mode == BeamformerDecodeMode_Hadamard) oracle_hadamard_transpose(A, e.hadamard);
	if (bp->readi_group_count > 1) {
		int G = (int)bp->readi_group_count;
		e.readi_hadamard = (float *)calloc((size_t)G * G, sizeof(float));
		oracle_hadamard_transpose(G, e.readi_hadamard);
	}

	uint32_t chunk_total = (bp->channel_count + BeamformerChunkChannelCount - 1) / BeamformerChunkChannelCount;
	if (rows_outermost && das_stage < plan.first_image_stage) {
		e.deferred        = (OracleDAS *)calloc(chunk_total, sizeof(OracleDAS));
		e.deferred_inputs = (uint8_t *)malloc((size_t)chunk_total * e.slot_bytes);
		e.defer_das       = e.deferred && e.deferred_inputs;
	}

	size_t raw_channel_byte_stride = bytes * bp->sample_count * bp->acquisition_count;   /* :574 */
	for (uint32_t channel_offset = 0; channel_offset < bp->channel_count;
	     channel_offset += BeamformerChunkChannelCount)                                   /* :1604-1614 */
	{
		const uint8_t *rf_pointer = mapped + raw_channel_byte_stride * channel_offset;
		int64_t left = (int64_t)((rf_bytes - raw_channel_byte_stride * channel_offset) /
		                         (size_t)kind_byte_size[plan.stages[0].in_kind]);
		e.chunk_channels = (int)bp->channel_count - (int)channel_offset;
		if (e.chunk_channels > plan.chunk_channel_count) e.chunk_channels = plan.chunk_channel_count;   /* Q5 */
		if (das_stage == 0)
			memcpy(e.ping_pong + 2 * e.slot_bytes, rf_pointer, raw_channel_byte_stride * (size_t)e.chunk_channels);
		for (int i = 0; i < plan.first_image_stage; i++)
			run_stage(&e, i, (int)channel_offset, rf_pointer, left);
	}
	if (e.defer_das && e.deferred_count) {
		/* every thread takes rows of the (sub-)grid and, per row, walks the chunks in the order the reference runs them: each voxel
		 * receives exactly the additions of the chunk-by-chunk schedule, in the same order */
		uint32_t zn = z_count ? z_count : (uint32_t)plan.output_points[2], yn = y_count ? y_count : (uint32_t)plan.output_points[1];
		int64_t  rows = (int64_t)zn * yn;
		uint64_t pairs = 0;
		double   t0 = oracle_now();
		/* one parallel region; thread t owns rows t, t + T, t + 2 T, ... for the whole frame and walks the chunks OUTSIDE its rows, so
		 * that the threads -- never waiting for one another -- stay on the same 16-channel chunk most of the time and share it in
		 * cache, as the reference's chunk loop does (a row-outermost loop streams the whole DAS input per row: measured slower on a
		 * 256-thread host than 16 threads of the old schedule) */
		#pragma omp parallel reduction(+:pairs) num_threads(threads > 0 ? threads : omp_get_max_threads())
		{
			const int64_t T = omp_get_num_threads(), t = omp_get_thread_num();
			for (int k = 0; k < e.deferred_count; k++) {
				OracleDAS d = e.deferred[k];
				d.threads = 1; d.row_count = 1;
				const float *input = (const float *)(e.deferred_inputs + (size_t)k * e.slot_bytes);
				for (int64_t row = t; row < rows; row += T) {
					d.row_first = row;
					pairs += oracle_das(&d, input, e.frame, e.incoherent);
					if (e.frame64) oracle_das_f64(&d, input, e.frame64, e.incoherent64);
				}
			}
		}
		e.pairs += pairs;
		e.das_seconds += oracle_now() - t0;
	}
	free(e.deferred); free(e.deferred_inputs);
	for (int i = plan.first_image_stage; i < plan.stage_count; i++)                       /* :1616-1619 */
		run_stage(&e, i, 0, 0, 0);

	if (pairs_out) *pairs_out = e.pairs;
	if (das_seconds) *das_seconds = e.das_seconds;
	free(mapped); free(e.ping_pong); free(e.incoherent); free(e.incoherent64); free(e.hadamard); free(e.readi_hadamard);
	return 1;
}
