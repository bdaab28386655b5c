/* oracle_stages.c -- CPU ORACLE (test infrastructure): ingest, Reshape, Decode,
 * Filter/Demodulate, CoherencyWeighting, min/max.  Literal restatements of
 * lib/ogl_beamformer_lib.c:491-570 and shaders/{reshape,decode,filter,
 * coherency_weighting}.glsl.  PARITY UNPINNED by the reference (see oracle.h). */
#include "oracle.h"
#include "oracle_f16.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* generated/beamformer.c:489-521 */
static const int kind_byte_size[6]     = {2, 4, 4, 8, 2, 4};
static const int kind_element_count[6] = {1, 2, 1, 2, 1, 2};
enum { BASE_I16, BASE_F32, BASE_F16 };
static const int kind_base[6] = {BASE_I16, BASE_I16, BASE_F32, BASE_F32, BASE_F16, BASE_F16};

static inline void load_element(int kind, const void *buffer, int64_t index, float v[2])
{
	int n = kind_element_count[kind];
	v[0] = v[1] = 0;
	for (int c = 0; c < n; c++) {
		switch (kind_base[kind]) {
		case BASE_I16: v[c] = (float)((const int16_t *)buffer)[n * index + c]; break;
		case BASE_F32: v[c] = ((const float *)buffer)[n * index + c]; break;
		case BASE_F16: v[c] = oracle_f16_bits_to_f32(((const uint16_t *)buffer)[n * index + c]); break;
		}
	}
}

/* GLSL constructor conversion OutputDataType(value) */
static inline void store_scalar(int kind, void *buffer, int64_t scalar_index, float v)
{
	switch (kind_base[kind]) {
	case BASE_I16: ((int16_t *)buffer)[scalar_index]  = (int16_t)v; break;   /* truncates toward zero */
	case BASE_F32: ((float *)buffer)[scalar_index]    = v; break;
	case BASE_F16: ((uint16_t *)buffer)[scalar_index] = oracle_f32_to_f16_bits(v); break;
	}
}

static inline void store_element(int kind, void *buffer, int64_t index, const float v[2])
{
	int n = kind_element_count[kind];
	for (int c = 0; c < n; c++) store_scalar(kind, buffer, n * index + c, v[c]);
}

/* lib/ogl_beamformer_lib.c:515-559 */
void oracle_channel_map(const void *raw, void *dst, const BeamformerParameters *bp, int data_kind,
                        const int16_t *channel_mapping)
{
	size_t bytes      = (size_t)kind_byte_size[data_kind];
	size_t out_stride = bytes * bp->sample_count * bp->acquisition_count;
	size_t in_stride  = bytes * bp->raw_data_dimensions[0];
	for (uint32_t channel = 0; channel < bp->channel_count; channel++) {
		uint16_t       data_channel = (uint16_t)channel_mapping[channel];
		uint8_t       *out = (uint8_t *)dst + out_stride * channel;
		const uint8_t *in  = (const uint8_t *)raw + in_stride * data_channel;
		if (bp->contrast_mode != BeamformerContrastMode_A1S2) {
			memcpy(out, in, out_stride);
			continue;
		}
		/* lib .c:466-489, :531-557: only the first sample_count*element_count scalars of the
		 * row are produced, the rest of the (cleared) row stays zero */
		memset(out, 0, out_stride);
		uint32_t n = bp->sample_count * (uint32_t)kind_element_count[data_kind];
		for (uint32_t s = 0; s < n; s++) {
			switch (kind_base[data_kind]) {
			case BASE_I16:{
				const int16_t *a = (const int16_t *)in;
				((int16_t *)out)[s] = (int16_t)(a[s] - a[n + s] - a[2 * n + s]);
			}break;
			case BASE_F32:{
				const float *a = (const float *)in;
				((float *)out)[s] = a[s] - a[n + s] - a[2 * n + s];
			}break;
			case BASE_F16:{
				/* _Float16 arithmetic: each subtraction rounds to binary16 */
				const uint16_t *a = (const uint16_t *)in;
				float t = oracle_round_f16(oracle_f16_bits_to_f32(a[s]) - oracle_f16_bits_to_f32(a[n + s]));
				t = oracle_round_f16(t - oracle_f16_bits_to_f32(a[2 * n + s]));
				((uint16_t *)out)[s] = oracle_f32_to_f16_bits(t);
			}break;
			}
		}
	}
}

/* reshape.glsl:61-82.  Q2: Float16 input is IEEE half (the reference reads it through an
 * int16 view, reshape.glsl:7-10). */
void oracle_reshape(const OracleReshape *r, const void *left, const void *right, void *out)
{
	int out_n = kind_element_count[r->out_kind];
	for (int z = 0; z < r->size[2]; z++) {
		for (int y = 0; y < r->size[1]; y++) {
			for (int x = 0; x < r->size[0]; x++) {
				int64_t in_index  = (int64_t)r->in_stride[0]  * x + (int64_t)r->in_stride[1]  * y + (int64_t)r->in_stride[2]  * z;
				int64_t out_index = (int64_t)r->out_stride[0] * x + (int64_t)r->out_stride[1] * y + (int64_t)r->out_stride[2] * z;
				float v[2] = {0, 0}, t[2];
				if (r->interleave) {
					load_element(r->in_kind, left,  in_index, t); v[0] = t[0];
					load_element(r->in_kind, right, in_index, t); v[1] = t[0];
				} else {
					load_element(r->in_kind, left, in_index, v);
				}
				if (kind_base[r->in_kind] == BASE_I16 && kind_base[r->out_kind] == BASE_I16) {
					/* integer to integer: no float round trip */
					int in_n = kind_element_count[r->in_kind];
					for (int c = 0; c < out_n; c++) {
						int16_t s = 0;
						if (r->interleave) s = ((const int16_t *)(c ? right : left))[in_n * in_index];
						else if (c < in_n) s = ((const int16_t *)left)[in_n * in_index + c];
						((int16_t *)out)[out_n * out_index + c] = s;
					}
				} else {
					store_element(r->out_kind, out, out_index, v);
				}
			}
		}
	}
}

/* decode.glsl:24-73 (LDS variant) and :119-150 (register variant): for every
 * (sample, channel): out[i] = (sum_j in[j] * Ht[T*j + i]) / T, f32 accumulation,
 * j ascending.  Input is [sample][chunk_channel][transmit] (beamformer_core.c:652-657).
 * Q4: the shaders bound the sample by OutputTransmitStride (decode.glsl:46, :125),
 * which equals the sample count only for DAS-layout output; the sample count is used. */
void oracle_decode(const OracleDecode *d, const void *in, void *out)
{
	int T = d->transmit_count, Cc = d->chunk_channel_count;
	int Cn = d->active_channels > 0 ? d->active_channels : Cc;
	int n = kind_element_count[d->out_kind];
	float *row = (float *)malloc(sizeof(float) * 2 * (size_t)T);
	int out_base_f16 = kind_base[d->out_kind] == BASE_F16;
	for (int sample = 0; sample < d->sample_count; sample++) {
		for (int channel = 0; channel < Cn; channel++) {
			int64_t rf_offset = (int64_t)T * Cc * sample + (int64_t)T * channel;
			for (int j = 0; j < T; j++) load_element(d->in_kind, in, rf_offset + j, row + 2 * j);
			for (int i = 0; i < T; i++) {
				/* OutputDataType result: the running sum lives in the OUTPUT type
				 * (decode.glsl:46-56, :131-141): f32, or f16 with a rounding per operation */
				float acc[2] = {0, 0};
				for (int j = 0; j < T; j++) {
					float h = d->hadamard[T * j + i];
					float s0 = row[2 * j + 0], s1 = row[2 * j + 1];
					if (out_base_f16) { s0 = oracle_round_f16(s0); s1 = oracle_round_f16(s1); }
					acc[0] += s0 * h;
					acc[1] += s1 * h;
					if (out_base_f16) { acc[0] = oracle_round_f16(acc[0]); acc[1] = oracle_round_f16(acc[1]); }
				}
				acc[0] /= (float)T;
				acc[1] /= (float)T;
				int64_t out_off = (int64_t)d->out_stride[1] * channel + (int64_t)d->out_stride[2] * i
				                  + (int64_t)d->out_stride[0] * sample;
				if (n == 1) store_scalar(d->out_kind, out, out_off, acc[0]);
				else        store_element(d->out_kind, out, out_off, acc);
			}
		}
	}
	free(row);
}

/* filter.glsl:2-14 */
static int filter_sample_is_f16(int in_kind) { return kind_base[in_kind] != BASE_F32; }

/* filter.glsl:68-135, one invocation grid = (ceil(S'/64), channels, transmits) */
void oracle_filter(const OracleFilter *f, const void *in, void *out, uint32_t output_element_offset)
{
	int L = f->filter_length, D = f->decimation_rate, W = f->workgroup;
	int total_samples = D * W + L - 1;
	int in_n     = kind_element_count[f->in_kind];
	int complex_sample = in_n == 2 || f->demodulate;      /* ComplexSampleType, filter.glsl:16-19 */
	int f16_lds  = filter_sample_is_f16(f->in_kind);
	int out_n    = kind_element_count[f->out_kind];
	int groups   = (f->sample_count + W - 1) / W;
	float *lds   = (float *)malloc(sizeof(float) * 2 * (size_t)total_samples);
	/* scale = SAMPLE_TYPE(ComplexFilter ? 1 : sqrt(2)) (filter.glsl:98) */
	float scale = f->complex_filter ? 1.0f : sqrtf(2.0f);
	if (f16_lds) scale = oracle_round_f16(scale);

	for (int transmit = 0; transmit < f->transmits; transmit++) {
	for (int channel = 0; channel < f->channels; channel++) {
	for (int wg = 0; wg < groups; wg++) {
		int offset_wraps = (D * wg * W) < (L - 1);                                    /* :79 */
		/* element offset of the row; with Demodulate the strides are in real samples and the
		 * byte offset is halved (filter.glsl:81-87): row_start counts InputDataType elements */
		int64_t row_start = (int64_t)f->in_stride[1] * channel + (int64_t)f->in_stride[2] * transmit;
		if (f->demodulate) row_start /= 2;
		int64_t window_start = row_start + (int64_t)D * wg * W - (L - 1);            /* :90-92 */

		for (int index = 0; index < total_samples; index++) {                        /* :99-111 */
			float s[2] = {0, 0};
			if ((!offset_wraps || index >= L - 1) && window_start + index >= 0 &&
			    window_start + index < f->in_elements) {
				load_element(f->in_kind, in, window_start + index, s);
				if (f16_lds) { s[0] = oracle_round_f16(s[0]); s[1] = oracle_round_f16(s[1]); }
				if (f->demodulate) {
					/* s * (1,-1), rotate_iq(s, index) (filter.glsl:57-64), then * scale in SAMPLE_TYPE */
					float a[2] = {s[0], -s[1]};
					float arg  = 6.28318530717958647692f * f->demodulation_frequency * (float)index
					             / f->sampling_frequency;
					float b[2] = {cosf(arg), -sinf(arg)};
					float r[2] = {b[0] * a[0] - b[1] * a[1], b[1] * a[0] + b[0] * a[1]};
					if (f16_lds) {
						r[0] = oracle_round_f16(r[0]); r[1] = oracle_round_f16(r[1]);
						s[0] = oracle_round_f16(scale * r[0]); s[1] = oracle_round_f16(scale * r[1]);
					} else {
						s[0] = scale * r[0]; s[1] = scale * r[1];
					}
				}
			}
			lds[2 * index] = s[0]; lds[2 * index + 1] = s[1];
		}

		for (int t = 0; t < W; t++) {
			int out_sample = wg * W + t;
			if (out_sample >= f->sample_count / D) continue;                            /* :115 */
			float result[2] = {0, 0};
			int   offset = D * t;
			for (int j = 0; j < L; j++) {
				const float *iq = lds + 2 * (offset + j);
				if (f->complex_filter && complex_sample) {
					const float *h = f->coefficients + 2 * j;
					result[0] += h[0] * iq[0] - h[1] * iq[1];
					result[1] += h[1] * iq[0] + h[0] * iq[1];
				} else {
					float h = f->complex_filter ? f->coefficients[2 * j] : f->coefficients[j];
					result[0] += iq[0] * h;
					result[1] += iq[1] * h;
				}
			}
			int64_t out_offset = (int64_t)f->out_stride[1] * channel + (int64_t)f->out_stride[2] * transmit
			                     + (int64_t)f->out_stride[0] * out_sample + output_element_offset;
			if (f->batch_sample_count != 0) {                                          /* :126-130 */
				store_scalar(f->out_kind, out, out_offset, result[0]);
				store_scalar(f->out_kind, out, out_offset + f->batch_sample_count, result[1]);
			} else if (out_n == 2) {
				store_element(f->out_kind, out, out_offset, result);
			} else {
				store_scalar(f->out_kind, out, out_offset, result[0]);
			}
		}
	}}}
	free(lds);
}

/* coherency_weighting.glsl:28-37: c *= Scale * c / inc, component-wise for complex */
void oracle_coherency_weighting(float *coherent, const float *incoherent, uint32_t voxels,
                                int complex_data, float scale)
{
	int n = complex_data ? 2 : 1;
	for (uint32_t i = 0; i < voxels; i++)
		for (int c = 0; c < n; c++) {
			float v = coherent[n * (uint64_t)i + c];
			coherent[n * (uint64_t)i + c] = v * (scale * v / incoherent[i]);
		}
}

/* oracle.h: y[n] = sum_j h[j] x[n - (L-1) + j], real x (the real component of whatever kind
 * arrives), complex f32-accumulated y stored in the output kind; samples outside the row are 0 */
void oracle_hilbert(const OracleFilter *f, const void *in, void *out)
{
	const int L = f->filter_length;
	for (int t = 0; t < f->transmits; t++)
	for (int c = 0; c < f->channels; c++)
	for (int n = 0; n < f->sample_count; n++) {
		float re = 0.f, im = 0.f;
		for (int j = 0; j < L; j++) {
			int s = n - (L - 1) + j;
			if (s < 0 || s >= f->sample_count) continue;
			float x[2];
			load_element(f->in_kind, in, (int64_t)f->in_stride[0] * s + (int64_t)f->in_stride[1] * c + (int64_t)f->in_stride[2] * t, x);
			re += f->coefficients[2 * j]     * x[0];
			im += f->coefficients[2 * j + 1] * x[0];
		}
		float y[2] = {re, im};
		store_element(f->out_kind, out, (int64_t)f->out_stride[0] * n + (int64_t)f->out_stride[1] * c + (int64_t)f->out_stride[2] * t, y);
	}
}

void oracle_sum(float *out, const float *in, float prescale, uint64_t floats)
{
	for (uint64_t i = 0; i < floats; i++) out[i] = out[i] + prescale * in[i];
}

void oracle_display(const float *frame, uint64_t voxels, int complex_data, float threshold_db, float gamma,
                    float db_cutoff, float *out)
{
	float threshold_val = powf(10.0f, threshold_db / 20.0f);
	for (uint64_t i = 0; i < voxels; i++) {
		float result = complex_data ? sqrtf(frame[2 * i] * frame[2 * i] + frame[2 * i + 1] * frame[2 * i + 1])
		                            : fabsf(frame[i]);
		result = fminf(fmaxf(result, 0.0f), threshold_val);
		result = result / threshold_val;
		result = powf(result, gamma);
		if (db_cutoff > 0) {
			result = 20 * logf(result) / logf(10);
			result = fminf(fmaxf(result, -db_cutoff), 0) / -db_cutoff;
			result = 1 - result;
		}
		out[i] = result;
	}
}

void oracle_min_max(const float *frame, uint64_t voxels, int complex_data, float *out2)
{
	float lo = INFINITY, hi = -INFINITY;
	for (uint64_t i = 0; i < voxels; i++) {
		float v = complex_data ? sqrtf(frame[2 * i] * frame[2 * i] + frame[2 * i + 1] * frame[2 * i + 1])
		                       : frame[i];
		if (v < lo) lo = v;
		if (v > hi) hi = v;
	}
	out2[0] = lo; out2[1] = hi;
}
