/* oracle_das_body.h -- CPU ORACLE (test infrastructure).  Included twice by oracle_das.c:
 * once with REAL = float (the restatement of shaders/das.glsl as the GPU runs it) and once
 * with REAL = double (a truth twin used to budget tolerances).  Statement order follows
 * the shader so that the float build performs the same operations in the same order.
 *
 * Macros supplied by the includer: REAL, FN(name), R_SQRT, R_SIN, R_COS, R_FLOOR, R_FABS,
 * R_ROUND, R_MODF, R_ISINF, R_PI.
 */

typedef struct { REAL x, y; } FN(c2);

/* das.glsl:54-61; Q3: the rotation angle is reduced to [0,1) turns before sin/cos */
static inline FN(c2) FN(rotate_iq)(const OracleDAS *p, FN(c2) iq, REAL time)
{
	if (!p->complex_data) return iq;
	REAL turns = (REAL)p->demodulation_frequency * time;
	turns     -= R_FLOOR(turns);
	REAL arg   = (REAL)2 * R_PI * turns;
	REAL c = R_COS(arg), s = R_SIN(arg);
	FN(c2) r = {c * iq.x - s * iq.y, s * iq.x + c * iq.y};
	return r;
}

static inline FN(c2) FN(load)(const OracleDAS *p, const float *rf, int index)
{
	FN(c2) r;
	if (p->complex_data) { r.x = rf[2 * (int64_t)index]; r.y = rf[2 * (int64_t)index + 1]; }
	else                 { r.x = rf[index];              r.y = 0; }
	return r;
}

/* das.glsl:66-97 */
static inline FN(c2) FN(cubic)(const OracleDAS *p, const float *rf, int offset, REAL t)
{
	FN(c2) s0 = FN(load)(p, rf, offset + 0), s1 = FN(load)(p, rf, offset + 1);
	FN(c2) s2 = FN(load)(p, rf, offset + 2), s3 = FN(load)(p, rf, offset + 3);
	REAL S[4] = {t * t * t, t * t, t, 1};
	FN(c2) P1 = s1, P2 = s2;
	FN(c2) T1 = {(REAL)0.5 * (P2.x - s0.x), (REAL)0.5 * (P2.y - s0.y)};
	FN(c2) T2 = {(REAL)0.5 * (s3.x - P1.x), (REAL)0.5 * (s3.y - P1.y)};
	FN(c2) r;
	if (!p->complex_data) {
		/* dot(S, h * C), C = (P1, P2, T1, T2); h columns (2,-3,0,1) (-2,3,0,0) (1,-2,1,0) (1,-1,0,0) */
		REAL hc0 =  2 * P1.x - 2 * P2.x + 1 * T1.x + 1 * T2.x;
		REAL hc1 = -3 * P1.x + 3 * P2.x - 2 * T1.x - 1 * T2.x;
		REAL hc2 =                            T1.x;
		REAL hc3 =      P1.x;
		r.x = S[0] * hc0 + S[1] * hc1 + S[2] * hc2 + S[3] * hc3;
		r.y = 0;
	} else {
		/* (S * h) * C: row vector times h first */
		REAL b0 =  2 * S[0] - 3 * S[1] + S[3];
		REAL b1 = -2 * S[0] + 3 * S[1];
		REAL b2 =      S[0] - 2 * S[1] + S[2];
		REAL b3 =      S[0] -     S[1];
		r.x = b0 * P1.x + b1 * P2.x + b2 * T1.x + b3 * T2.x;
		r.y = b0 * P1.y + b1 * P2.y + b2 * T1.y + b3 * T2.y;
	}
	return r;
}

/* das.glsl:99-124 */
static inline FN(c2) FN(sample_rf)(const OracleDAS *p, const float *rf, int rf_offset, REAL index)
{
	FN(c2) result = {0, 0};
	REAL fs = (REAL)p->sampling_frequency;
	switch (p->interpolation_mode) {
	case BeamformerInterpolationMode_Nearest:{
		if (index >= 0 && index < ((REAL)p->sample_count - (REAL)0.5))
			result = FN(rotate_iq)(p, FN(load)(p, rf, rf_offset + (int)R_ROUND(index)), index / fs);
		/* test infrastructure, not part of the shader: a tap within 2^-10 of a rounding boundary adds the
		 * size of the possible flip to the voxel's ambiguity budget (oracle_das.c) */
		if (oracle_near_half_buffer) {
			REAL f = index - R_FLOOR(index), S = (REAL)p->sample_count;
			int  k = (int)R_ROUND(index), other = -1, valid = index >= 0 && index < S - (REAL)0.5;
			int  boundary = 0;
			if (R_FABS(f - (REAL)0.5) < (REAL)ORACLE_NEAR_HALF) { boundary = 1; other = f < (REAL)0.5 ? k + 1 : k - 1; }
			if (R_FABS(index) < (REAL)ORACLE_NEAR_HALF || R_FABS(index - (S - (REAL)0.5)) < (REAL)ORACLE_NEAR_HALF) boundary = 1;
			if (boundary) {
				FN(c2) a = {0, 0}, b = {0, 0};
				if (valid) a = FN(load)(p, rf, rf_offset + k);
				if (other >= 0 && other < p->sample_count) b = FN(load)(p, rf, rf_offset + other);
				REAL dx = a.x - b.x, dy = a.y - b.y;
				oracle_near_half_budget += (float)R_SQRT(dx * dx + dy * dy) + 1e-30f;
			}
		}
	}break;
	case BeamformerInterpolationMode_Linear:{
		if (index >= 0 && index < (REAL)(p->sample_count - 1)) {
			REAL tk, t = R_MODF(index, &tk);
			int  n = rf_offset + (int)tk;
			FN(c2) a = FN(load)(p, rf, n), b = FN(load)(p, rf, n + 1);
			result.x = (1 - t) * a.x + t * b.x;
			result.y = (1 - t) * a.y + t * b.y;
			result   = FN(rotate_iq)(p, result, index / fs);
		}
	}break;
	case BeamformerInterpolationMode_Cubic:{
		if (index >= 1 && index < (REAL)(p->sample_count - 2)) {
			REAL tk, t = R_MODF(index, &tk);
			result = FN(rotate_iq)(p, FN(cubic)(p, rf, rf_offset + (int)index, t), index / fs);
		}
	}break;
	}
	return result;
}

/* das.glsl:126-130 */
static inline REAL FN(sample_index)(const OracleDAS *p, REAL distance)
{
	REAL time = distance / (REAL)p->speed_of_sound + (REAL)p->time_offset;
	return time * (REAL)p->sampling_frequency;
}

/* das.glsl:138-152 */
static inline REAL FN(apodize)(REAL arg)
{
	REAL a = R_COS(R_PI * arg);
	return a * a;
}

/* das.glsl:172-185 */
static inline uint32_t FN(orientation_for)(const OracleDAS *p, int acquisition)
{
	uint32_t r = p->transmit_receive_orientation & 0xFF;
	if (!p->single_orientation) r = p->transmit_receive_orientations[acquisition];
	return r;
}

static inline void FN(focal_vector_for)(const OracleDAS *p, int acquisition, REAL *angle, REAL *depth)
{
	if (p->single_focus) { *angle = p->transmit_angle; *depth = p->focus_depth; }
	else { *angle = p->focal_vectors[2 * acquisition]; *depth = p->focal_vectors[2 * acquisition + 1]; }
}

/* das.glsl:154-202 */
static inline REAL FN(rca_transmit_distance)(const REAL *world, REAL angle_deg, REAL focal_depth, uint32_t txrx)
{
	REAL result = 0;
	uint32_t tx = (txrx >> 4) & 0xF;
	if (tx != BeamformerRCAOrientation_None) {
		int  tx_rows = tx == BeamformerRCAOrientation_Rows;
		REAL angle   = angle_deg * (R_PI / (REAL)180);
		REAL px = world[tx_rows ? 1 : 0], pz = world[2];
		if (R_ISINF(focal_depth)) {
			result = px * R_SIN(angle) + pz * R_COS(angle);
		} else {
			REAL fx = focal_depth * R_SIN(angle), fz = focal_depth * R_COS(angle);
			REAL dx = px - fx, dz = pz - fz;
			result  = R_SQRT(dx * dx + dz * dz);
		}
	}
	return result;
}

typedef struct { REAL c[3]; uint64_t pairs; } FN(acc);

/* RESULT_STORE (das.glsl:28-32): coherent sum plus, with coherency weighting, |value| */
static inline void FN(accumulate)(const OracleDAS *p, FN(acc) *a, FN(c2) v)
{
	a->c[0] += v.x;
	a->c[1] += v.y;
	if (p->coherency_weighting)
		a->c[2] += p->complex_data ? R_SQRT(v.x * v.x + v.y * v.y) : R_FABS(v.x);
	a->pairs++;
}

/* das.glsl:204-231 */
static void FN(rca)(const OracleDAS *p, const float *rf, const REAL *world, const REAL *xdc, FN(acc) *acc)
{
	int S = p->sample_count, A = p->acquisition_count;
	for (int acquisition = 0; acquisition < A; acquisition++) {
		uint32_t txrx    = FN(orientation_for)(p, acquisition);
		int      rx_rows = (txrx & 0xF) == BeamformerRCAOrientation_Rows;
		REAL angle, depth;
		FN(focal_vector_for)(p, acquisition, &angle, &depth);
		REAL xw[2] = {xdc[rx_rows ? 1 : 0], xdc[2]};
		REAL transmit_distance = FN(rca_transmit_distance)(world, angle, depth, txrx);

		int rf_offset = (int)p->rf_element_offset + acquisition * S;
		rf_offset    -= p->interpolation_mode == BeamformerInterpolationMode_Cubic;
		for (int chunk_channel = 0; chunk_channel < p->chunk_channel_count; chunk_channel++) {
			REAL rx_channel = (REAL)(p->channel_offset + chunk_channel);
			REAL rx_lateral = rx_channel * (REAL)p->xdc_element_pitch[rx_rows ? 1 : 0];
			REAL rv[2]      = {xw[0] - rx_lateral, xw[1] - 0};
			REAL a_arg      = R_FABS((REAL)p->f_number * rv[0] / R_FABS(xw[1]));
			if (a_arg < (REAL)0.5) {
				REAL sidx = FN(sample_index)(p, transmit_distance + R_SQRT(rv[0] * rv[0] + rv[1] * rv[1]));
				REAL apod = FN(apodize)(a_arg);
				FN(c2) v  = FN(sample_rf)(p, rf, rf_offset, sidx);
				v.x *= apod; v.y *= apod;
				FN(accumulate)(p, acc, v);
			}
			rf_offset += S * A;
		}
	}
}

/* das.glsl:233-286 */
static void FN(hercules)(const OracleDAS *p, const float *rf, const REAL *world, const REAL *xdc, FN(acc) *acc)
{
	int S = p->sample_count, A = p->acquisition_count, sparse = p->sparse != 0;
	uint32_t txrx    = FN(orientation_for)(p, 0);
	int      rx_cols = (txrx & 0xF) == BeamformerRCAOrientation_Columns;
	REAL angle, depth;
	FN(focal_vector_for)(p, 0, &angle, &depth);

	REAL transmit_index   = FN(sample_index)(p, FN(rca_transmit_distance)(world, angle, depth, txrx));
	REAL z_delta_squared  = xdc[2] * xdc[2];
	REAL f_number_over_z  = R_FABS((REAL)p->f_number / xdc[2]);
	REAL apodization_test = (REAL)0.25 / (f_number_over_z * f_number_over_z);
	REAL pitch_x = p->xdc_element_pitch[0], pitch_y = p->xdc_element_pitch[1];
	REAL fs = p->sampling_frequency, c = p->speed_of_sound;

	for (int chunk_channel = 0; chunk_channel < p->chunk_channel_count; chunk_channel++) {
		REAL rx_channel = (REAL)(p->channel_offset + chunk_channel);
		int  rf_offset  = (int)p->rf_element_offset + chunk_channel * S * A + sparse * S;
		rf_offset      -= p->interpolation_mode == BeamformerInterpolationMode_Cubic;

		REAL ex = xdc[0], ey = xdc[1];
		if (rx_cols) { ex -= rx_channel * pitch_x; ex *= ex; }
		else         { ey -= rx_channel * pitch_y; ey *= ey; }

		for (int transmit = sparse; transmit < A; transmit++) {
			REAL tx_channel = sparse ? (REAL)p->sparse_elements[transmit - sparse] : (REAL)transmit;
			if (rx_cols) { ey = xdc[1] - tx_channel * pitch_y; ey *= ey; }
			else         { ex = xdc[0] - tx_channel * pitch_x; ex *= ex; }

			REAL element_delta_squared = ex + ey;
			if (element_delta_squared < apodization_test) {
				REAL apodization = transmit == 0 ? (REAL)1 / R_SQRT((REAL)A) : (REAL)1;
				apodization     *= FN(apodize)(f_number_over_z * R_SQRT(element_delta_squared));
				REAL index = transmit_index + R_SQRT(z_delta_squared + element_delta_squared) * fs / c;
				FN(c2) v = FN(sample_rf)(p, rf, rf_offset, index);
				v.x *= apodization; v.y *= apodization;
				FN(accumulate)(p, acc, v);
			}
			rf_offset += S;
		}
	}
}

/* das.glsl:288-321 (ReadiGroupCount <= 1) and :323-366 (READI) */
static void FN(forces)(const OracleDAS *p, const float *rf, const REAL *xdc, FN(acc) *acc)
{
	int S = p->sample_count, A = p->acquisition_count, sparse = p->sparse != 0;
	int readi = p->readi_group_count > 1;
	REAL pitch_x = p->xdc_element_pitch[0], pitch_y = p->xdc_element_pitch[1];
	REAL fs = p->sampling_frequency, c = p->speed_of_sound;

	REAL z_delta_squared     = xdc[2] * xdc[2];
	REAL transmit_y_delta    = xdc[1] - pitch_y * (REAL)p->channel_count / 2;
	REAL transmit_yz_squared = transmit_y_delta * transmit_y_delta + z_delta_squared;
	int  hadamard_offset     = (int)p->readi_group * (int)p->readi_group_count;

	for (int chunk_channel = 0; chunk_channel < p->chunk_channel_count; chunk_channel++) {
		REAL rx_channel      = (REAL)(p->channel_offset + chunk_channel);
		REAL receive_x_delta = xdc[0] - rx_channel * pitch_x;
		REAL a_arg           = R_FABS((REAL)p->f_number * receive_x_delta / xdc[2]);
		if (!(a_arg < (REAL)0.5)) continue;

		REAL receive_index = FN(sample_index)(p, R_SQRT(receive_x_delta * receive_x_delta + z_delta_squared));
		REAL apodization   = FN(apodize)(a_arg);

		if (!readi) {
			int rf_offset = (int)p->rf_element_offset + chunk_channel * S * A + sparse * S;
			rf_offset    -= p->interpolation_mode == BeamformerInterpolationMode_Cubic;
			for (int transmit = sparse; transmit < A; transmit++) {
				REAL tx_channel       = sparse ? (REAL)p->sparse_elements[transmit - sparse] : (REAL)transmit;
				REAL transmit_x_delta = xdc[0] - pitch_x * tx_channel;
				REAL transmit_index   = R_SQRT(transmit_yz_squared + transmit_x_delta * transmit_x_delta) * fs / c;
				FN(c2) v = FN(sample_rf)(p, rf, rf_offset, receive_index + transmit_index);
				v.x *= apodization; v.y *= apodization;
				FN(accumulate)(p, acc, v);
				rf_offset += S;
			}
		} else {
			int channel_rf_offset = (int)p->rf_element_offset + chunk_channel * S * A;
			channel_rf_offset    -= p->interpolation_mode == BeamformerInterpolationMode_Cubic;
			for (int tx_group = 0; tx_group < (int)p->readi_group_count; tx_group++) {
				REAL group_apodization = apodization * (REAL)p->readi_hadamard[hadamard_offset + tx_group];
				int  rf_offset = channel_rf_offset;
				for (int tx_event = 0; tx_event < A; tx_event++) {
					REAL tx_element       = (REAL)tx_group * (REAL)A + (REAL)tx_event;
					REAL transmit_x_delta = xdc[0] - pitch_x * tx_element;
					REAL transmit_index   = R_SQRT(transmit_yz_squared + transmit_x_delta * transmit_x_delta) * fs / c;
					FN(c2) v = FN(sample_rf)(p, rf, rf_offset, receive_index + transmit_index);
					v.x *= group_apodization; v.y *= group_apodization;
					FN(accumulate)(p, acc, v);
					rf_offset += S;
				}
			}
		}
	}
}

static inline void FN(m4_point)(const float *m, const REAL *v, REAL *out)
{
	for (int i = 0; i < 3; i++)
		out[i] = (REAL)m[i] * v[0] + (REAL)m[4 + i] * v[1] + (REAL)m[8 + i] * v[2] + (REAL)m[12 + i];
}

/* das.glsl:368-407 */
static uint64_t FN(das_run)(const OracleDAS *p, const float *rf, REAL *output, REAL *incoherent)
{
	uint32_t X = p->output_size[0], Y = p->output_size[1], Z = p->output_size[2];
	uint32_t z0 = p->z_count ? p->z_first : 0, zn = p->z_count ? p->z_count : Z;
	uint32_t y0 = p->y_count ? p->y_first : 0, yn = p->y_count ? p->y_count : Y;
	int      elements = p->complex_data ? 2 : 1;
	uint64_t pairs = 0;
	int64_t  rows  = (int64_t)zn * yn;
	int64_t  row_begin = 0, row_end = rows;
	if (p->row_count > 0) { row_begin = p->row_first; row_end = p->row_first + p->row_count < rows ? p->row_first + p->row_count : rows; }

	#pragma omp parallel for schedule(dynamic, 1) reduction(+:pairs) num_threads(oracle_thread_count(p->threads))
	for (int64_t row = row_begin; row < row_end; row++) {
		uint32_t zl = (uint32_t)(row / yn), yl = (uint32_t)(row % yn);
		uint32_t z = z0 + zl * (p->z_stride ? p->z_stride : 1u), y = y0 + yl * (p->y_stride ? p->y_stride : 1u);
		for (uint32_t x = 0; x < X; x++) {
			REAL point[3] = {
				(REAL)x / (REAL)(X > 2 ? X - 1 : 1),
				(REAL)y / (REAL)(Y > 2 ? Y - 1 : 1),
				(REAL)z / (REAL)(Z > 2 ? Z - 1 : 1),
			};
			REAL world[3], xdc[3];
			FN(m4_point)(p->voxel_transform, point, world);
			FN(acc) acc = {{0, 0, 0}, 0};
			oracle_near_half_budget = 0.f;
			switch (p->acquisition_kind) {
			case BeamformerAcquisitionKind_FORCES:
			case BeamformerAcquisitionKind_UFORCES:
				FN(forces)(p, rf, world, &acc);
				break;
			case BeamformerAcquisitionKind_HERCULES:
			case BeamformerAcquisitionKind_UHERCULES:
			case BeamformerAcquisitionKind_HERO_PA:
				FN(m4_point)(p->xdc_transform, world, xdc);
				FN(hercules)(p, rf, world, xdc, &acc);
				break;
			case BeamformerAcquisitionKind_Flash:
			case BeamformerAcquisitionKind_RCA_TPW:
			case BeamformerAcquisitionKind_RCA_VLS:
				FN(m4_point)(p->xdc_transform, world, xdc);
				FN(rca)(p, rf, world, xdc, &acc);
				break;
			default: break;
			}
			uint64_t out_index = (uint64_t)X * yn * zl + (uint64_t)X * yl + x;
			if (oracle_near_half_buffer) oracle_near_half_buffer[out_index] += oracle_near_half_budget;
			if (p->coherency_weighting) incoherent[out_index] += acc.c[2];
			output[elements * out_index] += acc.c[0];
			if (elements == 2) output[elements * out_index + 1] += acc.c[1];
			pairs += acc.pairs;
		}
	}
	return pairs;
}
