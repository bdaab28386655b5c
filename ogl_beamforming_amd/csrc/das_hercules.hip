/* das_hercules.hip -- delay-and-sum fast path for the HERCULES family on gfx950 (MI355X).
 *
 * Same arithmetic contract as das.hip's das_hercules (shaders/das.glsl:233-286 of the
 * reference): one transmit focus, a 2-D aperture of (receive channel c) x (decoded transmit
 * element t), and for every element pair
 *     e^2   = (lateral_rx - c pitch_rx)^2 + (lateral_tx - t pitch_tx)^2
 *     index = T0(voxel) + sqrt(z^2 + e^2) fs / c          pass: e^2 < 0.25 / (F# / z)^2
 *     out  += w_t cos^2(pi F#/z sqrt(e^2)) * rotate_iq(interpolate(rf[c][t], index), index).
 *
 * The general kernel is VALU bound at ~130 VALU clocks per pair (tools/microbench.hip: a wave64 f32
 * instruction holds its SIMD for 2.5 clocks, a packed one 4.3, v_fract / v_cvt 4.1, a transcendental
 * 8.15; the texture-address path takes 16.3 clocks per CU for the pair's one gather).  This kernel
 * applies when the volume's lateral axes are aligned with the array's (exact-zero matrix
 * coefficients, host check plan_hercules): the 64 lanes of a wave lie along the output's x axis and
 * share y and z, so one of the two lateral terms of e^2 is per lane and constant over the inner
 * loop, and the other is WAVE UNIFORM: it comes from a small global table D2[y][n] (built per
 * launch by hercules_table_kernel) through scalar loads and enters the vector ALU as an SGPR
 * operand.  Per pair that leaves
 *     two packed adds for (e^2, z^2 + e^2) of two elements, v_sqrt, a packed fma for two indices;
 *     cos^2(sqrt(w)), w = (pi F#/z)^2 e^2, as a degree-5 polynomial in w evaluated for two
 *       elements per packed instruction (|error| < 7e-7) instead of v_sqrt + v_cos;
 *     floor / fraction / address, ONE 16-byte gather, the interpolation as two packed ops;
 *     v_fract + v_sin + v_cos of the demodulation phase, the phasor scaled by the apodization in
 *       one packed multiply, two packed FMAs of rotate-accumulate;
 *     |sample| for coherency weighting (multiply, fma, v_sqrt, fma).
 * That is ~95 VALU clocks per pair, four of its instructions transcendental (the distance and |sample|
 * square roots, sin, cos): the kernel runs with the vector ALU 99 % busy.
 * The f-number test and the row range test are decided per wave and outer element from the
 * table row's extremes: when every lane passes both for the whole inner loop (the common case)
 * the loop runs without compares, selects or branches; otherwise a checked instantiation runs.
 * Gathers of four pairs are in flight together.  No LDS, no MFMA: gather-accumulate.
 *
 * "outer"/"inner": the array axis that runs along the output's x axis is the outer loop (its
 * per-lane term is computed once per element), the other one the inner loop.  Either may be the
 * receive or the transmit axis; only the RF row strides and the first-transmit weight differ.
 */
#include "das_exact.h"

#define BF_HERC_BATCH 4
/* cos^2(sqrt(w)) on [0, (pi/2)^2]: degree-5 fit at the Chebyshev nodes, |error| < 4.0e-7 (6.7e-7 as an
 * f32 Horner chain) -- next to 1e-4 of parity tolerance and the ~1e-6 of the hardware's v_cos */
#define BF_APOD_C0  0.9999996f
#define BF_APOD_C1 -0.9999883f
#define BF_APOD_C2  0.33327785f
#define BF_APOD_C3 -0.044347722f
#define BF_APOD_C4  0.0030977894f
#define BF_APOD_C5 -0.000112471265f

namespace {

__device__ __forceinline__ f32x2 splat(float v) { return f32x2{v, v}; }

__device__ __forceinline__ f32x2 apod_poly(f32x2 w)
{
	f32x2 r = splat(BF_APOD_C5);
	r = r * w + splat(BF_APOD_C4);
	r = r * w + splat(BF_APOD_C3);
	r = r * w + splat(BF_APOD_C2);
	r = r * w + splat(BF_APOD_C1);
	r = r * w + splat(BF_APOD_C0);
	return r;
}

/* das.glsl:187-202 with the per-transmit constants precomputed (same as das.hip) */
__device__ __forceinline__ float transmit_distance(const BfTransmit &t, float wx, float wy, float wz)
{
	float result = 0.f;
	if (!(t.flags & BF_TX_NONE)) {
		float px = (t.flags & BF_TX_ROWS) ? wy : wx;
		if (t.flags & BF_TX_PLANE) {
			result = px * t.sin_a + wz * t.cos_a;
		} else {
			float dx = px - t.focus_x, dz = wz - t.focus_z;
			result = hw_sqrt(dx * dx + dz * dz);
		}
	}
	return result;
}

/* transducer-space lateral coordinate `coord` (0 = x, 1 = y) of voxel (x, y, z) */
__device__ __forceinline__ void voxel_to_xdc(const BfDasArgs &p, uint32_t x, uint32_t y, uint32_t z,
                                             float &wx, float &wy, float &wz, float &xx, float &xy, float &xz)
{
	float px = (float)x / fmaxf(1.0f, (float)p.size[0] - 1.0f);       /* das.glsl:374-376 */
	float py = (float)y / fmaxf(1.0f, (float)p.size[1] - 1.0f);
	float pz = (float)z / fmaxf(1.0f, (float)p.size[2] - 1.0f);
	m4_point(p.voxel_transform, px, py, pz, wx, wy, wz);
	m4_point(p.xdc_transform, wx, wy, wz, xx, xy, xz);
}

} /* namespace */

/* pairs[i] = {rf[i], rf[i + 1] - rf[i]} for every complex sample of every row (the last sample of a row gets a zero
 * difference: it is never a pair's first tap); 32 zero bytes behind the last row for the range-checked loop */
__global__ __launch_bounds__(256) void hercules_pair_kernel(const f32x2 *__restrict__ rf, f32x4 *__restrict__ pairs, uint32_t total, uint32_t samples)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= total + 2u) return;
	f32x4 out = {0.f, 0.f, 0.f, 0.f};
	if (i < total) {
		const f32x2 s0 = rf[i];
		f32x2 ds = {0.f, 0.f};
		if ((i + 1u) % samples != 0u) ds = rf[i + 1u] - s0;
		out = f32x4{s0.x, s0.y, ds.x, ds.y};
	}
	pairs[i] = out;
}

/* poly[2 i], poly[2 i + 1] = {a0, a1}, {a2, a3} of the Catmull-Rom segment between samples i and i + 1 of its row
 * (das.glsl:67-97); segments that are never valid (the first and the last two of a row: index < 1 or >= S - 2) are zero, and
 * so are the 32 bytes behind the last row, the checked loop's target for invalid pairs */
__global__ __launch_bounds__(256) void hercules_cubic_kernel(const f32x2 *__restrict__ rf, f32x4 *__restrict__ poly, uint32_t total, uint32_t samples)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i > total) return;
	f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
	const uint32_t k = i % samples;
	if (i < total && k >= 1u && k + 2u < samples) {
		const f32x2 P0 = rf[i - 1u], P1 = rf[i], P2 = rf[i + 1u], P3 = rf[i + 2u];
		const f32x2 T1 = 0.5f * (P2 - P0), T2 = 0.5f * (P3 - P1);
		const f32x2 a2 = 3.f * (P2 - P1) - 2.f * T1 - T2, a3 = 2.f * (P1 - P2) + T1 + T2;
		lo = f32x4{P1.x, P1.y, T1.x, T1.y};
		hi = f32x4{a2.x, a2.y, a3.x, a3.y};
	}
	poly[2u * i] = lo; poly[2u * i + 1u] = hi;
}

/* D2[y * pitch + n] = (uniform lateral coordinate of output row y - position of inner element n)^2,
 * n < inner_count; entries up to the row pitch repeat the last element (never used for sums);
 * extremes[y] = {min, max} of the row.  One block per output row. */
__global__ __launch_bounds__(256) void hercules_table_kernel(const BfDasArgs p, const BfHerculesArgs q)
{
	const uint32_t y = blockIdx.x;
	float wx, wy, wz, xx, xy, xz;
	voxel_to_xdc(p, 0, y, p.z_first, wx, wy, wz, xx, xy, xz);
	const float lateral = q.inner_coord ? xy : xx;
	const float pitch   = q.inner_coord ? p.pitch[1] : p.pitch[0];
	float lo = __builtin_inff(), hi = -__builtin_inff();
	for (uint32_t n = threadIdx.x; n < q.table_pitch; n += blockDim.x) {
		uint32_t m = n < q.inner_count ? n : q.inner_count - 1;
		float element = (q.inner_is_transmit && p.sparse) ? (float)p.sparse_elements[m] : (float)m;
		float delta = lateral - element * pitch;
		float d2 = delta * delta * q.unit_scale2;
		q.table[(size_t)y * q.table_pitch + n] = d2;
		lo = fminf(lo, d2); hi = fmaxf(hi, d2);
	}
	__shared__ float red[2][4];
	for (int off = 32; off > 0; off >>= 1) {
		lo = fminf(lo, __shfl_xor(lo, off, 64));
		hi = fmaxf(hi, __shfl_xor(hi, off, 64));
	}
	if ((threadIdx.x & 63u) == 0) { red[0][threadIdx.x >> 6] = lo; red[1][threadIdx.x >> 6] = hi; }
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < 4; w++) { lo = fminf(lo, red[0][w]); hi = fmaxf(hi, red[1][w]); }
		q.extremes[2 * y] = lo; q.extremes[2 * y + 1] = hi;
	}
}

/* Grid: q.tiles[0] * q.tiles[1] * q.tiles[2] blocks of 64 * q.rows threads; a block is q.rows waves = q.rows output rows
 * (y) x 64 voxels along x of one z plane. */
/* PL ("phase local", IQ only): the demodulation phase of a pair is sin / cos of turns = turns_per_sample x index, which the
 * hardware takes in revolutions within +-256.  Absolute indices reach thousands, so the phase has to be reduced: by a v_fract per
 * pair (PL = false), or -- when the host knows the indices of one inner loop stay within ~400 turns of each other
 * (BfHerculesArgs::phase_local) -- by subtracting ONE integer per lane and outer element inside the fma that forms the turns:
 * turns' = fma(index, turns_per_sample, -B), B = round(turns at the middle of the lane's index range).  One instruction less per
 * pair, and the product is rounded at |turns'| <= a few hundred instead of at the absolute phase. */
/* PD ("paired data", linear interpolation of IQ samples): the kernel reads q.pairs, a copy of the DAS input in which every
 * sample carries the difference to its successor -- {s_k, s_(k+1) - s_k}, 16 bytes, written by hercules_pair_kernel in ~0.2 ms
 * per frame -- so the two taps of a pair are one 16-byte ALIGNED gather and the interpolation one packed fma (the difference the
 * loop used to form per pair is formed once per sample; the arithmetic and its rounding are the same).  With cubic
 * interpolation (what the reference's throughput harness runs, tests/throughput.c:451) the copy holds, per sample k, the four
 * complex coefficients of the Catmull-Rom segment [k, k + 1] as a polynomial in the fraction (das.glsl:67-97: a0 = P1,
 * a1 = T1, a2 = 3 (P2 - P1) - 2 T1 - T2, a3 = 2 (P1 - P2) + T1 + T2) -- 32 bytes, two aligned gathers -- and the Hermite
 * weights, the tangents and the four-term sum of every pair become a three-step Horner chain of packed fmas. */
/* ROW_ENDS: the instantiation for launches in which a pair can come within reach of an end of its RF row (BfDasArgs::row_ends, a host
 * bound: das_select.cpp); the other one carries none of that code -- its loops and its register allocation are round 3's. */
template <int INTERP, bool CPLX, bool CW, bool PL, bool PD, bool ROW_ENDS>
__global__ __launch_bounds__(1024) void das_hercules_kernel(const BfDasArgs p, const BfHerculesArgs q)
{
	constexpr bool EDGES = ROW_ENDS && INTERP != BF_INTERP_NEAREST;     /* (nearest flips at every half-integer: budgeted per voxel by the tests) */
	static_assert(!PD || (CPLX && INTERP != BF_INTERP_NEAREST), "prepared data: linear or cubic interpolation of complex samples");
	constexpr bool POLY = PD && INTERP == BF_INTERP_CUBIC;       /* q.pairs holds the cubic segment polynomials (hercules_cubic_kernel) */
	constexpr uint32_t ES = POLY ? 32 : PD ? 16 : CPLX ? 8 : 4;
	/* cubic interpolation of IQ samples out of the RF itself (coarse grids, where the prepared polynomial copy costs more in memory traffic
	 * than it saves -- the reference harness's view plane): index arithmetic as for the prepared forms, Catmull-Rom as a Horner cubic
	 * (das_factored.hip: 12 packed operations) instead of four Hermite weights per pair */
	constexpr bool RAWC = !PD && CPLX && INTERP == BF_INTERP_CUBIC;
	using VT = sample_t<CPLX>;

	/* blockIdx -> tile with each XCD walking a contiguous run of tiles (das.hip) */
	const uint32_t total = q.tiles[0] * q.tiles[1] * q.tiles[2];
	const uint32_t per   = (total + 7u) / 8u;
	const uint32_t tile  = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
	if (q.depth_major != 3u && tile >= total) return;
	/* depth-major walk (as das_separable.hip): the tiles an XCD has in flight together are one lateral
	 * column at consecutive depths, whose RF windows overlap almost entirely */
	uint32_t zl, tx_, ty_;
	if (q.depth_major == 3u) {                /* view planes in XCD-balanced bands (bf_kernels.h) */
		zl = 0;
		if (!bf_plane_walk<true>(blockIdx.x, q.tiles[0], q.tiles[1], q.band_rows, tx_, ty_)) return;
	} else if (q.depth_major == 2u) {                /* view planes: rows (= depth) fastest, an XCD's run is a lateral column (das.hip) */
		ty_ = tile % q.tiles[1];
		tx_ = (tile / q.tiles[1]) % q.tiles[0];
		zl  = tile / (q.tiles[1] * q.tiles[0]);
	} else if (q.depth_major) {
		zl  = tile % q.tiles[2];
		tx_ = (tile / q.tiles[2]) % q.tiles[0];
		ty_ = tile / (q.tiles[2] * q.tiles[0]);
	} else {
		tx_ = tile % q.tiles[0];
		ty_ = (tile / q.tiles[0]) % q.tiles[1];
		zl  = tile / (q.tiles[0] * q.tiles[1]);
	}
	const uint32_t z   = p.z_first + zl;

	const uint32_t lane = threadIdx.x & 63u;
	/* the wave's output row: wave uniform, told to the compiler so that table reads become scalar loads */
	/* (a wave whose row lies beyond the grid repeats the last row and stores nothing: every wave of the block reaches the barriers below) */
	const uint32_t y_real = __builtin_amdgcn_readfirstlane(ty_ * q.rows + (threadIdx.x >> 6));
	const uint32_t y = y_real < p.size[1] ? y_real : p.size[1] - 1u;
	const uint32_t x_real = tx_ * 64u + lane;
	const bool     inside = x_real < p.size[0] && y_real < p.size[1];
	const uint32_t x = x_real < p.size[0] ? x_real : p.size[0] - 1u;           /* idle lanes repeat the last voxel, store nothing */

	float wx, wy, wz, xx, xy, xz;
	voxel_to_xdc(p, x, y, z, wx, wy, wz, xx, xy, xz);

	const int S = p.sample_count, A = p.acquisition_count;
	const BfTransmit t0 = p.transmits[0];
	/* squared distances are held in the host-chosen unit (BfHerculesArgs::unit_scale2) in which the
	 * distance -> samples factor is free of rounding bias; the per-voxel transmit term divides by c */
	const float s2        = q.unit_scale2;
	const float fs_over_c = q.samples_per_unit;
	const float T0  = (div_speed_of_sound(transmit_distance(t0, wx, wy, wz), p) + p.time_offset) * p.sampling_frequency;
	const float z2  = xz * xz * s2;
	const float f_number_over_z  = __builtin_fabsf(p.f_number * hw_rcp(xz));
	const float apodization_test = 0.25f / (f_number_over_z * f_number_over_z) * s2;
	const float w_scale = (3.14159265358979f * f_number_over_z) * (3.14159265358979f * f_number_over_z) * hw_rcp(s2);
	const float turns_per_sample = p.turns_per_sample;

	const float outer_lateral = q.inner_coord ? xx : xy;           /* the other coordinate */
	const float outer_pitch   = q.inner_coord ? p.pitch[0] : p.pitch[1];
	const float *__restrict__ row_d2 = q.table + (size_t)y * q.table_pitch;
	const float d2_min = q.extremes[2 * y], d2_max = q.extremes[2 * y + 1];

	const char *rf = PD ? (const char *)q.pairs : (const char *)p.rf;
	const uint32_t ulast = (uint32_t)(S - 1);
	const int   n_inner = (int)q.inner_count, n_outer = (int)q.outer_count;
	/* byte strides of the RF rows along the two loops: rf[c][t] rows of S samples */
	const uint32_t row_bytes    = (uint32_t)S * ES;
	const uint32_t inner_stride = q.inner_is_transmit ? row_bytes : row_bytes * (uint32_t)A;
	const uint32_t outer_stride = q.inner_is_transmit ? row_bytes * (uint32_t)A : row_bytes;
	const uint32_t sparse_rows  = p.sparse ? row_bytes : 0u;       /* UHERCULES: transmit rows start at 1 */

	VT    coherent   = zero_sample<CPLX>();
	float incoherent = 0.f;
	[[maybe_unused]] const float edge_margin = p.edge_margin;
	[[maybe_unused]] unsigned long long edge_outer[4] = {0, 0, 0, 0};       /* the outer elements (bit m; at most 256) whose checked loop met such a pair: scalars */

	for (int m = 0; m < n_outer; m++) {
		/* the waves of a block -- adjacent output rows, whose RF windows share their cache lines -- walk the outer elements IN STEP: left to
		 * themselves they drift apart over the 128-256 elements and every wave pulls its own copy of the lines through L1.  Same-box A/B
		 * (das_select.cpp kHerculesRows): config 5 2896 -> 2638 ms with 8 rows, the reference harness's HERCULES plane 18.46 -> 17.47 ms */
		__syncthreads();
		float outer_element = (!q.inner_is_transmit && p.sparse) ? (float)p.sparse_elements[m] : (float)m;
		float od  = outer_lateral - outer_element * outer_pitch;
		float od2 = od * od * s2;
		/* wave-level decisions for this outer element from the table row's extremes */
		const bool  lane_none = !(od2 + d2_min < apodization_test);
		if (__builtin_amdgcn_ballot_w64(!lane_none) == 0) continue;             /* nobody passes anything */
		const bool  lane_all  = (od2 + d2_max < apodization_test);
		const float i_lo = T0 + hw_sqrt(z2 + (od2 + d2_min)) * fs_over_c;
		const float i_hi = T0 + hw_sqrt(z2 + (od2 + d2_max)) * fs_over_c;
		bool lane_range;
		if constexpr (INTERP == BF_INTERP_LINEAR)       lane_range = i_lo >= 0.5f && i_hi < (float)(S - 1) - 0.5f;
		else if constexpr (INTERP == BF_INTERP_CUBIC)   lane_range = i_lo >= 1.5f && i_hi < (float)(S - 2) - 0.5f;
		else                                            lane_range = i_lo >= 0.5f && i_hi < (float)S - 1.0f;
		const bool wave_fast = __builtin_amdgcn_ballot_w64(!(lane_all && lane_range)) == 0;

		const uint32_t row0 = (uint32_t)m * outer_stride + sparse_rows;
		/* partial sums of this outer element; acc1/acc2 hold re*(cos,sin) and im*(cos,sin) */
		f32x2 acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f};
		float accr = 0.f, mag = 0.f;
		[[maybe_unused]] unsigned long long edge_here = 0;      /* checked loop, row-end instantiation: lanes with a pair within the margin of a row end (a scalar) */
		const f32x2 od2p = splat(od2), z2p = splat(z2), T0p = splat(T0), kp = splat(fs_over_c),
		            wsp = splat(w_scale), tpsp = splat(turns_per_sample),
		            oz2p = splat(od2 + z2), wodp = splat(w_scale * od2);
		[[maybe_unused]] const f32x2 btp = splat(PL ? __builtin_rintf(turns_per_sample * 0.5f * (i_lo + i_hi)) : 0.f);

		/* B = 1 or BF_HERC_BATCH elements starting at inner element n; d2in = their table entries */
		auto group = [&](auto checked_c, auto count_c, int n, const float *d2in, float first_weight) {
			constexpr bool CHECK = decltype(checked_c)::value;
			constexpr int  B     = decltype(count_c)::value;
			constexpr int  P     = (B + 1) / 2;
			float d2[2 * P];
			#pragma unroll
			for (int k = 0; k < 2 * P; k++) d2[k] = d2in[k < B ? k : B - 1];
			f32x2 e2[P], dist[P], index[P], apod[P], turns[P];
			#pragma unroll
			for (int k = 0; k < P; k++) {
				const f32x2 dn = f32x2{d2[2 * k], d2[2 * k + 1]};
				f32x2 dd;
				if constexpr (CHECK) {
					e2[k] = od2p + dn;                       /* the reference's association: the f-number test reads it */
					dd    = z2p + e2[k];
					apod[k] = apod_poly(e2[k] * wsp);
				} else {
					dd      = oz2p + dn;                     /* (od2 + z2) + d2: one packed add */
					apod[k] = apod_poly(dn * wsp + wodp);    /* w = ws e2 as one packed fma */
				}
				dist[k]  = f32x2{hw_sqrt(dd.x), hw_sqrt(dd.y)};
				/* (checked loop: explicit fmas -- the row-end pass at the end of the kernel forms the same index again, bit for bit) */
				if constexpr (CHECK && EDGES) index[k] = f32x2{__builtin_fmaf(dist[k].x, fs_over_c, T0), __builtin_fmaf(dist[k].y, fs_over_c, T0)};
				else                 index[k] = dist[k] * kp + T0p;
				if constexpr (CPLX) { if constexpr (PL) turns[k] = index[k] * tpsp - btp; else turns[k] = index[k] * tpsp; }
			}
			/* (floor / fraction of two indices by packed adds -- index + 2^23 - 1/2 leaves floor(index) in the low
			 * mantissa bits -- measured the same as v_cvt_flr + v_fract per index, 179.1 against 178.7 ms on 32 planes of
			 * config 5, and needs margins at the row ends: not used) */
			float   frac[B], ap[B];
			Tap<INTERP> tap[B];
			uint32_t off[B];
			#pragma unroll
			for (int k = 0; k < B; k++) {
				[[maybe_unused]] float idx = (k & 1) ? index[k >> 1].y : index[k >> 1].x;
				float e   = 0.f;
				if constexpr (CHECK) e = (k & 1) ? e2[k >> 1].y : e2[k >> 1].x;
				ap[k]     = (k & 1) ? apod[k >> 1].y : apod[k >> 1].x;
				if (k == 0) ap[k] *= first_weight;
				uint32_t row = row0 + (uint32_t)(n + k) * inner_stride;
				if constexpr (INTERP == BF_INTERP_LINEAR || POLY || RAWC) {
					frac[k] = hw_fract(idx);
					uint32_t ki = (uint32_t)cvt_floor_i32(idx);
					const uint32_t first = RAWC ? ki - 1u : ki;              /* raw cubic taps start one sample early */
					off[k] = CHECK ? row + first * ES : first * ES;
					if constexpr (CHECK) {
						/* linear: 0 <= index < S - 1; cubic: 1 <= index < S - 2 (das.glsl:99-124) */
						bool ok = ((POLY || RAWC) ? (ki - 1u) < (uint32_t)(S - 3) : ki < ulast) && (e < apodization_test);
						if constexpr (EDGES) edge_here |= __builtin_amdgcn_ballot_w64(bfx::edge_near<INTERP>(idx, S, edge_margin) && e < apodization_test);
						off[k] = ok ? off[k] : q.zero_offset;
						ap[k]  = ok ? ap[k] : 0.f;
					}
				} else {
					tap[k] = tap_setup<INTERP, CPLX>(idx, (float)S, S - 1);
					off[k] = row + tap[k].off;
					if constexpr (CHECK) ap[k] = (e < apodization_test) ? ap[k] : 0.f;
					if constexpr (CHECK && EDGES) edge_here |= __builtin_amdgcn_ballot_w64(bfx::edge_near<INTERP>(idx, S, edge_margin) && e < apodization_test);
				}
			}
			TapData<INTERP, CPLX> d[B];
			#pragma unroll
			for (int k = 0; k < B; k++) {
				if constexpr (POLY) {
					const char *rowp = CHECK ? rf : rf + (row0 + (uint32_t)(n + k) * inner_stride);
					d[k].a = gather<f32x4>(rowp, off[k]);
					d[k].b = gather_at<f32x4, 16>(rowp, off[k]);
				} else if constexpr (RAWC) {
					const char *rowp = CHECK ? rf : rf + (row0 + (uint32_t)(n + k) * inner_stride);
					d[k].a = gather<f32x4_a8>(rowp, off[k]);
					d[k].b = gather_at<f32x4_a8, 16>(rowp, off[k]);
				} else if constexpr (!CHECK && INTERP == BF_INTERP_LINEAR) {
					/* the row (wave uniform) rides in the load's scalar base and the lane offset is one full-rate shift: a
					 * three-operand v_lshl_add_u32 is a half-rate instruction (tools/microbench.hip) */
					const char *rowp = rf + (row0 + (uint32_t)(n + k) * inner_stride);
					d[k] = tap_load<INTERP, CPLX>(rowp, off[k]);
				} else {
					d[k] = tap_load<INTERP, CPLX>(rf, off[k]);
				}
			}
			#pragma unroll
			for (int k = 0; k < B; k++) {
				VT sv;
				if constexpr (POLY) {
					const float t = frac[k];
					sv = f32x2{d[k].b.z, d[k].b.w} * t + f32x2{d[k].b.x, d[k].b.y};
					sv = sv * t + f32x2{d[k].a.z, d[k].a.w};
					sv = sv * t + f32x2{d[k].a.x, d[k].a.y};
				} else if constexpr (RAWC) {
					/* Catmull-Rom as four weights of the taps (bf_catmull_rom, das_common.h: nine scalar operations + one packed multiply and three
					 * packed fmas; the Horner form in the samples this had through round 3 was twelve packed operations) */
					const f32x2 s0 = {d[k].a.x, d[k].a.y}, s1 = {d[k].a.z, d[k].a.w}, s2 = {d[k].b.x, d[k].b.y}, s3 = {d[k].b.z, d[k].b.w};
					float w0, w1, w2, w3;
					bf_catmull_rom(frac[k], w0, w1, w2, w3);
					sv = w0 * s0 + w1 * s1 + w2 * s2 + w3 * s3;
				} else if constexpr (INTERP == BF_INTERP_LINEAR) {
					if constexpr (PD)        { f32x2 s0 = {d[k].a.x, d[k].a.y}, ds = {d[k].a.z, d[k].a.w}; sv = s0 + frac[k] * ds; }
					else if constexpr (CPLX) { f32x2 s0 = {d[k].a.x, d[k].a.y}, s1 = {d[k].a.z, d[k].a.w}; sv = s0 + frac[k] * (s1 - s0); }
					else                { sv = d[k].a.x + frac[k] * (d[k].a.y - d[k].a.x); }
				} else {
					sv = tap_finish<INTERP, CPLX>(tap[k], d[k]);
				}
				if constexpr (CPLX) {
					float tr = (k & 1) ? turns[k >> 1].y : turns[k >> 1].x;
					if constexpr (!PL) tr = hw_fract(tr);
					f32x2 cs = f32x2{hw_cos_turns(tr), hw_sin_turns(tr)} * ap[k];
					acc1 += sv.x * cs;
					acc2 += sv.y * cs;
					/* plain multiply + two FMAs: a packed square followed by scalar adds is slower here */
					if constexpr (CW) mag = __builtin_fmaf(ap[k], hw_sqrt(__builtin_fmaf(sv.y, sv.y, sv.x * sv.x)), mag);
				} else {
					accr = __builtin_fmaf(ap[k], sv, accr);
					if constexpr (CW) mag = __builtin_fmaf(ap[k], __builtin_fabsf(sv), mag);
				}
			}
		};

		auto inner = [&](auto checked_c) {
			constexpr int NB = BF_HERC_BATCH;
			int n = 0;
			/* the reference's weight of transmit 0 (das.glsl:272-273); UHERCULES never visits it */
			if (q.inner_is_transmit && !p.sparse) {
				float d2 = row_d2[0];
				group(checked_c, std::integral_constant<int, 1>{}, 0, &d2, p.first_transmit_weight);
				n = 1;
			}
			/* table entries of the next batch are fetched (scalar loads) while this one computes; the
			 * row is padded so that reading one batch past the end stays inside it */
			float cur[NB];
			#pragma unroll
			for (int k = 0; k < NB; k++) cur[k] = row_d2[n + k];
			for (; n + NB <= n_inner; n += NB) {
				float nxt[NB];
				#pragma unroll
				for (int k = 0; k < NB; k++) nxt[k] = row_d2[n + NB + k];
				group(checked_c, std::integral_constant<int, NB>{}, n, cur, 1.0f);
				#pragma unroll
				for (int k = 0; k < NB; k++) cur[k] = nxt[k];
			}
			for (; n < n_inner; n++) {
				float d2 = row_d2[n];
				group(checked_c, std::integral_constant<int, 1>{}, n, &d2, 1.0f);
			}
		};
		if (wave_fast) inner(std::false_type{});
		else           inner(std::true_type{});
		if constexpr (EDGES) { if (edge_here != 0ull) edge_outer[(m >> 6) & 3] |= 1ull << (m & 63); }

		/* fold this outer element (its weight when the outer loop is the transmit loop) */
		const float outer_weight = (!q.inner_is_transmit && !p.sparse && m == 0) ? p.first_transmit_weight : 1.0f;
		if constexpr (CPLX) {
			coherent.x += outer_weight * (acc1.x - acc2.y);
			coherent.y += outer_weight * (acc1.y + acc2.x);
		} else {
			coherent += outer_weight * accr;
		}
		if constexpr (CW) incoherent += outer_weight * mag;
	}
	if constexpr (EDGES) {
		{
			/* ---- row ends (das_exact.h).  The loops above are round 3's, untouched: they decide sample_rf's range test with their own index.
			 * Here the outer elements whose checked loop met a pair within the margin of an end (noted there: one compare per pair) are
			 * walked again -- the same index arithmetic as the checked loop's (explicit fmas there and here) --; such a pair is evaluated from the
			 * voxel's integer coordinates with the shader's own index, and the voxel's sums are corrected by the difference to what the
			 * loops decided and added for it.  Its registers are not the loops'. */
			const int sparse = p.sparse != 0;
			uint32_t ex = x;
			asm volatile("" : "+v"(ex));                       /* (or hipcc evaluates the exact voxel transform at the top of the kernel and keeps it across the loops) */
			for (int m = 0; m < n_outer; m++) {
				if (!((edge_outer[(m >> 6) & 3] >> (m & 63)) & 1ull)) continue;
				const float outer_element = (!q.inner_is_transmit && p.sparse) ? (float)p.sparse_elements[m] : (float)m;
				const float od  = outer_lateral - outer_element * outer_pitch;
				const float od2 = od * od * s2;
				for (int n = 0; n < n_inner; n++) {
					const float e2  = od2 + row_d2[n];
					const float idx = __builtin_fmaf(hw_sqrt(z2 + e2), fs_over_c, T0);
					if (e2 < apodization_test && bfx::edge_near<INTERP>(idx, S, edge_margin)) {
						const int channel  = q.inner_is_transmit ? m : n;
						const int transmit = (q.inner_is_transmit ? n : m) + sparse;
						bfx::edge_correct<BF_DAS_HERCULES, INTERP, CPLX, CW>(bfx::kernel_args(), ex, y, z, channel, transmit, idx, coherent, incoherent);
					}
				}
			}
		}
	}
	if (!inside) return;

	uint64_t out_index = (uint64_t)p.size[0] * p.size[1] * zl + (uint64_t)p.size[0] * y + x;
	if constexpr (CW) coherent = coherent * (coherent / incoherent);   /* coherency_weighting.glsl:36 */
	reinterpret_cast<VT *>(p.out)[out_index] = coherent;
}

template <int INTERP, bool CPLX, bool CW, bool PL, bool PD>
static hipError_t launch_herc(const BfDasArgs *a, const BfHerculesArgs *q, hipStream_t s)
{
	hipLaunchKernelGGL(hercules_table_kernel, dim3(a->size[1]), dim3(256), 0, s, *a, *q);
	if constexpr (PD) {
		const uint32_t total = (uint32_t)a->channel_count * (uint32_t)a->acquisition_count * (uint32_t)a->sample_count;
		if constexpr (INTERP == BF_INTERP_CUBIC)
			hipLaunchKernelGGL(hercules_cubic_kernel, dim3((total + 1u + 255u) / 256u), dim3(256), 0, s,
			                   (const f32x2 *)a->rf, (f32x4 *)q->pairs, total, (uint32_t)a->sample_count);
		else
			hipLaunchKernelGGL(hercules_pair_kernel, dim3((total + 2u + 255u) / 256u), dim3(256), 0, s,
			                   (const f32x2 *)a->rf, (f32x4 *)q->pairs, total, (uint32_t)a->sample_count);
	}
	uint32_t total = q->tiles[0] * q->tiles[1] * q->tiles[2];
	uint32_t grid  = q->depth_major == 3u ? bf_plane_walk_blocks(q->tiles[0], q->tiles[1], q->band_rows) : ((total + 7u) / 8u) * 8u;
	if (a->row_ends && INTERP != BF_INTERP_NEAREST) hipLaunchKernelGGL((das_hercules_kernel<INTERP, CPLX, CW, PL, PD, true>), dim3(grid), dim3(64u * q->rows), 0, s, *a, *q);
	else                                            hipLaunchKernelGGL((das_hercules_kernel<INTERP, CPLX, CW, PL, PD, false>), dim3(grid), dim3(64u * q->rows), 0, s, *a, *q);
	return hipGetLastError();
}

template <int INTERP, bool CW>
static hipError_t launch_herc_iq(const BfDasArgs *a, const BfHerculesArgs *q, hipStream_t s)
{
	if constexpr (INTERP != BF_INTERP_NEAREST) {
		if (q->pairs) return q->phase_local ? launch_herc<INTERP, true, CW, true, true>(a, q, s) : launch_herc<INTERP, true, CW, false, true>(a, q, s);
	}
	return q->phase_local ? launch_herc<INTERP, true, CW, true, false>(a, q, s) : launch_herc<INTERP, true, CW, false, false>(a, q, s);
}

template <int INTERP>
static hipError_t launch_herc_kind(const BfDasArgs *a, const BfHerculesArgs *q, hipStream_t s)
{
	if (a->complex_data) return a->coherency_weighting ? launch_herc_iq<INTERP, true>(a, q, s) : launch_herc_iq<INTERP, false>(a, q, s);
	return a->coherency_weighting ? launch_herc<INTERP, false, true, false, false>(a, q, s) : launch_herc<INTERP, false, false, false, false>(a, q, s);
}

extern "C" hipError_t bf_launch_das_hercules(const BfDasArgs *a, const BfHerculesArgs *q, hipStream_t s)
{
	switch (a->interpolation) {
	case BF_INTERP_NEAREST: return launch_herc_kind<BF_INTERP_NEAREST>(a, q, s);
	case BF_INTERP_LINEAR:  return launch_herc_kind<BF_INTERP_LINEAR>(a, q, s);
	case BF_INTERP_CUBIC:   return launch_herc_kind<BF_INTERP_CUBIC>(a, q, s);
	}
	return hipErrorInvalidValue;
}
