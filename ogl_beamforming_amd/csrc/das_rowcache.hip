/* das_rowcache.hip -- delay-and-sum for gfx950 (MI355X): the per-voxel factored kernel
 * (das_factored.hip) with the RF row segments a block can touch cached in LDS.
 *
 * das_factored.hip is bound by the L1 gather path: every tap of every (voxel, channel,
 * transmit) triple is a per-lane global load, 16 B (linear) or 32 B (cubic) per lane, and the
 * texture-address unit retires 64 B per clock per CU.  The 256 voxels of a block lie close
 * together, so for one (channel, transmit) row they touch a short run of samples: the sample
 * index is T(a) + R(ch), and over the block T spans [min T, max T] and R spans [min R, max R].
 * This kernel finds those four numbers by block reductions (no assumption about how the grid
 * lies), copies samples floor(min T) + floor(min R) - 1 ... + W of each row of a pass (CH
 * channels x AG transmits) into LDS with coalesced loads, and then serves the taps of the pass
 * from LDS (128 B per clock per CU, two or four 8-byte reads per triple).  Rows whose spread does
 * not fit the W-sample window -- steep geometries, huge tiles -- are gathered from global
 * memory exactly as das_factored.hip does, row by row (the decision is uniform over the block).
 *
 * Arithmetic per triple is das_factored.hip's: receive factors of a channel chunk in registers,
 * transmit phasors of a transmit group in registers, packed linear / Horner-cubic interpolation,
 * two packed FMAs of rotate-accumulate, |s| for coherency weighting.  Only the index is formed
 * as (T - floor(min T)) + (R - floor(min R)), which moves it by float rounding (~1e-4 samples).
 * IQ data, linear or cubic interpolation, RCA-family frames; everything else stays on
 * das_factored.hip / das.hip.
 *
 * STATUS: experiment, opt-in (beamformer_hip_set_das_path(5)).  Parity-tested and correct, but
 * slower than das_factored.hip as written: config 2 (cubic, no coherency weighting) 6.5 ms against
 * 3.6 ms, 64 planes of config 4 459 ms against 155 ms.  A pass (CH x AG = 32 triples per voxel)
 * is small next to its fixed costs -- two block barriers with the staging loads' latency between
 * them, the receive factors recomputed per transmit group -- and fully unrolling 32 triples with
 * both tap sources makes ~40 KB of straight-line code.  What a second attempt should change:
 * larger passes (more LDS per block or double-buffered windows), the global fallback per pass
 * instead of per row, transmit data that does not force the unroll.
 */
#include "das_common.h"

namespace {

constexpr int CH = 2;        /* channels per pass */
constexpr int AG = 16;       /* transmits per pass */
constexpr int W  = 64;       /* samples per cached row segment: one per lane of the staging wave */

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

/* a value every lane of the block holds identically (it came out of a block reduction): keep
 * it in a scalar register */
__device__ __forceinline__ float uniform(float v)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ float wave_min(float v)
{
	for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
	return v;
}
__device__ __forceinline__ float wave_max(float v)
{
	for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
	return v;
}

template <int INTERP, bool CW>
__global__ __launch_bounds__(256) void das_rowcache_kernel(const BfDasArgs p)
{
	constexpr uint32_t ES  = 8;
	constexpr int      OFS = INTERP == BF_INTERP_CUBIC ? 1 : 0;   /* first tap sits OFS samples below floor(index) */
	constexpr int      TAPS = INTERP == BF_INTERP_CUBIC ? 4 : 2;

	__shared__ f32x2 window[CH * AG][W];
	__shared__ f32x2 zeros[4];
	__shared__ float red_lo[4][5], red_hi[4][5];

	/* blockIdx -> tile and thread -> voxel exactly as das.hip */
	uint32_t total = p.blocks[0] * p.blocks[1] * p.blocks[2];
	uint32_t bid   = blockIdx.x;
	uint32_t per   = (total + 7u) / 8u;
	uint32_t tile  = (bid & 7u) * per + (bid >> 3);
	if (tile >= total) return;                                      /* whole block: no barrier is skipped */
	uint32_t bx = tile % p.blocks[0];
	uint32_t by = (tile / p.blocks[0]) % p.blocks[1];
	uint32_t bz = tile / (p.blocks[0] * p.blocks[1]);

	const uint32_t tid  = threadIdx.x;
	const uint32_t wave = tid >> 6;
	uint32_t lx = tid & ((1u << p.tile_shift[0]) - 1u);
	uint32_t ly = (tid >> p.tile_shift[0]) & ((1u << p.tile_shift[1]) - 1u);
	uint32_t lz = tid >> (p.tile_shift[0] + p.tile_shift[1]);
	uint32_t x = (bx << p.tile_shift[0]) + lx;
	uint32_t y = (by << p.tile_shift[1]) + ly;
	uint32_t zl = (bz << p.tile_shift[2]) + lz;
	const bool inside = x < p.size[0] && y < p.size[1] && zl < p.z_count;

	if (tid < 4) zeros[tid] = f32x2{0.f, 0.f};

	const uint32_t z = p.z_first + zl;
	float px = (float)x / fmaxf(1.0f, (float)p.size[0] - 1.0f);         /* das.glsl:374-376 */
	float py = (float)y / fmaxf(1.0f, (float)p.size[1] - 1.0f);
	float pz = (float)z / fmaxf(1.0f, (float)p.size[2] - 1.0f);
	float wx, wy, wz, xx, xy, xz;
	m4_point(p.voxel_transform, px, py, pz, wx, wy, wz);
	m4_point(p.xdc_transform, wx, wy, wz, xx, xy, xz);

	const char *rf = (const char *)p.rf;
	const int   S = p.sample_count, A = p.acquisition_count, C = p.channel_count;
	const float fs_over_c = p.sampling_frequency * p.inv_speed_of_sound;
	const float zz = xz * xz;
	const bool  rx_rows  = (p.transmits[0].flags & BF_RX_ROWS) != 0;
	const float lateral  = rx_rows ? xy : xx;
	const float pitch    = rx_rows ? p.pitch[1] : p.pitch[0];
	const float f_over_z = p.f_number * hw_rcp(__builtin_fabsf(xz));
	const float INF = __builtin_inff();

	f32x2 coherent = {0.f, 0.f};
	float incoherent = 0.f;

	/* Block-wide ranges of the five coordinates the indices depend on, once per block (wave
	 * shuffles + LDS).  Everything a pass needs -- where each row segment starts, how far it can
	 * spread -- follows from them in closed form, so the passes themselves need no reduction. */
	float g_lo[5], g_hi[5];
	{
		const float g[5] = {wx, wy, wz, lateral, __builtin_fabsf(xz)};
		#pragma unroll
		for (int i = 0; i < 5; i++) {
			float lo = wave_min(inside ? g[i] : INF), hi = wave_max(inside ? g[i] : -INF);
			if ((tid & 63u) == 0) { red_lo[wave][i] = lo; red_hi[wave][i] = hi; }
		}
		__syncthreads();
		#pragma unroll
		for (int i = 0; i < 5; i++) {
			g_lo[i] = uniform(fminf(fminf(red_lo[0][i], red_lo[1][i]), fminf(red_lo[2][i], red_lo[3][i])));
			g_hi[i] = uniform(fmaxf(fmaxf(red_hi[0][i], red_hi[1][i]), fmaxf(red_hi[2][i], red_hi[3][i])));
		}
	}
	/* distance from a point to an interval, nearest and farthest */
	auto nearest  = [](float lo, float hi, float q) { return fmaxf(0.f, fmaxf(lo - q, q - hi)); };
	auto farthest = [](float lo, float hi, float q) { return fmaxf(__builtin_fabsf(lo - q), __builtin_fabsf(hi - q)); };
	constexpr float MARGIN = 0.02f;                                /* float rounding of the bounds, in samples */

	for (int a0 = 0; a0 < A; a0 += AG) {
		/* ---- transmit group: index and phasor per voxel; [floor(min), max] of the index over the
		 * block from the coordinate ranges (the index is monotone in the path length) */
		float t_rel[AG], t_floor[AG], t_spread[AG];
		f32x2 cs[AG];
		#pragma unroll
		for (int j = 0; j < AG; j++) {
			float t_index = 0.f, d_lo = 0.f, d_hi = 0.f;
			if (a0 + j < A) {
				const BfTransmit t = p.transmits[a0 + j];
				t_index = (transmit_distance(t, wx, wy, wz) * p.inv_speed_of_sound + p.time_offset) * p.sampling_frequency;
				if (!(t.flags & BF_TX_NONE)) {
					const int   ax = (t.flags & BF_TX_ROWS) ? 1 : 0;
					if (t.flags & BF_TX_PLANE) {
						d_lo = fminf(g_lo[ax] * t.sin_a, g_hi[ax] * t.sin_a) + fminf(g_lo[2] * t.cos_a, g_hi[2] * t.cos_a);
						d_hi = fmaxf(g_lo[ax] * t.sin_a, g_hi[ax] * t.sin_a) + fmaxf(g_lo[2] * t.cos_a, g_hi[2] * t.cos_a);
					} else {
						float nx = nearest(g_lo[ax], g_hi[ax], t.focus_x), nz = nearest(g_lo[2], g_hi[2], t.focus_z);
						float fx = farthest(g_lo[ax], g_hi[ax], t.focus_x), fz = farthest(g_lo[2], g_hi[2], t.focus_z);
						d_lo = hw_sqrt(nx * nx + nz * nz); d_hi = hw_sqrt(fx * fx + fz * fz);
					}
				}
			}
			float turns = hw_fract(p.turns_per_sample * t_index);
			cs[j] = f32x2{hw_cos_turns(turns), hw_sin_turns(turns)};
			float lo = (d_lo * p.inv_speed_of_sound + p.time_offset) * p.sampling_frequency;
			float hi = (d_hi * p.inv_speed_of_sound + p.time_offset) * p.sampling_frequency;
			t_floor[j]  = uniform(__builtin_floorf(lo - MARGIN));
			t_spread[j] = uniform(hi + MARGIN - t_floor[j]);
			t_rel[j]    = t_index - t_floor[j];
		}

		for (int c0 = 0; c0 < C; c0 += CH) {
			/* ---- receive factors of the chunk (das_factored.hip); range of the receive index over
			 * the block from the lateral and depth ranges */
			float r_rel[CH], r_re[CH], r_im[CH], r_apod[CH], r_floor[CH], r_spread[CH], r_abs[CH];
			#pragma unroll
			for (int k = 0; k < CH; k++) {
				int   channel = c0 + k;
				float element = (float)channel * pitch;
				float dx      = lateral - element;
				float a_arg   = __builtin_fabsf(dx * f_over_z);
				bool  pass    = inside && a_arg < 0.5f && channel < C;
				float index   = hw_sqrt(dx * dx + zz) * fs_over_c;
				float apod    = pass ? apodize(a_arg) : 0.f;
				float turns   = hw_fract(p.turns_per_sample * index);
				r_abs[k]  = pass ? index : -1.0e9f;
				r_apod[k] = apod;
				r_re[k]   = apod * hw_cos_turns(turns);
				r_im[k]   = apod * hw_sin_turns(turns);
				float nx = nearest(g_lo[3], g_hi[3], element), fx = farthest(g_lo[3], g_hi[3], element);
				float lo = hw_sqrt(nx * nx + g_lo[4] * g_lo[4]) * fs_over_c, hi = hw_sqrt(fx * fx + g_hi[4] * g_hi[4]) * fs_over_c;
				r_floor[k]  = uniform(__builtin_floorf(lo - MARGIN));
				r_spread[k] = uniform(hi + MARGIN - r_floor[k]);
				r_rel[k]    = r_abs[k] - r_floor[k];
			}
			__syncthreads();                                           /* the previous pass has finished reading the windows */

			/* ---- stage the row segments of the pass: W samples from floor(min T) + floor(min R) - OFS.
			 * Rows are dealt to the four waves, one sample per lane (W == 64): coalesced loads */
			#pragma unroll
			for (int k = 0; k < CH; k++) {
				#pragma unroll
				for (int j = 0; j < AG; j++) {
					if (((k * AG + j) & 3) != (int)wave) continue;             /* wave-uniform */
					const bool cached = r_spread[k] + t_spread[j] + (float)(OFS + TAPS) < (float)W;
					int   channel = c0 + k, a = a0 + j;
					int   sample  = (int)r_floor[k] + (int)t_floor[j] - OFS + (int)(tid & 63u);
					f32x2 v = {0.f, 0.f};
					if (cached && channel < C && a < A && sample >= 0 && sample < S)
						v = gather<f32x2>(rf, (((uint32_t)channel * (uint32_t)A + (uint32_t)a) * (uint32_t)S + (uint32_t)sample) * ES);
					window[k * AG + j][tid & 63u] = v;
				}
			}
			__syncthreads();

			/* ---- the pass: CH x AG triples per voxel */
			f32x2 acc1[CH], acc2[CH];
			float mag[CH];
			#pragma unroll
			for (int k = 0; k < CH; k++) { acc1[k] = f32x2{0.f, 0.f}; acc2[k] = f32x2{0.f, 0.f}; mag[k] = 0.f; }

			#pragma unroll
			for (int j = 0; j < AG; j++) {
				if (a0 + j >= A) break;                                  /* uniform */
				#pragma unroll
				for (int k = 0; k < CH; k++) {
					const bool cached = r_spread[k] + t_spread[j] + (float)(OFS + TAPS) < (float)W;   /* block-uniform */
					f32x2 s0, s1, s2 = {0.f, 0.f}, s3 = {0.f, 0.f};
					float frac;
					if (cached) {
						float rel   = t_rel[j] + r_rel[k];                 /* index - floor(min T) - floor(min R) */
						frac        = hw_fract(rel);
						int   ki    = cvt_floor_i32(rel);
						int   k_abs = ki + (int)t_floor[j] + (int)r_floor[k];
						/* valid as in sample_rf: linear 0 <= floor < S-1, cubic 1 <= floor < S-2 */
						bool  valid = (uint32_t)(k_abs - OFS) < (uint32_t)(S - (INTERP == BF_INTERP_CUBIC ? 3 : 1));
						const f32x2 *src = valid ? &window[k * AG + j][ki] : zeros;   /* ki is the first tap: floor - OFS + OFS */
						s0 = src[0]; s1 = src[1];
						if constexpr (INTERP == BF_INTERP_CUBIC) { s2 = src[2]; s3 = src[3]; }
					} else {
						float index = (t_rel[j] + t_floor[j]) + r_abs[k];
						frac        = hw_fract(index);
						uint32_t ki = (uint32_t)(cvt_floor_i32(index) - OFS);
						uint32_t row = (((uint32_t)(c0 + k) * (uint32_t)A + (uint32_t)(a0 + j)) * (uint32_t)S) * ES;
						uint32_t off = ki < (uint32_t)(S - (INTERP == BF_INTERP_CUBIC ? 3 : 1)) && c0 + k < C ? row + (ki << 3) : p.zero_offset;
						f32x4 d0 = gather<f32x4_a8>(rf, off);
						s0 = f32x2{d0.x, d0.y}; s1 = f32x2{d0.z, d0.w};
						if constexpr (INTERP == BF_INTERP_CUBIC) {
							f32x4 d1 = gather<f32x4_a8>(rf, off + 16);
							s2 = f32x2{d1.x, d1.y}; s3 = f32x2{d1.z, d1.w};
						}
					}
					f32x2 sv;
					if constexpr (INTERP == BF_INTERP_LINEAR) {
						sv = s0 + frac * (s1 - s0);
					} else {
						/* Catmull-Rom as a Horner cubic (das_factored.hip) */
						f32x2 T1 = 0.5f * (s2 - s0), T2 = 0.5f * (s3 - s1), D = s2 - s1;
						f32x2 c3 = (T1 + T2) - 2.0f * D;
						f32x2 c2 = (D - T1) - c3;
						sv = s1 + frac * (T1 + frac * (c2 + frac * c3));
					}
					acc1[k] += sv.x * cs[j];
					acc2[k] += sv.y * cs[j];
					if constexpr (CW) { f32x2 sq = sv * sv; mag[k] += hw_sqrt(sq.x + sq.y); }
				}
			}
			#pragma unroll
			for (int k = 0; k < CH; k++) {
				f32x2 part = {acc1[k].x - acc2[k].y, acc1[k].y + acc2[k].x};
				coherent.x += r_re[k] * part.x - r_im[k] * part.y;
				coherent.y += r_im[k] * part.x + r_re[k] * part.y;
				if constexpr (CW) incoherent += r_apod[k] * mag[k];
			}
		}
	}

	if (inside) {
		uint64_t out_index = (uint64_t)p.size[0] * p.size[1] * zl + (uint64_t)p.size[0] * y + x;
		f32x2 v = coherent;
		if constexpr (CW) v = v * (v / incoherent);                      /* coherency_weighting.glsl:36 */
		reinterpret_cast<f32x2 *>(p.out)[out_index] = v;
	}
}

template <int INTERP>
hipError_t launch(const BfDasArgs *a, hipStream_t s)
{
	uint32_t total = a->blocks[0] * a->blocks[1] * a->blocks[2];
	uint32_t grid  = ((total + 7u) / 8u) * 8u;
	if (a->coherency_weighting) hipLaunchKernelGGL((das_rowcache_kernel<INTERP, true>),  dim3(grid), dim3(256), 0, s, *a);
	else                        hipLaunchKernelGGL((das_rowcache_kernel<INTERP, false>), dim3(grid), dim3(256), 0, s, *a);
	return hipGetLastError();
}

} // namespace

/* RCA-family IQ frames with linear or cubic interpolation, one receive orientation for all
 * transmits, 256-voxel tiles (no channel split): the caller has checked all of that and has set
 * zero_offset. */
extern "C" hipError_t bf_launch_das_rowcache(const BfDasArgs *a, hipStream_t s)
{
	if (a->family != BF_DAS_RCA || !a->complex_data || a->split_shift) return hipErrorInvalidValue;
	switch (a->interpolation) {
	case BF_INTERP_LINEAR: return launch<BF_INTERP_LINEAR>(a, s);
	case BF_INTERP_CUBIC:  return launch<BF_INTERP_CUBIC>(a, s);
	}
	return hipErrorInvalidValue;
}
