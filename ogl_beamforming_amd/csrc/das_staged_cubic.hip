/* das_staged_cubic.hip -- the LDS-staged row-column DAS kernel (das_staged.hip) for CUBIC interpolation.
 *
 * The reference's own throughput harness beamforms with Catmull-Rom cubic interpolation (tests/throughput.c:451;
 * shaders/das.glsl:67-97: Hermite basis over taps k-1 .. k+2, tangents (P2 - P0)/2 and (P3 - P1)/2, valid for
 * 1 <= index < S - 2).  Through L1 that is 32 bytes of gathers per (voxel, channel, transmit) term -- two wave64 gather
 * instructions, 32.6 clk per CU -- plus the four Hermite weights; here the interpolant of every window segment is a
 * cubic POLYNOMIAL whose four complex coefficients the staging threads compute once per window element and channel:
 *   * window element j holds {b0, b1, b2, b3}, the segment between window samples j and j + 1 expanded around its
 *     MIDDLE: value(g) = ((b3 g + b2) g + b1) g + b0 for g in [-1/2, 1/2] -- 32 bytes, two aligned ds_read_b128 and
 *     three packed fmas per term (around the middle, not in window coordinates: p^3 would cost 3 digits);
 *   * everything else is das_staged.hip: window position from the float tables (T'' = T - floor(tmin) + 1/2 here: the
 *     window starts one sample early for tap k - 1), magic-number rounding y = p + M for the element index and one
 *     v_mul_u32_u24 for its LDS address, g = p - (y - M) as two packed adds per pair of terms, transmit tables in
 *     pairs, the per-lane range flag in the sign of the receive weight (1 <= index < S - 2 for every transmit of the
 *     tile), buffer-load staging with the next channel's windows in flight; the neighbours a staging thread needs
 *     (samples j - 1, j + 1, j + 2) come from its neighbour lanes by one-lane wave shifts.
 * 78 KB of windows per block at 76 transmits: one 1024-thread block per CU, 128 VGPRs per lane.
 */
#include "das_common.h"

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef __attribute__((address_space(3))) f32x3 lds_f32x3;

__device__ __forceinline__ float cubic_phase_turns(float k, float index)
{
	float p = k * index;
	float e = __builtin_fmaf(k, index, -p);
	return hw_fract(p) + e;
}

/* LDS (A4 = transmits rounded up to a multiple of 4; transmits in PAIRS):
 *   stage[a*W + j]   = { b0, b1, b2, b3 }: the Catmull-Rom segment between window samples j and j + 1 of window (c, a) as a
 *                      cubic around the segment's middle; window sample j = sample floor(rmin_c) + floor(tmin_a) - 1 + j;
 *                      valid for 1 <= j <= W - 3; two unused elements in front, one zero element behind      2 x f32x4
 *   Tcs, R, Tz, tfl, rfloor, wave_range: as das_staged.hip (Tz holds T'' = t_index - floor(tmin_a) + 1/2) */
/* NL: window elements a thread stages per channel, ceil(A4 * W / threads) */
template <bool CW, int VS, int WS, int NL>
__global__ __launch_bounds__(1024, 4) void das_rca_staged_cubic_kernel(const BfDasArgs p, const BfSeparableArgs q)
{
	extern __shared__ __attribute__((aligned(16))) f32x4 staged_cubic_lds[];
	constexpr uint32_t V = 1u << VS, W = 1u << WS;
	const uint32_t U = 1u << q.u_shift;
	const int C = p.channel_count, A = p.acquisition_count, S = p.sample_count;
	const int A4 = (A + 3) & ~3;
	const int chunk = (int)q.channel_chunk;
	/* the staging area comes first and the kernel has no static LDS: 16 x (a window element's index + 2) IS its LDS
	 * address, which the inner loop forms with one shift */
	f32x4 *stage  = staged_cubic_lds + 4;                          /* (two unused 32-byte elements in front: see the rounding of the inner loop) */
	f32x4 *Tcs    = stage + 2 * ((size_t)A4 * W + 1);
	f32x4 *R      = Tcs + (size_t)(A4 / 2) * V;
	f32x2 *Tz     = reinterpret_cast<f32x2 *>(R + (size_t)chunk * U);
	int   *tfl    = reinterpret_cast<int *>(Tz + (size_t)(A4 / 2) * V);
	int   *rfloor = tfl + A4;
	f32x2 *wave_range = reinterpret_cast<f32x2 *>(rfloor + ((chunk + 1) & ~1));      /* 16 entries, 8-byte aligned */
	const uint32_t stage_elements = (uint32_t)A4 * W;

	const uint32_t total = q.tiles[0] * q.tiles[1] * q.tiles[2];
	const uint32_t per   = (total + 7u) / 8u;
	const uint32_t tile  = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
	if (tile >= total) return;                               /* whole block */
	uint32_t tu, tv, zl;                                     /* walk order: das_separable.hip */
	if (q.depth_major & 1u) {
		bf_column_walk(tile, q.tiles[0], q.tiles[2], q.walk_columns, tu, tv, zl);
	} else {
		tu = tile % q.tiles[0];
		tv = (tile / q.tiles[0]) % q.tiles[1];
		zl = tile / (q.tiles[0] * q.tiles[1]);
	}
	const uint32_t z  = p.z_first + zl;

	const uint32_t u_axis = q.u_axis, v_axis = 1u - q.u_axis;
	const float denom[3] = {fmaxf(1.0f, (float)p.size[0] - 1.0f), fmaxf(1.0f, (float)p.size[1] - 1.0f),
	                        fmaxf(1.0f, (float)p.size[2] - 1.0f)};
	const float pz = (float)z / denom[2];
	const float phase_k = p.demodulation_frequency * p.inv_sampling_frequency;
	const BfTransmit t0 = p.transmits[0];
	const bool  rx_rows = (t0.flags & BF_RX_ROWS) != 0;
	[[maybe_unused]] const float rx_pitch = rx_rows ? p.pitch[1] : p.pitch[0];
	const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
	if (q.depth_major & 2u) staged_violation_clear(tid);       /* STAGED_CHECKED: das_common.h */

	/* ---- transmit tables (absolute delays first) */
	for (uint32_t e = tid; e < (uint32_t)A4 * V; e += nthreads) {
		uint32_t a = e >> VS, iv = e & (V - 1);
		float cs_c = 0.f, cs_s = 0.f, t_idx = 0.f;           /* padding transmits: zero phasor, window position 0 */
		if (a < (uint32_t)A) {
			float coord[3] = {0.f, 0.f, pz};
			coord[v_axis] = (float)(tv * V + iv) / denom[v_axis];
			float wx, wy, wz;
			m4_point(p.voxel_transform, coord[0], coord[1], coord[2], wx, wy, wz);
			const BfTransmit t = p.transmits[a];
			float dist = 0.f;
			if (!(t.flags & BF_TX_NONE)) {
				float px = (t.flags & BF_TX_ROWS) ? wy : wx;
				if (t.flags & BF_TX_PLANE) dist = px * t.sin_a + wz * t.cos_a;
				else { float ddx = px - t.focus_x, ddz = wz - t.focus_z; dist = hw_sqrt(ddx * ddx + ddz * ddz); }
			}
			t_idx = (div_speed_of_sound(dist, p) + p.time_offset) * p.sampling_frequency;
			float turns = cubic_phase_turns(phase_k, t_idx);
			cs_c = hw_cos_turns(turns); cs_s = hw_sin_turns(turns);
		}
		const uint32_t pair = (a >> 1) * V + iv, half = a & 1u;
		reinterpret_cast<f32x2 *>(Tcs + pair)[half] = f32x2{cs_c, cs_s};
		reinterpret_cast<float *>(Tz + pair)[half]  = t_idx;
	}
	if (tid < 2) stage[2 * stage_elements + tid] = f32x4{0.f, 0.f, 0.f, 0.f};
	/* tile-wide extremes of the absolute transmit delay (range-test shortcut, as das_separable.hip) */
	__syncthreads();
	{
		float lo = __builtin_inff(), hi = -__builtin_inff();
		for (uint32_t e = tid; e < (uint32_t)A * V; e += nthreads) {
			uint32_t a = e >> VS, iv = e & (V - 1);
			float v = reinterpret_cast<const float *>(Tz + (a >> 1) * V + iv)[a & 1u];
			lo = fminf(lo, v); hi = fmaxf(hi, v);
		}
		for (int off = 32; off > 0; off >>= 1) {
			lo = fminf(lo, __shfl_xor(lo, off, 64));
			hi = fmaxf(hi, __shfl_xor(hi, off, 64));
		}
		if ((tid & 63u) == 0) wave_range[tid >> 6] = f32x2{lo, hi};
	}
	__syncthreads();
	f32x2 range = wave_range[0];
	for (uint32_t w = 1; w < (nthreads >> 6); w++) {
		range.x = fminf(range.x, wave_range[w].x);
		range.y = fmaxf(range.y, wave_range[w].y);
	}
	/* the same for every lane: keep it in scalar registers.  (Through scalar temporaries: __builtin_bit_cast applied
	 * directly to a vector component reads the vector's FIRST component with this hipcc -- range.y silently became
	 * range.x, and waves whose lanes reach the end of the RF row for the tile's largest transmit delay only took the
	 * unchecked loop; found by the focused-transmit parity case, whose delays differ by hundreds of samples.) */
	{
		const float lo = range.x, hi = range.y;
		range.x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lo)));
		range.y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hi)));
	}
	/* per transmit: floor of the smallest delay of its table row; the row becomes window-relative */
	for (uint32_t a = tid; a < (uint32_t)A4; a += nthreads) {
		float *row = reinterpret_cast<float *>(Tz + (size_t)(a >> 1) * V) + (a & 1u);
		float  m   = row[0];
		#pragma unroll 4
		for (uint32_t iv = 1; iv < V; iv++) m = fminf(m, row[2 * iv]);
		float fl = __builtin_floorf(m);
		#pragma unroll 4
		for (uint32_t iv = 0; iv < V; iv++) row[2 * iv] = (row[2 * iv] - fl) + 0.5f;      /* both steps exact */
		tfl[a] = (int)fl;
	}
	__syncthreads();                                         /* the floors are read below */

	uint32_t lu, lv;
	if (u_axis == 0) { lu = tid & (U - 1); lv = tid >> q.u_shift; }
	else             { lv = tid & (V - 1); lu = tid >> VS; }
	const uint32_t gu = tu * U + lu, gv = tv * V + lv;
	const uint32_t x = u_axis == 0 ? gu : gv, y = u_axis == 0 ? gv : gu;
	const bool inside = x < p.size[0] && y < p.size[1];

	f32x2 coherent   = {0.f, 0.f};
	float incoherent = 0.f;
	const f32x4   *Rl = R + lu;
	/* LDS byte addresses */
	uint32_t tcs_base = (uint32_t)(uintptr_t)(lds_f32x4 *)Tcs;
	uint32_t tz_base  = (uint32_t)(uintptr_t)(lds_f32x2 *)Tz;
	/* opaque to the compiler: otherwise the static LDS in front of the dynamic block is re-added as a constant
	 * to every address of the inner loop instead of once here */
	asm("" : "+s"(tcs_base), "+s"(tz_base));

	/* Staging.  Thread tid copies element j = tid % W of windows a_n = tid / W + n * (threads / W), n < NL:
	 * sample rfl + floor(tmin_a) + j of row (channel, a).  The loads are buffer loads over the whole DAS
	 * input: an offset outside it (a window that starts before the first row or ends behind the last)
	 * returns zero instead of faulting, and samples a window holds from a NEIGHBOURING row are never
	 * consumed -- a term is only evaluated (unchecked loop) or only kept (checked loop) when both of
	 * its taps lie inside its own row.  Per thread and n one loop-invariant byte offset; per channel one add. */
	const __amdgpu_buffer_rsrc_t rf_rsrc = __builtin_amdgcn_make_buffer_rsrc(
		const_cast<void *>(p.rf), 0, (int)((uint32_t)C * (uint32_t)A * (uint32_t)S * 8u), 0x00020000);
	const uint32_t windows_per_pass = nthreads >> WS;
	uint32_t stage_inv[NL];
	#pragma unroll
	for (int n = 0; n < NL; n++) {
		uint32_t a = (tid >> WS) + (uint32_t)n * windows_per_pass;
		/* transmits of the padding (a >= A) point far outside the buffer: they stage zeros */
		stage_inv[n] = a < (uint32_t)A ? (a * (uint32_t)S + (uint32_t)(tfl[a] - 1 + (int)(tid & (W - 1)))) * 8u : 0x80000000u;
	}
	auto stage_load = [&](int channel, int rfl, f32x2 (&regs)[NL]) {
		const uint32_t at = ((uint32_t)channel * (uint32_t)A * (uint32_t)S + (uint32_t)rfl) * 8u;
		#pragma unroll
		for (int n = 0; n < NL; n++) {
			/* (the padding's 0x80000000 + at stays out of range: the host refuses inputs of 2 GiB and more here) */
			i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rf_rsrc, (int)(stage_inv[n] + at), 0, 0);
			regs[n] = __builtin_bit_cast(f32x2, v);
		}
	};
	/* Element j keeps the Catmull-Rom segment between window samples j and j + 1 (das.glsl:67-97: a0 = P1, a1 = T1,
	 * a2 = 3 (P2 - P1) - 2 T1 - T2, a3 = 2 (P1 - P2) + T1 + T2 in the local coordinate t, tangents T1 = (P2 - P0) / 2,
	 * T2 = (P3 - P1) / 2) re-expanded around t = 1/2.  P0, P2, P3 sit in the neighbouring lanes (a wave stages whole
	 * windows, consecutive lanes consecutive samples); the first and the last two elements of a window get meaningless
	 * coefficients and are never selected (the host's window bound, plan_staged). */
	auto lane_shift = [](float v, bool up) {
		return up ? __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true))    /* lane i <- i + 1 */
		          : __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));   /* lane i <- i - 1 */
	};
	auto stage_store = [&](const f32x2 (&regs)[NL]) {
		#pragma unroll
		for (int n = 0; n < NL; n++) {
			const float p1x = regs[n].x, p1y = regs[n].y;
			const float p2x = lane_shift(p1x, true),  p2y = lane_shift(p1y, true);
			const float p3x = lane_shift(p2x, true),  p3y = lane_shift(p2y, true);
			const float p0x = lane_shift(p1x, false), p0y = lane_shift(p1y, false);
			const f32x2 P0 = {p0x, p0y}, P1 = {p1x, p1y}, P2 = {p2x, p2y}, P3 = {p3x, p3y};
			const f32x2 T1 = 0.5f * (P2 - P0), T2 = 0.5f * (P3 - P1);
			const f32x2 a2 = 3.f * (P2 - P1) - 2.f * T1 - T2, a3 = 2.f * (P1 - P2) + T1 + T2;
			const f32x2 b0 = P1 + 0.5f * T1 + 0.25f * a2 + 0.125f * a3;
			const f32x2 b1 = T1 + a2 + 0.75f * a3;
			const f32x2 b2 = a2 + 1.5f * a3;
			uint32_t e = tid + (uint32_t)n * nthreads;
			if (e < stage_elements) {
				stage[2 * e]     = f32x4{b0.x, b0.y, b1.x, b1.y};
				stage[2 * e + 1] = f32x4{b2.x, b2.y, a3.x, a3.y};
			}
		}
	};

	for (int c0 = 0; c0 < C; c0 += chunk) {
		const int cn = (C - c0) < chunk ? (C - c0) : chunk;
		__syncthreads();        /* readers of the previous chunk's R / stage are done; the transmit tables are complete */
		{
		/* the ~45 scalars of the receive-table build come from the kernel-argument segment at the top of every chunk instead of
		 * living in SGPRs across the channel loop (das_staged.hip) */
		typedef __attribute__((address_space(4))) const BfDasArgs const_args;
		const_args *ka = (const_args *)__builtin_amdgcn_kernarg_segment_ptr();
		asm volatile("" : "+s"(ka));
		const float k_denom_u = fmaxf(1.0f, (float)ka->size[u_axis] - 1.0f);
		const float k_pz = (float)z / fmaxf(1.0f, (float)ka->size[2] - 1.0f);
		const float k_fs = ka->sampling_frequency, k_inv_c = ka->inv_speed_of_sound, k_c = ka->speed_of_sound, k_fnum = ka->f_number;
		const float k_phase = ka->demodulation_frequency * ka->inv_sampling_frequency;
		const float k_pitch = rx_rows ? ka->pitch[1] : ka->pitch[0];
		for (uint32_t e = tid; e < (uint32_t)cn * U; e += nthreads) {
			uint32_t c = (uint32_t)c0 + (e >> q.u_shift), iu = e & (U - 1);
			float coord[3] = {0.f, 0.f, k_pz};
			coord[u_axis] = (float)(tu * U + iu) / k_denom_u;
			float wx, wy, wz, xx, xy, xz;
			m4_point(ka->voxel_transform, coord[0], coord[1], coord[2], wx, wy, wz);
			m4_point(ka->xdc_transform, wx, wy, wz, xx, xy, xz);
			float lateral = rx_rows ? xy : xx;
			float dx      = lateral - (float)c * k_pitch;
			float a_arg   = __builtin_fabsf(dx * (k_fnum * hw_rcp(__builtin_fabsf(xz))));
			/* the delay is kept for lanes outside the aperture too: it keeps their (discarded)
			 * LDS reads inside the window */
			float r_idx = div_speed_of_sound(hw_sqrt(dx * dx + xz * xz), k_inv_c, k_c) * k_fs;
			f32x4 entry = {r_idx, 0.f, 0.f, 0.f};
			if (a_arg < 0.5f) {
				float cs    = hw_cos_turns(0.5f * a_arg);
				float apod  = cs * cs;
				float turns = cubic_phase_turns(k_phase, r_idx);
				entry.y = apod * hw_cos_turns(turns);
				entry.z = apod * hw_sin_turns(turns);
				entry.w = apod;
			}
			R[e] = entry;
		}
		}
		__syncthreads();
		for (uint32_t cl = tid; cl < (uint32_t)cn; cl += nthreads) {
			const float *row = reinterpret_cast<const float *>(R + (size_t)cl * U);
			float m = row[0];
			#pragma unroll 4
			for (uint32_t iu = 1; iu < U; iu++) m = fminf(m, row[4 * iu]);
			rfloor[cl] = (int)__builtin_floorf(m);
		}
		__syncthreads();
		/* the entries become what the channel loop consumes with no arithmetic: the delay relative to the channel's window
		 * (exact) and, in the SIGN of the weight, whether the lane can leave the RF row for some transmit of the tile
		 * (r + min T < 0 or r + max T >= S - 1: such a wave runs the checked loop) */
		for (uint32_t e = tid; e < (uint32_t)cn * U; e += nthreads) {
			f32x4 entry = R[e];
			const bool lane_safe = (entry.x + range.x >= 1.f) && (entry.x + range.y < (float)(S - 2));      /* das.glsl: 1 <= index < S - 2 */
			entry.x -= (float)rfloor[e >> q.u_shift];
			if (!lane_safe) entry.w = -entry.w;          /* -0.0f for a lane outside the aperture: still "unsafe" to the sign test */
			R[e] = entry;
		}
		__syncthreads();

		f32x2 regs[NL];
		stage_load(c0, rfloor[0], regs);
		for (int cl = 0; cl < cn; cl++) {
			__syncthreads();                   /* everyone is done with the previous channel's windows */
			stage_store(regs);
			__syncthreads();
			if (cl + 1 < cn) stage_load(c0 + cl + 1, rfloor[cl + 1], regs);   /* in flight during the arithmetic */
			if (!inside) continue;

			/* (register budget: 64 per lane at 8 waves per SIMD with the next channel's windows in flight.  The
			 * receive entry is read twice -- delay and aperture test here, phasor and weight after the loop -- and the
			 * lane's table addresses are rebuilt per channel rather than kept) */
			float r_rel, r_w;
			{
				const f32x4 r = Rl[(size_t)cl * U];
				r_rel = r.x; r_w = r.w;
			}
			if (__builtin_amdgcn_ballot_w64(r_w != 0.f) == 0) continue;    /* F# culling per wave */
			const bool wave_safe = !(q.depth_major & 2u) && __builtin_amdgcn_ballot_w64(__builtin_signbitf(r_w)) == 0;   /* bit 1: test hook, checked loop everywhere */
			f32x2 acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f};
			f32x2 mag2 = {0.f, 0.f};
			/* one term: g in [-1/2, 1/2] = position relative to the middle of the selected segment, lo / hi = its coefficients */
			auto term = [&](f32x2 cs, float g, f32x4 lo, f32x4 hi) -> float {
				f32x2 sv = f32x2{hi.z, hi.w} * g + f32x2{hi.x, hi.y};
				sv = sv * g + f32x2{lo.z, lo.w};
				sv = sv * g + f32x2{lo.x, lo.y};
				acc1 += sv.x * cs;
				acc2 += sv.y * cs;
				if constexpr (CW) return hw_sqrt(__builtin_fmaf(sv.y, sv.y, sv.x * sv.x));
				else return 0.f;
			};
			auto batches = [&](auto checked) {
				constexpr bool CHECK = decltype(checked)::value;
				uint32_t lane_id = tid;
				asm volatile("" : "+v"(lane_id));
				const uint32_t lane_v = u_axis == 0 ? lane_id >> q.u_shift : lane_id & (V - 1);
				uint32_t tcs_at = tcs_base + (lane_v << 4), tz_at = tz_base + (lane_v << 3);
				/* y = p + M, M = 2^23 + 2 + a*W: the element index 2 + a*W + round(p) in the low mantissa bits (das_staged.hip
				 * explains the bias and the ties); the element's LDS byte address is (bits(y) & 0xFFFFFF) * 32; g = p - (y - M). */
				uint32_t m_bits = 0x4B000002u;
				[[maybe_unused]] bool window_left = false;    /* range-checked loop: some term selected an element outside its window */
				const f32x2 rr = {r_rel, r_rel};
				for (int a = 0; a < A4; a += 4, tcs_at += 2u * V * 16u, tz_at += 2u * V * 8u, m_bits += 4u * W) {
					uint32_t at[4]; f32x4 lo[4], hi[4];
					const float M = __builtin_bit_cast(float, m_bits);
					const f32x2 M2 = {M, M};
					const f32x4 cs01 = *(lds_f32x4 *)(uintptr_t)tcs_at;
					const f32x4 cs23 = *(lds_f32x4 *)(uintptr_t)(tcs_at + V * 16u);
					const f32x2 tz01 = *(lds_f32x2 *)(uintptr_t)tz_at;
					const f32x2 tz23 = *(lds_f32x2 *)(uintptr_t)(tz_at + V * 8u);
					const f32x2 p01 = rr + tz01, p23 = rr + tz23;
					const f32x2 y01 = p01 + M2,  y23 = p23 + M2;
					const f32x2 g01 = p01 - (y01 - M2), g23 = p23 - (y23 - M2);
					const float ys[4] = {y01.x, y01.y, y23.x, y23.y};
					#pragma unroll
					for (int k = 0; k < 4; k++) {
						const uint32_t yb = __builtin_bit_cast(uint32_t, ys[k]);
						asm("v_mul_u32_u24 %0, 32, %1" : "=v"(at[k]) : "v"(yb));
						if constexpr (CHECK) {
							/* segment n = round(p) is window sample n: absolute tap k = n - 1 + the two floors; valid for 1 <= k < S - 2 */
							uint32_t k_abs = (uint32_t)((int)(yb - m_bits) - 1 + rfloor[cl] + tfl[a + k]);
							at[k] = (k_abs - 1u) < (uint32_t)(S - 3) ? at[k] + (uint32_t)k * W * 32u : (stage_elements + 2u) * 32u;
							window_left |= __builtin_amdgcn_ballot_w64((yb - m_bits) - 1u > W - 4u) != 0ull;      /* (wave uniform: a scalar) never, unless plan_staged's bound is wrong */
						}
					}
					#pragma unroll
					for (int k = 0; k < 4; k++) {
						const uint32_t row_k = CHECK ? 0u : (uint32_t)k * W * 32u;       /* immediate */
						lo[k] = *(lds_f32x4 *)(uintptr_t)(at[k] + row_k);
						hi[k] = *(lds_f32x4 *)(uintptr_t)(at[k] + row_k + 16u);
					}
					const float q0 = term(f32x2{cs01.x, cs01.y}, g01.x, lo[0], hi[0]);
					const float q1 = term(f32x2{cs01.z, cs01.w}, g01.y, lo[1], hi[1]);
					const float q2 = term(f32x2{cs23.x, cs23.y}, g23.x, lo[2], hi[2]);
					const float q3 = term(f32x2{cs23.z, cs23.w}, g23.y, lo[3], hi[3]);
					if constexpr (CW) { mag2 += f32x2{q0, q1}; mag2 += f32x2{q2, q3}; }
				}
				if constexpr (CHECK) { if (window_left) staged_violation_raise(); }
			};
			if (wave_safe) batches(std::false_type{});
			else           batches(std::true_type{});
			/* per-channel fold, written scalar (hipcc otherwise builds it from packed ops and six register moves) */
			float sum_x = acc1.x - acc2.y, sum_y = acc1.y + acc2.x;
			asm volatile("" : "+v"(sum_x), "+v"(sum_y));
			const f32x4 r = *(volatile lds_f32x4 *)(uintptr_t)((uint32_t)(uintptr_t)(lds_f32x4 *)Rl + (uint32_t)cl * U * 16u);
			coherent.x = __builtin_fmaf(sum_x, r.y, __builtin_fmaf(-sum_y, r.z, coherent.x));
			coherent.y = __builtin_fmaf(sum_x, r.z, __builtin_fmaf(sum_y, r.y, coherent.y));
			if constexpr (CW) incoherent = __builtin_fmaf(__builtin_fabsf(r.w), mag2.x + mag2.y, incoherent);
		}
	}
	if (q.depth_major & 2u) staged_violation_report(tid);      /* (block uniform: every thread reaches it) */
	if (!inside) return;

	uint64_t out_index = (uint64_t)p.size[0] * p.size[1] * zl + (uint64_t)p.size[0] * y + x;
	if constexpr (CW) coherent = coherent * (coherent / incoherent);   /* coherency_weighting.glsl:36 */
	reinterpret_cast<f32x2 *>(p.out)[out_index] = coherent;
}

template <bool CW, int VS, int WS, int NL>
static hipError_t launch_cubic(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	uint32_t total = q->tiles[0] * q->tiles[1] * q->tiles[2];
	uint32_t grid  = ((total + 7u) / 8u) * 8u;
	auto kernel = das_rca_staged_cubic_kernel<CW, VS, WS, NL>;
	hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q->lds_bytes);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(kernel, dim3(grid), dim3(q->threads), q->lds_bytes, s, *a, *q);
	return hipGetLastError();
}

template <bool CW, int VS, int WS>
static hipError_t launch_cubic_loads(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	const uint32_t A4 = ((uint32_t)a->acquisition_count + 3u) & ~3u;
	switch (((A4 << WS) + q->threads - 1) / q->threads) {
	case 1: return launch_cubic<CW, VS, WS, 1>(a, q, s);
	case 2: return launch_cubic<CW, VS, WS, 2>(a, q, s);
	case 3: return launch_cubic<CW, VS, WS, 3>(a, q, s);
	case 4: return launch_cubic<CW, VS, WS, 4>(a, q, s);
	}
	return hipErrorInvalidValue;
}

template <bool CW>
static hipError_t launch_cubic_shape(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	switch ((q->v_shift << 4) | q->window_shift) {
	case (4 << 4) | 5: return launch_cubic_loads<CW, 4, 5>(a, q, s);
	case (5 << 4) | 5: return launch_cubic_loads<CW, 5, 5>(a, q, s);
	case (6 << 4) | 5: return launch_cubic_loads<CW, 6, 5>(a, q, s);
	case (4 << 4) | 6: return launch_cubic_loads<CW, 4, 6>(a, q, s);
	case (5 << 4) | 6: return launch_cubic_loads<CW, 5, 6>(a, q, s);
	case (6 << 4) | 6: return launch_cubic_loads<CW, 6, 6>(a, q, s);
	}
	return hipErrorInvalidValue;
}

/* complex samples, cubic interpolation; the caller checked q->window_shift */
extern "C" hipError_t bf_launch_das_staged_cubic(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	if (!a->complex_data || a->interpolation != BF_INTERP_CUBIC) return hipErrorInvalidValue;
	/* the staging loads address the DAS input through 32-bit buffer offsets with out-of-range padding at 2^31 */
	if ((uint64_t)a->channel_count * (uint64_t)a->acquisition_count * (uint64_t)a->sample_count * 8u >= (1ull << 31)) return hipErrorInvalidValue;
	return a->coherency_weighting ? launch_cubic_shape<true>(a, q, s) : launch_cubic_shape<false>(a, q, s);
}
