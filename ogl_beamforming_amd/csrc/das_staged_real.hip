/* das_staged_real.hip -- the LDS-staged row-column DAS kernel (das_staged.hip) for REAL samples.
 *
 * Same design, half the data: pipelines without Demodulate (e.g. {Decode, DAS} on Int16 RF) hand DAS
 * real float samples, sample_rf has no IQ rotation (shaders/das.glsl:99-124 with SAMPLE_TYPE float), and
 * the coherency weight sums |sample| (das.glsl:29, length() of a scalar).  Everything das_staged.hip's
 * header explains applies -- window position in the float tables, window elements stored as lines
 * {c_j, d_j} in window coordinates (8 bytes here: ONE ds_read_b64 per term and the interpolation one fma
 * of the position itself), magic-number rounding for the tap address, transmit delays in pairs, the
 * per-lane range flag in the sign of the receive weight, buffer-load staging with the next channel's
 * windows in flight -- minus the phasor tables and the complex multiply-accumulate.  Per term the inner
 * loop is: half a packed add (position), half a packed add (rounding), one v_lshlrev_b16 (address), one
 * fma (interpolation), one add (sum), and with coherency weighting one add of |sample| (a free modifier).
 * The gather kernel (das_separable.hip) pays 16.3 clk per wave64 gather for the same term.
 */
#include "das_common.h"

typedef __attribute__((address_space(3))) f32x2 lds_f32x2;

/* LDS (A4 = transmits rounded up to a multiple of 4):
 *   stage[a*W + j]   = { c_j, d_j }: the line through samples j and j + 1 of window (c, a) in window coordinates;
 *                      j < W, a < A4; two unused elements in front, one zero element behind          f32x2
 *   R[cl*U + u]      = { R' = r_index - floor(rmin_c), +-apod }   (sign bit set: the lane may leave the RF row)   f32x2
 *   Tz[(a/2)*V + v]  = { T'' = t_index - floor(tmin_a) - 1/2 of transmit a & ~1, of transmit a | 1 }   f32x2
 *   tfl[a] = floor(tmin_a),  rfloor[cl] = floor(rmin_c)                                                int
 *   wave_range[16]                                                                                     f32x2 */
template <bool CW, int VS, int WS, int NL>
__global__ __launch_bounds__(1024, 8) void das_rca_staged_real_kernel(const BfDasArgs p, const BfSeparableArgs q)
{
	extern __shared__ __attribute__((aligned(16))) f32x2 staged_real_lds[];
	constexpr uint32_t V = 1u << VS, W = 1u << WS;
	const uint32_t U = 1u << q.u_shift;
	const int C = p.channel_count, A = p.acquisition_count, S = p.sample_count;
	const int A4 = (A + 3) & ~3;
	const int chunk = (int)q.channel_chunk;
	/* the staging area comes first and the kernel has no static LDS: 8 x (a window element's index + 2) IS its LDS address */
	f32x2 *stage  = staged_real_lds + 2;
	f32x2 *R      = stage + (size_t)A4 * W + 2;              /* (+ the zero element, + one to keep 16-byte alignment) */
	f32x2 *Tz     = R + (size_t)chunk * U;
	int   *tfl    = reinterpret_cast<int *>(Tz + (size_t)(A4 / 2) * V);
	int   *rfloor = tfl + A4;
	f32x2 *wave_range = reinterpret_cast<f32x2 *>(rfloor + ((chunk + 1) & ~1));
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
	const BfTransmit t0 = p.transmits[0];
	const bool  rx_rows = (t0.flags & BF_RX_ROWS) != 0;
	[[maybe_unused]] const float rx_pitch = rx_rows ? p.pitch[1] : p.pitch[0];
	const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
	if (q.depth_major & 2u) staged_violation_clear(tid);       /* STAGED_CHECKED: das_common.h */

	/* ---- transmit delays (absolute first) */
	for (uint32_t e = tid; e < (uint32_t)A4 * V; e += nthreads) {
		uint32_t a = e >> VS, iv = e & (V - 1);
		float t_idx = 0.f;                                   /* padding transmits: window position 0 over a zero row */
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
		}
		reinterpret_cast<float *>(Tz + (a >> 1) * V + iv)[a & 1u] = t_idx;
	}
	if (tid == 0) stage[stage_elements] = f32x2{0.f, 0.f};
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
	float range_lo, range_hi;
	{
		/* (scalar temporaries throughout: __builtin_bit_cast on a vector component reads component 0 with this hipcc) */
		float lo = __builtin_inff(), hi = -__builtin_inff();
		for (uint32_t w = 0; w < (nthreads >> 6); w++) {
			const f32x2 r = wave_range[w];
			const float rl = r.x, rh = r.y;
			lo = fminf(lo, rl); hi = fmaxf(hi, rh);
		}
		range_lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lo)));
		range_hi = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hi)));
	}
	for (uint32_t a = tid; a < (uint32_t)A4; a += nthreads) {
		float *row = reinterpret_cast<float *>(Tz + (size_t)(a >> 1) * V) + (a & 1u);
		float  m   = row[0];
		#pragma unroll 4
		for (uint32_t iv = 1; iv < V; iv++) m = fminf(m, row[2 * iv]);
		float fl = __builtin_floorf(m);
		#pragma unroll 4
		for (uint32_t iv = 0; iv < V; iv++) row[2 * iv] = (row[2 * iv] - fl) - 0.5f;      /* both steps exact */
		tfl[a] = (int)fl;
	}
	__syncthreads();                                         /* the floors are read below */

	uint32_t lu, lv;
	if (u_axis == 0) { lu = tid & (U - 1); lv = tid >> q.u_shift; }
	else             { lv = tid & (V - 1); lu = tid >> VS; }
	const uint32_t gu = tu * U + lu, gv = tv * V + lv;
	const uint32_t x = u_axis == 0 ? gu : gv, y = u_axis == 0 ? gv : gu;
	const bool inside = x < p.size[0] && y < p.size[1];

	float coherent = 0.f, incoherent = 0.f;
	const f32x2   *Rl = R + lu;
	const uint32_t ulast = (uint32_t)(S - 1);
	uint32_t tz_base = (uint32_t)(uintptr_t)(lds_f32x2 *)Tz;
	asm("" : "+s"(tz_base));

	/* staging: as das_staged.hip, 4-byte samples */
	const __amdgpu_buffer_rsrc_t rf_rsrc = __builtin_amdgcn_make_buffer_rsrc(
		const_cast<void *>(p.rf), 0, (int)((uint32_t)C * (uint32_t)A * (uint32_t)S * 4u), 0x00020000);
	const uint32_t windows_per_pass = nthreads >> WS;
	uint32_t stage_inv[NL];
	#pragma unroll
	for (int n = 0; n < NL; n++) {
		uint32_t a = (tid >> WS) + (uint32_t)n * windows_per_pass;
		stage_inv[n] = a < (uint32_t)A ? (a * (uint32_t)S + (uint32_t)(tfl[a] + (int)(tid & (W - 1)))) * 4u : 0x80000000u;
	}
	auto stage_load = [&](int channel, int rfl, float (&regs)[NL]) {
		const uint32_t at = ((uint32_t)channel * (uint32_t)A * (uint32_t)S + (uint32_t)rfl) * 4u;
		#pragma unroll
		for (int n = 0; n < NL; n++)
			regs[n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rf_rsrc, (int)(stage_inv[n] + at), 0, 0));
	};
	const float half_minus_j = 0.5f - (float)(tid & (W - 1));
	auto stage_store = [&](const float (&regs)[NL]) {
		#pragma unroll
		for (int n = 0; n < NL; n++) {
			const float s0 = regs[n];
			const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0x130, 0xf, 0xf, true));
			uint32_t e = tid + (uint32_t)n * nthreads;
			const float d = s1 - s0;
			if (e < stage_elements) stage[e] = f32x2{__builtin_fmaf(half_minus_j, d, s0), d};
		}
	};

	for (int c0 = 0; c0 < C; c0 += chunk) {
		const int cn = (C - c0) < chunk ? (C - c0) : chunk;
		__syncthreads();
		{
		/* the ~45 scalars of the receive-table build come from the kernel-argument segment at the top of every chunk instead of
		 * living in SGPRs across the channel loop (das_staged.hip) */
		typedef __attribute__((address_space(4))) const BfDasArgs const_args;
		const_args *ka = (const_args *)__builtin_amdgcn_kernarg_segment_ptr();
		asm volatile("" : "+s"(ka));
		const float k_denom_u = fmaxf(1.0f, (float)ka->size[u_axis] - 1.0f);
		const float k_pz = (float)z / fmaxf(1.0f, (float)ka->size[2] - 1.0f);
		const float k_fs = ka->sampling_frequency, k_inv_c = ka->inv_speed_of_sound, k_c = ka->speed_of_sound, k_fnum = ka->f_number;
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
			float r_idx   = div_speed_of_sound(hw_sqrt(dx * dx + xz * xz), k_inv_c, k_c) * k_fs;
			float apod    = 0.f;
			if (a_arg < 0.5f) { float cs = hw_cos_turns(0.5f * a_arg); apod = cs * cs; }
			R[e] = f32x2{r_idx, apod};
		}
		}
		__syncthreads();
		for (uint32_t cl = tid; cl < (uint32_t)cn; cl += nthreads) {
			const float *row = reinterpret_cast<const float *>(R + (size_t)cl * U);
			float m = row[0];
			#pragma unroll 4
			for (uint32_t iu = 1; iu < U; iu++) m = fminf(m, row[2 * iu]);
			rfloor[cl] = (int)__builtin_floorf(m);
		}
		__syncthreads();
		for (uint32_t e = tid; e < (uint32_t)cn * U; e += nthreads) {
			const f32x2 entry = R[e];
			float r_abs = entry.x, w = entry.y;
			const bool lane_safe = (r_abs + range_lo >= 0.f) && (r_abs + range_hi < (float)(S - 1));
			r_abs -= (float)rfloor[e >> q.u_shift];
			if (!lane_safe) w = -w;                          /* -0.0f outside the aperture: still "unsafe" to the sign test */
			R[e] = f32x2{r_abs, w};
		}
		__syncthreads();

		float regs[NL];
		stage_load(c0, rfloor[0], regs);
		for (int cl = 0; cl < cn; cl++) {
			__syncthreads();
			stage_store(regs);
			__syncthreads();
			if (cl + 1 < cn) stage_load(c0 + cl + 1, rfloor[cl + 1], regs);
			if (!inside) continue;

			float r_rel, r_w;
			{
				const f32x2 r = Rl[(size_t)cl * U];
				r_rel = r.x; r_w = r.y;
			}
			if (__builtin_amdgcn_ballot_w64(r_w != 0.f) == 0) continue;    /* F# culling per wave */
			const bool wave_safe = !(q.depth_major & 2u) && __builtin_amdgcn_ballot_w64(__builtin_signbitf(r_w)) == 0;
			f32x2 sum2 = {0.f, 0.f}, mag2 = {0.f, 0.f};
			auto batches = [&](auto checked) {
				constexpr bool CHECK = decltype(checked)::value;
				uint32_t lane_id = tid;
				asm volatile("" : "+v"(lane_id));
				const uint32_t lane_v = u_axis == 0 ? lane_id >> q.u_shift : lane_id & (V - 1);
				uint32_t tz_at = tz_base + (lane_v << 3);
				uint32_t m_bits = 0x4B000002u;               /* 2^23 + 2: das_staged.hip explains the rounding and the bias */
				[[maybe_unused]] bool window_left = false;    /* range-checked loop: some term selected an element outside its window */
				const f32x2 rr = {r_rel, r_rel};
				for (int a = 0; a < A4; a += 4, tz_at += 2u * V * 8u, m_bits += 4u * W) {
					uint32_t at[4]; f32x2 tap[4];
					const float M = __builtin_bit_cast(float, m_bits);
					const f32x2 M2 = {M, M};
					const f32x2 tz01 = *(lds_f32x2 *)(uintptr_t)tz_at;
					const f32x2 tz23 = *(lds_f32x2 *)(uintptr_t)(tz_at + V * 8u);
					const f32x2 p01 = rr + tz01, p23 = rr + tz23;
					const f32x2 y01 = p01 + M2,  y23 = p23 + M2;
					const float ys[4] = {y01.x, y01.y, y23.x, y23.y}, ps[4] = {p01.x, p01.y, p23.x, p23.y};
					#pragma unroll
					for (int k = 0; k < 4; k++) {
						const uint32_t yb = __builtin_bit_cast(uint32_t, ys[k]);
						asm("v_lshlrev_b16 %0, 3, %1" : "=v"(at[k]) : "v"(yb));
						if constexpr (CHECK) {
							uint32_t k_abs = (uint32_t)((int)(yb - m_bits) + rfloor[cl] + tfl[a + k]);      /* yb - m_bits = round(p) */
							at[k] = k_abs < ulast ? at[k] + (uint32_t)k * W * 8u : (stage_elements + 2u) * 8u;
							window_left |= __builtin_amdgcn_ballot_w64((yb - m_bits) > W - 2u) != 0ull;      /* (wave uniform: a scalar) never, unless plan_staged's bound is wrong */
						}
					}
					#pragma unroll
					for (int k = 0; k < 4; k++) tap[k] = *(lds_f32x2 *)(uintptr_t)(at[k] + (CHECK ? 0u : (uint32_t)k * W * 8u));   /* immediate */
					float sv[4];
					#pragma unroll
					for (int k = 0; k < 4; k++) {
						sv[k] = __builtin_fmaf(ps[k], tap[k].y, tap[k].x);
						asm("" : "+v"(sv[k]));       /* four plain fmas: packed, hipcc spends six moves pairing their operands */
					}
					sum2 += f32x2{sv[0], sv[1]}; sum2 += f32x2{sv[2], sv[3]};
					if constexpr (CW) {
						mag2 += f32x2{__builtin_fabsf(sv[0]), __builtin_fabsf(sv[1])};
						mag2 += f32x2{__builtin_fabsf(sv[2]), __builtin_fabsf(sv[3])};
					}
				}
				if constexpr (CHECK) { if (window_left) staged_violation_raise(); }
			};
			if (wave_safe) batches(std::false_type{});
			else           batches(std::true_type{});
			const float apod = __builtin_fabsf(r_w);
			coherent = __builtin_fmaf(sum2.x + sum2.y, apod, coherent);
			if constexpr (CW) incoherent = __builtin_fmaf(mag2.x + mag2.y, apod, incoherent);
		}
	}
	if (q.depth_major & 2u) staged_violation_report(tid);      /* (block uniform: every thread reaches it) */
	if (!inside) return;

	uint64_t out_index = (uint64_t)p.size[0] * p.size[1] * zl + (uint64_t)p.size[0] * y + x;
	if constexpr (CW) coherent = coherent * (coherent / incoherent);   /* coherency_weighting.glsl:36 */
	reinterpret_cast<float *>(p.out)[out_index] = coherent;
}

template <bool CW, int VS, int WS, int NL>
static hipError_t launch_staged_real(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	uint32_t total = q->tiles[0] * q->tiles[1] * q->tiles[2];
	uint32_t grid  = ((total + 7u) / 8u) * 8u;
	auto kernel = das_rca_staged_real_kernel<CW, VS, WS, NL>;
	hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q->lds_bytes);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(kernel, dim3(grid), dim3(q->threads), q->lds_bytes, s, *a, *q);
	return hipGetLastError();
}

template <bool CW, int VS, int WS>
static hipError_t launch_staged_real_loads(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	const uint32_t A4 = ((uint32_t)a->acquisition_count + 3u) & ~3u;
	switch (((A4 << WS) + q->threads - 1) / q->threads) {
	case 1: return launch_staged_real<CW, VS, WS, 1>(a, q, s);
	case 2: return launch_staged_real<CW, VS, WS, 2>(a, q, s);
	case 3: return launch_staged_real<CW, VS, WS, 3>(a, q, s);
	case 4: return launch_staged_real<CW, VS, WS, 4>(a, q, s);
	case 5: case 6: return launch_staged_real<CW, VS, WS, 6>(a, q, s);       /* (a staging width larger than needed only loads zeros) */
	case 7: case 8: return launch_staged_real<CW, VS, WS, 8>(a, q, s);
	}
	return hipErrorInvalidValue;
}

template <bool CW>
static hipError_t launch_staged_real_shape(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	switch ((q->v_shift << 4) | q->window_shift) {
	case (4 << 4) | 5: return launch_staged_real_loads<CW, 4, 5>(a, q, s);
	case (5 << 4) | 5: return launch_staged_real_loads<CW, 5, 5>(a, q, s);
	case (6 << 4) | 5: return launch_staged_real_loads<CW, 6, 5>(a, q, s);
	case (4 << 4) | 6: return launch_staged_real_loads<CW, 4, 6>(a, q, s);
	case (5 << 4) | 6: return launch_staged_real_loads<CW, 5, 6>(a, q, s);
	case (6 << 4) | 6: return launch_staged_real_loads<CW, 6, 6>(a, q, s);
	}
	return hipErrorInvalidValue;
}

/* real samples, linear interpolation; the caller (bf_launch_das_staged) checked the rest */
extern "C" hipError_t bf_launch_das_staged_real(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	if (a->complex_data || a->interpolation != BF_INTERP_LINEAR) return hipErrorInvalidValue;
	if ((uint64_t)a->channel_count * (uint64_t)a->acquisition_count * (uint64_t)a->sample_count * 4u >= (1ull << 31)) return hipErrorInvalidValue;
	return a->coherency_weighting ? launch_staged_real_shape<true>(a, q, s) : launch_staged_real_shape<false>(a, q, s);
}
