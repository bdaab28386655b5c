/* das_factored.hip -- delay-and-sum for gfx950 (MI355X), per-voxel factored path.
 *
 * Same result as das.hip's general kernel (shaders/das.glsl RCA :204-231, FORCES :288-321)
 * for the families whose sample index is a SUM of a receive term and a transmit term for
 * every voxel:
 *
 *     RCA / TPW / VLS / Flash   index = [ (d_tx(a) / c + t0) fs ] + [ d_rx(ch) / c * fs ]
 *     FORCES / UFORCES          index = [ d_tx(a) fs / c ]        + [ (d_rx(ch) / c + t0) fs ]
 *
 * The general kernel spends ~45 VALU slots per (voxel, channel, transmit) triple, most of it
 * on work that depends on only one of the two loop variables: a square root and the cos^2
 * apodization per channel, and -- for IQ data -- the sin/cos of the demodulation phase, which
 * is e^{j phi(T + R)} = e^{j phi(T)} e^{j phi(R)}.  Here a thread (one voxel) keeps the receive
 * factors of CH channels in registers {R index, apod e^{j phi(R)}, apod}, walks the transmits
 * once per chunk computing {T index, e^{j phi(T)}} per transmit, and for each of the CH
 * triples of a transmit does only: one add, the interpolation, a complex multiply-accumulate
 * into that channel's partial sum and |sample| for coherency weighting.  The receive factor is
 * applied once per channel when the chunk is folded into the voxel's accumulators.
 * No LDS tables and no constraint on how the voxel grid lies relative to the array, unlike
 * das_separable.hip: this is the path of 2-D plane-wave compounding (tx and rx on the same
 * axis), tilted volumes and FORCES.  HERCULES (joint square root) and READI stay on das.hip.
 *
 * The CH gathers of a transmit are independent and issued back to back (tap_setup -> tap_load
 * x CH -> tap_finish, das_common.h); a chunk none of whose channels passes the f-number test
 * for any lane of the wave is skipped.  Summation order differs from the shader's (per channel
 * over transmits, then over channels): results agree to float rounding, tests state the
 * tolerance.  Channel split for small frames as in das.hip.
 */
#include "das_exact.h"

/* channels per register-resident chunk.  Measured on MI355X (config 4 geometry, 64 planes):
 * 2 -> 172 ms, 4 -> 160, 6 -> 154, 8 -> 154 with spills; cubic holds twice the gather data
 * per triple and is best at 4 (108 VGPRs). */
#ifndef BF_FACTORED_CHUNK
/* Every gather of a transmit's CH terms is issued before the first of them is consumed.  hipcc's scheduler otherwise decides by
 * itself whether to interleave (load, load, wait, arithmetic, load ...) or to batch, and flips with changes as remote as the order
 * of BfDasArgs' fields: on the harness's coarse grid, where a gather misses L1 more often than not, the batched form measured
 * 15.8 ms against the interleaved one's 18.9 on the same frame (profiles/r03_harness.json, round 3) -- so it is pinned. */
#define BF_ALL_GATHERS_ISSUED() __builtin_amdgcn_sched_barrier(0)
#define BF_FACTORED_CHUNK(interp) ((interp) == BF_INTERP_LINEAR ? 6 : 4)
#endif
namespace {

template <bool CPLX, bool CW>
struct ChannelFactor {
	float index;           /* receive part of the sample index; -1e9 when the channel fails the f-number test */
	float re, im;          /* apod * e^{j phi(R)} (CPLX) -- re alone holds apod for real data */
	float apod;            /* for the incoherent sum */
};

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

/* ROW_ENDS: the instantiation for launches in which a term can come within reach of an end of its RF row (BfDasArgs::row_ends, a host
 * bound: das_select.cpp); the other one carries none of that code -- its loops and its register allocation are round 3's. */
template <int FAMILY, int INTERP, bool CPLX, bool CW, bool ROW_ENDS>
__global__ __launch_bounds__(1024) void das_factored_kernel(const BfDasArgs p)
{
	constexpr int      CH = BF_FACTORED_CHUNK(INTERP);
	constexpr uint32_t ES = CPLX ? 8 : 4;
	extern __shared__ __attribute__((aligned(16))) unsigned char factored_lds[];

	/* blockIdx -> tile and thread -> voxel exactly as das.hip */
	uint32_t total = p.blocks[0] * p.blocks[1] * p.blocks[2];
	uint32_t bid   = blockIdx.x;
	uint32_t per   = (total + 7u) / 8u;
	uint32_t tile  = (bid & 7u) * per + (bid >> 3);
	if (p.depth_major != 3u && tile >= total) return;
	/* depth-major walk: consecutive tiles (in flight together on an XCD) are one lateral column at
	 * consecutive depths, whose RF windows overlap almost entirely (das_separable.hip) */
	uint32_t bx, by, bz;
	if (p.depth_major == 3u) {
		bz = 0;
		if (!bf_plane_walk(bid, p.blocks[0], p.blocks[1], p.band_rows, bx, by)) return;     /* whole block */
	} else if (p.depth_major == 2u) {
		/* view planes (depth on voxel y, one voxel along z): y fastest, so that each XCD's run of tiles is a lateral COLUMN
		 * at every depth -- the work per tile grows with depth (f-number culling), a run of depth ROWS would leave the XCDs
		 * that hold the shallow rows idle for a fifth of the launch */
		by = tile % p.blocks[1];
		bx = (tile / p.blocks[1]) % p.blocks[0];
		bz = tile / (p.blocks[1] * p.blocks[0]);
	} else if (p.depth_major) {
		bz = tile % p.blocks[2];
		bx = (tile / p.blocks[2]) % p.blocks[0];
		by = tile / (p.blocks[2] * p.blocks[0]);
	} else {
		bx = tile % p.blocks[0];
		by = (tile / p.blocks[0]) % p.blocks[1];
		bz = tile / (p.blocks[0] * p.blocks[1]);
	}

	uint32_t tid = threadIdx.x;
	uint32_t lx  = tid & ((1u << p.tile_shift[0]) - 1u);
	uint32_t ly  = (tid >> p.tile_shift[0]) & ((1u << p.tile_shift[1]) - 1u);
	uint32_t lz  = (tid >> (p.tile_shift[0] + p.tile_shift[1])) & ((1u << p.tile_shift[2]) - 1u);
	uint32_t split = tid >> (p.tile_shift[0] + p.tile_shift[1] + p.tile_shift[2]);
	/* (the voxel is worked out again where it is needed after the channel loop -- the row-end pass, the store -- rather than held in
	 * registers across it: the linear IQ instances run at their 128-register limit) */
	auto voxel_of = [&](uint32_t thread, uint32_t &vx, uint32_t &vy, uint32_t &vz) {
		vx = (bx << p.tile_shift[0]) + (thread & ((1u << p.tile_shift[0]) - 1u));
		vy = (by << p.tile_shift[1]) + ((thread >> p.tile_shift[0]) & ((1u << p.tile_shift[1]) - 1u));
		vz = (bz << p.tile_shift[2]) + ((thread >> (p.tile_shift[0] + p.tile_shift[1])) & ((1u << p.tile_shift[2]) - 1u));
	};
	uint32_t x = (bx << p.tile_shift[0]) + lx;
	uint32_t y = (by << p.tile_shift[1]) + ly;
	uint32_t zl = (bz << p.tile_shift[2]) + lz;
	const bool inside = x < p.size[0] && y < p.size[1] && zl < p.z_count;
	/* every lane takes part in the loops (lanes beyond the grid repeat its last voxel and store nothing), and a wave wholly beyond the grid
	 * skips the work but not the barriers: the waves of a block walk the channel chunks in step (below) */
	const bool wave_active = __builtin_amdgcn_ballot_w64(inside) != 0ull;
	x  = x  < p.size[0] ? x  : p.size[0] - 1u;
	y  = y  < p.size[1] ? y  : p.size[1] - 1u;
	zl = zl < p.z_count ? zl : p.z_count - 1u;

	sample_t<CPLX> coherent = zero_sample<CPLX>();
	float          incoherent = 0.f;

	{
		uint32_t z = p.z_first + zl;
		float px = (float)x / fmaxf(1.0f, (float)p.size[0] - 1.0f);       /* das.glsl:374-376 */
		float py = (float)y / fmaxf(1.0f, (float)p.size[1] - 1.0f);
		float pz = (float)z / fmaxf(1.0f, (float)p.size[2] - 1.0f);
		float wx, wy, wz;
		m4_point(p.voxel_transform, px, py, pz, wx, wy, wz);

		const char *rf = (const char *)p.rf;
		const int   S = p.sample_count, A = p.acquisition_count, C = p.channel_count;
		const int   last = S - 1;
		const float Sf = (float)S;
		const float turns_per_sample = p.demodulation_frequency * p.inv_sampling_frequency;

		/* transducer-space point; FORCES voxels arrive already transformed (beamformer_core.c:913-915) */
		float xx, xy, xz;
		if constexpr (FAMILY == BF_DAS_RCA) m4_point(p.xdc_transform, wx, wy, wz, xx, xy, xz);
		else { xx = wx; xy = wy; xz = wz; }
		const float zz = xz * xz;

		/* receive geometry: one orientation for all transmits (the host checks) */
		float lateral, pitch, f_over_z;
		if constexpr (FAMILY == BF_DAS_RCA) {
			const bool rx_rows = (p.transmits[0].flags & BF_RX_ROWS) != 0;
			lateral  = rx_rows ? xy : xx;
			pitch    = rx_rows ? p.pitch[1] : p.pitch[0];
			f_over_z = p.f_number * hw_rcp(__builtin_fabsf(xz));
		} else {
			lateral  = xx;
			pitch    = p.pitch[0];
			f_over_z = p.f_number * hw_rcp(xz);
		}
		/* FORCES transmit geometry (das.glsl:292-296) */
		const int   first_transmit = FAMILY == BF_DAS_RCA ? 0 : (p.sparse != 0);
		float transmit_yz_squared = 0.f;
		if constexpr (FAMILY != BF_DAS_RCA) {
			float dy = xy - p.pitch[1] * (float)C * 0.5f;
			transmit_yz_squared = dy * dy + zz;
		}

		const int per_split = (C + (1 << p.split_shift) - 1) >> p.split_shift;
		const int ch_begin  = (int)split * per_split;
		const int ch_end    = ch_begin + per_split < C ? ch_begin + per_split : C;

		/* the transmit part of the sample index of this lane's voxel (RCA: with the time offset; FORCES: das.glsl:312) */
		/* (the tables are read through the constant address space: with LDS-DMA stores in the loop the compiler can no longer
		 * prove global memory unchanged and would turn these wave-uniform reads into VECTOR loads -- whose results it then
		 * waits for with vmcnt(0), draining the DMA queue every transmit) */
		typedef __attribute__((address_space(4))) const f32x4   const_f32x4;
		typedef __attribute__((address_space(4))) const int16_t const_i16;
		const_f32x4 *transmits_c = (const_f32x4 *)(uintptr_t)p.transmits;      /* BfTransmit: {sin, cos, focus x, focus z}, {flags, pad x 3} */
		const_i16   *sparse_c    = (const_i16 *)(uintptr_t)p.sparse_elements;
		auto transmit_index = [&](int a) -> float {
			if constexpr (FAMILY == BF_DAS_RCA) {
				static_assert(sizeof(BfTransmit) == 32, "two 16-byte scalar loads per transmit");
				const f32x4 t_lo = transmits_c[2 * a], t_hi = transmits_c[2 * a + 1];
				BfTransmit t;
				t.sin_a = t_lo.x; t.cos_a = t_lo.y; t.focus_x = t_lo.z; t.focus_z = t_lo.w;
				{ const float f = t_hi.x; t.flags = __builtin_bit_cast(uint32_t, f); }
				return (div_speed_of_sound(transmit_distance(t, wx, wy, wz), p) + p.time_offset) * p.sampling_frequency;
			} else {
				float tx_channel = p.sparse ? (float)sparse_c[a - first_transmit] : (float)a;
				float tdx        = xx - p.pitch[0] * tx_channel;
				return div_speed_of_sound(hw_sqrt(transmit_yz_squared + tdx * tdx) * p.sampling_frequency, p);
			}
		};

		/* the receive part of the sample index of channel `channel` for this lane's voxel (RCA: the time offset rides with the transmit
		 * term; FORCES: with the receive term -- sample_index, das.glsl:126-130).  One function, explicit fmas: the row-end pass at the end
		 * of the kernel forms the same value again, bit for bit */
		auto receive_index = [&](int channel, float &dx) -> float {
			dx = __builtin_fmaf(-(float)channel, pitch, lateral);
			const float dist = hw_sqrt(__builtin_fmaf(dx, dx, zz));
			return FAMILY == BF_DAS_RCA ? div_speed_of_sound(dist, p) * p.sampling_frequency
			                            : (div_speed_of_sound(dist, p) + p.time_offset) * p.sampling_frequency;
		};

		/* Row ends (das_exact.h).  This kernel's index is a rounded receive term plus a rounded transmit term: within p.edge_margin of an
		 * end of sample_rf's valid range it is not trusted to decide keep-or-drop.  One pass over the transmits gives the lane the
		 * extremes of its transmit term; a chunk in which no lane of the wave can come that close to an end for any transmit (nearly
		 * every chunk of nearly every frame) needs nothing; in the others every transmit gets one wave-level test before its terms, and
		 * the rare term inside the margin is corrected with the shader's own index on the spot.  The gathers, the interpolation and the
		 * accumulation are round 3's, untouched. */
		constexpr bool EDGES = ROW_ENDS && INTERP != BF_INTERP_NEAREST;      /* (nearest flips at every half-integer: budgeted per voxel by the tests) */
		[[maybe_unused]] const float edge_margin = p.edge_margin;
		/* does the index interval [lo_index, hi_index] contain a point within the margin of an end of the row?  (Terms wholly inside the
		 * row need no test; terms wholly BEYOND it -- the outermost channels of the deepest pixels -- are dropped by the ordinary range
		 * test and need none either: only intervals that straddle an end do.) */
		[[maybe_unused]] auto straddles = [&](float lo_index, float hi_index) -> bool {
			const float e_hi = bfx::edge_hi<INTERP>(S), e_lo = bfx::edge_lo<INTERP>();
			return (lo_index < e_hi + edge_margin && hi_index > e_hi - edge_margin) || (lo_index < e_lo + edge_margin && hi_index > e_lo - edge_margin);
		};
		[[maybe_unused]] float t_lo = 0.f, t_hi = 0.f;
		if constexpr (EDGES) {
			t_lo = __builtin_inff(); t_hi = -__builtin_inff();
			for (int a = first_transmit; a < A; a++) {
				const float t = transmit_index(a);
				t_lo = fminf(t_lo, t); t_hi = fmaxf(t_hi, t);
			}
		}

		for (int c_rel = 0; c_rel < per_split; c_rel += CH) {          /* (the same number of turns in every wave of a block: the barrier) */
			const int c0 = ch_begin + c_rel;
			/* the waves of a block -- neighbours along x at one depth -- start every chunk together: left alone they drift apart over the
			 * 64 chunks x 128 transmits of a frame and stop sharing the lines they pull through L1.  Same-box A/B on the reference
			 * harness's plane: FORCES 16.88 -> 15.89 ms, VLS 16.86 -> 16.22, TPW 15.11 -> 15.35 */
			__syncthreads();
			if (!wave_active || c0 >= ch_end) continue;
			ChannelFactor<CPLX, CW> R[CH];
			bool any = false;
			#pragma unroll
			for (int k = 0; k < CH; k++) {
				int   channel = c0 + k;
				float dx;
				float index   = receive_index(channel, dx);
				float a_arg   = __builtin_fabsf(dx * f_over_z);
				bool  pass    = a_arg < 0.5f && channel < ch_end;
				float apod    = pass ? apodize(a_arg) : 0.f;
				R[k].index = pass ? index : -1.0e9f;
				R[k].apod  = apod;
				if constexpr (CPLX) {
					float turns = hw_fract(turns_per_sample * index);
					R[k].re = apod * hw_cos_turns(turns);
					R[k].im = apod * hw_sin_turns(turns);
				} else {
					R[k].re = apod; R[k].im = 0.f;
				}
				any |= pass;
			}
			if (!__builtin_amdgcn_ballot_w64(any)) continue;          /* wave-uniform */
			[[maybe_unused]] bool edges = false;                      /* wave-uniform: some lane's index range over the transmits contains an end of its row in this chunk */
			if constexpr (EDGES) {
				bool lane_safe = true;
				#pragma unroll
				for (int k = 0; k < CH; k++)
					lane_safe = lane_safe && (R[k].index < -1.0e8f || !straddles(R[k].index + t_lo, R[k].index + t_hi));
				edges = __builtin_amdgcn_ballot_w64(!lane_safe) != 0ull;
			}
			/* ... and, for such a chunk, the extremes of the lane's receive terms over its channels: per transmit ONE wave-level test of
			 * [r_lo, r_hi] + T against the row's ends decides whether any term of that transmit needs the per-term test at all */
			/* (as bounds on T itself -- T in (hi_a, hi_b): the interval [r_lo, r_hi] + T reaches the row's far end; (lo_a, lo_b): its near end --
			 * widened by a second margin for the roundings of the bounds: two compares per transmit, and the near end only in chunks where
			 * the smallest transmit term can get there at all) */
			[[maybe_unused]] float hi_a = __builtin_inff(), hi_b = -__builtin_inff(), lo_a = __builtin_inff(), lo_b = -__builtin_inff();
			[[maybe_unused]] bool near_end_too = false;
			if constexpr (EDGES) {
				if (edges) {
					float r_lo = __builtin_inff(), r_hi = -__builtin_inff();
					#pragma unroll
					for (int k = 0; k < CH; k++)
						if (R[k].index > -1.0e8f) { r_lo = fminf(r_lo, R[k].index); r_hi = fmaxf(r_hi, R[k].index); }
					const float e_hi = bfx::edge_hi<INTERP>(S), e_lo = bfx::edge_lo<INTERP>(), m2 = 2.0f * edge_margin;
					hi_a = (e_hi - m2) - r_hi; hi_b = (e_hi + m2) - r_lo;
					lo_a = (e_lo - m2) - r_hi; lo_b = (e_lo + m2) - r_lo;
					near_end_too = __builtin_amdgcn_ballot_w64(t_lo < lo_b) != 0ull;
				}
			}
			sample_t<CPLX> part[CH];
			float          part_abs[CH];
			#pragma unroll
			for (int k = 0; k < CH; k++) { part[k] = zero_sample<CPLX>(); part_abs[k] = 0.f; }
			/* IQ data, linear / cubic: the rotate-accumulate runs as two packed FMAs per triple on
			 *   acc1 += s.re (cos, sin)(T),  acc2 += s.im (cos, sin)(T),
			 * combined as (acc1.x - acc2.y, acc1.y + acc2.x) when the chunk is folded */
			constexpr bool PACKED = CPLX && INTERP != BF_INTERP_NEAREST;
			f32x2 acc1[CH], acc2[CH];
			#pragma unroll
			for (int k = 0; k < CH; k++) { acc1[k] = f32x2{0.f, 0.f}; acc2[k] = f32x2{0.f, 0.f}; }

			/* (the chunk's first channel is the same in every lane -- the channel split is per wave -- but derives from the thread id: told
			 * to the compiler, the row offsets of the loop are scalars and a tap's byte offset is one v_lshl_add_u32) */
			const uint32_t chunk_row = (uint32_t)__builtin_amdgcn_readfirstlane(c0) * (uint32_t)A * (uint32_t)S * ES;
			for (int a = first_transmit; a < A; a++) {
				float t_index = transmit_index(a);
				asm volatile("" : "+v"(t_index));           /* not fused into the per-channel adds: every kernel variant rounds the same way */
				if constexpr (EDGES) {
					if (edges && __builtin_amdgcn_ballot_w64((t_index > hi_a && t_index < hi_b) || (near_end_too && t_index > lo_a && t_index < lo_b)) != 0ull) {
						/* ---- row ends (rare): a term of this transmit within the margin of an end of its row is evaluated from the voxel's
						 * integer coordinates with the shader's own index, and the voxel's sums are CORRECTED by the difference to what the
						 * loop below decides and adds for it with its own index (das_exact.h: edge_correct) */
						bool some = false;
						#pragma unroll
						for (int k = 0; k < CH; k++) some = some || bfx::edge_near<INTERP>(t_index + R[k].index, S, edge_margin);
						if (__builtin_amdgcn_ballot_w64(some) != 0ull) {           /* (else: the intervals contain an end, but no term comes near it) */
						uint32_t ex, ey, ezl, thread = threadIdx.x;
						asm volatile("" : "+v"(thread));             /* (worked out here, not held across the loop) */
						voxel_of(thread, ex, ey, ezl);
						#pragma unroll 1
						for (int k = 0; k < CH; k++) {
							/* (R[k] by a chain of selects on the scalar k: a run-time index would put the register array on the stack) */
							float r_k = R[0].index;
							#pragma unroll
							for (int j = 1; j < CH; j++) r_k = k == j ? R[j].index : r_k;
							const float index = t_index + r_k;             /* (a channel outside the aperture: -1e9, never near) */
							if (bfx::edge_near<INTERP>(index, S, edge_margin))
								bfx::edge_correct<FAMILY, INTERP, CPLX, CW>(bfx::kernel_args(), ex, ey, p.z_first + ezl, c0 + k, a, index, coherent, incoherent);
						}
						}
					}
				}
				float tc = 1.f, ts = 0.f;
				if constexpr (CPLX) {
					float turns = hw_fract(turns_per_sample * t_index);
					tc = hw_cos_turns(turns); ts = hw_sin_turns(turns);
				}

				const uint32_t row0 = chunk_row + (uint32_t)a * (uint32_t)S * ES;
				const uint32_t row_step = (uint32_t)A * (uint32_t)S * ES;
				if constexpr (PACKED) {
					/* Written out so that every step is one instruction: index = T + R;
					 * frac = v_fract; k = v_cvt_flr_i32; offset = in range ? row + 8k : zero block
					 * (lanes outside the row, channels outside the aperture and the padding of a
					 * ragged chunk all gather zeros), then the interpolation on packed pairs. */
					const f32x2 cs = {tc, ts};
					float frac[CH]; uint32_t off[CH];
					#pragma unroll
					for (int k = 0; k < CH; k++) {
						float index = t_index + R[k].index;
						frac[k] = hw_fract(index);
						if constexpr (INTERP == BF_INTERP_LINEAR) {
							uint32_t ki = (uint32_t)cvt_floor_i32(index), tap;            /* valid: 0 <= index < S-1 */
							asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(tap) : "v"(ki), "s"(row0 + (uint32_t)k * row_step));
							off[k] = ki < (uint32_t)(S - 1) ? tap : p.zero_offset;
						} else {
							uint32_t kf = (uint32_t)cvt_floor_i32(index);                 /* valid: 1 <= index < S-2 */
							uint32_t tap;                                                 /* (one instruction; the compiler re-associates the C expression into two) */
							asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(tap) : "v"(kf), "s"(row0 + (uint32_t)k * row_step - ES));
							off[k] = kf - 1u < (uint32_t)(S - 3) ? tap : p.zero_offset;
						}
					}
					if constexpr (INTERP == BF_INTERP_LINEAR) {
						f32x4 d[CH];
						#pragma unroll
						for (int k = 0; k < CH; k++) d[k] = gather<f32x4_a8>(rf, off[k]);
						BF_ALL_GATHERS_ISSUED();
						#pragma unroll
						for (int k = 0; k < CH; k++) {
							f32x2 s0 = {d[k].x, d[k].y}, s1 = {d[k].z, d[k].w};
							f32x2 sv = s0 + frac[k] * (s1 - s0);
							acc1[k] += sv.x * cs;
							acc2[k] += sv.y * cs;
							if constexpr (CW) { f32x2 sq = sv * sv; part_abs[k] += hw_sqrt(sq.x + sq.y); }
						}
					} else {
						f32x4 d0[CH], d1[CH];
						#pragma unroll
						for (int k = 0; k < CH; k++) { d0[k] = gather<f32x4_a8>(rf, off[k]); d1[k] = gather_at<f32x4_a8, 16>(rf, off[k]); }
						BF_ALL_GATHERS_ISSUED();
						#pragma unroll
						for (int k = 0; k < CH; k++) {
							/* Catmull-Rom Hermite (das.glsl:67-97) as four weights of the taps (bf_catmull_rom, das_common.h: nine scalar
							 * operations) and one packed multiply + three packed fmas -- 38 clk of issue against the 56 of the Horner form
							 * in the samples (four subtractions + eleven packed operations) this loop had through round 3 */
							f32x2 s0 = {d0[k].x, d0[k].y}, s1 = {d0[k].z, d0[k].w}, s2 = {d1[k].x, d1[k].y}, s3 = {d1[k].z, d1[k].w};
							float w0, w1, w2, w3;
							bf_catmull_rom(frac[k], w0, w1, w2, w3);
							f32x2 sv = w0 * s0 + w1 * s1 + w2 * s2 + w3 * s3;
							acc1[k] += sv.x * cs;
							acc2[k] += sv.y * cs;
							if constexpr (CW) { f32x2 sq = sv * sv; part_abs[k] += hw_sqrt(sq.x + sq.y); }
						}
					}
					continue;
				}
				Tap<INTERP>           tap[CH];
				TapData<INTERP, CPLX> data[CH];
				#pragma unroll
				for (int k = 0; k < CH; k++) tap[k] = tap_setup<INTERP, CPLX>(t_index + R[k].index, Sf, last);
				#pragma unroll
				for (int k = 0; k < CH; k++) {
					/* rows past the last channel of a ragged chunk are not read: their weights are
					 * zero, so any legal row will do */
					uint32_t row = c0 + k < C ? row0 + (uint32_t)k * row_step : row0;
					data[k] = tap_load<INTERP, CPLX>(rf, row + tap[k].off);
				}
				#pragma unroll
				for (int k = 0; k < CH; k++) {
					sample_t<CPLX> s = tap_finish<INTERP, CPLX>(tap[k], data[k]);
					if constexpr (CPLX) {
						part[k].x += tc * s.x - ts * s.y;                /* rotate_iq, das.glsl:54-61 */
						part[k].y += ts * s.x + tc * s.y;
						if constexpr (CW) part_abs[k] += hw_sqrt(s.x * s.x + s.y * s.y);
					} else {
						part[k] += s;
						if constexpr (CW) part_abs[k] += __builtin_fabsf(s);
					}
				}
			}

			#pragma unroll
			for (int k = 0; k < CH; k++) {
				if constexpr (PACKED) part[k] = f32x2{acc1[k].x - acc2[k].y, acc1[k].y + acc2[k].x};
				if constexpr (CPLX) {
					coherent.x += R[k].re * part[k].x - R[k].im * part[k].y;
					coherent.y += R[k].im * part[k].x + R[k].re * part[k].y;
				} else {
					coherent += R[k].re * part[k];
				}
				if constexpr (CW) incoherent += R[k].apod * part_abs[k];
			}
		}

	}

	if (p.split_shift) {
		/* partial sums of waves 1..K-1 go through LDS; wave 0 adds them in split order */
		float *partial = reinterpret_cast<float *>(factored_lds);  /* [K-1][3][64] */
		const uint32_t lane = tid & 63u;
		if (split) {
			float *row = partial + (split - 1) * 192 + lane;
			if constexpr (CPLX) { row[0] = coherent.x; row[64] = coherent.y; }
			else                { row[0] = coherent; }
			if constexpr (CW) row[128] = incoherent;
		}
		__syncthreads();
		if (split) return;
		for (uint32_t k = 1; k < (1u << p.split_shift); k++) {
			const float *row = partial + (k - 1) * 192 + lane;
			if constexpr (CPLX) { coherent.x += row[0]; coherent.y += row[64]; }
			else                { coherent += row[0]; }
			if constexpr (CW) incoherent += row[128];
		}
	}

	if (inside) {
		uint32_t sx, sy, szl, thread = threadIdx.x;
		asm volatile("" : "+v"(thread));
		voxel_of(thread, sx, sy, szl);
		uint64_t out_index = (uint64_t)p.size[0] * p.size[1] * szl + (uint64_t)p.size[0] * sy + sx;
		sample_t<CPLX> v = coherent;
		if constexpr (CW) v = v * (v / incoherent);                      /* coherency_weighting.glsl:36 */
		reinterpret_cast<sample_t<CPLX> *>(p.out)[out_index] = v;
	}
}

template <int FAMILY, int INTERP, bool CPLX, bool CW>
hipError_t launch_one(const BfDasArgs *a, hipStream_t s)
{
	uint32_t total   = a->blocks[0] * a->blocks[1] * a->blocks[2];
	uint32_t grid    = a->depth_major == 3u ? bf_plane_walk_blocks(a->blocks[0], a->blocks[1], a->band_rows) : ((total + 7u) / 8u) * 8u;
	uint32_t threads = a->split_shift ? 64u << a->split_shift : 256u;
	uint32_t lds     = a->split_shift ? ((1u << a->split_shift) - 1u) * 192u * (uint32_t)sizeof(float) : 0u;
	if (a->row_ends && INTERP != BF_INTERP_NEAREST) hipLaunchKernelGGL((das_factored_kernel<FAMILY, INTERP, CPLX, CW, true>), dim3(grid), dim3(threads), lds, s, *a);
	else                                            hipLaunchKernelGGL((das_factored_kernel<FAMILY, INTERP, CPLX, CW, false>), dim3(grid), dim3(threads), lds, s, *a);
	return hipGetLastError();
}

template <int FAMILY, int INTERP>
hipError_t launch_kind(const BfDasArgs *a, hipStream_t s)
{
	if (a->complex_data) return a->coherency_weighting ? launch_one<FAMILY, INTERP, true,  true >(a, s)
	                                                   : launch_one<FAMILY, INTERP, true,  false>(a, s);
	else                 return a->coherency_weighting ? launch_one<FAMILY, INTERP, false, true >(a, s)
	                                                   : launch_one<FAMILY, INTERP, false, false>(a, s);
}

template <int FAMILY>
hipError_t launch_interp(const BfDasArgs *a, hipStream_t s)
{
	switch (a->interpolation) {
	case BF_INTERP_NEAREST: return launch_kind<FAMILY, BF_INTERP_NEAREST>(a, s);
	case BF_INTERP_LINEAR:  return launch_kind<FAMILY, BF_INTERP_LINEAR >(a, s);
	case BF_INTERP_CUBIC:   return launch_kind<FAMILY, BF_INTERP_CUBIC  >(a, s);
	}
	return hipErrorInvalidValue;
}

} // namespace

/* RCA-family and FORCES/UFORCES frames only; the caller has checked that all transmits share
 * one receive orientation (RCA). */
extern "C" hipError_t bf_launch_das_factored(const BfDasArgs *a, hipStream_t s)
{
	switch (a->family) {
	case BF_DAS_RCA:    return launch_interp<BF_DAS_RCA>(a, s);
	case BF_DAS_FORCES: return launch_interp<BF_DAS_FORCES>(a, s);
	}
	return hipErrorInvalidValue;
}
