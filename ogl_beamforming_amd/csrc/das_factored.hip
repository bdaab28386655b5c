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
 *
 * WAVE-SPAN STAGING (template parameter SPAN, round 3; IQ samples, linear / cubic).  On a COARSE grid -- the
 * reference harness's own 512 x 1024 view plane has 0.23 mm pixels, 1.5 samples of delay per voxel along x -- the
 * 64 lanes of a gather instruction land in 64 different places of an RF row: the texture path serves such a wave
 * in ~25 clocks instead of 16 and a cubic term needs two (rocprofv3: 210 clocks per wave-term per SIMD, VALU 57 %
 * busy).  But the span a wave touches is short: for one (channel, transmit) row the indices of its 64 voxels lie
 * within [min R + min T, max R + max T], ~30-100 samples.  So the wave copies that span -- ONE coalesced
 * 1 KB LDS-DMA load (`buffer_load_dwordx4 ... lds`: 128 samples, no VGPR, no address arithmetic: the row and the
 * window start are scalars) -- into its own LDS slot and every lane reads its taps from there (4 x ds_read_b64 for
 * Catmull-Rom).  No block barrier: the slot is the wave's own; the loads of transmit a + 1 are in flight while
 * transmit a is consumed (two buffers of CH slots, counted s_waitcnt).  The window start is exact, not estimated:
 * per channel the wave reduces min / max of the receive index over the lanes inside the aperture, per transmit
 * (once per wave, kept in LDS) the floor of the minimum transmit index; a chunk whose spread does not fit 128
 * samples takes the gather loop instead, so nothing depends on a host-side bound.  Same arithmetic as the gather
 * loop: frames are bit-identical (tests/test_gpu_parity.py).  On request only (das path flag 0x40) since the end of
 * round 3: with all gathers of a transmit issued before the first is consumed (BF_ALL_GATHERS_ISSUED below) the
 * gather loop itself runs the harness planes at 15.7-17.0 ms against 17.3-20.1 with span staging.
 */
#include "das_common.h"

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
/* ... and per chunk of the wave-span variant (its loop waits for LDS-DMA and LDS reads, not for gathers: occupancy counts) */
#ifndef BF_SPAN_CHUNK
#define BF_SPAN_CHUNK(interp) ((interp) == BF_INTERP_LINEAR ? 6 : 4)
#endif

namespace {

template <bool CPLX, bool CW>
struct ChannelFactor {
	float index;           /* receive part of the sample index; -1e9 when the channel fails the f-number test */
	float re, im;          /* apod * e^{j phi(R)} (CPLX) -- re alone holds apod for real data */
	float apod;            /* for the incoherent sum */
	float spare;           /* wave-span staging, unchecked loop: the index with lanes outside the aperture parked inside the window */
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

/* wave-wide minimum / maximum as scalars (DPP row shifts: 4 VALU ops + 4 v_readlane, no LDS) */
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)       /* lanes with no source lane keep their own value */
{
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <bool MAX>
__device__ __forceinline__ float wave_extreme(float v)
{
	auto pick = [](float a, float b) { return MAX ? fmaxf(a, b) : fminf(a, b); };
	v = pick(v, dpp_f<0x111>(v));      /* row_shr:1 */
	v = pick(v, dpp_f<0x112>(v));      /* row_shr:2 */
	v = pick(v, dpp_f<0x114>(v));      /* row_shr:4 */
	v = pick(v, dpp_f<0x118>(v));      /* row_shr:8: lane 15 of every row of 16 holds the row's extreme */
	float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 15));
	float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
	float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 47));
	float r4 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
	return pick(pick(r1, r2), pick(r3, r4));
}

constexpr uint32_t kSpanSamples = 128;                      /* one LDS-DMA wave instruction: 64 lanes x 16 bytes */
constexpr uint32_t kSpanSlotBytes = kSpanSamples * 8;

template <int FAMILY, int INTERP, bool CPLX, bool CW, bool SPAN = false>
__global__ __launch_bounds__(1024) void das_factored_kernel(const BfDasArgs p)
{
	constexpr int      CH = SPAN ? BF_SPAN_CHUNK(INTERP) : BF_FACTORED_CHUNK(INTERP);
	constexpr uint32_t ES = CPLX ? 8 : 4;
	static_assert(!SPAN || (CPLX && INTERP != BF_INTERP_NEAREST), "wave-span staging: linear or cubic interpolation of IQ samples");
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
	uint32_t x = (bx << p.tile_shift[0]) + lx;
	uint32_t y = (by << p.tile_shift[1]) + ly;
	uint32_t zl = (bz << p.tile_shift[2]) + lz;
	bool inside = x < p.size[0] && y < p.size[1] && zl < p.z_count;
	[[maybe_unused]] const bool store = inside;
	if constexpr (SPAN) {
		/* the LDS-DMA loads and the wave reductions need every lane: lanes outside the grid repeat its last voxel (and
		 * store nothing); a wave with no voxel at all leaves (SPAN launches have no channel split, hence no barrier) */
		if (__builtin_amdgcn_ballot_w64(inside) == 0) return;
		x = x < p.size[0] ? x : p.size[0] - 1u;
		y = y < p.size[1] ? y : p.size[1] - 1u;
		zl = zl < p.z_count ? zl : p.z_count - 1u;
		inside = true;
	}

	sample_t<CPLX> coherent = zero_sample<CPLX>();
	float          incoherent = 0.f;

	if (inside) {
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

		/* SPAN: this wave's LDS -- 64 zero bytes (where invalid taps read), floor(min over the lanes of the transmit index) per
		 * transmit, two buffers of CH one-KB slots */
		[[maybe_unused]] uint32_t span_lds = 0, span_tfloor = 0, span_slots = 0;
		[[maybe_unused]] int span_tspread = 0;
		[[maybe_unused]] float span_tlo = 0.f, span_thi = 0.f;       /* extremes of the transmit index over the wave's lanes and all transmits */
		if constexpr (SPAN) {
			const uint32_t A_pad = ((uint32_t)A + 15u) & ~15u;
			const uint32_t per_wave = 64u + 4u * A_pad + 2u * CH * kSpanSlotBytes;
			span_lds    = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)factored_lds +
			              (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)) * per_wave;
			span_tfloor = span_lds + 64u;
			span_slots  = span_tfloor + 4u * A_pad;
			const uint32_t lane = tid & 63u;
			if (lane < 16u) *(__attribute__((address_space(3))) float *)(uintptr_t)(span_lds + 4u * lane) = 0.f;
			float spread = 0.f;
			span_tlo = __builtin_inff(); span_thi = -__builtin_inff();
			for (int a = first_transmit; a < A; a++) {
				const float t  = transmit_index(a);
				const float tmin = wave_extreme<false>(t), tmax = wave_extreme<true>(t);
				const float lo = __builtin_floorf(tmin), hi = __builtin_floorf(tmax);
				if (lane == 0) *(__attribute__((address_space(3))) int *)(uintptr_t)(span_tfloor + 4u * (uint32_t)a) = (int)lo;
				spread = fmaxf(spread, hi - lo);
				span_tlo = fminf(span_tlo, tmin); span_thi = fmaxf(span_thi, tmax);
			}
			span_tspread = spread < 1.0e6f ? (int)spread : 1000000;       /* (wave uniform; a NaN index never fits) */
			if (!(spread == spread)) span_tspread = 1000000;
		}

		for (int c0 = ch_begin; c0 < ch_end; c0 += CH) {
			ChannelFactor<CPLX, CW> R[CH];
			bool any = false;
			#pragma unroll
			for (int k = 0; k < CH; k++) {
				int   channel = c0 + k;
				float dx      = lateral - (float)channel * pitch;
				float a_arg   = __builtin_fabsf(dx * f_over_z);
				bool  pass    = a_arg < 0.5f && channel < ch_end;
				float dist    = hw_sqrt(dx * dx + zz);
				/* RCA: the time offset rides with the transmit term; FORCES: with the receive term
				 * (sample_index, das.glsl:126-130) */
				float index   = FAMILY == BF_DAS_RCA ? div_speed_of_sound(dist, p) * p.sampling_frequency
				                                     : (div_speed_of_sound(dist, p) + p.time_offset) * p.sampling_frequency;
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

			/* ---- wave-span staging: the taps of this chunk come out of the wave's LDS slots */
			bool span_done = false;
			if constexpr (SPAN) {
				int  rfl[CH], rsp[CH];
				bool fits = true, safe = true;
				#pragma unroll
				for (int k = 0; k < CH; k++) {
					const bool  pass = R[k].index > -1.0e8f;
					const float lo = wave_extreme<false>(pass ? R[k].index :  __builtin_inff());
					const float hi = wave_extreme<true >(pass ? R[k].index : -__builtin_inff());
					const bool  on = lo <= hi;                           /* some lane of the wave is inside this channel's aperture */
					const float flo = __builtin_floorf(lo);
					rfl[k] = on ? (int)flo : 0;
					rsp[k] = on ? (int)(__builtin_floorf(hi) - flo) : 0;
					/* taps floor - 1 .. floor + 2 of every lane inside [window start, + 128): receive spread + the wave's largest
					 * transmit spread + 6 (two floors, the sum's rounding, the taps) */
					fits = fits && (!on || (__builtin_floorf(hi) - flo) + (float)span_tspread <= (float)(kSpanSamples - 7));
					/* no lane inside the aperture can leave the RF row for any transmit (one sample of margin for the sum's rounding):
					 * the loop then runs without the range test of sample_rf, and lanes OUTSIDE the aperture -- their weight is zero
					 * and stays zero -- take the index of one inside it instead of being steered to the zero block */
					safe = safe && on && lo + span_tlo >= (INTERP == BF_INTERP_CUBIC ? 2.0f : 1.0f) &&
					       hi + span_thi < (float)(S - (INTERP == BF_INTERP_CUBIC ? 3 : 2));      /* (a chunk with a channel NO lane uses: checked loop) */
					if (on && !pass) R[k].spare = lo;
					else             R[k].spare = R[k].index;
				}
				if (fits && S >= (int)kSpanSamples) {
					const uint32_t lane16 = (tid & 63u) * 16u;
					const uint32_t chunk_rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)(((uint32_t)c0 * (uint32_t)A) * (uint32_t)S * ES));
					int ws[CH], ws_next[CH];
					/* lanes that load: 2 samples each.  The unchecked loop needs exactly [window start, + spread + taps): lanes beyond it are
					 * switched off for the load (the texture path charges a wave's LDS-DMA by its bytes), and its window is not clamped into the
					 * row -- what it may hold of a neighbouring row is never read.  The checked loop loads all 128 samples of a clamped window. */
					unsigned long long lanes_on[CH];
					#pragma unroll
					for (int k = 0; k < CH; k++) {
						int need = safe ? (rsp[k] + span_tspread + 8) / 2 : 64;
						need = need < 1 ? 1 : (need > 64 ? 64 : need);
						lanes_on[k] = need >= 64 ? ~0ull : ((1ull << need) - 1ull);
					}
					typedef int i32x4 __attribute__((ext_vector_type(4)));
					/* buffer resource over the whole DAS input, by hand for the asm statement: base, stride 0, bytes, raw 32-bit data format */
					const uint64_t rf_address = (uint64_t)(uintptr_t)p.rf;
					const i32x4 rsrc_words = {__builtin_amdgcn_readfirstlane((int)(uint32_t)rf_address), __builtin_amdgcn_readfirstlane((int)((uint32_t)(rf_address >> 32) & 0xffffu)),
					                          __builtin_amdgcn_readfirstlane((int)((uint32_t)C * (uint32_t)A * (uint32_t)S * ES)), 0x00020000};
					auto issue = [&](int a, int buf, int (&ws_out)[CH]) {
						const int tf = __builtin_amdgcn_readfirstlane(*(__attribute__((address_space(3))) int *)(uintptr_t)(span_tfloor + 4u * (uint32_t)a));
						#pragma unroll
						for (int k = 0; k < CH; k++) {
							int w = rfl[k] + tf - 2;
							if (!safe) w = w < 0 ? 0 : (w > S - (int)kSpanSamples ? S - (int)kSpanSamples : w);
							ws_out[k] = w;
							/* rows past the last channel lie behind the buffer: the DMA writes zeros, nobody reads them (a channel no lane
							 * of the wave uses is loaded all the same: the counted wait below wants CH loads per transmit) */
							const uint32_t soff = chunk_rows + ((uint32_t)k * (uint32_t)A + (uint32_t)a) * (uint32_t)S * ES + (uint32_t)w * ES;
							const uint32_t dst  = span_slots + (uint32_t)(buf * CH + k) * kSpanSlotBytes;
							uint32_t keep_m0;
							/* M0 = LDS address of the slot; EXEC = the lanes that load; both restored (the kernel runs with every lane on here) */
							asm volatile("s_mov_b32 %0, m0\n\t"
							             "s_mov_b32 m0, %1\n\t"
							             "s_mov_b64 exec, %2\n\t"
							             "s_nop 0\n\t"
							             "buffer_load_dwordx4 %3, %4, %5 offen lds\n\t"
							             "s_mov_b64 exec, -1\n\t"
							             "s_mov_b32 m0, %0"
							             : "=&s"(keep_m0) : "s"(dst), "s"(lanes_on[k]), "v"(lane16), "s"(rsrc_words), "s"(soff) : "memory");
						}
					};
					typedef __attribute__((address_space(3))) f32x2 lds2;
					auto consume = [&](auto checked, int buf, float t_index, f32x2 cs) {
						constexpr bool CHECK = decltype(checked)::value;
						float frac[CH]; uint32_t at[CH];
						#pragma unroll
						for (int k = 0; k < CH; k++) {
							const float index = t_index + (CHECK ? R[k].index : R[k].spare);
							frac[k] = hw_fract(index);
							const uint32_t ki = (uint32_t)(cvt_floor_i32(index) - (INTERP == BF_INTERP_CUBIC ? 1 : 0));
							const uint32_t slot = span_slots + (uint32_t)(buf * CH + k) * kSpanSlotBytes;
							if constexpr (CHECK) {
								/* valid: 0 <= index < S - 1 (linear), 1 <= index < S - 2 (cubic); inside the staged window by construction
								 * (the second test only keeps a violated bound from reading a neighbour's slot) */
								const uint32_t rel = ki - (uint32_t)ws[k];
								const bool ok = ki < (uint32_t)(INTERP == BF_INTERP_CUBIC ? S - 3 : S - 1) && rel <= kSpanSamples - (INTERP == BF_INTERP_CUBIC ? 4u : 2u);
								at[k] = ok ? slot + rel * ES : span_lds;
							} else {
								at[k] = (ki << 3) + (slot - (uint32_t)ws[k] * ES);   /* one v_lshl_add_u32: the bracket is a scalar */
							}
						}
						/* every read of the batch is issued before the first is consumed */
						if constexpr (INTERP == BF_INTERP_LINEAR) {
							f32x2 s0[CH], s1[CH];
							#pragma unroll
							for (int k = 0; k < CH; k++) { s0[k] = *(lds2 *)(uintptr_t)at[k]; s1[k] = *(lds2 *)(uintptr_t)(at[k] + 8u); }
							#pragma unroll
							for (int k = 0; k < CH; k++) {
								f32x2 sv = s0[k] + frac[k] * (s1[k] - s0[k]);
								acc1[k] += sv.x * cs;
								acc2[k] += sv.y * cs;
								if constexpr (CW) { f32x2 sq = sv * sv; part_abs[k] += hw_sqrt(sq.x + sq.y); }
							}
						} else {
							f32x2 q0[CH], q1[CH], q2[CH], q3[CH];
							#pragma unroll
							for (int k = 0; k < CH; k++) {
								q0[k] = *(lds2 *)(uintptr_t)at[k];          q1[k] = *(lds2 *)(uintptr_t)(at[k] + 8u);
								q2[k] = *(lds2 *)(uintptr_t)(at[k] + 16u);  q3[k] = *(lds2 *)(uintptr_t)(at[k] + 24u);
							}
							#pragma unroll
							for (int k = 0; k < CH; k++) {
								f32x2 T1 = 0.5f * (q2[k] - q0[k]), T2 = 0.5f * (q3[k] - q1[k]), D = q2[k] - q1[k];
								f32x2 c3 = (T1 + T2) - 2.0f * D;
								f32x2 c2 = (D - T1) - c3;
								float t  = frac[k];
								f32x2 sv = q1[k] + t * (T1 + t * (c2 + t * c3));
								acc1[k] += sv.x * cs;
								acc2[k] += sv.y * cs;
								if constexpr (CW) { f32x2 sq = sv * sv; part_abs[k] += hw_sqrt(sq.x + sq.y); }
							}
						}
					};
					issue(first_transmit, 0, ws);
					for (int a = first_transmit; a < A; a++) {
						const int buf = (a - first_transmit) & 1;
						const bool more = a + 1 < A;
						if (more) issue(a + 1, buf ^ 1, ws_next);
						float t_index = transmit_index(a);
						asm volatile("" : "+v"(t_index));       /* as in the gather loop: the index is a SUM of two rounded terms, not an fma */
						const float turns = hw_fract(turns_per_sample * t_index);
						const f32x2 cs = {hw_cos_turns(turns), hw_sin_turns(turns)};
						/* the CH loads of transmit a + 1 may stay in flight; everything older has landed */
						if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CH) : "memory");
						else      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
						if (safe) consume(std::false_type{}, buf, t_index, cs);
						else      consume(std::true_type{},  buf, t_index, cs);
						#pragma unroll
						for (int k = 0; k < CH; k++) ws[k] = ws_next[k];
					}
					span_done = true;
				}
			}

			for (int a = first_transmit; a < A && !span_done; a++) {
				float t_index = transmit_index(a);
				asm volatile("" : "+v"(t_index));           /* not fused into the per-channel adds: every kernel variant rounds the same way */
				float tc = 1.f, ts = 0.f;
				if constexpr (CPLX) {
					float turns = hw_fract(turns_per_sample * t_index);
					tc = hw_cos_turns(turns); ts = hw_sin_turns(turns);
				}

				const uint32_t row0 = ((uint32_t)c0 * (uint32_t)A + (uint32_t)a) * (uint32_t)S * ES;
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
							uint32_t ki = (uint32_t)cvt_floor_i32(index);                 /* valid: 0 <= index < S-1 */
							off[k] = ki < (uint32_t)(S - 1) ? row0 + (uint32_t)k * row_step + (ki << 3) : p.zero_offset;
						} else {
							uint32_t ki = (uint32_t)(cvt_floor_i32(index) - 1);           /* valid: 1 <= index < S-2 */
							off[k] = ki < (uint32_t)(S - 3) ? row0 + (uint32_t)k * row_step + (ki << 3) : p.zero_offset;
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
						for (int k = 0; k < CH; k++) { d0[k] = gather<f32x4_a8>(rf, off[k]); d1[k] = gather<f32x4_a8>(rf, off[k] + 16); }
						BF_ALL_GATHERS_ISSUED();
						#pragma unroll
						for (int k = 0; k < CH; k++) {
							/* Catmull-Rom Hermite (das.glsl:67-97) as a cubic in t by Horner:
							 * p = s1 + t (T1 + t (c2 + t c3)),  T1 = (s2-s0)/2, T2 = (s3-s1)/2,
							 * c3 = T1 + T2 - 2 (s2-s1),  c2 = (s2-s1) - T1 - c3 */
							f32x2 s0 = {d0[k].x, d0[k].y}, s1 = {d0[k].z, d0[k].w}, s2 = {d1[k].x, d1[k].y}, s3 = {d1[k].z, d1[k].w};
							f32x2 T1 = 0.5f * (s2 - s0), T2 = 0.5f * (s3 - s1), D = s2 - s1;
							f32x2 c3 = (T1 + T2) - 2.0f * D;
							f32x2 c2 = (D - T1) - c3;
							float t  = frac[k];
							f32x2 sv = s1 + t * (T1 + t * (c2 + t * c3));
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

	if (SPAN ? store : inside) {
		uint64_t out_index = (uint64_t)p.size[0] * p.size[1] * zl + (uint64_t)p.size[0] * y + x;
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
	if constexpr (CPLX && INTERP != BF_INTERP_NEAREST) {
		if (a->span_stage && !a->split_shift) {
			/* per wave: the zero block, one floor per transmit, two buffers of CH slots */
			constexpr uint32_t CH = BF_SPAN_CHUNK(INTERP);
			const uint32_t per_wave = 64u + 4u * (((uint32_t)a->acquisition_count + 15u) & ~15u) + 2u * CH * kSpanSlotBytes;
			auto kernel = das_factored_kernel<FAMILY, INTERP, CPLX, CW, true>;
			hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4u * per_wave));
			if (e != hipSuccess) return e;
			hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 4u * per_wave, s, *a);
			return hipGetLastError();
		}
	}
	hipLaunchKernelGGL((das_factored_kernel<FAMILY, INTERP, CPLX, CW>), dim3(grid), dim3(threads), lds, s, *a);
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
