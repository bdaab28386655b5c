/* das_tile.hip -- delay-and-sum for gfx950 (MI355X): the per-voxel factored kernel with BLOCK-WIDE LDS staging, for fine
 * grids on which receive and transmit delays both move with both tile axes -- 2-D plane-wave compounding (tx and rx on one
 * array axis: BASELINE config 2), view planes through row-column data, FORCES -- with cubic interpolation of IQ samples.
 *
 * Same arithmetic contract as das_factored.hip (shaders/das.glsl RCA :204-231, FORCES :288-321, cubic :67-97): the sample
 * index of a voxel is a receive term plus a transmit term, so a thread keeps the receive index of CH channels in registers,
 * walks the transmits computing {T index, e^{j phi(T)}} once per transmit and chunk, per (voxel, channel, transmit) term adds
 * the two indices, interpolates and rotate-accumulates, and applies the channel's apod e^{j phi(R)} once, when the chunk folds.
 *
 * das_factored.hip gathers the four taps of a term through L1: two wave64 gather instructions, 32.6 clk per CU per term
 * (tools/microbench.hip), which is what config 2 waits for (texture path 0.88 busy; 59 VALU clk per term beside it).  Neither
 * das_staged_cubic.hip's tables (T[a][v], R[c][u]: nothing factorises over the tile axes here) nor wave-span staging (a fine
 * grid: L1 already serves the gathers at their floor) apply.  Here a 1024-thread block owns a 64 x 16 tile of the image and
 * stages, per chunk of CH channels and group of AT transmits, the CH x AT RF windows its 1024 voxels touch -- as cubic
 * POLYNOMIALS (das_staged_cubic.hip: the Catmull-Rom segment between window samples j and j + 1 expanded around its middle,
 * four complex coefficients, 32 bytes; the staging thread gets its neighbours' samples by one-lane wave shifts):
 *   * per term: position p = R' + T' (both relative to the window: exact differences), y = p + M rounds it and leaves the
 *     element index in the low mantissa bits (M = 2^23 + element base; das_staged.hip explains the trick), one v_mul_u32_u24
 *     gives the LDS address, g = p - (y - M) the offset from the segment's middle; two aligned ds_read_b128, a three-step
 *     Horner chain and two packed fmas of rotate-accumulate;
 *   * the window of row (c, a) starts at floor(min R_c) + floor(min T_a) - 1, minima over the BLOCK's voxels, computed by
 *     the block itself (wave reductions + one LDS exchange): no host bound exists that could be wrong -- a chunk whose spread
 *     does not fit the window (near field, steep grids) takes das_factored.hip's gather loop instead, block-uniformly;
 *   * two LDS buffers of CH x AT windows: the loads of group g + 1 are in flight (registers) while group g is consumed,
 *     converted and written behind it; one barrier per group;
 *   * the transmit loop is rotated and unrolled by two: a transmit's eight reads are issued, the next transmit's delay, phase and
 *     addresses are computed while they fly, then the arithmetic;
 *   * sample_rf's range test (1 <= index < S - 2) per wave: waves that cannot leave the RF row run an unchecked loop, the
 *     others test every term and read a zero element instead.
 * One block per CU (131 KB of windows, <= 128 VGPRs).  No MFMA: gather-accumulate.
 */
#include "das_common.h"

namespace {

constexpr int      kTileCH      = 4;            /* channels per register chunk */
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) int   lds_i32;

constexpr uint32_t kTileThreads = 1024;
constexpr uint32_t kTileElems   = 2048;         /* window elements per LDS buffer: CH x AT x W; two per thread */

/* das.glsl:187-202 with the per-transmit constants precomputed (das_factored.hip) */
__device__ __forceinline__ float tile_transmit_distance(const BfTransmit &t, float wx, float wy, float wz)
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

/* minimum (MAX: maximum) over each row of 16 lanes, valid in lane 15 of the row: four fused DPP steps (lanes with no source lane
 * are left alone).  Written out: from fminf() over a DPP move hipcc makes four instructions a step (copy, shift, canonicalise,
 * min), and these reductions run per channel and wave.  The s_nop are the two wait states between a VALU write and a DPP read
 * of the same register, which nobody inserts inside an asm block. */
#define BF_DPP_STEP(op, ctrl) "s_nop 1\n\t" op " %0, %0, %0 " ctrl "\n\t"
template <bool MAX>
__device__ __forceinline__ float row16_extreme(float v)
{
	if constexpr (MAX)
		asm(BF_DPP_STEP("v_max_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf") BF_DPP_STEP("v_max_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
		    BF_DPP_STEP("v_max_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf") BF_DPP_STEP("v_max_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf") : "+v"(v));
	else
		asm(BF_DPP_STEP("v_min_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf") BF_DPP_STEP("v_min_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf")
		    BF_DPP_STEP("v_min_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf") BF_DPP_STEP("v_min_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf") : "+v"(v));
	return v;
}
/* minimum (MAX: maximum) over the wave, valid in lane 63: the row steps, then the last lane of a row into the rows behind it
 * (gfx9's row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) */
template <bool MAX>
__device__ __forceinline__ float wave64_extreme_lane63(float v)
{
	v = row16_extreme<MAX>(v);
	if constexpr (MAX)
		asm(BF_DPP_STEP("v_max_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf") BF_DPP_STEP("v_max_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf") : "+v"(v));
	else
		asm(BF_DPP_STEP("v_min_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf") BF_DPP_STEP("v_min_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf") : "+v"(v));
	return v;
}
#undef BF_DPP_STEP
template <bool MAX>
__device__ __forceinline__ float wave64_extreme(float v)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wave64_extreme_lane63<MAX>(v)), 63));
}

/* 32 x the low 24 bits: the byte address of the polynomial element whose index sits in the low mantissa bits of y */
__device__ __forceinline__ uint32_t element_address(uint32_t y_bits)
{
	uint32_t at;
	asm("v_mul_u32_u24 %0, 32, %1" : "=v"(at) : "v"(y_bits));
	return at;
}

/* LDS (byte address 0 = start of the dynamic segment; the kernel has no static LDS):
 *   [0, 64)                         two unused 32-byte elements (a rounded position of -1 lands there, never consumed)
 *   [64, 64 + 2 x 2048 x 32)        two buffers of CH x AT windows of W polynomial elements
 *   then one zero element (checked loop), tfl[A padded to 16] (floor of the block's smallest transmit index, per transmit),
 *   exch[2][CH][16] (per-wave receive extremes of a chunk) */
template <int FAMILY, bool CW, int WS>
__global__ __launch_bounds__(1024, 4) void das_tile_kernel(const BfDasArgs p)
{
	/* Four channels per chunk.  (Eight would halve the per-transmit arithmetic a term carries -- about 3 of its 18 VALU instructions --
	 * but need some 140 registers in the transmit loop: tried, 60 spilled.) */
	constexpr int      CH = kTileCH;
	constexpr uint32_t W  = 1u << WS;
	constexpr uint32_t AT = kTileElems / (CH * W);            /* transmits per staged group: 16 (W = 32) or 8 (W = 64) */
	constexpr uint32_t ES = 8;
	extern __shared__ __attribute__((aligned(16))) unsigned char tile_lds[];

	/* blockIdx -> tile, thread -> voxel: das_factored.hip */
	const uint32_t total = p.blocks[0] * p.blocks[1] * p.blocks[2];
	const uint32_t bid   = blockIdx.x;
	const uint32_t per   = (total + 7u) / 8u;
	const uint32_t tile  = (bid & 7u) * per + (bid >> 3);
	if (p.depth_major != 3u && tile >= total) return;
	uint32_t bx, by, bz;
	if (p.depth_major == 3u) {
		bz = 0;
		if (!bf_plane_walk(bid, p.blocks[0], p.blocks[1], p.band_rows, bx, by)) return;     /* whole block */
	} else if (p.depth_major == 2u) {
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
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
	const uint32_t lx  = tid & ((1u << p.tile_shift[0]) - 1u);
	const uint32_t ly  = (tid >> p.tile_shift[0]) & ((1u << p.tile_shift[1]) - 1u);
	const uint32_t lz  = (tid >> (p.tile_shift[0] + p.tile_shift[1])) & ((1u << p.tile_shift[2]) - 1u);
	uint32_t x = (bx << p.tile_shift[0]) + lx, y = (by << p.tile_shift[1]) + ly, zl = (bz << p.tile_shift[2]) + lz;
	/* every thread takes part in the staging and the reductions: threads outside the grid repeat its last voxel (and store nothing) */
	x = x < p.size[0] ? x : p.size[0] - 1u;
	y = y < p.size[1] ? y : p.size[1] - 1u;
	zl = zl < p.z_count ? zl : p.z_count - 1u;

	const int   S = p.sample_count, A = p.acquisition_count, C = p.channel_count;
	const float turns_per_sample = p.demodulation_frequency * p.inv_sampling_frequency;
	float wx, wy, wz, xx, xy, xz;
	{
		const uint32_t z = p.z_first + zl;
		const float px = (float)x / fmaxf(1.0f, (float)p.size[0] - 1.0f);       /* das.glsl:374-376 */
		const float py = (float)y / fmaxf(1.0f, (float)p.size[1] - 1.0f);
		const float pz = (float)z / fmaxf(1.0f, (float)p.size[2] - 1.0f);
		m4_point(p.voxel_transform, px, py, pz, wx, wy, wz);
	}
	if constexpr (FAMILY == BF_DAS_RCA) m4_point(p.xdc_transform, wx, wy, wz, xx, xy, xz);
	else { xx = wx; xy = wy; xz = wz; }
	const float zz = xz * xz;
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
	const int first_transmit = FAMILY == BF_DAS_RCA ? 0 : (p.sparse != 0);
	float transmit_yz_squared = 0.f;
	if constexpr (FAMILY != BF_DAS_RCA) {
		float dy = xy - p.pitch[1] * (float)C * 0.5f;
		transmit_yz_squared = dy * dy + zz;
	}
	/* (through the constant address space: with buffer loads in flight hipcc would otherwise wait vmcnt(0) for vector copies of
	 * these wave-uniform reads -- das_factored.hip) */
	typedef __attribute__((address_space(4))) const f32x4   const_f32x4;
	typedef __attribute__((address_space(4))) const int16_t const_i16;
	const_f32x4 *transmits_c = (const_f32x4 *)(uintptr_t)p.transmits;
	const_i16   *sparse_c    = (const_i16 *)(uintptr_t)p.sparse_elements;
	auto transmit_index = [&](int a) -> float {
		if constexpr (FAMILY == BF_DAS_RCA) {
			const f32x4 t_lo = transmits_c[2 * a], t_hi = transmits_c[2 * a + 1];
			BfTransmit t;
			t.sin_a = t_lo.x; t.cos_a = t_lo.y; t.focus_x = t_lo.z; t.focus_z = t_lo.w;
			{ const float f = t_hi.x; t.flags = __builtin_bit_cast(uint32_t, f); }
			return (div_speed_of_sound(tile_transmit_distance(t, wx, wy, wz), p) + p.time_offset) * p.sampling_frequency;
		} else {
			float tx_channel = p.sparse ? (float)sparse_c[a - first_transmit] : (float)a;
			float tdx        = xx - p.pitch[0] * tx_channel;
			return div_speed_of_sound(hw_sqrt(transmit_yz_squared + tdx * tdx) * p.sampling_frequency, p);   /* das.glsl:312 */
		}
	};

	/* ---- LDS */
	const uint32_t A_pad = ((uint32_t)A + 15u) & ~15u;
	const uint32_t stage_base = 64u;                                   /* element e of buffer b sits at 64 + (b * 2048 + e) * 32 */
	const uint32_t zero_at    = stage_base + 2u * kTileElems * 32u;    /* one zero element */
	const uint32_t tfl_at     = zero_at + 32u;
	const uint32_t exch_at    = tfl_at + 4u * A_pad;                   /* [2][CH][16] floats */
	auto lds_f = [](uint32_t at) -> lds_f32 * { return (lds_f32 *)(uintptr_t)at; };
	auto lds_i = [](uint32_t at) -> lds_i32 * { return (lds_i32 *)(uintptr_t)at; };
	if (tid < 8) *lds_f(zero_at + 4u * tid) = 0.f;

	/* ---- per transmit: floor of the block's smallest transmit index, and the block's largest spread.  Wave extremes meet in the
	 * (still unused) staging area: [a][wave] minima at byte 64, maxima behind them. */
	{
		const uint32_t mins = stage_base, maxs = stage_base + 4u * 16u * A_pad;
		for (int a = first_transmit; a < A; a++) {
			const float t = transmit_index(a);
			const float lo = wave64_extreme<false>(t), hi = wave64_extreme<true>(t);
			if (lane == 0) { *lds_f(mins + 4u * ((uint32_t)a * 16u + wave)) = lo; *lds_f(maxs + 4u * ((uint32_t)a * 16u + wave)) = hi; }
		}
		__syncthreads();
		float spread = 0.f, tlo_all = __builtin_inff(), thi_all = -__builtin_inff();
		for (uint32_t a = (uint32_t)first_transmit + tid; a < (uint32_t)A; a += kTileThreads) {
			float lo = __builtin_inff(), hi = -__builtin_inff();
			for (uint32_t w = 0; w < 16u; w++) { lo = fminf(lo, *lds_f(mins + 4u * (a * 16u + w))); hi = fmaxf(hi, *lds_f(maxs + 4u * (a * 16u + w))); }
			const float fl = __builtin_floorf(lo);
			*lds_i(tfl_at + 4u * a) = (int)fl;
			spread = fmaxf(spread, __builtin_floorf(hi) - fl);
			tlo_all = fminf(tlo_all, lo); thi_all = fmaxf(thi_all, hi);
		}
		/* the three block-wide scalars: through the exchange area (threads that own no transmit contribute the identities) */
		const float s1 = wave64_extreme<true>(spread), s2 = wave64_extreme<false>(tlo_all), s3 = wave64_extreme<true>(thi_all);
		if (lane == 0) { *lds_f(exch_at + 4u * wave) = s1; *lds_f(exch_at + 4u * (16u + wave)) = s2; *lds_f(exch_at + 4u * (32u + wave)) = s3; }
		__syncthreads();
	}
	float tspread_f = 0.f, t_lo = __builtin_inff(), t_hi = -__builtin_inff();
	for (uint32_t w = 0; w < 16u; w++) {
		tspread_f = fmaxf(tspread_f, *lds_f(exch_at + 4u * w));
		t_lo = fminf(t_lo, *lds_f(exch_at + 4u * (16u + w)));
		t_hi = fmaxf(t_hi, *lds_f(exch_at + 4u * (32u + w)));
	}
	tspread_f = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tspread_f)));
	t_lo      = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t_lo)));
	t_hi      = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t_hi)));
	const bool t_finite = tspread_f == tspread_f && tspread_f < 1.0e6f;
	__syncthreads();                                                   /* the staging area is free again */

	/* ---- staging: thread tid owns elements e = tid and tid + 1024 of a group: window e / W (= k * AT + a_local), sample e % W */
	const __amdgpu_buffer_rsrc_t rf_rsrc = __builtin_amdgcn_make_buffer_rsrc(
		const_cast<void *>(p.rf), 0, (int)((uint32_t)C * (uint32_t)A * (uint32_t)S * ES), 0x00020000);
	auto lane_shift = [](float v, bool up) {
		return up ? __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true))    /* lane i <- i + 1 */
		          : __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));   /* lane i <- i - 1 */
	};

	const char *rf = (const char *)p.rf;
	sample_t<true> coherent = {0.f, 0.f};
	float incoherent = 0.f;

	for (int c0 = 0; c0 < C; c0 += CH) {
		/* receive index of the chunk's channels (das_factored.hip); the weights -- apodization and e^{j phi(R)} -- are only needed when the
		 * chunk's sums are folded in, and are computed there: twelve registers the transmit loop does not have to carry */
		float r_index[CH];
		#pragma unroll
		for (int k = 0; k < CH; k++) {
			const int   channel = c0 + k;
			const float dx      = lateral - (float)channel * pitch;
			const float a_arg   = __builtin_fabsf(dx * f_over_z);
			const bool  pass    = a_arg < 0.5f && channel < C;
			const float dist    = hw_sqrt(dx * dx + zz);
			const float index   = FAMILY == BF_DAS_RCA ? div_speed_of_sound(dist, p) * p.sampling_frequency
			                                           : (div_speed_of_sound(dist, p) + p.time_offset) * p.sampling_frequency;
			r_index[k] = pass ? index : -1.0e9f;
		}
		/* block-wide extremes of the receive index over the voxels inside each channel's aperture; and whether this wave's lanes inside the
		 * apertures stay inside the RF row for every transmit (one sample of margin for the sum's rounding): lane 63's verdict counts */
		bool wave_safe = true;
		__syncthreads();                                               /* (the exchange area's previous readers are done) */
		#pragma unroll
		for (int k = 0; k < CH; k++) {
			const bool  pass = r_index[k] > -1.0e8f;
			const float wlo = wave64_extreme_lane63<false>(pass ? r_index[k] :  __builtin_inff());       /* (valid in lane 63) */
			const float whi = wave64_extreme_lane63<true >(pass ? r_index[k] : -__builtin_inff());
			if (lane == 63) { *lds_f(exch_at + 4u * ((uint32_t)k * 16u + wave)) = wlo; *lds_f(exch_at + 4u * ((uint32_t)(CH + k) * 16u + wave)) = whi; }
			wave_safe = wave_safe && (wlo > whi || (wlo + t_lo >= 2.0f && whi + t_hi < (float)(S - 3)));
		}
		__syncthreads();
		float rlo[CH], rhi[CH];
		bool  fits = t_finite, some = false;
		#pragma unroll
		for (int k = 0; k < CH; k++) {
			/* every wave reduces the 16 wave values itself (lanes 0-15 hold them; one row of DPP shifts) */
			const float vlo = *lds_f(exch_at + 4u * ((uint32_t)k * 16u + (lane & 15u)));
			const float vhi = *lds_f(exch_at + 4u * ((uint32_t)(CH + k) * 16u + (lane & 15u)));
			rlo[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, row16_extreme<false>(vlo)), 15));
			rhi[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, row16_extreme<true >(vhi)), 15));
			const bool on = rlo[k] <= rhi[k];
			some |= on;
			/* segment floor(index) - (window start) must lie in [1, W - 3]: receive spread + transmit spread + 5 (two floors, the sum's
			 * rounding, the early sample, the taps) */
			fits = fits && (!on || (__builtin_floorf(rhi[k]) - __builtin_floorf(rlo[k])) + tspread_f <= (float)(W - 6u));
		}
		if (!some) continue;                                           /* block uniform: nobody is inside any aperture of the chunk */
		if (tid == 0 && p.tile_counters) atomicAdd(p.tile_counters + (fits ? 0 : 1), 1u);

		float part_abs[CH];
		f32x2 acc[CH];                                                 /* sum over the transmits of s e^{j phi(T)} */
		#pragma unroll
		for (int k = 0; k < CH; k++) { acc[k] = f32x2{0.f, 0.f}; part_abs[k] = 0.f; }

		if (fits) {
			/* ---------------- staged path */
			int   rfl[CH];
			f32x2 r_rel[CH / 2];
			#pragma unroll
			for (int k = 0; k < CH; k++) {
				const bool on = rlo[k] <= rhi[k];
				const float flo = on ? __builtin_floorf(rlo[k]) : 0.f;
				rfl[k] = __builtin_amdgcn_readfirstlane((int)flo);
				const bool pass = r_index[k] > -1.0e8f;
				/* lanes outside the aperture (weight zero, and it stays zero) take the smallest index inside it: their reads stay in the
				 * window; a channel nobody uses reads from the window's start */
				const float idx = pass ? r_index[k] : (on ? rlo[k] : 0.f);
				r_rel[k / 2][k & 1] = idx - flo;                          /* exact */
			}
			const uint32_t chunk_rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)c0 * (uint32_t)A));
			const uint32_t groups = ((uint32_t)(A - first_transmit) + AT - 1u) / AT;
			/* window of element e: k = e / (AT W), a_local = (e / W) % AT; the thread's two elements are tid and tid + 1024, and AT W = 512:
			 * both channels are wave uniform.  (The floors go through an opaque copy: hipcc otherwise folds the selects below into ONE load
			 * from rfl[] at a run-time index, keeps rfl[] in memory for it and "promotes" that memory to 16 KB of static LDS in front of
			 * the dynamic segment, whose address 0 this kernel relies on.) */
			const uint32_t my_j = tid & (W - 1u);
			const uint32_t e0 = tid, e1 = tid + kTileThreads;
			const uint32_t k0 = wave / (AT * W / 64u), k1 = k0 + kTileThreads / (AT * W);        /* k0 < CH / 2 <= k1 */
			const uint32_t al0 = (e0 / W) % AT, al1 = (e1 / W) % AT;
			int rfl_0 = 0, rfl_1 = 0;
			#pragma unroll
			for (int k = 0; k < CH / 2; k++) {
				int lower = rfl[k], upper = rfl[CH / 2 + k];
				asm volatile("" : "+v"(lower), "+v"(upper));
				rfl_0 = k0 == (uint32_t)k ? lower : rfl_0;
				rfl_1 = k1 == (uint32_t)(CH / 2 + k) ? upper : rfl_1;
			}
			auto load_one = [&](uint32_t g, uint32_t kk, uint32_t al, int rk) -> f32x2 {
				const uint32_t a = (uint32_t)first_transmit + g * AT + al;
				uint32_t off = 0x80000000u;                                /* transmits past the last (a ragged final group) stage zeros */
				if (a < (uint32_t)A && c0 + (int)kk < C) {
					const int first = rk + *lds_i(tfl_at + 4u * a) - 1 + (int)my_j;              /* window sample j = sample floor(rmin) + floor(tmin) - 1 + j */
					/* a window that starts before its row or ends behind it holds samples of the neighbouring rows (or zeros beyond the buffer):
					 * never consumed -- unchecked waves stay inside their row, checked waves test every term */
					off = ((chunk_rows + kk * (uint32_t)A + a) * (uint32_t)S + (uint32_t)first) * ES;
				}
				i32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rf_rsrc, (int)off, 0, 0);
				return __builtin_bit_cast(f32x2, v);
			};
			auto store_one = [&](uint32_t buf, uint32_t e, f32x2 s1) {
				/* das_staged_cubic.hip: Catmull-Rom segment [j, j + 1] (das.glsl:67-97) re-expanded around t = 1/2 */
				const float p1x = s1.x, p1y = s1.y;
				const float p2x = lane_shift(p1x, true),  p2y = lane_shift(p1y, true);
				const float p3x = lane_shift(p2x, true),  p3y = lane_shift(p2y, true);
				const float p0x = lane_shift(p1x, false), p0y = lane_shift(p1y, false);
				const f32x2 P0 = {p0x, p0y}, P1 = {p1x, p1y}, P2 = {p2x, p2y}, P3 = {p3x, p3y};
				/* p(1/2 + g) = b0 + b1 g + b2 g^2 + a3 g^3, the four coefficients straight from the taps:
				 *   b0 = (9 (P1 + P2) - (P0 + P3)) / 16     b1 = (11 (P2 - P1) - (P3 - P0)) / 8
				 *   b2 = ((P0 + P3) - (P1 + P2)) / 4        a3 = ((P3 - P0) - 3 (P2 - P1)) / 2 */
				const f32x2 S12 = P1 + P2, S03 = P0 + P3, D12 = P2 - P1, D03 = P3 - P0;
				const f32x2 b0 = 0.5625f * S12 - 0.0625f * S03;
				const f32x2 b1 = 1.375f * D12 - 0.125f * D03;
				const f32x2 b2 = 0.25f * (S03 - S12);
				const f32x2 a3 = 0.5f * D03 - 1.5f * D12;
				const uint32_t at = stage_base + (buf * kTileElems + e) * 32u;
				*(lds_f32x4 *)(uintptr_t)at         = f32x4{b0.x, b0.y, b1.x, b1.y};
				*(lds_f32x4 *)(uintptr_t)(at + 16u) = f32x4{b2.x, b2.y, a3.x, a3.y};
			};
			/* a transmit of a group: its index relative to the window start, its phase factor; and for four of the chunk's channels the LDS
			 * address of the term's polynomial and the offset from the segment's middle */
			struct Tx   { float t_rel; f32x2 cs; int tf; };
			struct Four { f32x2 gm[2]; uint32_t at[4]; };
			auto transmit = [&](uint32_t g, uint32_t al, int tf_lanes, int tfh_lanes) -> Tx {
				Tx tx;
				const int a = first_transmit + (int)(g * AT + al);
				float t_index = transmit_index(a);
				asm volatile("" : "+v"(t_index));
				tx.tf = __builtin_amdgcn_readlane(tf_lanes, (int)al);
				const float turns = hw_fract(turns_per_sample * t_index);
				tx.cs = f32x2{hw_cos_turns(turns), hw_sin_turns(turns)};
				/* + 1/2: y below rounds to the SEGMENT (the window starts one sample early); exact (multiples of an ulp of t_index, small) */
				tx.t_rel = t_index - __builtin_bit_cast(float, __builtin_amdgcn_readlane(tfh_lanes, (int)al));
				return tx;
			};
			auto four = [&](auto checked, const Tx &tx, uint32_t buf, uint32_t al) -> Four {
				constexpr bool CHECK = decltype(checked)::value;
				Four f;
				#pragma unroll
				for (int h = 0; h < 2; h++) {
					/* element index 2 + buf * 2048 + (k * AT + al) * W + segment in the low mantissa bits of y = p + M */
					const uint32_t m0 = 0x4B000002u + buf * kTileElems + ((uint32_t)(2 * h) * AT + al) * W, m1 = m0 + AT * W;
					const f32x2 M = {__builtin_bit_cast(float, m0), __builtin_bit_cast(float, m1)};
					const f32x2 pos = r_rel[h] + tx.t_rel;
					const f32x2 yv = pos + M;
					f.gm[h] = pos - (yv - M);
					/* (copies first: hipcc's __builtin_bit_cast of a vector ELEMENT reads element 0 whichever one is named) */
					const float    y0 = yv.x, y1 = yv.y;
					const uint32_t yb[2] = {__builtin_bit_cast(uint32_t, y0), __builtin_bit_cast(uint32_t, y1)};
					uint32_t at[2] = {element_address(yb[0]), element_address(yb[1])};
					if constexpr (CHECK) {
						/* absolute tap k_abs = segment - 1 + the two floors; valid for 1 <= k_abs < S - 2; and -- the window is the block's own
						 * construction -- never outside it: a segment beyond [1, W - 3] would be a bug in this kernel, so it reads zeros too */
						const uint32_t seg[2]   = {yb[0] - m0, yb[1] - m1};
						const uint32_t k_abs[2] = {(uint32_t)((int)seg[0] - 1 + rfl[2 * h] + tx.tf), (uint32_t)((int)seg[1] - 1 + rfl[2 * h + 1] + tx.tf)};
						at[0] = ((k_abs[0] - 1u) < (uint32_t)(S - 3) && (seg[0] - 1u) <= W - 4u) ? at[0] : zero_at;
						at[1] = ((k_abs[1] - 1u) < (uint32_t)(S - 3) && (seg[1] - 1u) <= W - 4u) ? at[1] : zero_at;
					}
					f.at[2 * h] = at[0]; f.at[2 * h + 1] = at[1];
				}
				return f;
			};
			/* one group of AT transmits out of buffer `buf`, four terms (one transmit, four channels) a step: the reads of a step's four
			 * polynomials are in flight while the addresses of the next step's are computed */
			auto consume = [&](auto checked, uint32_t g, uint32_t buf) {
				const uint32_t a0 = (uint32_t)first_transmit + g * AT;
				const uint32_t n  = (uint32_t)A - a0 < AT ? (uint32_t)A - a0 : AT;            /* block uniform; >= 1 */
				uint32_t a_lane = a0 + (lane & (AT - 1u));
				a_lane = a_lane < (uint32_t)A ? a_lane : (uint32_t)A - 1u;
				const int tf_lanes  = *lds_i(tfl_at + 4u * a_lane);      /* lane l: floor of the block's smallest index of transmit a0 + l % AT */
				const int tfh_lanes = __builtin_bit_cast(int, (float)tf_lanes - 0.5f);
				/* the arithmetic of four terms out of the registers the reads filled */
				auto reads = [&](const Four &f, f32x4 (&lo)[4], f32x4 (&hi)[4]) {
					#pragma unroll
					for (int j = 0; j < 4; j++) { lo[j] = *(lds_f32x4 *)(uintptr_t)f.at[j]; hi[j] = *(lds_f32x4 *)(uintptr_t)(f.at[j] + 16u); }
				};
				auto sums = [&](const Four &f, const Tx &tx, const f32x4 (&lo)[4], const f32x4 (&hi)[4]) {
					const f32x2 cs = tx.cs, csr = {-cs.y, cs.x};
					#pragma unroll
					for (int j = 0; j < 4; j++) {
						const float gk = f.gm[j / 2][j & 1];
						f32x2 sv = f32x2{hi[j].z, hi[j].w} * gk + f32x2{hi[j].x, hi[j].y};
						sv = sv * gk + f32x2{lo[j].z, lo[j].w};
						sv = sv * gk + f32x2{lo[j].x, lo[j].y};
						acc[j] += sv.x * cs;                             /* s e^{j phi}: the real part times (cos, sin) ... */
						acc[j] += sv.y * csr;                            /* ... the imaginary part times (-sin, cos) */
						if constexpr (CW) part_abs[j] += hw_sqrt(__builtin_fmaf(sv.y, sv.y, sv.x * sv.x));
					}
				};
				/* two transmits per turn, their terms in two fixed sets of registers (handing `next` over to `cur` costs six moves a transmit) */
				Tx   tx   = transmit(g, 0, tf_lanes, tfh_lanes), tx_odd = tx;
				Four even = four(checked, tx, buf, 0), odd = even;
				for (uint32_t al = 0; ; al += 2u) {
					f32x4 lo[4], hi[4];
					reads(even, lo, hi);
					bool more = al + 1u < n;                                     /* block uniform */
					if (more) { tx_odd = transmit(g, al + 1u, tf_lanes, tfh_lanes); odd = four(checked, tx_odd, buf, al + 1u); }
					sums(even, tx, lo, hi);
					if (!more) break;
					reads(odd, lo, hi);
					more = al + 2u < n;
					if (more) { tx = transmit(g, al + 2u, tf_lanes, tfh_lanes); even = four(checked, tx, buf, al + 2u); }
					sums(odd, tx_odd, lo, hi);
					if (!more) break;
				}
			};
			/* (the range-checked and the unchecked loops as two copies of the whole group loop -- the same barriers in both: accumulators
			 * that meet after every group cost sixteen moves each time) */
			auto run_groups = [&](auto checked) {
				f32x2 in0 = load_one(0, k0, al0, rfl_0), in1 = load_one(0, k1, al1, rfl_1);
				__syncthreads();                       /* the previous chunk's last group has been consumed by everyone */
				store_one(0, e0, in0); store_one(0, e1, in1);
				__syncthreads();
				for (uint32_t g = 0; g < groups; g++) {
					const uint32_t buf = g & 1u;
					const bool more = g + 1 < groups;
					if (more) { in0 = load_one(g + 1, k0, al0, rfl_0); in1 = load_one(g + 1, k1, al1, rfl_1); }      /* in flight during the arithmetic */
					consume(checked, g, buf);
					if (more) { store_one(buf ^ 1u, e0, in0); store_one(buf ^ 1u, e1, in1); }    /* the other buffer: its readers passed the barrier below a group ago */
					__syncthreads();
				}
			};
			if (__builtin_amdgcn_readlane((int)wave_safe, 63)) run_groups(std::false_type{});
			else                                               run_groups(std::true_type{});
		} else {
			/* ---------------- das_factored.hip's gather loop (near field, steep grids: the spread does not fit the window) */
			for (int a = first_transmit; a < A; a++) {
				float t_index = transmit_index(a);
				asm volatile("" : "+v"(t_index));
				const float turns = hw_fract(turns_per_sample * t_index);
				const f32x2 cs = {hw_cos_turns(turns), hw_sin_turns(turns)}, csr = {-cs.y, cs.x};
				const uint32_t row0 = ((uint32_t)c0 * (uint32_t)A + (uint32_t)a) * (uint32_t)S * ES;
				const uint32_t row_step = (uint32_t)A * (uint32_t)S * ES;
				{
					float frac[4]; uint32_t off[4];
					#pragma unroll
					for (int j = 0; j < 4; j++) {
						const int   k = j;
						const float index = t_index + r_index[k];
						frac[j] = hw_fract(index);
						const uint32_t ki = (uint32_t)(cvt_floor_i32(index) - 1);           /* valid: 1 <= index < S-2 */
						off[j] = ki < (uint32_t)(S - 3) ? row0 + (uint32_t)k * row_step + (ki << 3) : p.zero_offset;
					}
					f32x4 d0[4], d1[4];
					#pragma unroll
					for (int j = 0; j < 4; j++) { d0[j] = gather<f32x4_a8>(rf, off[j]); d1[j] = gather_at<f32x4_a8, 16>(rf, off[j]); }
					__builtin_amdgcn_sched_barrier(0);                            /* all eight issued before the first is consumed (das_factored.hip) */
					#pragma unroll
					for (int j = 0; j < 4; j++) {
						f32x2 s0 = {d0[j].x, d0[j].y}, s1 = {d0[j].z, d0[j].w}, s2 = {d1[j].x, d1[j].y}, s3 = {d1[j].z, d1[j].w};
						float w0, w1, w2, w3;
						bf_catmull_rom(frac[j], w0, w1, w2, w3);                     /* (das_factored.hip: tap weights, not a Horner cubic in the samples) */
						f32x2 sv = w0 * s0 + w1 * s1 + w2 * s2 + w3 * s3;
						acc[j] += sv.x * cs;
						acc[j] += sv.y * csr;
						if constexpr (CW) { f32x2 sq = sv * sv; part_abs[j] += hw_sqrt(sq.x + sq.y); }
					}
				}
			}
		}

		/* fold the chunk in: sum_a s e^{j phi(T)} times the channel's apod e^{j phi(R)} (das_factored.hip).  The receive index is worked
		 * out again (the same expressions, the same value): carried through the transmit loop it would cost a register per channel */
		#pragma unroll
		for (int k = 0; k < CH; k++) {
			const int   channel = c0 + k;
			float dx = lateral - (float)channel * pitch;
			asm volatile("" : "+v"(dx));
			const float a_arg = __builtin_fabsf(dx * f_over_z);
			const bool  pass  = a_arg < 0.5f && channel < C;
			const float dist  = hw_sqrt(dx * dx + zz);
			const float index = FAMILY == BF_DAS_RCA ? div_speed_of_sound(dist, p) * p.sampling_frequency
			                                         : (div_speed_of_sound(dist, p) + p.time_offset) * p.sampling_frequency;
			const float apod  = pass ? apodize(a_arg) : 0.f;
			const float turns = hw_fract(turns_per_sample * index);
			const float r_re  = apod * hw_cos_turns(turns), r_im = apod * hw_sin_turns(turns);
			const f32x2 part  = acc[k];
			coherent.x += r_re * part.x - r_im * part.y;
			coherent.y += r_im * part.x + r_re * part.y;
			if constexpr (CW) incoherent += apod * part_abs[k];
		}
	}

	/* (the voxel is worked out again rather than carried through the loops) */
	{
		uint32_t t = threadIdx.x;
		asm volatile("" : "+v"(t));                                      /* (or hipcc keeps the first computation's registers alive instead) */
		const uint32_t sx = (bx << p.tile_shift[0]) + (t & ((1u << p.tile_shift[0]) - 1u));
		const uint32_t sy = (by << p.tile_shift[1]) + ((t >> p.tile_shift[0]) & ((1u << p.tile_shift[1]) - 1u));
		const uint32_t sz = (bz << p.tile_shift[2]) + ((t >> (p.tile_shift[0] + p.tile_shift[1])) & ((1u << p.tile_shift[2]) - 1u));
		if (sx < p.size[0] && sy < p.size[1] && sz < p.z_count) {
			const uint64_t out_index = (uint64_t)p.size[0] * p.size[1] * sz + (uint64_t)p.size[0] * sy + sx;
			sample_t<true> v = coherent;
			if constexpr (CW) v = v * (v / incoherent);                      /* coherency_weighting.glsl:36 */
			reinterpret_cast<sample_t<true> *>(p.out)[out_index] = v;
		}
	}
}

template <int FAMILY, bool CW, int WS>
hipError_t launch_tile(const BfDasArgs *a, hipStream_t s)
{
	const uint32_t total = a->blocks[0] * a->blocks[1] * a->blocks[2];
	const uint32_t grid  = a->depth_major == 3u ? bf_plane_walk_blocks(a->blocks[0], a->blocks[1], a->band_rows) : ((total + 7u) / 8u) * 8u;
	const uint32_t A_pad = ((uint32_t)a->acquisition_count + 15u) & ~15u;
	uint32_t lds = 64u + 2u * kTileElems * 32u + 32u + 4u * A_pad + 4u * 2u * kTileCH * 16u + 64u;
	/* the transmit pass borrows the staging area for its [transmit][wave] minima and maxima: they must end before the zero element and the
	 * floors behind it (true for every count the planner admits -- BeamformerMaxEmissionsCount = 256 -- and checked rather than assumed) */
	const uint32_t scratch = 64u + 2u * 4u * 16u * A_pad;
	if (scratch > 64u + 2u * kTileElems * 32u) return hipErrorInvalidValue;
	auto kernel = das_tile_kernel<FAMILY, CW, WS>;
	hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(kernel, dim3(grid), dim3(kTileThreads), lds, s, *a);
	return hipGetLastError();
}

template <int FAMILY, bool CW>
hipError_t launch_tile_window(const BfDasArgs *a, hipStream_t s)
{
	return a->tile_window_shift == 6 ? launch_tile<FAMILY, CW, 6>(a, s) : launch_tile<FAMILY, CW, 5>(a, s);
}

} // namespace

/* IQ samples, cubic interpolation, RCA-family (one receive orientation) and FORCES frames whose tile is 1024 voxels (tile_shift
 * sums to 10) with the lanes of a wave consecutive voxels of one axis; DAS input under 2 GiB (staging offsets) */
extern "C" hipError_t bf_launch_das_tile(const BfDasArgs *a, hipStream_t s)
{
	if (!a->complex_data || a->interpolation != BF_INTERP_CUBIC || a->split_shift) return hipErrorInvalidValue;
	if (a->tile_shift[0] + a->tile_shift[1] + a->tile_shift[2] != 10 || a->sample_count < 8) return hipErrorInvalidValue;
	if ((uint64_t)a->channel_count * (uint64_t)a->acquisition_count * (uint64_t)a->sample_count * 8u >= (1ull << 31)) return hipErrorInvalidValue;
	switch (a->family) {
	case BF_DAS_RCA:    return a->coherency_weighting ? launch_tile_window<BF_DAS_RCA, true>(a, s)    : launch_tile_window<BF_DAS_RCA, false>(a, s);
	case BF_DAS_FORCES: return a->coherency_weighting ? launch_tile_window<BF_DAS_FORCES, true>(a, s) : launch_tile_window<BF_DAS_FORCES, false>(a, s);
	}
	return hipErrorInvalidValue;
}
