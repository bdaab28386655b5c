/* das_staged.hip -- row-column DAS with the RF staged in LDS (gfx950 / MI355X): the headline kernel.
 *
 * Same arithmetic contract and the same delay factorisation as das_separable.hip
 * (idx = T(a; v_tx, z) + R(c; v_rx, z), shaders/das.glsl:204-231 of the reference), for
 * linear interpolation of complex samples -- the configuration the headline metric runs.
 *
 * das_separable.hip gathers 16 bytes per (voxel, channel, transmit) term straight from global
 * memory and sits on the CU's vector-memory path: 16.3 clk per wave64 gather instruction
 * (tools/microbench.hip).  But the 1024 voxels of a 32 x 32 tile touch only a short window of
 * every RF row: the receive delay moves by at most pitch*fs/c (0.6 sample at config 4) per voxel
 * along the receive axis and less along the transmit axis.  So per channel the block copies,
 * for every transmit, one W-sample window (W = 32 or 64) of the RF row into LDS -- 19 KB of
 * coalesced loads per channel instead of 1.2 MB of gathers through L1 -- and every lane then
 * interpolates out of LDS.
 *
 * Round 1's version of this kernel lost to the gather kernel (1327 against 1165 ms) because it
 * paid ~20 VALU instructions per term against the gather kernel's 14 (integer window bookkeeping,
 * an always-on range test, address arithmetic per tap), and VALU issue is what both kernels wait
 * for.  This version pays 9 (36 per batch of 4 terms):
 *   * the window position is folded into the FLOAT tables: element j of window (c, a) holds sample
 *     floor(rmin_c) + floor(tmin_a) + j (rmin / tmin: the delay minima over the tile); the receive
 *     table hands the lane R' = R - floor(rmin_c), the transmit table holds
 *     T'' = T - floor(tmin_a) - 1/2 -- all exact in f32 -- so ONE add gives the position p in the
 *     window, less 1/2.  (A sum of two small numbers is rounded at 2^-19 of a sample instead of the
 *     absolute index's 2^-12: closer to exact arithmetic than the shader it restates.)
 *   * window elements are LINES in window coordinates, {c_j, d_j} with d_j = s_(j+1) - s_j and
 *     c_j = s_j + (1/2 - j) d_j, 16 bytes: the two taps are ONE aligned ds_read_b128 and the
 *     interpolation ONE packed fma of the position itself, c_j + p d_j -- no fraction is formed;
 *   * no v_fract, no v_cvt: adding M = 2^23 + 2 + a*W rounds p to the nearest integer and leaves the
 *     window ELEMENT INDEX in the low mantissa bits of y = p + M (a packed add over two terms); the
 *     tap's LDS address is (bits(y) & 0xFFF) * 16, one v_lshlrev_b16 by 4 (full rate; a 16-bit op leaves the
 *     upper half of its result zero on gfx950, which drops M's exponent bits), because the staging area
 *     starts two elements into an LDS that holds nothing static and A4 * W <= 4096 elements;
 *   * transmits in pairs: one ds_read2_b64 serves four terms' delays, one ds_read_b128 two terms'
 *     phasors; the transmit table is padded to a multiple of 4 with zero phasors over a zero window
 *     row, so the last batch needs no select;
 *   * the range test of sample_rf (0 <= index < S - 1) is decided per lane and channel when the
 *     receive table is built (tile-wide extremes of T, as in das_separable.hip) and kept in the sign
 *     of the entry's weight; only waves with such a lane run the checked loop (absolute tap = window
 *     tap + the two floors; invalid taps read a zero element).
 *
 * Pipeline per channel: the buffer loads of the NEXT channel's windows are issued into registers
 * before the current channel is consumed and written to LDS after it (barrier - ds_write -
 * barrier); two 1024-thread blocks share a CU, so one block's barriers hide under the other's
 * arithmetic.  The host only launches this kernel when its bound on the delay spread of a tile fits
 * the window (plan_staged, executor.cpp).  No MFMA: gather-accumulate.  Measured (config 4, one
 * MI355X): 785-850 ms per 512^3 frame against 1116-1191 ms for the gather kernel; VALU 94 % busy,
 * 0.83-0.86 of the rate of its own VALU stream run without memory instructions.
 */
#include "das_common.h"

typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef __attribute__((address_space(3))) f32x3 lds_f32x3;

__device__ __forceinline__ float staged_phase_turns(float k, float index)
{
	float p = k * index;
	float e = __builtin_fmaf(k, index, -p);
	return hw_fract(p) + e;
}

/* LDS (A4 = transmits rounded up to a multiple of 4; transmits are kept in PAIRS so that one read serves two terms):
 *   stage[a*W + j]   = { c_j, d_j }: the line through samples j and j + 1 of window (c, a) in window coordinates
 *                      (d_j = s' - s, c_j = s + (1/2 - j) d_j; s = sample floor(rmin_c) + floor(tmin_a) + j of row (c, a),
 *                      s' the next one); j < W, a < A4; two unused elements in front, one zero element behind   f32x4
 *   Tcs[(a/2)*V + v] = { cos(phi_t), sin(phi_t) of transmit a & ~1, then of transmit a | 1 }                   f32x4
 *   R[cl*U + u]      = { R' = r_index - floor(rmin_c), apod*cos(phi_r), apod*sin(phi_r), +-apod }   cl: channel in chunk;
 *                      the weight's sign bit set = the lane may leave the RF row (checked loop)                  f32x4
 *   Tz[(a/2)*V + v]  = { T'' = t_index - floor(tmin_a) - 1/2 of transmit a & ~1, of transmit a | 1 }             f32x2
 *   tfl[a]           = floor(tmin_a)  (checked loop and staging only),  rfloor[cl] = floor(rmin_c)               int */
/* NL: window elements a thread stages per channel, ceil(A4 * W / threads)
 * UNI: the tile is 64 voxels along the receive axis (= x), so a wave's lanes share ONE row of the transmit axis: the transmit
 *      delays and phasors are wave uniform.  They then come from a table in global memory (staged_tables_kernel below writes it
 *      once per frame, the same arithmetic the block otherwise does per tile) through SCALAR loads and enter the packed
 *      instructions as scalar operands: the LDS serves the four taps of a batch and nothing else (tools/microbench.hip
 *      loop_probe_uniform: 40.2 clk per term against 43.7, at a higher sustained clock). */
typedef __attribute__((address_space(4))) const f32x4 const_f32x4;
template <bool CW, int VS, int WS, int NL, bool UNI>
__global__ __launch_bounds__(1024, 8) void das_rca_staged_kernel(const BfDasArgs p, const BfSeparableArgs q)
{
	extern __shared__ __attribute__((aligned(16))) f32x4 staged_lds[];
	/* WS: log2 of the window length (5, 6) */
	constexpr uint32_t V = 1u << VS, W = 1u << WS;
	const uint32_t U = 1u << q.u_shift;
	const int C = p.channel_count, A = p.acquisition_count, S = p.sample_count;
	const int A4 = (A + 3) & ~3;
	const int chunk = (int)q.channel_chunk;
	/* the staging area comes first and the kernel has no static LDS: 16 x (a window element's index + 2) IS its LDS
	 * address, which the inner loop forms with one shift */
	f32x4 *stage  = staged_lds + 2;                          /* (two unused elements in front: see the rounding of the inner loop) */
	f32x4 *Tcs    = stage + (size_t)A4 * W + 1;
	const size_t table_rows = UNI ? 0 : (size_t)(A4 / 2) * V;  /* (UNI: no transmit tables in LDS) */
	f32x4 *R      = Tcs + table_rows;
	f32x2 *Tz     = reinterpret_cast<f32x2 *>(R + (size_t)chunk * U);
	int   *tfl    = reinterpret_cast<int *>(Tz + table_rows);
	int   *rfloor = tfl + A4;
	f32x2 *wave_range = reinterpret_cast<f32x2 *>(rfloor + ((chunk + 1) & ~1));      /* 16 entries, 8-byte aligned */
	const uint32_t stage_elements = (uint32_t)A4 * W;

	/* depth_major bit 2 (UNI): planes in chunks of 32, so that blocks j and j + 32 of an XCD's sequence -- the two a CU holds
	 * (dispatch is breadth first over an XCD's 32 CUs) -- are NEIGHBOURS ALONG u in one plane: they read the same 16 rows of the
	 * global transmit table (14.6 KB at 76 transmits: the scalar cache holds 16 KB) and adjacent RF windows */
	const bool paired = UNI && (q.depth_major & 4u);
	const uint32_t zchunks = (q.tiles[2] + 31u) >> 5;
	const uint32_t total = paired ? q.tiles[0] * q.tiles[1] * zchunks * 32u : q.tiles[0] * q.tiles[1] * q.tiles[2];
	const uint32_t per   = (total + 7u) / 8u;
	const uint32_t tile  = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
	if (tile >= total) return;                               /* whole block */
	uint32_t tu, tv, zl;                                     /* walk order: das_separable.hip */
	if (paired) {
		uint32_t r = tile >> 5;
		tu = r % q.tiles[0]; r /= q.tiles[0];
		zl = (r % zchunks) * 32u + (tile & 31u);
		tv = r / zchunks;
		if (zl >= q.tiles[2]) return;                        /* whole block: the last chunk's padding */
	} else if (q.depth_major & 1u) {
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

	/* UNI: the tile's slice of the global table: [A4] floors, {lo, hi} of the absolute delays, then per lateral row of the tile and
	 * batch of 4 transmits 48 bytes: {T'' x 4}, {cos, sin} x 4 */
	const unsigned char *tile_tab = UNI ? reinterpret_cast<const unsigned char *>(q.tables) + (size_t)(zl * q.tiles[1] + tv) * q.table_stride : nullptr;
	f32x2 range;
	if constexpr (UNI) {
		for (uint32_t a = tid; a < (uint32_t)A4; a += nthreads) tfl[a] = reinterpret_cast<const int *>(tile_tab)[a];
		if (tid == 0) stage[stage_elements] = f32x4{0.f, 0.f, 0.f, 0.f};
		range = *reinterpret_cast<const f32x2 *>(tile_tab + 4u * (uint32_t)A4);
	} else {
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
				float turns = staged_phase_turns(phase_k, t_idx);
				cs_c = hw_cos_turns(turns); cs_s = hw_sin_turns(turns);
			}
			const uint32_t pair = (a >> 1) * V + iv, half = a & 1u;
			reinterpret_cast<f32x2 *>(Tcs + pair)[half] = f32x2{cs_c, cs_s};
			reinterpret_cast<float *>(Tz + pair)[half]  = t_idx;
		}
		if (tid == 0) stage[stage_elements] = f32x4{0.f, 0.f, 0.f, 0.f};
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
		range = wave_range[0];
		for (uint32_t w = 1; w < (nthreads >> 6); w++) {
			range.x = fminf(range.x, wave_range[w].x);
			range.y = fmaxf(range.y, wave_range[w].y);
		}
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
	for (uint32_t a = tid; !UNI && a < (uint32_t)A4; a += nthreads) {
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

	/* the lane's voxel: needed for `inside` here and for the store at the very end -- recomputed there rather than held in two
	 * vector registers across the channel loop (the NL = 4 instances had none to spare) */
	auto voxel_of = [&](uint32_t thread, uint32_t &vx, uint32_t &vy, uint32_t &lane_u) {
		uint32_t lv_;
		if (u_axis == 0) { lane_u = thread & (U - 1); lv_ = thread >> q.u_shift; }
		else             { lv_ = thread & (V - 1); lane_u = thread >> VS; }
		const uint32_t gu = tu * U + lane_u, gv = tv * V + lv_;
		vx = u_axis == 0 ? gu : gv; vy = u_axis == 0 ? gv : gu;
	};
	uint32_t lu;
	bool inside;
	{
		uint32_t x0, y0;
		voxel_of(tid, x0, y0, lu);
		inside = x0 < p.size[0] && y0 < p.size[1];
	}

	f32x2 coherent   = {0.f, 0.f};
	float incoherent = 0.f;
	const f32x4   *Rl = R + lu;
	const uint32_t ulast = (uint32_t)(S - 1);
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
	uint32_t stage_inv[NL];
	{
		const uint32_t windows_per_pass = nthreads >> WS;
		#pragma unroll
		for (int n = 0; n < NL; n++) {
			uint32_t a = (tid >> WS) + (uint32_t)n * windows_per_pass;
			/* transmits of the padding (a >= A) point far outside the buffer: they stage zeros */
			stage_inv[n] = a < (uint32_t)A ? (a * (uint32_t)S + (uint32_t)(tfl[a] + (int)(tid & (W - 1)))) * 8u : 0x80000000u;
		}
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
	/* Element j keeps the LINE through samples j and j + 1 in window coordinates, {c_j, d_j} with d_j = s_(j+1) - s_j and
	 * c_j = s_j + (1/2 - j) d_j, so that the interpolated sample at position p (measured from half a sample into the window,
	 * as the tables hold it) is c_j + p d_j for j = round(p): one packed fma of the position itself, no fraction needed.
	 * (At an integer position both neighbouring lines give the same value, so the tie of the rounding is harmless.)  The next
	 * sample sits in the next lane (a wave stages whole windows, consecutive lanes consecutive samples), fetched with a
	 * one-lane wave shift.  The last element of a window gets a meaningless line and is never selected (the host's window
	 * bound, plan_staged). */
	const float half_minus_j = 0.5f - (float)(tid & (W - 1));
	auto stage_store = [&](const f32x2 (&regs)[NL]) {
		#pragma unroll
		for (int n = 0; n < NL; n++) {
			const float sx = regs[n].x, sy = regs[n].y;
			const float nx = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sx), 0x130, 0xf, 0xf, true));
			const float ny = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sy), 0x130, 0xf, 0xf, true));
			const float dx = nx - sx, dy = ny - sy;
			const uint32_t e = tid + (uint32_t)n * nthreads;
			if (e < stage_elements) stage[e] = f32x4{__builtin_fmaf(half_minus_j, dx, sx), __builtin_fmaf(half_minus_j, dy, sy), dx, dy};
		}
	};

	/* The receive table is rebuilt once per chunk of channels from ~50 scalars of the launch arguments (two 4 x 4 transforms, pitch,
	 * f-number, speed of sound, ...).  Held in SGPRs across the channel loop they cost this kernel 140 scalar spills (v_writelane /
	 * v_readlane into two of its 64 VGPRs, which in turn pushed 5 vector registers to scratch: 3.4 GiB written per 1 GiB frame).  They are
	 * read from the kernel-argument segment instead, through a pointer the compiler cannot see through, at the top of every chunk: a
	 * few s_load per 16 channels, dead again before the channel loop. */
	typedef __attribute__((address_space(4))) const BfDasArgs const_args;
	const_args *kernel_args = (const_args *)__builtin_amdgcn_kernarg_segment_ptr();
	static_assert(__builtin_offsetof(BfDasArgs, xdc_transform) == 0, "BfDasArgs is the kernel's first argument: it sits at offset 0 of the segment");
	for (int c0 = 0; c0 < C; c0 += chunk) {
		const int cn = (C - c0) < chunk ? (C - c0) : chunk;
		__syncthreads();        /* readers of the previous chunk's R / stage are done; the transmit tables are complete */
		{
		const_args *ka = kernel_args;
		asm volatile("" : "+s"(ka));
		const uint32_t k_size[3] = {ka->size[0], ka->size[1], ka->size[2]};
		const float k_denom_u = fmaxf(1.0f, (float)k_size[u_axis] - 1.0f);
		const float k_pz = (float)z / fmaxf(1.0f, (float)k_size[2] - 1.0f);
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
				float turns = staged_phase_turns(k_phase, r_idx);
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
			const bool lane_safe = (entry.x + range.x >= 0.f) && (entry.x + range.y < (float)(S - 1));
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
			/* one term: pos = position in the window (minus 1/2), tap = the line {c, d} of the element round(pos) selects */
			auto term = [&](f32x2 cs, float pos, f32x4 tap) -> float {
				f32x2 sv = f32x2{tap.x, tap.y} + pos * f32x2{tap.z, tap.w};
				acc1 += sv.x * cs;
				acc2 += sv.y * cs;
				if constexpr (CW) return hw_sqrt(__builtin_fmaf(sv.y, sv.y, sv.x * sv.x));
				else return 0.f;
			};
			auto batches = [&](auto checked) {
				constexpr bool CHECK = decltype(checked)::value;
				uint32_t lane_id = tid;
				asm volatile("" : "+v"(lane_id));                          /* not hoisted: see the register budget above */
				const uint32_t lane_v = u_axis == 0 ? lane_id >> q.u_shift : lane_id & (V - 1);
				uint32_t tcs_at = tcs_base + (lane_v << 4), tz_at = tz_base + (lane_v << 3);
				/* Position -> tap without v_fract / v_cvt / a fraction: adding M = 2^23 + 2 + (first window element of the batch)
				 * rounds the position to the nearest integer and leaves the ELEMENT INDEX 2 + a*W + round(p) in the low mantissa
				 * bits of y = p + M (packed: two terms per instruction); the tap's LDS byte address is (bits(y) & 0xFFF) * 16 --
				 * one 16-bit shift (full rate where 32-bit shifts and 24-bit multiplies are half rate; the upper half of a 16-bit
				 * result is zero on gfx950, so M's exponent bits drop out), no add: M's own bit pattern (0x4B000002 + a*W) contributes
				 * exactly 2 + a*W to the mantissa, a batch's base element 2 + a*W + round(p) is below 4096 (plan_staged: A4 * W <= 4096;
				 * rows 1-3 of the batch ride in the read's immediate offset), and the staging area starts two elements into an LDS
				 * that holds nothing static.  M is a scalar, stepped by 4*W
				 * per batch as an integer (the mantissa of a float in [2^23, 2^24) counts integers); term k's row k*W is the read's
				 * immediate offset.  The element is a line in window coordinates, so the interpolation uses p itself.
				 * Why 2^23 + 2: (1) p = -1/2 (the lane with the smallest delays of the tile) must round inside [2^23, 2^24) --
				 * just below 2^23 floats step by 1/2 and 2^23 - 1/2 would come back exact, with garbage in the low mantissa bits;
				 * (2) that tie must not round DOWN to the element in front of the row: 2 + a*W is even, so round-to-even takes it
				 * up to element 0.  At every other tie both neighbouring lines give the same value.
				 * (Tried and measured no faster: y = fma(p, 2^-149, B) into a denormal whose bit pattern is the index, then a
				 * 32-bit shift -- itself half rate, as it turned out -- instead of the 24-bit multiply: 0.721 of the gather kernel's
				 * time against 0.710.  The multiply before the 16-bit shift: 0.701 against 0.700.) */
				uint32_t m_bits = 0x4B000002u;
				[[maybe_unused]] bool window_left = false;    /* range-checked loop: some term selected an element outside its window */
				const f32x2 rr = {r_rel, r_rel};
				/* UNI: the wave's row of the global table (lv = tid >> 6 for a 64-wide tile), read through the constant address
				 * space so that the uniform reads become s_load_dwordx8 + s_load_dwordx4 per batch */
				const_f32x4 *uni_row = nullptr;
				if constexpr (UNI)
					uni_row = (const_f32x4 *)(uintptr_t)(tile_tab + 4u * (uint32_t)A4 + 16u +
					                                      (size_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)) * (size_t)(A4 / 4) * 48u);
				for (int a = 0; a < A4; a += 4, m_bits += 4u * W) {
					uint32_t at[4]; f32x4 tap[4];
					const float M = __builtin_bit_cast(float, m_bits);
					const f32x2 M2 = {M, M};
					f32x4 cs01, cs23; f32x2 tz01, tz23;
					if constexpr (UNI) {
						const f32x4 tz = uni_row[0];
						cs01 = uni_row[1]; cs23 = uni_row[2];
						tz01 = f32x2{tz.x, tz.y}; tz23 = f32x2{tz.z, tz.w};
						uni_row += 3;
					} else {
						cs01 = *(lds_f32x4 *)(uintptr_t)tcs_at;
						cs23 = *(lds_f32x4 *)(uintptr_t)(tcs_at + V * 16u);
						tz01 = *(lds_f32x2 *)(uintptr_t)tz_at;
						tz23 = *(lds_f32x2 *)(uintptr_t)(tz_at + V * 8u);
						tcs_at += 2u * V * 16u; tz_at += 2u * V * 8u;
					}
					const f32x2 p01 = rr + tz01, p23 = rr + tz23;
					const f32x2 y01 = p01 + M2,  y23 = p23 + M2;
					const float ys[4] = {y01.x, y01.y, y23.x, y23.y};
					#pragma unroll
					for (int k = 0; k < 4; k++) {
						const uint32_t yb = __builtin_bit_cast(uint32_t, ys[k]);
						asm("v_lshlrev_b16 %0, 4, %1" : "=v"(at[k]) : "v"(yb));     /* upper half of the result: zero */
						if constexpr (CHECK) {
							uint32_t k_abs = (uint32_t)((int)(yb - m_bits) + rfloor[cl] + tfl[a + k]);      /* yb - m_bits = round(p) */
							at[k] = k_abs < ulast ? at[k] + (uint32_t)k * W * 16u : (stage_elements + 2u) * 16u;
							window_left |= __builtin_amdgcn_ballot_w64((yb - m_bits) > W - 2u) != 0ull;      /* (wave uniform: a scalar) never, unless plan_staged's bound is wrong */
						}
					}
					#pragma unroll
					for (int k = 0; k < 4; k++) tap[k] = *(lds_f32x4 *)(uintptr_t)(at[k] + (CHECK ? 0u : (uint32_t)k * W * 16u));   /* immediate */
					const float q0 = term(f32x2{cs01.x, cs01.y}, p01.x, tap[0]);
					const float q1 = term(f32x2{cs01.z, cs01.w}, p01.y, tap[1]);
					const float q2 = term(f32x2{cs23.x, cs23.y}, p23.x, tap[2]);
					const float q3 = term(f32x2{cs23.z, cs23.w}, p23.y, tap[3]);
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

	uint32_t x, y, lane_u_unused, thread = tid;
	asm volatile("" : "+v"(thread));                      /* not the values computed before the loop */
	voxel_of(thread, x, y, lane_u_unused);
	uint64_t out_index = (uint64_t)p.size[0] * p.size[1] * zl + (uint64_t)p.size[0] * y + x;
	if constexpr (CW) coherent = coherent * (coherent / incoherent);   /* coherency_weighting.glsl:36 */
	reinterpret_cast<f32x2 *>(p.out)[out_index] = coherent;
}

template <bool CW, int VS, int WS, int NL, bool UNI>
static hipError_t launch_staged(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	uint32_t total = q->tiles[0] * q->tiles[1] * q->tiles[2];
	if (UNI && (q->depth_major & 4u)) total = q->tiles[0] * q->tiles[1] * ((q->tiles[2] + 31u) >> 5) * 32u;   /* the paired walk pads the planes to chunks of 32 */
	uint32_t grid  = ((total + 7u) / 8u) * 8u;
	auto kernel = das_rca_staged_kernel<CW, VS, WS, NL, UNI>;
	hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q->lds_bytes);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(kernel, dim3(grid), dim3(q->threads), q->lds_bytes, s, *a, *q);
	return hipGetLastError();
}

template <bool CW, int VS, int WS, bool UNI>
static hipError_t launch_staged_loads(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	const uint32_t A4 = ((uint32_t)a->acquisition_count + 3u) & ~3u;
	/* passes a thread stages per channel: whole windows per wave */
	const uint32_t passes = ((A4 << WS) + q->threads - 1) / q->threads;
	switch (passes) {
	case 1: return launch_staged<CW, VS, WS, 1, UNI>(a, q, s);
	case 2: return launch_staged<CW, VS, WS, 2, UNI>(a, q, s);
	case 3: return launch_staged<CW, VS, WS, 3, UNI>(a, q, s);
	case 4: return launch_staged<CW, VS, WS, 4, UNI>(a, q, s);
	}
	return hipErrorInvalidValue;
}

template <bool CW>
static hipError_t launch_staged_shape(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	if (q->uniform) {
		/* wave-uniform transmit tables: a 64 x 16 tile with x along the receive axis, 1024 threads, tables written by bf_launch_das_staged_tables */
		if (q->u_axis != 0 || q->u_shift != 6 || q->v_shift != 4 || q->threads != 1024 || !q->tables) return hipErrorInvalidValue;
		if (q->window_samples == 32) return launch_staged_loads<CW, 4, 5, true>(a, q, s);
		if (q->window_samples == 64) return launch_staged_loads<CW, 4, 6, true>(a, q, s);
		return hipErrorInvalidValue;
	}
	switch ((q->v_shift << 4) | q->window_shift) {
	case (4 << 4) | 5: return launch_staged_loads<CW, 4, 5, false>(a, q, s);
	case (5 << 4) | 5: return launch_staged_loads<CW, 5, 5, false>(a, q, s);
	case (6 << 4) | 5: return launch_staged_loads<CW, 6, 5, false>(a, q, s);
	case (4 << 4) | 6: return launch_staged_loads<CW, 4, 6, false>(a, q, s);
	case (5 << 4) | 6: return launch_staged_loads<CW, 5, 6, false>(a, q, s);
	case (6 << 4) | 6: return launch_staged_loads<CW, 6, 6, false>(a, q, s);
	}
	return hipErrorInvalidValue;
}

/* complex samples, linear interpolation only; the caller checked q->window_shift */
extern "C" hipError_t bf_launch_das_staged(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	if (!a->complex_data || a->interpolation != BF_INTERP_LINEAR) return hipErrorInvalidValue;
	/* the staging loads address the DAS input through 32-bit buffer offsets with out-of-range padding at 2^31 */
	if ((uint64_t)a->channel_count * (uint64_t)a->acquisition_count * (uint64_t)a->sample_count * 8u >= (1ull << 31)) return hipErrorInvalidValue;
	return a->coherency_weighting ? launch_staged_shape<true>(a, q, s) : launch_staged_shape<false>(a, q, s);
}

/* ---- the transmit tables of the UNI variant, once per frame: one block per (lateral tile row tv, plane zl), the arithmetic of the
 * kernel's own table build (same functions, same order: the entries are bit-identical to what a block would compute in LDS).
 * Layout per tile slice of q.table_stride bytes: int floor(tmin_a)[A4] | {lo, hi} of the absolute delays + 8 bytes of padding |
 * per lateral row iv < 16 and batch b < A4 / 4: {T'' of transmits 4b .. 4b + 3}, {cos, sin} of 4b, 4b + 1, {cos, sin} of 4b + 2, 4b + 3. */
__global__ __launch_bounds__(256) void staged_tables_kernel(const BfDasArgs p, const BfSeparableArgs q)
{
	extern __shared__ __attribute__((aligned(16))) f32x4 tables_lds[];
	constexpr uint32_t VS = 4, V = 1u << VS;
	const int A = p.acquisition_count;
	const int A4 = (A + 3) & ~3;
	float *t  = reinterpret_cast<float *>(tables_lds);                 /* [A4][V] */
	f32x2 *cs = reinterpret_cast<f32x2 *>(t + (size_t)A4 * V);          /* [A4][V] */
	f32x2 *wave_range = cs + (size_t)A4 * V;                            /* [4] */
	const uint32_t tv = blockIdx.x % q.tiles[1], zl = blockIdx.x / q.tiles[1];
	const uint32_t z  = p.z_first + zl;
	const uint32_t v_axis = 1u - q.u_axis;
	const float denom[3] = {fmaxf(1.0f, (float)p.size[0] - 1.0f), fmaxf(1.0f, (float)p.size[1] - 1.0f),
	                        fmaxf(1.0f, (float)p.size[2] - 1.0f)};
	const float pz = (float)z / denom[2];
	const float phase_k = p.demodulation_frequency * p.inv_sampling_frequency;
	const uint32_t tid = threadIdx.x, nthreads = blockDim.x;
	for (uint32_t e = tid; e < (uint32_t)A4 * V; e += nthreads) {
		uint32_t a = e >> VS, iv = e & (V - 1);
		float cs_c = 0.f, cs_s = 0.f, t_idx = 0.f;
		if (a < (uint32_t)A) {
			float coord[3] = {0.f, 0.f, pz};
			coord[v_axis] = (float)(tv * V + iv) / denom[v_axis];
			float wx, wy, wz;
			m4_point(p.voxel_transform, coord[0], coord[1], coord[2], wx, wy, wz);
			const BfTransmit t_a = p.transmits[a];
			float dist = 0.f;
			if (!(t_a.flags & BF_TX_NONE)) {
				float px = (t_a.flags & BF_TX_ROWS) ? wy : wx;
				if (t_a.flags & BF_TX_PLANE) dist = px * t_a.sin_a + wz * t_a.cos_a;
				else { float ddx = px - t_a.focus_x, ddz = wz - t_a.focus_z; dist = hw_sqrt(ddx * ddx + ddz * ddz); }
			}
			t_idx = (div_speed_of_sound(dist, p) + p.time_offset) * p.sampling_frequency;
			float turns = staged_phase_turns(phase_k, t_idx);
			cs_c = hw_cos_turns(turns); cs_s = hw_sin_turns(turns);
		}
		t[e] = t_idx; cs[e] = f32x2{cs_c, cs_s};
	}
	__syncthreads();
	{
		float lo = __builtin_inff(), hi = -__builtin_inff();
		for (uint32_t e = tid; e < (uint32_t)A * V; e += nthreads) { lo = fminf(lo, t[e]); hi = fmaxf(hi, t[e]); }
		for (int off = 32; off > 0; off >>= 1) {
			lo = fminf(lo, __shfl_xor(lo, off, 64));
			hi = fmaxf(hi, __shfl_xor(hi, off, 64));
		}
		if ((tid & 63u) == 0) wave_range[tid >> 6] = f32x2{lo, hi};
	}
	__syncthreads();
	unsigned char *tile_tab = reinterpret_cast<unsigned char *>(q.tables) + (size_t)blockIdx.x * q.table_stride;
	if (tid == 0) {
		f32x2 range = wave_range[0];
		for (uint32_t w = 1; w < (nthreads >> 6); w++) {
			range.x = fminf(range.x, wave_range[w].x);
			range.y = fmaxf(range.y, wave_range[w].y);
		}
		*reinterpret_cast<f32x4 *>(tile_tab + 4u * (uint32_t)A4) = f32x4{range.x, range.y, 0.f, 0.f};
	}
	for (uint32_t a = tid; a < (uint32_t)A4; a += nthreads) {
		float *row = t + (size_t)a * V;
		float  m   = row[0];
		for (uint32_t iv = 1; iv < V; iv++) m = fminf(m, row[iv]);
		float fl = __builtin_floorf(m);
		for (uint32_t iv = 0; iv < V; iv++) row[iv] = (row[iv] - fl) - 0.5f;      /* both steps exact */
		reinterpret_cast<int *>(tile_tab)[a] = (int)fl;
	}
	__syncthreads();
	const uint32_t batches = (uint32_t)A4 / 4u;
	f32x4 *rows = reinterpret_cast<f32x4 *>(tile_tab + 4u * (uint32_t)A4 + 16u);
	for (uint32_t e = tid; e < V * batches; e += nthreads) {
		const uint32_t iv = e / batches, b = e % batches;
		const float *t4 = t + (size_t)(4u * b) * V + iv;
		const f32x2 *c4 = cs + (size_t)(4u * b) * V + iv;
		f32x4 *row = rows + (size_t)e * 3u;
		row[0] = f32x4{t4[0], t4[V], t4[2 * V], t4[3 * V]};
		row[1] = f32x4{c4[0].x, c4[0].y, c4[V].x, c4[V].y};
		row[2] = f32x4{c4[2 * V].x, c4[2 * V].y, c4[3 * V].x, c4[3 * V].y};
	}
}

/* one launch per frame and shard, before bf_launch_das_staged with q->uniform set; q->tables holds q->tiles[1] * q->tiles[2] slices */
extern "C" hipError_t bf_launch_das_staged_tables(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	if (!q->uniform || !q->tables || q->v_shift != 4) return hipErrorInvalidValue;
	const uint32_t A4 = ((uint32_t)a->acquisition_count + 3u) & ~3u;
	if (q->table_stride < 4u * A4 + 16u + 16u * (A4 / 4u) * 48u) return hipErrorInvalidValue;
	const uint32_t lds = A4 * 16u * 12u + 64u;
	hipLaunchKernelGGL(staged_tables_kernel, dim3(q->tiles[1] * q->tiles[2]), dim3(256), lds, s, *a, *q);
	return hipGetLastError();
}
