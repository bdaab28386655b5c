/* das_separable.hip -- delay-and-sum fast path for row-column geometries on gfx950.
 *
 * Same arithmetic contract as das.hip's RCA family (shaders/das.glsl:204-231 of the
 * reference): out(v) = sum over transmits a and channels c of
 *     apodize(c, v) * rotate_iq( interpolate( rf[c][a], idx(v, a, c) ) ).
 *
 * What the reference (and das.hip) recompute for every (voxel, channel, transmit) triple
 * factors when the receive aperture and the transmit steering lie along different
 * transducer axes (rows vs columns) and the volume's z axis is the depth axis:
 *     idx(v, a, c)  = T(a; v_tx, z) + R(c; v_rx, z)          delay: transmit part + receive part
 *     phasor        = e^{j phi(T)} * e^{j phi(R)}            demodulation phase of each part
 *     apodization   = A(c; v_rx, z)                          receive part only
 * where v_rx / v_tx are the voxel coordinates along the receive / transmit lateral axes.
 * A block therefore owns a tile of U (receive axis) x V (transmit axis) voxels of one
 * z-plane and builds two small LDS tables -- R[c][u] (C*U entries) and T[a][v] (A*V
 * entries) -- costing C*U + A*V square roots and sin/cos pairs instead of U*V*C*A.
 * The inner loop per triple is then: one broadcast LDS read, one add, the index split, one
 * 16-byte gather, the interpolation and a complex multiply-accumulate; the channel's
 * apodization and receive phasor are applied once per channel to the transmit-summed value.
 *
 * MI355X mapping: wave64; 512 or 1024 threads per block share the tables (up to 160 KB of
 * LDS per CU); lanes are laid along the output's x axis so voxel stores are 128-B or 512-B
 * segments; blocks are dealt to XCDs in contiguous runs so that neighbouring tiles (which
 * read neighbouring RF windows) share an L2.  Gather-accumulate, VALU/L1-bound: no MFMA.
 */
#include "das_exact.h"

/* (voxel, channel, transmit) terms whose gathers are in flight together per lane */
#ifndef BF_SEP_BATCH
#define BF_SEP_BATCH 4
#endif

/* demodulation phase of a partial sample index, in turns in [0,1): fract(k * index) with the
 * rounding error of the product recovered by an fma, so that splitting the phase in two
 * parts does not cost precision (Q3 of oracle/oracle.h: the phase is defined range-reduced) */
__device__ __forceinline__ float phase_turns(float k, float index)
{
	float p = k * index;
	float e = __builtin_fmaf(k, index, -p);
	return hw_fract(p) + e;
}

/* LDS tables, 16-byte entries so that every read is one ds_read_b128:
 *   R[(c - c0)*U + u] = { r_index, apod*cos(phi_r), apod*sin(phi_r), apod }   (apod == 0: fails the F# test)
 *   T[a*V + v]        = { cos(phi_t), sin(phi_t), t_index, 0 }   (phasor first: it lands in an
 *                       even-aligned register pair, which packed FMAs need)
 * For real data the phasors are (1, 0).  T covers every transmit; R covers a chunk of
 * q.channel_chunk channels at a time and is rebuilt between chunks, which keeps the block's
 * LDS under 80 KB so that two 1024-thread blocks (8 waves per SIMD) share a CU: the gathers
 * are latency bound at lower occupancy. */
/* VS: log2 of the tile extent along the transmit axis as a compile-time constant (0: read it
 * from q) -- lets the table reads of a batch use immediate LDS offsets. */
template <int INTERP, bool CPLX, bool CW, int VS>
__global__ __launch_bounds__(1024, 8) void das_rca_separable_kernel(const BfDasArgs p, const BfSeparableArgs q)
{
	extern __shared__ __attribute__((aligned(16))) f32x4 sep_lds[];
	const uint32_t v_shift = VS ? (uint32_t)VS : q.v_shift;
	const uint32_t U = 1u << q.u_shift, V = 1u << v_shift;
	const int C = p.channel_count, A = p.acquisition_count, S = p.sample_count;
	const int chunk = (int)q.channel_chunk;
	f32x4 *R = sep_lds;
	f32x4 *T = sep_lds + (size_t)chunk * U;

	/* blockIdx -> tile with each XCD walking a contiguous run of tiles (das.hip) */
	const uint32_t total = q.tiles[0] * q.tiles[1] * q.tiles[2];
	const uint32_t per   = (total + 7u) / 8u;
	const uint32_t tile  = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
	if (tile >= total) return;                               /* whole block: no barrier is skipped */
	/* walk order of the tile list.  Depth-major (default): consecutive tiles -- the ones an XCD has in
	 * flight together -- are a few columns adjacent along u at consecutive depths, whose RF windows overlap
	 * (the window moves ~1.5 samples per plane and ~14 per tile laterally at config 4), so the lines one tile
	 * pulls into the XCD's L2 serve its neighbours (bf_column_walk, bf_kernels.h).  Plane-major: x, then y, then z. */
	uint32_t tu, tv, zl;                                         /* along the receive axis, along the transmit axis, plane inside the shard */
	if (q.depth_major) {
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
	const bool  rx_rows = (p.transmits[0].flags & BF_RX_ROWS) != 0;

	/* ---- transmit table: A x V entries, built once */
	for (uint32_t e = threadIdx.x; e < (uint32_t)A * V; e += blockDim.x) {
		uint32_t a = e >> v_shift, iv = e & (V - 1);
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
		float t_idx = (div_speed_of_sound(dist, p) + p.time_offset) * p.sampling_frequency;
		f32x4 entry = {1.f, 0.f, t_idx, 0.f};
		if constexpr (CPLX) {
			float turns = phase_turns(phase_k, t_idx);
			entry.x = hw_cos_turns(turns);
			entry.y = hw_sin_turns(turns);
		}
		T[e] = entry;
	}

	/* tile-wide extremes of the transmit delay (for the range-test shortcut): every wave
	 * reduces its share of the table, the block combines the per-wave results */
	__shared__ f32x2 wave_range[16];
	__syncthreads();
	{
		float lo = __builtin_inff(), hi = -__builtin_inff();
		for (uint32_t e = threadIdx.x; e < (uint32_t)A * V; e += blockDim.x) {
			float v = T[e].z;
			lo = fminf(lo, v); hi = fmaxf(hi, v);
		}
		for (int off = 32; off > 0; off >>= 1) {
			lo = fminf(lo, __shfl_xor(lo, off, 64));
			hi = fmaxf(hi, __shfl_xor(hi, off, 64));
		}
		if ((threadIdx.x & 63u) == 0) wave_range[threadIdx.x >> 6] = f32x2{lo, hi};
	}
	__syncthreads();
	f32x2 range = wave_range[0];
	for (uint32_t w = 1; w < (blockDim.x >> 6); w++) {
		range.x = fminf(range.x, wave_range[w].x);
		range.y = fmaxf(range.y, wave_range[w].y);
	}
	/* the same for every lane: kept in scalar registers (through scalar temporaries -- das_staged.hip says why) */
	{
		const float lo = range.x, hi = range.y;
		range.x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lo)));
		range.y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hi)));
	}

	/* thread -> voxel: lanes run along the output's x axis */
	/* (the voxel is worked out again where it is needed -- the row-end fix-up, the store at the very end -- rather than held in
	 * two vector registers across the channel loop: the cubic instances have none to spare) */
	auto voxel_of = [&](uint32_t thread, uint32_t &vx, uint32_t &vy, uint32_t &lane_u, uint32_t &lane_v) {
		if (u_axis == 0) { lane_u = thread & (U - 1); lane_v = thread >> q.u_shift; }
		else             { lane_v = thread & (V - 1); lane_u = thread >> v_shift; }
		const uint32_t gu = tu * U + lane_u, gv = tv * V + lane_v;
		vx = u_axis == 0 ? gu : gv; vy = u_axis == 0 ? gv : gu;
	};
	uint32_t lu, lv;
	bool inside;
	{
		uint32_t x0, y0;
		voxel_of(threadIdx.x, x0, y0, lu, lv);
		inside = x0 < p.size[0] && y0 < p.size[1];
	}

	using VT = sample_t<CPLX>;
	VT    coherent   = zero_sample<CPLX>();
	float incoherent = 0.f;

	const char *rf = (const char *)p.rf;
	constexpr uint32_t ES = CPLX ? 8 : 4;
	const float    fS = (float)S;
	const uint32_t row_bytes = (uint32_t)S * ES;
	const f32x4   *Rl = R + lu, *Tl = T + lv;
	const float    edge_margin = p.edge_margin;
	/* the terms of one channel that the loops below left out because their index came within the margin of a row end (the same
	 * sum of the same two table entries, so the same terms): evaluated with the shader's own index, added to the voxel's sums */
	auto edge_fixup = [&](int channel, const f32x4 &r) {
		if constexpr (INTERP != BF_INTERP_NEAREST) {
			uint32_t x, y, unused_u, unused_v, thread = threadIdx.x;
			asm volatile("" : "+v"(thread));                  /* not the values computed before the loop */
			voxel_of(thread, x, y, unused_u, unused_v);
			for (int a = 0; a < A; a++) {
				const float index = r.x + Tl[(size_t)a * V].z;
				if (bfx::edge_near<INTERP>(index, S, edge_margin) && r.w != 0.f)
					bfx::edge_term<BF_DAS_RCA, INTERP, CPLX, CW>(bfx::kernel_args(), x, y, z, channel, a, coherent, incoherent);
			}
		}
	};

	for (int c0 = 0; c0 < C; c0 += chunk) {
		const int cn = (C - c0) < chunk ? (C - c0) : chunk;
		__syncthreads();            /* previous chunk's readers are done (and T is complete) */
		/* ---- receive table for channels [c0, c0 + cn).  The ~50 scalars of the launch arguments it is built from (two 4 x 4 transforms,
		 * pitch, f-number, speed of sound, ...) are read from the kernel-argument segment here, at the top of every chunk, instead of being
		 * held in SGPRs across the channel loop (das_staged.hip: the same trick; at 8 waves per SIMD a wave has 80 SGPRs) */
		{
		const bfx::KernelArgs &ka = bfx::kernel_args();
		const float k_denom_u = fmaxf(1.0f, (float)ka.size[u_axis] - 1.0f);
		const float k_pz = (float)z / fmaxf(1.0f, (float)ka.size[2] - 1.0f);
		const float k_fs = ka.sampling_frequency, k_inv_c = ka.inv_speed_of_sound, k_c = ka.speed_of_sound, k_fnum = ka.f_number;
		const float k_phase = ka.demodulation_frequency * ka.inv_sampling_frequency;
		const float k_pitch = rx_rows ? ka.pitch[1] : ka.pitch[0];
		for (uint32_t e = threadIdx.x; e < (uint32_t)cn * U; e += blockDim.x) {
			uint32_t c = (uint32_t)c0 + (e >> q.u_shift), iu = e & (U - 1);
			float coord[3] = {0.f, 0.f, k_pz};
			coord[u_axis] = (float)(tu * U + iu) / k_denom_u;
			float wx, wy, wz, xx, xy, xz;
			m4_point(ka.voxel_transform, coord[0], coord[1], coord[2], wx, wy, wz);
			m4_point(ka.xdc_transform, wx, wy, wz, xx, xy, xz);
			float lateral = rx_rows ? xy : xx;
			float dx      = lateral - (float)c * k_pitch;
			float a_arg   = __builtin_fabsf(dx * (k_fnum * hw_rcp(__builtin_fabsf(xz))));
			f32x4 entry   = {0.f, 0.f, 0.f, 0.f};
			if (a_arg < 0.5f) {
				float cs    = hw_cos_turns(0.5f * a_arg);
				float apod  = cs * cs;
				float r_idx = div_speed_of_sound(hw_sqrt(dx * dx + xz * xz), k_inv_c, k_c) * k_fs;
				entry.x = r_idx;
				entry.w = apod;
				if constexpr (CPLX) {
					float turns = phase_turns(k_phase, r_idx);
					entry.y = apod * hw_cos_turns(turns);
					entry.z = apod * hw_sin_turns(turns);
				} else {
					entry.y = apod;
				}
			}
			R[e] = entry;
		}
		}
		__syncthreads();
		if (!inside) continue;

		for (int cl = 0; cl < cn; cl++) {
			const f32x4 r = Rl[(size_t)cl * U];
			/* F# culling: skip the channel when no lane of the wave is inside the aperture */
			if (__builtin_amdgcn_ballot_w64(r.w != 0.f) == 0) continue;
			VT    sum = zero_sample<CPLX>();
			float mag = 0.f;
			uint32_t row = (uint32_t)(c0 + cl) * (uint32_t)A * row_bytes;
			if constexpr (INTERP == BF_INTERP_LINEAR && CPLX) {
				/* The headline case, written out so that every step is one instruction:
				 *   idx = R + T; frac = v_fract(idx); k = v_cvt_flr_i32(idx);
				 *   valid = (unsigned)k < S-1; offset = valid ? row + 8k : zero block
				 *   s = (1-frac) s[k] + frac s[k+1]                       2 packed ops
				 *   acc1 += s.re * (cos, sin)(T); acc2 += s.im * (cos, sin)(T)   2 packed FMAs
				 * with sum = (acc1.x - acc2.y, acc1.y + acc2.x) formed once per channel.  Lanes whose
				 * index falls outside the row read 16 zero bytes placed behind the RF by the host. */
				constexpr int B = BF_SEP_BATCH;
				f32x2 acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f};
				const uint32_t ulast = (uint32_t)(S - 1);
				auto term = [&](f32x2 cs, float frac, f32x4 d) {
					f32x2 s0 = {d.x, d.y}, s1 = {d.z, d.w};
					f32x2 sv = s0 + frac * (s1 - s0);
					acc1 += sv.x * cs;
					acc2 += sv.y * cs;
					/* |s|: multiply + fma (5 clk) rather than a packed square + add (6.8 clk, tools/microbench.hip):
					 * 1154 -> 1135 ms per frame on one box */
					if constexpr (CW) mag += hw_sqrt(__builtin_fmaf(sv.y, sv.y, sv.x * sv.x));
				};
				/* When every lane of the wave stays inside the RF row for every transmit of the tile
				 * (r + min T >= 0 and r + max T < S - 1, the tile-wide extremes of T are in `range`),
				 * the per-term range test and the zero-block select are dropped.
				 * "Stays inside" with p.edge_margin to spare at both ends: a term of the unchecked loop is never one whose keep-or-drop
				 * this kernel's index could decide differently from the shader's (das_exact.h).  In the checked loop a term within that
				 * margin of an end reads the zero block -- left out -- and is evaluated exactly, from scratch, after the loop. */
				const bool lane_safe = (r.x + range.x >= edge_margin) && (r.x + range.y < (float)(S - 1) - edge_margin);
				const bool wave_safe = __builtin_amdgcn_ballot_w64(!lane_safe) == 0;
				bool edge_seen = false;
				auto batches = [&](auto checked) {
					constexpr bool CHECK = decltype(checked)::value;
					for (int a = 0; a < A; a += B, row += B * row_bytes) {
						f32x4 t[B]; float frac[B]; uint32_t off[B]; f32x4 d[B];
						#pragma unroll
						for (int k = 0; k < B; k++) t[k] = Tl[(size_t)(a + k < A ? a + k : A - 1) * V];
						#pragma unroll
						for (int k = 0; k < B; k++) {
							float index = r.x + t[k].z;
							frac[k] = hw_fract(index);
							uint32_t ki = (uint32_t)cvt_floor_i32(index);
							off[k] = row + (uint32_t)k * row_bytes + (ki << 3);
							if constexpr (CHECK) {
								const bool edge = bfx::edge_near<INTERP>(index, S, edge_margin) && r.w != 0.f && a + k < A;
								edge_seen |= edge;
								off[k] = (ki < ulast && !edge) ? off[k] : q.zero_offset;
							}
							if (a + k >= A) off[k] = q.zero_offset;      /* wave-uniform: padding of the last batch */
						}
						#pragma unroll
						for (int k = 0; k < B; k++) d[k] = gather<f32x4_a8>(rf, off[k]);
						#pragma unroll
						for (int k = 0; k < B; k++) term(f32x2{t[k].x, t[k].y}, frac[k], d[k]);
					}
				};
				if (wave_safe) batches(std::false_type{});
				else           batches(std::true_type{});
				sum = f32x2{acc1.x - acc2.y, acc1.y + acc2.x};
				if (!wave_safe && __builtin_amdgcn_ballot_w64(edge_seen) != 0) edge_fixup(c0 + cl, r);
			} else {
			/* transmits in batches of B: B broadcast table reads, B index splits, B gathers in
			 * flight, then B interpolate + rotate-accumulate steps (cubic holds two 16-byte loads
			 * and four weights per term: a smaller batch keeps it inside 64 VGPRs) */
			constexpr int B = INTERP == BF_INTERP_CUBIC ? 2 : 4;
			auto term = [&](const f32x4 &t, VT sv) {
				if constexpr (CPLX) {
					sum.x += sv.x * t.x - sv.y * t.y;
					sum.y += sv.x * t.y + sv.y * t.x;
					if constexpr (CW) mag += hw_sqrt(sv.x * sv.x + sv.y * sv.y);
				} else {
					sum += sv;
					if constexpr (CW) mag += __builtin_fabsf(sv);
				}
			};
			/* row ends as above (linear, cubic; nearest interpolation flips at every half-integer, not only there: the parity tests
			 * budget those): a wave with a lane within reach of an end of its RF row drops the terms inside the margin -- index -8 is
			 * outside every mode's range, so tap_setup gives them zero weights -- and evaluates them exactly after the loop */
			bool wave_safe = true, edge_seen = false;
			if constexpr (INTERP != BF_INTERP_NEAREST) {
				const bool lane_safe = (r.x + range.x >= bfx::edge_lo<INTERP>() + edge_margin) && (r.x + range.y < bfx::edge_hi<INTERP>(S) - edge_margin);
				wave_safe = __builtin_amdgcn_ballot_w64(!lane_safe && r.w != 0.f) == 0;
			}
			auto index_of = [&](auto checked, const f32x4 &t) -> float {
				float index = r.x + t.z;
				if constexpr (decltype(checked)::value) {
					const bool edge = bfx::edge_near<INTERP>(index, S, edge_margin) && r.w != 0.f;
					edge_seen |= edge;
					index = edge ? -8.0f : index;
				}
				return index;
			};
			auto batches = [&](auto checked) {
				int a = 0;
				for (; a + B <= A; a += B, row += B * row_bytes) {
					f32x4 t[B];
					Tap<INTERP> tap[B];
					TapData<INTERP, CPLX> d[B];
					#pragma unroll
					for (int k = 0; k < B; k++) t[k] = Tl[(size_t)(a + k) * V];
					#pragma unroll
					for (int k = 0; k < B; k++) tap[k] = tap_setup<INTERP, CPLX>(index_of(checked, t[k]), fS, S - 1);
					#pragma unroll
					for (int k = 0; k < B; k++) d[k] = tap_load<INTERP, CPLX>(rf, row + (uint32_t)k * row_bytes + tap[k].off);
					#pragma unroll
					for (int k = 0; k < B; k++) term(t[k], tap_finish<INTERP, CPLX>(tap[k], d[k]));
				}
				for (; a < A; a++, row += row_bytes) {
					const f32x4 t = Tl[(size_t)a * V];
					term(t, interpolate<INTERP, CPLX>(rf, row, index_of(checked, t), fS, S - 1));
				}
			};
			if (wave_safe) batches(std::false_type{});
			else           batches(std::true_type{});
			if (!wave_safe && __builtin_amdgcn_ballot_w64(edge_seen) != 0) edge_fixup(c0 + cl, r);
			}
			if constexpr (CPLX) {
				coherent.x += sum.x * r.y - sum.y * r.z;
				coherent.y += sum.x * r.z + sum.y * r.y;
			} else {
				coherent += sum * r.y;
			}
			if constexpr (CW) incoherent += r.w * mag;
		}
	}
	if (!inside) return;

	uint32_t x, y, unused_u, unused_v, thread = threadIdx.x;
	asm volatile("" : "+v"(thread));
	voxel_of(thread, x, y, unused_u, unused_v);
	uint64_t out_index = (uint64_t)p.size[0] * p.size[1] * zl + (uint64_t)p.size[0] * y + x;
	if constexpr (CW) coherent = coherent * (coherent / incoherent);   /* coherency_weighting.glsl:36 */
	reinterpret_cast<VT *>(p.out)[out_index] = coherent;
}

template <int INTERP, bool CPLX, bool CW, int VS>
static hipError_t launch_sep(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	uint32_t total = q->tiles[0] * q->tiles[1] * q->tiles[2];
	uint32_t grid  = ((total + 7u) / 8u) * 8u;
	auto kernel = das_rca_separable_kernel<INTERP, CPLX, CW, VS>;
	hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q->lds_bytes);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(kernel, dim3(grid), dim3(q->threads), q->lds_bytes, s, *a, *q);
	return hipGetLastError();
}

template <int INTERP>
static hipError_t launch_sep_kind(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	if (a->complex_data) {
		if constexpr (INTERP == BF_INTERP_LINEAR) {
			/* compile-time tile extents for the headline case */
			if (a->coherency_weighting) switch (q->v_shift) {
			case 4: return launch_sep<INTERP, true, true, 4>(a, q, s);
			case 5: return launch_sep<INTERP, true, true, 5>(a, q, s);
			case 6: return launch_sep<INTERP, true, true, 6>(a, q, s);
			} else switch (q->v_shift) {
			case 4: return launch_sep<INTERP, true, false, 4>(a, q, s);
			case 5: return launch_sep<INTERP, true, false, 5>(a, q, s);
			case 6: return launch_sep<INTERP, true, false, 6>(a, q, s);
			}
		}
		return a->coherency_weighting ? launch_sep<INTERP, true, true, 0>(a, q, s) : launch_sep<INTERP, true, false, 0>(a, q, s);
	}
	return a->coherency_weighting ? launch_sep<INTERP, false, true, 0>(a, q, s) : launch_sep<INTERP, false, false, 0>(a, q, s);
}

extern "C" hipError_t bf_launch_das_separable(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	switch (a->interpolation) {
	case BF_INTERP_NEAREST: return launch_sep_kind<BF_INTERP_NEAREST>(a, q, s);
	case BF_INTERP_LINEAR:  return launch_sep_kind<BF_INTERP_LINEAR>(a, q, s);
	case BF_INTERP_CUBIC:   return launch_sep_kind<BF_INTERP_CUBIC>(a, q, s);
	}
	return hipErrorInvalidValue;
}
