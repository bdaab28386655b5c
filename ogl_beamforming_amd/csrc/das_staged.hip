/* das_staged.hip -- row-column DAS with the RF staged in LDS (gfx950 / MI355X).
 *
 * Same arithmetic contract and the same delay factorisation as das_separable.hip
 * (idx = T(a; v_tx, z) + R(c; v_rx, z), shaders/das.glsl:204-231 of the reference), for
 * linear interpolation of complex samples -- the configuration the headline metric runs.
 *
 * das_separable.hip gathers 16 bytes per (voxel, channel, transmit) term straight from
 * global memory and saturates the CU's vector L1 (one 64-byte access per four lanes,
 * ~17 accesses per wave-gather, measured 1.04 accesses/clk/CU).  But the 1024 voxels of a
 * 32 x 32 tile touch only a short window of every RF row: the receive delay moves by at most
 * pitch*fs/c (0.6 sample at config 4) per voxel along the receive axis and less along the
 * transmit axis.  So per channel the block copies, for every transmit, one W-sample window
 * (W = 32 or 64, 256 or 512 bytes) of the RF row into LDS -- 19 KB per channel instead of
 * 1.2 MB of gathers through L1 -- and every lane then interpolates out of LDS:
 *   two ds_read_b64 (conflict free: a window is at most one 256-B bank row and the lanes of a
 *   half-wave sit on one voxel row) + the broadcast table read = 8 LDS cycles per 64 terms
 * against 17 L1 cycles before.
 *
 * Pipeline per channel: the global loads of the NEXT channel's windows are issued into
 * registers before the current channel is consumed, and written to LDS after it
 * (barrier - ds_write - barrier); two 1024-thread blocks share a CU so one block's barriers
 * hide under the other's arithmetic.
 *
 * Window bookkeeping is integer only: element e of window a holds sample
 * floor(rmin_c) + floor(tmin_a) + e, where rmin_c / tmin_a are the minima of the receive /
 * transmit delays over the tile (kept with the tables).  Since floor(x + y) >= floor(x) +
 * floor(y), a lane's tap k = floor(R + T) is never left of the window; the host only launches
 * this kernel when its bound on the delay spread fits the window (plan_staged, executor.cpp).
 * Out-of-range sample indices read a zero pair kept behind the windows, so there is no
 * weight arithmetic (das_common.h explains why no branch either).  No MFMA: gather-accumulate.
 */
#include "das_common.h"

#define BF_STAGE_MAX_LOADS 4      /* window elements a thread stages per channel (A*W <= 4096) */
#ifndef BF_STAGED_BATCH
#define BF_STAGED_BATCH 4         /* terms whose LDS reads are in flight together per lane */
#endif

__device__ __forceinline__ float staged_phase_turns(float k, float index)
{
	float p = k * index;
	float e = __builtin_fmaf(k, index, -p);
	return hw_fract(p) + e;
}

/* LDS (16-byte units first):
 *   T[a*V + v]   = { cos(phi_t), sin(phi_t), t_index, bits(a*W - floor(tmin_a)) }
 *   R[cl*U + u]  = { apod*cos(phi_r), apod*sin(phi_r), r_index, apod }       cl: channel in chunk
 *   stage[a*W + e], e < W, then 2 zero elements                              f32x2
 *   tfloor[a], rfloor[cl]                                                     int
 */
template <bool CW, int VS, int WS>
__global__ __launch_bounds__(1024, 8) void das_rca_staged_kernel(const BfDasArgs p, const BfSeparableArgs q)
{
	extern __shared__ __attribute__((aligned(16))) f32x4 staged_lds[];
	constexpr uint32_t V = 1u << VS, W = 1u << WS;
	const uint32_t U = 1u << q.u_shift;
	const int C = p.channel_count, A = p.acquisition_count, S = p.sample_count;
	const int chunk = (int)q.channel_chunk;
	f32x4 *T      = staged_lds;
	f32x4 *R      = T + (size_t)A * V;
	f32x2 *stage  = reinterpret_cast<f32x2 *>(R + (size_t)chunk * U);
	int   *tfloor = reinterpret_cast<int *>(stage + (size_t)A * W + 2);
	int   *rfloor = tfloor + A;
	const uint32_t zero_element = (uint32_t)A * W;

	const uint32_t total = q.tiles[0] * q.tiles[1] * q.tiles[2];
	const uint32_t per   = (total + 7u) / 8u;
	const uint32_t tile  = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
	if (tile >= total) return;                               /* whole block */
	const uint32_t tu = tile % q.tiles[0];
	const uint32_t tv = (tile / q.tiles[0]) % q.tiles[1];
	const uint32_t zl = tile / (q.tiles[0] * q.tiles[1]);
	const uint32_t z  = p.z_first + zl;

	const uint32_t u_axis = q.u_axis, v_axis = 1u - q.u_axis;
	const float denom[3] = {fmaxf(1.0f, (float)p.size[0] - 1.0f), fmaxf(1.0f, (float)p.size[1] - 1.0f),
	                        fmaxf(1.0f, (float)p.size[2] - 1.0f)};
	const float pz = (float)z / denom[2];
	const float phase_k = p.demodulation_frequency * p.inv_sampling_frequency;
	const BfTransmit t0 = p.transmits[0];
	const bool  rx_rows = (t0.flags & BF_RX_ROWS) != 0;
	const float rx_pitch = rx_rows ? p.pitch[1] : p.pitch[0];
	const uint32_t tid = threadIdx.x, nthreads = blockDim.x;

	/* ---- transmit table */
	for (uint32_t e = tid; e < (uint32_t)A * V; e += nthreads) {
		uint32_t a = e >> VS, iv = e & (V - 1);
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
		float t_idx = (dist * p.inv_speed_of_sound + p.time_offset) * p.sampling_frequency;
		float turns = staged_phase_turns(phase_k, t_idx);
		T[e] = f32x4{hw_cos_turns(turns), hw_sin_turns(turns), t_idx, 0.f};
	}
	if (tid < 2) stage[zero_element + tid] = f32x2{0.f, 0.f};
	__syncthreads();
	/* floor of the smallest transmit delay of each window, folded with the window's position */
	for (uint32_t a = tid; a < (uint32_t)A; a += nthreads) {
		float *row = reinterpret_cast<float *>(T + (size_t)a * V);
		float  m   = row[2];
		#pragma unroll 4
		for (uint32_t iv = 1; iv < V; iv++) m = fminf(m, row[4 * iv + 2]);
		int f = (int)__builtin_floorf(m);
		tfloor[a] = f;
		float bits = __builtin_bit_cast(float, (int)(a * W) - f);
		#pragma unroll 4
		for (uint32_t iv = 0; iv < V; iv++) row[4 * iv + 3] = bits;
	}

	uint32_t lu, lv;
	if (u_axis == 0) { lu = tid & (U - 1); lv = tid >> q.u_shift; }
	else             { lv = tid & (V - 1); lu = tid >> VS; }
	const uint32_t gu = tu * U + lu, gv = tv * V + lv;
	const uint32_t x = u_axis == 0 ? gu : gv, y = u_axis == 0 ? gv : gu;
	const bool inside = x < p.size[0] && y < p.size[1];

	f32x2 coherent   = {0.f, 0.f};
	float incoherent = 0.f;
	const char    *rf_bytes = (const char *)p.rf;
	const f32x4   *Rl = R + lu, *Tl = T + lv;
	const uint32_t ulast = (uint32_t)(S - 1);
	const uint32_t stage_elements = (uint32_t)A * W;

	/* the window elements this thread stages: e = tid + n*nthreads -> (a, j) */
	auto stage_load = [&](int channel, int rfl, f32x2 (&regs)[BF_STAGE_MAX_LOADS]) {
		#pragma unroll
		for (int n = 0; n < BF_STAGE_MAX_LOADS; n++) {
			uint32_t e = tid + (uint32_t)n * nthreads;
			regs[n] = f32x2{0.f, 0.f};
			if (e < stage_elements) {
				uint32_t a = e >> WS, j = e & (W - 1);
				int s = rfl + tfloor[a] + (int)j;
				if ((uint32_t)s < (uint32_t)S)
					regs[n] = gather<f32x2>(rf_bytes, (((uint32_t)channel * (uint32_t)A + a) * (uint32_t)S + (uint32_t)s) * 8u);
			}
		}
	};
	auto stage_store = [&](const f32x2 (&regs)[BF_STAGE_MAX_LOADS]) {
		#pragma unroll
		for (int n = 0; n < BF_STAGE_MAX_LOADS; n++) {
			uint32_t e = tid + (uint32_t)n * nthreads;
			if (e < stage_elements) stage[e] = regs[n];
		}
	};

	for (int c0 = 0; c0 < C; c0 += chunk) {
		const int cn = (C - c0) < chunk ? (C - c0) : chunk;
		__syncthreads();        /* readers of the previous chunk's R / stage are done; T is complete */
		for (uint32_t e = tid; e < (uint32_t)cn * U; e += nthreads) {
			uint32_t c = (uint32_t)c0 + (e >> q.u_shift), iu = e & (U - 1);
			float coord[3] = {0.f, 0.f, pz};
			coord[u_axis] = (float)(tu * U + iu) / denom[u_axis];
			float wx, wy, wz, xx, xy, xz;
			m4_point(p.voxel_transform, coord[0], coord[1], coord[2], wx, wy, wz);
			m4_point(p.xdc_transform, wx, wy, wz, xx, xy, xz);
			float lateral = rx_rows ? xy : xx;
			float dx      = lateral - (float)c * rx_pitch;
			float a_arg   = __builtin_fabsf(dx * (p.f_number * hw_rcp(__builtin_fabsf(xz))));
			/* the delay is kept for lanes outside the aperture too: it keeps their (discarded)
			 * LDS reads inside the window */
			float r_idx = hw_sqrt(dx * dx + xz * xz) * p.inv_speed_of_sound * p.sampling_frequency;
			f32x4 entry = {0.f, 0.f, r_idx, 0.f};
			if (a_arg < 0.5f) {
				float cs    = hw_cos_turns(0.5f * a_arg);
				float apod  = cs * cs;
				float turns = staged_phase_turns(phase_k, r_idx);
				entry.x = apod * hw_cos_turns(turns);
				entry.y = apod * hw_sin_turns(turns);
				entry.w = apod;
			}
			R[e] = entry;
		}
		__syncthreads();
		for (uint32_t cl = tid; cl < (uint32_t)cn; cl += nthreads) {
			const float *row = reinterpret_cast<const float *>(R + (size_t)cl * U);
			float m = row[2];
			#pragma unroll 4
			for (uint32_t iu = 1; iu < U; iu++) m = fminf(m, row[4 * iu + 2]);
			rfloor[cl] = (int)__builtin_floorf(m);
		}
		__syncthreads();

		f32x2 regs[BF_STAGE_MAX_LOADS];
		stage_load(c0, rfloor[0], regs);
		for (int cl = 0; cl < cn; cl++) {
			__syncthreads();                   /* everyone is done with the previous channel's windows */
			stage_store(regs);
			__syncthreads();
			if (cl + 1 < cn) stage_load(c0 + cl + 1, rfloor[cl + 1], regs);   /* in flight during the arithmetic */
			if (!inside) continue;

			const f32x4 r = Rl[(size_t)cl * U];
			if (__builtin_amdgcn_ballot_w64(r.w != 0.f) == 0) continue;    /* F# culling per wave */
			const int rbase = -rfloor[cl];
			f32x2 acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f};
			float mag = 0.f;
			constexpr int B = BF_STAGED_BATCH;
			for (int a = 0; a < A; a += B) {
				f32x4 t[B]; float frac[B]; uint32_t el[B]; f32x2 s0[B], s1[B];
				#pragma unroll
				for (int k = 0; k < B; k++) t[k] = Tl[(size_t)(a + k < A ? a + k : A - 1) * V];
				#pragma unroll
				for (int k = 0; k < B; k++) {
					float index = r.z + t[k].z;
					frac[k] = hw_fract(index);
					uint32_t ki = (uint32_t)cvt_floor_i32(index);
					/* (through a scalar temporary: __builtin_bit_cast applied directly to a vector
					 * component reads the vector's first component with this hipcc) */
					const float wbits = t[k].w;
					uint32_t e  = ki + (uint32_t)(__builtin_bit_cast(int, wbits) + rbase);
					el[k] = (ki < ulast && a + k < A) ? e : zero_element;
				}
				#pragma unroll
				for (int k = 0; k < B; k++) {
					/* two independent 8-byte reads (2 LDS cycles each); the empty asm keeps hipcc from
					 * fusing them into one ds_read2_b64 (8 cycles) */
					uint32_t e1 = el[k] + 1;
					asm("" : "+v"(e1));
					s0[k] = stage[el[k]];
					s1[k] = stage[e1];
				}
				#pragma unroll
				for (int k = 0; k < B; k++) {
					f32x2 sv = (1.f - frac[k]) * s0[k] + frac[k] * s1[k];
					f32x2 cs = {t[k].x, t[k].y};
					acc1 += sv.x * cs;
					acc2 += sv.y * cs;
					if constexpr (CW) { f32x2 sq = sv * sv; mag += hw_sqrt(sq.x + sq.y); }
				}
			}
			f32x2 sum = {acc1.x - acc2.y, acc1.y + acc2.x};
			coherent.x += sum.x * r.x - sum.y * r.y;
			coherent.y += sum.x * r.y + sum.y * r.x;
			if constexpr (CW) incoherent += r.w * mag;
		}
	}
	if (!inside) return;

	uint64_t out_index = (uint64_t)p.size[0] * p.size[1] * zl + (uint64_t)p.size[0] * y + x;
	if constexpr (CW) coherent = coherent * (coherent / incoherent);   /* coherency_weighting.glsl:36 */
	reinterpret_cast<f32x2 *>(p.out)[out_index] = coherent;
}

template <bool CW, int VS, int WS>
static hipError_t launch_staged(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	uint32_t total = q->tiles[0] * q->tiles[1] * q->tiles[2];
	uint32_t grid  = ((total + 7u) / 8u) * 8u;
	auto kernel = das_rca_staged_kernel<CW, VS, WS>;
	hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q->lds_bytes);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(kernel, dim3(grid), dim3(q->threads), q->lds_bytes, s, *a, *q);
	return hipGetLastError();
}

template <bool CW>
static hipError_t launch_staged_shape(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	switch ((q->v_shift << 4) | q->window_shift) {
	case (4 << 4) | 5: return launch_staged<CW, 4, 5>(a, q, s);
	case (5 << 4) | 5: return launch_staged<CW, 5, 5>(a, q, s);
	case (6 << 4) | 5: return launch_staged<CW, 6, 5>(a, q, s);
	case (4 << 4) | 6: return launch_staged<CW, 4, 6>(a, q, s);
	case (5 << 4) | 6: return launch_staged<CW, 5, 6>(a, q, s);
	case (6 << 4) | 6: return launch_staged<CW, 6, 6>(a, q, s);
	}
	return hipErrorInvalidValue;
}

/* complex samples, linear interpolation only; the caller checked q->window_shift */
extern "C" hipError_t bf_launch_das_staged(const BfDasArgs *a, const BfSeparableArgs *q, hipStream_t s)
{
	if (!a->complex_data || a->interpolation != BF_INTERP_LINEAR) return hipErrorInvalidValue;
	return a->coherency_weighting ? launch_staged_shape<true>(a, q, s) : launch_staged_shape<false>(a, q, s);
}
