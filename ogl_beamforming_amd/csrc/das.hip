/* das.hip -- delay-and-sum for gfx950 (MI355X), general path.
 *
 * Replaces shaders/das.glsl of the reference (main :368-407, RCA :204-231, HERCULES
 * :233-286, FORCES :288-321, READI_FORCES :323-366, sample_rf :99-124, cubic :67-97,
 * rotate_iq :54-61, apodize :138-152) and the DAS leg of do_compute_shader
 * (beamformer_core.c:1353-1369).
 *
 * Differences in structure (results agree within the float tolerance stated in
 * tests/test_gpu_parity.py):
 *   - ONE launch covers every receive channel; the reference launches the whole volume
 *     once per 16-channel chunk and read-modify-writes the frame channel_count/16 times
 *     (beamformer_core.c:1604-1614).  Here each voxel is written exactly once, so the
 *     frame and incoherent clears (beamformer_core.c:1573-1585) disappear too.
 *   - coherency weighting (coherency_weighting.glsl:28-37) is the epilogue of this kernel;
 *     the incoherent sum never reaches HBM.
 *   - per-transmit trigonometry is hoisted to a host-prepared BfTransmit table that the
 *     wave reads through scalar loads.
 *   - a 256-thread block covers a tile that is one voxel thick along the axis with the
 *     steepest delay gradient (chosen by the host), so that the 64 lanes of a wave gather
 *     neighbouring RF samples of one (channel, transmit) row: a few 128-B lines per
 *     wave-instruction instead of 64.
 *   - wave64 throughout; transcendental work uses the hardware's turn-based v_sin/v_cos
 *     with the argument reduced by v_fract (Q3 in oracle/oracle.h).
 * The gather-accumulate is memory/VALU bound: no MFMA.
 */
#include "das_exact.h"

template <bool CPLX, bool CW, bool COUNT>
struct Accumulator {
	sample_t<CPLX> coherent;
	float          incoherent;
	unsigned long long pairs;
	__device__ __forceinline__ void init() { coherent = zero_sample<CPLX>(); incoherent = 0.f; pairs = 0; }
	/* RESULT_STORE (das.glsl:28-32) */
	__device__ __forceinline__ void add(sample_t<CPLX> v)
	{
		coherent += v;
		if constexpr (CW) {
			if constexpr (CPLX) incoherent += hw_sqrt(v.x * v.x + v.y * v.y);
			else                incoherent += __builtin_fabsf(v);
		}
	}
};

/* das.glsl:187-202 with the per-transmit constants precomputed */
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

/* das.glsl:126-130 */
__device__ __forceinline__ float sample_index(float distance, const BfDasArgs &p)
{
	return (div_speed_of_sound(distance, p) + p.time_offset) * p.sampling_frequency;
}

/* A term at an end of its RF row (das_exact.h): this kernel's index -- hardware square root, fused multiply-adds -- may differ
 * from the shader's by an ulp, and sample_rf's range test is a step.  Within p.edge_margin of either end the index is therefore
 * formed again, exactly as the shader's text forms it; everything else about the term stays as it is.  (Nearest interpolation:
 * the index decides the tap at every half-integer, not only at the row ends -- the parity tests budget those flips per voxel.) */
template <int FAMILY, int INTERP>
__device__ __forceinline__ float settle_index(float index, const BfDasArgs &p, uint32_t x, uint32_t y, uint32_t z, int channel, int transmit)
{
	if constexpr (INTERP != BF_INTERP_NEAREST) {
		if (bfx::edge_near<INTERP>(index, p.sample_count, p.edge_margin))
			index = bfx::exact_index<FAMILY>(p, bfx::exact_voxel<FAMILY>(p, x, y, z), channel, transmit);
	}
	return index;
}

/* das.glsl:204-231 */
template <int INTERP, bool CPLX, bool CW, bool COUNT>
__device__ __forceinline__ void das_rca(const BfDasArgs &p, const char *rf, float wx, float wy, float wz, uint32_t x, uint32_t y, uint32_t z,
                                        int ch0, int ch1, Accumulator<CPLX, CW, COUNT> &acc)
{
	float xx, xy, xz;
	m4_point(p.xdc_transform, wx, wy, wz, xx, xy, xz);
	const int S = p.sample_count, A = p.acquisition_count;
	const float inv_abs_z = hw_rcp(__builtin_fabsf(xz));
	const float zz = xz * xz;

	for (int acquisition = 0; acquisition < A; acquisition++) {
		const BfTransmit t = p.transmits[acquisition];
		const bool  rx_rows = (t.flags & BF_RX_ROWS) != 0;
		const float lateral = rx_rows ? xy : xx;
		const float pitch   = rx_rows ? p.pitch[1] : p.pitch[0];
		const float tx_dist = transmit_distance(t, wx, wy, wz);
		const float f_over_z = p.f_number * inv_abs_z;

		int rf_offset = acquisition * S + ch0 * S * A;
		for (int channel = ch0; channel < ch1; channel++) {
			float dx    = lateral - (float)channel * pitch;
			float a_arg = __builtin_fabsf(dx * f_over_z);
			bool  pass  = a_arg < 0.5f;
			if constexpr (COUNT) {
				acc.pairs += pass;
			} else if (pass) {
				float sidx = sample_index(tx_dist + hw_sqrt(dx * dx + zz), p);
				sidx = settle_index<BF_DAS_RCA, INTERP>(sidx, p, x, y, z, channel, acquisition);
				acc.add(apodize(a_arg) * sample_rf<INTERP, CPLX>(rf, rf_offset, sidx, p));
			}
			rf_offset += S * A;
		}
	}
}

/* das.glsl:233-286 */
template <int INTERP, bool CPLX, bool CW, bool COUNT>
__device__ __forceinline__ void das_hercules(const BfDasArgs &p, const char *rf, float wx, float wy, float wz, uint32_t x, uint32_t y, uint32_t z,
                                             int ch0, int ch1, Accumulator<CPLX, CW, COUNT> &acc)
{
	float xx, xy, xz;
	m4_point(p.xdc_transform, wx, wy, wz, xx, xy, xz);
	const int S = p.sample_count, A = p.acquisition_count;
	const int sparse = p.sparse != 0;
	const BfTransmit t0 = p.transmits[0];
	const bool  rx_cols = (t0.flags & BF_RX_COLUMNS) != 0;

	const float transmit_index   = sample_index(transmit_distance(t0, wx, wy, wz), p);
	const float z_delta_squared  = xz * xz;
	const float f_number_over_z  = __builtin_fabsf(p.f_number * hw_rcp(xz));
	const float apodization_test = 0.25f / (f_number_over_z * f_number_over_z);
	/* the axis the decoded transmit elements run along, chosen once */
	const float tx_lateral       = rx_cols ? xy : xx;
	const float tx_pitch         = rx_cols ? p.pitch[1] : p.pitch[0];

	for (int channel = ch0; channel < ch1; channel++) {
		int rf_offset = channel * S * A + sparse * S;
		/* squared lateral distance to the receive element along the receive axis */
		float rx_delta = rx_cols ? xx - (float)channel * p.pitch[0] : xy - (float)channel * p.pitch[1];
		float rx_sq    = rx_delta * rx_delta;

		for (int transmit = sparse; transmit < A; transmit++) {
			float tx_channel = sparse ? (float)p.sparse_elements[transmit - sparse] : (float)transmit;
			float tx_delta   = tx_lateral - tx_channel * tx_pitch;
			float element_delta_squared = tx_delta * tx_delta + rx_sq;
			bool pass = element_delta_squared < apodization_test;
			if constexpr (COUNT) {
				acc.pairs += pass;
			} else if (pass) {
				/* "tribal knowledge" weight of the first transmit (das.glsl:272-273): a scalar */
				const float weight = transmit == 0 ? p.first_transmit_weight : 1.0f;
				float apodization = weight * apodize(f_number_over_z * hw_sqrt(element_delta_squared));
				float index = transmit_index + div_speed_of_sound(hw_sqrt(z_delta_squared + element_delta_squared) * p.sampling_frequency, p);   /* das.glsl:277 */
				index = settle_index<BF_DAS_HERCULES, INTERP>(index, p, x, y, z, channel, transmit);
				acc.add(apodization * sample_rf<INTERP, CPLX>(rf, rf_offset, index, p));
			}
			rf_offset += S;
		}
	}
}

/* das.glsl:288-321 and :323-366.  (wx, wy, wz) is already in transducer space: the host
 * pre-multiplies the voxel transform (beamformer_core.c:913-915). */
template <int INTERP, bool CPLX, bool CW, bool COUNT, bool READI>
__device__ __forceinline__ void das_forces(const BfDasArgs &p, const char *rf, float xx, float xy, float xz, uint32_t x, uint32_t y, uint32_t z,
                                           int ch0, int ch1, Accumulator<CPLX, CW, COUNT> &acc)
{
	const int S = p.sample_count, A = p.acquisition_count, C = p.channel_count;
	const int sparse = p.sparse != 0;
	const float z_delta_squared     = xz * xz;
	const float transmit_y_delta    = xy - p.pitch[1] * (float)C * 0.5f;
	const float transmit_yz_squared = transmit_y_delta * transmit_y_delta + z_delta_squared;
	const float f_over_z            = p.f_number * hw_rcp(xz);
	const int   hadamard_offset     = (int)p.readi_group * (int)p.readi_group_count;

	for (int channel = ch0; channel < ch1; channel++) {
		float receive_x_delta = xx - (float)channel * p.pitch[0];
		float a_arg           = __builtin_fabsf(receive_x_delta * f_over_z);
		const bool pass       = a_arg < 0.5f;
		if (!pass) continue;

		float receive_index = sample_index(hw_sqrt(receive_x_delta * receive_x_delta + z_delta_squared), p);
		float apodization   = COUNT ? 0.f : apodize(a_arg);

		if constexpr (!READI) {
			int rf_offset = channel * S * A + sparse * S;
			for (int transmit = sparse; transmit < A; transmit++) {
				if constexpr (COUNT) {
					acc.pairs += pass;
				} else {
					float tx_channel       = sparse ? (float)p.sparse_elements[transmit - sparse] : (float)transmit;
					float transmit_x_delta = xx - p.pitch[0] * tx_channel;
					float transmit_index   = div_speed_of_sound(hw_sqrt(transmit_yz_squared + transmit_x_delta * transmit_x_delta) * p.sampling_frequency, p);
					const float index = settle_index<BF_DAS_FORCES, INTERP>(receive_index + transmit_index, p, x, y, z, channel, transmit);
					acc.add(apodization * sample_rf<INTERP, CPLX>(rf, rf_offset, index, p));
				}
				rf_offset += S;
			}
		} else {
			int channel_rf_offset = channel * S * A;
			for (int tx_group = 0; tx_group < (int)p.readi_group_count; tx_group++) {
				_Float16 h = __builtin_bit_cast(_Float16, p.readi_hadamard[hadamard_offset + tx_group]);
				float group_apodization = apodization * (float)h;
				int   rf_offset = channel_rf_offset;
				for (int tx_event = 0; tx_event < A; tx_event++) {
					if constexpr (COUNT) {
						acc.pairs += pass;
					} else {
						float tx_element       = (float)tx_group * (float)A + (float)tx_event;
						float transmit_x_delta = xx - p.pitch[0] * tx_element;
						float transmit_index   = div_speed_of_sound(hw_sqrt(transmit_yz_squared + transmit_x_delta * transmit_x_delta) * p.sampling_frequency, p);
						const float index = settle_index<BF_DAS_READI, INTERP>(receive_index + transmit_index, p, x, y, z, channel, tx_group * A + tx_event);
						acc.add(group_apodization * sample_rf<INTERP, CPLX>(rf, rf_offset, index, p));
					}
					rf_offset += S;
				}
			}
		}
	}
}

/* das.glsl:368-407.  Grid: p.blocks[0]*p.blocks[1]*p.blocks[2] blocks; a block is a
 * (1<<tile_shift[0]) x (1<<tile_shift[1]) x (1<<tile_shift[2]) voxel tile: 256 voxels and
 * 256 threads, or -- for frames too small to fill the chip -- 64 voxels and K = 1<<split_shift
 * waves that each sum a contiguous range of C/K channels and combine through LDS. */
#ifndef BF_DAS_MAX_THREADS
#define BF_DAS_MAX_THREADS 1024
#endif
template <int FAMILY, int INTERP, bool CPLX, bool CW, bool COUNT>
__global__ __launch_bounds__(BF_DAS_MAX_THREADS) void das_kernel(const BfDasArgs p)
{
	/* blockIdx -> tile: consecutive block ids go round-robin over the 8 XCDs, so ids that
	 * share (id % 8) share an L2.  Deal the tile list out so that each XCD walks a
	 * contiguous run of tiles (neighbouring tiles read neighbouring RF windows). */
	uint32_t total  = p.blocks[0] * p.blocks[1] * p.blocks[2];
	uint32_t bid    = blockIdx.x;
	uint32_t per    = (total + 7u) / 8u;
	uint32_t tile   = (bid & 7u) * per + (bid >> 3);
	if (p.depth_major != 3u && tile >= total) {
		/* ragged tail: ids whose run is shorter map onto the unassigned remainder */
		return;
	}
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
	uint32_t split = tid >> (p.tile_shift[0] + p.tile_shift[1] + p.tile_shift[2]);   /* wave-uniform when K > 1 */
	uint32_t x = (bx << p.tile_shift[0]) + lx;
	uint32_t y = (by << p.tile_shift[1]) + ly;
	uint32_t zl = (bz << p.tile_shift[2]) + lz;       /* z inside the shard */
	bool inside = x < p.size[0] && y < p.size[1] && zl < p.z_count;

	Accumulator<CPLX, CW, COUNT> acc;
	acc.init();
	if (inside) {
		uint32_t z = p.z_first + zl;
		/* das.glsl:374-376 */
		float px = (float)x / fmaxf(1.0f, (float)p.size[0] - 1.0f);
		float py = (float)y / fmaxf(1.0f, (float)p.size[1] - 1.0f);
		float pz = (float)z / fmaxf(1.0f, (float)p.size[2] - 1.0f);
		float wx, wy, wz;
		m4_point(p.voxel_transform, px, py, pz, wx, wy, wz);
		const char *rf = (const char *)p.rf;
		const int C   = p.channel_count;
		const int per = (C + (1 << p.split_shift) - 1) >> p.split_shift;
		const int ch0 = (int)split * per;
		const int ch1 = ch0 + per < C ? ch0 + per : C;

		if constexpr (FAMILY == BF_DAS_RCA)           das_rca<INTERP, CPLX, CW, COUNT>(p, rf, wx, wy, wz, x, y, z, ch0, ch1, acc);
		else if constexpr (FAMILY == BF_DAS_HERCULES) das_hercules<INTERP, CPLX, CW, COUNT>(p, rf, wx, wy, wz, x, y, z, ch0, ch1, acc);
		else if constexpr (FAMILY == BF_DAS_FORCES)   das_forces<INTERP, CPLX, CW, COUNT, false>(p, rf, wx, wy, wz, x, y, z, ch0, ch1, acc);
		else                                          das_forces<INTERP, CPLX, CW, COUNT, true>(p, rf, wx, wy, wz, x, y, z, ch0, ch1, acc);
	}

	if constexpr (!COUNT) {
		if (p.split_shift) {
			/* partial sums of waves 1..K-1 go through LDS; wave 0 adds them in split order */
			extern __shared__ float partial[];                 /* [K-1][3][64] */
			const uint32_t lane = tid & 63u;
			if (split) {
				float *row = partial + (split - 1) * 192 + lane;
				if constexpr (CPLX) { row[0] = acc.coherent.x; row[64] = acc.coherent.y; }
				else                { row[0] = acc.coherent; }
				if constexpr (CW) row[128] = acc.incoherent;
			}
			__syncthreads();
			if (split) return;
			for (uint32_t k = 1; k < (1u << p.split_shift); k++) {
				const float *row = partial + (k - 1) * 192 + lane;
				if constexpr (CPLX) { acc.coherent.x += row[0]; acc.coherent.y += row[64]; }
				else                { acc.coherent += row[0]; }
				if constexpr (CW) acc.incoherent += row[128];
			}
		}
	}

	if constexpr (COUNT) {
		unsigned long long n = acc.pairs;
		for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
		if ((tid & 63u) == 0 && n) atomicAdd(p.pair_counter, n);
	} else if (inside) {
		uint64_t out_index = (uint64_t)p.size[0] * p.size[1] * zl + (uint64_t)p.size[0] * y + x;
		sample_t<CPLX> v = acc.coherent;
		/* coherency_weighting.glsl:36 with Scale = 1 (beamformer_core.c:949):
		 * c *= c / incoherent, component-wise; incoherent == 0 gives NaN as in the reference */
		if constexpr (CW) v = v * (v / acc.incoherent);
		reinterpret_cast<sample_t<CPLX> *>(p.out)[out_index] = v;
	}
}

template <int FAMILY, int INTERP, bool CPLX, bool CW, bool COUNT>
static hipError_t launch_one(const BfDasArgs *a, hipStream_t s)
{
	uint32_t total = a->blocks[0] * a->blocks[1] * a->blocks[2];
	uint32_t grid  = a->depth_major == 3u ? bf_plane_walk_blocks(a->blocks[0], a->blocks[1], a->band_rows) : ((total + 7u) / 8u) * 8u;
	uint32_t threads = a->split_shift ? 64u << a->split_shift : 256u;
	uint32_t lds     = a->split_shift ? ((1u << a->split_shift) - 1u) * 192u * (uint32_t)sizeof(float) : 0u;
	hipLaunchKernelGGL((das_kernel<FAMILY, INTERP, CPLX, CW, COUNT>), dim3(grid), dim3(threads), lds, s, *a);
	return hipGetLastError();
}

template <int FAMILY, int INTERP, bool COUNT>
static hipError_t launch_kind(const BfDasArgs *a, hipStream_t s)
{
	if constexpr (COUNT) return launch_one<FAMILY, INTERP, false, false, true>(a, s);
	else {
		if (a->complex_data) return a->coherency_weighting ? launch_one<FAMILY, INTERP, true,  true,  false>(a, s)
		                                                   : launch_one<FAMILY, INTERP, true,  false, false>(a, s);
		else                 return a->coherency_weighting ? launch_one<FAMILY, INTERP, false, true,  false>(a, s)
		                                                   : launch_one<FAMILY, INTERP, false, false, false>(a, s);
	}
}

template <int FAMILY, bool COUNT>
static hipError_t launch_interp(const BfDasArgs *a, hipStream_t s)
{
	if constexpr (COUNT) return launch_kind<FAMILY, BF_INTERP_NEAREST, true>(a, s);
	else switch (a->interpolation) {
	case BF_INTERP_NEAREST: return launch_kind<FAMILY, BF_INTERP_NEAREST, false>(a, s);
	case BF_INTERP_LINEAR:  return launch_kind<FAMILY, BF_INTERP_LINEAR,  false>(a, s);
	case BF_INTERP_CUBIC:   return launch_kind<FAMILY, BF_INTERP_CUBIC,   false>(a, s);
	}
	return hipErrorInvalidValue;
}

template <bool COUNT>
static hipError_t launch_family(const BfDasArgs *a, hipStream_t s)
{
	switch (a->family) {
	case BF_DAS_RCA:      return launch_interp<BF_DAS_RCA,      COUNT>(a, s);
	case BF_DAS_HERCULES: return launch_interp<BF_DAS_HERCULES, COUNT>(a, s);
	case BF_DAS_FORCES:   return launch_interp<BF_DAS_FORCES,   COUNT>(a, s);
	case BF_DAS_READI:    return launch_interp<BF_DAS_READI,    COUNT>(a, s);
	}
	return hipErrorInvalidValue;
}

extern "C" hipError_t bf_launch_das(const BfDasArgs *a, hipStream_t s)       { return launch_family<false>(a, s); }
extern "C" hipError_t bf_launch_das_count(const BfDasArgs *a, hipStream_t s) { return launch_family<true>(a, s); }
