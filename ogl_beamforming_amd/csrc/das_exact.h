/* das_exact.h -- the shader's OWN sample index, operation for operation, for the terms that sit at an end of an RF row.
 *
 * sample_rf (shaders/das.glsl:99-124) keeps a term while lo <= index < hi (linear: [0, S - 1), cubic: [1, S - 2)) and drops it
 * otherwise: a STEP.  Every kernel of this library forms the index its own way -- the fast ones as a rounded receive term plus a
 * rounded transmit term out of tables, the general one with the hardware's 1-ulp square root and fused multiply-adds -- and an index
 * that differs from the shader's by an ulp keeps or drops a term that lies within that ulp of `hi` or `lo`: the voxel then moves by
 * one whole tap (round 3's out-of-sample fuzz: 33 of 1050 random acquisitions off by up to 4e-2 of the frame maximum).
 *
 * So: a term whose fast index comes within BfDasArgs::edge_margin (a few dozen ulps of the largest index, das_select.cpp) of `lo`
 * or `hi` is NOT decided by that index.  It is evaluated here, from the voxel's integer coordinates, with the index formed exactly
 * as the shader's text forms it -- main :368-380 (voxel -> world -> transducer), :187-202 (transmit distance), :126-130
 * (sample_index), RCA :204-231, HERCULES :233-286, FORCES :288-321, READI :323-366 -- in IEEE single arithmetic: every product and sum
 * rounded on its own (no fused multiply-add: `#pragma clang fp contract(off)` in every function below), division and square root
 * correctly rounded (hipcc's default for `/` and __builtin_sqrtf; the kernels' hw_rcp / hw_sqrt are the 1-ulp instructions).  That is
 * the arithmetic of the CPU oracle's float build (oracle/oracle_das_body.h, gcc -ffp-contract=off), so the terms kept and dropped at
 * the row ends are the oracle's, bit for bit, in EVERY kernel -- and the unchecked loops, whose waves provably stay a margin away
 * from both ends, are untouched.
 *
 * Cost: only waves with a lane within reach of a row end run a checked loop at all (decided per wave from table extremes, as
 * before); there the test is a few VALU instructions per term, and the ~150-instruction evaluation below runs for the few terms
 * inside the margin.
 */
#ifndef BF_DAS_EXACT_H
#define BF_DAS_EXACT_H

#include "das_common.h"

namespace bfx {

/* valid range of sample_rf as floats: lo <= index < hi */
template <int INTERP> __device__ __forceinline__ float edge_lo()        { return INTERP == BF_INTERP_CUBIC ? 1.0f : 0.0f; }
template <int INTERP> __device__ __forceinline__ float edge_hi(int S)   { return (float)(S - (INTERP == BF_INTERP_CUBIC ? 2 : 1)); }

/* within `margin` of an end of the valid range (false for a NaN).  One expression for both ends -- the distance from the middle of the
 * range, less half its length, is within the margin of zero: two subtractions and a compare (the loops that carry this test per term are
 * VALU bound).  The subtraction from the middle is rounded at an ulp of S, far inside the margin; every loop and its fix-up use this one
 * function, so they agree on which terms are "near". */
template <int INTERP>
__device__ __forceinline__ bool edge_near(float index, int S, float margin)
{
	const float half = 0.5f * (edge_hi<INTERP>(S) - edge_lo<INTERP>()), mid = edge_lo<INTERP>() + half;        /* (scalars: exact halves of integers) */
	return __builtin_fabsf(__builtin_fabsf(index - mid) - half) < margin;
}
/* strictly inside the valid range, a margin away from both ends: what a fast kernel may decide by its own index */
template <int INTERP>
__device__ __forceinline__ bool edge_clear(float index, int S, float margin)
{
	return index >= edge_lo<INTERP>() + margin && index < edge_hi<INTERP>(S) - margin;
}

struct Voxel { float wx, wy, wz, xx, xy, xz; };

/* `c ? v.a : v.b` on two members is an LVALUE in C++: clang selects the ADDRESS and loads through it, which turns the struct into
 * a dynamically indexed stack object (28 bytes of scratch in every kernel that inlined this header).  By-value parameters force the
 * two loads first and a plain select after them. */
__device__ __forceinline__ float pick(bool c, float a, float b) { return c ? a : b; }

/* das.glsl:374-380: unit cube -> world (voxel_transform) -> transducer (xdc_transform); FORCES / READI voxels arrive already in
 * transducer space (the host pre-multiplies, beamformer_core.c:913-915) */
template <int FAMILY, typename P>
__device__ __forceinline__ Voxel exact_voxel(const P &p, uint32_t x, uint32_t y, uint32_t z)
{
	#pragma clang fp contract(off)
	Voxel v;
	const float px = (float)x / (float)(p.size[0] > 2u ? p.size[0] - 1u : 1u);
	const float py = (float)y / (float)(p.size[1] > 2u ? p.size[1] - 1u : 1u);
	const float pz = (float)z / (float)(p.size[2] > 2u ? p.size[2] - 1u : 1u);
	const auto &m = p.voxel_transform;
	v.wx = m[0] * px + m[4] * py + m[8]  * pz + m[12];
	v.wy = m[1] * px + m[5] * py + m[9]  * pz + m[13];
	v.wz = m[2] * px + m[6] * py + m[10] * pz + m[14];
	if constexpr (FAMILY == BF_DAS_RCA || FAMILY == BF_DAS_HERCULES) {
		const auto &t = p.xdc_transform;
		v.xx = t[0] * v.wx + t[4] * v.wy + t[8]  * v.wz + t[12];
		v.xy = t[1] * v.wx + t[5] * v.wy + t[9]  * v.wz + t[13];
		v.xz = t[2] * v.wx + t[6] * v.wy + t[10] * v.wz + t[14];
		} else {
		v.xx = v.wx; v.xy = v.wy; v.xz = v.wz;
	}
	return v;
}

/* das.glsl:187-202 (sin / cos / focus of the steering angle: the host's table, glibc sinf / cosf of the same radians the oracle takes) */
struct Tx { float sin_a, cos_a, focus_x, focus_z; uint32_t flags; };      /* BfTransmit without its padding (an array member would pin the copy to scratch) */

__device__ __forceinline__ float exact_transmit_distance(const Tx &t, const Voxel &v)
{
	#pragma clang fp contract(off)
	if (t.flags & BF_TX_NONE) return 0.f;
	const float px = pick((t.flags & BF_TX_ROWS) != 0, v.wy, v.wx), pz = v.wz;
	if (t.flags & BF_TX_PLANE) return px * t.sin_a + pz * t.cos_a;
	const float dx = px - t.focus_x, dz = pz - t.focus_z;
	return __builtin_sqrtf(dx * dx + dz * dz);
}

/* das.glsl:126-130 */
template <typename P>
__device__ __forceinline__ float exact_sample_index(float distance, const P &p)
{
	#pragma clang fp contract(off)
	const float time = distance / p.speed_of_sound + p.time_offset;
	return time * p.sampling_frequency;
}

template <typename P>
__device__ __forceinline__ Tx load_transmit(const P &p, int a)
{
	typedef __attribute__((address_space(4))) const f32x4 const_f32x4;
	const_f32x4 *tc = (const_f32x4 *)(uintptr_t)p.transmits;
	const f32x4 lo = tc[2 * a], hi = tc[2 * a + 1];
	Tx t;
	t.sin_a = lo.x; t.cos_a = lo.y; t.focus_x = lo.z; t.focus_z = lo.w;
	{ const float f = hi.x; t.flags = __builtin_bit_cast(uint32_t, f); }
	return t;
}

/* The sample index of term (channel, transmit) of voxel (x, y, z) as the shader forms it.
 *   RCA:      transmit = acquisition (das.glsl:204-231)
 *   HERCULES: transmit = acquisition index t >= sparse; its element is sparse_elements[t - sparse] or t (das.glsl:233-286)
 *   FORCES:   likewise (das.glsl:288-321)
 *   READI:    transmit = tx_group * acquisition_count + tx_event, the element index itself (das.glsl:323-366) */
template <int FAMILY, typename P>
__device__ __forceinline__ float exact_index(const P &p, const Voxel &v, int channel, int transmit)
{
	#pragma clang fp contract(off)
	const float rx_channel = (float)channel;
	if constexpr (FAMILY == BF_DAS_RCA) {
		const Tx t = load_transmit(p, transmit);
		const bool  rx_rows = (t.flags & BF_RX_ROWS) != 0;
		const float xw0 = pick(rx_rows, v.xy, v.xx), xw1 = v.xz;
		const float rx_lateral = rx_channel * pick(rx_rows, p.pitch[1], p.pitch[0]);
		const float rv0 = xw0 - rx_lateral, rv1 = xw1 - 0.f;
		return exact_sample_index(exact_transmit_distance(t, v) + __builtin_sqrtf(rv0 * rv0 + rv1 * rv1), p);
	} else if constexpr (FAMILY == BF_DAS_HERCULES) {
		const Tx t0 = load_transmit(p, 0);
		const bool  rx_cols = (t0.flags & BF_RX_COLUMNS) != 0;
		const float transmit_index = exact_sample_index(exact_transmit_distance(t0, v), p);
		const float z2 = v.xz * v.xz;
		const float tx_channel = p.sparse ? (float)p.sparse_elements[transmit - 1] : (float)transmit;
		float ex = v.xx, ey = v.xy;
		if (rx_cols) { ex -= rx_channel * p.pitch[0]; ex *= ex; ey = v.xy - tx_channel * p.pitch[1]; ey *= ey; }
		else         { ey -= rx_channel * p.pitch[1]; ey *= ey; ex = v.xx - tx_channel * p.pitch[0]; ex *= ex; }
		const float eds = ex + ey;
		return transmit_index + __builtin_sqrtf(z2 + eds) * p.sampling_frequency / p.speed_of_sound;
	} else {
		const float z2  = v.xz * v.xz;
		const float tyd = v.xy - p.pitch[1] * (float)p.channel_count / 2.f;
		const float tyz2 = tyd * tyd + z2;
		const float rxd = v.xx - rx_channel * p.pitch[0];
		const float receive_index = exact_sample_index(__builtin_sqrtf(rxd * rxd + z2), p);
		float tx_element;
		if constexpr (FAMILY == BF_DAS_READI) {
			const int A = p.acquisition_count;
			tx_element = (float)(transmit / A) * (float)A + (float)(transmit % A);
		} else {
			tx_element = p.sparse ? (float)p.sparse_elements[transmit - 1] : (float)transmit;
		}
		const float txd = v.xx - p.pitch[0] * tx_element;
		const float transmit_index = __builtin_sqrtf(tyz2 + txd * txd) * p.sampling_frequency / p.speed_of_sound;
		return receive_index + transmit_index;
	}
}

/* The launch arguments as they lie in the kernel-argument segment (BfDasArgs is every DAS kernel's first argument).  The fast kernels
 * hand THIS to the functions of this file from their fix-up loops -- `const BfDasArgs p` itself would keep ~60 scalars alive across
 * their inner loops for the sake of a path that almost never runs (the separable kernel: 62 scalar and 5 vector spills at its 64-VGPR
 * limit); read through a pointer the compiler cannot see through, the values are loaded when the path runs and dead again after. */
typedef __attribute__((address_space(4))) const BfDasArgs KernelArgs;
__device__ __forceinline__ KernelArgs &kernel_args()
{
	KernelArgs *ka = (KernelArgs *)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(ka));
	return *ka;
}

/* One whole term -- apodization x rotate_iq(sample_rf(exact index)) -- added to a voxel's sums: what a fast kernel's fix-up loop
 * calls for a term it declined to decide (and therefore left out of its own sums).  RCA, HERCULES, FORCES (no fast kernel takes
 * READI).  The weight follows das.hip's general kernel: continuous at the aperture's edge, so not a step. */
template <int FAMILY, int INTERP, bool CPLX, bool CW, typename P>
__device__ __forceinline__ void edge_term(const P &p, uint32_t x, uint32_t y, uint32_t z, int channel, int transmit,
                                          sample_t<CPLX> &coherent, float &incoherent)
{
	static_assert(FAMILY != BF_DAS_READI, "READI runs on the general kernel only");
	const Voxel v = exact_voxel<FAMILY>(p, x, y, z);
	const int   S = p.sample_count, A = p.acquisition_count;
	float weight;
	if constexpr (FAMILY == BF_DAS_RCA) {
		const Tx t = load_transmit(p, transmit);
		const bool  rx_rows = (t.flags & BF_RX_ROWS) != 0;
		const float dx = pick(rx_rows, v.xy, v.xx) - (float)channel * pick(rx_rows, p.pitch[1], p.pitch[0]);
		const float a_arg = __builtin_fabsf(dx * (p.f_number * hw_rcp(__builtin_fabsf(v.xz))));
		if (!(a_arg < 0.5f)) return;
		weight = apodize(a_arg);
	} else if constexpr (FAMILY == BF_DAS_HERCULES) {
		const bool  rx_cols = (load_transmit(p, 0).flags & BF_RX_COLUMNS) != 0;
		const float tx_channel = p.sparse ? (float)p.sparse_elements[transmit - 1] : (float)transmit;
		const float rd = rx_cols ? v.xx - (float)channel * p.pitch[0] : v.xy - (float)channel * p.pitch[1];
		const float td = rx_cols ? v.xy - tx_channel * p.pitch[1]     : v.xx - tx_channel * p.pitch[0];
		const float eds = td * td + rd * rd;
		const float f_over_z = __builtin_fabsf(p.f_number * hw_rcp(v.xz));
		if (!(eds < 0.25f / (f_over_z * f_over_z))) return;
		weight = (transmit == 0 ? p.first_transmit_weight : 1.0f) * apodize(f_over_z * hw_sqrt(eds));
	} else {
		const float dx = v.xx - (float)channel * p.pitch[0];
		const float a_arg = __builtin_fabsf(dx * (p.f_number * hw_rcp(v.xz)));
		if (!(a_arg < 0.5f)) return;
		weight = apodize(a_arg);
	}
	const float index = exact_index<FAMILY>(p, v, channel, transmit);
	const sample_t<CPLX> s = weight * sample_rf<INTERP, CPLX>((const char *)p.rf, (channel * A + transmit) * S, index, p);
	coherent += s;
	if constexpr (CW) {
		if constexpr (CPLX) incoherent += hw_sqrt(s.x * s.x + s.y * s.y);
		else                incoherent += __builtin_fabsf(s);
	}
}

/* The same for a kernel whose loops did NOT leave the term out but decided it with their own index `fast_index` (the factored and
 * HERCULES kernels: their loops stay exactly as they are; a pass at the end of the kernel walks the few chunks that can hold such terms):
 * the voxel's sums are corrected by  weight x (sample_rf(exact index) - sample_rf(fast_index)).  sample_rf applies the very range test the
 * loops applied to the very same index bits, so the subtracted term is what the loops added (or nothing, if they dropped it) up to the
 * rounding of one term's arithmetic -- 1e-7 of a single tap; where both indices lie on the same side of the row's end the correction is
 * that small too, where they straddle it, it is the whole tap: the oracle's decision. */
template <int FAMILY, int INTERP, bool CPLX, bool CW, typename P>
__device__ __forceinline__ void edge_correct(const P &p, uint32_t x, uint32_t y, uint32_t z, int channel, int transmit, float fast_index,
                                             sample_t<CPLX> &coherent, float &incoherent)
{
	static_assert(FAMILY != BF_DAS_READI, "READI runs on the general kernel only");
	const Voxel v = exact_voxel<FAMILY>(p, x, y, z);
	const int   S = p.sample_count, A = p.acquisition_count;
	float weight;
	if constexpr (FAMILY == BF_DAS_RCA) {
		const Tx t = load_transmit(p, transmit);
		const bool  rx_rows = (t.flags & BF_RX_ROWS) != 0;
		const float dx = pick(rx_rows, v.xy, v.xx) - (float)channel * pick(rx_rows, p.pitch[1], p.pitch[0]);
		const float a_arg = __builtin_fabsf(dx * (p.f_number * hw_rcp(__builtin_fabsf(v.xz))));
		if (!(a_arg < 0.5f)) return;
		weight = apodize(a_arg);
	} else if constexpr (FAMILY == BF_DAS_HERCULES) {
		const bool  rx_cols = (load_transmit(p, 0).flags & BF_RX_COLUMNS) != 0;
		const float tx_channel = p.sparse ? (float)p.sparse_elements[transmit - 1] : (float)transmit;
		const float rd = rx_cols ? v.xx - (float)channel * p.pitch[0] : v.xy - (float)channel * p.pitch[1];
		const float td = rx_cols ? v.xy - tx_channel * p.pitch[1]     : v.xx - tx_channel * p.pitch[0];
		const float eds = td * td + rd * rd;
		const float f_over_z = __builtin_fabsf(p.f_number * hw_rcp(v.xz));
		if (!(eds < 0.25f / (f_over_z * f_over_z))) return;
		weight = (transmit == 0 ? p.first_transmit_weight : 1.0f) * apodize(f_over_z * hw_sqrt(eds));
	} else {
		const float dx = v.xx - (float)channel * p.pitch[0];
		const float a_arg = __builtin_fabsf(dx * (p.f_number * hw_rcp(v.xz)));
		if (!(a_arg < 0.5f)) return;
		weight = apodize(a_arg);
	}
	const float index = exact_index<FAMILY>(p, v, channel, transmit);
	const int   row = (channel * A + transmit) * S;
	/* (one evaluation after the other, as a loop the compiler may not unroll: interleaved, the two cubic interpolations of IQ samples hold
	 * twice the registers, and the kernel's occupancy is set by its most expensive point -- this rare one) */
	#pragma unroll 1
	for (int which = 0; which < 2; which++) {
		const float at = which ? fast_index : index, sign = which ? -weight : weight;
		const sample_t<CPLX> s = sign * sample_rf<INTERP, CPLX>((const char *)p.rf, row, at, p);
		coherent += s;
		if constexpr (CW) {
			float magnitude;
			if constexpr (CPLX) magnitude = hw_sqrt(s.x * s.x + s.y * s.y); else magnitude = __builtin_fabsf(s);
			incoherent += which ? -magnitude : magnitude;
		}
	}
}

} /* namespace bfx */
#endif
