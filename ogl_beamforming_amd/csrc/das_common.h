/* das_common.h -- device helpers shared by das.hip (general path) and das_separable.hip
 * (row-column fast path): hardware transcendental wrappers, element-aligned gathers and the
 * branch-free restatement of sample_rf (shaders/das.glsl:99-124, cubic :67-97).
 *
 * Branch-free on purpose: the reference tests the sample index and only then loads.  On
 * CDNA4 a load inside a divergent branch has to be waited for inside that branch, which
 * serialises every gather of a wave.  Here the tap address is clamped into the row (always a
 * legal address) and the validity test becomes a 0/1 weight folded into the interpolation
 * weights, so the compiler can issue the gathers of several (channel, transmit) terms
 * back to back and wait once.  The result is bit-identical to the branchy form: a valid
 * sample is multiplied by 1.0f, an invalid one contributes +0.
 */
#ifndef BF_DAS_COMMON_H
#define BF_DAS_COMMON_H

#include <hip/hip_runtime.h>
#include <type_traits>
#include "bf_kernels.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
/* gathers are only element aligned: 8 B for complex, 4 B for real samples */
typedef f32x4 f32x4_a8 __attribute__((aligned(8)));
typedef f32x4 f32x4_a4 __attribute__((aligned(4)));
typedef f32x2 f32x2_a4 __attribute__((aligned(4)));

#define BF_INTERP_NEAREST 0
#define BF_INTERP_LINEAR  1
#define BF_INTERP_CUBIC   2

template <bool CPLX> using sample_t = typename std::conditional<CPLX, f32x2, float>::type;

__device__ __forceinline__ float hw_sqrt(float x)      { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float hw_rcp(float x)       { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float hw_rsq(float x)       { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float hw_fract(float x)     { return __builtin_amdgcn_fractf(x); }
__device__ __forceinline__ float hw_sin_turns(float x) { return __builtin_amdgcn_sinf(x); }   /* sin(2 pi x) */
__device__ __forceinline__ float hw_cos_turns(float x) { return __builtin_amdgcn_cosf(x); }   /* cos(2 pi x) */

/* distance / speed_of_sound, correctly rounded in all but rare cases: the product with the reciprocal
 * plus one Newton correction (2 FMAs).  The plain product carries the reciprocal's own rounding error as
 * a SYSTEMATIC relative bias of every delay (1.9e-8 for c = 1540 m/s): 4e-5 of a sample at index 2000,
 * i.e. 1.2e-4 rad of demodulation phase common to all taps -- which is 1.2e-4 of a coherent peak, the
 * whole parity budget.  Used wherever a delay is computed once per table entry or per voxel term (the
 * reference divides: sample_index, das.glsl:126-130); the per-pair loops of das.hip / das_hercules.hip keep
 * the reciprocal. */
__device__ __forceinline__ float div_speed_of_sound(float distance, const BfDasArgs &p)
{
	float q = distance * p.inv_speed_of_sound;
	float e = __builtin_fmaf(-q, p.speed_of_sound, distance);
	return __builtin_fmaf(e, p.inv_speed_of_sound, q);
}

__device__ __forceinline__ float div_speed_of_sound(float distance, float inv_speed_of_sound, float speed_of_sound)
{
	float q = distance * inv_speed_of_sound;
	float e = __builtin_fmaf(-q, speed_of_sound, distance);
	return __builtin_fmaf(e, inv_speed_of_sound, q);
}

/* The LDS-staged kernels (das_staged*.hip) trust the host's bound on the delay spread of a tile (plan_staged, das_select.cpp): a
 * position outside the staged window would read a neighbouring transmit's window -- wrong voxels, no fault.  Their range-checked loop
 * (every wave under the STAGED_CHECKED hook) therefore tests every term; an offending wave raises a flag in the first LDS word (the
 * kernels keep two unused window elements at LDS address 0 and have no static LDS), and at the end of the block one thread adds it to
 * BfSeparableArgs::violations -- read through the kernel-argument segment there, so that neither flag nor pointer costs the loops a
 * register (the headline instance sits at its 64-VGPR limit).  The staged kernels take (BfDasArgs, BfSeparableArgs) in that order. */
__device__ __forceinline__ void staged_violation_clear(uint32_t tid)
{
	uint32_t at = 0;
	asm volatile("" : "+v"(at));
	if (tid == 0) *(__attribute__((address_space(3))) volatile uint32_t *)(uintptr_t)at = 0u;
}
__device__ __forceinline__ void staged_violation_raise()
{
	uint32_t at = 0;
	asm volatile("" : "+v"(at));
	*(__attribute__((address_space(3))) volatile uint32_t *)(uintptr_t)at = 1u;
}
/* every thread of the block calls this once, after its last channel */
__device__ __forceinline__ void staged_violation_report(uint32_t tid)
{
	__syncthreads();
	uint32_t at = 0;
	asm volatile("" : "+v"(at));
	if (tid == 0 && *(__attribute__((address_space(3))) volatile uint32_t *)(uintptr_t)at) {
		typedef __attribute__((address_space(4))) const BfSeparableArgs const_sep;
		const_sep *q = (const_sep *)((__attribute__((address_space(4))) const char *)__builtin_amdgcn_kernarg_segment_ptr() + ((sizeof(BfDasArgs) + 7u) & ~(size_t)7u));
		uint32_t *counter = q->violations;
		if (counter) atomicAdd(counter, 1u);
	}
}

/* (int)floor(x) in one instruction (hipcc emits v_floor_f32 + v_cvt_i32_f32) */
__device__ __forceinline__ int cvt_floor_i32(float x)
{
	int r;
	asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
	return r;
}

/* das.glsl:138-152: cos(pi a)^2 */
__device__ __forceinline__ float apodize(float a)
{
	float c = hw_cos_turns(0.5f * a);
	return c * c;
}

template <bool CPLX>
__device__ __forceinline__ sample_t<CPLX> zero_sample()
{
	if constexpr (CPLX) return f32x2{0.f, 0.f}; else return 0.f;
}

/* byte offsets stay in 32 bits: the host rejects DAS inputs of 4 GiB and more */
template <typename T>
__device__ __forceinline__ T gather(const char *rf, uint32_t byte_offset)
{
	return *reinterpret_cast<const T *>(rf + byte_offset);
}

/* the same load `BYTES` further on: the constant rides in the instruction's offset field (base + zero-extended 32-bit offset + constant),
 * where `byte_offset + BYTES` in 32 bits would be a VALU addition the compiler may not fold (it can wrap) */
template <typename T, int BYTES>
__device__ __forceinline__ T gather_at(const char *rf, uint32_t byte_offset)
{
	return *reinterpret_cast<const T *>(rf + (uint64_t)byte_offset + BYTES);
}

/* Catmull-Rom weights of the taps s0..s3 at t in [0, 1) between s1 and s2 (das.glsl:67-97 multiplied out):
 *   w0 = -t u^2 / 2,  w3 = -t^2 u / 2,  w1 = u + t u (1 - 3t/2),  w2 = t + t u (3t/2 - 1/2),   u = 1 - t
 * (w0 + w1 + w2 + w3 = 1; t = 1/2: -1/16, 9/16, 9/16, -1/16).  Nine scalar operations. */
__device__ __forceinline__ void bf_catmull_rom(float t, float &w0, float &w1, float &w2, float &w3)
{
	const float u = 1.0f - t, tu = t * u, a = -0.5f * tu;
	w0 = a * u;
	w3 = a * t;
	w1 = __builtin_fmaf(tu, __builtin_fmaf(-1.5f, t, 1.0f), u);
	w2 = __builtin_fmaf(tu, __builtin_fmaf(1.5f, t, -0.5f), t);
}

template <typename M>      /* const float * in any address space */
__device__ __forceinline__ void m4_point(M m, float x, float y, float z, float &ox, float &oy, float &oz)
{
	ox = m[0] * x + m[4] * y + m[8]  * z + m[12];
	oy = m[1] * x + m[5] * y + m[9]  * z + m[13];
	oz = m[2] * x + m[6] * y + m[10] * z + m[14];
}

/* The interpolation of sample_rf split in three steps so that a caller can issue the
 * gathers of several terms before consuming any of them (software pipelining by hand; hipcc
 * otherwise waits for each load right after issuing it once registers are capped):
 *   tap_setup  : index -> byte offset inside the row (clamped to a legal address) + weights
 *                (zero when the index is outside the valid range)
 *   tap_load   : the 1 or 2 vector loads
 *   tap_finish : weighted sum
 *   Nearest: valid 0 <= index < S - 0.5, tap round(index)
 *   Linear:  valid 0 <= index < S - 1,   taps k, k+1
 *   Cubic:   valid 1 <= index < S - 2,   taps k-1 .. k+2 (Catmull-Rom Hermite, das.glsl:67-97)
 * last = SampleCount - 1, sample_count_f = (float)SampleCount. */
template <int INTERP> struct Tap;
template <> struct Tap<BF_INTERP_NEAREST> { uint32_t off; float w0; };
template <> struct Tap<BF_INTERP_LINEAR>  { uint32_t off; float w0, w1; };
template <> struct Tap<BF_INTERP_CUBIC>   { uint32_t off; float w0, w1, w2, w3; };

template <int INTERP, bool CPLX> struct TapData;
template <> struct TapData<BF_INTERP_NEAREST, true>  { f32x2 a; };
template <> struct TapData<BF_INTERP_NEAREST, false> { float a; };
template <> struct TapData<BF_INTERP_LINEAR,  true>  { f32x4 a; };
template <> struct TapData<BF_INTERP_LINEAR,  false> { f32x2 a; };
template <> struct TapData<BF_INTERP_CUBIC,   true>  { f32x4 a, b; };
template <> struct TapData<BF_INTERP_CUBIC,   false> { f32x4 a; };

template <int INTERP, bool CPLX>
__device__ __forceinline__ Tap<INTERP> tap_setup(float index, float sample_count_f, int last)
{
	constexpr uint32_t ES = CPLX ? 8 : 4;
	Tap<INTERP> tap;
	if constexpr (INTERP == BF_INTERP_NEAREST) {
		tap.w0 = (index >= 0.f && index < sample_count_f - 0.5f) ? 1.f : 0.f;
		int k  = (int)__builtin_roundf(index);
		k = k < 0 ? 0 : (k > last ? last : k);
		tap.off = (uint32_t)k * ES;
	} else if constexpr (INTERP == BF_INTERP_LINEAR) {
		/* 0 <= index < S-1  <=>  (unsigned)floor(index) < S-1 */
		float    t  = hw_fract(index);
		uint32_t k  = (uint32_t)(int)__builtin_floorf(index);
		float    w  = k < (uint32_t)last ? 1.f : 0.f;
		k = k < (uint32_t)(last - 1) ? k : (uint32_t)(last - 1);
		tap.off = k * ES;
		tap.w1  = w * t;               /* t and (1 - t) when w == 1 */
		tap.w0  = w - tap.w1;
	} else {
		/* 1 <= index < S-2  <=>  (unsigned)(floor(index) - 1) < S-3 */
		float    t  = hw_fract(index);
		uint32_t k  = (uint32_t)((int)__builtin_floorf(index) - 1);
		float    w  = k < (uint32_t)(last - 2) ? 1.f : 0.f;
		k = k < (uint32_t)(last - 3) ? k : (uint32_t)(last - 3);
		tap.off = k * ES;
		float t2 = t * t, t3 = t2 * t;
		/* Hermite basis, tangents 0.5 (P2 - P0) and 0.5 (P3 - P1) */
		tap.w0 = w * ( 2.f * t3 - 3.f * t2 + 1.f);
		tap.w1 = w * (-2.f * t3 + 3.f * t2);
		tap.w2 = w * (       t3 - 2.f * t2 + t);
		tap.w3 = w * (       t3 -       t2);
	}
	return tap;
}

template <int INTERP, bool CPLX>
__device__ __forceinline__ TapData<INTERP, CPLX> tap_load(const char *rf, uint32_t byte_offset)
{
	TapData<INTERP, CPLX> d;
	if constexpr (INTERP == BF_INTERP_NEAREST) {
		d.a = gather<sample_t<CPLX>>(rf, byte_offset);
	} else if constexpr (INTERP == BF_INTERP_LINEAR) {
		if constexpr (CPLX) d.a = gather<f32x4_a8>(rf, byte_offset);
		else                d.a = gather<f32x2_a4>(rf, byte_offset);
	} else {
		if constexpr (CPLX) { d.a = gather<f32x4_a8>(rf, byte_offset); d.b = gather<f32x4_a8>(rf, byte_offset + 16); }
		else                { d.a = gather<f32x4_a4>(rf, byte_offset); }
	}
	return d;
}

template <int INTERP, bool CPLX>
__device__ __forceinline__ sample_t<CPLX> tap_finish(const Tap<INTERP> &tap, const TapData<INTERP, CPLX> &d)
{
	if constexpr (INTERP == BF_INTERP_NEAREST) {
		return tap.w0 * d.a;
	} else if constexpr (INTERP == BF_INTERP_LINEAR) {
		if constexpr (CPLX) return tap.w0 * f32x2{d.a.x, d.a.y} + tap.w1 * f32x2{d.a.z, d.a.w};
		else                return tap.w0 * d.a.x + tap.w1 * d.a.y;
	} else {
		if constexpr (CPLX) {
			f32x2 s0 = {d.a.x, d.a.y}, s1 = {d.a.z, d.a.w}, s2 = {d.b.x, d.b.y}, s3 = {d.b.z, d.b.w};
			f32x2 T1 = 0.5f * (s2 - s0), T2 = 0.5f * (s3 - s1);
			return tap.w0 * s1 + tap.w1 * s2 + tap.w2 * T1 + tap.w3 * T2;
		} else {
			float T1 = 0.5f * (d.a.z - d.a.x), T2 = 0.5f * (d.a.w - d.a.y);
			return tap.w0 * d.a.y + tap.w1 * d.a.z + tap.w2 * T1 + tap.w3 * T2;
		}
	}
}

/* Interpolated sample of the row starting at row_byte_offset, WITHOUT the IQ rotation. */
template <int INTERP, bool CPLX>
__device__ __forceinline__ sample_t<CPLX> interpolate(const char *rf, uint32_t row_byte_offset, float index,
                                                      float sample_count_f, int last)
{
	Tap<INTERP> tap = tap_setup<INTERP, CPLX>(index, sample_count_f, last);
	TapData<INTERP, CPLX> d = tap_load<INTERP, CPLX>(rf, row_byte_offset + tap.off);
	return tap_finish<INTERP, CPLX>(tap, d);
}

/* das.glsl:54-61 with the angle in turns, reduced to [0,1) */
template <typename P>
__device__ __forceinline__ f32x2 rotate_iq(f32x2 iq, float index, const P &p)
{
	float turns = hw_fract(index * p.turns_per_sample);
	float c = hw_cos_turns(turns), s = hw_sin_turns(turns);
	return f32x2{c * iq.x - s * iq.y, s * iq.x + c * iq.y};
}

/* das.glsl:99-124 (+ cubic :67-97), with the IQ rotation.  rf_offset is the element index of the row's sample 0.
 * The reference's test-then-load form: what the general kernel (das.hip) runs -- at 8 waves per SIMD and VALU bound, the extra
 * select / clamp work of the branch-free form above costs it more than the serialised gathers do -- and what the row-end
 * fix-up of every fast kernel evaluates its few terms with (das_exact.h). */
template <int INTERP, bool CPLX, typename P>
__device__ __forceinline__ sample_t<CPLX> sample_rf(const char *rf, int rf_offset, float index, const P &p)
{
	constexpr uint32_t ES = CPLX ? 8 : 4;
	sample_t<CPLX> result = zero_sample<CPLX>();
	const float S = (float)p.sample_count;
	if constexpr (INTERP == BF_INTERP_NEAREST) {
		if (index >= 0.f && index < S - 0.5f) {
			int k = (int)__builtin_roundf(index);
			result = gather<sample_t<CPLX>>(rf, (uint32_t)(rf_offset + k) * ES);
			if constexpr (CPLX) result = rotate_iq(result, index, p);
		}
	} else if constexpr (INTERP == BF_INTERP_LINEAR) {
		/* 0 <= index < S-1  <=>  (unsigned)floor(index) < S-1: one convert and one compare */
		uint32_t k = (uint32_t)cvt_floor_i32(index);
		if (k < (uint32_t)(p.sample_count - 1)) {
			float t = hw_fract(index);
			uint32_t off = ((uint32_t)rf_offset + k) * ES;
			if constexpr (CPLX) {
				f32x4 v = gather<f32x4_a8>(rf, off);
				f32x2 a = {v.x, v.y}, b = {v.z, v.w};
				result = a + t * (b - a);
				result = rotate_iq(result, index, p);
			} else {
				f32x2 v = gather<f32x2_a4>(rf, off);
				result = v.x + t * (v.y - v.x);
			}
		}
	} else {
		/* 1 <= index < S-2  <=>  (unsigned)(floor(index) - 1) < S-3 */
		uint32_t k = (uint32_t)(cvt_floor_i32(index) - 1);
		if (k < (uint32_t)(p.sample_count - 3)) {
			float t = hw_fract(index);
			uint32_t off = ((uint32_t)rf_offset + k) * ES;
			float t2 = t * t, t3 = t2 * t;
			/* Hermite basis with tangents 0.5 (P2 - P0), 0.5 (P3 - P1) */
			float b0 =  2.f * t3 - 3.f * t2 + 1.f;
			float b1 = -2.f * t3 + 3.f * t2;
			float b2 =        t3 - 2.f * t2 + t;
			float b3 =        t3 -       t2;
			if constexpr (CPLX) {
				f32x4 lo = gather<f32x4_a8>(rf, off), hi = gather<f32x4_a8>(rf, off + 16);
				f32x2 s0 = {lo.x, lo.y}, s1 = {lo.z, lo.w}, s2 = {hi.x, hi.y}, s3 = {hi.z, hi.w};
				f32x2 T1 = 0.5f * (s2 - s0), T2 = 0.5f * (s3 - s1);
				result = b0 * s1 + b1 * s2 + b2 * T1 + b3 * T2;
				result = rotate_iq(result, index, p);
			} else {
				f32x4 v = gather<f32x4_a4>(rf, off);
				float T1 = 0.5f * (v.z - v.x), T2 = 0.5f * (v.w - v.y);
				result = b0 * v.y + b1 * v.z + b2 * T1 + b3 * T2;
			}
		}
	}
	return result;
}

#endif
