/* microbench.hip -- the per-CU ceilings DESIGN.md prices the DAS kernels against, measured.
 *
 * One standalone program (hipcc --offload-arch=gfx950 tools/microbench.hip -o tools/bin/microbench),
 * prints one JSON object.  Every probe is a loop of inline-asm instructions timed INSIDE the
 * kernel with s_memtime (shader clock) next to s_memrealtime (100 MHz), so rates come out per
 * clock per CU at the clock the part actually sustains, and the sustained clock itself is
 * reported.  Probes (each at several waves per SIMD):
 *   valu      v_fma_f32, v_pk_fma_f32, v_sqrt_f32, v_sin_f32, v_rcp_f32, v_cvt_flr_i32_f32,
 *             v_fract_f32: wave-instructions per clock per SIMD
 *   gather    per-lane global_load_dword / dwordx2 / dwordx4 with 4 loads in flight per wave,
 *             from a window resident in the CU's L1 (8 KB per block), in the XCD's L2 (2 MB
 *             shared) -- address patterns: contiguous (coalesced), DAS-like (8-byte aligned,
 *             neighbouring lanes 0..8 B apart: the interpolation taps of neighbouring voxels),
 *             random inside the window: bytes per clock per CU
 *   lds       ds_read_b64, ds_read_b128 (16-B aligned), ds_read2_b64 (16 B at 8-B alignment) with
 *             the same three patterns: bytes per clock per CU
 * Nothing here is on the product path; it is the evidence behind roofline.binding in bench.py.
 */
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Stamp { uint64_t cycles, realtime; };   /* per wave: shader-clock ticks and 100 MHz ticks over the timed loop */

__device__ __forceinline__ uint64_t memtime()     { return __builtin_amdgcn_s_memtime(); }
__device__ __forceinline__ uint64_t memrealtime() { return __builtin_amdgcn_s_memrealtime(); }

/* ------------------------------------------------------------------ VALU issue */
enum { OP_FMA, OP_PK_FMA, OP_SQRT, OP_SIN, OP_RCP, OP_CVT_FLR, OP_FRACT, OP_MUL, OP_ADD, OP_LSHL_ADD, OP_PK_ADD, OP_PK_ADD_SGPR, OP_PK_ADD_NEG,
       OP_PK_FMA_SEL, OP_MUL_U24, OP_LSHL_B16, OP_PK_LSHL_B16, OP_LSHL_B32, OP_COUNT };
static const char *op_name[OP_COUNT] = {"v_fma_f32", "v_pk_fma_f32", "v_sqrt_f32", "v_sin_f32", "v_rcp_f32",
                                        "v_cvt_flr_i32_f32", "v_fract_f32", "v_mul_f32", "v_add_f32", "v_lshl_add_u32", "v_pk_add_f32",
                                        "v_pk_add_f32 (scalar-pair operand, op_sel_hi:[1,0])", "v_pk_add_f32 (neg_lo neg_hi on one operand)",
                                        "v_pk_fma_f32 (op_sel_hi:[0,1,1]: one half broadcast)", "v_mul_u32_u24",
                                        "v_lshlrev_b16", "v_pk_lshlrev_b16", "v_lshlrev_b32"};

template <int OP>
__global__ __launch_bounds__(1024) void valu_probe(Stamp *stamps, float *sink, int iters)
{
	/* 8 independent chains per lane, 4 rounds per iteration = 32 instructions per iteration */
	float a[8]; f32x2 p[8];
	for (int k = 0; k < 8; k++) { a[k] = 1.0f + 0.001f * (float)(threadIdx.x + k); p[k] = f32x2{a[k], a[k] + 0.5f}; }
	float b = 0.999f, c = 0.0001f;
	f32x2 pb = {0.999f, 0.998f}, pc = {0.0001f, 0.0002f};
	uint64_t spair = 0x3a83126f3a83126full;                  /* two small floats in a scalar register pair */
	asm volatile("" : "+s"(spair));
	uint32_t shifts = 0x000f0004u;                           /* v_pk_lshlrev_b16: low half << 4, high half << 15 */
	asm volatile("" : "+v"(shifts));
	__syncthreads();
	uint64_t t0 = memtime(), r0 = memrealtime();
	for (int i = 0; i < iters; i++) {
		#pragma unroll
		for (int r = 0; r < 4; r++) {
			#pragma unroll
			for (int k = 0; k < 8; k++) {
				if constexpr (OP == OP_FMA)     asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
				if constexpr (OP == OP_MUL)     asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
				if constexpr (OP == OP_ADD)     asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
				if constexpr (OP == OP_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[k]) : "v"(b));
				if constexpr (OP == OP_PK_ADD)  asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc));
				if constexpr (OP == OP_PK_ADD_SGPR) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[k]) : "s"(spair));
				if constexpr (OP == OP_PK_ADD_NEG)  asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[k]) : "v"(pc));
				if constexpr (OP == OP_PK_FMA_SEL)  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[k]) : "v"(pb), "v"(pc));
				if constexpr (OP == OP_MUL_U24) asm volatile("v_mul_u32_u24 %0, 3, %0" : "+v"(a[k]));
				if constexpr (OP == OP_LSHL_B16) asm volatile("v_lshlrev_b16 %0, 4, %0" : "+v"(a[k]));
				if constexpr (OP == OP_PK_LSHL_B16) asm volatile("v_pk_lshlrev_b16 %0, %1, %0" : "+v"(a[k]) : "v"(shifts));
				if constexpr (OP == OP_LSHL_B32) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[k]));
				if constexpr (OP == OP_PK_FMA)  asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pb), "v"(pc));
				if constexpr (OP == OP_SQRT)    asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
				if constexpr (OP == OP_SIN)     asm volatile("v_sin_f32 %0, %0" : "+v"(a[k]));
				if constexpr (OP == OP_RCP)     asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
				if constexpr (OP == OP_CVT_FLR) asm volatile("v_cvt_flr_i32_f32 %0, %0" : "+v"(a[k]));
				if constexpr (OP == OP_FRACT)   asm volatile("v_fract_f32 %0, %0" : "+v"(a[k]));
			}
		}
	}
	uint64_t t1 = memtime(), r1 = memrealtime();
	float s = 0.f;
	for (int k = 0; k < 8; k++) s += a[k] + p[k].x + p[k].y;
	if (s == 12345.678f) sink[0] = s;
	if ((threadIdx.x & 63) == 0) {
		uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
		stamps[wave] = Stamp{t1 - t0, r1 - r0};
	}
}

__global__ void shift_semantics(uint32_t *out)
{
	uint32_t a = 0x4b000923u, b = 0x4b000923u, shifts = 0x000f0004u;
	asm volatile("v_lshlrev_b16 %0, 4, %0" : "+v"(a));
	asm volatile("v_pk_lshlrev_b16 %0, %1, %0" : "+v"(b) : "v"(shifts));
	if (threadIdx.x == 0) { out[0] = a; out[1] = b; }
}

/* ------------------------------------------------------------------ the VALU stream of one DAS term
 * What das_staged.hip's unchecked inner loop issues per (voxel, channel, transmit) term once the taps and
 * the table entries are in registers -- window position, tap address, interpolation, rotate-accumulate,
 * |s| for coherency weighting: 11 VALU instructions, here with four independent terms per iteration and
 * no memory instruction at all.  Its rate is the ceiling of a kernel that is bound by VALU issue. */
__global__ __launch_bounds__(1024) void term_probe(Stamp *stamps, float *sink, int iters)
{
	float r = 3.25f + 0.01f * (float)(threadIdx.x & 63), tz[4], mag = 0.f;
	f32x2 tap_s[4], tap_d[4], cs[4], acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f};
	for (int k = 0; k < 4; k++) {
		tz[k] = 1.5f + (float)k; tap_s[k] = f32x2{0.5f + k, 0.25f}; tap_d[k] = f32x2{0.125f, -0.5f};
		cs[k] = f32x2{0.6f, 0.8f};
	}
	uint32_t base = 4096;
	asm volatile("" : "+s"(base));
	__syncthreads();
	uint64_t t0 = memtime(), r0 = memrealtime();
	for (int i = 0; i < iters; i++) {
		#pragma unroll
		for (int k = 0; k < 4; k++) {
			/* plain C++ so that hipcc schedules the four terms of an iteration against each other exactly as it does in
			 * the kernel; the empty asm statements make the inputs opaque (nothing is hoisted) without issuing anything */
			asm volatile("" : "+v"(tz[k]), "+v"(tap_s[k]), "+v"(tap_d[k]), "+v"(cs[k]));
			float rel  = r + tz[k];
			float frac = __builtin_amdgcn_fractf(rel);
			int ki; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ki) : "v"(rel));
			uint32_t at = ((uint32_t)ki << 4) + base;
			f32x2 sv = tap_s[k] + frac * tap_d[k];
			acc1 += sv.x * cs[k];
			acc2 += sv.y * cs[k];
			mag += __builtin_amdgcn_sqrtf(__builtin_fmaf(sv.y, sv.y, sv.x * sv.x));
			asm volatile("" :: "v"(at));
		}
	}
	uint64_t t1 = memtime(), r1 = memrealtime();
	if (mag + acc1.x + acc1.y + acc2.x + acc2.y == 12345.678f) sink[0] = mag;
	if ((threadIdx.x & 63) == 0) {
		uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
		stamps[wave] = Stamp{t1 - t0, r1 - r0};
	}
}

/* The stream das_staged.hip ships (round 2, final form): per batch of four terms 4 packed adds (window position,
 * round-by-magic-number), 4 v_mul_u32_u24 (tap address), 4 + 8 packed fmas (interpolation, rotate-accumulate),
 * 4 x (v_mul, v_fmac, v_sqrt) and 2 packed adds for |s|: 36 VALU instructions per 4 terms. */
/* PARTS: bit 0 = position / rounding / address (4 packed adds + 4 v_lshlrev_b16 per 4 terms), bit 1 = interpolation and
 * rotate-accumulate (12 packed fmas), bit 2 = |s| (4 x v_mul, v_fmac, v_sqrt + 2 packed adds); 7 = the whole stream.  The
 * parts run alone tell which of them the whole costs more than. */
template <int PARTS>
__global__ __launch_bounds__(1024) void term_probe_packed(Stamp *stamps, float *sink, int iters)
{
	float r = 3.25f + 0.01f * (float)(threadIdx.x & 63);
	f32x2 tz01 = {1.5f, 2.5f}, tz23 = {3.5f, 4.5f}, mag2 = {0.f, 0.f}, acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f};
	f32x2 g01 = {0.25f, -0.25f}, g23 = {0.125f, -0.125f}, svs[4];
	f32x4 tap[4], cs01 = {0.6f, 0.8f, 0.8f, 0.6f}, cs23 = cs01;
	for (int k = 0; k < 4; k++) { tap[k] = f32x4{0.5f + k, 0.25f, 0.125f, -0.5f}; svs[k] = f32x2{0.5f + k, 0.25f}; }
	uint32_t m_bits = 0x4B000001u;
	asm volatile("" : "+s"(m_bits));
	__syncthreads();
	uint64_t t0 = memtime(), r0 = memrealtime();
	for (int i = 0; i < iters; i++) {
		asm volatile("" : "+v"(tz01), "+v"(tz23), "+v"(cs01), "+v"(cs23), "+v"(tap[0]), "+v"(tap[1]), "+v"(tap[2]), "+v"(tap[3]));
		if constexpr (PARTS & 1) {
			const float M = __builtin_bit_cast(float, m_bits);
			const f32x2 M2 = {M, M}, rr = {r, r};
			const f32x2 p01 = rr + tz01, p23 = rr + tz23;
			const f32x2 y01 = p01 + M2,  y23 = p23 + M2;
			g01 = p01; g23 = p23;                        /* the interpolation uses the position itself (line-form elements) */
			const float ys[4] = {y01.x, y01.y, y23.x, y23.y};
			#pragma unroll
			for (int k = 0; k < 4; k++) {
				uint32_t at;
				asm("v_lshlrev_b16 %0, 4, %1" : "=v"(at) : "v"(__builtin_bit_cast(uint32_t, ys[k])));
				asm volatile("" :: "v"(at));
			}
			if constexpr (!(PARTS & 2)) asm volatile("" :: "v"(g01), "v"(g23));
		} else {
			asm volatile("" : "+v"(g01), "+v"(g23));
		}
		const float gs[4] = {g01.x, g01.y, g23.x, g23.y};
		const f32x2 cs[4] = {{cs01.x, cs01.y}, {cs01.z, cs01.w}, {cs23.x, cs23.y}, {cs23.z, cs23.w}};
		float q[4];
		#pragma unroll
		for (int k = 0; k < 4; k++) {
			f32x2 sv;
			if constexpr (PARTS & 2) {
				sv = f32x2{tap[k].x, tap[k].y} + gs[k] * f32x2{tap[k].z, tap[k].w};
				acc1 += sv.x * cs[k];
				acc2 += sv.y * cs[k];
				if constexpr (!(PARTS & 4)) asm volatile("" :: "v"(sv));
			} else {
				asm volatile("" : "+v"(svs[k]));
				sv = svs[k];
			}
			if constexpr (PARTS & 4) q[k] = __builtin_amdgcn_sqrtf(__builtin_fmaf(sv.y, sv.y, sv.x * sv.x));
		}
		if constexpr (PARTS & 4) { mag2 += f32x2{q[0], q[1]}; mag2 += f32x2{q[2], q[3]}; }
		m_bits += 128;
		if (m_bits > 0x4B000801u) m_bits = 0x4B000001u;
	}
	uint64_t t1 = memtime(), r1 = memrealtime();
	if (mag2.x + mag2.y + acc1.x + acc1.y + acc2.x + acc2.y == 12345.678f) sink[0] = mag2.x;
	if ((threadIdx.x & 63) == 0) {
		uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
		stamps[wave] = Stamp{t1 - t0, r1 - r0};
	}
}

/* ------------------------------------------------------------------ the VALU stream of das_hercules.hip's inner loop
 * What the aligned-grid HERCULES kernel issues per batch of four (channel, transmit-element) pairs in its unchecked IQ loop with
 * coherency weighting and the per-lane phase reduction (config 5's instantiation) once the gathered samples are in registers:
 * (od2 + z2) + D2 and w = ws D2 + ws od2 as packed ops over two pairs, v_sqrt, the packed index fma, the degree-5 apodization
 * polynomial (packed), v_fract + v_cvt_flr of the index, the tap offset, the interpolation (one packed fma of the prepared
 * {sample, difference} pair; CUBIC: the three-step Horner chain of the prepared segment polynomial), turns fma, v_sin, v_cos, the
 * phasor scaled by the apodization, two packed fmas of rotate-accumulate and mul / fma / v_sqrt / fma for |s|.  Written in the
 * kernel's own C++ (the expressions of das_hercules.hip's `group` lambda, CHECK = false, B = 4) with the table entries and the
 * gathered data made opaque every iteration; no memory instruction.  Its rate is the ceiling of that formulation. */
__device__ __forceinline__ f32x2 mb_splat(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 mb_apod_poly(f32x2 w)
{
	f32x2 r = mb_splat(-0.000112471265f);
	r = r * w + mb_splat(0.0030977894f);
	r = r * w + mb_splat(-0.044347722f);
	r = r * w + mb_splat(0.33327785f);
	r = r * w + mb_splat(-0.9999883f);
	r = r * w + mb_splat(0.9999996f);
	return r;
}
/* MODE 0: linear, prepared {sample, difference}; 1: cubic, prepared segment polynomial; 2: cubic out of the four RAW taps (coarse grids: the reference
 * harness's view plane) -- Catmull-Rom as the kernel's four tap weights.  CW: with the |s| sum of coherency weighting. */
template <int MODE, bool CW>
__global__ __launch_bounds__(256) void hercules_probe(Stamp *stamps, float *sink, int iters)
{
	const float lane = (float)(threadIdx.x & 63);
	float d2[4] = {1.0e-6f, 2.0e-6f, 3.0e-6f, 4.0e-6f};
	f32x4 lo[4], hi[4];
	for (int k = 0; k < 4; k++) { lo[k] = f32x4{0.5f + k, 0.25f, 0.125f, -0.5f}; hi[k] = f32x4{0.3f, 0.1f + k, -0.2f, 0.05f}; }
	const f32x2 oz2p = mb_splat(4.0e-4f + 1.0e-7f * lane), wodp = mb_splat(0.01f), wsp = mb_splat(1.0e3f), T0p = mb_splat(100.f + lane),
	            kp = mb_splat(8117.f), tpsp = mb_splat(0.5f), btp = mb_splat(130.f);
	f32x2 acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f};
	float mag = 0.f;
	uint32_t row = 4096;
	asm volatile("" : "+s"(row));
	__syncthreads();
	uint64_t t0 = memtime(), r0 = memrealtime();
	for (int i = 0; i < iters; i++) {
		asm volatile("" : "+s"(d2[0]), "+s"(d2[1]), "+s"(d2[2]), "+s"(d2[3]));
		asm volatile("" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]));
		if constexpr (MODE != 0) asm volatile("" : "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
		f32x2 index[2], apod[2], turns[2];
		#pragma unroll
		for (int k = 0; k < 2; k++) {
			const f32x2 dn = f32x2{d2[2 * k], d2[2 * k + 1]};
			const f32x2 dd = oz2p + dn;
			apod[k] = mb_apod_poly(dn * wsp + wodp);
			const f32x2 dist = f32x2{__builtin_amdgcn_sqrtf(dd.x), __builtin_amdgcn_sqrtf(dd.y)};
			index[k] = dist * kp + T0p;
			turns[k] = index[k] * tpsp - btp;
		}
		#pragma unroll
		for (int k = 0; k < 4; k++) {
			const float idx = (k & 1) ? index[k >> 1].y : index[k >> 1].x;
			const float ap  = (k & 1) ? apod[k >> 1].y : apod[k >> 1].x;
			const float frac = __builtin_amdgcn_fractf(idx);
			int ki; asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ki) : "v"(idx));
			const uint32_t off = MODE == 2 ? ((uint32_t)ki - 1u) * 8u : (uint32_t)ki * (MODE == 1 ? 32u : 16u);
			asm volatile("" :: "v"(off));
			f32x2 sv;
			if constexpr (MODE == 1) {
				sv = f32x2{hi[k].z, hi[k].w} * frac + f32x2{hi[k].x, hi[k].y};
				sv = sv * frac + f32x2{lo[k].z, lo[k].w};
				sv = sv * frac + f32x2{lo[k].x, lo[k].y};
			} else if constexpr (MODE == 2) {
				const f32x2 s0 = {lo[k].x, lo[k].y}, s1 = {lo[k].z, lo[k].w}, s2 = {hi[k].x, hi[k].y}, s3 = {hi[k].z, hi[k].w};
				const float u = 1.0f - frac, tu = frac * u, a_ = -0.5f * tu;          /* bf_catmull_rom (das_common.h) */
				const float w0 = a_ * u, w3 = a_ * frac;
				const float w1 = __builtin_fmaf(tu, __builtin_fmaf(-1.5f, frac, 1.0f), u);
				const float w2 = __builtin_fmaf(tu, __builtin_fmaf(1.5f, frac, -0.5f), frac);
				sv = w0 * s0 + w1 * s1 + w2 * s2 + w3 * s3;
			} else {
				sv = f32x2{lo[k].x, lo[k].y} + frac * f32x2{lo[k].z, lo[k].w};
			}
			const float tr = (k & 1) ? turns[k >> 1].y : turns[k >> 1].x;
			const f32x2 cs = f32x2{__builtin_amdgcn_cosf(tr), __builtin_amdgcn_sinf(tr)} * ap;
			acc1 += sv.x * cs;
			acc2 += sv.y * cs;
			if constexpr (CW) mag = __builtin_fmaf(ap, __builtin_amdgcn_sqrtf(__builtin_fmaf(sv.y, sv.y, sv.x * sv.x)), mag);
		}
	}
	uint64_t t1 = memtime(), r1 = memrealtime();
	if (mag + acc1.x + acc1.y + acc2.x + acc2.y == 12345.678f) sink[0] = mag;
	if ((threadIdx.x & 63) == 0) {
		uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
		stamps[wave] = Stamp{t1 - t0, r1 - r0};
	}
}

/* ------------------------------------------------------------------ the whole inner loop of das_staged.hip: VALU stream AND its LDS reads
 * Config 4's shape (76 padded transmits, 32-element windows, a 32 x 32 tile, 16 waves per block, two blocks per CU): per batch
 * of 4 terms one ds_read2_b64 (delays), two ds_read_b128 (phasors) and four ds_read_b128 taps at addresses formed from the
 * positions, as the kernel forms them -- everything but the per-channel staging and its barriers.
 * MODE bit 0: address by v_lshlrev_b16 instead of v_mul_u32_u24; bit 1: no LDS reads at all (operands stay in registers);
 * bit 2: taps by ds_read_b64 (half the returned bytes: how sensitive the loop is to LDS return traffic);
 * bit 3: LDS reads and address arithmetic only (no interpolation / accumulate / |s|);
 * bit 4: the delays of the NEXT batch are read a batch ahead (4 more registers); bit 5: the phasors too (8 more). */
typedef __attribute__((address_space(3))) f32x2 mb_lds_f32x2;
typedef __attribute__((address_space(3))) f32x4 mb_lds_f32x4;
template <int MODE>
__global__ __launch_bounds__(1024, 8) void loop_probe(Stamp *stamps, float *sink, int iters)
{
	extern __shared__ __attribute__((aligned(16))) f32x4 probe_lds[];
	constexpr uint32_t A4 = 76, W = 32, V = 32;
	f32x4 *stage = probe_lds + 2;
	f32x4 *Tcs = stage + A4 * W + 1;
	f32x2 *Tz = reinterpret_cast<f32x2 *>(Tcs + (A4 / 2) * V);
	for (uint32_t i = threadIdx.x; i < A4 * W + 3; i += blockDim.x) probe_lds[i] = f32x4{0.5f + 0.001f * i, 0.25f, 0.125f, -0.5f};
	for (uint32_t i = threadIdx.x; i < (A4 / 2) * V; i += blockDim.x) {
		uint32_t v = i % V, a2 = i / V;
		Tcs[i] = f32x4{0.6f, 0.8f, 0.8f, 0.6f};
		Tz[i] = f32x2{1.f + 0.3f * v + 0.7f * (float)((2 * a2) % 5), 1.f + 0.3f * v + 0.7f * (float)((2 * a2 + 1) % 5)};
	}
	__syncthreads();
	const uint32_t lane_v = threadIdx.x >> 5, lane_u = threadIdx.x & 31u;
	const float r_rel = 0.5f * (float)lane_u;
	const uint32_t tcs_base = (uint32_t)(uintptr_t)(mb_lds_f32x4 *)Tcs, tz_base = (uint32_t)(uintptr_t)(mb_lds_f32x2 *)Tz;
	f32x2 acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f}, mag2 = {0.f, 0.f};
	uint64_t t0 = memtime(), r0 = memrealtime();
	for (int i = 0; i < iters; i++) {
		uint32_t tcs_at = tcs_base + (lane_v << 4), tz_at = tz_base + (lane_v << 3);
		uint32_t m_bits = 0x4B000002u;
		const f32x2 rr = {r_rel, r_rel};
		f32x4 cs01 = {0.6f, 0.8f, 0.8f, 0.6f}, cs23 = cs01, tap[4];
		f32x2 tz01 = {1.5f, 2.5f}, tz23 = {3.5f, 4.5f};
		for (int k = 0; k < 4; k++) tap[k] = f32x4{0.5f + k, 0.25f, 0.125f, -0.5f};
		f32x2 ntz01 = tz01, ntz23 = tz23; f32x4 ncs01 = cs01, ncs23 = cs23;
		if constexpr (MODE & 16) { ntz01 = *(mb_lds_f32x2 *)(uintptr_t)tz_at; ntz23 = *(mb_lds_f32x2 *)(uintptr_t)(tz_at + V * 8u); }
		if constexpr (MODE & 32) { ncs01 = *(mb_lds_f32x4 *)(uintptr_t)tcs_at; ncs23 = *(mb_lds_f32x4 *)(uintptr_t)(tcs_at + V * 16u); }
		for (uint32_t a = 0; a < A4; a += 4, tcs_at += 2u * V * 16u, tz_at += 2u * V * 8u, m_bits += 4u * W) {
			const float M = __builtin_bit_cast(float, m_bits);
			const f32x2 M2 = {M, M};
			if constexpr (!(MODE & 2)) {
				if constexpr (MODE & 32) {
					cs01 = ncs01; cs23 = ncs23;              /* (the table is padded by a batch, as the kernel's is by the zero rows) */
					ncs01 = *(mb_lds_f32x4 *)(uintptr_t)(tcs_at + 2u * V * 16u);
					ncs23 = *(mb_lds_f32x4 *)(uintptr_t)(tcs_at + 3u * V * 16u);
				} else {
					cs01 = *(mb_lds_f32x4 *)(uintptr_t)tcs_at;
					cs23 = *(mb_lds_f32x4 *)(uintptr_t)(tcs_at + V * 16u);
				}
				if constexpr (MODE & 16) {
					tz01 = ntz01; tz23 = ntz23;
					ntz01 = *(mb_lds_f32x2 *)(uintptr_t)(tz_at + 2u * V * 8u);
					ntz23 = *(mb_lds_f32x2 *)(uintptr_t)(tz_at + 3u * V * 8u);
				} else {
					tz01 = *(mb_lds_f32x2 *)(uintptr_t)tz_at;
					tz23 = *(mb_lds_f32x2 *)(uintptr_t)(tz_at + V * 8u);
				}
			} else {
				asm volatile("" : "+v"(tz01), "+v"(tz23), "+v"(cs01), "+v"(cs23), "+v"(tap[0]), "+v"(tap[1]), "+v"(tap[2]), "+v"(tap[3]));
			}
			const f32x2 p01 = rr + tz01, p23 = rr + tz23;
			const f32x2 y01 = p01 + M2,  y23 = p23 + M2;
			const float ys[4] = {y01.x, y01.y, y23.x, y23.y}, ps[4] = {p01.x, p01.y, p23.x, p23.y};
			uint32_t at[4];
			#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t yb = __builtin_bit_cast(uint32_t, ys[k]);
				if constexpr (MODE & 1) asm("v_lshlrev_b16 %0, 4, %1" : "=v"(at[k]) : "v"(yb));
				else                    asm("v_mul_u32_u24 %0, 16, %1" : "=v"(at[k]) : "v"(yb));
				if constexpr (MODE & 2) asm volatile("" :: "v"(at[k]));
			}
			if constexpr (!(MODE & 2)) {
				#pragma unroll
				for (int k = 0; k < 4; k++) {
					if constexpr (MODE & 4) { f32x2 h = *(mb_lds_f32x2 *)(uintptr_t)(at[k] + (uint32_t)k * W * 16u); tap[k].x = h.x; tap[k].y = h.y; }
					else tap[k] = *(mb_lds_f32x4 *)(uintptr_t)(at[k] + (uint32_t)k * W * 16u);
				}
			}
			if constexpr (MODE & 8) {
				asm volatile("" :: "v"(tap[0]), "v"(tap[1]), "v"(tap[2]), "v"(tap[3]), "v"(cs01), "v"(cs23));
			} else {
				const f32x2 cs[4] = {{cs01.x, cs01.y}, {cs01.z, cs01.w}, {cs23.x, cs23.y}, {cs23.z, cs23.w}};
				float q[4];
				#pragma unroll
				for (int k = 0; k < 4; k++) {
					f32x2 sv = f32x2{tap[k].x, tap[k].y} + ps[k] * f32x2{tap[k].z, tap[k].w};
					acc1 += sv.x * cs[k];
					acc2 += sv.y * cs[k];
					q[k] = __builtin_amdgcn_sqrtf(__builtin_fmaf(sv.y, sv.y, sv.x * sv.x));
				}
				mag2 += f32x2{q[0], q[1]}; mag2 += f32x2{q[2], q[3]};
			}
		}
	}
	uint64_t t1 = memtime(), r1 = memrealtime();
	if (mag2.x + mag2.y + acc1.x + acc1.y + acc2.x + acc2.y == 12345.678f) sink[0] = mag2.x;
	if ((threadIdx.x & 63) == 0) {
		uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
		stamps[wave] = Stamp{t1 - t0, r1 - r0};
	}
}

/* The same loop when a wave's 64 voxels share one lateral row of the transmit axis (a 64 x 16 tile): delays and phasors are wave
 * uniform, come from a table in global memory through SCALAR loads (3 x s_load_dwordx4 per batch) and enter the packed ops as
 * scalar operands; the LDS serves the four taps only. */
__global__ __launch_bounds__(1024, 8) void loop_probe_uniform(const f32x4 *table, Stamp *stamps, float *sink, int iters)
{
	extern __shared__ __attribute__((aligned(16))) f32x4 probe_lds[];
	constexpr uint32_t A4 = 76, W = 32;
	for (uint32_t i = threadIdx.x; i < A4 * W + 3; i += blockDim.x) probe_lds[i] = f32x4{0.5f + 0.001f * i, 0.25f, 0.125f, -0.5f};
	__syncthreads();
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const float r_rel = 0.4f * (float)(threadIdx.x & 63u);
	typedef __attribute__((address_space(4))) const f32x4 const_f32x4;        /* constant address space: uniform reads become s_load */
	const_f32x4 *row = (const_f32x4 *)(uintptr_t)(table + (size_t)wave * (A4 / 4) * 3);   /* per wave and batch: {tz0..3}, {cs0, cs1}, {cs2, cs3} */
	f32x2 acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f}, mag2 = {0.f, 0.f};
	uint64_t t0 = memtime(), r0 = memrealtime();
	for (int i = 0; i < iters; i++) {
		uint32_t m_bits = 0x4B000002u;
		const f32x2 rr = {r_rel, r_rel};
		const_f32x4 *at_row = row;
		for (uint32_t a = 0; a < A4; a += 4, at_row += 3, m_bits += 4u * W) {
			const float M = __builtin_bit_cast(float, m_bits);
			const f32x2 M2 = {M, M};
			const f32x4 tz = at_row[0], cs01 = at_row[1], cs23 = at_row[2];
			const f32x2 p01 = rr + f32x2{tz.x, tz.y}, p23 = rr + f32x2{tz.z, tz.w};
			const f32x2 y01 = p01 + M2,  y23 = p23 + M2;
			const float ys[4] = {y01.x, y01.y, y23.x, y23.y}, ps[4] = {p01.x, p01.y, p23.x, p23.y};
			uint32_t at[4]; f32x4 tap[4];
			#pragma unroll
			for (int k = 0; k < 4; k++) asm("v_lshlrev_b16 %0, 4, %1" : "=v"(at[k]) : "v"(__builtin_bit_cast(uint32_t, ys[k])));
			#pragma unroll
			for (int k = 0; k < 4; k++) tap[k] = *(mb_lds_f32x4 *)(uintptr_t)(at[k] + (uint32_t)k * W * 16u);
			const f32x2 cs[4] = {{cs01.x, cs01.y}, {cs01.z, cs01.w}, {cs23.x, cs23.y}, {cs23.z, cs23.w}};
			float q[4];
			#pragma unroll
			for (int k = 0; k < 4; k++) {
				f32x2 sv = f32x2{tap[k].x, tap[k].y} + ps[k] * f32x2{tap[k].z, tap[k].w};
				acc1 += sv.x * cs[k];
				acc2 += sv.y * cs[k];
				q[k] = __builtin_amdgcn_sqrtf(__builtin_fmaf(sv.y, sv.y, sv.x * sv.x));
			}
			mag2 += f32x2{q[0], q[1]}; mag2 += f32x2{q[2], q[3]};
		}
	}
	uint64_t t1 = memtime(), r1 = memrealtime();
	if (mag2.x + mag2.y + acc1.x + acc1.y + acc2.x + acc2.y == 12345.678f) sink[0] = mag2.x;
	if ((threadIdx.x & 63) == 0) {
		uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
		stamps[w] = Stamp{t1 - t0, r1 - r0};
	}
}

/* ------------------------------------------------------------------ per-lane gathers from global memory */
enum { PAT_COALESCED, PAT_DAS, PAT_RANDOM, PAT_COUNT };
static const char *pat_name[PAT_COUNT] = {"contiguous", "das_like", "random"};

/* byte offset of lane `lane`'s access number `n` inside a window of `mask + 1` bytes, aligned to `align` */
template <int PAT>
__device__ __forceinline__ uint32_t pattern(uint32_t lane, uint32_t n, uint32_t width, uint32_t mask, uint32_t align)
{
	uint32_t off;
	if constexpr (PAT == PAT_COALESCED) off = lane * width + n * 64u * width;
	else if constexpr (PAT == PAT_DAS)  off = ((lane * 5u) >> 3) * 8u + n * 1000u;      /* 0..8 B between neighbours, 8-B aligned */
	else { uint32_t h = (lane * 2654435761u) ^ (n * 40503u + 0x9E3779B9u); h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; off = h; }
	return (off & mask) & ~(align - 1u);
}

template <int WIDTH, int PAT>
__global__ __launch_bounds__(1024) void gather_probe(const char *base, uint32_t window_bytes, uint32_t per_block_window,
                                                     Stamp *stamps, float *sink, int iters)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;
	const char *win = base + (per_block_window ? (size_t)blockIdx.x * window_bytes : 0);
	const uint32_t mask = window_bytes / 2u - 1u;       /* offsets fall in the first half of the (power-of-two) window: the footprint */
	const uint32_t align = PAT == PAT_COALESCED ? (uint32_t)WIDTH : (WIDTH >= 8 ? 8u : 4u);
	/* warm the window into the cache */
	float warm = 0.f;
	for (uint32_t o = threadIdx.x * 16u; o < window_bytes; o += blockDim.x * 16u) warm += *(const float *)(win + o);
	__syncthreads();
	f32x4 acc = {warm, 0.f, 0.f, 0.f};
	uint64_t t0 = memtime(), r0 = memrealtime();
	for (int i = 0; i < iters; i++) {
		uint32_t off[4];
		#pragma unroll
		for (int k = 0; k < 4; k++) off[k] = pattern<PAT>(lane + wave * 7u, (uint32_t)(i * 4 + k), WIDTH, mask, align);
		if constexpr (WIDTH == 16) {
			f32x4 d0, d1, d2, d3;
			asm volatile("global_load_dwordx4 %0, %4, %8\n\tglobal_load_dwordx4 %1, %5, %8\n\t"
			             "global_load_dwordx4 %2, %6, %8\n\tglobal_load_dwordx4 %3, %7, %8\n\ts_waitcnt vmcnt(0)"
			             : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
			             : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(win) : "memory");
			acc += d0 + d1 + d2 + d3;
		} else if constexpr (WIDTH == 8) {
			f32x2 d0, d1, d2, d3;
			asm volatile("global_load_dwordx2 %0, %4, %8\n\tglobal_load_dwordx2 %1, %5, %8\n\t"
			             "global_load_dwordx2 %2, %6, %8\n\tglobal_load_dwordx2 %3, %7, %8\n\ts_waitcnt vmcnt(0)"
			             : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
			             : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(win) : "memory");
			acc.x += d0.x + d1.x + d2.x + d3.x; acc.y += d0.y + d1.y + d2.y + d3.y;
		} else {
			float d0, d1, d2, d3;
			asm volatile("global_load_dword %0, %4, %8\n\tglobal_load_dword %1, %5, %8\n\t"
			             "global_load_dword %2, %6, %8\n\tglobal_load_dword %3, %7, %8\n\ts_waitcnt vmcnt(0)"
			             : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
			             : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(win) : "memory");
			acc.x += d0 + d1 + d2 + d3;
		}
	}
	uint64_t t1 = memtime(), r1 = memrealtime();
	if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
	if (lane == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{t1 - t0, r1 - r0};
}

/* ------------------------------------------------------------------ LDS reads */
enum { LDS_B64, LDS_B128, LDS_READ2_B64, LDS_B32, LDS_BPERMUTE, LDS_COUNT };
static const char *lds_name[LDS_COUNT] = {"ds_read_b64", "ds_read_b128", "ds_read2_b64(16B@8)", "ds_read_b32",
                                          "ds_bpermute_b32 (lane crossbar, no LDS memory)"};

template <int KIND, int PAT>
__global__ __launch_bounds__(1024) void lds_probe(Stamp *stamps, float *sink, int iters, uint32_t window_bytes)
{
	extern __shared__ __attribute__((aligned(16))) char lds[];
	for (uint32_t o = threadIdx.x * 4u; o < window_bytes; o += blockDim.x * 4u) *(float *)(lds + o) = (float)o;
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t mask = window_bytes / 2u - 1u;
	constexpr uint32_t WIDTH = KIND == LDS_B64 ? 8 : ((KIND == LDS_B32 || KIND == LDS_BPERMUTE) ? 4 : 16);
	const uint32_t align = KIND == LDS_B128 ? 16u : ((KIND == LDS_B32 || KIND == LDS_BPERMUTE) ? 4u : 8u);
	f32x4 acc = {0.f, 0.f, 0.f, 0.f};
	uint64_t t0 = memtime(), r0 = memrealtime();
	for (int i = 0; i < iters; i++) {
		uint32_t off[4];
		#pragma unroll
		for (int k = 0; k < 4; k++) {
			uint32_t o = pattern<PAT>(lane + wave * 7u, (uint32_t)(i * 4 + k), WIDTH, mask, align);
			if (PAT == PAT_COALESCED) o &= ~(WIDTH - 1u);
			off[k] = o;
		}
		if constexpr (KIND == LDS_B128) {
			f32x4 d0, d1, d2, d3;
			asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
			             : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]) : "memory");
			acc += d0 + d1 + d2 + d3;
		} else if constexpr (KIND == LDS_READ2_B64) {
			f32x4 d0, d1, d2, d3;
			asm volatile("ds_read2_b64 %0, %4 offset1:1\n\tds_read2_b64 %1, %5 offset1:1\n\tds_read2_b64 %2, %6 offset1:1\n\t"
			             "ds_read2_b64 %3, %7 offset1:1\n\ts_waitcnt lgkmcnt(0)"
			             : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]) : "memory");
			acc += d0 + d1 + d2 + d3;
		} else if constexpr (KIND == LDS_B64) {
			f32x2 d0, d1, d2, d3;
			asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
			             : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]) : "memory");
			acc.x += d0.x + d1.x + d2.x + d3.x; acc.y += d0.y + d1.y + d2.y + d3.y;
		} else if constexpr (KIND == LDS_BPERMUTE) {
			/* each lane pulls a dword from the lane named by its byte address / 4 (mod 64) */
			float d0, d1, d2, d3, src = acc.y + (float)lane;
			asm volatile("ds_bpermute_b32 %0, %4, %8\n\tds_bpermute_b32 %1, %5, %8\n\tds_bpermute_b32 %2, %6, %8\n\tds_bpermute_b32 %3, %7, %8\n\ts_waitcnt lgkmcnt(0)"
			             : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(src) : "memory");
			acc.x += d0 + d1 + d2 + d3;
		} else {
			float d0, d1, d2, d3;
			asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b32 %3, %7\n\ts_waitcnt lgkmcnt(0)"
			             : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]) : "memory");
			acc.x += d0 + d1 + d2 + d3;
		}
	}
	uint64_t t1 = memtime(), r1 = memrealtime();
	if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
	if (lane == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{t1 - t0, r1 - r0};
}

/* ------------------------------------------------------------------ host side */
struct Result { double cycles_per_wave, clock_ghz, wall_ms; };

static Stamp *d_stamps; static float *d_sink; static std::vector<Stamp> h_stamps;
static int n_cu = 256;

template <typename F>
static Result run(F launch, int waves_total)
{
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	launch();                                         /* warm up (code load, clocks) */
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(e0));
	launch();
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
	h_stamps.resize(waves_total);
	CHECK(hipMemcpy(h_stamps.data(), d_stamps, sizeof(Stamp) * waves_total, hipMemcpyDeviceToHost));
	std::vector<double> cyc(waves_total), clk(waves_total);
	for (int i = 0; i < waves_total; i++) {
		cyc[i] = (double)h_stamps[i].cycles;
		clk[i] = h_stamps[i].realtime ? (double)h_stamps[i].cycles / ((double)h_stamps[i].realtime * 10e-9) * 1e-9 : 0.0;
	}
	std::nth_element(cyc.begin(), cyc.begin() + waves_total / 2, cyc.end());
	std::nth_element(clk.begin(), clk.begin() + waves_total / 2, clk.end());
	CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
	return Result{cyc[waves_total / 2], clk[waves_total / 2], ms};
}

static std::string json;
static void emit(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
#include <cstdarg>
static void emit(const char *fmt, ...) { char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap); json += buf; }

template <int OP> static void valu_case(bool first)
{
	const int iters = 16000;                           /* 512k instructions per wave */
	for (int wps : {1, 2, 4, 8}) {
		int waves_per_block = wps <= 4 ? 4 * wps : 16, blocks_per_cu = wps <= 4 ? 1 : 2;
		int blocks = n_cu * blocks_per_cu, waves = blocks * waves_per_block;
		Result r = run([&] { hipLaunchKernelGGL((valu_probe<OP>), dim3(blocks), dim3(64 * waves_per_block), 0, 0, d_stamps, d_sink, iters); }, waves);
		double inst = 32.0 * iters;
		double per_simd = wps * inst / r.cycles_per_wave;          /* wave-instructions per clock per SIMD, from the waves' own stamps */
		/* the same from the launch's wall time (HIP events) at the measured clock: includes launch ramp and
		 * waves that do not all overlap, so it is the conservative figure (the one DESIGN.md quotes) */
		double wall_cycles = r.wall_ms * 1e-3 * r.clock_ghz * 1e9;
		double cyc_wall = wall_cycles / (inst * wps);
		emit("%s{\"op\":\"%s\",\"waves_per_simd\":%d,\"cycles_per_wave_inst_per_simd_wall\":%.3f,\"cycles_per_wave_inst_per_simd_stamps\":%.3f,"
		     "\"lanes_per_clk_per_cu_wall\":%.1f,\"clock_ghz\":%.3f,\"wall_ms\":%.3f}", (first && wps == 1) ? "" : ",\n  ",
		     op_name[OP], wps, cyc_wall, 1.0 / per_simd, 64.0 * 4 / cyc_wall, r.clock_ghz, r.wall_ms);
	}
}

static void term_case()
{
	const int iters = 40000;                           /* 160k terms per wave */
	bool first = true;
	for (int wps : {4, 8}) {
		int waves_per_block = wps <= 4 ? 4 * wps : 16, blocks_per_cu = wps <= 4 ? 1 : 2;
		int blocks = n_cu * blocks_per_cu, waves = blocks * waves_per_block;
		Result r = run([&] { hipLaunchKernelGGL(term_probe, dim3(blocks), dim3(64 * waves_per_block), 0, 0, d_stamps, d_sink, iters); }, waves);
		double terms = 4.0 * iters;
		double wall_cycles = r.wall_ms * 1e-3 * r.clock_ghz * 1e9;
		emit("%s{\"stream\":\"das_staged term: v_add_f32, v_fract_f32, v_cvt_flr_i32_f32, v_lshl_add_u32, 3 x v_pk_fma_f32, v_mul_f32, v_fmac_f32, "
		     "v_sqrt_f32, v_add_f32 (+ v_add_u32 of the loop counter per 4 terms)\",\"waves_per_simd\":%d,\"cycles_per_term_per_simd_wall\":%.3f,"
		     "\"cycles_per_term_per_simd_stamps\":%.3f,\"clock_ghz\":%.3f,\"wall_ms\":%.3f}", first ? "" : ",\n  ",
		     wps, wall_cycles / (terms * wps), r.cycles_per_wave / (terms * wps), r.clock_ghz, r.wall_ms);
		first = false;
	}
}

template <int PARTS> static void term_packed_case(const char *what)
{
	const int iters = 40000;                           /* 160k terms per wave */
	for (int wps : {4, 8}) {
		int waves_per_block = wps <= 4 ? 4 * wps : 16, blocks_per_cu = wps <= 4 ? 1 : 2;
		int blocks = n_cu * blocks_per_cu, waves = blocks * waves_per_block;
		Result r = run([&] { hipLaunchKernelGGL(term_probe_packed<PARTS>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, d_stamps, d_sink, iters); }, waves);
		double terms = 4.0 * iters;
		double wall_cycles = r.wall_ms * 1e-3 * r.clock_ghz * 1e9;
		emit(",\n  {\"stream\":\"%s\",\"waves_per_simd\":%d,"
		     "\"cycles_per_term_per_simd_wall\":%.3f,\"cycles_per_term_per_simd_stamps\":%.3f,\"clock_ghz\":%.3f,\"wall_ms\":%.3f}",
		     what, wps, wall_cycles / (terms * wps), r.cycles_per_wave / (terms * wps), r.clock_ghz, r.wall_ms);
	}
}

template <int MODE, bool CW> static void hercules_case(const char *what, bool first)
{
	const int iters = 20000;                           /* 80k pairs per wave */
	for (int wps : {4, 7}) {                            /* the kernel holds 7 waves per SIMD (64-66 VGPRs) */
		int waves_per_block = 4, blocks_per_cu = wps;
		int blocks = n_cu * blocks_per_cu, waves = blocks * waves_per_block;
		Result r = run([&] { hipLaunchKernelGGL((hercules_probe<MODE, CW>), dim3(blocks), dim3(256), 0, 0, d_stamps, d_sink, iters); }, waves);
		double pairs = 4.0 * iters;
		double wall_cycles = r.wall_ms * 1e-3 * r.clock_ghz * 1e9;
		emit("%s{\"stream\":\"%s\",\"interpolation\":\"%s\",\"taps\":\"%s\",\"coherency_weighting\":%s,\"waves_per_simd\":%d,\"cycles_per_pair_per_simd_wall\":%.3f,\"cycles_per_pair_per_simd_stamps\":%.3f,"
		     "\"clock_ghz\":%.3f,\"wall_ms\":%.3f}", (first && wps == 4) ? "" : ",\n  ", what, MODE == 0 ? "linear" : "cubic", MODE == 2 ? "raw" : "prepared", CW ? "true" : "false", wps, wall_cycles / (pairs * wps), r.cycles_per_wave / (pairs * wps), r.clock_ghz, r.wall_ms);
	}
}

template <int MODE> static void loop_case(const char *what)
{
	const int iters = 1000;                            /* 76k terms per wave */
	const uint32_t lds = 16u * (76u * 32u + 3u) + 16u * 38u * 32u + 8u * 38u * 32u + 64u + 2048u;   /* (+ a batch of table rows: the read-ahead variants) */
	CHECK(hipFuncSetAttribute((const void *)loop_probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	int blocks = n_cu * 2, waves = blocks * 16;
	Result r = run([&] { hipLaunchKernelGGL(loop_probe<MODE>, dim3(blocks), dim3(1024), lds, 0, d_stamps, d_sink, iters); }, waves);
	double terms = 76.0 * iters;
	double wall_cycles = r.wall_ms * 1e-3 * r.clock_ghz * 1e9;
	emit(",\n  {\"stream\":\"%s\",\"waves_per_simd\":8,"
	     "\"cycles_per_term_per_simd_wall\":%.3f,\"cycles_per_term_per_simd_stamps\":%.3f,\"clock_ghz\":%.3f,\"wall_ms\":%.3f}",
	     what, wall_cycles / (terms * 8), r.cycles_per_wave / (terms * 8), r.clock_ghz, r.wall_ms);
}

static void loop_uniform_case(const char *what)
{
	const int iters = 1000;
	const uint32_t lds = 16u * (76u * 32u + 3u) + 64u;
	CHECK(hipFuncSetAttribute((const void *)loop_probe_uniform, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	std::vector<float> h(16 * 19 * 12);
	for (int w = 0; w < 16; w++) for (int b = 0; b < 19; b++) {
		float *e = &h[(size_t)(w * 19 + b) * 12];
		for (int k = 0; k < 4; k++) e[k] = 1.f + 0.3f * w + 0.7f * (float)((4 * b + k) % 5);
		for (int k = 0; k < 4; k++) { e[4 + 2 * k] = 0.6f; e[5 + 2 * k] = 0.8f; }
	}
	f32x4 *d_table;
	CHECK(hipMalloc(&d_table, h.size() * 4));
	CHECK(hipMemcpy(d_table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
	int blocks = n_cu * 2, waves = blocks * 16;
	Result r = run([&] { hipLaunchKernelGGL(loop_probe_uniform, dim3(blocks), dim3(1024), lds, 0, d_table, d_stamps, d_sink, iters); }, waves);
	double terms = 76.0 * iters;
	double wall_cycles = r.wall_ms * 1e-3 * r.clock_ghz * 1e9;
	emit(",\n  {\"stream\":\"%s\",\"waves_per_simd\":8,"
	     "\"cycles_per_term_per_simd_wall\":%.3f,\"cycles_per_term_per_simd_stamps\":%.3f,\"clock_ghz\":%.3f,\"wall_ms\":%.3f}",
	     what, wall_cycles / (terms * 8), r.cycles_per_wave / (terms * 8), r.clock_ghz, r.wall_ms);
	CHECK(hipFree(d_table));
}

static char *d_window;

template <int WIDTH, int PAT> static void gather_case(const char *level, uint32_t window, bool per_block, bool &first)
{
	const int iters = 2000;
	for (int wps : {2, 4, 8}) {
		int waves_per_block = wps <= 4 ? 4 * wps : 16, blocks_per_cu = wps <= 4 ? 1 : 2;
		int blocks = n_cu * blocks_per_cu, waves = blocks * waves_per_block;
		Result r = run([&] { hipLaunchKernelGGL((gather_probe<WIDTH, PAT>), dim3(blocks), dim3(64 * waves_per_block), 0, 0,
		                                        d_window, window, per_block ? 1u : 0u, d_stamps, d_sink, iters); }, waves);
		double bytes_per_wave = 4.0 * iters * 64 * WIDTH;
		double b_clk_cu = 4.0 * wps * bytes_per_wave / r.cycles_per_wave;            /* from the waves' own stamps */
		double wall_cycles = r.wall_ms * 1e-3 * r.clock_ghz * 1e9;
		double b_clk_cu_wall = bytes_per_wave * waves / wall_cycles / n_cu;          /* from the launch's wall time: conservative */
		emit("%s{\"inst\":\"global_load_%s\",\"level\":\"%s\",\"window_bytes\":%u,\"pattern\":\"%s\",\"waves_per_simd\":%d,"
		     "\"bytes_per_clk_per_cu_wall\":%.2f,\"bytes_per_clk_per_cu_stamps\":%.2f,\"clk_per_wave_inst_per_cu_wall\":%.2f,"
		     "\"clock_ghz\":%.3f,\"chip_TBps_wall\":%.2f,\"wall_ms\":%.3f}",
		     first ? "" : ",\n  ", WIDTH == 16 ? "dwordx4" : (WIDTH == 8 ? "dwordx2" : "dword"), level, window, pat_name[PAT], wps,
		     b_clk_cu_wall, b_clk_cu, 64.0 * WIDTH / b_clk_cu_wall, r.clock_ghz, b_clk_cu_wall * n_cu * r.clock_ghz * 1e9 / 1e12, r.wall_ms);
		first = false;
	}
}

template <int KIND, int PAT> static void lds_case(bool &first)
{
	const int iters = 4000;
	const uint32_t window = 32768;
	constexpr int WIDTH = KIND == LDS_B64 ? 8 : ((KIND == LDS_B32 || KIND == LDS_BPERMUTE) ? 4 : 16);
	for (int wps : {1, 2, 4, 8}) {
		int waves_per_block = wps <= 4 ? 4 * wps : 16, blocks_per_cu = wps <= 4 ? 1 : 2;
		int blocks = n_cu * blocks_per_cu, waves = blocks * waves_per_block;
		Result r = run([&] { hipLaunchKernelGGL((lds_probe<KIND, PAT>), dim3(blocks), dim3(64 * waves_per_block), window, 0,
		                                        d_stamps, d_sink, iters, window); }, waves);
		double bytes_per_wave = 4.0 * iters * 64 * WIDTH;
		double b_clk_cu = 4.0 * wps * bytes_per_wave / r.cycles_per_wave;
		double wall_cycles = r.wall_ms * 1e-3 * r.clock_ghz * 1e9;
		double b_clk_cu_wall = bytes_per_wave * waves / wall_cycles / n_cu;
		emit("%s{\"inst\":\"%s\",\"pattern\":\"%s\",\"waves_per_simd\":%d,\"bytes_per_clk_per_cu_wall\":%.2f,"
		     "\"bytes_per_clk_per_cu_stamps\":%.2f,\"clk_per_wave_inst_per_cu_wall\":%.2f,\"clock_ghz\":%.3f,\"wall_ms\":%.3f}",
		     first ? "" : ",\n  ", lds_name[KIND], pat_name[PAT], wps, b_clk_cu_wall, b_clk_cu, 64.0 * WIDTH / b_clk_cu_wall, r.clock_ghz, r.wall_ms);
		first = false;
	}
}

int main(int argc, char **argv)
{
	CHECK(hipSetDevice(0));
	hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
	n_cu = prop.multiProcessorCount;
	CHECK(hipMalloc(&d_stamps, sizeof(Stamp) * 16384));
	CHECK(hipMalloc(&d_sink, 64));
	CHECK(hipMalloc(&d_window, 512u << 20));
	CHECK(hipMemset(d_window, 0, 512u << 20));
	bool quick = argc > 1 && !strcmp(argv[1], "--quick");

	emit("{\"device\":\"%s\",\"arch\":\"%s\",\"compute_units\":%d,\"clock_rate_khz\":%d,\n", prop.name, prop.gcnArchName, n_cu, prop.clockRate);
	emit(" \"method\":\"loops of inline-asm instructions, every CU busy with the stated waves per SIMD; clock_ghz = in-kernel s_memtime / s_memrealtime "
	     "(100 MHz), median over waves; *_wall rates = work / (HIP-event wall time x that clock), *_stamps rates = from each wave's own s_memtime span "
	     "(optimistic when the waves of a CU do not all overlap)\",\n");

	emit(" \"valu\":[\n  ");
	valu_case<OP_FMA>(true);
	valu_case<OP_MUL>(false);
	valu_case<OP_PK_FMA>(false);
	valu_case<OP_SQRT>(false);
	valu_case<OP_SIN>(false);
	valu_case<OP_RCP>(false);
	valu_case<OP_CVT_FLR>(false);
	valu_case<OP_FRACT>(false);
	valu_case<OP_ADD>(false);
	valu_case<OP_LSHL_ADD>(false);
	valu_case<OP_PK_ADD>(false);
	valu_case<OP_PK_ADD_SGPR>(false);
	valu_case<OP_PK_ADD_NEG>(false);
	valu_case<OP_PK_FMA_SEL>(false);
	valu_case<OP_MUL_U24>(false);
	valu_case<OP_LSHL_B16>(false);
	valu_case<OP_PK_LSHL_B16>(false);
	valu_case<OP_LSHL_B32>(false);
	emit("],\n");
	{   /* what the 16-bit shifts leave in the upper half of the destination (the staged kernels' tap address relies on it) */
		uint32_t *d_bits, h_bits[2] = {0, 0};
		CHECK(hipMalloc(&d_bits, 8));
		hipLaunchKernelGGL(shift_semantics, dim3(1), dim3(64), 0, 0, d_bits);
		CHECK(hipMemcpy(h_bits, d_bits, 8, hipMemcpyDeviceToHost));
		emit(" \"shift_semantics\":{\"input\":\"0x4b000923\",\"v_lshlrev_b16 by 4\":\"0x%08x\",\"v_pk_lshlrev_b16 by {4, 15}\":\"0x%08x\"},\n", h_bits[0], h_bits[1]);
		CHECK(hipFree(d_bits));
	}
	emit(" \"valu_stream\":[\n  ");
	term_case();
	term_packed_case<7>("das_staged term, shipping form: per 4 terms 4 x v_pk_add_f32 (position, magic-number rounding), 4 x v_lshlrev_b16, "
	                    "12 x v_pk_fma_f32, 4 x (v_mul_f32, v_fmac_f32, v_sqrt_f32), 2 x v_pk_add_f32 = 36 VALU instructions");
	term_packed_case<1>("part of the shipping form alone: position / rounding / address (4 x v_pk_add_f32 + 4 x v_lshlrev_b16 per 4 terms)");
	term_packed_case<2>("part of the shipping form alone: interpolation + rotate-accumulate (12 x v_pk_fma_f32 per 4 terms)");
	term_packed_case<4>("part of the shipping form alone: |s| (4 x (v_mul_f32, v_fmac_f32, v_sqrt_f32) + 2 x v_pk_add_f32 per 4 terms)");
	term_packed_case<6>("parts of the shipping form: interpolation + rotate-accumulate + |s| (no position part)");
	loop_case<0>("das_staged inner loop with its LDS reads (config 4's shape, no staging / barriers): address by v_mul_u32_u24");
	loop_case<1>("das_staged inner loop with its LDS reads: address by v_lshlrev_b16");
	loop_case<3>("das_staged inner loop, VALU only (operands stay in registers): address by v_lshlrev_b16");
	loop_case<2>("das_staged inner loop, VALU only: address by v_mul_u32_u24");
	loop_uniform_case("das_staged inner loop, wave-uniform delays and phasors through scalar loads (a 64 x 16 tile), LDS serves the taps only");
	loop_case<17>("das_staged inner loop with its LDS reads, delays read a batch ahead (v_lshlrev_b16)");
	loop_case<49>("das_staged inner loop with its LDS reads, delays and phasors read a batch ahead (v_lshlrev_b16)");
	loop_case<5>("das_staged inner loop, taps read as ds_read_b64 (half the returned bytes; v_lshlrev_b16)");
	loop_case<9>("das_staged inner loop, LDS reads and position / address arithmetic only (v_lshlrev_b16)");
	emit("],\n");

	emit(" \"hercules_stream\":[\n  ");
	hercules_case<0, true>("das_hercules inner loop (IQ, linear interpolation of the prepared {sample, difference} pairs, coherency weighting, per-lane phase reduction): VALU only", true);
	hercules_case<1, true>("das_hercules inner loop (IQ, cubic: three-step Horner chain of the prepared segment polynomial, coherency weighting): VALU only", false);
	hercules_case<2, true>("das_hercules inner loop (IQ, cubic out of the four raw taps: Catmull-Rom tap weights per pair, coherency weighting): VALU only", false);
	hercules_case<2, false>("das_hercules inner loop (IQ, cubic out of the four raw taps, no coherency weighting: the reference harness's frame): VALU only", false);
	hercules_case<0, false>("das_hercules inner loop (IQ, linear, prepared pairs, no coherency weighting): VALU only", false);
	emit("],\n");

	bool first = true;
	emit(" \"gather\":[\n  ");
	/* L1: each block owns an 8 KB window (<= 2 blocks per CU: 16 KB of the 32 KB L1) */
	gather_case<16, PAT_COALESCED>("L1", 8192, true, first);
	gather_case<16, PAT_DAS>("L1", 8192, true, first);
	gather_case<16, PAT_RANDOM>("L1", 8192, true, first);
	gather_case<8, PAT_COALESCED>("L1", 8192, true, first);
	gather_case<8, PAT_DAS>("L1", 8192, true, first);
	gather_case<8, PAT_RANDOM>("L1", 8192, true, first);
	gather_case<4, PAT_COALESCED>("L1", 8192, true, first);
	gather_case<4, PAT_DAS>("L1", 8192, true, first);
	gather_case<4, PAT_RANDOM>("L1", 8192, true, first);
	if (!quick) {
		/* L2: all blocks share one 2 MB window (fits each XCD's 4 MB L2, not the 32 KB L1) */
		gather_case<16, PAT_COALESCED>("L2", 2u << 20, false, first);
		gather_case<16, PAT_RANDOM>("L2", 2u << 20, false, first);
		gather_case<8, PAT_RANDOM>("L2", 2u << 20, false, first);
		/* Infinity Cache: 128 MB shared window */
		gather_case<16, PAT_RANDOM>("MALL", 128u << 20, false, first);
	}
	emit("],\n");

	first = true;
	emit(" \"lds\":[\n  ");
	lds_case<LDS_B128, PAT_COALESCED>(first);
	lds_case<LDS_B128, PAT_RANDOM>(first);
	lds_case<LDS_READ2_B64, PAT_COALESCED>(first);
	lds_case<LDS_READ2_B64, PAT_DAS>(first);
	lds_case<LDS_READ2_B64, PAT_RANDOM>(first);
	lds_case<LDS_B64, PAT_COALESCED>(first);
	lds_case<LDS_B64, PAT_DAS>(first);
	lds_case<LDS_B64, PAT_RANDOM>(first);
	lds_case<LDS_B32, PAT_DAS>(first);
	lds_case<LDS_B32, PAT_RANDOM>(first);
	lds_case<LDS_BPERMUTE, PAT_DAS>(first);
	lds_case<LDS_BPERMUTE, PAT_RANDOM>(first);
	emit("]}\n");
	fputs(json.c_str(), stdout);
	return 0;
}
