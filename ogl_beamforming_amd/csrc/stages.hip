/* stages.hip -- the stages around DAS for gfx950: RF ingest (channel map + A1S2),
 * Reshape, Hadamard Decode, Filter/Demodulate, frame min/max.
 *
 * Replaces lib/ogl_beamformer_lib.c:515-559 (the client's per-channel CPU copy),
 * shaders/reshape.glsl, shaders/decode.glsl, shaders/filter.glsl of the reference and the
 * matching legs of do_compute_shader (beamformer_core.c:1312-1351, :1378-1396).
 * All kernels process every receive channel in one launch (no 16-channel chunk loop) and
 * keep the reference's inter-stage element kinds (f16 staging where the reference stages in
 * f16) so that results agree within the tolerances of tests/test_gpu_parity.py.
 * These stages are bandwidth bound (< 2 % of a frame); no MFMA.
 */
#include <hip/hip_runtime.h>
#include <type_traits>
#include "bf_kernels.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

/* ------------------------------------------------------------------ element access */

__device__ __forceinline__ f32x2 load_element(int kind, const void *buffer, int64_t index)
{
	f32x2 v = {0.f, 0.f};
	switch (kind) {
	case 0: v.x = (float)((const int16_t *)buffer)[index]; break;
	case 1: { const int16_t *p = (const int16_t *)buffer + 2 * index; v.x = (float)p[0]; v.y = (float)p[1]; } break;
	case 2: v.x = ((const float *)buffer)[index]; break;
	case 3: v = ((const f32x2 *)buffer)[index]; break;
	case 4: v.x = (float)((const _Float16 *)buffer)[index]; break;
	case 5: { const _Float16 *p = (const _Float16 *)buffer + 2 * index; v.x = (float)p[0]; v.y = (float)p[1]; } break;
	}
	return v;
}

/* GLSL constructor conversion OutputDataType(value): float->int truncates, float->half RTE */
__device__ __forceinline__ void store_scalar(int kind, void *buffer, int64_t scalar_index, float v)
{
	switch (kind >> 1) {            /* 0: int16, 1: float32, 2: float16 (kinds 0,1 / 2,3 / 4,5) */
	case 0: ((int16_t *)buffer)[scalar_index]  = (int16_t)v; break;
	case 1: ((float *)buffer)[scalar_index]    = v; break;
	case 2: ((_Float16 *)buffer)[scalar_index] = (_Float16)v; break;
	}
}

__device__ __forceinline__ void store_element(int kind, void *buffer, int64_t index, f32x2 v)
{
	if (kind & 1) { store_scalar(kind, buffer, 2 * index, v.x); store_scalar(kind, buffer, 2 * index + 1, v.y); }
	else          { store_scalar(kind, buffer, index, v.x); }
}

/* ------------------------------------------------------------------ ingest */

/* out[ch][0..out_row_bytes) = raw[channel_mapping[ch]][...]  (lib .c:519-529).
 * V = bytes moved per lane per step. */
template <typename V>
__global__ __launch_bounds__(256) void ingest_copy_kernel(const BfIngestArgs a)
{
	uint32_t channel = blockIdx.y;
	uint16_t row     = (uint16_t)a.channel_mapping[channel];
	const V *in  = (const V *)((const char *)a.raw + a.in_row_bytes * row);
	V       *out = (V *)((char *)a.out + a.out_row_bytes * channel);
	uint64_t n   = a.out_row_bytes / sizeof(V);
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
		out[i] = in[i];
}

/* A1S2 contrast: out[s] = a[s] - b[s] - c[s] on the first n scalars of the row, the rest of
 * the row is zero (lib .c:478-487, :553-556).  T arithmetic rounds per operation, as the
 * reference's typed C loops do. */
template <typename T>
__global__ __launch_bounds__(256) void ingest_a1s2_kernel(const BfIngestArgs a)
{
	uint32_t channel = blockIdx.y;
	uint16_t row     = (uint16_t)a.channel_mapping[channel];
	const T *in  = (const T *)((const char *)a.raw + a.in_row_bytes * row);
	T       *out = (T *)((char *)a.out + a.out_row_bytes * channel);
	uint64_t total = a.out_row_bytes / sizeof(T);
	uint32_t n     = a.a1s2_scalars;
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
		T v = (T)0;
		if (i < n) v = (T)((T)(in[i] - in[n + i]) - in[2 * (uint64_t)n + i]);
		out[i] = v;
	}
}

extern "C" hipError_t bf_launch_ingest(const BfIngestArgs *a, hipStream_t s)
{
	if (a->channels == 0 || a->out_row_bytes == 0) return hipSuccess;
	typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
	if (a->a1s2) {
		uint64_t n = a->out_row_bytes / (a->base == BF_BASE_F32 ? 4 : 2);
		dim3 grid((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256), a->channels);
		switch (a->base) {
		case BF_BASE_I16: hipLaunchKernelGGL(ingest_a1s2_kernel<int16_t>,  grid, dim3(256), 0, s, *a); break;
		case BF_BASE_F32: hipLaunchKernelGGL(ingest_a1s2_kernel<float>,    grid, dim3(256), 0, s, *a); break;
		default:          hipLaunchKernelGGL(ingest_a1s2_kernel<_Float16>, grid, dim3(256), 0, s, *a); break;
		}
		return hipGetLastError();
	}
	uint64_t align = a->in_row_bytes | a->out_row_bytes | (uint64_t)(uintptr_t)a->raw | (uint64_t)(uintptr_t)a->out;
	uint32_t v = (align % 16 == 0) ? 16 : (align % 4 == 0) ? 4 : 2;
	uint64_t n = a->out_row_bytes / v;
	dim3 grid((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256), a->channels);
	switch (v) {
	case 16: hipLaunchKernelGGL(ingest_copy_kernel<u32x4>,    grid, dim3(256), 0, s, *a); break;
	case 4:  hipLaunchKernelGGL(ingest_copy_kernel<uint32_t>, grid, dim3(256), 0, s, *a); break;
	default: hipLaunchKernelGGL(ingest_copy_kernel<uint16_t>, grid, dim3(256), 0, s, *a); break;
	}
	return hipGetLastError();
}

/* ------------------------------------------------------------------ reshape */

/* reshape.glsl:61-82.  One thread per element; `order` lists the axes fastest first so
 * that consecutive lanes walk the axis whose OUTPUT stride is 1.  Float16 data is IEEE
 * half (quirk Q2). */
struct ReshapeOrder { uint32_t axis[3]; };

__global__ __launch_bounds__(256) void reshape_kernel(const BfReshapeArgs a, const ReshapeOrder order)
{
	uint64_t total = (uint64_t)a.size[0] * a.size[1] * a.size[2];
	uint32_t n0 = a.size[order.axis[0]], n1 = a.size[order.axis[1]];
	for (uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (uint64_t)gridDim.x * 256) {
		uint32_t c[3];
		c[order.axis[0]] = (uint32_t)(id % n0);
		c[order.axis[1]] = (uint32_t)((id / n0) % n1);
		c[order.axis[2]] = (uint32_t)(id / ((uint64_t)n0 * n1));
		int64_t in_index  = a.in_stride[0]  * c[0] + a.in_stride[1]  * c[1] + a.in_stride[2]  * c[2];
		int64_t out_index = a.out_stride[0] * c[0] + a.out_stride[1] * c[1] + a.out_stride[2] * c[2];
		f32x2 v;
		if (a.interleave) {
			v.x = load_element(a.in_kind, a.left,  in_index).x;
			v.y = load_element(a.in_kind, a.right, in_index).x;
		} else {
			v = load_element(a.in_kind, a.left, in_index);
		}
		store_element(a.out_kind, a.out, out_index, v);
	}
}

extern "C" hipError_t bf_launch_reshape(const BfReshapeArgs *a, hipStream_t s)
{
	uint64_t total = (uint64_t)a->size[0] * a->size[1] * a->size[2];
	if (!total) return hipSuccess;
	ReshapeOrder order = {{0, 1, 2}};
	int fast = 0;
	for (int i = 0; i < 3; i++) if (a->out_stride[i] == 1 && a->size[i] > 1) fast = i;
	order.axis[0] = (uint32_t)fast;
	order.axis[1] = (uint32_t)((fast + 1) % 3);
	order.axis[2] = (uint32_t)((fast + 2) % 3);
	uint64_t blocks = (total + 255) / 256;
	if (blocks > 65536) blocks = 65536;
	hipLaunchKernelGGL(reshape_kernel, dim3((unsigned)blocks), dim3(256), 0, s, *a, order);
	return hipGetLastError();
}

/* ------------------------------------------------------------------ decode */

/* decode.glsl:24-73 / :119-150: out[ch][i][s] = (1/T) sum_j in[s][ch][j] * Ht[T*j + i].
 *
 * Block = 64 consecutive samples of one channel; the 64 x T input tile is staged in LDS
 * (rows padded by one element: lane s reads column j with stride T+1, conflict free).
 * Wave w owns the outputs i = 4(w + 4g) .. +3; the four Hadamard rows it needs come from a
 * host-transposed float copy HtT[i*T + j] through wave-uniform (scalar) loads, so the inner
 * loop is one ds_read per 4 (real) or 8 (complex) v_fma.  Accumulation happens in the
 * OUTPUT element type, as in the shader (OutputDataType result[]): f32, or f16 rounded
 * after every operation when the planner leaves the output in half precision. */
template <bool CPLX, bool ACC16>
__global__ __launch_bounds__(256) void decode_kernel(const BfDecodeArgs a)
{
	const float *__restrict__ hadamard_t = a.hadamard_t;
	extern __shared__ __attribute__((aligned(16))) float decode_lds[];
	const uint32_t T = a.transmit_count, C = a.channel_count;
	const uint32_t channel = blockIdx.y;
	const uint32_t s0      = blockIdx.x * 64;
	const uint32_t pitch   = T + 1;
	constexpr uint32_t N   = CPLX ? 2 : 1;

	for (uint32_t e = threadIdx.x; e < 64 * T; e += 256) {
		uint32_t sl = e / T, j = e - sl * T;
		uint32_t sample = s0 + sl;
		f32x2 v = {0.f, 0.f};
		if (sample < a.sample_count)
			v = load_element(a.in_kind, a.in, ((int64_t)sample * C + channel) * T + j);
		if (ACC16) { v.x = (float)(_Float16)v.x; v.y = (float)(_Float16)v.y; }   /* OutputDataType(rf[j]) */
		decode_lds[(sl * pitch + j) * N] = v.x;
		if (CPLX) decode_lds[(sl * pitch + j) * N + 1] = v.y;
	}
	__syncthreads();

	const uint32_t lane   = threadIdx.x & 63u;
	const uint32_t wave   = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t sample = s0 + lane;
	const float   *x = decode_lds + (size_t)lane * pitch * N;

	for (uint32_t i0 = wave * 4; i0 < T; i0 += 16) {
		float acc[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
		const float *h0 = hadamard_t + (size_t)(i0 + 0 < T ? i0 + 0 : 0) * T;
		const float *h1 = hadamard_t + (size_t)(i0 + 1 < T ? i0 + 1 : 0) * T;
		const float *h2 = hadamard_t + (size_t)(i0 + 2 < T ? i0 + 2 : 0) * T;
		const float *h3 = hadamard_t + (size_t)(i0 + 3 < T ? i0 + 3 : 0) * T;
		for (uint32_t j = 0; j < T; j++) {
			float xr = x[j * N], xi = CPLX ? x[j * N + 1] : 0.f;
			float h[4] = {h0[j], h1[j], h2[j], h3[j]};
			#pragma unroll
			for (int k = 0; k < 4; k++) {
				acc[k][0] += xr * h[k];
				if (CPLX) acc[k][1] += xi * h[k];
				if (ACC16) {
					acc[k][0] = (float)(_Float16)acc[k][0];
					if (CPLX) acc[k][1] = (float)(_Float16)acc[k][1];
				}
			}
		}
		if (sample < a.sample_count) {
			#pragma unroll
			for (int k = 0; k < 4; k++) {
				uint32_t i = i0 + k;
				if (i >= T) break;
				int64_t off = a.out_stride[1] * channel + a.out_stride[2] * i + a.out_stride[0] * sample;
				f32x2 v = {acc[k][0] / (float)T, acc[k][1] / (float)T};
				if (a.out_kind & 1) store_element(a.out_kind, a.out, off, v);
				else                store_scalar(a.out_kind, a.out, off, v.x);
			}
		}
	}
}

/* The same sums as decode_kernel in O(T (b + log2(T/b))) operations per sample instead of
 * O(T^2), for matrices of the form Sylvester(n) (x) B (n = T/b a power of two; the host has
 * checked the uploaded matrix entry for entry): a dense b x b transform inside every group of
 * b consecutive transmits, then a fast Walsh-Hadamard transform across the groups, both in
 * place on the LDS tile.  Every partial sum of Int16 RF is an integer below 2^24, exact in
 * f32 whatever the order, so for the reference's usual raw data the result is bit-identical to
 * the O(T^2) loop; for float RF the two differ by float rounding only.  f32 accumulation only
 * (the per-operation f16 rounding of ACC16 pipelines depends on the order: those stay on
 * decode_kernel).  Lanes are samples (row pitch T+1: conflict free); the butterflies of a
 * stage are dealt to the four waves. */
template <bool CPLX, uint32_t B>
__global__ __launch_bounds__(256) void decode_fwht_kernel(const BfDecodeArgs a)
{
	typedef typename std::conditional<CPLX, f32x2, float>::type V;
	extern __shared__ __attribute__((aligned(16))) float decode_lds[];
	V *tile = reinterpret_cast<V *>(decode_lds);
	const uint32_t T = a.transmit_count, C = a.channel_count;
	const uint32_t channel = blockIdx.y;
	const uint32_t s0      = blockIdx.x * 64;
	const uint32_t pitch   = T + 1;

	for (uint32_t e = threadIdx.x; e < 64 * T; e += 256) {
		uint32_t sl = e / T, j = e - sl * T;
		uint32_t sample = s0 + sl;
		f32x2 v = {0.f, 0.f};
		if (sample < a.sample_count)
			v = load_element(a.in_kind, a.in, ((int64_t)sample * C + channel) * T + j);
		if constexpr (CPLX) tile[sl * pitch + j] = v; else tile[sl * pitch + j] = v.x;
	}
	__syncthreads();

	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	V *x = tile + (size_t)lane * pitch;

	if constexpr (B > 1) {
		/* y[g*B + i] = sum_j Bm[i][j] x[g*B + j] for every group g; B is 12 or 20, unrolled into
		 * registers, the matrix through wave-uniform loads */
		for (uint32_t g = wave; g < T / B; g += 4) {
			V in[B];
			#pragma unroll
			for (uint32_t j = 0; j < B; j++) in[j] = x[g * B + j];
			#pragma unroll
			for (uint32_t i = 0; i < B; i++) {
				V acc = in[0] * a.hadamard_base[i * B];
				#pragma unroll
				for (uint32_t j = 1; j < B; j++) acc += in[j] * a.hadamard_base[i * B + j];
				x[g * B + i] = acc;
			}
		}
		__syncthreads();
	}
	/* Sylvester butterflies across groups: element (g, i) pairs with (g ^ h, i) */
	for (uint32_t h = 1; h < T / B; h <<= 1) {
		for (uint32_t p = wave; p < T / 2; p += 4) {
			uint32_t i  = p % B, q = p / B;                      /* q enumerates the T/(2B) group pairs */
			uint32_t g  = ((q / h) * 2 * h) + (q % h);           /* group with bit h clear */
			uint32_t lo = g * B + i, hi = (g + h) * B + i;
			V u = x[lo], v = x[hi];
			x[lo] = u + v; x[hi] = u - v;
		}
		__syncthreads();
	}

	const uint32_t sample = s0 + lane;
	if (sample < a.sample_count) {
		for (uint32_t i = wave; i < T; i += 4) {
			V acc = x[i];
			int64_t off = a.out_stride[1] * channel + a.out_stride[2] * i + a.out_stride[0] * sample;
			f32x2 v;
			if constexpr (CPLX) v = f32x2{acc.x / (float)T, acc.y / (float)T}; else v = f32x2{acc / (float)T, 0.f};
			if (a.out_kind & 1) store_element(a.out_kind, a.out, off, v);
			else                store_scalar(a.out_kind, a.out, off, v.x);
		}
	}
}

extern "C" hipError_t bf_launch_decode(const BfDecodeArgs *a, hipStream_t s)
{
	if (!a->sample_count || !a->channel_count || !a->transmit_count) return hipSuccess;
	bool cplx  = (a->in_kind & 1) != 0;
	bool acc16 = (a->out_kind >> 1) == 2;
	dim3 grid((a->sample_count + 63) / 64, a->channel_count);
	size_t lds = (size_t)64 * (a->transmit_count + 1) * (cplx ? 8 : 4);
	if (lds > 160 * 1024) return hipErrorInvalidValue;
	if (!acc16 && a->hadamard_base_order && a->hadamard_base && a->transmit_count >= 4) {
		switch (a->hadamard_base_order) {
		case 1:
			if (cplx) hipLaunchKernelGGL((decode_fwht_kernel<true,  1>),  grid, dim3(256), lds, s, *a);
			else      hipLaunchKernelGGL((decode_fwht_kernel<false, 1>),  grid, dim3(256), lds, s, *a);
			return hipGetLastError();
		case 12:
			if (cplx) hipLaunchKernelGGL((decode_fwht_kernel<true,  12>), grid, dim3(256), lds, s, *a);
			else      hipLaunchKernelGGL((decode_fwht_kernel<false, 12>), grid, dim3(256), lds, s, *a);
			return hipGetLastError();
		case 20:
			if (cplx) hipLaunchKernelGGL((decode_fwht_kernel<true,  20>), grid, dim3(256), lds, s, *a);
			else      hipLaunchKernelGGL((decode_fwht_kernel<false, 20>), grid, dim3(256), lds, s, *a);
			return hipGetLastError();
		default: break;       /* unknown base: dense kernel below */
		}
	}
	if (cplx) {
		if (acc16) hipLaunchKernelGGL((decode_kernel<true,  true>),  grid, dim3(256), lds, s, *a);
		else       hipLaunchKernelGGL((decode_kernel<true,  false>), grid, dim3(256), lds, s, *a);
	} else {
		if (acc16) hipLaunchKernelGGL((decode_kernel<false, true>),  grid, dim3(256), lds, s, *a);
		else       hipLaunchKernelGGL((decode_kernel<false, false>), grid, dim3(256), lds, s, *a);
	}
	return hipGetLastError();
}

/* ------------------------------------------------------------------ filter / demodulate */

/* filter.glsl:68-135.  A wave owns one group of 64 outputs and its own LDS window of
 * D*64 + L - 1 samples, exactly the shader's workgroup (the demodulation phase is indexed
 * by the position inside that window, filter.glsl:99-107).  Four such groups share a
 * 256-thread block.  F16 selects the shader's SAMPLE_TYPE (filter.glsl:2-14): binary16 for
 * every 16-bit input kind, binary32 otherwise; products and sums are f32. */
/* TRANSPOSE: the stage feeds Decode, whose input layout is [sample][channel][transmit] (transmit fastest,
 * beamformer_core.c:645-664): a wave's 64 outputs then lie C x A elements apart and the plain store writes 64 separate
 * 4- / 8-byte pieces (the harness's HERCULES frame: 1.9 ms of a 28 ms frame for a 0.8 GB stage).  Here the 16 waves of a
 * block take the SAME 64 samples of 16 CONSECUTIVE transmits, meet in an LDS tile [64 samples][16 transmits] and the block
 * stores it with 16 transmits contiguous per sample -- 64- / 128-byte pieces.  Same arithmetic, same values. */
constexpr uint32_t kFilterTransposeWaves = 16;
template <int IN_KIND, bool DEMOD, bool TRANSPOSE = false>
__global__ __launch_bounds__(TRANSPOSE ? 1024 : 256) void filter_kernel(const BfFilterArgs a)
{
	constexpr bool F16 = (IN_KIND >> 1) != 1;                       /* every 16-bit kind stages through binary16 */
	extern __shared__ __attribute__((aligned(16))) float filter_lds[];
	const uint32_t L = a.filter_length, D = a.decimation;
	const uint32_t window = D * 64 + L - 1;
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t wg   = TRANSPOSE ? blockIdx.x : blockIdx.x * 4 + wave;          /* the shader's gl_WorkGroupID.x */
	const uint32_t channel = blockIdx.y;
	const uint32_t transmit_wanted = TRANSPOSE ? blockIdx.z * kFilterTransposeWaves + wave : blockIdx.z;
	const uint32_t transmit = transmit_wanted < a.transmits ? transmit_wanted : a.transmits - 1;     /* (TRANSPOSE: a ragged last block computes a copy, stores nothing) */
	float *w = filter_lds + (size_t)wave * window * 2;
	constexpr bool in_complex     = (IN_KIND & 1) != 0;
	constexpr bool complex_sample = in_complex || DEMOD;             /* filter.glsl:16-19 */

	const bool offset_wraps = (D * wg * 64) < (L - 1);             /* filter.glsl:79 */
	int64_t row_start = a.in_stride[1] * channel + a.in_stride[2] * transmit;
	if (DEMOD) row_start /= 2;                                      /* filter.glsl:81-87 */
	const int64_t window_start = row_start + (int64_t)D * wg * 64 - (int64_t)(L - 1);

	float scale = a.complex_filter ? 1.0f : __builtin_sqrtf(2.0f);   /* filter.glsl:98 */
	if (F16) scale = (float)(_Float16)scale;

	for (uint32_t index = lane; index < window; index += 64) {
		f32x2 s = {0.f, 0.f};
		int64_t e = window_start + index;
		if ((!offset_wraps || index >= L - 1) && e >= 0 && e < a.in_elements) {
			s = load_element(IN_KIND, a.in, e);
			if (F16) { s.x = (float)(_Float16)s.x; s.y = (float)(_Float16)s.y; }
			if (DEMOD) {
				/* s * (1,-1); rotate_iq (filter.glsl:57-64); * scale, all in SAMPLE_TYPE */
				/* the phase depends only on the position inside the window: the host tabulates it once
				 * per plan with this very expression (libm sin/cos cost more than the whole FIR here) */
				float c, sn;
				if (a.phasors) {
					c = a.phasors[2 * index]; sn = a.phasors[2 * index + 1];
				} else {
					float arg = 6.28318530717958647692f * a.demodulation_frequency * (float)index / a.sampling_frequency;
					c = cosf(arg); sn = -sinf(arg);
				}
				f32x2 q = {s.x, -s.y};
				f32x2 r = {c * q.x - sn * q.y, sn * q.x + c * q.y};
				if (F16) {
					_Float16 rx = (_Float16)r.x, ry = (_Float16)r.y, sc = (_Float16)scale;
					s.x = (float)(_Float16)(sc * rx);
					s.y = (float)(_Float16)(sc * ry);
				} else {
					s = scale * r;
				}
			}
		}
		reinterpret_cast<f32x2 *>(w)[index] = s;
	}
	/* each wave reads back only the window it wrote: ordering inside the wave is enough, the
	 * four groups of a block never wait for one another */
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

	const uint32_t out_sample = wg * 64 + lane;
	[[maybe_unused]] f32x2 *tile = reinterpret_cast<f32x2 *>(filter_lds + (size_t)kFilterTransposeWaves * window * 2);   /* TRANSPOSE: [64][16] */
	if (out_sample < a.sample_count / D) {                            /* filter.glsl:115 */
		f32x2 result = {0.f, 0.f};
		const float *x = w + 2 * (size_t)(D * lane);
		/* (unrolled: the taps' LDS reads and coefficient loads of eight turns are issued together and waited for once -- scalar loads and LDS
		 * reads share one counter, so a turn of its own waits out both latencies 36 times per output; the sums keep their order) */
		if (a.complex_filter && complex_sample) {
			#pragma unroll 8
			for (uint32_t j = 0; j < L; j++) {
				float hr = a.coefficients[2 * j], hi = a.coefficients[2 * j + 1];
				float xr = x[2 * j], xi = x[2 * j + 1];
				result.x += hr * xr - hi * xi;
				result.y += hi * xr + hr * xi;
			}
		} else {
			const uint32_t hs = a.complex_filter ? 2 : 1;
			#pragma unroll 8
			for (uint32_t j = 0; j < L; j++) {
				float h = a.coefficients[hs * j];
				result.x += x[2 * j]     * h;
				result.y += x[2 * j + 1] * h;
			}
		}
		if constexpr (TRANSPOSE) {
			tile[lane * kFilterTransposeWaves + wave] = result;
		} else {
		int64_t off = a.out_stride[1] * channel + a.out_stride[2] * transmit + a.out_stride[0] * out_sample;
		if (a.batch_sample_count == 0 && a.out_kind == 3) {           /* the common case: f32 complex, one 8-byte store */
			reinterpret_cast<f32x2 *>(a.out)[off] = result;
		} else if (a.batch_sample_count != 0) {                       /* filter.glsl:126-130 */
			store_scalar(a.out_kind, a.out, off, result.x);
			store_scalar(a.out_kind, a.out, off + a.batch_sample_count, result.y);
		} else if (a.out_kind & 1) {
			store_element(a.out_kind, a.out, off, result);
		} else {
			store_scalar(a.out_kind, a.out, off, result.x);
		}
		}
	}
	if constexpr (TRANSPOSE) {
		__syncthreads();
		/* thread t stores sample t / 16 of transmit t % 16: 16 consecutive elements of the output per sample */
		const uint32_t ts = threadIdx.x / kFilterTransposeWaves, tt = threadIdx.x % kFilterTransposeWaves;
		const uint32_t sample = wg * 64 + ts, tx = blockIdx.z * kFilterTransposeWaves + tt;
		if (sample < a.sample_count / D && tx < a.transmits) {
			const f32x2 v = tile[ts * kFilterTransposeWaves + tt];
			const int64_t off = a.out_stride[1] * channel + a.out_stride[2] * tx + a.out_stride[0] * sample;
			if (a.out_kind == 3) reinterpret_cast<f32x2 *>(a.out)[off] = v;
			else                 store_element(a.out_kind, a.out, off, v);
		}
	}
}

extern "C" hipError_t bf_launch_filter(const BfFilterArgs *a, hipStream_t s)
{
	if (!a->sample_count || !a->channels || !a->transmits) return hipSuccess;
	uint32_t groups = (a->sample_count + 63) / 64;
	dim3 grid((groups + 3) / 4, a->channels, a->transmits);
	size_t lds = (size_t)4 * (a->decimation * 64 + a->filter_length - 1) * 2 * sizeof(float);
	if (lds > 160 * 1024) return hipErrorInvalidValue;
	/* output with the transmits contiguous and the samples far apart (Decode's input layout), complex elements, several transmits:
	 * the transposing form */
	const size_t lds_t = (size_t)kFilterTransposeWaves * (a->decimation * 64 + a->filter_length - 1) * 2 * sizeof(float) + 64 * kFilterTransposeWaves * sizeof(f32x2);
	const bool transpose = a->out_stride[2] == 1 && a->out_stride[0] >= (int64_t)a->transmits && a->transmits >= 4 && a->batch_sample_count == 0 &&
	                       (a->out_kind & 1) && lds_t <= 64 * 1024;
	if (transpose) {
		dim3 grid_t(groups, a->channels, (a->transmits + kFilterTransposeWaves - 1) / kFilterTransposeWaves);
		#define BF_FILTER_T(kind) \
			case kind: if (a->demodulate) hipLaunchKernelGGL((filter_kernel<kind, true,  true>), grid_t, dim3(1024), lds_t, s, *a); \
			           else               hipLaunchKernelGGL((filter_kernel<kind, false, true>), grid_t, dim3(1024), lds_t, s, *a); break;
		switch (a->in_kind) {
		BF_FILTER_T(0) BF_FILTER_T(1) BF_FILTER_T(2) BF_FILTER_T(3) BF_FILTER_T(4) BF_FILTER_T(5)
		default: return hipErrorInvalidValue;
		}
		#undef BF_FILTER_T
		return hipGetLastError();
	}
	#define BF_FILTER_CASE(kind) \
		case kind: if (a->demodulate) hipLaunchKernelGGL((filter_kernel<kind, true>),  grid, dim3(256), lds, s, *a); \
		           else               hipLaunchKernelGGL((filter_kernel<kind, false>), grid, dim3(256), lds, s, *a); break;
	switch (a->in_kind) {
	BF_FILTER_CASE(0) BF_FILTER_CASE(1) BF_FILTER_CASE(2) BF_FILTER_CASE(3) BF_FILTER_CASE(4) BF_FILTER_CASE(5)
	default: return hipErrorInvalidValue;
	}
	#undef BF_FILTER_CASE
	return hipGetLastError();
}

/* ------------------------------------------------------------------ Hilbert (build-defined) */

/* Analytic signal along samples: y[n] = sum_{j<L} h[j] x[n - (L-1) + j] with the complex taps of
 * host_math.cpp hilbert_fir (real part = x delayed by 31 samples, imaginary part = its Hilbert
 * transform).  The reference has no implementation to follow (include/ogl_beamformer_hip.h).
 * Same shape as the filter kernel: a wave owns 64 outputs and an LDS window of 64 + L - 1 real
 * samples; f32 products and sums; samples outside the row are zero. */
__global__ __launch_bounds__(256) void hilbert_kernel(const BfFilterArgs a)
{
	extern __shared__ __attribute__((aligned(16))) float filter_lds[];
	const uint32_t L = a.filter_length, window = 64 + L - 1;
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t wg   = blockIdx.x * 4 + wave;
	const uint32_t channel = blockIdx.y, transmit = blockIdx.z;
	float *w = filter_lds + (size_t)wave * window;
	const int64_t row = a.in_stride[1] * channel + a.in_stride[2] * transmit;
	const int64_t first = (int64_t)wg * 64 - (int64_t)(L - 1);
	for (uint32_t index = lane; index < window; index += 64) {
		int64_t s = first + index;
		float v = 0.f;
		if (s >= 0 && s < (int64_t)a.sample_count) {
			int64_t e = row + a.in_stride[0] * s;
			if (e < a.in_elements) v = load_element(a.in_kind, a.in, e).x;
		}
		w[index] = v;
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

	const uint32_t n = wg * 64 + lane;
	if (n < a.sample_count) {
		f32x2 y = {0.f, 0.f};
		for (uint32_t j = 0; j < L; j++) {
			float x = w[lane + j];
			y.x += a.coefficients[2 * j] * x;
			y.y += a.coefficients[2 * j + 1] * x;
		}
		store_element(a.out_kind, a.out, a.out_stride[1] * channel + a.out_stride[2] * transmit + a.out_stride[0] * n, y);
	}
}

extern "C" hipError_t bf_launch_hilbert(const BfFilterArgs *a, hipStream_t s)
{
	if (!a->sample_count || !a->channels || !a->transmits) return hipSuccess;
	uint32_t groups = (a->sample_count + 63) / 64;
	dim3 grid((groups + 3) / 4, a->channels, a->transmits);
	size_t lds = (size_t)4 * (64 + a->filter_length - 1) * sizeof(float);
	hipLaunchKernelGGL(hilbert_kernel, dim3(grid), dim3(256), lds, s, *a);
	return hipGetLastError();
}

/* ------------------------------------------------------------------ sum */

/* shaders/sum.glsl:7-12: out += prescale * in over every component of the frame.  The
 * reference's dispatch (beamformer_core.c:1417-1448, dead code there) clears the output and
 * applies this once per input frame, oldest first, with prescale = 1/frame_count; the
 * order of the additions is kept.  Frames are whole multiples of 64 bytes. */
__global__ __launch_bounds__(256) void sum_kernel(f32x4 *out, const f32x4 *in, float prescale, uint64_t count4)
{
	#pragma clang fp contract(off)
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count4; i += (uint64_t)gridDim.x * 256) {
		f32x4 o = out[i], v = __builtin_nontemporal_load(in + i);
		o.x = o.x + prescale * v.x; o.y = o.y + prescale * v.y;
		o.z = o.z + prescale * v.z; o.w = o.w + prescale * v.w;
		out[i] = o;
	}
}

extern "C" hipError_t bf_launch_sum(void *out, const void *in, float prescale, uint64_t bytes, hipStream_t s)
{
	uint64_t count4 = bytes / 16;
	uint64_t blocks = (count4 + 255) / 256;
	if (blocks > 256 * 16) blocks = 256 * 16;
	if (!blocks) return hipSuccess;
	hipLaunchKernelGGL(sum_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, (f32x4 *)out, (const f32x4 *)in, prescale, count4);
	return hipGetLastError();
}

/* ------------------------------------------------------------------ display reduction */

/* sample_value of shaders/render_3d.frag.glsl:50-73, the step on the far side of the path:
 * magnitude -> clamp to 10^(threshold/20) -> normalise -> gamma -> optional dB window
 * 1 - clamp(20 log10 v, -cutoff, 0) / -cutoff.  One float in [0, 1] per voxel. */
__global__ __launch_bounds__(256) void display_kernel(const float *frame, uint64_t voxels, int cplx, float threshold_value,
                                                      float gamma, float db_cutoff, float *out)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < voxels; i += (uint64_t)gridDim.x * 256) {
		float v;
		if (cplx) { f32x2 c = ((const f32x2 *)frame)[i]; v = __builtin_sqrtf(c.x * c.x + c.y * c.y); }
		else      { v = __builtin_fabsf(frame[i]); }
		v = fminf(fmaxf(v, 0.0f), threshold_value);
		v = v / threshold_value;
		v = powf(v, gamma);
		if (db_cutoff > 0) {
			v = 20.0f * logf(v) / logf(10.0f);
			v = fminf(fmaxf(v, -db_cutoff), 0.0f) / -db_cutoff;
			v = 1.0f - v;
		}
		out[i] = v;
	}
}

extern "C" hipError_t bf_launch_display(const void *frame, uint64_t voxels, int complex_data, float threshold_db,
                                        float gamma, float db_cutoff, float *out, hipStream_t s)
{
	uint64_t blocks = (voxels + 255) / 256;
	if (blocks > 256 * 16) blocks = 256 * 16;
	if (!blocks) return hipSuccess;
	hipLaunchKernelGGL(display_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, (const float *)frame, voxels, complex_data,
	                   powf(10.0f, threshold_db / 20.0f), gamma, db_cutoff, out);
	return hipGetLastError();
}

/* ------------------------------------------------------------------ min / max */

/* Build-defined reduction (the reference's shaders/min_max.glsl is never dispatched,
 * beamformer_core.c:632-637): min and max over the frame of |v| (complex) or v (real).
 * Two passes, no atomics: bitwise reproducible. */
__global__ __launch_bounds__(256) void min_max_partial_kernel(const float *frame, uint64_t voxels, int cplx, float *scratch)
{
	float lo = __builtin_inff(), hi = -__builtin_inff();
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < voxels; i += (uint64_t)gridDim.x * 256) {
		float v;
		if (cplx) { f32x2 c = ((const f32x2 *)frame)[i]; v = __builtin_sqrtf(c.x * c.x + c.y * c.y); }
		else      { v = frame[i]; }
		lo = fminf(lo, v); hi = fmaxf(hi, v);
	}
	for (int off = 32; off > 0; off >>= 1) {
		lo = fminf(lo, __shfl_xor(lo, off, 64));
		hi = fmaxf(hi, __shfl_xor(hi, off, 64));
	}
	__shared__ float slo[4], shi[4];
	if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
	__syncthreads();
	if (threadIdx.x == 0) {
		scratch[2 * blockIdx.x]     = fminf(fminf(slo[0], slo[1]), fminf(slo[2], slo[3]));
		scratch[2 * blockIdx.x + 1] = fmaxf(fmaxf(shi[0], shi[1]), fmaxf(shi[2], shi[3]));
	}
}

__global__ __launch_bounds__(64) void min_max_final_kernel(const float *scratch, uint32_t n, float *out2)
{
	float lo = __builtin_inff(), hi = -__builtin_inff();
	for (uint32_t i = threadIdx.x; i < n; i += 64) { lo = fminf(lo, scratch[2 * i]); hi = fmaxf(hi, scratch[2 * i + 1]); }
	for (int off = 32; off > 0; off >>= 1) {
		lo = fminf(lo, __shfl_xor(lo, off, 64));
		hi = fmaxf(hi, __shfl_xor(hi, off, 64));
	}
	if (threadIdx.x == 0) { out2[0] = lo; out2[1] = hi; }
}

extern "C" hipError_t bf_launch_min_max(const void *frame, uint64_t voxels, int complex_data,
                                        float *scratch, float *out2, hipStream_t s)
{
	uint32_t blocks = (uint32_t)((voxels + 255) / 256 > 1024 ? 1024 : (voxels + 255) / 256);
	if (!blocks) blocks = 1;
	hipLaunchKernelGGL(min_max_partial_kernel, dim3(blocks), dim3(256), 0, s, (const float *)frame, voxels, complex_data, scratch);
	hipLaunchKernelGGL(min_max_final_kernel, dim3(1), dim3(64), 0, s, scratch, blocks, out2);
	return hipGetLastError();
}


/* Position-weighted 64-bit checksum of a device buffer: sum over 8-byte words w[i] of w[i] * (i + 1) (mod 2^64; integer addition:
 * any order gives the same bits).  What bench.py all-reduces over the ranks of a multi-GPU run -- here for the devices of
 * beamformer_hip_set_devices: a peer that beamformed a stale, partial or misplaced RF copy is caught. */
__global__ __launch_bounds__(256) void rf_checksum_kernel(const unsigned long long *words, uint64_t count, unsigned long long *out)
{
	unsigned long long sum = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256u) sum += words[i] * (i + 1u);
	for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
	__shared__ unsigned long long part[4];
	if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = sum;
	__syncthreads();
	if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

extern "C" hipError_t bf_launch_rf_checksum(const void *data, uint64_t bytes, unsigned long long *out, hipStream_t s)
{
	hipError_t e = hipMemsetAsync(out, 0, sizeof(unsigned long long), s);
	if (e != hipSuccess) return e;
	const uint64_t words = bytes / 8u;
	if (!words) return hipSuccess;
	uint32_t blocks = (uint32_t)((words + 255u) / 256u > 2048u ? 2048u : (words + 255u) / 256u);
	hipLaunchKernelGGL(rf_checksum_kernel, dim3(blocks), dim3(256), 0, s, (const unsigned long long *)data, words, out);
	return hipGetLastError();
}
