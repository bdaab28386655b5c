/* oracle_math.c -- CPU ORACLE (test infrastructure): host DSP math.
 * Restates math.c and beamformer_core.c:366-398 of the reference.  Pinned against the
 * compiled reference by tests/test_oracle_math.py using tests/golden/host_math.npz. */
#include "oracle.h"
#include "oracle_f16.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_PI 3.14159265358979323846f   /* base_types.h:33-35: PI is a float literal */

static int is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

/* +-1 core of a normalised Hadamard matrix of order q+1 (q = 11, 19) as it appears in the
 * reference's literal tables (math.c:38-76): row 0 and column 0 are +1, the q x q core is
 * (back-)circulant in c[k] = -1 for k == 0 or k a quadratic residue mod q, +1 otherwise.
 * Order 12 rotates right with the row (c[(j-i) mod 11]), order 20 rotates left
 * (c[(j+i) mod 19]). */
static void paley_base(int q, float *out)
{
	int n = q + 1;
	int residue[32] = {0};
	for (int k = 1; k < q; k++) residue[(k * k) % q] = 1;
	for (int i = 0; i < n; i++) {
		for (int j = 0; j < n; j++) {
			float v = 1.0f;
			if (i > 0 && j > 0) {
				int k = (q == 11) ? ((j - 1) - (i - 1) + q) % q : ((j - 1) + (i - 1)) % q;
				v = (k == 0 || residue[k]) ? -1.0f : 1.0f;
			}
			out[i * n + j] = v;
		}
	}
}

/* math.c:35-134.  Sylvester doubling (math.c:102-112) then Kronecker product with the
 * order-12 / order-20 base (math.c:20-33, :114-121).  Quirk Q1: the reference's guard
 * (math.c:96) makes it return NULL for every non power of two; the intended construction
 * is produced here. */
int oracle_hadamard_transpose(int order, float *out)
{
	int dim = order, base = 0;
	if (is_pow2(order))                                  base = order;
	else if (order % 20 == 0 && is_pow2(order / 20))   { base = 20; dim = order / 20; }
	else if (order % 12 == 0 && is_pow2(order / 12))   { base = 12; dim = order / 12; }
	if (!base) return 0;

	int kron = !is_pow2(order);
	float *m = kron ? (float *)malloc(sizeof(float) * dim * dim) : out;
	m[0] = 1;
	for (int k = 1; k < dim; k *= 2) {
		for (int i = 0; i < k; i++) {
			for (int j = 0; j < k; j++) {
				float v = m[i * dim + j];
				m[(i + k) * dim + j]     =  v;
				m[i * dim + j + k]       =  v;
				m[(i + k) * dim + j + k] = -v;
			}
		}
	}
	if (kron) {
		float b[400];
		paley_base(base - 1, b);
		for (int i = 0; i < dim; i++)
			for (int j = 0; j < dim; j++)
				for (int r = 0; r < base; r++)
					for (int c = 0; c < base; c++)
						out[(i * base + r) * order + j * base + c] = m[i * dim + j] * b[r * base + c];
		free(m);
	}
	return 1;
}

/* Modified Bessel I0.  The reference evaluates Cephes' Chebyshev fits
 * (external/cephes.c:24-103, Cephes Math Library 2.8); this is the defining power series
 * sum_k ((x/2)^k / k!)^2 in float64, which agrees with those fits to ~1e-16 relative for
 * the beta values a Kaiser window uses. */
double oracle_bessel_i0(double x)
{
	double q = 0.25 * x * x, term = 1.0, sum = 1.0;
	for (int k = 1; k < 500; k++) {
		term *= q / ((double)k * (double)k);
		sum  += term;
		if (term < 1e-18 * sum) break;
	}
	return sum;
}

/* util.h:86 */
static int f32_equal(float x, float y)
{
	float ax = fabsf(x), ay = fabsf(y);
	float m  = ax > ay ? ax : ay; if (m < 1.0f) m = 1.0f;
	return fabsf(x - y) <= 1e-6f * m;
}

/* math.c:750-767 */
void oracle_kaiser_low_pass(float cutoff, float fs, float beta, int length, float *out)
{
	float wc      = 2 * ORACLE_PI * cutoff / fs;
	float a       = (float)length / 2.0f;
	float pi_i0_b = ORACLE_PI * (float)oracle_bessel_i0(beta);
	for (int n = 0; n < length; n++) {
		float t       = (float)n - a;
		float impulse = !f32_equal(t, 0) ? sinf(wc * t) / t : wc;
		t             = t / a;
		float window  = (float)oracle_bessel_i0(beta * sqrtf(1 - t * t)) / pi_i0_b;
		out[n]        = impulse * window;
	}
}

/* math.c:739-747 */
float oracle_tukey_window(float t, float tapering)
{
	float r = tapering, result = 1;
	if (t < r / 2)      result = 0.5f * (1 + cosf(2 * ORACLE_PI * (t - r / 2)     / r));
	if (t >= 1 - r / 2) result = 0.5f * (1 + cosf(2 * ORACLE_PI * (t - 1 + r / 2) / r));
	return result;
}

/* math.c:769-781 */
void oracle_rf_chirp(float fmin, float fmax, float fs, int length, int reverse, float *out)
{
	for (int i = 0; i < length; i++) {
		int   index = reverse ? length - 1 - i : i;
		float fc    = fmin + (float)i * (fmax - fmin) / (2 * (float)length);
		float arg   = 2 * ORACLE_PI * fc * (float)i / fs;
		out[index]  = sinf(arg) * oracle_tukey_window((float)i / (float)length, 0.2f);
	}
}

/* math.c:783-797 */
void oracle_baseband_chirp(float fmin, float fmax, float fs, int length, int reverse,
                           float scale, float *out)
{
	float conjugate = reverse ? -1 : 1;
	for (int i = 0; i < length; i++) {
		int   index = reverse ? length - 1 - i : i;
		float fc    = fmin + (float)i * (fmax - fmin) / (2 * (float)length);
		float arg   = 2 * ORACLE_PI * fc * (float)i / fs;
		float w     = oracle_tukey_window((float)i / (float)length, 0.2f);
		out[2 * index + 0] = (scale * cosf(arg)) * w;
		out[2 * index + 1] = (conjugate * scale * sinf(arg)) * w;
	}
}

/* math.c:726-737 */
float oracle_real_filter_first_moment(const float *h, int length, float fs)
{
	float n = 0, d = 0;
	for (int i = 0; i < length; i++) {
		float t = h[i] * h[i];
		n += (float)i * t;
		d += t;
	}
	return n / d / fs;
}

/* math.c:713-724 */
float oracle_complex_filter_first_moment(const float *h, int length, float fs)
{
	float n = 0, d = 0;
	for (int i = 0; i < length; i++) {
		float t = h[2 * i] * h[2 * i] + h[2 * i + 1] * h[2 * i + 1];
		n += (float)i * t;
		d += t;
	}
	return n / d / fs;
}

/* beamformer_core.c:366-398 */
int oracle_filter_create(const BeamformerFilterParameters *fp, float *coefficients, int cap,
                         float *time_delay)
{
	int length = 0;
	switch (fp->kind) {
	case BeamformerFilterKind_Kaiser:{
		length = (int)fp->kaiser.length;
		if (length > cap) return -1;
		oracle_kaiser_low_pass(fp->kaiser.cutoff_frequency, fp->sampling_frequency, fp->kaiser.beta, length,
		                       coefficients);
		*time_delay = (float)length / 2.0f / fp->sampling_frequency;
	}break;
	case BeamformerFilterKind_MatchedChirp:{
		float fs = fp->sampling_frequency;
		length   = (int)(fp->matched_chirp.duration * fs);
		if (length * (fp->complex ? 2 : 1) > cap) return -1;
		if (fp->complex) {
			oracle_baseband_chirp(fp->matched_chirp.min_frequency, fp->matched_chirp.max_frequency, fs,
			                      length, 1, 0.5f, coefficients);
			*time_delay = oracle_complex_filter_first_moment(coefficients, length, fs);
		} else {
			oracle_rf_chirp(fp->matched_chirp.min_frequency, fp->matched_chirp.max_frequency, fs,
			                length, 1, coefficients);
			*time_delay = oracle_real_filter_first_moment(coefficients, length, fs);
		}
	}break;
	default: return -1;
	}
	return length;
}

/* math.c:448-458; column major: out.c[i][j] = dot(row_j(a), b.c[i]) */
void oracle_m4_mul(const float *a, const float *b, float *out)
{
	float r[16];
	for (int i = 0; i < 4; i++)
		for (int j = 0; j < 4; j++)
			r[4 * i + j] = a[j] * b[4 * i] + a[4 + j] * b[4 * i + 1] + a[8 + j] * b[4 * i + 2]
			               + a[12 + j] * b[4 * i + 3];
	memcpy(out, r, sizeof(r));
}

/* math.c:799-829 */
static void das_output_dimension(int *p)
{
	for (int i = 0; i < 3; i++) if (p[i] < 1) p[i] = 1;
	int dim = (p[0] > 1) + (p[1] > 1) + (p[2] > 1);
	if (dim == 1) {
		if (p[1] > 1) p[0] = p[1];
		if (p[2] > 1) p[0] = p[2];
		p[1] = p[2] = 1;
	} else if (dim == 2) {
		if (p[0] > 1) { if (p[2] > 1) p[1] = p[2]; }
		else          { p[0] = p[2]; }
		p[2] = 1;
	}
}

/* math.c:844-869 */
static void das_transform_2d_with_normal(const float *normal, const float *mn, const float *mx,
                                         float offset, float *out)
{
	float U[3] = {0, 1.0f, 0};
	if (f32_equal(U[0] * normal[0] + U[1] * normal[1] + U[2] * normal[2], 1.0f)) {
		U[0] = 1.0f; U[1] = 0; U[2] = 0;
	}
	const float *N = normal;
	float V[3] = {U[1] * N[2] - U[2] * N[1], U[2] * N[0] - U[0] * N[2], U[0] * N[1] - U[1] * N[0]};
	float lo[3], hi[3], extent[3];
	for (int i = 0; i < 3; i++) {
		lo[i]     = U[i] * mn[0] + V[i] * mn[1];
		hi[i]     = U[i] * mx[0] + V[i] * mx[1];
		extent[i] = hi[i] - lo[i];
	}
	float ue = U[0] * extent[0] + U[1] * extent[1] + U[2] * extent[2];
	float ve = V[0] * extent[0] + V[1] * extent[1] + V[2] * extent[2];
	for (int i = 0; i < 3; i++) {
		out[0 + i]  = U[i] * ue;
		out[4 + i]  = V[i] * ve;
		out[8 + i]  = N[i];
		out[12 + i] = N[i] * offset + lo[i];
	}
	out[3] = out[7] = out[11] = 0.0f;
	out[15] = 1.0f;
}

/* math.c:871-892 */
void oracle_das_transform_2d(int plane, const float *min2, const float *max2, float offset, float *out16)
{
	float n_xz[3] = {0, 1.0f, 0}, n_yz[3] = {-1.0f, 0, 0}, n_xy[3] = {0, 0, 1.0f};
	das_transform_2d_with_normal(plane == 0 ? n_xz : plane == 1 ? n_yz : n_xy, min2, max2, offset, out16);
}

/* math.c:894-904 */
void oracle_das_transform_3d(const float *mn, const float *mx, float *out)
{
	memset(out, 0, 16 * sizeof(float));
	out[0]  = mx[0] - mn[0];
	out[5]  = mx[1] - mn[1];
	out[10] = mx[2] - mn[2];
	out[12] = mn[0]; out[13] = mn[1]; out[14] = mn[2]; out[15] = 1.0f;
}

/* math.c:906-920 (1-D: math.c:831-842) */
void oracle_das_transform(const float *mn, const float *mx, int *points, float *out)
{
	das_output_dimension(points);
	int dim = (points[0] > 1) + (points[1] > 1) + (points[2] > 1);
	switch (dim) {
	case 1:{
		memset(out, 0, 16 * sizeof(float));
		out[0]  = mx[0] - mn[0]; out[1]  = mx[1] - mn[1]; out[2]  = mx[2] - mn[2];
		out[12] = mn[0];         out[13] = mn[1];         out[14] = mn[2];  out[15] = 1.0f;
	}break;
	case 2: oracle_das_transform_2d(0, mn, mx, 0, out); break;
	case 3: oracle_das_transform_3d(mn, mx, out);       break;
	default: memset(out, 0, 16 * sizeof(float));        break;
	}
}


/* the software binary16 of oracle_f16.h, exported so that tests can pin it against an independent
 * IEEE implementation (numpy) */
void oracle_f16_bits_from_f32(const float *in, uint16_t *out, uint64_t n)
{
	for (uint64_t i = 0; i < n; i++) out[i] = oracle_f32_to_f16_bits(in[i]);
}

void oracle_f16_roundtrip(const uint16_t *bits, float *out, uint64_t n)
{
	for (uint64_t i = 0; i < n; i++) out[i] = oracle_f16_bits_to_f32(bits[i]);
}


/* oracle.h: the build-defined Hilbert FIR, stored in the correlation order oracle_hilbert uses */
void oracle_hilbert_fir(float *taps)
{
	const int L = ORACLE_HILBERT_LENGTH, M = (ORACLE_HILBERT_LENGTH - 1) / 2;
	for (int j = 0; j < L; j++) {
		int    m = j - M;
		double w = 0.54 - 0.46 * cos(2.0 * 3.14159265358979323846 * (double)j / (double)(L - 1));
		taps[2 * j]     = j == M ? 1.0f : 0.0f;
		taps[2 * j + 1] = (m & 1) ? (float)(-(2.0 / (3.14159265358979323846 * (double)m)) * w) : 0.0f;
	}
}
