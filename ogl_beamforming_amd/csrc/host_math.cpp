/* host_math.cpp -- host-side DSP math of the beamformer core: Hadamard matrices, Kaiser /
 * matched-chirp filter design and small matrix helpers.  The reference computes these on
 * the host too (math.c:35-134, :713-797, :448-458; beamformer_core.c:366-398) and uploads
 * the tables; so does this library.  Pinned bit-for-bit (power-of-two Hadamard, Kaiser,
 * chirps, moments) against the compiled reference by tests/test_host_math.py through the
 * bf_host_* debug exports at the bottom of lib_api.cpp. */
#include "host_math.h"
#include <cmath>
#include <cstring>

namespace bf {

static constexpr float kPi = 3.14159265358979323846f;   /* base_types.h:33-35 */

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

/* Normalised Hadamard matrix of order q+1 (q prime, q = 3 mod 4) with a (back-)circulant
 * core built from quadratic residues; reproduces the two literal base matrices of the
 * reference (math.c:38-76): order 12 shifts the core right per row, order 20 left. */
static std::vector<float> residue_hadamard(int q, bool shift_right)
{
	int n = q + 1;
	std::vector<char>  qr(q, 0);
	for (int k = 1; k < q; k++) qr[(k * k) % q] = 1;
	std::vector<float> h((size_t)n * n, 1.0f);
	for (int i = 1; i < n; i++)
		for (int j = 1; j < n; j++) {
			int k = shift_right ? ((j - i) % q + q) % q : (i + j - 2) % q;
			h[(size_t)i * n + j] = (k == 0 || qr[k]) ? -1.0f : 1.0f;
		}
	return h;
}

/* math.c:35-134; orders 2^k, 12*2^k, 20*2^k.  (The reference's snapshot returns NULL for
 * the last two through the guard at math.c:96; the intended Kronecker construction
 * math.c:114-121 is produced.) */
std::vector<float> hadamard_transpose(int order)
{
	int dim = order, base = 0;
	if (order <= 0 || order > 4096) return {};            /* callers pass transmit / group counts (<= 256) */
	if (is_pow2(order))                                base = 1;
	else if (order % 20 == 0 && is_pow2(order / 20)) { base = 20; dim = order / 20; }
	else if (order % 12 == 0 && is_pow2(order / 12)) { base = 12; dim = order / 12; }
	if (!base) return {};

	std::vector<float> sylvester((size_t)dim * dim, 0.0f);
	sylvester[0] = 1.0f;
	for (int k = 1; k < dim; k *= 2)
		for (int i = 0; i < k; i++)
			for (int j = 0; j < k; j++) {
				float v = sylvester[(size_t)i * dim + j];
				sylvester[(size_t)(i + k) * dim + j]     =  v;
				sylvester[(size_t)i * dim + j + k]       =  v;
				sylvester[(size_t)(i + k) * dim + j + k] = -v;
			}
	if (base == 1) return sylvester;

	std::vector<float> b = residue_hadamard(base - 1, base == 12);
	std::vector<float> out((size_t)order * order);
	for (int i = 0; i < dim; i++)
		for (int j = 0; j < dim; j++)
			for (int r = 0; r < base; r++)
				for (int c = 0; c < base; c++)
					out[(size_t)(i * base + r) * order + j * base + c] =
						sylvester[(size_t)i * dim + j] * b[(size_t)r * base + c];
	return out;
}

/* I0 by its power series in double; the reference uses Cephes' Chebyshev fits
 * (external/cephes.c:24-103), equal to ~1e-16 relative */
double bessel_i0(double x)
{
	double q = 0.25 * x * x, term = 1.0, sum = 1.0;
	for (int k = 1; k < 500; k++) {
		term *= q / ((double)k * (double)k);
		sum  += term;
		if (term < 1e-18 * sum) break;
	}
	return sum;
}

static bool nearly_equal(float x, float y)                /* util.h:86 */
{
	float m = std::fmax(1.0f, std::fmax(std::fabs(x), std::fabs(y)));
	return std::fabs(x - y) <= 1e-6f * m;
}

/* math.c:750-767 */
std::vector<float> kaiser_low_pass(float cutoff, float fs, float beta, int length)
{
	std::vector<float> h((size_t)(length > 0 ? length : 0));
	float wc = 2 * kPi * cutoff / fs;
	float a  = (float)length / 2.0f;
	float norm = kPi * (float)bessel_i0(beta);
	for (int n = 0; n < length; n++) {
		float t       = (float)n - a;
		float impulse = !nearly_equal(t, 0) ? sinf(wc * t) / t : wc;
		t             = t / a;
		float window  = (float)bessel_i0(beta * sqrtf(1 - t * t)) / norm;
		h[(size_t)n]  = impulse * window;
	}
	return h;
}

/* math.c:739-747 */
float tukey_window(float t, float tapering)
{
	float r = tapering, w = 1;
	if (t < r / 2)      w = 0.5f * (1 + cosf(2 * kPi * (t - r / 2)     / r));
	if (t >= 1 - r / 2) w = 0.5f * (1 + cosf(2 * kPi * (t - 1 + r / 2) / r));
	return w;
}

/* math.c:769-781 */
std::vector<float> rf_chirp(float fmin, float fmax, float fs, int length, bool reverse)
{
	std::vector<float> h((size_t)(length > 0 ? length : 0));
	for (int i = 0; i < length; i++) {
		int   index = reverse ? length - 1 - i : i;
		float fc    = fmin + (float)i * (fmax - fmin) / (2 * (float)length);
		float arg   = 2 * kPi * fc * (float)i / fs;
		h[(size_t)index] = sinf(arg) * tukey_window((float)i / (float)length, 0.2f);
	}
	return h;
}

/* math.c:783-797; interleaved re, im */
std::vector<float> baseband_chirp(float fmin, float fmax, float fs, int length, bool reverse, float scale)
{
	std::vector<float> h((size_t)(length > 0 ? 2 * length : 0));
	float conjugate = reverse ? -1.0f : 1.0f;
	for (int i = 0; i < length; i++) {
		int   index = reverse ? length - 1 - i : i;
		float fc    = fmin + (float)i * (fmax - fmin) / (2 * (float)length);
		float arg   = 2 * kPi * fc * (float)i / fs;
		float w     = tukey_window((float)i / (float)length, 0.2f);
		h[(size_t)2 * index]     = (scale * cosf(arg)) * w;
		h[(size_t)2 * index + 1] = (conjugate * scale * sinf(arg)) * w;
	}
	return h;
}

/* math.c:713-737 */
float filter_first_moment(const std::vector<float> &h, bool complex_taps, float fs)
{
	float n = 0, d = 0;
	size_t length = complex_taps ? h.size() / 2 : h.size();
	for (size_t i = 0; i < length; i++) {
		float t = complex_taps ? h[2 * i] * h[2 * i] + h[2 * i + 1] * h[2 * i + 1] : h[i] * h[i];
		n += (float)i * t;
		d += t;
	}
	return n / d / fs;
}

/* beamformer_core.c:366-398 */
bool filter_create(const BeamformerFilterParameters &fp, Filter &out)
{
	out = Filter{};
	out.complex_taps = fp.complex != 0;
	switch (fp.kind) {
	case BeamformerFilterKind_Kaiser:
		/* the reference's Kaiser path always produces real taps, whatever `complex` says
		 * (beamformer_core.c:372); the ComplexFilter compile flag is still taken from the
		 * parameters (beamformer_core.c:833), so real taps would be read as pairs.  Real
		 * taps + real flag is the only self-consistent combination: enforce it. */
		out.complex_taps = false;
		out.length       = (int)fp.kaiser.length;
		if (out.length <= 0 || out.length > 4096) return false;
		out.taps         = kaiser_low_pass(fp.kaiser.cutoff_frequency, fp.sampling_frequency, fp.kaiser.beta, out.length);
		out.time_delay   = (float)out.length / 2.0f / fp.sampling_frequency;
		return true;
	case BeamformerFilterKind_MatchedChirp:{
		float fs   = fp.sampling_frequency;
		out.length = (int)(fp.matched_chirp.duration * fs);
		if (out.length <= 0 || out.length > 4096) return false;
		if (out.complex_taps) out.taps = baseband_chirp(fp.matched_chirp.min_frequency, fp.matched_chirp.max_frequency, fs, out.length, true, 0.5f);
		else                  out.taps = rf_chirp(fp.matched_chirp.min_frequency, fp.matched_chirp.max_frequency, fs, out.length, true);
		out.time_delay = filter_first_moment(out.taps, out.complex_taps, fs);
		return true;
	}
	default: return false;
	}
}

/* math.c:448-458, column major */
void m4_mul(const float *a, const float *b, float *out)
{
	float r[16];
	for (int i = 0; i < 4; i++)
		for (int j = 0; j < 4; j++)
			r[4 * i + j] = a[j] * b[4 * i] + a[4 + j] * b[4 * i + 1] + a[8 + j] * b[4 * i + 2] + a[12 + j] * b[4 * i + 3];
	std::memcpy(out, r, sizeof(r));
}

/* The build-defined Hilbert stage (include/ogl_beamformer_hip.h, beamformer_hip_enable_hilbert):
 * taps of the analytic-signal FIR  y[n] = sum_{j<63} h[j] x[n - 62 + j]:  h[31] = 1 and, for odd m,
 * h[31 + m] = -j (2 / (pi m)) w[31 + m] with a Hamming window w -- a type-III Hilbert transformer
 * next to a pure delay of 31 samples. */
std::vector<float> hilbert_fir()
{
	const int L = kHilbertLength, M = (kHilbertLength - 1) / 2;
	std::vector<float> taps(2 * L);
	for (int j = 0; j < L; j++) {
		int    m = j - M;
		double w = 0.54 - 0.46 * std::cos(2.0 * 3.14159265358979323846 * (double)j / (double)(L - 1));
		taps[2 * j]     = j == M ? 1.0f : 0.0f;
		taps[2 * j + 1] = (m & 1) ? (float)(-(2.0 / (3.14159265358979323846 * (double)m)) * w) : 0.0f;
	}
	return taps;
}

/* das_transform (math.c:906-920) with das_output_dimension (:799-829): the grid keeps its
 * extents but is renumbered so that a line is (n,1,1) and a plane is (a,b,1); a plane is the
 * world x-z plane at y = 0 (das_transform_2d_xz :872-877 -> das_transform_2d_with_normal
 * :844-870 with N = +y: U = x, V = U x N = z), min/max giving (x, z) in their first two
 * components; a box maps the unit cube onto [min, max] (:894-904); a line runs from min to
 * max (:831-842).  Column major. */
void das_transform(const float mn[3], const float mx[3], int32_t points[3], float out[16])
{
	for (int i = 0; i < 3; i++) if (points[i] < 1) points[i] = 1;
	const int dim = (points[0] > 1) + (points[1] > 1) + (points[2] > 1);
	for (int i = 0; i < 16; i++) out[i] = 0.0f;
	if (dim == 1) {
		if (points[1] > 1) points[0] = points[1];
		if (points[2] > 1) points[0] = points[2];
		points[1] = points[2] = 1;
		for (int i = 0; i < 3; i++) { out[i] = mx[i] - mn[i]; out[12 + i] = mn[i]; }
		out[15] = 1.0f;
	} else if (dim == 2) {
		if (points[0] > 1) { if (points[2] > 1) points[1] = points[2]; }
		else               { points[0] = points[2]; }
		points[2] = 1;
		const float U[3] = {1.0f, 0.0f, 0.0f}, N[3] = {0.0f, 1.0f, 0.0f};
		const float V[3] = {U[1] * N[2] - U[2] * N[1], U[2] * N[0] - U[0] * N[2], U[0] * N[1] - U[1] * N[0]};
		float lo[3], extent[3];
		for (int i = 0; i < 3; i++) {
			lo[i]     = U[i] * mn[0] + V[i] * mn[1];
			extent[i] = (U[i] * mx[0] + V[i] * mx[1]) - lo[i];
		}
		const float ue = U[0] * extent[0] + U[1] * extent[1] + U[2] * extent[2];
		const float ve = V[0] * extent[0] + V[1] * extent[1] + V[2] * extent[2];
		for (int i = 0; i < 3; i++) {
			out[i] = U[i] * ue; out[4 + i] = V[i] * ve; out[8 + i] = N[i];
			out[12 + i] = N[i] * 0.0f + lo[i];
		}
		out[15] = 1.0f;
	} else if (dim == 3) {
		out[0] = mx[0] - mn[0]; out[5] = mx[1] - mn[1]; out[10] = mx[2] - mn[2];
		out[12] = mn[0]; out[13] = mn[1]; out[14] = mn[2]; out[15] = 1.0f;
	}
}

} // namespace bf
