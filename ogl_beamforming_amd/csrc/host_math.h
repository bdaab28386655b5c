/* host_math.h -- see host_math.cpp */
#ifndef BF_HOST_MATH_H
#define BF_HOST_MATH_H
#include <vector>
#include "../../include/ogl_beamformer_lib.h"

namespace bf {

struct Filter {
	std::vector<float> taps;        /* length floats, or 2*length (re, im) when complex_taps */
	int   length       = 0;
	bool  complex_taps = false;
	float time_delay   = 0;         /* added to the DAS time offset (beamformer_core.c:835) */
};

std::vector<float> hadamard_transpose(int order);                 /* row major, Ht[T*j + i]; empty: no construction */
double             bessel_i0(double x);
std::vector<float> kaiser_low_pass(float cutoff, float fs, float beta, int length);
float              tukey_window(float t, float tapering);
std::vector<float> rf_chirp(float fmin, float fmax, float fs, int length, bool reverse);
std::vector<float> baseband_chirp(float fmin, float fmax, float fs, int length, bool reverse, float scale);
float              filter_first_moment(const std::vector<float> &h, bool complex_taps, float fs);
bool               filter_create(const BeamformerFilterParameters &fp, Filter &out);
constexpr int      kHilbertLength = 63;
std::vector<float> hilbert_fir();                                     /* 2*63 floats (re, im), correlation order */
void               m4_mul(const float *a, const float *b, float *out);
void               das_transform(const float mn[3], const float mx[3], int32_t points[3], float out16[16]);

} // namespace bf
#endif
