/* oracle_das.c -- CPU ORACLE (test infrastructure): delay-and-sum, restating
 * shaders/das.glsl.  The loops live in oracle_das_body.h, instantiated for float (the
 * checker) and double (tolerance truth).  PARITY UNPINNED by the reference (see oracle.h). */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int oracle_thread_count(int requested)
{
#ifdef _OPENMP
	return requested > 0 ? requested : omp_get_max_threads();
#else
	(void)requested;
	return 1;
#endif
}

/* Nearest interpolation is discontinuous at index = k + 1/2 (and at the two ends of the valid range):
 * an implementation whose index differs by float rounding picks the other sample there.  For every
 * such tap the oracle adds |other sample - chosen sample| (the most the flip can move the coherent
 * sum; apodization and weights are <= 1) to a per-thread budget; das_run adds each voxel's budget to
 * the caller's buffer (when set), so a parity test can demand |gpu - oracle| <= tolerance + budget on
 * EVERY voxel instead of allowing a fraction of mismatches.  Test infrastructure, not the shader. */
#define ORACLE_NEAR_HALF (1.0 / 1024.0)
static _Thread_local float oracle_near_half_budget;
static float *oracle_near_half_buffer;          /* one float per sub-grid voxel, or NULL */
void oracle_set_nearest_ambiguity_buffer(float *budget) { oracle_near_half_buffer = budget; }

#define REAL     float
#define FN(n)    n##_f32
#define R_SQRT   sqrtf
#define R_SIN    sinf
#define R_COS    cosf
#define R_FLOOR  floorf
#define R_FABS   fabsf
#define R_ROUND  roundf
#define R_MODF   modff
#define R_ISINF  isinf
#define R_PI     3.14159265358979323846f
#include "oracle_das_body.h"
#undef REAL
#undef FN
#undef R_SQRT
#undef R_SIN
#undef R_COS
#undef R_FLOOR
#undef R_FABS
#undef R_ROUND
#undef R_MODF
#undef R_ISINF
#undef R_PI

#define REAL     double
#define FN(n)    n##_f64
#define R_SQRT   sqrt
#define R_SIN    sin
#define R_COS    cos
#define R_FLOOR  floor
#define R_FABS   fabs
#define R_ROUND  round
#define R_MODF   modf
#define R_ISINF  isinf
#define R_PI     3.14159265358979323846
#include "oracle_das_body.h"

uint64_t oracle_das(const OracleDAS *p, const float *rf, float *output, float *incoherent)
{
	return das_run_f32(p, rf, output, incoherent);
}

uint64_t oracle_das_f64(const OracleDAS *p, const float *rf, double *output, double *incoherent)
{
	return das_run_f64(p, rf, output, incoherent);
}
