/* oracle_f16.h -- TEST INFRASTRUCTURE.  IEEE binary16 <-> binary32 in portable C
 * (gcc 11 has no _Float16 on x86).  Round-to-nearest-even, subnormals kept.
 * f16 arithmetic in the shaders (filter.glsl:2-10, :99-107) is emulated as
 * round16(f32 op): exact for one add/mul because binary32 carries > 2*11+2 bits. */
#ifndef ORACLE_F16_H
#define ORACLE_F16_H
#include <stdint.h>
#include <string.h>

static inline uint16_t oracle_f32_to_f16_bits(float f)
{
	uint32_t x; memcpy(&x, &f, 4);
	uint32_t sign = (x >> 16) & 0x8000u;
	uint32_t mant = x & 0x007FFFFFu;
	int32_t  exp  = (int32_t)((x >> 23) & 0xFF);
	if (exp == 0xFF) return (uint16_t)(sign | 0x7C00u | (mant ? 0x0200u | (mant >> 13) : 0));
	exp = exp - 127 + 15;
	if (exp >= 0x1F) return (uint16_t)(sign | 0x7C00u);          /* overflow -> inf */
	if (exp <= 0) {                                              /* subnormal / zero */
		if (exp < -10) return (uint16_t)sign;
		mant |= 0x00800000u;
		uint32_t shift = (uint32_t)(14 - exp);
		uint32_t half  = mant >> shift;
		uint32_t rem   = mant & ((1u << shift) - 1);
		uint32_t mid   = 1u << (shift - 1);
		if (rem > mid || (rem == mid && (half & 1))) half++;
		return (uint16_t)(sign | half);
	}
	uint32_t half = (uint32_t)(exp << 10) | (mant >> 13);
	uint32_t rem  = mant & 0x1FFFu;
	if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) half++; /* may carry into exp: ok */
	return (uint16_t)(sign | half);
}

static inline float oracle_f16_bits_to_f32(uint16_t h)
{
	uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
	uint32_t exp  = (h >> 10) & 0x1F;
	uint32_t mant = h & 0x3FFu;
	uint32_t x;
	if (exp == 0) {
		if (mant == 0) x = sign;
		else {
			int e = -1;
			do { e++; mant <<= 1; } while (!(mant & 0x400u));
			x = sign | (uint32_t)((127 - 15 - e) << 23) | ((mant & 0x3FFu) << 13);
		}
	} else if (exp == 0x1F) {
		x = sign | 0x7F800000u | (mant << 13);
	} else {
		x = sign | ((exp + 127 - 15) << 23) | (mant << 13);
	}
	float f; memcpy(&f, &x, 4);
	return f;
}

/* value of f after a round trip through binary16 */
static inline float oracle_round_f16(float f)
{
	return oracle_f16_bits_to_f32(oracle_f32_to_f16_bits(f));
}
#endif
