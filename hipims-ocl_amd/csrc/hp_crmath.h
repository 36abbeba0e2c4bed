/* hp_crmath.h -- one libm-independent cube root for host AND device.
 *
 * Why.  The friction term evaluates pow(h, 1.0/3.0) (reference: Schemes/CLFriction.clc:43; also
 * Boundaries/CLBoundaries.clc:81 and, as pow(d, 10.0/3.0), Schemes/CLSchemeInertial.clc:352).  An OpenCL device's pow
 * is whatever its vendor library computes (<= 16 ulp by the standard); glibc's, the ROCm device library's and any CPU
 * OpenCL runtime's all differ in the last bit on a few per cent of arguments, so no two runs of the reference on
 * different platforms agree bit for bit once friction is on -- and neither could a STRICT HIP kernel and a host
 * oracle.  This header fixes ONE result for that operation: the correctly rounded cube root, computed with IEEE
 * +, -, * and fma only (no table, no libm call), so that gcc on x86-64 and hipcc on gfx950 produce the same bits.
 * A correctly rounded cbrt(x) differs from the infinitely precise pow(x, fl(1/3)) by |ln x| * 1.85e-17 relative
 * (fl(1/3) = 1/3 - 2^-54/3): at most 4.3 ulp at a depth of 1e-10 m, 1.3 ulp at 1 mm, 0.5 ulp at 1 m (its own rounding
 * included; measured in tests/test_crmath.py) -- a conforming `pow` result by a wide margin.
 *
 * Who uses it.  oracle/swe_oracle.c (R_POW13 / R_POW103), oracle/ref_build/shim.cpp (the `pow` it hands the reference's
 * kernels) and the STRICT flavour of the HIP kernels (hp_math.hpp).  The FAST flavour keeps its own rcbrt_fast.
 * tests/test_crmath.py checks correct rounding against exact integer arithmetic on a few hundred thousand arguments.
 *
 * The including translation unit must not contract a*b+c (all three are built with -ffp-contract=off); every fused
 * operation below is an explicit __builtin_fma.
 */
#ifndef HP_CRMATH_H
#define HP_CRMATH_H

#include <stdint.h>

#ifndef HP_CR_FN
#define HP_CR_FN static inline
#endif

HP_CR_FN double hp_cr_from_bits(uint64_t b) { double d; __builtin_memcpy(&d, &b, 8); return d; }
HP_CR_FN uint64_t hp_cr_to_bits(double d) { uint64_t b; __builtin_memcpy(&b, &d, 8); return b; }

/* cbrt(x), correctly rounded (round to nearest even), for x >= 0; NaN for x < 0 and NaN (what pow(x, 1/3) returns). */
HP_CR_FN double hp_cr_cbrt(double x)
{
	if (!(x > 0.0)) return (x == 0.0) ? 0.0 : hp_cr_from_bits(0x7ff8000000000000ull);
	uint64_t bits = hp_cr_to_bits(x);
	int e = (int)((bits >> 52) & 0x7ff);
	if (e == 0x7ff) return x;                                  /* +inf */
	int bias = 0;
	if (e == 0) {                                              /* subnormal: scale by 2^54 (exact), cbrt by 2^-18 */
		x = x * 18014398509481984.0;
		bits = hp_cr_to_bits(x);
		e = (int)((bits >> 52) & 0x7ff);
		bias = 18;
	}
	/* x = m * 2^(3k), m in [1, 8) */
	const int E = e - 1023;
	int k = E / 3, r = E - 3 * k;
	if (r < 0) { r += 3; k -= 1; }
	const double m = hp_cr_from_bits((bits & 0x000fffffffffffffull) | ((uint64_t)(1023 + r) << 52));

	/* Division-free (round 4; the results are those of the round-3 routine, which spent four IEEE divisions here -- 4e9
	 * random arguments compared bit for bit, tools/README.md).  s ~ m^(-1/3): a minimax cubic on [1, 8) (relative error
	 * < 1.7e-2), then two steps of the third-order iteration s <- s (1 + d/3 + 2 d^2/9), d = 1 - m s^3 (error
	 * e -> 14/81 e^3: 1.7e-2 -> 8e-7 -> rounding level).  y = m s^2 is the root to a few ulp, w = s^2/3 is 1/(3 y^2) to
	 * ~1e-15, and one Newton step with a fused residual brings y within an ulp. */
	double s = 1.2370148232167815 + m * (-0.2972025836621341 + m * (0.04589330067811839 + m * -0.002548799012225146));
	for (int i = 0; i < 2; ++i) {
		const double s3 = (s * s) * s;
		const double d = __builtin_fma(-m, s3, 1.0);
		const double t = d * __builtin_fma(d, 2.0 / 9.0, 1.0 / 3.0);
		s = __builtin_fma(s, t, s);
	}
	const double w = (s * s) * (1.0 / 3.0);
	double y = m * (s * s);
	{ const double y2 = y * y; y = __builtin_fma(__builtin_fma(-y2, y, m), w, y); }
	/* last step with the residual m - y^3 carried in double-double: y^2 = y2h + y2l and y2h*y = y3h + y3l exactly, so
	 * y^3 = y3h + y3l + y2l*y up to 2^-106; m - y3h is exact (Sterbenz).  c = (m - y^3) w is the Newton correction, good
	 * to 2^-49 of itself, and |c| <= 1 ulp(y): RN(y + c) is the correctly rounded root unless the root lies within
	 * ~2^-102 (relative) of a rounding boundary. */
	const double y2h = y * y, y2l = __builtin_fma(y, y, -y2h);
	const double y3h = y2h * y, y3l = __builtin_fma(y2h, y, -y3h);
	const double res = ((m - y3h) - y3l) - y2l * y;
	y = __builtin_fma(res, w, y);
	/* scale by 2^(k - bias): exact (the result is a normal number for every finite positive double) */
	return y * hp_cr_from_bits((uint64_t)(1023 + k - bias) << 52);
}

/* float: the double result rounded once more (correctly rounded except for double-rounding ties, ~1e-9 of arguments;
 * deterministic either way, which is what the parity tests need) */
HP_CR_FN float hp_cr_cbrtf(float x) { return (float)hp_cr_cbrt((double)x); }

/* x^(10/3) as the inertial scheme uses it: x^3 * cbrt(x), products in this order */
HP_CR_FN double hp_cr_pow103(double x) { return ((x * x) * x) * hp_cr_cbrt(x); }
HP_CR_FN float hp_cr_pow103f(float x) { return ((x * x) * x) * hp_cr_cbrtf(x); }

#endif /* HP_CRMATH_H */
