// hp_math.hpp -- device-side arithmetic of the shallow-water step (gfx950, wave64).
//
// Written from the numerical specification in SURVEY.md Appendix A; each block cites the reference
// statement it must agree with (paths relative to the reference's src/).  Unlike the reference, a cell
// face is solved ONCE and finished twice -- once for the cell on its left/south side and once for the
// cell on its right/north side -- because everything expensive in the reconstruction + HLLC solve
// (depths, velocities, celerities, wave speeds: all the divisions and square roots) is independent of
// which cell is "own"; only the vertical shift (CLSchemeGodunov.clc:85-86, :136-139) differs.
//
// The translation unit is compiled with -ffp-contract=off: STRICT code therefore rounds exactly like
// the oracle; FAST code asks for every fused multiply-add explicitly (fma_()).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define HP_CR_FN __host__ __device__ __forceinline__
#include "hp_crmath.h"          // the cube root STRICT shares with the oracle and the reference build (see there)

namespace hp {

enum : int { AXIS_X = 0, AXIS_Y = 1 };

template <typename T> struct Params {
	long cols, rows;             // local array size
	long row_offset, global_rows;
	T    dx, inv_dx, vs, qs, courant, t_end, dt_fixed;
	T    inv_dx_pow2;            // 1 / dx where dx is a power of two (then x / dx == x * (1 / dx) bit for bit: STRICT multiplies), else 0
	int  friction, dynamic_dt;
	int  manning_uniform;        // every cell has the same Manning n (the usual "constant" data source): not re-read per step
	T    manning_value;
	int  simplified_cfl;         // TIMESTEP_SIMPLIFIED (CLSchemeInertial.clh:25): wave speed = sqrt(g h) only
	int  muscl_nb_bed;           // HP_QUIRK_MUSCL_NEIGHBOUR_Y_IS_BED: the predictor's first-order test reads the neighbours' bed
};

// device-resident time-control block ("Time", "Timestep", ... buffers, CSchemeGodunov.cpp:852-872)
template <typename T> struct Scalars {
	T        t, dt, t_hydro, t_sync, batch_dt;
	uint32_t batch_ok, batch_skipped;
};

template <typename T> struct alignas(sizeof(T) * 4) State4 { T z, zmax, qx, qy; };

template <typename T> __device__ __forceinline__ constexpr T gravity() { return T(9.81); }   // CLUniversalHeader.clh:33

__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float  fma_(float a, float b, float c)    { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double sqrt_(double x) { return __builtin_sqrt(x); }
__device__ __forceinline__ float  sqrt_(float x)  { return __builtin_sqrtf(x); }
// STRICT's pow: the reference asks for three exponents only (1/3, 10/3, 2); the first two come from hp_crmath.h --
// correctly rounded cube root from IEEE basic operations, bit-identical on host and device -- so that a STRICT run
// equals the oracle WITH friction on (round 2 used the device library's pow: last-bit differences from glibc's)
__device__ __forceinline__ double pow13_(double x)  { return hp_cr_cbrt(x); }
__device__ __forceinline__ float  pow13_(float x)   { return hp_cr_cbrtf(x); }
__device__ __forceinline__ double pow103_(double x) { return hp_cr_pow103(x); }
__device__ __forceinline__ float  pow103_(float x)  { return hp_cr_pow103f(x); }
__device__ __forceinline__ double fabs_(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float  fabs_(float x)  { return __builtin_fabsf(x); }
__device__ __forceinline__ double fmax_(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float  fmax_(float a, float b)   { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double fmin_(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float  fmin_(float a, float b)   { return __builtin_fminf(a, b); }
__device__ __forceinline__ double fmod_(double a, double b) { return fmod(a, b); }
__device__ __forceinline__ float  fmod_(float a, float b)   { return fmodf(a, b); }
__device__ __forceinline__ double floor_(double x) { return __builtin_floor(x); }
__device__ __forceinline__ float  floor_(float x)  { return __builtin_floorf(x); }

// Wave-uniform votes straight from the compare's lane mask (round 5): HIP's __all / __any go through a per-lane 0/1 register
// (v_cndmask + v_cmp_ne + s_cmp -- two VALU instructions per vote, five votes per row step); the ballot IS the compare's SGPR
// pair ANDed with exec.  Same values; all active lanes vote, as with __all / __any.
__device__ __forceinline__ bool wave_all(const bool p) { return __builtin_amdgcn_ballot_w64(p) == __builtin_amdgcn_ballot_w64(true); }
__device__ __forceinline__ bool wave_any(const bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
// a value every lane holds alike, moved to scalar registers (v_readfirstlane): vector instructions take it as a scalar operand
__device__ __forceinline__ float uniform_value(const float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// HIDE: an empty statement in front hides where v comes from.  Without it the compiler moves the readfirstlane in front of a
// multiplication by a constant (and drops it behind operands it knows to be uniform): the product is a vector instruction again and
// sits in a pair of vector registers for the whole kernel -- harmless where there is room (the plain pair kernel measures 0.8 %
// FASTER that way: fewer scalar spill moves in its loop), a spill with a scratch re-load in front of every friction term where not
// (the pair kernels with area boundaries and with stamps: S-RAIN 4096^2 fp64 0.2676 -> 0.2544 ms, every pair exact on S-ROUGH
// 0.2766 -> 0.2556: profiles/r06ag_opaque_uniform_value_ab.txt)
template <bool HIDE = false> __device__ __forceinline__ double uniform_value(double v)
{
	if constexpr (HIDE) asm("" : "+v"(v));
	return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <bool HIDE> __device__ __forceinline__ float uniform_value(const float v) { return uniform_value(v); }

// v_max_f64 / v_min_f64 WITHOUT the canonicalising v_max x, x, x the compiler puts in front of fmax / fmin whenever an operand
// comes out of memory, a lane move or a select (it must assume a signalling NaN; the hardware instruction quiets one by itself).
// FAST flavour only, and only where no source modifier would have been folded in.
__device__ __forceinline__ double fmax_raw(const double a, const double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double fmin_raw(const double a, const double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double fmax0_raw(const double a) { double r; asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(a)); return r; }
__device__ __forceinline__ double fmin0_raw(const double a) { double r; asm("v_min_f64 %0, %1, 0" : "=v"(r) : "v"(a)); return r; }
__device__ __forceinline__ float fmax_raw(const float a, const float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float fmin_raw(const float a, const float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float fmax0_raw(const float a) { float r; asm("v_max_f32 %0, %1, 0" : "=v"(r) : "v"(a)); return r; }
__device__ __forceinline__ float fmin0_raw(const float a) { float r; asm("v_min_f32 %0, %1, 0" : "=v"(r) : "v"(a)); return r; }

// ---- FAST-flavour primitives: hardware seed + Newton steps instead of the IEEE expansions ----
// 1/x to about 1 ulp: v_rcp_f64 seed r, residual e = 1 - x r, ONE third-order step r (1 + e + e^2) -- what is left is e^3 (1e-22)
// and the last fma's rounding (the IEEE division adds scaling + fix-up; two Newton-Raphson steps, rounds 2-4, are one fma more).
// Measured on gfx950 (tools/seedcheck): seed 4.6e-8 relative (2^-24.4), one Newton step 2.1e-15, two steps 1.1e-16;
// v_rsq_f64 seed 5.2e-8.
__device__ __forceinline__ double rcp_fast(double x)
{
	const double r = __builtin_amdgcn_rcp(x);
	double e = __builtin_fma(-x, r, 1.0);
	e = __builtin_fma(e, e, e);
	return __builtin_fma(r, e, r);
}
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }   // v_rcp_f32 is 1 ulp already
// sqrt(x), x >= 0, to about 1 ulp: v_rsq_f64 seed, Goldschmidt step + residual correction.  x is clamped away
// from zero (sqrt(1e-300) = 1e-150 stands in for 0; depths below VERY_SMALL never reach a division by it).
__device__ __forceinline__ double sqrt_fast(double x)
{
	x = __builtin_fmax(x, 1e-300);
	const double y = __builtin_amdgcn_rsq(x);
	double g = x * y;
	const double h = 0.5 * y;                         // (left at the seed's 5e-8: it only scales the last correction, itself 4e-15 of g)
	const double r = __builtin_fma(-h, g, 0.5);
	g = __builtin_fma(g, r, g);
	const double d = __builtin_fma(-g, g, x);
	return __builtin_fma(d, h, g);
}
__device__ __forceinline__ float sqrt_fast(float x) { return __builtin_amdgcn_sqrtf(x); }   // v_sqrt_f32: 1 ulp, no denormal rescaling
// the same for an argument the caller keeps away from zero (x >= 1e-300)
__device__ __forceinline__ double sqrt_fast_pos(double x)
{
	const double y = __builtin_amdgcn_rsq(x);
	double g = x * y;
	const double h = 0.5 * y;
	const double r = __builtin_fma(-h, g, 0.5);
	g = __builtin_fma(g, r, g);
	const double d = __builtin_fma(-g, g, x);
	return __builtin_fma(d, h, g);
}
// x^(-1/3), x > 0: fp32 exp2/log2 seed y (relative error e ~ 1e-6), residual e = 1 - x y^3, ONE third-order step
// y (1 + e/3 + 2 e^2/9) -- the series of (1 - e)^(-1/3); what is left is 14/81 e^3 (1e-19).  (Rounds 2-4: two Newton steps, four
// instructions more.)
__device__ __forceinline__ double rcbrt_fast(double x)
{
	const float xf = (float)x;
	const double y = (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf(xf) * (-1.0f / 3.0f));
	const double e = __builtin_fma(-x * y, y * y, 1.0);
	const double p = __builtin_fma(e, 2.0 / 9.0, 1.0 / 3.0);
	return __builtin_fma(y, p * e, y);
}
__device__ __forceinline__ float rcbrt_fast(float x)
{
	float y = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (-1.0f / 3.0f));
	const float e = __builtin_fmaf(-x * y, y * y, 1.0f);
	return __builtin_fmaf(y * (1.0f / 3.0f), e, y);
}

// ---- STRICT-flavour division with a SHARED reciprocal (round 4) ----
// What an fp64 `a / b` compiles to on gfx950 (the IEEE-correct expansion; hp_engine.s, any STRICT kernel):
//     d0 = v_div_scale(b, b, a)          d1, vcc = v_div_scale(a, b, a)
//     r  = v_rcp(d0);  e = fma(-d0, r, 1);  r = fma(r, e, r);  e = fma(-d0, r, 1);  r = fma(r, e, r)
//     q0 = d1 * r;  rem = fma(-d0, q0, d1);  q = v_div_fmas(rem, r, q0 | vcc);  result = v_div_fixup(q, b, a)
// -- 11 VALU instructions, one of them (v_rcp_f64) at a quarter of the rate, and the two v_div_scale only ever change an
// operand whose exponent is within ~50 of the ends of the range.  Where several quotients share ONE denominator -- qx/h and
// qy/h, the two HLL middle-state fluxes over s_R - s_L, the eight flux differences over dx -- the STRICT kernels now refine
// the reciprocal ONCE and run only the numerator's part per quotient, with the SAME instructions in the SAME order:
//     recip_of(b)      : r as above from b itself; flags b unless it is a normal number below 2^1000 (1/b normal)
//     div_shared(a, .) : d1, vcc = v_div_scale(a, b, a) and d0 = v_div_scale(b, b, a) as the compiler issues them; the lane is
//                        flagged if d0 differs from b -- the hardware rescaled the DENOMINATOR for this numerator (exponents
//                        768 apart, a quotient near the ends of the range), so r is not the reciprocal the compiler's
//                        sequence would have refined; then q0 = d1 r, rem = fma(-b, q0, d1), q = v_div_fmas(rem, r, q0 | vcc),
//                        v_div_fixup(q, b, a).  A rescaled NUMERATOR (below about 2^-960: denormal discharges) and a zero
//                        numerator (v_div_scale answers NaN, v_div_fixup delivers the signed zero) go through the hardware's
//                        own path, v_div_fmas and v_div_fixup, exactly as in the compiler's sequence.
// An unflagged lane has executed the compiler's own sequence on the compiler's own operands (d0 == b, hence the same r; the same
// d1 and vcc): the same bits by construction, not by an error analysis -- and tools/divcheck compares 2e9 random pairs over every
// exponent window plus a table of special operands on the GPU: no unflagged quotient differs from the plain division.  A flagged
// lane raises a word that makes the HOST re-run the batch with the plain divisions (below).  7 VALU per quotient after the first
// instead of 11 + the quarter-rate v_rcp.  fp32 keeps the plain division (its expansion is short).
template <typename T> struct Recip { T b, r; bool bad; };
// PLAIN = true: the plain `a / b` everywhere (the twin a flagged wavefront re-runs, and what fp32 always takes)
template <bool PLAIN> __device__ __forceinline__ Recip<double> recip_of(const double b)
{
	Recip<double> d;
	d.b = b; d.r = 0.0; d.bad = false;
	if (!PLAIN) {
		d.bad = !__builtin_amdgcn_class(b, 0x108) || !(__builtin_fabs(b) < 0x1p1000);      // 0x108: -normal | +normal
		double r = __builtin_amdgcn_rcp(b);
		double e = __builtin_fma(-b, r, 1.0);
		r = __builtin_fma(r, e, r);
		e = __builtin_fma(-b, r, 1.0);
		d.r = __builtin_fma(r, e, r);
	}
	return d;
}
// `use` == false: the caller selects the quotient away (b may hold anything there); such a lane never raises the flag
template <bool PLAIN> __device__ __forceinline__ double div_shared(const double a, const Recip<double>& d, bool& bad, const bool use = true)
{
	if (PLAIN) return a / d.b;
	bool vcc, vcc_den;
	const double d1 = __builtin_amdgcn_div_scale(a, d.b, true, &vcc);
	const double d0 = __builtin_amdgcn_div_scale(a, d.b, false, &vcc_den);
	// the one thing the shared reciprocal cannot follow: v_div_scale rescaling the DENOMINATOR for this numerator (exponents 768
	// apart, a quotient beyond the range) -- `<>` is the ordered not-equal, so the NaN it answers for a zero numerator does not
	// flag.  (A cheaper test -- "vcc set although d1 == a", without the second v_div_scale -- misses pairs for which the
	// hardware rescales BOTH operands: 2.5 M of 2.5e8 over the whole exponent range, tools/divcheck; it was not faster either.)
	bad = bad || ((d.bad || __builtin_islessgreater(d0, d.b)) && use);
	const double q0 = d1 * d.r;
	const double rem = __builtin_fma(-d.b, q0, d1);
	return __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(rem, d.r, q0, vcc), d.b, a);
}
// the end of a group of quotients: raise the domain's SLOT_SPEC word if any lane of the wavefront flagged.  The branch is
// predicted not taken and its body -- one store -- sits out of line: nothing of the group has to stay alive for it.
template <bool PLAIN, typename T> __device__ __forceinline__ void spec_raise(const bool bad, T* word)
{
	if (!PLAIN && sizeof(T) == 8)
		if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad) != 0, 0)) *word = T(1);
}
template <bool PLAIN> __device__ __forceinline__ Recip<float> recip_of(const float b) { return Recip<float>{b, 0.0f, false}; }
template <bool PLAIN> __device__ __forceinline__ float div_shared(const float a, const Recip<float>& d, bool&, const bool = true) { return a / d.b; }
// a0 / b and a1 / b
template <bool PLAIN, typename T>
__device__ __forceinline__ void div2_strict(const T a0, const T a1, const T b, const bool use, T& q0, T& q1, bool& bad)
{
	const Recip<T> d = recip_of<PLAIN>(b);
	q0 = div_shared<PLAIN>(a0, d, bad, use);
	q1 = div_shared<PLAIN>(a1, d, bad, use);
}
// Where the flag goes: out of the kernel.  Every in-kernel fall-back was built and measured (profiles/r04i_*): a branch behind
// every quotient that patches a flagged lane up at once (in line: forty-odd jumps over cold code per cell, S-DAM 0.544 -> 0.643
// ms; out of line with __builtin_expect: 1.07 ms -- the cold divisions' live ranges cost the kernel its registers); one
// wave-uniform test per GROUP of statements re-running the group's plain twin (1.80 ms: scratch); a plain division behind a real
// function call (1.25 ms); and a per-lane flag ORed across the whole tile (0.68 ms: SGPR spills in the row loop) -- against 0.426
// ms for the same kernel with no fall-back at all.  So a group of quotients only raises ONE WORD in the domain's slot block
// (spec_raise: a store, out of line, behind a branch that is never taken), and the fall-back is the HOST's: hp_step_batch takes
// a device-side snapshot in front of a speculative batch, and whoever next enters the library finds the word raised, puts the
// snapshot back and re-runs the batch with the plain kernels (hp_engine.hip: spec_begin / spec_resolve).
template <bool STRICT, typename T> constexpr bool shares_reciprocals() { return STRICT && sizeof(T) == 8; }

// a*b + c: two roundings in STRICT (what the reference computes without -cl-mad-enable), one FMA in FAST
template <bool STRICT, typename T> __device__ __forceinline__ T mad(const T a, const T b, const T c)
{
	return STRICT ? (a * b + c) : fma_(a, b, c);
}

// One side of a face as the cell owner prepares it: raw state + the cell-centre velocities of
// reconstructInterface (CLSchemeGodunov.clc:39-61): u0 = (Z - zb < VERY_SMALL) ? 0 : Qx / (Z - zb).
template <typename T> struct Side { T eta, zb, qx, qy, u0, v0; };

template <bool STRICT, bool PLAIN, typename T>
__device__ __forceinline__ Side<T> make_side_impl(T z, T qx, T qy, T zb, T vs, T* spec_word)
{
	bool bad = false;
	Side<T> s;
	s.eta = z; s.zb = zb; s.qx = qx; s.qy = qy;
	const T h0 = z - zb;
	if (STRICT) {
		T u, v;
		div2_strict<PLAIN>(qx, qy, h0, !(h0 < vs), u, v, bad);
		spec_raise<PLAIN>(bad, spec_word);
		s.u0 = (h0 < vs ? T(0) : u);
		s.v0 = (h0 < vs ? T(0) : v);
	} else {
		const T inv = (h0 < vs ? T(0) : rcp_fast(h0));
		s.u0 = qx * inv;
		s.v0 = qy * inv;
	}
	return s;
}
template <bool STRICT, typename T>                       // the plain flavour (everything but the speculative K1 / K2 instantiations)
__device__ __forceinline__ Side<T> make_side(T z, T qx, T qy, T zb, T vs)
{
	return make_side_impl<STRICT, true>(z, qx, qy, zb, vs, (T*)nullptr);
}

// What a cell needs from one of its four faces: the flux vector (mass, x-momentum, y-momentum), the
// neighbour-side reconstructed level and bed for the bed-slope source term (CLSchemeGodunov.clc:268-269,
// :323-325) and whether a stopping condition fired (:107-130).
template <typename T> struct FaceFlux { T f0, fx, fy, eta_nb, zb_nb; bool stop; };

// Shift-dependent tail of a face solve for the DRY-DRY case (CLSolverHLLC.clc:45-61): pressure-like term
// only, left bed on both sides (Q4).  `s` is the vertical shift of the cell the face is finished for.
template <int AXIS, typename T>
__device__ __forceinline__ FaceFlux<T> finish_dry(const T etaL, const T etaR, const T zbm, const T s,
                                                  const bool own_left, const bool stop)
{
	const T half_g = T(0.5) * gravity<T>();
	const T a = etaL - s, b = etaR - s, zb = zbm - s;
	const T p = half_g * (((a + b) / 2) * ((a + b) / 2) - zb * (a + b));
	FaceFlux<T> o;
	o.f0 = T(0);
	o.fx = (AXIS == AXIS_X ? p : T(0));
	o.fy = (AXIS == AXIS_Y ? p : T(0));
	o.eta_nb = own_left ? b : a;
	o.zb_nb = zb;
	o.stop = stop;
	return o;
}

// Everything of the HLLC solve that does not depend on the shift.
template <typename T> struct FaceCore {
	T etaL, etaR, zbm;            // reconstructed levels before the shift, common bed
	T unL, unR, utL, utR;         // normal / tangential velocities
	T qnL, qnR, qtL, qtR;         // normal / tangential discharges
	T sL, sR, sLsR, inv_ds;       // wave speeds
	bool bLeft, bRight, bMid1;
};

// Shift-dependent tail of a STRICT face solve for one of the two cells (the FAST flavour finishes a face once, see face_solve)
template <int AXIS, bool PLAIN, typename T>
__device__ __forceinline__ FaceFlux<T> finish_wet(const FaceCore<T> k, const Recip<T> rds, const T s, const bool own_left, const bool stop, bool& bad)
{
	// (k and rds by VALUE, the tangential velocity picked into a local first: with references the speculative instantiations kept
	// the face core in scratch memory -- the select between two of its fields had become a load from a selected address)
	const T ut_mid = k.bMid1 ? k.utL : k.utR;
	const T half_g = T(0.5) * gravity<T>();
	const T a = k.etaL - s, b = k.etaR - s, zb = k.zbm - s;
	// normal-momentum flux of each side in free-surface form, left bed on both sides (:146-157, Q4)
	const T fnL = k.unL * k.qnL + half_g * (a * a - 2 * zb * a);
	const T fnR = k.unR * k.qnR + half_g * (b * b - 2 * zb * b);
	T f0, fn, ft;
	if (k.bLeft)       { f0 = k.qnL; fn = fnL; ft = k.unL * k.qtL; }
	else if (k.bRight) { f0 = k.qnR; fn = fnR; ft = k.unR * k.qtR; }
	else {
		// HLL middle state (:200-224); both quotients (and both finishes of the face) share s_R - s_L
		const T n1 = k.sR * k.qnL - k.sL * k.qnR + k.sLsR * (b - a);
		const T n2 = k.sR * fnL - k.sL * fnR + k.sLsR * (k.qnR - k.qnL);
		const T f1m = div_shared<PLAIN>(n1, rds, bad), f2m = div_shared<PLAIN>(n2, rds, bad);   // (rds: s_R - s_L and its refined reciprocal)
		f0 = f1m; fn = f2m; ft = f1m * ut_mid;
	}
	FaceFlux<T> o;
	o.f0 = f0;
	o.fx = (AXIS == AXIS_X ? fn : ft);
	o.fy = (AXIS == AXIS_X ? ft : fn);
	o.eta_nb = own_left ? b : a;
	o.zb_nb = zb;
	o.stop = stop;
	return o;
}

template <typename T> struct FacePair { FaceFlux<T> forL, forR; bool wet; };   // wet (FAST, wave-uniform): every lane has water on both sides of its face

// ---- FAST flavour of a face solve: DEPTH FORM (round 5) ----------------------------------------------------------------------
// The reference evaluates the pressure-like part of the normal-momentum flux in free-surface form on SHIFTED levels
// (CLSolverHLLC.clc:146-157 on the output of CLSchemeGodunov.clc:84-97, :134-139): with a = eta_L - s, b = eta_R - s, zb = zbm - s
// (s = the vertical shift of the cell the face is finished for, left bed on both sides, quirk Q4),
//     F_n,L = u_L q_L + g/2 (a^2 - 2 zb a)          and, because a - zb = eta_L - zbm = h_L whatever s is,
//           = u_L q_L + g/2 h_L^2 - g/2 Z^2,        Z = zbm - s                         (s = max(0, zbm - eta_own)),
// likewise on the right with h_R.  The HLL average of two fluxes that both carry -g/2 Z^2 is the average minus g/2 Z^2, and the mass
// flux only sees b - a = h_R - h_L: EVERYTHING the solver computes is independent of the shift except that one additive constant.
// The cell update (CLSchemeGodunov.clc:323-336) then adds the bed-slope source g/2 (eta_E' + eta_W')(zb_E' - zb_W') on the
// neighbour-side shifted values eta' = h_nb + Z, zb' = Z, and
//     (F_E - F_W) + g/2 (eta_E' + eta_W')(Z_E - Z_W) = (F~_E - F~_W) - g/2 (Z_E^2 - Z_W^2) + g/2 (h_E + h_W + Z_E + Z_W)(Z_E - Z_W)
//                                                     = (F~_E - F~_W) + g/2 (h_E + h_W)(Z_E - Z_W),           F~ = F + g/2 Z^2:
// the constants cancel against the source term identically.  So the FAST face hands over the shift-free flux F~ (ONE value for
// both cells: the second finish of round 2 and the per-cell C(s) of round 3 are gone), the neighbour-side DEPTH h_nb in
// FaceFlux::eta_nb and Z in FaceFlux::zb_nb, and godunov_update's FAST branch -- the same expression as before on fields that
// now mean (h_nb, Z) -- delivers the same sum.  Same mathematics, different rounding: the terms of size g/2 zb^2 that the
// free-surface form adds and takes away again (5e4 at a 100 m bed: the reference's own noise floor, what its -cl-mad-enable and
// strict builds differ by) never arise.  STRICT keeps the reference's statements.
//
// One formula for the whole fan: with s_L' = min(s_L, 0), s_R' = max(s_R, 0) the HLL expression
//     F = (s_R' F_L - s_L' F_R + s_L' s_R' (U_R - U_L)) / (s_R' - s_L')
// is F_L for s_L >= 0 and F_R for s_R <= 0 (:174-198) -- no region selects, no wave-uniform test for them; the tangential
// momentum is the mass flux times the upwind tangential velocity in all three regions (:206-224; sign(s_M) = sign(F_1) as in
// round 3, and in the supercritical regions u_n q_t = q_n u_t with q_n of the sign the region implies).  A face between two dry
// cells (:45-61: zero mass and tangential flux, pressure of the mean level) goes through the same formula: its depths are below
// VERY_SMALL, so F~ is below g/2 1e-20 and F_1 below 1e-14 -- against the reference's exact zeros both differences vanish under
// the update's VERY_SMALL flush (:340-348); the celerity is clamped away from zero so that s_R' - s_L' never is.
// A wavefront whose lanes all have water on both sides skips what only a dry side can need: zeroed velocities (:87-92), dry-side
// wave speeds (:129-140), the shift and the stopping conditions (:101-133); a wet-wet lane gets the same bits either way.
template <typename T> __device__ __forceinline__ T celerity_fast(const T gh);
template <> __device__ __forceinline__ double celerity_fast(const double gh) { return sqrt_fast_pos(gh); }    // (gh >= 1e-300: face_solve_fast keeps the depths above 1.02e-301)
template <> __device__ __forceinline__ float  celerity_fast(const float gh)  { return __builtin_amdgcn_sqrtf(__builtin_fmaxf(gh, 1e-30f)); }   // (fp32 keeps its own clamp: below)

template <int AXIS, typename T>
__device__ __forceinline__ FacePair<T> face_solve_fast(const Side<T>& L, const Side<T>& R, const T vs)
{
	const T g = gravity<T>(), half_g = T(0.5) * g;
	const T zbm = fmax_raw(L.zb, R.zb);
	// :84-97.  fp64: the two clamps -- the depths at zero, the celerities away from zero -- are one, and only a wavefront with a dry
	// side somewhere pays for it: a dry side's depth is held at 1.02e-301 instead of 0, so that g h >= 1e-300 under the root;
	// everything such a depth multiplies underflows to the zero it stood for.  (fp32 keeps the two clamps on every face: the merged
	// form measured 1-3 % SLOWER there, profiles/r05fm_three_builds_ab.txt -- its instructions are single-issue either way and the
	// longer dry-side block costs more than four of them save.)
	constexpr bool MERGED = sizeof(T) == 8;
	T hL = L.eta - zbm, hR = R.eta - zbm;
	if (!MERGED) { hL = fmax_(hL, T(0)); hR = fmax_(hR, T(0)); }
	T unL = (AXIS == AXIS_X ? L.u0 : L.v0), unR = (AXIS == AXIS_X ? R.u0 : R.v0);
	T utL = (AXIS == AXIS_X ? L.v0 : L.u0), utR = (AXIS == AXIS_X ? R.v0 : R.u0);
	const bool all_wet = wave_all(hL > vs && hR > vs);
	const bool dryL = hL < vs, dryR = hR < vs;
	T ZL = zbm, ZR = zbm;
	bool stopL = false, stopR = false;
	if (!all_wet) {
		if (MERGED) { hL = fmax_(hL, T(1.02e-301)); hR = fmax_(hR, T(1.02e-301)); }
		// stopping conditions (:101-133) on the cell-centre velocities and raw discharges
		const T qrawL = (AXIS == AXIS_X ? L.qx : L.qy), qrawR = (AXIS == AXIS_X ? R.qx : R.qy);
		const bool shared = (hR <= vs && unL < T(0)) || (hL <= vs && unR > T(0));
		stopL = shared || (hL <= vs && qrawL > T(0));
		stopR = shared || (hR <= vs && qrawR < T(0));
		unL = dryL ? T(0) : unL; utL = dryL ? T(0) : utL;                                 // :87-92
		unR = dryR ? T(0) : unR; utR = dryR ? T(0) : utR;
		// zbm - shift, the shift formed as the reference forms it (:85-86, :136-139): mathematically min(zbm, eta_own), but where
		// a level lies far below the common bed -- the 9999.9 m walls of every closed edge -- fl(zbm - fl(zbm - eta)) is eta + delta
		// with delta up to half an ulp OF THE WALL (1e-12 m), and that delta is the reference's dominant rounding term there: its
		// wall pressure is -g/2 (eta + delta)^2, the momentum balance of a cell at rest against a wall comes out as -g/2 delta h
		// instead of zero, survives the VERY_SMALL flush, and its SIGN decides whether the stopping conditions (:101-133) fire on
		// the cells along a wall in a flow that is symmetric about them.  Two more instructions keep that term, sign and size.
		ZL = zbm - fmax_(zbm - L.eta, T(0)); ZR = zbm - fmax_(zbm - R.eta, T(0));
	}
	const T qnL = hL * unL, qnR = hR * unR;                                               // :99-102
	const T ghL = g * hL, ghR = g * hR;
	const T aL = celerity_fast(ghL), aR = celerity_fast(ghR);                             // :103-106
	// :123-126 through the two Riemann invariants (halved): u* = (unL + unR)/2 + aL - aR = pL + mR, a* = |(unL - unR)/4 + (aL + aR)/2|
	// = |pL - mR| / 2 -- five operations for the reference's seven
	const T pL = fma_(T(0.5), unL, aL), mR = fma_(T(0.5), unR, -aR);
	const T u_star = pL + mR;
	const T a_star = fabs_(T(0.5) * (pL - mR));
	T sL = fmin_(unL - aL, u_star - a_star), sR = fmax_(unR + aR, u_star + a_star);       // :129-140
	if (!all_wet) {
		sL = dryL ? fma_(T(-2), aR, unR) : sL;
		sR = dryR ? fma_(T(2), aL, unL) : sR;
	}
	sL = fmin0_raw(sL); sR = fmax0_raw(sR);                                               // the whole fan (:174-198)
	const T inv_ds = rcp_fast(sR - sL);
	const T sLsR = sL * sR;
	const T fnL = fma_(unL, qnL, (T(0.5) * ghL) * hL), fnR = fma_(unR, qnR, (T(0.5) * ghR) * hR);   // u q + g/2 h^2
	const T f0 = fma_(sLsR, hR - hL, fma_(sR, qnL, -(sL * qnR))) * inv_ds;                // :200-203
	const T fn = fma_(sLsR, qnR - qnL, fma_(sR, fnL, -(sL * fnR))) * inv_ds;
	const T ft = f0 * ((f0 >= T(0)) ? utL : utR);                                         // :206-224
	(void)half_g;
	FacePair<T> o;
	o.forL.f0 = f0; o.forL.fx = (AXIS == AXIS_X ? fn : ft); o.forL.fy = (AXIS == AXIS_X ? ft : fn);
	o.forL.eta_nb = hR; o.forL.zb_nb = ZL; o.forL.stop = stopL;
	o.forR.f0 = f0; o.forR.fx = o.forL.fx; o.forR.fy = o.forL.fy;
	o.forR.eta_nb = hL; o.forR.zb_nb = ZR; o.forR.stop = stopR;
	o.wet = all_wet;
	return o;
}

// Solve the face between cell L (west/south) and cell R (east/north).
//   forL : the face as cell L sees it (its E or N face; "own" = left,  ucDirection < DOMAIN_DIR_S)
//   forR : the face as cell R sees it (its W or S face; "own" = right)
// Reference: reconstructInterface (CLSchemeGodunov.clc:27-159) + riemannSolver (CLSolverHLLC.clc:27-248).
template <int AXIS, bool STRICT, bool WANT_L, bool WANT_R, bool PLAIN, typename T>
__device__ __forceinline__ FacePair<T> face_solve_impl(const Side<T>& L, const Side<T>& R, const T vs, T* spec_word)
{
	if (!STRICT) return face_solve_fast<AXIS>(L, R, vs);            // (depth form, above; everything below is the exact flavour)
	bool bad = false;
	const T g = gravity<T>();

	// ---- reconstruction (:84-97) ----
	T zbm, hL, hR, shL, shR;
	if (STRICT) {
		zbm = (L.zb > R.zb ? L.zb : R.zb);
		hL = (L.eta - zbm > T(0) ? (L.eta - zbm) : T(0));
		hR = (R.eta - zbm > T(0) ? (R.eta - zbm) : T(0));
		shL = zbm - L.eta; if (shL < T(0)) shL = T(0);        // shift when the LEFT cell is own (:85-86)
		shR = zbm - R.eta; if (shR < T(0)) shR = T(0);        // shift when the RIGHT cell is own
	} else {                                                  // same values through v_max_f64
		zbm = fmax_(L.zb, R.zb);
		hL = fmax_(L.eta - zbm, T(0));
		hR = fmax_(R.eta - zbm, T(0));
		shL = fmax_(zbm - L.eta, T(0));
		shR = fmax_(zbm - R.eta, T(0));
	}
	const T etaL = hL + zbm, etaR = hR + zbm;
	const T qxL = hL * L.u0, qyL = hL * L.v0;
	const T qxR = hR * R.u0, qyR = hR * R.v0;

	// ---- stopping conditions (:101-133): first test is direction specific, the other two shared ----
	const T vnL = (AXIS == AXIS_X ? L.u0 : L.v0), vnR = (AXIS == AXIS_X ? R.u0 : R.v0);
	const T qrawL = (AXIS == AXIS_X ? L.qx : L.qy), qrawR = (AXIS == AXIS_X ? R.qx : R.qy);
	const bool shared = (hR <= vs && vnL < T(0)) || (hL <= vs && vnR > T(0));
	const bool stopL = shared || (hL <= vs && qrawL > T(0));      // N / E case
	const bool stopR = shared || (hR <= vs && qrawR < T(0));      // S / W case

	// ---- HLLC ----
	FaceFlux<T> oL, oR;
	if (hL < vs && hR < vs) {
		// ---- the reference's statements in the reference's order ----
		oL = finish_dry<AXIS>(etaL, etaR, zbm, shL, true, stopL);
		oR = finish_dry<AXIS>(etaL, etaR, zbm, shR, false, stopR);
	} else {
		// velocities recomputed from the reconstructed discharges as the reference does (:87-92); the two quotients of a side
		// share its depth (div2_strict)
		T uL, vL, uR, vR;
		div2_strict<PLAIN>(qxL, qyL, hL, !(hL < vs), uL, vL, bad);
		div2_strict<PLAIN>(qxR, qyR, hR, !(hR < vs), uR, vR, bad);
		uL = (hL < vs ? T(0) : uL); vL = (hL < vs ? T(0) : vL);
		uR = (hR < vs ? T(0) : uR); vR = (hR < vs ? T(0) : vR);
		FaceCore<T> k;
		k.etaL = etaL; k.etaR = etaR; k.zbm = zbm;
		k.unL = (AXIS == AXIS_X ? uL : vL); k.unR = (AXIS == AXIS_X ? uR : vR);           // dVel   (:95-98)
		k.utL = (AXIS == AXIS_X ? vL : uL); k.utR = (AXIS == AXIS_X ? vR : uR);           // tangential velocity
		k.qnL = (AXIS == AXIS_X ? qxL : qyL); k.qnR = (AXIS == AXIS_X ? qxR : qyR);       // dDis   (:99-102)
		k.qtL = (AXIS == AXIS_X ? qyL : qxL); k.qtR = (AXIS == AXIS_X ? qyR : qxR);
		const T aL = sqrt_(g * hL), aR = sqrt_(g * hR);                                   // dA     (:103-106)

		// two-rarefaction star state and wave speeds (:123-142)
		const T a_avg = (aL + aR) / 2;
		const T tmp = a_avg + (k.unL - k.unR) / 4;
		const T u_star = (k.unL + k.unR) / 2 + aL - aR;
		const T hstar = div_shared<PLAIN>(tmp * tmp, recip_of<PLAIN>(g), bad);            // (tmp * tmp) / g: a constant denominator
		const T a_star = sqrt_(g * hstar);
		T sL, sR;
		if (hL < vs) sL = k.unR - 2 * aR;
		else         sL = (((k.unL - aL) > (u_star - a_star)) ? (u_star - a_star) : (k.unL - aL));
		if (hR < vs) sR = k.unL + 2 * aL;
		else         sR = (((k.unR + aR) < (u_star + a_star)) ? (u_star + a_star) : (k.unR + aR));
		k.sL = sL; k.sR = sR; k.sLsR = sL * sR;

		// region selection (:174-177); NaN wave speeds fall through to "right" exactly as in the reference
		const T sm_num = sL * hR * (k.unR - sR) - sR * hL * (k.unL - sL);
		const T sm_den = hR * (k.unR - sR) - hL * (k.unL - sL);
		const bool sm_nonneg = (sm_num / sm_den) >= T(0);
		k.bLeft = sL >= T(0);
		k.bMid1 = sL < T(0) && sR >= T(0) && sm_nonneg;
		const bool bMid2 = sL < T(0) && sR >= T(0) && !k.bMid1;
		k.bRight = !k.bLeft && !k.bMid1 && !bMid2;
		k.inv_ds = T(0);
		const Recip<T> rds = recip_of<PLAIN>(sR - sL);

		oL = finish_wet<AXIS, PLAIN>(k, rds, shL, true, stopL, bad);
		// Where the two cells of a face see the same vertical shift -- everywhere except where a cell's level lies below its
		// neighbour's bed -- the second finish would repeat the first on the same operands: same statements, same bits.  It is
		// skipped when that holds for every lane that is here (round 4; FAST has had its own form of this since round 1).
		if (WANT_L && WANT_R && wave_all(shL == shR)) {
			oR = oL;
			oR.eta_nb = k.etaL - shR;                                                     // finish_wet's `a` (own cell = right)
			oR.stop = stopR;
		} else {
			oR = finish_wet<AXIS, PLAIN>(k, rds, shR, false, stopR, bad);
		}
	}
	if (STRICT) spec_raise<PLAIN>(bad, spec_word);
	FacePair<T> out;
	out.forL = oL;
	out.forR = oR;
	out.wet = false;
	return out;
}
template <int AXIS, bool STRICT, bool WANT_L, bool WANT_R, typename T>
__device__ __forceinline__ FacePair<T> face_solve(const Side<T>& L, const Side<T>& R, const T vs)
{
	return face_solve_impl<AXIS, STRICT, WANT_L, WANT_R, true>(L, R, vs, (T*)nullptr);
}

// The face between two DRY cells as the cell on its right / north side sees it: exactly what face_solve returns in
// .forR for a lane whose two reconstructed depths are below VERY_SMALL (its dry-dry branch, CLSolverHLLC.clc:45-61) --
// the same statements in the same order, without the wet solve around them.  A cell's depth bounds its face depth
// (h_face = max(eta - max(zbL, zbR), 0) <= eta - zb), so two dry cells always take that branch.  Used by K1 on rows
// that are dry land throughout, where this face is the only product of the row anyone may still need.
template <int AXIS, bool STRICT, typename T>
__device__ __forceinline__ FaceFlux<T> face_dry_for_right(const Side<T>& L, const Side<T>& R, const T vs)
{
	// FAST (depth form, round 5) has no dry-dry branch: a face between dry cells goes through the one formula, here as wherever
	// else the face might be solved -- a face has the same bits whichever tile, and whichever loop of a tile, produces it
	if (!STRICT) return face_solve_fast<AXIS>(L, R, vs).forR;
	T zbm, hL, hR, shR;
	if (STRICT) {
		zbm = (L.zb > R.zb ? L.zb : R.zb);
		hL = (L.eta - zbm > T(0) ? (L.eta - zbm) : T(0));
		hR = (R.eta - zbm > T(0) ? (R.eta - zbm) : T(0));
		shR = zbm - R.eta; if (shR < T(0)) shR = T(0);
	} else {
		zbm = fmax_(L.zb, R.zb);
		hL = fmax_(L.eta - zbm, T(0));
		hR = fmax_(R.eta - zbm, T(0));
		shR = fmax_(zbm - R.eta, T(0));
	}
	const T etaL = hL + zbm, etaR = hR + zbm;
	const T vnL = (AXIS == AXIS_X ? L.u0 : L.v0), vnR = (AXIS == AXIS_X ? R.u0 : R.v0);
	const T qrawR = (AXIS == AXIS_X ? R.qx : R.qy);
	const bool shared = (hR <= vs && vnL < T(0)) || (hL <= vs && vnR > T(0));
	const bool stopR = shared || (hR <= vs && qrawR < T(0));
	return finish_dry<AXIS>(etaL, etaR, zbm, shR, false, stopR);
}

// Point-implicit Manning friction (Schemes/CLFriction.clc:26-72)
template <bool STRICT, bool PLAIN, typename T>
__device__ __forceinline__ void friction(T& qx, T& qy, const T z, const T zb, const T n, const T dt, const T vs, bool& bad)
{
	static_assert(STRICT, "the FAST flavour goes through friction_fast (below)");
	const T g = gravity<T>();
	const T q = sqrt_(qx * qx + qy * qy);
	const T h = z - zb;
	if (h < vs || q < vs) return;
	const T cf  = (g * n * n) / pow13_(h);                           // CLFriction.clc:43
	// (-cf) / (h h) is -(cf / (h h)) exactly: one quotient serves the four places it is written (:44-50); the two `/ q` share q,
	// and the two clamps -qx / dt, -qy / dt share dt (:52-65) -- div2_strict, same bits as the separate divisions
	const T k   = cf / (h * h);
	const T sfx = (-k) * qx * q;
	const T sfy = (-k) * qy * q;
	T tx, ty;
	div2_strict<PLAIN>(dt * k * (2 * (qx * qx) + (qy * qy)), dt * k * ((qx * qx) + 2 * (qy * qy)), q, true, tx, ty, bad);
	const T dx  = T(1.0) + tx;
	const T dy  = T(1.0) + ty;
	T fx = sfx / dx, fy = sfy / dy;
	// The clamps (:52-65) keep friction from REVERSING the flow: F is replaced by -q / dt where |F| exceeds |q| / dt.  The implicit
	// denominators make |F| dt < |q| mathematically, so the clamp only ever acts through rounding, at |F| dt ~ |q|.  A lane whose
	// |F| dt (1 + 2^-30), evaluated in floating point (two roundings: relative 2^-52), stays below |q| has |F| < (|q| / dt)(1 - 2^-31)
	// < RN(|q| / dt) in exact arithmetic: neither `F < m` nor `F > m` can hold, whatever the quotient's last bit is; F = +-0 cannot
	// be clamped either (m has the opposite sign of q or is a zero).  Where every active lane of the wavefront is such a lane the two
	// quotients -qx / dt, -qy / dt are not needed -- the same bits without them (round 4; NaN or overflow fail the test and take
	// the reference's statements).
	const T margin = T(1) + (sizeof(T) == 8 ? T(9.313225746154785e-10) : T(6.103515625e-05));   // 1 + 2^-30 (fp64), 1 + 2^-14 (fp32)
	const bool free_x = (fabs_(fx) * dt) * margin < fabs_(qx) || fx == T(0);
	const bool free_y = (fabs_(fy) * dt) * margin < fabs_(qy) || fy == T(0);
	if (!wave_all(free_x && free_y)) {
		T mx, my;
		div2_strict<PLAIN>(-qx, -qy, dt, true, mx, my, bad);
		if (qx >= T(0)) { if (fx < mx) fx = mx; } else { if (fx > mx) fx = mx; }
		if (qy >= T(0)) { if (fy < my) fy = my; } else { if (fy > my) fy = my; }
	}
	qx = qx + dt * fx;
	qy = qy + dt * fy;
}

// FAST flavour of the same term, straight-line.  With A = dt g n^2 h^(-7/3) and Q = |q| the reference's update is
//   qx <- qx + max(dt Fx, -qx)  (qx >= 0; mirrored otherwise),  dt Fx = -qx A Q^2 / (Q + A (2 qx^2 + qy^2)),
// i.e. qx <- qx (1 - min(fx, 1)) with fx = A Q^2 / (Q + A (2 qx^2 + qy^2)) >= 0: "friction can stop the flow, not reverse it"
// (:52-65) is a clamp of fx at one -- no sign cases, no division by Q.  Cells the reference skips (h or Q below
// VERY_SMALL, :36-37) take fx = 0, and a wavefront without an active cell skips the arithmetic (still water).
template <typename T>
__device__ __forceinline__ void friction_fast(T& qx, T& qy, const T z, const T zb, const T n, const T dtg, const T vs)      // dtg = dt g
{
	const T h = z - zb;
	const T q = sqrt_fast(fma_(qx, qx, qy * qy));
	const bool active = !(h < vs || q < vs);
	if (!wave_any(active)) return;
	const T rc = rcbrt_fast(h);                              // h^(-1/3)
	const T rc2 = rc * rc, rc4 = rc2 * rc2;
	const T A = (dtg * (n * n)) * (rc4 * rc2 * rc);          // dt g n^2 h^(-7/3)
	const T qx2 = qx * qx, qy2 = qy * qy;                    // (recomputed here: two multiplications against two live registers)
	const T aq2 = A * (qx2 + qy2);
	const T denx = fma_(A, fma_(T(2), qx2, qy2), q), deny = fma_(A, fma_(T(2), qy2, qx2), q);
	const T rxy = rcp_fast(denx * deny);                     // 1/denx = rxy * deny, 1/deny = rxy * denx
	T fx = fmin_(aq2 * (rxy * deny), T(1)), fy = fmin_(aq2 * (rxy * denx), T(1));
	fx = active ? fx : T(0);
	fy = active ? fy : T(0);
	qx = fma_(-qx, fx, qx);
	qy = fma_(-qy, fy, qy);
}

template <bool STRICT = true, typename T>
__device__ __forceinline__ T small_to_zero(const T v, const T vs)
{
	if (STRICT) return ((v > T(0) && v < vs) || (v < T(0) && v > -vs)) ? T(0) : v;      // CLSchemeGodunov.clc:340-348
	return (fabs_(v) < vs) ? T(0) : v;                                   // same set of values, one compare
}

// Godunov cell update from its four finished faces (CLSchemeGodunov.clc:321-383).
// Returns the new state; `c` holds {Z, Zmax, Qx, Qy} of the cell before the step.
// (Probe: the pair kernel's look at "does this cell end the step dry, and changed" -- quirk Q3's stamps, hp_kernels.hpp: PairAux.  It
// sits where the old state is still in registers and everything that decides is known: behind the flux step of the level, in front of
// the friction step -- a cell that ends the step dry has no friction term (CLFriction.clc:36-37), so its discharge after the flux step
// is final.  `want` (per lane) joins the vote of the block that zeroes the discharge at a stopping condition: no branch of its own on
// the march's path -- one cost the pair kernel a tenth of its speed, masked stores 6 % -- and `hit` runs inside that block.  NoProbe:
// nothing, and the vote is the stopping conditions' alone.)
struct NoProbe {
	template <typename... A> __device__ __forceinline__ bool want(A&&...) const { return false; }
	template <typename... A> __device__ __forceinline__ void hit(A&&...) const {}
};
template <typename W, typename H> struct Probe2 { W want; H hit; };
template <typename W, typename H> __device__ __forceinline__ Probe2<W, H> make_probe(W w, H h) { return Probe2<W, H>{w, h}; }
// lam = -(dt / dx), dtg = dt g: wave-uniform and the same on every row of a launch.  `given`: the kernel formed them once and holds them
// in scalar registers (uniform_value); otherwise the update forms them itself and the compiler keeps them where it likes.
template <typename T> struct StepConsts { T lam, dtg; bool given; };
template <bool STRICT, bool CLAMP_FIRST, bool PLAIN, typename T, typename Probe = NoProbe>
__device__ __forceinline__ State4<T> godunov_update_impl(State4<T> c, const T zb, const T n, const T dt,
                                                         const FaceFlux<T>& fN, const FaceFlux<T>& fE,
                                                         const FaceFlux<T>& fS, const FaceFlux<T>& fW,
                                                         const T dx, const T inv_dx, const T vs, const bool with_friction, T* spec_word,
                                                         const Probe& probe = Probe(), const StepConsts<T> pre = StepConsts<T>{T(0), T(0), false})
{
	bool bad = false;
	const T g = gravity<T>();
	T d0, d2, d3;
	if (STRICT) {
		// bed-slope source from the neighbour-side reconstructed values (:323-325)
		if (inv_dx != T(0)) {
			// STRICT is handed inv_dx only where dx is a POWER OF TWO (Params::inv_dx_pow2; the usual 0.5 / 1 / 2 / 4 m rasters):
			// x / 2^k and x * 2^-k are the correctly rounded value of the same real number -- the same bits in every case,
			// denormal results included -- so the eight divisions by dx (== dy) are eight multiplications (wave-uniform branch)
			const T sx = -1 * g * ((fE.eta_nb + fW.eta_nb) / 2) * ((fE.zb_nb - fW.zb_nb) * inv_dx);
			const T sy = -1 * g * ((fN.eta_nb + fS.eta_nb) / 2) * ((fN.zb_nb - fS.zb_nb) * inv_dx);
			d0 = (fE.f0 - fW.f0) * inv_dx + (fN.f0 - fS.f0) * inv_dx - T(0);      // :328-336
			d2 = (fE.fx - fW.fx) * inv_dx + (fN.fx - fS.fx) * inv_dx - sx;
			d3 = (fE.fy - fW.fy) * inv_dx + (fN.fy - fS.fy) * inv_dx - sy;
		} else {
		// eight quotients over the same dx (== dy): one refined reciprocal, hoisted out of the row loop by the compiler
		const Recip<T> rdx = recip_of<PLAIN>(dx);
		const T sx = -1 * g * ((fE.eta_nb + fW.eta_nb) / 2) * div_shared<PLAIN>(fE.zb_nb - fW.zb_nb, rdx, bad);
		const T sy = -1 * g * ((fN.eta_nb + fS.eta_nb) / 2) * div_shared<PLAIN>(fN.zb_nb - fS.zb_nb, rdx, bad);
		d0 = div_shared<PLAIN>(fE.f0 - fW.f0, rdx, bad) + div_shared<PLAIN>(fN.f0 - fS.f0, rdx, bad) - T(0);      // :328-336
		d2 = div_shared<PLAIN>(fE.fx - fW.fx, rdx, bad) + div_shared<PLAIN>(fN.fx - fS.fx, rdx, bad) - sx;
		d3 = div_shared<PLAIN>(fE.fy - fW.fy, rdx, bad) + div_shared<PLAIN>(fN.fy - fS.fy, rdx, bad) - sy;
		}
	} else {
		// dx == dy: the sums stay multiplied by dx -- the flush below compares them with VERY_SMALL dx, the update multiplies by
		// dt / dx (both wave-uniform and formed once per launch) -- instead of eight divisions
		const T hg = T(0.5) * g;
		const T sxd = hg * (fE.eta_nb + fW.eta_nb) * (fE.zb_nb - fW.zb_nb);      // = -sx * dx
		const T syd = hg * (fN.eta_nb + fS.eta_nb) * (fN.zb_nb - fS.zb_nb);
		d0 = (fE.f0 - fW.f0) + (fN.f0 - fS.f0);
		d2 = (fE.fx - fW.fx) + (fN.fx - fS.fx) + sxd;
		d3 = (fE.fy - fW.fy) + (fN.fy - fS.fy) + syd;
	}
	const T flush = STRICT ? vs : vs * dx;
	d0 = small_to_zero<STRICT>(d0, flush);
	d2 = small_to_zero<STRICT>(d2, flush);
	d3 = small_to_zero<STRICT>(d3, flush);

	const bool stop = fN.stop || fE.stop || fS.stop || fW.stop;
	if (STRICT) {
		T qx0 = c.qx, qy0 = c.qy;
		if (stop) { qx0 = T(0); qy0 = T(0); }                        // :351-355
		const T z1  = c.z - dt * d0;                                 // :358-360
		const T qx1 = qx0 - dt * d2;
		const T qy1 = qy0 - dt * d3;
		if (wave_any(probe.want(c, z1))) probe.hit(c, z1, qx1, qy1);
		c.z = z1; c.qx = qx1; c.qy = qy1;
		if (with_friction) friction<true, PLAIN>(c.qx, c.qy, c.z, zb, n, dt, vs, bad);        // :362-372
		spec_raise<PLAIN>(bad, spec_word);
	} else {
		T qx0 = c.qx, qy0 = c.qy;
		// (StepConsts: the two wave-uniform, row-invariant products of the step handed in as SCALAR register pairs by a kernel that has no
		// vector registers to keep them in -- the pair kernel, round 6)
		const T lam = pre.given ? pre.lam : -(dt * inv_dx);
		const T z1 = fma_(lam, d0, c.z);
		if (wave_any(stop || probe.want(c, z1))) {               // a stopping condition needs a dry side
			asm volatile("");                                    // (a real branch: if-converted, its eight selects ran on every row)
			qx0 = stop ? T(0) : qx0; qy0 = stop ? T(0) : qy0;
			probe.hit(c, z1, fma_(lam, d2, qx0), fma_(lam, d3, qy0));
		}
		c.z = z1;
		c.qx = fma_(lam, d2, qx0);
		c.qy = fma_(lam, d3, qy0);
		if (with_friction) friction_fast(c.qx, c.qy, c.z, zb, n, pre.given ? pre.dtg : dt * g, vs);
	}

	if (CLAMP_FIRST) {                                               // mch_2nd_cacheNone order (MUSCL :791-796)
		if (c.z - zb < vs) c.z = zb;
		if (c.z > c.zmax && c.zmax > T(-9990.0)) c.zmax = c.z;
	} else {
		if (c.z > c.zmax && c.zmax > T(-9990.0)) c.zmax = c.z;       // :375-376
		if (c.z - zb < vs) c.z = zb;                                 // :379-380
	}
	return c;
}
template <bool STRICT, bool CLAMP_FIRST = false, typename T>
__device__ __forceinline__ State4<T> godunov_update(State4<T> c, const T zb, const T n, const T dt,
                                                    const FaceFlux<T>& fN, const FaceFlux<T>& fE,
                                                    const FaceFlux<T>& fS, const FaceFlux<T>& fW,
                                                    const T dx, const T inv_dx, const T vs, const bool with_friction)
{
	return godunov_update_impl<STRICT, CLAMP_FIRST, true>(c, zb, n, dt, fN, fE, fS, fW, dx, inv_dx, vs, with_friction, (T*)nullptr);
}

// =================================================================================================
//  MUSCL-Hancock (Schemes/CLSchemeMUSCLHancock.clc, Schemes/Limiters/CLSlopeLimiterMINMOD.clc)
// =================================================================================================
template <typename T> struct Face4 { T z, h, qx, qy; };              // extrapolated face state {Z, H, Qx, Qy}
template <typename T> struct Faces { Face4<T> n, e, s, w; };
template <typename T> struct Raw { T z, zmax, qx, qy, zb; };         // what a neighbour contributes

// calculateLimitedSlope (CLSlopeLimiterMINMOD.clc:51-72), MINBEE_BETA = 1 -> MINMOD
template <bool STRICT, typename T>
__device__ __forceinline__ T limited_slope(const T l, const T c, const T r)
{
	const T dL = c - l, dR = r - c;
	if (STRICT) {
		const T rr = (fabs_(dL) <= T(0) ? T(0) : (dR / dL));
		return fmax_(fmax_(T(0), fmin_(T(1) * rr, T(1))), fmin_(rr, T(1))) * dL;
	}
	// clamp(dR/dL, 0, 1) * dL without the division: 0 if the differences disagree in sign, else the smaller one
	return (dL * dR <= T(0)) ? T(0) : ((fabs_(dR) < fabs_(dL)) ? dR : dL);
}

// slopeLimiter (:26-46): no slopes on a wet-dry front
template <bool STRICT, typename T>
__device__ __forceinline__ Face4<T> limiter(const Raw<T>& l, const Raw<T>& c, const Raw<T>& r, const T vs)
{
	Face4<T> s;
	if ((l.z - l.zb) < vs || (r.z - r.zb) < vs) { s.z = s.h = s.qx = s.qy = T(0); return s; }
	s.z  = limited_slope<STRICT>(l.z, c.z, r.z);
	s.h  = limited_slope<STRICT>(l.z - l.zb, c.z - c.zb, r.z - r.zb);
	s.qx = limited_slope<STRICT>(l.qx, c.qx, r.qx);
	s.qy = limited_slope<STRICT>(l.qy, c.qy, r.qy);
	return s;
}

// faceExtrapolate (CLSchemeMUSCLHancock.clc:389-403); `c.h` is unused: H is rebuilt from Z - zb
template <bool STRICT, typename T>
__device__ __forceinline__ Face4<T> face_extrapolate(const T zb, const Face4<T>& c, const Face4<T>& slope, const T coef)
{
	Face4<T> f;
	f.z = mad<STRICT>(coef, slope.z, c.z);
	f.h = mad<STRICT>(coef, slope.h, c.z - zb);
	f.qx = mad<STRICT>(coef, slope.qx, c.qx);
	f.qy = mad<STRICT>(coef, slope.qy, c.qy);
	return f;
}

// mch_1st (:301-382): limited slopes, face extrapolation, half-step evolution, re-extrapolation
template <bool STRICT, bool PLAIN, typename T>
__device__ __forceinline__ Faces<T> muscl_predict_impl(const Raw<T>& c, const Raw<T>& n, const Raw<T>& e, const Raw<T>& s,
                                                       const Raw<T>& w, const T dt, const T dx, const T inv_dx, const T vs,
                                                       const bool nb_y_is_bed, bool& quiet_row, bool& same_row, T* spec_word)
{
	bool bad = false;
	const T g = gravity<T>();
	Face4<T> cc; cc.z = c.z; cc.h = c.z - c.zb; cc.qx = c.qx; cc.qy = c.qy;           // :333
	// FAST does not produce the faces of a quiet row at all (quiet_row: they ARE the cell state, and the kernel builds what it
	// needs from the cell state itself): pre-setting them cost sixteen v_mov_b64 in front of every predictor, dead on a live row
	Faces<T> f;
	if (STRICT) { f.n = cc; f.e = cc; f.s = cc; f.w = cc; }
	// :325-330.  `pNeigData*.y` is the neighbour's BED in the reference's default configuration (kCachePrediction: the LDS tile
	// of mch_1st_cachePrediction holds {Z, bed, Qx, Qy}, :201, :232-239) and its Zmax in mch_1st_cacheNone (:109-125);
	// nb_y_is_bed is wave-uniform (HP_QUIRK_MUSCL_NEIGHBOUR_Y_IS_BED)
	const bool nb_null = nb_y_is_bed ? (n.zb <= T(-9998.0) || e.zb <= T(-9998.0) || s.zb <= T(-9998.0) || w.zb <= T(-9998.0))
	                                 : (n.zmax <= T(-9998.0) || e.zmax <= T(-9998.0) || s.zmax <= T(-9998.0) || w.zmax <= T(-9998.0));
	const bool first = (c.z - c.zb < T(1E-5)) || nb_null;

	// Cheapest sufficient test for a quiet row (see below): a cell whose four neighbours carry exactly its own level, bed
	// and discharges has all differences zero, hence all limited slopes zero -- sixteen compares instead of the eight
	// limiter evaluations that would discover the same thing.
	// (round 5: on live water ONE lane whose northern neighbour stands at another level settles both votes -- the other fifteen
	// compares are only made where no such lane exists)
	bool same = false;
	quiet_row = same_row = false;
	if (!wave_any(!first && n.z != c.z)) {
		same = n.z == c.z && n.zb == c.zb && n.qx == c.qx && n.qy == c.qy &&
		       e.z == c.z && e.zb == c.zb && e.qx == c.qx && e.qy == c.qy &&
		       s.z == c.z && s.zb == c.zb && s.qx == c.qx && s.qy == c.qy &&
		       w.z == c.z && w.zb == c.zb && w.qx == c.qx && w.qy == c.qy;
		quiet_row = wave_all(first || same);
		same_row = wave_all(same);              // every lane's neighbourhood is one state (used by the kernel's inert-row test)
	}
	if (quiet_row) return f;

	Face4<T> sx, sy;                                                                    // :343-346
	sx.z = sx.h = sx.qx = sx.qy = sy.z = sy.h = sy.qx = sy.qy = T(0);
	if (STRICT) {
		if (!first) { sx = limiter<STRICT>(w, c, e, vs); sy = limiter<STRICT>(s, c, n, vs); }
	} else {
		// FAST (round 4): every reason a limited slope is zero -- a first-order cell (:325-330), a dry neighbour on the axis
		// (slopeLimiter :26-46), differences that disagree in sign (MINMOD) -- goes into ONE select per slope; the slope itself is
		// the smaller difference with the sign of the left one (v_min_f64 with |.| modifiers + v_bfi_b32), no second select.
		// Was: three layers of selects on doubles, two v_cndmask_b32 each (64 per cell; now 16).
		const bool zx = first || (w.z - w.zb) < vs || (e.z - e.zb) < vs, zy = first || (s.z - s.zb) < vs || (n.z - n.zb) < vs;
		auto mm = [](const T l, const T m, const T r, const bool zero) {
			const T dL = m - l, dR = r - m;
			const T v = __builtin_copysign(fmin_(fabs_(dL), fabs_(dR)), dL);
			return (zero || !(dL * dR > T(0))) ? T(0) : v;
		};
		const T hw = w.z - w.zb, hc = c.z - c.zb, he = e.z - e.zb, hs = s.z - s.zb, hn = n.z - n.zb;
		sx.z = mm(w.z, c.z, e.z, zx);   sx.h = mm(hw, hc, he, zx);   sx.qx = mm(w.qx, c.qx, e.qx, zx);   sx.qy = mm(w.qy, c.qy, e.qy, zx);
		sy.z = mm(s.z, c.z, n.z, zy);   sy.h = mm(hs, hc, hn, zy);   sy.qx = mm(s.qx, c.qx, n.qx, zy);   sy.qy = mm(s.qy, c.qy, n.qy, zy);
	}
	// Quiescent water (wave-uniform fast path).  Where every limited slope of every lane is zero -- still or uniformly
	// moving water over a flat bed, wet/dry fronts (no slopes there, :26-46), first-order cells -- the four face states
	// equal the cell state, the face fluxes cancel pairwise, the bed-slope term is g/2 (2 eta)(zb - zb) = 0, the half step
	// changes nothing and the re-extrapolated faces are the cell state again: exactly, in both arithmetic flavours
	// (+-0.5 * 0 + c = c; x - x = 0; 0 * anything finite = 0).  The rest of the predictor is skipped for the whole
	// wavefront.  Flood models are mostly such water (and dry land) most of the time.
	bool flat = sx.z == T(0) && sy.z == T(0);                                           // staged like `same`
	quiet_row = false;
	if (wave_all(first || flat)) {
		flat = flat && sx.h == T(0) && sx.qx == T(0) && sx.qy == T(0) && sy.h == T(0) && sy.qx == T(0) && sy.qy == T(0);
		quiet_row = wave_all(first || flat);   // wave-uniform: every lane's four faces ARE its cell state
	}
	if (quiet_row) return f;
	// STRICT leaves a first-order lane here (its faces are the cell state, :333-339).  FAST lets it run along: its slopes are
	// zero, so its four faces ARE the cell state (c + 0.5 * 0), the face fluxes cancel pairwise (x - x), the bed-slope term is
	// g/2 (2 z)(zb - zb) = 0, and with the half-step increments forced to zero below the re-extrapolated faces are the cell state
	// again -- the same values as the early exit, without the sixteen selects on doubles that merging the two paths cost
	if (STRICT && first) return f;

	T d0, d2, d3;
	if (!STRICT) {
		// ---- FAST: the half step in DEPTH FORM (round 5; see face_solve_fast for the same step on the corrector's faces) ----
		// estimateFluxVectorX / Y (:420-471) evaluate u q + g/2 (z^2 - 2 (z - h) z) on the extrapolated face states and evolveCellState
		// (:476-526) adds the face-bed source g/2 (z_E + z_W)(zb_E - zb_W), zb = z - h.  With z^2 - 2 zb z = h^2 - zb^2 the pressure
		// difference and the source of one axis add up to g/2 (h_E + h_W)(z_E - z_W) -- and the two faces of a cell are the cell
		// state +- half a slope, so h_E + h_W = 2 h and z_E - z_W = slope.z: the whole term is g h slope.z.  The level faces of the
		// first extrapolation are not needed at all, nor the eight products of the free-surface form.
		// Face depths: a lane that gets here with slopes has h >= 1e-5 (:325-330) and MINMOD keeps a face value between the cell and
		// the midpoint to its neighbour, so h_face >= h / 2 > VERY_SMALL: the reference's `h < VERY_SMALL ? 0 : q / h` never takes
		// its first arm on a lane whose increments are used; first-order lanes ride along and have their increments forced to
		// zero below (a SELECT: whatever 1 / 0 made of them does not propagate).  Four quotients, one reciprocal.
		const T hc = cc.h;
		const T hE = fma_(T(0.5), sx.h, hc), hW = fma_(T(-0.5), sx.h, hc), hN = fma_(T(0.5), sy.h, hc), hS = fma_(T(-0.5), sy.h, hc);
		const T qxE = fma_(T(0.5), sx.qx, c.qx), qxW = fma_(T(-0.5), sx.qx, c.qx), qyE = fma_(T(0.5), sx.qy, c.qy), qyW = fma_(T(-0.5), sx.qy, c.qy);
		const T qxN = fma_(T(0.5), sy.qx, c.qx), qxS = fma_(T(-0.5), sy.qx, c.qx), qyN = fma_(T(0.5), sy.qy, c.qy), qyS = fma_(T(-0.5), sy.qy, c.qy);
		T iE, iW, iN, iS;
		if (sizeof(T) == 8) {
			const T pEW = hE * hW, pNS = hN * hS;
			const T r = rcp_fast(pEW * pNS);                       // >= (h/2)^4 >= 6e-23 on the lanes that count
			const T rEW = r * pNS, rNS = r * pEW;                  // 1 / (hE hW), 1 / (hN hS)
			iE = rEW * hW; iW = rEW * hE; iN = rNS * hS; iS = rNS * hN;
		} else {
			iE = rcp_fast(hE); iW = rcp_fast(hW); iN = rcp_fast(hN); iS = rcp_fast(hS);     // (fp32: v_rcp_f32 is one instruction, the product would underflow)
		}
		const T uE = qxE * iE, uW = qxW * iW, vN = qyN * iN, vS = qyS * iS;
		const T gh = g * hc;
		d0 = ((qxE - qxW) + (qyN - qyS)) * inv_dx;
		d2 = fma_(gh, sx.z, fma_(uE, qxE, -(uW * qxW)) + fma_(vN, qxN, -(vS * qxS))) * inv_dx;
		d3 = fma_(gh, sy.z, fma_(uE, qyE, -(uW * qyW)) + fma_(vN, qyN, -(vS * qyS))) * inv_dx;
		d0 = (first || fabs_(d0) < vs) ? T(0) : d0;               // (first-order lanes: no half step, whatever their fluxes came to)
		d2 = (first || fabs_(d2) < vs) ? T(0) : d2;
		d3 = (first || fabs_(d3) < vs) ? T(0) : d3;
		const T mh = T(-0.5) * dt;
		cc.z = fma_(mh, d0, cc.z); cc.qx = fma_(mh, d2, cc.qx); cc.qy = fma_(mh, d3, cc.qy);
	} else {
	f.n = face_extrapolate<STRICT>(c.zb, cc, sy, T(+0.5));                                         // :349-352
	f.e = face_extrapolate<STRICT>(c.zb, cc, sx, T(+0.5));
	f.s = face_extrapolate<STRICT>(c.zb, cc, sy, T(-0.5));
	f.w = face_extrapolate<STRICT>(c.zb, cc, sx, T(-0.5));

	// estimateFluxVectorX / Y (:420-471): FSL form with zb = Z - H
	auto press = [&](const Face4<T>& a) { return T(0.5) * g * ((a.z * a.z) - 2 * (a.z - a.h) * a.z); };
	auto vel = [&](const T q, const T h) { return h < vs ? T(0) : q / h; };
	const T uE = vel(f.e.qx, f.e.h), uW = vel(f.w.qx, f.w.h), vN = vel(f.n.qy, f.n.h), vS = vel(f.s.qy, f.s.h);
	const T FE0 = f.e.qx, FE1 = mad<STRICT>(uE, f.e.qx, press(f.e)), FE2 = uE * f.e.qy;
	const T FW0 = f.w.qx, FW1 = mad<STRICT>(uW, f.w.qx, press(f.w)), FW2 = uW * f.w.qy;
	const T FN0 = f.n.qy, FN1 = vN * f.n.qx, FN2 = mad<STRICT>(vN, f.n.qy, press(f.n));
	const T FS0 = f.s.qy, FS1 = vS * f.s.qx, FS2 = mad<STRICT>(vS, f.s.qy, press(f.s));

	// evolveCellState (:476-526)
	{
		if (inv_dx != T(0)) {                                              // dx a power of two: products, as in godunov_update
			const T s1 = -1 * g * ((f.e.z + f.w.z) / 2) * (((f.e.z - f.e.h) - (f.w.z - f.w.h)) * inv_dx);
			const T s2 = -1 * g * ((f.n.z + f.s.z) / 2) * (((f.n.z - f.n.h) - (f.s.z - f.s.h)) * inv_dx);
			d0 = (FE0 - FW0) * inv_dx + (FN0 - FS0) * inv_dx - T(0);
			d2 = (FE1 - FW1) * inv_dx + (FN1 - FS1) * inv_dx - s1;
			d3 = (FE2 - FW2) * inv_dx + (FN2 - FS2) * inv_dx - s2;
		} else {
		const Recip<T> rdx = recip_of<PLAIN>(dx);                          // eight quotients over dx, as in godunov_update
		const T s1 = -1 * g * ((f.e.z + f.w.z) / 2) * div_shared<PLAIN>((f.e.z - f.e.h) - (f.w.z - f.w.h), rdx, bad);
		const T s2 = -1 * g * ((f.n.z + f.s.z) / 2) * div_shared<PLAIN>((f.n.z - f.n.h) - (f.s.z - f.s.h), rdx, bad);
		d0 = div_shared<PLAIN>(FE0 - FW0, rdx, bad) + div_shared<PLAIN>(FN0 - FS0, rdx, bad) - T(0);
		d2 = div_shared<PLAIN>(FE1 - FW1, rdx, bad) + div_shared<PLAIN>(FN1 - FS1, rdx, bad) - s1;
		d3 = div_shared<PLAIN>(FE2 - FW2, rdx, bad) + div_shared<PLAIN>(FN2 - FS2, rdx, bad) - s2;
		}
		spec_raise<PLAIN>(bad, spec_word);
	}
	d0 = small_to_zero<STRICT>(d0, vs);
	d2 = small_to_zero<STRICT>(d2, vs);
	d3 = small_to_zero<STRICT>(d3, vs);
	cc.z  = cc.z  - T(0.5) * dt * d0;
	cc.qx = cc.qx - T(0.5) * dt * d2;
	cc.qy = cc.qy - T(0.5) * dt * d3;
	}

	f.n = face_extrapolate<STRICT>(c.zb, cc, sy, T(+0.5));                                         // :376-379
	f.e = face_extrapolate<STRICT>(c.zb, cc, sx, T(+0.5));
	f.s = face_extrapolate<STRICT>(c.zb, cc, sy, T(-0.5));
	f.w = face_extrapolate<STRICT>(c.zb, cc, sx, T(-0.5));
	return f;
}
template <bool STRICT, typename T>
__device__ __forceinline__ Faces<T> muscl_predict(const Raw<T>& c, const Raw<T>& n, const Raw<T>& e, const Raw<T>& s,
                                                  const Raw<T>& w, const T dt, const T dx, const T inv_dx, const T vs,
                                                  const bool nb_y_is_bed, bool& quiet_row, bool& same_row)
{
	Faces<T> f = muscl_predict_impl<STRICT, true>(c, n, e, s, w, dt, dx, inv_dx, vs, nb_y_is_bed, quiet_row, same_row, (T*)nullptr);
	if (!STRICT && quiet_row) {                  // (callers of this wrapper get the faces of a quiet row as well)
		Face4<T> cc; cc.z = c.z; cc.h = c.z - c.zb; cc.qx = c.qx; cc.qy = c.qy;
		f.n = cc; f.e = cc; f.s = cc; f.w = cc;
	}
	return f;
}

// One side of a corrector face: the extrapolated face state + the RAW cell discharges the stopping conditions
// test (2nd-order reconstructInterface, CLSchemeMUSCLHancock.clc:1119-1230; note `<=` in the velocity guard)
template <bool STRICT, bool PLAIN, typename T>
__device__ __forceinline__ Side<T> side_from_face_impl(const Face4<T>& f, const T qx_raw, const T qy_raw, const T vs, T* spec_word)
{
	bool bad = false;
	Side<T> s;
	s.eta = f.z; s.zb = f.z - f.h; s.qx = qx_raw; s.qy = qy_raw;
	if (STRICT) {
		T u, v;
		div2_strict<PLAIN>(f.qx, f.qy, f.h, !(f.h <= vs), u, v, bad);
		spec_raise<PLAIN>(bad, spec_word);
		s.u0 = (f.h <= vs ? T(0) : u);
		s.v0 = (f.h <= vs ? T(0) : v);
	} else {
		const T inv = (f.h <= vs ? T(0) : rcp_fast(f.h));
		s.u0 = f.qx * inv;
		s.v0 = f.qy * inv;
	}
	return s;
}
template <bool STRICT, typename T>
__device__ __forceinline__ Side<T> side_from_face(const Face4<T>& f, const T qx_raw, const T qy_raw, const T vs)
{
	return side_from_face_impl<STRICT, true>(f, qx_raw, qy_raw, vs, (T*)nullptr);
}

// ---- partial-inertial scheme (CLSchemeInertial.clc) ----
// calculateInertialFlux (:331-378): discharge per unit width across one face from the previous discharge, the
// water-level slope and a semi-implicit Manning term, limited to Froude 0.8 (CLSchemeInertial.clh:24).
// STRICT keeps the reference's expression order (device pow stands in for the host's); FAST writes
// g d dt n^2 |q| / d^(10/3) as g dt n^2 |q| d^(-7/3) with the cube-root seed, and the two Froude tests as a clamp
// to +-0.8 d sqrt(g d).
template <bool STRICT, typename T>
__device__ __forceinline__ T inertial_flux(const T n, const T dt, const T q_prev, const T z_up, const T b_up,
                                           const T z_down, const T b_down, const T dx, const T inv_dx, const T vs)
{
	const T g = gravity<T>();
	const T d = fmax_(z_down, z_up) - fmax_(b_up, b_down);                                // :342
	T q;
	if (STRICT) {
		const T slope = (z_down - z_up) / dx;                                             // :343
		q = (q_prev - (g * d * dt * slope)) /                                             // :346-348
		    (T(1.0) + g * d * dt * n * n * fabs_(q_prev) / pow103_(d));
		const T c = sqrt_(g * d);
		if (q > T(0) && ((fabs_(q) / d) / c) > T(0.8)) q = d * c * T(0.8);                // :351-356
		if (q < T(0) && ((fabs_(q) / d) / c) > T(0.8)) q = T(0) - d * c * T(0.8);
	} else {
		const T slope = (z_down - z_up) * inv_dx;
		const T rc = rcbrt_fast(d);                                                       // d^(-1/3)
		const T rc2 = rc * rc, rc4 = rc2 * rc2;
		const T gdt = g * dt;
		const T den = fma_(gdt * (n * n) * fabs_(q_prev), rc4 * rc2 * rc, T(1));
		q = fma_(-gdt * d, slope, q_prev) * rcp_fast(den);
		const T lim = T(0.8) * d * sqrt_fast(g * d);
		q = fmin_(fmax_(q, -lim), lim);
	}
	if (d < vs) q = T(0);                                                                 // :371-372
	return q;
}

// ine_cacheDisabled's update (:144-163): new face discharges are stored in the cell, the level moves by their
// divergence.
template <bool STRICT, typename T>
__device__ __forceinline__ State4<T> inertial_update(const State4<T>& c, const T zb, const T dt, const T qN, const T qE,
                                                     const T qS, const T qW, const T dx, const T inv_dx, const T vs)
{
	State4<T> o;
	o.qx = qW; o.qy = qS;
	const T delta = STRICT ? (qE - qW + qN - qS) / dx : (qE - qW + qN - qS) * inv_dx;      // :148-149
	o.z = STRICT ? c.z + dt * delta : fma_(dt, delta, c.z);                                // :152
	o.zmax = (o.z > c.zmax) ? o.z : c.zmax;                                                // :155-156
	if (o.z - zb < vs) o.z = zb;                                                           // :159-160
	return o;
}

// Wave speed of one cell for the CFL reduction (CLDynamicTimestep.clc:185-216)
template <bool STRICT, bool PLAIN, typename T>
__device__ __forceinline__ T cfl_speed_impl(const T z, const T zmax, const T qx, const T qy, const T zb, const T qs,
                                            const bool simplified, T* spec_word)
{
	bool bad = false;
	const T h = z - zb;
	if (h > qs && zmax > T(-9999.0)) {
		if (simplified)                                       // :205-210 (partial-inertial scheme)
			return STRICT ? sqrt_(gravity<T>() * h) : sqrt_fast(gravity<T>() * h);
		if (STRICT) {
			// max(|qx / h| + a, |qy / h| + a) (:222-236) with ONE quotient: IEEE rounding is monotonic and symmetric in sign, so for
			// h > 0 the larger magnitude has the larger (or equal) rounded quotient, |RN(q / h)| = RN(|q| / h), and adding the same
			// `a` keeps the order -- the value the reference's two quotients, two negations and final select deliver, bit for bit
			// (a NaN discharge takes the same way through the compare as through the reference's `vx < vy`).
			(void)bad; (void)spec_word;
			const T ax = fabs_(qx), ay = fabs_(qy);
			const T m = (ax < ay) ? ay : ax;
			return m / h + sqrt_(gravity<T>() * h);
		} else {
			// one reciprocal square root for both terms (round 5): y = h^(-1/2) from the v_rsq seed and one Newton step (relative
			// error 4e-15; fp32: the seed is 1 ulp), then 1 / h = y y and sqrt(g h) = sqrt(g) h y -- 13 issue slots against the
			// 24 of a refined reciprocal plus a refined square root; the timestep follows the maximum to a relative 1e-14
			const T m = fmax_(fabs_(qx), fabs_(qy));
			if (sizeof(T) == 8) {
				const double hd = (double)h;
				double y = __builtin_amdgcn_rsq(hd);
				const double e = __builtin_fma(-(hd * y), y, 1.0);
				y = __builtin_fma(0.5 * y, e, y);
				return (T)__builtin_fma((double)m, y * y, 3.1320919526731650 * (hd * y));      // sqrt(9.81), correctly rounded
			}
			const float y = __builtin_amdgcn_rsqf((float)h);
			return (T)__builtin_fmaf((float)m, y * y, 3.1320920f * ((float)h * y));
		}
	}
	return T(0);
}
template <bool STRICT, typename T>
__device__ __forceinline__ T cfl_speed(const T z, const T zmax, const T qx, const T qy, const T zb, const T qs,
                                       const bool simplified = false)
{
	return cfl_speed_impl<STRICT, true>(z, zmax, qx, qy, zb, qs, simplified, (T*)nullptr);
}

// ---- order-preserving unsigned image of a non-negative float, for an exact atomic max ----
__device__ __forceinline__ void atomic_max_nonneg(double* slot, double v)
{
	atomicMax(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)__double_as_longlong(v));
}
__device__ __forceinline__ void atomic_max_nonneg(float* slot, float v)
{
	atomicMax(reinterpret_cast<unsigned int*>(slot), __float_as_uint(v));
}

// layout of the CFL slot block (elements of T): [0] running max | [SLOT_SAVED] last max used | [SLOT_EDGE..+1] ring maxima
constexpr int SLOT_SAVED = 32, SLOT_EDGE = 64;          // separate 256-B apart so line [0] only ever sees atomics
// strips: this rank's own last maximum (what it contributes when its buffer was not priced anew), the all-reduced maximum
// (the collective writes it next to, not over, the local one), and the batch-start handshake's 2 x 8 elements
constexpr int SLOT_LOCAL = 33, SLOT_GLOBAL = 34, SLOT_HANDSHAKE = 112;
// raised (non-zero) by a SPECULATIVE flux launch in which some quotient fell outside what the shared-reciprocal division covers
constexpr int SLOT_SPEC = 100;
// raised (non-zero, sticky) by a launch's tail block that gave up waiting for a flux block's word (hp_kernels.hpp: launch_tail)
constexpr int SLOT_TAIL_ERR = 104;
constexpr int SLOT_BDY = 96;            // cfl_slot[SLOT_BDY] != 0: the next iteration's area boundaries are already in its source buffer (K1 FUSED, pairs)
constexpr int SLOT_M1 = 40;             // iteration pairs on domains with area boundaries: the maximum of the primary buffer WITH the next iteration's boundaries applied

__device__ __forceinline__ double atomic_exchange_zero(double* slot)
{
	return __longlong_as_double((long long)atomicExch(reinterpret_cast<unsigned long long*>(slot), 0ull));
}
__device__ __forceinline__ float atomic_exchange_zero(float* slot)
{
	return __uint_as_float(atomicExch(reinterpret_cast<unsigned int*>(slot), 0u));
}

template <typename T>
__device__ __forceinline__ T wave_max(T v)
{
	// `>`-style max: NaN never wins (CLDynamicTimestep.clc:215-216)
	for (int off = 32; off > 0; off >>= 1) {
		const T o = __shfl_xor(v, off, 64);
		v = (o > v) ? o : v;
	}
	return v;
}

} // namespace hp
