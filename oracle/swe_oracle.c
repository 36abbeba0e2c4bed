/*
 * oracle/swe_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See swe_oracle.h.
 *
 * Scalar restatement of the reference's device kernels and of the host-side iteration graph.
 * Expression order follows the reference statement by statement so that, built with
 * -ffp-contract=off, results are bit-identical to the reference's kernel source compiled for the
 * host the same way (oracle/_ref, checked in tests/test_oracle_vs_ref.py).
 * Citations are relative to /root/reference/src/.
 */
#include "swe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* pow() is the one operation of the path whose result is implementation defined (an OpenCL device's pow is good to
 * 16 ulp); the reference only ever asks for x^(1/3), x^(10/3) and x^2.  The oracle, the reference build's shim and the
 * STRICT HIP kernels all take the first two from this header (correctly rounded cube root, IEEE basic operations only:
 * same bits on every platform), and x^2 as x*x. */
#include "../hipims-ocl_amd/csrc/hp_crmath.h"

#ifdef ORC_FP32
#define R_SQRT  sqrtf
#define R_POW13(x)  hp_cr_cbrtf(x)
#define R_POW103(x) hp_cr_pow103f(x)
#define R_FABS  fabsf
#define R_FMAX  fmaxf
#define R_FMIN  fminf
#define R_FLOOR floorf
#define R_FMOD  fmodf
#else
#define R_SQRT  sqrt
#define R_POW13(x)  hp_cr_cbrt(x)
#define R_POW103(x) hp_cr_pow103(x)
#define R_FABS  fabs
#define R_FMAX  fmax
#define R_FMIN  fmin
#define R_FLOOR floor
#define R_FMOD  fmod
#endif

#define RC(x) ((real)(x))

/* OpenCL/Executors/CLUniversalHeader.clh:33 */
static const real GRAVITY = RC(9.81);

/* Schemes/CLDynamicTimestep.clh:24-29 ; Boundaries/CLBoundaries.clh:28 */
static const real TIMESTEP_EARLY_LIMIT            = RC(0.1);
static const real TIMESTEP_EARLY_LIMIT_DURATION   = RC(60.0);
static const real TIMESTEP_START_MINIMUM          = RC(1E-10);
static const real TIMESTEP_START_MINIMUM_DURATION = RC(1.0);
static const real TIMESTEP_MINIMUM                = RC(1E-10);
static const real TIMESTEP_MAXIMUM                = RC(15.0);
static const real TIMESTEP_HYDROLOGICAL           = RC(1.0);

int orc_real_bytes(void) { return (int)sizeof(real); }

/* =============================================================================================
 *  1st-order non-negative reconstruction -- Schemes/CLSchemeGodunov.clc:27-159
 *  out vectors: {Z, H, Qx, Qy, U, V, Zb, 0}
 * ========================================================================================== */
static int stop_tests(real VS, int dir, real* L, real* R, const real* sL, const real* sR)
{
	int stop = 0;
	switch (dir) {
	case ORC_DIR_N:                                                   /* :103-111 */
		if (L[1] <= VS && sL[3] > RC(0.0)) { stop++; }
		if (R[1] <= VS && L[5] < RC(0.0))  { stop++; L[5] = RC(0.0); }
		if (L[1] <= VS && R[5] > RC(0.0))  { stop++; R[5] = RC(0.0); }
		break;
	case ORC_DIR_S:                                                   /* :112-118 */
		if (R[1] <= VS && sR[3] < RC(0.0)) { stop++; }
		if (R[1] <= VS && L[5] < RC(0.0))  { stop++; L[5] = RC(0.0); }
		if (L[1] <= VS && R[5] > RC(0.0))  { stop++; R[5] = RC(0.0); }
		break;
	case ORC_DIR_E:                                                   /* :119-125 */
		if (L[1] <= VS && sL[2] > RC(0.0)) { stop++; }
		if (R[1] <= VS && L[4] < RC(0.0))  { stop++; L[4] = RC(0.0); }
		if (L[1] <= VS && R[4] > RC(0.0))  { stop++; R[4] = RC(0.0); }
		break;
	case ORC_DIR_W:                                                   /* :126-132 */
		if (R[1] <= VS && sR[2] < RC(0.0)) { stop++; }
		if (R[1] <= VS && L[4] < RC(0.0))  { stop++; L[4] = RC(0.0); }
		if (L[1] <= VS && R[4] > RC(0.0))  { stop++; R[4] = RC(0.0); }
		break;
	}
	return stop;
}

int orc_reconstruct(const orc_params* p, int dir, const real sL[4], real bL, const real sR[4], real bR,
                    real oL[8], real oR[8])
{
	const real VS = p->very_small;
	real L[8], R[8];
	real dDepthL = sL[0] - bL;                                        /* :39-40 */
	real dDepthR = sR[0] - bR;

	L[0] = sL[0]; L[1] = dDepthL; L[2] = sL[2]; L[3] = sL[3];         /* :44-52 */
	L[4] = (dDepthL < VS ? RC(0.0) : sL[2] / dDepthL);
	L[5] = (dDepthL < VS ? RC(0.0) : sL[3] / dDepthL);
	L[6] = bL; L[7] = RC(0.0);
	R[0] = sR[0]; R[1] = dDepthR; R[2] = sR[2]; R[3] = sR[3];         /* :53-61 */
	R[4] = (dDepthR < VS ? RC(0.0) : sR[2] / dDepthR);
	R[5] = (dDepthR < VS ? RC(0.0) : sR[3] / dDepthR);
	R[6] = bR; R[7] = RC(0.0);

	real dBedMaximum = (L[6] > R[6] ? L[6] : R[6]);                   /* :84 */
	real dShiftV = dBedMaximum - (dir < ORC_DIR_S ? sL : sR)[0];      /* :85 */
	if (dShiftV < RC(0.0)) dShiftV = RC(0.0);

	L[1] = (sL[0] - dBedMaximum > RC(0.0) ? (sL[0] - dBedMaximum) : RC(0.0));   /* :89-92 */
	L[0] = L[1] + dBedMaximum;
	L[2] = L[1] * L[4];
	L[3] = L[1] * L[5];
	R[1] = (sR[0] - dBedMaximum > RC(0.0) ? (sR[0] - dBedMaximum) : RC(0.0));   /* :94-97 */
	R[0] = R[1] + dBedMaximum;
	R[2] = R[1] * R[4];
	R[3] = R[1] * R[5];

	int stop = stop_tests(VS, dir, L, R, sL, sR);                     /* :101-133 */

	L[6] = dBedMaximum - dShiftV;                                     /* :136-139 */
	R[6] = dBedMaximum - dShiftV;
	L[0] -= dShiftV;
	R[0] -= dShiftV;

	memcpy(oL, L, sizeof L);
	memcpy(oR, R, sizeof R);
	return stop;
}

/* =============================================================================================
 *  HLLC approximate Riemann solver -- Solvers/CLSolverHLLC.clc:27-248
 * ========================================================================================== */
void orc_hllc(const orc_params* p, int dir, const real Lin[8], const real Rin[8], real F[4])
{
	const real VS = p->very_small;
	real L[8], R[8];
	memcpy(L, Lin, sizeof L);
	memcpy(R, Rin, sizeof R);

	/* uint2 direction vector: N/S -> (0,1), E/W -> (1,0)  (:42) */
	const unsigned nx = (dir == ORC_DIR_N || dir == ORC_DIR_S) ? 0u : 1u;
	const unsigned ny = (dir == ORC_DIR_N || dir == ORC_DIR_S) ? 1u : 0u;

	/* both sides dry (:45-61); note the LEFT bed on both (quirk Q4) */
	if (L[1] < VS && R[1] < VS) {
		F[0] = RC(0.0);
		F[1] = (real)nx * RC(0.5) * GRAVITY * (((L[0] + R[0]) / 2) * ((L[0] + R[0]) / 2) - L[6] * (L[0] + R[0]));
		F[2] = (real)ny * RC(0.5) * GRAVITY * (((L[0] + R[0]) / 2) * ((L[0] + R[0]) / 2) - L[6] * (L[0] + R[0]));
		F[3] = RC(0.0);
		return;
	}

	/* velocities recomputed from the reconstructed discharges (:87-92) */
	L[4] = (L[1] < VS ? RC(0.0) : L[2] / L[1]);
	L[5] = (L[1] < VS ? RC(0.0) : L[3] / L[1]);
	R[4] = (R[1] < VS ? RC(0.0) : R[2] / R[1]);
	R[5] = (R[1] < VS ? RC(0.0) : R[3] / R[1]);

	real dVel0 = (real)nx * L[4] + (real)ny * L[5];                   /* :95-106 */
	real dVel1 = (real)nx * R[4] + (real)ny * R[5];
	real dDis0 = (real)nx * L[2] + (real)ny * L[3];
	real dDis1 = (real)nx * R[2] + (real)ny * R[3];
	real dA0 = R_SQRT(GRAVITY * L[1]);
	real dA1 = R_SQRT(GRAVITY * R[1]);

	real a_Avg  = (dA0 + dA1) / 2;                                    /* :123-126 */
	real H_star = ((a_Avg + (dVel0 - dVel1) / 4) * (a_Avg + (dVel0 - dVel1) / 4)) / GRAVITY;
	real U_star = (dVel0 + dVel1) / 2 + dA0 - dA1;
	real A_star = R_SQRT(GRAVITY * H_star);

	real s_L, s_R, s_M;                                               /* :129-142 */
	if (L[1] < VS) {
		s_L = dVel1 - 2 * dA1;
	} else {
		s_L = (((dVel0 - dA0) > (U_star - A_star)) ? (U_star - A_star) : (dVel0 - dA0));
	}
	if (R[1] < VS) {
		s_R = dVel0 + 2 * dA0;
	} else {
		s_R = (((dVel1 + dA1) < (U_star + A_star)) ? (U_star + A_star) : (dVel1 + dA1));
	}
	s_M = (s_L * R[1] * (dVel1 - s_R) - s_R * L[1] * (dVel0 - s_L)) /
	      (R[1] * (dVel1 - s_R) - L[1] * (dVel0 - s_L));

	real FL[4], FR[4];                                                /* :146-157, left bed on the right too */
	FL[0] = dDis0;
	FL[1] = dVel0 * L[2] + (real)nx * RC(0.5) * GRAVITY * (L[0] * L[0] - 2 * L[6] * L[0]);
	FL[2] = dVel0 * L[3] + (real)ny * RC(0.5) * GRAVITY * (L[0] * L[0] - 2 * L[6] * L[0]);
	FL[3] = RC(0.0);
	FR[0] = dDis1;
	FR[1] = dVel1 * R[2] + (real)nx * RC(0.5) * GRAVITY * (R[0] * R[0] - 2 * L[6] * R[0]);
	FR[2] = dVel1 * R[3] + (real)ny * RC(0.5) * GRAVITY * (R[0] * R[0] - 2 * L[6] * R[0]);
	FR[3] = RC(0.0);

	int bLeft     = s_L >= RC(0.0);                                   /* :174-177 */
	int bMiddle_1 = s_L < RC(0.0) && s_R >= RC(0.0) && s_M >= RC(0.0);
	int bMiddle_2 = s_L < RC(0.0) && s_R >= RC(0.0) && !bMiddle_1;
	int bRight    = !bLeft && !bMiddle_1 && !bMiddle_2;

	if (bLeft)  { memcpy(F, FL, sizeof FL); return; }
	if (bRight) { memcpy(F, FR, sizeof FR); return; }

	real FM_L = (real)nx * FL[1] + (real)ny * FL[2];                  /* :200-203 */
	real FM_R = (real)nx * FR[1] + (real)ny * FR[2];
	real F1_M = (s_R * FL[0] - s_L * FR[0] + s_L * s_R * (R[0] - L[0])) / (s_R - s_L);
	real F2_M = (s_R * FM_L  - s_L * FM_R  + s_L * s_R * (dDis1 - dDis0)) / (s_R - s_L);

	if (bMiddle_1) {                                                  /* :206-224 */
		F[0] = F1_M;
		F[1] = (real)nx * F2_M + (real)ny * F1_M * L[4];
		F[2] = (real)nx * F1_M * L[5] + (real)ny * F2_M;
		F[3] = RC(0.0);
	}
	if (bMiddle_2) {
		F[0] = F1_M;
		F[1] = (real)nx * F2_M + (real)ny * F1_M * R[4];
		F[2] = (real)nx * F1_M * R[5] + (real)ny * F2_M;
		F[3] = RC(0.0);
	}
}

/* =============================================================================================
 *  Point-implicit Manning friction -- Schemes/CLFriction.clc:26-72
 * ========================================================================================== */
void orc_friction(const orc_params* p, const real s[4], real bed, real n, real dt, real out[4])
{
	const real VS = p->very_small;
	out[0] = s[0]; out[1] = s[1]; out[2] = s[2]; out[3] = s[3];

	real dQ = R_SQRT(s[2] * s[2] + s[3] * s[3]);                      /* :36-37 */
	real dDepth = s[0] - bed;
	if (dDepth < VS || dQ < VS) return;                               /* :40 */

	real dCf  = (GRAVITY * n * n) / (R_POW13(dDepth));     /* :43 */
	real dSfx = (-dCf / (dDepth * dDepth)) * s[2] * dQ;                              /* :44-45 */
	real dSfy = (-dCf / (dDepth * dDepth)) * s[3] * dQ;
	real dDx  = RC(1.0) + dt * (dCf / (dDepth * dDepth)) * (2 * (s[2] * s[2]) + (s[3] * s[3])) / dQ;   /* :46-47 */
	real dDy  = RC(1.0) + dt * (dCf / (dDepth * dDepth)) * ((s[2] * s[2]) + 2 * (s[3] * s[3])) / dQ;
	real dFx  = dSfx / dDx;                                           /* :48-49 */
	real dFy  = dSfy / dDy;

	if (s[2] >= RC(0.0)) {                                            /* :52-65 */
		if (dFx < -s[2] / dt) dFx = -s[2] / dt;
	} else {
		if (dFx > -s[2] / dt) dFx = -s[2] / dt;
	}
	if (s[3] >= RC(0.0)) {
		if (dFy < -s[3] / dt) dFy = -s[3] / dt;
	} else {
		if (dFy > -s[3] / dt) dFy = -s[3] / dt;
	}

	out[2] = s[2] + dt * dFx;                                         /* :68-69 */
	out[3] = s[3] + dt * dFy;
}

/* =============================================================================================
 *  MINMOD limiter -- Schemes/Limiters/CLSlopeLimiterMINMOD.clc:26-72 (MINBEE_BETA = 1, .clh:23)
 * ========================================================================================== */
real orc_limited_slope(real dLeft, real dCenter, real dRight)
{
	const real MINBEE_BETA = RC(1.0);
	real dRegionL = dCenter - dLeft;                                  /* :64-65 */
	real dRegionR = dRight - dCenter;
	real dR = (R_FABS(dRegionL) <= RC(0.0) ? RC(0.0) : (dRegionR / dRegionL));   /* :68 */
	real dPhi = R_FMAX(R_FMAX(RC(0.0), R_FMIN(MINBEE_BETA * dR, RC(1.0))), R_FMIN(dR, MINBEE_BETA)) * dRegionL;
	return dPhi;
}

void orc_limiter(const orc_params* p, const real sL[4], const real sC[4], const real sR[4],
                 real bL, real bC, real bR, real out[4])
{
	const real VS = p->very_small;
	if ((sL[0] - bL) < VS || (sR[0] - bR) < VS) {                     /* :38-39 */
		out[0] = out[1] = out[2] = out[3] = RC(0.0);
		return;
	}
	out[0] = orc_limited_slope(sL[0], sC[0], sR[0]);                  /* :41-44 */
	out[1] = orc_limited_slope(sL[0] - bL, sC[0] - bC, sR[0] - bR);
	out[2] = orc_limited_slope(sL[2], sC[2], sR[2]);
	out[3] = orc_limited_slope(sL[3], sC[3], sR[3]);
}

/* =============================================================================================
 *  MUSCL-Hancock predictor -- Schemes/CLSchemeMUSCLHancock.clc:301-526
 *  face vectors: {Z, H, Qx, Qy}
 * ========================================================================================== */
static void face_extrapolate(real bed, const real c[4], const real slope[4], real coef, real out[4])
{                                                                     /* :389-403 */
	out[0] = c[0] + coef * slope[0];
	out[1] = c[1] + coef * slope[1];
	out[2] = c[2] + coef * slope[2];
	out[3] = c[3] + coef * slope[3];
	out[1] = (c[0] - bed) + coef * slope[1];
}

static void flux_vector_x(real VS, const real f[4], real F[3])
{                                                                     /* :420-443 */
	real u = (f[1] < VS ? RC(0.0) : f[2] / f[1]);
	F[0] = f[2];
	F[1] = u * f[2] + RC(0.5) * GRAVITY * ((f[0] * f[0]) - 2 * (f[0] - f[1]) * f[0]);
	F[2] = u * f[3];
}

static void flux_vector_y(real VS, const real f[4], real F[3])
{                                                                     /* :448-471 */
	real v = (f[1] < VS ? RC(0.0) : f[3] / f[1]);
	F[0] = f[3];
	F[1] = v * f[2];
	F[2] = v * f[3] + RC(0.5) * GRAVITY * ((f[0] * f[0]) - 2 * (f[0] - f[1]) * f[0]);
}

static real small_to_zero(real v, real VS)
{                                                                     /* e.g. CLSchemeGodunov.clc:340-348 */
	if ((v > RC(0.0) && v < VS) || (v < RC(0.0) && v > -VS)) return RC(0.0);
	return v;
}

static void evolve_cell(const orc_params* p, real dt, real c[4], const real fN[4], const real fE[4],
                        const real fS[4], const real fW[4], const real FN[3], const real FE[3],
                        const real FS[3], const real FW[3])
{                                                                     /* :476-526 */
	const real VS = p->very_small, DX = p->dx, DY = p->dx;
	real Sx = RC(0.0);
	real Sy = -1 * GRAVITY * ((fE[0] + fW[0]) / 2) * (((fE[0] - fE[1]) - (fW[0] - fW[1])) / DX);
	real Sz = -1 * GRAVITY * ((fN[0] + fS[0]) / 2) * (((fN[0] - fN[1]) - (fS[0] - fS[1])) / DY);

	real dx_ = (FE[0] - FW[0]) / DX + (FN[0] - FS[0]) / DY - Sx;
	real dz_ = (FE[1] - FW[1]) / DX + (FN[1] - FS[1]) / DY - Sy;
	real dw_ = (FE[2] - FW[2]) / DX + (FN[2] - FS[2]) / DY - Sz;
	dx_ = small_to_zero(dx_, VS);
	dz_ = small_to_zero(dz_, VS);
	dw_ = small_to_zero(dw_, VS);

	c[0] = c[0] - RC(0.5) * dt * dx_;
	c[2] = c[2] - RC(0.5) * dt * dz_;
	c[3] = c[3] - RC(0.5) * dt * dw_;
}

int orc_mch_1st(const orc_params* p, real dt, const real states[20], const real beds[5], real faces[16])
{                                                                     /* :301-382 */
	const real VS = p->very_small;
	const real *sC = states, *sN = states + 4, *sE = states + 8, *sS = states + 12, *sW = states + 16;
	const real bC = beds[0], bN = beds[1], bE = beds[2], bS = beds[3], bW = beds[4];
	real *fN = faces, *fE = faces + 4, *fS = faces + 8, *fW = faces + 12;
	int first = 0;

	/* pNeigData*.y: the neighbour's Zmax as mch_1st_cacheNone passes it (:109-125), its BED as the default
	 * mch_1st_cachePrediction does (the LDS tile is {Z, bed, Qx, Qy}, :201, :232-239, :251-266) */
	const real yN = p->muscl_nb_y_is_bed ? bN : sN[1], yE = p->muscl_nb_y_is_bed ? bE : sE[1];
	const real yS = p->muscl_nb_y_is_bed ? bS : sS[1], yW = p->muscl_nb_y_is_bed ? bW : sW[1];
	if (sC[0] - bC < RC(1E-5) ||                                      /* :325-330 (hard-coded 1E-5) */
	    yN <= RC(-9998.0) || yE <= RC(-9998.0) || yS <= RC(-9998.0) || yW <= RC(-9998.0))
		first = 1;

	real c[4] = { sC[0], sC[0] - bC, sC[2], sC[3] };                  /* :333 second element becomes depth */
	for (int k = 0; k < 4; ++k) { fN[k] = c[k]; fE[k] = c[k]; fS[k] = c[k]; fW[k] = c[k]; }
	if (first) return 1;

	real slopeX[4], slopeY[4];                                        /* :343-346 */
	orc_limiter(p, sW, c, sE, bW, bC, bE, slopeX);
	orc_limiter(p, sS, c, sN, bS, bC, bN, slopeY);

	face_extrapolate(bC, c, slopeY, RC(+0.5), fN);                    /* :349-352 */
	face_extrapolate(bC, c, slopeX, RC(+0.5), fE);
	face_extrapolate(bC, c, slopeY, RC(-0.5), fS);
	face_extrapolate(bC, c, slopeX, RC(-0.5), fW);

	real FN[3], FE[3], FS[3], FW[3];                                  /* :355-358 */
	flux_vector_y(VS, fN, FN);
	flux_vector_x(VS, fE, FE);
	flux_vector_y(VS, fS, FS);
	flux_vector_x(VS, fW, FW);

	evolve_cell(p, dt, c, fN, fE, fS, fW, FN, FE, FS, FW);            /* :362-373 */

	face_extrapolate(bC, c, slopeY, RC(+0.5), fN);                    /* :376-379 */
	face_extrapolate(bC, c, slopeX, RC(+0.5), fE);
	face_extrapolate(bC, c, slopeY, RC(-0.5), fS);
	face_extrapolate(bC, c, slopeX, RC(-0.5), fW);
	return 0;
}

/* 2nd-order reconstruction -- Schemes/CLSchemeMUSCLHancock.clc:1119-1230 */
int orc_reconstruct2(const orc_params* p, int dir, const real sL[4], real bL, const real sR[4], real bR,
                     const real eL[4], const real eR[4], real oL[8], real oR[8])
{
	const real VS = p->very_small;
	real L[8], R[8];
	(void)bL; (void)bR;                                               /* beds come from the face estimates */

	L[0] = eL[0]; L[1] = eL[1]; L[2] = eL[2]; L[3] = eL[3];           /* :1136-1144 (note <=) */
	L[4] = (eL[1] <= VS ? RC(0.0) : eL[2] / eL[1]);
	L[5] = (eL[1] <= VS ? RC(0.0) : eL[3] / eL[1]);
	L[6] = eL[0] - eL[1]; L[7] = RC(0.0);
	R[0] = eR[0]; R[1] = eR[1]; R[2] = eR[2]; R[3] = eR[3];           /* :1145-1153 */
	R[4] = (eR[1] <= VS ? RC(0.0) : eR[2] / eR[1]);
	R[5] = (eR[1] <= VS ? RC(0.0) : eR[3] / eR[1]);
	R[6] = eR[0] - eR[1]; R[7] = RC(0.0);

	real dBedMaximum = (L[6] > R[6] ? L[6] : R[6]);                   /* :1156-1158 */
	real dShiftV = dBedMaximum - (dir < ORC_DIR_S ? eL : eR)[0];
	if (dShiftV < RC(0.0)) dShiftV = RC(0.0);

	L[1] = (eL[0] - dBedMaximum > RC(0.0) ? (eL[0] - dBedMaximum) : RC(0.0));   /* :1161-1169 */
	L[0] = L[1] + dBedMaximum;
	L[2] = L[1] * L[4];
	L[3] = L[1] * L[5];
	R[1] = (eR[0] - dBedMaximum > RC(0.0) ? (eR[0] - dBedMaximum) : RC(0.0));
	R[0] = R[1] + dBedMaximum;
	R[2] = R[1] * R[4];
	R[3] = R[1] * R[5];

	int stop = stop_tests(VS, dir, L, R, sL, sR);                     /* :1173-1205: raw CELL discharges */

	L[6] = dBedMaximum - dShiftV;                                     /* :1208-1211 */
	R[6] = dBedMaximum - dShiftV;
	L[0] -= dShiftV;
	R[0] -= dShiftV;

	memcpy(oL, L, sizeof L);
	memcpy(oR, R, sizeof R);
	return stop;
}

/* =============================================================================================
 *  Grid kernels
 * ========================================================================================== */
#define IDX(p, x, y) ((size_t)(y) * (size_t)(p)->cols + (size_t)(x))          /* CLDomainCartesian.clc:26-30 */

/* One work-item of gts_cacheDisabled -- Schemes/CLSchemeGodunov.clc:164-384 */
static void godunov_cell(const orc_params* p, real dt, long x, long y, const real* bed, const real* src,
                         real* dst, const real* manning)
{
	const real VS = p->very_small, DX = p->dx, DY = p->dx;
	if (x >= p->cols - 1 || y >= p->rows - 1 || x <= 0 || y <= 0) return;       /* :183-187 */
	const size_t id = IDX(p, x, y);

	if (dt <= RC(0.0)) {                                              /* :201-206 */
		memcpy(dst + 4 * id, src + 4 * id, 4 * sizeof(real));
		return;
	}

	real c[4];
	memcpy(c, src + 4 * id, sizeof c);                                /* :209-211 */
	real bC = bed[id], n = manning[id];

	if (c[1] <= RC(-9999.0) || c[0] == RC(-9999.0)) {                 /* :214-218 */
		memcpy(dst + 4 * id, c, sizeof c);
		return;
	}

	const size_t iW = IDX(p, x - 1, y), iS = IDX(p, x, y - 1), iN = IDX(p, x, y + 1), iE = IDX(p, x + 1, y);
	real sW[4], sS[4], sN[4], sE[4];
	memcpy(sW, src + 4 * iW, sizeof sW); memcpy(sS, src + 4 * iS, sizeof sS);   /* :220-235 */
	memcpy(sN, src + 4 * iN, sizeof sN); memcpy(sE, src + 4 * iE, sizeof sE);
	real bW = bed[iW], bS = bed[iS], bN = bed[iN], bE = bed[iE];

	int dry = 0;                                                      /* :248-255 */
	if (c[0]  - bC < VS) dry++;
	if (sN[0] - bN < VS) dry++;
	if (sE[0] - bE < VS) dry++;
	if (sS[0] - bS < VS) dry++;
	if (sW[0] - bW < VS) dry++;
	if (dry >= 5) return;                                             /* dst NOT written (quirk Q3) */

	real L[8], R[8], FN[4], FS[4], FE[4], FW[4];
	int stop = 0;

	stop += orc_reconstruct(p, ORC_DIR_N, c, bC, sN, bN, L, R);       /* :259-277 */
	sN[0] = R[0]; bN = R[6];
	orc_hllc(p, ORC_DIR_N, L, R, FN);

	stop += orc_reconstruct(p, ORC_DIR_S, sS, bS, c, bC, L, R);       /* :280-291 */
	sS[0] = L[0]; bS = L[6];
	orc_hllc(p, ORC_DIR_S, L, R, FS);

	stop += orc_reconstruct(p, ORC_DIR_E, c, bC, sE, bE, L, R);       /* :294-305 */
	sE[0] = R[0]; bE = R[6];
	orc_hllc(p, ORC_DIR_E, L, R, FE);

	stop += orc_reconstruct(p, ORC_DIR_W, sW, bW, c, bC, L, R);       /* :308-319 */
	sW[0] = L[0]; bW = L[6];
	orc_hllc(p, ORC_DIR_W, L, R, FW);

	real Sx = RC(0.0);                                                /* :323-325 */
	real Sy = -1 * GRAVITY * ((sE[0] + sW[0]) / 2) * ((bE - bW) / DX);
	real Sz = -1 * GRAVITY * ((sN[0] + sS[0]) / 2) * ((bN - bS) / DY);

	real d0 = (FE[0] - FW[0]) / DX + (FN[0] - FS[0]) / DY - Sx;       /* :328-336 */
	real d2 = (FE[1] - FW[1]) / DX + (FN[1] - FS[1]) / DY - Sy;
	real d3 = (FE[2] - FW[2]) / DX + (FN[2] - FS[2]) / DY - Sz;
	d0 = small_to_zero(d0, VS);                                       /* :340-348 */
	d2 = small_to_zero(d2, VS);
	d3 = small_to_zero(d3, VS);

	if (stop > 0) { c[2] = RC(0.0); c[3] = RC(0.0); }                 /* :351-355 */

	c[0] = c[0] - dt * d0;                                            /* :358-360 */
	c[2] = c[2] - dt * d2;
	c[3] = c[3] - dt * d3;

	if (p->friction) {                                                /* :362-372 */
		real o[4];
		orc_friction(p, c, bC, n, dt, o);
		memcpy(c, o, sizeof c);
	}

	if (c[0] > c[1] && c[1] > RC(-9990.0)) c[1] = c[0];               /* :375-376 */
	if (c[0] - bC < VS) c[0] = bC;                                    /* :379-380 */

	memcpy(dst + 4 * id, c, sizeof c);                                /* :383 */
}

void orc_godunov_step(const orc_params* p, real dt, const real* bed, const real* src, real* dst,
                      const real* manning)
{
	long y;
#pragma omp parallel for schedule(static) num_threads(p->threads > 0 ? p->threads : 1)
	for (y = 0; y < p->rows; ++y)
		for (long x = 0; x < p->cols; ++x)
			godunov_cell(p, dt, x, y, bed, src, dst, manning);
}

/* calculateInertialFlux -- Schemes/CLSchemeInertial.clc:331-378 (FROUDE_LIMIT 0.8: CLSchemeInertial.clh:24) */
real orc_inertial_flux(const orc_params* p, real dManningCoef, real dTimestep, real dPreviousDischarge,
                       real dLevelUpstream, real dBedUpstream, real dLevelDownstream, real dBedDownstream)
{
	const real FROUDE_LIMIT = RC(0.8);
	real dDischarge;
	real dDepth = R_FMAX(dLevelDownstream, dLevelUpstream) - R_FMAX(dBedUpstream, dBedDownstream);   /* :342 */
	real dSlope = (dLevelDownstream - dLevelUpstream) / p->dx;                                        /* :343 */

	dDischarge = (dPreviousDischarge - (GRAVITY * dDepth * dTimestep * dSlope)) /                     /* :346-348 */
	             (RC(1.0) + GRAVITY * dDepth * dTimestep * dManningCoef * dManningCoef * R_FABS(dPreviousDischarge) /
	              R_POW103(dDepth));

	if (dDischarge > RC(0.0) &&                                                                       /* :351-356 */
	    ((R_FABS(dDischarge) / dDepth) / R_SQRT(GRAVITY * dDepth)) > FROUDE_LIMIT)
		dDischarge = dDepth * R_SQRT(GRAVITY * dDepth) * FROUDE_LIMIT;
	if (dDischarge < RC(0.0) &&
	    ((R_FABS(dDischarge) / dDepth) / R_SQRT(GRAVITY * dDepth)) > FROUDE_LIMIT)
		dDischarge = RC(0.0) - dDepth * R_SQRT(GRAVITY * dDepth) * FROUDE_LIMIT;

	if (dDepth < p->very_small)                                                                       /* :371-372 */
		dDischarge = RC(0.0);
	return dDischarge;
}

/* One work-item of ine_cacheDisabled -- Schemes/CLSchemeInertial.clc:26-169.  State = {Z, Zmax, Q across the
 * west face, Q across the south face}. */
static void inertial_cell(const orc_params* p, real dt, long x, long y, const real* bed, const real* src,
                          real* dst, const real* manning)
{
	const real VS = p->very_small;
	if (x >= p->cols - 1 || y >= p->rows - 1 || x <= 0 || y <= 0) return;       /* :45-49 */
	if (dt <= RC(0.0)) return;                                        /* :61-62: dst NOT written (unlike Godunov) */

	const size_t id = IDX(p, x, y);
	real c[4];
	memcpy(c, src + 4 * id, sizeof c);                                /* :65-67 */
	real bC = bed[id], n = manning[id];
	if (c[1] <= RC(-9999.0) || c[0] == RC(-9999.0)) {                 /* :70-74 */
		memcpy(dst + 4 * id, c, sizeof c);
		return;
	}
	const size_t iW = IDX(p, x - 1, y), iS = IDX(p, x, y - 1), iN = IDX(p, x, y + 1), iE = IDX(p, x + 1, y);
	const real *sW = src + 4 * iW, *sS = src + 4 * iS, *sN = src + 4 * iN, *sE = src + 4 * iE;   /* :76-91 */
	const real bW = bed[iW], bS = bed[iS], bN = bed[iN], bE = bed[iE];

	int dry = 0;                                                      /* :93-100 */
	if (c[0]  - bC < VS) dry++;
	if (sN[0] - bN < VS) dry++;
	if (sE[0] - bE < VS) dry++;
	if (sS[0] - bS < VS) dry++;
	if (sW[0] - bW < VS) dry++;
	if (dry >= 5) return;                                             /* dst NOT written (quirk Q3) */

	const real qN = orc_inertial_flux(p, n, dt, sN[3], sN[0], bN, c[0], bC);    /* :104-112 */
	const real qE = orc_inertial_flux(p, n, dt, sE[2], sE[0], bE, c[0], bC);    /* :114-122 */
	const real qS = orc_inertial_flux(p, n, dt, c[3], c[0], bC, sS[0], bS);     /* :124-132 */
	const real qW = orc_inertial_flux(p, n, dt, c[2], c[0], bC, sW[0], bW);     /* :134-142 */

	c[2] = qW;                                                        /* :144-145 */
	c[3] = qS;
	const real dDeltaFSL = (qE - qW + qN - qS) / p->dx;               /* :148-149 (DOMAIN_DELTAY) */
	c[0] = c[0] + dt * dDeltaFSL;                                     /* :152 */
	if (c[0] > c[1]) c[1] = c[0];                                     /* :155-156 */
	if (c[0] - bC < VS) c[0] = bC;                                    /* :159-160 */
	memcpy(dst + 4 * id, c, sizeof c);                                /* :163 */
}

void orc_inertial_step(const orc_params* p, real dt, const real* bed, const real* src, real* dst,
                       const real* manning)
{
	long y;
#pragma omp parallel for schedule(static) num_threads(p->threads > 0 ? p->threads : 1)
	for (y = 0; y < p->rows; ++y)
		for (long x = 0; x < p->cols; ++x)
			inertial_cell(p, dt, x, y, bed, src, dst, manning);
}

/* mch_1st_cacheNone -- Schemes/CLSchemeMUSCLHancock.clc:28-152; with p->muscl_nb_y_is_bed the reference's default,
 * mch_1st_cachePrediction -- :157-296: same cells (the overlapping 16 x 16 groups cover 1..n-2 once), the cell's own state
 * from global memory, the four neighbours from the LDS tile {Z, bed, Qx, Qy} */
void orc_muscl_predict(const orc_params* p, real dt, const real* bed, const real* state,
                       real* fN, real* fE, real* fS, real* fW)
{
	long y;
#pragma omp parallel for schedule(static) num_threads(p->threads > 0 ? p->threads : 1)
	for (y = 0; y < p->rows; ++y)
		for (long x = 0; x < p->cols; ++x) {
			if (x >= p->cols - 1 || y >= p->rows - 1 || x <= 0 || y <= 0) continue;   /* :57-61 */
			if (dt <= RC(0.0)) continue;                              /* :69-70 */
			const size_t id = IDX(p, x, y), iN = IDX(p, x, y + 1), iE = IDX(p, x + 1, y),
			             iS = IDX(p, x, y - 1), iW = IDX(p, x - 1, y);
			real st[20], bd[5], faces[16];
			memcpy(st,      state + 4 * id, 4 * sizeof(real));
			memcpy(st + 4,  state + 4 * iN, 4 * sizeof(real));
			memcpy(st + 8,  state + 4 * iE, 4 * sizeof(real));
			memcpy(st + 12, state + 4 * iS, 4 * sizeof(real));
			memcpy(st + 16, state + 4 * iW, 4 * sizeof(real));
			bd[0] = bed[id]; bd[1] = bed[iN]; bd[2] = bed[iE]; bd[3] = bed[iS]; bd[4] = bed[iW];
			if (p->muscl_nb_y_is_bed) {                               /* :243-248: own Zmax, neighbours' BED, and `==` */
				if (st[1] == RC(-9999.0) && bd[1] == RC(-9999.0) && bd[2] == RC(-9999.0) &&
				    bd[3] == RC(-9999.0) && bd[4] == RC(-9999.0))
					continue;
			} else
			if (st[1] <= RC(-9999.0) && st[5] <= RC(-9999.0) && st[9] <= RC(-9999.0) &&   /* :95-100 */
			    st[13] <= RC(-9999.0) && st[17] <= RC(-9999.0))
				continue;
			orc_mch_1st(p, dt, st, bd, faces);                        /* :109-125 */
			memcpy(fN + 4 * id, faces,      4 * sizeof(real));        /* :138-143 */
			memcpy(fE + 4 * id, faces + 4,  4 * sizeof(real));
			memcpy(fS + 4 * id, faces + 8,  4 * sizeof(real));
			memcpy(fW + 4 * id, faces + 12, 4 * sizeof(real));
		}
}

/* One work-item of mch_2nd_cacheNone -- Schemes/CLSchemeMUSCLHancock.clc:533-801 */
static void muscl_correct_cell(const orc_params* p, real dt, long x, long y, const real* src, real* dst,
                               const real* bed, const real* manning, const real* fN, const real* fE,
                               const real* fS, const real* fW)
{
	const real VS = p->very_small, DX = p->dx, DY = p->dx;
	if (x >= p->cols - 2 || y >= p->rows - 2 || x <= 1 || y <= 1) return;       /* :569-573 */
	if (dt <= RC(0.0)) return;                                        /* :576-577 */

	const size_t id = IDX(p, x, y);
	real c[4];
	memcpy(c, src + 4 * id, sizeof c);                                /* :588-591 */
	real bC = bed[id], n = manning[id];
	if (c[1] <= RC(-9999.0) || c[0] == RC(-9999.0)) return;           /* :594-595 */

	int dry = 0;
	if (c[0] - bC < VS) ++dry;                                        /* :597-598 */

	const size_t in_[4] = { IDX(p, x, y + 1), IDX(p, x + 1, y), IDX(p, x, y - 1), IDX(p, x - 1, y) };
	const real* fint[4] = { fN, fE, fS, fW };                         /* :581-584 */
	const real* fext[4] = { fS, fW, fN, fE };
	real sn[4][4], bn[4], ei[4][4], ee[4][4];
	for (int d = 0; d < 4; ++d) {                                     /* :601-635 */
		memcpy(sn[d], src + 4 * in_[d], sizeof sn[d]);
		bn[d] = bed[in_[d]];
		memcpy(ei[d], fint[d] + 4 * id, sizeof ei[d]);
		memcpy(ee[d], fext[d] + 4 * in_[d], sizeof ee[d]);
		if (sn[d][1] < VS) ++dry;                                     /* :633: tests Zmax, not depth (quirk Q6) */
	}
	if (dry >= 5) return;                                             /* :638 */

	real L[8], R[8], F[4][4];
	int stop = 0;
	stop += orc_reconstruct2(p, ORC_DIR_N, c, bC, sn[0], bn[0], ei[0], ee[0], L, R);   /* :642-655 */
	sn[0][0] = R[0]; sn[0][1] = R[6];
	orc_hllc(p, ORC_DIR_N, L, R, F[0]);
	stop += orc_reconstruct2(p, ORC_DIR_E, c, bC, sn[1], bn[1], ei[1], ee[1], L, R);   /* :658-671 */
	sn[1][0] = R[0]; sn[1][1] = R[6];
	orc_hllc(p, ORC_DIR_E, L, R, F[1]);
	stop += orc_reconstruct2(p, ORC_DIR_S, sn[2], bn[2], c, bC, ee[2], ei[2], L, R);   /* :674-687 */
	sn[2][0] = L[0]; sn[2][1] = L[6];
	orc_hllc(p, ORC_DIR_S, L, R, F[2]);
	stop += orc_reconstruct2(p, ORC_DIR_W, sn[3], bn[3], c, bC, ee[3], ei[3], L, R);   /* :690-703 */
	sn[3][0] = L[0]; sn[3][1] = L[6];
	orc_hllc(p, ORC_DIR_W, L, R, F[3]);

	real Sx = RC(0.0);                                                /* :707-709 */
	real Sy = -1 * GRAVITY * ((sn[1][0] + sn[3][0]) / 2) * ((sn[1][1] - sn[3][1]) / DX);
	real Sz = -1 * GRAVITY * ((sn[0][0] + sn[2][0]) / 2) * ((sn[0][1] - sn[2][1]) / DY);

	real d0 = (F[1][0] - F[3][0]) / DX + (F[0][0] - F[2][0]) / DY - Sx;          /* :712-720 */
	real d2 = (F[1][1] - F[3][1]) / DX + (F[0][1] - F[2][1]) / DY - Sy;
	real d3 = (F[1][2] - F[3][2]) / DX + (F[0][2] - F[2][2]) / DY - Sz;
	d0 = small_to_zero(d0, VS);
	d2 = small_to_zero(d2, VS);
	d3 = small_to_zero(d3, VS);

	if (stop > 0) { c[3] = RC(0.0); c[2] = RC(0.0); }                 /* :733-737 */

	c[0] = c[0] - dt * d0;                                            /* :742-744 */
	c[2] = c[2] - dt * d2;
	c[3] = c[3] - dt * d3;

	if (p->friction) {                                                /* :778-788 */
		real o[4];
		orc_friction(p, c, bC, n, dt, o);
		memcpy(c, o, sizeof c);
	}

	if (c[0] - bC < VS) c[0] = bC;                                    /* :791-792 (clamp BEFORE max here) */
	if (c[0] > c[1] && c[1] > RC(-9990.0)) c[1] = c[0];               /* :795-796 */

	memcpy(dst + 4 * id, c, sizeof c);                                /* :799 */
}

void orc_muscl_correct(const orc_params* p, real dt, const real* src, real* dst, const real* bed,
                       const real* manning, const real* fN, const real* fE, const real* fS, const real* fW)
{
	if (src == dst) {
		/* in place: order matters (quirk Q6) -> strictly serial, x fastest then y */
		for (long y = 0; y < p->rows; ++y)
			for (long x = 0; x < p->cols; ++x)
				muscl_correct_cell(p, dt, x, y, src, dst, bed, manning, fN, fE, fS, fW);
		return;
	}
	long y;
#pragma omp parallel for schedule(static) num_threads(p->threads > 0 ? p->threads : 1)
	for (y = 0; y < p->rows; ++y)
		for (long x = 0; x < p->cols; ++x)
			muscl_correct_cell(p, dt, x, y, src, dst, bed, manning, fN, fE, fS, fW);
}

/* tst_Reduce -- Schemes/CLDynamicTimestep.clc:166-249.  The reference takes a strided per-work-item
 * max, an LDS tree max and (in tst_Advance_Normal :75-80) a serial max over the group results;
 * max over finite non-negative values is exact and order-independent, so a single pass is identical.
 * Comparisons are `>` so NaN speeds never enter (:215-216). */
real orc_cfl_max_speed(const orc_params* p, const real* state, const real* bed)
{
	const real QS = p->quite_small;
	const size_t cells = (size_t)p->cols * (size_t)p->rows;
	real dMaxSpeed = RC(0.0);
	long i;
#pragma omp parallel for schedule(static) reduction(max : dMaxSpeed) num_threads(p->threads > 0 ? p->threads : 1)
	for (i = 0; i < (long)cells; ++i) {
		const real* s = state + 4 * i;
		real dDepth = s[0] - bed[i];                                  /* :191 */
		real dCellSpeed;
		if (dDepth > QS && s[1] > RC(-9999.0)) {                      /* :193 */
			real dVelX, dVelY;
			if (!p->simplified_cfl) {                                 /* :195-203 */
				dVelX = s[2] / dDepth;
				dVelY = s[3] / dDepth;
				if (dVelX < RC(0.0)) dVelX = -dVelX;
				if (dVelY < RC(0.0)) dVelY = -dVelY;
				dVelX += R_SQRT(GRAVITY * dDepth);
				dVelY += R_SQRT(GRAVITY * dDepth);
			} else {                                                  /* :205-210 TIMESTEP_SIMPLIFIED */
				dVelX = R_SQRT(GRAVITY * dDepth);
				dVelY = R_SQRT(GRAVITY * dDepth);
			}
			dCellSpeed = (dVelX < dVelY) ? dVelY : dVelX;             /* :211 */
		} else {
			dCellSpeed = RC(0.0);
		}
		if (dCellSpeed > dMaxSpeed) dMaxSpeed = dCellSpeed;
	}
	return dMaxSpeed;
}

/* tst_Advance_Normal -- Schemes/CLDynamicTimestep.clc:27-146 */
void orc_advance(const orc_params* p, orc_scalars* s, real dMaxSpeed)
{
	const real VS = p->very_small;
	real dLclTime = s->t;
	real dLclTimestep = R_FMAX(RC(0.0), s->dt);                       /* :42 */
	real dLclTimeHydrological = s->t_hydro;
	real dLclSyncTime = s->t_sync;
	real dLclBatchTimesteps = s->batch_dt;
	unsigned ok = s->batch_ok, skipped = s->batch_skipped;

	dLclTime += dLclTimestep;                                         /* :50-51 */
	dLclBatchTimesteps += dLclTimestep;
	if (dLclTimestep > RC(0.0)) ok++; else skipped++;                 /* :53-58 */

	if (dLclTimeHydrological > TIMESTEP_HYDROLOGICAL)                 /* :61-66 */
		dLclTimeHydrological = dLclTimestep;
	else
		dLclTimeHydrological += dLclTimestep;

	if (p->dynamic_dt) {                                              /* :68-92 */
		real dMinTime = p->dx / dMaxSpeed;
		if (dLclTime < TIMESTEP_START_MINIMUM_DURATION && dMinTime < TIMESTEP_START_MINIMUM)
			dMinTime = TIMESTEP_START_MINIMUM;
		dLclTimestep = p->courant * dMinTime;
	} else {
		dLclTimestep = p->fixed_dt;                                   /* :93-97 */
	}

	if (dLclTimestep > RC(0.0) && dLclTimestep < TIMESTEP_MINIMUM)    /* :112-113 */
		dLclTimestep = TIMESTEP_MINIMUM;

	if ((dLclTime + dLclTimestep) >= dLclSyncTime) {                  /* :118-124 */
		if (dLclSyncTime - dLclTime > VS)
			dLclTimestep = dLclSyncTime - dLclTime;
		if (dLclSyncTime - dLclTime <= VS)
			dLclTimestep = -dLclTimestep;
	}

	if (dLclTime < TIMESTEP_EARLY_LIMIT_DURATION && dLclTimestep > TIMESTEP_EARLY_LIMIT)   /* :128-129 */
		dLclTimestep = TIMESTEP_EARLY_LIMIT;

	if ((dLclTime + dLclTimestep) > p->end_time)                      /* :132-133 */
		dLclTimestep = p->end_time - dLclTime;

	if (dLclTimestep > TIMESTEP_MAXIMUM)                              /* :136-137 */
		dLclTimestep = TIMESTEP_MAXIMUM;

	s->t = dLclTime; s->dt = dLclTimestep; s->t_hydro = dLclTimeHydrological;   /* :140-145 */
	s->batch_dt = dLclBatchTimesteps; s->batch_ok = ok; s->batch_skipped = skipped;
}

/* tst_UpdateTimestep -- Schemes/CLDynamicTimestep.clc:255-317 */
void orc_update_timestep(const orc_params* p, orc_scalars* s, real dMaxSpeed)
{
	real dLclTime = s->t;
	real dLclOriginalTimestep = R_FABS(s->dt);                        /* :264 */
	real dLclSyncTime = s->t_sync;
	real dLclBatchTimesteps = s->batch_dt;
	/* :268 declares dLclTimestep WITHOUT a value; with TIMESTEP_FIXED nothing assigns it before the fmin at :297
	 * (undefined behaviour in the reference).  The reference's host build (oracle/_ref, -DREF_FIXED_DT) returns the
	 * original timestep there -- what fmin gives for any garbage >= original or a NaN -- so that is what is restated. */
	real dLclTimestep = dLclOriginalTimestep;

	if (p->dynamic_dt) {                                              /* :269-292 */
		real dMinTime = p->dx / dMaxSpeed;
		if (dLclTime < TIMESTEP_START_MINIMUM_DURATION && dMinTime < TIMESTEP_START_MINIMUM)
			dMinTime = TIMESTEP_START_MINIMUM;
		dLclTimestep = p->courant * dMinTime;
	}

	dLclTimestep = R_FMIN(dLclTimestep, dLclOriginalTimestep);        /* :297-298 */
	dLclBatchTimesteps = dLclBatchTimesteps - dLclOriginalTimestep + dLclTimestep;

	if (dLclTime < TIMESTEP_EARLY_LIMIT_DURATION && dLclTimestep > TIMESTEP_EARLY_LIMIT)   /* :301-302 */
		dLclTimestep = TIMESTEP_EARLY_LIMIT;

	if ((dLclTime + dLclTimestep) >= dLclSyncTime)                    /* :305-306 */
		dLclTimestep = R_FMAX(RC(0.0), dLclSyncTime - dLclTime);

	if (dLclTimestep > TIMESTEP_MAXIMUM)                              /* :309-310 */
		dLclTimestep = TIMESTEP_MAXIMUM;

	s->dt = dLclTimestep;                                             /* :313-314 */
	s->batch_dt = dLclBatchTimesteps;
}

/* bdy_Uniform -- Boundaries/CLBoundaries.clc:130-184; NDRange CBoundaryUniform.cpp:294-295 (quirk Q9) */
void orc_bdy_uniform(const orc_params* p, const orc_scalars* s, int definition,
                     const real* series, unsigned entries, real interval, real length,
                     real* state, const real* bed, int truncated_range)
{
	(void)entries;
	const long gx = truncated_range ? (p->cols / 8) * 8 : p->cols;
	const long gy = truncated_range ? (p->rows / 8) * 8 : p->rows;
	for (long y = 0; y < gy; ++y)
		for (long x = 0; x < gx; ++x) {
			if (x >= p->cols - 1 || y >= p->rows - 1 || x <= 0 || y <= 0) continue;   /* :148-152 */
			const size_t id = IDX(p, x, y);
			real* c = state + 4 * id;
			real dLclTime = s->t, dLclRealTimestep = s->dt, dLclTimestep = s->t_hydro;
			if (dLclTimestep < TIMESTEP_HYDROLOGICAL || dLclRealTimestep <= RC(0.0)) continue;   /* :165-166 */
			if (dLclTime >= length || c[1] <= RC(-9999.0)) continue;  /* :168-169 */
			unsigned long ts = (unsigned long)R_FLOOR(dLclTime / interval);           /* :172-173 */
			real rate = series[2 * ts + 1];
			if (definition == ORC_UNIFORM_RAIN_INTENSITY)             /* :176-177 */
				c[0] += rate / RC(3600000.0) * dLclTimestep;
			if (definition == ORC_UNIFORM_LOSS_RATE)                  /* :179-180 */
				c[0] = R_FMAX(bed[id], c[0] - rate / RC(3600000.0) * dLclTimestep);
		}
}

/* bdy_Gridded -- Boundaries/CLBoundaries.clc:186-246; NDRange CBoundaryGridded.cpp:298-299 */
void orc_bdy_gridded(const orc_params* p, const orc_scalars* s, int definition,
                     const real* grids, unsigned long entries, unsigned long grows, unsigned long gcols,
                     real resolution, real off_x, real off_y, real interval,
                     real* state, const real* bed, int truncated_range)
{
	(void)bed;
	const long gx = truncated_range ? (p->cols / 8) * 8 : p->cols;
	const long gy = truncated_range ? (p->rows / 8) * 8 : p->rows;
	for (long y = 0; y < gy; ++y)
		for (long x = 0; x < gx; ++x) {
			if (x >= p->cols - 1 || y >= p->rows - 1 || x <= 0 || y <= 0) continue;   /* :204-208 */
			const size_t id = IDX(p, x, y);
			real* c = state + 4 * id;
			real dLclTime = s->t, dLclTimestep = s->t_hydro;
			if (c[1] <= RC(-9999.0) || c[0] == RC(-9999.0)) continue; /* :220-221 */
			if (dLclTimestep < TIMESTEP_HYDROLOGICAL) continue;       /* :224-225 */
			unsigned long ts = (unsigned long)R_FLOOR(dLclTime / interval);           /* :228 */
			/* :229 clamps to `entries` (one slice PAST the end -- an out-of-bounds read in the
			 * reference); a defined restatement clamps to the last slice instead. */
			if (ts >= entries) ts = entries - 1;
			real col = R_FLOOR((((real)x * p->dx) - off_x) / resolution);             /* :231-232 */
			real row = R_FLOOR((((real)y * p->dx) - off_y) / resolution);
			unsigned long cell = (grows * gcols) * ts + (gcols * (unsigned long)row) + (unsigned long)col;   /* :233-234 */
			real rate = grids[cell];
			if (definition == ORC_GRIDDED_RAIN_INTENSITY)             /* :238-239 */
				c[0] += rate / RC(3600000.0) * dLclTimestep;
			if (definition == ORC_GRIDDED_MASS_FLUX)                  /* :241-242 */
				c[0] += rate / (p->dx * p->dx) * dLclTimestep;
		}
}

/* bdy_Cell -- Boundaries/CLBoundaries.clc:23-128; 1-D over the relation list (CBoundaryCell.cpp:442-443) */
void orc_bdy_cell(const orc_params* p, const orc_scalars* s, int depth_def, int discharge_def,
                  const unsigned long* relations, unsigned long count, const real* series, unsigned long entries,
                  real interval, real length, real* state, const real* bed)
{
	(void)entries;
	const real VS = p->very_small, DX = p->dx, DY = p->dx;
	const real dLocalTime = s->t, dLocalTimestep = s->dt;
	for (unsigned long r = 0; r < count; ++r) {
		if (dLocalTime >= length || dLocalTimestep <= RC(0.0)) return;            /* :38-39 */
		const unsigned long base = (unsigned long)R_FLOOR(dLocalTime / interval), next = base + 1;   /* :41-42 */
		const unsigned long id = relations[r];
		real* c = state + 4 * id;
		const real dCellBed = bed[id];
		const real w = R_FMOD(dLocalTime, interval) / interval;                   /* :50 */
		real ts[4];
		for (int k = 0; k < 4; ++k) ts[k] = series[4 * base + k] + (series[4 * next + k] - series[4 * base + k]) * w;

		if (depth_def == ORC_DEPTH_IS_DEPTH) {                                    /* :53-59 */
			c[0] = dCellBed + ts[1];
		} else if (depth_def == ORC_DEPTH_IS_FSL) {                               /* :60-66 */
			c[0] = R_FMAX(dCellBed, ts[1]);
		} else {                                                                  /* :67-98 */
			if (R_FABS(ts[2]) > VS || R_FABS(ts[3]) > VS || discharge_def == ORC_DISCHARGE_IS_VOLUME) {
				real dDepth = (R_FABS(ts[2]) * dLocalTimestep) / DY + (R_FABS(ts[3]) * dLocalTimestep) / DX;
				real dCriticalDepth = R_FMAX(R_POW13((ts[2] * ts[2]) / GRAVITY),
				                             R_POW13((ts[3] * ts[3]) / GRAVITY));
				if (discharge_def == ORC_DISCHARGE_IS_VOLUME) {
					dDepth = (R_FABS(ts[2]) * dLocalTimestep) / (DX * DY);
					dCriticalDepth = RC(0.0);
					ts[2] = RC(0.0);
					ts[3] = RC(0.0);
				}
				c[0] = R_FMAX(dCellBed + dCriticalDepth, c[0] + dDepth);
			}
		}
		if (discharge_def == ORC_DISCHARGE_IS_DISCHARGE) c[2] = ts[2];            /* :100-108 */
		else if (discharge_def == ORC_DISCHARGE_IS_VELOCITY) c[2] = ts[2] * (c[0] - dCellBed);
		if (discharge_def == ORC_DISCHARGE_IS_DISCHARGE) c[3] = ts[3];            /* :110-118 */
		else if (discharge_def == ORC_DISCHARGE_IS_VELOCITY) c[3] = ts[3] * (c[0] - dCellBed);
	}
}

/* =============================================================================================
 *  Simulation-level driver: the reference's per-iteration kernel graph
 *  CSchemeGodunov::scheduleIteration (CSchemeGodunov.cpp:1617-1666) and
 *  CSchemeMUSCLHancock::scheduleIteration (CSchemeMUSCLHancock.cpp:646-680)
 * ========================================================================================== */
typedef struct {
	int   kind;           /* 0 uniform, 1 gridded, 2 cell */
	int   definition;
	int   discharge_def;
	unsigned long* relations;
	unsigned long  count;
	real* data;
	unsigned long entries, grows, gcols;
	real  interval, length, resolution, off_x, off_y;
} orc_bdy;

struct orc_sim {
	orc_params  p;
	int         scheme;
	unsigned    quirks;
	size_t      cells;
	real       *primary, *alt, *bed, *manning;
	real       *face[4];
	orc_scalars sc;
	int         use_alt;          /* bUseAlternateKernel (CSchemeGodunov.cpp:1090) */
	orc_bdy    *bdy;
	int         nbdy;
};

orc_sim* orc_sim_create(const orc_params* p, int scheme, unsigned quirks, real dt_initial)
{
	orc_sim* s = (orc_sim*)calloc(1, sizeof *s);
	s->p = *p;
	s->scheme = scheme;
	if (scheme == ORC_SCHEME_INERTIAL) s->p.simplified_cfl = 1;       /* CLSchemeInertial.clh:25, always defined */
	s->quirks = quirks;
	s->p.muscl_nb_y_is_bed = (quirks & ORC_Q11_MUSCL_NB_Y_IS_BED) ? 1 : 0;
	s->cells = (size_t)p->cols * (size_t)p->rows;
	s->primary = (real*)calloc(s->cells * 4, sizeof(real));
	s->alt     = (real*)calloc(s->cells * 4, sizeof(real));
	s->bed     = (real*)calloc(s->cells, sizeof(real));
	s->manning = (real*)calloc(s->cells, sizeof(real));
	if (scheme == ORC_SCHEME_MUSCL)
		for (int d = 0; d < 4; ++d) s->face[d] = (real*)calloc(s->cells * 4, sizeof(real));
	/* CSchemeGodunov.cpp:862-867: t = 0, dt = timestepInitial (CScheme.cpp:49), t_hydro = 0, t_target = 0 */
	s->sc.t = RC(0.0); s->sc.dt = dt_initial; s->sc.t_hydro = RC(0.0); s->sc.t_sync = RC(0.0);
	s->sc.batch_dt = RC(0.0); s->sc.batch_ok = 0; s->sc.batch_skipped = 0;
	return s;
}

void orc_sim_destroy(orc_sim* s)
{
	if (!s) return;
	free(s->primary); free(s->alt); free(s->bed); free(s->manning);
	for (int d = 0; d < 4; ++d) free(s->face[d]);
	for (int i = 0; i < s->nbdy; ++i) { free(s->bdy[i].data); free(s->bdy[i].relations); }
	free(s->bdy);
	free(s);
}

/* prepareSimulation (CSchemeGodunov.cpp:1064-1072): BOTH state buffers receive the same host array */
void orc_sim_upload(orc_sim* s, const real* state, const real* bed, const real* manning)
{
	/* the ping-pong phase restarts with the cell states only (prepareSimulation writes both state buffers and clears
	 * bUseAlternateKernel, CSchemeGodunov.cpp:1064-1075); a write of the bed or Manning buffer alone is a plain
	 * COCLBuffer::queueWriteAll and leaves it where it is */
	if (state)   { memcpy(s->primary, state, s->cells * 4 * sizeof(real)); memcpy(s->alt, state, s->cells * 4 * sizeof(real)); s->use_alt = 0; }
	if (bed)     memcpy(s->bed, bed, s->cells * sizeof(real));
	if (manning) memcpy(s->manning, manning, s->cells * sizeof(real));
}

static orc_bdy* new_bdy(orc_sim* s)
{
	s->bdy = (orc_bdy*)realloc(s->bdy, (size_t)(s->nbdy + 1) * sizeof(orc_bdy));
	memset(&s->bdy[s->nbdy], 0, sizeof(orc_bdy));
	return &s->bdy[s->nbdy++];
}

int orc_sim_add_uniform(orc_sim* s, int definition, const real* series, unsigned entries, real interval, real length)
{
	orc_bdy* b = new_bdy(s);
	b->kind = 0; b->definition = definition; b->entries = entries; b->interval = interval; b->length = length;
	b->data = (real*)malloc((size_t)entries * 2 * sizeof(real));
	memcpy(b->data, series, (size_t)entries * 2 * sizeof(real));
	return s->nbdy - 1;
}

int orc_sim_add_gridded(orc_sim* s, int definition, const real* grids, unsigned long entries,
                        unsigned long grows, unsigned long gcols, real resolution, real off_x, real off_y,
                        real interval)
{
	orc_bdy* b = new_bdy(s);
	b->kind = 1; b->definition = definition; b->entries = entries; b->grows = grows; b->gcols = gcols;
	b->resolution = resolution; b->off_x = off_x; b->off_y = off_y; b->interval = interval;
	size_t n = (size_t)entries * grows * gcols;
	b->data = (real*)malloc(n * sizeof(real));
	memcpy(b->data, grids, n * sizeof(real));
	return s->nbdy - 1;
}

int orc_sim_add_cell(orc_sim* s, int depth_def, int discharge_def, const unsigned long* relations,
                     unsigned long count, const real* series, unsigned long entries, real interval, real length)
{
	orc_bdy* b = new_bdy(s);
	b->kind = 2; b->definition = depth_def; b->discharge_def = discharge_def; b->count = count; b->entries = entries;
	b->interval = interval; b->length = length;
	b->relations = (unsigned long*)malloc(count * sizeof(unsigned long));
	memcpy(b->relations, relations, count * sizeof(unsigned long));
	b->data = (real*)malloc((size_t)entries * 4 * sizeof(real));
	memcpy(b->data, series, (size_t)entries * 4 * sizeof(real));
	return s->nbdy - 1;
}

void orc_sim_set_target(orc_sim* s, real t_sync) { s->sc.t_sync = t_sync; }       /* CSchemeGodunov.cpp:1166-1176 */
void orc_sim_force_dt(orc_sim* s, real dt)       { s->sc.dt = dt; }               /* :1213-1232 */
void orc_sim_reset_counters(orc_sim* s)                                           /* tst_ResetCounters, CLDynamicTimestep.clc:151-161 */
{ s->sc.batch_dt = RC(0.0); s->sc.batch_ok = 0; s->sc.batch_skipped = 0; }

void orc_sim_update_timestep(orc_sim* s)
{
	real vmax = RC(0.0);
	if (s->p.dynamic_dt) vmax = orc_cfl_max_speed(&s->p, s->primary, s->bed);
	orc_update_timestep(&s->p, &s->sc, vmax);
}

static void apply_boundaries(orc_sim* s, real* target)
{
	/* CBoundaryMap::applyBoundaries (CBoundaryMap.cpp:76-80): one kernel per boundary, unordered in the
	 * reference (quirk Q7); here: the order they were added. */
	const int trunc = (s->quirks & ORC_Q9_BDY_TRUNCATED) ? 1 : 0;
	for (int i = 0; i < s->nbdy; ++i) {
		orc_bdy* b = &s->bdy[i];
		if (b->kind == 2)
			orc_bdy_cell(&s->p, &s->sc, b->definition, b->discharge_def, b->relations, b->count, b->data, b->entries,
			             b->interval, b->length, target, s->bed);
		else if (b->kind == 0)
			orc_bdy_uniform(&s->p, &s->sc, b->definition, b->data, (unsigned)b->entries, b->interval, b->length,
			                target, s->bed, trunc);
		else
			orc_bdy_gridded(&s->p, &s->sc, b->definition, b->data, b->entries, b->grows, b->gcols,
			                b->resolution, b->off_x, b->off_y, b->interval, target, s->bed, trunc);
	}
}

static void iterate_godunov(orc_sim* s)
{
	real* src = s->use_alt ? s->alt : s->primary;                     /* :1624-1635 */
	real* dst = s->use_alt ? s->primary : s->alt;
	apply_boundaries(s, src);                                         /* :1638 */
	if (s->scheme == ORC_SCHEME_INERTIAL)                             /* CSchemeInertial inherits the graph; its */
		orc_inertial_step(&s->p, s->sc.dt, s->bed, src, dst, s->manning);   /* kernel is oclKernelFullTimestep */
	else
		orc_godunov_step(&s->p, s->sc.dt, s->bed, src, dst, s->manning);  /* :1642 */
	real vmax = RC(0.0);
	if (s->p.dynamic_dt)                                              /* :1653-1657; arg 3 does not exist -> primary (Q1) */
		vmax = orc_cfl_max_speed(&s->p, (s->quirks & ORC_Q1_CFL_READS_PRIMARY) ? s->primary : dst, s->bed);
	orc_advance(&s->p, &s->sc, vmax);                                 /* :1660 */
	s->use_alt = !s->use_alt;                                         /* Threaded_runBatch :1300 */
}

static void iterate_muscl(orc_sim* s)
{
	/* no applyBoundaries call at all (quirk Q8) */
	orc_muscl_predict(&s->p, s->sc.dt, s->bed, s->primary, s->face[0], s->face[1], s->face[2], s->face[3]);
	if (s->quirks & ORC_Q6_MUSCL_SERIAL) {
		orc_muscl_correct(&s->p, s->sc.dt, s->primary, s->primary, s->bed, s->manning,
		                  s->face[0], s->face[1], s->face[2], s->face[3]);
	} else {
		memcpy(s->alt, s->primary, s->cells * 4 * sizeof(real));
		orc_muscl_correct(&s->p, s->sc.dt, s->primary, s->alt, s->bed, s->manning,
		                  s->face[0], s->face[1], s->face[2], s->face[3]);
		real* t = s->primary; s->primary = s->alt; s->alt = t;
	}
	real vmax = RC(0.0);
	if (s->p.dynamic_dt) vmax = orc_cfl_max_speed(&s->p, s->primary, s->bed);
	orc_advance(&s->p, &s->sc, vmax);
}

void orc_sim_run(orc_sim* s, long n, real* dt_trace)
{
	for (long i = 0; i < n; ++i) {
		if (dt_trace) dt_trace[i] = s->sc.dt;
		if (s->scheme == ORC_SCHEME_MUSCL) iterate_muscl(s); else iterate_godunov(s);
	}
}

void orc_sim_scalars(const orc_sim* s, orc_scalars* out) { *out = s->sc; }

void orc_sim_download(const orc_sim* s, real* state)
{
	const real* cur = (s->scheme != ORC_SCHEME_MUSCL && s->use_alt) ? s->alt : s->primary;
	memcpy(state, cur, s->cells * 4 * sizeof(real));
}
