/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build) -- not product code.
 *
 * Scalar-pointer entry points appended AFTER the reference's own .clc text so that the
 * reference's inner functions (which take OpenCL vector types by value) can be called from
 * plain C / ctypes.  Nothing here computes anything: each wrapper loads vectors, calls the
 * reference function, and stores the result.
 *
 * Functions wrapped (all compiled from /root/reference/src, unmodified):
 *   reconstructInterface  Schemes/CLSchemeGodunov.clc:27        (REF_GODUNOV build)
 *                         Schemes/CLSchemeMUSCLHancock.clc:1119 (REF_MUSCL build)
 *   riemannSolver         Solvers/CLSolverHLLC.clc:27
 *   implicitFriction      Schemes/CLFriction.clc:26
 *   slopeLimiter          Schemes/Limiters/CLSlopeLimiterMINMOD.clc:26   (REF_MUSCL)
 *   mch_1st               Schemes/CLSchemeMUSCLHancock.clc:301           (REF_MUSCL)
 *   calculateInertialFlux Schemes/CLSchemeInertial.clc:331               (REF_INERTIAL)
 */

#define LD4(p, i)  (cl_double4)((p)[4*(i)+0], (p)[4*(i)+1], (p)[4*(i)+2], (p)[4*(i)+3])
#define LD8(p)     (cl_double8)((p)[0], (p)[1], (p)[2], (p)[3], (p)[4], (p)[5], (p)[6], (p)[7])
#define ST4(v, p, i) do { (p)[4*(i)+0] = (v).x; (p)[4*(i)+1] = (v).y; (p)[4*(i)+2] = (v).z; (p)[4*(i)+3] = (v).w; } while (0)
#define ST8(v, p)  do { (p)[0] = (v).s0; (p)[1] = (v).s1; (p)[2] = (v).s2; (p)[3] = (v).s3; \
                        (p)[4] = (v).s4; (p)[5] = (v).s5; (p)[6] = (v).s6; (p)[7] = (v).s7; } while (0)

#ifndef REF_INERTIAL            /* the inertial program does not include the HLLC solver (CSchemeInertial.cpp:202-221) */
__kernel void refw_hllc(int dir, __global const cl_double* L, __global const cl_double* R,
                        __global cl_double* out)
{
	cl_double8 l = LD8(L), r = LD8(R);
	cl_double4 f = riemannSolver((cl_uchar)dir, l, r, false);
	{ cl_double4 tmp_ = f; ST4(tmp_, out, 0); }
}
#endif

__kernel void refw_friction(__global const cl_double* state, cl_double bed, cl_double n,
                            cl_double dt, __global cl_double* out)
{
	cl_double4 s = LD4(state, 0);
	{ cl_double4 tmp_ = implicitFriction(s, bed, n, dt); ST4(tmp_, out, 0); }
}

#ifdef REF_GODUNOV
__kernel void refw_reconstruct(int dir, __global const cl_double* sL, cl_double bL,
                               __global const cl_double* sR, cl_double bR,
                               __global cl_double* oL, __global cl_double* oR,
                               __global int* stop)
{
	cl_double8 l, r;
	cl_uchar s = reconstructInterface(LD4(sL, 0), bL, LD4(sR, 0), bR, &l, &r, (cl_uchar)dir);
	ST8(l, oL);
	ST8(r, oR);
	*stop = s;
}
#endif

#ifdef REF_MUSCL
__kernel void refw_reconstruct2(int dir, __global const cl_double* sL, cl_double bL,
                                __global const cl_double* sR, cl_double bR,
                                __global const cl_double* eL, __global const cl_double* eR,
                                __global cl_double* oL, __global cl_double* oR,
                                __global int* stop)
{
	cl_double8 l, r;
	cl_uchar s = reconstructInterface(LD4(sL, 0), bL, LD4(sR, 0), bR,
	                                  LD4(eL, 0), LD4(eR, 0), &l, &r, (cl_uchar)dir);
	ST8(l, oL);
	ST8(r, oR);
	*stop = s;
}

__kernel void refw_limiter(__global const cl_double* sL, __global const cl_double* sC,
                           __global const cl_double* sR, cl_double bL, cl_double bC, cl_double bR,
                           __global cl_double* out)
{
	{ cl_double4 tmp_ = slopeLimiter(LD4(sL, 0), LD4(sC, 0), LD4(sR, 0), bL, bC, bR); ST4(tmp_, out, 0); }
}

/* states: C,N,E,S,W (4 each); beds: C,N,E,S,W; faces out: N,E,S,W (4 each) */
__kernel void refw_mch_1st(cl_double dt, __global const cl_double* states,
                           __global const cl_double* beds, __global cl_double* faces,
                           __global int* firstOrder)
{
	cl_double4 fN, fE, fS, fW;
	bool fo = mch_1st(dt, LD4(states, 0), LD4(states, 1), LD4(states, 2),
	                  LD4(states, 3), LD4(states, 4),
	                  beds[0], beds[1], beds[2], beds[3], beds[4], &fN, &fE, &fS, &fW);
	{ cl_double4 tmp_ = fN; ST4(tmp_, faces, 0); }
	{ cl_double4 tmp_ = fE; ST4(tmp_, faces, 1); }
	{ cl_double4 tmp_ = fS; ST4(tmp_, faces, 2); }
	{ cl_double4 tmp_ = fW; ST4(tmp_, faces, 3); }
	*firstOrder = fo ? 1 : 0;
}
#endif

#ifdef REF_INERTIAL
/* in: {manning, dt, previous discharge, level up, bed up, level down, bed down} */
__kernel void refw_inertial_flux(__global const cl_double* in, __global cl_double* out)
{
	*out = calculateInertialFlux(in[0], in[1], in[2], in[3], in[4], in[5], in[6]);
}
#endif
