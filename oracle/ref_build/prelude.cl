/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build) -- not product code.
 *
 * Prelude placed in front of the reference's own OpenCL C files when they are compiled,
 * unmodified and from where they lie under /root/reference/src, into a host shared object.
 *
 * The reference builds this text at run time from its XML parameters:
 *   - extension/typedef header : src/OpenCL/Executors/COCLProgram.cpp:359-399
 *   - "#define" constants      : src/Schemes/CSchemeGodunov.cpp:666-784 (prepare1OConstants),
 *                                src/Schemes/CSchemeMUSCLHancock.cpp:404-523,
 *                                src/Schemes/CSchemeInertial.cpp:226-251
 * Here every run-time parameter is an `extern __constant` variable instead of a literal so one
 * shared object serves every grid size / threshold; the arithmetic is unchanged (IEEE division
 * by a variable 1.0 == division by the literal).
 *
 * REF_FP32 selects the reference's single-precision mode (typedef float cl_double, built with
 * -cl-single-precision-constant exactly as COCLProgram.cpp:66-70 does).
 */
#pragma OPENCL EXTENSION cl_khr_fp64 : enable

#ifdef REF_FP32
typedef float   cl_double;
typedef float2  cl_double2;
typedef float4  cl_double4;
typedef float8  cl_double8;
#else
typedef double  cl_double;
typedef double2 cl_double2;
typedef double4 cl_double4;
typedef double8 cl_double8;
#endif

/* run-time parameters (defined in shim.cpp, set by ref_configure()) */
extern __constant cl_double REFP_VERY_SMALL, REFP_QUITE_SMALL, REFP_DELTAX, REFP_DELTAY;
extern __constant cl_double REFP_ENDTIME, REFP_OUTPUTTIME, REFP_COURANT, REFP_FIXED_DT;
extern __constant long      REFP_COLS, REFP_ROWS, REFP_CELLCOUNT;
extern __constant unsigned  REFP_WORKERS;

#define VERY_SMALL            REFP_VERY_SMALL
#define QUITE_SMALL           REFP_QUITE_SMALL
#define DOMAIN_CELLCOUNT      REFP_CELLCOUNT
#define DOMAIN_COLS           REFP_COLS
#define DOMAIN_ROWS           REFP_ROWS
#define DOMAIN_DELTAX         REFP_DELTAX
#define DOMAIN_DELTAY         REFP_DELTAY
#define SCHEME_ENDTIME        REFP_ENDTIME
#define SCHEME_OUTPUTTIME     REFP_OUTPUTTIME
#define COURANT_NUMBER        REFP_COURANT
#define TIMESTEP_WORKERS      REFP_WORKERS

/* The oracle drives one work-item at a time: every kernel is built for 1x1x1 groups -- except tst_Reduce when the host asks for
 * threads: its team runs as one work-group with a real barrier (shim.cpp: ref_reduce; max is exact, so the result is identical). */
#define REQD_WG_SIZE_FULL_TS  __attribute__((reqd_work_group_size(1, 1, 1)))
#define REQD_WG_SIZE_HALF_TS  __attribute__((reqd_work_group_size(1, 1, 1)))
#define REQD_WG_SIZE_LINE     __attribute__((reqd_work_group_size(1, 1, 1)))
#define TIMESTEP_GROUPSIZE    64   /* size of tst_Reduce's __local array: groups of up to 64 work-items (shim.cpp: ref_reduce) */
#define GTS_DIM1              16
#define GTS_DIM2              16
#define MCH_STG1_DIM1         16
#define MCH_STG1_DIM2         16
#define MCH_STG2_DIM1         16
#define MCH_STG2_DIM2         16
#define MEM_SEPARATE_FACES    1
#define INE_DIM1              16
#define INE_DIM2              16

#ifdef REF_FIXED_DT
#define TIMESTEP_FIXED        REFP_FIXED_DT
#else
#define TIMESTEP_DYNAMIC      1
#endif
#ifndef REF_NO_FRICTION
#define FRICTION_ENABLED      1
#endif
#define FRICTION_IN_FLUX_KERNEL 1
