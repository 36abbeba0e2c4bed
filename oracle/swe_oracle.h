/*
 * oracle/swe_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, scalar, CPU restatement of the reference's shallow-water hot path (HiPIMS-OCL,
 * lukeshope/hipims-ocl): Godunov / MUSCL-Hancock / partial-inertial timestep + HLLC + MINMOD +
 * friction + rainfall source terms + CFL reduction + time control.  Every function cites the reference
 * file:line it follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may use it -- as the checker / reported baseline, never as the thing shipped.
 *
 * PINNING: the reference has no golden vectors or tests of its own for this path
 * ("Expected results: To be completed").  The oracle is instead pinned
 * against the reference ITSELF: the reference's OpenCL C kernel sources are compiled, unmodified,
 * for the host (oracle/ref_build -> oracle/_ref/ *.so) and this restatement is checked
 * bit-for-bit against them (tests/test_oracle_vs_ref.py, runs where /root/reference exists) and
 * against the committed fixtures those binaries produced (tests/golden/ *.npz, travel everywhere).
 *
 * Built twice: real = double (liboracle_f64.so) and real = float (liboracle_f32.so, the
 * reference's "typedef float cl_double" + -cl-single-precision-constant mode,
 * src/OpenCL/Executors/COCLProgram.cpp:66-70, :387-399).  Compile with -ffp-contract=off.
 */
#ifndef SWE_ORACLE_H
#define SWE_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#ifdef ORC_FP32
typedef float real;
#else
typedef double real;
#endif

/* direction codes: src/Domain/Cartesian/CLDomainCartesian.clh:33-36 */
enum { ORC_DIR_N = 0, ORC_DIR_E = 1, ORC_DIR_S = 2, ORC_DIR_W = 3 };

enum { ORC_SCHEME_GODUNOV = 0, ORC_SCHEME_MUSCL = 1, ORC_SCHEME_INERTIAL = 2 };

/* quirk switches (SURVEY.md section 8 a-quirks); the defaults reproduce the reference */
enum {
	ORC_Q1_CFL_READS_PRIMARY = 1,   /* CSchemeGodunov.cpp:1629/:1634: tst_Reduce always reads the primary buffer */
	ORC_Q9_BDY_TRUNCATED     = 2,   /* CBoundaryUniform.cpp:294: boundary NDRange = floor(n/8)*8 */
	ORC_Q6_MUSCL_SERIAL      = 4,   /* in-place corrector driven row-major (what oracle/_ref does); off = snapshot */
	ORC_Q11_MUSCL_NB_Y_IS_BED = 8,  /* the reference's DEFAULT MUSCL configuration (kCachePrediction, CSchemeMUSCLHancock.cpp:46)
	                                 * runs mch_1st_cachePrediction, whose LDS tile holds {Z, BED, Qx, Qy}
	                                 * (CLSchemeMUSCLHancock.clc:201): the neighbours it hands to mch_1st carry their bed in .y,
	                                 * so the first-order fallback (:325-330) and the all-disabled skip (:243-248) test the
	                                 * neighbours' BED; off = mch_1st_cacheNone (neighbours' Zmax, :95-100) */
	ORC_QUIRKS_REFERENCE     = 1 | 2 | 4 | 8
};

/* uniform-boundary definitions: src/Boundaries/CLBoundaries.clh:44-50 */
enum { ORC_UNIFORM_RAIN_INTENSITY = 0, ORC_UNIFORM_LOSS_RATE = 1 };
enum { ORC_GRIDDED_RAIN_INTENSITY = 0, ORC_GRIDDED_RAIN_ACCUMUL = 1, ORC_GRIDDED_MASS_FLUX = 2 };
/* cell-boundary definitions: src/Boundaries/CLBoundaries.clh:34-42 */
enum { ORC_DEPTH_IGNORE = 0, ORC_DEPTH_IS_FSL = 1, ORC_DEPTH_IS_DEPTH = 2, ORC_DEPTH_IS_CRITICAL = 3 };
enum { ORC_DISCHARGE_IGNORE = 0, ORC_DISCHARGE_IS_DISCHARGE = 1, ORC_DISCHARGE_IS_VELOCITY = 2, ORC_DISCHARGE_IS_VOLUME = 3 };

typedef struct {
	long  cols, rows;
	real  dx;            /* DOMAIN_DELTAX == DOMAIN_DELTAY (CSchemeGodunov.cpp:780-781) */
	real  very_small;    /* VERY_SMALL  (dryThreshold, default 1e-10, CSchemeGodunov.cpp:56) */
	real  quite_small;   /* QUITE_SMALL = 10*VERY_SMALL (CSchemeGodunov.cpp:57, :523) */
	real  courant;       /* COURANT_NUMBER (default 0.5, CScheme.cpp:48) */
	real  end_time;      /* SCHEME_ENDTIME */
	real  fixed_dt;      /* TIMESTEP_FIXED when dynamic_dt == 0 */
	int   dynamic_dt;    /* TIMESTEP_DYNAMIC */
	int   friction;      /* FRICTION_ENABLED (fused: FRICTION_IN_FLUX_KERNEL) */
	int   threads;       /* worker threads for the grid loops (1 = scalar) */
	int   simplified_cfl;/* TIMESTEP_SIMPLIFIED (CLSchemeInertial.clh:25): wave speed = sqrt(g h) only */
	int   muscl_nb_y_is_bed; /* ORC_Q11 (orc_sim_create sets it from the quirk mask): predictor = mch_1st_cachePrediction */
} orc_params;

typedef struct {
	real     t, dt, t_hydro, t_sync, batch_dt;
	unsigned batch_ok, batch_skipped;
} orc_scalars;

int   orc_real_bytes(void);

/* ---- function level (registers in, registers out) ---- */
int   orc_reconstruct(const orc_params* p, int dir, const real sL[4], real bL, const real sR[4], real bR,
                      real oL[8], real oR[8]);
void  orc_hllc(const orc_params* p, int dir, const real L[8], const real R[8], real F[4]);
void  orc_friction(const orc_params* p, const real s[4], real bed, real n, real dt, real out[4]);
real  orc_limited_slope(real l, real c, real r);
void  orc_limiter(const orc_params* p, const real sL[4], const real sC[4], const real sR[4],
                  real bL, real bC, real bR, real out[4]);
/* states/beds ordered C,N,E,S,W ; faces ordered N,E,S,W ; returns 1 if first-order fallback */
int   orc_mch_1st(const orc_params* p, real dt, const real states[20], const real beds[5], real faces[16]);
int   orc_reconstruct2(const orc_params* p, int dir, const real sL[4], real bL, const real sR[4], real bR,
                       const real eL[4], const real eR[4], real oL[8], real oR[8]);

/* calculateInertialFlux (CLSchemeInertial.clc:331-378) */
real  orc_inertial_flux(const orc_params* p, real n, real dt, real q_prev, real z_up, real b_up, real z_down, real b_down);

/* ---- grid level (one kernel of the reference each) ---- */
void  orc_godunov_step(const orc_params* p, real dt, const real* bed, const real* src, real* dst,
                       const real* manning);
/* ine_cacheDisabled (CLSchemeInertial.clc:26-169) */
void  orc_inertial_step(const orc_params* p, real dt, const real* bed, const real* src, real* dst,
                        const real* manning);
void  orc_muscl_predict(const orc_params* p, real dt, const real* bed, const real* state,
                        real* fN, real* fE, real* fS, real* fW);
/* src == dst: in place, row-major serial (quirk Q6); src != dst: snapshot (dst must start as a copy of src) */
void  orc_muscl_correct(const orc_params* p, real dt, const real* src, real* dst, const real* bed,
                        const real* manning, const real* fN, const real* fE, const real* fS, const real* fW);
real  orc_cfl_max_speed(const orc_params* p, const real* state, const real* bed);
void  orc_advance(const orc_params* p, orc_scalars* s, real max_speed);
void  orc_update_timestep(const orc_params* p, orc_scalars* s, real max_speed);
void  orc_bdy_uniform(const orc_params* p, const orc_scalars* s, int definition,
                      const real* series /* [n][2] time,value */, unsigned entries, real interval, real length,
                      real* state, const real* bed, int truncated_range);
void  orc_bdy_gridded(const orc_params* p, const orc_scalars* s, int definition,
                      const real* grids /* [entries][grows][gcols] */, unsigned long entries,
                      unsigned long grows, unsigned long gcols, real resolution, real off_x, real off_y,
                      real interval, real* state, const real* bed, int truncated_range);

void  orc_bdy_cell(const orc_params* p, const orc_scalars* s, int depth_def, int discharge_def,
                   const unsigned long* relations, unsigned long count,
                   const real* series /* [entries][4] time, depth/fsl, qx, qy */, unsigned long entries,
                   real interval, real length, real* state, const real* bed);

/* ---- simulation level: scheduleIteration (CSchemeGodunov.cpp:1617-1666,
 *      CSchemeMUSCLHancock.cpp:646-680) incl. ping-pong and quirks ---- */
typedef struct orc_sim orc_sim;
orc_sim* orc_sim_create(const orc_params* p, int scheme, unsigned quirks, real dt_initial);
void     orc_sim_destroy(orc_sim* s);
void     orc_sim_upload(orc_sim* s, const real* state, const real* bed, const real* manning);
int      orc_sim_add_uniform(orc_sim* s, int definition, const real* series, unsigned entries, real interval, real length);
int      orc_sim_add_gridded(orc_sim* s, int definition, const real* grids, unsigned long entries,
                             unsigned long grows, unsigned long gcols, real resolution, real off_x, real off_y,
                             real interval);
int      orc_sim_add_cell(orc_sim* s, int depth_def, int discharge_def, const unsigned long* relations,
                          unsigned long count, const real* series, unsigned long entries, real interval, real length);
void     orc_sim_set_target(orc_sim* s, real t_sync);
void     orc_sim_force_dt(orc_sim* s, real dt);
void     orc_sim_reset_counters(orc_sim* s);
/* tst_Reduce (primary buffer) + tst_UpdateTimestep, as Threaded_runBatch queues them after a new target time
 * (CSchemeGodunov.cpp:1189-1195) */
void     orc_sim_update_timestep(orc_sim* s);
/* run n iterations; if dt_trace != NULL it receives the dt USED by each iteration */
void     orc_sim_run(orc_sim* s, long n, real* dt_trace);
void     orc_sim_scalars(const orc_sim* s, orc_scalars* out);
/* the buffer the NEXT iteration would read (CSchemeGodunov::getNextCellSourceBuffer, :1705-1715) */
void     orc_sim_download(const orc_sim* s, real* state);

#ifdef __cplusplus
}
#endif
#endif
