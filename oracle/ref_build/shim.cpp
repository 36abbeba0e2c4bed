/*
 * TEST INFRASTRUCTURE ONLY (oracle/_ref build) -- not product code.
 *
 * Work-item shim + NDRange drivers for the reference's OpenCL C kernels when they are compiled
 * for the host with `clang -x cl -target x86_64` (recipe: SURVEY.md Appendix B).
 *
 * What is here and why:
 *  1. the OpenCL work-item / math built-ins the compiled kernel text leaves undefined
 *     (get_global_id ..., sqrt, pow, fmax ...) -- each is the obvious one-liner over libm;
 *  2. the run-time parameter variables that prelude.cl declares (the reference bakes the same
 *     values in as #defines, CSchemeGodunov.cpp:666-784);
 *  3. drivers that walk an NDRange (x fastest, then y) and call the kernel once per work-item:
 *     serially by default; the kernels whose work-items are independent can be spread over host
 *     threads by rows (ref_set_threads) the way a CPU OpenCL runtime spreads work-groups over cores --
 *     bench.py's cpu_baseline uses that.  The in-place MUSCL corrector and the boundary kernels stay
 *     serial (their result depends on the order).  The launch geometry of each driver follows the reference's host code, cited
 *     per function.  No arithmetic of the scheme is restated here.
 *
 * Built three times per precision: -DREF_GODUNOV (Godunov program) / -DREF_MUSCL (MUSCL program) /
 * -DREF_INERTIAL (partial-inertial program).
 */
#include "../../hipims-ocl_amd/csrc/hp_crmath.h"   // shared with the oracle and the STRICT HIP kernels (see there)
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <omp.h>
#include <ucontext.h>
#include <vector>

#ifdef REF_FP32
typedef float real;
#else
typedef double real;
#endif

// ---------------------------------------------------------------------------------------------
// 1. built-ins (C++ linkage on purpose: the kernel object references the Itanium-mangled names)
// ---------------------------------------------------------------------------------------------
static thread_local size_t t_gid[3], t_lid[3], t_grp[3], t_lsz[3] = {1, 1, 1}, t_gsz[3] = {1, 1, 1};

size_t get_global_id(unsigned d)   { return t_gid[d]; }
size_t get_local_id(unsigned d)    { return t_lid[d]; }
size_t get_group_id(unsigned d)    { return t_grp[d]; }
size_t get_local_size(unsigned d)  { return t_lsz[d]; }
size_t get_global_size(unsigned d) { return t_gsz[d]; }
// barrier(): a no-op for the kernels driven one work-item at a time (groups of 1 x 1 x 1).  The one kernel driven as a real
// work-group -- mch_1st_cachePrediction, ref_mch_1st_cached below -- runs its work-items as coroutines (ucontext fibers) of
// the calling thread: barrier() hands control back to the group's scheduler, which resumes a work-item only after every
// work-item of the group has arrived -- the semantics of the OpenCL barrier, deterministically and without 256 kernel threads.
static thread_local ucontext_t* t_fiber = nullptr;      // the running work-item's context (nullptr: not inside a group)
static thread_local ucontext_t* t_sched = nullptr;      // the group scheduler's context
static thread_local int         t_at_barrier = 0;
static thread_local int         t_omp_group = 0;        // this thread is one work-item of an OpenMP team run as ONE work-group (ref_reduce)
void   barrier(unsigned)
{
	if (t_omp_group) {                                    // every thread of the team is a work-item of the group and comes through here
		#pragma omp barrier
		return;
	}
	if (!t_fiber) return;
	t_at_barrier = 1;
	swapcontext(t_fiber, t_sched);                        // resumed (ids restored by the scheduler) once the whole group is here
}

long   max(long a, long b)         { return a > b ? a : b; }
long   min(long a, long b)         { return a < b ? a : b; }

double sqrt(double x)              { return __builtin_sqrt(x); }
// pow: the one built-in whose result is implementation defined (<= 16 ulp in OpenCL).  The reference calls it with three
// exponents only -- 1.0/3.0 (CLFriction.clc:43, CLBoundaries.clc:81), 10.0/3.0 (CLSchemeInertial.clc:352) and 2
// (CLBoundaries.clc:81) -- and this "device" answers with the libm-independent routines of hp_crmath.h (correctly rounded
// cube root; x^3 * cbrt(x); x * x), so that the reference kernels, the C oracle and the STRICT HIP kernels can agree bit
// for bit.  Any other exponent would be a use this build does not know about: it stops.
double pow(double x, double y)
{
#ifdef REF_LIBM_POW
	return __builtin_pow(x, y);       // a DIFFERENT conforming pow (this host's libm): the *_libm builds measure how far two
	                                  // platforms' runs of the reference itself drift apart through this one built-in
#endif
	if (y == 1.0 / 3.0)  return hp_cr_cbrt(x);
	if (y == 10.0 / 3.0) return hp_cr_pow103(x);
	if (y == 2.0)        return x * x;
	__builtin_trap();
}
double pown(double x, int n)       { return __builtin_powi(x, n); }
double fabs(double x)              { return __builtin_fabs(x); }
double fmax(double a, double b)    { return __builtin_fmax(a, b); }
double fmin(double a, double b)    { return __builtin_fmin(a, b); }
double fmod(double a, double b)    { return __builtin_fmod(a, b); }
double floor(double x)             { return __builtin_floor(x); }
double max(double a, double b)     { return __builtin_fmax(a, b); }
double min(double a, double b)     { return __builtin_fmin(a, b); }

float  sqrt(float x)               { return __builtin_sqrtf(x); }
float  pow(float x, float y)
{
	if (y == 1.0f / 3.0f)  return hp_cr_cbrtf(x);
	if (y == 10.0f / 3.0f) return hp_cr_pow103f(x);
	if (y == 2.0f)         return x * x;
	__builtin_trap();
}
float  pown(float x, int n)        { return __builtin_powif(x, n); }
float  fabs(float x)               { return __builtin_fabsf(x); }
float  fmax(float a, float b)      { return __builtin_fmaxf(a, b); }
float  fmin(float a, float b)      { return __builtin_fminf(a, b); }
float  fmod(float a, float b)      { return __builtin_fmodf(a, b); }
float  floor(float x)              { return __builtin_floorf(x); }
float  max(float a, float b)       { return __builtin_fmaxf(a, b); }
float  min(float a, float b)       { return __builtin_fminf(a, b); }

// ---------------------------------------------------------------------------------------------
// 2. run-time parameters
// ---------------------------------------------------------------------------------------------
extern "C" {
real     REFP_VERY_SMALL = (real)1e-10, REFP_QUITE_SMALL = (real)1e-9, REFP_DELTAX = 1, REFP_DELTAY = 1;
real     REFP_ENDTIME = (real)1e30, REFP_OUTPUTTIME = (real)1e30, REFP_COURANT = (real)0.5, REFP_FIXED_DT = 0;
long     REFP_COLS = 0, REFP_ROWS = 0, REFP_CELLCOUNT = 0;
unsigned REFP_WORKERS = 1;

// ---- the reference kernels (defined in the compiled .clc text) ----
void tst_Reduce(real* state, const real* bed, real* scratch);
void tst_Advance_Normal(real* t, real* dt, real* thydro, real* scratch, real* state, real* bed,
                        real* tsync, real* batchdt, unsigned* ok, unsigned* skipped);
void tst_UpdateTimestep(real* t, real* dt, real* scratch, real* tsync, real* batchdt);
void tst_ResetCounters(real* batchdt, unsigned* ok, unsigned* skipped);
void per_Friction(const real* dt, real* state, real* bed, real* manning, real* time);
void bdy_Uniform(const void* cfg, const real* series, real* t, real* dt, real* thydro,
                 real* state, real* bed, real* manning);
void bdy_Gridded(const void* cfg, const real* series, real* t, real* dt, real* thydro,
                 real* state, real* bed, real* manning);
void bdy_Cell(const void* cfg, const uint64_t* relations, const real* series, real* t, real* dt,
              real* thydro, real* state, real* bed, real* manning);
#ifdef REF_GODUNOV
void gts_cacheDisabled(const real* dt, const real* bed, real* src, real* dst, const real* manning);
#endif
#ifdef REF_INERTIAL
void ine_cacheDisabled(const real* dt, const real* bed, real* src, real* dst, const real* manning);
#endif
#ifdef REF_MUSCL
void mch_1st_cacheNone(const real* dt, const real* bed, real* state, real* fN, real* fE, real* fS, real* fW);
void mch_2nd_cacheNone(const real* dt, real* state, const real* bed, const real* manning,
                       real* fN, real* fE, real* fS, real* fW);
void mch_1st_cachePrediction(const real* dt, const real* bed, real* state, real* fN, real* fE, real* fS, real* fW);
#endif

// ---------------------------------------------------------------------------------------------
// 3. drivers
// ---------------------------------------------------------------------------------------------
int ref_real_bytes(void) { return (int)sizeof(real); }

static int g_threads = 1;
void ref_set_threads(int n) { g_threads = n > 0 ? n : 1; }
#define REF_GROUPSIZE_MAX 64                                /* == TIMESTEP_GROUPSIZE of prelude.cl (the scratch array's size) */

/* CSchemeGodunov.cpp:666-784 -- the constants every kernel is compiled against.
 * `workers` = TIMESTEP_WORKERS = reduction GLOBAL size (CSchemeGodunov.cpp:764, quirk Q5). */
void ref_configure(long cols, long rows, double dx, double very_small, double courant,
                   double endtime, unsigned workers, double fixed_dt)
{
	REFP_COLS = cols; REFP_ROWS = rows; REFP_CELLCOUNT = cols * rows;
	REFP_DELTAX = (real)dx; REFP_DELTAY = (real)dx;
	REFP_VERY_SMALL = (real)very_small; REFP_QUITE_SMALL = (real)(very_small * 10);   /* CSchemeGodunov.cpp:522-523 */
	REFP_COURANT = (real)courant; REFP_ENDTIME = (real)endtime; REFP_OUTPUTTIME = (real)endtime;
	REFP_WORKERS = workers; REFP_FIXED_DT = (real)fixed_dt;
}

static inline void set_item_2d(size_t x, size_t y, size_t gx, size_t gy)
{
	t_gid[0] = x; t_gid[1] = y; t_gid[2] = 0;
	t_grp[0] = x; t_grp[1] = y; t_grp[2] = 0;
	t_lid[0] = t_lid[1] = t_lid[2] = 0;
	t_lsz[0] = t_lsz[1] = t_lsz[2] = 1;
	t_gsz[0] = gx; t_gsz[1] = gy; t_gsz[2] = 1;
}

#ifdef REF_GODUNOV
/* global size = cols x rows rounded up to the group (CSchemeGodunov.cpp:636-637, :958-965);
 * work-items beyond the grid return at the bounds guard, so cols x rows is equivalent. */
void ref_gts(const real* dt, const real* bed, real* src, real* dst, const real* manning)
{
	#pragma omp parallel for schedule(static) num_threads(g_threads)
	for (long y = 0; y < REFP_ROWS; ++y)
		for (long x = 0; x < REFP_COLS; ++x) {
			set_item_2d((size_t)x, (size_t)y, (size_t)REFP_COLS, (size_t)REFP_ROWS);
			gts_cacheDisabled(dt, bed, src, dst, manning);
		}
}
#endif

#ifdef REF_INERTIAL
/* same launch geometry as the Godunov kernel: CSchemeInertial.cpp:269-271 reuses prepare1OExecDimensions */
void ref_ine(const real* dt, const real* bed, real* src, real* dst, const real* manning)
{
	#pragma omp parallel for schedule(static) num_threads(g_threads)
	for (long y = 0; y < REFP_ROWS; ++y)
		for (long x = 0; x < REFP_COLS; ++x) {
			set_item_2d((size_t)x, (size_t)y, (size_t)REFP_COLS, (size_t)REFP_ROWS);
			ine_cacheDisabled(dt, bed, src, dst, manning);
		}
}
#endif

#ifdef REF_MUSCL
void ref_mch_1st(const real* dt, const real* bed, real* state, real* fN, real* fE, real* fS, real* fW)
{
	#pragma omp parallel for schedule(static) num_threads(g_threads)
	for (long y = 0; y < REFP_ROWS; ++y)
		for (long x = 0; x < REFP_COLS; ++x) {
			set_item_2d((size_t)x, (size_t)y, (size_t)REFP_COLS, (size_t)REFP_ROWS);
			mch_1st_cacheNone(dt, bed, state, fN, fE, fS, fW);
		}
}
/* The reference's DEFAULT predictor (ucConfiguration = kCachePrediction, CSchemeMUSCLHancock.cpp:46, kernels :523-531):
 * mch_1st_cachePrediction, work-groups of MCH_STG1_DIM1 x MCH_STG1_DIM2 = 16 x 16 work-items (floor(sqrt(256)),
 * CSchemeMUSCLHancock.cpp:361-373) that overlap by two cells, one __local tile per group and a barrier between filling and
 * reading it (CLSchemeMUSCLHancock.clc:176-296).  Run here as a REAL work-group: 256 coroutines, one per work-item, that all
 * run up to the barrier before any of them runs beyond it; the kernel's __local array is one static object of the compiled
 * text, shared by the work-items exactly as LDS is shared by a group.  Groups run one after the other.  Global size =
 * ceil(n * 16/14) rounded up to the group size (CSchemeMUSCLHancock.cpp:375-380, COCLKernel.cpp:343-344). */
struct GroupCall { const real* dt; const real* bed; real* state; real* f[4]; int done; };
static thread_local GroupCall* t_call = nullptr;
static void work_item_entry()
{
	GroupCall* c = t_call;
	mch_1st_cachePrediction(c->dt, c->bed, c->state, c->f[0], c->f[1], c->f[2], c->f[3]);
	c->done = 1;                                          /* returns to the scheduler through uc_link */
}
void ref_mch_1st_cached(const real* dt, const real* bed, real* state, real* fN, real* fE, real* fS, real* fW)
{
	/* one call at a time per library: the kernel's __local tile is ONE static object of the compiled text (there is no per-thread
	 * copy to be had), so two simulations stepping MUSCL from different threads take turns here */
	static std::mutex one_group_at_a_time;
	std::lock_guard<std::mutex> lock(one_group_at_a_time);
	const size_t L = 16;                                  /* == MCH_STG1_DIM1/2 of prelude.cl */
	auto gsize = [&](long n) {
		size_t g = (size_t)__builtin_ceil((double)n * ((double)L / (double)(L - 2)));
		return (size_t)(__builtin_ceil((double)g / (double)L) * (double)L);
	};
	const size_t gsx = gsize(REFP_COLS), gsy = gsize(REFP_ROWS), ngx = gsx / L, ngy = gsy / L, items = L * L;
	const size_t STACK = 128 * 1024;
	static thread_local std::vector<char> stacks;         /* allocated once per calling thread (re-zeroing 32 MiB per call would dominate); thread_local
	                                                         like the scheduler state around it: two simulations stepping MUSCL from different threads
	                                                         must not run their fibers on the same stacks */
	if (stacks.size() < items * STACK) stacks.resize(items * STACK);
	std::vector<ucontext_t> ctx(items);
	std::vector<GroupCall> call(items);
	ucontext_t sched;
	auto set_ids = [&](size_t gx, size_t gy, size_t lid) {
		const size_t lx = lid % L, ly = lid / L;
		t_gid[0] = gx * L + lx; t_gid[1] = gy * L + ly; t_gid[2] = 0;
		t_grp[0] = gx; t_grp[1] = gy; t_grp[2] = 0;
		t_lid[0] = lx; t_lid[1] = ly; t_lid[2] = 0;
		t_lsz[0] = L; t_lsz[1] = L; t_lsz[2] = 1;
		t_gsz[0] = gsx; t_gsz[1] = gsy; t_gsz[2] = 1;
	};
	t_sched = &sched;
	for (size_t gy = 0; gy < ngy; ++gy)
		for (size_t gx = 0; gx < ngx; ++gx) {
			for (size_t i = 0; i < items; ++i) {
				call[i] = GroupCall{dt, bed, state, {fN, fE, fS, fW}, 0};
				getcontext(&ctx[i]);
				ctx[i].uc_stack.ss_sp = &stacks[i * STACK];
				ctx[i].uc_stack.ss_size = STACK;
				ctx[i].uc_link = &sched;
				makecontext(&ctx[i], work_item_entry, 0);
			}
			/* rounds: every unfinished work-item runs to its next barrier (or to its end); the next round starts only when all
			 * of them have -- nobody passes a barrier before everybody has reached it */
			for (bool any = true; any;) {
				any = false;
				for (size_t i = 0; i < items; ++i) {
					if (call[i].done) continue;
					set_ids(gx, gy, i);
					t_call = &call[i]; t_fiber = &ctx[i]; t_at_barrier = 0;
					swapcontext(&sched, &ctx[i]);
					any = any || !call[i].done;
				}
			}
		}
	t_fiber = nullptr; t_sched = nullptr; t_call = nullptr;
}
/* In-place corrector: result depends on work-item order (quirk Q6).  Driven row-major, x fastest. */
void ref_mch_2nd(const real* dt, real* state, const real* bed, const real* manning,
                 real* fN, real* fE, real* fS, real* fW)
{
	for (long y = 0; y < REFP_ROWS; ++y)
		for (long x = 0; x < REFP_COLS; ++x) {
			set_item_2d((size_t)x, (size_t)y, (size_t)REFP_COLS, (size_t)REFP_ROWS);
			mch_2nd_cacheNone(dt, state, bed, manning, fN, fE, fS, fW);
		}
}
#endif

/* 1-D, global size = TIMESTEP_WORKERS.  One thread: groups of one work-item (group id == global id), the LDS tree degenerates
 * to a copy.  Several threads: the kernel's __local scratch array is ONE static object of the compiled text, so independent
 * work-items on different threads would race on it -- the store of a work-item's maximum and the read-back four statements
 * later (round 5: found as a timestep that ignored the fastest cell, one call in ~5000, by running config C1 to its end).
 * The team therefore runs as the work-group the kernel was written for: N threads = N work-items of one group, barrier() is a
 * real barrier, the tree reduction does its job; groups one after the other.  N = the largest power of two <= the thread count
 * (the tree needs one), TIMESTEP_GROUPSIZE in prelude.cl bounds it.  The maximum is exact and order independent: same result. */
void ref_reduce(real* state, const real* bed, real* scratch)
{
	int n = 1;
	while (n * 2 <= g_threads && n * 2 <= REF_GROUPSIZE_MAX && (size_t)(n * 2) <= (size_t)REFP_WORKERS && REFP_WORKERS % (unsigned)(n * 2) == 0) n *= 2;
	if (n == 1) {
		for (size_t g = 0; g < (size_t)REFP_WORKERS; ++g) {
			t_gid[0] = g; t_grp[0] = g; t_lid[0] = 0; t_lsz[0] = 1; t_gsz[0] = REFP_WORKERS;
			t_gid[1] = t_gid[2] = 0;
			tst_Reduce(state, bed, scratch);
		}
		return;
	}
	const size_t groups = (size_t)REFP_WORKERS / (size_t)n;
	#pragma omp parallel num_threads(n)
	{
		const size_t lid = (size_t)omp_get_thread_num();
		t_omp_group = 1;
		for (size_t grp = 0; grp < groups; ++grp) {
			t_gid[0] = grp * (size_t)n + lid; t_grp[0] = grp; t_lid[0] = lid; t_lsz[0] = (size_t)n; t_gsz[0] = REFP_WORKERS;
			t_gid[1] = t_gid[2] = 0;
			tst_Reduce(state, bed, scratch);
			#pragma omp barrier                               // the next group reuses the scratch array
		}
		t_omp_group = 0;
	}
}

void ref_advance(real* t, real* dt, real* thydro, real* scratch, real* state, real* bed,
                 real* tsync, real* batchdt, unsigned* ok, unsigned* skipped)
{
	set_item_2d(0, 0, 1, 1);
	tst_Advance_Normal(t, dt, thydro, scratch, state, bed, tsync, batchdt, ok, skipped);
}

void ref_update_timestep(real* t, real* dt, real* scratch, real* tsync, real* batchdt)
{
	set_item_2d(0, 0, 1, 1);
	tst_UpdateTimestep(t, dt, scratch, tsync, batchdt);
}

/* Global size floor(cols/8)*8 x floor(rows/8)*8 : integer division inside ceil(), applied before
 * the 8x8 group size is set (CBoundaryUniform.cpp:294-295, CBoundaryGridded.cpp:298-299 -- quirk Q9).
 * `full_range` != 0 covers the whole grid instead. */
void ref_bdy_uniform(const void* cfg, const real* series, real* t, real* dt, real* thydro,
                     real* state, real* bed, real* manning, int full_range)
{
	long gx = full_range ? REFP_COLS : (REFP_COLS / 8) * 8;
	long gy = full_range ? REFP_ROWS : (REFP_ROWS / 8) * 8;
	for (long y = 0; y < gy; ++y)
		for (long x = 0; x < gx; ++x) {
			set_item_2d((size_t)x, (size_t)y, (size_t)gx, (size_t)gy);
			bdy_Uniform(cfg, series, t, dt, thydro, state, bed, manning);
		}
}

void ref_bdy_gridded(const void* cfg, const real* series, real* t, real* dt, real* thydro,
                     real* state, real* bed, real* manning, int full_range)
{
	long gx = full_range ? REFP_COLS : (REFP_COLS / 8) * 8;
	long gy = full_range ? REFP_ROWS : (REFP_ROWS / 8) * 8;
	for (long y = 0; y < gy; ++y)
		for (long x = 0; x < gx; ++x) {
			set_item_2d((size_t)x, (size_t)y, (size_t)gx, (size_t)gy);
			bdy_Gridded(cfg, series, t, dt, thydro, state, bed, manning);
		}
}

/* 1-D over the relation list (CBoundaryCell.cpp:442-443: global = (n/8+1)*8, group 8; items >= n return). */
void ref_bdy_cell(const void* cfg, const uint64_t* relations, const real* series, real* t, real* dt,
                  real* thydro, real* state, real* bed, real* manning, long count)
{
	for (long i = 0; i < count; ++i) {
		set_item_2d((size_t)i, 0, (size_t)count, 1);
		bdy_Cell(cfg, relations, series, t, dt, thydro, state, bed, manning);
	}
}

} // extern "C"
