/*
 * hipims_mi.h -- C ABI of the MI355X-native shallow-water step engine (libhipims_mi.so).
 *
 * This is the drop-in boundary for ONE path of HiPIMS-OCL (lukeshope/hipims-ocl): the per-timestep
 * update that `CSchemeGodunov` / `CSchemeMUSCLHancock` / `CSchemeInertial` drive through the OpenCL executor classes
 * (`COCLProgram`, `COCLKernel`, `COCLBuffer`, `COCLDevice`).  The HIP engine owns the kernel graph,
 * so the reference's generic program/kernel/argument layer collapses into semantic calls; each
 * entry point below cites the reference interface it replaces (paths relative to the reference's
 * `src/`).  INTEGRATION.md shows the binding a HiPIMS maintainer would add.
 *
 * Conventions
 *  - plain C, plain pointers and sizes; no C++/torch types cross this boundary;
 *  - every function returns 0 (HP_OK) or a negative hp_status; hp_last_error() gives the message of
 *    the calling thread's last failure.  Nothing throws, nothing calls exit() (the reference's
 *    model::doError levels, main.cpp:631-652, are left to the host);
 *  - host arrays use the reference's layout unchanged (Domain/CDomain.h:26-33, CDomain.cpp:171-180):
 *    state = cols*rows x {Z free-surface level, Zmax, Qx, Qy}, row-major, row 0 = south;
 *    bed / manning = cols*rows scalars.  Element type = double (precision 8) or float (precision 4);
 *  - transfers and steps are ASYNCHRONOUS on the domain's HIP stream, exactly like the reference's
 *    non-blocking queueWrite / queueRead / scheduleExecution: host memory passed to upload/download
 *    must stay alive until hp_sync() (COCLDevice::blockUntilFinished) returns;
 *  - threading contract = the reference's (Schemes/CSchemeGodunov.cpp:1116-1370): one worker thread
 *    per domain may call step/read/sync; hp_is_busy may be called from another thread; everything
 *    else requires the domain to be idle.
 *  - there is NO CPU fallback: every call fails with HP_ERR_NO_DEVICE when no HIP device is usable.
 */
#ifndef HIPIMS_MI_H
#define HIPIMS_MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HP_ABI_VERSION 2          /* 2: hp_domain_desc_t.ghost_rows */

typedef enum {
	HP_OK              =  0,
	HP_ERR_INVALID     = -1,    /* bad argument / descriptor */
	HP_ERR_NO_DEVICE   = -2,    /* no usable HIP device (never falls back to the CPU) */
	HP_ERR_HIP         = -3,    /* a HIP runtime call failed; see hp_last_error() */
	HP_ERR_UNSUPPORTED = -4,
	HP_ERR_STATE       = -5     /* call not valid in the domain's current state */
} hp_status;

/* Schemes: Schemes/CScheme.cpp:141-190 (createFromConfig "Godunov" / "MUSCL-Hancock") */
enum { HP_SCHEME_GODUNOV = 0, HP_SCHEME_MUSCL_HANCOCK = 1, HP_SCHEME_INERTIAL = 2 };   /* model::schemeTypes, CScheme.h:33-37 */

/* which array an upload / download moves: COCLBuffer "Cell states", "Bed elevations",
 * "Manning coefficients" (Schemes/CSchemeGodunov.cpp:832-845) */
enum { HP_ARRAY_STATE = 0, HP_ARRAY_BED = 1, HP_ARRAY_MANNING = 2 };

/* Behaviours of the reference a parity run reproduces; each can be switched off.
 * HP_QUIRKS_REFERENCE is the default and is what the parity tests run. */
enum {
	HP_QUIRK_CFL_READS_PRIMARY = 1u << 0,  /* Q1: tst_Reduce always reads the primary state buffer
	                                          (CSchemeGodunov.cpp:1629/:1634 set a non-existent arg) */
	HP_QUIRK_BDY_TRUNCATED     = 1u << 1,  /* Q9: boundary kernels cover floor(n/8)*8 cells per axis
	                                          (Boundaries/CBoundaryUniform.cpp:294-295) */
	HP_QUIRK_MUSCL_NEIGHBOUR_Y_IS_BED = 1u << 2,
	                                       /* Q11: the reference's DEFAULT MUSCL-Hancock configuration is kCachePrediction
	                                          (Schemes/CSchemeMUSCLHancock.cpp:46): mch_1st_cachePrediction keeps {Z, BED, Qx, Qy}
	                                          in LDS (CLSchemeMUSCLHancock.clc:201) and hands the four neighbours to mch_1st with
	                                          their bed in .y (:232-239), so the predictor's first-order fallback
	                                          `pNeigData*.y <= -9998.0` (:325-330) tests the neighbour's BED -- a cell next to a
	                                          disabled (Zmax = -9999) cell with an ordinary bed stays second order.  Off = the
	                                          non-default mch_1st_cacheNone variant, which tests the neighbour's Zmax (:109-125). */
	HP_QUIRKS_REFERENCE        = 7u
};

/* Arithmetic flavour of the device kernels.
 *  STRICT: the reference's expression order, IEEE division/sqrt, no FMA contraction -- bit-identical to
 *          the oracle and to the reference's own kernels as built by oracle/ref_build, friction included: pow(), the one
 *          built-in whose result is implementation defined, is in all three the correctly rounded cube root of
 *          csrc/hp_crmath.h (i.e. the identity holds against the reference run on a platform with THAT pow; a
 *          platform with another conforming pow differs in the last bit of the friction term, fixture f10_*_libm);
 *  FAST  : algebraically identical, fewer divisions, explicit FMAs (the reference itself ships with
 *          -cl-mad-enable, OpenCL/Executors/COCLProgram.cpp:73, so its own results are compiler
 *          dependent at this level).  Default. */
enum { HP_MATH_FAST = 0, HP_MATH_STRICT = 1 };

/* Kernel selection (diagnostics): AUTO = the tuned kernel; BASIC = one thread per cell, four face
 * solves per cell -- the same dataflow as gts_cacheDisabled, kept as an on-device cross-check. */
enum { HP_KERNEL_AUTO = 0, HP_KERNEL_BASIC = 1 };

typedef struct hp_domain hp_domain_t;

/* ---- device discovery: COCLDevice capability fields (OpenCL/Executors/COCLDevice.h:53-99),
 *      CExecutorControlOpenCL::createDevices (CExecutorControlOpenCL.cpp:211) ---- */
typedef struct {
	char     name[128];          /* clDeviceName */
	char     arch[32];           /* gcnArchName, e.g. "gfx950" */
	int32_t  compute_units;      /* clDeviceComputeUnits */
	int32_t  clock_mhz;          /* clDeviceClockFrequency */
	uint64_t global_mem_bytes;   /* clDeviceGlobalMemSize */
	uint64_t lds_bytes_per_cu;   /* clDeviceLocalSize */
	int32_t  wavefront;          /* 64 on gfx950 */
	int32_t  fp64;               /* COCLDevice::isDoubleCompatible */
} hp_device_info_t;

int hp_abi_version(void);
int hp_device_count(int* count);
int hp_device_info(int device, hp_device_info_t* info);
const char* hp_last_error(void);

/* Optional log sink, the place model::doError (src/main.cpp:631-652 -> CLog::writeError) has in the reference: every
 * failing call hands its message to `sink` with the reference's error level (common.h:62-68; failures of this
 * library are kLevelModelStop = 2) before returning its code, so an adapter can forward to pManager->log without
 * polling hp_last_error().  Process-wide; NULL removes it.  Called on the thread that made the failing call. */
#define HP_LOG_FATAL        1
#define HP_LOG_MODEL_STOP   2
#define HP_LOG_CONTINUE     4
#define HP_LOG_WARNING      8
#define HP_LOG_INFORMATION 16
typedef void (*hp_log_sink_t)(int level, const char* message, void* user);
int hp_set_log_sink(hp_log_sink_t sink, void* user);

/* ---- domain: CScheme::prepareAll + the registerConstant set
 *      (Schemes/CSchemeGodunov.cpp:386-470, :666-784; XML parameters :128-333, CScheme.cpp:46-55) ---- */
typedef struct {
	uint32_t struct_size;        /* = sizeof(hp_domain_desc_t) */
	int32_t  device;             /* <domain deviceNumber=...> (Domain/CDomainManager.cpp:203-220) */
	int64_t  cols, rows;         /* LOCAL array size (all rows this rank stores, ghosts included) */
	double   dx;                 /* cell resolution; DOMAIN_DELTAX == DOMAIN_DELTAY */
	int32_t  precision;          /* 8 (fp64) or 4 (fp32): <simulation floatingPointPrecision> */
	int32_t  scheme;             /* HP_SCHEME_* */
	double   courant;            /* courantNumber, default 0.5 */
	double   dry_threshold;      /* dryThreshold = VERY_SMALL, default 1e-10; QUITE_SMALL = 10x */
	int32_t  friction;           /* frictionEffects (fused into the flux kernel) */
	int32_t  dynamic_dt;         /* timestepMode: 1 = CFL, 0 = fixed */
	double   dt_fixed;           /* TIMESTEP_FIXED */
	double   dt_initial;         /* timestepInitial, default 0.001 */
	double   t_end;              /* simulation duration = SCHEME_ENDTIME */
	uint32_t quirks;             /* HP_QUIRK_* mask */
	int32_t  math_mode;          /* HP_MATH_* */
	int32_t  kernel;             /* HP_KERNEL_* */
	/* 1-D row-strip decomposition (replaces Domain/Links/CDomainLink + MPI).
	 * A single-GPU domain has global_rows == rows and row_offset == 0. */
	int64_t  global_rows;        /* rows of the whole logical grid */
	int64_t  row_offset;         /* global row index of local row 0 */
	/* Ghost rows stored per interior side of a strip: 0 = one stencil reach g (1 row Godunov / inertial, 2 MUSCL-Hancock) and a
	 * ghost-row exchange after every iteration; 2g = two reaches and an exchange after every SECOND iteration (the strip
	 * recomputes g of its neighbour's rows redundantly in between; hp_strip_step_batch only).  What is left of the reference's
	 * wide overlap zones with several iterations between synchronisations (Domain/Links/CDomainLink.cpp:297-328,
	 * Domain/CDomainBase.cpp:163-174), here without forecast and rollback: results stay bit-identical to the single domain. */
	int32_t  ghost_rows;
	int32_t  reserved0;
} hp_domain_desc_t;

void hp_domain_desc_default(hp_domain_desc_t* desc);   /* reference defaults, CScheme.cpp:46-55 */
int  hp_domain_create(const hp_domain_desc_t* desc, hp_domain_t** out);
int  hp_domain_destroy(hp_domain_t* d);                /* CSchemeGodunov::release1OResources, :993-1048 */

/* COCLBuffer::queueWriteAll (COCLBuffer.h:40): HP_ARRAY_STATE fills BOTH ping-pong buffers, as
 * prepareSimulation does (CSchemeGodunov.cpp:1064-1065), and resets the ping-pong phase. */
int hp_domain_upload(hp_domain_t* d, int which, const void* host, size_t bytes);
/* COCLBuffer::queueReadAll / queueReadPartial (COCLBuffer.h:38-43); for HP_ARRAY_STATE reads the buffer
 * the NEXT iteration would use as its source (CSchemeGodunov::getNextCellSourceBuffer, :1705-1715,
 * which is what readDomainAll / saveCurrentState / CDomainLink::pullFromBuffer read). */
int hp_domain_download(hp_domain_t* d, int which, void* host, int64_t row0, int64_t nrows);
/* COCLBuffer::queueWritePartial into the next source buffer (CDomainLink::pushToBuffer, CDomainLink.cpp:252-270) */
int hp_domain_upload_rows(hp_domain_t* d, const void* host, int64_t row0, int64_t nrows);

/* Device-side checkpoint: what saveCurrentState + rollbackSimulation do through host memory (CSchemeGodunov.cpp:1720-1736,
 * :1474-1518), kept in HBM instead (two more copies of a 4096^2 fp64 state are 1 GB of 288).  hp_state_save copies BOTH
 * ping-pong buffers and the time-control block; hp_state_restore puts each buffer, the time-control block, the ping-pong phase
 * and the remembered CFL maxima back: the steps that follow repeat the original ones bit for bit.  (The reference's rollback
 * writes the saved next-source state into both buffers; cells whose whole neighbourhood is dry, which the flux kernel leaves
 * untouched -- quirk Q3 --, then continue from other values than in the original run.  A host that wants exactly that uploads
 * the state it downloaded, as CSchemeMI::rollbackSimulation does.)  The host may then adjust time / target / timestep with the calls
 * below, exactly as the reference's rollback sequence does. */
int hp_state_save(hp_domain_t* d);
int hp_state_restore(hp_domain_t* d);

/* ---- boundaries: CBoundaryUniform / CBoundaryGridded::prepareBoundary
 *      (Boundaries/CBoundaryUniform.cpp:180-296, CBoundaryGridded.cpp:166-300); the arrays are the packed
 *      buffers those functions build.  Applied in the order added (the reference's order is unspecified, Q7) */
enum { HP_UNIFORM_RAIN_INTENSITY = 0, HP_UNIFORM_LOSS_RATE = 1 };           /* Boundaries/CLBoundaries.clh:44-45 */
enum { HP_GRIDDED_RAIN_INTENSITY = 0, HP_GRIDDED_RAIN_ACCUMUL = 1, HP_GRIDDED_MASS_FLUX = 2 };   /* :47-50 */
/* series: `entries` x {time, value} pairs (value in mm/h), element type = domain precision */
int hp_boundary_add_uniform(hp_domain_t* d, int definition, const void* series, uint32_t entries,
                            double interval, double length);
/* grids: [entries][grid_rows][grid_cols] rates (mm/h), element type = domain precision */
int hp_boundary_add_gridded(hp_domain_t* d, int definition, const void* grids, uint64_t entries,
                            uint64_t grid_rows, uint64_t grid_cols, double resolution,
                            double offset_x, double offset_y, double interval);
/* CBoundaryCell::prepareBoundary (Boundaries/CBoundaryCell.cpp:300-445) -- "next" row N2 of SURVEY.md 8(f).
 * cells: flat ids y*cols + x in the GLOBAL grid (CDomainCartesian::getCellID); series: `entries` x {time, depth or
 * level, Qx, Qy}, already divided by the cell count for "total" discharges as the reference's packer does (:398-402). */
enum { HP_DEPTH_IGNORE = 0, HP_DEPTH_IS_FSL = 1, HP_DEPTH_IS_DEPTH = 2, HP_DEPTH_IS_CRITICAL = 3 };       /* CLBoundaries.clh:34-37 */
enum { HP_DISCHARGE_IGNORE = 0, HP_DISCHARGE_IS_DISCHARGE = 1, HP_DISCHARGE_IS_VELOCITY = 2, HP_DISCHARGE_IS_VOLUME = 3 };   /* :39-42 */
int hp_boundary_add_cell(hp_domain_t* d, int depth_definition, int discharge_definition, const uint64_t* cells,
                         uint64_t count, const void* series, uint64_t entries, double interval, double length);
int hp_boundary_clear(hp_domain_t* d);
/* Are the domain's area boundaries carried by the flux kernel itself (rain / loss of iteration n+1 added to the state
 * iteration n stores, no separate pass over the grid)?  True for the Godunov scheme with up to three uniform / gridded
 * boundaries, no cell boundary among them, and rain grids of at least 64 model cells per grid cell; everything else runs
 * the boundary kernels as separate launches like the reference (CSchemeGodunov.cpp:1637-1643).  Results are identical
 * either way; bench.py reports it. */
int hp_boundaries_fused(hp_domain_t* d, int* fused);

/* ---- time control ---- */
int hp_set_target_time(hp_domain_t* d, double t);     /* CScheme::setTargetTime -> "Target time (sync)" buffer (:1166-1176) */
int hp_set_time(hp_domain_t* d, double t);            /* rewrite the "Time" buffer: rollbackSimulation (:1480-1494) */
int hp_force_timestep(hp_domain_t* d, double dt);     /* CSchemeGodunov::forceTimestep (:1803-1811) + write (:1213-1232) */
int hp_reset_counters(hp_domain_t* d);                /* tst_ResetCounters (CLDynamicTimestep.clc:151-161) */
int hp_update_timestep(hp_domain_t* d);               /* tst_Reduce + tst_UpdateTimestep (:1189-1195, :1254-1260) */

/* ---- the hot loop: `n` x CScheme*::scheduleIteration (CSchemeGodunov.cpp:1617-1666,
 *      CSchemeMUSCLHancock.cpp:646-680), enqueued without blocking ---- */
int hp_step_batch(hp_domain_t* d, uint32_t n_iterations);

typedef struct {
	double   time;               /* "Time" buffer */
	double   timestep;           /* "Timestep" buffer; negative = suspended at the sync point */
	double   time_hydrological;
	double   time_target;
	double   batch_timesteps;    /* "Batch timesteps cumulative" */
	uint32_t batch_successful;   /* "Batch successful iterations" */
	uint32_t batch_skipped;      /* "Batch skipped iterations" */
	uint64_t cells_calculated;   /* ulCurrentCellsCalculated: cols*rows per queued iteration (:1299) */
	uint64_t iterations;
} hp_scalars_t;

/* CSchemeGodunov::readKeyStatistics (:1817-1835) after the five queueReadAll calls (:1309-1313); BLOCKS */
int hp_read_scalars(hp_domain_t* d, hp_scalars_t* out);
int hp_sync(hp_domain_t* d);                          /* COCLDevice::blockUntilFinished */
/* COCLDevice::isBusy.  With HP_STRICT_SPECULATE=1 the call that finds a speculative batch finished also looks at its verdict
 * and may re-queue the batch with the plain divisions: *busy is then 1 again (the call may ENQUEUE work; it never blocks on it). */
int hp_is_busy(hp_domain_t* d, int* busy);
/* Inside a batch an iteration is one launch whose last block waits for the flux blocks' maxima and advances the time.  That wait
 * is bounded (HP_TAIL_TIMEOUT_MS, environment, default 20000): a block that never reports -- only possible if blocks were ever
 * dispatched out of order -- freezes time and timestep instead of hanging the GPU, and hp_sync / hp_read_scalars and every later
 * call on the domain fail with HP_ERR_STATE.  (No reference counterpart: its queue is a sequence of separate kernels.) */

/* ---- multi-GPU strips (one process per GPU).  The engine never talks to other ranks itself: the host
 *      moves ghost rows and the wave-speed maximum with its collective library (RCCL through
 *      torch.distributed in this repo) on the domain's stream, between these calls.  Replaces
 *      CDomainLink::pullFromBuffer/pushToBuffer + CMPIManager (MPI/CMPIManager.cpp:555-709, :852-861). ---- */
/* Split form of one iteration: boundaries + flux kernel (+ local CFL maximum) ... */
int hp_step_begin(hp_domain_t* d);
/* ... then, after the host has all-reduced (MAX) the 8-byte value at hp_device_ptr(HP_PTR_CFL_MAX),
 * the time advance (tst_Advance_Normal) and the ping-pong flip. */
int hp_step_end(hp_domain_t* d);
/* Between hp_step_begin and hp_step_end: does this iteration carry a NEW local maximum that has to be all-reduced?
 * (0 on iterations whose reduction re-reads an unchanged buffer -- quirk Q1 -- and in fixed-timestep mode.) */
int hp_step_needs_reduction(hp_domain_t* d, int* needed);

enum {
	HP_PTR_STATE_NEXT_SRC = 0,   /* state buffer the next iteration reads (ghost rows are written here) */
	HP_PTR_STATE_OTHER    = 1,
	HP_PTR_BED            = 2,
	HP_PTR_MANNING        = 3,
	HP_PTR_CFL_MAX        = 4,   /* one element of the domain precision: local max wave speed */
	HP_PTR_SCALARS        = 5
};
int hp_device_ptr(hp_domain_t* d, int which, void** ptr);
int hp_stream(hp_domain_t* d, void** hip_stream);     /* the domain's hipStream_t (COCLDevice's command queue) */
/* Halo overlap.  When on, hp_step_begin runs the row segments that hold the rows the strip neighbours need
 * (the first/last owned rows) on a second, higher-priority stream and the interior segments on the domain's
 * stream; the domain's stream joins the second one before anything that follows (CFL maximum, hp_step_end).  The
 * host starts its halo send/recv on hp_stream_halo right after hp_step_begin -- ordered after the halo rows only,
 * so the transfer overlaps the interior compute -- and makes the domain's stream wait for the transfer before
 * hp_step_end.  Results are identical either way.  Replaces the reference's overlap-by-wide-ghost-zones scheme
 * (CDomainLink.cpp:297-328), which exists to amortise host-staged copies.  Off by default. */
int hp_set_halo_overlap(hp_domain_t* d, int on);
int hp_stream_halo(hp_domain_t* d, void** hip_stream);

/* ---- the strip loop driven from C++ (one process per GPU).  The per-iteration protocol above -- flux launches, ghost
 *      rows to and from the two strip neighbours, all-reduce(MAX) of the wave speed, time advance -- queued by the
 *      library itself on RCCL, so that a C++ host (HiPIMS's CModel / CMPIManager) needs no collective code of its own and
 *      the host cost per iteration is a handful of enqueues.  Replaces CDomainLink::pullFromBuffer / pushToBuffer
 *      (Domain/Links/CDomainLink.cpp:168-270) and CMPIManager's block exchange and MPI_Allreduce(MIN)
 *      (MPI/CMPIManager.cpp:555-709, :852-861).  Ranks are ordered south to north: rank k's neighbours are k-1 and k+1.
 *      The collective library is loaded at run time (dlopen): pass the copy the process already uses (a torch process:
 *      torch/lib/librccl.so), or NULL for the system's.  The unique id is created on rank 0 and handed to the other
 *      ranks by the host's own means (MPI_Bcast in HiPIMS, a torch broadcast in this repository's tests). ---- */
#define HP_COMM_ID_BYTES 128
int hp_comm_load(const char* rccl_library_path);   /* a path is tried alone; NULL tries librccl.so, librccl.so.1, /opt/rocm/lib/librccl.so */
int hp_comm_unique_id(void* id_out /* HP_COMM_ID_BYTES */);
int hp_strip_comm_init(hp_domain_t* d, const void* id, int rank, int world);   /* collective: every rank calls it; turns the halo overlap on unless hp_set_halo_overlap chose */
int hp_strip_step_batch(hp_domain_t* d, uint32_t n_iterations);               /* hp_step_batch for a strip */
int hp_strip_update_timestep(hp_domain_t* d);                                 /* hp_update_timestep for a strip (collective) */
int hp_strip_comm_destroy(hp_domain_t* d);
/* What the strip loop is really running on (reporting: bench.py puts it into its JSON line).  `library` is the file the
 * dynamic loader resolved the collective library to, `comm_ranks` the rank count that library reports for the domain's
 * communicator (ncclCommCount; -1 without a communicator).  d == NULL: library path only.  No reference counterpart
 * (CMPIManager logs its node count, MPI/CMPIManager.cpp:63-80). */
typedef struct {
	char    library[256];
	int32_t comm_ranks;
	int32_t comm_rank;
	int32_t halo_overlap;        /* 1: halo rows on their own stream, the transfer overlaps the interior launch */
	int32_t ghost_rows;          /* ghost rows per interior side: the stencil reach (exchange every iteration) or twice that (every second) */
	int32_t peer_max;            /* 1: the maximum over the strips travels through peer-written mailboxes, 0: through the library's all-reduce */
	int32_t peer_halo;           /* 1: the ghost rows are written by the strips into each other's state buffers, 0: sent and received through the library */
} hp_strip_info_t;
int hp_strip_info(hp_domain_t* d, hp_strip_info_t* out);

/* ---- the maximum over all strips without a collective ("one-shot all-gather-to-all over direct links, then a
 * local max"; stands where MPI_Allreduce(MIN dt) is, MPI/CMPIManager.cpp:852-861) ----
 * Every strip owns a small mailbox in uncached device memory which the other strips' GPUs write directly over xGMI; the
 * advance kernel that runs after every flux launch anyway (or the flux launch's own tail block) stores its maximum into every peer's mailbox, waits for theirs
 * and folds them -- no collective kernel on the critical path of an iteration.
 *   hp_strip_peer_ticket   allocates the mailbox and describes it and the domain's two state buffers (addresses for ranks
 *                          of the same process, IPC handles for other processes) in HP_PEER_TICKET_BYTES that the host
 *                          hands to every rank by its own means, like the communicator id;
 *   hp_strip_peer_connect  takes all ranks' tickets in rank order (count <= 64), maps the peers' mailboxes and the strip
 *                          neighbours' state buffers and runs a connection test (two reductions with known answers,
 *                          bounded wait).  COLLECTIVE.  With a communicator the ranks then agree (one all-reduce) on what
 *                          ALL of them can do:
 *                            *active = 2  mailboxes and ghost rows: hp_strip_step_batch runs ONE launch per iteration and calls
 *                                         nothing of the collective library inside an iteration -- the tiles that compute the
 *                                         strip's edge rows store them into the neighbours' ghost rows as well (CDomainLink's
 *                                         push / pull, Domain/Links/CDomainLink.cpp:168-270), and the launch's tail block
 *                                         holds the mailbox round, on every iteration, which is the hand-over;
 *                            *active = 1  mailboxes only (a rank could not map a neighbour's buffers, or HP_PEER_DIRECT=0
 *                                         in some rank's environment): the rows keep going through ncclSend / ncclRecv;
 *                            *active = 0  nothing: everything stays with the library (the reason goes to the log sink as
 *                                         a warning; the call still returns HP_OK).
 *                          Without a communicator (diagnostic use) *active is this rank's own verdict on its mailboxes;
 *   hp_strip_peer_round    one reduction of a caller-given value (diagnostic; COLLECTIVE over the connected ranks);
 *   hp_strip_peer_disconnect  unmaps and frees (also done by hp_strip_comm_destroy and hp_domain_destroy).
 * A strip that is not heard from within HP_PEER_TIMEOUT_MS (environment, default 10000) raises a sticky error instead of
 * hanging the GPU: hp_read_scalars then fails with HP_ERR_HIP. */
#define HP_PEER_TICKET_BYTES 384
int hp_strip_peer_ticket(hp_domain_t* d, void* ticket_out);
int hp_strip_peer_connect(hp_domain_t* d, const void* tickets, int count, int rank, int* active);
int hp_strip_peer_round(hp_domain_t* d, double value, double* max_out);
int hp_strip_peer_disconnect(hp_domain_t* d);

/* ---- measurement hooks (no reference counterpart; COCLDevice has no profiling queue, COCLDevice.cpp:283-288) ----
 * Bracket a region of the domain's stream with HIP events and return the elapsed milliseconds. */
int hp_timer_start(hp_domain_t* d);
int hp_timer_stop(hp_domain_t* d, float* elapsed_ms);  /* BLOCKS until the stop event completes */
/* Average device time of the dominant (flux) kernel: every `stride`-th launch is bracketed with events from a pool of
 * 16 pairs created by this call (so nothing is created inside a timed region); sampling stops when the pool is used up.
 * The time an EMPTY event pair takes (measured by this call on the idle stream) is taken off every sample. */
int hp_kernel_timing(hp_domain_t* d, int enable_stride);
int hp_kernel_timing_read(hp_domain_t* d, double* avg_ms, uint32_t* samples);   /* BLOCKS */
/* The cost of an empty event pair that the last hp_kernel_timing() measured and hp_kernel_timing_read() takes off every
 * sample (raw average = avg_ms + this): reported so that lines with and without the correction can be compared. */
int hp_kernel_timing_overhead(hp_domain_t* d, double* overhead_ms);
/* Iteration PAIRS.  Where it is the same computation -- Godunov scheme, the tuned kernel, quirk Q1 on, no boundary conditions or only
 * area boundaries the flux kernel can carry (hp_boundaries_fused), a grid of about a round of blocks and more -- hp_step_batch runs two
 * iterations as ONE launch (hp_kernels.hpp: godunov_march2; half the HBM traffic).  Everything observable through this interface is what
 * single iterations leave, with ONE condition in FAST arithmetic: a cell the reference leaves untouched at a pair's first step (quirk Q3,
 * CLSchemeGodunov.clc:248-255) passes its current state on where the reference keeps the stale one of the iteration before -- different
 * only if the cell dried out that very step.  The EXACT flavour writes those stale values down between launches (4-9 % slower) and runs
 * where they matter: STRICT arithmetic always, domains whose boundaries remove water (loss rate, mass flux), and on request.
 *   HP_TWO_STEP=0 / 1    pairs off / on wherever eligible (default: by grid size; STRICT: by the engine's own measurement, the two being
 *                        the same bits)                    HP_PAIR_EXACT=0 / 1   the exact flavour nowhere / everywhere
 *   HP_PAIR_BDY=0        domains with area boundaries keep single iterations      HP_PAIR_STRICT=0   so does STRICT arithmetic
 * hp_launch_counts and hp_pair_stats show what ran. */
/* How many whole-domain flux launches the domain has queued since it was created, and how many of them carried their own tail
 * block (reduction + time advance inside the flux launch: an iteration is then ONE launch; otherwise the flux launch is followed
 * by an advance launch).  bench.py states its roofline basis from the difference of two readings around the timed region. */
int hp_launch_counts(hp_domain_t* d, uint64_t* flux_launches, uint64_t* with_tail);

/* Diagnostics of the iteration pairs, for tests and A/B runs; blocks (a stream synchronisation; the stamps are counted by a small kernel,
 * nothing of their size is copied).  out[0] iteration pairs run, out[1] pairs that started cold on a
 * domain with area boundaries (stand-alone boundary pass + reduction in front: after single iterations, an upload, a new target
 * time), out[2] cells stamped by the LAST pair launch, out[3] cells that carry a stamp of any launch; the exact mode's choice between
 * pairs and single iterations (the same bits; chosen by measurement, hp_engine.hip: tuner_poll): out[4] samples taken, out[5] changes
 * of mind, out[6] 1 if pairs are the current choice, out[7] the last sample's pair time over its two single iterations' time, x 1000;
 * out[8] (exact flavour) stale values taken from a stamp that DIFFERED from the cell's current state -- the cells the flavour without
 * stamps would have got wrong on this run (0: it would have left the same bits); out[9..11] reserved. */
int hp_pair_stats(hp_domain_t* d, uint64_t out[12]);

#ifdef __cplusplus
}
#endif
#endif /* HIPIMS_MI_H */
