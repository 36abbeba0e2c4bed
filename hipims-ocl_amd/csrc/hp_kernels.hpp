// hp_kernels.hpp -- HIP kernels of the step engine (gfx950).  See DESIGN.md for the kernel graph.
#pragma once

#include "hp_math.hpp"

namespace hp {

// -------------------------------------------------------------------------------------------------
// K0  godunov_basic : one thread per cell, four face solves per cell.
//     Same dataflow as the reference's gts_cacheDisabled (CLSchemeGodunov.clc:164-384); kept as the
//     on-device cross-check for the tuned kernel (HP_KERNEL_BASIC) -- not the performance path.
// -------------------------------------------------------------------------------------------------
template <bool STRICT, typename T>
__global__ __launch_bounds__(256) void godunov_basic(const Params<T> p, const Scalars<T>* __restrict__ sc,
                                                     const T* __restrict__ bed, const State4<T>* __restrict__ src,
                                                     State4<T>* __restrict__ dst, const T* __restrict__ manning,
                                                     const long y_lo, const long y_hi)
{
	// rows [y_lo, y_hi): all but the edge ring of a whole domain; for a row strip what launch_rows says -- a strip's ghost rows
	// are its neighbours' to write (and, with the strips' own transport, written by them while this launch still runs)
	const long x = (long)blockIdx.x * blockDim.x + threadIdx.x;
	const long y = (long)blockIdx.y * blockDim.y + threadIdx.y + y_lo;
	if (x >= p.cols - 1 || y >= y_hi || x <= 0) return;                           // :183-187
	const size_t id = (size_t)y * p.cols + x;
	const T dt = sc->dt;

	if (dt <= T(0)) { dst[id] = src[id]; return; }                                // :201-206

	const State4<T> c = src[id];
	const T zb = bed[id], n = manning[id];
	if (c.zmax <= T(-9999.0) || c.z == T(-9999.0)) { dst[id] = c; return; }       // :214-218

	const size_t iW = id - 1, iE = id + 1, iS = id - p.cols, iN = id + p.cols;
	const State4<T> cW = src[iW], cE = src[iE], cS = src[iS], cN = src[iN];
	const T zW = bed[iW], zE = bed[iE], zS = bed[iS], zN = bed[iN];

	int dry = 0;                                                                  // :248-255
	if (c.z  - zb < p.vs) dry++;
	if (cN.z - zN < p.vs) dry++;
	if (cE.z - zE < p.vs) dry++;
	if (cS.z - zS < p.vs) dry++;
	if (cW.z - zW < p.vs) dry++;
	if (dry >= 5) return;                                                         // dst untouched (Q3)

	const Side<T> sC = make_side<STRICT>(c.z, c.qx, c.qy, zb, p.vs);
	const Side<T> sN = make_side<STRICT>(cN.z, cN.qx, cN.qy, zN, p.vs);
	const Side<T> sE = make_side<STRICT>(cE.z, cE.qx, cE.qy, zE, p.vs);
	const Side<T> sS = make_side<STRICT>(cS.z, cS.qx, cS.qy, zS, p.vs);
	const Side<T> sW = make_side<STRICT>(cW.z, cW.qx, cW.qy, zW, p.vs);

	const FaceFlux<T> fN = face_solve<AXIS_Y, STRICT, true, false>(sC, sN, p.vs).forL;
	const FaceFlux<T> fS = face_solve<AXIS_Y, STRICT, true, true>(sS, sC, p.vs).forR;
	const FaceFlux<T> fE = face_solve<AXIS_X, STRICT, true, false>(sC, sE, p.vs).forL;
	const FaceFlux<T> fW = face_solve<AXIS_X, STRICT, false, true>(sW, sC, p.vs).forR;

	dst[id] = godunov_update<STRICT>(c, zb, n, dt, fN, fE, fS, fW, p.dx, STRICT ? p.inv_dx_pow2 : p.inv_dx, p.vs, p.friction != 0);
}

// -------------------------------------------------------------------------------------------------
// K4  advance_time : tst_Advance_Normal (CLDynamicTimestep.clc:27-146), one lane.
//     `slot` holds the (all-reduced) maximum wave speed; it is cleared for the next accumulation.
//     UPDATE_ONLY = tst_UpdateTimestep (:255-317).
//     slot[0] = running maximum of this iteration (all-reduced by the host across strips), cleared here;
//     slot[SLOT_SAVED] = maximum last used (re-used when the primary buffer was not touched: `fresh` == 0, Q1);
//     slot[SLOT_EDGE], slot[SLOT_EDGE+1] = edge-ring maxima of the two state buffers (read by the march kernels).
// -------------------------------------------------------------------------------------------------
// tst_Advance_Normal's arithmetic (CLDynamicTimestep.clc:42-145) on a register copy of the time-control block: `vmax` is the
// maximum wave speed the reduction delivered.  Used by advance_body (the block in memory) and by the two-iterations kernel, whose
// every wavefront needs the SECOND iteration's timestep before the first has been launched (quirk Q1 makes it computable).
template <typename T>
__device__ __forceinline__ void advance_scalars(const Params<T>& p, Scalars<T>& s, const T vmax)
{
	const T EARLY_LIMIT = T(0.1), EARLY_DURATION = T(60.0), START_MIN = T(1E-10), START_DURATION = T(1.0);
	const T DT_MIN = T(1E-10), DT_MAX = T(15.0), HYDRO = T(1.0);                 // CLDynamicTimestep.clh:24-29
	T t = s.t, t_sync = s.t_sync, batch = s.batch_dt;
	T dt = fmax_(T(0), s.dt);                                                     // :42
	T t_hydro = s.t_hydro;
	uint32_t ok = s.batch_ok, skipped = s.batch_skipped;
	t += dt;                                                                      // :50-51
	batch += dt;
	if (dt > T(0)) ok++; else skipped++;                                          // :53-58
	if (t_hydro > HYDRO) t_hydro = dt; else t_hydro += dt;                        // :61-66

	if (p.dynamic_dt) {                                                           // :68-92
		T tmin = p.dx / vmax;
		if (t < START_DURATION && tmin < START_MIN) tmin = START_MIN;
		dt = p.courant * tmin;
	} else {
		dt = p.dt_fixed;                                                          // :93-97
	}
	if (dt > T(0) && dt < DT_MIN) dt = DT_MIN;                                    // :112-113
	if ((t + dt) >= t_sync) {                                                     // :118-124
		const T dt_in = dt;
		if (t_sync - t > p.vs)  dt = t_sync - t;
		if (t_sync - t <= p.vs) dt = -dt_in;
	}
	if (t < EARLY_DURATION && dt > EARLY_LIMIT) dt = EARLY_LIMIT;                 // :128-129
	if ((t + dt) > p.t_end) dt = p.t_end - t;                                     // :132-133
	if (dt > DT_MAX) dt = DT_MAX;                                                 // :136-137

	s.t = t; s.dt = dt; s.t_hydro = t_hydro; s.batch_dt = batch;                  // :140-145
	s.batch_ok = ok; s.batch_skipped = skipped;
}

template <bool UPDATE_ONLY, typename T>
__device__ __forceinline__ void advance_body(const Params<T>& p, Scalars<T>* sc, T* slot, const int fresh)
{
	const T EARLY_LIMIT = T(0.1), EARLY_DURATION = T(60.0), START_MIN = T(1E-10), START_DURATION = T(1.0), DT_MAX = T(15.0);   // CLDynamicTimestep.clh:24-29
	// slot[0] is only ever touched by agent-scope atomics (performed at the memory side, so no XCD's L2 holds a
	// stale copy); the remembered maximum lives one cache line further (SLOT_SAVED)
	// fresh: bit 0 = this iteration priced a buffer anew (slot[0] holds this rank's maximum); bit 1 = the maximum over all
	// strips stands in slot[SLOT_GLOBAL] (hp_strip_step_batch's all-reduce writes it there, next to the local one)
	T vmax;
	if (fresh) {
		vmax = atomic_exchange_zero(slot);
		slot[SLOT_LOCAL] = vmax;
		if (fresh & 2) vmax = slot[SLOT_GLOBAL];
		slot[SLOT_SAVED] = vmax;
	} else {
		vmax = slot[SLOT_SAVED];
	}

	T t = sc->t, t_sync = sc->t_sync, batch = sc->batch_dt;
	if (UPDATE_ONLY) {
		const T dt_orig = fabs_(sc->dt);                                          // :264
		// TIMESTEP_FIXED leaves the reference's dLclTimestep UNINITIALISED here (:268, :269-292 compiled out) before
		// fmin(dLclTimestep, original); its host build yields the original value (what any garbage >= original, or a
		// NaN, gives through fmin) -- restated as that
		T dt = dt_orig;
		if (p.dynamic_dt) {
			T tmin = p.dx / vmax;
			if (t < START_DURATION && tmin < START_MIN) tmin = START_MIN;
			dt = p.courant * tmin;
		}
		dt = fmin_(dt, dt_orig);                                                  // :297-298
		batch = batch - dt_orig + dt;
		if (t < EARLY_DURATION && dt > EARLY_LIMIT) dt = EARLY_LIMIT;             // :301-302
		if ((t + dt) >= t_sync) dt = fmax_(T(0), t_sync - t);                     // :305-306
		if (dt > DT_MAX) dt = DT_MAX;                                             // :309-310
		sc->dt = dt;
		sc->batch_dt = batch;
		return;
	}

	Scalars<T> s = *sc;
	advance_scalars(p, s, vmax);
	*sc = s;
}

// -------------------------------------------------------------------------------------------------
// The maximum over all strips, written by the strips themselves ("one-shot all-gather-to-all over direct
// links, then a local max"; stands where the reference's MPI_Allreduce is, MPI/CMPIManager.cpp:852-861).
//
//   Every rank owns a MAILBOX in uncached device memory that its peers address directly (xGMI peer access, through
//   an IPC mapping when the peer is another process): 2 sets x 64 words, word [set][r] written by rank r only.  A
//   reduction is one pass of the advance kernel's wavefront: lane r stores this rank's value into rank r's mailbox
//   (one 8-byte system-scope store per peer), then polls its own mailbox's word r until rank r's value has arrived,
//   empties it, and the wavefront folds the 64 lanes.  No collective kernel, no launch besides the advance kernel
//   that runs anyway.
//
//   Two sets are enough: rank P can only write reduction n+2 after it has finished reduction n+1, which needs THIS
//   rank's value n+1, which this rank publishes (release) after it emptied set n%2 in reduction n -- so the write
//   of n+2 is ordered behind the emptying it must not precede.
//
//   The poll is bounded: a peer that never shows up (a rank that died, a mapping that does not do what it should)
//   raises the mailbox's sticky error word instead of hanging the GPU; later reductions then return at once and the
//   host reads the error with the scalars (hp_read_scalars).
// -------------------------------------------------------------------------------------------------
constexpr int PEER_MAX_RANKS = 64, PEER_WORD_ERROR = 2 * PEER_MAX_RANKS, PEER_WORD_RESULT = PEER_WORD_ERROR + 1,
              PEER_WORDS = PEER_WORD_ERROR + 8;
constexpr unsigned long long PEER_EMPTY = ~0ull;       // no finite value, and not the NaN arithmetic produces

struct PeerBox {
	unsigned long long*        mine;      // this rank's mailbox
	unsigned long long* const* peer;      // device table: rank r's mailbox as this device addresses it
	int                        world, rank, set;
	unsigned long long         timeout;   // wall_clock64 ticks (100 MHz)
};

__device__ __forceinline__ unsigned long long peer_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ unsigned long long peer_bits(float v)  { return (unsigned long long)__float_as_uint(v); }
__device__ __forceinline__ void peer_value(unsigned long long w, double& v) { v = __longlong_as_double((long long)w); }
__device__ __forceinline__ void peer_value(unsigned long long w, float& v)  { v = __uint_as_float((unsigned)w); }

__device__ __forceinline__ double atomic_peek(double* s)
{
	return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ float atomic_peek(float* s)
{
	return __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned int*>(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// all 64 lanes of one wavefront; returns the maximum over the ranks' values (wave_max's `>`: a NaN never wins)
template <typename T>
__device__ __forceinline__ T peer_reduce_max(const PeerBox& box, const T local)
{
	const int lane = threadIdx.x & 63;
	T v = local;
	if (lane < box.world) {
		unsigned long long* inbox = box.mine + box.set * PEER_MAX_RANKS + lane;
		__hip_atomic_store(box.peer[lane] + box.set * PEER_MAX_RANKS + box.rank, peer_bits(local), __ATOMIC_RELEASE,
		                   __HIP_MEMORY_SCOPE_SYSTEM);
		const bool broken = __hip_atomic_load(box.mine + PEER_WORD_ERROR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
		const unsigned long long t0 = wall_clock64();
		unsigned long long w;
		for (;;) {
			w = __hip_atomic_load(inbox, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
			if (w != PEER_EMPTY) break;
			if (broken || wall_clock64() - t0 > box.timeout) break;
			__builtin_amdgcn_s_sleep(1);
		}
		if (w != PEER_EMPTY) {
			__hip_atomic_store(inbox, PEER_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			peer_value(w, v);
		} else {
			__hip_atomic_store(box.mine + PEER_WORD_ERROR, 1ull + (unsigned long long)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
	return wave_max(v);
}

// Ghost rows written straight into the strip neighbours' memory (CDomainLink::pushToBuffer / pullFromBuffer and CMPIManager's
// block exchange, Domain/Links/CDomainLink.cpp:168-270, MPI/CMPIManager.cpp:555-709, without a transfer library in between):
// the blocks of the advance kernel copy this strip's first / last owned rows of the NEW state into the neighbours' ghost
// rows (peer addresses over xGMI), and the block that finishes last runs the mailbox round and the time advance.  The
// round doubles as the hand-over: a neighbour's advance kernel returns only after THIS kernel has published, which is after
// every row has left (release at system scope), and its next flux launch starts after that; and this strip's next flux
// launch -- whose rows will be written into the neighbours' OTHER buffer -- starts only after the neighbours have published,
// i.e. after the flux launch that still read that buffer.  Hence one round per iteration, also on the iterations that need
// no new maximum (quirk Q1).
struct PeerPush {
	const uint4* from[2];      // [0]: rows for the south neighbour, [1]: for the north one (this strip's new state)
	uint4*       to[2];        // the neighbours' ghost rows, as this device addresses them; nullptr = no neighbour / nothing to send
	unsigned     count;        // uint4 per side
	unsigned*    arrived;      // blocks done (zero between launches)
};

// fresh bit 2: a round through the mailboxes; bit 1 then says whether its result is the maximum to use
template <bool UPDATE_ONLY, typename T>
__global__ void advance_time(const Params<T> p, Scalars<T>* sc, T* slot, const int fresh, const PeerBox box, const PeerPush push)
{
	if (gridDim.x > 1 || push.to[0] || push.to[1]) {
		for (int side = 0; side < 2; ++side) {
			if (!push.to[side]) continue;
			for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < push.count; i += gridDim.x * blockDim.x)
				push.to[side][i] = push.from[side][i];
		}
		__shared__ unsigned last;
		__threadfence_system();                                   // this thread's rows are out ...
		__syncthreads();                                          // ... and so are the whole block's
		if (threadIdx.x == 0) last = __hip_atomic_fetch_add(push.arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
		__syncthreads();
		if (!last) return;
		if (threadIdx.x == 0) __hip_atomic_store(push.arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	} else if (blockIdx.x != 0) return;
	if (threadIdx.x >= 64) return;
	if (fresh & 4) {
		const T local = atomic_peek(slot);            // slot[0] is only ever touched by memory-side atomics
		const T all = peer_reduce_max(box, local);
		if (threadIdx.x == 0 && (fresh & 2)) slot[SLOT_GLOBAL] = all;
	}
	if (threadIdx.x != 0) return;
	advance_body<UPDATE_ONLY>(p, sc, slot, fresh & 3);
}

// -------------------------------------------------------------------------------------------------
// The launch's own tail.  Two dependent launches per iteration (flux, advance) cost more in hand-over than in work on
// anything but the largest grids (the example's 342 x 195: 11.7 us per iteration for 3 us of arithmetic).
// Instead of a separate advance launch behind the flux launch, the flux launch carries ONE more block.  Every flux block ends by storing its maximum into its own word of `done`
// (a plain 8-byte store: the word doubles as the block's "I am through" flag, so there is no atomic to wait for and no
// ordering between two memory operations to arrange); the tail block polls all the words, folds them, empties them for the
// next launch, and then does what advance_time does (mailbox round, advance_body; the ghost rows have left with the tiles).  The tail
// block has the launch's highest index: when it is dispatched every flux block of its XCD has been, and the other XCDs'
// blocks never wait for it -- it cannot starve what it waits for.  It needs nothing of the flux blocks but their maxima: the
// new state is the NEXT launch's business, which the stream orders behind this whole launch.
// -------------------------------------------------------------------------------------------------
template <typename T> struct LaunchTail {
	unsigned long long* done;      // nullptr: no tail (the classic two launches)
	unsigned            flux_blocks;
	int                 pair;          // the launch covered TWO iterations (godunov_march2): the first advance re-uses the remembered maximum (quirk Q1), the second takes this launch's;
	                                   // 2: ... on a domain with area boundaries -- the first advance takes slot[SLOT_M1] (the primary buffer priced WITH the first
	                                   // iteration's rain, by the launch before), and this launch's second word per block (done1) becomes the next slot[SLOT_M1]
	unsigned long long* done1;         // (pair == 2) per flux block: its maximum with the boundaries of the iteration after the pair applied
	int                 bdy_flag;      // (pair == 2) what cfl_slot[SLOT_BDY] holds after this launch: 1 = the state it stored carries the next iteration's boundaries
	unsigned            poll_blocks;   // words the tail block waits for: flux_blocks (HP_DEBUG_TAIL_EXTRA_WORD=1: one more, which nobody writes -- the time-out's test)
	unsigned long long  timeout;       // wall_clock64 ticks (100 MHz) the tail block waits for ONE word before it gives up (HP_TAIL_TIMEOUT_MS)
	int                 fresh;     // advance_time's `fresh`
	Scalars<T>*         sc;
	T*                  slot;
	PeerBox             box;
	// ghost rows (TAIL == 2 instantiations of the flux kernels): the strip's first / last owned rows are stored TWICE by the tile that
	// computes them -- into this strip's new state and, through peer_rows[side], into the neighbour's ghost rows of the same
	// ping-pong buffer.  peer_rows[side] is shifted so that THIS strip's cell index addresses the neighbour's copy of the cell.
	// Cells the kernel leaves untouched (quirk Q3) are left untouched there too: both copies started equal (the host uploads
	// strips with their ghost rows) and receive the same stores ever after.
	State4<T>*          peer_rows[2];   // [0] south neighbour, [1] north; nullptr: no neighbour / nothing to send this iteration
	int                 edge_rows[4];   // [lo, hi) of the rows that go south, [lo, hi) of those that go north (empty: lo >= hi)
};

#ifndef HP_TAIL_FORMAL_ORDER
#define HP_TAIL_FORMAL_ORDER 0
#endif
// every wavefront of every flux block, at its very end; `m` = the wavefront's maximum (0 when this launch prices nothing)
template <bool TWO = false, typename T>
__device__ __forceinline__ void tail_block_done(const LaunchTail<T>& tail, const T m, const int wave, const int lane, const long y0, const long y1, const T m1 = T(0))
{
	__shared__ T part[4];
	__shared__ T part1[4];
	// a tile that stored rows into a neighbour reports only once those stores have been acknowledged (its own state's stores
	// need no such wait: nobody reads them before the next launch); all the other tiles report at once
	const bool sends = (y0 < tail.edge_rows[1] && y1 > tail.edge_rows[0]) || (y0 < tail.edge_rows[3] && y1 > tail.edge_rows[2]);   // wave-uniform
	// Ordering of the hand-over.  What the protocol needs: the edge tiles' rows are in the neighbour's memory before this strip's
	// tail block publishes into the neighbour's mailbox (system-scope release store in peer_reduce_max).  How it is had: the rows
	// are stored WRITTEN THROUGH at system scope (sc0 sc1: they pass this XCD's L2 and are acknowledged by the memory they are
	// for), the tile waits for those acknowledgements (s_waitcnt vmcnt(0)) before its block reports "through", and the tail
	// block reads every block's report before it publishes.  In the HSA memory model's own terms that chain would be a
	// system-scope RELEASE of the done word and an ACQUIRE in the tail block -- built and measured in round 4
	// (HP_TAIL_FORMAL_ORDER=1): a system-scope release makes the block write back its XCD's whole L2 (the 4 MiB of ordinary state
	// stores the other tiles have left there), and the two-strip probe went from 44.5 to 48.6 us per iteration on one box
	// (profiles/r04f_pair_probe_bisect.txt).  The write-through
	// + acknowledged-store form stays the default: it orders exactly the stores that cross to the neighbour and nothing else
	// (30 000-iteration soaks, the strip fuzz and the process-rank tests all run on it).
	if (sends) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	if (lane == 0) { part[wave] = m; if (TWO) part1[wave] = m1; }
	__syncthreads();
	bool block_sends = false;
	if (threadIdx.x == 0) {
		// (the tile rows of the block's four waves are the same: `sends` of wave 0 is the block's)
		block_sends = sends;
		T b = part[0];
		for (int w = 1; w < 4; ++w) if (part[w] > b) b = part[w];
		if (TWO) {                                               // (a word of its own with its own "arrived": no order between the two stores is needed)
			T b1 = part1[0];
			for (int w = 1; w < 4; ++w) if (part1[w] > b1) b1 = part1[w];
			__hip_atomic_store(tail.done1 + blockIdx.x, peer_bits(b1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
#if HP_TAIL_FORMAL_ORDER
		if (block_sends) __hip_atomic_store(tail.done + blockIdx.x, peer_bits(b), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		else
#endif
		__hip_atomic_store(tail.done + blockIdx.x, peer_bits(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		(void)block_sends;
	}
}

#ifndef HP_TAIL_POLL_SLEEP
#define HP_TAIL_POLL_SLEEP 1
#endif
// The tail block's wait is BOUNDED (round 5).  It is correct under in-order block dispatch -- the tail block has the launch's
// highest index, every flux block is resident or dispatched before it -- and a dispatcher that ever did otherwise would leave it
// spinning on a block that cannot start: a hung GPU, on a shared box a reset.  So a word that has not arrived `timeout` ticks after
// the tail block first found it empty (HP_TAIL_TIMEOUT_MS, generous: the tail block only starts once its XCD has dispatched every
// flux block of the launch) raises the sticky word cfl_slot[SLOT_TAIL_ERR] instead: the tail block returns WITHOUT the mailbox round
// and WITHOUT advance_body -- t and dt stay frozen, the flux blocks that were still to come run and end on their own -- later tail
// blocks of the same batch see the word and return at once, and the host finds it wherever it next blocks on the stream
// (hp_sync, hp_read_scalars; hp_engine.hip: tail_error_check) and fails the domain with HP_ERR_STATE.  The clock is only read on
// the slow path (a word found empty), the common case is one load per word as before.
template <typename T>
__device__ __forceinline__ void launch_tail(const Params<T>& p, const LaunchTail<T>& tail)
{
	__shared__ T part[4];
	__shared__ int failed_before, late_any;     // two words: a wave that times out early must not look like "an earlier launch failed" to a slower wave's first read
	if (threadIdx.x == 0) { failed_before = atomic_peek(tail.slot + SLOT_TAIL_ERR) != T(0); late_any = 0; }   // an earlier launch of this domain gave up
	__syncthreads();
	if (failed_before) return;
	T m = T(0), m1 = T(0);
	bool late = false;
	// (pair == 2: every flux block reports TWO maxima, each in a word of its own -- the second array is polled the same way)
	for (int set = 0; set < (tail.pair == 2 ? 2 : 1) && !late; ++set) {
		unsigned long long* const words = set ? tail.done1 : tail.done;
		const unsigned count = set ? tail.flux_blocks : tail.poll_blocks;
		for (unsigned i = threadIdx.x; i < count; i += blockDim.x) {
			unsigned long long w = __hip_atomic_load(words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (w == PEER_EMPTY) {
				const unsigned long long t0 = wall_clock64();
				for (;;) {
					__builtin_amdgcn_s_sleep(HP_TAIL_POLL_SLEEP);
					w = __hip_atomic_load(words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if (w != PEER_EMPTY) break;
					if (wall_clock64() - t0 > tail.timeout) { late = true; break; }
				}
				if (late) break;
			}
			__hip_atomic_store(words + i, PEER_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			T v;
			peer_value(w, v);
			if (set) { if (v > m1) m1 = v; } else { if (v > m) m = v; }
		}
	}
	if (late) late_any = 1;                                     // (benign race: every writer writes 1)
	__syncthreads();
	if (late_any) {
		if (threadIdx.x == 0) tail.slot[SLOT_TAIL_ERR] = T(1);
		return;
	}
	// every done word has been seen: acquire what their writers released (the edge tiles' rows in the neighbours' memory)
	// before this block publishes into the neighbours' mailboxes
#if HP_TAIL_FORMAL_ORDER
	if (tail.peer_rows[0] || tail.peer_rows[1]) __atomic_thread_fence(__ATOMIC_ACQUIRE);
#endif
	m = wave_max(m);
	__shared__ T part1[4];
	if (tail.pair == 2) { m1 = wave_max(m1); if ((threadIdx.x & 63) == 0) part1[threadIdx.x >> 6] = m1; }
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < 4; ++w) if (part[w] > m) m = part[w];
		if (tail.fresh & 1) tail.slot[0] = m;                   // where the atomic maxima of a classic launch would have gathered
		part[0] = m;
		if (tail.pair == 2) for (int w = 1; w < 4; ++w) if (part1[w] > m1) m1 = part1[w];
	}
	__syncthreads();                                            // (part[0] for wave 0)
	if (threadIdx.x >= 64) return;
	if (tail.fresh & 4) {
		const T local = (tail.fresh & 1) ? part[0] : atomic_peek(tail.slot);
		const T all = peer_reduce_max(tail.box, local);
		if (threadIdx.x == 0 && (tail.fresh & 2)) tail.slot[SLOT_GLOBAL] = all;
	}
	if (threadIdx.x != 0) return;
	if (tail.pair == 2) {
		// area boundaries: the first iteration's reduction priced the primary buffer with ITS rain in it -- the maximum the launch
		// before left in slot[SLOT_M1] (every wavefront of this launch has taken the second timestep from it: godunov_march2)
		Scalars<T> s = *tail.sc;
		advance_scalars(p, s, tail.slot[SLOT_M1]);
		*tail.sc = s;
	} else if (tail.pair) advance_body<false>(p, tail.sc, tail.slot, 0);      // the pair's first iteration: its reduction re-read the primary buffer (fresh == 0)
	advance_body<false>(p, tail.sc, tail.slot, tail.fresh & 3);
	if (tail.pair == 2) {
		tail.slot[SLOT_M1] = m1;                                 // what the NEXT pair's first advance will be told
		tail.slot[SLOT_BDY] = tail.bdy_flag ? T(1) : T(0);       // (for the stand-alone boundary pass of a single iteration that may follow)
	}
}

// diagnostic / connection test: one reduction of a caller-given word (hp_strip_peer_round)
__global__ void peer_round(const PeerBox box, const double value)
{
	const double all = peer_reduce_max(box, value);
	if (threadIdx.x == 0) box.mine[PEER_WORD_RESULT] = peer_bits(all);
}

// -------------------------------------------------------------------------------------------------
// K1  godunov_march : the tuned Godunov + HLLC step.
//
//  * One wavefront owns a strip of 64 consecutive columns (lane = column, so every row of the strip is one
//    coalesced 2 KiB state load + 512 B bed + 512 B Manning) and MARCHES north over `rseg` rows.
//  * Every face is solved once: a lane solves the face to its EAST and the face to its NORTH (face_solve
//    finishes each for both adjacent cells).  The west-face result arrives from lane-1 through a cross-lane
//    move, the south-face result is carried in registers from the previous row.  Lanes 0 and 63 are halo lanes
//    (62 updated columns per wave); the row below the segment costs one extra north-face solve per segment.
//  * The 4-neighbour stencil therefore never re-reads a cell from memory inside a tile: W/E neighbours come
//    from the wave's own registers via DPP wavefront shifts, N/S from the register pipeline.
//  * CFL_MODE fuses tst_Reduce into the epilogue: 1 = speeds of what this launch leaves in `dst`
//    (incl. the stale value of all-dry cells the reference does not write, Q3), 2 = speeds of the source
//    state (after the boundary kernels), 0 = none.  Edge-ring cells are priced once at upload (edge_max).
//  * blockIdx is remapped so that each XCD (blocks b, b+8, ... share one) works on a contiguous run of tiles
//    and neighbouring tiles' halo rows/columns hit the same L2.
// -------------------------------------------------------------------------------------------------
// ---- buffer-resource addressing (gfx9 SRD): base in 4 SGPRs, per-lane byte offset in ONE VGPR that never changes,
//      per-row offset in an SGPR -> no 64-bit VALU address arithmetic in the row loop, and stores whose offset lies
//      beyond num_records are dropped by the hardware range check (how non-writing lanes are masked).
typedef unsigned int hp_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int hp_u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned HP_SRD_FLAGS = 0x00020000u;        // raw buffer, 32-bit data format (gfx90a/gfx94x/gfx950)
constexpr unsigned HP_OOB = 0x80000000u;              // any offset >= num_records: access dropped

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void* base, const size_t bytes)
{
	const unsigned n = bytes > 0x7fffffffull ? 0x7fffffffu : (unsigned)bytes;
	return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)n, (int)HP_SRD_FLAGS);
}

// cache policy of the three access streams (aux operand: bit 1 = nt on gfx940+); all default -- see DESIGN 4 K1 for what
// selective non-temporal streams measured
#ifndef HP_AUX_STATE_LD
#define HP_AUX_STATE_LD 0
#endif
#ifndef HP_AUX_STATE_ST
#define HP_AUX_STATE_ST 0
#endif
#ifndef HP_AUX_SCALAR_LD
#define HP_AUX_SCALAR_LD 0
#endif
__device__ __forceinline__ State4<double> buf_load_state(__amdgpu_buffer_rsrc_t r, const unsigned voff, const unsigned soff, double)
{
	const hp_u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, HP_AUX_STATE_LD);
	const hp_u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff + 16, (int)soff, HP_AUX_STATE_LD);
	State4<double> s;
	s.z    = __hiloint2double((int)a.y, (int)a.x); s.zmax = __hiloint2double((int)a.w, (int)a.z);
	s.qx   = __hiloint2double((int)b.y, (int)b.x); s.qy   = __hiloint2double((int)b.w, (int)b.z);
	return s;
}
__device__ __forceinline__ State4<float> buf_load_state(__amdgpu_buffer_rsrc_t r, const unsigned voff, const unsigned soff, float)
{
	const hp_u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, HP_AUX_STATE_LD);
	State4<float> s;
	s.z = __uint_as_float(a.x); s.zmax = __uint_as_float(a.y); s.qx = __uint_as_float(a.z); s.qy = __uint_as_float(a.w);
	return s;
}
__device__ __forceinline__ double buf_load_scalar(__amdgpu_buffer_rsrc_t r, const unsigned voff, const unsigned soff, double)
{
	const hp_u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, HP_AUX_SCALAR_LD);
	return __hiloint2double((int)a.y, (int)a.x);
}
__device__ __forceinline__ float buf_load_scalar(__amdgpu_buffer_rsrc_t r, const unsigned voff, const unsigned soff, float)
{
	return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, HP_AUX_SCALAR_LD));
}
// Store hazard.  On gfx950 a 16-byte buffer store reads its VGPR operands -- the data AND the per-lane offset -- over
// several cycles AFTER it has issued; a VALU instruction that overwrites one of them in the next slots changes what lanes
// 12-15 of every 16-lane row store, or where.  Seen in round 2 as depth instead of level in z (fp32, 4096^2 and up: `h = z -
// zb` of the CFL epilogue reused z's register directly behind the store) and in round 3 as cells that were not stored at
// all (strips with rain, fp32: the compiler had moved a `v_mov` that recycled the OFFSET register between the store and the
// fence that then only covered the data).  The compiler's hazard recogniser exempts stores whose soffset is an SGPR, as
// ours is.  All operands of the store are therefore kept alive through the wait states behind it: nothing may overwrite
// them before this fence, whatever else gets scheduled in between (four wait states: round 2's two were found by trial, and
// the cost is two cycles per row).
// The fence binds VALUES, not registers (round 6, tools/isa_store_hazard.py): where the stored values also live in other registers -- the
// state that is stored is priced, carried to the next row, stored a second time into a strip neighbour -- the compiler satisfied the
// fence's inputs from THOSE copies and recycled the store's own data registers one slot behind it (found in the shipped code of round 5:
// godunov_march2<0, 2, float>, inertial_march<true, 0, 2, double>, both behind a peer store).  store_operands_own makes the four data
// words values of their own first: defined by an (empty) asm, they have no other copy the fence could be fed from, so keeping them
// alive to the fence means keeping the store's registers.  The build's assembly is scanned for the pattern on every build
// (tests/test_resource_usage.py): a store whose data or offset register is written inside the window fails the CPU suite.
#ifndef HP_K1B_STRICT_WAVES
#define HP_K1B_STRICT_WAVES 2        // waves per SIMD of the exact mode's fp64 pair kernel (3: 168 registers, spills -- see LAB_NOTES R6.6)
#endif
#ifndef HP_STORE_OPERANDS_OWN
#define HP_STORE_OPERANDS_OWN 1         // 0: round 5's stores (the study of LAB_NOTES R6.x builds both)
#endif
__device__ __forceinline__ void store_operands_own(hp_u32x4& a)
{
	if (HP_STORE_OPERANDS_OWN) asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));
}
__device__ __forceinline__ void store_fence(const hp_u32x4& a, const unsigned voff, const unsigned soff)
{
	asm volatile("s_nop 3" : : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(voff), "s"(soff) : "memory");
}

// `voff` selects per lane between the cell's offset and HP_OOB (dropped by the range check); it is pinned in a VGPR so
// that the compiler cannot turn the select into two exec-masked stores (memory instructions under divergent control
// flow make the s_waitcnt vmcnt bookkeeping conservative).
// AUX: the store's cache policy.  HP_AUX_THROUGH (sc0 sc1: written through to memory at system scope) is for rows stored into a
// strip neighbour's buffer: they must not linger in this XCD's L2 -- the reader is another GPU, or another launch that
// starts as soon as this strip's tail block has spoken, before THIS launch ends and writes its L2 back.
constexpr int HP_AUX_THROUGH = 1 | 16;
template <int AUX = HP_AUX_STATE_ST>
__device__ __forceinline__ void buf_store_state(const State4<double>& s, __amdgpu_buffer_rsrc_t r, unsigned voff, const unsigned soff)
{
	asm volatile("" : "+v"(voff));
	hp_u32x4 a, b;
	a.x = (unsigned)__double2loint(s.z);  a.y = (unsigned)__double2hiint(s.z);
	a.z = (unsigned)__double2loint(s.zmax); a.w = (unsigned)__double2hiint(s.zmax);
	b.x = (unsigned)__double2loint(s.qx); b.y = (unsigned)__double2hiint(s.qx);
	b.z = (unsigned)__double2loint(s.qy); b.w = (unsigned)__double2hiint(s.qy);
	store_operands_own(a); store_operands_own(b);
	__builtin_amdgcn_raw_buffer_store_b128(a, r, (int)voff, (int)soff, AUX);
	__builtin_amdgcn_raw_buffer_store_b128(b, r, (int)voff + 16, (int)soff, AUX);
	store_fence(a, voff, soff);
	store_fence(b, voff, soff);
}
template <int AUX = HP_AUX_STATE_ST>
__device__ __forceinline__ void buf_store_state(const State4<float>& s, __amdgpu_buffer_rsrc_t r, unsigned voff, const unsigned soff)
{
	asm volatile("" : "+v"(voff));
	hp_u32x4 a;
	a.x = __float_as_uint(s.z); a.y = __float_as_uint(s.zmax); a.z = __float_as_uint(s.qx); a.w = __float_as_uint(s.qy);
	store_operands_own(a);
	__builtin_amdgcn_raw_buffer_store_b128(a, r, (int)voff, (int)soff, AUX);
	store_fence(a, voff, soff);
}

constexpr int MARCH_COLS = 62;          // updated columns per wavefront (lanes 1..62)

// Neighbour-lane moves.  A lane's column neighbours live in the adjacent lanes of the same wavefront; their values
// arrive through DPP wavefront ROTATES (one v_mov_b32_dpp per dword, a VALU instruction) instead of a trip through the
// LDS crossbar (ds_bpermute: an LDS instruction, an address VGPR and an lgkmcnt wait per batch).  A rotate, not a shift
// (round 4): a shift leaves the lane without a neighbour (63 for east, 0 for west) its old contents, and keeping the lane's
// own value there cost a v_mov_b32 in front of every v_mov_b32_dpp (the destination is tied to `old`).  With the rotate lane
// 63's "east" is lane 0's cell and lane 0's "west" is lane 63's: both are halo lanes whose results are never stored, what
// they receive is a real cell of the same row (finite, sane), and a wave-uniform test of the form "every lane and its
// neighbours ..." is unchanged because the neighbour is some lane of the same wavefront either way.
// (Directions verified on MI355X: tools/dpp_probe.)
constexpr int DPP_WAVE_ROL1 = 0x134, DPP_WAVE_ROR1 = 0x13C;     // dst[i] = src[(i+1) % 64] / dst[i] = src[(i-1) % 64]
template <int CTRL> __device__ __forceinline__ int lane_move(const int v)
{
	return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, false);
}
template <int CTRL> __device__ __forceinline__ float lane_move(const float v)
{
	return __int_as_float(lane_move<CTRL>(__float_as_int(v)));
}
template <int CTRL> __device__ __forceinline__ double lane_move(const double v)
{
	return __hiloint2double(lane_move<CTRL>(__double2hiint(v)), lane_move<CTRL>(__double2loint(v)));
}
// lane 0's value in every lane (a scalar)
__device__ __forceinline__ float first_lane(const float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double first_lane(const double v)
{
	return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
template <typename V> __device__ __forceinline__ V from_east(const V v) { return lane_move<DPP_WAVE_ROL1>(v); }   // lane + 1
template <typename V> __device__ __forceinline__ V from_west(const V v) { return lane_move<DPP_WAVE_ROR1>(v); }   // lane - 1

template <typename T> struct RowRegs { State4<T> c; T zb, n; };

template <typename T>
__device__ __forceinline__ Side<T> side_from_east(const Side<T>& s)
{
	Side<T> r;
	r.eta = from_east(s.eta); r.zb = from_east(s.zb);
	r.qx = from_east(s.qx);   r.qy = from_east(s.qy);
	r.u0 = from_east(s.u0);   r.v0 = from_east(s.v0);
	return r;
}

// How one launch's blocks map to tiles.  A launch covers the updated rows [y_begin, y_end) of the domain (all of them,
// or the interior / one halo block of a strip-decomposed step: hp_engine.hip).  The rows are cut into 8 bands, one per
// XCD (blocks are dealt round-robin to the XCDs, so blockIdx & 7 IS the XCD: its blocks walk one contiguous band and
// the halo rows of neighbouring tiles are hits in that XCD's L2).  Inside a band the first `nbig` row segments are
// `rseg` rows tall and the last `ntail` segments only `rseg_tail`: blocks start in blockIdx order, so the short
// tiles are what is running when the band drains, and the tail of the launch is one SHORT tile long instead of one
// tall tile (a tall tile lasts ~48 us at 4096^2 -- 18 % of the launch, measured as the intercept of
// time-vs-cells between 4096^2 and 16384 x 8192).
struct TileMap {
	int  nstrips, groups;        // 62/60-column strips of the grid, and 4-wave groups of them
	long y_begin, y_end;         // updated rows covered by this launch
	int  nbands;                 // 8 (one band per XCD); 2 for the halo launch of a strip (south block, north block)
	long band_stride;            // distance between band starts (= band_rows, except for the halo launch)
	int  band_rows;              // rows per band
	int  rseg, nbig;             // tall segments per band
	int  rseg_tail, ntail;       // short segments per band (after the tall ones)
	int  price_lo, price_hi;     // rows whose cells the fused CFL epilogue prices: the rows this rank OWNS (a strip with two
	                             // reaches of ghost rows also updates rows it does not own, hp_engine.hip: strip loop); 32-bit so
	                             // that the per-row test is two scalar compares
	int  flip;                   // this launch visits the tiles of every band from the top down (hp_engine.hip: sweep_flip)
};

// rows [y0, y1) and the column strip of this wave; false if the block / wave has nothing to do (wave-uniform)
__device__ __forceinline__ bool tile_rows(const TileMap& tm, const int wave, long& strip, long& y0, long& y1)
{
	const unsigned band = blockIdx.x % (unsigned)tm.nbands, i = blockIdx.x / (unsigned)tm.nbands;
	const unsigned nbig_tiles = (unsigned)(tm.groups * tm.nbig);
	const long band_y0 = tm.y_begin + (long)band * tm.band_stride;
	const long band_y1 = (band_y0 + tm.band_rows < tm.y_end) ? (band_y0 + tm.band_rows) : tm.y_end;
	unsigned g;
	long h;
	if (i < nbig_tiles) {
		g = i % (unsigned)tm.groups;
		h = tm.rseg;
		y0 = band_y0 + (long)(i / (unsigned)tm.groups) * tm.rseg;
	} else {
		const unsigned j = i - nbig_tiles;
		g = j % (unsigned)tm.groups;
		h = tm.rseg_tail;
		y0 = band_y0 + (long)tm.nbig * tm.rseg + (long)(j / (unsigned)tm.groups) * tm.rseg_tail;
	}
	y1 = (y0 + h < band_y1) ? (y0 + h) : band_y1;
	strip = (long)g * 4 + wave;
	const bool any = y0 < band_y1 && strip < tm.nstrips;
	if (tm.flip) {                               // the same tiles, mirrored within the band (rows still march south to north inside a tile)
		const long top = band_y0 + (band_y1 - y1);
		y1 = band_y0 + (band_y1 - y0);
		y0 = top;
	}
	return any;
}

// ---- area boundaries (bdy_Uniform / bdy_Gridded, Boundaries/CLBoundaries.clc:130-246): descriptors shared by the stand-alone
//      pass (bdy_area, further down) and by the flux kernel's fused epilogue ----
template <typename T> struct UniformBdy { const T* series; uint32_t entries; int definition; T interval, length; };
template <typename T> struct GriddedBdy {
	const T* grids; uint64_t entries, grows, gcols; int definition; T resolution, off_x, off_y, interval;
};

template <typename T>
__device__ __forceinline__ bool bdy_in_range(const Params<T>& p, const long x, const long gy, const bool truncated)
{
	if (x >= p.cols - 1 || gy >= p.global_rows - 1 || x <= 0 || gy <= 0) return false;
	// NDRange = floor(n/8)*8 per axis (CBoundaryUniform.cpp:294-295, CBoundaryGridded.cpp:298-299; Q9)
	if (truncated && (x >= (p.cols / 8) * 8 || gy >= (p.global_rows / 8) * 8)) return false;
	return true;
}

constexpr int AREA_BDY_MAX = 8;
template <typename T> struct AreaBdy {
	int kind;                      // 0 uniform, 1 gridded
	UniformBdy<T> u;
	GriddedBdy<T> g;
};
template <typename T> struct AreaBdyList { int count; AreaBdy<T> b[AREA_BDY_MAX]; };


// What the fused epilogue of K1 carries per area boundary: the level increment of the tile's lower / upper rain-grid row
// (a uniform boundary has one value for both), the first row of the upper one, and how it is applied.
constexpr int FUSED_BDY_MAX = 3;
template <typename T> struct FusedBdy {
	T    inc_lo, inc_hi;                // per lane: metres added per application (uniform loss: metres removed)
	int  y_switch;                      // local row from which inc_hi applies
	int  mode;                          // -1 inactive, 0 add (uniform rain), 1 loss (floor at the bed), 2 add with the gridded kernel's extra null test
};

// What an iteration pair needs besides K1's arguments (round 6).
//  * Quirk Q3, exactly.  A cell the reference leaves untouched at the pair's FIRST step keeps what its destination buffer held: the
//    state of the iteration before the pair, with that iteration's boundaries -- a value that, when pairs follow each other, never
//    left the registers of the launch before.  That launch therefore writes it down where it could matter: a cell that may be dry at
//    the next step (only a dry cell can be left untouched) and whose first-step value differs from what the next launch will read
//    for it goes into z_state, stamped with the launch's number in z_gen; the launch's word haz[gen & 1] is set to gen.  A launch whose
//    source was written by pair launch `prev_gen` looks a first-step-untouched cell up (only if haz[prev_gen & 1] == prev_gen, i.e.
//    hardly ever) and takes the stamped value instead of the cell's current one -- which is what it takes, exactly, for every other
//    cell.  (Round 5 took the current value everywhere and had 881 181 probe cells to show for it)  The first single
//    iteration behind pairs (K1's FILL flag) reads the same stamps.
//  * Area boundaries (BDY).  See the kernel's header.
// A stamp record: the cell's first-step state followed by the number of the launch that wrote it (16-byte aligned: 48 B fp64, 32 B fp32) --
// one buffer, so that the march's masked stores of both go through ONE buffer resource.
template <typename T> constexpr unsigned stamp_rec() { return (unsigned)sizeof(State4<T>) + 16u; }
template <typename T> struct StampBufs { char* rec; unsigned long long* haz; };
template <typename T> __device__ __forceinline__ unsigned stamp_gen(const StampBufs<T>& b, const size_t id)
{
	return *reinterpret_cast<const unsigned*>(b.rec + id * stamp_rec<T>() + sizeof(State4<T>));
}
template <typename T> __device__ __forceinline__ State4<T> stamp_state(const StampBufs<T>& b, const size_t id)
{
	const T* v = reinterpret_cast<const T*>(b.rec + id * stamp_rec<T>());
	State4<T> c; c.z = v[0]; c.zmax = v[1]; c.qx = v[2]; c.qy = v[3];
	return c;
}
template <typename T> struct PairAux {
	StampBufs<T>          stamps;      // rec == nullptr: no stamps (nothing to look up, nothing written down).  (By value: kernel arguments are
	                                   // uniform by construction -- loaded through a pointer the addresses came back as per-lane values and every
	                                   // buffer access that used them was wrapped in a waterfall loop)
	unsigned              gen, prev_gen;
	const AreaBdyList<T>* list;        // (BDY) the domain's area boundaries, as K1's fused epilogue gets them
	int                   fuse_next;   // (BDY) another iteration of the same batch follows the pair: store the state with ITS boundaries applied
	int                   in_place;    // (BDY) the first iteration's boundaries are in the source buffer already (the launch before stored them / the stand-alone pass ran)
	int                   truncated;   // HP_QUIRK_BDY_TRUNCATED
};

// FUSED (round 3): the instantiation a domain with fusable area boundaries runs.  The reference applies rain / loss IN PLACE
// to an iteration's source buffer before its flux kernel (CSchemeGodunov.cpp:1637-1643).  Iteration n+1's source buffer
// is what iteration n writes, and everything the boundary kernels of iteration n+1 read -- t, t_hydro and the sign of
// dt -- follows from scalars that are known when the flux kernel of iteration n starts (tst_Advance_Normal:
// t += dt, t_hydro accumulates or restarts, CLDynamicTimestep.clc:42-66), EXCEPT the new dt, of which the uniform
// kernel only asks whether it is positive (CLBoundaries.clc:165-166).  So this kernel adds iteration n+1's increments to
// the state it is about to store: same operations per cell as the stand-alone pass, no second trip through HBM.  Cells it
// does not store (all-dry neighbourhoods, quirk Q3) get theirs read-modify-write in the cold pass after the row loop.
// The host asks for it (`fuse_next`) on every iteration of a batch but the last -- what a download sees between batches
// is the reference's buffer, without the next iteration's rain -- and the kernel declines when dt's sign is not certain
// (within VERY_SMALL of the sync point, at the end time): the word at cfl_slot[SLOT_BDY] tells the stand-alone pass of
// the next iteration whether there is anything left for it to do.
// waves per SIMD the register allocator is asked to make room for (as muscl_waves below): fp64 runs three (FAST needs 147
// VGPRs; STRICT with the shared reciprocals of round 4 would take 181 if left alone -- the scheduler interleaves the quotient
// chains -- and lose the third wave; STRICT at TWO waves, no spills, measured 5.5 % behind three waves with 30-44 spilled registers:
// profiles/r04s_dpp_rotate_and_k1_strict_waves_ab.txt, -DHP_K1_STRICT_WAVES=2)
#ifndef HP_K1_STRICT_WAVES
#define HP_K1_STRICT_WAVES 3
#endif
// (FAST at FOUR waves: 128 VGPRs, 28-70 spilled, S-DAM 0.25 -> 0.35 ms, S-RAIN 0.32 -> 0.49: -DHP_K1_FAST_WAVES=4, round 4)
#ifndef HP_K1_FAST_WAVES
#define HP_K1_FAST_WAVES 3
#endif
#ifndef HP_K1_F32_WAVES
#define HP_K1_F32_WAVES 5
#endif
template <bool STRICT, typename T> constexpr int march_waves() { return sizeof(T) == 4 ? HP_K1_F32_WAVES : (STRICT ? HP_K1_STRICT_WAVES : HP_K1_FAST_WAVES); }

// SPEC (STRICT fp64 only): the speculative flavour of a STRICT batch -- quotients that share a denominator share its refined
// reciprocal (hp_math.hpp: div_shared), a lane whose operands fall outside what that covers raises the domain's SLOT_SPEC
// word at the end of its tile, and the host re-runs the batch with the plain instantiation (hp_engine.hip: spec_resolve)
// (round 5: the FAST fp64 flavour in depth form needs 125-129 VGPRs -- on the edge of a FOURTH wave per SIMD, which the hardware would
// grant by itself below 129 and which measured SLOWER: S-DAM 0.251 against 0.240 ms, S-ROUGH 0.283 against 0.276, the 4096 x 514 strip
// 36.7 against 34.2 us, profiles/r05k_three_way_ab.txt -- a row march wants its rows' requests in flight, not more marchers per SIMD.
// So the attribute names the maximum too: three waves, whatever the register count comes to.)
template <bool STRICT, typename T> constexpr int march_waves_max() { return sizeof(T) == 4 ? 8 : march_waves<STRICT, T>(); }
template <bool STRICT, int CFL_MODE, bool FUSED, int TAIL, typename T, bool SPEC = false>          // TAIL: 0 none, 1 tail block, 2 tail block + ghost rows stored into the neighbours
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(march_waves<STRICT, T>(), march_waves_max<STRICT, T>()))) void godunov_march(const Params<T> p, const Scalars<T>* sc,
                                                     const T* __restrict__ bed, const State4<T>* __restrict__ src,
                                                     State4<T>* __restrict__ dst, const T* __restrict__ manning,
                                                     T* cfl_slot, const T* __restrict__ edge_max,
                                                     const TileMap tm, const AreaBdyList<T>* __restrict__ fused_list,
                                                     const int fuse_next, const int flags, const LaunchTail<T> tail)
{
	if (TAIL != 0 && blockIdx.x >= tail.flux_blocks) {                             // the launch's own tail (LaunchTail above)
		launch_tail(p, tail);
		return;
	}
	// the wave index is made a scalar explicitly: everything derived from it (tile rows, buffer descriptors, row
	// offsets) then lives in SGPRs and the buffer accesses need no waterfall loop
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	long strip, y0, y1;                                                            // rows [y0, y1) are updated
	T wave_vmax = T(0);                                                            // (for the tail block, when there is one)
	if (tile_rows(tm, wave, strip, y0, y1)) {                                      // wave-uniform

	const long x = strip * MARCH_COLS + lane;
	const long xc = (x < p.cols) ? x : (p.cols - 1);                               // clamp halo lanes past the grid
	const bool out_x = lane >= 1 && lane <= MARCH_COLS && x <= p.cols - 2;         // x >= 1 is implied

	const T dt = sc->dt, vs = p.vs;
	constexpr bool PL = !SPEC;                 // plain divisions (everything but a speculative STRICT fp64 batch)
	T* const spec_bad = cfl_slot + SLOT_SPEC;  // (SPEC) raised when a quotient falls outside what the shared-reciprocal division covers
	const bool skip_step = dt <= T(0);                                             // CLSchemeGodunov.clc:201-206
	const bool with_friction = p.friction != 0;
	T vmax = T(0);
	unsigned long long stale_rows = 0;   // bit i: row y0+i of this lane was left untouched (all-dry, Q3); rseg <= 64
	const bool truncated = (flags & 1) != 0;                                       // HP_QUIRK_BDY_TRUNCATED
	// FILL: the destination buffer holds nothing to build on (iteration pairs have run since it was last written: hp_engine.hip,
	// run_pair) -- the cells the reference leaves untouched (Q3) are stored too, with the value the host's repair copy would have put
	// there, the source's
	const bool fill = (flags & 2) != 0;

	// ---- fused area boundaries of the NEXT iteration (see above) ----
	FusedBdy<T> fb[FUSED_BDY_MAX];
	bool fuse = false, fuse_flag = false;                                          // wave-uniform
	bool fuse_x = false;                                                           // this lane's column is inside the boundary kernels' range
	if (FUSED) {
		#pragma unroll
		for (int k = 0; k < FUSED_BDY_MAX; ++k) { fb[k].inc_lo = fb[k].inc_hi = T(0); fb[k].y_switch = 0; fb[k].mode = -1; }
		if (fuse_next) {
			// t and t_hydro as tst_Advance_Normal will leave them (CLDynamicTimestep.clc:42-66, the statements of advance_body)
			const T dt_n = fmax_(T(0), dt);
			const T t1 = sc->t + dt_n;
			const T th0 = sc->t_hydro;
			const T th1 = (th0 > T(1.0)) ? dt_n : (th0 + dt_n);
			// dt of the next iteration is positive whatever the new maximum is, unless the sync point or the end time
			// intervene (:112-137); only the uniform kernel asks (CLBoundaries.clc:165-166)
			const bool dt_positive = (sc->t_sync - t1 > p.vs) && (t1 < p.t_end) && (p.dynamic_dt || p.dt_fixed > T(0));
			const int nb = fused_list->count;
			bool has_uniform = false;
			#pragma unroll
			for (int k = 0; k < FUSED_BDY_MAX; ++k) has_uniform = has_uniform || (k < nb && fused_list->b[k].kind == 0);
			fuse_flag = !has_uniform || dt_positive;
			fuse = fuse_flag && th1 >= T(1.0);                                         // the hydrological gate (:165, :224)
			if (fuse) {
				const long tile_rows_n = y1 - y0;
				fuse_x = x >= 1 && x <= p.cols - 2 && (!truncated || x < (p.cols / 8) * 8);     // bdy_in_range, column part
				#pragma unroll
				for (int k = 0; k < FUSED_BDY_MAX; ++k) {                                 // (static indices: fb[] lives in registers)
					if (k >= nb) continue;
					const AreaBdy<T>& b = fused_list->b[k];
					if (b.kind == 0) {
						if (t1 >= b.u.length) continue;                                       // :168
						unsigned long ts = (unsigned long)floor_(t1 / b.u.interval);          // :172-173
						if (ts >= b.u.entries) ts = b.u.entries - 1;
						const T amount = b.u.series[2 * ts + 1] / T(3600000.0) * th1;
						fb[k].inc_lo = fb[k].inc_hi = amount;
						fb[k].y_switch = (int)y0;
						fb[k].mode = b.u.definition == 1 ? 1 : (b.u.definition == 0 ? 0 : -1);
					} else {
						if (b.g.definition != 0 && b.g.definition != 2) continue;           // accumulated depths: nothing is applied (:238-242)
						unsigned long ts = (unsigned long)floor_(t1 / b.g.interval);          // :228
						if (ts >= b.g.entries) ts = b.g.entries - 1;
						// this lane's rain-grid column, and the rain-grid row of tile row y0 + lane (one evaluation per
						// lane covers the tile's rows: the host only fuses grids whose cells are at least 64 model cells
						// wide and high, so a tile meets at most two grid rows and a wavefront at most two grid columns)
						T colf = floor_((((T)xc * p.dx) - b.g.off_x) / b.g.resolution);          // :231
						if (colf < T(0)) colf = T(0);
						const long gy_l = y0 + lane + p.row_offset;
						const T rowf = floor_((((T)gy_l * p.dx) - b.g.off_y) / b.g.resolution);  // :232
						const unsigned long col = (unsigned long)colf;
						const unsigned long c0 = (unsigned long)__builtin_amdgcn_readfirstlane((int)col);
						const unsigned long c1 = (c0 + 1 < b.g.gcols) ? c0 + 1 : c0;
						const long row_l = (long)rowf;
						const long r_lo = (long)__builtin_amdgcn_readfirstlane((int)row_l);
						const unsigned long long in_lo = __ballot(lane < tile_rows_n && row_l == r_lo);
						const long n_lo = (long)__popcll(in_lo);
						const long r_hi = (n_lo < tile_rows_n) ? r_lo + 1 : r_lo;
						const unsigned long base = (b.g.grows * b.g.gcols) * ts;
						const unsigned long rl = (unsigned long)(r_lo < 0 ? 0 : r_lo), rh = (unsigned long)(r_hi < 0 ? 0 : ((unsigned long)r_hi < b.g.grows ? r_hi : (long)b.g.grows - 1));
						const T g00 = b.g.grids[base + b.g.gcols * rl + c0], g01 = b.g.grids[base + b.g.gcols * rl + c1];
						const T g10 = b.g.grids[base + b.g.gcols * rh + c0], g11 = b.g.grids[base + b.g.gcols * rh + c1];
						const T rate_lo = (col == c0) ? g00 : g01, rate_hi = (col == c0) ? g10 : g11;
						if (b.g.definition == 0) {                                            // :238-239
							fb[k].inc_lo = rate_lo / T(3600000.0) * th1;
							fb[k].inc_hi = rate_hi / T(3600000.0) * th1;
						} else {                                                              // :241-242
							fb[k].inc_lo = rate_lo / (p.dx * p.dx) * th1;
							fb[k].inc_hi = rate_hi / (p.dx * p.dx) * th1;
						}
						fb[k].y_switch = (int)(y0 + n_lo);
						fb[k].mode = 2;
					}
				}
			}
		}
	}
	// one cell's share of the next iteration's boundaries, in the order added: the statements of bdy_area
	auto apply_fused = [&](State4<T> c, const T zb, const long y) {
		const long gy = y + p.row_offset;
		const bool row_ok = gy >= 1 && gy <= p.global_rows - 2 && (!truncated || gy < (p.global_rows / 8) * 8);
		if (!row_ok) return c;                                                    // wave-uniform
		const bool cell_ok = fuse_x && !(c.zmax <= T(-9999.0));                   // :168-169, :220-221 (first half)
		#pragma unroll
		for (int k = 0; k < FUSED_BDY_MAX; ++k) {
			if (fb[k].mode < 0) continue;                                         // wave-uniform
			const T inc = ((int)y >= fb[k].y_switch) ? fb[k].inc_hi : fb[k].inc_lo;
			T z2;
			if (fb[k].mode == 1) z2 = fmax_(zb, c.z - inc);                       // :179-180
			else                 z2 = c.z + inc;                                  // :176-177, :238-242
			const bool ok = cell_ok && !(fb[k].mode == 2 && c.z == T(-9999.0));   // :220-221 (second half; re-tested as the level changes)
			c.z = ok ? z2 : c.z;
		}
		return c;
	};

	// the wave's window of the four arrays, addressed from its first cell (row y0-1, column of lane 0)
	const size_t cell0 = (size_t)(y0 - 1) * p.cols + (size_t)(strip * MARCH_COLS);
	const size_t cells_left = (size_t)p.cols * p.rows - cell0;
	const __amdgpu_buffer_rsrc_t srd_src = make_srd(src + cell0, cells_left * sizeof(State4<T>));
	const __amdgpu_buffer_rsrc_t srd_dst = make_srd(dst + cell0, cells_left * sizeof(State4<T>));
	const __amdgpu_buffer_rsrc_t srd_bed = make_srd(bed + cell0, cells_left * sizeof(T));
	const __amdgpu_buffer_rsrc_t srd_man = make_srd(manning + cell0, cells_left * sizeof(T));
	// (TAIL == 2) the same window of the neighbours' buffers: a row of the edge ranges is stored there as well, every other
	// row presents the out-of-range offset -- one more store instruction per row, no branch around it (see the store below)
	const __amdgpu_buffer_rsrc_t srd_peer0 = make_srd((TAIL == 2 && tail.peer_rows[0] ? tail.peer_rows[0] : dst) + cell0, cells_left * sizeof(State4<T>));
	const __amdgpu_buffer_rsrc_t srd_peer1 = make_srd((TAIL == 2 && tail.peer_rows[1] ? tail.peer_rows[1] : dst) + cell0, cells_left * sizeof(State4<T>));
	const unsigned lane_col = (unsigned)(xc - strip * MARCH_COLS);                // clamped column within the window
	const unsigned voff_state = lane_col * (unsigned)sizeof(State4<T>), voff_scalar = lane_col * (unsigned)sizeof(T);
	const unsigned row_state = (unsigned)p.cols * (unsigned)sizeof(State4<T>), row_scalar = (unsigned)p.cols * (unsigned)sizeof(T);
	auto store_peer = [&](const State4<T>& v, const long y, const bool write_lane) {
		const bool e0 = (int)y >= tail.edge_rows[0] && (int)y < tail.edge_rows[1];   // wave-uniform
		const bool e1 = (int)y >= tail.edge_rows[2] && (int)y < tail.edge_rows[3];
		buf_store_state<HP_AUX_THROUGH>(v, e1 ? srd_peer1 : srd_peer0, (write_lane && (e0 || e1)) ? voff_state : HP_OOB, (unsigned)(y - (y0 - 1)) * row_state);
	};

	// `live` = false: a prefetch slot that no row step will read (the row beyond the tile's north halo row).  The lanes
	// then present an out-of-range offset: the range check answers with zeros and NO memory request is made -- the slot
	// keeps its place in the wave's vmcnt sequence without fetching a row from HBM (round 2 fetched 21 rows per 18-row
	// tile; 20 are needed).
	auto load_row = [&](const long y, const bool live = true) {
		RowRegs<T> r;
		const unsigned k = (unsigned)(y - (y0 - 1));                               // wave-uniform -> SGPR offsets
		unsigned vs_ = live ? voff_state : HP_OOB, vc_ = live ? voff_scalar : HP_OOB;
		asm volatile("" : "+v"(vs_), "+v"(vc_));                                   // a select, never a branch around the loads
		r.c = buf_load_state(srd_src, vs_, k * row_state, T());
		r.zb = buf_load_scalar(srd_bed, vc_, k * row_scalar, T());
		r.n = p.manning_uniform ? p.manning_value : buf_load_scalar(srd_man, vc_, k * row_scalar, T());
		return r;
	};

	// pipeline fill: row y0-1 (south of the segment) gives the first south face.  Rows are fetched two ahead of the
	// one being updated, into two landing sets P and Q that alternate (the loop is unrolled twice): the row in flight
	// is never copied, so the only wait for it is at its first real use one iteration later.  (A rolled loop has to
	// copy "in flight" -> "next" at the bottom of every iteration, which parks the wave on vmcnt right there and
	// collapses the prefetch distance to less than one row -- measured with in-kernel stamps: 30-50 % of a row's
	// cycles sat in that copy.)
	RowRegs<T> rc = load_row(y0);
	RowRegs<T> rP = load_row(y0 + 1), rQ;                                          // y0+1 <= rows-1 always
	Side<T> sC;
	FaceFlux<T> fS = {};
	bool dryS;
	// Still water (wave-uniform, round 4; K2 has had the same skip since round 3).  A row whose 64 cells hold ONE wet state at rest
	// -- one level, one bed, zero discharge -- between a row south and a row north that hold the same state cell for cell comes out
	// of the update exactly as it went in: each of a cell's four faces is solved on identical left and right states, opposite faces
	// return identical values, every flux difference is x - x = +0, the bed-slope term is (zb - zb) = +0 times something finite,
	// friction does not act on zero discharge, and Z - dt (+0) = Z, Q - dt (+0) = Q bit for bit, in FAST and in STRICT; only Zmax
	// may have to follow Z (:375-376).  Such a row skips its faces, the north side and the update -- and leaves what the rows hand
	// each other, the current row's side `sC` and its south flux `fS`, untouched: the row south of a still row holds the still row's
	// state cell for cell (that is part of the test), so fS already IS the face between two such cells and sC the side of such a
	// cell, and the row north of it holds that state too: whoever comes next, still or not, inherits exactly the values the full
	// path would have produced for it (same values in, same bits out).  Lakes, reservoirs, the sea and both pools of the dam-break
	// benchmark are such water.
	// STRICT only.  The exact flavour's row costs 1300 VALU instructions and a still row 60: S-DAM 4096^2 0.436 -> 0.303 ms, the
	// same bits.  The FAST flavour's launch is bound by the memory system in such water, not by its 420 instructions per row:
	// with the skip it was 2 % SLOWER on S-DAM and 3-4 % slower where nothing is still (S-RAIN, S-ROUGH: the tests and what they
	// did to the schedule) -- profiles/r04w_still_and_sweep_ab.txt.
	constexpr bool STILL_SKIP = STRICT;
	// (the wave-uniform facts below share ONE scalar register: as separate bools -- an SGPR pair each -- they pushed the fused
	// flavour's scalar spills into the vector registers and those into scratch)
	constexpr unsigned REST_S = 1, REST_C = 2, EQ_S = 4;   // every cell of the row south of the current one / of the current row is at
	                                                       // rest (Q = 0); the two rows hold the same level and bed, cell for cell
	constexpr unsigned PRICED = 16;                        // the still row below has put this (one) state's wave speed into vmax
	unsigned st = 0;
	auto at_rest = [&](const RowRegs<T>& r) { return wave_all(r.c.qx == T(0) && r.c.qy == T(0)) != 0; };
	auto same_level = [&](const RowRegs<T>& a, const RowRegs<T>& b) { return wave_all(a.c.z == b.c.z && a.zb == b.zb) != 0; };
	{
		const RowRegs<T> rs = load_row(y0 - 1);
		const Side<T> sS = make_side_impl<STRICT, PL>(rs.c.z, rs.c.qx, rs.c.qy, rs.zb, vs, spec_bad);
		sC = make_side_impl<STRICT, PL>(rc.c.z, rc.c.qx, rc.c.qy, rc.zb, vs, spec_bad);
		dryS = (rs.c.z - rs.zb) < vs;
		if (STILL_SKIP) {
			if (at_rest(rs)) st |= REST_S;
			if (at_rest(rc)) st |= REST_C;
			if (st == (REST_S | REST_C) && same_level(rs, rc)) st |= EQ_S;
		}
		// both sides are asked for although only the north cell's is used: the tile below finishes this same face as
		// ITS north face, and the two must take the same code path or results would depend on where tiles (and strip
		// boundaries) fall
		if (!skip_step) fS = face_solve_impl<AXIS_Y, STRICT, true, true, PL>(sS, sC, vs, spec_bad).forR;
	}

	// one row: `rc` is updated, `rn` is its northern neighbour (already landed), `pre` receives the prefetch of row y+2
	auto row_step = [&](const long y, const RowRegs<T>& rn, RowRegs<T>& pre) {
		pre = load_row((y + 2 <= y1) ? (y + 2) : y1, y + 2 <= y1);                 // prefetch; nothing beyond the north halo row y1
		State4<T> out = rc.c;
		bool write = out_x;
		Side<T> sN = sC;
		bool skip_cfl = false;

		// still water (see above): one state across the wavefront, the same state north and south.  Staged: moving water pays for the
		// two compares of `rest_n` only
		const bool rest_n = STILL_SKIP && at_rest(rn);
		const bool eq_n = STILL_SKIP && (st & REST_C) && rest_n && same_level(rc, rn);
		bool still = false;
		if (STILL_SKIP && !skip_step && eq_n && (st & (EQ_S | REST_S)) == (EQ_S | REST_S)) {
			const T z_first = first_lane(rc.c.z), b_first = first_lane(rc.zb);
			still = wave_all(rc.c.z == z_first && rc.zb == b_first && (rc.c.z - rc.zb) > vs &&
			              !(rc.c.zmax <= T(-9999.0) || rc.c.z == T(-9999.0))) != 0;
		}
		st = (st & PRICED) | ((st & REST_C) ? REST_S : 0u) | (rest_n ? REST_C : 0u) | (eq_n ? EQ_S : 0u);

		if (still) {
			if (out.z > out.zmax && out.zmax > T(-9990.0)) out.zmax = out.z;              // :375-376, all that is left of the update
			dryS = false;
			// CFL epilogue: ONE state across the wave; if the still row below (the same state, by its own test) has priced it, pricing
			// it again cannot change the maximum
			skip_cfl = (st & PRICED) != 0;
			if ((int)y >= tm.price_lo && (int)y < tm.price_hi && wave_any(out_x && out.zmax > T(-9999.0))) st |= PRICED;
		} else if (!skip_step) {
			st &= ~PRICED;
			// east face first: its result has to travel to the next lane while the north face is solved
			const Side<T> sE = side_from_east(sC);
			const FacePair<T> fx = face_solve_impl<AXIS_X, STRICT, true, true, PL>(sC, sE, vs, spec_bad);
			const FaceFlux<T> fE = fx.forL, forW = fx.forR;
			FaceFlux<T> fW;
			fW.f0 = from_west(forW.f0); fW.fx = from_west(forW.fx);
			fW.fy = from_west(forW.fy); fW.eta_nb = from_west(forW.eta_nb);
			fW.zb_nb = from_west(forW.zb_nb);
			// the five dry tests of :248-255.  (FAST) a face whose lanes ALL have water on both sides (FacePair::wet) lies between two wet
			// cells in every lane -- a cell's depth is at least its depth above the face's common bed --, so such a wavefront skips the tests
			// (fp64 only: in fp32 the two branches cost what the six instructions save -- profiles/r05fm_three_builds_ab.txt)
			constexpr bool LAZY_DRY = !STRICT && sizeof(T) == 8;
			bool dryC = false, dryE = false, dryN = false;
			if (!LAZY_DRY || !fx.wet) {
				asm volatile("");                                                     // (a real branch: speculated, its six instructions ran on every row)
				dryC = (rc.c.z - rc.zb) < vs;
				dryE = (sE.eta - sE.zb) < vs;
			}
			const int flags = from_west((int)forW.stop | ((int)dryC << 1));
			fW.stop = (flags & 1) != 0;
			const bool dryW = (flags & 2) != 0;

			sN = make_side_impl<STRICT, PL>(rn.c.z, rn.c.qx, rn.c.qy, rn.zb, vs, spec_bad);
			const FacePair<T> fy = face_solve_impl<AXIS_Y, STRICT, true, true, PL>(sC, sN, vs, spec_bad);
			const FaceFlux<T> fN = fy.forL;
			if (!LAZY_DRY || !fy.wet) {
				asm volatile("");
				dryN = (rn.c.z - rn.zb) < vs;
			}

			const bool disabled = rc.c.zmax <= T(-9999.0) || rc.c.z == T(-9999.0);     // :214-218
			const bool dry5 = dryC && dryN && dryE && dryS && dryW;                   // :248-255
			const State4<T> upd = godunov_update_impl<STRICT, false, PL>(rc.c, rc.zb, rc.n, dt, fN, fE, fS, fW, p.dx, STRICT ? p.inv_dx_pow2 : p.inv_dx, vs,
			                                                            with_friction, spec_bad);
			if (!disabled) {
				if (dry5) {                                                               // dst untouched (Q3)
					if (!fill) {                                                          // (FILL: stored, priced and rained on right here, as `out` = the source's value)
						write = false;
						if (out_x) stale_rows |= 1ull << (unsigned)(y - y0);
					}
				} else {
					out = upd;
				}
			}
			fS = fy.forR;
			dryS = dryC;
		}

		// One unconditional store per row: lanes that must not write (halo lanes, all-dry cells) present an
		// out-of-range offset, which the buffer range check drops, instead of branching around the store.  That
		// keeps the wave's vmcnt bookkeeping static so the wait for the prefetched row never has to drain the
		// stores behind it.
		// (FUSED) what is stored already carries the next iteration's rain / loss; the CFL epilogue prices the state WITHOUT
		// it, as the reference's reduction runs before the next iteration's boundary kernels
		State4<T> stored = out;
		if (FUSED && fuse) stored = apply_fused(out, rc.zb, y);
		buf_store_state(stored, srd_dst, write ? voff_state : HP_OOB, (unsigned)(y - (y0 - 1)) * row_state);
		if (TAIL == 2) store_peer(stored, y, write);
		const bool priced = (int)y >= tm.price_lo && (int)y < tm.price_hi && !skip_cfl;       // wave-uniform
		if (!priced) {
		} else if (CFL_MODE == 1) {
			if (write) {
				const T s = cfl_speed_impl<STRICT, PL>(out.z, out.zmax, out.qx, out.qy, rc.zb, p.qs, false, spec_bad);
				if (s > vmax) vmax = s;
			}
		} else if (CFL_MODE == 2) {
			if (out_x) {
				const T s = cfl_speed_impl<STRICT, PL>(rc.c.z, rc.c.zmax, rc.c.qx, rc.c.qy, rc.zb, p.qs, false, spec_bad);
				if (s > vmax) vmax = s;
			}
		}
		rc = rn;                     // a copy of values that have already been used: no wait attached to it
		sC = sN;
	};

	long y = y0;
	// Dry land at the start of the tile (most tiles of a flood domain away from the water, entirely).  While every lane's
	// cell and the cells north and south of it are dry -- for the updated lanes 1..62 the east and west neighbours are
	// lanes of this wave, so all five dry tests of :248-255 hold -- the reference updates nothing (:254-255): neither the
	// east faces nor the update are needed, and the north face (the next row's south flux, should that row turn out to
	// be live) lies between two dry cells: the dry-dry form of the solver, taken directly (face_dry_for_right, the
	// same statements face_solve executes for such a lane).  The run ends at the first row that is not dry throughout;
	// the general loop below takes over from there and is not touched by any of this.
	if (!skip_step) {
		while (y < y1) {
			const RowRegs<T>& rn = rP;
			const bool dryC = (rc.c.z - rc.zb) < vs, dryN = (rn.c.z - rn.zb) < vs;
			if (!wave_all(dryC && dryN && dryS)) break;
			rQ = load_row((y + 2 <= y1) ? (y + 2) : y1, y + 2 <= y1);
			const bool disabled = rc.c.zmax <= T(-9999.0) || rc.c.z == T(-9999.0);     // :214-218
			const bool write = out_x && (disabled || fill);                            // nulls are carried, dry cells untouched (Q3; FILL: carried too)
			if (out_x && !disabled) stale_rows |= 1ull << (unsigned)(y - y0);
			const Side<T> sN = make_side_impl<STRICT, PL>(rn.c.z, rn.c.qx, rn.c.qy, rn.zb, vs, spec_bad);
			fS = face_dry_for_right<AXIS_Y, STRICT>(sC, sN, vs);
			// (FUSED: the only cells this loop stores are nulls, which no boundary kernel touches -- and, with FILL, the dry cells, which
			// the cold pass below stores again with their rain)
			buf_store_state(rc.c, srd_dst, write ? voff_state : HP_OOB, (unsigned)(y - (y0 - 1)) * row_state);
			if (TAIL == 2) store_peer(rc.c, y, write);
			if (!((int)y >= tm.price_lo && (int)y < tm.price_hi)) {
			} else if (CFL_MODE == 1) {
				if (write) {
					const T s = cfl_speed_impl<STRICT, PL>(rc.c.z, rc.c.zmax, rc.c.qx, rc.c.qy, rc.zb, p.qs, false, spec_bad);
					if (s > vmax) vmax = s;
				}
			} else if (CFL_MODE == 2) {
				if (out_x) {
					const T s = cfl_speed_impl<STRICT, PL>(rc.c.z, rc.c.zmax, rc.c.qx, rc.c.qy, rc.zb, p.qs, false, spec_bad);
					if (s > vmax) vmax = s;
				}
			}
			dryS = dryC;
			rc = rP; sC = sN; rP = rQ;
			st = 0;                                     // (dry rows are not still WATER; the general loop finds out for itself)
			++y;
		}
	}
	for (; y + 2 <= y1; y += 2) {
		row_step(y, rP, rQ);
		row_step(y + 1, rQ, rP);
	}
	if (y < y1) row_step(y, rP, rQ);

	if ((CFL_MODE == 1 || (FUSED && fuse)) && wave_any(stale_rows != 0)) {
		// cells the reference leaves untouched still hold their two-steps-old value in dst, and tst_Reduce prices it;
		// (FUSED) the next iteration's boundary kernels would change exactly that value in place: read-modify-write
		for (long y = y0; y < y1; ++y) {
			if ((stale_rows >> (unsigned)(y - y0)) & 1ull) {
				const size_t id = (size_t)y * p.cols + xc;
				const State4<T> c = fill ? src[id] : dst[id];
				const T zb = bed[id];
				if (CFL_MODE == 1 && (int)y >= tm.price_lo && (int)y < tm.price_hi) {
					const T s = cfl_speed_impl<STRICT, PL>(c.z, c.zmax, c.qx, c.qy, zb, p.qs, false, spec_bad);
					if (s > vmax) vmax = s;
				}
				if (FUSED && fuse) {
					const State4<T> rained = apply_fused(c, zb, y);
					dst[id] = rained;
					if (TAIL == 2) store_peer(rained, y, true);                          // (the neighbour's copy of the cell gets the same store)
				}
			}
		}
	}

	if (CFL_MODE != 0) {
		// the edge ring (never written, priced at upload) joins the maximum before any cross-rank all-reduce
		if (blockIdx.x == 0 && wave == 0) { const T e = *edge_max; if (e > vmax) vmax = e; }
		vmax = wave_max(vmax);
		if (TAIL != 0) wave_vmax = vmax;
		else if (lane == 0 && vmax > T(0)) atomic_max_nonneg(cfl_slot, vmax);
	}
	// (FUSED) tell the next iteration's stand-alone boundary pass whether anything is left for it to do.  Every wavefront
	// reaches the same decision from the same scalars; one of them writes it down.
	if (FUSED && blockIdx.x == 0 && wave == 0 && lane == 0) cfl_slot[SLOT_BDY] = fuse_flag ? T(1) : T(0);
	}   // tile / strip guard
	if (TAIL != 0) tail_block_done(tail, wave_vmax, wave, lane, y0, y1);
}

// -------------------------------------------------------------------------------------------------
// K1b godunov_march2 : TWO Godunov iterations in one pass (round 5; FAST flavour, no boundary conditions; single domains with
//     TAIL 1, row strips that store two reaches of ghost rows with TAIL 2).
//
//  Why it is possible.  Quirk Q1: the reference's reduction always prices the PRIMARY state buffer.  An iteration that reads the
//  primary buffer (every other one) therefore re-prices the state it started from: the timestep of the iteration after it follows
//  from a maximum that is already known (cfl_slot[SLOT_SAVED]) and from time-control scalars -- every wavefront can work it out
//  before anything is launched (advance_scalars).  So the pair (k: primary -> other, k + 1: other -> primary) is one pass:
//  read state k once, write state k + 2 once -- 40 B per cell-step instead of 80 (tools/membench: the march in this shape moves a
//  pair in 0.245-0.255 ms at 4096^2, where one step in K1's shape takes 0.246).
//  What it costs.  A two-cell halo: 60 updated columns per wavefront instead of 62, and the first step is evaluated on two more
//  rows than the tile has; the second step's registers.
//  How.  The march of K1 twice, one row apart: stage A turns source row r into the intermediate row U1(r) exactly as K1 would
//  have stored it; stage B, one row behind, turns U1(r - 2 .. r) into the final row r - 1.  Each stage carries its own side,
//  south flux and dry flags; the intermediate state never leaves the registers.  Edge-ring cells pass through stage A unchanged
//  (no kernel ever writes them).
//  Cells the reference leaves untouched (quirk Q3).  First step: its destination would keep what it held, and that is what the
//  second step would read; here the cell's CURRENT state stands in -- on every wet/dry workload tried (S-ROUGH, the dry-bed dam
//  break, config C1; 1750 iterations of the reference's kernels, tools/history/r05_q3_stale_probe.py) the two were equal in every such
//  cell at every step.  Second step: the primary buffer keeps state k, and this kernel stores exactly that.
//  The pass writes into the OTHER buffer; the host then swaps the two pointers (hp_engine.hip: run_pair), so "primary" is again
//  the buffer that holds the newest state, as after two single iterations.
// -------------------------------------------------------------------------------------------------
constexpr int MARCH2_COLS = 60;          // updated columns per wavefront (lanes 2..61)
template <typename T> struct RowU1 { State4<T> c; T zb; bool plain; };      // plain: c IS the cell's source value, bit for bit (nothing touched it at the first step)

// HZ: the instantiation that keeps quirk Q3 EXACT across pair launches (the stamps, PairAux).  What it costs the march -- a vote that joins
// an existing one, the bookkeeping of which intermediate cells are plain copies, the registers both take from a kernel that has none to
// spare -- measured 3.7 % on still water, 9 % where every tile is live and 8 % on the fp32 rain workload (profiles/r06f_stamps_ab.txt;
// a branch of its own per row cost 9-34 %, masked stores 6 %), so it runs where the stale values matter: domains whose boundaries
// REMOVE water (a loss rate dries whole regions at once, and their cells then sit untouched with stale values in the other buffer), or
// on request (HP_PAIR_EXACT=1).  Without it a first-step-untouched cell passes its current state on, as in round 5: equal to the stale
// one in every such cell-iteration of the probes (tools/history/r05_q3_stale_probe.py: 881 181 of them) -- unless the cell dried that very step.
template <bool STRICT, int CFL_MODE, bool BDY, bool HZ, int TAIL, typename T>   // CFL_MODE 1: price what the pair leaves in the primary buffer; 0: fixed timestep
// (three waves per SIMD: 167 VGPRs and four spilled registers measured 0.193 ms per iteration at 4096^2 against 0.215 at two waves
// and 171 registers without spills -- profiles/r05n_two_step.txt; -DHP_K1B_WAVES_MIN=2 builds the other one)
#ifndef HP_K1B_WAVES_MIN
#define HP_K1B_WAVES_MIN 3
#endif
// (STRICT: two waves per SIMD in fp64, three in fp32 -- no vector register of the exact flavour is spilled: tests/test_resource_usage.py)
// (fp32: FIVE waves per SIMD, 96 VGPRs -- against the 4-5 of round 5, where the allocator settled for four: C5's shape 8192^2 S-RAIN 0.5003 ->
// 0.4795 ms, S-DAM 4096^2 fp32 0.1062 -> 0.1005; three waves 0.560 / 0.118, six (80 VGPRs, 29-78 spilled) 0.775 / 0.152 --
// profiles/r06x_f32_pair_waves.txt.  The fp32 march is bound by latency, not by issue: a vector instruction takes it two cycles.)
#ifndef HP_K1B_F32_WAVES_MIN
#define HP_K1B_F32_WAVES_MIN 5
#endif
#ifndef HP_K1B_F32_WAVES_MAX
#define HP_K1B_F32_WAVES_MAX 5
#endif
// (the exact flavour, HZ, keeps round 5's four: at five its fp32 instantiations spill 20-70 vector registers)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sizeof(T) == 4 ? (STRICT ? 3 : (HZ ? 4 : HP_K1B_F32_WAVES_MIN)) : (STRICT ? HP_K1B_STRICT_WAVES : HP_K1B_WAVES_MIN), sizeof(T) == 4 ? (STRICT ? 3 : HP_K1B_F32_WAVES_MAX) : (STRICT ? HP_K1B_STRICT_WAVES : 3)))) void godunov_march2(
	const Params<T> p, const Scalars<T>* sc, const T* __restrict__ bed, const State4<T>* __restrict__ src,
	State4<T>* __restrict__ dst, const T* __restrict__ manning, T* cfl_slot, const T* __restrict__ edge_max,
	const TileMap tm, const LaunchTail<T> tail, const PairAux<T> aux)
{
	if (TAIL != 0 && blockIdx.x >= tail.flux_blocks) {
		launch_tail(p, tail);
		return;
	}
	// (BDY) increments of the three boundary applications a pair can meet -- [0] the first iteration's, on the source rows as they are
	// loaded (only when they are not in the buffer yet: the first pair of a batch), [1] the second iteration's, on the intermediate
	// state in registers, [2] the iteration's after the pair, on the state as it is stored -- per boundary, for the tile's lower / upper
	// rain-grid row, per lane.  They wait in LDS (one read per row and application) instead of eighteen registers.
	__shared__ T inc_tab[BDY ? 4 * 3 * FUSED_BDY_MAX * 2 * 64 : 1];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	long strip, y0, y1;                                                            // rows [y0, y1) receive state k + 2
	T wave_vmax = T(0), wave_vmax1 = T(0);
	if (tile_rows(tm, wave, strip, y0, y1)) {

	const long x = strip * MARCH2_COLS - 1 + lane;                                 // lane 2 is the strip's first updated column
	const long xc = x < 0 ? 0 : (x < p.cols ? x : (p.cols - 1));
	const bool out_x = lane >= 2 && lane <= MARCH2_COLS + 1 && x <= p.cols - 2;    // x >= 1 is implied
	const bool ring_x = x <= 0 || x >= p.cols - 1;                                 // an edge-ring column (or a lane beyond the grid): never updated
	const T vs = p.vs;
	const bool with_friction = p.friction != 0;
	const T inv_dx = STRICT ? p.inv_dx_pow2 : p.inv_dx;

	// the two timesteps: this iteration's, and the next one's as tst_Advance_Normal will leave it (see above).  With area boundaries
	// the first iteration's reduction prices the primary buffer with that iteration's rain in it: slot[SLOT_M1], left by the launch before
	Scalars<T> s1 = *sc;
	const Scalars<T> s0 = s1;
	const T dt_a = s1.dt;
	advance_scalars(p, s1, p.dynamic_dt ? cfl_slot[BDY ? SLOT_M1 : SLOT_SAVED] : T(0));
	const T dt_b = s1.dt;
	const bool skip_a = dt_a <= T(0), skip_b = dt_b <= T(0);                       // CLSchemeGodunov.clc:201-206: the state is copied
	// (FAST fp64: -(dt / dx) and dt g of the two steps, formed once at the kernel's start and handed to the update instead of being formed
	// there -- an fp64 product has no scalar instruction, so left alone they sit in eight vector registers for the whole march, which
	// this kernel does not have: they were what it spilled and re-loaded from scratch in front of every friction term.  The plain kernel
	// leaves the rest to the compiler (which keeps -(dt / dx) in scalar pairs and dt g in vector ones: S-DAM 4096^2 0.1861 -> 0.1829 ms,
	// S-ROUGH 0.246 -> 0.2397 on one box, no scratch left; profiles/r06af_scalar_step_constants_ab.txt); the kernels with area
	// boundaries or stamps force all four into scalar registers (uniform_value<HIDE>, hp_math.hpp: S-RAIN 0.2676 -> 0.2544 ms).
	// fp32 has the registers and lost 1.4 % with it)
	constexpr bool SCALAR_CONSTS = !STRICT && sizeof(T) == 8;
	constexpr bool HIDE = BDY || HZ;                                               // (hp_math.hpp: uniform_value)
	const StepConsts<T> ka{uniform_value<HIDE>(-(dt_a * p.inv_dx)), uniform_value<HIDE>(dt_a * gravity<T>()), SCALAR_CONSTS};
	const StepConsts<T> kb{uniform_value<HIDE>(-(dt_b * p.inv_dx)), uniform_value<HIDE>(dt_b * gravity<T>()), SCALAR_CONSTS};

	// the wave's window: rows from y0 - 2 (as far as the grid goes), columns from the strip's first halo column
	const long row_base = y0 - 2 < 0 ? 0 : y0 - 2, col_base = strip * MARCH2_COLS - 1 < 0 ? 0 : strip * MARCH2_COLS - 1;
	const size_t cell0 = (size_t)row_base * p.cols + (size_t)col_base;
	const size_t cells_left = (size_t)p.cols * p.rows - cell0;
	const __amdgpu_buffer_rsrc_t srd_src = make_srd(src + cell0, cells_left * sizeof(State4<T>));
	const __amdgpu_buffer_rsrc_t srd_dst = make_srd(dst + cell0, cells_left * sizeof(State4<T>));
	const __amdgpu_buffer_rsrc_t srd_bed = make_srd(bed + cell0, cells_left * sizeof(T));
	const __amdgpu_buffer_rsrc_t srd_man = make_srd(manning + cell0, cells_left * sizeof(T));
	const unsigned lane_col = (unsigned)(xc - col_base);
	const unsigned voff_state = lane_col * (unsigned)sizeof(State4<T>), voff_scalar = lane_col * (unsigned)sizeof(T);
	const unsigned row_state = (unsigned)p.cols * (unsigned)sizeof(State4<T>), row_scalar = (unsigned)p.cols * (unsigned)sizeof(T);
	const long last_row = p.rows - 1;
	// (TAIL == 2, a row strip) the final rows of the edge ranges are stored into the strip neighbours' ghost rows as well, written
	// through -- as in K1; a pair hands over two reaches of rows at once
	const __amdgpu_buffer_rsrc_t srd_peer0 = make_srd((TAIL == 2 && tail.peer_rows[0] ? tail.peer_rows[0] : dst) + cell0, cells_left * sizeof(State4<T>));
	const __amdgpu_buffer_rsrc_t srd_peer1 = make_srd((TAIL == 2 && tail.peer_rows[1] ? tail.peer_rows[1] : dst) + cell0, cells_left * sizeof(State4<T>));
	auto store_peer = [&](const State4<T>& v, const long y, const bool write_lane) {
		const bool e0 = (int)y >= tail.edge_rows[0] && (int)y < tail.edge_rows[1];   // wave-uniform
		const bool e1 = (int)y >= tail.edge_rows[2] && (int)y < tail.edge_rows[3];
		buf_store_state<HP_AUX_THROUGH>(v, e1 ? srd_peer1 : srd_peer0, (write_lane && (e0 || e1)) ? voff_state : HP_OOB, (unsigned)(y - row_base) * row_state);
	};
	auto load_row = [&](long y, const bool live = true) {                          // (rows beyond the grid: the nearest one -- only ring rows ask)
		y = y < 0 ? 0 : (y > last_row ? last_row : y);
		RowRegs<T> r;
		const unsigned k = (unsigned)(y - row_base);
		unsigned vs_ = live ? voff_state : HP_OOB, vc_ = live ? voff_scalar : HP_OOB;
		asm volatile("" : "+v"(vs_), "+v"(vc_));
		r.c = buf_load_state(srd_src, vs_, k * row_state, T());
		r.zb = buf_load_scalar(srd_bed, vc_, k * row_scalar, T());
		r.n = p.manning_uniform ? p.manning_value : buf_load_scalar(srd_man, vc_, k * row_scalar, T());
		return r;
	};
	constexpr bool LAZY_DRY = !STRICT && sizeof(T) == 8;                           // (K1's row step: the dry tests only where a face is not wet throughout)
	// west flux of a cell = what the lane to its west found for its east face (as in K1)
	auto flux_from_west = [&](const FaceFlux<T>& forW, const bool dryC, bool& dryW) {
		FaceFlux<T> fW;
		fW.f0 = from_west(forW.f0); fW.fx = from_west(forW.fx); fW.fy = from_west(forW.fy);
		fW.eta_nb = from_west(forW.eta_nb); fW.zb_nb = from_west(forW.zb_nb);
		const int flags = from_west((int)forW.stop | ((int)dryC << 1));
		fW.stop = (flags & 1) != 0;
		dryW = (flags & 2) != 0;
		return fW;
	};

	// ---- quirk Q3 at the first step: the stamps of the launch that wrote `src` (PairAux) ----
	const bool hz_on = HZ && aux.stamps.rec != nullptr;
	constexpr unsigned REC = stamp_rec<T>();
	bool stamped = false;                                  // (LIVE launches, per lane) wrote a stamp: the launch's word is raised after the march
	const bool hz_any = hz_on && aux.prev_gen != 0 && aux.stamps.haz[aux.prev_gen & 1u] == (unsigned long long)aux.prev_gen;     // wave-uniform (scalar loads)

	// ---- (BDY) which of the three applications are live in THIS launch: scalar arithmetic on the time-control block ----
	unsigned act = 0;                       // bit j: application j has at least one boundary that acts
	unsigned modes = 0x3ffffu;              // 2 bits per (j, k): 0 add (uniform rain), 1 loss (floor at the bed), 2 add with the gridded kernel's null test, 3 inactive
	int ysw[FUSED_BDY_MAX] = {0, 0, 0};     // per boundary: the tile row from which the upper rain-grid row's increment applies
	bool bdy_x = false;
	long gy_lim = 0;
	const bool truncated = BDY && aux.truncated != 0;
	T* const tab = inc_tab + (BDY ? wave * (3 * FUSED_BDY_MAX * 2 * 64) : 0);
	if (BDY) {
		// time, hydrological time and the sign of the timestep as the boundary kernels of iterations k, k + 1, k + 2 see them
		// (tst_Advance_Normal, CLDynamicTimestep.clc:42-66; the sign of dt(k + 2) is decided by the sync point and the end time
		// alone, whatever this launch's maximum turns out to be: :112-137 -- K1's fused epilogue rests on the same fact)
		const T dtn = fmax_(T(0), s1.dt);
		const T t2 = s1.t + dtn;
		const T th2 = (s1.t_hydro > T(1.0)) ? dtn : (s1.t_hydro + dtn);
		const bool pos2 = (s1.t_sync - t2 > p.vs) && (t2 < p.t_end) && (p.dynamic_dt || p.dt_fixed > T(0));
		const T tj[3] = {s0.t, s1.t, t2}, thj[3] = {s0.t_hydro, s1.t_hydro, th2};
		const bool posj[3] = {s0.dt > T(0), s1.dt > T(0), pos2};
		bdy_x = x >= 1 && x <= p.cols - 2 && (!truncated || x < (p.cols / 8) * 8);     // bdy_in_range, column part
		gy_lim = truncated ? (p.global_rows / 8) * 8 : p.global_rows;
		const int nb = aux.list->count;
		const long n_rows = ((y1 + 1 < last_row ? y1 + 1 : last_row) - row_base) + 1;   // rows the wavefront touches (<= 36)
		#pragma unroll
		for (int j = 0; j < 3; ++j) {
			if (j == 0 && aux.in_place) continue;                                     // in the buffer already
			if (!(thj[j] >= T(1.0))) continue;                                        // the hydrological gate (CLBoundaries.clc:165, :224)
			#pragma unroll
			for (int k = 0; k < FUSED_BDY_MAX; ++k) {
				if (k >= nb) continue;
				const AreaBdy<T>& b = aux.list->b[k];
				T inc_lo = T(0), inc_hi = T(0);
				unsigned mode = 3u;
				if (b.kind == 0) {
					if (!posj[j] || tj[j] >= b.u.length) continue;                    // :165-168
					unsigned long ts = (unsigned long)floor_(tj[j] / b.u.interval);   // :172-173
					if (ts >= b.u.entries) ts = b.u.entries - 1;
					inc_lo = inc_hi = b.u.series[2 * ts + 1] / T(3600000.0) * thj[j];
					mode = b.u.definition == 1 ? 1u : (b.u.definition == 0 ? 0u : 3u);
				} else {
					if (b.g.definition != 0 && b.g.definition != 2) continue;         // accumulated depths: nothing is applied (:238-242)
					unsigned long ts = (unsigned long)floor_(tj[j] / b.g.interval);   // :228
					if (ts >= b.g.entries) ts = b.g.entries - 1;
					// this lane's rain-grid column, and the rain-grid row of window row row_base + lane (K1's fused set-up: the host
					// only fuses grids whose cells are at least 64 model cells wide and high, so a wavefront's rows meet at most two
					// grid rows and its columns two grid columns)
					T colf = floor_((((T)xc * p.dx) - b.g.off_x) / b.g.resolution);   // :231
					if (colf < T(0)) colf = T(0);
					const long gy_l = row_base + lane + p.row_offset;
					const T rowf = floor_((((T)gy_l * p.dx) - b.g.off_y) / b.g.resolution);   // :232
					const unsigned long col = (unsigned long)colf;
					const unsigned long c0 = (unsigned long)__builtin_amdgcn_readfirstlane((int)col);
					const unsigned long c1 = (c0 + 1 < b.g.gcols) ? c0 + 1 : c0;
					const long row_l = (long)rowf;
					const long r_lo = (long)__builtin_amdgcn_readfirstlane((int)row_l);
					const unsigned long long in_lo = __ballot(lane < n_rows && row_l == r_lo);
					const long n_lo = (long)__popcll(in_lo);
					const long r_hi = (n_lo < n_rows) ? r_lo + 1 : r_lo;
					const unsigned long base = (b.g.grows * b.g.gcols) * ts;
					const unsigned long rl = (unsigned long)(r_lo < 0 ? 0 : r_lo), rh = (unsigned long)(r_hi < 0 ? 0 : ((unsigned long)r_hi < b.g.grows ? r_hi : (long)b.g.grows - 1));
					const T g00 = b.g.grids[base + b.g.gcols * rl + c0], g01 = b.g.grids[base + b.g.gcols * rl + c1];
					const T g10 = b.g.grids[base + b.g.gcols * rh + c0], g11 = b.g.grids[base + b.g.gcols * rh + c1];
					const T rate_lo = (col == c0) ? g00 : g01, rate_hi = (col == c0) ? g10 : g11;
					if (b.g.definition == 0) {                                        // :238-239
						inc_lo = rate_lo / T(3600000.0) * thj[j];
						inc_hi = rate_hi / T(3600000.0) * thj[j];
					} else {                                                          // :241-242
						inc_lo = rate_lo / (p.dx * p.dx) * thj[j];
						inc_hi = rate_hi / (p.dx * p.dx) * thj[j];
					}
					ysw[k] = (int)(row_base + n_lo);
					mode = 2u;
				}
				if (mode == 3u) continue;
				tab[((j * FUSED_BDY_MAX + k) * 2 + 0) * 64 + lane] = inc_lo;
				tab[((j * FUSED_BDY_MAX + k) * 2 + 1) * 64 + lane] = inc_hi;
				modes = (modes & ~(3u << (2 * (j * FUSED_BDY_MAX + k)))) | (mode << (2 * (j * FUSED_BDY_MAX + k)));
				act |= 1u << j;
			}
		}
	}
	const bool fuse = BDY && aux.fuse_next != 0;           // the state is stored with application [2] in it

	T vmax = T(0), vmax1 = T(0);
	unsigned stale_rows = 0;                                                       // bit i: row y0 + i of this lane keeps state k (quirk Q3 at the second step); tiles are at most 32 rows

	// The march, in two builds of the same statements: LIVE carries the boundary applications (a launch in which the hydrological
	// gate opens at one of its three moments), !LIVE is the kernel as it is without boundaries -- which is what a BDY launch runs on
	// all the iterations in between (the gate opens once per second of model time).
	auto march = [&](auto live_tag, auto look_tag) {
	constexpr bool LIVE = decltype(live_tag)::value;
	constexpr bool LOOK = decltype(look_tag)::value;         // the launch that wrote `src` stamped cells: stage A looks them up (hardly ever)
	// one cell's share of application J, in the order the boundaries were added: the statements of bdy_area
	auto apply = [&](auto j_tag, State4<T> c, const T zb, const long y) {
		constexpr int J = decltype(j_tag)::value;
		if (!LIVE || !((act >> J) & 1u)) return c;                                 // wave-uniform
		const long gy = y + p.row_offset;
		if (!(gy >= 1 && gy <= p.global_rows - 2 && gy < gy_lim)) return c;       // bdy_in_range, row part (wave-uniform)
		const bool cell_ok = bdy_x && !(c.zmax <= T(-9999.0));                    // :168-169, :220-221 (first half)
		#pragma unroll
		for (int k = 0; k < FUSED_BDY_MAX; ++k) {
			const unsigned m = (modes >> (2 * (J * FUSED_BDY_MAX + k))) & 3u;
			if (m == 3u) continue;                                                // wave-uniform
			const T inc = tab[((J * FUSED_BDY_MAX + k) * 2 + ((int)y >= ysw[k] ? 1 : 0)) * 64 + lane];
			const T z2 = (m == 1u) ? fmax_(zb, c.z - inc) : (c.z + inc);          // :179-180 | :176-177, :238-242
			const bool ok = cell_ok && !(m == 2u && c.z == T(-9999.0));           // :220-221 (second half; re-tested as the level changes)
			c.z = ok ? z2 : c.z;
		}
		return c;
	};
	using J0 = std::integral_constant<int, 0>; using J1 = std::integral_constant<int, 1>; using J2 = std::integral_constant<int, 2>;
	const bool live0 = LIVE && (act & 1u) != 0, live1 = LIVE && (act & 2u) != 0, live2 = LIVE && (act & 4u) != 0;

	// ---- stage A: source row r -> U1(r), K1's row step with the result kept in registers ----
	RowRegs<T> rc = load_row(y0 - 1);                                              // the first row stage A produces
	RowRegs<T> rP = load_row(y0), rQ;
	Side<T> sCa;
	FaceFlux<T> fSa = {};
	bool drySa;
	// Still water (STRICT; K1's skip, see there: a row whose 64 cells hold ONE wet state at rest between two rows that hold the same
	// state cell for cell comes out of the update as it went in, bit for bit).  Both stages have it -- stage B on the intermediate rows
	// -- and on such rows a pair is what it is for the FAST flavour: a copy at half the bytes.
	constexpr bool STILL = STRICT;
	constexpr unsigned REST_S = 1, REST_C = 2, EQ_S = 4, PRICED = 16;              // (K1's flags, one word per stage)
	unsigned stA = 0, stB = 0;
	auto at_rest = [&](const State4<T>& c) { return wave_all(c.qx == T(0) && c.qy == T(0)) != 0; };
	auto same_level = [&](const State4<T>& a, const T za, const State4<T>& b, const T zb_) { return wave_all(a.z == b.z && za == zb_) != 0; };
	auto one_wet_state = [&](const State4<T>& c, const T zb_) {
		const T z_first = first_lane(c.z), b_first = first_lane(zb_);
		return wave_all(c.z == z_first && zb_ == b_first && (c.z - zb_) > vs && !(c.zmax <= T(-9999.0) || c.z == T(-9999.0))) != 0;
	};
	{
		RowRegs<T> rs = load_row(y0 - 2);
		if (live0) { rs.c = apply(J0(), rs.c, rs.zb, y0 - 2 < 0 ? 0 : y0 - 2); rc.c = apply(J0(), rc.c, rc.zb, y0 - 1); }
		const Side<T> sS = make_side<STRICT>(rs.c.z, rs.c.qx, rs.c.qy, rs.zb, vs);
		sCa = make_side<STRICT>(rc.c.z, rc.c.qx, rc.c.qy, rc.zb, vs);
		drySa = (rs.c.z - rs.zb) < vs;
		if (STILL) {
			if (at_rest(rs.c)) stA |= REST_S;
			if (at_rest(rc.c)) stA |= REST_C;
			if (stA == (REST_S | REST_C) && same_level(rs.c, rs.zb, rc.c, rc.zb)) stA |= EQ_S;
		}
		if (!skip_a) fSa = face_solve<AXIS_Y, STRICT, true, true>(sS, sCa, vs).forR;
	}
	auto stage_a = [&](const long r, const RowRegs<T>& rn_in, RowRegs<T>& pre) {
		pre = load_row(r + 2, r + 2 <= y1 + 1);                                   // nothing beyond the row north of the tile's halo row
		RowRegs<T> rn = rn_in;
		if (live0) rn.c = apply(J0(), rn.c, rn.zb, r + 1 > last_row ? last_row : r + 1);   // (the row enters the march here: boundaries first, CSchemeGodunov.cpp:1638)
		RowU1<T> u; u.c = rc.c; u.zb = rc.zb; u.plain = true;
		Side<T> sN = sCa;
		const bool ring_row = r <= 0 || r >= last_row;                             // wave-uniform: passes through, but its north face is needed
		bool still = false;
		if (STILL) {
			const bool rest_n = at_rest(rn.c);
			const bool eq_n = (stA & REST_C) && rest_n && same_level(rc.c, rc.zb, rn.c, rn.zb);
			if (!skip_a && !ring_row && eq_n && (stA & (EQ_S | REST_S)) == (EQ_S | REST_S)) still = one_wet_state(rc.c, rc.zb);
			stA = ((stA & REST_C) ? REST_S : 0u) | (rest_n ? REST_C : 0u) | (eq_n ? EQ_S : 0u);
		}
		if (still) {
			// (the side and the south flux the rows hand each other stay as they are: the rows hold one state)
			if (u.c.z > u.c.zmax && u.c.zmax > T(-9990.0)) u.c.zmax = u.c.z;          // :375-376, all that is left of the update
			u.plain = false;                                                       // (never asked: a still cell is wet)
			drySa = false;
		} else if (!skip_a) {
			sN = make_side<STRICT>(rn.c.z, rn.c.qx, rn.c.qy, rn.zb, vs);
			const FacePair<T> fy = face_solve<AXIS_Y, STRICT, true, true>(sCa, sN, vs);
			// (a face whose lanes all have water on both sides lies between wet cells: K1's row step)
			bool dryC = false, dryN = false;
			if (!LAZY_DRY || !fy.wet) {
				asm volatile("");
				dryC = (rc.c.z - rc.zb) < vs;
				dryN = (rn.c.z - rn.zb) < vs;
			}
			if (!ring_row) {
				const Side<T> sE = side_from_east(sCa);
				const FacePair<T> fx = face_solve<AXIS_X, STRICT, true, true>(sCa, sE, vs);
				bool dryW;
				const FaceFlux<T> fW = flux_from_west(fx.forR, dryC, dryW);
				bool dryE = false;
				if (!LAZY_DRY || !fx.wet) {
					asm volatile("");
					dryE = (sE.eta - sE.zb) < vs;
				}
				const bool disabled = rc.c.zmax <= T(-9999.0) || rc.c.z == T(-9999.0); // :214-218
				const bool dry5 = dryC && dryN && dryE && drySa && dryW;               // :248-255 (untouched: see PairAux)
				const State4<T> upd = godunov_update_impl<STRICT, false, true>(rc.c, rc.zb, rc.n, dt_a, fy.forL, fx.forL, fSa, fW, p.dx, inv_dx, vs, with_friction,
				                                                               (T*)nullptr, NoProbe(), ka);
				// (lanes 0 and 63 have no west / east neighbour of their own -- the rotate hands them a cell from the far end of the wavefront --
				// and nothing reads what the first step makes of them except stage B's wave-wide votes: they keep their source state, a real
				// cell of the row, instead of an update from a foreign flux that on a thin film drains them dry and sends every wavefront of
				// the second step down the dry-side paths.  S-RAIN 8192^2 fp32: 640 M -> see profiles/r06u_* VALU per pair)
				const bool touched = !(ring_x || disabled || dry5 || lane == 0 || lane == 63);
				if (touched) { u.c = upd; u.plain = false; }
				// untouched by the reference (Q3): its destination keeps the state of the iteration before the pair.  That is the cell's
				// current state -- unless the launch before said otherwise
				if (LOOK && wave_any(dry5 && !ring_x && !disabled)) {
					if (dry5 && !ring_x && !disabled) {
						const size_t id = (size_t)r * p.cols + xc;
						if (stamp_gen(aux.stamps, id) == aux.prev_gen) {
							const State4<T> kept = stamp_state(aux.stamps, id);
							// (audit: a stale value that is NOT the cell's current state -- what the flavour without stamps would have got wrong)
							if (kept.z != u.c.z || kept.zmax != u.c.zmax || kept.qx != u.c.qx || kept.qy != u.c.qy) atomicAdd(aux.stamps.haz + 2, 1ull);
							u.c = kept; u.plain = false;
						}
					}
				}
			}
			fSa = fy.forR;
			drySa = dryC;
		}
		if (live1) {                                                               // the second iteration's boundaries, on its source state
			const T z_in = u.c.z;
			u.c = apply(J1(), u.c, u.zb, r);
			u.plain = u.plain && u.c.z == z_in;
		}
		rc = rn;
		sCa = sN;
		return u;
	};

	// ---- stage B: U1(y - 1 .. y + 1) -> the final row y ----
	RowU1<T> uc;                                                                   // U1 of the row stage B works on
	T n_c = rc.n;                                                                  // (its Manning n: one row behind stage A's)
	Side<T> sCb;
	FaceFlux<T> fSb = {};
	bool drySb = false;
	auto stage_b = [&](const long y, const RowU1<T>& un, const T n_next, const bool update) {
		State4<T> out = uc.c;
		bool write = out_x;
		bool untouched = false;                                                    // left alone by the second step (Q3): the primary buffer keeps state k
		// (the row's offset, pinned as a scalar)
		const unsigned row_k = (unsigned)__builtin_amdgcn_readfirstlane((int)(y - row_base));
		bool still = false, skip_cfl = false;
		if (STILL) {
			const bool rest_n = at_rest(un.c);
			const bool eq_n = (stB & REST_C) && rest_n && same_level(uc.c, uc.zb, un.c, un.zb);
			if (update && !skip_b && eq_n && (stB & (EQ_S | REST_S)) == (EQ_S | REST_S)) still = one_wet_state(uc.c, uc.zb);
			stB = (stB & PRICED) | ((stB & REST_C) ? REST_S : 0u) | (rest_n ? REST_C : 0u) | (eq_n ? EQ_S : 0u);
		}
		Side<T> sN = sCb;
		if (still) {
			if (out.z > out.zmax && out.zmax > T(-9990.0)) out.zmax = out.z;          // :375-376
			drySb = false;
			// ONE state across the wave: if the still row below (the same state, by its own test) has priced it, pricing it again
			// cannot change the maximum
			skip_cfl = (stB & PRICED) != 0;
			if ((int)y >= tm.price_lo && (int)y < tm.price_hi && wave_any(out_x && out.zmax > T(-9999.0))) stB |= PRICED;
		} else {
			stB &= ~PRICED;
			sN = make_side<STRICT>(un.c.z, un.c.qx, un.c.qy, un.zb, vs);
		}
		if (!still && !skip_b) {
			const FacePair<T> fy = face_solve<AXIS_Y, STRICT, true, true>(sCb, sN, vs);
			bool dryC = false, dryN = false;
			if (!LAZY_DRY || !fy.wet) {
				asm volatile("");
				dryC = (uc.c.z - uc.zb) < vs;
				dryN = (un.c.z - un.zb) < vs;
			}
			if (update) {
				const Side<T> sE = side_from_east(sCb);
				const FacePair<T> fx = face_solve<AXIS_X, STRICT, true, true>(sCb, sE, vs);
				bool dryW;
				const FaceFlux<T> fW = flux_from_west(fx.forR, dryC, dryW);
				bool dryE = false;
				if (!LAZY_DRY || !fx.wet) {
					asm volatile("");
					dryE = (sE.eta - sE.zb) < vs;
				}
				const bool disabled = uc.c.zmax <= T(-9999.0) || uc.c.z == T(-9999.0);
				const bool dry5 = dryC && dryN && dryE && drySb && dryW;
				// Q3 for the launch that follows (PairAux): the first-step value uc.c is written down where that launch could need it and
				// cannot have it -- a cell that may be dry at its first step and whose value there differs from uc.c.
				//  * a cell this step leaves alone holds state k afterwards: equal to uc.c if nothing touched it at the first step either
				//    (`plain`), unknown here otherwise -- written down;
				//  * a cell this step updates: looked at between the flux step and the friction step (the update's probe), where the old
				//    state is in registers anyway and everything that decides is known -- a cell that ends the step dry has no friction
				//    term, so its level (clamped to the bed), its discharge and its maximum level are final there.
				//  (LIVE launches, whose stored state also receives the next iteration's boundaries, do the same after the update, on
				//  the state with those applied: further down.)
				const bool keeps_k = dry5 && out_x && !disabled;                       // this step leaves the cell alone
				const bool probed = out_x && !disabled && !dry5;
				const bool pending = !LIVE && hz_on && keeps_k && !uc.plain;
				const auto probe = make_probe(
					[&](const State4<T>&, const T z1) { return pending || (!LIVE && hz_on && probed && (z1 - uc.zb) < vs); },
					[&](const State4<T>& c0, const T z1, const T qx1, const T qy1) {
						if (LIVE || !hz_on) return;
						const bool ends_dry = (z1 - uc.zb) < vs;                       // (then the level is clamped to the bed: CLSchemeGodunov.clc:379-380)
						const bool changed = c0.z != uc.zb || c0.qx != qx1 || c0.qy != qy1 || (z1 > c0.zmax && c0.zmax > T(-9990.0));
						const bool need = pending || (probed && ends_dry && changed);
						if (wave_any(need)) {
							if (need) {
								const size_t id = (size_t)y * p.cols + xc;
								T* v = reinterpret_cast<T*>(aux.stamps.rec + id * REC);
								v[0] = c0.z; v[1] = c0.zmax; v[2] = c0.qx; v[3] = c0.qy;
								*reinterpret_cast<unsigned*>(aux.stamps.rec + id * REC + sizeof(State4<T>)) = aux.gen;
								aux.stamps.haz[aux.gen & 1u] = (unsigned long long)aux.gen;        // (the launch's word: every writer writes the same value)
							}
						}
					});
				const State4<T> upd = godunov_update_impl<STRICT, false, true>(uc.c, uc.zb, n_c, dt_b, fy.forL, fx.forL, fSb, fW, p.dx, inv_dx, vs, with_friction,
				                                                               (T*)nullptr, probe, kb);
				if (!disabled) {
					if (dry5) {                                                        // the primary buffer keeps state k: copied in the cold pass below
						write = false;
						untouched = out_x;
						if (out_x) stale_rows |= 1u << (unsigned)(y - y0);
					} else {
						out = upd;
					}
				}
			}
			fSb = fy.forR;
			drySb = dryC;
		}
		if (update) {
			// what the NEXT launch will read for this cell: the state with the boundaries of the iteration after the pair (stored like
			// that if another iteration of the batch follows; applied by the next launch as it loads the row otherwise)
			State4<T> next = out;
			if (live2) next = apply(J2(), out, uc.zb, y);
			// (LIVE: Q3's stamps on the state with the next iteration's boundaries applied -- what the next launch will read for the cell,
			// or, when the batch ends here and that launch applies them itself as it loads, possibly without them)
			if (LIVE && hz_on) {
				const bool dry_next = (next.z - uc.zb) < vs || (!fuse && (out.z - uc.zb) < vs);
				bool stamp = write && dry_next && (uc.c.z != next.z || uc.c.qx != next.qx || uc.c.qy != next.qy || uc.c.zmax != next.zmax ||
				                                   (!fuse && uc.c.z != out.z));
				stamp = stamp || (untouched && (!uc.plain || (next.z != out.z && (next.z - uc.zb) < vs)));
				if (wave_any(stamp)) {
					asm volatile("");
					if (stamp) {
						const size_t id = (size_t)y * p.cols + xc;
						T* v = reinterpret_cast<T*>(aux.stamps.rec + id * REC);
						v[0] = uc.c.z; v[1] = uc.c.zmax; v[2] = uc.c.qx; v[3] = uc.c.qy;
						*reinterpret_cast<unsigned*>(aux.stamps.rec + id * REC + sizeof(State4<T>)) = aux.gen;
					}
					stamped = stamped || stamp;
				}
			}
			buf_store_state(fuse ? next : out, srd_dst, write ? voff_state : HP_OOB, row_k * row_state);
			if (TAIL == 2) store_peer(fuse ? next : out, y, write);
			if (CFL_MODE == 1 && write && !skip_cfl && (int)y >= tm.price_lo && (int)y < tm.price_hi) {
				const T s = cfl_speed<STRICT>(out.z, out.zmax, out.qx, out.qy, uc.zb, p.qs);
				if (s > vmax) vmax = s;
				if (live2) {                                                           // ... and what the next iteration's reduction will find (slot[SLOT_M1])
					const T s2 = cfl_speed<STRICT>(next.z, next.zmax, next.qx, next.qy, uc.zb, p.qs);
					if (s2 > vmax1) vmax1 = s2;
				}
			}
		}
		uc = un; n_c = n_next;
		sCb = sN;
	};

	// prologue: U1(y0 - 1) and U1(y0), and the face between them
	{
		const T n0 = rc.n;
		uc = stage_a(y0 - 1, rP, rQ);                                              // (rc = row y0, rP -> rQ holds row y0 + 1)
		n_c = n0;
		sCb = make_side<STRICT>(uc.c.z, uc.c.qx, uc.c.qy, uc.zb, vs);
		if (STILL && at_rest(uc.c)) stB |= REST_C;
		const T n1 = rc.n;
		const RowU1<T> u0 = stage_a(y0, rQ, rP);
		stage_b(y0 - 1, u0, n1, false);                                            // stage B without an update: the face below row y0 and its dry flag
	}
	// steady state: stage A on row r, stage B on row r - 1
	long r = y0 + 1;
	for (; r + 1 <= y1; r += 2) {
		{ const T nn = rc.n; const RowU1<T> u = stage_a(r, rP, rQ); stage_b(r - 1, u, nn, true); }
		{ const T nn = rc.n; const RowU1<T> u = stage_a(r + 1, rQ, rP); stage_b(r, u, nn, true); }
	}
	if (r <= y1) { const T nn = rc.n; const RowU1<T> u = stage_a(r, rP, rQ); stage_b(r - 1, u, nn, true); }

	// cells the second step leaves untouched: the primary buffer keeps state k with the first iteration's boundaries (and the
	// reduction prices it; the next iteration's boundaries act on it in place)
	if (wave_any(stale_rows != 0)) {
		for (long y = y0; y < y1; ++y) {
			if ((stale_rows >> (unsigned)(y - y0)) & 1u) {
				const size_t id = (size_t)y * p.cols + xc;
				const T zb = bed[id];
				State4<T> c = src[id];
				if (live0) c = apply(J0(), c, zb, y);
				State4<T> next = c;
				if (live2) next = apply(J2(), c, zb, y);
				dst[id] = fuse ? next : c;
				if (TAIL == 2) store_peer(fuse ? next : c, y, true);                // (the neighbour's copy of the cell gets the same store)
				if (CFL_MODE == 1 && (int)y >= tm.price_lo && (int)y < tm.price_hi) {
					const T s = cfl_speed<STRICT>(c.z, c.zmax, c.qx, c.qy, zb, p.qs);
					if (s > vmax) vmax = s;
					if (live2) {
						const T s2 = cfl_speed<STRICT>(next.z, next.zmax, next.qx, next.qy, zb, p.qs);
						if (s2 > vmax1) vmax1 = s2;
					}
				}
			}
		}
	}
	if (!live2) vmax1 = vmax;                              // nothing acts on the stored state before the next reduction: one maximum
	};   // march

	auto raise_word = [&]() { if (hz_on && wave_any(stamped) && lane == 0) aux.stamps.haz[aux.gen & 1u] = (unsigned long long)aux.gen; };
	if (BDY && act != 0) { if (hz_any) march(std::true_type(), std::true_type()); else march(std::true_type(), std::false_type()); }
	else                 { if (hz_any) march(std::false_type(), std::true_type()); else march(std::false_type(), std::false_type()); }
	raise_word();

	if (CFL_MODE != 0) {
		if (blockIdx.x == 0 && wave == 0) { const T e = *edge_max; if (e > vmax) vmax = e; if (e > vmax1) vmax1 = e; }
		vmax = wave_max(vmax);
		if (BDY) vmax1 = wave_max(vmax1);
		if (TAIL != 0) { wave_vmax = vmax; wave_vmax1 = vmax1; }
		else if (lane == 0 && vmax > T(0)) atomic_max_nonneg(cfl_slot, vmax);
	}
	}   // tile / strip guard
	if (TAIL != 0) tail_block_done<BDY>(tail, wave_vmax, wave, lane, y0, y1, wave_vmax1);
}

// repair_other_buffer's second half (hp_engine.hip): after the device copy "other := current", the cells the last pair launch stamped get
// the value the reference's other buffer holds there (PairAux) -- so that whatever builds on that buffer next finds what single iterations
// would have left in it.
template <typename T>
__global__ __launch_bounds__(256) void stamps_to_buffer(State4<T>* __restrict__ other, const StampBufs<T> b, const unsigned gen, const size_t cells)
{
	if (b.haz[gen & 1u] != (unsigned long long)gen) return;
	for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < cells; id += (size_t)gridDim.x * blockDim.x)
		if (stamp_gen(b, id) == gen) other[id] = stamp_state(b, id);
}

// hp_pair_stats: cells stamped by launch `gen` (counts[0]) and cells that carry any launch's stamp (counts[1]) -- counted where the records
// are (they are 48 bytes a cell: a copy to the host idles the device for tens of milliseconds on a large grid)
template <typename T>
__global__ __launch_bounds__(256) void stamps_count(const StampBufs<T> b, const unsigned gen, const size_t cells, unsigned long long* counts)
{
	unsigned long long last = 0, any = 0;
	for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < cells; id += (size_t)gridDim.x * blockDim.x) {
		const unsigned g = stamp_gen(b, id);
		last += (g != 0 && g == gen); any += (g != 0);
	}
	for (int off = 32; off > 0; off >>= 1) { last += __shfl_down(last, off); any += __shfl_down(any, off); }
	if ((threadIdx.x & 63) == 0 && (last | any)) { atomicAdd(counts, last); atomicAdd(counts + 1, any); }
}

// Cold start of iteration pairs on a domain with area boundaries (hp_engine.hip: pair_cold_start): the stand-alone boundary pass and the
// stand-alone reduction have just done the first half of iteration k the reference's way -- boundaries in place, the primary buffer
// priced -- and this moves the result to where the pair kernel looks for it.
template <typename T>
__global__ void pair_cold_start_words(T* slot)
{
	slot[SLOT_M1] = atomic_exchange_zero(slot);
	slot[SLOT_BDY] = T(1);
}

// Do the edge rings of the two state buffers hold the same bits?  (hp_engine.hip: rings_really_differ.  No flux kernel writes ring cells,
// so each buffer keeps the ring it was given -- the same in both after a full upload, possibly different after a PARTIAL one, which goes
// to the current buffer only, CSchemeGodunov.cpp:1064-1065 vs queueWritePartial.  A pair carries the primary buffer's ring through both
// of its steps, single iterations read the other buffer's on every second one: pairs need equal rings.)  Raises *differ.
template <typename T>
__global__ __launch_bounds__(256) void rings_compare(const State4<T>* __restrict__ a, const State4<T>* __restrict__ b, const long cols, const long rows,
                                                     unsigned long long* differ)
{
	const long n = 2 * cols + 2 * (rows - 2);
	bool bad = false;
	for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
		long x, y;
		if (i < cols) { x = i; y = 0; }
		else if (i < 2 * cols) { x = i - cols; y = rows - 1; }
		else { const long j = i - 2 * cols; x = (j & 1) ? cols - 1 : 0; y = 1 + (j >> 1); }
		const size_t id = (size_t)y * cols + x;
		// bit patterns, not values: -0.0 / +0.0 and NaNs would otherwise pass or fail for the wrong reason
		const unsigned* wa = reinterpret_cast<const unsigned*>(a + id);
		const unsigned* wb = reinterpret_cast<const unsigned*>(b + id);
		#pragma unroll
		for (unsigned w = 0; w < sizeof(State4<T>) / 4; ++w) bad = bad || wa[w] != wb[w];
	}
	if (bad) *differ = 1ull;
}

// -------------------------------------------------------------------------------------------------
// K2  muscl_march : MUSCL-Hancock (MINMOD) predictor + HLLC corrector in ONE pass, double buffered.
//
//  Replaces mch_1st_cacheNone/cachePrediction + mch_2nd_cacheNone / mch_cacheMaximum
//  (CLSchemeMUSCLHancock.clc:28-296, :533-1114) and their four 32 B/cell face buffers: the predictor's face states
//  live only in registers.  Same wavefront-marching scheme as K1 with a 2-cell halo: lanes 0,1,62,63 are halo
//  lanes (60 updated columns per wave), the predictor runs one row ahead of the corrector, and each segment starts
//  with two extra predictor rows.  Each corrector face is solved once and finished for both adjacent cells.
//  Unlike the reference's in-place corrector (whose result depends on work-item order, quirk Q6) the update reads
//  `src` and writes `dst`; cells the reference leaves untouched are copied.
// -------------------------------------------------------------------------------------------------
constexpr int MUSCL_COLS = 60;

template <typename T>
__device__ __forceinline__ Raw<T> raw_of(const RowRegs<T>& r) { return Raw<T>{r.c.z, r.c.zmax, r.c.qx, r.c.qy, r.zb}; }

template <typename T>
__device__ __forceinline__ Raw<T> raw_from_east(const Raw<T>& r)
{
	return Raw<T>{from_east(r.z), from_east(r.zmax), from_east(r.qx), from_east(r.qy), from_east(r.zb)};
}
template <typename T>
__device__ __forceinline__ Raw<T> raw_from_west(const Raw<T>& r)
{
	return Raw<T>{from_west(r.z), from_west(r.zmax), from_west(r.qx), from_west(r.qy), from_west(r.zb)};
}

// waves per SIMD the register allocator is asked to make room for: the FAST fp64 flavour fits three once its waiting
// values sit in LDS (168 VGPRs, no scratch with a uniform Manning n); STRICT fp64 (IEEE division / sqrt expansions)
// would spill there and keeps two; fp32 has room for four
template <bool STRICT, typename T> constexpr int muscl_waves() { return sizeof(T) == 4 ? 4 : (STRICT ? 2 : 3); }

template <bool STRICT, int CFL_MODE, bool UNIFORM_N, int TAIL, typename T, bool SPEC = false>      // SPEC: see K1
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(muscl_waves<STRICT, T>()))) void muscl_march(const Params<T> p, const Scalars<T>* sc,
                                                   const T* __restrict__ bed, const State4<T>* __restrict__ src,
                                                   State4<T>* __restrict__ dst, const T* __restrict__ manning,
                                                   T* cfl_slot, const T* __restrict__ edge_max,
                                                   const TileMap tm, const LaunchTail<T> tail)
{
	if (TAIL != 0 && blockIdx.x >= tail.flux_blocks) {                                  // the launch's own tail block (LaunchTail, K4)
		launch_tail(p, tail);
		return;
	}
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: see K1
	long strip, y0, y1;                                                            // corrector domain 2..n-3 (:569-573)
	T wave_vmax = T(0);
	if (tile_rows(tm, wave, strip, y0, y1)) {

	// Register diet for a third wave per SIMD: what a row does not need while its two faces are being solved -- the
	// predicted N/E/W face values of a row (12 values) -- waits in LDS ([value][lane]: conflict-free 8-byte columns, 24 KiB
	// per block in fp64) from the moment the predictor has produced them until the row's own faces are solved.
	__shared__ T lds_stash[4][18][64];
	T (*const stash)[64] = lds_stash[wave];

	const long x = strip * MUSCL_COLS + lane;
	const long xc = (x < p.cols) ? x : (p.cols - 1);
	const bool out_x = lane >= 2 && lane <= MUSCL_COLS + 1 && x <= p.cols - 3;

	const T dt = sc->dt, vs = p.vs;
	constexpr bool PL = !SPEC;
	T* const spec_bad = cfl_slot + SLOT_SPEC;
	const bool skip_step = dt <= T(0);                                             // :576-577, :69-70
	const bool with_friction = p.friction != 0;
	T vmax = T(0);

	// buffer-resource addressing as in K1: window base = row y0-2, column of lane 0
	const size_t cell0 = (size_t)(y0 - 2) * p.cols + (size_t)(strip * MUSCL_COLS);
	const size_t cells_left = (size_t)p.cols * p.rows - cell0;
	const __amdgpu_buffer_rsrc_t srd_src = make_srd(src + cell0, cells_left * sizeof(State4<T>));
	const __amdgpu_buffer_rsrc_t srd_dst = make_srd(dst + cell0, cells_left * sizeof(State4<T>));
	const __amdgpu_buffer_rsrc_t srd_bed = make_srd(bed + cell0, cells_left * sizeof(T));
	const __amdgpu_buffer_rsrc_t srd_man = make_srd(manning + cell0, cells_left * sizeof(T));
	const unsigned lane_col = (unsigned)(xc - strip * MUSCL_COLS);
	const unsigned voff_state = lane_col * (unsigned)sizeof(State4<T>), voff_scalar = lane_col * (unsigned)sizeof(T);
	const unsigned row_state = (unsigned)p.cols * (unsigned)sizeof(State4<T>), row_scalar = (unsigned)p.cols * (unsigned)sizeof(T);
	// (TAIL == 2) the rows of the edge ranges are stored into the strip neighbours as well, written through (see K1)
	const __amdgpu_buffer_rsrc_t srd_peer0 = make_srd((TAIL == 2 && tail.peer_rows[0] ? tail.peer_rows[0] : dst) + cell0, cells_left * sizeof(State4<T>));
	const __amdgpu_buffer_rsrc_t srd_peer1 = make_srd((TAIL == 2 && tail.peer_rows[1] ? tail.peer_rows[1] : dst) + cell0, cells_left * sizeof(State4<T>));
	auto store_peer = [&](const State4<T>& v, const long y, const bool write_lane) {
		const bool e0 = (int)y >= tail.edge_rows[0] && (int)y < tail.edge_rows[1];   // wave-uniform
		const bool e1 = (int)y >= tail.edge_rows[2] && (int)y < tail.edge_rows[3];
		buf_store_state<HP_AUX_THROUGH>(v, e1 ? srd_peer1 : srd_peer0, (write_lane && (e0 || e1)) ? voff_state : HP_OOB, (unsigned)(y - (y0 - 2)) * row_state);
	};

	auto load_row = [&](const long y, const bool live = true) {                 // `live`: see K1
		RowRegs<T> r;
		const unsigned k = (unsigned)(y - (y0 - 2));
		unsigned vs_ = live ? voff_state : HP_OOB, vc_ = live ? voff_scalar : HP_OOB;
		asm volatile("" : "+v"(vs_), "+v"(vc_));
		r.c = buf_load_state(srd_src, vs_, k * row_state, T());
		r.zb = buf_load_scalar(srd_bed, vc_, k * row_scalar, T());
		if (UNIFORM_N) r.n = p.manning_value;               // one value everywhere (found at upload): stays a scalar
		else           r.n = buf_load_scalar(srd_man, vc_, k * row_scalar, T());
		return r;
	};
	auto predict = [&](const RowRegs<T>& south, const RowRegs<T>& mid, const RowRegs<T>& north, bool& dry_e, bool& dry_w,
	                   bool& quiet, bool& same) {
		const Raw<T> c = raw_of(mid);
		const Raw<T> e = raw_from_east(c), w = raw_from_west(c);
		dry_e = e.zmax < vs;                                                       // :633 tests Zmax, not depth (Q6)
		dry_w = w.zmax < vs;
		return muscl_predict_impl<STRICT, PL>(c, raw_of(north), e, raw_of(south), w, dt, p.dx, STRICT ? p.inv_dx_pow2 : p.inv_dx, vs, p.muscl_nb_bed != 0, quiet, same, spec_bad);
	};
	// A "quiet" row (muscl_predict's wave-uniform fast path: all four face states of every lane equal the cell state)
	// needs neither the LDS round trip of its face values nor three separate sides: one side built from the cell state
	// serves its E, W and N faces -- the same values the general path would produce, with a third of the reciprocals.
	auto cell_side = [&](const RowRegs<T>& r) {
		Face4<T> cc; cc.z = r.c.z; cc.h = r.c.z - r.zb; cc.qx = r.c.qx; cc.qy = r.c.qy;
		return side_from_face_impl<STRICT, PL>(cc, r.c.qx, r.c.qy, vs, spec_bad);
	};

	// one side of a north / south face from a predicted face state; FAST predicts no faces for a quiet row (they are the cell state)
	auto face_side = [&](const Face4<T>& fc, const RowRegs<T>& r, const bool quiet) {
		if (!STRICT && quiet) return cell_side(r);
		return side_from_face_impl<STRICT, PL>(fc, r.c.qx, r.c.qy, vs, spec_bad);
	};

	RowRegs<T> rA = load_row(y0);
	RowRegs<T> rB = load_row(y0 + 1);                                              // y0+1 <= rows-2
	RowRegs<T> rP = load_row((y0 + 2 < p.rows) ? (y0 + 2) : (p.rows - 1)), rQ;    // landing sets of the row two ahead

	FaceFlux<T> fS = {};
	bool dryE = false, dryW = false, dryS;
	bool quiet_c = false;                                                          // the row being corrected is a quiet row
	bool quiet_s = false, same_c = false;                                          // ... the row south of it; every lane's neighbourhood is one state
	bool fS_ok = true;                                                             // fS holds the south face of the row being corrected
	bool still_priced = false;                 // the row below was a still row whose (one) wave speed is already in vmax
	{
		const RowRegs<T>& rc = rA; const RowRegs<T>& rn = rB;
		const RowRegs<T> rs2 = load_row(y0 - 2);
		const RowRegs<T> rs = load_row(y0 - 1);
		dryS = rs.c.zmax < vs;
		if (!skip_step) {
			bool de, dw, ss;
			const Faces<T> ps = predict(rs2, rs, rc, de, dw, quiet_s, ss);
			const Faces<T> pc = predict(rs, rc, rn, dryE, dryW, quiet_c, same_c);
			if (!quiet_c) {
			stash[0][lane] = pc.n.z; stash[1][lane] = pc.n.h; stash[2][lane] = pc.n.qx; stash[3][lane] = pc.n.qy;
			stash[4][lane] = pc.e.z; stash[5][lane] = pc.e.h; stash[6][lane] = pc.e.qx; stash[7][lane] = pc.e.qy;
			stash[8][lane] = pc.w.z; stash[9][lane] = pc.w.h; stash[10][lane] = pc.w.qx; stash[11][lane] = pc.w.qy;
			}
			const Side<T> sS = face_side(ps.n, rs, quiet_s);
			const Side<T> sC = face_side(pc.s, rc, quiet_c);
			fS = face_solve_impl<AXIS_Y, STRICT, true, true, PL>(sS, sC, vs, spec_bad).forR;
		}
	}

	// one row: `rc` is corrected, `rn` is its northern neighbour, `rnn` the row after (landed last iteration),
	// `pre` receives the prefetch of row y+3; the row in flight is never copied (see K1).  On return `rn` is the row to
	// correct next and `rc`'s registers hold ITS northern neighbour (back from LDS): the caller swaps the two names.
	auto row_step = [&](const long y, RowRegs<T>& rc, RowRegs<T>& rn, const RowRegs<T>& rnn, RowRegs<T>& pre) {
		// prefetch; the predictor of row y1 is the last consumer and reads up to row y1 + 1 (<= rows - 1)
		pre = load_row((y + 3 <= y1 + 1) ? (y + 3) : (y1 + 1), y + 3 <= y1 + 1);
		State4<T> out = rc.c;
		const T zb_c = rc.zb;
		bool skip_cfl = false;

		if (!skip_step) {
			// predictor of the next row (needs rows y, y+1, y+2)
			bool dryE_n, dryW_n, quiet_n, same_n;
			const Faces<T> pn = predict(rc, rn, rnn, dryE_n, dryW_n, quiet_n, same_n);
			const Face4<T> pn_s = pn.s;

			const bool disabled = rc.c.zmax <= T(-9999.0) || rc.c.z == T(-9999.0);     // :594-595
			const bool dryC = (rc.c.z - rc.zb) < vs;                                  // :597-598
			const bool dryN = rn.c.zmax < vs;
			const bool dry5 = dryC && dryN && dryE && dryS && dryW;                   // :638

			// Inert rows (wave-uniform).  Two kinds of row provably come out of the corrector exactly as they went in:
			//  dry land -- every lane's cell and its four neighbours are dry (dry5): the reference does not update such
			//    cells at all (:638), so no face of the row is needed for the row itself;
			//  still water -- every lane's cell and its four neighbours hold ONE wet state with zero discharge, and the
			//    rows north and south are quiet (their face states are their cell states): the four faces of a cell are
			//    then solved on identical left/right states, opposite faces return identical values, every flux
			//    difference is x - x = +0, the bed-slope term is (zb - zb) = +0 times something finite, friction does
			//    not act on zero discharge, and Z - dt*(+0) = Z, Q - dt*(+0) = Q bit for bit (FAST and STRICT alike;
			//    only Zmax may still have to follow Z).
			// Such a row skips both face solves and the update.  The one thing another row may want from it is its north
			// face (the next row's south flux): a dry row solves it unless the next row is dry land as well (known
			// here: that row's five dry tests need rows y..y+2, all in registers); a still row never does -- if the
			// next row turns out not to be inert, it re-derives that face from its own cell state, which the still
			// row's `same` test has shown to be the still row's state too (same values in, same bits out).
			// (the votes are taken only where they can matter: live rows pay for `dry5` alone)
			const bool inertD = wave_all(dry5);
			bool inertS = false, inertD_n = false;
			if (same_c && quiet_n && quiet_s) {
				const bool wet_q0 = (rc.c.z - rc.zb) > vs && rc.c.qx == T(0) && rc.c.qy == T(0);
				inertS = wave_all(wet_q0);
			}
			if (inertD) {
				const bool dry5_n = ((rn.c.z - rn.zb) < vs) && (rnn.c.zmax < vs) && dryE_n && (rc.c.zmax < vs) && dryW_n;
				inertD_n = wave_all(dry5_n);
			}

			const bool inert = inertS || inertD;

			// the current row's faces come back from LDS only now that the predictor's temporaries are dead, and the
			// next row's take their place there
			Side<T> sE_mine, sW_mine, sN_mine;
			if (!inert) {
				if (quiet_c) {
					sE_mine = cell_side(rc);
					sW_mine = sE_mine; sN_mine = sE_mine;
				} else {
					Face4<T> pc_n, pc_e, pc_w;
					pc_n.z = stash[0][lane]; pc_n.h = stash[1][lane]; pc_n.qx = stash[2][lane]; pc_n.qy = stash[3][lane];
					pc_e.z = stash[4][lane]; pc_e.h = stash[5][lane]; pc_e.qx = stash[6][lane]; pc_e.qy = stash[7][lane];
					pc_w.z = stash[8][lane]; pc_w.h = stash[9][lane]; pc_w.qx = stash[10][lane]; pc_w.qy = stash[11][lane];
					sE_mine = side_from_face_impl<STRICT, PL>(pc_e, rc.c.qx, rc.c.qy, vs, spec_bad);
					sW_mine = side_from_face_impl<STRICT, PL>(pc_w, rc.c.qx, rc.c.qy, vs, spec_bad);
					sN_mine = side_from_face_impl<STRICT, PL>(pc_n, rc.c.qx, rc.c.qy, vs, spec_bad);
				}
			}
			if (!quiet_n) {
				stash[0][lane] = pn.n.z; stash[1][lane] = pn.n.h; stash[2][lane] = pn.n.qx; stash[3][lane] = pn.n.qy;
				stash[4][lane] = pn.e.z; stash[5][lane] = pn.e.h; stash[6][lane] = pn.e.qx; stash[7][lane] = pn.e.qy;
				stash[8][lane] = pn.w.z; stash[9][lane] = pn.w.h; stash[10][lane] = pn.w.qx; stash[11][lane] = pn.w.qy;
			}
			// the row two ahead has served the predictor; it is next needed as the northern row of the next iteration
			stash[12][lane] = rnn.c.z; stash[13][lane] = rnn.c.zmax; stash[14][lane] = rnn.c.qx; stash[15][lane] = rnn.c.qy;
			stash[16][lane] = rnn.zb;
			if (!UNIFORM_N) stash[17][lane] = rnn.n;

			if (inert) {
				if (inertD && !inertD_n) {
					const Side<T> sN_dry = cell_side(rc);                                 // a dry row is a quiet row
					const Side<T> sN_nb = face_side(pn_s, rn, quiet_n);
					fS = face_solve_impl<AXIS_Y, STRICT, true, true, PL>(sN_dry, sN_nb, vs, spec_bad).forR;
					fS_ok = true;
				} else {
					fS_ok = false;
				}
				if (inertS && out.z > out.zmax && out.zmax > T(-9990.0)) out.zmax = out.z;   // :791-796, all that is left of the update
				// CFL epilogue: dry cells price at zero; a still row has ONE state across the wave, and if the still row below
				// (the same state, by its neighbourhood test) has already put that wave speed into vmax, pricing it again
				// cannot change the maximum
				skip_cfl = inertD || still_priced;
				still_priced = inertS && (still_priced || ((int)y >= tm.price_lo && (int)y < tm.price_hi && wave_any(out_x && out.zmax > T(-9999.0))));
			} else {
			still_priced = false;
			if (!fS_ok) {                                   // the row below was a still row: its state is this row's state
				const Side<T> cs = cell_side(rc);
				fS = face_solve_impl<AXIS_Y, STRICT, true, true, PL>(cs, cs, vs, spec_bad).forR;
			}
			// east face: my E-face state against the east neighbour's W-face state
			const Side<T> sE_nb = side_from_east(sW_mine);
			const FacePair<T> fx = face_solve_impl<AXIS_X, STRICT, true, true, PL>(sE_mine, sE_nb, vs, spec_bad);
			const FaceFlux<T> fE = fx.forL, forW = fx.forR;
			FaceFlux<T> fW;
			fW.f0 = from_west(forW.f0); fW.fx = from_west(forW.fx);
			fW.fy = from_west(forW.fy); fW.eta_nb = from_west(forW.eta_nb);
			fW.zb_nb = from_west(forW.zb_nb);
			fW.stop = from_west((int)forW.stop) != 0;

			// north face: my N-face state against the north neighbour's S-face state
			const Side<T> sN_nb = face_side(pn_s, rn, quiet_n);
			const FacePair<T> fy = face_solve_impl<AXIS_Y, STRICT, true, true, PL>(sN_mine, sN_nb, vs, spec_bad);
			const FaceFlux<T> fN = fy.forL;

			const State4<T> upd = godunov_update_impl<STRICT, true, PL>(rc.c, rc.zb, rc.n, dt, fN, fE, fS, fW, p.dx, STRICT ? p.inv_dx_pow2 : p.inv_dx,
			                                                             vs, with_friction, spec_bad);
			if (!disabled && !dry5) out = upd;

			fS = fy.forR;
			fS_ok = true;
			}

			dryS = rc.c.zmax < vs;
			dryE = dryE_n; dryW = dryW_n;
			quiet_s = quiet_c; quiet_c = quiet_n; same_c = same_n;
		}

		buf_store_state(out, srd_dst, out_x ? voff_state : HP_OOB, (unsigned)(y - (y0 - 2)) * row_state);
		if (TAIL == 2) store_peer(out, y, out_x);
		if (CFL_MODE == 1 && !skip_cfl && out_x && (int)y >= tm.price_lo && (int)y < tm.price_hi) {
			const T s = cfl_speed_impl<STRICT, PL>(out.z, out.zmax, out.qx, out.qy, zb_c, p.qs, false, spec_bad);
			if (s > vmax) vmax = s;
		}
		if (!skip_step) {            // row y+2 comes back into the registers of the row that is finished
			rc.c.z = stash[12][lane]; rc.c.zmax = stash[13][lane]; rc.c.qx = stash[14][lane]; rc.c.qy = stash[15][lane];
			rc.zb = stash[16][lane];
			rc.n = UNIFORM_N ? p.manning_value : stash[17][lane];
		} else {
			rc = rnn;
		}
	};

	long y = y0;
	for (; y + 2 <= y1; y += 2) {
		row_step(y, rA, rB, rP, rQ);
		row_step(y + 1, rB, rA, rQ, rP);
	}
	if (y < y1) row_step(y, rA, rB, rP, rQ);

	if (CFL_MODE != 0) {
		if (blockIdx.x == 0 && wave == 0) { const T e = *edge_max; if (e > vmax) vmax = e; }
		vmax = wave_max(vmax);
		if (TAIL != 0) wave_vmax = vmax;
		else if (lane == 0 && vmax > T(0)) atomic_max_nonneg(cfl_slot, vmax);
	}
	}   // tile / strip guard
	if (TAIL != 0) tail_block_done(tail, wave_vmax, wave, lane, y0, y1);
}

// -------------------------------------------------------------------------------------------------
// K6  inertial_march : partial-inertial scheme (ine_cacheDisabled / ine_cacheEnabled, CLSchemeInertial.clc:26-326).
//
//  State = {Z, Zmax, discharge across the WEST face, discharge across the SOUTH face}.  Same wavefront march as K1
//  (lane = column, 62 updated columns per wave, rows streamed through a two-deep register pipeline).  The reference
//  evaluates calculateInertialFlux four times per cell with the CELL's Manning n, so a face gets two values when n
//  varies; with a uniform n (wave-uniform flag, found at upload) the north discharge is carried to the next row as
//  its south discharge and the east discharge goes to the east lane as its west discharge: two evaluations per
//  cell, bit-identical to four.  About 60 VALU instructions per evaluation: the kernel is HBM bound.
//  dt <= 0 returns WITHOUT writing dst (:61-62, unlike the Godunov kernel), all-dry cells likewise (:103, Q3):
//  both leave dst stale, and the fused CFL epilogue prices what dst really holds.
// -------------------------------------------------------------------------------------------------
template <bool STRICT, int CFL_MODE, int TAIL, typename T>
__global__ __launch_bounds__(256) void inertial_march(const Params<T> p, const Scalars<T>* sc,
                                                      const T* __restrict__ bed, const State4<T>* __restrict__ src,
                                                      State4<T>* __restrict__ dst, const T* __restrict__ manning,
                                                      T* cfl_slot, const T* __restrict__ edge_max,
                                                      const TileMap tm, const LaunchTail<T> tail)
{
	if (TAIL != 0 && blockIdx.x >= tail.flux_blocks) {                                  // the launch's own tail block (LaunchTail, K4)
		launch_tail(p, tail);
		return;
	}
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: see K1
	long strip, y0, y1;
	T wave_vmax = T(0);
	if (tile_rows(tm, wave, strip, y0, y1)) {

	const long x = strip * MARCH_COLS + lane;
	const long xc = (x < p.cols) ? x : (p.cols - 1);
	const bool out_x = lane >= 1 && lane <= MARCH_COLS && x <= p.cols - 2;

	const T dt = sc->dt, vs = p.vs;
	const bool skip_step = dt <= T(0);                                             // :61-62
	const bool uniform_n = p.manning_uniform != 0;
	const bool simplified = p.simplified_cfl != 0;
	T vmax = T(0);
	unsigned long long stale_rows = 0;

	const size_t cell0 = (size_t)(y0 - 1) * p.cols + (size_t)(strip * MARCH_COLS);
	const size_t cells_left = (size_t)p.cols * p.rows - cell0;
	const __amdgpu_buffer_rsrc_t srd_src = make_srd(src + cell0, cells_left * sizeof(State4<T>));
	const __amdgpu_buffer_rsrc_t srd_dst = make_srd(dst + cell0, cells_left * sizeof(State4<T>));
	const __amdgpu_buffer_rsrc_t srd_bed = make_srd(bed + cell0, cells_left * sizeof(T));
	const __amdgpu_buffer_rsrc_t srd_man = make_srd(manning + cell0, cells_left * sizeof(T));
	const unsigned lane_col = (unsigned)(xc - strip * MARCH_COLS);
	const unsigned voff_state = lane_col * (unsigned)sizeof(State4<T>), voff_scalar = lane_col * (unsigned)sizeof(T);
	const unsigned row_state = (unsigned)p.cols * (unsigned)sizeof(State4<T>), row_scalar = (unsigned)p.cols * (unsigned)sizeof(T);
	// (TAIL == 2) the rows of the edge ranges are stored into the strip neighbours as well, written through (see K1)
	const __amdgpu_buffer_rsrc_t srd_peer0 = make_srd((TAIL == 2 && tail.peer_rows[0] ? tail.peer_rows[0] : dst) + cell0, cells_left * sizeof(State4<T>));
	const __amdgpu_buffer_rsrc_t srd_peer1 = make_srd((TAIL == 2 && tail.peer_rows[1] ? tail.peer_rows[1] : dst) + cell0, cells_left * sizeof(State4<T>));
	auto store_peer = [&](const State4<T>& v, const long y, const bool write_lane) {
		const bool e0 = (int)y >= tail.edge_rows[0] && (int)y < tail.edge_rows[1];   // wave-uniform
		const bool e1 = (int)y >= tail.edge_rows[2] && (int)y < tail.edge_rows[3];
		buf_store_state<HP_AUX_THROUGH>(v, e1 ? srd_peer1 : srd_peer0, (write_lane && (e0 || e1)) ? voff_state : HP_OOB, (unsigned)(y - (y0 - 1)) * row_state);
	};

	auto load_row = [&](const long y, const bool live = true) {                 // `live`: see K1
		RowRegs<T> r;
		const unsigned k = (unsigned)(y - (y0 - 1));
		unsigned vs_ = live ? voff_state : HP_OOB, vc_ = live ? voff_scalar : HP_OOB;
		asm volatile("" : "+v"(vs_), "+v"(vc_));
		r.c = buf_load_state(srd_src, vs_, k * row_state, T());
		r.zb = buf_load_scalar(srd_bed, vc_, k * row_scalar, T());
		r.n = uniform_n ? p.manning_value : buf_load_scalar(srd_man, vc_, k * row_scalar, T());
		return r;
	};
	auto flux = [&](const T n, const T q_prev, const T z_up, const T b_up, const T z_down, const T b_down) {
		return inertial_flux<STRICT>(n, dt, q_prev, z_up, b_up, z_down, b_down, p.dx, p.inv_dx, vs);
	};

	RowRegs<T> rc = load_row(y0);
	RowRegs<T> rP = load_row(y0 + 1), rQ;
	T zS, bS, qS_carry;                           // southern neighbour's level and bed; with uniform n: its north discharge
	{
		const RowRegs<T> rs = load_row(y0 - 1);
		zS = rs.c.z; bS = rs.zb;
		qS_carry = flux(rc.n, rc.c.qy, rc.c.z, rc.zb, zS, bS);                     // the first row's south face (:124-132)
	}

	auto row_step = [&](const long y, const RowRegs<T>& rn, RowRegs<T>& pre) {
		pre = load_row((y + 2 <= y1) ? (y + 2) : y1, y + 2 <= y1);
		State4<T> out = rc.c;
		bool write = out_x;

		const T zE = from_east(rc.c.z), bE = from_east(rc.zb), qxE = from_east(rc.c.qx);
		const T zW = from_west(rc.c.z), bW = from_west(rc.zb);
		const T qN = flux(rc.n, rn.c.qy, rn.c.z, rn.zb, rc.c.z, rc.zb);            // :104-112
		const T qE = flux(rc.n, qxE, zE, bE, rc.c.z, rc.zb);                       // :114-122
		T qS, qW;
		if (uniform_n) {                                                           // wave-uniform
			qS = qS_carry;
			qW = from_west(qE);
		} else {
			qS = flux(rc.n, rc.c.qy, rc.c.z, rc.zb, zS, bS);                       // :124-132
			qW = flux(rc.n, rc.c.qx, rc.c.z, rc.zb, zW, bW);                       // :134-142
		}
		const bool disabled = rc.c.zmax <= T(-9999.0) || rc.c.z == T(-9999.0);     // :70-74
		const bool dry5 = (rc.c.z - rc.zb) < vs && (rn.c.z - rn.zb) < vs && (zE - bE) < vs && (zS - bS) < vs &&
		                  (zW - bW) < vs;                                          // :93-103
		const State4<T> upd = inertial_update<STRICT>(rc.c, rc.zb, dt, qN, qE, qS, qW, p.dx, p.inv_dx, vs);
		if (skip_step || (!disabled && dry5)) {                                    // dst keeps what it holds
			write = false;
			if (out_x) stale_rows |= 1ull << (unsigned)(y - y0);
		} else if (!disabled) {
			out = upd;
		}
		buf_store_state(out, srd_dst, write ? voff_state : HP_OOB, (unsigned)(y - (y0 - 1)) * row_state);
		if (TAIL == 2) store_peer(out, y, write);
		if (!((int)y >= tm.price_lo && (int)y < tm.price_hi)) {
		} else if (CFL_MODE == 1) {
			if (write) {
				const T s = cfl_speed<STRICT>(out.z, out.zmax, out.qx, out.qy, rc.zb, p.qs, simplified);
				if (s > vmax) vmax = s;
			}
		} else if (CFL_MODE == 2) {
			if (out_x) {
				const T s = cfl_speed<STRICT>(rc.c.z, rc.c.zmax, rc.c.qx, rc.c.qy, rc.zb, p.qs, simplified);
				if (s > vmax) vmax = s;
			}
		}
		zS = rc.c.z; bS = rc.zb; qS_carry = qN;
		rc = rn;
	};

	long y = y0;
	for (; y + 2 <= y1; y += 2) {
		row_step(y, rP, rQ);
		row_step(y + 1, rQ, rP);
	}
	if (y < y1) row_step(y, rP, rQ);

	if (CFL_MODE == 1 && wave_any(stale_rows != 0)) {
		for (long yy = y0; yy < y1; ++yy) {
			if (((stale_rows >> (unsigned)(yy - y0)) & 1ull) && (int)yy >= tm.price_lo && (int)yy < tm.price_hi) {
				const size_t id = (size_t)yy * p.cols + xc;
				const State4<T> c = dst[id];
				const T s = cfl_speed<STRICT>(c.z, c.zmax, c.qx, c.qy, bed[id], p.qs, simplified);
				if (s > vmax) vmax = s;
			}
		}
	}

	if (CFL_MODE != 0) {
		if (blockIdx.x == 0 && wave == 0) { const T e = *edge_max; if (e > vmax) vmax = e; }
		vmax = wave_max(vmax);
		if (TAIL != 0) wave_vmax = vmax;
		else if (lane == 0 && vmax > T(0)) atomic_max_nonneg(cfl_slot, vmax);
	}
	}   // tile / strip guard
	if (TAIL != 0) tail_block_done(tail, wave_vmax, wave, lane, y0, y1);
}

// max wave speed over the edge ring (cells no kernel ever writes): the `w` outermost columns on rows
// [row_lo,row_hi) plus the `w` outermost rows at the global south / north end when this strip holds them
// (south / north = first such local row, or -1).  w = 1 (Godunov) or 2 (MUSCL-Hancock).  Priced once per upload.
template <bool STRICT, typename T>
__global__ __launch_bounds__(256) void cfl_edge_ring(const Params<T> p, const State4<T>* __restrict__ state,
                                                     const T* __restrict__ bed, const long row_lo, const long row_hi,
                                                     const long south, const long north, const int w,
                                                     T* __restrict__ edge_max)
{
	const long n_side = (row_hi - row_lo) * w;
	const long n_row = (long)w * p.cols;
	const long total = 2 * n_side + (south >= 0 ? n_row : 0) + (north >= 0 ? n_row : 0);
	T m = T(0);
	for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
		long x, y;
		if (i < n_side) { x = i % w; y = row_lo + i / w; }
		else if (i < 2 * n_side) { const long j = i - n_side; x = p.cols - 1 - (j % w); y = row_lo + j / w; }
		else {
			long j = i - 2 * n_side;
			if (south >= 0 && j < n_row) { x = j % p.cols; y = south + j / p.cols; }
			else { if (south >= 0) j -= n_row; x = j % p.cols; y = north + j / p.cols; }
		}
		const size_t id = (size_t)y * p.cols + x;
		const State4<T> c = state[id];
		const T s = cfl_speed<STRICT>(c.z, c.zmax, c.qx, c.qy, bed[id], p.qs, p.simplified_cfl != 0);
		if (s > m) m = s;
	}
	m = wave_max(m);
	if ((threadIdx.x & 63) == 0 && m > T(0)) atomic_max_nonneg(edge_max, m);
}

// -------------------------------------------------------------------------------------------------
// K3  cfl_reduce : max wave speed over rows [row_lo, row_hi) of one state buffer.
//     tst_Reduce (CLDynamicTimestep.clc:166-249) + the serial max of tst_Advance_Normal (:75-80) as
//     wavefront shuffle -> LDS across the block's waves -> one exact atomic max per block.
// -------------------------------------------------------------------------------------------------
//     STRICT = the domain's arithmetic: a buffer must price to the same bits whichever kernel prices it -- this one after an
//     upload, the flux kernels' fused epilogue afterwards (FAST's reciprocal / square-root forms differ from the IEEE ones in
//     the last bit; a strip that re-uses its remembered maximum next to a single domain that re-prices showed it, round 3).
template <bool STRICT, typename T>
__global__ __launch_bounds__(256) void cfl_reduce(const Params<T> p, const State4<T>* __restrict__ state,
                                                  const T* __restrict__ bed, const long row_lo, const long row_hi,
                                                  T* __restrict__ slot)
{
	__shared__ T wave_part[4];
	const size_t first = (size_t)row_lo * p.cols, last = (size_t)row_hi * p.cols;
	T m = T(0);
	for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < last;
	     i += (size_t)gridDim.x * blockDim.x) {
		const State4<T> c = state[i];
		const T s = cfl_speed<STRICT>(c.z, c.zmax, c.qx, c.qy, bed[i], p.qs, p.simplified_cfl != 0);
		if (s > m) m = s;
	}
	m = wave_max(m);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (lane == 0) wave_part[wave] = m;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < (int)(blockDim.x >> 6); ++w) if (wave_part[w] > m) m = wave_part[w];
		if (m > T(0)) atomic_max_nonneg(slot, m);
	}
}

// -------------------------------------------------------------------------------------------------
// K5  boundary source terms applied in place to the source state before the flux kernel
//     bdy_Uniform (Boundaries/CLBoundaries.clc:130-184), bdy_Gridded (:186-246).
//     Rows are addressed globally so ghost rows of a strip receive the same rain as their owner gives them.
// -------------------------------------------------------------------------------------------------
// Both kernels run on a small grid-stride grid: the hydrological gate (t_hydro >= 1 s, a device scalar the host
// cannot see without a sync) is closed on most iterations, and a launch whose every wave exits after reading three
// scalars must cost microseconds, not a 16 M-thread dispatch.
template <typename T>
__global__ __launch_bounds__(256) void bdy_uniform(const Params<T> p, const Scalars<T>* __restrict__ sc,
                                                   const UniformBdy<T> b, State4<T>* __restrict__ state,
                                                   const T* __restrict__ bed, const bool truncated)
{
	const T t = sc->t, dt_real = sc->dt, dt = sc->t_hydro;
	if (dt < T(1.0) || dt_real <= T(0)) return;                                  // :165-166 (uniform over the grid)
	if (t >= b.length) return;                                                    // :168
	unsigned long ts = (unsigned long)floor_(t / b.interval);                     // :172-173
	if (ts >= b.entries) ts = b.entries - 1;          // a series shorter than its `length`: hold the last entry (the reference reads past the buffer)
	const T rate = b.series[2 * ts + 1];
	const T amount = rate / T(3600000.0) * dt;
	const size_t cells = (size_t)p.cols * p.rows;
	for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < cells; id += (size_t)gridDim.x * blockDim.x) {
		const long y = (long)(id / p.cols), x = (long)(id - (size_t)y * p.cols);
		if (!bdy_in_range(p, x, y + p.row_offset, truncated)) continue;
		State4<T> c = state[id];
		if (c.zmax <= T(-9999.0)) continue;                                       // :168-169
		if (b.definition == 0) c.z += amount;                                     // :176-177
		if (b.definition == 1) c.z = fmax_(bed[id], c.z - amount);                // :179-180
		state[id] = c;
	}
}

template <typename T>
__global__ __launch_bounds__(256) void bdy_gridded(const Params<T> p, const Scalars<T>* __restrict__ sc,
                                                   const GriddedBdy<T> b, State4<T>* __restrict__ state,
                                                   const bool truncated)
{
	const T t = sc->t, dt = sc->t_hydro;
	if (dt < T(1.0)) return;                                                      // :224-225
	unsigned long ts = (unsigned long)floor_(t / b.interval);                     // :228
	if (ts >= b.entries) ts = b.entries - 1;          // reference reads one slice past the end here (:229)
	const size_t cells = (size_t)p.cols * p.rows;
	for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < cells; id += (size_t)gridDim.x * blockDim.x) {
		const long y = (long)(id / p.cols), x = (long)(id - (size_t)y * p.cols);
		const long gy = y + p.row_offset;
		if (!bdy_in_range(p, x, gy, truncated)) continue;
		State4<T> c = state[id];
		if (c.zmax <= T(-9999.0) || c.z == T(-9999.0)) continue;                  // :220-221
		const T col = floor_((((T)x * p.dx) - b.off_x) / b.resolution);           // :231-232
		const T row = floor_((((T)gy * p.dx) - b.off_y) / b.resolution);
		const unsigned long cell = (b.grows * b.gcols) * ts + (b.gcols * (unsigned long)row) + (unsigned long)col;
		const T rate = b.grids[cell];
		if (b.definition == 0) c.z += rate / T(3600000.0) * dt;                   // :238-239
		if (b.definition == 2) c.z += rate / (p.dx * p.dx) * dt;                  // :241-242
		state[id] = c;
	}
}

// All consecutive uniform / gridded boundaries of an iteration in ONE launch: a cell is read once, every boundary is
// applied to it in the order added (each exactly as bdy_uniform / bdy_gridded would, gate and all), and written once.
// Per cell this is the same sequence of operations as the separate launches, so results are bit-identical; what goes
// away is one launch and one 64 B/cell pass per boundary (the reference's example model has two: rain and drainage).
template <typename T>
__global__ __launch_bounds__(256) void bdy_area(const Params<T> p, const Scalars<T>* __restrict__ sc, const AreaBdyList<T> list,
                                                State4<T>* __restrict__ state, const T* __restrict__ bed, const bool truncated,
                                                const T* __restrict__ applied)
{
	if (applied && *applied != T(0)) return;       // the previous iteration's flux kernel has applied them already (K1, FUSED)
	const T t = sc->t, dt_real = sc->dt, dt = sc->t_hydro;
	if (dt < T(1.0)) return;                                                      // CLBoundaries.clc:165, :224 (uniform over the grid)
	// per boundary: is it active this iteration, and (uniform) how much does it add / remove
	bool active[AREA_BDY_MAX];
	T amount[AREA_BDY_MAX];
	unsigned long slice[AREA_BDY_MAX];
	bool any = false;
	for (int k = 0; k < list.count; ++k) {
		const AreaBdy<T>& b = list.b[k];
		active[k] = false; amount[k] = T(0); slice[k] = 0;
		if (b.kind == 0) {
			if (dt_real <= T(0) || t >= b.u.length) continue;                     // :165-168
			unsigned long ts = (unsigned long)floor_(t / b.u.interval);           // :172-173
			if (ts >= b.u.entries) ts = b.u.entries - 1;
			amount[k] = b.u.series[2 * ts + 1] / T(3600000.0) * dt;
			active[k] = true;
		} else {
			unsigned long ts = (unsigned long)floor_(t / b.g.interval);           // :228
			if (ts >= b.g.entries) ts = b.g.entries - 1;
			slice[k] = ts;
			active[k] = true;
		}
		any = any || active[k];
	}
	if (!any) return;
	bool need_bed = false;                                                        // only the loss rate looks at the bed (:179-180)
	for (int k = 0; k < list.count; ++k) need_bed = need_bed || (active[k] && list.b[k].kind == 0 && list.b[k].u.definition == 1);
	const size_t cells = (size_t)p.cols * p.rows;
	for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < cells; id += (size_t)gridDim.x * blockDim.x) {
		const long y = (long)(id / p.cols), x = (long)(id - (size_t)y * p.cols);
		const long gy = y + p.row_offset;
		if (!bdy_in_range(p, x, gy, truncated)) continue;
		State4<T> c = state[id];
		if (c.zmax <= T(-9999.0)) continue;                                       // :168-169, :220-221 (first half)
		const T zb = need_bed ? bed[id] : T(0);
		for (int k = 0; k < list.count; ++k) {
			if (!active[k]) continue;
			const AreaBdy<T>& b = list.b[k];
			if (b.kind == 0) {
				if (b.u.definition == 0) c.z += amount[k];                        // :176-177
				if (b.u.definition == 1) c.z = fmax_(zb, c.z - amount[k]);        // :179-180
			} else {
				if (c.z == T(-9999.0)) continue;                                  // :220-221 (second half; re-tested as the level changes)
				const T col = floor_((((T)x * p.dx) - b.g.off_x) / b.g.resolution);   // :231-232
				const T row = floor_((((T)gy * p.dx) - b.g.off_y) / b.g.resolution);
				const unsigned long cell = (b.g.grows * b.g.gcols) * slice[k] + (b.g.gcols * (unsigned long)row) + (unsigned long)col;
				const T rate = b.g.grids[cell];
				if (b.g.definition == 0) c.z += rate / T(3600000.0) * dt;         // :238-239
				if (b.g.definition == 2) c.z += rate / (p.dx * p.dx) * dt;        // :241-242
			}
		}
		state[id] = c;
	}
}

// bdy_Cell (Boundaries/CLBoundaries.clc:23-128): imposed depth / level / discharge / velocity / volume on a list of
// cells with linear interpolation in time.  Cell ids are global; a strip applies the ones it stores (ghost rows too).
template <typename T> struct CellBdy {
	const unsigned long long* cells; unsigned long long count;
	const T* series; unsigned long long entries; int depth_def, discharge_def; T interval, length;
};

template <typename T>
__global__ __launch_bounds__(64) void bdy_cell(const Params<T> p, const Scalars<T>* __restrict__ sc, const CellBdy<T> b,
                                               State4<T>* __restrict__ state, const T* __restrict__ bed)
{
	const T t = sc->t, dt = sc->dt;
	const unsigned long long r = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= b.count || t >= b.length || dt <= T(0)) return;                       // :38-39
	const unsigned long long gid = b.cells[r];
	const long gy = (long)(gid / (unsigned long long)p.cols), x = (long)(gid - (unsigned long long)gy * p.cols);
	const long y = gy - p.row_offset;
	if (y < 0 || y >= p.rows) return;                                              // not stored by this strip
	const size_t id = (size_t)y * p.cols + x;
	const unsigned long long base = (unsigned long long)floor_(t / b.interval), next = base + 1;   // :41-42
	State4<T> c = state[id];
	const T zb = bed[id];
	const T w = fmod_(t, b.interval) / b.interval;                                 // :50
	T ts[4];
	for (int k = 0; k < 4; ++k) ts[k] = b.series[4 * base + k] + (b.series[4 * next + k] - b.series[4 * base + k]) * w;

	const T g = gravity<T>();
	if (b.depth_def == 2) {                                                        // depth is fixed (:53-59)
		c.z = zb + ts[1];
	} else if (b.depth_def == 1) {                                                 // level is fixed (:60-66)
		c.z = fmax_(zb, ts[1]);
	} else if (fabs_(ts[2]) > p.vs || fabs_(ts[3]) > p.vs || b.discharge_def == 3) {   // :67-98
		T depth = (fabs_(ts[2]) * dt) / p.dx + (fabs_(ts[3]) * dt) / p.dx;
		T crit = fmax_(pow13_((ts[2] * ts[2]) / g), pow13_((ts[3] * ts[3]) / g));      // :81
		if (b.discharge_def == 3) {                                                // volume: no direction, no scaling
			depth = (fabs_(ts[2]) * dt) / (p.dx * p.dx);
			crit = T(0);
			ts[2] = T(0);
			ts[3] = T(0);
		}
		c.z = fmax_(zb + crit, c.z + depth);
	}
	if (b.discharge_def == 1) { c.qx = ts[2]; c.qy = ts[3]; }                      // :100-118
	else if (b.discharge_def == 2) { c.qx = ts[2] * (c.z - zb); c.qy = ts[3] * (c.z - zb); }
	state[id] = c;
}

} // namespace hp
