// hp_engine.hip -- host side of libhipims_mi.so: the C ABI of include/hipims_mi.h over the HIP kernels.
//
// Mirrors what the reference's CSchemeGodunov / CSchemeMUSCLHancock do against COCLDevice/COCLBuffer/COCLKernel
// (buffers: CSchemeGodunov.cpp:789-892; iteration graph: :1617-1666) -- one HIP stream per domain stands in for
// the reference's per-device command queue, stream order for its explicit queueBarrier() calls.
#include "../../include/hipims_mi.h"
#include "hp_kernels.hpp"
#include <hip/hip_ext.h>
#include <rccl/rccl.h>          // types and prototypes only: the library itself is dlopen'ed (hp_comm_load)
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <unistd.h>

using namespace hp;

namespace {

thread_local std::string g_last_error;

constexpr size_t CFL_SLOT_BYTES = 1024;        // hp_math.hpp: SLOT_* elements, each on its own 256-B line

// RCCL entry points, resolved by hp_comm_load
struct Rccl {
	void* handle = nullptr;
	decltype(&ncclGetUniqueId)   GetUniqueId = nullptr;
	decltype(&ncclCommInitRank)  CommInitRank = nullptr;
	decltype(&ncclCommDestroy)   CommDestroy = nullptr;
	decltype(&ncclGroupStart)    GroupStart = nullptr;
	decltype(&ncclGroupEnd)      GroupEnd = nullptr;
	decltype(&ncclSend)          Send = nullptr;
	decltype(&ncclRecv)          Recv = nullptr;
	decltype(&ncclAllReduce)     AllReduce = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	decltype(&ncclCommCount)     CommCount = nullptr;        // optional (reporting only)
	std::string path;                                        // what the dynamic loader resolved the library to
} g_rccl;

std::atomic<hp_log_sink_t> g_log_sink{nullptr};
std::atomic<void*>         g_log_user{nullptr};

int fail(int code, const std::string& msg)
{
	g_last_error = msg;
	if (hp_log_sink_t sink = g_log_sink.load(std::memory_order_acquire))
		sink(HP_LOG_MODEL_STOP, g_last_error.c_str(), g_log_user.load(std::memory_order_acquire));
	return code;
}

void log_line(int level, const std::string& msg)
{
	if (hp_log_sink_t sink = g_log_sink.load(std::memory_order_acquire))
		sink(level, msg.c_str(), g_log_user.load(std::memory_order_acquire));
}

#define HIP_TRY(expr)                                                                                  \
	do {                                                                                               \
		hipError_t e_ = (expr);                                                                        \
		if (e_ != hipSuccess)                                                                          \
			return fail(HP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                \
	} while (0)

struct Boundary {
	int      kind;         // 0 uniform, 1 gridded, 2 cell
	int      definition;
	int      discharge_def;
	void*    cells;        // device: global cell ids (cell boundaries)
	uint64_t count;
	void*    data;         // device
	uint64_t entries, grows, gcols;
	double   interval, length, resolution, off_x, off_y;
};

} // namespace

struct hp_domain {
	hp_domain_desc_t desc;
	hipStream_t      stream = nullptr;
	size_t           cells = 0, esize = 0;
	void*            state[2] = {nullptr, nullptr};   // [0] = primary "Cell states", [1] = "Cell states (alternate)"
	void*            bed = nullptr;
	void*            manning = nullptr;
	void*            scalars = nullptr;               // Scalars<T> on the device
	void*            cfl_slot = nullptr;              // CFL_SLOT_BYTES: running max | SLOT_SAVED last used max | SLOT_EDGE ring maxima of [0], [1]
	bool             manning_uniform = false;         // found at upload: one value everywhere -> kernels skip the array
	double           manning_value = 0.0;
	bool             need_full_reduce = true;         // the remembered maximum is stale (upload / link import)
	bool             edge_dirty = true;               // edge-ring maxima must be re-priced
	bool             bdy_on_ring = false;             // a cell boundary imposes values on never-written ring cells: re-price every iteration
	int              adv_fresh = 1;                   // does hp_step_end's advance kernel read a new maximum?
	int              march_rseg = 16;                 // rows per wavefront tile of godunov_march
	int              muscl_rseg = 32;                 // ... of muscl_march (two warm-up rows per tile)
	int              inertial_rseg = 32;              // ... of inertial_march
	int              march_nbands = 8, muscl_nbands = 8, inertial_nbands = 8;   // row bands of a whole-domain launch (pick_tiling)
	int              march_rseg_parts = 16, inertial_rseg_parts = 16;            // tile height of the interior / halo parts of a split step (8 / 2 bands: the classic height)
	int              sweep_flip = 0;                  // parity of the whole-domain flux launches: every other one visits each band's tiles from
	                                                  // the top down, so that it starts on the rows its predecessor wrote last (sweep_alternates)
	int              tall_rseg = 18;                  // K1/K6 tile height where an XCD band has >= 256 rows (16 if a knob is set)
	int              tail_rseg = 8, tail_pct = 0;     // optional short tiles for the last tail_pct % of each XCD band (measured: no gain)
	void*            host_scalars = nullptr;          // pinned mirror
	int              use_alt = 0;                     // bUseAlternateKernel
	bool             in_step = false;
	std::vector<Boundary> bdy;
	// area boundaries carried by the flux kernel's fused epilogue (K1, FUSED): device copy of their descriptors
	void*            fused_list = nullptr;
	bool             fusable = false;                 // Godunov, tuned kernel, 1..FUSED_BDY_MAX uniform / coarse gridded boundaries, no cell boundary
	int              fuse_next = 0;                   // this iteration is followed by another one of the same batch
	uint64_t         cells_calculated = 0, iterations = 0;
	hipEvent_t       ev_start = nullptr, ev_stop = nullptr;
	// flux-kernel timing samples
	int              timing_stride = 0;
	uint64_t         timing_counter = 0;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> timing_events;   // pool, created by hp_kernel_timing
	size_t           timing_used = 0;
	double           timing_overhead_ms = 0.0;        // an empty event pair's own time (taken off every sample)
	long             own_lo = 0, own_hi = 0;          // rows this rank owns (CFL reduction range)
	// ghost rows of a strip: `ghost_rows` stored per interior side (g or 2g, g = the scheme's stencil reach), of which
	// `ghost_valid` currently hold their owners' values; an iteration consumes g of them, the exchange refills them all
	long             ghost_rows = 0, ghost_valid = 0, saved_ghost_valid = 0;
	bool             split_now = false;               // this iteration is followed by an exchange: halo part on its own stream
	// what the ranks told each other at the start of the batch (hp_strip_step_batch): which of them price a new maximum on
	// the iterations that the ping-pong phase alone would not make them price
	bool             strip_any_bdy = false, strip_any_full = false;
	bool             strip_pairs = false;             // the batch's handshake: every rank can run iteration pairs (godunov_march2 over two-reach ghost rows)
	// halo overlap (strip decomposition): the row segments next to the ghost rows run on their own stream so the
	// neighbours' halo transfer can start while the interior segments are still being computed
	bool             halo_overlap = false;
	bool             halo_overlap_set = false;        // the host chose explicitly (hp_set_halo_overlap): comm_init keeps it
	hipStream_t      stream_halo = nullptr;
	hipEvent_t       ev_fork = nullptr, ev_halo = nullptr;
	bool             fork_is_advance = false;         // ev_fork was recorded BY the last advance_time launch
	// strip decomposition driven from C++ (hp_strip_*): one RCCL communicator over the ranks, strip neighbours = rank +- 1
	// device-side checkpoint (hp_state_save / hp_state_restore)
	void*            saved_state = nullptr;
	void*            saved_scalars = nullptr;         // Scalars<T> + the four CFL slots
	bool             saved_valid = false;
	bool             saved_full_reduce = true, saved_edge_dirty = true;
	int              saved_use_alt = 0;
	ncclComm_t       comm = nullptr;
	int              comm_rank = 0, comm_world = 1;
	hipEvent_t       ev_xchg = nullptr;               // ghost rows of the iteration in flight have arrived
	// the maximum over all strips through peer-written mailboxes (hp_strip_peer_*; PeerBox in hp_kernels.hpp)
	unsigned long long*  peer_mine = nullptr;         // this rank's mailbox (uncached device memory)
	unsigned long long** peer_table = nullptr;        // device: every rank's mailbox as this device addresses it
	std::vector<void*>   peer_mapped;                 // IPC mappings to close
	int              peer_world = 0, peer_rank = 0;
	bool             peer_agreed = false;             // every rank of the communicator passed the connection test: the strip loop uses them
	uint64_t         peer_rounds = 0;                 // reductions so far (its parity picks the mailbox set; the same on every rank)
	// ghost rows written straight into the neighbours' state buffers (PeerPush): [side: 0 south, 1 north][ping-pong buffer]
	void*            peer_state[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
	long             peer_rows[2] = {0, 0};           // the neighbours' local row counts
	unsigned*        push_arrived = nullptr;          // device counter of the push blocks
	bool             peer_direct = false;             // every rank reached its neighbours' buffers: no transfer library inside an iteration
	bool             push_now = false;                // this iteration's advance kernel carries the ghost rows
	// the flux launch's own tail block instead of a separate advance launch (LaunchTail, hp_kernels.hpp): small launches only
	unsigned long long* tail_words = nullptr;         // one word per flux block (EMPTY between launches)
	bool             rings_differ = false;            // a partial state upload went into ONE buffer: the edge rings of the two may differ -- no iteration pairs until the next full upload (pair_eligible)
	bool             saved_rings_differ = false;
	bool             rings_checked = false;           // ... and have been compared since (rings_really_differ): the flag is a fact, not a maybe
	bool             fill_now = false;                // this iteration's K1 launch stores the cells the reference leaves untouched as well (dispatch_begin)
	bool             other_stale = false;             // pairs (godunov_march2) ran since the non-current state buffer last held a state the single-iteration kernels can build on
	int              march2_rseg = 24;                // tile height of the two-iterations kernel
	bool             march2_pays = false;             // the grid is big enough for it (hp_domain_create)
	int              march2_nbands = 8;               // its row bands (one-round grids: searched)
	bool             print_tiling = false;
	// quirk Q3 across pair launches, exactly (hp_kernels.hpp: PairAux): stamps of the cells whose first-step stale value the next launch needs
	void*            z_state = nullptr;               // stamp records, one per cell: State4<T> + the number of the pair launch that wrote it (allocated with the first pair)
	unsigned long long* haz_words = nullptr;          // two words: [g & 1] == g <=> pair launch g stamped something
	unsigned         pair_gen = 0;                    // number of the last pair launch -- the one that wrote the current state while other_stale holds
	bool             saved_m1_valid = false;
	bool             pair_fused_next = false;         // the last pair stored its state with the next iteration's boundaries applied (SLOT_BDY = 1)
	bool             m1_valid = false;                // area boundaries: cfl_slot[SLOT_M1] prices the primary buffer with the next iteration's boundaries (left by the last pair)
	uint64_t         pair_cold_starts = 0;
	// STRICT: pairs or single iterations, by measurement (pair_tuner: the two are the same bits; which is faster depends on how much of
	// the water stands still)
	hipEvent_t       tune_ev[4] = {nullptr, nullptr, nullptr, nullptr};
	int              tune_phase = 0;                  // 0: sample at the next opportunity, 1: a sample is in flight, 2: decided
	bool             tune_prefer_pairs = true;
	uint64_t         tune_next = 0;                   // iteration count from which the next sample is due
	uint64_t         tune_samples = 0, tune_switches = 0;
	float            tune_pair_ms = 0.f, tune_single_ms = 0.f;
	uint64_t         pairs = 0;                       // iteration pairs run by it
	unsigned         single_streak = 0;               // single iterations since the last pair (step_begin_impl: which launches hp_kernel_timing samples)
	uint64_t         flux_launches = 0, flux_launches_tailed = 0;   // whole-domain flux launches of hp_step_batch / hp_strip_step_batch, and how many carried their own tail block
	bool             tail_failed = false;             // a tail block gave up waiting (SLOT_TAIL_ERR seen by the host): the domain is unusable
	bool             strip_first = false;             // hp_strip_step_batch: the batch's first iteration
	bool             tail_allowed = false;            // inside hp_step_batch / hp_strip_step_batch (the host-driven split step has work in between)
	bool             tail_want = false;               // step_begin_impl: this iteration qualifies, if the launch is small enough
	int              tail_fresh = 0;                  // ... and this is what its advance would be told
	bool             tail_done = false;               // launch_march: the launch carried it -- hp_step_end launches nothing
	// speculative STRICT fp64 batches (hp_math.hpp: div_shared; spec_begin / spec_resolve below)
	bool             spec_now = false;                // the flux launches queued now are the shared-reciprocal instantiations
	uint32_t         spec_pending = 0;                // iterations of a speculative batch whose flag word has not been looked at yet
	void*            spec_state = nullptr;            // snapshot in front of the batch: both state buffers ...
	void*            spec_scalars = nullptr;          // ... Scalars<T> + the slot block
	struct { int use_alt, adv_fresh; bool need_full_reduce, edge_dirty; long ghost_valid; uint64_t cells_calculated, iterations; } spec_host;
	uint64_t         spec_batches = 0, spec_replays = 0;
};

namespace {

void peer_release(hp_domain* d);      // (mailboxes of the peer-written maximum, further down)
PeerBox peer_box(hp_domain* d, bool use, long timeout_ms = 0);
PeerPush make_push(const hp_domain* d);
constexpr unsigned TAIL_MAX_BLOCKS = 65536;    // words allocated per domain
static unsigned tail_limit()
{
	static const unsigned v = std::getenv("HP_TAIL_MAX_BLOCKS") ? (unsigned)std::atoi(std::getenv("HP_TAIL_MAX_BLOCKS")) : TAIL_MAX_BLOCKS;   // (a knob for A/B runs; 768 = one round of blocks)
	return v < TAIL_MAX_BLOCKS ? v : TAIL_MAX_BLOCKS;
}

// stencil reach of the scheme = ghost rows one iteration consumes
inline long strip_ghosts(const hp_domain* d) { return d->desc.scheme == HP_SCHEME_MUSCL_HANCOCK ? 2 : 1; }

template <typename T> Params<T> make_params(const hp_domain* d)
{
	Params<T> p;
	p.cols = d->desc.cols; p.rows = d->desc.rows;
	p.row_offset = d->desc.row_offset; p.global_rows = d->desc.global_rows;
	p.dx = (T)d->desc.dx;
	p.inv_dx = T(1) / p.dx;
	p.vs = (T)d->desc.dry_threshold;
	p.qs = (T)d->desc.dry_threshold * T(10);             // CSchemeGodunov.cpp:57, :523
	p.courant = (T)d->desc.courant;
	p.t_end = (T)d->desc.t_end;
	p.dt_fixed = (T)d->desc.dt_fixed;
	p.friction = d->desc.friction;
	p.dynamic_dt = d->desc.dynamic_dt;
	p.manning_uniform = d->manning_uniform ? 1 : 0;
	p.manning_value = (T)d->manning_value;
	p.simplified_cfl = d->desc.scheme == HP_SCHEME_INERTIAL ? 1 : 0;       // CLSchemeInertial.clh:25
	p.muscl_nb_bed = (d->desc.quirks & HP_QUIRK_MUSCL_NEIGHBOUR_Y_IS_BED) ? 1 : 0;
	{
		// dx a power of two (in T's arithmetic): STRICT's divisions by dx become multiplications by the exact 1 / dx
		int e = 0;
		const double m = std::frexp((double)(T)d->desc.dx, &e);
		static const bool off = std::getenv("HP_STRICT_DX_POW2") && std::atoi(std::getenv("HP_STRICT_DX_POW2")) == 0;
		p.inv_dx_pow2 = (m == 0.5 && !off && e > -100 && e < 100) ? (T)(1.0 / (double)(T)d->desc.dx) : T(0);
	}
	return p;
}

inline dim3 grid2d(long cols, long rows, dim3 block)
{
	return dim3((unsigned)((cols + block.x - 1) / block.x), (unsigned)((rows + block.y - 1) / block.y));
}

template <typename T> int apply_boundaries(hp_domain* d, void* target)
{
	const Params<T> p = make_params<T>(d);
	const bool truncated = (d->desc.quirks & HP_QUIRK_BDY_TRUNCATED) != 0;
	size_t want = ((size_t)p.cols * p.rows + 255) / 256;
	const dim3 block(256), grid((unsigned)(want < 1024 ? want : 1024));
	// consecutive uniform / gridded boundaries go out as ONE launch (bdy_area); cell boundaries keep their place in
	// the order added (the reference's order is unspecified, quirk Q7; ours is the order of the hp_boundary_add_* calls)
	AreaBdyList<T> list;
	list.count = 0;
	auto flush = [&]() {
		if (list.count == 0) return;
		hipLaunchKernelGGL(bdy_area<T>, grid, block, 0, d->stream, p, (const Scalars<T>*)d->scalars, list, (State4<T>*)target,
		                   (const T*)d->bed, truncated, d->fusable ? (const T*)d->cfl_slot + SLOT_BDY : (const T*)nullptr);
		list.count = 0;
	};
	for (const Boundary& b : d->bdy) {
		if (b.kind == 2) {
			flush();
			CellBdy<T> c{(const unsigned long long*)b.cells, b.count, (const T*)b.data, b.entries, b.definition,
			             b.discharge_def, (T)b.interval, (T)b.length};
			hipLaunchKernelGGL(bdy_cell<T>, dim3((unsigned)((b.count + 63) / 64)), dim3(64), 0, d->stream, p,
			                   (const Scalars<T>*)d->scalars, c, (State4<T>*)target, (const T*)d->bed);
			continue;
		}
		if (list.count == AREA_BDY_MAX) flush();
		AreaBdy<T>& a = list.b[list.count++];
		a = AreaBdy<T>{};
		if (b.kind == 0) {
			a.kind = 0;
			a.u = UniformBdy<T>{(const T*)b.data, (uint32_t)b.entries, b.definition, (T)b.interval, (T)b.length};
		} else {
			a.kind = 1;
			a.g = GriddedBdy<T>{(const T*)b.data, b.entries, b.grows, b.gcols, b.definition,
			                    (T)b.resolution, (T)b.off_x, (T)b.off_y, (T)b.interval};
		}
	}
	flush();
	HIP_TRY(hipGetLastError());
	return HP_OK;
}

template <typename T> int launch_reduce(hp_domain* d, const void* state, long row_lo, long row_hi)
{
	const Params<T> p = make_params<T>(d);
	const size_t n = (size_t)(row_hi - row_lo) * p.cols;
	unsigned blocks = (unsigned)((n + 256 * 8 - 1) / (256 * 8));
	if (blocks > 2048) blocks = 2048;
	if (blocks < 1) blocks = 1;
	// in the domain's own arithmetic: the same buffer prices to the same bits here and in the flux kernels' fused epilogue
	if (d->desc.math_mode == HP_MATH_STRICT)
		hipLaunchKernelGGL((cfl_reduce<true, T>), dim3(blocks), dim3(256), 0, d->stream, p, (const State4<T>*)state,
		                   (const T*)d->bed, row_lo, row_hi, (T*)d->cfl_slot);
	else
		hipLaunchKernelGGL((cfl_reduce<false, T>), dim3(blocks), dim3(256), 0, d->stream, p, (const State4<T>*)state,
		                   (const T*)d->bed, row_lo, row_hi, (T*)d->cfl_slot);
	HIP_TRY(hipGetLastError());
	return HP_OK;
}

template <typename T> int price_edge_ring(hp_domain* d)
{
	const Params<T> p = make_params<T>(d);
	HIP_TRY(hipMemsetAsync((T*)d->cfl_slot + SLOT_EDGE, 0, 2 * sizeof(T), d->stream));
	const int w = (d->desc.scheme == HP_SCHEME_MUSCL_HANCOCK) ? 2 : 1;
	const bool at_south = d->desc.row_offset == 0;
	const bool at_north = d->desc.row_offset + d->desc.rows == d->desc.global_rows;
	const long south = at_south ? 0 : -1;
	const long north = at_north ? d->desc.rows - w : -1;
	// side columns of the owned rows, minus the rows already covered by south/north
	const long lo = d->own_lo + (at_south ? w : 0), hi = d->own_hi - (at_north ? w : 0);
	for (int b = 0; b < 2; ++b) {
		if (d->desc.math_mode == HP_MATH_STRICT)
			hipLaunchKernelGGL((cfl_edge_ring<true, T>), dim3(64), dim3(256), 0, d->stream, p, (const State4<T>*)d->state[b],
			                   (const T*)d->bed, lo, hi, south, north, w, (T*)d->cfl_slot + SLOT_EDGE + b);
		else
			hipLaunchKernelGGL((cfl_edge_ring<false, T>), dim3(64), dim3(256), 0, d->stream, p, (const State4<T>*)d->state[b],
			                   (const T*)d->bed, lo, hi, south, north, w, (T*)d->cfl_slot + SLOT_EDGE + b);
	}
	HIP_TRY(hipGetLastError());
	d->edge_dirty = false;
	return HP_OK;
}

// Which rows of the domain one launch covers
enum { PART_ALL = 0, PART_HALO = 1, PART_INTERIOR = 2 };

// Row ranges of the three parts.  [lo, hi) are the updated rows of the domain; with the halo overlap the `halo` rows
// next to each ghost block (one tile height, at least the g rows the neighbours need) form the halo part -- a south
// and a north block, covered by ONE launch with two bands -- and the rest the interior part.  A domain too thin for
// an interior part runs whole in the halo part.
struct RowRange { long lo, hi; };

// TileMap of one launch (see hp_kernels.hpp): 8 XCD bands over [lo, hi), tall tiles first and optionally short tiles
// for the last `tail_pct` percent of each band; or, for the halo part, the two blocks of `halo` rows as two bands.
inline bool make_tile_map(long lo, long hi, int g, int part, int nstrips, int rseg, int rseg_tail, int tail_pct,
                          TileMap& tm, unsigned& blocks, int rseg_tall = 16, long need = 0, long price_lo = 0, long price_hi = 0x7fffffffL,
                          int nbands = 8)
{
	static const long halo_env = std::getenv("HP_HALO_ROWS") ? std::atol(std::getenv("HP_HALO_ROWS")) : 0;
	if (need < g) need = g;                                             // rows at each end that a strip neighbour is sent
	const long halo = halo_env >= need ? halo_env : (rseg > need ? rseg : need);
	tm.price_lo = (int)price_lo; tm.price_hi = (int)(price_hi < 0x7fffffffL ? price_hi : 0x7fffffffL);
	tm.flip = 0;
	const bool can_split = hi - lo > 2 * halo;
	tm.nstrips = nstrips;
	tm.groups = (nstrips + 3) / 4;
	if (part == PART_HALO && can_split) {
		tm.y_begin = lo; tm.y_end = hi;
		tm.nbands = 2; tm.band_stride = (hi - halo) - lo; tm.band_rows = (int)halo;
		tm.rseg = rseg < tm.band_rows ? rseg : tm.band_rows;
		tm.nbig = (tm.band_rows + tm.rseg - 1) / tm.rseg;
		tm.rseg_tail = rseg_tail; tm.ntail = 0;
		blocks = 2u * (unsigned)(tm.groups * tm.nbig);
		return true;
	}
	if (part == PART_INTERIOR) {
		if (!can_split) return false;                                   // the halo launch took everything
		lo += halo; hi -= halo;
	}
	tm.y_begin = lo; tm.y_end = hi;
	const long rows = hi - lo;
	// Row bands: 8, one per XCD -- or, for a whole-domain launch that fits the chip in ONE round, the number the tiling
	// search of hp_domain_create found (pick_tiling: such a launch is bound by its most loaded CU, and 8 x groups x segments
	// only offers coarse block counts).  HP_NBANDS forces a number (tools/history/r04_band_sweep.py).
	static const int nbands_env = std::getenv("HP_NBANDS") ? std::atoi(std::getenv("HP_NBANDS")) : 0;
	tm.nbands = part != PART_ALL ? 8 : (nbands_env >= 1 && nbands_env <= 64 ? nbands_env : (nbands >= 1 && nbands <= 64 ? nbands : 8));
	tm.band_rows = (int)((rows + tm.nbands - 1) / tm.nbands);
	tm.band_stride = tm.band_rows;
	// K1/K6 on tall bands: 18-row tiles measured 1.9 % / 2.8 % ahead of 16 at 4096^2 (band of 512 rows) and 0.6 % at
	// 8192 x 2050 (256), but 3 % behind at 16384 x 1026 (128 rows = eight exact 16-row tiles); interleaved repeats
	// on one box, tools/rseg_fine_sweep*.sh
	if (rseg == 16 && g == 1 && tm.band_rows >= 256) rseg = rseg_tall;
	tm.rseg = rseg < tm.band_rows ? rseg : tm.band_rows;
	tm.rseg_tail = rseg_tail;
	if (rseg_tail >= tm.rseg || tail_pct <= 0) {
		tm.nbig = (tm.band_rows + tm.rseg - 1) / tm.rseg;
		tm.ntail = 0;
	} else {
		const int tail_rows = (int)((long)tm.band_rows * tail_pct / 100);
		tm.nbig = (tm.band_rows - tail_rows + tm.rseg - 1) / tm.rseg;
		const int rest = tm.band_rows - tm.nbig * tm.rseg;
		tm.ntail = rest > 0 ? (rest + rseg_tail - 1) / rseg_tail : 0;
	}
	blocks = (unsigned)tm.nbands * (unsigned)(tm.groups * (tm.nbig + tm.ntail));
	return true;
}

// Rows one flux launch updates: everything between the never-written edge ring / the outermost ghost rows.  A strip with
// two reaches of ghost rows (ghost_rows = 2g) also updates, on the iteration that follows an exchange, the g ghost rows
// next to its owned rows -- redundantly with their owner, from the same values, to the same bits -- and on the next
// iteration, which finds only those still valid, its owned rows alone.
inline void launch_rows(const hp_domain* d, long g, long& lo, long& hi)
{
	const bool south = d->desc.row_offset > 0, north = d->desc.row_offset + d->desc.rows < d->desc.global_rows;
	const long keep = d->ghost_rows - (d->ghost_valid - g);             // ghost rows at each interior side NOT updated now
	lo = south ? keep : g;
	hi = d->desc.rows - (north ? keep : g);
}

// The launch's own tail block (LaunchTail, hp_kernels.hpp), and with it the rows that leave with this iteration: the tiles that
// compute the strip's first / last `ghost_rows` owned rows store them into the neighbours as well -- the neighbour's pointer
// is shifted so that this strip's cell index lands on the neighbour's copy of the cell.  `limit`: launches of more blocks keep
// the advance launch.  Returns 0 (no tail), 1 (tail block) or 2 (tail block + rows stored into the neighbours).
// how long a launch's tail block waits for ONE flux block's word before it raises SLOT_TAIL_ERR (hp_kernels.hpp: launch_tail).
// Generous: the wait only starts once the tail block runs, i.e. once its XCD has dispatched every flux block of the launch.
static long tail_timeout_ms()
{
	static const long ms = std::getenv("HP_TAIL_TIMEOUT_MS") ? std::atol(std::getenv("HP_TAIL_TIMEOUT_MS")) : 20000;
	return ms > 0 ? ms : 20000;
}
// (tests) the tail block also waits for one word that no block writes: the time-out must fire, the call sequence must fail
static unsigned tail_debug_extra_word()
{
	static const unsigned v = std::getenv("HP_DEBUG_TAIL_EXTRA_WORD") && std::atoi(std::getenv("HP_DEBUG_TAIL_EXTRA_WORD")) != 0 ? 1u : 0u;
	return v;
}
template <typename T> int make_tail(hp_domain* d, unsigned& blocks, int part, hipStream_t stream, unsigned limit, LaunchTail<T>& tail)
{
	tail = LaunchTail<T>{};
	if (part == PART_ALL) d->flux_launches++;
	if (!(d->tail_want && part == PART_ALL && stream == d->stream && blocks <= limit)) return 0;
	tail.done = d->tail_words;
	tail.done1 = d->tail_words + TAIL_MAX_BLOCKS + 8;
	tail.bdy_flag = 0;
	tail.flux_blocks = blocks;
	tail.poll_blocks = blocks + tail_debug_extra_word();
	tail.timeout = (unsigned long long)tail_timeout_ms() * 100000ull;       // wall_clock64: 100 MHz
	tail.fresh = d->tail_fresh;
	tail.sc = (Scalars<T>*)d->scalars;
	tail.slot = (T*)d->cfl_slot;
	tail.box = peer_box(d, (d->tail_fresh & 4) != 0);
	tail.edge_rows[0] = tail.edge_rows[1] = tail.edge_rows[2] = tail.edge_rows[3] = 0;
	tail.peer_rows[0] = tail.peer_rows[1] = nullptr;
	int kind = 1;
	if (d->push_now) {
		const long G = d->ghost_rows, rows = d->desc.rows, cols = d->desc.cols;
		const int b = d->use_alt ^ 1;
		if (d->peer_state[0][b]) {
			tail.peer_rows[0] = (State4<T>*)d->peer_state[0][b] + (d->peer_rows[0] - 2 * G) * cols;     // my row G -> its row rows_s - G
			tail.edge_rows[0] = (int)G; tail.edge_rows[1] = (int)(2 * G);
		}
		if (d->peer_state[1][b]) {
			tail.peer_rows[1] = (State4<T>*)d->peer_state[1][b] - (rows - 2 * G) * cols;                // my row rows - 2G -> its row 0
			tail.edge_rows[2] = (int)(rows - 2 * G); tail.edge_rows[3] = (int)(rows - G);
		}
		kind = 2;
	}
	blocks += 1;
	d->tail_done = true;
	d->flux_launches_tailed++;
	return kind;
}
// (K2 / K6 have their own knob for A/B runs.  A first probe had them lose 0.5 / 1.7 % at the 4608 blocks of 4096^2 and a limit of
// two rounds of blocks was set; bench.py's interleaved repeats on one box then showed the opposite -- MUSCL-Hancock 4096^2
// 0.311 -> 0.302 ms, fp32 0.215 -> 0.192, inertial 0.268 -> 0.249-0.263 -- so there is no limit of their own any more)
static unsigned tail_limit_k2k6()
{
	static const unsigned v = std::getenv("HP_TAIL_MAX_BLOCKS_K2K6") ? (unsigned)std::atoi(std::getenv("HP_TAIL_MAX_BLOCKS_K2K6")) : TAIL_MAX_BLOCKS;
	return std::min(tail_limit(), v);
}

// Alternating sweep direction.  A launch leaves the last ~250 MiB it loaded or stored in the Infinity Cache (256 MiB, stores
// allocate: MI355X_MICROARCH.md), and what it stored last is the top of every row band -- exactly what the NEXT launch, which reads
// the state this one wrote, would reach last.  Every other whole-domain launch therefore visits each band's tiles from the top down
// (TileMap::flip: the same tiles mirrored within the band): it starts on rows whose new state and bed may still be on the die.  Pure
// scheduling: which wavefront solves a face never changes its bits.  Measured (profiles/r04zv_sweep_alternate_again.txt,
// r04zw_sweep_alternate_f64.txt; same box, interleaved): fp32, whose launch is 0.6 GB: S-DAM 4096^2 +5 %, S-RAIN +2.5 %, S-ROUGH
// +3 %, MUSCL +2.3 %, 8192^2 +0.7 %; fp64 MUSCL +2.5 % on the dam break, +0.5 % on live water; fp64 Godunov and inertial: 0 to -3 %
// (2048^2, 16384 x 1026, 16384 x 8192) and erratic at 4096^2 and 8192^2.  Box-dependent: a third box showed no difference at all
// for fp32 (0.1290 / 0.1300 / 0.1291 ms).  So: on for fp32 and for fp64 MUSCL, where it was never behind; off for fp64 Godunov /
// inertial; HP_SWEEP_ALTERNATE=0 / 1 forces it.
inline bool sweep_alternates(const hp_domain* d)
{
	static const int forced = std::getenv("HP_SWEEP_ALTERNATE") ? (std::atoi(std::getenv("HP_SWEEP_ALTERNATE")) != 0 ? 1 : 0) : -1;
	if (forced >= 0) return forced != 0;
	return d->desc.precision == 4 || d->desc.scheme == HP_SCHEME_MUSCL_HANCOCK;
}
inline void sweep_direction(hp_domain* d, const int part, TileMap& tm)
{
	if (part != PART_ALL || !sweep_alternates(d)) return;
	tm.flip = d->sweep_flip;
	d->sweep_flip ^= 1;
}

template <typename T, bool STRICT, int CFL_MODE>
int launch_muscl(hp_domain* d, const void* src, void* dst, int edge_buffer, int part, hipStream_t stream)
{
	const Params<T> p = make_params<T>(d);
	TileMap tm;
	unsigned blocks;
	long lo, hi;
	launch_rows(d, 2, lo, hi);
	if (!make_tile_map(lo, hi, 2, part, (int)((p.cols - 4 + MUSCL_COLS - 1) / MUSCL_COLS), d->muscl_rseg, d->tail_rseg < 8 ? 8 : d->tail_rseg, d->tail_pct, tm, blocks,
	                   16, d->ghost_rows, d->own_lo, d->own_hi, d->muscl_nbands))
		return HP_OK;
	sweep_direction(d, part, tm);
	LaunchTail<T> tail;
	const int tail_kind = make_tail<T>(d, blocks, part, stream, tail_limit_k2k6(), tail);
#define HP_LAUNCH_K2S(UNIFORM_, TAIL_, SPEC_)                                                                                        \
	hipLaunchKernelGGL((muscl_march<STRICT, CFL_MODE, UNIFORM_, TAIL_, T, SPEC_>), dim3(blocks), dim3(256), 0, stream, p,            \
	                   (const Scalars<T>*)d->scalars, (const T*)d->bed, (const State4<T>*)src, (State4<T>*)dst,                     \
	                   (const T*)d->manning, (T*)d->cfl_slot, (const T*)d->cfl_slot + SLOT_EDGE + edge_buffer, tm, tail)
#define HP_LAUNCH_K2(UNIFORM_, TAIL_) HP_LAUNCH_K2S(UNIFORM_, TAIL_, false)
	// a speculative STRICT fp64 batch (spec_begin): the shared-reciprocal instantiation; single domains only (TAIL 0 / 1)
	constexpr bool CAN_SPEC = STRICT && sizeof(T) == 8;
	if (CAN_SPEC && d->spec_now && tail_kind != 2 && part == PART_ALL) {
		if (d->manning_uniform) { if (tail_kind == 1) HP_LAUNCH_K2S(true, 1, CAN_SPEC); else HP_LAUNCH_K2S(true, 0, CAN_SPEC); }
		else                    { if (tail_kind == 1) HP_LAUNCH_K2S(false, 1, CAN_SPEC); else HP_LAUNCH_K2S(false, 0, CAN_SPEC); }
	} else
	if (d->manning_uniform) { if (tail_kind == 2) HP_LAUNCH_K2(true, 2); else if (tail_kind == 1) HP_LAUNCH_K2(true, 1); else HP_LAUNCH_K2(true, 0); }
	else                    { if (tail_kind == 2) HP_LAUNCH_K2(false, 2); else if (tail_kind == 1) HP_LAUNCH_K2(false, 1); else HP_LAUNCH_K2(false, 0); }
#undef HP_LAUNCH_K2
#undef HP_LAUNCH_K2S
	HIP_TRY(hipGetLastError());
	return HP_OK;
}

template <typename T, bool STRICT, int CFL_MODE>
int launch_march(hp_domain* d, const void* src, void* dst, int edge_buffer, int part, hipStream_t stream)
{
	const Params<T> p = make_params<T>(d);
	TileMap tm;
	unsigned blocks;
	long lo, hi;
	launch_rows(d, 1, lo, hi);
	if (!make_tile_map(lo, hi, 1, part, (int)((p.cols - 2 + MARCH_COLS - 1) / MARCH_COLS), part == PART_ALL ? d->march_rseg : d->march_rseg_parts, d->tail_rseg, d->tail_pct, tm, blocks, d->tall_rseg,
	                   d->ghost_rows, d->own_lo, d->own_hi, d->march_nbands))
		return HP_OK;
	sweep_direction(d, part, tm);
	const int truncated = ((d->desc.quirks & HP_QUIRK_BDY_TRUNCATED) != 0 ? 1 : 0) | (d->fill_now ? 2 : 0);   // the kernel's `flags`

	LaunchTail<T> tail;
	const int tail_kind = make_tail<T>(d, blocks, part, stream, tail_limit(), tail);
#define HP_LAUNCH_K1S(FUSED_, TAIL_, LIST_, NEXT_, SPEC_)                                                                                   \
	hipLaunchKernelGGL((godunov_march<STRICT, CFL_MODE, FUSED_, TAIL_, T, SPEC_>), dim3(blocks), dim3(256), 0, stream, p,                   \
	                   (const Scalars<T>*)d->scalars, (const T*)d->bed, (const State4<T>*)src, (State4<T>*)dst,                     \
	                   (const T*)d->manning, (T*)d->cfl_slot, (const T*)d->cfl_slot + SLOT_EDGE + edge_buffer, tm, LIST_, NEXT_,   \
	                   truncated, tail)
#define HP_LAUNCH_K1(FUSED_, TAIL_, LIST_, NEXT_) HP_LAUNCH_K1S(FUSED_, TAIL_, LIST_, NEXT_, false)
	constexpr bool CAN_SPEC = STRICT && sizeof(T) == 8;       // a speculative STRICT fp64 batch: see launch_muscl
	if (CAN_SPEC && d->spec_now && !d->fusable && tail_kind != 2 && part == PART_ALL) {        // (never with fused boundaries: spec_wanted)
		if (tail_kind == 1) HP_LAUNCH_K1S(false, 1, (const AreaBdyList<T>*)nullptr, 0, CAN_SPEC);
		else                HP_LAUNCH_K1S(false, 0, (const AreaBdyList<T>*)nullptr, 0, CAN_SPEC);
	} else
#ifdef HP_KEEP_SPEC_FUSED
	// (the instantiation round 5 deleted -- the speculative flavour with FUSED boundaries, LAB_NOTES R5.13 -- kept buildable for the
	// study of what went wrong with it: a library built with -DHP_KEEP_SPEC_FUSED speculates on such domains too)
	if (CAN_SPEC && d->spec_now && d->fusable && tail_kind != 2 && part == PART_ALL) {
		if (tail_kind == 1) HP_LAUNCH_K1S(true, 1, (const AreaBdyList<T>*)d->fused_list, d->fuse_next, CAN_SPEC);
		else                HP_LAUNCH_K1S(true, 0, (const AreaBdyList<T>*)d->fused_list, d->fuse_next, CAN_SPEC);
	} else
#endif
	if (d->fusable) {
		if (tail_kind == 2)      HP_LAUNCH_K1(true, 2, (const AreaBdyList<T>*)d->fused_list, d->fuse_next);
		else if (tail_kind == 1) HP_LAUNCH_K1(true, 1, (const AreaBdyList<T>*)d->fused_list, d->fuse_next);
		else                     HP_LAUNCH_K1(true, 0, (const AreaBdyList<T>*)d->fused_list, d->fuse_next);
	} else {
		if (tail_kind == 2)      HP_LAUNCH_K1(false, 2, (const AreaBdyList<T>*)nullptr, 0);
		else if (tail_kind == 1) HP_LAUNCH_K1(false, 1, (const AreaBdyList<T>*)nullptr, 0);
		else                     HP_LAUNCH_K1(false, 0, (const AreaBdyList<T>*)nullptr, 0);
	}
#undef HP_LAUNCH_K1S
#undef HP_LAUNCH_K1
	HIP_TRY(hipGetLastError());
	return HP_OK;
}

template <typename T, bool STRICT, int CFL_MODE>
int launch_inertial(hp_domain* d, const void* src, void* dst, int edge_buffer, int part, hipStream_t stream)
{
	const Params<T> p = make_params<T>(d);
	TileMap tm;
	unsigned blocks;
	long lo, hi;
	launch_rows(d, 1, lo, hi);
	if (!make_tile_map(lo, hi, 1, part, (int)((p.cols - 2 + MARCH_COLS - 1) / MARCH_COLS), part == PART_ALL ? d->inertial_rseg : d->inertial_rseg_parts, d->tail_rseg, d->tail_pct, tm, blocks, d->tall_rseg,
	                   d->ghost_rows, d->own_lo, d->own_hi, d->inertial_nbands))
		return HP_OK;
	sweep_direction(d, part, tm);
	LaunchTail<T> tail;
	const int tail_kind = make_tail<T>(d, blocks, part, stream, tail_limit_k2k6(), tail);
#define HP_LAUNCH_K6(TAIL_)                                                                                                         \
	hipLaunchKernelGGL((inertial_march<STRICT, CFL_MODE, TAIL_, T>), dim3(blocks), dim3(256), 0, stream, p,                          \
	                   (const Scalars<T>*)d->scalars, (const T*)d->bed, (const State4<T>*)src, (State4<T>*)dst,                     \
	                   (const T*)d->manning, (T*)d->cfl_slot, (const T*)d->cfl_slot + SLOT_EDGE + edge_buffer, tm, tail)
	if (tail_kind == 2) HP_LAUNCH_K6(2); else if (tail_kind == 1) HP_LAUNCH_K6(1); else HP_LAUNCH_K6(0);
#undef HP_LAUNCH_K6
	HIP_TRY(hipGetLastError());
	return HP_OK;
}

template <typename T, bool STRICT>
int launch_flux(hp_domain* d, const void* src, void* dst, int cfl_mode, int part, hipStream_t stream)
{
	const Params<T> p = make_params<T>(d);
	if (d->desc.scheme == HP_SCHEME_MUSCL_HANCOCK) {
		if (p.cols < 5 || p.rows < 5) return fail(HP_ERR_INVALID, "MUSCL-Hancock needs at least a 5x5 grid");
		return cfl_mode ? launch_muscl<T, STRICT, 1>(d, src, dst, d->use_alt ^ 1, part, stream)
		                : launch_muscl<T, STRICT, 0>(d, src, dst, 0, part, stream);
	}
	if (d->desc.scheme == HP_SCHEME_INERTIAL) {
		if (p.cols < 3 || p.rows < 3) return fail(HP_ERR_INVALID, "the inertial scheme needs at least a 3x3 grid");
		switch (cfl_mode) {
		case 1:  return launch_inertial<T, STRICT, 1>(d, src, dst, d->use_alt ^ 1, part, stream);
		case 2:  return launch_inertial<T, STRICT, 2>(d, src, dst, d->use_alt, part, stream);
		default: return launch_inertial<T, STRICT, 0>(d, src, dst, 0, part, stream);
		}
	}
	if (d->desc.kernel == HP_KERNEL_BASIC) {
		if (part == PART_INTERIOR) return HP_OK;                          // no split for the cross-check kernel
		long lo, hi;
		launch_rows(d, 1, lo, hi);
		const dim3 block(64, 4), grid = grid2d(p.cols, hi - lo, block);
		hipLaunchKernelGGL((godunov_basic<STRICT, T>), grid, block, 0, stream, p, (const Scalars<T>*)d->scalars,
		                   (const T*)d->bed, (const State4<T>*)src, (State4<T>*)dst, (const T*)d->manning, lo, hi);
		HIP_TRY(hipGetLastError());
		return HP_OK;
	}
	// ring of the buffer that is priced: mode 1 -> dst, mode 2 -> src (= primary)
	switch (cfl_mode) {
	case 1:  return launch_march<T, STRICT, 1>(d, src, dst, d->use_alt ^ 1, part, stream);
	case 2:  return launch_march<T, STRICT, 2>(d, src, dst, d->use_alt, part, stream);
	default: return launch_march<T, STRICT, 0>(d, src, dst, 0, part, stream);
	}
}

static bool pairs_possible(const hp_domain* d);
// boundaries -> flux -> local CFL maximum   (CSchemeGodunov.cpp:1637-1657)
template <typename T, bool STRICT> int step_begin_impl(hp_domain* d)
{
	void* src = d->state[d->use_alt];
	void* dst = d->state[d->use_alt ^ 1];
	int rc;
	const bool has_bdy = !d->bdy.empty() && d->desc.scheme != HP_SCHEME_MUSCL_HANCOCK;   // MUSCL never applies them (Q8)
	if (has_bdy) {
		d->fork_is_advance = false;
		if ((rc = apply_boundaries<T>(d, src)) != HP_OK) return rc;
	}

	// Which buffer does the CFL reduction price?  Q1: always the primary one (CSchemeGodunov.cpp:1629/:1634);
	// otherwise what this iteration writes.
	const bool q1 = (d->desc.quirks & HP_QUIRK_CFL_READS_PRIMARY) != 0;
	const bool muscl = d->desc.scheme == HP_SCHEME_MUSCL_HANCOCK;
	const bool basic = d->desc.kernel == HP_KERNEL_BASIC && d->desc.scheme == HP_SCHEME_GODUNOV;
	const bool dst_is_primary = d->use_alt == 1;
	int cfl_mode = 0;                                    // fused epilogue of the tuned kernel
	if (d->desc.dynamic_dt && !basic) {
		// MUSCL-Hancock has ONE state buffer in the reference (updated in place), so its reduction always sees
		// the new state: here that is dst, whichever physical buffer it is
		if (muscl || !q1 || dst_is_primary) cfl_mode = 1; // price what lands in dst
		else if (has_bdy)          cfl_mode = 2;         // primary = source, changed in place by the boundaries
		else                       cfl_mode = 0;         // primary untouched: last maximum still holds
		// the ring maxima are cached from the last upload -- unless a cell boundary rewrites ring cells in place
		// (bdy_cell accepts any cell; tst_Reduce re-reads every cell each iteration, CLDynamicTimestep.clc:166-249)
		if (cfl_mode != 0 && (d->edge_dirty || (has_bdy && d->bdy_on_ring)))
			{ d->fork_is_advance = false; if ((rc = price_edge_ring<T>(d)) != HP_OK) return rc; }
	}

	// flux-kernel timing: every `stride`-th launch is bracketed by a pair of events taken from a pool created by
	// hp_kernel_timing (nothing is created inside a timed region; once the pool is used up sampling stops)
	// (a domain whose batches are made of iteration pairs samples THOSE launches -- run_pair -- not the odd single iteration at a
	// batch's ends: hp_kernel_timing reports one kernel's average, the dominant one's)
	// (... while pairs really run: a domain that COULD pair but does not -- batches of one iteration, a launch too big for the tail
	// block, a maximum that stays stale -- is seen by its run of single iterations and samples them)
	d->single_streak++;
	const bool sample = d->timing_stride > 0 && d->timing_used < d->timing_events.size() && (!pairs_possible(d) || d->single_streak > 3) &&
	                    (d->timing_counter++ % (uint64_t)d->timing_stride) == 0;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	if (sample) {
		e0 = d->timing_events[d->timing_used].first;
		e1 = d->timing_events[d->timing_used].second;
		HIP_TRY(hipEventRecord(e0, d->stream));
	}
	// The launch's own tail block instead of a separate advance launch (LaunchTail): inside a batch call, the tuned Godunov
	// kernel, one launch over all rows, nothing between the flux launch and the advance (no stand-alone reduction), and
	// either no reduction over strips at all or the strips' own transport in its plain shape (no rank with boundaries or a
	// stale maximum: those iterations pass remembered maxima around, which stays with the advance launch).
	{
		static const bool enabled = !(std::getenv("HP_LAUNCH_TAIL") && std::atoi(std::getenv("HP_LAUNCH_TAIL")) == 0);
		static const bool lonely_too = std::getenv("HP_STRIP_REDUCE_ALWAYS") && std::atoi(std::getenv("HP_STRIP_REDUCE_ALWAYS")) != 0;
		const bool reduce_after = d->desc.dynamic_dt && cfl_mode == 0 && (basic || d->need_full_reduce);
		const int priced = (d->desc.dynamic_dt && cfl_mode != 0) ? 1 : 0;
		// strips: the ranks reduce on the iterations strip_allreduce_max says (`everyone`); the tail block serves the two plain
		// shapes -- this rank priced and everyone reduces, or nobody does -- and leaves the rest (a rank that passes its
		// REMEMBERED maximum around: boundaries or a stale maximum somewhere else) to the advance launch
		bool strips_ok = d->comm_world <= 1 ? !(d->comm && lonely_too) : d->peer_direct;
		int fresh = priced;
		if (d->comm_world > 1 && d->peer_direct) {
			const bool everyone = d->desc.dynamic_dt && (muscl || !q1 || dst_is_primary || basic || d->strip_any_bdy ||
			                                             (d->strip_first && d->strip_any_full));
			strips_ok = (priced != 0) == everyone;
			fresh = everyone ? 7 : 4;
		}
		d->tail_want = enabled && d->tail_allowed && !basic && !reduce_after && strips_ok &&
		               !(d->halo_overlap && d->split_now) && d->tail_words != nullptr;
		d->tail_fresh = fresh;
		d->tail_done = false;
	}
	if (d->halo_overlap && d->split_now) {
		// fork: everything queued so far (previous advance, boundaries, ring pricing) happens-before the halo
		// segments; they run on their own stream, next to the interior segments on the domain's stream
		// (when nothing was queued since the previous advance_time, that kernel's own completion event serves:
		// a separate marker packet in front of the interior launch costs a few microseconds on the critical path)
		if (!d->fork_is_advance) HIP_TRY(hipEventRecord(d->ev_fork, d->stream));
		d->fork_is_advance = false;
		HIP_TRY(hipStreamWaitEvent(d->stream_halo, d->ev_fork, 0));
		if ((rc = launch_flux<T, STRICT>(d, src, dst, cfl_mode, PART_HALO, d->stream_halo)) != HP_OK) return rc;
		HIP_TRY(hipEventRecord(d->ev_halo, d->stream_halo));
		if ((rc = launch_flux<T, STRICT>(d, src, dst, cfl_mode, PART_INTERIOR, d->stream)) != HP_OK) return rc;
		// join: the CFL maximum (and anything after it on the domain's stream) needs both launches
		HIP_TRY(hipStreamWaitEvent(d->stream, d->ev_halo, 0));
	} else if ((rc = launch_flux<T, STRICT>(d, src, dst, cfl_mode, PART_ALL, d->stream)) != HP_OK) return rc;
	d->tail_want = false;
	if (sample) {
		HIP_TRY(hipEventRecord(e1, d->stream));
		d->timing_used++;
	}

	d->adv_fresh = 0;
	if (d->desc.dynamic_dt) {
		if (cfl_mode != 0) {
			d->adv_fresh = 1;
			d->need_full_reduce = false;
		} else if (basic || d->need_full_reduce) {
			const void* reduce_buf = q1 ? d->state[0] : dst;
			if ((rc = launch_reduce<T>(d, reduce_buf, d->own_lo, d->own_hi)) != HP_OK) return rc;
			d->adv_fresh = 1;
			d->need_full_reduce = false;
		}
	}
	return HP_OK;
}

// the ghost rows an advance kernel (or a launch's tail block) carries to the strip neighbours: the first / last `ghost_rows`
// owned rows of the state this iteration wrote go into the neighbours' ghost rows of the same ping-pong buffer
PeerPush make_push(const hp_domain* d)
{
	const long G = d->ghost_rows, rows = d->desc.rows;
	const size_t row_bytes = (size_t)d->desc.cols * 4 * d->esize;
	const int b = d->use_alt ^ 1;
	const char* mine = (const char*)d->state[b];
	PeerPush push{};
	push.count = (unsigned)((size_t)G * row_bytes / 16);
	push.arrived = d->push_arrived;
	if (d->peer_state[0][b]) {
		push.from[0] = (const uint4*)(mine + (size_t)G * row_bytes);
		push.to[0] = (uint4*)((char*)d->peer_state[0][b] + (size_t)(d->peer_rows[0] - G) * row_bytes);
	}
	if (d->peer_state[1][b]) {
		push.from[1] = (const uint4*)(mine + (size_t)(rows - 2 * G) * row_bytes);
		push.to[1] = (uint4*)d->peer_state[1][b];
	}
	return push;
}

// kernel argument of a reduction through the mailboxes (one per reduction: its parity picks the mailbox set)
static long peer_timeout_ms()
{
	static const long ms = std::getenv("HP_PEER_TIMEOUT_MS") ? std::atol(std::getenv("HP_PEER_TIMEOUT_MS")) : 10000;
	return ms > 0 ? ms : 10000;
}
PeerBox peer_box(hp_domain* d, bool use, long timeout_ms)
{
	PeerBox box{};
	if (!use) return box;
	box.mine = d->peer_mine; box.peer = d->peer_table;
	box.world = d->peer_world; box.rank = d->peer_rank;
	box.set = (int)(d->peer_rounds++ & 1);
	box.timeout = (unsigned long long)(timeout_ms > 0 ? timeout_ms : peer_timeout_ms()) * 100000ull;   // wall_clock64: 100 MHz
	return box;
}

template <typename T> int step_end_impl(hp_domain* d)
{
	const Params<T> p = make_params<T>(d);
	const PeerBox box = d->tail_done ? PeerBox{} : peer_box(d, (d->adv_fresh & 4) != 0);   // (the tail block took its own: one per round)
	if (d->tail_done) {
		// the flux launch's tail block has done it all (LaunchTail): nothing to launch
		d->tail_done = false;
		d->push_now = false;
		d->fork_is_advance = false;
	} else if (d->push_now) {
		// the ghost rows travel inside the advance kernel (PeerPush)
		const PeerPush push = make_push(d);
		static const unsigned per_block = std::getenv("HP_PUSH_PER_BLOCK") ? (unsigned)std::atoi(std::getenv("HP_PUSH_PER_BLOCK")) : 1024u;
		const unsigned blocks = (push.to[0] || push.to[1]) ? std::min(32u, (push.count + per_block - 1u) / per_block) : 1u;
		hipLaunchKernelGGL((advance_time<false, T>), dim3(blocks), dim3(256), 0, d->stream, p, (Scalars<T>*)d->scalars,
		                   (T*)d->cfl_slot, d->adv_fresh, box, push);
		d->push_now = false;
		d->fork_is_advance = false;
	} else if (d->halo_overlap && !d->peer_direct) {   // (also after an unsplit iteration: the NEXT one may be split and fork from here)
		hipExtLaunchKernelGGL((advance_time<false, T>), dim3(1), dim3(64), 0, d->stream, nullptr, d->ev_fork, 0, p,
		                      (Scalars<T>*)d->scalars, (T*)d->cfl_slot, d->adv_fresh, box, PeerPush{});
		d->fork_is_advance = true;                                        // cleared by anything else queued on the stream
	} else {
		hipLaunchKernelGGL((advance_time<false, T>), dim3(1), dim3(64), 0, d->stream, p, (Scalars<T>*)d->scalars,
		                   (T*)d->cfl_slot, d->adv_fresh, box, PeerPush{});
	}
	HIP_TRY(hipGetLastError());
	d->use_alt ^= 1;                                                      // Threaded_runBatch :1300
	d->cells_calculated += (uint64_t)d->desc.cols * (uint64_t)d->desc.rows;   // :1299
	d->iterations += 1;
	return HP_OK;
}

// ---- two iterations per pass (hp_kernels.hpp: godunov_march2) -------------------------------------------------------------------
// A pair is two consecutive iterations (k reads the primary buffer, k + 1 writes it) of a batch on a single domain run by ONE
// launch.  Eligible: Godunov scheme, tuned kernel, FAST arithmetic, no boundary conditions (they act between the two iterations),
// no strips, the reference's reduction quirk Q1 on (it is what makes the second timestep known in advance), the remembered maximum
// valid, the launch's own tail block available (it advances the time twice), and the next iteration reads the primary buffer.
// HP_TWO_STEP=0 switches it off, =1 forces it on wherever it is eligible; default: grids whose launch is at least two rounds of
// blocks at 12-row tiles (hp_domain_create: march2_pays -- below that the extra halo rows and the shorter tiles cost more than the
// bytes save: 1024^2 and the 4096 x 514 strip lose 3-5 %, 2048^2 gains 13 %, 4096^2 22 %, profiles/r05n_two_step.txt).
static int two_step_mode()
{
	static const int v = std::getenv("HP_TWO_STEP") ? std::atoi(std::getenv("HP_TWO_STEP")) : -1;
	return v;
}
// (the part of the test that does not change from one iteration to the next: this domain's batches are made of pairs wherever two
// iterations are to be had)
static bool pairs_possible_common(const hp_domain* d)
{
	const int mode = two_step_mode();
	if (mode == 0 || (mode < 0 && !d->march2_pays)) return false;
	static const bool tail_enabled = !(std::getenv("HP_LAUNCH_TAIL") && std::atoi(std::getenv("HP_LAUNCH_TAIL")) == 0);
	// boundary conditions: none, or area boundaries the flux kernel can carry itself (`fusable`: uniform / coarse gridded ones; round 6) --
	// the pair kernel applies them between its two steps and prices the state it stores with and without the next iteration's
	// (godunov_march2, BDY); single domains only.  HP_PAIR_BDY=0 keeps such domains on single iterations (A/B runs).
	static const bool bdy_enabled = !(std::getenv("HP_PAIR_BDY") && std::atoi(std::getenv("HP_PAIR_BDY")) == 0);
	// (fp32: from 30 M cells on -- S-RAIN 4096^2 fp32 LOSES 3 % in pairs, 0.1436 -> 0.1478 ms, where fp64 gains 7.5 % and 8192^2 gains 8-9 %
	// in both precisions: profiles/r06w_srain_pairs_by_size.txt; with five waves per SIMD, the final binary: level at 16.8 M and 25 M cells,
	// +5 % at 37.7 M, +13 % at 67 M: profiles/r06ah_f32_bdy_pairs_by_size.txt; HP_TWO_STEP=1 overrides)
	const bool bdy_size_ok = d->desc.precision == 8 || d->cells >= 30000000 || mode > 0;
	const bool bdy_ok = d->bdy.empty() || (bdy_enabled && bdy_size_ok && d->fusable && !d->comm && d->comm_world <= 1 && d->desc.dynamic_dt);
	// STRICT (round 6): the same statements in the same order as K1's, so the pair is the same computation there too -- always with
	// the stamps (the exact mode promises the reference's bits); no boundaries, single domains, dynamic timestep.  What it is for: rows
	// of still water are a copy in STRICT (K1's skip), and a pair copies them at half the bytes.  HP_PAIR_STRICT=0 switches it off.
	static const bool strict_enabled = !(std::getenv("HP_PAIR_STRICT") && std::atoi(std::getenv("HP_PAIR_STRICT")) == 0);
	const bool math_ok = d->desc.math_mode == HP_MATH_FAST ||
	                     (strict_enabled && d->bdy.empty() && !d->comm && d->comm_world <= 1 && d->desc.dynamic_dt && !d->spec_now);
	return d->desc.scheme == HP_SCHEME_GODUNOV && d->desc.kernel != HP_KERNEL_BASIC && math_ok &&
	       bdy_ok && (!d->desc.dynamic_dt || (d->desc.quirks & HP_QUIRK_CFL_READS_PRIMARY) != 0) &&
	       tail_enabled && d->tail_words != nullptr && d->desc.rows >= 5 && d->desc.cols >= 5;
}
static bool pairs_possible(const hp_domain* d)          // a single domain
{
	return pairs_possible_common(d) && !d->comm && d->comm_world <= 1 && !d->peer_mine && d->desc.row_offset == 0 &&
	       d->desc.global_rows == d->desc.rows;
}
// After partial uploads (rings_differ: the two buffers' edge rings MAY differ) the rings are compared once, the first time pairs are
// wanted: block-wise loads of a whole grid (hp_domain_upload_rows over all rows: how grids too big for one host array arrive) leave
// them equal to the zero-filled other buffer's wherever the ring is the usual closed wall (state 0), and such domains pair.  Blocks.
constexpr size_t HOST_RINGS = 496;          // byte offset in the pinned block
static int rings_really_differ(hp_domain* d)
{
	if (!d->rings_differ || d->rings_checked) return HP_OK;
	unsigned long long* word = (unsigned long long*)((char*)d->scalars + 192);     // (device scratch behind the Scalars block)
	HIP_TRY(hipMemsetAsync(word, 0, 8, d->stream));
	if (d->desc.precision == 8)
		hipLaunchKernelGGL((rings_compare<double>), dim3(32), dim3(256), 0, d->stream, (const State4<double>*)d->state[0],
		                   (const State4<double>*)d->state[1], (long)d->desc.cols, (long)d->desc.rows, word);
	else
		hipLaunchKernelGGL((rings_compare<float>), dim3(32), dim3(256), 0, d->stream, (const State4<float>*)d->state[0],
		                   (const State4<float>*)d->state[1], (long)d->desc.cols, (long)d->desc.rows, word);
	HIP_TRY(hipGetLastError());
	volatile unsigned long long* host = (volatile unsigned long long*)((char*)d->host_scalars + HOST_RINGS);
	*host = 1;
	HIP_TRY(hipMemcpyAsync((void*)host, word, 8, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(hipStreamSynchronize(d->stream));
	d->rings_checked = true;
	if (*host == 0) d->rings_differ = false;
	return HP_OK;
}
// (not after a PARTIAL upload -- rings_differ, until the next full one: queueWritePartial goes to the current buffer only, so the two
// buffers' edge rings may differ, and single iterations read the OTHER buffer's ring on every second iteration while a pair carries the
// primary buffer's through both steps: next to ring cells that matter hydraulically -- an edge without walls -- the two part.  ADVICE r05)
static bool pair_eligible(const hp_domain* d)
{
	return pairs_possible(d) && d->use_alt == 0 && !d->rings_differ && (!d->desc.dynamic_dt || (!d->need_full_reduce && !d->edge_dirty));
}
// A ROW STRIP runs pairs where it stores two reaches of ghost rows (one exchange per two iterations anyway) and the strips write
// their rows into each other themselves (transport level 2: the pair's tail block holds the one mailbox round, which is the
// hand-over).  What THIS rank can do goes into the batch's handshake (strip_handshake); the ranks pair up only if all of them can.
static bool strip_pairs_possible_here(const hp_domain* d)
{
	return pairs_possible_common(d) && d->comm && d->comm_world > 1 && d->peer_direct && d->ghost_rows == 2 && !d->rings_differ;
}
// The stamps' buffers (PairAux), allocated with a domain's first exact pair.
// Which pairs keep quirk Q3 exact (godunov_march2's HZ instantiation: 4-9 % slower): HP_PAIR_EXACT=1 all of them, =0 none; default: the
// domains whose area boundaries can REMOVE water -- a loss rate, a mass flux -- where whole regions dry at once and sit untouched, with
// stale values in the reference's other buffer, for as long as they stay dry.
static bool pair_exact(const hp_domain* d)
{
	if (d->desc.math_mode == HP_MATH_STRICT) return true;
	static const int forced = std::getenv("HP_PAIR_EXACT") ? (std::atoi(std::getenv("HP_PAIR_EXACT")) != 0 ? 1 : 0) : -1;
	if (forced >= 0) return forced != 0;
	for (const Boundary& b : d->bdy)
		if ((b.kind == 0 && b.definition == HP_UNIFORM_LOSS_RATE) || (b.kind == 1 && b.definition == HP_GRIDDED_MASS_FLUX)) return true;
	return false;
}
static int pair_stamps_alloc(hp_domain* d)
{
	if (!pair_exact(d) || d->z_state) return HP_OK;
	const size_t rec = (size_t)4 * d->esize + 16;                         // stamp_rec<T>(): the state, then the launch's number
	HIP_TRY(hipMalloc(&d->z_state, d->cells * rec));
	HIP_TRY(hipMalloc((void**)&d->haz_words, 64));
	HIP_TRY(hipMemsetAsync(d->z_state, 0, d->cells * rec, d->stream));
	HIP_TRY(hipMemsetAsync(d->haz_words, 0, 64, d->stream));
	return HP_OK;
}
// Area boundaries: the pair kernel needs the primary buffer priced WITH the first iteration's boundaries (slot[SLOT_M1]).  Every BDY
// pair leaves that figure for its successor; where there is no such predecessor -- the first pair after single iterations, an upload,
// a new target time -- the first half of iteration k is done the reference's own way: the stand-alone boundary pass on the primary
// buffer (it declines by itself if a fused store has applied them), the stand-alone reduction, and the result moved to SLOT_M1.
template <typename T> int pair_cold_start(hp_domain* d)
{
	int rc;
	if ((rc = apply_boundaries<T>(d, d->state[0])) != HP_OK) return rc;
	if ((rc = launch_reduce<T>(d, d->state[0], d->own_lo, d->own_hi)) != HP_OK) return rc;
	hipLaunchKernelGGL((pair_cold_start_words<T>), dim3(1), dim3(1), 0, d->stream, (T*)d->cfl_slot);
	HIP_TRY(hipGetLastError());
	d->m1_valid = true;
	d->pair_cold_starts++;
	return HP_OK;
}
template <typename T> int run_pair_t(hp_domain* d, const bool strip, const bool followed)
{
	const Params<T> p = make_params<T>(d);
	TileMap tm;
	unsigned blocks;
	long lo, hi;
	if (strip) {                                                          // the strip's OWNED rows: the pair consumes both reaches of ghost rows
		const bool south = d->desc.row_offset > 0, north = d->desc.row_offset + d->desc.rows < d->desc.global_rows;
		lo = south ? d->ghost_rows : 1;
		hi = d->desc.rows - (north ? d->ghost_rows : 1);
	} else launch_rows(d, 1, lo, hi);
	// (STRICT: 12-row tiles -- its live tiles cost 1300 instructions per row and stage, and a 24-row one is what the launch waits for at its
	// end: S-DAM 4096^2 0.275 ms per iteration at 24 rows, 0.2645 at 12, 0.281 at 8; profiles/r06i_strict_pair_tile_sweep.txt)
	static const bool rseg_forced = std::getenv("HP_MARCH2_RSEG") != nullptr;
	const int rseg2 = (d->desc.math_mode == HP_MATH_STRICT && !rseg_forced) ? std::min(d->march2_rseg, 12) : d->march2_rseg;
	if (!make_tile_map(lo, hi, 1, PART_ALL, (int)((p.cols - 2 + MARCH2_COLS - 1) / MARCH2_COLS), rseg2, rseg2, 0, tm, blocks, rseg2,
	                   0, d->own_lo, d->own_hi, d->march2_nbands))
		return HP_ERR_STATE;
	if (blocks > tail_limit()) return HP_ERR_STATE;                      // (the caller falls back to single iterations)
	d->tail_want = true; d->tail_allowed = true; d->push_now = strip;    // (strip: the final rows of the edge ranges leave with the launch)
	// single domain: the second iteration prices the primary buffer (1); a strip: ... and every rank reduces, through the mailboxes,
	// whose round is also the hand-over of the rows (7; fixed timestep: the round alone, 4) -- step_begin_impl's `fresh`
	d->tail_fresh = strip ? (d->desc.dynamic_dt ? 7 : 4) : (d->desc.dynamic_dt ? 1 : 0);
	d->tail_done = false;
	LaunchTail<T> tail;
	const int kind = make_tail<T>(d, blocks, PART_ALL, d->stream, tail_limit(), tail);
	d->tail_want = false; d->tail_allowed = false; d->push_now = false;
	if (kind != (strip ? 2 : 1)) return HP_ERR_STATE;
	const bool bdy = !d->bdy.empty();
	int rc;
	if ((rc = pair_stamps_alloc(d)) != HP_OK) return rc;
	bool in_place = false;
	if (bdy && !d->m1_valid) {                                            // (queued in front of the pair launch: same stream)
		if ((rc = pair_cold_start<T>(d)) != HP_OK) return rc;
		in_place = true;
	} else if (bdy) in_place = d->pair_fused_next;                        // the pair before stored this iteration's boundaries with its state
	tail.pair = bdy ? 2 : 1;
	tail.bdy_flag = bdy && followed ? 1 : 0;
	PairAux<T> aux{};
	const bool exact = pair_exact(d) && d->z_state != nullptr;
	if (exact) aux.stamps = StampBufs<T>{(char*)d->z_state, d->haz_words};
	aux.prev_gen = d->other_stale ? d->pair_gen : 0u;                     // `src` was written by a pair launch: its stamps apply
	aux.gen = ++d->pair_gen;
	if (aux.gen == 0) aux.gen = ++d->pair_gen;                            // (0 means "no launch")
	aux.list = (const AreaBdyList<T>*)d->fused_list; aux.fuse_next = bdy && followed ? 1 : 0; aux.in_place = in_place ? 1 : 0;
	aux.truncated = (d->desc.quirks & HP_QUIRK_BDY_TRUNCATED) != 0 ? 1 : 0;
	const void* src = d->state[0];
	void* dst = d->state[1];
	// flux-kernel timing (hp_kernel_timing): as in step_begin_impl -- a sampled launch here covers two iterations
	const bool sample = d->timing_stride > 0 && d->timing_used < d->timing_events.size() &&
	                    (d->timing_counter++ % (uint64_t)d->timing_stride) == 0;
	if (sample) HIP_TRY(hipEventRecord(d->timing_events[d->timing_used].first, d->stream));
#define HP_LAUNCH_K1B2(CFL_, BDY_, HZ_, TAIL_)                                                                                            \
	hipLaunchKernelGGL((godunov_march2<false, CFL_, BDY_, HZ_, TAIL_, T>), dim3(blocks), dim3(256), 0, d->stream, p, (const Scalars<T>*)d->scalars, \
	                   (const T*)d->bed, (const State4<T>*)src, (State4<T>*)dst, (const T*)d->manning, (T*)d->cfl_slot,                    \
	                   (const T*)d->cfl_slot + SLOT_EDGE, tm, tail, aux)
#define HP_LAUNCH_K1B(CFL_, BDY_, TAIL_) do { if (exact) HP_LAUNCH_K1B2(CFL_, BDY_, true, TAIL_); else HP_LAUNCH_K1B2(CFL_, BDY_, false, TAIL_); } while (0)
	if (d->desc.math_mode == HP_MATH_STRICT)
		hipLaunchKernelGGL((godunov_march2<true, 1, false, true, 1, T>), dim3(blocks), dim3(256), 0, d->stream, p, (const Scalars<T>*)d->scalars,
		                   (const T*)d->bed, (const State4<T>*)src, (State4<T>*)dst, (const T*)d->manning, (T*)d->cfl_slot,
		                   (const T*)d->cfl_slot + SLOT_EDGE, tm, tail, aux);
	else
	if (strip)    { if (d->desc.dynamic_dt) HP_LAUNCH_K1B2(1, false, false, 2); else HP_LAUNCH_K1B2(0, false, false, 2); }     // (strips: no stamps -- a ghost row's would be the neighbour's to write)
	else if (bdy) HP_LAUNCH_K1B(1, true, 1);                              // (area boundaries pair with a dynamic timestep only: pairs_possible_common)
	else          { if (d->desc.dynamic_dt) HP_LAUNCH_K1B(1, false, 1); else HP_LAUNCH_K1B(0, false, 1); }
#undef HP_LAUNCH_K1B
#undef HP_LAUNCH_K1B2
	d->pair_fused_next = bdy && followed;
	d->m1_valid = bdy;
	HIP_TRY(hipGetLastError());
	if (sample) { HIP_TRY(hipEventRecord(d->timing_events[d->timing_used].second, d->stream)); d->timing_used++; }
	// the pass wrote state k + 2 into the other buffer: that buffer IS the primary one from here on (two single iterations would
	// have left the newest state in the primary buffer; the two buffers' edge rings are equal -- pair_eligible -- and so are their maxima)
	std::swap(d->state[0], d->state[1]);
	if (strip) {                                                          // every rank swaps: a neighbour's buffer b is the one it calls b now
		std::swap(d->peer_state[0][0], d->peer_state[0][1]);
		std::swap(d->peer_state[1][0], d->peer_state[1][1]);
		d->ghost_valid = d->ghost_rows;                                   // the exchange happened inside the launch
	}
	d->other_stale = true;
	d->single_streak = 0;
	d->tail_done = false; d->fork_is_advance = false;
	d->adv_fresh = d->desc.dynamic_dt ? 1 : 0;
	d->cells_calculated += 2 * (uint64_t)d->desc.cols * (uint64_t)d->desc.rows;
	d->iterations += 2;
	d->pairs += 1;
	return HP_OK;
}
// `followed`: another iteration of the same batch comes after the pair (area boundaries: the pair stores its state with that iteration's
// boundaries applied, as K1's fused epilogue does; a batch's last launch never does -- what a download sees is the reference's buffer)
static int run_pair(hp_domain* d, const bool strip = false, const bool followed = false)
{
	return d->desc.precision == 8 ? run_pair_t<double>(d, strip, followed) : run_pair_t<float>(d, strip, followed);
}
// Before a single-iteration kernel builds on the non-current buffer again (it leaves all-dry cells of its destination untouched,
// quirk Q3): that buffer holds a state two iterations old after pairs -- bring it up to date with one device copy.
static int repair_other_buffer(hp_domain* d)
{
	if (!d->other_stale) return HP_OK;
	// (the edge rings of the two buffers are equal whenever pairs have run: pair_eligible, hp_domain_upload_rows)
	HIP_TRY(hipMemcpyAsync(d->state[d->use_alt ^ 1], d->state[d->use_alt], d->cells * 4 * d->esize, hipMemcpyDeviceToDevice, d->stream));
	if (d->z_state) {                                                     // ... except where the last pair launch stamped a different value (PairAux)
		if (d->desc.precision == 8)
			hipLaunchKernelGGL((stamps_to_buffer<double>), dim3(1024), dim3(256), 0, d->stream, (State4<double>*)d->state[d->use_alt ^ 1],
			                   StampBufs<double>{(char*)d->z_state, d->haz_words}, d->pair_gen, d->cells);
		else
			hipLaunchKernelGGL((stamps_to_buffer<float>), dim3(1024), dim3(256), 0, d->stream, (State4<float>*)d->state[d->use_alt ^ 1],
			                   StampBufs<float>{(char*)d->z_state, d->haz_words}, d->pair_gen, d->cells);
		HIP_TRY(hipGetLastError());
	}
	d->other_stale = false;
	return HP_OK;
}

int dispatch_begin(hp_domain* d)
{
	// after iteration pairs the non-current buffer is out of date; K1 brings it up to date by itself (its FILL flag: every cell of the
	// launch's rows is stored), any other kernel gets the device copy first
	static const bool fill_enabled = !(std::getenv("HP_FILL_AFTER_PAIRS") && std::atoi(std::getenv("HP_FILL_AFTER_PAIRS")) == 0);
	// (exact pairs: the device copy, then the stamps on top -- repair_other_buffer -- so that K1 needs no knowledge of them)
	const bool fill = fill_enabled && d->other_stale && d->desc.scheme == HP_SCHEME_GODUNOV && d->desc.kernel != HP_KERNEL_BASIC && !(pair_exact(d) && d->z_state);
	if (!fill) { const int rc0 = repair_other_buffer(d); if (rc0 != HP_OK) return rc0; }
	d->fill_now = fill;
	const bool strict = d->desc.math_mode == HP_MATH_STRICT;
	const int rc = d->desc.precision == 8 ? (strict ? step_begin_impl<double, true>(d) : step_begin_impl<double, false>(d))
	                                      : (strict ? step_begin_impl<float, true>(d) : step_begin_impl<float, false>(d));
	d->fill_now = false;
	if (fill && rc == HP_OK) d->other_stale = false;
	return rc;
}

int dispatch_end(hp_domain* d)
{
	return d->desc.precision == 8 ? step_end_impl<double>(d) : step_end_impl<float>(d);
}

template <typename T> int write_scalars_initial(hp_domain* d)
{
	Scalars<T> s;
	s.t = T(0); s.dt = (T)d->desc.dt_initial; s.t_hydro = T(0); s.t_sync = T(0); s.batch_dt = T(0);
	s.batch_ok = 0; s.batch_skipped = 0;                                  // CSchemeGodunov.cpp:805-867
	HIP_TRY(hipMemcpy(d->scalars, &s, sizeof s, hipMemcpyHostToDevice));
	return HP_OK;
}

// One time-control scalar rewritten in stream order, without blocking: the value travels as a kernel argument, so
// no host staging buffer has to outlive the call (the reference enqueues these writes non-blocking too,
// CSchemeGodunov.cpp:1166-1176, :1213-1232).
template <typename T> __global__ void store_scalar(T* where, const T value) { *where = value; }

template <typename T> int set_scalar_field(hp_domain* d, size_t offset, double value)
{
	hipLaunchKernelGGL(store_scalar<T>, dim3(1), dim3(1), 0, d->stream, (T*)((char*)d->scalars + offset), (T)value);
	HIP_TRY(hipGetLastError());
	return HP_OK;
}


// Which area boundaries the flux kernel carries itself (K1 FUSED, hp_kernels.hpp): Godunov scheme, tuned kernel, nothing but
// uniform / gridded boundaries (a cell boundary in between has its place in the order added), at most FUSED_BDY_MAX of
// them, and rain grids coarse enough for a tile to meet at most two grid rows and a wavefront two grid columns.  Everything
// else keeps the stand-alone pass.  HP_FUSE_BDY=0 switches the fusion off (A/B runs).
template <typename T> int refresh_fusable_t(hp_domain* d)
{
	static const bool enabled = !(std::getenv("HP_FUSE_BDY") && std::atoi(std::getenv("HP_FUSE_BDY")) == 0);
	d->fusable = false;
	if (!enabled || d->desc.scheme != HP_SCHEME_GODUNOV || d->desc.kernel == HP_KERNEL_BASIC) return HP_OK;
	if (d->bdy.empty() || d->bdy.size() > (size_t)FUSED_BDY_MAX) return HP_OK;
	AreaBdyList<T> list;
	list.count = 0;
	for (const Boundary& b : d->bdy) {
		if (b.kind == 2) return HP_OK;
		AreaBdy<T>& a = list.b[list.count++];
		a = AreaBdy<T>{};
		if (b.kind == 0) {
			a.kind = 0;
			a.u = UniformBdy<T>{(const T*)b.data, (uint32_t)b.entries, b.definition, (T)b.interval, (T)b.length};
		} else {
			if (b.resolution < 64.0 * d->desc.dx) return HP_OK;
			a.kind = 1;
			a.g = GriddedBdy<T>{(const T*)b.data, b.entries, b.grows, b.gcols, b.definition,
			                    (T)b.resolution, (T)b.off_x, (T)b.off_y, (T)b.interval};
		}
	}
	if (!d->fused_list) HIP_TRY(hipMalloc(&d->fused_list, sizeof(AreaBdyList<double>)));
	// the flux launches of a batch still in flight read this list: wait for them before it changes (hp_step_batch returns
	// without waiting, and a blocking copy is not ordered behind a non-blocking stream -- a boundary added right after a
	// batch call used to rain on that batch's remaining iterations; found by the fuzz's boundary-added-mid-run cases)
	HIP_TRY(hipStreamSynchronize(d->stream));
	if (d->stream_halo) HIP_TRY(hipStreamSynchronize(d->stream_halo));
	HIP_TRY(hipMemcpy(d->fused_list, &list, sizeof list, hipMemcpyHostToDevice));
	d->fusable = true;
	return HP_OK;
}

int refresh_fusable(hp_domain* d)
{
	int rc = d->desc.precision == 8 ? refresh_fusable_t<double>(d) : refresh_fusable_t<float>(d);
	if (rc != HP_OK) return rc;
	// nothing of a next iteration is in any buffer at this point (a batch's last iteration never fuses)
	HIP_TRY(hipMemsetAsync((char*)d->cfl_slot + (size_t)SLOT_BDY * d->esize, 0, d->esize, d->stream));
	return HP_OK;
}

int spec_resolve(hp_domain* d);
int tail_failed_error()
{
	return fail(HP_ERR_STATE, "a flux launch's tail block gave up waiting for a flux block (HP_TAIL_TIMEOUT_MS): time and timestep "
	                          "are frozen from that iteration on, the domain's state is not valid; destroy the domain");
}
int check_domain(hp_domain* d)
{
	if (d) d->fork_is_advance = false;        // any entry point may queue work behind the last advance_time
	if (!d) return fail(HP_ERR_INVALID, "null domain");
	hipError_t e = hipSetDevice(d->desc.device);
	if (e != hipSuccess) return fail(HP_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
	if (d->tail_failed) return tail_failed_error();      // (hp_domain_destroy does not come through here)
	// a speculative STRICT batch is still to be looked at: whoever enters the library next does that first, so that nothing is
	// ever queued behind -- or read from -- a batch that has to be re-run
	if (d->spec_pending) return spec_resolve(d);
	return HP_OK;
}

} // namespace

// =================================================================================================
extern "C" {

int hp_abi_version(void) { return HP_ABI_VERSION; }

const char* hp_last_error(void) { return g_last_error.c_str(); }

int hp_set_log_sink(hp_log_sink_t sink, void* user)
{
	g_log_sink.store(nullptr, std::memory_order_release);
	g_log_user.store(user, std::memory_order_release);
	g_log_sink.store(sink, std::memory_order_release);
	return HP_OK;
}

int hp_device_count(int* count)
{
	if (!count) return fail(HP_ERR_INVALID, "count == NULL");
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) {
		*count = 0;
		return fail(HP_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
	}
	*count = n;
	return HP_OK;
}

int hp_device_info(int device, hp_device_info_t* info)
{
	if (!info) return fail(HP_ERR_INVALID, "info == NULL");
	int n = 0, rc = hp_device_count(&n);
	if (rc != HP_OK) return rc;
	if (device < 0 || device >= n) return fail(HP_ERR_INVALID, "device index out of range");
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	std::memset(info, 0, sizeof *info);
	std::snprintf(info->name, sizeof info->name, "%s", prop.name);
	std::snprintf(info->arch, sizeof info->arch, "%s", prop.gcnArchName);
	info->compute_units = prop.multiProcessorCount;
	info->clock_mhz = prop.clockRate / 1000;
	info->global_mem_bytes = prop.totalGlobalMem;
	info->lds_bytes_per_cu = prop.maxSharedMemoryPerMultiProcessor;
	info->wavefront = prop.warpSize;
	info->fp64 = 1;
	return HP_OK;
}

void hp_domain_desc_default(hp_domain_desc_t* desc)
{
	if (!desc) return;
	std::memset(desc, 0, sizeof *desc);
	desc->struct_size = (uint32_t)sizeof *desc;
	desc->device = 0;
	desc->dx = 1.0;
	desc->precision = 8;
	desc->scheme = HP_SCHEME_GODUNOV;
	desc->courant = 0.5;                 // CScheme.cpp:48
	desc->dry_threshold = 1e-10;         // CSchemeGodunov.cpp:56
	desc->friction = 1;                  // CScheme.cpp:51
	desc->dynamic_dt = 1;                // CScheme.cpp:50
	desc->dt_fixed = 0.001;
	desc->dt_initial = 0.001;            // CScheme.cpp:49
	desc->t_end = 1e30;
	desc->quirks = HP_QUIRKS_REFERENCE;
	desc->math_mode = HP_MATH_FAST;
	desc->kernel = HP_KERNEL_AUTO;
}

int hp_domain_create(const hp_domain_desc_t* desc, hp_domain_t** out)
{
	if (!desc || !out) return fail(HP_ERR_INVALID, "null argument");
	if (desc->struct_size != sizeof(hp_domain_desc_t)) return fail(HP_ERR_INVALID, "hp_domain_desc_t size mismatch");
	if (desc->cols < 3 || desc->rows < 3) return fail(HP_ERR_INVALID, "grid must be at least 3x3");
	if (desc->precision != 8 && desc->precision != 4) return fail(HP_ERR_INVALID, "precision must be 8 or 4");
	if (desc->scheme != HP_SCHEME_GODUNOV && desc->scheme != HP_SCHEME_MUSCL_HANCOCK &&
	    desc->scheme != HP_SCHEME_INERTIAL)
		return fail(HP_ERR_INVALID, "unknown scheme");
	if (!(desc->dx > 0)) return fail(HP_ERR_INVALID, "dx must be positive");
	int n = 0, rc = hp_device_count(&n);
	if (rc != HP_OK) return rc;
	if (desc->device < 0 || desc->device >= n) return fail(HP_ERR_INVALID, "device index out of range");
	HIP_TRY(hipSetDevice(desc->device));

	hp_domain* d = new hp_domain();
	d->desc = *desc;
	if (d->desc.global_rows <= 0) { d->desc.global_rows = d->desc.rows; d->desc.row_offset = 0; }
	if (d->desc.row_offset < 0 || d->desc.row_offset + d->desc.rows > d->desc.global_rows) {
		delete d;
		return fail(HP_ERR_INVALID, "strip does not fit the global grid");
	}
	const long g = (desc->scheme == HP_SCHEME_MUSCL_HANCOCK) ? 2 : 1;     // stencil reach = ghost rows per exchange
	d->ghost_rows = desc->ghost_rows > 0 ? desc->ghost_rows : g;
	if (d->ghost_rows != g && d->ghost_rows != 2 * g) {
		delete d;
		return fail(HP_ERR_INVALID, "ghost_rows must be 0, the scheme's stencil reach or twice that");
	}
	d->ghost_valid = d->ghost_rows;
	d->split_now = true;                                                  // (hp_step_begin: every iteration is followed by an exchange)
	d->own_lo = (d->desc.row_offset > 0) ? d->ghost_rows : 0;
	d->own_hi = d->desc.rows - ((d->desc.row_offset + d->desc.rows < d->desc.global_rows) ? d->ghost_rows : 0);
	if (d->own_hi - d->own_lo < 1) { delete d; return fail(HP_ERR_INVALID, "strip has no owned rows"); }
	d->cells = (size_t)desc->cols * (size_t)desc->rows;
	d->esize = (size_t)desc->precision;
	// Tile height: a wavefront marches its tile's rows one after the other, so a grid that yields few tiles is bound
	// by that serial walk, not by bandwidth (342 x 195: 35 us per step at 16 rows, 12 us at 2).  Take the tallest
	// tile that still gives every CU some blocks (tools/small_grid_probe.py: 4096^2 and 2048^2 want 16, 1024^2 8,
	// 512^2 4, the 342 x 195 example 2).
	{
		int cus = 256;
		hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, desc->device);
		static const bool refine = !(std::getenv("HP_RSEG_REFINE") && std::atoi(std::getenv("HP_RSEG_REFINE")) == 0);
		// (read per domain, not once per process: tools/strong_probe_pair.py creates domains both ways)
		const bool one_round_search = !(std::getenv("HP_TILING_SEARCH") && std::atoi(std::getenv("HP_TILING_SEARCH")) == 0);
		auto pick = [&](long updated_rows, long updated_cols, int tile_cols, int tallest, int shortest, int blocks_per_cu, int& nbands_out, bool searchable,
		                int* classic_out = nullptr) {
			const long groups = ((updated_cols + tile_cols - 1) / tile_cols + 3) / 4;
			nbands_out = 8;
			int rseg = tallest;
			while (rseg > shortest && groups * ((updated_rows + rseg - 1) / rseg) < 350) rseg /= 2;
			if (rseg < shortest) rseg = shortest;
			// Round 3: a launch whose blocks all fit the chip at once (a row strip of a strong-scaling run: 4096 x 514 is 544
			// tiles of 16 rows on 768 block slots) lasts as long as ONE tile does, so the shortest tile that still fits in one
			// round wins: 13-row tiles put that strip into 680 blocks (56 -> 51 us per iteration).  Fine-tune within
			// (rseg/2, rseg] where the halving above left a middle-sized tile; launches of more than one round are left alone
			// (tools/history/r03k.sh: 2048^2, 4096 x 1026 and 8192 x 514 lose 0-9 % with shorter tiles).
			if (refine && rseg >= 8) {
				const long band_rows = (updated_rows + 7) / 8, slots = (long)cus * blocks_per_cu;
				auto cost = [&](int r) {
					const long blocks = groups * 8 * ((band_rows + r - 1) / r);
					return ((blocks + slots - 1) / slots) * (long)(2 * r + 7);        // rounds x (rows + 3.5), doubled to stay integral
				};
				if (groups * 8 * ((band_rows + rseg - 1) / rseg) <= slots) {        // measured (profiles/r03k): no gain beyond one round
					int best = rseg;
					for (int r = rseg - 1; r > rseg / 2; --r) if (cost(r) < cost(best)) best = r;
					rseg = best;
					// Round 4: what such a launch really lasts is what its MOST LOADED CU has to walk: blocks are dealt to the CUs
					// layer by layer (block b lands on CU b mod cus while every CU has room), a block costs its rows plus ~3.5 rows
					// of pipeline fill, and the blocks of one CU share its SIMDs.  8 bands x groups x segments only offers coarse
					// block counts (680 blocks on 256 CUs: 168 CUs walk three 13-row tiles, 88 walk two); with another number
					// of bands the last segment of each band can be a SHORT tile, and "two tall + one short on every CU" (15
					// bands, 15 + 15 + 5 rows: 765 blocks) measured 39.1 us per iteration on the 4096 x 514 strip against 43.4
					// (profiles/r04b_band_sweep_4096x514.txt).  The search simulates the dealing for every (bands, rows) pair that
					// fits one round and takes the cheapest; a CU with fewer than three resident blocks hides less latency
					// (2 blocks: +20 %, 1 block: +60 %, fitted on the same sweep).
					// (the searched height is only right TOGETHER with its band count, which only a whole-domain launch gets: the interior /
					// halo parts of a split step keep the classic height -- ADVICE r04)
					if (classic_out) *classic_out = rseg;
					if (one_round_search && searchable) {
						// a tile's fixed cost in row-times: 3.5 for fp64 K1 / K6 (the pipeline fill and the extra south face)
						const double fill = std::getenv("HP_TILING_FILL") ? std::atof(std::getenv("HP_TILING_FILL")) : 3.5;
						double best_cost = 1e30;
						int best_nb = 8, best_r = rseg;
						long best_pref = -2;
						std::vector<double> load((size_t)cus);
						std::vector<int> count((size_t)cus);
						for (int nb = 1; nb <= 32; ++nb) {
							const long brows = (updated_rows + nb - 1) / nb;
							for (int r = 4; r <= 64 && r <= brows; ++r) {
								const long nseg = (brows + r - 1) / r, blocks = (long)nb * groups * nseg;
								if (blocks > slots) continue;
								std::fill(load.begin(), load.end(), 0.0);
								std::fill(count.begin(), count.end(), 0);
								for (long b = 0; b < blocks; ++b) {                  // tile_rows() of hp_kernels.hpp
									const long band = b % nb, i = b / nb, seg = i / groups;
									const long y0 = band * brows + seg * r;
									const long band_end = std::min(updated_rows, (band + 1) * brows);
									const long h = std::min(y0 + r, band_end) - y0;
									if (h <= 0) continue;
									load[(size_t)(b % cus)] += (double)h + fill;
									count[(size_t)(b % cus)] += 1;
								}
								double worst = 0.0;
								for (int c = 0; c < cus; ++c) {
									const double f = count[(size_t)c] >= 3 ? 1.0 : count[(size_t)c] == 2 ? 1.2 : 1.6;
									worst = std::max(worst, load[(size_t)c] * f);
								}
								// (ties: round 5 -- the last tile of a band as tall as it can be up to half a full tile (the pair kernel's rule below;
								// K1 on the 4096 x 514 strip: 14 + 14 + 7 33.3 us, 13 + 13 + 9 33.1, round 4's 15 + 15 + 5 34.3, profiles/r05fq_*) --,
								// then the taller first tiles, then the tiling nearest to the 8-band one, whose bands keep to their XCD's L2)
								const long last = brows - (nseg - 1) * r;
								const long pref = (nseg >= 2 && nseg <= 3 && 2 * last <= r) ? last : -1;
								if (worst < best_cost - 1e-9 || (worst < best_cost + 1e-9 && (pref > best_pref || (pref == best_pref &&
								    (r > best_r || (r == best_r && std::abs(nb - 8) < std::abs(best_nb - 8))))))) {
									best_cost = worst; best_nb = nb; best_r = r; best_pref = pref;
								}
							}
						}
						nbands_out = best_nb; rseg = best_r;
					}
				}
			}
			return rseg;
		};
		// fp32 (round 4, profiles/r04za_rseg_f32.txt, r04z_band_sweep_f32_shapes.txt): a tile's fill and extra south face want tall
		// tiles, the dispatcher wants many -- 32 rows is best only where that still leaves six rounds of blocks (8192^2: 6.8); below
		// that shorter tiles win by more than the fill costs: 4096^2 S-DAM 0.143 -> 0.129 ms and S-RAIN 0.180 -> 0.168 at 12 rows,
		// the 8192 x 1026 strip of config C5 0.113 -> 0.089 ms (S-RAIN) and 83 -> 66 us (S-DAM); 8 rows lose again.  (Round 2's "32
		// rows, 13-47 % ahead of 16" was measured on kernels with a dearer fill.)  fp64 is flat from 10 to 20 rows.
		// fp64 (profiles/r04zc_rseg_f64_shapes.txt): flat from 10 to 20 rows where a launch is many rounds (4096^2, 16384 x 1026); between
		// one round and 2.5 (2048^2, 4096 x 1026, 8192 x 1026 ...) 12 rows are 5-9 % ahead of 16 for K1, 10 rows 5-7 % ahead of 12 for K2.
		const bool f32 = desc->precision == 4;
		auto tallest_by_rounds = [&](long updated_rows, long updated_cols, int tile_cols, int blocks_per_cu, std::initializer_list<int> tall,
		                             double rounds, int otherwise) {
			const long groups = ((updated_cols + tile_cols - 1) / tile_cols + 3) / 4, band_rows = (updated_rows + 7) / 8;
			// (fp64: a launch that fits the chip in about ONE round at the classic height belongs to the one-round logic inside pick();
			// the 4096 x 516 MUSCL strip, 1.1 rounds at 12 rows, measured 3 % behind with 10)
			if (!f32 && (double)(groups * 8 * ((band_rows + *tall.begin() - 1) / *tall.begin())) <= 1.25 * cus * blocks_per_cu) return *tall.begin();
			for (int r : tall)
				if ((double)(groups * 8 * ((band_rows + r - 1) / r)) >= rounds * cus * blocks_per_cu) return r;
			return otherwise;
		};
		const int k1_tallest = f32 ? tallest_by_rounds(desc->rows - 2, desc->cols - 2, MARCH_COLS, 5, {32, 24, 16}, 6.0, 12)
		                           : tallest_by_rounds(desc->rows - 2, desc->cols - 2, MARCH_COLS, 3, {16}, 2.5, 12);
		const int k2_tallest = f32 ? tallest_by_rounds(desc->rows - 4, desc->cols - 4, MUSCL_COLS, 4, {32, 24, 16}, 6.0, 12)
		                           : tallest_by_rounds(desc->rows - 4, desc->cols - 4, MUSCL_COLS, 3, {12}, 3.0, 10);
		int classic = 0;
		d->march_rseg    = pick(desc->rows - 2, desc->cols - 2, MARCH_COLS, k1_tallest, 2, desc->precision == 4 ? 5 : 3, d->march_nbands,
		                        desc->precision == 8 || std::getenv("HP_TILING_SEARCH_F32") != nullptr, &classic);      // (fp32: 8192 x 1026 measured 83.2 -> 84.4 us with the searched tiling: its fill is not 3.5 rows)
		d->march_rseg_parts = classic > 0 ? classic : d->march_rseg;
		d->inertial_rseg = d->march_rseg; d->inertial_nbands = d->march_nbands; d->inertial_rseg_parts = d->march_rseg_parts;
		// K2 after the inert-row cut (round 2): a tile of still water or dry land costs a fifth of a tile on the flood front,
		// so fp64 wants more, shorter tiles for the dispatcher to balance (16-20 rows: 0.319 ms against 0.355 at 32 on the
		// 4096^2 dam break, 0.355 against 0.395 on the developed flood, +2-5 % at 8192^2 and 16384 x 1028)
		// (round 4, profiles/r04l_k2_rseg.txt: 12 rows: the 4096^2 dam break 0.300 -> 0.287 ms, developed flood 0.343 -> 0.340, 2048^2 -1.4 %,
		// 8192^2 +0.6 %, every tile live +1.7 %; 10 and below lose again; fp32: as K1 -- 4096^2 dam break 0.192 -> 0.161 ms at 12 rows)
		d->muscl_rseg    = pick(desc->rows - 4, desc->cols - 4, MUSCL_COLS, k2_tallest, 4, desc->precision == 4 ? 4 : 3, d->muscl_nbands,
		                        false);                     // (K2's tiles differ fivefold in cost -- inert rows --: the searched tiling lost 20 % on the 4096 x 514 dam break)
		if (d->muscl_rseg < 4) d->muscl_rseg = 4;
	}
	if (std::getenv("HP_PRINT_TILING"))
		std::fprintf(stderr, "[hipims_mi] tiling %ld x %ld: K1/K6 %d rows x %d bands, K2 %d rows x %d bands\n", (long)desc->cols, (long)desc->rows,
		             d->march_rseg, d->march_nbands, d->muscl_rseg, d->muscl_nbands);
	d->print_tiling = std::getenv("HP_PRINT_TILING") != nullptr;
	if (const char* e = std::getenv("HP_MARCH_RSEG")) {                   // tuning knob: rows per wavefront tile
		const int v = std::atoi(e);
		if (v >= 1 && v <= 64) { d->march_rseg = d->march_rseg_parts = v; d->tall_rseg = 16; }   // a forced 16 stays 16
	}
	{
		// the two-iterations kernel: the tallest tile that leaves four rounds of blocks (three 4-wave blocks per CU); worth taking at
		// all from two rounds at 12-row tiles (profiles/r05n_two_step.txt)
		int cus = 256;
		hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, desc->device);
		const long groups = (((long)desc->cols - 2 + MARCH2_COLS - 1) / MARCH2_COLS + 3) / 4, band_rows = ((long)desc->rows - 2 + 7) / 8;
		const long slots = (long)cus * 3;
		auto blocks_at = [&](int r) { return groups * 8 * ((band_rows + r - 1) / r); };
		d->march2_rseg = 12;
		for (int r : {32, 24, 18}) if (blocks_at(r) >= 4 * slots) { d->march2_rseg = r; break; }
		d->march2_pays = blocks_at(12) >= 2 * slots;
		// One round of blocks (the 4096 x 514 strip of a strong-scaling run): as for K1 (pick: one_round_search), what such a launch
		// lasts is what its most loaded CU walks -- (bands, rows) by simulating the dealing, a tile costing its rows plus the two
		// extra first-step rows and the pipeline fill.  profiles/r05u_strip_pair_sweep.txt: 14 bands x 15 rows 31.6 us per iteration
		// against 33.7 in single iterations and 36.0 with 8 bands x 12 rows; taken from 1.5 M cells on (below: untested).
		static const bool search = !(std::getenv("HP_TILING_SEARCH") && std::atoi(std::getenv("HP_TILING_SEARCH")) == 0);
		const long updated_rows = (long)desc->rows - 2;
		if (!d->march2_pays && search && d->cells >= 1500000 && blocks_at(12) >= slots / 2) {
			const double fill = std::getenv("HP_MARCH2_FILL") ? std::atof(std::getenv("HP_MARCH2_FILL")) : 4.0;
			double best_cost = 1e30;
			int best_nb = 0, best_r = 0;
			long best_pref = -2;
			std::vector<double> load((size_t)cus);
			std::vector<int> count((size_t)cus);
			for (int nb = 8; nb <= 32; ++nb) {                              // (at least a band per XCD: 4 bands x 14 rows lost 2 % at 1448^2)
				const long brows = (updated_rows + nb - 1) / nb;
				for (int r = 8; r <= 32 && r <= brows; ++r) {
					const long nseg = (brows + r - 1) / r, blocks = (long)nb * groups * nseg;
					if (blocks > slots) continue;
					std::fill(load.begin(), load.end(), 0.0);
					std::fill(count.begin(), count.end(), 0);
					for (long b = 0; b < blocks; ++b) {                      // tile_rows() of hp_kernels.hpp
						const long band = b % nb, i = b / nb, seg = i / groups;
						const long y0 = band * brows + seg * r;
						const long h = std::min(y0 + r, std::min(updated_rows, (band + 1) * brows)) - y0;
						if (h <= 0) continue;
						load[(size_t)(b % cus)] += (double)h + fill;
						count[(size_t)(b % cus)] += 1;
					}
					double worst = 0.0;
					for (int c = 0; c < cus; ++c)
						worst = std::max(worst, load[(size_t)c] * (count[(size_t)c] >= 3 ? 1.0 : count[(size_t)c] == 2 ? 1.2 : 1.6));
					// Ties -- the same rows and the same number of tiles on every CU, cut differently (37 rows as 16 + 16 + 5, 15 + 15 + 7,
					// 14 + 14 + 9 ...): measured over eight one-round shapes (profiles/r05fp_tie_sweep.txt), the pair kernel wants the LAST
					// tile of a band as tall as it can be without exceeding half a full tile -- 15 + 15 + 7 is 4 % ahead of 16 + 16 + 5 and
					// 6 % ahead of 13 + 13 + 11 on the 4096 x 514 strip (30.0 against 31.4 / 32.0 us), 11 + 11 + 5 6 % ahead of 12 + 12 + 3
					// on 3072 x 514 -- and otherwise the taller tiles.
					// (bands of two or three tiles: with more the rule made 1448^2 slower)
					const long last = brows - (nseg - 1) * r;
					const long pref = (nseg >= 2 && nseg <= 3 && 2 * last <= r) ? last : -1;
					if (worst < best_cost - 1e-9 || (worst < best_cost + 1e-9 && (pref > best_pref || (pref == best_pref && r > best_r)))) {
						best_cost = worst; best_nb = nb; best_r = r; best_pref = pref;
					}
				}
			}
			if (best_nb > 0) { d->march2_rseg = best_r; d->march2_nbands = best_nb; d->march2_pays = true; }
		}
	}
	if (const char* e = std::getenv("HP_MARCH2_RSEG")) { const int v = std::atoi(e); if (v >= 2 && v <= 32) d->march2_rseg = v; }
	if (d->print_tiling)
		std::fprintf(stderr, "[hipims_mi] tiling %ld x %ld: pair kernel %d rows x %d bands, %s\n", (long)desc->cols, (long)desc->rows, d->march2_rseg,
		             d->march2_nbands, d->march2_pays ? "taken by default" : "not taken by default");
	if (const char* e = std::getenv("HP_TAIL_RSEG")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) d->tail_rseg = v; }
	if (const char* e = std::getenv("HP_TAIL_PCT"))  { const int v = std::atoi(e); if (v >= 0 && v <= 100) d->tail_pct = v; }
	if (const char* e = std::getenv("HP_INERTIAL_RSEG")) {
		const int v = std::atoi(e);
		if (v >= 1 && v <= 64) { d->inertial_rseg = d->inertial_rseg_parts = v; d->tall_rseg = 16; }
	}
	if (const char* e = std::getenv("HP_MUSCL_RSEG")) {
		const int v = std::atoi(e);
		if (v >= 1 && v <= 4096) d->muscl_rseg = v;
	}

	{
		// a wavefront addresses its tile through a buffer resource whose range field is 31 bits wide: the tile's rows
		// (segment + halo rows) must fit in it
		const int rseg_max = std::max(std::max(d->march_rseg, d->march2_rseg), std::max(d->muscl_rseg, d->inertial_rseg));
		if ((double)(rseg_max + 4) * (double)desc->cols * 4.0 * (double)desc->precision >= 2147483647.0) {
			delete d;
			return fail(HP_ERR_UNSUPPORTED, "grid too wide for the tile addressing (cols * 32 B * (rows per tile + 4) >= 2 GiB)");
		}
	}

	auto cleanup = [&](int code) { hp_domain_destroy(d); return code; };
#define HIP_TRY_C(expr)                                                                                \
	do {                                                                                               \
		hipError_t e_ = (expr);                                                                        \
		if (e_ != hipSuccess)                                                                          \
			return cleanup(fail(HP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)));       \
	} while (0)
	HIP_TRY_C(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
	HIP_TRY_C(hipMalloc(&d->state[0], d->cells * 4 * d->esize));
	HIP_TRY_C(hipMalloc(&d->state[1], d->cells * 4 * d->esize));
	HIP_TRY_C(hipMalloc(&d->bed, d->cells * d->esize));
	HIP_TRY_C(hipMalloc(&d->manning, d->cells * d->esize));
	HIP_TRY_C(hipMalloc(&d->scalars, 256));
	HIP_TRY_C(hipMalloc(&d->cfl_slot, CFL_SLOT_BYTES));
	if (std::getenv("HP_PRINT_BUFFERS"))
		std::fprintf(stderr, "[hipims_mi] buffers: state[0] %p state[1] %p bed %p manning %p\n", d->state[0], d->state[1], d->bed, d->manning);

	HIP_TRY_C(hipMalloc((void**)&d->tail_words, 2 * (TAIL_MAX_BLOCKS + 8) * sizeof(unsigned long long)));
	HIP_TRY_C(hipMemset(d->tail_words, 0xff, 2 * (TAIL_MAX_BLOCKS + 8) * sizeof(unsigned long long)));      // every word EMPTY (two arrays: LaunchTail::done, done1)
	HIP_TRY_C(hipHostMalloc(&d->host_scalars, 512, hipHostMallocDefault));
	HIP_TRY_C(hipMemset(d->state[0], 0, d->cells * 4 * d->esize));
	HIP_TRY_C(hipMemset(d->state[1], 0, d->cells * 4 * d->esize));
	HIP_TRY_C(hipMemset(d->bed, 0, d->cells * d->esize));
	HIP_TRY_C(hipMemset(d->manning, 0, d->cells * d->esize));
	HIP_TRY_C(hipMemset(d->cfl_slot, 0, CFL_SLOT_BYTES));
	HIP_TRY_C(hipMemset(d->scalars, 0, 256));
	HIP_TRY_C(hipEventCreate(&d->ev_start));
	HIP_TRY_C(hipEventCreate(&d->ev_stop));
	{
		int lo = 0, hi = 0;                                               // numerically lower = higher priority
		HIP_TRY_C(hipDeviceGetStreamPriorityRange(&lo, &hi));
		HIP_TRY_C(hipStreamCreateWithPriority(&d->stream_halo, hipStreamNonBlocking, hi));
	}
	HIP_TRY_C(hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming));
	HIP_TRY_C(hipEventCreateWithFlags(&d->ev_halo, hipEventDisableTiming));
#undef HIP_TRY_C
	rc = (d->desc.precision == 8) ? write_scalars_initial<double>(d) : write_scalars_initial<float>(d);
	if (rc != HP_OK) return cleanup(rc);
	*out = d;
	return HP_OK;
}

int hp_domain_destroy(hp_domain_t* d)
{
	if (!d) return HP_OK;
	hipSetDevice(d->desc.device);
	if (d->stream) hipStreamSynchronize(d->stream);
	if (d->stream_halo) hipStreamSynchronize(d->stream_halo);
	peer_release(d);
	for (auto& b : d->bdy) { hipFree(b.data); hipFree(b.cells); }
	for (auto& ev : d->timing_events) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
	hipFree(d->state[0]); hipFree(d->state[1]); hipFree(d->bed); hipFree(d->manning);
	hipFree(d->scalars); hipFree(d->cfl_slot);
	hipFree(d->saved_state); hipFree(d->saved_scalars); hipFree(d->fused_list); hipFree(d->tail_words);
	hipFree(d->z_state); hipFree(d->haz_words);
	for (hipEvent_t e : d->tune_ev) if (e) hipEventDestroy(e);
	hipFree(d->spec_state); hipFree(d->spec_scalars);
	if (d->host_scalars) hipHostFree(d->host_scalars);
	if (d->ev_start) hipEventDestroy(d->ev_start);
	if (d->ev_stop) hipEventDestroy(d->ev_stop);
	if (d->ev_fork) hipEventDestroy(d->ev_fork);
	if (d->ev_halo) hipEventDestroy(d->ev_halo);
	if (d->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(d->comm);
	if (d->ev_xchg) hipEventDestroy(d->ev_xchg);
	if (d->stream_halo) hipStreamDestroy(d->stream_halo);
	if (d->stream) hipStreamDestroy(d->stream);
	delete d;
	return HP_OK;
}

int hp_domain_upload(hp_domain_t* d, int which, const void* host, size_t bytes)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	d->m1_valid = false;                                                 // (pairs with area boundaries start cold: pair_cold_start)
	if (!host) return fail(HP_ERR_INVALID, "host == NULL");
	if (d->in_step) return fail(HP_ERR_STATE, "upload between hp_step_begin and hp_step_end");
	switch (which) {
	case HP_ARRAY_STATE:
		if (bytes != d->cells * 4 * d->esize) return fail(HP_ERR_INVALID, "state size mismatch");
		// both ping-pong buffers get the same host array (CSchemeGodunov.cpp:1064-1065)
		HIP_TRY(hipMemcpyAsync(d->state[0], host, bytes, hipMemcpyHostToDevice, d->stream));
		HIP_TRY(hipMemcpyAsync(d->state[1], host, bytes, hipMemcpyHostToDevice, d->stream));
		d->other_stale = false;
		d->rings_differ = false;
		d->tune_phase = 0;                                                // (a sample still in flight measured the state that has just been replaced)
		d->use_alt = 0;                                                   // :1075
		d->need_full_reduce = true;
		d->edge_dirty = true;
		d->ghost_valid = d->ghost_rows;                                   // the host uploads strips with all their ghost rows
		HIP_TRY(hipMemsetAsync((char*)d->cfl_slot + (size_t)SLOT_BDY * d->esize, 0, d->esize, d->stream));
		return HP_OK;
	case HP_ARRAY_BED:
		if (bytes != d->cells * d->esize) return fail(HP_ERR_INVALID, "bed size mismatch");
		HIP_TRY(hipMemcpyAsync(d->bed, host, bytes, hipMemcpyHostToDevice, d->stream));
		d->need_full_reduce = true;
		d->edge_dirty = true;
		return HP_OK;
	case HP_ARRAY_MANNING:
		if (bytes != d->cells * d->esize) return fail(HP_ERR_INVALID, "manning size mismatch");
		HIP_TRY(hipMemcpyAsync(d->manning, host, bytes, hipMemcpyHostToDevice, d->stream));
		{
			// "constant" Manning sources are the norm (<dataSource type="constant" value="manningCoefficient">):
			// a uniform array is passed to the kernels as a scalar and its 8 B/cell are not streamed every step
			bool same = true;
			if (d->esize == 8) {
				const double* v = (const double*)host;
				for (size_t i = 1; i < d->cells && same; ++i) same = (v[i] == v[0]);
				d->manning_value = v[0];
			} else {
				const float* v = (const float*)host;
				for (size_t i = 1; i < d->cells && same; ++i) same = (v[i] == v[0]);
				d->manning_value = v[0];
			}
			d->manning_uniform = same;
		}
		return HP_OK;
	}
	return fail(HP_ERR_INVALID, "unknown array id");
}

// saveCurrentState / rollbackSimulation without the PCIe round trip: the copy stays in HBM
int hp_state_save(hp_domain_t* d)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (d->in_step) return fail(HP_ERR_STATE, "hp_state_save between hp_step_begin and hp_step_end");
	if ((rc = repair_other_buffer(d)) != HP_OK) return rc;           // (after iteration pairs: see run_pair)
	const size_t bytes = d->cells * 4 * d->esize;
	const size_t sc_bytes = d->desc.precision == 8 ? sizeof(Scalars<double>) : sizeof(Scalars<float>);
	// BOTH ping-pong buffers: the one the next iteration writes is not dead -- cells whose whole neighbourhood is dry are left
	// untouched by the flux kernel (quirk Q3) and keep what that buffer held, so a replay needs it too (a fuzz over random
	// configurations found 6 % of them replaying differently when only the next-source buffer was kept, round 3)
	if (!d->saved_state) HIP_TRY(hipMalloc(&d->saved_state, 2 * bytes));
	if (!d->saved_scalars) HIP_TRY(hipMalloc(&d->saved_scalars, sc_bytes + CFL_SLOT_BYTES));
	HIP_TRY(hipMemcpyAsync(d->saved_state, d->state[d->use_alt], bytes, hipMemcpyDeviceToDevice, d->stream));
	HIP_TRY(hipMemcpyAsync((char*)d->saved_state + bytes, d->state[d->use_alt ^ 1], bytes, hipMemcpyDeviceToDevice, d->stream));
	HIP_TRY(hipMemcpyAsync(d->saved_scalars, d->scalars, sc_bytes, hipMemcpyDeviceToDevice, d->stream));
	// the WHOLE slot block: running maximum, last maximum used (SLOT_SAVED), ring maxima (SLOT_EDGE) -- round 2 kept the first
	// four elements only, which since the slots moved 256 B apart no longer included the remembered maximum
	HIP_TRY(hipMemcpyAsync((char*)d->saved_scalars + sc_bytes, d->cfl_slot, CFL_SLOT_BYTES, hipMemcpyDeviceToDevice, d->stream));
	d->saved_full_reduce = d->need_full_reduce;
	d->saved_edge_dirty = d->edge_dirty;
	d->saved_use_alt = d->use_alt;
	d->saved_rings_differ = d->rings_differ;
	d->saved_m1_valid = d->m1_valid;
	d->saved_ghost_valid = d->ghost_valid;
	d->saved_valid = true;
	return HP_OK;
}

int hp_state_restore(hp_domain_t* d)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (d->in_step) return fail(HP_ERR_STATE, "hp_state_restore between hp_step_begin and hp_step_end");
	if (!d->saved_valid) return fail(HP_ERR_STATE, "hp_state_restore without a saved state");
	const size_t bytes = d->cells * 4 * d->esize;
	const size_t sc_bytes = d->desc.precision == 8 ? sizeof(Scalars<double>) : sizeof(Scalars<float>);
	// both ping-pong buffers return to what EACH of them held (rollbackSimulation writes the saved next-source state into both,
	// CSchemeGodunov.cpp:1496-1499, after which its cells with an all-dry neighbourhood no longer continue as the original run
	// did), and so do the ping-pong phase and the remembered CFL maxima: the steps that follow repeat the original ones bit for bit
	HIP_TRY(hipMemcpyAsync(d->state[d->saved_use_alt], d->saved_state, bytes, hipMemcpyDeviceToDevice, d->stream));
	HIP_TRY(hipMemcpyAsync(d->state[d->saved_use_alt ^ 1], (char*)d->saved_state + bytes, bytes, hipMemcpyDeviceToDevice, d->stream));
	HIP_TRY(hipMemcpyAsync(d->scalars, d->saved_scalars, sc_bytes, hipMemcpyDeviceToDevice, d->stream));
	// (the slot block AROUND the sticky "a tail block gave up" word: a word raised since the checkpoint, and not yet seen by the host
	// through hp_sync / hp_read_scalars, must survive the roll-back -- ADVICE r05)
	{
		const size_t lo = (size_t)SLOT_TAIL_ERR * d->esize, hi = lo + d->esize;
		const char* from = (const char*)d->saved_scalars + sc_bytes;
		HIP_TRY(hipMemcpyAsync(d->cfl_slot, from, lo, hipMemcpyDeviceToDevice, d->stream));
		HIP_TRY(hipMemcpyAsync((char*)d->cfl_slot + hi, from + hi, CFL_SLOT_BYTES - hi, hipMemcpyDeviceToDevice, d->stream));
	}
	d->use_alt = d->saved_use_alt;
	d->other_stale = false;                                               // (a checkpoint is taken with both buffers brought up to date)
	d->rings_differ = d->saved_rings_differ;
	d->rings_checked = false;
	d->m1_valid = d->saved_m1_valid;                                      // (slot[SLOT_M1] has come back with the slot block)
	d->tune_phase = 0;                                                    // (another state: the exact mode's pairs-or-singles choice is measured anew -- a sample
	                                                                      // still in flight measured the state that is being replaced: with samples every 128
	                                                                      // iterations bench.py's restore found one in flight nearly every time and the timed
	                                                                      // region ran on a choice made for the pre-warmed flood)
	d->pair_fused_next = false;                                           // (a checkpoint is taken between batches: nothing fused is in the buffer)
	d->ghost_valid = d->saved_ghost_valid;
	// a bed or state upload between save and restore has left its own marks: they stay
	d->need_full_reduce = d->need_full_reduce || d->saved_full_reduce;
	d->edge_dirty = d->edge_dirty || d->saved_edge_dirty;
	d->fork_is_advance = false;
	return HP_OK;
}

int hp_domain_download(hp_domain_t* d, int which, void* host, int64_t row0, int64_t nrows)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (!host) return fail(HP_ERR_INVALID, "host == NULL");
	if (row0 < 0 || nrows < 0 || row0 + nrows > d->desc.rows) return fail(HP_ERR_INVALID, "row range out of bounds");
	const size_t per_row = (size_t)d->desc.cols * d->esize * (which == HP_ARRAY_STATE ? 4 : 1);
	const void* base;
	switch (which) {
	case HP_ARRAY_STATE:   base = d->state[d->use_alt]; break;            // getNextCellSourceBuffer (:1705-1715)
	case HP_ARRAY_BED:     base = d->bed; break;
	case HP_ARRAY_MANNING: base = d->manning; break;
	default: return fail(HP_ERR_INVALID, "unknown array id");
	}
	HIP_TRY(hipMemcpyAsync(host, (const char*)base + (size_t)row0 * per_row, (size_t)nrows * per_row,
	                       hipMemcpyDeviceToHost, d->stream));
	return HP_OK;
}

int hp_domain_upload_rows(hp_domain_t* d, const void* host, int64_t row0, int64_t nrows)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	d->m1_valid = false;                                                 // (pairs with area boundaries start cold: pair_cold_start)
	if (!host) return fail(HP_ERR_INVALID, "host == NULL");
	if (row0 < 0 || nrows < 0 || row0 + nrows > d->desc.rows) return fail(HP_ERR_INVALID, "row range out of bounds");
	const size_t per_row = (size_t)d->desc.cols * d->esize * 4;
	// queueWritePartial goes to the CURRENT buffer only: the other one keeps what it held -- after iteration pairs it has to be
	// brought up to date first, or the rows written now would reach it with the next single iteration's FILL
	if ((rc = repair_other_buffer(d)) != HP_OK) return rc;
	HIP_TRY(hipMemcpyAsync((char*)d->state[d->use_alt] + (size_t)row0 * per_row, host, (size_t)nrows * per_row,
	                       hipMemcpyHostToDevice, d->stream));
	d->need_full_reduce = true;
	d->edge_dirty = true;
	d->rings_differ = true;
	d->rings_checked = false;
	return HP_OK;
}

int hp_boundary_add_uniform(hp_domain_t* d, int definition, const void* series, uint32_t entries,
                            double interval, double length)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	d->m1_valid = false;                                                 // (pairs with area boundaries start cold: pair_cold_start)
	if (!series || entries == 0 || !(interval > 0)) return fail(HP_ERR_INVALID, "bad uniform boundary");
	if (definition != HP_UNIFORM_RAIN_INTENSITY && definition != HP_UNIFORM_LOSS_RATE)
		return fail(HP_ERR_INVALID, "unknown uniform boundary definition");
	Boundary b{};
	b.kind = 0; b.definition = definition; b.entries = entries; b.interval = interval; b.length = length;
	const size_t bytes = (size_t)entries * 2 * d->esize;
	HIP_TRY(hipMalloc(&b.data, bytes));
	HIP_TRY(hipMemcpy(b.data, series, bytes, hipMemcpyHostToDevice));
	d->bdy.push_back(b);
	return refresh_fusable(d);
}

int hp_boundary_add_gridded(hp_domain_t* d, int definition, const void* grids, uint64_t entries,
                            uint64_t grid_rows, uint64_t grid_cols, double resolution,
                            double offset_x, double offset_y, double interval)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	d->m1_valid = false;                                                 // (pairs with area boundaries start cold: pair_cold_start)
	if (!grids || entries == 0 || grid_rows == 0 || grid_cols == 0 || !(resolution > 0) || !(interval > 0))
		return fail(HP_ERR_INVALID, "bad gridded boundary");
	{
		// bdy_Gridded indexes floor((x dx - off) / res) without a range check (CLBoundaries.clc:231-236): a grid that
		// does not cover the interior cells reads outside the buffer.  Rejected here instead.
		const double dx = d->desc.dx;
		const double x_lo = 1.0 * dx - offset_x, x_hi = (double)(d->desc.cols - 2) * dx - offset_x;
		const double y_lo = 1.0 * dx - offset_y, y_hi = (double)(d->desc.global_rows - 2) * dx - offset_y;
		if (x_lo < 0.0 || y_lo < 0.0 || std::floor(x_hi / resolution) >= (double)grid_cols ||
		    std::floor(y_hi / resolution) >= (double)grid_rows)
			return fail(HP_ERR_INVALID, "gridded boundary does not cover the domain's interior cells");
	}
	Boundary b{};
	b.kind = 1; b.definition = definition; b.entries = entries; b.grows = grid_rows; b.gcols = grid_cols;
	b.resolution = resolution; b.off_x = offset_x; b.off_y = offset_y; b.interval = interval;
	const size_t bytes = (size_t)entries * grid_rows * grid_cols * d->esize;
	HIP_TRY(hipMalloc(&b.data, bytes));
	HIP_TRY(hipMemcpy(b.data, grids, bytes, hipMemcpyHostToDevice));
	d->bdy.push_back(b);
	return refresh_fusable(d);
}

int hp_boundary_add_cell(hp_domain_t* d, int depth_definition, int discharge_definition, const uint64_t* cells,
                         uint64_t count, const void* series, uint64_t entries, double interval, double length)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	d->m1_valid = false;                                                 // (pairs with area boundaries start cold: pair_cold_start)
	if (!cells || count == 0 || !series || entries < 2 || !(interval > 0)) return fail(HP_ERR_INVALID, "bad cell boundary");
	if (depth_definition < 0 || depth_definition > 3 || discharge_definition < 0 || discharge_definition > 3)
		return fail(HP_ERR_INVALID, "unknown cell boundary definition");
	const uint64_t global_cells = (uint64_t)d->desc.cols * (uint64_t)d->desc.global_rows;
	bool on_ring = false;
	{
		const uint64_t w = (d->desc.scheme == HP_SCHEME_MUSCL_HANCOCK) ? 2 : 1, cols = (uint64_t)d->desc.cols,
		               grows = (uint64_t)d->desc.global_rows;
		for (uint64_t i = 0; i < count; ++i) {
			if (cells[i] >= global_cells) return fail(HP_ERR_INVALID, "cell boundary id outside the grid");
			const uint64_t gy = cells[i] / cols, x = cells[i] - gy * cols;
			on_ring = on_ring || x < w || x + w >= cols || gy < w || gy + w >= grows;
		}
	}
	if (length > (double)(entries - 1) * interval + 1e-9)
		return fail(HP_ERR_INVALID, "cell boundary series shorter than its length (interpolation reads entry n+1)");
	Boundary b{};
	b.kind = 2; b.definition = depth_definition; b.discharge_def = discharge_definition; b.count = count;
	b.entries = entries; b.interval = interval; b.length = length;
	HIP_TRY(hipMalloc(&b.cells, count * sizeof(uint64_t)));
	HIP_TRY(hipMemcpy(b.cells, cells, count * sizeof(uint64_t), hipMemcpyHostToDevice));
	const size_t bytes = (size_t)entries * 4 * d->esize;
	HIP_TRY(hipMalloc(&b.data, bytes));
	HIP_TRY(hipMemcpy(b.data, series, bytes, hipMemcpyHostToDevice));
	d->bdy.push_back(b);
	d->bdy_on_ring = d->bdy_on_ring || on_ring;
	return refresh_fusable(d);
}

int hp_boundary_clear(hp_domain_t* d)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	d->m1_valid = false;                                                 // (pairs with area boundaries start cold: pair_cold_start)
	HIP_TRY(hipStreamSynchronize(d->stream));
	for (auto& b : d->bdy) { hipFree(b.data); hipFree(b.cells); }
	d->bdy.clear();
	d->bdy_on_ring = false;
	return refresh_fusable(d);
}

int hp_boundaries_fused(hp_domain_t* d, int* fused)
{
	if (!d || !fused) return fail(HP_ERR_INVALID, "null argument");
	*fused = d->fusable ? 1 : 0;
	return HP_OK;
}

int hp_set_target_time(hp_domain_t* d, double t)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	d->m1_valid = false;                                                 // (pairs with area boundaries start cold: pair_cold_start)
	return d->desc.precision == 8 ? set_scalar_field<double>(d, offsetof(Scalars<double>, t_sync), t)
	                              : set_scalar_field<float>(d, offsetof(Scalars<float>, t_sync), t);
}

int hp_set_time(hp_domain_t* d, double t)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	d->m1_valid = false;                                                 // (pairs with area boundaries start cold: pair_cold_start)
	return d->desc.precision == 8 ? set_scalar_field<double>(d, offsetof(Scalars<double>, t), t)
	                              : set_scalar_field<float>(d, offsetof(Scalars<float>, t), t);
}

int hp_force_timestep(hp_domain_t* d, double dt)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	d->m1_valid = false;                                                 // (pairs with area boundaries start cold: pair_cold_start)
	return d->desc.precision == 8 ? set_scalar_field<double>(d, offsetof(Scalars<double>, dt), dt)
	                              : set_scalar_field<float>(d, offsetof(Scalars<float>, dt), dt);
}

int hp_reset_counters(hp_domain_t* d)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	const size_t off = d->desc.precision == 8 ? offsetof(Scalars<double>, batch_dt) : offsetof(Scalars<float>, batch_dt);
	const size_t end = d->desc.precision == 8 ? sizeof(Scalars<double>) : sizeof(Scalars<float>);
	HIP_TRY(hipMemsetAsync((char*)d->scalars + off, 0, end - off, d->stream));
	return HP_OK;
}

int hp_update_timestep(hp_domain_t* d)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	d->m1_valid = false;                                                 // (pairs with area boundaries start cold: pair_cold_start)
	if (d->in_step) return fail(HP_ERR_STATE, "inside a split step");
	if (d->desc.dynamic_dt && d->desc.global_rows != d->desc.rows)
		return fail(HP_ERR_UNSUPPORTED, "hp_update_timestep on a row strip: the maximum must be all-reduced across "
		                                "ranks not available through this call)");
	// tst_Reduce reads the primary buffer (arg wiring CSchemeGodunov.cpp:922, :927), then tst_UpdateTimestep.  MUSCL-Hancock
	// has ONE state buffer in the reference, updated in place -- its primary buffer is always the current state, which here
	// is whichever ping-pong buffer the next iteration reads (found by the fuzz's update-timestep calls between batches)
	const void* priced = d->desc.scheme == HP_SCHEME_MUSCL_HANCOCK ? d->state[d->use_alt] : d->state[0];
	if (d->desc.precision == 8) {
		if (d->desc.dynamic_dt && (rc = launch_reduce<double>(d, priced, d->own_lo, d->own_hi)) != HP_OK) return rc;
		hipLaunchKernelGGL((advance_time<true, double>), dim3(1), dim3(64), 0, d->stream, make_params<double>(d),
		                   (Scalars<double>*)d->scalars, (double*)d->cfl_slot, 1, PeerBox{}, PeerPush{});
	} else {
		if (d->desc.dynamic_dt && (rc = launch_reduce<float>(d, priced, d->own_lo, d->own_hi)) != HP_OK) return rc;
		hipLaunchKernelGGL((advance_time<true, float>), dim3(1), dim3(64), 0, d->stream, make_params<float>(d),
		                   (Scalars<float>*)d->scalars, (float*)d->cfl_slot, 1, PeerBox{}, PeerPush{});
	}
	HIP_TRY(hipGetLastError());
	return HP_OK;
}

int hp_step_begin(hp_domain_t* d)
{
	const bool fork_ready = d && d->fork_is_advance;     // nothing was queued since the last hp_step_end
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (d->in_step) return fail(HP_ERR_STATE, "hp_step_begin called twice");
	if (d->ghost_rows != strip_ghosts(d))
		return fail(HP_ERR_UNSUPPORTED, "the split step exchanges after every iteration: it needs ghost_rows = the stencil reach "
		                                "(two reaches of ghost rows are hp_strip_step_batch's)");
	d->split_now = true;
	d->fork_is_advance = fork_ready;
	if ((rc = dispatch_begin(d)) != HP_OK) return rc;
	d->in_step = true;
	return HP_OK;
}

int hp_step_needs_reduction(hp_domain_t* d, int* needed)
{
	if (!d || !needed) return fail(HP_ERR_INVALID, "null argument");
	if (!d->in_step) return fail(HP_ERR_STATE, "hp_step_needs_reduction outside a split step");
	*needed = d->adv_fresh;
	return HP_OK;
}

int hp_step_end(hp_domain_t* d)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (!d->in_step) return fail(HP_ERR_STATE, "hp_step_end without hp_step_begin");
	d->in_step = false;
	return dispatch_end(d);
}

} // extern "C"
namespace {
// ---- speculative STRICT fp64 batches -------------------------------------------------------------------------------------
// The STRICT flux kernels have a flavour whose quotients share refined reciprocals (hp_math.hpp: div_shared): bit-identical to
// the plain divisions except for operands at the ends of the exponent range, which it DETECTS (SLOT_SPEC) but does not handle
// -- every in-kernel fall-back cost more than the sharing gains (profiles/r04i).  A batch of at least SPEC_MIN iterations on a
// single domain therefore runs speculatively: snapshot (both state buffers, scalars, slot block: as hp_state_save), the
// shared-reciprocal kernels, the flag word copied to pinned memory behind the last launch.  The next entry into the library
// (check_domain) waits for the batch, looks at the word and, if it is raised, puts the snapshot back and runs the same
// iterations with the plain kernels.  Cost of the snapshot: two device copies of the state per batch (0.45 ms at 4096^2, i.e.
// one iteration's worth per batch); HP_STRICT_SPECULATE=1 switches it on (see spec_wanted for why it is off by default),
// HP_STRICT_SPEC_FORCE=k pretends every k-th batch raised the word (tests).
constexpr uint32_t SPEC_MIN = 8;
constexpr size_t HOST_SPEC_FLAG = 464;      // byte offset in the pinned block (0..127 scalars, 256..447 handshake, 480..495 read-backs)
bool spec_wanted(const hp_domain* d, uint32_t n)
{
	// OPT-IN (HP_STRICT_SPECULATE=1).  Measured with the rigorous flag (profiles/r04j_strict_lines_*): the Godunov kernel gains 2-3 %
	// (S-DAM 4096^2 0.490 -> 0.479 ms, S-RAIN 0.716 -> 0.697), the MUSCL-Hancock kernel LOSES 9-13 % (0.580 -> 0.657, 1.069 -> 1.164):
	// the denominator-side v_div_scale + compare that make the detection exact cost most of what the shared reciprocal saves
	// (without any detection the same kernels run at 0.426 / 0.657 / 0.518 / 0.926 ms).  Not worth a snapshot per batch by default.
	static const bool enabled = std::getenv("HP_STRICT_SPECULATE") && std::atoi(std::getenv("HP_STRICT_SPECULATE")) != 0;
	// Not with FUSED area boundaries (round 5).  The speculative flavour of the fused kernel with its own tail block is the engine's
	// largest instantiation -- 168 VGPRs with 58-66 of them spilled and 103-114 spilled SGPRs -- and what the compiler made of it
	// stopped being the plain kernel's arithmetic when an unrelated flag was added to the kernel (deterministic, relative 1e-7 ...
	// 1e-4, no word raised; the same source built for two waves per SIMD, without the vector spills, is exact again:
	// profiles/r05fg_spec_fused_tail_miscompare.txt).  A 2-3 % experiment is not worth an instantiation that only holds while the
	// register allocator is lucky: such domains run the plain STRICT kernels.
#ifdef HP_KEEP_SPEC_FUSED
	const bool fusable_ok = true;
#else
	const bool fusable_ok = !d->fusable;
#endif
	return enabled && n >= SPEC_MIN && d->desc.math_mode == HP_MATH_STRICT && d->desc.precision == 8 && !d->comm && fusable_ok &&
	       d->desc.kernel != HP_KERNEL_BASIC && (d->desc.scheme == HP_SCHEME_GODUNOV || d->desc.scheme == HP_SCHEME_MUSCL_HANCOCK);
}
int spec_begin(hp_domain* d)
{
	const size_t bytes = d->cells * 4 * d->esize, sc_bytes = sizeof(Scalars<double>);
	if (!d->spec_state) HIP_TRY(hipMalloc(&d->spec_state, 2 * bytes));
	if (!d->spec_scalars) HIP_TRY(hipMalloc(&d->spec_scalars, sc_bytes + CFL_SLOT_BYTES));
	HIP_TRY(hipMemsetAsync((char*)d->cfl_slot + (size_t)SLOT_SPEC * d->esize, 0, d->esize, d->stream));
	HIP_TRY(hipMemcpyAsync(d->spec_state, d->state[0], bytes, hipMemcpyDeviceToDevice, d->stream));
	HIP_TRY(hipMemcpyAsync((char*)d->spec_state + bytes, d->state[1], bytes, hipMemcpyDeviceToDevice, d->stream));
	HIP_TRY(hipMemcpyAsync(d->spec_scalars, d->scalars, sc_bytes, hipMemcpyDeviceToDevice, d->stream));
	HIP_TRY(hipMemcpyAsync((char*)d->spec_scalars + sc_bytes, d->cfl_slot, CFL_SLOT_BYTES, hipMemcpyDeviceToDevice, d->stream));
	d->spec_host.use_alt = d->use_alt; d->spec_host.adv_fresh = d->adv_fresh;
	d->spec_host.need_full_reduce = d->need_full_reduce; d->spec_host.edge_dirty = d->edge_dirty;
	d->spec_host.ghost_valid = d->ghost_valid;
	d->spec_host.cells_calculated = d->cells_calculated; d->spec_host.iterations = d->iterations;
	d->spec_now = true;
	return HP_OK;
}
// The exact mode's choice between pairs and single iterations (round 6).  STRICT pairs are STRICT single iterations bit for bit, so the
// choice is free -- and workload dependent: rows of still water are a copy in STRICT (K1's skip) which a pair makes at half the bytes
// (S-DAM 4096^2: 0.299 -> 0.265 ms per iteration), while water that moves everywhere is bound by instruction issue, where the pair's two
// extra first-step rows per tile and its narrower wavefronts cost 13 % (S-ROUGH: 0.60 -> 0.68 ms; profiles/r06h_strict_pairs.txt).  So the
// engine measures: every 128 iterations two pairs and four single iterations are bracketed by events on the domain's stream
// (nothing blocks: the events are looked at every sixteen iterations of the batch loop and when a batch starts -- a host that queues far
// ahead of the device sees them complete at the next batch call) and the faster flavour runs until the next sample.
// (Round 6, late: the period was 512 and the events were only looked at when a batch STARTED -- on a developing S-DAM flood the crossover
// lies near iteration 1000 and the engine ran pairs until the sample of iteration 1750, 9 % behind single iterations all the while;
// now, in batches of 250, it switches in the batch that contains the crossover: tools/strict_tuner_probe.py, profiles/r06ak_*.  A sample costs one copy of the state and six iterations in the slower flavour: 0.1-0.2 % at this period.)
// HP_PAIR_TUNE=0: always pairs where eligible; HP_PAIR_TUNE_PERIOD=n: iterations between samples.
static bool tuner_on(const hp_domain* d)
{
	static const bool enabled = !(std::getenv("HP_PAIR_TUNE") && std::atoi(std::getenv("HP_PAIR_TUNE")) == 0);
	return enabled && d->desc.math_mode == HP_MATH_STRICT && two_step_mode() != 1;
}
static int tuner_poll(hp_domain* d)
{
	if (d->tune_phase == 1 && hipEventQuery(d->tune_ev[3]) == hipSuccess) {
		float pair_ms = 0.f, single_ms = 0.f;
		HIP_TRY(hipEventElapsedTime(&pair_ms, d->tune_ev[0], d->tune_ev[1]));
		HIP_TRY(hipEventElapsedTime(&single_ms, d->tune_ev[2], d->tune_ev[3]));
		const bool prefer = pair_ms < 0.98f * single_ms;
		if (prefer != d->tune_prefer_pairs) d->tune_switches++;
		d->tune_prefer_pairs = prefer;
		d->tune_pair_ms = pair_ms; d->tune_single_ms = single_ms;
		d->tune_samples++;
		d->tune_phase = 2;
		static const uint64_t period = std::getenv("HP_PAIR_TUNE_PERIOD") ? (uint64_t)std::max(12L, std::atol(std::getenv("HP_PAIR_TUNE_PERIOD"))) : 128;
		d->tune_next = d->iterations + period;
	}
	if (d->tune_phase == 2 && d->iterations >= d->tune_next) d->tune_phase = 0;
	return HP_OK;
}
int run_single(hp_domain* d, const bool followed)
{
	// (K1 FUSED) every iteration but the last carries its successor's rain / loss: between batches the buffers are
	// what the reference's are -- a download never sees rain of an iteration that has not begun
	d->fuse_next = followed;
	d->m1_valid = false;                           // (a single iteration prices one maximum: the next pair starts cold)
	d->tail_allowed = true;                        // nothing is queued between the flux launch and the advance in this loop
	int rc = dispatch_begin(d);
	d->tail_allowed = false;
	if (rc != HP_OK) return rc;
	return dispatch_end(d);
}
int run_iterations(hp_domain* d, uint32_t n_iterations)
{
	int rc;
	const bool tune = tuner_on(d);
	if (tune && (rc = tuner_poll(d)) != HP_OK) return rc;
	for (uint32_t i = 0; i < n_iterations; ++i) {
		if (tune && d->tune_phase != 0 && (i & 15u) == 15u && (rc = tuner_poll(d)) != HP_OK) return rc;   // (two event queries: host time, behind the queue)
		// (STRICT) a sample: three pairs, then six single iterations -- the first pair and the first two single iterations warm the
		// kernel's code and are not timed, the rest are bracketed by events
		if (tune && d->tune_phase == 0 && i + 12 <= n_iterations && !d->rings_differ && pair_eligible(d)) {
			for (hipEvent_t& e : d->tune_ev) if (!e) HIP_TRY(hipEventCreate(&e));
			if ((rc = pair_stamps_alloc(d)) != HP_OK) return rc;           // (the first pair's allocation and its memset are not the pair's price)
			rc = run_pair(d, false, true);
			if (rc == HP_OK) {
				HIP_TRY(hipEventRecord(d->tune_ev[0], d->stream));
				if ((rc = run_pair(d, false, true)) != HP_OK || (rc = run_pair(d, false, true)) != HP_OK) return rc;
				HIP_TRY(hipEventRecord(d->tune_ev[1], d->stream));
				if ((rc = repair_other_buffer(d)) != HP_OK) return rc;    // (not part of an iteration's price: a run of single iterations pays it once)
				if ((rc = run_single(d, true)) != HP_OK || (rc = run_single(d, true)) != HP_OK) return rc;
				HIP_TRY(hipEventRecord(d->tune_ev[2], d->stream));
				for (int k = 0; k < 4; ++k)
					if ((rc = run_single(d, k < 3 || i + 12 < n_iterations)) != HP_OK) return rc;
				HIP_TRY(hipEventRecord(d->tune_ev[3], d->stream));
				d->tune_phase = 1;
				i += 11;
				continue;
			}
			if (rc != HP_ERR_STATE) return rc;
		}
		if (i + 2 <= n_iterations && (!tune || d->tune_prefer_pairs) && pair_eligible(d)) {
			rc = run_pair(d, false, i + 2 < n_iterations);
			if (rc == HP_OK) { ++i; continue; }
			if (rc != HP_ERR_STATE) return rc;                 // (HP_ERR_STATE: not launchable as a pair -- single iterations)
		}
		if ((rc = run_single(d, i + 1 < n_iterations)) != HP_OK) return rc;
	}
	d->fuse_next = 0;
	return HP_OK;
}
int spec_resolve(hp_domain* d)
{
	const uint32_t n = d->spec_pending;
	d->spec_pending = 0;
	HIP_TRY(hipStreamSynchronize(d->stream));
	static const long force = std::getenv("HP_STRICT_SPEC_FORCE") ? std::atol(std::getenv("HP_STRICT_SPEC_FORCE")) : 0;
	const double flag = *(volatile double*)((char*)d->host_scalars + HOST_SPEC_FLAG);
	const bool raised = flag != 0.0 || (force > 0 && d->spec_batches % (uint64_t)force == 0);
	if (!raised) return HP_OK;
	// some quotient of the batch fell outside what the shared-reciprocal division covers: the batch never happened
	d->spec_replays++;
	log_line(HP_LOG_INFORMATION, "a speculative STRICT batch of " + std::to_string(n) + " iterations is re-run with the plain divisions");
	const size_t bytes = d->cells * 4 * d->esize, sc_bytes = sizeof(Scalars<double>);
	HIP_TRY(hipMemcpyAsync(d->state[0], d->spec_state, bytes, hipMemcpyDeviceToDevice, d->stream));
	HIP_TRY(hipMemcpyAsync(d->state[1], (char*)d->spec_state + bytes, bytes, hipMemcpyDeviceToDevice, d->stream));
	HIP_TRY(hipMemcpyAsync(d->scalars, d->spec_scalars, sc_bytes, hipMemcpyDeviceToDevice, d->stream));
	HIP_TRY(hipMemcpyAsync(d->cfl_slot, (char*)d->spec_scalars + sc_bytes, CFL_SLOT_BYTES, hipMemcpyDeviceToDevice, d->stream));
	d->use_alt = d->spec_host.use_alt; d->adv_fresh = d->spec_host.adv_fresh;
	d->need_full_reduce = d->spec_host.need_full_reduce; d->edge_dirty = d->spec_host.edge_dirty;
	d->ghost_valid = d->spec_host.ghost_valid;
	d->cells_calculated = d->spec_host.cells_calculated; d->iterations = d->spec_host.iterations;
	d->fork_is_advance = false;
	return run_iterations(d, n);
}
} // namespace
extern "C" {

int hp_step_batch(hp_domain_t* d, uint32_t n_iterations)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (d->in_step) return fail(HP_ERR_STATE, "inside a split step");
	const bool speculate = spec_wanted(d, n_iterations);
	if (speculate && (rc = spec_begin(d)) != HP_OK) return rc;
	rc = run_iterations(d, n_iterations);
	if (speculate) {
		d->spec_now = false;
		if (rc != HP_OK) return rc;
		d->spec_batches++;
		HIP_TRY(hipMemcpyAsync((char*)d->host_scalars + HOST_SPEC_FLAG, (char*)d->cfl_slot + (size_t)SLOT_SPEC * d->esize, 8,
		                       hipMemcpyDeviceToHost, d->stream));
		d->spec_pending = n_iterations;
	}
	return rc;
}

// The mailboxes' sticky error word (a strip that was not heard from within HP_PEER_TIMEOUT_MS: the tail / advance kernel then
// went on with a partial maximum and the ghost rows were not handed over).  Read wherever the host blocks on this domain's
// stream anyway -- hp_read_scalars, hp_sync (what a download or a checkpoint is followed by), the strips' batch-start
// handshake -- so that no host call sequence gets rasters of a broken exchange with HP_OK.  Blocks.
static int peer_error_check(hp_domain* d)
{
	if (!d->peer_agreed || !d->peer_mine) return HP_OK;
	uint64_t* peer_error = (uint64_t*)((char*)d->host_scalars + 480);
	*peer_error = 0;
	HIP_TRY(hipMemcpyAsync(peer_error, d->peer_mine + PEER_WORD_ERROR, 8, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(hipStreamSynchronize(d->stream));
	if (*peer_error)
		return fail(HP_ERR_HIP, "the exchange between the strips is incomplete: rank " + std::to_string((long)*peer_error - 1) +
		                        " was not heard from within the time limit (HP_PEER_TIMEOUT_MS); states and ghost rows after that point are not valid");
	return HP_OK;
}

// The sticky word of a launch's tail block that gave up waiting for a flux block (hp_kernels.hpp: launch_tail; time and timestep
// were left frozen from that launch on).  Read wherever the host blocks on the domain's stream anyway; once seen, every later call
// on the domain fails with HP_ERR_STATE (check_domain) -- its state is not a state of the model.  `queued`: the read-back has been
// queued by the caller (hp_read_scalars), only the verdict is left.
constexpr size_t HOST_TAIL_ERR = 448;       // byte offset in the pinned block (next to HOST_SPEC_FLAG)
static int tail_error_queue(hp_domain* d)
{
	*(volatile double*)((char*)d->host_scalars + HOST_TAIL_ERR) = 0.0;
	HIP_TRY(hipMemcpyAsync((char*)d->host_scalars + HOST_TAIL_ERR, (char*)d->cfl_slot + (size_t)SLOT_TAIL_ERR * d->esize, d->esize,
	                       hipMemcpyDeviceToHost, d->stream));
	return HP_OK;
}
static int tail_error_verdict(hp_domain* d)
{
	const unsigned char* w = (const unsigned char*)d->host_scalars + HOST_TAIL_ERR;
	bool raised = false;
	for (size_t i = 0; i < d->esize; ++i) raised = raised || w[i] != 0;
	if (!raised) return HP_OK;
	d->tail_failed = true;
	return tail_failed_error();
}

int hp_read_scalars(hp_domain_t* d, hp_scalars_t* out)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (!out) return fail(HP_ERR_INVALID, "out == NULL");
	HIP_TRY(hipMemcpyAsync(d->host_scalars, d->scalars, 128, hipMemcpyDeviceToHost, d->stream));
	if ((rc = tail_error_queue(d)) != HP_OK) return rc;
	uint64_t* peer_error = (uint64_t*)((char*)d->host_scalars + 480);
	*peer_error = 0;
	if (d->peer_agreed)                      // the mailboxes' sticky error word: a strip that was not heard from in time
		HIP_TRY(hipMemcpyAsync(peer_error, d->peer_mine + PEER_WORD_ERROR, 8, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(hipStreamSynchronize(d->stream));
	if ((rc = tail_error_verdict(d)) != HP_OK) return rc;
	if (*peer_error)
		return fail(HP_ERR_HIP, "the maximum over the strips is incomplete: rank " + std::to_string((long)*peer_error - 1) +
		                        " was not heard from within the time limit (HP_PEER_TIMEOUT_MS)");
	if (d->desc.precision == 8) {
		const Scalars<double>* s = (const Scalars<double>*)d->host_scalars;
		out->time = s->t; out->timestep = s->dt; out->time_hydrological = s->t_hydro; out->time_target = s->t_sync;
		out->batch_timesteps = s->batch_dt; out->batch_successful = s->batch_ok; out->batch_skipped = s->batch_skipped;
	} else {
		const Scalars<float>* s = (const Scalars<float>*)d->host_scalars;
		out->time = s->t; out->timestep = s->dt; out->time_hydrological = s->t_hydro; out->time_target = s->t_sync;
		out->batch_timesteps = s->batch_dt; out->batch_successful = s->batch_ok; out->batch_skipped = s->batch_skipped;
	}
	out->cells_calculated = d->cells_calculated;
	out->iterations = d->iterations;
	return HP_OK;
}

int hp_sync(hp_domain_t* d)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if ((rc = tail_error_queue(d)) != HP_OK) return rc;
	HIP_TRY(hipStreamSynchronize(d->stream));
	if ((rc = tail_error_verdict(d)) != HP_OK) return rc;                // (nor may the state of a domain whose time stands still)
	return peer_error_check(d);                                          // (a download / checkpoint of a broken exchange must not look fine)
}

int hp_is_busy(hp_domain_t* d, int* busy)
{
	if (!d || !busy) return fail(HP_ERR_INVALID, "null argument");
	hipError_t e = hipStreamQuery(d->stream);
	if (e == hipSuccess && d->stream_halo) e = hipStreamQuery(d->stream_halo);       // halo segments of a split step
	if (e == hipSuccess) {
		*busy = 0;
		// (an idle domain whose speculative batch has not been looked at yet: do that now -- the caller polls this to learn
		// whether the batch's results can be read)
		// -- and if that look re-queues the batch (the plain divisions' replay), the domain IS busy again: say so
		if (d->spec_pending) {
			hipSetDevice(d->desc.device);
			const int rc = spec_resolve(d);
			if (rc == HP_OK && hipStreamQuery(d->stream) == hipErrorNotReady) *busy = 1;
			return rc;
		}
		return HP_OK;
	}
	if (e == hipErrorNotReady) { *busy = 1; return HP_OK; }
	return fail(HP_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(e));
}

int hp_device_ptr(hp_domain_t* d, int which, void** ptr)
{
	if (!d || !ptr) return fail(HP_ERR_INVALID, "null argument");
	d->fork_is_advance = false;          // the caller may queue work on these buffers behind the last advance_time
	switch (which) {
	case HP_PTR_STATE_NEXT_SRC: *ptr = d->state[d->use_alt]; return HP_OK;
	case HP_PTR_STATE_OTHER:    *ptr = d->state[d->use_alt ^ 1]; return HP_OK;
	case HP_PTR_BED:            *ptr = d->bed; return HP_OK;
	case HP_PTR_MANNING:        *ptr = d->manning; return HP_OK;
	case HP_PTR_CFL_MAX:        *ptr = d->cfl_slot; return HP_OK;
	case HP_PTR_SCALARS:        *ptr = d->scalars; return HP_OK;
	}
	return fail(HP_ERR_INVALID, "unknown pointer id");
}

int hp_stream(hp_domain_t* d, void** hip_stream)
{
	if (!d || !hip_stream) return fail(HP_ERR_INVALID, "null argument");
	d->fork_is_advance = false;
	*hip_stream = (void*)d->stream;
	return HP_OK;
}

int hp_set_halo_overlap(hp_domain_t* d, int on)
{
	if (!d) return fail(HP_ERR_INVALID, "null domain");
	if (d->in_step) return fail(HP_ERR_STATE, "hp_set_halo_overlap inside an iteration");
	d->halo_overlap = on != 0;
	d->halo_overlap_set = true;
	return HP_OK;
}

int hp_stream_halo(hp_domain_t* d, void** hip_stream)
{
	if (!d || !hip_stream) return fail(HP_ERR_INVALID, "null argument");
	*hip_stream = (void*)d->stream_halo;
	return HP_OK;
}

} // extern "C"

// ---- RCCL, loaded at run time: the engine has no link-time dependency on a collective library, and a process that
//      already carries one (torch bundles its own librccl, bound to its own HIP runtime) keeps using that copy ----
namespace {

int rccl_ready()
{
	return g_rccl.handle ? HP_OK : fail(HP_ERR_STATE, "no collective library loaded: call hp_comm_load first");
}
#define RCCL_TRY(expr)                                                                                 \
	do {                                                                                               \
		ncclResult_t r_ = (expr);                                                                      \
		if (r_ != ncclSuccess)                                                                         \
			return fail(HP_ERR_HIP, std::string(#expr) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "RCCL error")); \
	} while (0)


// Ghost-row exchange of the iteration in flight, queued behind the flux launches of hp_step_begin: the first / last
// `ghost_rows` owned rows of the new state go to the strip neighbours' ghost rows (CDomainLink::pullFromBuffer /
// pushToBuffer, CDomainLink.cpp:168-270; CMPIManager's block exchange, CMPIManager.cpp:555-709).
int strip_halo_exchange(hp_domain* d)
{
	const long G = d->ghost_rows, rows = d->desc.rows;
	const size_t row_elems = (size_t)d->desc.cols * 4, count = (size_t)G * row_elems;
	const ncclDataType_t type = d->desc.precision == 8 ? ncclDouble : ncclFloat;
	char* state = (char*)d->state[d->use_alt ^ 1];                    // the buffer the iteration in flight writes
	const size_t row_bytes = row_elems * d->esize;
	const bool south = d->comm_rank > 0, north = d->comm_rank < d->comm_world - 1;
	// the rows the neighbours need come from the halo part of the step: the transfer is ordered after that stream only
	hipStream_t xs = (d->halo_overlap && d->split_now) ? d->stream_halo : d->stream;
	if (south || north) {
		RCCL_TRY(g_rccl.GroupStart());
		if (south) {
			RCCL_TRY(g_rccl.Send(state + (size_t)G * row_bytes, count, type, d->comm_rank - 1, d->comm, xs));            // my first owned rows
			RCCL_TRY(g_rccl.Recv(state, count, type, d->comm_rank - 1, d->comm, xs));                                     // into my south ghost rows
		}
		if (north) {
			RCCL_TRY(g_rccl.Send(state + (size_t)(rows - 2 * G) * row_bytes, count, type, d->comm_rank + 1, d->comm, xs));  // my last owned rows
			RCCL_TRY(g_rccl.Recv(state + (size_t)(rows - G) * row_bytes, count, type, d->comm_rank + 1, d->comm, xs));       // into my north ghost rows
		}
		RCCL_TRY(g_rccl.GroupEnd());
		if (xs != d->stream) {
			HIP_TRY(hipEventRecord(d->ev_xchg, xs));
			HIP_TRY(hipStreamWaitEvent(d->stream, d->ev_xchg, 0));      // hp_step_end (and the next iteration) need the rows
		}
	}
	return HP_OK;
}

template <typename T> __global__ void copy_scalar(T* to, const T* from) { *to = *from; }

// The wave-speed maximum over all strips (MPI_Allreduce(MIN dt), CMPIManager.cpp:852-861).  Whether an iteration carries a
// NEW maximum is decided from what is the same on every rank -- the scheme, quirk Q1 and the ping-pong phase, and what the
// ranks told each other at the start of the batch (does ANY of them have boundaries, or a stale maximum after an upload) --
// so either every rank enters the collective or none does, even when the host has given a boundary to one strip only.  A
// rank whose own buffer was not priced this iteration contributes the local maximum it remembers.
int strip_allreduce_max(hp_domain* d, bool first_of_batch)
{
	// (measurement knob, tools/strong_probe.py: a communicator of ONE rank normally skips the reduction; with the knob it goes
	// through it -- the library's all-reduce or its own mailbox -- so that the cost of the step can be timed on one GPU)
	static const bool lonely_too = std::getenv("HP_STRIP_REDUCE_ALWAYS") && std::atoi(std::getenv("HP_STRIP_REDUCE_ALWAYS")) != 0;
	if (d->comm_world <= 1 && !lonely_too) return HP_OK;
	if (!d->desc.dynamic_dt) {
		if (d->peer_direct) d->adv_fresh = 4;                          // fixed timestep: nothing to reduce, the round still hands the rows over
		return HP_OK;
	}
	const bool q1 = (d->desc.quirks & HP_QUIRK_CFL_READS_PRIMARY) != 0, muscl = d->desc.scheme == HP_SCHEME_MUSCL_HANCOCK;
	const bool basic = d->desc.kernel == HP_KERNEL_BASIC && d->desc.scheme == HP_SCHEME_GODUNOV;
	const bool dst_is_primary = d->use_alt == 1;
	const bool everyone = muscl || !q1 || dst_is_primary || basic || d->strip_any_bdy || (first_of_batch && d->strip_any_full);
	if (!everyone) {
		if (d->adv_fresh) return fail(HP_ERR_STATE, "a strip priced a new maximum on an iteration the other ranks do not reduce "
		                                            "(boundaries or uploads changed inside a batch?)");
		if (d->peer_direct) d->adv_fresh = 4;                          // a round all the same: it is what hands the ghost rows over (PeerPush)
		return HP_OK;
	}
	const ncclDataType_t type = d->desc.precision == 8 ? ncclDouble : ncclFloat;
	if (!d->adv_fresh) {                                               // nothing new here: the remembered local maximum stands in
		if (d->desc.precision == 8) hipLaunchKernelGGL(copy_scalar<double>, dim3(1), dim3(1), 0, d->stream, (double*)d->cfl_slot, (const double*)d->cfl_slot + SLOT_LOCAL);
		else                        hipLaunchKernelGGL(copy_scalar<float>, dim3(1), dim3(1), 0, d->stream, (float*)d->cfl_slot, (const float*)d->cfl_slot + SLOT_LOCAL);
		HIP_TRY(hipGetLastError());
	}
	if (d->peer_agreed) {                                              // the advance kernel itself trades the values with the peers
		d->adv_fresh = 7;
		return HP_OK;
	}
	RCCL_TRY(g_rccl.AllReduce(d->cfl_slot, (char*)d->cfl_slot + (size_t)SLOT_GLOBAL * d->esize, 1, type, ncclMax, d->comm, d->stream));
	d->adv_fresh = 3;                                                  // advance_time: take the maximum from SLOT_GLOBAL
	return HP_OK;
}

// Start of a batch: the ranks compare notes (one 8-element all-reduce and a read-back).  Boundaries and uploads are host
// calls, which nothing forces to be the same on every rank; the ping-pong phase and the ghost-row state must be.
int strip_handshake(hp_domain* d)
{
	d->strip_any_bdy = !d->bdy.empty() && d->desc.scheme != HP_SCHEME_MUSCL_HANCOCK;
	d->strip_any_full = d->need_full_reduce;
	d->strip_pairs = false;
	if (d->comm_world <= 1) return HP_OK;
	// (element 0 also carries the mailboxes' sticky error word of THIS rank as a value above 1: the all-reduce then tells every
	// rank that the exchange broke somewhere, and all of them fail together instead of one leaving the others in a collective)
	const bool broken = peer_error_check(d) != HP_OK;
	// (element 0: 3 = this rank's exchange is broken, 1 = it has boundary conditions, 0.5 = it has none but cannot run iteration
	// pairs, 0 = it can: the maximum over the ranks says what ALL of them may do)
	const double mine[8] = {broken ? 3.0 : (d->strip_any_bdy ? 1.0 : (strip_pairs_possible_here(d) ? 0.0 : 0.5)), d->strip_any_full ? 1.0 : 0.0, (double)d->use_alt, -(double)d->use_alt,
	                        (double)d->ghost_valid, -(double)d->ghost_valid, (double)d->ghost_rows, -(double)d->ghost_rows};
	double all[8];
	char* slot = (char*)d->cfl_slot + (size_t)SLOT_HANDSHAKE * d->esize;
	const ncclDataType_t type = d->desc.precision == 8 ? ncclDouble : ncclFloat;
	// staged through the domain's pinned block (bytes 256..511; hp_read_scalars uses the first 128): an asynchronous copy
	// must not read from, or land in, this function's stack
	char* pinned = (char*)d->host_scalars + 256;
	HIP_TRY(hipStreamSynchronize(d->stream));                           // nothing of an earlier handshake is still in flight
	if (d->desc.precision == 8) std::memcpy(pinned, mine, sizeof mine);
	else { float* f = (float*)pinned; for (int i = 0; i < 8; ++i) f[i] = (float)mine[i]; }
	HIP_TRY(hipMemcpyAsync(slot, pinned, 8 * d->esize, hipMemcpyHostToDevice, d->stream));
	RCCL_TRY(g_rccl.AllReduce(slot, slot + 8 * d->esize, 8, type, ncclMax, d->comm, d->stream));
	HIP_TRY(hipMemcpyAsync(pinned + 128, slot + 8 * d->esize, 8 * d->esize, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(hipStreamSynchronize(d->stream));
	if (d->desc.precision == 8) std::memcpy(all, pinned + 128, sizeof all);
	else { const float* f = (const float*)(pinned + 128); for (int i = 0; i < 8; ++i) all[i] = f[i]; }
	if (all[0] > 2.0)
		return fail(HP_ERR_HIP, broken ? std::string(hp_last_error())
		                               : std::string("the exchange between the strips broke on another rank (a strip was not heard from in time)"));
	if (all[2] != -all[3]) return fail(HP_ERR_STATE, "the strips disagree on the ping-pong phase (different iteration counts?)");
	if (all[4] != -all[5] || all[6] != -all[7]) return fail(HP_ERR_STATE, "the strips disagree on their ghost rows");
	d->strip_any_bdy = all[0] >= 1.0;
	d->strip_any_full = all[1] > 0.0;
	d->strip_pairs = all[0] == 0.0;
	return HP_OK;
}

// ---- peer-written mailboxes: set-up and tear-down ----
struct PeerTicket {                     // HP_PEER_TICKET_BYTES, travels between the ranks by the host's own means
	uint64_t          magic;
	uint64_t          process;          // a number drawn once per process: same number = same address space
	uint64_t          address;          // the mailbox in its owner's address space
	int32_t           device, rank;
	hipIpcMemHandle_t handle;           // ... and for everybody else
	uint64_t          state[2];         // the two state buffers (the strip neighbours write their ghost rows), likewise
	hipIpcMemHandle_t state_handle[2];
	int64_t           rows, cols, ghost_rows, row_offset;
	int32_t           precision, state_shareable;
	unsigned char     pad[HP_PEER_TICKET_BYTES - 32 - 3 * sizeof(hipIpcMemHandle_t) - 16 - 32 - 8];
};
static_assert(sizeof(PeerTicket) == HP_PEER_TICKET_BYTES, "HP_PEER_TICKET_BYTES");
constexpr uint64_t PEER_MAGIC = 0x68702d7065657231ull;           // "hp-peer1"

uint64_t process_token()
{
	static const uint64_t token = [] {
		std::random_device rd;
		return ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ ((uint64_t)getpid() << 20);
	}();
	return token;
}

void peer_release(hp_domain* d)
{
	d->peer_agreed = false;
	if (!d->peer_mine && !d->peer_table && d->peer_mapped.empty() && !d->push_arrived) return;
	if (d->stream) hipStreamSynchronize(d->stream);
	for (void* m : d->peer_mapped) hipIpcCloseMemHandle(m);
	d->peer_mapped.clear();
	d->peer_direct = false; d->push_now = false;
	for (auto& side : d->peer_state) side[0] = side[1] = nullptr;
	if (d->push_arrived) { hipFree(d->push_arrived); d->push_arrived = nullptr; }
	if (d->peer_table) { hipFree(d->peer_table); d->peer_table = nullptr; }
	if (d->peer_mine)  { hipFree(d->peer_mine);  d->peer_mine = nullptr; }
	d->peer_world = 0; d->peer_rank = 0; d->peer_rounds = 0;
}

// one reduction of a caller-given value through the mailboxes, result and error word read back
int peer_round_now(hp_domain* d, double value, long timeout_ms, double* result, uint64_t* error)
{
	const PeerBox box = peer_box(d, true, timeout_ms);
	hipLaunchKernelGGL(peer_round, dim3(1), dim3(64), 0, d->stream, box, value);
	HIP_TRY(hipGetLastError());
	uint64_t* pinned = (uint64_t*)((char*)d->host_scalars + 480);
	HIP_TRY(hipMemcpyAsync(pinned, d->peer_mine + PEER_WORD_ERROR, 16, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(hipStreamSynchronize(d->stream));
	*error = pinned[0];
	std::memcpy(result, &pinned[1], 8);
	return HP_OK;
}
} // namespace

extern "C" {

int hp_comm_load(const char* library_path)
{
	if (g_rccl.handle) return HP_OK;
	// an explicit path is the only candidate (the caller names the copy its process already uses; silently picking
	// another one could bind a second HIP runtime); NULL = the system's
	const bool given = library_path && *library_path;
	const char* candidates[] = {given ? library_path : "librccl.so", given ? nullptr : "librccl.so.1",
	                            given ? nullptr : "/opt/rocm/lib/librccl.so"};
	void* h = nullptr;
	std::string why = "not found";
	for (const char* c : candidates) {
		if (!c || !*c) continue;
		h = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
		if (h) break;
		const char* e = dlerror();           // returns the message ONCE and clears it
		if (e) why = e;
	}
	if (!h) return fail(HP_ERR_UNSUPPORTED, "cannot load the RCCL library: " + why);
	Rccl r;
	r.handle = h;
#define SYM(name) r.name = (decltype(r.name))dlsym(h, "nccl" #name); if (!r.name) { dlclose(h); return fail(HP_ERR_UNSUPPORTED, "RCCL library lacks nccl" #name); }
	SYM(GetUniqueId) SYM(CommInitRank) SYM(CommDestroy) SYM(GroupStart) SYM(GroupEnd) SYM(Send) SYM(Recv) SYM(AllReduce) SYM(GetErrorString)
#undef SYM
	r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
	Dl_info where;
	if (dladdr((void*)r.AllReduce, &where) && where.dli_fname) r.path = where.dli_fname;
	g_rccl = r;
	return HP_OK;
}

int hp_strip_info(hp_domain_t* d, hp_strip_info_t* out)
{
	if (!out) return fail(HP_ERR_INVALID, "out == NULL");
	std::memset(out, 0, sizeof *out);
	std::snprintf(out->library, sizeof out->library, "%s", g_rccl.path.c_str());
	out->comm_ranks = -1;
	if (!d) return HP_OK;                                                // library path only
	out->comm_rank = d->comm_rank;
	out->halo_overlap = d->halo_overlap ? 1 : 0;
	out->ghost_rows = (int32_t)d->ghost_rows;
	out->peer_max = d->peer_agreed ? 1 : 0;
	out->peer_halo = d->peer_direct ? 1 : 0;
	if (d->comm && g_rccl.CommCount) {
		int n = -1;
		RCCL_TRY(g_rccl.CommCount(d->comm, &n));
		out->comm_ranks = n;                                             // what the LIBRARY says, not what the caller passed
	}
	return HP_OK;
}

int hp_comm_unique_id(void* id_out)
{
	int rc = rccl_ready();
	if (rc != HP_OK) return rc;
	if (!id_out) return fail(HP_ERR_INVALID, "id_out == NULL");
	static_assert(sizeof(ncclUniqueId) == HP_COMM_ID_BYTES, "HP_COMM_ID_BYTES");
	ncclUniqueId id;
	RCCL_TRY(g_rccl.GetUniqueId(&id));
	std::memcpy(id_out, &id, sizeof id);
	return HP_OK;
}

int hp_strip_comm_init(hp_domain_t* d, const void* id, int rank, int world)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if ((rc = rccl_ready()) != HP_OK) return rc;
	if (!id || world < 1 || rank < 0 || rank >= world) return fail(HP_ERR_INVALID, "bad communicator arguments");
	if (d->comm) return fail(HP_ERR_STATE, "the domain already has a communicator");
	// the strip's place in the global grid must agree with its rank: neighbours are rank - 1 (south) and rank + 1 (north)
	const bool has_south = d->desc.row_offset > 0, has_north = d->desc.row_offset + d->desc.rows < d->desc.global_rows;
	if (has_south != (rank > 0) || has_north != (rank < world - 1))
		return fail(HP_ERR_INVALID, "rank does not match the strip's position in the global grid");
	ncclUniqueId uid;
	std::memcpy(&uid, id, sizeof uid);
	RCCL_TRY(g_rccl.CommInitRank(&d->comm, world, uid, rank));
	d->comm_rank = rank; d->comm_world = world;
	if (!d->ev_xchg) HIP_TRY(hipEventCreateWithFlags(&d->ev_xchg, hipEventDisableTiming));
	if (!d->halo_overlap_set) d->halo_overlap = world > 1;             // default only: an explicit choice stands
	return HP_OK;
}

int hp_strip_comm_destroy(hp_domain_t* d)
{
	if (!d) return HP_OK;
	peer_release(d);
	if (d->comm) {
		hipStreamSynchronize(d->stream);
		if (d->stream_halo) hipStreamSynchronize(d->stream_halo);
		if (g_rccl.CommDestroy) g_rccl.CommDestroy(d->comm);
		d->comm = nullptr;
	}
	d->comm_world = 1; d->comm_rank = 0;
	return HP_OK;
}

// ---- the maximum over all strips without a collective: peer-written mailboxes (PeerBox, hp_kernels.hpp) ----
int hp_strip_peer_ticket(hp_domain_t* d, void* ticket_out)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (!ticket_out) return fail(HP_ERR_INVALID, "ticket_out == NULL");
	if (d->in_step) return fail(HP_ERR_STATE, "inside a split step");
	peer_release(d);
	// uncached: a peer's store must be what this device's next load sees, whichever cache it would have sat in
	void* box = nullptr;
	HIP_TRY(hipExtMallocWithFlags(&box, PEER_WORDS * sizeof(unsigned long long), hipDeviceMallocUncached));
	d->peer_mine = (unsigned long long*)box;
	HIP_TRY(hipMemsetAsync(d->peer_mine, 0xff, 2 * PEER_MAX_RANKS * sizeof(unsigned long long), d->stream));       // every word EMPTY
	HIP_TRY(hipMemsetAsync(d->peer_mine + PEER_WORD_ERROR, 0, (PEER_WORDS - PEER_WORD_ERROR) * sizeof(unsigned long long), d->stream));
	HIP_TRY(hipStreamSynchronize(d->stream));                            // ... before anybody learns where it is
	PeerTicket t;
	std::memset(&t, 0, sizeof t);
	t.magic = PEER_MAGIC; t.process = process_token(); t.address = (uint64_t)(uintptr_t)d->peer_mine;
	t.device = d->desc.device; t.rank = -1;
	if (hipIpcGetMemHandle(&t.handle, d->peer_mine) != hipSuccess) {
		(void)hipGetLastError();                                         // other processes cannot map it; ranks of THIS process still can
		std::memset(&t.handle, 0, sizeof t.handle);
	}
	t.state_shareable = 1;
	for (int b = 0; b < 2; ++b) {
		t.state[b] = (uint64_t)(uintptr_t)d->state[b];
		if (hipIpcGetMemHandle(&t.state_handle[b], d->state[b]) != hipSuccess) {
			(void)hipGetLastError();
			std::memset(&t.state_handle[b], 0, sizeof t.state_handle[b]);
			t.state_shareable = 0;
		}
	}
	t.rows = d->desc.rows; t.cols = d->desc.cols; t.ghost_rows = d->ghost_rows; t.row_offset = d->desc.row_offset;
	t.precision = d->desc.precision;
	std::memcpy(ticket_out, &t, sizeof t);
	return HP_OK;
}

int hp_strip_peer_connect(hp_domain_t* d, const void* tickets, int count, int rank, int* active)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (active) *active = 0;
	// This call is COLLECTIVE over the communicator's ranks: whatever is wrong on this rank -- arguments included -- it must
	// still reach the agreement below, or the other ranks wait in that all-reduce for ever.  A local fault
	// therefore only marks this rank's verdict as "no"; the error code is returned after the agreement.
	std::string fault;
	if (!tickets || count < 1 || count > PEER_MAX_RANKS || rank < 0 || rank >= count)
		fault = "bad mailbox arguments (1.." + std::to_string(PEER_MAX_RANKS) + " ranks)";
	else if (!d->peer_mine) fault = "hp_strip_peer_connect without hp_strip_peer_ticket";
	else if (d->comm && (count != d->comm_world || rank != d->comm_rank)) fault = "mailbox ranks do not match the communicator's";
	const PeerTicket* t = (const PeerTicket*)tickets;
	if (fault.empty() && (t[rank].magic != PEER_MAGIC || t[rank].address != (uint64_t)(uintptr_t)d->peer_mine || t[rank].process != process_token()))
		fault = "tickets[rank] is not this domain's ticket";
	if (!fault.empty() && !d->comm) return fail(HP_ERR_INVALID, fault);   // nobody to keep waiting
	if (!fault.empty()) count = 0;                                       // nothing is mapped below
	// map every peer's mailbox; a failure here is not an error of the call: the connection test below fails on every
	// rank alike (the others never hear from this one) and the run stays on the collective
	std::vector<unsigned long long*> table((size_t)(count > 0 ? count : 1), nullptr);
	bool mapped = fault.empty();
	std::string why = fault;
	for (int r = 0; r < count && mapped; ++r) {
		if (t[r].magic != PEER_MAGIC) { mapped = false; why = "ticket " + std::to_string(r) + " is not a ticket"; break; }
		if (t[r].process == process_token()) {                           // same address space (ranks as threads, or one process driving several GPUs)
			table[(size_t)r] = (unsigned long long*)(uintptr_t)t[r].address;
			if (t[r].device != d->desc.device) {
				const hipError_t e = hipDeviceEnablePeerAccess(t[r].device, 0);
				if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { mapped = false; why = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e); }
				(void)hipGetLastError();
			}
		} else {
			void* m = nullptr;
			const hipError_t e = hipIpcOpenMemHandle(&m, t[r].handle, hipIpcMemLazyEnablePeerAccess);
			if (e != hipSuccess) { (void)hipGetLastError(); mapped = false; why = std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e); }
			else { d->peer_mapped.push_back(m); table[(size_t)r] = (unsigned long long*)m; }
		}
	}
	d->peer_world = count > 0 ? count : 1; d->peer_rank = fault.empty() ? rank : 0; d->peer_rounds = 0;
	// the strip neighbours' state buffers, for the ghost rows (PeerPush).  Optional on top of the mailboxes: when any rank
	// cannot have it the rows keep travelling through the collective library's send / receive
	const bool direct_wanted = !(std::getenv("HP_PEER_DIRECT") && std::atoi(std::getenv("HP_PEER_DIRECT")) == 0);
	bool direct = mapped && direct_wanted && d->comm != nullptr;
	std::string why_not_direct = direct_wanted ? "" : "switched off (HP_PEER_DIRECT=0)";
	for (int side = 0; side < 2 && direct; ++side) {
		const int r = side == 0 ? rank - 1 : rank + 1;
		if (r < 0 || r >= count) continue;
		const PeerTicket& n = t[r];
		if (n.cols != d->desc.cols || n.precision != d->desc.precision || n.ghost_rows != d->ghost_rows || d->own_hi - d->own_lo < d->ghost_rows ||
		    (side == 1 && n.row_offset != d->desc.row_offset + d->desc.rows - 2 * d->ghost_rows) ||
		    (side == 0 && n.row_offset + n.rows - 2 * d->ghost_rows != d->desc.row_offset)) {
			direct = false; why_not_direct = "rank " + std::to_string(r) + "'s strip does not adjoin this one"; break;
		}
		d->peer_rows[side] = (long)n.rows;
		for (int b = 0; b < 2 && direct; ++b) {
			if (n.process == process_token()) {
				d->peer_state[side][b] = (void*)(uintptr_t)n.state[b];
				if (n.device != d->desc.device) {
					const hipError_t e = hipDeviceEnablePeerAccess(n.device, 0);
					if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { direct = false; why_not_direct = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e); }
					(void)hipGetLastError();
				}
			} else if (!n.state_shareable) {
				direct = false; why_not_direct = "rank " + std::to_string(r) + " could not export its state buffers";
			} else {
				void* m = nullptr;
				const hipError_t e = hipIpcOpenMemHandle(&m, n.state_handle[b], hipIpcMemLazyEnablePeerAccess);
				if (e != hipSuccess) { (void)hipGetLastError(); direct = false; why_not_direct = std::string("hipIpcOpenMemHandle (state): ") + hipGetErrorString(e); }
				else { d->peer_mapped.push_back(m); d->peer_state[side][b] = m; }
			}
		}
	}
	// (from here to the agreement nothing returns early: a rank that left now would leave the others waiting in the all-reduce)
	if (direct && !d->push_arrived) {
		if (hipMalloc((void**)&d->push_arrived, 64) != hipSuccess || hipMemset(d->push_arrived, 0, 64) != hipSuccess) {
			(void)hipGetLastError();
			direct = false; why_not_direct = "no memory for the push counter";
		}
	}
	uint64_t error = 1;
	double got = 0.0;
	if (mapped && !d->peer_table && hipMalloc((void**)&d->peer_table, PEER_MAX_RANKS * sizeof(unsigned long long*)) != hipSuccess) {
		(void)hipGetLastError();
		mapped = false; why = "no memory for the mailbox table";
	}
	if (mapped && hipMemcpy(d->peer_table, table.data(), (size_t)count * sizeof(unsigned long long*), hipMemcpyHostToDevice) != hipSuccess) {
		(void)hipGetLastError();
		mapped = false; why = "mailbox table upload failed";
	}
	if (mapped) {
		// connection test: two reductions (one per mailbox set) of rank + 1 and -(rank + 1) must give `count` and -1
		const long test_ms = std::getenv("HP_PEER_TEST_MS") ? std::atol(std::getenv("HP_PEER_TEST_MS")) : 3000;
		bool ok = peer_round_now(d, (double)(rank + 1), test_ms, &got, &error) == HP_OK && error == 0 && got == (double)count;
		if (ok) ok = peer_round_now(d, -(double)(rank + 1), test_ms, &got, &error) == HP_OK && error == 0 && got == -1.0;
		if (!ok && why.empty()) why = error ? "rank " + std::to_string((long)error - 1) + " was not heard from" : "wrong maximum, or the test launch failed";
		error = ok ? 0 : 1;
	}
	if (error) log_line(HP_LOG_WARNING, "peer-written maximum not available on rank " + std::to_string(rank) + ": " + why);
	if (!error && !direct && d->comm) log_line(HP_LOG_INFORMATION, "ghost rows stay with the collective library on rank " + std::to_string(rank) + ": " + why_not_direct);
	if (d->comm) {
		// every rank must take the same road: one MAX over the ranks' verdicts through the collective library
		double verdict[2] = {error ? 1.0 : 0.0, direct ? 0.0 : 1.0};
		if (d->comm_world > 1) {
			char* slot = (char*)d->cfl_slot + (size_t)SLOT_HANDSHAKE * d->esize;
			char* pinned = (char*)d->host_scalars + 256;
			const ncclDataType_t type = d->desc.precision == 8 ? ncclDouble : ncclFloat;
			if (d->desc.precision == 8) std::memcpy(pinned, verdict, 16);
			else { const float v[2] = {(float)verdict[0], (float)verdict[1]}; std::memcpy(pinned, v, 8); }
			HIP_TRY(hipMemcpyAsync(slot, pinned, 2 * d->esize, hipMemcpyHostToDevice, d->stream));
			RCCL_TRY(g_rccl.AllReduce(slot, slot + 8 * d->esize, 2, type, ncclMax, d->comm, d->stream));
			HIP_TRY(hipMemcpyAsync(pinned + 128, slot + 8 * d->esize, 2 * d->esize, hipMemcpyDeviceToHost, d->stream));
			HIP_TRY(hipStreamSynchronize(d->stream));
			if (d->desc.precision == 8) std::memcpy(verdict, pinned + 128, 16);
			else { float f[2]; std::memcpy(f, pinned + 128, 8); verdict[0] = f[0]; verdict[1] = f[1]; }
		}
		d->peer_agreed = verdict[0] == 0.0;
		d->peer_direct = d->peer_agreed && verdict[1] == 0.0;
		if (active) *active = d->peer_agreed ? (d->peer_direct ? 2 : 1) : 0;
		if (!d->peer_agreed) peer_release(d);
		else if (!d->peer_direct) for (auto& side : d->peer_state) side[0] = side[1] = nullptr;
		if (!fault.empty()) return fail(HP_ERR_INVALID, fault);            // (every rank has left the agreement by now)
	} else {
		// no communicator: nothing to agree through (diagnostic use, hp_strip_peer_round); the strip loop is not touched
		if (active) *active = error ? 0 : 1;
		if (error) peer_release(d);
	}
	return HP_OK;
}

int hp_strip_peer_round(hp_domain_t* d, double value, double* max_out)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (!max_out) return fail(HP_ERR_INVALID, "max_out == NULL");
	if (!d->peer_mine || !d->peer_table || d->peer_world < 1) return fail(HP_ERR_STATE, "no mailboxes connected");
	if (d->in_step) return fail(HP_ERR_STATE, "inside a split step");
	uint64_t error = 0;
	if ((rc = peer_round_now(d, value, 0, max_out, &error)) != HP_OK) return rc;
	if (error) return fail(HP_ERR_HIP, "peer-written maximum: rank " + std::to_string((long)error - 1) + " was not heard from in time");
	return HP_OK;
}

int hp_strip_peer_disconnect(hp_domain_t* d)
{
	if (!d) return HP_OK;
	if (d->in_step) return fail(HP_ERR_STATE, "inside a split step");
	peer_release(d);
	return HP_OK;
}

int hp_strip_step_batch(hp_domain_t* d, uint32_t n_iterations)
{
	const bool fork_ready = d && d->fork_is_advance;
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (!d->comm) return fail(HP_ERR_STATE, "hp_strip_step_batch without hp_strip_comm_init");
	if (d->in_step) return fail(HP_ERR_STATE, "inside a split step");
	if (n_iterations == 0) return HP_OK;
	const long g = strip_ghosts(d);
	if ((rc = strip_handshake(d)) != HP_OK) return rc;
	d->fork_is_advance = fork_ready && d->comm_world <= 1;            // (the handshake queued work behind the last advance_time)
	// (a maximum that some rank has yet to price anew -- an upload since the last batch -- keeps the ranks on single iterations
	// until the first iteration that prices on every rank has run: a decision every rank takes from the same, handshaken facts)
	bool full_pending = d->strip_any_full;
	for (uint32_t i = 0; i < n_iterations; ++i) {
		if (i + 2 <= n_iterations && d->strip_pairs && !full_pending && d->use_alt == 0 && d->ghost_valid == d->ghost_rows) {
			rc = run_pair(d, true);
			if (rc != HP_OK) return rc == HP_ERR_STATE ? fail(HP_ERR_STATE, "a strip could not launch an iteration pair the ranks had agreed on") : rc;
			++i;
			d->fork_is_advance = false;
			continue;
		}
		if (d->use_alt == 1) full_pending = false;    // this iteration writes the primary buffer: every rank prices it
		d->fuse_next = i + 1 < n_iterations;          // as hp_step_batch: the rows sent to the neighbours carry the rain too
		if (d->ghost_valid < g) return fail(HP_ERR_STATE, "ghost rows exhausted");
		// an iteration consumes g layers of ghost rows; when fewer than g are left afterwards, the new state's rows are
		// exchanged (ghost_rows = g: every iteration; 2g: every second one, and only those iterations are split into a
		// halo and an interior launch)
		const bool exchange = d->ghost_valid - g < g;
		d->split_now = exchange && !d->peer_direct;   // (rows that travel inside the advance kernel need no launch of their own)
		d->push_now = exchange && d->peer_direct;     // ... and leave with the advance kernel, or with the flux launch's tail block
		d->strip_first = i == 0;
		d->tail_allowed = true;
		rc = dispatch_begin(d);
		d->tail_allowed = false;
		if (rc != HP_OK) return rc;
		if (exchange) {
			if (!d->peer_direct && (rc = strip_halo_exchange(d)) != HP_OK) return rc;
			d->ghost_valid = d->ghost_rows;
		} else {
			d->ghost_valid -= g;
		}
		if ((rc = strip_allreduce_max(d, i == 0)) != HP_OK) return rc;
		if ((rc = dispatch_end(d)) != HP_OK) return rc;
		// the exchange sits between advance_time and the next flux launch on the stream graph: its event, not
		// advance_time's, is what the next halo launch may fork from only if nothing else was queued -- keep it simple
		// and let hp_step_begin record its own fork event whenever a transfer was queued on the domain's stream
		if (!d->halo_overlap || d->peer_direct) d->fork_is_advance = false;
	}
	d->fuse_next = 0;
	d->split_now = true;
	return HP_OK;
}

int hp_strip_update_timestep(hp_domain_t* d)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (!d->comm) return fail(HP_ERR_STATE, "hp_strip_update_timestep without hp_strip_comm_init");
	if (d->in_step) return fail(HP_ERR_STATE, "inside a split step");
	// tst_Reduce over the owned rows of the primary buffer, the maximum over all strips, then tst_UpdateTimestep on every
	// rank redundantly (CSchemeGodunov.cpp:1189-1195, :1254-1260 with CMPIManager's reduction in between)
	const ncclDataType_t type = d->desc.precision == 8 ? ncclDouble : ncclFloat;
	// with the mailboxes the advance kernel WAITS for the other strips' kernels (for a bounded time): launch it only once every
	// rank's host has arrived here -- the library's all-reduce of the handshake waits without a limit, a strip still busy with
	// a long upload is no error
	if (d->peer_agreed && d->comm_world > 1 && (rc = strip_handshake(d)) != HP_OK) return rc;
	const void* priced = d->desc.scheme == HP_SCHEME_MUSCL_HANCOCK ? d->state[d->use_alt] : d->state[0];     // (as hp_update_timestep)
	if (d->desc.precision == 8) {
		if (d->desc.dynamic_dt && (rc = launch_reduce<double>(d, priced, d->own_lo, d->own_hi)) != HP_OK) return rc;
	} else {
		if (d->desc.dynamic_dt && (rc = launch_reduce<float>(d, priced, d->own_lo, d->own_hi)) != HP_OK) return rc;
	}
	const bool reduced = d->desc.dynamic_dt && d->comm_world > 1, by_peers = reduced && d->peer_agreed;
	if (reduced && !by_peers)
		RCCL_TRY(g_rccl.AllReduce(d->cfl_slot, (char*)d->cfl_slot + (size_t)SLOT_GLOBAL * d->esize, 1, type, ncclMax, d->comm, d->stream));
	const int fresh = by_peers ? 7 : reduced ? 3 : 1;
	const PeerBox box = peer_box(d, by_peers);
	if (d->desc.precision == 8)
		hipLaunchKernelGGL((advance_time<true, double>), dim3(1), dim3(64), 0, d->stream, make_params<double>(d),
		                   (Scalars<double>*)d->scalars, (double*)d->cfl_slot, fresh, box, PeerPush{});
	else
		hipLaunchKernelGGL((advance_time<true, float>), dim3(1), dim3(64), 0, d->stream, make_params<float>(d),
		                   (Scalars<float>*)d->scalars, (float*)d->cfl_slot, fresh, box, PeerPush{});
	HIP_TRY(hipGetLastError());
	return HP_OK;
}

} // extern "C"

extern "C" {

int hp_timer_start(hp_domain_t* d)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	HIP_TRY(hipEventRecord(d->ev_start, d->stream));
	return HP_OK;
}

int hp_timer_stop(hp_domain_t* d, float* elapsed_ms)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (!elapsed_ms) return fail(HP_ERR_INVALID, "elapsed_ms == NULL");
	HIP_TRY(hipEventRecord(d->ev_stop, d->stream));
	HIP_TRY(hipEventSynchronize(d->ev_stop));
	HIP_TRY(hipEventElapsedTime(elapsed_ms, d->ev_start, d->ev_stop));
	return HP_OK;
}

int hp_kernel_timing(hp_domain_t* d, int enable_stride)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	HIP_TRY(hipStreamSynchronize(d->stream));
	constexpr size_t POOL = 16;                                          // samples per measurement: sparse on purpose
	while (enable_stride > 0 && d->timing_events.size() < POOL) {
		hipEvent_t a, b;
		HIP_TRY(hipEventCreate(&a));
		HIP_TRY(hipEventCreate(&b));
		d->timing_events.emplace_back(a, b);
	}
	// What a pair of events costs by itself (two marker packets with nothing between them): since an iteration became ONE launch
	// the step time has nothing but that launch in it, and a sample that includes the markers' own time came out ABOVE the step
	// time.  Measured here, on the idle stream, as the smallest of eight empty pairs, and taken off every sample when they are read.
	d->timing_overhead_ms = 0.0;
	if (enable_stride > 0 && !d->timing_events.empty()) {
		float best = 1e9f;
		for (int i = 0; i < 8; ++i) {
			HIP_TRY(hipEventRecord(d->timing_events[0].first, d->stream));
			HIP_TRY(hipEventRecord(d->timing_events[0].second, d->stream));
			HIP_TRY(hipEventSynchronize(d->timing_events[0].second));
			float ms = 0.f;
			HIP_TRY(hipEventElapsedTime(&ms, d->timing_events[0].first, d->timing_events[0].second));
			if (ms < best) best = ms;
		}
		d->timing_overhead_ms = best < 1e8f ? best : 0.0;
	}
	d->timing_used = 0;
	d->timing_counter = 0;
	d->timing_stride = enable_stride > 0 ? enable_stride : 0;
	return HP_OK;
}

int hp_kernel_timing_read(hp_domain_t* d, double* avg_ms, uint32_t* samples)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (!avg_ms || !samples) return fail(HP_ERR_INVALID, "null argument");
	HIP_TRY(hipStreamSynchronize(d->stream));
	double total = 0.0;
	uint32_t n = 0;
	for (size_t i = 0; i < d->timing_used; ++i) {
		float ms = 0.f;
		HIP_TRY(hipEventElapsedTime(&ms, d->timing_events[i].first, d->timing_events[i].second));
		total += std::max(0.0, (double)ms - d->timing_overhead_ms);
		++n;
	}
	*avg_ms = n ? total / n : 0.0;
	*samples = n;
	return HP_OK;
}

int hp_pair_stats(hp_domain_t* d, uint64_t out[12])
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (!out) return fail(HP_ERR_INVALID, "out == NULL");
	out[0] = d->pairs; out[1] = d->pair_cold_starts; out[2] = out[3] = 0;
	HIP_TRY(hipStreamSynchronize(d->stream));
	if (tuner_on(d) && (rc = tuner_poll(d)) != HP_OK) return rc;
	out[4] = d->tune_samples; out[5] = d->tune_switches; out[6] = d->tune_prefer_pairs ? 1 : 0;
	out[7] = d->tune_single_ms > 0.f ? (uint64_t)(1000.0 * d->tune_pair_ms / d->tune_single_ms) : 0;
	out[8] = out[9] = out[10] = out[11] = 0;
	if (d->haz_words) {
		unsigned long long used = 0;
		HIP_TRY(hipMemcpy(&used, d->haz_words + 2, sizeof used, hipMemcpyDeviceToHost));
		out[8] = used;
	}
	if (!d->z_state) return HP_OK;
	// (counted on the device: the records are 48 bytes a cell)
	unsigned long long* counts = d->haz_words + 4;                        // (words 4 and 5 of the 64-byte block: 0-1 the launches' words, 2 the audit)
	HIP_TRY(hipMemsetAsync(counts, 0, 16, d->stream));
	if (d->desc.precision == 8)
		hipLaunchKernelGGL((stamps_count<double>), dim3(1024), dim3(256), 0, d->stream, StampBufs<double>{(char*)d->z_state, d->haz_words}, d->pair_gen, d->cells, counts);
	else
		hipLaunchKernelGGL((stamps_count<float>), dim3(1024), dim3(256), 0, d->stream, StampBufs<float>{(char*)d->z_state, d->haz_words}, d->pair_gen, d->cells, counts);
	HIP_TRY(hipGetLastError());
	unsigned long long host_counts[2] = {0, 0};
	HIP_TRY(hipMemcpyAsync(host_counts, counts, sizeof host_counts, hipMemcpyDeviceToHost, d->stream));
	HIP_TRY(hipStreamSynchronize(d->stream));
	out[2] = host_counts[0]; out[3] = host_counts[1];
	return HP_OK;
}

int hp_launch_counts(hp_domain_t* d, uint64_t* flux_launches, uint64_t* with_tail)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (!flux_launches || !with_tail) return fail(HP_ERR_INVALID, "null argument");
	*flux_launches = d->flux_launches;
	*with_tail = d->flux_launches_tailed;
	return HP_OK;
}

int hp_kernel_timing_overhead(hp_domain_t* d, double* overhead_ms)
{
	int rc = check_domain(d);
	if (rc != HP_OK) return rc;
	if (!overhead_ms) return fail(HP_ERR_INVALID, "null argument");
	*overhead_ms = d->timing_overhead_ms;
	return HP_OK;
}

} // extern "C"
