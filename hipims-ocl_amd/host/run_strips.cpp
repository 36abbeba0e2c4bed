// run_strips.cpp -- the multi-domain shape of CModel::runModelMain (src/CModel.cpp:1041-1139 with CDomainManager's
// domain set, CDomainLink and CMPIManager) over CSchemeMI in strip mode: ONE grid cut into `world` row strips, one
// CSchemeMI per strip, the per-iteration ghost-row exchange and timestep reduction inside the library
// (hp_strip_step_batch).  In production each strip is a process on its own GPU and the communicator id travels by
// MPI_Bcast; here the ranks are THREADS of this process sharing one GPU, so the collective library has to be one that
// allows that -- the tests pass tests/fake_rccl (RCCL itself refuses two ranks on one device).
//   usage: run_strips <collective library> <world> <cols> <rows> <duration_s> <output_frequency_s> [godunov|muscl] [batch]
// Output (stdout, one line per output time):  t  iterations  volume  checksum(Z)     -- same columns as run_dambreak's
// 1st, 2nd, 4th and 5th, summed over the rows each rank OWNS.
#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "hp_scheme.hpp"

using namespace hipims_mi;

namespace {
struct Rendezvous {                       // all ranks meet at every output time (what CModel's sync point is)
	std::mutex m; std::condition_variable cv; int world, arrived = 0; unsigned long gen = 0;
	void wait() {
		std::unique_lock<std::mutex> l(m);
		const unsigned long g = gen;
		if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); } else cv.wait(l, [&] { return gen != g; });
	}
};
}

int main(int argc, char** argv)
{
	if (argc < 7) { std::fprintf(stderr, "usage: run_strips <lib> <world> <cols> <rows> <duration> <freq> [godunov|muscl] [batch]\n"); return 1; }
	const char* lib = argv[1];
	const int world = std::atoi(argv[2]);
	const long cols = std::atol(argv[3]), rows = std::atol(argv[4]);
	const double duration = std::atof(argv[5]), freq = std::atof(argv[6]);
	const bool muscl = argc > 7 && std::strcmp(argv[7], "muscl") == 0;
	const unsigned batch = argc > 8 ? (unsigned)std::atoi(argv[8]) : 25;
	const long g = muscl ? 2 : 1;                                   // ghost rows per interior side

	if (hp_comm_load(lib) != HP_OK) { std::fprintf(stderr, "hp_comm_load: %s\n", hp_last_error()); return 2; }
	char id[HP_COMM_ID_BYTES];
	if (hp_comm_unique_id(id) != HP_OK) { std::fprintf(stderr, "hp_comm_unique_id: %s\n", hp_last_error()); return 2; }

	const int outputs = (int)(duration / freq + 0.5);
	std::vector<std::vector<double>> vol(world, std::vector<double>(outputs, 0.0)), sum(world, std::vector<double>(outputs, 0.0));
	std::vector<double> times(outputs, 0.0);
	std::vector<unsigned> iters(outputs, 0);
	std::vector<int> failed(world, 0);
	Rendezvous meet; meet.world = world;
	// the maximum over the strips: peer-written mailboxes unless HIPIMS_MI_PEER_MAX=0 (then the collective library's all-reduce)
	const bool peer_max = !(std::getenv("HIPIMS_MI_PEER_MAX") && std::atoi(std::getenv("HIPIMS_MI_PEER_MAX")) == 0) && world > 1;
	std::vector<char> tickets((size_t)world * HP_PEER_TICKET_BYTES, 0);
	std::vector<int> peers_active(world, 0);

	auto rank_main = [&](const int r) {
		const long own_lo = (long)r * rows / world, own_hi = (long)(r + 1) * rows / world;
		const long lo = std::max(0L, own_lo - g), hi = std::min(rows, own_hi + g);
		DomainArrays dom;
		dom.resize(cols, hi - lo);
		dom.resolution = 1.0;
		for (long y = lo; y < hi; ++y)
			for (long x = 0; x < cols; ++x) {
				const size_t i = (size_t)(y - lo) * cols + x;
				const bool edge = x == 0 || y == 0 || x == cols - 1 || y == rows - 1;
				const double z = edge ? 0.0 : (x < cols / 2 ? 10.0 : 1.0);
				dom.cellStates[4 * i] = z; dom.cellStates[4 * i + 1] = z;
				dom.manningValues[i] = 0.03;
			}
		dom.closeEdges(lo == 0, hi == rows);

		CSchemeMI scheme(muscl ? schemeTypes::kMUSCLHancock : schemeTypes::kGodunov, &dom);
		scheme.setSimulationLength(duration);
		if (batch > 0) { scheme.setQueueMode(false); scheme.setQueueSize(batch); }      // else: the automatic queue (several-domains formula)
		scheme.setStrip(r, world, id, rows, lo);
		scheme.prepareAll();
		if (!scheme.isReady()) { std::fprintf(stderr, "rank %d prepareAll failed: %s\n", r, scheme.lastError().c_str()); failed[r] = 1; }
		if (peer_max && !failed[r] && !scheme.getPeerTicket(&tickets[(size_t)r * HP_PEER_TICKET_BYTES])) failed[r] = 1;
		meet.wait();                                            // (the tickets have travelled: shared memory stands in for the host's broadcast)
		if (std::any_of(failed.begin(), failed.end(), [](int f) { return f != 0; })) return;
		if (peer_max) peers_active[r] = scheme.connectPeers(tickets.data());
		scheme.prepareSimulation();

		double target = freq;
		int out = 0;
		double pass = 0.0;                                    // stands in for the wall clock: only "has time passed" matters to a strip
		while (out < outputs) {
			scheme.runSimulation(target, pass += 1.0);
			scheme.waitUntilIdle();                                   // this driver has a host thread per strip (run_model_strips.cpp has ONE for all)
			if (!scheme.isReady()) {                                  // the other ranks are inside a collective: leave as a process
				std::fprintf(stderr, "rank %d step failed: %s\n", r, scheme.lastError().c_str());
				std::_Exit(3);
			}
			if (scheme.isSimulationSyncReady(target)) {
				scheme.saveCurrentState();
				double s = 0.0, v = 0.0;
				for (long y = own_lo; y < own_hi; ++y)
					for (long x = 0; x < cols; ++x) {
						const size_t i = (size_t)(y - lo) * cols + x;
						s += dom.cellStates[4 * i];
						const double h = dom.cellStates[4 * i] - dom.bedElevations[i];
						if (h > 0.0 && dom.bedElevations[i] < 9999.0) v += h;
					}
				sum[r][out] = s; vol[r][out] = v;
				if (r == 0) { times[out] = scheme.getCurrentTime(); iters[out] = scheme.getIterationsSuccessful(); }
				++out;
				target = std::min(duration, target + freq);
				meet.wait();                                        // every strip has reached the output time
			}
		}
		scheme.cleanupSimulation();
	};

	std::vector<std::thread> threads;
	for (int r = 0; r < world; ++r) threads.emplace_back(rank_main, r);
	for (auto& t : threads) t.join();
	if (std::any_of(failed.begin(), failed.end(), [](int f) { return f != 0; })) return 3;
	const int level = peer_max ? *std::min_element(peers_active.begin(), peers_active.end()) : 0;
	std::fprintf(stderr, "maximum over the strips: %s; ghost rows: %s\n", level >= 1 ? "peer-written mailboxes" : "all-reduce",
	             level >= 2 ? "written by the strips" : "send / receive");
	for (int o = 0; o < outputs; ++o) {
		double v = 0.0, s = 0.0;
		for (int r = 0; r < world; ++r) { v += vol[r][o]; s += sum[r][o]; }
		std::printf("%.9f %u %.9f %.12e\n", times[o], iters[o], v, s);
	}
	return 0;
}
