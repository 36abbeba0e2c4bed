// run_model_strips.cpp -- CModel::runModelMain (src/CModel.cpp:1041-1139) over several CSchemeMI in strip mode, driven by
// ONE host thread: the shape of the program the adapter drops into -- several <domain deviceNumber=...> of one process
// (Domain/CDomainManager.cpp:203-220), every domain's batch on its scheme's own worker (CSchemeGodunov.cpp:1116-1139), the
// main thread assessing, synchronising, writing outputs, scheduling every idle domain in turn and polling isRunning()
// (runModelDomainAssess :552-697, runModelSync :775-852, runModelOutputs :869-891, runModelSchedule :906-955).
// ONE grid is cut into `world` row strips; the per-iteration ghost rows and the timestep reduction are the library's
// (hp_strip_step_batch), so the model-level synchronisation that is left is the output times.
// The strips share one GPU here, so the collective library has to allow that: the tests pass tests/fake_rccl.
//   usage: run_model_strips <collective library> <world> <cols> <rows> <duration_s> <output_frequency_s> [godunov|muscl] [batch]
// Output (stdout, one line per output time):  t  iterations  volume  checksum(Z)     -- as run_strips
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "hp_scheme.hpp"

using namespace hipims_mi;

int main(int argc, char** argv)
{
	if (argc < 7) { std::fprintf(stderr, "usage: run_model_strips <lib> <world> <cols> <rows> <duration> <freq> [godunov|muscl] [batch]\n"); return 1; }
	const char* lib = argv[1];
	const int world = std::atoi(argv[2]);
	const long cols = std::atol(argv[3]), rows = std::atol(argv[4]);
	const double dSimulationTime = std::atof(argv[5]), dOutputFrequency = std::atof(argv[6]);
	const bool muscl = argc > 7 && std::strcmp(argv[7], "muscl") == 0;
	const unsigned batch = argc > 8 ? (unsigned)std::atoi(argv[8]) : 25;
	const long g = muscl ? 2 : 1;                                   // ghost rows per interior side

	if (hp_comm_load(lib) != HP_OK) { std::fprintf(stderr, "hp_comm_load: %s\n", hp_last_error()); return 2; }
	char id[HP_COMM_ID_BYTES];
	if (hp_comm_unique_id(id) != HP_OK) { std::fprintf(stderr, "hp_comm_unique_id: %s\n", hp_last_error()); return 2; }
	CSchemeMI::setLogSink([](int level, const char* msg, void*) { if (level >= HP_LOG_WARNING) std::fprintf(stderr, "[hipims_mi level %d] %s\n", level, msg); }, nullptr);

	// ---- CDomainManager: the domain set (one DomainArrays + one scheme per strip) ----
	struct Strip { long own_lo, own_hi, lo, hi; DomainArrays dom; std::unique_ptr<CSchemeMI> scheme; };
	std::vector<Strip> domains((size_t)world);
	std::vector<CSchemeMI*> schemes;
	for (int r = 0; r < world; ++r) {
		Strip& s = domains[(size_t)r];
		s.own_lo = (long)r * rows / world; s.own_hi = (long)(r + 1) * rows / world;
		s.lo = std::max(0L, s.own_lo - g); s.hi = std::min(rows, s.own_hi + g);
		s.dom.resize(cols, s.hi - s.lo);
		s.dom.resolution = 1.0;
		for (long y = s.lo; y < s.hi; ++y)
			for (long x = 0; x < cols; ++x) {
				const size_t i = (size_t)(y - s.lo) * cols + x;
				const bool edge = x == 0 || y == 0 || x == cols - 1 || y == rows - 1;
				const double z = edge ? 0.0 : (x < cols / 2 ? 10.0 : 1.0);
				s.dom.cellStates[4 * i] = z; s.dom.cellStates[4 * i + 1] = z;
				s.dom.manningValues[i] = 0.03;
			}
		s.dom.closeEdges(s.lo == 0, s.hi == rows);
		s.scheme.reset(new CSchemeMI(muscl ? schemeTypes::kMUSCLHancock : schemeTypes::kGodunov, &s.dom));
		s.scheme->setSimulationLength(dSimulationTime);
		if (batch > 0) { s.scheme->setQueueMode(false); s.scheme->setQueueSize(batch); }   // else: the several-domains formula
		s.scheme->setStrip(r, world, id, rows, s.lo);
		schemes.push_back(s.scheme.get());
	}
	const bool peer_max = !(std::getenv("HIPIMS_MI_PEER_MAX") && std::atoi(std::getenv("HIPIMS_MI_PEER_MAX")) == 0) && world > 1;
	const int level = CSchemeMI::prepareStripSet(schemes, peer_max);          // the collective part of the set-up, all strips at once
	if (level < 0) {
		for (int r = 0; r < world; ++r) if (!schemes[(size_t)r]->isReady()) std::fprintf(stderr, "strip %d: %s\n", r, schemes[(size_t)r]->lastError().c_str());
		return 3;
	}
	std::fprintf(stderr, "maximum over the strips: %s; ghost rows: %s\n", level >= 1 ? "peer-written mailboxes" : "all-reduce",
	             level >= 2 ? "written by the strips" : "send / receive");

	// ---- CModel::runModelPrepare (:499-523) ----
	for (CSchemeMI* s : schemes) s->prepareSimulation();
	bool   bSynchronised = true, bAllIdle = true, bRollbackRequired = false;
	double dTargetTime = 0.0, dLastSyncTime = -1.0, dLastOutputTime = 0.0, dCurrentTime = 0.0, dEarliestTime = 0.0;
	std::vector<char> bSyncReady((size_t)world, 0), bIdle((size_t)world, 0);
	const auto t0 = std::chrono::steady_clock::now();
	unsigned long polls = 0, schedules = 0;

	// ---- CModel::runModelMain (:1069-1108): one thread, no blocking call inside the loop ----
	while (dCurrentTime < dSimulationTime - 1E-5 || !bAllIdle) {
		// runModelDomainAssess (:552-697)
		bRollbackRequired = false;
		dEarliestTime = 0.0;
		for (int i = 0; i < world; ++i) {
			CSchemeMI* sch = schemes[(size_t)i];
			const bool running = sch->isRunning();                  // (acquire: the statistics below are the finished batch's)
			bIdle[(size_t)i] = !running;
			if (running) { bSyncReady[(size_t)i] = 0; continue; }   // (the reference reads a running scheme's stale time; an idle one's here)
			if (!sch->isReady()) { std::fprintf(stderr, "strip %d failed: %s\n", i, sch->lastError().c_str()); std::_Exit(3); }
			if (dEarliestTime == 0.0 || dEarliestTime > sch->getCurrentTime()) dEarliestTime = sch->getCurrentTime();
			if (!sch->isSimulationSyncReady(dTargetTime) || bSynchronised || dLastSyncTime == dEarliestTime) {
				bSyncReady[(size_t)i] = 0;
				if (sch->isSimulationFailure(dTargetTime)) bRollbackRequired = true;
			} else bSyncReady[(size_t)i] = 1;
		}
		bSynchronised = true; bAllIdle = true;
		for (int i = 0; i < world; ++i) { if (!bSyncReady[(size_t)i]) bSynchronised = false; if (!bIdle[(size_t)i]) bAllIdle = false; }
		if (bAllIdle) dCurrentTime = dEarliestTime;
		if (bRollbackRequired) { std::fprintf(stderr, "a strip ran past the target time\n"); return 4; }

		// runModelSync (:775-852)
		if (bSynchronised && bAllIdle) {
			dCurrentTime = dEarliestTime;
			dLastSyncTime = dCurrentTime;
			// runModelOutputs (:869-891)
			const bool output_due = std::fabs(dCurrentTime - dLastOutputTime - dOutputFrequency) < 1E-5 && dCurrentTime > dLastOutputTime;
			if (output_due) {
				double v = 0.0, sum = 0.0;
				for (Strip& s : domains) {
					s.scheme->saveCurrentState();                       // (:836-846: only when an output is due)
					for (long y = s.own_lo; y < s.own_hi; ++y)
						for (long x = 0; x < cols; ++x) {
							const size_t i = (size_t)(y - s.lo) * cols + x;
							sum += s.dom.cellStates[4 * i];
							const double h = s.dom.cellStates[4 * i] - s.dom.bedElevations[i];
							if (h > 0.0 && s.dom.bedElevations[i] < 9999.0) v += h;
						}
				}
				std::printf("%.9f %u %.9f %.12e\n", schemes[0]->getCurrentTime(), schemes[0]->getIterationsSuccessful(), v, sum);
				dLastOutputTime = dCurrentTime;
				for (CSchemeMI* s : schemes) s->forceTimeAdvance();
			}
			// runModelUpdateTarget (:725-770).  The strips' exchange is the library's, every iteration: nothing at model level has to
			// be synchronised between outputs ("otherwise run free, for as long as possible (i.e. until outputs needed)")
			double dEarliestSyncProposal = dSimulationTime;
			if (std::floor(dEarliestSyncProposal / dOutputFrequency) > std::floor(dLastSyncTime / dOutputFrequency))
				dEarliestSyncProposal = (std::floor(dLastSyncTime / dOutputFrequency) + 1) * dOutputFrequency;
			dTargetTime = dEarliestSyncProposal;
		}

		// runModelSchedule (:906-955): every idle domain gets its next batch; runSimulation returns at once
		const double real = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		for (int i = 0; i < world; ++i)
			if (!bSynchronised && bIdle[(size_t)i]) { schemes[(size_t)i]->runSimulation(dTargetTime, real); ++schedules; }
		++polls;
		if (!bAllIdle) std::this_thread::yield();                  // (the reference's loop spins; be polite to the workers on a small host)
	}
	std::fprintf(stderr, "main thread: %lu passes of the management loop, %lu batches scheduled, no strip driven from a thread of its own\n", polls, schedules);
	for (CSchemeMI* s : schemes) s->cleanupSimulation();
	return 0;
}
