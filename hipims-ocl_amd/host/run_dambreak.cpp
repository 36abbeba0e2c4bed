// run_dambreak.cpp -- a CModel::runModelMain-shaped driver (src/CModel.cpp:1041-1139, reduced to one domain) over
// CSchemeMI: run a closed-basin dam break to successive output times and print what CModel's progress box reports.
//   usage: run_dambreak <cols> <rows> <duration_s> <output_frequency_s> [godunov|muscl] [fixed_batch_size (0 = automatic queue)]
//                       [inflow]   (a CBoundaryCell inflow hydrograph on a column of cells next to the west wall)
// Output (stdout, one line per output time):  t  iterations  cells_calculated  volume  checksum(Z)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hp_scheme.hpp"

using namespace hipims_mi;

int main(int argc, char** argv)
{
	const long cols = argc > 1 ? std::atol(argv[1]) : 512, rows = argc > 2 ? std::atol(argv[2]) : 256;
	const double duration = argc > 3 ? std::atof(argv[3]) : 5.0, freq = argc > 4 ? std::atof(argv[4]) : 1.0;
	const bool muscl = argc > 5 && std::strcmp(argv[5], "muscl") == 0;
	const unsigned fixedBatch = argc > 6 ? (unsigned)std::atoi(argv[6]) : 0;
	const bool inflow = argc > 7 && std::strcmp(argv[7], "inflow") == 0;

	DomainArrays dom;
	dom.resize(cols, rows);
	dom.resolution = 1.0;
	for (long y = 0; y < rows; ++y)
		for (long x = 0; x < cols; ++x) {
			const size_t i = (size_t)y * cols + x;
			const bool edge = x == 0 || y == 0 || x == cols - 1 || y == rows - 1;
			const double z = edge ? 0.0 : (x < cols / 2 ? 10.0 : 1.0);
			dom.cellStates[4 * i] = z; dom.cellStates[4 * i + 1] = z;
			dom.manningValues[i] = 0.03;
		}
	dom.closeEdges();

	CSchemeMI scheme(muscl ? schemeTypes::kMUSCLHancock : schemeTypes::kGodunov, &dom);
	scheme.setSimulationLength(duration);
	if (fixedBatch > 0) { scheme.setQueueMode(false); scheme.setQueueSize(fixedBatch); }      // <scheme queueMode="fixed" queueSize=...>
	if (inflow) {
		// <boundary type="cell" depthValue="ignore" dischargeValue="volume">: 40 m3/s rising to 120 m3/s, shared by the column's cells
		std::vector<uint64_t> cells;
		for (long y = rows / 4; y < 3 * rows / 4; ++y) cells.push_back((uint64_t)y * cols + 1);
		const std::vector<double> series = {0.0, 0.0, 40.0, 0.0,  1.0, 0.0, 120.0, 0.0,  2.0, 0.0, 120.0, 0.0};
		scheme.addBoundaryCell(HP_DEPTH_IGNORE, HP_DISCHARGE_IS_VOLUME, cells, series, 1.0, 2.0);
	}
	CSchemeMI::setLogSink([](int level, const char* msg, void*) { std::fprintf(stderr, "[hipims_mi level %d] %s\n", level, msg); }, nullptr);
	scheme.prepareAll();
	if (!scheme.isReady()) { std::fprintf(stderr, "prepareAll failed: %s\n", scheme.lastError().c_str()); return 2; }
	scheme.prepareSimulation();

	const auto t0 = std::chrono::steady_clock::now();
	double target = freq;
	while (scheme.getCurrentTime() < duration - 1e-9) {
		const double real = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		scheme.runSimulation(target, real);                       // runModelSchedule (:906-961): returns at once, the batch runs on the scheme's worker
		scheme.waitUntilIdle();                                   // (one domain, nothing else to schedule: block instead of polling isRunning())
		if (!scheme.isReady()) { std::fprintf(stderr, "step failed: %s\n", scheme.lastError().c_str()); return 3; }
		if (scheme.isSimulationSyncReady(target)) {               // runModelSync (:775-868) -> outputs, next target
			scheme.saveCurrentState();
			double sum = 0.0;
			for (size_t i = 0; i < dom.cellCount(); ++i) sum += dom.cellStates[4 * i];
			std::printf("%.9f %u %llu %.9f %.12e\n", scheme.getCurrentTime(), scheme.getIterationsSuccessful(),
			            scheme.getCellsCalculated(), dom.volume(), sum);
			target = std::min(duration, target + freq);
		}
	}
	const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	std::fprintf(stderr, "rate: %.1f Mcell-steps/s over %.3f s wall\n", scheme.getCellsCalculated() / wall / 1e6, wall);
	scheme.cleanupSimulation();
	return 0;
}
