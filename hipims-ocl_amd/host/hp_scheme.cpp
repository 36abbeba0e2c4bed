// hp_scheme.cpp -- see hp_scheme.hpp.  Behaviour follows src/Schemes/CSchemeGodunov.cpp of the reference, its worker
// thread included (runBatchThread / Threaded_runBatch, :1116-1139, :1147-1372): runSimulation returns at once with
// bRunning set, the scheme's own thread queues the batch (hp_step_batch / hp_strip_step_batch), blocks in the read of the
// key statistics (hp_read_scalars: clFinish + five reads there) and clears bRunning -- so one host thread can drive
// several strips, as CModel's main loop drives several domains.
#include "hp_scheme.hpp"

#include <algorithm>
#include <chrono>
#include <utility>
#include <cmath>
#include <cstring>

namespace hipims_mi {

void DomainArrays::closeEdges(bool south, bool north)
{
	for (long x = 0; x < cols; ++x) {
		if (south) bedElevations[x] = 9999.9;
		if (north) bedElevations[(size_t)(rows - 1) * cols + x] = 9999.9;
	}
	for (long y = 0; y < rows; ++y) { bedElevations[(size_t)y * cols] = 9999.9; bedElevations[(size_t)y * cols + cols - 1] = 9999.9; }
}

double DomainArrays::volume() const
{
	double v = 0.0;
	for (size_t i = 0; i < cellCount(); ++i) {
		const double h = cellStates[4 * i] - bedElevations[i];
		if (h > 0.0 && bedElevations[i] < 9999.0) v += h * resolution * resolution;
	}
	return v;
}

CSchemeMI::CSchemeMI(unsigned char schemeType, DomainArrays* domain, int deviceNumber)
	: pDomain(domain), ucScheme(schemeType), iDevice(deviceNumber) {}

CSchemeMI::~CSchemeMI() { cleanupSimulation(); }

bool CSchemeMI::check(int rc, const char* what)
{
	if (rc == HP_OK) return true;
	sLastError = std::string(what) + ": " + hp_last_error();      // model::doError(..., kLevelModelStop) in the reference
	bReady = false;
	return false;
}

void CSchemeMI::addBoundaryUniform(int definition, const std::vector<double>& pairs, double interval, double length)
{
	boundaries.push_back({0, definition, pairs, pairs.size() / 2, 0, 0, interval, length, 0, 0, 0});
}

void CSchemeMI::addBoundaryGridded(int definition, const std::vector<double>& grids, uint64_t entries, uint64_t gridRows,
                                   uint64_t gridCols, double resolution, double offsetX, double offsetY, double interval)
{
	boundaries.push_back({1, definition, grids, entries, gridRows, gridCols, interval, 0, resolution, offsetX, offsetY});
}

void CSchemeMI::addBoundaryCell(int depthDefinition, int dischargeDefinition, const std::vector<uint64_t>& cells,
                                const std::vector<double>& series, double interval, double length)
{
	PendingBoundary b{2, depthDefinition, series, series.size() / 4, 0, 0, interval, length, 0, 0, 0};
	b.dischargeDefinition = dischargeDefinition;
	b.cells = cells;
	boundaries.push_back(std::move(b));
}

void CSchemeMI::setStrip(int rank, int world, const void* commId, long globalRows, long rowOffset)
{
	bStrip = true; iStripRank = rank; iStripWorld = world; lGlobalRows = globalRows; lRowOffset = rowOffset;
	std::memcpy(cCommId, commId, HP_COMM_ID_BYTES);
	// batch sizes must agree across the ranks: the automatic queue of a strip uses the reference's several-domains
	// formula (runSimulation below), which is made of simulated quantities only
}

// prepare1OExecDimensions / Constants / Code / Memory / Kernels / Boundaries collapse into one descriptor
void CSchemeMI::prepareAll()
{
	hp_domain_desc_t desc;
	hp_domain_desc_default(&desc);
	desc.device = iDevice - 1;                                        // deviceNumber is 1-based (CDomainManager.cpp:203-220)
	desc.cols = pDomain->cols; desc.rows = pDomain->rows; desc.dx = pDomain->resolution;
	desc.precision = 8;
	desc.scheme = (ucScheme == schemeTypes::kMUSCLHancock)           ? HP_SCHEME_MUSCL_HANCOCK
	            : (ucScheme == schemeTypes::kInertialSimplification) ? HP_SCHEME_INERTIAL
	                                                                 : HP_SCHEME_GODUNOV;
	desc.courant = dCourantNumber;
	desc.dry_threshold = dThresholdVerySmall;
	desc.friction = bFrictionEffects ? 1 : 0;
	desc.dynamic_dt = bDynamicTimestep ? 1 : 0;
	desc.dt_fixed = dTimestep; desc.dt_initial = dTimestep;           // CSchemeGodunov.cpp:862-867
	desc.t_end = dSimulationLength;
	desc.math_mode = iMathMode;
	if (bStrip) { desc.global_rows = lGlobalRows; desc.row_offset = lRowOffset; }
	if (!check(hp_domain_create(&desc, &hpDomain), "hp_domain_create")) return;
	for (const PendingBoundary& b : boundaries) {
		const int rc = b.kind == 0
			? hp_boundary_add_uniform(hpDomain, b.definition, b.data.data(), (uint32_t)b.entries, b.interval, b.length)
			: b.kind == 1
			? hp_boundary_add_gridded(hpDomain, b.definition, b.data.data(), b.entries, b.rows, b.cols, b.resolution,
			                          b.offx, b.offy, b.interval)
			: hp_boundary_add_cell(hpDomain, b.definition, b.dischargeDefinition, b.cells.data(), b.cells.size(),
			                       b.data.data(), b.entries, b.interval, b.length);
		if (!check(rc, "hp_boundary_add")) return;
	}
	if (bStrip && !check(hp_strip_comm_init(hpDomain, cCommId, iStripRank, iStripWorld), "hp_strip_comm_init")) return;
	dCurrentTimestep = dTimestep;
	bReady = true;
}

bool CSchemeMI::getPeerTicket(void* ticketOut)
{
	return bReady && bStrip && check(hp_strip_peer_ticket(hpDomain, ticketOut), "hp_strip_peer_ticket");
}

int CSchemeMI::connectPeers(const void* tickets)
{
	int active = 0;
	if (!bReady || !bStrip || !check(hp_strip_peer_connect(hpDomain, tickets, iStripWorld, iStripRank, &active), "hp_strip_peer_connect"))
		return 0;
	return active;
}

int CSchemeMI::prepareStripSet(const std::vector<CSchemeMI*>& strips, bool peerMax)
{
	const size_t n = strips.size();
	std::vector<std::thread> th;
	for (CSchemeMI* s : strips) th.emplace_back([s] { s->prepareAll(); });       // collective: the communicator
	for (auto& t : th) t.join();
	for (CSchemeMI* s : strips) if (!s->isReady()) return -1;
	if (!peerMax || n < 2) return 0;
	std::vector<char> tickets(n * HP_PEER_TICKET_BYTES, 0);
	for (size_t r = 0; r < n; ++r) if (!strips[r]->getPeerTicket(&tickets[r * HP_PEER_TICKET_BYTES])) return -1;
	std::vector<int> level(n, 0);
	th.clear();
	for (size_t r = 0; r < n; ++r) th.emplace_back([&, r] { level[r] = strips[r]->connectPeers(tickets.data()); });   // collective: test + agreement
	for (auto& t : th) t.join();
	for (CSchemeMI* s : strips) if (!s->isReady()) return -1;
	return *std::min_element(level.begin(), level.end());
}

void CSchemeMI::prepareSimulation()
{
	if (!bReady) return;
	const size_t n = pDomain->cellCount();
	if (!check(hp_domain_upload(hpDomain, HP_ARRAY_BED, pDomain->bedElevations.data(), n * 8), "upload bed")) return;
	if (!check(hp_domain_upload(hpDomain, HP_ARRAY_MANNING, pDomain->manningValues.data(), n * 8), "upload manning")) return;
	if (!check(hp_domain_upload(hpDomain, HP_ARRAY_STATE, pDomain->cellStates.data(), n * 32), "upload state")) return;
	check(hp_sync(hpDomain), "hp_sync");                              // blockUntilFinished (:1071)
	bOverrideTimestep = false; bUseForcedTimeAdvance = true; bCellStatesSynced = true;
	dBatchStartedTime = 0.0; ulCurrentCellsCalculated = 0; uiIterationsSinceSync = 0; dLastSyncTime = 0.0;
	bRunning.store(false);
}

void CSchemeMI::setTargetTime(double t)
{
	if (t == dTargetTime) return;
	dTargetTime = t;
	bUpdateTargetTime = true;
}

void CSchemeMI::forceTimestep(double dt)
{
	if (dt == dCurrentTimestep) return;
	dCurrentTimestep = dt;
	bOverrideTimestep = true;
}

void CSchemeMI::runSimulation(double dTarget, double dRealTime)
{
	if (!bReady || isRunning()) return;
	int busy = 0;
	if (hp_is_busy(hpDomain, &busy) != HP_OK || busy) return;         // :1377-1378
	if (dTargetTime != dTarget) setTargetTime(dTarget);               // :1381-1382
	if (dTarget <= 0.0) return;                                       // :1385-1386
	if (dCurrentTime > dTarget + 1E-5) return;                        // :1389-1407 (warning in the reference)

	// batch size: aim for about a second of work, no silly jumps, never beyond the rollback limit (:1420-1448)
	if (bAutomaticQueue && dRealTime > 1E-5 && ucSyncMethod != syncMethod::kSyncTimestep) {
		const double dBatchDuration = dRealTime - dBatchStartedTime;
		const unsigned int uiOld = uiQueueAdditionSize;
		if (bStrip) {
			// getDomainCount() > 1 (:1428-1429): the iterations left to the target at the batch's mean timestep, plus one.
			// Time, target, timestep sum and iteration count are the same numbers on every rank, so is the result.
			if (uiBatchSuccessful > 0 && dBatchTimesteps > 0.0)
				uiQueueAdditionSize = (unsigned int)((dTargetTime - dCurrentTime) / (dBatchTimesteps / (double)uiBatchSuccessful) + 1.0);
		} else
		uiQueueAdditionSize = std::max(1u, std::min(uiBatchRate * 3,
			(unsigned int)std::ceil(1.0 / (dBatchDuration / (double)uiQueueAdditionSize))));
		if (uiQueueAdditionSize > uiOld * 2 && uiQueueAdditionSize > 40)
			uiQueueAdditionSize = std::min(uiBatchRate * 3, uiOld * 2);
		if (uiQueueAdditionSize > uiRollbackLimit - uiIterationsSinceSync)
			uiQueueAdditionSize = uiRollbackLimit - uiIterationsSinceSync;
		if (uiQueueAdditionSize < 1) uiQueueAdditionSize = 1;
	}
	dBatchStartedTime = dRealTime;
	{
		std::lock_guard<std::mutex> l(mtxWorker);
		bRunning.store(true, std::memory_order_release);
	}
	runBatchThread();                                                 // :1452; returns at once
}

// runBatchThread (:1116-1139): the worker is created on the first batch and kept ("because of the overhead associated with
// creating a thread", :1149-1150); later calls only wake it
void CSchemeMI::runBatchThread()
{
	if (!bThreadRunning) {
		bThreadRunning = true;
		thWorker = std::thread([this] { Threaded_runBatch(); });
	}
	cvWorker.notify_one();
}

void CSchemeMI::Threaded_runBatch()
{
	for (;;) {
		{
			std::unique_lock<std::mutex> l(mtxWorker);                // "Are we expected to run?" (:1153-1161), asleep instead of spinning
			cvWorker.wait(l, [this] { return !bThreadRunning || bRunning.load(std::memory_order_acquire); });
			if (!bThreadRunning) return;
		}
		runBatch();
		bRunning.store(false, std::memory_order_release);             // "Wait until further work is scheduled" (:1366)
		cvWorker.notify_all();                                        // (waitUntilIdle)
	}
}

void CSchemeMI::waitUntilIdle()
{
	std::unique_lock<std::mutex> l(mtxWorker);
	cvWorker.wait_for(l, std::chrono::milliseconds(1), [this] { return !bRunning.load(std::memory_order_acquire); });
	while (bRunning.load(std::memory_order_acquire)) cvWorker.wait_for(l, std::chrono::milliseconds(1));
}

// ---- Threaded_runBatch (:1147-1372), one pass of its loop body; runs on the worker thread ----
void CSchemeMI::runBatch()
{
	if (bUpdateTargetTime) {                                          // :1163-1209
		bUpdateTargetTime = false;
		check(hp_set_target_time(hpDomain, dTargetTime), "hp_set_target_time");
		bCellStatesSynced = false;
		uiIterationsSinceSync = 0;
		bUseForcedTimeAdvance = true;
		if (dCurrentTimestep <= 0.0 && ucSyncMethod == syncMethod::kSyncForecast)
			check(bStrip ? hp_strip_update_timestep(hpDomain) : hp_update_timestep(hpDomain), "hp_update_timestep");   // tst_Reduce + tst_UpdateTimestep
		if (dCurrentTime + dCurrentTimestep > dTargetTime + 1E-5) {
			dCurrentTimestep = dTargetTime - dCurrentTime;
			bOverrideTimestep = true;
		}
	}
	if (dCurrentTime < dTargetTime && bOverrideTimestep) {            // :1213-1232
		check(hp_force_timestep(hpDomain, dCurrentTimestep), "hp_force_timestep");
		bOverrideTimestep = false;
	}
	unsigned int uiQueueAmount = uiQueueAdditionSize;
	if (ucSyncMethod == syncMethod::kSyncTimestep) uiQueueAmount = 1; // :1274-1276
	if (uiIterationsSinceSync < uiRollbackLimit && dCurrentTime < dTargetTime) {   // :1285-1304
		check(bStrip ? hp_strip_step_batch(hpDomain, uiQueueAmount) : hp_step_batch(hpDomain, uiQueueAmount), "hp_step_batch");
		uiIterationsSinceSync += uiQueueAmount;
		bCellStatesSynced = false;
	}
	readKeyStatistics();                                              // :1309-1313 + blockUntilFinished + :1350
}

void CSchemeMI::readKeyStatistics()
{
	const unsigned int uiLast = uiBatchSuccessful;
	hp_scalars_t s;
	if (!check(hp_read_scalars(hpDomain, &s), "hp_read_scalars")) return;
	dCurrentTimestep = s.timestep; dCurrentTime = s.time; dBatchTimesteps = s.batch_timesteps;
	uiBatchSuccessful = s.batch_successful; uiBatchSkipped = s.batch_skipped;
	uiBatchRate = uiBatchSuccessful > uiLast ? (uiBatchSuccessful - uiLast) : 1;   // :1834
	ulCurrentCellsCalculated = s.cells_calculated;
}

void CSchemeMI::readDomainAll()
{
	if (!hpDomain) return;
	check(hp_domain_download(hpDomain, HP_ARRAY_STATE, pDomain->cellStates.data(), 0, pDomain->rows), "hp_domain_download");
	check(hp_sync(hpDomain), "hp_sync");
}

void CSchemeMI::saveCurrentState()
{
	readDomainAll();
	uiIterationsSinceSync = 0;
	bCellStatesSynced = true;
}

void CSchemeMI::rollbackSimulation(double dTime, double dTarget)
{
	check(hp_sync(hpDomain), "hp_sync");
	uiIterationsSinceSync = 0;
	dCurrentTime = dTime; dTargetTime = dTarget;
	check(hp_set_time(hpDomain, dTime), "hp_set_time");
	check(hp_set_target_time(hpDomain, dTarget), "hp_set_target_time");
	check(hp_domain_upload(hpDomain, HP_ARRAY_STATE, pDomain->cellStates.data(), pDomain->cellCount() * 32), "upload state");
	if (ucSyncMethod != syncMethod::kSyncTimestep)
		check(bStrip ? hp_strip_update_timestep(hpDomain) : hp_update_timestep(hpDomain), "hp_update_timestep");
	bUseForcedTimeAdvance = true;
	check(hp_reset_counters(hpDomain), "hp_reset_counters");
	check(hp_sync(hpDomain), "hp_sync");
	readKeyStatistics();
}

bool CSchemeMI::isSimulationFailure(double dExpected)
{
	if (isRunning()) return false;
	if (ucSyncMethod == syncMethod::kSyncForecast && uiBatchSuccessful >= uiRollbackLimit && dExpected - dCurrentTime > 1E-5) return true;
	if (ucSyncMethod == syncMethod::kSyncTimestep && uiBatchSuccessful > uiRollbackLimit) return true;
	if (dCurrentTime > dExpected + 1E-5) return true;
	return false;
}

bool CSchemeMI::isSimulationSyncReady(double dExpected)
{
	if (isRunning()) return false;
	if (ucSyncMethod != syncMethod::kSyncTimestep && dExpected - dCurrentTime > 1E-5) return false;
	if (ucSyncMethod == syncMethod::kSyncTimestep && uiIterationsSinceSync < uiRollbackLimit - 1 &&
	    dExpected - dCurrentTime > 1E-5 && dCurrentTime > 0.0) return false;
	return true;
}

double CSchemeMI::proposeSyncPoint(double dTime)
{
	double dProposal = dTime + std::fabs(dTimestep);
	if (dTime > 1E-5 && uiBatchSuccessful > 0) {
		dProposal = dTime + std::max(std::fabs(dTimestep),
			uiRollbackLimit * (dBatchTimesteps / uiBatchSuccessful) * (((double)uiRollbackLimit - 3.0) / uiRollbackLimit));
		if (uiBatchSuccessful >= uiRollbackLimit) dProposal = dTime + dBatchTimesteps * 0.95;
	} else if (dProposal - dTime < 1E-5) {
		dProposal = dTime + std::fabs(dTimestep);
	}
	return dProposal;
}

void CSchemeMI::cleanupSimulation()
{
	// "Kill the worker thread ... wait for the thread to terminate before returning" (:1458-1469): a batch in flight finishes first
	if (thWorker.joinable()) {
		waitUntilIdle();
		{
			std::lock_guard<std::mutex> l(mtxWorker);
			bThreadRunning = false;
		}
		cvWorker.notify_all();
		thWorker.join();
	}
	bThreadRunning = false;
	if (hpDomain) { hp_domain_destroy(hpDomain); hpDomain = nullptr; }
	bRunning.store(false); bReady = false; dBatchStartedTime = 0.0;
}

} // namespace hipims_mi
