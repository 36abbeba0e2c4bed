// hp_scheme.hpp -- C++ host side of the path above the C ABI: the part of HiPIMS's `CScheme` surface
// (src/Schemes/CScheme.h:82-129) that `CModel` and `CDomain*` call to drive a domain, re-implemented over
// libhipims_mi.so.  Method names, argument meaning and the batch / sync-point behaviour follow
// CSchemeGodunov (src/Schemes/CSchemeGodunov.cpp); file parsing, logging and the OpenCL executor are not here.
//
// This is what INTEGRATION.md's adapter looks like when it does not depend on the rest of HiPIMS (boost, GDAL,
// TinyXML): host arrays in the reference's layout stand in for CDomain (src/Domain/CDomain.cpp:143-191).
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hipims_mi.h"

namespace hipims_mi {

// CDomain's host arrays for one Cartesian domain (CDomain.h:26-33, CDomain.cpp:171-180): fp64 only here.
struct DomainArrays {
	long cols = 0, rows = 0;
	double resolution = 1.0;
	std::vector<double> cellStates;      // cols*rows x {Z, Zmax, Qx, Qy}, row 0 = south
	std::vector<double> bedElevations;   // cols*rows
	std::vector<double> manningValues;   // cols*rows
	void resize(long c, long r) {
		cols = c; rows = r;
		cellStates.assign((size_t)c * r * 4, 0.0);
		bedElevations.assign((size_t)c * r, 0.0);
		manningValues.assign((size_t)c * r, 0.0);
	}
	size_t cellCount() const { return (size_t)cols * rows; }
	// CDomainCartesian::imposeBoundaryModification (Cartesian/CDomainCartesian.cpp:773-799): closed edges; a row strip
	// of a larger grid closes its south / north edge only where that is the grid's edge
	void closeEdges(bool south = true, bool north = true);
	// CDomain::getVolume
	double volume() const;
};

namespace schemeTypes { enum : unsigned char { kGodunov = 0, kMUSCLHancock = 1, kInertialSimplification = 2 }; }   // CScheme.h:33-37
namespace timestepMode { enum : unsigned char { kCFL = 0, kFixed = 1 }; }                       // CScheme.h:48-52
namespace syncMethod { enum : unsigned char { kSyncTimestep = 0, kSyncForecast = 1 }; }        // CScheme.h:57-62

class CSchemeMI {
public:
	CSchemeMI(unsigned char schemeType, DomainArrays* domain, int deviceNumber = 1);
	~CSchemeMI();

	// ---- configuration (CScheme::setupFromConfig, CScheme.cpp:60-135; CSchemeGodunov.cpp:128-333) ----
	void   setCourantNumber(double c)        { dCourantNumber = c; }
	double getCourantNumber() const          { return dCourantNumber; }
	void   setTimestepMode(unsigned char m)  { bDynamicTimestep = (m == timestepMode::kCFL); }
	void   setTimestep(double dt)            { dTimestep = dt; }
	void   setFrictionStatus(bool b)         { bFrictionEffects = b; }
	void   setDryThreshold(double d)         { dThresholdVerySmall = d; }
	void   setQueueMode(bool automatic)      { bAutomaticQueue = automatic; }
	void   setQueueSize(unsigned int n)      { uiQueueAdditionSize = n; }
	void   setSimulationLength(double t)     { dSimulationLength = t; }
	void   setSyncMethod(unsigned char m)    { ucSyncMethod = m; }
	void   setRollbackLimit(unsigned int n)  { uiRollbackLimit = n; }
	void   setMathMode(int m)                { iMathMode = m; }
	// CBoundaryUniform / CBoundaryGridded (atmospheric time series) and CBoundaryCell (per-cell level / discharge series,
	// Boundaries/CBoundaryCell.cpp:140-330: `cells` are ids into cellStates, series = entries x {t, depth/FSL, Qx|Q, Qy})
	void   addBoundaryUniform(int definition, const std::vector<double>& timeValuePairs, double interval, double length);
	void   addBoundaryGridded(int definition, const std::vector<double>& grids, uint64_t entries, uint64_t gridRows,
	                          uint64_t gridCols, double resolution, double offsetX, double offsetY, double interval);
	void   addBoundaryCell(int depthDefinition, int dischargeDefinition, const std::vector<uint64_t>& cells,
	                       const std::vector<double>& series, double interval, double length);
	// Row-strip decomposition (the reference's multi-domain set + CDomainLink + CMPIManager, CDomainManager.cpp:139-260):
	// this scheme is rank `rank` of `world` strips of ONE grid of `globalRows` rows; its DomainArrays hold rows
	// [rowOffset, rowOffset + rows) including the ghost rows next to its neighbours.  `commId` is the HP_COMM_ID_BYTES
	// blob of hp_comm_unique_id, obtained on rank 0 and handed to every rank by the host's own means (MPI_Bcast in the
	// reference's world).  The per-iteration ghost-row exchange and the all-reduce of the wave-speed maximum then run
	// inside the library (hp_strip_step_batch).  Every rank must ask for the same batch sizes: the automatic queue
	// (sized from each process's own wall clock) is off in this mode.
	void   setStrip(int rank, int world, const void* commId, long globalRows, long rowOffset);
	// The maximum over the strips through peer-written mailboxes instead of the collective library's all-reduce
	// (hp_strip_peer_*): after prepareAll every rank hands out its ticket (HP_PEER_TICKET_BYTES), the host gathers them in
	// rank order -- as it distributed commId -- and every rank calls connectPeers with all of them.  Returns what all ranks
	// agreed on: 2 = ghost rows and maxima are written by the strips into each other's memory (nothing of the collective
	// library inside an iteration), 1 = the maxima only, 0 = everything stays with the library.
	bool   getPeerTicket(void* ticketOut);
	int    connectPeers(const void* tickets);
	// Several strips driven by ONE host thread of ONE process (the reference's several <domain deviceNumber=...> of one
	// CDomainManager, CDomainManager.cpp:203-220): the collective parts of the set-up -- prepareAll's communicator
	// (ncclCommInitRank blocks until every rank has joined) and connectPeers' connection test and agreement -- cannot be
	// called strip after strip from one thread.  prepareStripSet runs them for all strips of this process at once (one
	// short-lived thread per strip, as ncclCommInitAll does inside the collective library) and returns the transport level
	// all strips agreed on (connectPeers' value; `peerMax` = false keeps everything on the collective library), or -1 on failure.
	static int prepareStripSet(const std::vector<CSchemeMI*>& strips, bool peerMax = true);
	// model::doError's place (main.cpp:631-652): every failure of the library is also handed to this sink
	static void setLogSink(hp_log_sink_t sink, void* user) { hp_set_log_sink(sink, user); }

	// ---- the CScheme virtuals CModel drives (CScheme.h:82-129) ----
	// Threading contract of the reference (CSchemeGodunov.cpp:1116-1139, :1147-1372, :1374-1453): runSimulation
	// sizes the batch, sets bRunning and returns AT ONCE; the scheme's own worker thread (runBatchThread, kept alive between
	// batches) queues the batch, blocks until the device has finished and the key statistics are back, and clears bRunning.
	// The caller -- CModel's single main thread, which schedules every idle domain in turn (CModel.cpp:906-955) -- polls
	// isRunning(); everything else here (setTargetTime, saveCurrentState, rollbackSimulation, readDomainAll ...) is called
	// only while the scheme is idle.  In strip mode that is what lets ONE host thread drive several strips: each strip's
	// batch (handshake, hp_strip_step_batch, hp_read_scalars) blocks its own worker, never the caller.
	bool   isReady() const                   { return bReady; }
	bool   isRunning() const                 { return bRunning.load(std::memory_order_acquire); }
	void   waitUntilIdle();                                      // a host without a polling loop: COCLDevice::blockUntilFinished + "until !bRunning"
	void   prepareAll();                                         // CSchemeGodunov::prepareAll (:386-470)
	void   prepareSimulation();                                  // :1053-1092
	void   setTargetTime(double t);                              // :1741-1753
	double getTargetTime() const             { return dTargetTime; }
	void   forceTimestep(double dt);                             // :1803-1811
	void   forceTimeAdvance()                { bUseForcedTimeAdvance = true; }
	void   runSimulation(double dTargetTime, double dRealTime);  // :1374-1453 + Threaded_runBatch (:1147-1372)
	void   readKeyStatistics();                                  // :1817-1835
	void   readDomainAll();                                      // :1671-1679
	void   saveCurrentState();                                   // :1720-1736
	void   rollbackSimulation(double dCurrentTime, double dTargetTime);   // :1474-1518
	bool   isSimulationFailure(double dExpectedTargetTime);      // :1523-1555
	bool   isSimulationSyncReady(double dExpectedTargetTime);    // :1568-1612
	double proposeSyncPoint(double dCurrentTime);                // :1758-1790
	void   cleanupSimulation();                                  // :1458-1469

	double             getCurrentTime() const          { return dCurrentTime; }
	double             getCurrentTimestep() const      { return dCurrentTimestep; }
	bool               getCurrentSuspendedState() const{ return dCurrentTimestep < 0.0; }
	double             getAverageTimestep() const      { return uiBatchSuccessful < 1 ? 0.0 : dBatchTimesteps / uiBatchSuccessful; }
	unsigned long long getCellsCalculated() const      { return ulCurrentCellsCalculated; }
	unsigned int       getBatchSize() const            { return uiQueueAdditionSize; }
	unsigned int       getIterationsSuccessful() const { return uiBatchSuccessful; }
	unsigned int       getIterationsSkipped() const    { return uiBatchSkipped; }
	const std::string& lastError() const               { return sLastError; }
	hp_domain_t*       handle()                        { return hpDomain; }

private:
	bool check(int rc, const char* what);
	void runBatchThread();                                       // :1116-1139
	void Threaded_runBatch();                                    // :1147-1372
	void runBatch();                                             // one pass of Threaded_runBatch's loop body

	DomainArrays* pDomain;
	hp_domain_t*  hpDomain = nullptr;
	unsigned char ucScheme;
	int           iDevice;
	int           iMathMode = HP_MATH_FAST;
	std::string   sLastError;

	// CScheme.cpp:46-55 / CSchemeGodunov.cpp:42-75 defaults
	bool         bReady = false;
	std::atomic<bool> bRunning{false};                           // set by runSimulation (caller), cleared by the worker
	std::thread  thWorker;                                       // the batch thread; bThreadRunning keeps it alive between batches
	bool         bThreadRunning = false;
	std::mutex   mtxWorker;
	std::condition_variable cvWorker;                            // (the reference's worker spins on bRunning; this one sleeps)
	bool         bAutomaticQueue = true;
	unsigned int uiQueueAdditionSize = 1;
	double       dCourantNumber = 0.5;
	double       dTimestep = 0.001;
	bool         bDynamicTimestep = true;
	bool         bFrictionEffects = true;
	double       dThresholdVerySmall = 1e-10;
	double       dSimulationLength = 1e30;
	unsigned char ucSyncMethod = syncMethod::kSyncForecast;      // CDomainManager.cpp:64-100 default
	unsigned int uiRollbackLimit = 999999999;                    // single domain: CDomainBase.cpp:163-174

	double       dTargetTime = 0.0, dCurrentTime = 0.0, dCurrentTimestep = 0.001;
	double       dBatchTimesteps = 0.0, dBatchStartedTime = 0.0, dLastSyncTime = 0.0;
	unsigned int uiBatchSuccessful = 0, uiBatchSkipped = 0, uiBatchRate = 1;
	unsigned int uiIterationsSinceSync = 0;
	unsigned long long ulCurrentCellsCalculated = 0;
	bool         bUpdateTargetTime = false, bOverrideTimestep = false, bUseForcedTimeAdvance = true;
	bool         bCellStatesSynced = true;

	bool         bStrip = false;
	int          iStripRank = 0, iStripWorld = 1;
	long         lGlobalRows = 0, lRowOffset = 0;
	char         cCommId[HP_COMM_ID_BYTES] = {};

	struct PendingBoundary { int kind, definition; std::vector<double> data; uint64_t entries, rows, cols;
	                         double interval, length, resolution, offx, offy;
	                         int dischargeDefinition = 0; std::vector<uint64_t> cells; };   // kind 0 uniform, 1 gridded, 2 cell
	std::vector<PendingBoundary> boundaries;
};

} // namespace hipims_mi
