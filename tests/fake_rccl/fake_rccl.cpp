// fake_rccl -- TEST DOUBLE for the collective library behind hp_strip_* (tests only; never shipped or linked by the
// product).  RCCL refuses two ranks on one device, and the GPU boxes the tests run on have one: this library gives
// the nine entry points hp_comm_load resolves the semantics they have in RCCL for RANKS THAT ARE THREADS OF ONE
// PROCESS sharing one GPU, so that hp_strip_step_batch's multi-rank logic (which rows go where, stream ordering,
// which iterations all-reduce) runs with 2-4 real ranks (tests/strip_threads_worker.py).
//   ncclSend / ncclRecv inside a group: the receiver copies device-to-device on ITS stream after the sender's stream
//   has reached the send (event), and the sender's stream then waits for the copy (event) -- the ordering RCCL gives.
//   ncclAllReduce(MAX, 1..8 elements, in or out of place): through the host, with a generation barrier over the ranks.
// RANKS THAT ARE PROCESSES (tests/strip_procs_worker.py: the ghost rows and maxima then travel through IPC mappings, which
// is what that test is about): with FAKE_RCCL_SHM=<file> in the environment the all-reduce's meeting point is that file,
// mapped by every process (created zero-filled by the test); send / receive are not available in this mode.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

namespace {
struct Parcel { const void* buf = nullptr; size_t bytes = 0; hipEvent_t ready = nullptr, done = nullptr; bool posted = false, copied = false; };
struct Shared {
	std::mutex m;
	std::condition_variable cv;
	std::map<std::pair<int, int>, std::vector<Parcel>> box;      // (src, dst) -> parcels in flight, FIFO
	int world = 0, arrived = 0;
	unsigned long long generation = 0;
	double acc[8] = {0}, result[8] = {0};
};
struct ShmBlock { unsigned long long lock, generation; int arrived, pad; double acc[8], result[8]; };   // zero-filled file
struct ShmLock {
	ShmBlock* b;
	explicit ShmLock(ShmBlock* blk) : b(blk) { while (__atomic_exchange_n(&b->lock, 1ull, __ATOMIC_ACQUIRE)) sched_yield(); }
	~ShmLock() { __atomic_store_n(&b->lock, 0ull, __ATOMIC_RELEASE); }
};
struct Op { bool send; void* buf; size_t bytes; int peer; hipStream_t stream; };
thread_local std::vector<Op> t_ops;
thread_local int t_depth = 0;
size_t type_size(ncclDataType_t t) { return t == ncclDouble ? 8 : 4; }
}

struct ncclComm { int rank, world; Shared* sh; ShmBlock* shm; };

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
	Shared* sh = new Shared;                                      // lives for the test process
	std::memset(id, 0, sizeof *id);
	std::memcpy(id->internal, &sh, sizeof sh);
	return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank)
{
	Shared* sh;
	std::memcpy(&sh, id.internal, sizeof sh);
	if (const char* path = std::getenv("FAKE_RCCL_SHM")) {        // ranks are processes: `sh` belongs to whoever made the id
		const int fd = open(path, O_RDWR);
		if (fd < 0) return ncclSystemError;
		void* m = mmap(nullptr, sizeof(ShmBlock), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		close(fd);
		if (m == MAP_FAILED) return ncclSystemError;
		*comm = new ncclComm{rank, world, nullptr, (ShmBlock*)m};
		return ncclSuccess;
	}
	{ std::lock_guard<std::mutex> l(sh->m); sh->world = world; }
	*comm = new ncclComm{rank, world, sh, nullptr};
	return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t) { return "fake_rccl error"; }
ncclResult_t ncclGroupStart() { ++t_depth; return ncclSuccess; }

static ncclResult_t flush(ncclComm_t comm)
{
	Shared* sh = comm->sh;
	if (!sh) return ncclInvalidUsage;                             // (process ranks: all-reduce only)
	std::vector<Op> ops; ops.swap(t_ops);
	// 1. post every send: the parcel becomes visible together with an event that marks the sender's stream position
	for (const Op& o : ops) if (o.send) {
		Parcel p; p.buf = o.buf; p.bytes = o.bytes; p.posted = true;
		if (hipEventCreateWithFlags(&p.ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
		if (hipEventCreateWithFlags(&p.done, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
		if (hipEventRecord(p.ready, o.stream) != hipSuccess) return ncclUnhandledCudaError;
		std::lock_guard<std::mutex> l(sh->m);
		sh->box[{comm->rank, o.peer}].push_back(p);
		sh->cv.notify_all();
	}
	// 2. every receive: wait for the matching parcel, copy on the receiver's stream behind the sender's event
	for (const Op& o : ops) if (!o.send) {
		std::unique_lock<std::mutex> l(sh->m);
		auto& q = sh->box[{o.peer, comm->rank}];
		size_t idx = 0;
		sh->cv.wait(l, [&] { for (idx = 0; idx < q.size(); ++idx) if (q[idx].posted && !q[idx].copied) return true; return false; });
		Parcel& p = q[idx];
		if (p.bytes != o.bytes) return ncclInvalidArgument;
		if (hipStreamWaitEvent(o.stream, p.ready, 0) != hipSuccess) return ncclUnhandledCudaError;
		if (hipMemcpyAsync(o.buf, p.buf, p.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) return ncclUnhandledCudaError;
		if (hipEventRecord(p.done, o.stream) != hipSuccess) return ncclUnhandledCudaError;
		p.copied = true;
		sh->cv.notify_all();
	}
	// 3. every send: the sender's stream must not run ahead of the copy out of its buffer
	for (const Op& o : ops) if (o.send) {
		std::unique_lock<std::mutex> l(sh->m);
		auto& q = sh->box[{comm->rank, o.peer}];
		sh->cv.wait(l, [&] { return !q.empty() && q.front().copied; });
		Parcel p = q.front();
		q.erase(q.begin());
		l.unlock();
		if (hipStreamWaitEvent(o.stream, p.done, 0) != hipSuccess) return ncclUnhandledCudaError;
		hipEventDestroy(p.ready);                                 // destruction is deferred by the runtime until completion
		hipEventDestroy(p.done);
	}
	return ncclSuccess;
}

static thread_local ncclComm_t t_comm = nullptr;
ncclResult_t ncclGroupEnd()
{
	if (--t_depth > 0) return ncclSuccess;
	return t_comm ? flush(t_comm) : ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
	t_comm = comm;
	t_ops.push_back({true, const_cast<void*>(buf), count * type_size(type), peer, stream});
	return t_depth > 0 ? ncclSuccess : flush(comm);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
	t_comm = comm;
	t_ops.push_back({false, buf, count * type_size(type), peer, stream});
	return t_depth > 0 ? ncclSuccess : flush(comm);
}

ncclResult_t ncclAllReduce(const void* sendbuf, void* recvbuf, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
	if (count < 1 || count > 8 || op != ncclMax) return ncclInvalidArgument;
	double v[8] = {0}; float vf[8] = {0};
	if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
	if (type == ncclDouble) { if (hipMemcpy(v, sendbuf, 8 * count, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError; }
	else { if (hipMemcpy(vf, sendbuf, 4 * count, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError; for (size_t i = 0; i < count; ++i) v[i] = vf[i]; }
	Shared* sh = comm->sh;
	double result[8];
	if (ShmBlock* b = comm->shm) {
		unsigned long long gen;
		{
			ShmLock l(b);
			gen = b->generation;
			for (size_t i = 0; i < count; ++i) if (b->arrived == 0 || v[i] > b->acc[i]) b->acc[i] = v[i];
			if (++b->arrived == comm->world) { for (size_t i = 0; i < count; ++i) b->result[i] = b->acc[i]; b->arrived = 0; __atomic_store_n(&b->generation, gen + 1, __ATOMIC_RELEASE); }
		}
		const auto t0 = std::chrono::steady_clock::now();
		while (__atomic_load_n(&b->generation, __ATOMIC_ACQUIRE) == gen) {
			if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return ncclSystemError;   // a rank went missing
			sched_yield();
		}
		ShmLock l(b);
		for (size_t i = 0; i < count; ++i) result[i] = b->result[i];
	} else {
		std::unique_lock<std::mutex> l(sh->m);
		const unsigned long long gen = sh->generation;
		for (size_t i = 0; i < count; ++i) if (sh->arrived == 0 || v[i] > sh->acc[i]) sh->acc[i] = v[i];
		if (++sh->arrived == comm->world) { for (size_t i = 0; i < count; ++i) sh->result[i] = sh->acc[i]; sh->arrived = 0; ++sh->generation; sh->cv.notify_all(); }
		else sh->cv.wait(l, [&] { return sh->generation != gen; });
		for (size_t i = 0; i < count; ++i) result[i] = sh->result[i];
	}
	if (type == ncclDouble) { if (hipMemcpy(recvbuf, result, 8 * count, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError; }
	else { for (size_t i = 0; i < count; ++i) vf[i] = (float)result[i]; if (hipMemcpy(recvbuf, vf, 4 * count, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError; }
	return ncclSuccess;
}

}
