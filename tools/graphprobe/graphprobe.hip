// graphprobe -- round 5: would capturing a batch's launches in a hipGraph shorten a small grid's iteration?  (not product code)
// N dependent launches of a kernel of B blocks that does ~t microseconds of work, on one stream: (a) plain launches, (b) the same
// launches captured once into a graph and replayed.  Reports microseconds per launch, host side (time to enqueue) and end to end.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void work(double* a, int iters)
{
	double v = a[blockIdx.x * blockDim.x + threadIdx.x];
	for (int i = 0; i < iters; ++i) v = v * 1.0000001 + 1e-9;
	a[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main(int argc, char** argv)
{
	const int n = argc > 1 ? atoi(argv[1]) : 512;
	double* a; CK(hipMalloc(&a, 1024 * 256 * sizeof(double))); CK(hipMemset(a, 0, 1024 * 256 * sizeof(double)));
	hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	for (int blocks : {16, 120, 512}) for (int iters : {50, 1000}) {
		auto now = [] { return std::chrono::steady_clock::now(); };
		auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
		for (int w = 0; w < 64; ++w) hipLaunchKernelGGL(work, dim3(blocks), dim3(256), 0, s, a, iters);
		CK(hipStreamSynchronize(s));
		double best_plain = 1e30, best_plain_host = 1e30, best_graph = 1e30, best_graph_host = 1e30;
		for (int rep = 0; rep < 5; ++rep) {
			auto t0 = now();
			for (int i = 0; i < n; ++i) hipLaunchKernelGGL(work, dim3(blocks), dim3(256), 0, s, a, iters);
			auto t1 = now();
			CK(hipStreamSynchronize(s));
			auto t2 = now();
			best_plain = std::min(best_plain, us(t0, t2) / n); best_plain_host = std::min(best_plain_host, us(t0, t1) / n);
		}
		hipGraph_t g; hipGraphExec_t ge;
		CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
		for (int i = 0; i < n; ++i) hipLaunchKernelGGL(work, dim3(blocks), dim3(256), 0, s, a, iters);
		CK(hipStreamEndCapture(s, &g));
		CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
		CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
		for (int rep = 0; rep < 5; ++rep) {
			auto t0 = now();
			CK(hipGraphLaunch(ge, s));
			auto t1 = now();
			CK(hipStreamSynchronize(s));
			auto t2 = now();
			best_graph = std::min(best_graph, us(t0, t2) / n); best_graph_host = std::min(best_graph_host, us(t0, t1) / n);
		}
		printf("%4d blocks x %4d iterations of work, %d dependent launches:  plain %.2f us per launch (host %.2f)   graph replay %.2f us per launch (host %.2f)\n",
		       blocks, iters, n, best_plain, best_plain_host, best_graph, best_graph_host);
		CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
	}
	return 0;
}
