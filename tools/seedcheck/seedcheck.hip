// Accuracy of the fp64 hardware seeds used by the FAST flavour (v_rcp_f64, v_rsq_f64) and of the Newton steps on top.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* out, int n)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double v = x[i];
	double r = __builtin_amdgcn_rcp(v);
	out[i] = r;
	double e = __builtin_fma(-v, r, 1.0);
	double r1 = __builtin_fma(r, e, r);
	out[n + i] = r1;
	e = __builtin_fma(-v, r1, 1.0);
	out[2 * n + i] = __builtin_fma(r1, e, r1);
	out[3 * n + i] = __builtin_amdgcn_rsq(v);
}
int main()
{
	const int n = 1 << 20;
	std::vector<double> h(n), o(4 * n);
	std::mt19937_64 g(1); std::uniform_real_distribution<double> u(-8, 8);
	for (auto& v : h) v = std::pow(10.0, u(g));
	double *dx, *dout;
	hipMalloc(&dx, n * 8); hipMalloc(&dout, 4 * n * 8);
	hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
	hipMemcpy(o.data(), dout, 4 * n * 8, hipMemcpyDeviceToHost);
	double e0 = 0, e1 = 0, e2 = 0, es = 0;
	for (int i = 0; i < n; ++i) {
		const long double ex = 1.0L / h[i];
		e0 = std::fmax(e0, (double)fabsl((o[i] - ex) / ex));
		e1 = std::fmax(e1, (double)fabsl((o[n + i] - ex) / ex));
		e2 = std::fmax(e2, (double)fabsl((o[2 * n + i] - ex) / ex));
		const long double sx = 1.0L / sqrtl((long double)h[i]);
		es = std::fmax(es, (double)fabsl((o[3 * n + i] - sx) / sx));
	}
	printf("max rel error: v_rcp_f64 %.3e (2^%.1f)  +1 Newton %.3e  +2 Newton %.3e   v_rsq_f64 %.3e (2^%.1f)\n",
	       e0, std::log2(e0), e1, e2, es, std::log2(es));
	return 0;
}
