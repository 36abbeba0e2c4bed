// divcheck -- the STRICT kernels' shared-reciprocal division (hp_math.hpp: recip_of / div_shared) against the plain `a / b`:
// (1) a table of special operands with the flags each raises; (2) N random pairs per exponent window: every UNFLAGGED quotient
// must equal the plain one bit for bit (and the flagged share is reported).   usage: divcheck [millions of pairs, default 200]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include "../../hipims-ocl_amd/csrc/hp_math.hpp"

__global__ void table(const double* a, const double* b, double* q, double* qp, int* flags, int n)
{
	const int i = threadIdx.x;
	if (i >= n) return;
	bool bad = false;
	const hp::Recip<double> d = hp::recip_of<false>(b[i]);
	q[i] = hp::div_shared<false>(a[i], d, bad);
	qp[i] = a[i] / b[i];
	flags[i] = (bad ? 1 : 0) | (d.bad ? 2 : 0);
}

__device__ inline uint64_t rng(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__device__ inline double make(uint64_t& s, int emin, int emax)       // random sign, mantissa, exponent in [emin, emax]
{
	const uint64_t m = rng(s) & 0x000fffffffffffffull, sign = rng(s) & 0x8000000000000000ull;
	const uint64_t e = (uint64_t)(1023 + emin + (int)(rng(s) % (uint64_t)(emax - emin + 1)));
	return __longlong_as_double((long long)(sign | (e << 52) | m));
}
__global__ void sweep(uint64_t seed, int emin_a, int emax_a, int emin_b, int emax_b, int per_thread, unsigned long long* counts)
{
	uint64_t s = seed ^ ((uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1);
	unsigned long long mismatch = 0, flagged = 0, missed = 0;
	for (int k = 0; k < per_thread; ++k) {
		const double b = make(s, emin_b, emax_b);
		const hp::Recip<double> d = hp::recip_of<false>(b);
		for (int j = 0; j < 4; ++j) {                   // four numerators per denominator, as the kernels share it
			const double a = make(s, emin_a, emax_a);
			bool bad = false;
			const double q = hp::div_shared<false>(a, d, bad);
			const double p = a / b;
			bool v;
			const double d0 = __builtin_amdgcn_div_scale(a, b, false, &v);     // what the compiler's sequence takes the reciprocal of
			if (bad) ++flagged;
			else {
				if (__double_as_longlong(q) != __double_as_longlong(p)) ++mismatch;
				if (__builtin_islessgreater(d0, b)) ++missed;                       // the denominator WAS rescaled and the lane did not flag
			}
		}
	}
	atomicAdd(&counts[0], mismatch);
	atomicAdd(&counts[1], flagged);
	atomicAdd(&counts[2], missed);
}

int main(int argc, char** argv)
{
	const double inf = __builtin_inf(), nan = __builtin_nan("");
	const double A[] = {0.0, -0.0, 1.0, 3.0, 1e-310, -4e-320, 1e-292, 1e-291, inf, -inf, nan, 1e300, 1e-300, 1.0, 1.0, 1.0, 1.0, 1.0, 5e-324, 1e200, 0.0};
	const double B[] = {3.0, 7.0, 3.0, -7.0, 3.0, 3.0, 3.0, 3.0, 3.0, 3.0, 3.0, 1e-100, 1e100, 0.0, inf, nan, 1e-310, 1e305, 1e-5, 1e-200, 0.0};
	const int n = (int)(sizeof A / sizeof A[0]);
	double *da, *db, *dq, *dp; int* df;
	hipMalloc(&da, sizeof A); hipMalloc(&db, sizeof B); hipMalloc(&dq, sizeof A); hipMalloc(&dp, sizeof A); hipMalloc(&df, n * sizeof(int));
	hipMemcpy(da, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(db, B, sizeof B, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(table, dim3(1), dim3(64), 0, 0, da, db, dq, dp, df, n);
	double Q[64], P[64]; int F[64];
	hipMemcpy(Q, dq, sizeof A, hipMemcpyDeviceToHost); hipMemcpy(P, dp, sizeof A, hipMemcpyDeviceToHost); hipMemcpy(F, df, n * sizeof(int), hipMemcpyDeviceToHost);
	std::printf("%12s %12s   %-24s %-24s flagged(1) b-flag(2)  same bits\n", "a", "b", "shared", "plain");
	int wrong_unflagged = 0;
	for (int i = 0; i < n; ++i) {
		const bool same = std::memcmp(&Q[i], &P[i], 8) == 0 || (Q[i] != Q[i] && P[i] != P[i]);
		std::printf("%12.4g %12.4g   %-24.17g %-24.17g %d          %s\n", A[i], B[i], Q[i], P[i], F[i], same ? "yes" : "NO");
		if (!F[i] && !same) ++wrong_unflagged;
	}
	std::printf("table: %d unflagged quotients differ from the plain division\n\n", wrong_unflagged);

	const long millions = argc > 1 ? std::atol(argv[1]) : 200;
	unsigned long long* dc; hipMalloc(&dc, 24);
	struct W { int ea0, ea1, eb0, eb1; const char* what; } win[] = {
		{-40, 40, -40, 40, "the kernels' working range (1e-12 .. 1e12)"},
		{-300, 300, -300, 300, "wide"},
		{-1022, 1023, -1022, 1023, "the whole normal range"},
		{-1022, -960, -60, 60, "tiny numerators (below and above 2^-969)"},
		{-60, 60, -1022, -990, "tiny denominators"},
		{-60, 60, 990, 1023, "huge denominators"},
		{600, 1023, -400, -1, "quotients that overflow or nearly"},
		{-700, -300, 300, 700, "quotients that underflow or nearly"},
	};
	unsigned long long total = 0, total_mismatch = 0;
	for (const W& w : win) {
		hipMemset(dc, 0, 24);
		const int blocks = 4096, threads = 256;
		const int per_thread = (int)((millions * 1000000L / 4) / (blocks * threads) / (long)(sizeof win / sizeof win[0])) + 1;
		hipLaunchKernelGGL(sweep, dim3(blocks), dim3(threads), 0, 0, 0x1234567ull + total, w.ea0, w.ea1, w.eb0, w.eb1, per_thread, dc);
		unsigned long long c[3];
		hipMemcpy(c, dc, 24, hipMemcpyDeviceToHost);
		const unsigned long long pairs = (unsigned long long)blocks * threads * per_thread * 4;
		std::printf("%-48s %12llu pairs  flagged %10llu (%.4f %%)  unflagged-and-different %llu  unflagged-with-rescaled-denominator %llu\n", w.what, pairs, c[1], 100.0 * c[1] / pairs, c[0], c[2]);
		total += pairs; total_mismatch += c[0] + c[2];
	}
	std::printf("total %llu pairs, %llu unflagged quotients differ from the plain IEEE division\n", total, total_mismatch);
	return (wrong_unflagged || total_mismatch) ? 1 : 0;
}
