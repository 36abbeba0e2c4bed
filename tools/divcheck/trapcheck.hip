// trapcheck -- can a wave's sticky IEEE exception bits (TRAPSTS.EXCP) stand guard over a division WITHOUT v_div_scale / v_div_fmas?
// (1) which bits each kind of operation raises on gfx950 with traps disabled; (2) N random pairs per exponent window: the bare
// sequence (v_rcp_f64, two Newton steps, q0 = a r, rem = a - b q0, q = q0 + rem r, v_div_fixup) against the compiler's `a / b`:
// every wave with a differing quotient must have raised invalid / input-denormal / div0 / overflow / underflow.
// usage: trapcheck [millions of pairs, default 200]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>

__device__ inline unsigned excp_read(double dep)
{
	unsigned v;
	asm volatile("s_nop 7\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 7)" : "=s"(v) : "v"(dep) : "memory");
	return v;
}
__device__ inline void excp_clear(double& a, double& b)
{
	asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 7), 0\n\ts_nop 3" : "+v"(a), "+v"(b) :: "memory");
}
__device__ inline double rcp_refined(double b)
{
	double r = __builtin_amdgcn_rcp(b);
	double e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r);
	e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r);
	return r;
}
__device__ inline double div_bare(double a, double b, double r)
{
	const double q0 = a * r;
	const double rem = __builtin_fma(-b, q0, a);
	return __builtin_amdgcn_div_fixup(__builtin_fma(rem, r, q0), b, a);
}

// op: 0 a*b  1 a+b  2 fma(a,b,b)  3 rcp(a)  4 sqrt-ish rsq(a)  5 bare a/b  6 compiler a/b  7 nothing
__global__ void table(const double* A, const double* B, const int* OP, double* out, unsigned* flags, unsigned* initial, int n)
{
	if (threadIdx.x == 0) initial[0] = excp_read(0.0);
	for (int i = 0; i < n; ++i) {
		double a = A[i], b = B[i], q = 0.0;
		excp_clear(a, b);
		if (threadIdx.x == 0) {                          // one lane raises; the bits belong to the wave
			switch (OP[i]) {
			case 0: q = a * b; break;
			case 1: q = a + b; break;
			case 2: q = __builtin_fma(a, b, b); break;
			case 3: q = __builtin_amdgcn_rcp(a); break;
			case 4: q = __builtin_amdgcn_rsq(a); break;
			case 5: q = div_bare(a, b, rcp_refined(b)); break;
			case 6: q = a / b; break;
			default: q = a; break;
			}
		}
		const unsigned f = excp_read(q);
		if (threadIdx.x == 0) { out[i] = q; flags[i] = f; }
	}
}

__device__ inline uint64_t rng(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__device__ inline double make(uint64_t& s, int emin, int emax)
{
	const uint64_t m = rng(s) & 0x000fffffffffffffull, sign = rng(s) & 0x8000000000000000ull;
	const uint64_t e = (uint64_t)(1023 + emin + (int)(rng(s) % (uint64_t)(emax - emin + 1)));
	return __longlong_as_double((long long)(sign | (e << 52) | m));
}
constexpr unsigned GUARD = 0x1f;                      // invalid, input denormal, float div0, overflow, underflow
__global__ void sweep(uint64_t seed, int emin_a, int emax_a, int emin_b, int emax_b, int per_thread, unsigned long long* counts)
{
	uint64_t s = seed ^ ((uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1);
	unsigned long long wrong_lanes = 0, flagged_waves = 0, missed_waves = 0, waves = 0;
	for (int k = 0; k < per_thread; ++k) {
		double b = make(s, emin_b, emax_b);
		double a0 = make(s, emin_a, emax_a), a1 = make(s, emin_a, emax_a);
		double a2 = make(s, emin_a, emax_a), a3 = make(s, emin_a, emax_a);
		const double p0 = a0 / b, p1 = a1 / b, p2 = a2 / b, p3 = a3 / b;             // the compiler's IEEE sequence
		double dep = p0 + p1 + p2 + p3;
		excp_clear(b, dep);
		const double r = rcp_refined(b);
		const double q0 = div_bare(a0, b, r), q1 = div_bare(a1, b, r), q2 = div_bare(a2, b, r), q3 = div_bare(a3, b, r);
		const unsigned f = excp_read((q0 + q1) + (q2 + q3)) & GUARD;
		const int wrong = (__double_as_longlong(q0) != __double_as_longlong(p0)) + (__double_as_longlong(q1) != __double_as_longlong(p1))
			+ (__double_as_longlong(q2) != __double_as_longlong(p2)) + (__double_as_longlong(q3) != __double_as_longlong(p3));
		const bool any_wrong = __builtin_amdgcn_ballot_w64(wrong != 0) != 0;
		wrong_lanes += wrong;
		if ((threadIdx.x & 63) == 0) { ++waves; if (f) ++flagged_waves; else if (any_wrong) ++missed_waves; }
	}
	atomicAdd(&counts[0], wrong_lanes);
	atomicAdd(&counts[1], flagged_waves);
	atomicAdd(&counts[2], missed_waves);
	atomicAdd(&counts[3], waves);
}

int main(int argc, char** argv)
{
	const double inf = __builtin_inf();
	struct Row { double a, b; int op; const char* what; } rows[] = {
		{1.5, 2.5, 0, "1.5 * 2.5 (exact)"}, {1.0, 3.0, 6, "1 / 3 compiler (inexact)"}, {1.0, 3.0, 5, "1 / 3 bare"},
		{1e-310, 2.0, 0, "denormal * 2"}, {1e-200, 1e-200, 0, "1e-200 * 1e-200 (underflow)"}, {1e-160, 1e-160, 0, "1e-160^2 (denormal result)"},
		{0x1p-1000, 0x1p-50, 0, "2^-1000 * 2^-50 (exact denormal result)"},
		{1e200, 1e200, 0, "overflow"}, {0.0, inf, 0, "0 * inf"}, {inf, -inf, 1, "inf - inf"},
		{0.0, 0.0, 3, "rcp(0)"}, {1e-310, 0.0, 3, "rcp(denormal)"}, {0x1p1023, 0.0, 3, "rcp(2^1023)"}, {3.0, 0.0, 3, "rcp(3)"},
		{0.0, 0.0, 4, "rsq(0)"}, {-1.0, 0.0, 4, "rsq(-1)"},
		{0.0, 3.0, 5, "0 / 3 bare"}, {-0.0, 3.0, 5, "-0 / 3 bare"}, {1.0, 0.0, 5, "1 / 0 bare"}, {0.0, 0.0, 5, "0 / 0 bare"},
		{1e-300, 3.0, 5, "1e-300 / 3 bare (tiny numerator)"}, {1.0, 0x1.8p1022, 5, "1 / 1.5*2^1022 bare"}, {1.0, 1e-310, 5, "1 / denormal bare"},
		{1e300, 1e-10, 5, "overflowing quotient bare"}, {1e-300, 1e10, 5, "underflowing quotient bare"},
		{1.0, 3.0, 7, "nothing"},
	};
	const int n = (int)(sizeof rows / sizeof rows[0]);
	double A[64], B[64]; int OP[64];
	for (int i = 0; i < n; ++i) { A[i] = rows[i].a; B[i] = rows[i].b; OP[i] = rows[i].op; }
	double *da, *db, *dq; int* dop; unsigned *df, *di;
	hipMalloc(&da, sizeof A); hipMalloc(&db, sizeof B); hipMalloc(&dq, sizeof A); hipMalloc(&dop, sizeof OP); hipMalloc(&df, 64 * 4); hipMalloc(&di, 4);
	hipMemcpy(da, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(db, B, sizeof B, hipMemcpyHostToDevice); hipMemcpy(dop, OP, sizeof OP, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(table, dim3(1), dim3(64), 0, 0, da, db, dop, dq, df, di, n);
	double Q[64]; unsigned F[64], I0;
	hipMemcpy(Q, dq, sizeof Q, hipMemcpyDeviceToHost); hipMemcpy(F, df, sizeof F, hipMemcpyDeviceToHost); hipMemcpy(&I0, di, 4, hipMemcpyDeviceToHost);
	if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernel failed\n"); return 2; }
	std::printf("TRAPSTS.EXCP of a fresh wave: 0x%x\n", I0);
	std::printf("%-44s %-24s bits (1 invalid, 2 input denormal, 4 div0, 8 overflow, 16 underflow, 32 inexact, 64 int div0)\n", "operation", "result");
	for (int i = 0; i < n; ++i) std::printf("%-44s %-24.17g 0x%02x\n", rows[i].what, Q[i], F[i]);

	const long millions = argc > 1 ? std::atol(argv[1]) : 200;
	unsigned long long* dc; hipMalloc(&dc, 32);
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
	unsigned long long total = 0, total_missed = 0;
	for (const W& w : win) {
		hipMemset(dc, 0, 32);
		const int blocks = 4096, threads = 256;
		const int per_thread = (int)((millions * 1000000L / 4) / (blocks * threads) / (long)(sizeof win / sizeof win[0])) + 1;
		hipLaunchKernelGGL(sweep, dim3(blocks), dim3(threads), 0, 0, 0x1234567ull + total, w.ea0, w.ea1, w.eb0, w.eb1, per_thread, dc);
		unsigned long long c[4];
		hipMemcpy(c, dc, 32, hipMemcpyDeviceToHost);
		const unsigned long long pairs = (unsigned long long)blocks * threads * per_thread * 4;
		std::printf("%-48s %12llu pairs  differing quotients %10llu  waves %llu flagged %llu (%.4f %%)  UNFLAGGED waves with a differing quotient %llu\n",
		            w.what, pairs, c[0], c[3], c[1], 100.0 * c[1] / (c[3] ? c[3] : 1), c[2]);
		total += pairs; total_missed += c[2];
	}
	std::printf("total %llu pairs, %llu unflagged waves held a quotient that differs from the plain IEEE division\n", total, total_missed);
	return total_missed ? 1 : 0;
}
