// pkrate -- issue rate and dependent-issue latency of v_fma_f32, v_pk_fma_f32, v_fma_f64 and v_cndmask on one SIMD
// (one 64-lane wave per SIMD, then 2 and 4): cycles per wave-instruction from s_memtime.  Decides whether packed
// fp32 can pay on this chip.    build: hipcc --offload-arch=gfx950 -O2 -o pkrate pkrate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND, int CHAINS>
__global__ void k(float* out, unsigned long long* cyc, int iters)
{
	f2 a[4]; double d[4]; float s[4];
	for (int i = 0; i < 4; ++i) { a[i] = (f2)(1.0f + threadIdx.x * 1e-3f + i); d[i] = 1.0 + i + threadIdx.x * 1e-3; s[i] = 1.0f + i + threadIdx.x * 1e-3f; }
	const f2 m = (f2)(0.999f), c = (f2)(1e-3f);
	const unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < iters; ++it) {
		if (KIND == 0) { REP16(for (int j = 0; j < CHAINS; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[j]) : "v"(m.x), "v"(c.x));) }
		if (KIND == 1) { REP16(for (int j = 0; j < CHAINS; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(m), "v"(c));) }
		if (KIND == 2) { REP16(for (int j = 0; j < CHAINS; ++j) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[j]) : "v"((double)m.x), "v"((double)c.x));) }
		if (KIND == 3) { REP16(for (int j = 0; j < CHAINS; ++j) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(m));) }
		if (KIND == 4) { REP16(for (int j = 0; j < CHAINS; ++j) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c));) }
	}
	const unsigned long long t1 = __builtin_readcyclecounter();
	float r = 0; for (int i = 0; i < 4; ++i) r += a[i].x + a[i].y + (float)d[i] + s[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = r;
	if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND, int CHAINS> void run(const char* name, int waves_per_simd)
{
	float* out; unsigned long long* cyc; hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
	const int iters = 2000, blocks = 256 * waves_per_simd;       // 256-thread blocks = 4 waves = one per SIMD of a CU
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<KIND, CHAINS><<<blocks, 256>>>(out, cyc, iters); hipDeviceSynchronize();
	hipEventRecord(e0); k<KIND, CHAINS><<<blocks, 256>>>(out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
	const double n = (double)iters * 16 * CHAINS;
	std::printf("%-14s chains %d  waves/SIMD %d : %6.2f counter ticks per instr per wave, %7.3f ns per instr per SIMD (all waves)\n", name, CHAINS,
	            waves_per_simd, (double)h / n, ms * 1e6 / (n * waves_per_simd));
	hipFree(out); hipFree(cyc);
}
int main()
{
	for (int w : {1, 2, 4}) {
		if (w == 1) { run<0,1>("v_fma_f32", 1); run<0,4>("v_fma_f32", 1); run<1,1>("v_pk_fma_f32", 1); run<1,4>("v_pk_fma_f32", 1); run<2,1>("v_fma_f64", 1); run<2,4>("v_fma_f64", 1); run<3,4>("v_pk_mul_f32", 1); run<4,4>("v_pk_add_f32", 1); }
		if (w == 2) { run<0,1>("v_fma_f32", 2); run<1,1>("v_pk_fma_f32", 2); run<2,1>("v_fma_f64", 2); run<0,4>("v_fma_f32", 2); run<1,4>("v_pk_fma_f32", 2); }
		if (w == 4) { run<0,1>("v_fma_f32", 4); run<1,1>("v_pk_fma_f32", 4); run<2,1>("v_fma_f64", 4); run<0,4>("v_fma_f32", 4); run<1,4>("v_pk_fma_f32", 4); }
	}
	return 0;
}
