// which lane a DPP wavefront shift / rotate reads from, on the hardware (the kernels' from_east / from_west rely on it)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out)
{
	int v = threadIdx.x * 10;
	out[threadIdx.x]       = __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false);   // wave_shl:1, old = own
	out[64 + threadIdx.x]  = __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);   // wave_shr:1
	out[128 + threadIdx.x] = __builtin_amdgcn_mov_dpp(v, 0x134, 0xf, 0xf, false);         // wave_rol:1
	out[192 + threadIdx.x] = __builtin_amdgcn_mov_dpp(v, 0x13C, 0xf, 0xf, false);         // wave_ror:1
}
int main()
{
	int* d; (void)hipMalloc(&d, 1024);
	k<<<1, 64>>>(d);
	int h[256]; (void)hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int i = 0; i < 64; ++i) {
		const bool show = i < 3 || i > 60 || (i >= 15 && i <= 16) || (i >= 31 && i <= 32);
		if (show) std::printf("lane %2d  shl %3d  shr %3d  rol %3d  ror %3d\n", i, h[i] / 10, h[64 + i] / 10, h[128 + i] / 10, h[192 + i] / 10);
		if (h[128 + i] != ((i + 1) % 64) * 10 || h[192 + i] != ((i + 63) % 64) * 10) ++bad;
	}
	std::printf("rol:1 reads lane (i+1)%%64 and ror:1 reads lane (i-1)%%64 on every lane: %s\n", bad ? "NO" : "yes");
	return bad ? 1 : 0;
}
