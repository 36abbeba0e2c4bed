#include <hip/hip_runtime.h>
__global__ void k(int* out)
{
	int v = threadIdx.x * 10;
	int e = __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false);
	int w = __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
	out[threadIdx.x] = e; out[64 + threadIdx.x] = w;
}
int main(){ int* d; hipMalloc(&d, 512); k<<<1,64>>>(d); int h[128]; hipMemcpy(h,d,512,hipMemcpyDeviceToHost); for(int i=0;i<4;i++) printf("lane %d east %d west %d\n", i, h[i], h[64+i]); for(int i=60;i<64;i++) printf("lane %d east %d west %d\n", i, h[i], h[64+i]); for (int i=14;i<18;i++) printf("lane %d east %d west %d\n", i, h[i], h[64+i]); return 0; }
