// which CU each block of a 1024-thread, 75 KB-LDS launch lands on (first wave of blocks): dispatch policy probe.
// hipcc --offload-arch=gfx950 -O2 tools/cu_map.hip -o tools/bin/cu_map; output: profiles/r02_cu_map.txt (first 520 blocks: 65 per XCD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(1024) void probe(uint32_t *out, int spin)
{
	extern __shared__ float lds[];
	uint32_t hw, xcc;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	uint64_t t0 = __builtin_amdgcn_s_memtime();
	lds[threadIdx.x] = (float)hw;
	while (__builtin_amdgcn_s_memtime() - t0 < (uint64_t)spin) { }
	if (threadIdx.x == 0) { out[blockIdx.x * 4 + 0] = hw; out[blockIdx.x * 4 + 1] = xcc; out[blockIdx.x * 4 + 2] = (uint32_t)t0; out[blockIdx.x * 4 + 3] = (uint32_t)(t0 >> 32); }
}
int main()
{
	const int blocks = 8 * 64 * 2;
	uint32_t *d, *h = new uint32_t[blocks * 4];
	hipMalloc(&d, blocks * 16);
	hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 76000);
	hipLaunchKernelGGL(probe, dim3(blocks), dim3(1024), 76000, 0, d, 2000000);
	hipDeviceSynchronize();
	hipMemcpy(h, d, blocks * 16, hipMemcpyDeviceToHost);
	for (int b = 0; b < blocks; b++) {
		uint32_t hw = h[b * 4], xcc = h[b * 4 + 1] & 0xf;
		uint64_t t = ((uint64_t)h[b * 4 + 3] << 32) | h[b * 4 + 2];
		printf("%d xcc %u se %u sh %u cu %u raw %08x t %llu\n", b, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, hw, (unsigned long long)t);
	}
	return 0;
}
