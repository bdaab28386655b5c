/* Host-side cost of the pinned H2D path per frame (event wait, memcpy into the pinned slot,
 * hipMemcpyAsync, event record, kernel launch) for 0.25 - 32 MiB frames: the measurement behind
 * the 8 MiB copy-stream threshold in csrc/executor.cpp.
 *   hipcc --offload-arch=gfx950 -O2 -o h2d_probe tools/h2d_probe.cpp && ./h2d_probe */
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(float *p, int n) { float v = p[threadIdx.x]; for (int i = 0; i < n; i++) v = v * 1.0001f + 0.5f; p[threadIdx.x] = v; }
int main()
{
	hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	for (size_t size : {262144ul, 1ul << 20, 4ul << 20, 32ul << 20}) {
		void *pinned[3], *dev; std::vector<char> user(size, 1);
		for (auto &p : pinned) hipHostMalloc(&p, size, hipHostMallocDefault);
		hipMalloc(&dev, size);
		hipEvent_t e[3]; for (auto &x : e) hipEventCreateWithFlags(&x, hipEventDisableTiming);
		float *w; hipMalloc(&w, 1024);
		hipStreamSynchronize(s);
		double t_memcpy = 0, t_async = 0, t_rec = 0, t_sync = 0, t_kernel = 0;
		const int N = 200;
		double t0 = now();
		for (int i = 0; i < N; i++) {
			int k = i % 3;
			double a = now(); if (i >= 3) hipEventSynchronize(e[k]); double b = now(); t_sync += b - a;
			std::memcpy(pinned[k], user.data(), size); a = now(); t_memcpy += a - b;
			hipMemcpyAsync(dev, pinned[k], size, hipMemcpyHostToDevice, s); b = now(); t_async += b - a;
			hipEventRecord(e[k], s); a = now(); t_rec += a - b;
			hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, w, 3000); b = now(); t_kernel += b - a;
		}
		hipStreamSynchronize(s);
		double el = now() - t0;
		printf("size %9zu: per-iter %8.1f us | evsync %7.1f memcpy %7.1f async %7.1f record %6.1f launch %6.1f\n",
		       size, el / N * 1e6, t_sync / N * 1e6, t_memcpy / N * 1e6, t_async / N * 1e6, t_rec / N * 1e6, t_kernel / N * 1e6);
		for (auto &p : pinned) hipHostFree(p);
		hipFree(dev); hipFree(w);
	}
	return 0;
}
