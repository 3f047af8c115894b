// Microbenchmark: how fast does the GPU start tiny workgroups?  1 M wavefronts as 64-thread and as 256-thread workgroups, each wave doing
// a short dependent chain of global loads (like the per-read finalisation kernels).   hipcc -O3 --offload-arch=gfx950 scripts/launch_rate.hip -o scripts/launch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAIN>
__global__ void k(const int *p, int *out, int n)
{
	const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	if (w >= n) return;
	int v = w;
	for (int i = 0; i < CHAIN; ++i) v = p[(v * 97 + i) & 0xfffff];
	if ((threadIdx.x & 63) == 0) out[w] = v;
}
template <int CHAIN> static void run(const int *p, int *out, int n, int bs)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int blocks = (int)(((long long)n * 64 + bs - 1) / bs);
	hipLaunchKernelGGL(k<CHAIN>, dim3(blocks), dim3(bs), 0, 0, p, out, n);
	hipEventRecord(e0, 0);
	for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<CHAIN>, dim3(blocks), dim3(bs), 0, 0, p, out, n);
	hipEventRecord(e1, 0); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	printf("chain %2d  block %4d : %.3f ms per 1M waves  (%.0f M waves/s)\n", CHAIN, bs, ms / 5, n / (ms / 5) / 1e3);
}
int main()
{
	int *p, *out; const int n = 1 << 20;
	hipMalloc(&p, 4 << 20); hipMalloc(&out, n * 4); hipMemset(p, 1, 4 << 20);
	for (int bs : { 64, 128, 256, 512, 1024 }) run<0>(p, out, n, bs);
	for (int bs : { 64, 256, 1024 }) run<8>(p, out, n, bs);
	for (int bs : { 64, 256, 1024 }) run<24>(p, out, n, bs);
	return 0;
}
