// Microbenchmark: ceiling for the FM-index access pattern on this GPU -- dependent random 64-byte gathers
// (one lane reads one whole 64 B block with 4 x 16 B loads, next address depends on the data), K independent
// gathers per lane per step, optional 32 B scattered store per step (the prev/curr list traffic of k_smem).
//   hipcc -O3 --offload-arch=gfx950 scripts/gather_bw.hip -o gpurun_out/gather_bw && gpurun_out/gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

template <int K, int ST>
__global__ __launch_bounds__(256) void k_gather(const uint4 *tab, uint64_t nblk, int steps, uint4 *scratch, uint64_t per_lane, unsigned long long *sink)
{
	uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
	uint64_t s[K];
	for (int k = 0; k < K; ++k) s[k] = (tid * K + k) * 0x9E3779B97F4A7C15ull + 12345;
	uint64_t acc = 0;
	uint4 *my = scratch + tid * per_lane;
	for (int it = 0; it < steps; ++it) {
		uint4 v[K][4];
		for (int k = 0; k < K; ++k) {
			s[k] ^= s[k] >> 29; s[k] *= 0xBF58476D1CE4E5B9ull; s[k] ^= s[k] >> 32;
			const uint4 *b = tab + (s[k] % nblk) * 4;
			v[k][0] = b[0]; v[k][1] = b[1]; v[k][2] = b[2]; v[k][3] = b[3];
		}
		for (int k = 0; k < K; ++k) {
			uint64_t h = (uint64_t)(v[k][0].x ^ v[k][1].y ^ v[k][2].z ^ v[k][3].w);
			s[k] += h; acc += h;
		}
		if (ST) {
			uint4 *p = my + (size_t)(it % per_lane);
			p[0] = make_uint4((uint32_t)acc, it, 0, 0);
			if (ST > 1) p[1 % per_lane] = make_uint4(it, (uint32_t)acc, 0, 0);
		}
	}
	if (acc == 0x1234567) atomicAdd(sink, acc);
}

// the same gathers, loaded quad-cooperatively: in round r the four lanes of a quad read the four 16-byte quarters of quad-lane r's
// block with ONE instruction (a wavefront instruction then touches 16 cache lines instead of 64)
template <int K>
__global__ __launch_bounds__(256) void k_gather_coop(const uint4 *tab, uint64_t nblk, int steps, unsigned long long *sink)
{
	uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
	const int ql = threadIdx.x & 3;
	uint64_t s[K];
	for (int k = 0; k < K; ++k) s[k] = (tid * K + k) * 0x9E3779B97F4A7C15ull + 12345;
	uint64_t acc = 0;
	for (int it = 0; it < steps; ++it) {
		uint4 v[K][4];
		for (int k = 0; k < K; ++k) {
			s[k] ^= s[k] >> 29; s[k] *= 0xBF58476D1CE4E5B9ull; s[k] ^= s[k] >> 32;
			const uint64_t mine = (s[k] % nblk) * 4;
			for (int r = 0; r < 4; ++r) {
				const uint64_t blk = __shfl(mine, (threadIdx.x & ~3) + r);   // quad-lane r's block index
				v[k][r] = tab[blk + ql];
			}
		}
		for (int k = 0; k < K; ++k) {
			uint64_t h = (uint64_t)(v[k][0].x ^ v[k][1].y ^ v[k][2].z ^ v[k][3].w);
			s[k] += h; acc += h;
		}
	}
	if (acc == 0x1234567) atomicAdd(sink, acc);
}
template <int K>
static double run_coop(const uint4 *tab, uint64_t nblk, int waves_per_cu, int steps, unsigned long long *sink, const char *what)
{
	int blocks = 256 * waves_per_cu / 4;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL((k_gather_coop<K>), dim3(blocks), dim3(256), 0, 0, tab, nblk, steps / 4, sink);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL((k_gather_coop<K>), dim3(blocks), dim3(256), 0, 0, tab, nblk, steps, sink);
	hipEventRecord(e1, 0); hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	double lines = (double)blocks * 256 * K * steps;
	printf("%-28s table %6.0f MB  waves/CU %2d  K %d          : %8.3f ms  %7.2f G gathers/s  %7.1f GB/s (64 B each)\n", what,
	       nblk * 64 / 1e6, waves_per_cu, K, ms, lines / ms / 1e6, lines * 64 / ms / 1e6);
	fflush(stdout);
	return lines * 64 / ms / 1e6;
}

__global__ __launch_bounds__(256) void k_copy(const uint4 *src, uint4 *dst, size_t n)
{
	for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// streaming copy of `bytes` (read + write counted): the bandwidth ceiling a coalesced kernel sees
static double run_copy(const uint4 *src, uint4 *dst, size_t bytes)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k_copy, dim3(256 * 32), dim3(256), 0, 0, src, dst, bytes / 16);
	hipEventRecord(e0, 0);
	for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k_copy, dim3(256 * 32), dim3(256), 0, 0, src, dst, bytes / 16);
	hipEventRecord(e1, 0); hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	return 4.0 * 2.0 * bytes / ms / 1e6;
}

template <int K, int ST>
static double run(const uint4 *tab, uint64_t nblk, int waves_per_cu, int steps, uint4 *scratch, uint64_t per_lane, unsigned long long *sink, const char *what)
{
	int blocks = 256 * waves_per_cu / 4;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL((k_gather<K, ST>), dim3(blocks), dim3(256), 0, 0, tab, nblk, steps / 4, scratch, per_lane, sink);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL((k_gather<K, ST>), dim3(blocks), dim3(256), 0, 0, tab, nblk, steps, scratch, per_lane, sink);
	hipEventRecord(e1, 0); hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	double lines = (double)blocks * 256 * K * steps;
	if (what) printf("%-28s table %6.0f MB  waves/CU %2d  K %d  store %d : %8.3f ms  %7.2f G gathers/s  %7.1f GB/s (64 B each)\n", what,
	                 nblk * 64 / 1e6, waves_per_cu, K, ST, ms, lines / ms / 1e6, lines * 64 / ms / 1e6);
	fflush(stdout);
	return lines * 64 / ms / 1e6;
}

int main(int argc, char **argv)
{
	const uint64_t max_bytes = 8ull << 30;
	uint4 *tab; unsigned long long *sink; uint4 *scratch;
	if (hipMalloc(&tab, max_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
	hipMalloc(&sink, 8); hipMemset(sink, 0, 8);
	const uint64_t per_lane = 256;                         // 4 KB of scratch per lane
	hipMalloc(&scratch, (size_t)256 * 32 * 64 * per_lane * 16);
	hipMemset(tab, 0x5a, max_bytes);
	if (argc > 1 && argv[1][0] == 'q') {                     // quick mode for bench.py: one JSON line with the two measured ceilings
		const double cp = run_copy(tab, tab + (2ull << 30) / 16, 2ull << 30);
		const double ga = run<2, 0>(tab, 3072ull * (1ull << 20) / 64, 16, 400, scratch, per_lane, sink, nullptr);
		printf("{\"stream_copy_GBps\": %.1f, \"gather64_GBps\": %.1f, \"gather_table_MB\": 3221}\n", cp, ga);
		return 0;
	}
	if (argc > 1) {                                          // calibration run for the FETCH_SIZE counter: known bytes = gathers x 64
		run<1, 0>(tab, 3072ull * (1ull << 20) / 64, 16, 400, scratch, per_lane, sink, "calibration: dependent gather");
		printf("known bytes of the timed dispatch: %.0f (and a quarter of that for the warm-up dispatch)\n", 256.0 * 16 * 64 * 400 * 64);
		return 0;
	}
	for (uint64_t mb : { 64ull, 512ull, 3072ull, 8192ull }) {
		uint64_t nblk = mb * (1ull << 20) / 64;
		run_coop<1>(tab, nblk, 16, 400, sink, "quad-cooperative, 1/lane");
		run_coop<2>(tab, nblk, 16, 400, sink, "quad-cooperative, 2/lane");
		run_coop<2>(tab, nblk, 32, 400, sink, "quad-cooperative, 2/lane");
		run_coop<4>(tab, nblk, 16, 200, sink, "quad-cooperative, 4/lane");
		run<1, 0>(tab, nblk, 16, 400, scratch, per_lane, sink, "dependent gather");
		run<2, 0>(tab, nblk, 16, 400, scratch, per_lane, sink, "2 gathers/lane/step");
		run<2, 0>(tab, nblk, 32, 400, scratch, per_lane, sink, "2 gathers/lane/step");
		run<4, 0>(tab, nblk, 16, 200, scratch, per_lane, sink, "4 gathers/lane/step");
		run<2, 1>(tab, nblk, 16, 400, scratch, per_lane, sink, "2 gathers + 16 B store");
		run<2, 2>(tab, nblk, 16, 400, scratch, per_lane, sink, "2 gathers + 32 B store");
	}
	printf("streaming copy (2 GiB, read + write): %.1f GB/s\n", run_copy(tab, tab + (2ull << 30) / 16, 2ull << 30));
	return 0;
}
