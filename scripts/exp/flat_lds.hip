// experiment: 80-byte records in LDS reached through a generic pointer (flat instructions)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
struct Rec { int64_t a, b; float f; int32_t v[13]; int32_t pad; };
static_assert(sizeof(Rec) == 80, "");
__global__ __launch_bounds__(64) void k(Rec *g, int n, int use_lds, int *out)
{
	__shared__ uint8_t s_pad[8 + 700];
	__shared__ __attribute__((aligned(16))) Rec s_list[96];
	__shared__ __attribute__((aligned(16))) Rec s_tmp[96];
	const int l = threadIdx.x;
	s_pad[l] = (uint8_t)l;
	Rec *L = use_lds ? s_list : g;
	Rec *T = use_lds ? s_tmp : g + 96;
	if (use_lds) { __syncthreads(); for (int k = l; k < n; k += 64) s_list[k] = g[k]; __threadfence_block(); __syncthreads(); }
	for (int k = l; k < n; k += 64) { T[k] = L[n - 1 - k]; T[k].v[0] += s_pad[l]; }
	__threadfence_block(); __syncthreads();
	if (l == 0) { for (int i = 1; i < n; ++i) { Rec *p = &T[i]; if (p->a < T[i-1].a) p->b = p->a; } }
	__threadfence_block(); __syncthreads();
	for (int k = l; k < n; k += 64) L[k] = T[k];
	__threadfence_block(); __syncthreads();
	if (use_lds) for (int k = l; k < n; k += 64) g[k] = s_list[k];
	if (l == 0) *out = (int)L[0].a;
}
int main()
{
	Rec *g; int *out;
	hipMalloc(&g, sizeof(Rec) * 192); hipMalloc(&out, 4);
	Rec h[192]; for (int i = 0; i < 192; ++i) { h[i].a = i; h[i].b = -i; h[i].f = i; for (int j = 0; j < 13; ++j) h[i].v[j] = j; h[i].pad = 0; }
	for (int mode = 0; mode < 2; ++mode) {
		hipMemcpy(g, h, sizeof h, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(k, dim3(4), dim3(64), 0, 0, g, 50, mode, out);
		hipError_t e = hipDeviceSynchronize();
		int o = -1; hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost);
		printf("mode %d: %s out=%d\n", mode, hipGetErrorString(e), o);
	}
	return 0;
}
