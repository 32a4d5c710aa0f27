// Developer probe: how much LDS per 256-thread block still lets THREE blocks be resident on a CU of this device?
// (hipOccupancyMaxActiveBlocksPerMultiprocessor answered 3 for 54 048 bytes per block where the hardware ran 2: profiles/r4_ab_steps.txt,
// sessions 9-10.)  For every size: 3 x #CU blocks, each announces itself and waits - with a time limit - until all have; the blocks see
// each other only if they are all resident at once.   hipcc --offload-arch=gfx950 -O2 -o lds_probe lds_probe.hip && ./lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void probe(unsigned *arrived, unsigned *saw_all, unsigned want, long long limit_ticks) {
	extern __shared__ unsigned char lds[];
	if (threadIdx.x == 0) {
		lds[0] = 1; // (the allocation is used)
		__hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const long long t0 = wall_clock64();
		unsigned ok = 0;
		while (wall_clock64() - t0 < limit_ticks) {
			if (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { ok = 1; break; }
			__builtin_amdgcn_s_sleep(32);
		}
		saw_all[blockIdx.x] = ok;
	}
}

int main(int argc, char **argv) {
	hipDeviceProp_t pr;
	if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
	const int ncu = pr.multiProcessorCount, per_cu = argc > 1 ? atoi(argv[1]) : 3;
	int rate_khz = 100000;
	(void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
	const long long limit = (long long)rate_khz * 30; // 30 ms
	const unsigned grid = (unsigned)(ncu * per_cu);
	unsigned *d_arr, *d_saw;
	hipMalloc(&d_arr, 4); hipMalloc(&d_saw, grid * 4);
	hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
	printf("# %s: %d CUs, LDS per CU as reported %zu bytes (maxSharedMemoryPerMultiProcessor), per block %zu; %d blocks of 256 threads per CU\n", pr.gcnArchName, ncu,
	       (size_t)pr.maxSharedMemoryPerMultiProcessor, (size_t)pr.sharedMemPerBlock, per_cu);
	printf("# bytes_per_block  occupancy_query  blocks_that_saw_all_of_%u\n", grid);
	std::vector<unsigned> h(grid);
	for (int sz = 51200; sz <= 55296; sz += 128) {
		int occ = 0;
		(void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)probe, 256, (size_t)sz);
		hipMemset(d_arr, 0, 4); hipMemset(d_saw, 0, grid * 4);
		hipLaunchKernelGGL(probe, dim3(grid), dim3(256), (size_t)sz, 0, d_arr, d_saw, grid, limit);
		if (hipDeviceSynchronize() != hipSuccess) { printf("%d launch failed\n", sz); continue; }
		hipMemcpy(h.data(), d_saw, grid * 4, hipMemcpyDeviceToHost);
		unsigned n = 0;
		for (unsigned v : h) n += v;
		printf("%d %d %u%s\n", sz, occ, n, n == grid ? "" : "   <- not all resident");
	}
	return 0;
}
