// Experiment (round 2): what limits the one-bucket-per-lane gather on a GRCh37-scale index (6.85 GB) - table size (TLB reach),
// the 64-distinct-lines-per-instruction access shape, or the ALU of the rank?  Modes:
//   lane      : each lane loads its own 128-B bucket with 8 x global_load_dwordx4 (what kl_search / kl_calc_d do)
//   coop      : the wave loads the same 64 buckets cooperatively - instruction r: lane l loads slice (l & 7) of the bucket of
//               lane 8r + (l >> 3), so one instruction touches 8 lines instead of 64 - and transposes through LDS
//   *_nop     : the same without the rank ALU (xor of the words), to separate memory from ALU
//   meta8/16  : (round 3) the shape of the search kernel's per-lane METADATA - every lane loads 8 or 16 bytes from its own random
//               place of the table (a per-position record, a heap entry): 64 different 64-byte sectors per wave instruction.
//               Under `rocprofv3 --pmc FETCH_SIZE` (tools/pmc_traffic.sh) this calibrates how the counter tallies such requests:
//               FETCH_SIZE x 1024 / (loads x 64 B) - next to the 0.50 it shows for 128-byte bucket requests (k_coop, k_lane).
// usage: gather_bench <table_MiB> [window_MiB] [meta]  (window: positions are drawn from the first window_MiB only; meta: only the
//        calibration kernels k_coop<false,1>, k_meta<8>, k_meta<16>, one launch each, 2^27 loads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; return x; }

__device__ __forceinline__ unsigned long long rank15(const uint4 d[8], uint32_t off) {
  uint32_t cnt[16];
#pragma unroll
  for (int c = 0; c < 16; c++) cnt[c] = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const uint4 p = d[4 + w];
    const int nv = (int)off + 1 - 32 * w;
    const uint32_t m = nv <= 0 ? 0u : (nv >= 32 ? 0xFFFFFFFFu : ((1u << nv) - 1u));
    const uint32_t a[4] = {~p.x & ~p.y, p.x & ~p.y, ~p.x & p.y, p.x & p.y};
    const uint32_t m2 = m & ~p.z, m2p = m & p.z;
    const uint32_t b[4] = {m2 & ~p.w, m2p & ~p.w, m2 & p.w, m2p & p.w};
#pragma unroll
    for (int c = 1; c < 16; c++) cnt[c] += __popc(a[c & 3] & b[c >> 2]);
  }
  const uint32_t* cw = (const uint32_t*)&d[0];
  unsigned long long acc = 0;
#pragma unroll
  for (int c = 1; c < 16; c++) acc += (unsigned long long)(cnt[c] + cw[c]) * (c + 1);
  return acc;
}
__device__ __forceinline__ unsigned long long xor8(const uint4 d[8]) {
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) v ^= d[k].x ^ d[k].y ^ d[k].z ^ d[k].w;
  return v;
}

template <bool ALU, int U>
__global__ __launch_bounds__(256) void k_lane(const uint4* __restrict__ buckets, uint64_t npos, uint64_t n, uint64_t seed, unsigned long long* out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long acc = 0;
  for (uint64_t q = tid * U; q < n; q += nth * U) {
    uint4 d[U][8];
    uint32_t off[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint64_t pos = mix((q + u) * 0x9E3779B97F4A7C15ull + seed) % npos;
      off[u] = pos & 127;
      const uint4* b = buckets + (pos >> 7) * 8;
#pragma unroll
      for (int k = 0; k < 8; k++) d[u][k] = b[k];
    }
#pragma unroll
    for (int u = 0; u < U; u++) acc += ALU ? rank15(d[u], off[u]) : xor8(d[u]);
  }
  atomicAdd(out, acc);
}

// cooperative load + LDS transpose: per wave 64 lanes x 8 slices; row stride 9 uint4 (144 B) to spread the banks
template <bool ALU, int U>
__global__ __launch_bounds__(256) void k_coop(const uint4* __restrict__ buckets, uint64_t npos, uint64_t n, uint64_t seed, unsigned long long* out) {
  __shared__ uint4 stage[4][U][64 * 9];
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long acc = 0;
  for (uint64_t q = tid * U; q < n; q += nth * U) { // (n is a multiple of the thread count: all lanes of a wave iterate together)
    uint32_t off[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint64_t pos = mix((q + u) * 0x9E3779B97F4A7C15ull + seed) % npos;
      off[u] = pos & 127;
      const uint64_t blk = pos >> 7;
      const uint32_t blo = (uint32_t)blk, bhi = (uint32_t)(blk >> 32);
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const int owner = 8 * r + (lane >> 3);
        const uint64_t ob = ((uint64_t)(uint32_t)__shfl((int)bhi, owner) << 32) | (uint32_t)__shfl((int)blo, owner);
        stage[wv][u][owner * 9 + (lane & 7)] = buckets[ob * 8 + (lane & 7)];
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < U; u++) {
      uint4 d[8];
#pragma unroll
      for (int k = 0; k < 8; k++) d[k] = stage[wv][u][lane * 9 + k];
      acc += ALU ? rank15(d, off[u]) : xor8(d);
    }
    __builtin_amdgcn_wave_barrier();
  }
  atomicAdd(out, acc);
}

// one W-byte load per lane from a random W-aligned place of the table
template <int W>
__global__ __launch_bounds__(256) void k_meta(const uint4* __restrict__ table, uint64_t nbytes, uint64_t n, uint64_t seed, unsigned long long* out) {
  const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long acc = 0;
  for (uint64_t q = tid; q < n; q += nth) {
    const uint64_t off = (mix(q * 0x9E3779B97F4A7C15ull + seed) % (nbytes / W)) * W;
    if (W == 8) { const uint2 v = *(const uint2*)((const char*)table + off); acc += v.x ^ v.y; }
    else { const uint4 v = *(const uint4*)((const char*)table + off); acc += v.x ^ v.y ^ v.z ^ v.w; }
  }
  atomicAdd(out, acc);
}

int main(int argc, char** argv) {
  const uint64_t mib = argc > 1 ? strtoull(argv[1], 0, 10) : 1024, win = argc > 2 ? strtoull(argv[2], 0, 10) : mib;
  const uint64_t bytes = mib << 20, npos = (win << 20) - 1; // one position per byte of the window: bucket = pos >> 7
  uint4* d;
  if (hipMalloc(&d, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(d, 0x5a, bytes);
  unsigned long long* out; hipMalloc(&out, 16); hipMemset(out, 0, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const uint64_t n = 1ull << 27;
  printf("table %llu MiB, positions drawn from the first %llu MiB, %llu queries\n", (unsigned long long)mib, (unsigned long long)win, (unsigned long long)n);
  auto run = [&](auto kern, const char* name, int grid) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, npos, n, 7ull + rep, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("%-22s grid %5d: %8.3f ms  %6.2f Gvisit/s  %7.1f GB/s(128B)\n", name, grid, best, n / best / 1e6, n * 128.0 / best / 1e6);
  };
  if (argc > 3) { // calibration launches for the PMC passes: known numbers of requests of each shape
    hipLaunchKernelGGL((k_coop<false, 1>), dim3(1024), dim3(256), 0, 0, d, npos, n, 11ull, out);
    hipLaunchKernelGGL((k_meta<8>), dim3(2048), dim3(256), 0, 0, d, (win << 20), n, 12ull, out);
    hipLaunchKernelGGL((k_meta<16>), dim3(2048), dim3(256), 0, 0, d, (win << 20), n, 13ull, out);
    hipDeviceSynchronize();
    printf("calibration: k_coop<false,1> %llu x 128 B bucket requests; k_meta<8>, k_meta<16> %llu loads each (one 64-byte sector per load)\n", (unsigned long long)n, (unsigned long long)n);
    return 0;
  }
  for (int grid : {512, 1024, 2048}) {
    run(k_lane<false, 1>, "lane_nop U=1", grid);
    run(k_lane<false, 2>, "lane_nop U=2", grid);
    run(k_lane<true, 1>, "lane U=1", grid);
    run(k_lane<true, 2>, "lane U=2", grid);
    run(k_coop<false, 1>, "coop_nop U=1", grid);
    run(k_coop<false, 2>, "coop_nop U=2", grid);
    run(k_coop<true, 1>, "coop U=1", grid);
    run(k_coop<true, 2>, "coop U=2", grid);
  }
  return 0;
}
