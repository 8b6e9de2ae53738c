// fetch_calib.hip -- what rocprofv3's FETCH_SIZE counts for the access pattern of the seeding kernel (gfx950, MI355X).
//
// The guide's rule "FETCH_SIZE reports half the bytes" is calibrated for wide coalesced streaming reads (16 B per lane);
// bucket_hits_staged_kernel reads posting lists: runs of ~18 consecutive 2-byte genome numbers (counting pass) and of
// 8-byte postings (scatter pass), each run at an unrelated place of a multi-GB array, 32 lanes' worth of slots per list
// and eight lists per lane in flight.  This program reads a KNOWN set of such runs -- the host knows every byte asked for
// and every 64-byte line touched -- so that the counter's factor for THIS pattern is a measured number:
//     factor = FETCH_SIZE as counted / (64-byte lines touched x 64)
// Kernels (one dispatch each, names are what tools/fetch_calib.sh looks for):
//     calib_stream16   every lane 16 consecutive bytes, coalesced (the guide's case: expect 0.5)
//     calib_runs_u16   posting-list-shaped runs of 2-byte items
//     calib_runs_u64   posting-list-shaped runs of 8-byte items
// Build (build container): hipcc --offload-arch=gfx950 -O3 -o tools/fetch_calib tools/fetch_calib.hip
// Run (GPU box): bash tools/fetch_calib.sh <tag>   -> gpurun_out/<tag>_fetch_calibration.txt
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr uint64_t kFootprint = 2ull << 30;  // bytes of the array the runs lie in: far beyond the 256 MiB Infinity Cache
constexpr uint32_t kLists = 24u << 20;       // runs read per dispatch (x ~18 items: 450 million items)
constexpr int kLoads = 8;                    // lists per lane in flight, as in bucket_hits_staged_kernel

__host__ __device__ inline uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
// run i: its length (1 .. 35, mean 18: the benchmark's postings per list) and its first item
__host__ __device__ inline uint32_t run_len(uint32_t i) { return 1u + (uint32_t)(mix64(0x1234567ull + i) % 35u); }
__host__ __device__ inline uint64_t run_first(uint32_t i, uint64_t n_items) { return mix64(0x9e3779b97f4a7c15ull * (i + 1)) % (n_items - 64u); }

template <typename T>
__global__ __launch_bounds__(512) void calib_runs(const T *__restrict__ data, uint64_t n_items, uint32_t n_lists, unsigned long long *sink, unsigned long long never) {
  const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t list0 = wave * (2u * kLoads);  // sixteen lists per wave and step: two halves of 32 lanes, eight loads each
  unsigned long long acc = 0;
  uint64_t lo[kLoads];
  uint32_t n[kLoads];
#pragma unroll
  for (int u = 0; u < kLoads; ++u) {
    const uint32_t i = list0 + 2u * (uint32_t)u + (lane >> 5);
    lo[u] = i < n_lists ? run_first(i, n_items) : 0u;
    n[u] = i < n_lists ? run_len(i) : 0u;
  }
  for (uint32_t r = 0; r < 64u; r += 32u) {
    const uint32_t slot = r + (lane & 31u);
    T v[kLoads];
#pragma unroll
    for (int u = 0; u < kLoads; ++u) v[u] = slot < n[u] ? data[lo[u] + slot] : T(0);
#pragma unroll
    for (int u = 0; u < kLoads; ++u) acc += (unsigned long long)v[u];
  }
  if (acc == never) *sink = acc;  // (a value the sums never take, unknown to the compiler: keeps the loads)
}
__global__ __launch_bounds__(256) void calib_stream16(const uint4 *__restrict__ data, uint64_t n16, unsigned long long *sink, unsigned long long never) {
  unsigned long long acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 v = data[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == never) *sink = acc;
}

template <typename T>
static void host_counts(uint64_t n_items, double *asked, double *lines) {
  // bytes asked for, and distinct 64-byte lines touched, run by run (two runs touching the same line by chance: counted twice;
  // with 24 million runs of ~0.1 KB in 2 GiB that is about one line in forty)
  uint64_t a = 0, l = 0;
  for (uint32_t i = 0; i < kLists; ++i) {
    const uint64_t first = run_first(i, n_items) * sizeof(T), bytes = (uint64_t)run_len(i) * sizeof(T);
    a += bytes;
    l += (first + bytes - 1) / 64 - first / 64 + 1;
  }
  *asked = (double)a;
  *lines = (double)l * 64.0;
}

int main() {
  void *buf = nullptr;
  unsigned long long *sink = nullptr;
  CHECK(hipMalloc(&buf, kFootprint));
  CHECK(hipMalloc(&sink, 8));
  CHECK(hipMemset(buf, 1, kFootprint));
  CHECK(hipDeviceSynchronize());
  const uint32_t waves = (kLists + 2 * kLoads - 1) / (2 * kLoads), blocks = (waves + 7) / 8;
  for (int rep = 0; rep < 3; ++rep) {  // (the first dispatch of each kernel is a warm-up the summary drops)
    hipLaunchKernelGGL(calib_stream16, dim3(256 * 16), dim3(256), 0, 0, (const uint4 *)buf, kFootprint / 16, sink, ~0ull);
    hipLaunchKernelGGL(calib_runs<uint16_t>, dim3(blocks), dim3(512), 0, 0, (const uint16_t *)buf, kFootprint / 2, kLists, sink, ~0ull);
    hipLaunchKernelGGL(calib_runs<uint64_t>, dim3(blocks), dim3(512), 0, 0, (const uint64_t *)buf, kFootprint / 8, kLists, sink, ~0ull);
    CHECK(hipDeviceSynchronize());
  }
  double a16, l16, a64, l64;
  host_counts<uint16_t>(kFootprint / 2, &a16, &l16);
  host_counts<uint64_t>(kFootprint / 8, &a64, &l64);
  printf("calib_stream16 bytes_asked %.0f line_bytes %.0f\n", (double)kFootprint, (double)kFootprint);
  printf("calib_runs<unsigned short> bytes_asked %.0f line_bytes %.0f lists %u\n", a16, l16, kLists);
  printf("calib_runs<unsigned long> bytes_asked %.0f line_bytes %.0f lists %u\n", a64, l64, kLists);
  return 0;
}
