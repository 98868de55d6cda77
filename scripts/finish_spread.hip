// Where do the ~35 us per launch beyond (bytes / streaming rate) go in a mid-column sweep (profiles/r3_team_pattern.md, section 4)?
// The sweep's load pattern (256 workgroups x 4 waves x 16 KiB per step, chunks of 32 columns strided over the grid, three
// tiles in flight; loads only) with a wall-clock stamp (s_memrealtime, 100 MHz) when each workgroup starts and when it ends.
//   hipcc -O3 --offload-arch=gfx950 scripts/finish_spread.hip -o /tmp/finish_spread && /tmp/finish_spread
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int U = 16;

struct Tile {
  f4 v[U];
  __device__ __forceinline__ void load(const float* p) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const f4*)(p + u * 256));
  }
  __device__ __forceinline__ void sum(f4& acc) const {
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
};

// mode 0: workgroup b takes chunks b, b + G, b + 2 G, ... (the sweeps' CgMap); 1: b ^ 1 instead of b (is it the workgroup or the
// address that is slow?); 2: rotated -- round r gives workgroup b the chunk (b + r) mod G; 3: rotated by 3 r
__global__ __launch_bounds__(256) void runs_kernel(const float* __restrict__ A, long m, long n, float* out, unsigned long long* stamps, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) stamps[2 * blockIdx.x] = wall_clock64();
  const long steps = n / gridDim.x;
  auto addr = [&](long i) -> const float* {
    const long r = i / 32;
    const long G = gridDim.x;
    long b = blockIdx.x;
    if (mode == 1) b ^= 1;
    if (mode == 2) b = (b + r) % G;
    if (mode == 3) b = (b + 3 * r) % G;
    const long col = (r * G + b) * 32 + (i % 32);
    return A + col * m + (long)wave * U * 256 + lane * 4;
  };
  f4 acc = {0, 0, 0, 0};
  Tile t0, t1, t2;
  t0.load(addr(0));
  t1.load(addr(1));
  long i = 0;
  for (; i + 3 <= steps - 2; i += 3) {
    t2.load(addr(i + 2));
    t0.sum(acc);
    t0.load(addr(i + 3));
    t1.sum(acc);
    t1.load(addr(i + 4));
    t2.sum(acc);
  }
  t0.sum(acc);
  t1.sum(acc);
  const float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 1.2345e-30f) out[0] = s;
  __syncthreads();
  if (threadIdx.x == 0) stamps[2 * blockIdx.x + 1] = wall_clock64();
}

__global__ void fill_random(float* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32) * 40503u ^ seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    p[i] = ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) * 0.01f;
  }
}

int main() {
  const long m = 16384;
  const size_t bytes_max = (size_t)64 << 30;
  float *A, *out;
  unsigned long long* stamps;
  CK(hipMalloc(&A, bytes_max));
  CK(hipMalloc(&out, 4));
  CK(hipMalloc(&stamps, 2 * 256 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, A, bytes_max / 4, 1u);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("# m = 16384, loads only, 256 workgroups; times in us relative to the first workgroup's start\n");
  for (int mode : {0, 1, 2, 3, 0, 2})
  for (long n : {131072L, 1048576L}) {
    std::vector<unsigned long long> h(512);
    float ms = 0;
    for (int r = 0; r < 4; ++r) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(runs_kernel, dim3(256), dim3(256), 0, 0, A, m, n, out, stamps, mode);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipMemcpy(h.data(), stamps, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < 256; ++b) t0 = std::min(t0, h[2 * b]);
    std::vector<double> st(256), en(256);
    for (int b = 0; b < 256; ++b) st[b] = (h[2 * b] - t0) * 0.01, en[b] = (h[2 * b + 1] - t0) * 0.01;
    std::vector<double> ss = st, es = en;
    std::sort(ss.begin(), ss.end()); std::sort(es.begin(), es.end());
    const double bytes = (double)m * n * 4;
    printf("mode %d n = %ld (%.1f GiB): events %.1f us = %.2f TB/s | starts: median %.1f max %.1f | ends: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f | "
           "bytes / (median end) = %.2f TB/s\n", mode, n, bytes / (1 << 30), ms * 1e3, bytes / ms / 1e9, ss[128], ss[255], es[0], es[25], es[128], es[230], es[255],
           bytes / es[128] / 1e6);
    double xs[8] = {0}, xm[8] = {0};
    for (int b = 0; b < 256; ++b) xs[b % 8] += en[b] / 32.0, xm[b % 8] = std::max(xm[b % 8], en[b]);
    printf("   mean / max end by blockIdx %% 8:");
    for (int x = 0; x < 8; ++x) printf("  %.1f/%.1f", xs[x], xm[x]);
    printf("\n");
  }
  return 0;
}
