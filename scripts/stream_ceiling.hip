// HBM ceiling microbenchmark for MI355X ("report a measured stream-copy ceiling from the build's own
// microbenchmark").  Read-only (the GEMV passes are >99.9 % reads) and copy, swept over launch geometry.
//   hipcc -O3 --offload-arch=gfx950 scripts/stream_ceiling.hip -o /tmp/stream_ceiling && /tmp/stream_ceiling [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ void read_kernel(const f4* __restrict__ p, size_t n4, float* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  f4 acc = {0, 0, 0, 0};
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  for (; i < n4; i += stride) acc += p[i];
  float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 1.2345e-30f) out[0] = s;  // keep the loads alive
}

// The access pattern of the GEMV sweeps: every wave streams contiguous runs of U KiB (U loads of 64 lanes x 16 B), two
// runs in flight (the second is issued before the first is consumed), runs dealt round-robin over all waves of the grid.
template <int U>
__global__ void read_runs_kernel(const f4* __restrict__ p, size_t n4, float* out) {
  const size_t run = (size_t)U * 64;  // f4 elements per run
  const size_t nruns = n4 / run;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, nwaves = (size_t)gridDim.x * blockDim.x / 64;
  const int lane = threadIdx.x & 63;
  f4 acc = {0, 0, 0, 0};
  f4 a[U], b[U];
  size_t r = wave;
  if (r < nruns) {
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = __builtin_nontemporal_load(p + r * run + u * 64 + lane);
  }
  for (; r < nruns; r += 2 * nwaves) {
    const size_t r1 = r + nwaves, r2 = r + 2 * nwaves;
    if (r1 < nruns) {
#pragma unroll
      for (int u = 0; u < U; ++u) b[u] = __builtin_nontemporal_load(p + r1 * run + u * 64 + lane);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += a[u];
    if (r2 < nruns) {
#pragma unroll
      for (int u = 0; u < U; ++u) a[u] = __builtin_nontemporal_load(p + r2 * run + u * 64 + lane);
    }
    if (r1 < nruns) {
#pragma unroll
      for (int u = 0; u < U; ++u) acc += b[u];
    }
  }
  float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 1.2345e-30f) out[0] = s;
}

template <int U>
__global__ void copy_kernel(const f4* __restrict__ p, f4* __restrict__ q, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], q + i + u * stride);
  }
  for (; i < n4; i += stride) q[i] = p[i];
}

template <typename F>
static double time_ms(F launch, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

// data like the bench's matrix entries (every word different): a zero-filled buffer streams measurably faster than data
__global__ void fill_random(float* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32) * 40503u ^ seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    p[i] = ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) * 0.01f;
  }
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 16.0;
  const bool zeros = argc > 2 && argv[2][0] == 'z';
  const size_t bytes = (size_t)(gib * (1ull << 30));
  const size_t n4 = bytes / 16;
  f4 *p, *q;
  float* out;
  CK(hipMalloc(&p, bytes)); CK(hipMalloc(&q, bytes)); CK(hipMalloc(&out, 4));
  if (zeros) {
    CK(hipMemset(p, 0, bytes));
  } else {
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (float*)p, bytes / 4, 1u);
  }
  CK(hipMemset(q, 0, bytes));
  CK(hipDeviceSynchronize());
  printf("# source buffer: %s\n", zeros ? "zeros" : "random data (every word different)");
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cu = prop.multiProcessorCount;
  printf("# %s, %d CUs, buffer %.1f GiB\n# kind threads blocks unroll nt ms GB/s\n", prop.gcnArchName, cu, gib);
  double best_r = 0, best_c = 0;
  for (int threads : {256, 512, 1024})
    for (int bpc : {1, 2, 4, 8, 16}) {
      const int blocks = cu * bpc;
      if ((long)threads * bpc > 2048) continue;
#define RUN_R(U, NT) { double ms = time_ms([&] { hipLaunchKernelGGL((read_kernel<U, NT>), dim3(blocks), dim3(threads), 0, 0, p, n4, out); }, 3); \
        double g = bytes / ms / 1e6; if (g > best_r) best_r = g; printf("read %d %d %d %d %.3f %.1f\n", threads, blocks, U, (int)NT, ms, g); }
      RUN_R(1, true) RUN_R(2, true) RUN_R(4, true) RUN_R(8, true) RUN_R(4, false)
#define RUN_C(U) { double ms = time_ms([&] { hipLaunchKernelGGL((copy_kernel<U>), dim3(blocks), dim3(threads), 0, 0, p, q, n4); }, 3); \
        double g = 2.0 * bytes / ms / 1e6; if (g > best_c) best_c = g; printf("copy %d %d %d 1 %.3f %.1f\n", threads, blocks, U, ms, g); }
      RUN_C(1) RUN_C(4)
    }
  double best_w = 0;
  for (int threads : {256, 512})
    for (int bpc : {1, 2, 4}) {
      const int blocks = cu * bpc;
#define RUN_W(U) { double ms = time_ms([&] { hipLaunchKernelGGL((read_runs_kernel<U>), dim3(blocks), dim3(threads), 0, 0, p, n4, out); }, 3); \
        double g = bytes / ms / 1e6; if (g > best_w) best_w = g; printf("read_runs %d %d %d 1 %.3f %.1f\n", threads, blocks, U, ms, g); }
      RUN_W(4) RUN_W(8)
    }
  printf("# best read-only %.1f GB/s, best read-only in wave-contiguous runs (the sweeps' pattern) %.1f GB/s, best copy (read+write) %.1f GB/s\n",
         best_r, best_w, best_c);
  return 0;
}
