// Is the long-column sweep's gap to the mid-column one (0.87 against 0.91 of 8 TB/s) in its ACCESS PATTERN?  Loads only, no
// exchange, no LDS: the team kernel's pattern -- TM workgroups share a column, each streams its 64 KiB piece (four waves x
// 16 KiB) of one column per step, the next step is one column (m * 4 bytes) further on; teams take chunks of 32 columns,
// strided over the grid; three tiles in flight -- for column lengths 16384 (TM = 1) .. 262144 (TM = 16), same total bytes.
// (PG_TNT_EXPERIMENT: the team kernel without waits, LDS read-back and parking runs no faster -- profiles/r3_team_pattern.md.)
//   hipcc -O3 --offload-arch=gfx950 scripts/strided_runs.hip -o /tmp/strided_runs && /tmp/strided_runs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int U = 16;  // 1 KiB row groups per wave and step

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

// order 0: the team kernel's (a team = workgroups t, t + nteams, ...: same XCD under round-robin dispatch)
// order 1: every workgroup streams ONE contiguous region of the matrix (same bytes per workgroup, no sharing of columns)
template <int ORDER>
__global__ __launch_bounds__(256) void runs_kernel(const float* __restrict__ A, long m, long n, int tm, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nteams = gridDim.x / tm;
  const int team = blockIdx.x % nteams, member = blockIdx.x / nteams;
  const long steps = n / nteams;  // columns per team (n is a multiple of 32 * nteams)
  auto addr = [&](long i) -> const float* {
    if (ORDER == 1) {
      const long bytes_per_wg = m * n / gridDim.x;  // floats
      return A + (long)blockIdx.x * bytes_per_wg + i * (U * 4 * 256) + (long)wave * U * 256 + lane * 4;
    }
    const long col = ((i / 32) * nteams + team) * 32 + (i % 32);
    return A + col * m + ((long)member * 4 + wave) * U * 256 + lane * 4;
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
}

__global__ void fill_random(float* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32) * 40503u ^ seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    p[i] = ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) * 0.01f;
  }
}

int main() {
  const size_t bytes = (size_t)64 << 30;
  float *A, *out;
  CK(hipMalloc(&A, bytes));
  CK(hipMalloc(&out, 4));
  hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, A, bytes / 4, 1u);
  CK(hipDeviceSynchronize());
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cu = prop.multiProcessorCount;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("# %s, %d CUs, 64 GiB, loads only, 256 workgroups x 4 waves x 16 KiB per step, three tiles in flight\n", prop.gcnArchName, cu);
  printf("# rows  members  order  ms  TB/s\n");
  for (int rep = 0; rep < 2; ++rep)
    for (int tm = 1; tm <= 16; tm *= 2) {
      const long m = 16384L * tm, n = (long)(bytes / 4) / m;
      for (int order = 0; order < 2; ++order) {
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
          CK(hipEventRecord(e0));
          if (order == 0) hipLaunchKernelGGL(runs_kernel<0>, dim3(cu), dim3(256), 0, 0, A, m, n, tm, out);
          else hipLaunchKernelGGL(runs_kernel<1>, dim3(cu), dim3(256), 0, 0, A, m, n, tm, out);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (r > 0 && ms < best) best = ms;
        }
        printf("%7ld %3d  %s  %.3f  %.3f\n", m, tm, order == 0 ? "team pattern " : "contiguous/wg", best, bytes / best / 1e9);
      }
    }
  return 0;
}
