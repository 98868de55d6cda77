// Does a second read of a chunk of A come out of the 256 MB Infinity Cache (memory-side) instead of HBM?  If it does, a row-team sweep could read a chunk,
// exchange the chunk's dots in ONE message per device, and take the chunk a second time from that cache for A v -- no per-step posts, no parked tiles.
// One persistent launch: workgroup w reads its slice of chunk c (chunk bytes / workgroups) once or twice, chunk after chunk; the slices of a chunk
// together are `chunk` bytes, so between a workgroup's two reads of a line the device has pulled about one chunk through the L2s (32 MB in all).
//   hipcc -O3 --offload-arch=gfx950 scripts/kernel_lab/mall_reread.hip -o build/mall && build/mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void reread(const v4* __restrict__ a, size_t chunk_v4, size_t nchunks, int reads, float* out) {
  const size_t slice = chunk_v4 / gridDim.x;  // 16-byte elements per workgroup and chunk
  v4 acc = {0, 0, 0, 0};
  for (size_t c = 0; c < nchunks; ++c) {
    const v4* p = a + c * chunk_v4 + (size_t)blockIdx.x * slice;
    for (int r = 0; r < reads; ++r) {
      for (size_t i = threadIdx.x; i < slice; i += 1024) {  // four loads in flight per lane
        v4 x0, x1, x2, x3;
        if (NT) {
          x0 = __builtin_nontemporal_load(p + i), x1 = __builtin_nontemporal_load(p + i + 256), x2 = __builtin_nontemporal_load(p + i + 512), x3 = __builtin_nontemporal_load(p + i + 768);
        } else {
          x0 = p[i], x1 = p[i + 256], x2 = p[i + 512], x3 = p[i + 768];
        }
        acc += x0 + x1 + x2 + x3;
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}

int main() {
  const size_t total = (size_t)8 << 30;
  v4* a;
  float* out;
  if (hipMalloc((void**)&a, total + (1 << 20)) != hipSuccess || hipMalloc((void**)&out, 1 << 20) != hipSuccess) return printf("alloc failed\n"), 1;
  (void)hipMemset(a, 0, total);
  const int grid = 2048;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  for (int nt = 0; nt < 2; ++nt)
    for (size_t chunk_mb : {32, 64, 128, 192, 256, 512}) {
      const size_t chunk_v4 = (chunk_mb << 20) / 16, nchunks = total / (chunk_mb << 20);
      float ms[3] = {0, 0, 0};
      for (int reads = 1; reads <= 2; ++reads)
        for (int rep = 0; rep < 2; ++rep) {
          (void)hipEventRecord(e0);
          if (nt) hipLaunchKernelGGL(reread<true>, dim3(grid), dim3(256), 0, 0, a, chunk_v4, nchunks, reads, out);
          else hipLaunchKernelGGL(reread<false>, dim3(grid), dim3(256), 0, 0, a, chunk_v4, nchunks, reads, out);
          (void)hipEventRecord(e1);
          if (hipDeviceSynchronize() != hipSuccess) return printf("kernel failed\n"), 1;
          (void)hipEventElapsedTime(&ms[reads], e0, e1);
        }
      printf("%s loads, chunk %4zu MB: one read %.3f ms = %.2f TB/s | two reads %.3f ms = %.2f TB/s of loads, %.2f x the time of one\n", nt ? "nt   " : "plain", chunk_mb, ms[1],
             total / ms[1] / 1e9, ms[2], 2.0 * total / ms[2] / 1e9, ms[2] / ms[1]);
    }
  return 0;
}
