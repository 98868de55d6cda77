// Does a SCALAR store (s_store_dwordx4 + s_dcache_wb) on gfx950 reach uncached device memory while the kernel is still running -- i.e. could the
// row-team sweep post its granules outside the wave's vector-memory queue?  Workgroup 2k posts, workgroup 2k + 1 polls the same 16 bytes with
// system-scope vector loads (bounded).  Prints how many pairs saw the value and the mean number of polls.
//   hipcc -O3 --offload-arch=gfx950 scripts/kernel_lab/scalar_store_probe.hip -o /tmp/ssp && /tmp/ssp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void probe(unsigned long long* box, unsigned* seen, unsigned* polls, int rounds, int use_scalar) {
  const int pair = blockIdx.x >> 1;
  unsigned long long* p = box + (size_t)pair * 16;  // 128 bytes per pair
  if ((blockIdx.x & 1) == 0) {
    for (int r = 1; r <= rounds; ++r) {
      if (use_scalar) {
        u4 d = {(unsigned)r, 0xA5A50000u + (unsigned)pair, (unsigned)r, 0x5A5A0000u + (unsigned)pair};
        d.x = __builtin_amdgcn_readfirstlane(d.x), d.y = __builtin_amdgcn_readfirstlane(d.y);
        d.z = __builtin_amdgcn_readfirstlane(d.z), d.w = __builtin_amdgcn_readfirstlane(d.w);
        asm volatile("s_store_dwordx4 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::"s"(d), "s"(p) : "memory");
      } else if (threadIdx.x < 2) {
        const unsigned long long w = ((unsigned long long)((threadIdx.x ? 0x5A5A0000u : 0xA5A50000u) + (unsigned)pair) << 32) | (unsigned)r;
        __hip_atomic_store(p + threadIdx.x, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      // wait for the consumer's acknowledgement of round r (word 8 of the pair's line, written by a vector store)
      long long spins = 0;
      while (__hip_atomic_load(p + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned long long)r && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
      if (spins >= (1 << 22)) return;
    }
  } else {
    unsigned ok = 0, np = 0;
    for (int r = 1; r <= rounds; ++r) {
      long long spins = 0;
      bool got = false;
      while (++spins < (1 << 22)) {
        const unsigned long long w0 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long w1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)w0 == (unsigned)r && (unsigned)w1 == (unsigned)r && (unsigned)(w0 >> 32) == 0xA5A50000u + (unsigned)pair &&
            (unsigned)(w1 >> 32) == 0x5A5A0000u + (unsigned)pair) {
          got = true;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      np += (unsigned)spins;
      if (!got) break;
      ok++;
      if (threadIdx.x == 0) __hip_atomic_store(p + 8, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0) seen[pair] = ok, polls[pair] = np;
  }
}

int main() {
  const int pairs = 256, rounds = 200;
  unsigned long long* box;
  unsigned *seen, *polls;
  if (hipExtMallocWithFlags((void**)&box, pairs * 128, hipDeviceMallocUncached) != hipSuccess) return printf("no uncached memory\n"), 1;
  hipMalloc(&seen, pairs * 4), hipMalloc(&polls, pairs * 4);
  for (int use_scalar = 0; use_scalar < 2; ++use_scalar) {
    hipMemset(box, 0, pairs * 128), hipMemset(seen, 0, pairs * 4), hipMemset(polls, 0, pairs * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe, dim3(2 * pairs), dim3(64), 0, 0, box, seen, polls, rounds, use_scalar);
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) return printf("kernel failed\n"), 1;
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> s(pairs), q(pairs);
    hipMemcpy(s.data(), seen, pairs * 4, hipMemcpyDeviceToHost), hipMemcpy(q.data(), polls, pairs * 4, hipMemcpyDeviceToHost);
    long long all = 0, pl = 0;
    int full = 0;
    for (int i = 0; i < pairs; ++i) all += s[i], pl += q[i], full += s[i] == (unsigned)rounds;
    printf("%s post: %d of %d pairs saw all %d rounds (%lld round trips), %.1f polls per round, %.3f ms -> %.2f us per round trip\n", use_scalar ? "SCALAR" : "vector",
           full, pairs, rounds, all, all ? (double)pl / all : 0.0, ms, 1e3 * ms / rounds);
  }
  return 0;
}
