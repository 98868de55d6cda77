// Infinity-Cache (MALL, 256 MiB) experiment for north_star's ROW layout ("MALL-resident row panels").
// Question: if pass T streams a column panel of the row block and pass N re-reads the same panel a
// little later (after the all-reduce of the panel's gradient entries has landed), does the second read come out of the
// Infinity Cache faster than an HBM stream -- and how many bytes may pass in between before it does not?
//
//   hipcc -O3 --offload-arch=gfx950 scripts/mall_panel.hip -o /tmp/mall_panel && /tmp/mall_panel
//
// Kernel durations are taken INSIDE the kernel (wall_clock64 of the first workgroup to start / last to finish): the
// panels are 16..128 MiB = 2..20 us of streaming, too short for event pairs.
//   part E  resident read rate: one buffer of S MiB read 20x back to back (S = 8 .. 1024 MiB)
//   part C  re-read after distance: read cold panel X, stream d MiB of other cold memory, read X again (timed)
//   part D  additivity: a cold HBM stream and a resident-panel re-read loop on two streams, alone and together
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// the sweeps' access pattern (scripts/stream_ceiling.hip: read_runs): every wave streams contiguous 8 KiB runs, two in flight
template <bool NT>
__global__ void read_runs(const f4* __restrict__ p, size_t n4, float* out, unsigned long long* t0, unsigned long long* t1) {
  constexpr int U = 8;
  if (threadIdx.x == 0 && t0) t0[blockIdx.x] = wall_clock64();
  const size_t run = (size_t)U * 64;
  const size_t nruns = n4 / run;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, nwaves = (size_t)gridDim.x * blockDim.x / 64;
  const int lane = threadIdx.x & 63;
  f4 acc = {0, 0, 0, 0};
  f4 a[U], b[U];
  size_t r = wave;
#define LD(q) (NT ? __builtin_nontemporal_load(q) : *(q))
  if (r < nruns) {
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = LD(p + r * run + u * 64 + lane);
  }
  for (; r < nruns; r += 2 * nwaves) {
    const size_t r1 = r + nwaves, r2 = r + 2 * nwaves;
    if (r1 < nruns) {
#pragma unroll
      for (int u = 0; u < U; ++u) b[u] = LD(p + r1 * run + u * 64 + lane);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += a[u];
    if (r2 < nruns) {
#pragma unroll
      for (int u = 0; u < U; ++u) a[u] = LD(p + r2 * run + u * 64 + lane);
    }
    if (r1 < nruns) {
#pragma unroll
      for (int u = 0; u < U; ++u) acc += b[u];
    }
  }
#undef LD
  float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 1.2345e-30f) out[0] = s;
  __syncthreads();
  if (threadIdx.x == 0 && t1) t1[blockIdx.x] = wall_clock64();
}

static int g_blocks = 256, g_threads = 256;
static unsigned long long *d_t0, *d_t1;
static float* d_out;
static double g_tick_us = 0.01;

template <bool NT>
static void launch(const void* p, size_t bytes, hipStream_t s, bool timed) {
  hipLaunchKernelGGL((read_runs<NT>), dim3(g_blocks), dim3(g_threads), 0, s, (const f4*)p, bytes / 16, d_out,
                     timed ? d_t0 : nullptr, timed ? d_t1 : nullptr);
}
static void launch_nt(bool nt, const void* p, size_t bytes, hipStream_t s, bool timed) {
  if (nt) launch<true>(p, bytes, s, timed); else launch<false>(p, bytes, s, timed);
}

// duration of the last timed launch, microseconds
static double last_us() {
  std::vector<unsigned long long> a(g_blocks), b(g_blocks);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(a.data(), d_t0, g_blocks * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), d_t1, g_blocks * 8, hipMemcpyDeviceToHost));
  return (double)(*std::max_element(b.begin(), b.end()) - *std::min_element(a.begin(), a.end())) * g_tick_us;
}

int main(int argc, char** argv) {
  const size_t MiB = 1ull << 20;
  const size_t big = (argc > 1 ? (size_t)atoi(argv[1]) : 8192) * MiB;
  char* p;
  CK(hipMalloc(&p, big));
  CK(hipMemset(p, 0, big));
  CK(hipMalloc(&d_out, 4));
  CK(hipMalloc(&d_t0, 4096 * 8));
  CK(hipMalloc(&d_t1, 4096 * 8));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int khz = 100000;
  hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  g_tick_us = 1e3 / (double)khz;
  printf("# %s, %d CUs, L2 %d MiB, wall clock %d kHz, buffer %zu MiB\n", prop.gcnArchName, prop.multiProcessorCount,
         prop.l2CacheSize >> 20, khz, big / MiB);

  for (int geo = 0; geo < 2; ++geo) {
    g_blocks = geo == 0 ? 256 : 512;
    g_threads = 256;
    printf("\n## geometry: %d workgroups x %d threads\n", g_blocks, g_threads);

    printf("\n### part E: one buffer of S MiB read 20x back to back (in-kernel clock, median of the last 10)\n");
    printf("| S MiB | plain loads GB/s | nontemporal loads GB/s |\n|---:|---:|---:|\n");
    for (size_t S : {8, 16, 32, 48, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 4096}) {
      if (S * MiB > big) continue;
      double g[2];
      for (int nt = 0; nt < 2; ++nt) {
        std::vector<double> us;
        for (int r = 0; r < 20; ++r) {
          launch_nt(nt, p, S * MiB, 0, true);
          if (r >= 10) us.push_back(last_us());
        }
        std::sort(us.begin(), us.end());
        g[nt] = S * MiB / us[us.size() / 2] / 1e3;
      }
      printf("| %zu | %.0f | %.0f |\n", S, g[0], g[1]);
    }

    printf("\n### part C: read cold panel X, stream d MiB of other cold memory, read X again (second read timed; median of 8)\n");
    for (int nt = 0; nt < 2; ++nt) {
      printf("\n%s loads\n\n| panel MiB | d = 0 | 32 | 64 | 96 | 128 | 160 | 192 | 224 | 256 | 384 | first (cold) read |\n|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n",
             nt ? "nontemporal" : "plain");
      for (size_t S : {16, 32, 64, 128}) {
        printf("| %zu |", S);
        double cold = 0;
        for (size_t d : {0, 32, 64, 96, 128, 160, 192, 224, 256, 384}) {
          std::vector<double> us, usc;
          size_t off = 0;
          for (int r = 0; r < 8; ++r) {
            if (off + (S + d) * MiB > big) off = 0;
            launch_nt(nt, p + off, S * MiB, 0, true);  // first touch of X in this round (cold: the buffer is 8 GiB)
            usc.push_back(last_us());
            if (d) launch_nt(nt, p + off + S * MiB, d * MiB, 0, false);
            launch_nt(nt, p + off, S * MiB, 0, true);
            us.push_back(last_us());
            off += (S + d + 64) * MiB;
          }
          std::sort(us.begin(), us.end());
          std::sort(usc.begin(), usc.end());
          cold = S * MiB / usc[usc.size() / 2] / 1e3;
          printf(" %.0f |", S * MiB / us[us.size() / 2] / 1e3);
        }
        printf(" %.0f |\n", cold);
      }
    }
  }

  // part D: additivity of an HBM stream and a resident re-read (two streams; event-timed, the runs are milliseconds long)
  g_blocks = 256;
  printf("\n### part D: cold stream of 4 GiB on stream A, resident 64 MiB panel re-read 64x (4 GiB) on stream B -- alone and together\n");
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa));
  CK(hipStreamCreate(&sb));
  hipEvent_t e0, e1, f0, f1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
  const size_t cold_bytes = std::min(big / 2, (size_t)4096 * MiB);
  char* res = p + big - 64 * MiB;
  for (int mode = 0; mode < 3; ++mode) {
    float ms_a = 0, ms_b = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      if (mode != 1) {
        CK(hipEventRecord(e0, sa));
        launch<false>(p, cold_bytes, sa, false);
        CK(hipEventRecord(e1, sa));
      }
      if (mode != 0) {
        CK(hipEventRecord(f0, sb));
        for (int r = 0; r < 64; ++r) launch<false>(res, 64 * MiB, sb, false);
        CK(hipEventRecord(f1, sb));
      }
      CK(hipDeviceSynchronize());
      if (mode != 1) CK(hipEventElapsedTime(&ms_a, e0, e1));
      if (mode != 0) CK(hipEventElapsedTime(&ms_b, f0, f1));
    }
    printf("%s: cold stream %.3f ms (%.0f GB/s)   resident loop %.3f ms (%.0f GB/s)\n",
           mode == 0 ? "A alone   " : mode == 1 ? "B alone   " : "A and B   ", ms_a, ms_a > 0 ? cold_bytes / ms_a / 1e6 : 0.0, ms_b,
           ms_b > 0 ? 64.0 * 64 * MiB / ms_b / 1e6 : 0.0);
  }
  return 0;
}
