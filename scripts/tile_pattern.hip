// What bounds the single-sweep kernel at mid column lengths?  The sweep's ingredients are added one at a
// time on the SAME load pattern (workgroup of WAVES waves, wave w streams row groups w*U .. w*U+U-1 of C adjacent columns per
// step, column groups strided over the grid, nontemporal 16-byte loads):
//   L   loads only (tile summed into one register: the minimum that keeps the loads alive)
//   LV  + the sweep's arithmetic per element (dot with a resident r, multiply-add into a resident accumulator) and the wave reduction
//   LVB + the per-step cross-wave exchange: one LDS store per column, workgroup barrier, WAVES LDS loads per column
//   (the real kernel adds the 2 scalar loads / 5 scalar stores per column and the epilogue arithmetic)
// each with one register tile (load, wait, consume) and with two (the next tile's loads are issued before the current is consumed).
//   hipcc -O3 --offload-arch=gfx950 scripts/tile_pattern.hip -o /tmp/tile_pattern && /tmp/tile_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, int C>
struct Tile {
  f4 v[C][U];
  __device__ __forceinline__ void load(const float* A, long ld, long cg, int wave, int lane) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float* p = A + (cg * C + c) * ld + lane * 4;
#pragma unroll
      for (int u = 0; u < U; ++u) v[c][u] = __builtin_nontemporal_load((const f4*)(p + (long)(wave * U + u) * 256));
    }
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int U, int C, int WAVES, int LEVEL, bool DB>
__global__ __launch_bounds__(WAVES * 64) void tile_stream(const float* __restrict__ A, long ld, long n, const float* __restrict__ r,
                                                           float* __restrict__ out) {
  __shared__ float sm[2][C][WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long ncg = n / C;
  f4 rk[U], racc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    rk[u] = *(const f4*)(r + (long)(wave * U + u) * 256 + lane * 4);
    racc[u] = f4{0, 0, 0, 0};
  }
  f4 sink = {0, 0, 0, 0};
  auto consume = [&](const Tile<U, C>& t, int buf) {
    if constexpr (LEVEL == 0) {
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int u = 0; u < U; ++u) sink += t.v[c][u];
    } else {
      float dot[C];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float d = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) d = fmaf(t.v[c][u][e], rk[u][e], d);
        dot[c] = wave_sum(d);
      }
      if constexpr (LEVEL >= 2) {
        if (lane == 0) {
#pragma unroll
          for (int c = 0; c < C; ++c) sm[buf][c][wave] = dot[c];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float g = sm[buf][c][0];
#pragma unroll
          for (int w = 1; w < WAVES; ++w) g += sm[buf][c][w];
          dot[c] = g;
        }
      }
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float vj = dot[c] * 1e-3f;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) racc[u][e] = fmaf(t.v[c][u][e], vj, racc[u][e]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) asm volatile("" : "+v"(racc[u]));
    }
  };
  const long cnt = ncg > (long)blockIdx.x ? (ncg - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  auto at = [&](long i) { return (long)blockIdx.x + i * (long)gridDim.x; };
  if constexpr (DB) {
    Tile<U, C> ta, tb;
    long i = 0;
    if (i < cnt) ta.load(A, ld, at(i), wave, lane);
    while (i < cnt) {
      if (i + 1 < cnt) tb.load(A, ld, at(i + 1), wave, lane);
      consume(ta, 0);
      if (i + 1 >= cnt) break;
      if (i + 2 < cnt) ta.load(A, ld, at(i + 2), wave, lane);
      consume(tb, 1);
      i += 2;
    }
  } else {
    Tile<U, C> t;
    int buf = 0;
    for (long i = 0; i < cnt; ++i) {
      t.load(A, ld, at(i), wave, lane);
      consume(t, buf);
      buf ^= 1;
    }
  }
  float s = sink.x + sink.y + sink.z + sink.w;
#pragma unroll
  for (int u = 0; u < U; ++u) s += racc[u].x + racc[u].y + racc[u].z + racc[u].w;
  if (s == 1.2345e-30f) out[0] = s;
}

static float *dA, *dr, *dout;

template <int U, int C, int WAVES, int LEVEL, bool DB>
static void run(long m, long n, int blocks, const char* tag) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto go = [&] { hipLaunchKernelGGL((tile_stream<U, C, WAVES, LEVEL, DB>), dim3(blocks), dim3(WAVES * 64), 0, 0, dA, m, n, dr, dout); };
  go();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(a));
    for (int k = 0; k < 4; ++k) go();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms / 4 < best) best = ms / 4;
  }
  printf("| %ld x %ld | U=%d C=%d waves=%d %s | %d | %s | %.1f | %.0f |\n", m, n, U, C, WAVES, DB ? "two tiles" : "one tile", blocks, tag,
         best * 1e3, (double)m * n * 4 / best / 1e6);
}

template <int U, int C, int WAVES, bool DB>
static void levels(long m, long n, int blocks) {
  run<U, C, WAVES, 0, DB>(m, n, blocks, "L");
  run<U, C, WAVES, 1, DB>(m, n, blocks, "LV");
  run<U, C, WAVES, 2, DB>(m, n, blocks, "LVB");
}

int main() {
  const long n = 262144, mmax = 16384;
  CK(hipMalloc(&dA, (size_t)mmax * n * 4));
  CK(hipMemset(dA, 0, (size_t)mmax * n * 4));
  CK(hipMalloc(&dr, mmax * 4));
  CK(hipMemset(dr, 0, mmax * 4));
  CK(hipMalloc(&dout, 4));
  printf("| shape | geometry | workgroups | level | us | GB/s |\n|---|---|---:|---|---:|---:|\n");
  // the headline geometry at 16384 rows
  levels<16, 2, 4, true>(16384, n, 256);
  levels<16, 2, 4, false>(16384, n, 256);
  levels<16, 1, 4, false>(16384, n, 512);
  // 8192 rows: what the dispatch uses (<4,8,8>, one tile) and its neighbours
  levels<4, 8, 8, false>(8192, n, 256);
  levels<4, 4, 8, true>(8192, n, 256);
  levels<4, 4, 8, false>(8192, n, 256);
  levels<8, 4, 4, true>(8192, n, 256);
  levels<8, 4, 4, false>(8192, n, 256);
  levels<8, 2, 4, true>(8192, n, 512);
  levels<8, 2, 4, false>(8192, n, 512);
  levels<8, 4, 4, false>(8192, n, 512);
  levels<4, 8, 8, false>(8192, n, 512);
  // 10240 rows (U = 10, four waves: the single-member team's shape)
  levels<10, 2, 4, true>(10240, n, 256);
  levels<10, 2, 4, false>(10240, n, 256);
  levels<10, 2, 4, false>(10240, n, 512);
  levels<10, 4, 4, false>(10240, n, 256);
  levels<5, 4, 8, false>(10240, n, 256);
  levels<5, 8, 8, false>(10240, n, 256);
  return 0;
}
