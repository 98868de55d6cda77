// Generic elementwise / reduction driver shared by pg_vec.hip and pg_lbfgs.hip.
#pragma once
#include "pg_internal.h"

namespace pgew {

inline unsigned grid_for(int64_t n_items, int num_cu, bool reduces) {
  int64_t blocks = (n_items + 255) / 256;
  // <= 2048 blocks on MI355X for pure streams, grid-stride the rest; kernels that end in the ticketed grid
  // reduction use <= 2 per CU: its single-counter fan-in costs ~12 ns per arriving workgroup
  const int64_t cap = (int64_t)num_cu * (reduces ? 2 : 8);
  if (blocks > cap) blocks = cap;
  if (blocks > PG_RED_MAX_BLOCKS) blocks = PG_RED_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Generic elementwise driver: F is a functor with
//   template<int N> __device__ void operator()(int64_t i0, /*lane-private*/ Acc&) processing N consecutive
// elements starting at i0 (N = VEC for the vector body, 1 for tails / unaligned operands).
template <typename T, typename F, int NS, unsigned MAXMASK>
__global__ __launch_bounds__(256) void ew_kernel(int64_t n, bool vec_ok, F f, double* __restrict__ red_partials,
                                                 unsigned* __restrict__ red_counter, double* __restrict__ out) {
  constexpr int VEC = VecOf<T>::N;
  double acc[NS > 0 ? NS : 1];
#pragma unroll
  for (int k = 0; k < (NS > 0 ? NS : 1); ++k) acc[k] = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * 256;
  if (vec_ok) {
    const int64_t nvec = n / VEC;
    for (int64_t v = tid; v < nvec; v += nthreads) f.template apply<VEC>(v * VEC, acc);
    for (int64_t i = nvec * VEC + tid; i < n; i += nthreads) f.template apply<1>(i, acc);
  } else {
    for (int64_t i = tid; i < n; i += nthreads) f.template apply<1>(i, acc);
  }
  if constexpr (NS > 0) {
    double ps[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) ps[k] = f.post_scale(k);
    grid_reduce_finalize<NS, MAXMASK>(acc, red_partials, red_counter, out, ps);
  }
}

template <typename T, int N>
struct Pack {
  T v[N];
};

template <typename T, int N>
__device__ __forceinline__ Pack<T, N> ld(const T* __restrict__ p, int64_t i) {
  Pack<T, N> r;
  if constexpr (N == 1) {
    r.v[0] = p[i];
  } else {
    using V = typename VecOf<T>::type;
    V t = *reinterpret_cast<const V*>(p + i);
#pragma unroll
    for (int e = 0; e < N; ++e) r.v[e] = t[e];
  }
  return r;
}

template <typename T, int N>
__device__ __forceinline__ void st(T* __restrict__ p, int64_t i, const Pack<T, N>& r) {
  if constexpr (N == 1) {
    p[i] = r.v[0];
  } else {
    using V = typename VecOf<T>::type;
    V t;
#pragma unroll
    for (int e = 0; e < N; ++e) t[e] = r.v[e];
    *reinterpret_cast<V*>(p + i) = t;
  }
}


template <typename T, typename F, int NS, unsigned MAXMASK>
pg_status launch_ew(pg_ctx* c, int64_t n, bool vec_ok, const F& f, double* out_dev) {
  if (n <= 0 && NS == 0) return PG_OK;
  const unsigned blocks = grid_for(n / VecOf<T>::N + 1, c->num_cu, NS > 0);
  hipLaunchKernelGGL((ew_kernel<T, F, NS, MAXMASK>), dim3(blocks), dim3(256), 0, c->stream, n, vec_ok, f,
                     c->red_partials, c->red_counter, out_dev);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

}  // namespace pgew
