// Generic elementwise / reduction driver shared by pg_vec.hip and pg_lbfgs.hip.
#pragma once
#include <cstdlib>

#include "pg_internal.h"

namespace pgew {

// Launch geometry.  Pure streams: 256-thread workgroups, <= 8 per CU (grid-stride the rest).  Kernels that end in
// the ticketed grid reduction: 1024-thread workgroups (16 waves keep ~48 KiB of loads in flight per CU), ONE per CU:
// the single-counter fan-in of the reduction costs ~12 ns per arriving workgroup, so 256 arrivals (~3 us), not 2048.
constexpr int EW_BS_STREAM = 256;
constexpr int EW_BS_REDUCE = 1024;

inline unsigned grid_for(int64_t n_items, int num_cu, bool reduces) {
  const int bs = reduces ? EW_BS_REDUCE : EW_BS_STREAM;
  int64_t blocks = (n_items + bs - 1) / bs;
  static const int reduce_per_cu = getenv("PG_EW_REDUCE_BLOCKS_PER_CU") ? atoi(getenv("PG_EW_REDUCE_BLOCKS_PER_CU")) : 1;
  const int64_t cap = (int64_t)num_cu * (reduces ? reduce_per_cu : 8);
  if (blocks > cap) blocks = cap;
  if (blocks > PG_RED_MAX_BLOCKS) blocks = PG_RED_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// hook run by thread 0 of the finalizing workgroup with the reduced values (overload per functor when needed)
template <typename F>
__device__ __forceinline__ void ew_on_final(const F&, const double*) {}

// Generic elementwise driver: F is a functor with
//   template<int N> __device__ void operator()(int64_t i0, /*lane-private*/ Acc&) processing N consecutive
// elements starting at i0 (N = VEC for the vector body, 1 for tails / unaligned operands).
template <typename T, typename F, int NS, unsigned MAXMASK, int BS, int UNR = 2>
__global__ __launch_bounds__(BS) void ew_kernel(int64_t n, bool vec_ok, F f, double* __restrict__ red_partials,
                                                 unsigned* __restrict__ red_counter, double* __restrict__ out) {
  constexpr int VEC = VecOf<T>::N;
  double acc[NS > 0 ? NS : 1];
#pragma unroll
  for (int k = 0; k < (NS > 0 ? NS : 1); ++k) acc[k] = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * BS + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * BS;
  if (vec_ok) {
    const int64_t nvec = n / VEC;
    int64_t v = tid;
    for (; v + (UNR - 1) * nthreads < nvec; v += UNR * nthreads) {  // UNR independent vectors per trip: more loads in flight
#pragma unroll
      for (int k = 0; k < UNR; ++k) f.template apply<VEC>((v + k * nthreads) * VEC, acc);
    }
    for (; v < nvec; v += nthreads) f.template apply<VEC>(v * VEC, acc);
    for (int64_t i = nvec * VEC + tid; i < n; i += nthreads) f.template apply<1>(i, acc);
  } else {
    for (int64_t i = tid; i < n; i += nthreads) f.template apply<1>(i, acc);
  }
  if constexpr (NS > 0) {
    double ps[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) ps[k] = f.post_scale(k);
    double fin[NS];
    const bool last = grid_reduce_finalize<NS, MAXMASK, BS / 64>(acc, red_partials, red_counter, out, ps, fin);
    if (last && threadIdx.x == 0) ew_on_final(f, fin);
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

// non-temporal store: for streams the kernel chain never reads back (keeps them from displacing re-read operands
// in L2 / Infinity Cache)
template <typename T, int N>
__device__ __forceinline__ void st_nt(T* __restrict__ p, int64_t i, const Pack<T, N>& r) {
  if constexpr (N == 1) {
    __builtin_nontemporal_store(r.v[0], p + i);
  } else {
    using V = typename VecOf<T>::type;
    V t;
#pragma unroll
    for (int e = 0; e < N; ++e) t[e] = r.v[e];
    __builtin_nontemporal_store(t, reinterpret_cast<V*>(p + i));
  }
}


// BSR: workgroup size of a reducing kernel (default EW_BS_REDUCE); UNR: independent vectors per loop trip
template <typename T, typename F, int NS, unsigned MAXMASK, int BSR = EW_BS_REDUCE, int UNR = 2>
pg_status launch_ew(pg_ctx* c, int64_t n, bool vec_ok, const F& f, double* out_dev, int blocks_per_cu = 0) {
  if (n <= 0 && NS == 0) return PG_OK;
  constexpr int BS = NS > 0 ? BSR : EW_BS_STREAM;
  unsigned blocks = grid_for(n / VecOf<T>::N + 1, c->num_cu, NS > 0);
  if (blocks_per_cu > 0) {
    int64_t b = (n / VecOf<T>::N + BS) / BS;
    if (b > (int64_t)c->num_cu * blocks_per_cu) b = (int64_t)c->num_cu * blocks_per_cu;
    if (b > PG_RED_MAX_BLOCKS) b = PG_RED_MAX_BLOCKS;
    blocks = (unsigned)(b < 1 ? 1 : b);
  }
  hipLaunchKernelGGL((ew_kernel<T, F, NS, MAXMASK, BS, UNR>), dim3(blocks), dim3(BS), 0, c->stream, n, vec_ok, f,
                     c->red_partials, c->red_counter, out_dev);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

}  // namespace pgew
