// L-BFGS operator on the device: src/accel/lbfgs.jl:5-95 (LBFGSOperator, update!, reset!, mul!).
//
// The two-loop recursion is 2M *dependent* reductions.  Each loop step is ONE kernel that applies the
// previous step's AXPY and accumulates the next step's dot product in the same sweep; the coefficient is
// produced on the device by the finalizing workgroup of the grid reduction and consumed by the next launch,
// so mul! needs 1 + 2*currmem launches and no host synchronisation.
#include "pg_ew.h"

struct pg_lbfgs {
  pg_ctx* ctx = nullptr;
  int dtype = PG_F32;
  int M = 0;
  int64_t n = 0;
  int currmem = 0, curridx = 0;  // 1-based index like the reference; 0 = empty
  void* slab = nullptr;          // s, y, s_M[M], y_M[M]
  void *s = nullptr, *y = nullptr;
  void** s_M = nullptr;
  void** y_M = nullptr;
  double* ys_M = nullptr;   // host copy (lbfgs.jl:11)
  double* dcoef = nullptr;  // device: alphas[M] ++ betas[M] ++ scratch[4]
  double H = 1.0;
  // images under a linear map A (pg_lbfgs_images_*): A s_i, A y_i of the stored pairs, so that A (H v) follows from A v
  // and the two-loop coefficients without reading A -- d = H0 (v - sum alpha_i y_i) + sum (alpha_i - beta_i) s_i
  int64_t img_m = 0;
  void* img_slab = nullptr;  // As_M[M], Ay_M[M]
  size_t img_vb = 0;
  bool last_update_accepted = false;
  bool img_valid[64] = {};  // slot holds the images of the pair currently stored in it (set by images_update, cleared when
                            // update overwrites the slot, when images are (re-)enabled and on reset)
  int last_k = 0;        // currmem of the last apply
  int last_idx[64];      // its loop order (newest -> oldest, 1-based slots)
  double last_H = 1.0;
};

namespace {
using namespace pgew;

// d_out = scale * (d_in + c * w), c read from device scalars; optionally acc[0] = <u, d_out> and the
// finalizer stores acc[0] / out_div.
//   mode 0: c = 0 (plain copy/scale)    mode 1: c = -coef[a_idx]    mode 2: c = coef[a_idx] - coef[b_idx]
template <typename T>
struct AxpyDotF {
  T* d_out;
  const T* d_in;
  const T* w;
  const T* u;
  const double* coef;
  int mode, a_idx, b_idx;
  T scale;
  double inv_div;
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double* acc) const {
    T c = T(0);
    if (mode == 1) c = -(T)coef[a_idx];
    if (mode == 2) c = (T)coef[a_idx] - (T)coef[b_idx];
    Pack<T, N> dv = ld<T, N>(d_in, i), o;
    if (mode != 0) {
      Pack<T, N> wv = ld<T, N>(w, i);
#pragma unroll
      for (int e = 0; e < N; ++e) o.v[e] = scale * (dv.v[e] + c * wv.v[e]);
    } else {
#pragma unroll
      for (int e = 0; e < N; ++e) o.v[e] = scale * dv.v[e];
    }
    if (d_out != nullptr) st<T, N>(d_out, i, o);
    if (u != nullptr) {
      Pack<T, N> uv = ld<T, N>(u, i);
#pragma unroll
      for (int e = 0; e < N; ++e) acc[0] += (double)uv.v[e] * (double)o.v[e];
    }
  }
  __device__ double post_scale(int) const { return inv_div; }
};

template <typename T>
struct Dot2F {  // acc = { <s,y>, <y,y> }
  const T* s;
  const T* y;
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double* acc) const {
    Pack<T, N> sv = ld<T, N>(s, i), yv = ld<T, N>(y, i);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      acc[0] += (double)sv.v[e] * (double)yv.v[e];
      acc[1] += (double)yv.v[e] * (double)yv.v[e];
    }
  }
  __device__ double post_scale(int) const { return 1.0; }
};

template <typename T>
pg_status axpy_dot(pg_lbfgs* L, void* d_out, const void* d_in, int mode, int a_idx, int b_idx, const void* w,
                   double scale, const void* u, int out_idx, double out_div) {
  AxpyDotF<T> f{(T*)d_out, (const T*)d_in, (const T*)w, (const T*)u, L->dcoef, mode, a_idx, b_idx, (T)scale,
                u ? 1.0 / out_div : 1.0};
  const bool v = aligned16(d_out) && aligned16(d_in) && (!w || aligned16(w)) && (!u || aligned16(u));
  if (u) return launch_ew<T, AxpyDotF<T>, 1, 0u>(L->ctx, L->n, v, f, L->dcoef + out_idx);
  return launch_ew<T, AxpyDotF<T>, 0, 0u>(L->ctx, L->n, v, f, L->dcoef);
}

template <typename T>
pg_status lbfgs_apply_t(pg_lbfgs* L, void* d, const void* v) {
  const int M = L->M, k = L->currmem;
  const int BETA = M;  // betas[M] follow alphas[M] in dcoef
  const double H = (double)(T)L->H;
  if (k == 0) {  // d .= v ; d .*= H
    L->last_k = 0;
    L->last_H = H;
    return axpy_dot<T>(L, d, v, 0, 0, 0, nullptr, H, nullptr, 0, 1.0);
  }
  // order of loop1 (newest -> oldest): idx_t, t = 0..k-1   lbfgs.jl:72-83
  int idx[64];
  {
    int id = L->curridx;
    for (int t = 0; t < k; ++t) {
      idx[t] = id;
      id -= 1;
      if (id == 0) id = M;
    }
  }
  // alphas[idx0] = <s_idx0, v> / ys_idx0          (d .= v is folded into the first AXPY)
  PG_TRY((axpy_dot<T>(L, nullptr, v, 0, 0, 0, nullptr, 1.0, L->s_M[idx[0] - 1], idx[0] - 1, L->ys_M[idx[0] - 1])));
  for (int t = 0; t < k; ++t) {
    const void* din = (t == 0) ? v : d;
    const int i = idx[t] - 1;
    if (t + 1 < k) {  // d -= alpha_i y_i ; alpha_next = <s_next, d> / ys_next
      const int nx = idx[t + 1] - 1;
      PG_TRY((axpy_dot<T>(L, d, din, 1, i, 0, L->y_M[i], 1.0, L->s_M[nx], nx, L->ys_M[nx])));
    } else {  // last of loop1: d -= alpha_i y_i ; d *= H (:67) ; beta = <y_i, d> / ys_i (first of loop2 :85-95)
      PG_TRY((axpy_dot<T>(L, d, din, 1, i, 0, L->y_M[i], H, L->y_M[i], BETA + i, L->ys_M[i])));
    }
  }
  for (int t = k - 1; t >= 0; --t) {  // loop2 oldest -> newest
    const int i = idx[t] - 1;
    // every step's beta has its own slot (betas[slot of the pair]): nothing is overwritten while a kernel still reads it,
    // and the whole coefficient set survives the recursion (pg_lbfgs_images_apply reuses it)
    if (t > 0) {  // d += (alpha_i - beta_i) s_i ; beta_next = <y_next, d> / ys_next
      const int nx = idx[t - 1] - 1;
      PG_TRY((axpy_dot<T>(L, d, d, 2, i, BETA + i, L->s_M[i], 1.0, L->y_M[nx], BETA + nx, L->ys_M[nx])));
    } else {
      PG_TRY((axpy_dot<T>(L, d, d, 2, i, BETA + i, L->s_M[i], 1.0, nullptr, 0, 1.0)));
    }
  }
  L->last_k = k;
  for (int t = 0; t < k; ++t) L->last_idx[t] = idx[t];
  L->last_H = H;
  return PG_OK;
}

// Ad[i] = H (Av[i] - sum_t alpha_t Ay_t[i]) + sum_t (alpha_t - beta_t) As_t[i] with the coefficients the last apply left
// on the device: the image of d = H v under the linear map the stored images belong to
struct ImgOrder {
  int k;
  int idx[64];  // 0-based slots, newest -> oldest
};

template <typename T>
__global__ __launch_bounds__(256) void lbfgs_image_kernel(T* __restrict__ Ad, const T* __restrict__ Av, const T* __restrict__ As,
                                                         const T* __restrict__ Ay, int64_t stride, int64_t m, int M,
                                                         const double* __restrict__ coef, T H, ImgOrder ord) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
    T acc = Av[i];
    for (int t = 0; t < ord.k; ++t) acc -= (T)coef[ord.idx[t]] * Ay[(int64_t)ord.idx[t] * stride + i];
    acc *= H;
    for (int t = ord.k - 1; t >= 0; --t)
      acc += ((T)coef[ord.idx[t]] - (T)coef[M + ord.idx[t]]) * As[(int64_t)ord.idx[t] * stride + i];
    Ad[i] = acc;
  }
}

template <typename T>
pg_status lbfgs_images_apply_t(pg_lbfgs* L, void* Ad, const void* Av) {
  ImgOrder ord;
  ord.k = L->last_k;
  for (int t = 0; t < ord.k; ++t) ord.idx[t] = L->last_idx[t] - 1;
  const int64_t stride = (int64_t)(L->img_vb / sizeof(T));
  int64_t blocks = (L->img_m + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(lbfgs_image_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, L->ctx->stream, (T*)Ad, (const T*)Av,
                     (const T*)L->img_slab, (const T*)((char*)L->img_slab + (size_t)L->M * L->img_vb), stride, L->img_m, L->M,
                     (const double*)L->dcoef, (T)L->last_H, ord);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

template <typename T>
pg_status lbfgs_update_t(pg_lbfgs* L, const void* s, const void* y) {
  pg_ctx* c = L->ctx;
  const size_t nb = (size_t)L->n * sizeof(T);
  if (L->n > 0) {  // L.s .= s ; L.y .= y      lbfgs.jl:31-32
    PG_HIP(hipMemcpyAsync(L->s, s, nb, hipMemcpyDeviceToDevice, c->stream));
    PG_HIP(hipMemcpyAsync(L->y, y, nb, hipMemcpyDeviceToDevice, c->stream));
  }
  Dot2F<T> f{(const T*)L->s, (const T*)L->y};
  PG_TRY((launch_ew<T, Dot2F<T>, 2, 0u>(c, L->n, true, f, c->dscal + PG_S_MISC)));
  PG_TRY(pg_read_scalars(c, PG_S_MISC, 2));
  const T ys = (T)c->hscal[PG_S_MISC], yty = (T)c->hscal[PG_S_MISC + 1];
  L->last_update_accepted = ys > T(0);
  if (ys > T(0)) {  // :34-49
    L->curridx += 1;
    if (L->curridx > L->M) L->curridx = 1;
    L->currmem += 1;
    if (L->currmem > L->M) L->currmem = L->M;
    L->ys_M[L->curridx - 1] = (double)ys;
    L->img_valid[L->curridx - 1] = false;  // the slot's images (if any) belonged to the pair just overwritten
    if (L->n > 0) {
      PG_HIP(hipMemcpyAsync(L->s_M[L->curridx - 1], L->s, nb, hipMemcpyDeviceToDevice, c->stream));
      PG_HIP(hipMemcpyAsync(L->y_M[L->curridx - 1], L->y, nb, hipMemcpyDeviceToDevice, c->stream));
    }
    L->H = (double)(ys / yty);
  }
  return PG_OK;
}

}  // namespace

extern "C" {

pg_status pg_lbfgs_create(pg_ctx* c, int32_t dtype, int32_t M, int64_t n, pg_lbfgs** out) {
  PG_REQUIRE(c != nullptr && out != nullptr, "null argument");
  PG_REQUIRE(dtype == PG_F32 || dtype == PG_F64, "bad dtype");
  PG_REQUIRE(M >= 1 && M <= 64, "memory M must be in 1..64");
  PG_REQUIRE(n >= 0, "negative length");
  *out = nullptr;
  pg_lbfgs* L = new pg_lbfgs();
  L->ctx = c;
  L->dtype = dtype;
  L->M = M;
  L->n = n;
  const size_t vb = (size_t)pg_round_up((int64_t)((size_t)(n > 0 ? n : 1) * pg_sizeof(dtype)), 256);
  PG_HIP(hipSetDevice(c->device));
  if (hipMalloc(&L->slab, vb * (2 + 2 * (size_t)M)) != hipSuccess ||
      hipMalloc((void**)&L->dcoef, sizeof(double) * (2 * M + 4)) != hipSuccess) {
    pg_set_error("L-BFGS memory allocation failed (M=%d, n=%lld)", M, (long long)n);
    pg_lbfgs_destroy(L);
    return PG_ERR_ALLOC;
  }
  PG_HIP(hipMemsetAsync(L->slab, 0, vb * (2 + 2 * (size_t)M), c->stream));
  PG_HIP(hipMemsetAsync(L->dcoef, 0, sizeof(double) * (2 * M + 4), c->stream));
  char* base = (char*)L->slab;
  L->s = base;
  L->y = base + vb;
  L->s_M = new void*[M];
  L->y_M = new void*[M];
  L->ys_M = new double[M]();
  for (int i = 0; i < M; ++i) {
    L->s_M[i] = base + vb * (2 + i);
    L->y_M[i] = base + vb * (2 + M + i);
  }
  *out = L;
  return PG_OK;
}

pg_status pg_lbfgs_destroy(pg_lbfgs* L) {
  if (!L) return PG_OK;
  if (L->slab || L->dcoef) (void)hipStreamSynchronize(L->ctx->stream);
  if (L->slab) (void)hipFree(L->slab);
  if (L->dcoef) (void)hipFree(L->dcoef);
  if (L->img_slab) (void)hipFree(L->img_slab);
  delete[] L->s_M;
  delete[] L->y_M;
  delete[] L->ys_M;
  delete L;
  return PG_OK;
}

pg_status pg_lbfgs_update(pg_lbfgs* L, const void* s, const void* y) {
  PG_REQUIRE(L != nullptr, "operator is null");
  PG_REQUIRE(L->n == 0 || (s != nullptr && y != nullptr), "null vector");
  return L->dtype == PG_F32 ? lbfgs_update_t<float>(L, s, y) : lbfgs_update_t<double>(L, s, y);
}

pg_status pg_lbfgs_reset(pg_lbfgs* L) {  // lbfgs.jl:52-55
  PG_REQUIRE(L != nullptr, "operator is null");
  L->currmem = 0;
  L->curridx = 0;
  L->H = 1.0;
  L->last_k = 0;
  L->last_H = 1.0;
  for (bool& v : L->img_valid) v = false;
  return PG_OK;
}

pg_status pg_lbfgs_images_enable(pg_lbfgs* L, int64_t m) {
  PG_REQUIRE(L != nullptr && m >= 0, "bad argument");
  if (L->img_slab != nullptr && L->img_m == m) return PG_OK;
  if (L->img_slab) {
    (void)hipStreamSynchronize(L->ctx->stream);
    (void)hipFree(L->img_slab);
    L->img_slab = nullptr;
  }
  L->img_m = m;
  L->img_vb = (size_t)pg_round_up((int64_t)((size_t)(m > 0 ? m : 1) * pg_sizeof(L->dtype)), 256);
  if (hipMalloc(&L->img_slab, L->img_vb * 2 * (size_t)L->M) != hipSuccess) {
    pg_set_error("L-BFGS image allocation failed (M=%d, m=%lld)", L->M, (long long)m);
    return PG_ERR_ALLOC;
  }
  PG_HIP(hipMemsetAsync(L->img_slab, 0, L->img_vb * 2 * (size_t)L->M, L->ctx->stream));
  for (bool& v : L->img_valid) v = false;  // pairs already stored have no image: images_apply refuses them
  return PG_OK;
}

pg_status pg_lbfgs_images_update(pg_lbfgs* L, const void* As, const void* Ay) {
  PG_REQUIRE(L != nullptr && L->img_slab != nullptr, "images are not enabled");
  PG_REQUIRE(L->img_m == 0 || (As != nullptr && Ay != nullptr), "null vector");
  if (!L->last_update_accepted || L->curridx == 0) return PG_OK;  // the pair was not stored (<s, y> <= 0)
  if (L->img_m == 0) {
    L->img_valid[L->curridx - 1] = true;
    return PG_OK;
  }
  const size_t nb = (size_t)L->img_m * pg_sizeof(L->dtype);
  char* base = (char*)L->img_slab;
  PG_HIP(hipMemcpyAsync(base + (size_t)(L->curridx - 1) * L->img_vb, As, nb, hipMemcpyDeviceToDevice, L->ctx->stream));
  PG_HIP(hipMemcpyAsync(base + ((size_t)L->M + (size_t)(L->curridx - 1)) * L->img_vb, Ay, nb, hipMemcpyDeviceToDevice,
                        L->ctx->stream));
  L->img_valid[L->curridx - 1] = true;
  return PG_OK;
}

pg_status pg_lbfgs_images_apply(pg_lbfgs* L, void* Ad, const void* Av) {
  PG_REQUIRE(L != nullptr && L->img_slab != nullptr, "images are not enabled");
  PG_REQUIRE(L->img_m == 0 || (Ad != nullptr && Av != nullptr), "null vector");
  if (L->img_m == 0) return PG_OK;
  // every pair the last apply used needs the image of the pair it holds NOW: enabled after pairs were stored, or an
  // accepted update without its images_update, would otherwise combine stale / zero images into a wrong A d silently
  for (int t = 0; t < L->last_k; ++t) {
    const int slot = L->last_idx[t] - 1;
    if (slot < 0 || slot >= L->M || !L->img_valid[slot]) {
      pg_set_error("L-BFGS images: slot %d used by the last apply has no current image (call pg_lbfgs_images_update after every "
                   "accepted pg_lbfgs_update; enable images on an empty operator)", slot + 1);
      return PG_ERR_INVALID;
    }
  }
  return L->dtype == PG_F32 ? lbfgs_images_apply_t<float>(L, Ad, Av) : lbfgs_images_apply_t<double>(L, Ad, Av);
}

pg_status pg_lbfgs_images_ready(pg_lbfgs* L, int32_t* ready_out) {
  PG_REQUIRE(L != nullptr && ready_out != nullptr, "null argument");
  *ready_out = 0;
  if (L->img_slab == nullptr) return PG_OK;  // images were never enabled
  int id = L->curridx;  // the slots the next apply walks (lbfgs.jl:72-83): newest -> oldest
  for (int t = 0; t < L->currmem; ++t) {
    if (id < 1 || id > L->M || !L->img_valid[id - 1]) return PG_OK;
    id -= 1;
    if (id == 0) id = L->M;
  }
  *ready_out = 1;
  return PG_OK;
}

pg_status pg_lbfgs_apply(pg_lbfgs* L, void* d, const void* v) {
  PG_REQUIRE(L != nullptr, "operator is null");
  PG_REQUIRE(L->n == 0 || (d != nullptr && v != nullptr), "null vector");
  if (L->n == 0) return PG_OK;
  return L->dtype == PG_F32 ? lbfgs_apply_t<float>(L, d, v) : lbfgs_apply_t<double>(L, d, v);
}

}  // extern "C"
