// Elementwise / BLAS-1 kernels over n-vectors: the broadcasts, prox operators and reductions of the
// forward-backward iteration body (forward_backward.jl:117-120, fast_forward_backward.jl:135-142,
// fb_tools.jl:3-5,48-50) and the fused forward-backward epilogue.
//
// All are HBM-bound streams: 16-byte-per-lane loads/stores when every pointer is 16-byte aligned
// (always true for library- and torch-allocated vectors), scalar tail; reductions accumulate per thread in
// fp64 and finish with the deterministic grid reduction of pg_internal.h.
#include "pg_ew.h"

namespace {

using namespace pgew;

// ---- functors ---------------------------------------------------------------------------------
template <typename T>
struct AxpbyF {  // out = a x + b y   (y nullable)
  T* out;
  const T* x;
  const T* y;
  T a, b;
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double*) const {
    Pack<T, N> xv = ld<T, N>(x, i), o;
    if (y != nullptr) {
      Pack<T, N> yv = ld<T, N>(y, i);
#pragma unroll
      for (int e = 0; e < N; ++e) o.v[e] = a * xv.v[e] + b * yv.v[e];
    } else {
#pragma unroll
      for (int e = 0; e < N; ++e) o.v[e] = a * xv.v[e];
    }
    st<T, N>(out, i, o);
  }
  __device__ double post_scale(int) const { return 1.0; }
};

template <typename T>
struct AddScalarF {  // out = x + c  (x nullable: fill)
  T* out;
  const T* x;
  T c;
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double*) const {
    Pack<T, N> o;
    if (x != nullptr) {
      Pack<T, N> xv = ld<T, N>(x, i);
#pragma unroll
      for (int e = 0; e < N; ++e) o.v[e] = xv.v[e] + c;
    } else {
#pragma unroll
      for (int e = 0; e < N; ++e) o.v[e] = c;
    }
    st<T, N>(out, i, o);
  }
  __device__ double post_scale(int) const { return 1.0; }
};

template <typename T>
struct ExtrapolateF {  // x = z + beta (z - z_prev)      fast_forward_backward.jl:135
  T* x;
  const T* z;
  const T* zp;
  T beta;
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double*) const {
    Pack<T, N> zv = ld<T, N>(z, i), pv = ld<T, N>(zp, i), o;
#pragma unroll
    for (int e = 0; e < N; ++e) o.v[e] = zv.v[e] + beta * (zv.v[e] - pv.v[e]);
    st<T, N>(x, i, o);
  }
  __device__ double post_scale(int) const { return 1.0; }
};

template <typename T>
struct DotF {  // acc[0] = sum x y
  const T* x;
  const T* y;
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double* acc) const {
    Pack<T, N> xv = ld<T, N>(x, i), yv = ld<T, N>(y, i);
#pragma unroll
    for (int e = 0; e < N; ++e) acc[0] += (double)xv.v[e] * (double)yv.v[e];
  }
  __device__ double post_scale(int) const { return 1.0; }
};

template <typename T>
struct NrmInfF {  // acc[0] = max |x|
  const T* x;
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double* acc) const {
    Pack<T, N> xv = ld<T, N>(x, i);
#pragma unroll
    for (int e = 0; e < N; ++e) acc[0] = pg_maxn(acc[0], fabs((double)xv.v[e]));
  }
  __device__ double post_scale(int) const { return 1.0; }
};

template <typename T>
struct Norm1F {  // acc[0] = lam * sum |x|
  const T* x;
  double lam;
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double* acc) const {
    Pack<T, N> xv = ld<T, N>(x, i);
#pragma unroll
    for (int e = 0; e < N; ++e) acc[0] += fabs((double)xv.v[e]);
  }
  __device__ double post_scale(int) const { return lam; }
};

// soft threshold exactly as ProximalOperators' NormL1 prox!:
//   y = x + (x <= -gl ? gl : (x >= gl ? -gl : -x))
template <typename T>
__device__ __forceinline__ T soft_threshold(T x, T gl) {
  return x <= -gl ? x + gl : (x >= gl ? x - gl : T(0));
}

template <typename T>
struct ProxL1F {  // y = soft(x, gamma lam); acc[0] = lam sum |y|
  T* y;
  const T* x;
  T gl;
  double lam;
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double* acc) const {
    Pack<T, N> xv = ld<T, N>(x, i), o;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      o.v[e] = soft_threshold(xv.v[e], gl);
      acc[0] += fabs((double)o.v[e]);
    }
    st<T, N>(y, i, o);
  }
  __device__ double post_scale(int) const { return lam; }
};

// NormL1 with per-element weights (ProximalOperators.NormL1(lambda::AbstractArray)): g(x) = sum_i lam_i |x_i|,
// prox: y_i = soft(x_i, gamma lam_i)
template <typename T>
struct Norm1WF {
  const T* x;
  const T* lam;
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double* acc) const {
    Pack<T, N> xv = ld<T, N>(x, i), lv = ld<T, N>(lam, i);
#pragma unroll
    for (int e = 0; e < N; ++e) acc[0] += (double)lv.v[e] * fabs((double)xv.v[e]);
  }
  __device__ double post_scale(int) const { return 1.0; }
};

template <typename T>
struct ProxL1WF {
  T* y;
  const T* x;
  const T* lam;
  T gamma;
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double* acc) const {
    Pack<T, N> xv = ld<T, N>(x, i), lv = ld<T, N>(lam, i), o;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      o.v[e] = soft_threshold(xv.v[e], pg_l1w_threshold(gamma, lv.v[e]));
      acc[0] += (double)lv.v[e] * fabs((double)o.v[e]);
    }
    st<T, N>(y, i, o);
  }
  __device__ double post_scale(int) const { return 1.0; }
};

template <typename T>
struct ProxBoxF {  // y = min(hi, max(lo, x))   (vector bounds optional)
  T* y;
  const T* x;
  T lo, hi;
  const T* lov;
  const T* hiv;
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double*) const {
    Pack<T, N> xv = ld<T, N>(x, i), o;
    Pack<T, N> l, h;
    if (lov != nullptr) l = ld<T, N>(lov, i);
    if (hiv != nullptr) h = ld<T, N>(hiv, i);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const T le = lov != nullptr ? l.v[e] : lo;
      const T he = hiv != nullptr ? h.v[e] : hi;
      o.v[e] = fmin(he, fmax(le, xv.v[e]));
    }
    st<T, N>(y, i, o);
  }
  __device__ double post_scale(int) const { return 1.0; }
};

// fused forward-backward epilogue:
//   y = x - gamma grad ; z = prox_{gamma g}(y) ; res = x - z
//   acc = { g(z)/lam = sum|z| , max|res| , sum grad*res , sum res^2 }
template <typename T, int GKIND>
struct EpilogueF {
  const T* x;
  const T* grad;
  T* y;
  T* z;
  T* res;
  T* grad_copy;  // nullable: also materialise grad here (sharded path keeps it in the all-reduce buffer)
  T gamma, p0, p1;  // p0 = gamma*lam (NormL1) | lo (IndBox) ; p1 = hi
  double gscale;    // lam for NormL1 (1 with per-element weights) else 0
  const T* p0v = nullptr;  // IndBox: per-element bounds; NormL1: per-element weights lam_i (nullptr: the scalars)
  const T* p1v = nullptr;
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double* acc) const {
    Pack<T, N> xv = ld<T, N>(x, i), gv = ld<T, N>(grad, i), yv, zv, rv, lov, hiv;
    if constexpr (GKIND == PG_G_INDBOX) {
      if (p0v != nullptr) lov = ld<T, N>(p0v, i), hiv = ld<T, N>(p1v, i);
    }
    if constexpr (GKIND == PG_G_NORML1) {
      if (p0v != nullptr) lov = ld<T, N>(p0v, i);
    }
#pragma unroll
    for (int e = 0; e < N; ++e) {
      yv.v[e] = xv.v[e] - gamma * gv.v[e];
      if constexpr (GKIND == PG_G_NORML1)
        zv.v[e] = soft_threshold(yv.v[e], p0v != nullptr ? pg_l1w_threshold(gamma, lov.v[e]) : p0);
      else if constexpr (GKIND == PG_G_INDBOX)
        zv.v[e] = p0v != nullptr ? fmin(hiv.v[e], fmax(lov.v[e], yv.v[e])) : fmin(p1, fmax(p0, yv.v[e]));
      else
        zv.v[e] = yv.v[e];
      rv.v[e] = xv.v[e] - zv.v[e];
      if constexpr (GKIND == PG_G_NORML1)
        acc[0] += p0v != nullptr ? (double)lov.v[e] * fabs((double)zv.v[e]) : fabs((double)zv.v[e]);
      acc[1] = pg_maxn(acc[1], fabs((double)rv.v[e]));
      acc[2] += (double)gv.v[e] * (double)rv.v[e];
      acc[3] += (double)rv.v[e] * (double)rv.v[e];
    }
    st<T, N>(y, i, yv);
    st<T, N>(z, i, zv);
    st<T, N>(res, i, rv);
    if (grad_copy != nullptr) st<T, N>(grad_copy, i, gv);
  }
  __device__ double post_scale(int k) const { return k == 0 ? gscale : 1.0; }
};

// separable quadratic f(x) = sum_i d_i x_i^2 / 2 + q_i x_i  (ProximalOperators: Tilt(SqrNormL2(d), q) /
// Quadratic(Diagonal(d), q)); prox_{gamma f}(x)_i = (x_i - gamma q_i) / (1 + gamma d_i)
template <typename T>
struct SepQuadParams {
  const T* __restrict__ dv;  // nullable -> scalar ds
  const T* __restrict__ qv;  // nullable -> scalar qs
  T ds, qs;
};

// prox of the separable quadratic, (x - gamma q) / (1 + gamma d), with every product rounded (no FMA contraction) and
// the quotient CORRECTLY ROUNDED, i.e. the bits of the IEEE division an unfused CPU evaluation performs -- but the divider
// is paid once per element, not once per application: inv = RN(1 / den) is formed with one IEEE division, and a quotient
// a / den is then  q0 = RN(a inv) followed by two Markstein corrections  q <- RN(q + RN(a - den q) inv)  (the residual is
// exact in an FMA).  After the first correction q is within half an ulp + 2 u^2 of a / den, which is the precondition under
// which the second one returns RN(a / den) (Markstein 1990; Cornea, Harrison & Tang, "Scientific computing on Itanium-based
// systems", thm. 8.3), barring underflow of the residual (|a| below 2^(emin + p)).  Five multiply-adds against the ~25
// issue slots of v_div_scale / v_rcp / v_div_fmas / v_div_fixup: this is what took the K-iterations-per-sweep
// Douglas-Rachford kernel off the divider (profiles/r2_bench_dr.json.log).  The single-step kernel and the operator-level
// prox use the same function, so all three produce the same bits.
template <typename T>
struct SepQuadElem {
  T gq, den, inv;
};
template <typename T>
__device__ __forceinline__ SepQuadElem<T> sepquad_prepare(T gamma, T de, T qe) {
#pragma clang fp contract(off)
  SepQuadElem<T> p;
  p.gq = gamma * qe;
  p.den = T(1) + gamma * de;
  p.inv = T(1) / p.den;
  return p;
}
template <typename T>
__device__ __forceinline__ T sepquad_apply(const SepQuadElem<T>& p, T x) {
#pragma clang fp contract(off)
  const T a = x - p.gq;
  T q = a * p.inv;
  q = fma(fma(-p.den, q, a), p.inv, q);
  q = fma(fma(-p.den, q, a), p.inv, q);
  return q;
}
template <typename T>
__device__ __forceinline__ T sepquad_prox_elem(T x, T gamma, T de, T qe) {
  return sepquad_apply(sepquad_prepare(gamma, de, qe), x);
}

template <typename T, int N>
__device__ __forceinline__ void sepquad_prox(const SepQuadParams<T>& f, T gamma, int64_t i, const Pack<T, N>& x,
                                             Pack<T, N>& y, double* fval) {
  Pack<T, N> d, q;
  if (f.dv != nullptr) d = ld<T, N>(f.dv, i);
  if (f.qv != nullptr) q = ld<T, N>(f.qv, i);
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const T de = f.dv != nullptr ? d.v[e] : f.ds;
    const T qe = f.qv != nullptr ? q.v[e] : f.qs;
    y.v[e] = sepquad_prox_elem(x.v[e], gamma, de, qe);
    if (fval != nullptr) *fval += 0.5 * (double)de * (double)y.v[e] * (double)y.v[e] + (double)qe * (double)y.v[e];
  }
}

template <typename T>
struct ProxSepQuadF {  // y = prox_{gamma f}(x); acc[0] = f(y)
  T* y;
  const T* x;
  SepQuadParams<T> f;
  T gamma;
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double* acc) const {
    Pack<T, N> xv = ld<T, N>(x, i), yv;
    sepquad_prox<T, N>(f, gamma, i, xv, yv, &acc[0]);
    st<T, N>(y, i, yv);
  }
  __device__ double post_scale(int) const { return 1.0; }
};

// One Douglas-Rachford iteration (douglas_rachford.jl:53-63), f separable quadratic, g by kind:
//   y = prox_{gamma f}(x) ; r = 2y - x ; z = prox_{gamma g}(r) ; res = y - z ; x -= res
//   acc = { max|res| , f(y) , g(z)/lam }
// y is always written (it is the solution, :70); r / z / res only when the caller wants the full state.
template <typename T, int GKIND>
struct DRStepF {
  T* x;  // x_next (may be the vector x is read from: the in-place step of douglas_rachford.jl:62)
  T* __restrict__ y;
  T* __restrict__ r;    // nullable
  T* __restrict__ z;    // nullable
  T* __restrict__ res;  // nullable
  SepQuadParams<T> f;
  T gamma, p0, p1;  // g: p0 = gamma*lam (NormL1) | lo (IndBox) ; p1 = hi
  double gscale;
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double* acc) const {
    Pack<T, N> xv = ld<T, N>(x_in != nullptr ? x_in : x, i), yv, rv, zv, sv;
    sepquad_prox<T, N>(f, gamma, i, xv, yv, &acc[1]);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      rv.v[e] = T(2) * yv.v[e] - xv.v[e];
      if constexpr (GKIND == PG_G_NORML1)
        zv.v[e] = soft_threshold(rv.v[e], p0);
      else if constexpr (GKIND == PG_G_INDBOX)
        zv.v[e] = fmin(p1, fmax(p0, rv.v[e]));
      else
        zv.v[e] = rv.v[e];
      sv.v[e] = yv.v[e] - zv.v[e];
      xv.v[e] = xv.v[e] - sv.v[e];
      acc[0] = pg_maxn(acc[0], fabs((double)sv.v[e]));
      if constexpr (GKIND == PG_G_NORML1) acc[2] += fabs((double)zv.v[e]);
    }
    st<T, N>(x, i, xv);  // re-read by the next iteration: regular store
    if (nt_out) {
      st_nt<T, N>(y, i, yv);
      if (r != nullptr) st_nt<T, N>(r, i, rv);
      if (z != nullptr) st_nt<T, N>(z, i, zv);
      if (res != nullptr) st_nt<T, N>(res, i, sv);
    } else {
      st<T, N>(y, i, yv);
      if (r != nullptr) st<T, N>(r, i, rv);
      if (z != nullptr) st<T, N>(z, i, zv);
      if (res != nullptr) st<T, N>(res, i, sv);
    }
  }
  __device__ double post_scale(int k) const { return k == 2 ? gscale : 1.0; }
  bool nt_out = true;  // non-temporal stores for the streams the next iteration does not read (experiments: PG_DR_NT=0)
  const T* x_in = nullptr;  // out-of-place step (pg_dr_step_async): x is read here and the new x written to `x`
};

// K Douglas-Rachford iterations per HBM sweep (temporal blocking): f and g are separable, so an element's K updates
// depend on nothing but the element -- x, d, q are read once, the iterates stay in registers, and only the state of
// the LAST iteration (x, y and optionally r, z, res) is written.  The stop rule of every one of the K iterations is
// still evaluated: acc[j] = max|res| of inner iteration j (the host replays from x_in when an inner iteration other
// than the last one satisfies it, so the returned state is exactly the one the step-by-step loop stops at).
//   acc = { max|res|_0 .. max|res|_{K-1} , f(y_K) , g(z_K)/lam }
template <typename T, int GKIND, int K>
struct DRBlockArgs {
  const T* __restrict__ x_in;
  T* __restrict__ x_out;
  T* __restrict__ y;
  T* __restrict__ r;    // nullable
  T* __restrict__ z;    // nullable
  T* __restrict__ res;  // nullable
  SepQuadParams<T> f;
  T gamma, p0, p1;
  double gscale;
};

// two elements side by side: on gfx950 the multiply-adds of a Float32 pair issue as ONE v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 (Float64 pairs are two instructions; same source)
template <typename T>
struct Pair2 {
  typedef T type __attribute__((ext_vector_type(2)));
};
__device__ __forceinline__ float pg_med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
__device__ __forceinline__ double pg_med3(double x, double lo, double hi) { return fmin(hi, fmax(lo, x)); }

template <typename T, int N>
struct DRBlockIn {
  Pack<T, N> x, d, q;
};
// BOTH: d and q are known to be vectors -- three unconditional loads (no branch for the compiler to trip over)
template <typename T, int GKIND, int K, int N, bool BOTH = false>
__device__ __forceinline__ DRBlockIn<T, N> dr_block_load(const DRBlockArgs<T, GKIND, K>& a, int64_t i) {
  DRBlockIn<T, N> in;
  in.x = ld<T, N>(a.x_in, i);
  if (BOTH || a.f.dv != nullptr) in.d = ld<T, N>(a.f.dv, i);
  if (BOTH || a.f.qv != nullptr) in.q = ld<T, N>(a.f.qv, i);
  return in;
}

// The K running maxima of a thread: one working-precision register per inner iteration.
template <typename T, int K>
struct DRMaxRegs {
  T mx[K];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < K; ++j) mx[j] = T(0);
  }
  __device__ __forceinline__ void fold(int j, T m) {
    T v = fmax(mx[j], m);
    // the empty asm pins the maximum HERE: left alone the compiler sinks the K max operations to the end of the sweep
    // (their only use) and keeps every iteration's residuals alive for them -- 4 registers per iteration, 209 at K = 32
    asm volatile("" : "+v"(v));
    mx[j] = v;
  }
  __device__ __forceinline__ void fold2(int j, T a, T b) {  // max(mx, |a|, |b|): one v_max3 with abs modifiers
    T v = fmax(fmax(mx[j], fabs(a)), fabs(b));
    asm volatile("" : "+v"(v));
    mx[j] = v;
  }
  __device__ __forceinline__ T get(int j) const { return mx[j]; }
};

template <typename T, int GKIND, int K, int N, typename MX>
__device__ __forceinline__ void dr_block_compute(const DRBlockArgs<T, GKIND, K>& a, int64_t i, const DRBlockIn<T, N>& in,
                                                 MX& mx, double& fy, double& gz) {
#pragma clang fp contract(off)
  Pack<T, N> xv = in.x, yv, rv, zv, sv, d = in.d, q = in.q;
  if constexpr (N >= 2) {
    using P = typename Pair2<T>::type;
    constexpr int NP = N / 2;
    const P two = {T(2), T(2)}, gam = {a.gamma, a.gamma};
    // sepquad_prepare / sepquad_apply on pairs: same operations, same rounding, per lane.  The NP pairs of a vector advance
    // in lockstep (pair loop INSIDE the iteration loop): consecutive instructions then belong to different dependency
    // chains, which is what the packed-Float32 pipeline wants (a dependent v_pk_* needs a wait state; written pair by
    // pair the compiler filled those with s_nop, 276 of them against 358 arithmetic instructions).
    // Float32 quotient a / den = RN32(RN64(a * RN64(1 / den))): the double product is within 2^-52 of a / den, and the
    // quotient of two 24-bit significands is never closer than 2^-49 (relative) to a rounding boundary of the 24-bit
    // format, so the final rounding lands on RN32(a / den) -- the bits of the IEEE division, for three full-rate
    // instructions (convert, multiply, convert) instead of the five of the Markstein sequence.  Float64 keeps Markstein.
    constexpr bool VIA64 = false;  // measured: 67.3 us per 16 iterations against 65.1 us for the Markstein form (the conversions are not full rate)
    P de[NP], qe[NP], gq[NP], inv[NP], nden[NP], xe[NP], ye[NP], re[NP], ze[NP], se[NP];
    double inv64[NP][2];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      de[p] = (P){a.f.dv != nullptr ? d.v[2 * p] : a.f.ds, a.f.dv != nullptr ? d.v[2 * p + 1] : a.f.ds};
      qe[p] = (P){a.f.qv != nullptr ? q.v[2 * p] : a.f.qs, a.f.qv != nullptr ? q.v[2 * p + 1] : a.f.qs};
      gq[p] = gam * qe[p];
      const P den = (P){T(1), T(1)} + gam * de[p];
      if constexpr (VIA64) {
        inv64[p][0] = 1.0 / (double)den[0], inv64[p][1] = 1.0 / (double)den[1];  // one division per element per sweep
      } else {
        inv[p] = (P){T(1) / den[0], T(1) / den[1]};
        nden[p] = -den;
      }
      xe[p] = (P){xv.v[2 * p], xv.v[2 * p + 1]};
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
      P aa[NP], qq[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) aa[p] = xe[p] - gq[p];
      if constexpr (VIA64) {
#pragma unroll
        for (int p = 0; p < NP; ++p) qq[p] = (P){(T)((double)aa[p][0] * inv64[p][0]), (T)((double)aa[p][1] * inv64[p][1])};
      } else {
        // The scheduling fences keep the pairs in lockstep in the EMITTED code too: in the prefetching loop the scheduler
        // otherwise runs one pair's whole chain after the other's and pads every dependent v_pk_* with s_nop (four per
        // iteration, ~8 % of the loop's issue cycles)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < NP; ++p) qq[p] = aa[p] * inv[p];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < 2; ++it) {  // two Markstein corrections
          P rr[NP];
#pragma unroll
          for (int p = 0; p < NP; ++p) rr[p] = __builtin_elementwise_fma(nden[p], qq[p], aa[p]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int p = 0; p < NP; ++p) qq[p] = __builtin_elementwise_fma(rr[p], inv[p], qq[p]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        ye[p] = qq[p];
        re[p] = __builtin_elementwise_fma(two, ye[p], -xe[p]);  // 2 y - x: the product is exact, so this is the unfused value
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int l = 0; l < 2; ++l) {
          if constexpr (GKIND == PG_G_NORML1)
            ze[p][l] = soft_threshold(re[p][l], a.p0);
          else if constexpr (GKIND == PG_G_INDBOX)
            ze[p][l] = pg_med3(re[p][l], a.p0, a.p1);  // = min(hi, max(lo, .)) for lo <= hi (checked by the entry point)
          else
            ze[p][l] = re[p][l];
        }
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) se[p] = ye[p] - ze[p];
#pragma unroll
      for (int p = 0; p < NP; ++p) xe[p] = xe[p] - se[p];
#pragma unroll
      for (int p = 0; p < NP; ++p) mx.fold2(j, se[p][0], se[p][1]);  // one v_max3_f32 per pair
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int l = 0; l < 2; ++l) {
        const int e = 2 * p + l;
        xv.v[e] = xe[p][l], yv.v[e] = ye[p][l], rv.v[e] = re[p][l], zv.v[e] = ze[p][l], sv.v[e] = se[p][l];
        fy += 0.5 * (double)de[p][l] * (double)ye[p][l] * (double)ye[p][l] + (double)qe[p][l] * (double)ye[p][l];
        if constexpr (GKIND == PG_G_NORML1) gz += fabs((double)ze[p][l]);
      }
    }
  } else {
    const T de = a.f.dv != nullptr ? d.v[0] : a.f.ds;
    const T qe = a.f.qv != nullptr ? q.v[0] : a.f.qs;
    const SepQuadElem<T> pe = sepquad_prepare(a.gamma, de, qe);
    T xe = xv.v[0], ye, re, ze, se;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      ye = sepquad_apply(pe, xe);
      re = fma(T(2), ye, -xe);
      if constexpr (GKIND == PG_G_NORML1)
        ze = soft_threshold(re, a.p0);
      else if constexpr (GKIND == PG_G_INDBOX)
        ze = pg_med3(re, a.p0, a.p1);
      else
        ze = re;
      se = ye - ze;
      xe = xe - se;
      mx.fold(j, fabs(se));
    }
    xv.v[0] = xe, yv.v[0] = ye, rv.v[0] = re, zv.v[0] = ze, sv.v[0] = se;
    fy += 0.5 * (double)de * (double)ye * (double)ye + (double)qe * (double)ye;
    if constexpr (GKIND == PG_G_NORML1) gz += fabs((double)ze);
  }
  st<T, N>(a.x_out, i, xv);
  st_nt<T, N>(a.y, i, yv);
  if (a.r != nullptr) st_nt<T, N>(a.r, i, rv);
  if (a.z != nullptr) st_nt<T, N>(a.z, i, zv);
  if (a.res != nullptr) st_nt<T, N>(a.res, i, sv);
}

// 512-thread workgroups, two per CU, for every K: the K running maxima live in T registers over the whole sweep and are handed
// to the grid reduction one at a time (grid_reduce_finalize_streamed; the array form kept 4 (K + 2) registers live at the end of
// the kernel and forced 256-thread workgroups at K = 32: 125 us per block against 118 us now, K = 64: 218 us)
#ifndef DR_BS
#define DR_BS 512
#endif
template <int K>
constexpr int dr_block_bs() { return DR_BS; }

template <typename T, int GKIND, int K>
__global__ __launch_bounds__(dr_block_bs<K>()) void dr_block_kernel(int64_t n, bool vec_ok, DRBlockArgs<T, GKIND, K> a,
                                                               double* __restrict__ red_partials,
                                                               unsigned* __restrict__ red_counter,
                                                               double* __restrict__ out) {
  constexpr int VEC = VecOf<T>::N;
  constexpr int DR_BLOCK_BS = dr_block_bs<K>();
  DRMaxRegs<T, K> mx;
  mx.init();
  double fy = 0.0, gz = 0.0;
  const int64_t tid = (int64_t)blockIdx.x * DR_BLOCK_BS + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * DR_BLOCK_BS;
  if (vec_ok) {
    const int64_t nvec = n / VEC;
    // Software pipeline: the next vector's x, d, q are requested BEFORE the K iterations on the current one, so the
    // memory phase of a trip hides behind the arithmetic of the previous one (all waves of the grid start together and
    // would otherwise alternate between loading and computing in step: 21 us + 16 x 2.8 us per sweep at n = 10^7).
    // The prefetch is unconditional -- past the end it re-reads the thread's last vector -- because a branch around it
    // makes the compiler wait for it (vmcnt(0)) before the arithmetic, which undoes the overlap.
    if (a.f.dv != nullptr && a.f.qv != nullptr) {
      if (tid < nvec) {
        DRBlockIn<T, VEC> cur = dr_block_load<T, GKIND, K, VEC, true>(a, tid * VEC);
        for (int64_t v = tid; v < nvec; v += nthreads) {
          const int64_t vn = v + nthreads < nvec ? v + nthreads : v;
          const DRBlockIn<T, VEC> nxt = dr_block_load<T, GKIND, K, VEC, true>(a, vn * VEC);
          __builtin_amdgcn_sched_barrier(0);  // keep the loads up here (the scheduler sinks them below the arithmetic)
          dr_block_compute<T, GKIND, K, VEC>(a, v * VEC, cur, mx, fy, gz);
          cur = nxt;
        }
      }
    } else {  // scalar d or q: nothing worth prefetching beyond x
      for (int64_t v = tid; v < nvec; v += nthreads)
        dr_block_compute<T, GKIND, K, VEC>(a, v * VEC, dr_block_load<T, GKIND, K, VEC>(a, v * VEC), mx, fy, gz);
    }
    for (int64_t i = nvec * VEC + tid; i < n; i += nthreads)
      dr_block_compute<T, GKIND, K, 1>(a, i, dr_block_load<T, GKIND, K, 1>(a, i), mx, fy, gz);
  } else {
    for (int64_t i = tid; i < n; i += nthreads)
      dr_block_compute<T, GKIND, K, 1>(a, i, dr_block_load<T, GKIND, K, 1>(a, i), mx, fy, gz);
  }
  // the K maxima are reduced within the wave in working precision (DPP row steps + v_readlane: ~11 instructions each
  // against ~36 for an fp64 shuffle chain -- the end of the kernel was a fifth of its instruction count)
  if constexpr (K >= 32) {
    // one slot at a time, nothing held per thread (grid_reduce_finalize would keep 4 (K + 2) registers live: one wave per
    // SIMD at K = 64); summation / maximum order over waves and workgroups is the same
    const double fyw = pg_wave_allreduce<false, double>(fy), gzw = pg_wave_allreduce<false, double>(gz);
    auto wave_val = [&](int k) -> double {
      if (k < K) return (double)pg_wave_allreduce<true, T>((T)mx.get(k));
      return k == K ? fyw : gzw;
    };
    grid_reduce_finalize_streamed<K + 2, (K >= 64 ? ~0ull : ((1ull << (K & 63)) - 1ull)), DR_BLOCK_BS / 64>(
        wave_val, red_partials, red_counter, out, a.gscale);
  } else {
    double v[K + 2], ps[K + 2];
#pragma unroll
    for (int j = 0; j < K; ++j) v[j] = (double)pg_wave_allreduce<true, T>((T)mx.get(j)), ps[j] = 1.0;
    v[K] = pg_wave_allreduce<false, double>(fy), ps[K] = 1.0;
    v[K + 1] = pg_wave_allreduce<false, double>(gz), ps[K + 1] = a.gscale;
    grid_reduce_finalize<K + 2, (1ull << K) - 1ull, DR_BLOCK_BS / 64, true>(v, red_partials, red_counter, out, ps);
  }
}

// smooth losses on m-vectors (the `f` of PANOC's f(Ax)); acc[0] = f(u), grad written elementwise
//   LOSS 0: squared distance  f(u) = ||u - b||^2 / 2 , grad = u - b     (benchmark/benchmarks.jl:19-28)
//   LOSS 1: logistic          f(u) = sum log(1 + exp(-(u - b))), grad = -1 / (1 + exp(u - b))
//                             (test/problems/test_sparse_logistic_small.jl:20-26, labels all one)
template <typename T, int LOSS>
struct LossF {
  T* grad;
  const T* u;
  const T* b;
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double* acc) const {
    Pack<T, N> uv = ld<T, N>(u, i), bv = ld<T, N>(b, i), g;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const T t = uv.v[e] - bv.v[e];
      if constexpr (LOSS == 0) {
        g.v[e] = t;
        acc[0] += (double)t * (double)t;
      } else {
        acc[0] += (double)log(T(1) + exp(-t));
        g.v[e] = -T(1) / (T(1) + exp(t));
      }
    }
    st<T, N>(grad, i, g);
  }
  __device__ double post_scale(int) const { return LOSS == 0 ? 0.5 : 1.0; }
};

// r = a r1 + b r2 ; acc[0] = sum r^2  (residual at the extrapolated point from the residuals at z and z_prev:
// A x - b = (1 + beta)(A z - b) - beta (A z_prev - b), the b terms cancel exactly)
template <typename T>
struct ResidualComboF {
  T* r;
  const T* r1;
  const T* r2;
  T a, b;
  double fscale;
  T* f_typed;  // nullable: mirror of f in working precision (slot n of the all-reduce payload)
  template <int N>
  __device__ __forceinline__ void apply(int64_t i, double* acc) const {
    Pack<T, N> u = ld<T, N>(r1, i), w = ld<T, N>(r2, i), o;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      o.v[e] = a * u.v[e] + b * w.v[e];
      acc[0] += (double)o.v[e] * (double)o.v[e];
    }
    st<T, N>(r, i, o);
  }
  __device__ double post_scale(int) const { return fscale; }
};
template <typename T>
__device__ __forceinline__ void ew_on_final(const ResidualComboF<T>& f, const double* fin) {
  if (f.f_typed != nullptr) *f.f_typed = (T)fin[0];
}

template <typename T>
pg_status epilogue_t(pg_ctx* c, int64_t n, const T* x, const T* grad, double gamma, int g_kind, double g_p0,
                     double g_p1, T* y, T* z, T* res, T* grad_copy, const T* g_v0 = nullptr, const T* g_v1 = nullptr) {
  const bool v = aligned16(x) && aligned16(grad) && aligned16(y) && aligned16(z) && aligned16(res) &&
                 (grad_copy == nullptr || aligned16(grad_copy)) && (g_v0 == nullptr || aligned16(g_v0)) &&
                 (g_v1 == nullptr || aligned16(g_v1));
  const T gm = (T)gamma;
  pg_prof_scope prof(c, PG_K_EPILOGUE);
  if (g_kind == PG_G_NORML1) {
    EpilogueF<T, PG_G_NORML1> f{x, grad, y, z, res, grad_copy, gm, (T)(gm * (T)g_p0), T(0), g_v0 != nullptr ? 1.0 : (double)(T)g_p0,
                                g_v0, nullptr};
    return launch_ew<T, decltype(f), 4, 0x2u>(c, n, v, f, c->dscal + PG_S_GZ);
  }
  if (g_kind == PG_G_INDBOX) {
    EpilogueF<T, PG_G_INDBOX> f{x, grad, y, z, res, grad_copy, gm, (T)g_p0, (T)g_p1, 0.0, g_v0, g_v1};
    return launch_ew<T, decltype(f), 4, 0x2u>(c, n, v, f, c->dscal + PG_S_GZ);
  }
  if (g_kind == PG_G_ZERO) {
    EpilogueF<T, PG_G_ZERO> f{x, grad, y, z, res, grad_copy, gm, T(0), T(0), 0.0};
    return launch_ew<T, decltype(f), 4, 0x2u>(c, n, v, f, c->dscal + PG_S_GZ);
  }
  pg_set_error("unknown g_kind %d", g_kind);
  return PG_ERR_INVALID;
}

#define PG_DISPATCH(dtype, CALL_F32, CALL_F64)                    \
  do {                                                            \
    if ((dtype) == PG_F32) return CALL_F32;                       \
    if ((dtype) == PG_F64) return CALL_F64;                       \
    pg_set_error("dtype must be PG_F32 or PG_F64, got %d", dtype); \
    return PG_ERR_INVALID;                                        \
  } while (0)

template <typename T>
pg_status axpby_t(pg_ctx* c, int64_t n, void* out, double a, const void* x, double b, const void* y) {
  AxpbyF<T> f{(T*)out, (const T*)x, (b != 0.0 ? (const T*)y : nullptr), (T)a, (T)b};
  const bool v = aligned16(out) && aligned16(x) && (f.y == nullptr || aligned16(y));
  return launch_ew<T, AxpbyF<T>, 0, 0u>(c, n, v, f, c->dscal + PG_S_MISC);
}
template <typename T>
pg_status add_scalar_t(pg_ctx* c, int64_t n, void* out, const void* x, double cst) {
  AddScalarF<T> f{(T*)out, (const T*)x, (T)cst};
  const bool v = aligned16(out) && (x == nullptr || aligned16(x));
  return launch_ew<T, AddScalarF<T>, 0, 0u>(c, n, v, f, c->dscal + PG_S_MISC);
}
template <typename T>
pg_status extrapolate_t(pg_ctx* c, int64_t n, void* x, const void* z, const void* zp, double beta) {
  pg_prof_scope prof(c, PG_K_EXTRAPOLATE);
  ExtrapolateF<T> f{(T*)x, (const T*)z, (const T*)zp, (T)beta};
  const bool v = aligned16(x) && aligned16(z) && aligned16(zp);
  return launch_ew<T, ExtrapolateF<T>, 0, 0u>(c, n, v, f, c->dscal + PG_S_MISC);
}
template <typename T>
pg_status dot_t(pg_ctx* c, int64_t n, const void* x, const void* y) {
  DotF<T> f{(const T*)x, (const T*)y};
  return launch_ew<T, DotF<T>, 1, 0u>(c, n, aligned16(x) && aligned16(y), f, c->dscal + PG_S_MISC);
}
template <typename T>
pg_status nrminf_t(pg_ctx* c, int64_t n, const void* x) {
  NrmInfF<T> f{(const T*)x};
  return launch_ew<T, NrmInfF<T>, 1, 0x1u>(c, n, aligned16(x), f, c->dscal + PG_S_MISC);
}
template <typename T>
pg_status norm1_t(pg_ctx* c, int64_t n, const void* x, double lam) {
  Norm1F<T> f{(const T*)x, (double)(T)lam};
  return launch_ew<T, Norm1F<T>, 1, 0u>(c, n, aligned16(x), f, c->dscal + PG_S_MISC);
}
template <typename T>
pg_status prox_l1_t(pg_ctx* c, int64_t n, void* y, const void* x, double lam, double gamma) {
  ProxL1F<T> f{(T*)y, (const T*)x, (T)((T)gamma * (T)lam), (double)(T)lam};
  return launch_ew<T, ProxL1F<T>, 1, 0u>(c, n, aligned16(x) && aligned16(y), f, c->dscal + PG_S_MISC);
}
template <typename T>
pg_status norm1w_t(pg_ctx* c, int64_t n, const void* x, const void* lam) {
  Norm1WF<T> f{(const T*)x, (const T*)lam};
  return launch_ew<T, Norm1WF<T>, 1, 0u>(c, n, aligned16(x) && aligned16(lam), f, c->dscal + PG_S_MISC);
}
template <typename T>
pg_status prox_l1w_t(pg_ctx* c, int64_t n, void* y, const void* x, const void* lam, double gamma) {
  ProxL1WF<T> f{(T*)y, (const T*)x, (const T*)lam, (T)gamma};
  return launch_ew<T, ProxL1WF<T>, 1, 0u>(c, n, aligned16(x) && aligned16(y) && aligned16(lam), f, c->dscal + PG_S_MISC);
}
template <typename T>
pg_status prox_box_t(pg_ctx* c, int64_t n, void* y, const void* x, double lo, double hi, const void* lov,
                     const void* hiv) {
  ProxBoxF<T> f{(T*)y, (const T*)x, (T)lo, (T)hi, (const T*)lov, (const T*)hiv};
  const bool v = aligned16(x) && aligned16(y) && (!lov || aligned16(lov)) && (!hiv || aligned16(hiv));
  return launch_ew<T, ProxBoxF<T>, 0, 0u>(c, n, v, f, c->dscal + PG_S_MISC);
}

template <typename T>
pg_status prox_sepquad_t(pg_ctx* c, int64_t n, void* y, const void* x, const void* dv, double ds, const void* qv,
                         double qs, double gamma) {
  ProxSepQuadF<T> f{(T*)y, (const T*)x, {(const T*)dv, (const T*)qv, (T)ds, (T)qs}, (T)gamma};
  const bool v = aligned16(x) && aligned16(y) && (!dv || aligned16(dv)) && (!qv || aligned16(qv));
  return launch_ew<T, ProxSepQuadF<T>, 1, 0u>(c, n, v, f, c->dscal + PG_S_MISC);
}

// launch geometry of the stepping kernel; PG_DR_STEP_GEOM = "<threads>x<blocks per CU>x<vectors per trip>" for experiments
template <typename T, typename F>
pg_status dr_step_launch(pg_ctx* c, int64_t n, bool v, const F& f_in, int slot = PG_S_DR) {
  static const char* geom = getenv("PG_DR_STEP_GEOM");
  static const bool nt_out = !(getenv("PG_DR_NT") && atoi(getenv("PG_DR_NT")) == 0);
  F f = f_in;
  f.nt_out = nt_out;
  int bs = 1024, bpc = 1, unr = 2;
  if (geom != nullptr && *geom) sscanf(geom, "%dx%dx%d", &bs, &bpc, &unr);
#define PG_DR_GEOM(BB, UU) \
  if (bs == BB && unr == UU) return launch_ew<T, F, 3, 0x1u, BB, UU>(c, n, v, f, c->dscal + slot, bpc)
  PG_DR_GEOM(1024, 2);
  PG_DR_GEOM(1024, 1);
  PG_DR_GEOM(1024, 4);
  PG_DR_GEOM(512, 1);
  PG_DR_GEOM(512, 2);
  PG_DR_GEOM(512, 4);
  PG_DR_GEOM(256, 1);
  PG_DR_GEOM(256, 2);
  PG_DR_GEOM(256, 4);
#undef PG_DR_GEOM
  pg_set_error("PG_DR_STEP_GEOM: unsupported geometry %dx%dx%d", bs, bpc, unr);
  return PG_ERR_INVALID;
}

template <typename T>
pg_status dr_step_t(pg_ctx* c, int64_t n, void* x, void* y, void* r, void* z, void* res, const void* dv, double ds,
                    const void* qv, double qs, int g_kind, double g_p0, double g_p1, double gamma, const void* x_in = nullptr,
                    int slot = PG_S_DR) {
  const bool v = aligned16(x) && aligned16(y) && (!r || aligned16(r)) && (!z || aligned16(z)) &&
                 (!res || aligned16(res)) && (!dv || aligned16(dv)) && (!qv || aligned16(qv)) && (!x_in || aligned16(x_in));
  const T gm = (T)gamma;
  SepQuadParams<T> fp{(const T*)dv, (const T*)qv, (T)ds, (T)qs};
  pg_prof_scope prof(c, PG_K_DR_STEP);
  if (g_kind == PG_G_NORML1) {
    DRStepF<T, PG_G_NORML1> f{(T*)x, (T*)y, (T*)r, (T*)z, (T*)res, fp, gm, (T)(gm * (T)g_p0), T(0), (double)(T)g_p0};
    f.x_in = (const T*)x_in;
    return dr_step_launch<T>(c, n, v, f, slot);
  }
  if (g_kind == PG_G_INDBOX) {
    DRStepF<T, PG_G_INDBOX> f{(T*)x, (T*)y, (T*)r, (T*)z, (T*)res, fp, gm, (T)g_p0, (T)g_p1, 0.0};
    f.x_in = (const T*)x_in;
    return dr_step_launch<T>(c, n, v, f, slot);
  }
  if (g_kind == PG_G_ZERO) {
    DRStepF<T, PG_G_ZERO> f{(T*)x, (T*)y, (T*)r, (T*)z, (T*)res, fp, gm, T(0), T(0), 0.0};
    f.x_in = (const T*)x_in;
    return dr_step_launch<T>(c, n, v, f, slot);
  }
  pg_set_error("unknown g_kind %d", g_kind);
  return PG_ERR_INVALID;
}

template <typename T, int GKIND, int K>
pg_status dr_block_launch(pg_ctx* c, int64_t n, bool vec_ok, const DRBlockArgs<T, GKIND, K>& a, int slot_base) {
  constexpr int DR_BLOCK_BS = dr_block_bs<K>();
  int64_t blocks = (n / VecOf<T>::N + DR_BLOCK_BS) / DR_BLOCK_BS;
  if (blocks > (int64_t)c->num_cu * (1024 / DR_BLOCK_BS)) blocks = (int64_t)c->num_cu * (1024 / DR_BLOCK_BS);
  if (blocks > PG_RED_MAX_BLOCKS) blocks = PG_RED_MAX_BLOCKS;
  hipLaunchKernelGGL((dr_block_kernel<T, GKIND, K>), dim3((unsigned)blocks), dim3(DR_BLOCK_BS), 0, c->stream, n, vec_ok, a,
                     c->red_partials, c->red_counter, c->dscal + slot_base);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

template <typename T, int K>
pg_status dr_block_t(pg_ctx* c, int64_t n, const void* x_in, void* x_out, void* y, void* r, void* z, void* res,
                     const void* dv, double ds, const void* qv, double qs, int g_kind, double g_p0, double g_p1,
                     double gamma, int slot_base) {
  const bool v = aligned16(x_in) && aligned16(x_out) && aligned16(y) && (!r || aligned16(r)) && (!z || aligned16(z)) &&
                 (!res || aligned16(res)) && (!dv || aligned16(dv)) && (!qv || aligned16(qv));
  const T gm = (T)gamma;
  SepQuadParams<T> fp{(const T*)dv, (const T*)qv, (T)ds, (T)qs};
  pg_prof_scope prof(c, PG_K_DR_STEP);
  if (g_kind == PG_G_NORML1) {
    DRBlockArgs<T, PG_G_NORML1, K> a{(const T*)x_in, (T*)x_out, (T*)y, (T*)r, (T*)z, (T*)res, fp, gm, (T)(gm * (T)g_p0),
                                     T(0), (double)(T)g_p0};
    return dr_block_launch(c, n, v, a, slot_base);
  }
  if (g_kind == PG_G_INDBOX) {
    DRBlockArgs<T, PG_G_INDBOX, K> a{(const T*)x_in, (T*)x_out, (T*)y, (T*)r, (T*)z, (T*)res, fp, gm, (T)g_p0, (T)g_p1, 0.0};
    return dr_block_launch(c, n, v, a, slot_base);
  }
  if (g_kind == PG_G_ZERO) {
    DRBlockArgs<T, PG_G_ZERO, K> a{(const T*)x_in, (T*)x_out, (T*)y, (T*)r, (T*)z, (T*)res, fp, gm, T(0), T(0), 0.0};
    return dr_block_launch(c, n, v, a, slot_base);
  }
  pg_set_error("unknown g_kind %d", g_kind);
  return PG_ERR_INVALID;
}

// DouglasRachford driver loop (ProximalAlgorithms.jl:114-123 with the stop rule of douglas_rachford.jl:65-69) in
// blocks of K iterations per sweep; returns the number of iterations k and leaves exactly the state of iteration k.
//
// Two blocks are kept in flight: while the host waits for block b's K residual norms (an event, not a stream
// synchronisation) block b + 1 is already queued behind it, so the device does not idle for the ~12 us of a launch +
// read-back per block (a fifth of a 16-iteration sweep at n = 10^7).  x rotates through three buffers (x, x_alt and a
// context-owned workspace): block b + 1 must not overwrite the INPUT of block b, from which the state is replayed with
// single steps when one of b's inner iterations satisfies the stop rule (block b + 1 has by then overwritten y / r / z /
// res with a state past the stop).  No successor is queued behind a block that reaches maxit.
template <typename T>
pg_status dr_run_t(pg_ctx* c, int64_t n, void* x, void* x_alt, void* y, void* r, void* z, void* res, const void* dv,
                   double ds, const void* qv, double qs, int g_kind, double g_p0, double g_p1, double gamma, double tol,
                   int64_t maxit, int K, int64_t* k_out, double* scalars_out) {
  const T gm = (T)gamma, tl = (T)tol;
  auto stop = [&](double res_inf) { return (T)res_inf / gm <= tl; };  // norm(res, Inf) / gamma <= tol  in T
  auto step = [&](void* xx) { return dr_step_t<T>(c, n, xx, y, r, z, res, dv, ds, qv, qs, g_kind, g_p0, g_p1, gamma); };
  void* cur = x;
  int64_t k = 0;
  double sc[3] = {0, 0, 0};
  bool done = false;
  if (K > 1 && maxit >= 8) {
    const size_t nb = (size_t)(n > 0 ? n : 1) * sizeof(T);
    if (c->dr_ws_bytes < nb) {
      if (c->dr_ws) {
        PG_HIP(hipStreamSynchronize(c->stream));
        PG_HIP(hipFree(c->dr_ws));
        c->dr_ws = nullptr;
        c->dr_ws_bytes = 0;  // (a failed hipMalloc below must not leave the old size standing)
      }
      PG_HIP(hipMalloc(&c->dr_ws, nb));
      c->dr_ws_bytes = nb;
    }
    for (int e = 0; e < 2; ++e)
      if (c->dr_ev[e] == nullptr) PG_HIP(hipEventCreateWithFlags(&c->dr_ev[e], hipEventDisableTiming));
    void* bufs[3] = {x, x_alt, c->dr_ws};
    const int slot_base[2] = {PG_S_DRRUN, PG_S_DRRUN2};
    struct Blk {
      int in = 0, set = 0;
      bool live = false;
    };
    int in = 0;
    // blocks of Kc iterations while at least Kc are left; the remainder (< K) goes through the next smaller block sizes
    // (32, 16, 8) instead of up to K - 1 single steps with a host round trip each, then < 8 single steps
    for (int Kc = K; Kc >= 8 && !done; Kc >>= 1) {
      Blk A, B;
      auto launch = [&](Blk& b, int bin, int set) -> pg_status {
        b.in = bin, b.set = set, b.live = true;
        void *xi = bufs[bin], *xo = bufs[(bin + 1) % 3];
        PG_TRY(Kc == 64   ? (dr_block_t<T, 64>(c, n, xi, xo, y, r, z, res, dv, ds, qv, qs, g_kind, g_p0, g_p1, gamma, slot_base[set]))
               : Kc == 32 ? (dr_block_t<T, 32>(c, n, xi, xo, y, r, z, res, dv, ds, qv, qs, g_kind, g_p0, g_p1, gamma, slot_base[set]))
               : Kc == 16 ? (dr_block_t<T, 16>(c, n, xi, xo, y, r, z, res, dv, ds, qv, qs, g_kind, g_p0, g_p1, gamma, slot_base[set]))
                          : (dr_block_t<T, 8>(c, n, xi, xo, y, r, z, res, dv, ds, qv, qs, g_kind, g_p0, g_p1, gamma, slot_base[set])));
        PG_HIP(hipEventRecord(c->dr_ev[set], c->stream));
        return PG_OK;
      };
      while (!done && maxit - k >= Kc) {
        if (!A.live) PG_TRY(launch(A, in, 0));
        // a successor only if block A cannot be the last one of this size (it reaches maxit exactly when k + Kc >= maxit)
        if (!B.live && maxit - (k + Kc) >= Kc) PG_TRY(launch(B, (A.in + 1) % 3, 1 - A.set));
        PG_HIP(hipEventSynchronize(c->dr_ev[A.set]));
        const double* hs = c->hscal + slot_base[A.set];
        int hit = -1;
        for (int j = 0; j < Kc && hit < 0; ++j)
          if (k + j + 1 >= maxit || stop(hs[j])) hit = j;
        if (hit < 0) {  // no stop inside A: its output is the next input; B (if queued) becomes the block to wait for
          k += Kc;
          in = (A.in + 1) % 3;
          A = B;
          B.live = false;
        } else if (hit == Kc - 1 && !B.live) {  // stopped exactly at the end of A and nothing ran past it: the state is A's
          k += Kc;
          in = (A.in + 1) % 3;
          sc[0] = hs[Kc - 1], sc[1] = hs[Kc], sc[2] = hs[Kc + 1];
          done = true;
        } else {
          // an inner iteration stopped (or a successor has overwritten y / r / z / res): replay hit + 1 single steps from
          // A's input -- same arithmetic, same bits; stream order puts them behind the queued successor
          in = A.in;
          for (int j = 0; j <= hit; ++j) PG_TRY(step(bufs[in]));
          PG_TRY(pg_read_scalars(c, PG_S_DR, 3));
          for (int q3 = 0; q3 < 3; ++q3) sc[q3] = c->hscal[PG_S_DR + q3];
          k += hit + 1;
          done = true;
        }
      }
    }
    cur = bufs[in];
  }
  while (!done && k < maxit) {  // fewer than 8 iterations left (or block = 1): step by step
    PG_TRY(step(cur));
    PG_TRY(pg_read_scalars(c, PG_S_DR, 3));
    for (int q3 = 0; q3 < 3; ++q3) sc[q3] = c->hscal[PG_S_DR + q3];
    ++k;
    done = k >= maxit || stop(sc[0]);
  }
  if (cur != x && n > 0) {
    if (hipMemcpyAsync(x, cur, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) {
      pg_set_error("hipMemcpyAsync failed");
      return PG_ERR_HIP;
    }
  }
  PG_TRY(pg_ctx_sync(c));
  if (k_out) *k_out = k;
  if (scalars_out)
    for (int q3 = 0; q3 < 3; ++q3) scalars_out[q3] = sc[q3];
  return PG_OK;
}

template <typename T>
pg_status loss_t(pg_ctx* c, int loss, int64_t m, const void* u, const void* b, void* grad) {
  const bool v = aligned16(u) && aligned16(b) && aligned16(grad);
  if (loss == 0) {
    LossF<T, 0> f{(T*)grad, (const T*)u, (const T*)b};
    return launch_ew<T, decltype(f), 1, 0u>(c, m, v, f, c->dscal + PG_S_MISC);
  }
  LossF<T, 1> f{(T*)grad, (const T*)u, (const T*)b};
  return launch_ew<T, decltype(f), 1, 0u>(c, m, v, f, c->dscal + PG_S_MISC);
}

pg_status finish_scalar(pg_ctx* c, int slot, double* out) {
  if (!out) return PG_OK;
  PG_TRY(pg_read_scalars(c, slot, 1));
  *out = c->hscal[slot];
  return PG_OK;
}

}  // namespace

pg_status pg_residual_combo_async(pg_ctx* c, int dtype, int64_t m, void* r_out, double a, const void* r1, double b,
                                  const void* r2, double f_scale, void* f_typed, double* f_dst) {
  const bool v = aligned16(r_out) && aligned16(r1) && aligned16(r2);
  if (f_dst == nullptr) f_dst = c->dscal + PG_S_F;
  if (dtype == PG_F32) {
    ResidualComboF<float> f{(float*)r_out, (const float*)r1, (const float*)r2, (float)a, (float)b, f_scale, (float*)f_typed};
    return launch_ew<float, ResidualComboF<float>, 1, 0u>(c, m, v, f, f_dst);
  }
  ResidualComboF<double> f{(double*)r_out, (const double*)r1, (const double*)r2, a, b, f_scale, (double*)f_typed};
  return launch_ew<double, ResidualComboF<double>, 1, 0u>(c, m, v, f, f_dst);
}

pg_status pg_fb_epilogue_async(pg_ctx* c, int dtype, int64_t n, const void* x, const void* grad, double gamma,
                               int g_kind, double g_p0, double g_p1, void* y, void* z, void* res, const void* g_v0,
                               const void* g_v1) {
  PG_DISPATCH(dtype,
              epilogue_t<float>(c, n, (const float*)x, (const float*)grad, gamma, g_kind, g_p0, g_p1, (float*)y,
                                (float*)z, (float*)res, nullptr, (const float*)g_v0, (const float*)g_v1),
              epilogue_t<double>(c, n, (const double*)x, (const double*)grad, gamma, g_kind, g_p0, g_p1,
                                 (double*)y, (double*)z, (double*)res, nullptr, (const double*)g_v0, (const double*)g_v1));
}

extern "C" {

#define PG_VEC_ARGS_OK(c, n) \
  PG_REQUIRE((c) != nullptr, "ctx is null"); \
  PG_REQUIRE((n) >= 0, "negative length")

pg_status pg_axpby(pg_ctx* c, int32_t dtype, int64_t n, void* out, double a, const void* x, double b,
                   const void* y) {
  PG_VEC_ARGS_OK(c, n);
  if (n == 0) return PG_OK;
  PG_REQUIRE(out != nullptr && x != nullptr && (b == 0.0 || y != nullptr), "null vector");
  PG_DISPATCH(dtype, axpby_t<float>(c, n, out, a, x, b, y), axpby_t<double>(c, n, out, a, x, b, y));
}

pg_status pg_add_scalar(pg_ctx* c, int32_t dtype, int64_t n, void* out, const void* x, double cst) {
  PG_VEC_ARGS_OK(c, n);
  if (n == 0) return PG_OK;
  PG_REQUIRE(out != nullptr && x != nullptr, "null vector");
  PG_DISPATCH(dtype, add_scalar_t<float>(c, n, out, x, cst), add_scalar_t<double>(c, n, out, x, cst));
}

pg_status pg_fill(pg_ctx* c, int32_t dtype, int64_t n, void* out, double cst) {
  PG_VEC_ARGS_OK(c, n);
  if (n == 0) return PG_OK;
  PG_REQUIRE(out != nullptr, "null vector");
  PG_DISPATCH(dtype, add_scalar_t<float>(c, n, out, nullptr, cst), add_scalar_t<double>(c, n, out, nullptr, cst));
}

pg_status pg_extrapolate(pg_ctx* c, int32_t dtype, int64_t n, void* x, const void* z, const void* zp,
                         double beta) {
  PG_VEC_ARGS_OK(c, n);
  if (n == 0) return PG_OK;
  PG_REQUIRE(x != nullptr && z != nullptr && zp != nullptr, "null vector");
  PG_DISPATCH(dtype, extrapolate_t<float>(c, n, x, z, zp, beta), extrapolate_t<double>(c, n, x, z, zp, beta));
}

pg_status pg_dot(pg_ctx* c, int32_t dtype, int64_t n, const void* x, const void* y, double* out) {
  PG_VEC_ARGS_OK(c, n);
  PG_REQUIRE(n == 0 || (x != nullptr && y != nullptr), "null vector");
  PG_REQUIRE(dtype == PG_F32 || dtype == PG_F64, "bad dtype");
  PG_TRY(dtype == PG_F32 ? dot_t<float>(c, n, x, y) : dot_t<double>(c, n, x, y));
  return finish_scalar(c, PG_S_MISC, out);
}

pg_status pg_nrm2sq(pg_ctx* c, int32_t dtype, int64_t n, const void* x, double* out) {
  return pg_dot(c, dtype, n, x, x, out);
}

pg_status pg_nrminf(pg_ctx* c, int32_t dtype, int64_t n, const void* x, double* out) {
  PG_VEC_ARGS_OK(c, n);
  PG_REQUIRE(n == 0 || x != nullptr, "null vector");
  PG_REQUIRE(dtype == PG_F32 || dtype == PG_F64, "bad dtype");
  PG_TRY(dtype == PG_F32 ? nrminf_t<float>(c, n, x) : nrminf_t<double>(c, n, x));
  return finish_scalar(c, PG_S_MISC, out);
}

pg_status pg_norml1_value(pg_ctx* c, int32_t dtype, int64_t n, const void* x, double lam, double* out) {
  PG_VEC_ARGS_OK(c, n);
  PG_REQUIRE(n == 0 || x != nullptr, "null vector");
  PG_REQUIRE(dtype == PG_F32 || dtype == PG_F64, "bad dtype");
  PG_TRY(dtype == PG_F32 ? norm1_t<float>(c, n, x, lam) : norm1_t<double>(c, n, x, lam));
  return finish_scalar(c, PG_S_MISC, out);
}

pg_status pg_prox_norml1(pg_ctx* c, int32_t dtype, int64_t n, void* y, const void* x, double lam, double gamma,
                         double* gy_out) {
  PG_VEC_ARGS_OK(c, n);
  PG_REQUIRE(n == 0 || (x != nullptr && y != nullptr), "null vector");
  PG_REQUIRE(dtype == PG_F32 || dtype == PG_F64, "bad dtype");
  PG_TRY(dtype == PG_F32 ? prox_l1_t<float>(c, n, y, x, lam, gamma) : prox_l1_t<double>(c, n, y, x, lam, gamma));
  return finish_scalar(c, PG_S_MISC, gy_out);
}

pg_status pg_norml1w_value(pg_ctx* c, int32_t dtype, int64_t n, const void* x, const void* lam_vec, double* out) {
  PG_VEC_ARGS_OK(c, n);
  PG_REQUIRE(n == 0 || (x != nullptr && lam_vec != nullptr), "null vector");
  PG_REQUIRE(dtype == PG_F32 || dtype == PG_F64, "bad dtype");
  PG_TRY(dtype == PG_F32 ? norm1w_t<float>(c, n, x, lam_vec) : norm1w_t<double>(c, n, x, lam_vec));
  return finish_scalar(c, PG_S_MISC, out);
}

pg_status pg_prox_norml1w(pg_ctx* c, int32_t dtype, int64_t n, void* y, const void* x, const void* lam_vec, double gamma,
                          double* gy_out) {
  PG_VEC_ARGS_OK(c, n);
  PG_REQUIRE(n == 0 || (x != nullptr && y != nullptr && lam_vec != nullptr), "null vector");
  PG_REQUIRE(dtype == PG_F32 || dtype == PG_F64, "bad dtype");
  PG_TRY(dtype == PG_F32 ? prox_l1w_t<float>(c, n, y, x, lam_vec, gamma) : prox_l1w_t<double>(c, n, y, x, lam_vec, gamma));
  return finish_scalar(c, PG_S_MISC, gy_out);
}

pg_status pg_prox_indbox(pg_ctx* c, int32_t dtype, int64_t n, void* y, const void* x, double lo, double hi,
                         const void* lo_vec, const void* hi_vec, double* gy_out) {
  PG_VEC_ARGS_OK(c, n);
  PG_REQUIRE(n == 0 || (x != nullptr && y != nullptr), "null vector");
  PG_REQUIRE(dtype == PG_F32 || dtype == PG_F64, "bad dtype");
  if (n > 0)
    PG_TRY(dtype == PG_F32 ? prox_box_t<float>(c, n, y, x, lo, hi, lo_vec, hi_vec)
                           : prox_box_t<double>(c, n, y, x, lo, hi, lo_vec, hi_vec));
  if (gy_out) *gy_out = 0.0;
  return PG_OK;
}

pg_status pg_fb_epilogue(pg_ctx* c, int32_t dtype, int64_t n, const void* x, const void* grad, double gamma,
                         int32_t g_kind, double g_p0, double g_p1, void* y, void* z, void* res,
                         double* scalars_out) {
  PG_VEC_ARGS_OK(c, n);
  PG_REQUIRE(n == 0 || (x && grad && y && z && res), "null vector");
  PG_REQUIRE(dtype == PG_F32 || dtype == PG_F64, "bad dtype");
  PG_TRY(pg_fb_epilogue_async(c, dtype, n, x, grad, gamma, g_kind, g_p0, g_p1, y, z, res));
  if (scalars_out) {
    PG_TRY(pg_read_scalars(c, PG_S_GZ, 4));
    for (int k = 0; k < 4; ++k) scalars_out[k] = c->hscal[PG_S_GZ + k];
  }
  return PG_OK;
}

pg_status pg_prox_sepquad(pg_ctx* c, int32_t dtype, int64_t n, void* y, const void* x, const void* d_vec, double d,
                          const void* q_vec, double q, double gamma, double* fy_out) {
  PG_VEC_ARGS_OK(c, n);
  PG_REQUIRE(n == 0 || (x != nullptr && y != nullptr), "null vector");
  PG_REQUIRE(dtype == PG_F32 || dtype == PG_F64, "bad dtype");
  PG_TRY(dtype == PG_F32 ? prox_sepquad_t<float>(c, n, y, x, d_vec, d, q_vec, q, gamma)
                         : prox_sepquad_t<double>(c, n, y, x, d_vec, d, q_vec, q, gamma));
  return finish_scalar(c, PG_S_MISC, fy_out);
}

pg_status pg_dr_step(pg_ctx* c, int32_t dtype, int64_t n, void* x, void* y, void* r, void* z, void* res,
                     const void* d_vec, double d, const void* q_vec, double q, int32_t g_kind, double g_p0,
                     double g_p1, double gamma, double* scalars_out) {
  PG_VEC_ARGS_OK(c, n);
  PG_REQUIRE(n == 0 || (x != nullptr && y != nullptr), "null vector");
  PG_REQUIRE(dtype == PG_F32 || dtype == PG_F64, "bad dtype");
  PG_TRY(dtype == PG_F32 ? dr_step_t<float>(c, n, x, y, r, z, res, d_vec, d, q_vec, q, g_kind, g_p0, g_p1, gamma)
                         : dr_step_t<double>(c, n, x, y, r, z, res, d_vec, d, q_vec, q, g_kind, g_p0, g_p1, gamma));
  if (scalars_out) {
    PG_TRY(pg_read_scalars(c, PG_S_DR, 3));
    for (int k = 0; k < 3; ++k) scalars_out[k] = c->hscal[PG_S_DR + k];
  }
  return PG_OK;
}

// Stepping with the NEXT iteration already in flight.  `for state in iter` reads a scalar (norm(res, Inf), for the stop rule) after
// every iteration, and at n = 10^7 the round trip -- launch, synchronise, return to the host language -- is half as long as
// the 34 us kernel itself.  The state is separable from its successor: iteration k + 1 reads x_k and nothing else, so it can be
// launched, into a SECOND set of state vectors, before the host has looked at iteration k; the host's share then hides behind
// the kernel.  _async launches one iteration out of place (x_in -> x_out, y, r, z, res) with its scalars going to slot 0 | 1
// and records an event; _wait blocks on THAT iteration only (not on the one launched after it) and returns its scalars.
pg_status pg_dr_step_async(pg_ctx* c, int32_t dtype, int64_t n, const void* x_in, void* x_out, void* y, void* r, void* z, void* res,
                           const void* d_vec, double d, const void* q_vec, double q, int32_t g_kind, double g_p0, double g_p1,
                           double gamma, int32_t slot) {
  PG_VEC_ARGS_OK(c, n);
  PG_REQUIRE(n == 0 || (x_in != nullptr && x_out != nullptr && y != nullptr), "null vector");
  PG_REQUIRE(dtype == PG_F32 || dtype == PG_F64, "bad dtype");
  PG_REQUIRE(slot == 0 || slot == 1, "slot must be 0 or 1");
  if (c->dr_ev[slot] == nullptr) PG_HIP(hipEventCreateWithFlags(&c->dr_ev[slot], hipEventDisableTiming));
  const int base = slot == 0 ? PG_S_DRA : PG_S_DRB;
  PG_TRY(dtype == PG_F32 ? dr_step_t<float>(c, n, x_out, y, r, z, res, d_vec, d, q_vec, q, g_kind, g_p0, g_p1, gamma, x_in, base)
                         : dr_step_t<double>(c, n, x_out, y, r, z, res, d_vec, d, q_vec, q, g_kind, g_p0, g_p1, gamma, x_in, base));
  PG_HIP(hipEventRecord(c->dr_ev[slot], c->stream));
  return PG_OK;
}

pg_status pg_dr_step_wait(pg_ctx* c, int32_t slot, double* scalars_out) {
  PG_REQUIRE(c != nullptr && scalars_out != nullptr, "null argument");
  PG_REQUIRE((slot == 0 || slot == 1) && c->dr_ev[slot] != nullptr, "no iteration was launched into this slot");
  PG_HIP(hipEventSynchronize(c->dr_ev[slot]));
  const int base = slot == 0 ? PG_S_DRA : PG_S_DRB;
  for (int k = 0; k < 3; ++k) scalars_out[k] = c->hscal[base + k];
  return PG_OK;
}

pg_status pg_dr_run(pg_ctx* c, int32_t dtype, int64_t n, void* x, void* x_alt, void* y, void* r, void* z, void* res,
                    const void* d_vec, double d, const void* q_vec, double q, int32_t g_kind, double g_p0, double g_p1,
                    double gamma, double tol, int64_t maxit, int32_t block, int64_t* k_out, double* scalars_out) {
  PG_VEC_ARGS_OK(c, n);
  PG_REQUIRE(n == 0 || (x != nullptr && y != nullptr), "null vector");
  PG_REQUIRE(dtype == PG_F32 || dtype == PG_F64, "bad dtype");
  PG_REQUIRE(block == 1 || block == 8 || block == 16 || block == 32 || block == 64, "block must be 1, 8, 16, 32 or 64");
  PG_REQUIRE(block == 1 || (x_alt != nullptr && x_alt != x) || n == 0, "x_alt (a second n-vector) is required when block > 1");
  PG_REQUIRE(maxit >= 1, "maxit must be >= 1");
  PG_REQUIRE(gamma > 0, "gamma must be positive");
  // ProximalOperators.IndBox refuses lb > ub at construction; the blocked kernel clamps with v_med3_f32 on that premise
  PG_REQUIRE(g_kind != PG_G_INDBOX || g_p0 <= g_p1, "IndBox needs lo <= hi");
  return dtype == PG_F32 ? dr_run_t<float>(c, n, x, x_alt, y, r, z, res, d_vec, d, q_vec, q, g_kind, g_p0, g_p1, gamma,
                                           tol, maxit, block, k_out, scalars_out)
                         : dr_run_t<double>(c, n, x, x_alt, y, r, z, res, d_vec, d, q_vec, q, g_kind, g_p0, g_p1, gamma,
                                            tol, maxit, block, k_out, scalars_out);
}

pg_status pg_loss_value_and_gradient(pg_ctx* c, int32_t dtype, int32_t loss, int64_t m, const void* u, const void* b,
                                     void* grad, double* f_out) {
  PG_VEC_ARGS_OK(c, m);
  PG_REQUIRE(m == 0 || (u != nullptr && b != nullptr && grad != nullptr), "null vector");
  PG_REQUIRE(dtype == PG_F32 || dtype == PG_F64, "bad dtype");
  PG_REQUIRE(loss == PG_LOSS_SQDIST || loss == PG_LOSS_LOGISTIC, "unknown loss");
  PG_TRY(dtype == PG_F32 ? loss_t<float>(c, loss, m, u, b, grad) : loss_t<double>(c, loss, m, u, b, grad));
  return finish_scalar(c, PG_S_MISC, f_out);
}

}  // extern "C"
