/* Plain-C client of libproxgrad_hip.so: no Python, no torch -- exactly what a Julia `ccall` host sees.
 * Builds a small LASSO instance on the host, uploads it, runs FastForwardBackward (adaptive step) through
 * pg_iter_run, and checks the answer against a naive C restatement of the same iteration
 * (src/algorithms/fast_forward_backward.jl:73-145, fb_tools.jl:3-63, nesterov.jl:89-103) run on the CPU.
 * Exit code 0 = ok.  Usage: ffb_smoke [m n]   (compile: gcc -O2 -I include ffb_smoke.c -L... -lproxgrad_hip -lm) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "proxgrad_hip.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    pg_status s__ = (call);                                                      \
    if (s__ != PG_OK) {                                                          \
      fprintf(stderr, "%s failed (%d): %s\n", #call, s__, pg_last_error());      \
      return 2;                                                                  \
    }                                                                            \
  } while (0)

static double urand(unsigned* s) {
  *s = *s * 1664525u + 1013904223u;
  return ((*s >> 8) + 0.5) / 16777216.0;
}
static double nrand(unsigned* s) { return sqrt(-2.0 * log(urand(s))) * cos(6.283185307179586 * urand(s)); }

/* ---- naive CPU FFB (double), adaptive step, NormL1 ---- */
static double ls_vg(int m, int n, const double* A, const double* b, const double* x, double* grad, double* r) {
  double f = 0;
  for (int i = 0; i < m; ++i) {
    double s = -b[i];
    for (int j = 0; j < n; ++j) s += A[i + (size_t)j * m] * x[j];
    r[i] = s;
    f += s * s;
  }
  if (grad)
    for (int j = 0; j < n; ++j) {
      double s = 0;
      for (int i = 0; i < m; ++i) s += A[i + (size_t)j * m] * r[i];
      grad[j] = s;
    }
  return 0.5 * f;
}
static double epilogue(int n, const double* x, const double* g, double gamma, double lam, double* y, double* z, double* res,
                       double* dot, double* sq, double* rinf) {
  double gz = 0;
  *dot = *sq = *rinf = 0;
  for (int j = 0; j < n; ++j) {
    y[j] = x[j] - gamma * g[j];
    double gl = gamma * lam;
    z[j] = y[j] <= -gl ? y[j] + gl : (y[j] >= gl ? y[j] - gl : 0.0);
    res[j] = x[j] - z[j];
    gz += fabs(z[j]);
    *dot += g[j] * res[j];
    *sq += res[j] * res[j];
    if (fabs(res[j]) > *rinf) *rinf = fabs(res[j]);
  }
  return lam * gz;
}
static long cpu_ffb(int m, int n, const double* A, const double* b, double lam, double tol, long maxit, double* z_out) {
  double *x = calloc(n, 8), *g = calloc(n, 8), *y = calloc(n, 8), *z = calloc(n, 8), *res = calloc(n, 8), *zp = calloc(n, 8),
         *r = calloc(m, 8), *t = calloc(n, 8), *g2 = calloc(n, 8);
  double fx = ls_vg(m, n, A, b, x, g, r), dot, sq, rinf;
  for (int j = 0; j < n; ++j) t[j] = x[j] + 1;
  ls_vg(m, n, A, b, t, g2, r);
  double d = 0;
  for (int j = 0; j < n; ++j) d += (g2[j] - g[j]) * (g2[j] - g[j]);
  double gamma = 1.0 / (sqrt(d) / sqrt((double)n));
  epilogue(n, x, g, gamma, lam, y, z, res, &dot, &sq, &rinf);
  memcpy(zp, x, 8 * (size_t)n);
  double ss = -1, th = -1;
  long k = 1;
  while (!(k >= maxit || rinf / gamma <= tol)) {
    double upp = fx - dot + (1.0 / gamma / 2) * sq, fz = ls_vg(m, n, A, b, z, NULL, r);
    while (fz > upp + 10 * 2.220446049250313e-16 * (1 + fabs(fz)) && gamma >= 1e-7) {
      gamma *= 0.5;
      epilogue(n, x, g, gamma, lam, y, z, res, &dot, &sq, &rinf);
      upp = fx - dot + (1.0 / gamma / 2) * sq;
      fz = ls_vg(m, n, A, b, z, NULL, r);
    }
    if (ss < 0) { ss = gamma; th = 1; }
    double bb = th * th / ss, delta = bb * bb + 4 * th * th / (ss * gamma), thn = gamma * (-bb + sqrt(delta)) / 2;
    double beta = gamma * th * (1 - th) / (ss * thn + gamma * th * th);
    ss = gamma; th = thn;
    for (int j = 0; j < n; ++j) x[j] = z[j] + beta * (z[j] - zp[j]);
    double* tmp = zp; zp = z; z = tmp;
    fx = ls_vg(m, n, A, b, x, g, r);
    epilogue(n, x, g, gamma, lam, y, z, res, &dot, &sq, &rinf);
    ++k;
  }
  memcpy(z_out, z, 8 * (size_t)n);
  free(x); free(g); free(y); free(res); free(r); free(t); free(g2); free(z); free(zp);
  return k;
}

int main(int argc, char** argv) {
  const int m = argc > 2 ? atoi(argv[1]) : 60, n = argc > 2 ? atoi(argv[2]) : 150;
  unsigned seed = 12345;
  double *A = malloc(8 * (size_t)m * n), *b = malloc(8 * (size_t)m), *xt = calloc(n, 8);
  for (size_t k = 0; k < (size_t)m * n; ++k) A[k] = nrand(&seed) / sqrt((double)m);
  for (int j = 0; j < n; j += 17) xt[j] = nrand(&seed);
  for (int i = 0; i < m; ++i) {
    double s = 0.01 * nrand(&seed);
    for (int j = 0; j < n; ++j) s += A[i + (size_t)j * m] * xt[j];
    b[i] = s;
  }
  double lam = 0;
  for (int j = 0; j < n; ++j) {
    double s = 0;
    for (int i = 0; i < m; ++i) s += A[i + (size_t)j * m] * b[i];
    if (fabs(s) > lam) lam = fabs(s);
  }
  lam *= 0.1;

  if (pg_abi_version() != PG_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 2; }
  pg_ctx* ctx; pg_mat* Ad; pg_ls* f; pg_iter* it;
  CHECK(pg_ctx_create(0, NULL, &ctx));
  pg_device_info info;
  CHECK(pg_ctx_device_info(ctx, &info));
  CHECK(pg_mat_create(ctx, PG_F64, m, n, &Ad));
  CHECK(pg_mat_upload(Ad, A, m));
  void *bd, *x0d;
  CHECK(pg_malloc(ctx, 8 * (size_t)m, &bd));
  CHECK(pg_malloc(ctx, 8 * (size_t)n, &x0d));
  CHECK(pg_memcpy_h2d(ctx, bd, b, 8 * (size_t)m));
  CHECK(pg_memset_zero(ctx, x0d, 8 * (size_t)n));
  CHECK(pg_ls_create(ctx, Ad, bd, 1.0, &f));
  pg_iter_opts o;
  CHECK(pg_iter_opts_default(&o));
  o.fast = 1;
  o.g_kind = PG_G_NORML1;
  o.g_p0 = lam;
  CHECK(pg_iter_create(ctx, f, &o, &it));
  pg_iter_scalars sc;
  CHECK(pg_iter_init(it, x0d, &sc));
  int64_t k_gpu = 0;
  const double tol = 1e-8;
  CHECK(pg_iter_run(it, 1, 10000, tol, &k_gpu, &sc));
  pg_iter_state st;
  CHECK(pg_iter_state_view(it, &st));
  double *z_gpu = malloc(8 * (size_t)n), *z_cpu = malloc(8 * (size_t)n);
  CHECK(pg_memcpy_d2h(ctx, z_gpu, st.z, 8 * (size_t)n));
  long k_cpu = cpu_ffb(m, n, A, b, lam, tol, 10000, z_cpu);
  double err = 0, nz = 0;
  for (int j = 0; j < n; ++j) { if (fabs(z_gpu[j] - z_cpu[j]) > err) err = fabs(z_gpu[j] - z_cpu[j]); nz += z_gpu[j] != 0; }
  printf("device %s (%s, %d CUs)  m=%d n=%d lam=%.6f  k_gpu=%lld k_cpu=%ld  max|z_gpu-z_cpu|=%.3e  nnz=%.0f  gamma=%.6e\n",
         info.name, info.arch, info.compute_units, m, n, lam, (long long)k_gpu, k_cpu, err, nz, sc.gamma);
  /* error path: messages travel through pg_last_error() */
  pg_mat* bad = NULL;
  if (pg_mat_create(ctx, 7, 1, 1, &bad) == PG_OK || strlen(pg_last_error()) == 0) { fprintf(stderr, "bad dtype accepted\n"); return 3; }
  CHECK(pg_iter_destroy(it));
  CHECK(pg_ls_destroy(f));
  CHECK(pg_mat_destroy(Ad));
  CHECK(pg_free(ctx, bd));
  CHECK(pg_free(ctx, x0d));
  CHECK(pg_ctx_destroy(ctx));
  if (!(err <= 1e-9) || llabs((long long)k_gpu - k_cpu) > 2) { fprintf(stderr, "MISMATCH\n"); return 1; }
  printf("C_ABI_OK\n");
  return 0;
}
