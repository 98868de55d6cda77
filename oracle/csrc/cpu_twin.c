/* cpu_twin.c -- TEST INFRASTRUCTURE (oracle/): a C / OpenMP restatement of the FastForwardBackward fixed-step iteration on
 * LeastSquares + NormL1, used ONLY as the timed CPU leg of bench.py (`cpu_baseline`, kind "port") and checked against the
 * numpy oracle by tests/test_oracle_golden.py.  Nothing in the product package links or loads it.
 *
 * It follows the reference's op order UNFUSED, as SURVEY 8(d) prescribes for the CPU baseline:
 *   value_and_gradient(f, x)     benchmark/benchmarks.jl:11-17     res = A*x - b ; (norm(res)^2 / 2, A' * res)
 *   the step                     src/algorithms/fast_forward_backward.jl:131-142
 *   AdaptiveNesterovSequence     src/accel/nesterov.jl:89-103 (mf = 0)
 *   NormL1 prox                  soft threshold (ProximalOperators; pinned by the reference's known answers)
 * The two matrix-vector products are threaded over all cores (OpenMP), each as its own pass over A with a fresh
 * temporary like the reference's `A*x - b` and `A' * res`; the elementwise statements are single loops like Julia's
 * broadcasts.  A is column-major with leading dimension ld (a Julia Matrix{Float32}).
 *
 *   gcc -O3 -march=native -fopenmp -shared -fPIC oracle/csrc/cpu_twin.c -o oracle/_build/libcpu_twin.so
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

/* y = A x : every thread takes a contiguous range of columns and accumulates them into its own copy of y (streaming its
 * part of A once, contiguously); the copies are then summed in thread order (fixed order: deterministic for a given
 * thread count) */
static void gemv_n(const float* A, long m, long n, long ld, const float* x, float* y) {
  const int nt_max = omp_get_max_threads();
  float* part = malloc((size_t)nt_max * (size_t)m * sizeof(float));
  int nt_used = 1;
#pragma omp parallel
  {
    const int nt = omp_get_num_threads(), t = omp_get_thread_num();
#pragma omp single
    nt_used = nt;
    float* yy = part + (size_t)t * (size_t)m;
    memset(yy, 0, (size_t)m * sizeof(float));
    const long j0 = n * t / nt, j1 = n * (t + 1) / nt;
    long j = j0;
    for (; j + 3 < j1; j += 4) {  /* four columns per pass over the thread's y: one y load / store per four loads of A */
      const float x0 = x[j], x1 = x[j + 1], x2 = x[j + 2], x3 = x[j + 3];
      const float *a0 = A + j * ld, *a1 = a0 + ld, *a2 = a1 + ld, *a3 = a2 + ld;
#pragma omp simd
      for (long i = 0; i < m; ++i) yy[i] += (a0[i] * x0 + a1[i] * x1) + (a2[i] * x2 + a3[i] * x3);
    }
    for (; j < j1; ++j) {
      const float xj = x[j];
      const float* a = A + j * ld;
#pragma omp simd
      for (long i = 0; i < m; ++i) yy[i] += a[i] * xj;
    }
#pragma omp barrier
#pragma omp for schedule(static)
    for (long i = 0; i < m; ++i) {
      float s = part[i];
      for (int q = 1; q < nt; ++q) s += part[(size_t)q * (size_t)m + i];
      y[i] = s;
    }
  }
  (void)nt_used;
  free(part);
}

/* g = A' r : columns are independent dot products */
static void gemv_t(const float* A, long m, long n, long ld, const float* r, float* g) {
#pragma omp parallel for schedule(static)
  for (long j = 0; j < n; ++j) {
    const float* a = A + j * ld;
    float sum = 0.f;
#pragma omp simd reduction(+ : sum)
    for (long i = 0; i < m; ++i) sum += a[i] * r[i];
    g[j] = sum;
  }
}

/* benchmarks.jl:11-17 */
static float value_and_gradient(const float* A, long m, long n, long ld, const float* b, const float* x, float* res, float* grad) {
  gemv_n(A, m, n, ld, x, res);
  double nrm = 0.0;
  for (long i = 0; i < m; ++i) {
    res[i] -= b[i];
    nrm += (double)res[i] * (double)res[i];
  }
  gemv_t(A, m, n, ld, res, grad);
  return (float)(0.5 * nrm);
}

/* Read ceiling of this host on the SAME bytes: one pass over `count` floats summed with all threads (static chunks, like
 * gemv_t's column blocks), `reps` passes timed together.  Eight independent partial sums per thread: with ONE the loop is
 * bound by the latency of the dependent vector add (round 3: the pass ran at half of what sgemv sustained on the same matrix),
 * not by memory.  Returns the sum (so that the loads stay); seconds in *seconds_out. */
double cpu_twin_read_pass(const float* a, long count, int reps, double* seconds_out) {
  double total = 0.0;
  const double t0 = omp_get_wtime();
  for (int r = 0; r < reps; ++r) {
    double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (long blk = 0; blk < (count + 65535) / 65536; ++blk) {
      const long lo = blk * 65536, hi = lo + 65536 < count ? lo + 65536 : count;
      /* eight scalar accumulators under `omp simd reduction`: each becomes a vector register of its own, i.e. eight independent
       * dependency chains per thread (one chain -- or accumulators the compiler keeps in memory -- is latency-bound) */
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
      const long n8 = (hi - lo) / 8;  /* eight contiguous runs of n8 floats each: eight streams, no shuffles */
      const float* p = a + lo;
#pragma omp simd reduction(+ : s0, s1, s2, s3, s4, s5, s6, s7)
      for (long k = 0; k < n8; ++k) {
        s0 += p[k];
        s1 += p[n8 + k];
        s2 += p[2 * n8 + k];
        s3 += p[3 * n8 + k];
        s4 += p[4 * n8 + k];
        s5 += p[5 * n8 + k];
        s6 += p[6 * n8 + k];
        s7 += p[7 * n8 + k];
      }
      float tail = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
      for (long i = lo + 8 * n8; i < hi; ++i) tail += a[i];
      s += tail;
    }
    total += s;
  }
  *seconds_out = omp_get_wtime() - t0;
  return total;
}

/* The same pass in gemv_t's own access pattern: every thread walks its chunk as ONE contiguous stream, 16384-float runs summed
 * with a single `omp simd` reduction each (the compiler's unrolled vector accumulators) -- on some hosts this pattern streams
 * faster than eight interleaved streams per thread, on others slower; the CPU leg reports the better of the two as the host's
 * read rate. */
double cpu_twin_read_pass_seq(const float* a, long count, int reps, double* seconds_out) {
  double total = 0.0;
  const double t0 = omp_get_wtime();
  for (int r = 0; r < reps; ++r) {
    double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (long blk = 0; blk < (count + 16383) / 16384; ++blk) {
      const long lo = blk * 16384, hi = lo + 16384 < count ? lo + 16384 : count;
      float sum = 0.f;
#pragma omp simd reduction(+ : sum)
      for (long i = lo; i < hi; ++i) sum += a[i];
      s += sum;
    }
    total += s;
  }
  *seconds_out = omp_get_wtime() - t0;
  return total;
}

/* First touch of a buffer that a single-threaded copy is about to fill (the 64 GiB download of the device matrix): every
 * thread writes the pages of the static chunk it will later READ (gemv_t's column blocks, the read pass), so that on a
 * multi-socket host the pages live next to the cores that stream them instead of all on the copying thread's node. */
void cpu_twin_first_touch(float* a, long count) {
#pragma omp parallel for schedule(static)
  for (long blk = 0; blk < (count + 65535) / 65536; ++blk) {
    const long lo = blk * 65536, hi = lo + 65536 < count ? lo + 65536 : count;
    for (long i = lo; i < hi; i += 1024) a[i] = 0.0f; /* one store per 4 KiB page */
  }
}

int cpu_twin_threads(void) { return omp_get_max_threads(); }
void cpu_twin_set_threads(int t) { omp_set_num_threads(t); }

/* init (fast_forward_backward.jl:73-97) + `steps` iterations (:131-142) with gamma = 1 / Lf; z_out = state.z (n floats);
 * seconds_out = wall time of the `steps` iterations only; returns f(x) of the last state */
float cpu_twin_ffb(const float* A, long m, long n, long ld, const float* b, float lam, float Lf, int steps, float* z_out,
                   double* seconds_out) {
  float* x = calloc(n, sizeof(float));      /* x0 = 0 (benchmarks.jl:58) */
  float* grad = malloc(n * sizeof(float));
  float* y = malloc(n * sizeof(float));
  float* z = malloc(n * sizeof(float));
  float* z_prev = malloc(n * sizeof(float));
  float* res = malloc(m * sizeof(float));
  const float gamma = 1.0f / Lf;
  float f_x = value_and_gradient(A, m, n, ld, b, x, res, grad);            /* :75 */
  const float gl = gamma * lam;
  for (long j = 0; j < n; ++j) {                                            /* :79-80 */
    y[j] = x[j] - gamma * grad[j];
    z[j] = y[j] <= -gl ? y[j] + gl : (y[j] >= gl ? y[j] - gl : 0.0f);
    z_prev[j] = x[j];                                                       /* :69 z_prev = copy(x) */
  }
  float s_step = -1.0f, s_theta = -1.0f;                                    /* AdaptiveNesterovSequence(mf = 0) */
  const double t0 = omp_get_wtime();
  for (int k = 0; k < steps; ++k) {
    /* nesterov.jl:89-103 */
    if (s_step < 0.0f) {
      s_step = gamma;
      s_theta = 1.0f;
    }
    const float bq = s_theta * s_theta / s_step;
    const float delta = bq * bq + 4.0f * (s_theta * s_theta) / (s_step * gamma);
    const float theta = gamma * (-bq + sqrtf(delta)) / 2.0f;
    const float beta = gamma * s_theta * (1.0f - s_theta) / (s_step * theta + gamma * s_theta * s_theta);
    s_step = gamma;
    s_theta = theta;
    for (long j = 0; j < n; ++j) x[j] = z[j] + beta * (z[j] - z_prev[j]);   /* :135 */
    float* tmp = z_prev;                                                    /* :136 */
    z_prev = z;
    z = tmp;
    f_x = value_and_gradient(A, m, n, ld, b, x, res, grad);                 /* :138-139 */
    for (long j = 0; j < n; ++j) {                                          /* :140-142 */
      y[j] = x[j] - gamma * grad[j];
      z[j] = y[j] <= -gl ? y[j] + gl : (y[j] >= gl ? y[j] - gl : 0.0f);
    }
  }
  if (seconds_out) *seconds_out = omp_get_wtime() - t0;
  memcpy(z_out, z, n * sizeof(float));
  free(x), free(grad), free(y), free(z), free(z_prev), free(res);
  return f_x;
}
