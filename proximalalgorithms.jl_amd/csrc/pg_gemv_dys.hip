// Davis-Yin epilogue of the single-sweep kernel (gemv_tn_kernel<..., MODE = 1>) and its entry point, in a translation
// unit of their own so that the forward-backward instantiations of pg_gemv.hip (the headline kernel) are compiled
// independently of them.  Reference: davis_yin.jl:73-83 (outside SURVEY section 8; kept for the callers that exist).
#include "pg_gemv_tn.h"

using namespace pgtn;

namespace {

template <typename T>
pg_status launch_tn_dys(pg_mat* A, TNArgs<T>& a, int* blocks_out) {
  const int nrg = a.nrg;
  // the default geometries of launch_tn (pg_gemv.hip), no tuner variants
  int W = nrg <= 2 ? 1 : nrg <= 8 ? 2 : (nrg <= 16 || (nrg > 32 && nrg <= 64)) ? 4 : 8;
  int U = 1;
  while (U * W < nrg) U *= 2;
  int C = U >= 8 ? 32 / U : 16 / U;
  if (W == 8 && U == 16) C = 1;
  if (W == 8 && U == 4) C = 8;
  if (W <= 2) C = (W == 2 && U == 4) ? 4 : (U == 1 ? 16 : 8);
  if (sizeof(T) == 8 && U == 1 && C > 16) C = 16;
#define PG_TN_DYS(UU, CC, WW) \
  if (U == UU && C == CC && W == WW) return launch_tn_ucw<T, UU, CC, WW, 1>(A, a, blocks_out)
    PG_TN_DYS(16, 2, 4);
    PG_TN_DYS(16, 1, 8);
    PG_TN_DYS(4, 8, 8);
    PG_TN_DYS(4, 4, 4);
    PG_TN_DYS(4, 4, 2);
    PG_TN_DYS(2, 8, 2);
    PG_TN_DYS(2, 8, 1);
    PG_TN_DYS(1, 16, 1);
#undef PG_TN_DYS
    pg_set_error("no Davis-Yin sweep for U=%d C=%d WAVES=%d (launch-geometry overrides are not available in this mode)", U, C, W);
    return PG_ERR_UNSUPPORTED;
}

// One Davis-Yin iteration (davis_yin.jl:73-83) for f = loss o A in ONE read of A: given r = grad loss(A xg), the sweep forms
// grad = A' r, z_half, xh = prox_{gamma h}, res, z+ = z + relax res, the NEXT xg+ = prox_{gamma g}(z+) and A xg+.
template <typename T>
pg_status mat_fused_dys_t(pg_mat* A, const T* r, const T* xg, const T* z, double gamma, double relax, int g_kind, double g_p0,
                          double g_p1, int h_kind, double h_p0, double h_p1, T* grad, T* z_half, T* xh, T* res, T* z_next,
                          T* xg_next, T* A_xg_next) {
  pg_ctx* c = A->ctx;
  if (pg_row_sharded(c) || pg_col_sharded(c) || !tn_single_wg_supported<T>(A)) {
    pg_set_error("the single-sweep pass needs an unsharded operator with at most %d rows", (int)(128 * 1024 / sizeof(T)));
    return PG_ERR_UNSUPPORTED;
  }
  if (A->rpad == nullptr) {
    PG_HIP(hipMalloc(&A->rpad, (size_t)A->ld * sizeof(T)));
    PG_HIP(hipMemsetAsync(A->rpad, 0, (size_t)A->ld * sizeof(T), c->stream));
  }
  PG_HIP(hipMemcpyAsync(A->rpad, r, (size_t)A->m * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
  const T gm = (T)gamma;
  auto scaled = [&](int kind, double p0) -> T {
    if (kind == PG_G_NORML1) return (T)(gm * (T)p0);
    if (kind == PG_G_SQRNORML2) return T(1) / (T(1) + (T)p0 * gm);
    return (T)p0;
  };
  TNArgs<T> a;
  a.A = (const T*)A->data;
  a.ld = A->ld;
  a.n = A->n;
  a.m = A->m;
  a.nrg = (int)(A->ld / (1024 / (int64_t)sizeof(T)));
  a.r = (const T*)A->rpad;
  a.x = xg;
  a.z_old = z;
  a.gamma = gm;
  a.beta = T(0);
  a.p0 = scaled(g_kind, g_p0);
  a.p1 = (T)g_p1;
  a.lam_ls = T(1);
  a.g_kind = g_kind;
  a.gscale = 0.0;
  a.h_kind = h_kind;
  a.h_p0 = scaled(h_kind, h_p0);
  a.h_p1 = (T)h_p1;
  a.relax = (T)relax;
  a.g_out = grad;
  a.y = z_half;
  a.xh_out = xh;
  a.res = res;
  a.v_out = z_next;
  a.z_new = xg_next;
  a.partials = nullptr;
  a.red_partials = c->red_partials;
  a.red_counter = c->red_counter;
  a.scal_out = c->dscal + PG_S_GZ;
  a.line_cols = env_int("PG_TN_LINE_COLS", 0) > 0 ? env_int("PG_TN_LINE_COLS", 0) : 32;  // experiments: 1 = dealt one by one (rounds 1-2)
  int blocks = 0;
  PG_TRY((launch_tn_dys<T>(A, a, &blocks)));
  int64_t fb = (A->ld + 63) / 64;
  if (fb > 1024) fb = 1024;
  pg_prof_scope prof(c, PG_K_GEMV_N_FINISH);
  hipLaunchKernelGGL((gemv_n_finish_kernel<T, false>), dim3((unsigned)fb), dim3(1024), 0, c->stream,
                     (const T*)A->partials, A->ld, A->m, blocks, (const T*)nullptr, A_xg_next, A->m, 0.0, (double*)nullptr,
                     (unsigned*)nullptr, (double*)nullptr, (T*)nullptr, ColPack<T>{});
  PG_LAUNCH_CHECK();
  return PG_OK;
}

}  // namespace

extern "C" {

pg_status pg_mat_fused_dys(pg_mat* A, const void* r, const void* xg, const void* z, double gamma, double relax,
                           int32_t g_kind, double g_p0, double g_p1, int32_t h_kind, double h_p0, double h_p1, void* grad,
                           void* z_half, void* xh, void* res, void* z_next, void* xg_next, void* A_xg_next,
                           double* scalars_out) {
  PG_REQUIRE(A != nullptr, "matrix is null");
  PG_REQUIRE(r && xg && z && grad && z_half && xh && res && z_next && xg_next && A_xg_next, "null vector");
  PG_REQUIRE(g_kind >= PG_G_ZERO && g_kind <= PG_G_SQRNORML2 && h_kind >= PG_G_ZERO && h_kind <= PG_G_SQRNORML2, "unknown prox kind");
  PG_REQUIRE(gamma > 0, "gamma must be positive");
  PG_TRY(A->dtype == PG_F32
             ? mat_fused_dys_t<float>(A, (const float*)r, (const float*)xg, (const float*)z, gamma, relax, g_kind, g_p0, g_p1,
                                      h_kind, h_p0, h_p1, (float*)grad, (float*)z_half, (float*)xh, (float*)res,
                                      (float*)z_next, (float*)xg_next, (float*)A_xg_next)
             : mat_fused_dys_t<double>(A, (const double*)r, (const double*)xg, (const double*)z, gamma, relax, g_kind, g_p0,
                                       g_p1, h_kind, h_p0, h_p1, (double*)grad, (double*)z_half, (double*)xh, (double*)res,
                                       (double*)z_next, (double*)xg_next, (double*)A_xg_next));
  if (scalars_out) {
    PG_TRY(pg_read_scalars(A->ctx, PG_S_GZ, 4));
    for (int k = 0; k < 4; ++k) scalars_out[k] = A->ctx->hscal[PG_S_GZ + k];
  }
  return PG_OK;
}

}  // extern "C"
