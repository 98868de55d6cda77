// Mid-length columns (33 .. 128 row groups: 8448 .. 32768 rows in Float32): instantiations of the one-workgroup sweep
// (gemv_tn_kernel, pg_gemv_tn.h) for a row-group count per wave U that fits the column exactly (9 .. 16 rather than the
// next power of two) and with the register-tile count chosen per geometry, compiled in their own translation unit.
// scripts/tile_pattern.hip (profiles/r3_mid_columns_counters.md) is the measurement behind them: the load pattern of these
// geometries alone reaches 7.2-7.35 TB/s with the dot / barrier / accumulate structure of the sweep, so what the
// power-of-two geometries lose at these lengths is repeated or wasted loads, not the memory system.
#include "pg_gemv_tn.h"

namespace pgtn {

namespace {

template <typename T, int U, int C, int WAVES, bool DB>
pg_status launch_mid(pg_mat* A, TNArgs<T>& a, int* blocks_out, int bpc) {
  pg_ctx* c = A->ctx;
  const int64_t ncg = (A->n + C - 1) / C;
  int64_t blocks = (int64_t)c->num_cu * bpc;
  if (env_int("PG_TN_BLOCKS", 0) > 0) blocks = env_int("PG_TN_BLOCKS", 0);
  if (blocks > ncg) blocks = ncg;
  if (blocks > PG_RED_MAX_BLOCKS) blocks = PG_RED_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  PG_TRY(ensure_partials(A, (int)blocks));
  a.partials = (T*)A->partials;
  *blocks_out = (int)blocks;
  pg_prof_scope prof(c, PG_K_GEMV_TN);
  hipLaunchKernelGGL((gemv_tn_kernel<T, U, C, WAVES, DB, 0>), dim3((unsigned)blocks), dim3(WAVES * 64), 0, c->stream, a);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

}  // namespace

// U row groups per wave, C columns per step, W waves, db register tiles - 1; PG_ERR_UNSUPPORTED when not instantiated
template <typename T>
pg_status launch_tn_mid(pg_mat* A, TNArgs<T>& a, int* blocks_out, int U, int C, int W, int db, int bpc) {
#define PG_MID(UU, CC, WW, DD) \
  if (U == UU && C == CC && W == WW && db == DD) return launch_mid<T, UU, CC, WW, (DD != 0)>(A, a, blocks_out, bpc)
  // four waves, exact U (33 .. 64 row groups)
  PG_MID(9, 2, 4, 1); PG_MID(10, 2, 4, 1); PG_MID(11, 2, 4, 1); PG_MID(12, 2, 4, 1);
  PG_MID(13, 2, 4, 1); PG_MID(14, 2, 4, 1); PG_MID(15, 2, 4, 1);
  PG_MID(9, 2, 4, 0); PG_MID(10, 2, 4, 0); PG_MID(11, 2, 4, 0); PG_MID(12, 2, 4, 0);
  PG_MID(13, 2, 4, 0); PG_MID(14, 2, 4, 0); PG_MID(15, 2, 4, 0); PG_MID(16, 2, 4, 0);
  PG_MID(10, 4, 4, 0); PG_MID(12, 4, 4, 0);
  // 17 .. 32 row groups on four waves (U = 5 .. 8)
  PG_MID(8, 4, 4, 0); PG_MID(8, 2, 4, 0); PG_MID(8, 4, 4, 1); PG_MID(8, 2, 4, 1);
  PG_MID(5, 4, 4, 0); PG_MID(6, 4, 4, 0); PG_MID(7, 4, 4, 0);
  PG_MID(5, 4, 4, 1); PG_MID(6, 4, 4, 1); PG_MID(7, 4, 4, 1);
  // eight waves, exact U (65 .. 128 row groups)
  PG_MID(9, 2, 8, 0); PG_MID(10, 2, 8, 0); PG_MID(11, 2, 8, 0); PG_MID(12, 2, 8, 0);
  PG_MID(13, 2, 8, 0); PG_MID(14, 2, 8, 0); PG_MID(15, 2, 8, 0); PG_MID(16, 2, 8, 0);
  PG_MID(9, 1, 8, 0); PG_MID(10, 1, 8, 0); PG_MID(11, 1, 8, 0); PG_MID(12, 1, 8, 0);
  PG_MID(13, 1, 8, 0); PG_MID(14, 1, 8, 0); PG_MID(15, 1, 8, 0);
  PG_MID(9, 1, 8, 1); PG_MID(12, 1, 8, 1); PG_MID(16, 1, 8, 1);
#undef PG_MID
  pg_set_error("no mid-column gemv_tn instantiation for U=%d C=%d WAVES=%d tiles=%d", U, C, W, db + 1);
  return PG_ERR_UNSUPPORTED;
}

template pg_status launch_tn_mid<float>(pg_mat*, TNArgs<float>&, int*, int, int, int, int, int);
template pg_status launch_tn_mid<double>(pg_mat*, TNArgs<double>&, int*, int, int, int, int, int);

}  // namespace pgtn
