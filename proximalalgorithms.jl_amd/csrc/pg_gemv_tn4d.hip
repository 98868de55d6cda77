// The Float64 instantiations of the row-team sweep (pg_gemv_tn4.hip), compiled as a translation unit of their own so that the two
// element types build side by side: the file is pg_gemv_tn4.hip again with everything that is not a template left out.
#define PG_TN4_FLOAT64_UNIT 1
#include "pg_gemv_tn4.hip"
