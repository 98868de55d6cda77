#include <type_traits>
#include "pg_gemv_tn.h"
namespace pgtn {
namespace {
#include "pg_gemv_tnt.h"
#include "pg_gemv_tnp1.h"
template __global__ void gemv_tnp1_kernel<ONE_T, ONE_U, ONE_C, ONE_LAG, ONE_PF, ONE_LAGR, false, ONE_PAIR, ONE_AHEAD>(TNArgs<ONE_T>);
}
}
