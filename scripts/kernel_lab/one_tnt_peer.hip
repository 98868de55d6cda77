// compile ONE row-team instantiation for a look at its registers / ISA:  -DONE_U=8 -DONE_C=4 -DONE_LAG=1 -DONE_PF=1 -DONE_LAGR=1 -DONE_W=1
#include <type_traits>
#include "pg_gemv_tn.h"
namespace pgtn {
namespace {
#include "pg_gemv_tnt.h"
template __global__ void gemv_tnt_kernel<float, ONE_U, ONE_C, ONE_W, ONE_LAG, ONE_PF, true, ONE_LAGR, false, ONE_AHEAD>(TNArgs<float>);
}
}
