// fp32 instantiation of the stencil (one site per thread): spinor32 fields, float2 gauge copy.  See hopping.hip / hopping_impl.inc.
#include "hopping_common.h"

#define HOP_CTX_GAUGE(ctx) ((ctx)->gauge32)
#define HOP_CTX_GAUGE_READY(ctx) ((ctx)->gauge32_set)
#define HOP_CTX_OCC(ctx) ((ctx)->opt_occ32)
namespace hop32 {
TMHIP_SCALAR_COMPLEX_OPS(v2f, float)
#define HOP_SITES 1
#include "hopping_impl.inc"
#undef HOP_SITES
}  // namespace hop32

#undef HOP_CTX_OCC
#undef HOP_CTX_GAUGE
#undef HOP_CTX_GAUGE_READY
