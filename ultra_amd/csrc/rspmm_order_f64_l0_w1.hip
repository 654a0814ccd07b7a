// Explicit instantiations of the reference-order kernels (rspmm_order_kernels.hpp): one translation unit per
// (element type, relation slice in LDS or not, per-call edge weights or not) so that hipcc compiles them in parallel.
#include "rspmm_order_kernels.hpp"

namespace ultra {
ULTRA_DEFINE_ORDER_VARIANT(double, false, true)
}  // namespace ultra
