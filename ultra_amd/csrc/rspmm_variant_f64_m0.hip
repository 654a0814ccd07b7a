// Explicit instantiation: double, relation/input read through L2 (MODE_GLOBAL); VEC 4 and the VEC 1 fallback.
#include "rspmm_kernels.hpp"
namespace ultra {
ULTRA_DEFINE_VARIANT(double, 4, 0)
ULTRA_DEFINE_VARIANT(double, 1, 0)
}  // namespace ultra
