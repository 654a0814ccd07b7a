// Explicit instantiation: double, LDS-staged variant MODE 2 (1: relation slice in LDS, 2: relation + input slices in LDS).
#include "rspmm_kernels.hpp"
namespace ultra {
ULTRA_DEFINE_VARIANT(double, 4, 2)
}  // namespace ultra
