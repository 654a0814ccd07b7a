// Explicit instantiation: float, LDS-staged variant MODE 1 (1: relation slice in LDS, 2: relation + input slices in LDS).
#include "rspmm_kernels.hpp"
namespace ultra {
ULTRA_DEFINE_VARIANT(float, 4, 1)
}  // namespace ultra
