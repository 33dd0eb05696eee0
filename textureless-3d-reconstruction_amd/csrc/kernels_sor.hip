// kernels_sor.hip -- f1: statistical outlier removal (placeholder until the grid-kNN kernels land).
#include "tl3d_internal.h"
namespace tl3d {
int sor_run(tl3d_ctx *, const float *, long long, int, double, double, uint8_t *, long long *) {
    return set_err(TL3D_E_STATE, "statistical outlier filter not built in this library version");
}
}  // namespace tl3d
