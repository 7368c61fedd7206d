// kernel_optimize.hip -- placeholder until the refinement kernel lands.
#include <hip/hip_runtime.h>
#include "launch.h"
namespace hpmvs {
void launch_optimize(const DevScene&, const DevOptions&, const DevBatch&, int32_t*, int, hipStream_t) {}
}
