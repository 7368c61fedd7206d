// kernel_launch.hip -- which build of the refinement kernel a batch launch gets (kernel_optimize.hip is compiled
// twice: see kernel_optimize_big.hip), and the workspace that fits both.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "launch.h"

namespace hpmvs {

size_t optimize_workspace_bytes_std(int n_cus);
size_t optimize_workspace_bytes_big(int n_cus);
void launch_optimize_std(const DevScene& sc, const DevOptions& o, const DevBatch& b, int32_t* workspace, int n_cus, hipStream_t st);
void launch_optimize_big(const DevScene& sc, const DevOptions& o, const DevBatch& b, int32_t* workspace, int n_cus, hipStream_t st);

size_t optimize_workspace_bytes(int n_cus) {
    const size_t a = optimize_workspace_bytes_std(n_cus), b = optimize_workspace_bytes_big(n_cus);
    return a > b ? a : b;
}

void launch_optimize(const DevScene& sc, const DevOptions& o, const DevBatch& b, int32_t* workspace, int n_cus,
                     hipStream_t st) {
    // HPMVS_SLOTS=29 / 58 forces one build (measurements); otherwise the 58-slot build from 1600 patches per CU
    // (409 600 on an MI355X), where its lower cost per patch has paid for its longer ramp-down
    static const int forced = [] { const char* e = getenv("HPMVS_SLOTS"); return e ? atoi(e) : 0; }();
    const bool big = forced == 58 || (forced != 29 && (long long)b.n >= 1600ll * n_cus);
    if (big) launch_optimize_big(sc, o, b, workspace, n_cus, st);
    else launch_optimize_std(sc, o, b, workspace, n_cus, st);
}

}  // namespace hpmvs
