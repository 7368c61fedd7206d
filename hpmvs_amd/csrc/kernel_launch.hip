// kernel_launch.hip -- which build of the refinement kernel a batch launch gets (kernel_optimize.hip is compiled four
// times: see kernel_optimize_big.hip and kernel_optimize_tree.hip), and the workspace that fits all of them.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "launch.h"

namespace hpmvs {

#define HPMVS_DECLARE(sfx)                                                                                             \
    size_t optimize_workspace_bytes##sfx(int n_cus);                                                                   \
    void launch_optimize##sfx(const DevScene& sc, const DevOptions& o, const DevBatch& b, int32_t* workspace, int n_cus, hipStream_t st);
HPMVS_DECLARE(_std)
HPMVS_DECLARE(_big)
HPMVS_DECLARE(_tree)
HPMVS_DECLARE(_big_tree)
#undef HPMVS_DECLARE

size_t optimize_workspace_bytes(int n_cus) {
    size_t m = optimize_workspace_bytes_std(n_cus);
    const size_t o[3] = {optimize_workspace_bytes_big(n_cus), optimize_workspace_bytes_tree(n_cus), optimize_workspace_bytes_big_tree(n_cus)};
    for (size_t v : o) m = v > m ? v : m;
    return m;
}

// HPMVS_SUMS=tree: the throughput-mode window sums (photometric.hpp, HPMVS_FAST_SUMS).  Anything else, or nothing: the
// reference's sequential order (parity mode, the default; what `exact` spells out).
int optimize_sums_mode() {
    static const int mode = [] { const char* e = getenv("HPMVS_SUMS"); return (e && strcmp(e, "tree") == 0) ? 1 : 0; }();
    return mode;
}

void launch_optimize_wide(const DevScene& sc, const DevOptions& o, const DevBatch& b, int32_t* workspace, int32_t* counter, size_t list_cap,
                          int n_cus, hipStream_t st);
static void launch_optimize_fast(const DevScene& sc, const DevOptions& o, const DevBatch& b, int32_t* workspace, int n_cus, hipStream_t st);

// A refinement launch = the batch kernel (lists up to 64 ids) + the wide kernel behind it, which redoes the patches the batch
// kernel marked ok == 2 (list longer than 64 ids; kernel_wide.hip).  With nothing marked the wide kernel is a scan of ok[].
void launch_optimize(const DevScene& sc, const DevOptions& o, const DevBatch& b, int32_t* workspace, int n_cus,
                     hipStream_t st) {
    if (b.n <= 0) return;
    launch_optimize_fast(sc, o, b, workspace, n_cus, st);
    // (its three counters: words of the zeroed counter block nobody else uses; its list of marked patches: the rest of the
    //  workspace, which the batch kernel is through with)
    launch_optimize_wide(sc, o, b, workspace, workspace + 240, (optimize_workspace_bytes(n_cus) - 1024) / sizeof(int32_t), n_cus, st);
}

static void launch_optimize_fast(const DevScene& sc, const DevOptions& o, const DevBatch& b, int32_t* workspace, int n_cus,
                                 hipStream_t st) {
    // HPMVS_SLOTS=29 / 64 forces one build (measurements; 58 and 60, the big build's earlier slot counts, mean it too);
    // otherwise the 64-slot build from 768 patches per CU (196 608 on an MI355X), where its lower cost per patch has paid
    // for its longer ramp-down.  Round 5, 29 / 64 slots, same gpurun call (profiles/r05/slots_crossover.txt): C2 scene 100 k
    // patches 29.6 / 32.3 ms, 150 k 42.4 / 43.2, 250 k 61.3 / 59.6; C1 scene (half the grabs per patch) 150 k 27.5 / 32.4,
    // 250 k 40.4 / 43.2 -- the crossover moved up from round 4's 480 per CU (the list services and the photometric front end
    // got cheaper, the optimiser's blocks did not) and depends on the scene; 768 is the C2 scene's.
    static const int forced = [] { const char* e = getenv("HPMVS_SLOTS"); return e ? atoi(e) : 0; }();
    const bool big = forced == 58 || forced == 60 || forced == 64 || (forced != 29 && (long long)b.n >= 768ll * n_cus);
    if (optimize_sums_mode() == 1) {
        if (big) launch_optimize_big_tree(sc, o, b, workspace, n_cus, st);
        else launch_optimize_tree(sc, o, b, workspace, n_cus, st);
        return;
    }
    if (big) launch_optimize_big(sc, o, b, workspace, n_cus, st);
    else launch_optimize_std(sc, o, b, workspace, n_cus, st);
}

}  // namespace hpmvs
