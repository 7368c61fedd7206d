// kernel_optimize_tree.hip -- the refinement kernel (29 slots per wavefront) with the THROUGHPUT-mode window sums of
// photometric.hpp (HPMVS_FAST_SUMS: every lane sums its row, the row sums are added in row order).  Opt-in with
// HPMVS_SUMS=tree (kernel_launch.hip); not bit-identical to the oracle -- the default build and every parity test use the
// exact-order chains.  What the mode costs in agreement and buys in time: profiles/r04/sums_modes.json, DESIGN section 5.
#define HPMVS_FAST_SUMS 1
#define HPMVS_VARIANT_SUFFIX _tree
#include "kernel_optimize.hip"
