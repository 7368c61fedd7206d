// kernel_optimize_big_tree.hip -- kernel_optimize_big.hip (58 slots per wavefront, 2 regions) with the throughput-mode
// window sums (see kernel_optimize_tree.hip).
#define BQ_P 58
#define HPMVS_WG_REGIONS 2
#define HPMVS_FAST_SUMS 1
#define HPMVS_VARIANT_SUFFIX _big_tree
#include "kernel_optimize.hip"
