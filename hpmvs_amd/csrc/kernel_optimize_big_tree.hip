// kernel_optimize_big_tree.hip -- kernel_optimize_big.hip (64 slots per wavefront, 3 regions) with the throughput-mode
// window sums (see kernel_optimize_tree.hip).
#ifndef HPMVS_BIG_P
#define HPMVS_BIG_P 64
#endif
#define BQ_P HPMVS_BIG_P
#ifndef BQ_COLD_HI
#define BQ_COLD_HI 145
#endif
#ifndef HPMVS_COLD_GLOBAL
#define HPMVS_COLD_GLOBAL 1   // bmat, zmat, vlag, wsuma of the optimiser states in global memory
#endif
#ifndef HPMVS_WG_REGIONS
#define HPMVS_WG_REGIONS 3   /* (2 regions of 84 KB without BQ_COLD_GLOBAL: pass -DHPMVS_COLD_GLOBAL=0 -DHPMVS_WG_REGIONS=2 -DHPMVS_BIG_P=58) */
#endif
#define HPMVS_FAST_SUMS 1
// the lone-request shortcuts of the photometric service (photometric.hpp) stay out of this build: the extra code costs
// its 64-slot rounds more registers (spills) than its short ramp-down gets back (round 4: C2 +1 %)
#ifndef HPMVS_SPLIT_PREP
#define HPMVS_SPLIT_PREP 0
#endif
#ifndef HPMVS_SINGLE_FAST
#define HPMVS_SINGLE_FAST 0
#endif
#define HPMVS_VARIANT_SUFFIX _big_tree
#include "kernel_optimize.hip"
