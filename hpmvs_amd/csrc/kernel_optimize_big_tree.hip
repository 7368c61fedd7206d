// kernel_optimize_big_tree.hip -- kernel_optimize_big.hip (58 slots per wavefront, 2 regions) with the throughput-mode
// window sums (see kernel_optimize_tree.hip).
#ifndef HPMVS_BIG_P
#define HPMVS_BIG_P 60
#endif
#define BQ_P HPMVS_BIG_P
#ifndef HPMVS_COLD_GLOBAL
#define HPMVS_COLD_GLOBAL 1   // bmat, zmat, vlag of the optimiser states in global memory: 50 KB regions, three per CU
#endif
#ifndef HPMVS_WG_REGIONS
#define HPMVS_WG_REGIONS 3   /* 58 slots x 876 B = 50 KiB per region (2 of 79 KB without BQ_COLD_GLOBAL: pass -DHPMVS_COLD_GLOBAL=0 -DHPMVS_WG_REGIONS=2) */
#endif
#define HPMVS_FAST_SUMS 1
#define HPMVS_VARIANT_SUFFIX _big_tree
#include "kernel_optimize.hip"
