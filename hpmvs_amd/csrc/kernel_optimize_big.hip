// kernel_optimize_big.hip -- the refinement kernel once more, for batches that fill the machine many times over:
// 58 patch slots per wavefront and two 79 KB state regions per CU instead of 29 slots and four regions.  The float64
// optimiser blocks cost about the same per call whether 29 or 58 lanes run them, so a patch's share of them halves;
// the price is rounds twice as long (a patch is ~90 sequential rounds, up to 1000), i.e. a longer ramp-down at the
// end of a launch.  Measured on the C2 scene (1 MI355X): 60 k patches 23.5 ms (29 slots) against 28.2 ms (58 slots),
// 250 k 68.4 / 70.5, 500 k 129.5 / 127.4, 1 M 241.8 / 229.6; launch_optimize() in kernel_launch.hip takes this build
// from 1600 patches per CU upwards.
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
#define HPMVS_VARIANT_SUFFIX _big
#include "kernel_optimize.hip"
