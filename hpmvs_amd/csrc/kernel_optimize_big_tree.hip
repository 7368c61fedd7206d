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
// The parking traffic of this build (a wavefront's 52 KB of optimiser states out to its buffer and back in, once per round) carries the
// NON-TEMPORAL hint both ways: a line comes back a whole round (~250 us, hundreds of MB of other traffic) later, so keeping it in L2
// only evicts what IS reused within microseconds -- the spilled registers, the cold matrices, the result rows.  Round 6, same gpurun
// call: C2 199.8 -> 192.0 ms, 250 k seeds 59.5 -> 57.7, 2e6 seeds of the 200-view scene 369.5 -> 357.1; either direction alone 198-199;
// the hint on the cold matrices as well: 195-202 (they ARE reused).  The 29-slot build keeps plain accesses (C1 22.5 -> 23.4 ms with the
// hint, a 125 k-seed shard 34.7 -> 35.3): profiles/r06/park_nontemporal.txt.
#ifndef HPMVS_PARK_AUX
#define HPMVS_PARK_AUX 2          /* cache-policy bits of global_load_lds: 2 = nt */
#endif
#ifndef HPMVS_PARK_NT_STORE
#define HPMVS_PARK_NT_STORE 1
#endif
#define HPMVS_VARIANT_SUFFIX _big_tree
#include "kernel_optimize.hip"
