// kernel_optimize_big.hip -- the refinement kernel once more, for batches that fill the machine many times over:
// 64 patch slots per wavefront (every lane owns a patch) and three 52 KB state regions per CU instead of 29 slots and
// four regions.  The float64 optimiser blocks cost about the same per call whether 29 or 64 lanes run them, so a patch's
// share of them more than halves; the price is rounds twice as long (a patch is ~90 sequential rounds, up to 1000), i.e.
// a longer ramp-down at the end of a launch.  64 slots fit three regions because the optimiser's big matrices AND its
// wsuma/ptsid vector (bmat, zmat, vlag, wsuma: state indices [77, 145), 68 of 164 doubles) live in global memory behind
// buffer loads (BQ_COLD_GLOBAL, bobyqa3.hpp): 64 x (96 x 8 + 52) B = 52 480 B per region.
// Measured on the C2 scene (1 MI355X, round 4): 64 slots 213.6 ms against 219.0 ms for 60 slots with wsuma in LDS and
// 241 ms for the 29-slot build; launch_optimize() in kernel_launch.hip takes this build from 768 patches per CU upwards (round 5; 480 in round 4).
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
// Of the lone-request shortcuts of the photometric service (photometric.hpp) this build takes the ONE-REQUEST batch (a batch that
// is one request hands its images to the groups directly: the ramp-down of a launch) but not the split preparation of the job
// sets: round 5, after the kernel lost ~10 spilled registers elsewhere, C2 202.5 ms against 203.2 and 250 k patches 59.7 ms
// against 60.4-61.3 with it, 204.7 / 60.1 with both (round 4: either one +1 %, both +17 %; profiles/r05/ab_variants.txt)
#ifndef HPMVS_SPLIT_PREP
#define HPMVS_SPLIT_PREP 0
#endif
#ifndef HPMVS_SINGLE_FAST
#define HPMVS_SINGLE_FAST 1
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
#define HPMVS_VARIANT_SUFFIX _big
#include "kernel_optimize.hip"
