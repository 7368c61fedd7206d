// prof.h -- region timers for -DHPMVS_PROFILE_PHASES builds (tools/phase_profile.py); no-ops otherwise.
// Accumulated per wavefront in LDS by lane 0 (no atomics inside the kernel body), flushed to the
// global table once when the wave exits.
#pragma once
#if defined(HPMVS_PROFILE_PHASES) && defined(__HIPCC__)
#include <hip/hip_runtime.h>
static __device__ unsigned long long g_bqprof[64];
static __shared__ unsigned long long s_prof[64];
#endif
