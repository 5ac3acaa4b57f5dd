// lrf_env.h — the environment variables the library reads.
//
// Service knobs (every build): test hooks that move a size threshold so that small test inputs reach the large-call
// kernels, and the pipe's schedule overrides.  Read once per process.
//   LRF_PERSIST              0: never k_bcd_p; 1: k_bcd_p from LRF_BCDW_MIN_BLOCKS blocks on       (run_bcd)
//   LRF_FAMILY_SPLIT_BLOCKS  blocks from which a call's rank families run on kernels of their own  (plan_runs)
//   LRF_BCDW16_MIN_BLOCKS / LRF_BCDW32_MIN_BLOCKS  blocks from which the wave kernels of ranks 9..16 / 17..32 run
//   LRF_DEBUG_INIT_SWEEPS    stops k_init after a stage / forces the Sturm replacement loop       (tests/test_hip_parity.py)
//   LRF_PIPE_BULK / LRF_PIPE_TAIL  the pipe's piece sizes                                          (tests/test_pipeline.py)
//   LRF_FUSED_GRAM_MIN_CHUNKS  from how many luma Gram chunks on k_planes16_gram forms the patch matrices (tests/test_fused_gram.py)
//
// Developer comparison switches (LRF_PLANES_NO_TILED, LRF_BCD_WG, LRF_NO_BCDW32, LRF_ANY_*, ...: the kernels a later one
// replaced, kept for A/B timing) exist only in a -DLRF_DEV build (`make -C lrf_amd/csrc dev` -> liblrf_hip_dev.so): in
// the shipped library dev_flag() is the constant false and the replaced paths are dead code the compiler drops.
#ifndef LRF_ENV_H
#define LRF_ENV_H
#include <stdlib.h>

inline long env_long(const char* name, long dflt)
{
    const char* e = getenv(name);
    return (e && *e) ? atol(e) : dflt;
}
#ifdef LRF_DEV
inline bool dev_flag(const char* name)
{
    const char* e = getenv(name);
    return e && e[0] == '1';
}
inline long dev_long(const char* name, long dflt) { return env_long(name, dflt); }
#else
constexpr bool dev_flag(const char*) { return false; }
constexpr long dev_long(const char*, long dflt) { return dflt; }
#endif
#endif
