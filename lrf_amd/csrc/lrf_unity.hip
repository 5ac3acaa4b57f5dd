// lrf_unity.hip — the whole library as ONE translation unit, for the variant builds of the tools and tests (stamps, ablations,
// failure injection: `make unity VARIANT=... DEFS=...`).  The shipped library is built from its seven units separately.
#include "lrf_ctx.hip"
#include "lrf_pipe.hip"
#include "lrf_encode8.hip"
#include "lrf_planes_gram.hip"
#include "lrf_bcd32.hip"
#include "lrf_bcd_persist.hip"
#include "lrf_any.hip"
