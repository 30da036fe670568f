// macx_chain_fwd.hip -- the forward chain kernels of the read unit (macx_chain_h2.hip.h) as a translation unit of their own: they are
// a third of the library's compile time each, and build.py compiles the translation units in parallel.  Exports (C++ linkage,
// declared in macx_chain_api.hip.h): macx::chain_fwd_launch.
#include <hip/hip_runtime.h>
#define MACX_H2_NO_KERNELS
#define MACX_CHAIN_FWD_TU
#include "macx_chain_h2.hip.h"
