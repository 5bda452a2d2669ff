// Identity of the build: sha256 over every source, header and compiler flag of the library (lram_amd/build.py computes it
// and passes it as LRAM_BUILD_ID_HEX).  The marker string is also what build.py looks for in the .so file to decide whether
// a library on disk was built from the checked-out sources -- a stale one on a GPU box would run old kernels without a word.
#include "../../include/lram_hip.h"

#ifndef LRAM_BUILD_ID_HEX
#error "compile through lram_amd/build.py (it defines LRAM_BUILD_ID_HEX)"
#endif

namespace {
const char kBuildId[] = "LRAM_BUILD_ID=" LRAM_BUILD_ID_HEX;
}

extern "C" const char* lram_build_id(void) { return kBuildId + 14; }
