// rt_build_id.cpp -- the identity of what this library was built from (raytracing_simple_amd/_build.py source_hash: every source
// under csrc/, the public headers, the compiler flags).  Profiles are stamped with it, and bench.py prints counter-derived
// figures of a committed profile only beside the library they were measured on.
#include "../../include/rt_api.h"
#include "rt_build_id.h"        // generated: csrc/_obj/rt_build_id.h

extern "C" RT_API const char *rt_build_id(void) { return RT_BUILD_ID; }
