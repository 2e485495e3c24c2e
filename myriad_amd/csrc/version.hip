#include "common.h"
extern "C" const char* mh_version(void) { return "myriad_hip 0.1 (gfx950)"; }
extern "C" int mh_target_arch(void) { return 950; }
