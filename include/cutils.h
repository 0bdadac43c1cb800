// cutils.h -- forwarding header: iDivUp/iAlignUp, safeCall, InitCuda, TimerGPU/TimerCPU (reference
// cutils.h:15-140) are provided by cuSIFT.h of this build.
#ifndef CUSIFT_AMD_CUTILS_H
#define CUSIFT_AMD_CUTILS_H
#include "cuSIFT.h"
#endif
