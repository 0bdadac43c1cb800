// cuImage.h -- forwarding header: the reference's cuImage.h:8-26 surface lives in cuSIFT.h of this build.
#ifndef CUSIFT_AMD_CUIMAGE_H
#define CUSIFT_AMD_CUIMAGE_H
#include "cuSIFT.h"
#endif
