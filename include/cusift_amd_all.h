/*
 * cusift_amd_all.h -- the whole C ABI of libcusift_amd.so: the drop-in core and its three companions.
 */
#ifndef CUSIFT_AMD_ALL_H
#define CUSIFT_AMD_ALL_H
#include "cusift_amd.h"
#include "cusift_amd_stages.h"
#include "cusift_amd_multigpu.h"
#include "cusift_amd_extras.h"
#endif /* CUSIFT_AMD_ALL_H */
