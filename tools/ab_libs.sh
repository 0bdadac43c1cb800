#!/bin/bash
# Same-box A/B of library builds (make ab NAME=x DEFS=...): for each library in turn, REPS times, a fresh process runs
# tools/ab_pyramid.py with the default launch policy -- ms per 64 x 1080p batch for a lone caller and for four streams.
#   tools/ab_libs.sh base x y ...      (names under tools/ab/, or paths);  CONTENT=tile|blobs|raw  REPS=3
set -u
CONTENT=${CONTENT:-tile}
for rep in $(seq 1 ${REPS:-3}); do for v in "$@"; do
  lib=$v; [ -f "tools/ab/$v.so" ] && lib=$PWD/tools/ab/$v.so
  CUSIFT_AMD_LIB=$lib AB_POLICIES=${AB_POLICIES:--1} python tools/ab_pyramid.py 2 "$CONTENT" 2>&1 | grep "stream(s)" | sed "s/^/$(basename $v .so): /"
done; done
