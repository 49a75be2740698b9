#!/bin/bash
# GPU box: what do the mask's byte stores cost?  Two builds of the library -- the product, and one whose everyday kernels never
# store (RTS_EXPERIMENT_NO_MASK_STORE: the store is behind a condition that is never true, the walk stays) -- timed in
# alternating processes on the same frame.  The difference bounds what ANY better store pattern (row stores, 2x2 tiles per
# workgroup ...) could gain.  The experiment build lives in build_ab/ and is never installed.
#   usage: tools/mask_store_ab.sh <config> [kernel]
set -u
CFG=${1:-city_4k}; K=${2:-8}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
if [ ! -f build_ab/librts_nostore.so ]; then
  make -C raytracedshadows_amd/csrc -j8 BUILD=../../build_ab/obj_nostore OUT=../../build_ab/librts_nostore.so EXTRA=-DRTS_EXPERIMENT_NO_MASK_STORE > build_ab_nostore.log 2>&1 || { tail -5 build_ab_nostore.log; exit 1; }
fi
for rep in 1 2 3; do
  for lib in "" build_ab/librts_nostore.so; do
    RTS_LIB=${lib:+$REPO/$lib} KERNELS=$K N=200 OPTS=${OPTS:-} python tests/experiments/kernel_ab.py $CFG 2>&1 | tail -1 | sed "s#^#${lib:-product}: #"
  done
done
