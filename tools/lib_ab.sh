#!/bin/bash
# GPU box: the product library against experiment builds (build_ab/librts_<name>.so), alternating processes, same frame.
#   usage: tools/lib_ab.sh <config> <kernel> <name> [<name> ...]      (OPTS=key=value,... for context options)
set -u
CFG=$1; K=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for rep in 1 2 3; do
  for name in product "$@"; do
    lib=""; [ "$name" != product ] && lib=$REPO/build_ab/librts_$name.so
    RTS_LIB=$lib KERNELS=$K N=${N:-200} OPTS=${OPTS:-} python tests/experiments/kernel_ab.py $CFG 2>&1 | tail -1 | sed "s#^#$name: #"
  done
done
