#!/bin/bash
# Builds an experiment variant of the library beside the product: tools/build_ab.sh <name> <-Dflags...>  ->  build_ab/librts_<name>.so
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
make -C $REPO/raytracedshadows_amd/csrc -j8 BUILD=../../build_ab/obj_$NAME OUT=../../build_ab/librts_$NAME.so EXTRA="$*" > $REPO/build_ab/build_$NAME.log 2>&1 || { tail -5 $REPO/build_ab/build_$NAME.log; exit 1; }
ls -la $REPO/build_ab/librts_$NAME.so
