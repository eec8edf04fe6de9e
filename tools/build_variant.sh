#!/bin/bash
# Build an A/B variant of libfsnerf_hip.so into ab/<name>.so: one translation unit (SRC=render by default) recompiled
# with extra flags, the other objects taken from the in-tree build.
# usage: [SRC=train_fused] tools/build_variant.sh <name> [hipcc flags...]
set -e
R=$(cd $(dirname $0)/.. && pwd)
N=$1; shift
SRC=${SRC:-render}
mkdir -p $R/ab /tmp/var_$N
cd $R/fs-nerf_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -I../../include -I. "$@" -c $SRC.hip -o /tmp/var_$N/$SRC.o
OBJS=$(ls *.o | grep -v "^$SRC.o\$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/ab/$N.so /tmp/var_$N/$SRC.o $OBJS
echo built $R/ab/$N.so
