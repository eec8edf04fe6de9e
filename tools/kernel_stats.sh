#!/bin/bash
# Register / spill / scratch / LDS usage of the kernels of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/kernel_stats.sh <file.hip> [grep-pattern-on-mangled-name] [extra hipcc flags...]
R=$(cd $(dirname $0)/.. && pwd)
F=$1; P=${2:-.}; shift; shift
cd $R/fs-nerf_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -I../../include -I. "$@" \
  -c $F -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|Spill|ScratchSize|LDS Size|SGPRs:" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - - - - - | grep -E "$P"
