#!/bin/bash
# Register / spill / scratch / LDS figures of every kernel in an object file built by csrc/Makefile (llvm-readelf --notes
# on the gfx950 code object inside the fat binary).  usage: tools/kernel_regs.sh fs-nerf_amd/csrc/render.o [name filter]
set -e
O=$1; F=${2:-.}
T=$(mktemp -d)
LL=/opt/rocm/lib/llvm/bin
$LL/llvm-objcopy -O binary --only-section=.hip_fatbin $O $T/fat
$LL/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat --output=$T/dev.co --unbundle
$LL/llvm-readelf --notes $T/dev.co | python3 -c "
import sys, re, subprocess
txt = sys.stdin.read()
for b in txt.split('- .agpr_count')[1:]:
    name = re.search(r'\.name:\s+(\S+)', b).group(1)
    g = lambda k: re.search(r'\.' + k + r':\s+(\d+)', b).group(1)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if not re.search(r'$F', dem): continue
    print(f\"{dem[:70]:70s} vgpr {g('vgpr_count'):>3} vspill {g('vgpr_spill_count'):>3} sgpr {g('sgpr_count'):>3} sspill {g('sgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size')}\")
"
rm -rf $T
