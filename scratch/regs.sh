#!/bin/bash
# usage: regs.sh file.o [filter] -- VGPR / SGPR / spill / LDS of every kernel in the gfx950 code object of a hipcc object file
O=$1; F=${2:-.}
D=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section=.hip_fatbin=$D/fat.bin $O 2>/dev/null
T=$(/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$D/fat.bin 2>/dev/null | grep gfx950 | head -1)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=$T --input=$D/fat.bin --output=$D/k.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $D/k.co | python3 -c "
import sys, re
txt = sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    g = lambda k: (re.search(r'\.'+k+r':\s*(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    import subprocess
    if re.search(r'$F', name or ''):
        print(g('vgpr_count'), 'vgpr', g('sgpr_count'), 'sgpr', g('vgpr_spill_count'), 'vspill', g('private_segment_fixed_size'), 'scratch', g('group_segment_fixed_size'), 'lds', name)
" | while read a; do n=$(echo "$a" | awk '{print $NF}'); echo "$(echo "$a" | sed "s/ [^ ]*$//") $(echo $n | c++filt | cut -c1-100)"; done
rm -rf $D
