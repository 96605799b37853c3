#!/bin/bash
# VGPR / AGPR / SGPR / scratch / LDS of every kernel in the built objects (rcdms_amd/lib/*.o), from the code objects'
# metadata notes: the check that no shipped instantiation spills (.private_segment_fixed_size) and what occupancy it gets.
# usage: tools/kernel_resources.sh [pattern]   (pattern: grep -E filter on the demangled line)
cd "$(dirname "$0")/.."
LLVM=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
for o in rcdms_amd/lib/*.o; do
  b=$(basename "$o" .o)
  $LLVM/llvm-objcopy --dump-section .hip_fatbin="$tmp/$b.fb" "$o" 2>/dev/null || continue
  $LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$tmp/$b.fb" --output="$tmp/$b.co" --unbundle 2>/dev/null || continue
  $LLVM/llvm-readelf --notes "$tmp/$b.co" 2>/dev/null | B=$b python3 -c "
import os, re, sys
txt = sys.stdin.read()
for blk in re.split(r'\n\s*- \.agpr_count', txt)[1:]:
    blk = '.agpr_count' + blk
    g = lambda k: (re.search(r'\.' + k + r':\s*(\S+)', blk) or [None, '?'])[1]
    print('%5s v %4s a %4s s %5s scratch %6s lds  %-8s %s' % (g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size'), os.environ['B'], g('name')))
"
done | c++filt | sed -e 's/void (anonymous namespace):://' -e 's/(\(anonymous namespace\)::[A-Za-z]*Args)//' > "$tmp/all.txt"
if [ -n "$1" ]; then grep -E "$1" "$tmp/all.txt"; else cat "$tmp/all.txt"; fi
rm -rf "$tmp"
