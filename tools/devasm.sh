#!/bin/bash
# device code of one HIP object of the in-tree build: bash tools/devasm.sh <name, e.g. mbmap> -> /tmp/<name>.hsaco, /tmp/<name>.s + a table
# of (kernel, VGPRs, AGPRs, SGPRs, scratch bytes, LDS) from the AMDGPU metadata notes
set -e
L=/opt/rocm/lib/llvm/bin; n=$1; o=$(dirname $0)/../rust-birdnet-onnx_amd/csrc/$n.o
$L/llvm-objcopy --dump-section=.hip_fatbin=/tmp/$n.fatbin $o /tmp/$n.copy.o
$L/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/$n.fatbin --output=/tmp/$n.hsaco
$L/llvm-objdump -d /tmp/$n.hsaco > /tmp/$n.s
$L/llvm-readelf --notes /tmp/$n.hsaco | python3 -c "
import sys,re
cur={}
for l in sys.stdin:
    m=re.match(r'\s+\.(name|vgpr_count|agpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\S+)',l)
    if m: cur[m.group(1)]=m.group(2)
    if l.strip().startswith('.wavefront_size') or l.strip().startswith('- .agpr_count'):
        pass
    if m and m.group(1)=='vgpr_count':
        pass
    if l.strip().startswith('.vgpr_spill_count') or (m and m.group(1)=='vgpr_count'):
        if 'name' in cur and 'vgpr_count' in cur:
            import subprocess
            print(cur.get('vgpr_count'),cur.get('agpr_count'),cur.get('sgpr_count'),cur.get('private_segment_fixed_size'),cur.get('name')); cur={}
" | while read v a s p name; do echo "v=$v a=$a s=$s scratch=$p $(echo $name | c++filt | cut -c1-140)"; done
