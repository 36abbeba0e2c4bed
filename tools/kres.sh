#!/bin/bash
# resource usage of named kernel instantiations in the built library: tools/kres.sh <substring> ...
cd /tmp && mkdir -p co && cd co && python3 - <<'PY'
import re,struct
b=open("/root/repo/hipims-ocl_amd/lib/libhipims_mi.so",'rb').read()
for m in re.finditer(b'\x7fELF',b):
    i=m.start()
    if b[i+18:i+20]==b'\xe0\x00':
        shoff=struct.unpack_from('<Q',b,i+0x28)[0]; shentsize,shnum=struct.unpack_from('<HH',b,i+0x3a)
        open("cur_0.co",'wb').write(b[i:i+shoff+shentsize*shnum])
PY
for k in "$@"; do echo "== $k"; /opt/rocm/lib/llvm/bin/llvm-readelf --notes cur_0.co 2>/dev/null | grep -A40 "name:.*$k" | grep -E "vgpr_count|sgpr_spill|vgpr_spill|private_segment_fixed" | head -4 | tr '\n' ' '; echo; done
