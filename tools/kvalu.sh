#!/bin/bash
# static instruction mix of one kernel instantiation in a built library: tools/kvalu.sh <lib.so> <mangled-name substring>
LIB=$1; K=$2
cd /tmp && mkdir -p co && cd co && python3 - "$LIB" <<'PY'
import re,struct,sys
b=open(sys.argv[1],'rb').read()
for m in re.finditer(b'\x7fELF',b):
    i=m.start()
    if b[i+18:i+20]==b'\xe0\x00':
        shoff=struct.unpack_from('<Q',b,i+0x28)[0]; shentsize,shnum=struct.unpack_from('<HH',b,i+0x3a)
        open("kv.co",'wb').write(b[i:i+shoff+shentsize*shnum])
PY
SYM=$(/opt/rocm/lib/llvm/bin/llvm-readelf -s kv.co | grep "$K" | grep -v "\.kd" | awk '{print $8}' | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d --disassemble-symbols=$SYM kv.co | awk 'NR>7{print $1}' > kv.dis
echo "$SYM" | cut -c1-80
echo "VALU $(grep -c '^v_' kv.dis)  cndmask $(grep -c '^v_cndmask' kv.dis)  mov $(grep -cE '^v_mov_b(32|64)' kv.dis)  dpp-ish $(grep -c 'dpp' kv.dis)  SALU $(grep -c '^s_' kv.dis)  lds $(grep -c '^ds_' kv.dis)  rcp/rsq $(grep -cE '^v_(rcp|rsq|sqrt)' kv.dis) lanes $(grep -cE '^v_(readlane|writelane)' kv.dis)"
