#!/bin/bash
# resource usage + ISA of one kernel of a .hip file: tools/scans/kernel_asm.sh kernels_score.hip <mangled-name-prefix> [out.s]
# (compiles with the Makefile's flags into /tmp/kasm, prints VGPRs / spills / scratch / occupancy and instruction counts)
SRC=$1; SYM=$2; OUT=${3:-/tmp/kasm/kernel.s}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p /tmp/kasm && cd /tmp/kasm || exit 1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -I$ROOT/include -I$ROOT/common_amd/csrc -c $ROOT/common_amd/csrc/$SRC -o /tmp/kasm/o.o \
  -Rpass-analysis=kernel-resource-usage --save-temps=obj 2> /tmp/kasm/remarks.txt
grep -A10 "Function Name: $SYM" /tmp/kasm/remarks.txt | grep -E "Name|SGPRs|VGPRs|Scratch|Occupancy|LDS" | sed 's/.*remark: [^ ]* *//'
F=$(ls /tmp/kasm/*gfx950.s | head -1)
s=$(grep -n "^$SYM" $F | head -1 | cut -d: -f1)
e=$(awk -v s=$s 'NR>s && /s_endpgm/ {print NR; exit}' $F)
sed -n "${s},${e}p" $F > $OUT
echo "lines $(wc -l < $OUT)  s_load_x16 $(grep -c s_load_dwordx16 $OUT)  s_load_x8 $(grep -c s_load_dwordx8 $OUT)  scratch $(grep -c scratch_ $OUT)  lane-ops $(grep -c 'v_writelane\|v_readlane' $OUT)  v_mov $(grep -c v_mov_b32 $OUT)"
