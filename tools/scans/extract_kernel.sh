#!/bin/bash
# cut one kernel's ISA out of a --save-temps .s file: extract_kernel.sh file.s <mangled-name-prefix> out.s
F=$1; SYM=$2; OUT=$3
s=$(grep -n "^$SYM" $F | head -1 | cut -d: -f1)
e=$(awk -v s=$s 'NR>s && /^\.Lfunc_end/ {print NR; exit}' $F)
sed -n "${s},${e}p" $F > $OUT
echo "lines $(wc -l < $OUT) scratch $(grep -c scratch_ $OUT) writelane $(grep -c v_writelane $OUT) readlane $(grep -c v_readlane $OUT) s_load $(grep -c s_load_ $OUT) v_mov $(grep -c 'v_mov_b' $OUT)"
