#!/bin/bash
# C3's columns with and without the nich blocks (MSC_NO_NICH_BLOCKS=1: every plain nich feature on its own), whole and by family
for nb in 0 1; do
  for fam in all nich; do
    echo "== no_blocks=$nb family=$fam"
    if [ "$nb" = 1 ]; then export MSC_NO_NICH_BLOCKS=1; else unset MSC_NO_NICH_BLOCKS; fi
    if [ "$fam" = all ]; then python tools/scans/c3_pieces.py 256 "$@"; else python tools/scans/c3_pieces.py 256 --family=nich "$@"; fi
  done
done 2>&1 | grep -v amdgpu.ids
