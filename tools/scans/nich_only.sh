#!/bin/bash
# sixteen plain nich columns alone (C3's), 1M rows: the pack kernels (k_score_nich_pack / k_sweep_nich_pack) against the
# lane <-> row kernel and the LDS-staged tile kernels; with every library under common_amd/lib/variants too
for lib in "" common_amd/lib/variants/*.so; do
  if [ -z "$lib" ]; then unset MSC_LIB_PATH; echo "#### product"; else [ -f "$lib" ] || continue; export MSC_LIB_PATH=$PWD/$lib; echo "#### $lib"; fi
  for K in 256 128 64 32; do
    echo "== K $K default";              python tools/scans/c3_pieces.py $K --family=nich 2>&1 | grep -v amdgpu
    echo "== K $K no lane<->row kernel"; MSC_TAIL_MIN_ROWS=100000000 python tools/scans/c3_pieces.py $K --family=nich 2>&1 | grep -v amdgpu
  done
done
echo "#### no pack (MSC_NO_ROLES)"
unset MSC_LIB_PATH
for K in 256 128; do echo "== K $K"; MSC_NO_ROLES=1 python tools/scans/c3_pieces.py $K --family=nich 2>&1 | grep -v amdgpu; done
