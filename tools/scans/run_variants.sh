#!/bin/bash
# on the GPU box: C3's scoring pass and sweep step with every library under common_amd/lib/variants (and the product's)
for lib in "" common_amd/lib/variants/*.so; do
  if [ -z "$lib" ]; then unset MSC_LIB_PATH; echo "== product"; else export MSC_LIB_PATH=$PWD/$lib; echo "== $lib"; fi
  python tools/scans/c3_pieces.py "$@" 2>&1 | grep -v amdgpu.ids
done
