#!/bin/bash
# 4 x dm(4) with the product library and every variant under common_amd/lib/variants
for lib in "" common_amd/lib/variants/*.so; do
  if [ -z "$lib" ]; then unset MSC_LIB_PATH; echo "== product"; else export MSC_LIB_PATH=$PWD/$lib; echo "== $lib"; fi
  python tools/scans/dm_case.py 32 256 --small 2>&1 | grep -v amdgpu.ids
  python tools/scans/dm_case.py 256 2>&1 | grep -v amdgpu.ids
done
