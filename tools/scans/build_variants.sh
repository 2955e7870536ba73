#!/bin/bash
# experiment builds of the library (CPU): tools/scans/build_variants.sh name "flags" [name "flags" ...]
cd "$(dirname "$0")/../../common_amd/csrc" || exit 1
while [ $# -ge 2 ]; do
  echo "== variant $1: $2"
  make -s -j8 VARIANT="$1" EXTRA="$2" 2>&1 | grep -E "error" -A5 | head -20
  shift 2
done
ls -la ../lib/variants/
