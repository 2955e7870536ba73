#!/bin/bash
# does a pass get cheaper with fewer groups?  (the kernels are chosen by the plan's prices: launchers.hpp PlanCost)
for sp in bb:32 dd32:16 bb:8,dd8:8,gp:8 bb:16,gp:16,dd32:16,nich:16 nich:16 bb:8,nich:8; do
  for K in 256 128 100 64 32; do echo "== $sp K=$K"; python tools/scans/c3_pieces.py $K --spec=$sp 2>&1 | grep -v amdgpu; done
done
