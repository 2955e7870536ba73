#!/bin/bash
# counter evidence for profiles/r04_dm.txt: 4 x dm(4) on 1M rows, small counts (tables staged), K = 32 and 256
# (one counter group per pass; the groups are the ones tools/profile_round.sh collects without complaint)
python tools/scans/dm_case.py 32 256 --small 2>&1 | grep -v amdgpu.ids
python tools/scans/dm_case.py 32 256 2>&1 | grep -v amdgpu.ids
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  echo "== pass $i: $set"
  bash tools/scans/pmc.sh dm$i k_score_tile "$set" $GRAFT_REPO_ROOT/tools/scans/dm_case.py 32 256 --small
  echo "pass $i rc=$?"
done
