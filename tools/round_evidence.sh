#!/bin/bash
# One round's committed evidence, on the GPU box from the repo root: tools/round_evidence.sh <tag>
#   <tag>_bench_n1.json                the default bench line
#   <tag>_bench_n1_driver_shape.json   the driver's --steps 20 --warmup 5
#   <tag>_bench_2rank_gloo.json        two ranks on this one GPU over gloo (the N > 1 code path; NOT a scaling number)
#   <tag>_bench_c5_nccl_1rank.json     one rank over RCCL with the exchange forced
#   then tools/profile_round.sh <tag>  (kernel stats, timed region, PMC)
# everything under gpurun_out/<tag>/; copy what is to be judged into profiles/.
set -u
TAG=$1
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/bench_n1.err; echo "bench n1 rc=$?"
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_n1_driver_shape.json 2> $OUT/bench_driver.err; echo "bench driver shape rc=$?"
MSC_BENCH_BACKEND=gloo python bench.py --gpus 2 --no-cpu-baseline > $OUT/${TAG}_bench_2rank_gloo.json 2> $OUT/bench_2rank.err; echo "bench 2 ranks gloo rc=$?"
MSC_BENCH_FORCE_C5=1 python bench.py --no-cpu-baseline > $OUT/${TAG}_bench_c5_nccl_1rank.json 2> $OUT/bench_c5.err; echo "bench c5 nccl 1 rank rc=$?"
bash tools/profile_round.sh $TAG
