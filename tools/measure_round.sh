#!/bin/bash
# Round measurement batch (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats, PMC traffic / MFMA /
# shader-clock passes, value-chain and gradient profiles.  Results land in gpurun_out/; copy what should be judged into profiles/.
# usage: tools/measure_round.sh TAG        (TAG like r02)
TAG=${1:-r02}
set -x
mkdir -p gpurun_out/final_$TAG
F=gpurun_out/final_$TAG
python bench.py > $F/bench.json 2> $F/bench.err
tail -c 900 $F/bench.json
bash tools/profile_bench.sh ${TAG}_bench > $F/prof.txt 2>&1
bash tools/pmc_traffic.sh $TAG > $F/pmc_traffic.txt 2>&1
bash tools/pmc_mfma.sh $TAG > $F/pmc_mfma.json 2> $F/pmc_mfma.err
python tools/gemm_timeline.py > $F/gemm_timeline.txt 2>&1
bash tools/profile_cmd.sh ${TAG}_value tools/value_driver.py > /dev/null 2>&1
bash tools/profile_cmd.sh ${TAG}_vjp tools/grad_bench.py bcc_li 4096 vjp > /dev/null 2>&1
python tools/grad_bench.py bcc_li 4096 > $F/grad_bench.txt 2>&1
for s in h2 lih; do python bench.py --system $s --no-cpu-baseline > $F/bench_$s.json 2>> $F/bench.err; done
python bench.py --system graphene --batch 512 --steps 2 --no-cpu-baseline > $F/bench_graphene.json 2>> $F/bench.err
python bench.py --system diamond --dtype f32 --batch 1024 --steps 2 --no-cpu-baseline > $F/bench_diamond.json 2>> $F/bench.err
bash tools/profile_cmd.sh ${TAG}_graphene bench.py --system graphene --batch 512 --steps 2 --no-cpu-baseline --no-mcmc > /dev/null 2>&1
bash tools/profile_cmd.sh ${TAG}_diamond bench.py --system diamond --dtype f32 --batch 1024 --steps 2 --no-cpu-baseline --no-mcmc > /dev/null 2>&1
ls -la $F
