#!/bin/bash
# Round measurement batch (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats, PMC traffic / MFMA /
# shader-clock passes, value-chain and gradient profiles.  Results land in gpurun_out/; copy what should be judged into profiles/.
# usage: tools/measure_round.sh TAG        (TAG like r02)
TAG=${1:-r02}
set -x
mkdir -p gpurun_out/final_$TAG
F=gpurun_out/final_$TAG
python -m pytest tests -m gpu -q -k "int8" 2>&1 | tail -2 > $F/int8_tests.txt
# the traffic pass first, and its summary into profiles/ ON THIS BOX: the bench line's `roofline.traffic` then comes from the same box
bash tools/pmc_traffic.sh $TAG > $F/pmc_traffic.txt 2>&1
cp gpurun_out/pmc_$TAG/summary.json profiles/${TAG}_pmc_traffic.json
python bench.py > $F/bench.json 2> $F/bench.err
tail -c 900 $F/bench.json
bash tools/profile_bench.sh ${TAG}_bench > $F/prof.txt 2>&1
bash tools/pmc_mfma.sh $TAG > $F/pmc_mfma.json 2> $F/pmc_mfma.err
bash tools/profile_cmd.sh ${TAG}_value tools/value_driver.py > /dev/null 2>&1
bash tools/profile_cmd.sh ${TAG}_vjp tools/grad_bench.py bcc_li 4096 vjp > /dev/null 2>&1
python tools/grad_bench.py bcc_li 4096 > $F/grad_bench.txt 2>&1
for s in h2 lih; do python bench.py --system $s --no-cpu-baseline > $F/bench_$s.json 2>> $F/bench.err; done
python bench.py --system graphene --batch 512 --steps 2 --no-cpu-baseline > $F/bench_graphene.json 2>> $F/bench.err
python bench.py --system diamond --dtype f32 --batch 1024 --steps 2 --no-cpu-baseline > $F/bench_diamond.json 2>> $F/bench.err
bash tools/profile_cmd.sh ${TAG}_graphene bench.py --system graphene --batch 512 --steps 2 --no-cpu-baseline --no-mcmc > /dev/null 2>&1
bash tools/profile_cmd.sh ${TAG}_diamond bench.py --system diamond --dtype f32 --batch 1024 --steps 2 --no-cpu-baseline --no-mcmc > /dev/null 2>&1
# cells between and beyond the BASELINE sizes (float64), the float32 stage-loss table, the forward / mcmc_step timings
for a in "bcc_li 3,3,2 256" "bcc_li 4,3,2 256" "bcc_li 3 256" "graphene 3 128"; do set -- $a; python tools/kbench.py --system $1 --S $2 --batch $3 --steps 2 --check 0 2>&1 | grep -v "^/opt"; done > $F/large_cells.txt
python tools/kbench.py --system diamond --batch 256 --steps 2 --check 1 2>&1 | grep -v "^/opt" >> $F/large_cells.txt
for b in 256 512 1024 2048; do python tools/kbench.py --batch $b --steps 3 --check 0 2>&1 | grep -v "^/opt" | head -1; done > $F/batch_sweep.txt
python tools/f32_stage_loss.py diamond 2>&1 | grep -v "^/opt" > $F/f32_stage_loss.txt
ls -la $F
