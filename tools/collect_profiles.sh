#!/bin/bash
# Copy the outputs of tools/measure_round.sh TAG (gpurun_out/, scratch) into profiles/ (tracked).  usage: tools/collect_profiles.sh r04
TAG=${1:-r04}
G=gpurun_out; P=profiles; F=$G/final_$TAG
cp $F/bench.json $P/${TAG}_bench.json
for s in h2 lih graphene diamond; do cp $F/bench_$s.json $P/${TAG}_bench_$s.json; done
cp $G/${TAG}_bench/bench.json $P/${TAG}_bench_under_rocprof.json
for k in bench graphene diamond value vjp; do f=$(find $G/${TAG}_$k -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $P/${TAG}_${k}_kernel_stats.csv; done
cp $G/pmc_$TAG/summary.json $P/${TAG}_pmc_traffic.json
cp $F/pmc_mfma.json $P/${TAG}_pmc_mfma.json
cp $F/grad_bench.txt $P/${TAG}_grad_bench.txt
for f in large_cells batch_sweep f32_stage_loss; do [ -f $F/$f.txt ] && cp $F/$f.txt $P/${TAG}_$f.txt; done
ls -la $P | grep ${TAG}_
