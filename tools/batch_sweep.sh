#!/bin/bash
# frames/s of the timed workload against frame pairs per launch, on one box (python bench.py --batch B, short form)
set -e -o pipefail
mkdir -p gpurun_out
: > gpurun_out/batch_sweep.txt
for rep in 1 2; do
for b in 256 512 768 1024 2048; do
  python3 bench.py --batch $b --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --profile-events 0 --allow-stale-profile > gpurun_out/bs_$b.json 2> gpurun_out/bs_$b.err
  python3 -c "import json;d=json.load(open('gpurun_out/bs_$b.json'));print($b, round(d['value']), round(d['ms_per_step'],4))" | tee -a gpurun_out/batch_sweep.txt
done
done
