#!/bin/bash
# Times the fused kernels for each ablation variant (results of the variants are wrong on purpose).
cd "$(dirname "$0")/.."
for lib in "" $(ls diffusion-model_amd/isingdiff/abl/*.so); do
  echo "== ${lib:-baseline}"
  ISINGDIFF_LIB=${lib:+$PWD/$lib} timeout -k 10 120 python -u bench.py --batch 4096 --T 30 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), ' '.join(f\"{k['name'].split('(')[0]}={k['avg_us']:.0f}\" for k in d['kernels']))"
done
