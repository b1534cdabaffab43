#!/bin/bash
# GPU box: A/B of the cost-sorted wave slots (KMANIP_COST_SORT / KMANIP_COST_W = ik,work,near-cube,armtab,cubetab,binwidth) on the two-arm configs
run() {  # env n label
  python bench.py --env $1 --envs-per-gpu $2 --steps 128 --warmup 8 --no-variants --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-14s %-22s value %.4g ms/step %.4f k_step %.4f' % ('$1', '$3', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg']['k_step']))"
}
for e in KManipDualArm KManipTorso; do
  KMANIP_COST_SORT=0 run $e 8192 "off"
  for w in "10,1,0,0,0,50" "10,1,500,0,0,50" "10,1,1000,0,0,50" "18,1,1000,0,0,50" "18,1,2000,0,0,100"; do
    KMANIP_COST_SORT=1 KMANIP_COST_W=$w run $e 8192 "$w"
  done
done
