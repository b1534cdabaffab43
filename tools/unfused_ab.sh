run() {
  python bench.py --env $1 --envs-per-gpu $2 --steps 128 --warmup 8 --no-variants --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-14s %-28s value %.4g ms/step %.4f k_step %.4f' % ('$1', '$3', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg']['k_step']))"
}
for e in KManipDualArm KManipTorso; do
  KMANIP_COST_SORT=1 run $e 8192 "fused sort"
  KMANIP_IK_UNFUSED=1 KMANIP_COST_SORT=0 run $e 8192 "unfused nosort"
  KMANIP_IK_UNFUSED=1 KMANIP_COST_SORT=1 KMANIP_COST_W=0,1,0,0,0,25 run $e 8192 "unfused sort work-only"
done
