#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of two builds of the library (KMANIP_LIB), alternating, `reps` times each.
# Usage: bash tools/ab.sh <libA.so> <libB.so> [reps=3] [bench.py args...]      (default args: the headline workload, 1024 timed launches)
A=$1; B=$2; REPS=${3:-3}; shift 3 2>/dev/null || shift $#
ARGS="${@:---steps 1024 --warmup 16 --no-variants --no-cpu-baseline} --time-every 1"      # (events around EVERY launch on both sides: a library older than 0.30 knows no sampling)
one() {
  KMANIP_LIB=$1 python3 bench.py $ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
k = d['roofline']['kernel_ms_avg']
print('%-44s value %.4g  ms/step %.4f  k_step %.4f  k_render %.4f  %s' % ('$1', d['value'], d['ms_per_step'], k['k_step'], k['k_render'], d['config']['library']))"
}
for i in $(seq $REPS); do one $A; one $B; done
