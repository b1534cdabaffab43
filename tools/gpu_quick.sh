#!/bin/bash
# Run ON THE GPU BOX (via gpurun): the quick measurement loop of a kernel change -- headline bench line (no variants, no CPU leg),
# SQ instruction counters (two passes) and the phase-stamp profile of the diagnostic build, into gpurun_out/q_<tag>/
set -o pipefail
TAG=${1:-q}
OUT=gpurun_out/q_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --no-variants --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
python3 -c "
import json; d = json.load(open('$OUT/bench.json')); print('value %.4g  ms/step %.4f  k_step %.4f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg']['k_step']))"
if [ "$2" != "nosq" ]; then
ARGS="--steps 16 --warmup 4 --no-cpu-baseline --no-variants"
P2="SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVES"
P4="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH"
i=0
for P in "$P2" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT -o p$i -- python3 bench.py $ARGS > $OUT/p$i.out 2> $OUT/p$i.err || echo "pass $i failed (see $OUT/p$i.err)"
done
python3 - <<PY
import csv, collections, json, glob
out = {}
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in d.items():
        if "k_step" in k:
            out.setdefault(k, {})[c] = sum(v[-16:]) / len(v[-16:])
json.dump(out, open("$OUT/sq.json", "w"), indent=1)
for k, v in out.items():
    f64 = sum(v.get(c, 0) for c in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64"))
    print(k[:40], "VALU %.1fM  F64 %.1fM  other %.1fM  SALU %.1fM  LDS %.1fM  BR %.2fM  wait %.3f  lanes %.3f" % (
        v.get("SQ_INSTS_VALU", 0) / 1e6, f64 / 1e6, (v.get("SQ_INSTS_VALU", 0) - f64) / 1e6, v.get("SQ_INSTS_SALU", 0) / 1e6, v.get("SQ_INSTS_LDS", 0) / 1e6,
        v.get("SQ_INSTS_BRANCH", 0) / 1e6, v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1), v.get("SQ_THREAD_CYCLES_VALU", 0) / max(64 * v.get("SQ_INSTS_VALU", 1), 1)))
PY
fi
if [ -f gym_kmanip_amd/libkmanip_hip_prof.so ]; then
  KMANIP_LIB=gym_kmanip_amd/libkmanip_hip_prof.so python3 tools/phase_profile.py > $OUT/phase.txt 2> $OUT/phase.err || echo "phase profile failed"
  grep "last launch" $OUT/phase.txt || true
fi
