#!/bin/bash
# Run ON THE GPU BOX (via gpurun): SQ issue / stall / FLOP counters of the hot kernels, one rocprofv3 --pmc pass per
# group of <= 8 SQ counters (never combined with a trace).  Averages the LAST 16 launches of each kernel = the timed,
# desynchronised steady-state steps of bench.py.  Writes gpurun_out/sq_<tag>/sq.json with _meta.version = kmanip_version().
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 16 --warmup 4 --no-cpu-baseline --no-variants"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY"
P2="SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES"
P3="SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT"
P4="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_VSKIPPED"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT -o p$i -- python3 bench.py $ARGS > $OUT/p$i.out 2> $OUT/p$i.err || echo "pass $i failed (see $OUT/p$i.err)"
done
python3 - <<PY
import csv, collections, json, glob, ctypes, os
out = {}
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in d.items():
        if "k_" in k:
            out.setdefault(k, {})[c] = sum(v[-16:]) / len(v[-16:])
L = ctypes.CDLL(os.path.join("gym_kmanip_amd", "libkmanip_hip.so")); L.kmanip_version.restype = ctypes.c_char_p
out["_meta"] = {"version": L.kmanip_version().decode(), "command": "bench.py $ARGS", "averaged": "last 16 launches per kernel"}
json.dump(out, open("$OUT/sq.json", "w"), indent=1)
for k, v in out.items():
    if k.startswith("void k_step") and "SQ_WAVE_CYCLES" in v:
        wc = v["SQ_WAVE_CYCLES"]
        print(k, {c: round(v[c] / wc, 4) for c in v if c.startswith(("SQ_ACTIVE", "SQ_WAIT", "SQ_INST_CYCLES"))})
PY
