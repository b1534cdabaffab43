#!/bin/bash
# Run ON THE GPU BOX (via gpurun): SQ issue/stall counters of the hot kernels, two --pmc passes (8 SQ slots each).
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 16 --warmup 4 --no-cpu-baseline --no-pgs-variant --chunk 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT -o p1 -- python bench.py $ARGS > /dev/null 2> $OUT/p1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $OUT -o p2 -- python bench.py $ARGS > /dev/null 2> $OUT/p2.err
python - <<PY
import csv, collections, json, glob
out = {}
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in d.items():
        if "k_" in k:
            out.setdefault(k, {})[c] = sum(v) / len(v)
json.dump(out, open("$OUT/sq.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
