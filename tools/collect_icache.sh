#!/bin/bash
# Run ON THE GPU BOX (via gpurun): instruction-cache behaviour of the hot kernel (one --pmc pass).
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/ic_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT -o ic -- python bench.py --steps 16 --warmup 68 --no-cpu-baseline --no-pgs-variant --chunk 0 > /dev/null 2> $OUT/ic.err
python - <<PY
import csv, collections, json, glob
out = {}
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in d.items():
        if "k_step" in k:
            out.setdefault(k, {})[c] = sum(v[-16:]) / len(v[-16:])
json.dump(out, open("$OUT/ic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
