#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: bench line + rocprofv3 kernel stats + PMC HBM traffic
# (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes) -> gpurun_out/prof_<tag>/
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ktrace -- python bench.py --no-cpu-baseline --no-pgs-variant > $OUT/bench_under_prof.json 2> $OUT/ktrace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-pgs-variant --chunk 0 > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-pgs-variant --chunk 0 > /dev/null 2> $OUT/write.err
python - <<PY
import csv, collections, json
out = {}
for tag, ctr in [("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")]:
    rows = list(csv.DictReader(open("$OUT/%s_counter_collection.csv" % tag)))
    d = collections.defaultdict(list)
    for r in rows:
        d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        if "k_" in k:
            out.setdefault(k, {})[ctr + "_KB_avg_per_launch"] = sum(v) / len(v)
json.dump(out, open("$OUT/pmc_hbm.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
tail -c 400 $OUT/bench.json
