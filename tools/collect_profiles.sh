#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: bench line + rocprofv3 kernel stats of the SAME command + PMC HBM
# traffic (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes) -> gpurun_out/prof_<tag>/
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ktrace -- python3 bench.py --no-cpu-baseline --no-variants > $OUT/bench_under_prof.json 2> $OUT/ktrace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-variants > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-variants > /dev/null 2> $OUT/write.err
python3 - <<PY
import csv, collections, json, glob, ctypes, os
out = {}
for tag, ctr in [("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")]:
    files = glob.glob("$OUT/**/%s_counter_collection.csv" % tag, recursive=True)
    d = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        if "k_" in k:
            out.setdefault(k, {})[ctr + "_KB_avg_per_launch"] = sum(v[-16:]) / len(v[-16:])
L = ctypes.CDLL(os.path.join("gym_kmanip_amd", "libkmanip_hip.so")); L.kmanip_version.restype = ctypes.c_char_p
out["_meta"] = {"version": L.kmanip_version().decode(), "averaged": "last 16 launches per kernel (steady state)"}
json.dump(out, open("$OUT/pmc_hbm.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name "*kernel_stats.csv" | head -3
tail -c 300 $OUT/bench.json
