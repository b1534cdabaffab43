#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: every measurement profiles/r06_* is made from, into gpurun_out/r06/.
# Every file this script writes says which library it measured (VERDICT r5 task 8): JSON files carry config.library /
# _meta.version, text files get a first line "# library: <kmanip_version()>  (tools/collect_r06.sh <part>)", and the rocprofv3 CSV files
# -- kept byte for byte as rocprofv3 wrote them -- are listed with their library and command in MANIFEST.json.
# Usage: bash tools/collect_r06.sh [part ...]   parts: main sq final phase slow configs short soak misc waves multi ab  (default: all but ab)
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r06
mkdir -p $OUT
PARTS=${@:-main sq final phase slow configs short soak misc waves multi}
has() { [[ " $PARTS " == *" $1 "* ]]; }
VER=$(python3 -c "import ctypes; L = ctypes.CDLL('gym_kmanip_amd/libkmanip_hip.so'); L.kmanip_version.restype = ctypes.c_char_p; print(L.kmanip_version().decode())")
stamp() {  # <file> <part>: prepend the library line to a text file; a CSV goes into the manifest instead
  [ -f "$1" ] || return 0
  case "$1" in
    *.csv) python3 - "$1" "$2" <<PY
import json, os, sys
p = "$OUT/MANIFEST.json"
d = json.load(open(p)) if os.path.exists(p) else {}
d["r06_" + os.path.basename(sys.argv[1])] = {"library": "$VER", "made_by": "tools/collect_r06.sh " + sys.argv[2]}
json.dump(d, open(p, "w"), indent=1, sort_keys=True)
PY
    ;;
    *) sed -i "1i # library: $VER  (tools/collect_r06.sh $2)" "$1" ;;
  esac
}
stats() {  # <dir> <name> <part>: copy the kernel-stats CSV of a rocprofv3 --kernel-trace --stats run
  f=$(find $1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/$2 && stamp $OUT/$2 $3
}
if has main; then
  bash tools/collect_profiles.sh r06 > $OUT/collect_profiles.log 2>&1
  cp gpurun_out/prof_r06/bench.json $OUT/bench.json; cp gpurun_out/prof_r06/bench_under_prof.json $OUT/bench_under_rocprof.json
  cp gpurun_out/prof_r06/pmc_hbm.json $OUT/pmc_hbm.json; stats gpurun_out/prof_r06 kernel_stats.csv main
  echo "main done"
fi
if has sq; then
  bash tools/collect_sq.sh r06 > $OUT/collect_sq.log 2>&1; cp gpurun_out/sq_r06/sq.json $OUT/sq_counters.json; echo "sq done"
fi
if has final; then
  # the default bench line once more, now that the counter files of THIS library version exist (bench.py reads roofline.traffic /
  # roofline.valu from profiles/r06_pmc_hbm.json / r06_sq_counters.json and refuses files of another version)
  cp $OUT/pmc_hbm.json profiles/r06_pmc_hbm.json; cp $OUT/sq_counters.json profiles/r06_sq_counters.json
  python3 bench.py > $OUT/bench.json 2> $OUT/bench_final.err; echo "final done"
fi
if has phase; then
  L=gym_kmanip_amd/libkmanip_hip_prof.so
  for e in "KManipSoloArm:" "KManipDualArm:_dualarm" "KManipTorso:_torso"; do
    KMANIP_LIB=$L python3 tools/phase_profile.py newton ${e%%:*} > $OUT/phase_profile${e#*:}.txt 2>> $OUT/phase.err; stamp $OUT/phase_profile${e#*:}.txt "phase (the -DKM_PROFILE build of the same sources)"
  done
  echo "phase done"
fi
if has slow; then
  python3 tests/tools/slow_launches.py 512 > $OUT/slow_launches.txt 2> $OUT/slow.err; stamp $OUT/slow_launches.txt slow
  python3 tests/tools/slow_launches.py 512 64 > $OUT/slow_launches_ik_max_nfev64.txt 2>> $OUT/slow.err; stamp $OUT/slow_launches_ik_max_nfev64.txt slow
  echo "slow done"
fi
if has configs; then
  python3 bench.py --env KManipDualArm --envs-per-gpu 8192 --no-variants > $OUT/bench_dualarm_8192.json 2> $OUT/cfg.err
  python3 bench.py --env KManipTorso --envs-per-gpu 8192 --no-variants > $OUT/bench_torso_8192.json 2>> $OUT/cfg.err
  python3 bench.py --envs-per-gpu 2048 --depth 64 --no-variants > $OUT/bench_config5_depth64.json 2>> $OUT/cfg.err
  python3 bench.py --env KManipSoloArmVision --envs-per-gpu 2048 --steps 256 --no-variants > $OUT/bench_vision_2048.json 2>> $OUT/cfg.err
  for cfg in "dualarm_8192 --env KManipDualArm --envs-per-gpu 8192" "torso_8192 --env KManipTorso --envs-per-gpu 8192" "config5_depth64 --envs-per-gpu 2048 --depth 64" "vision_2048 --env KManipSoloArmVision --envs-per-gpu 2048 --steps 256"; do
    set -- $cfg; tag=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$tag -o ktrace -- python3 bench.py "$@" --no-cpu-baseline --no-variants > /dev/null 2>> $OUT/cfg.err
    stats $OUT/kt_$tag kernel_stats_$tag.csv configs; rm -rf $OUT/kt_$tag
  done
  echo "configs done"
fi
if has short; then
  for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline; done > $OUT/bench_short_x3.jsonl 2> $OUT/short.err
  python3 - <<PY > $OUT/bench_short_x3.txt
import json
v = [json.loads(l) for l in open("$OUT/bench_short_x3.jsonl") if l.strip()]
vals = [d["value"] for d in v]
print("# library: %s  (tools/collect_r06.sh short)" % v[0]["config"]["library"])
print("three back-to-back  python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline  (the driver's window):")
for d in v: print("  value %.4g env steps/s  ms_per_step %.4f  k_step %.4f ms" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"]["k_step"]))
print("  spread (max - min) / mean = %.2f %%" % (100 * (max(vals) - min(vals)) / (sum(vals) / len(vals))))
PY
  echo "short done"
fi
if has soak; then
  python3 tests/tools/parity_soak.py 4096 200 > $OUT/parity_soak.txt 2> $OUT/soak.err; stamp $OUT/parity_soak.txt soak; echo "soak done"
fi
if has misc; then
  python3 tests/tools/render_timing.py 2048 2>/dev/null | grep -v amdgpu.ids > $OUT/render_timing.txt; stamp $OUT/render_timing.txt misc
  python3 tools/kernel_resources.py gym_kmanip_amd/libkmanip_hip.so > $OUT/kernel_resources.txt; stamp $OUT/kernel_resources.txt misc
  echo "misc done"
fi
if has waves; then
  python3 tests/tools/wave_times_dispatch.py 4096 16 2>/dev/null | grep -v amdgpu > $OUT/wave_times.txt; stamp $OUT/wave_times.txt waves
  python3 tests/tools/wave_times.py KManipDualArm 8192 2>/dev/null | grep -v amdgpu > $OUT/wave_times_dualarm.txt; stamp $OUT/wave_times_dualarm.txt waves
  python3 tests/tools/wave_times.py KManipTorso 8192 2>/dev/null | grep -v amdgpu > $OUT/wave_times_torso.txt; stamp $OUT/wave_times_torso.txt waves
  python3 tests/tools/wave_chain_sums.py 256 2>/dev/null | grep -v amdgpu > $OUT/chain_sums.txt; stamp $OUT/chain_sums.txt waves
  echo "waves done"
fi
if has multi; then
  for e in KManipSoloArm KManipDualArm KManipTorso KManipSoloArmVision; do python3 tests/tools/multi_handle_timing.py $e 2>/dev/null | grep handles; done > $OUT/multi_handle_timing.txt; stamp $OUT/multi_handle_timing.txt multi
  echo "multi done"
fi
if has ab; then
  # same-box A/B against the round-5 library (built from commit f256a0c into gym_kmanip_amd/libkmanip_hip_base.so; not shipped)
  B=gym_kmanip_amd/libkmanip_hip_base.so; N=gym_kmanip_amd/libkmanip_hip.so
  { echo "# library: $VER against the round-5 library, same box, alternating (tools/ab.sh; tools/collect_r06.sh ab)"
    echo "== KManipSoloArm @ 4096 (headline), 1024 timed launches"; bash tools/ab.sh $B $N 3
    echo "== KManipDualArm @ 8192, 256 launches"; bash tools/ab.sh $B $N 2 --env KManipDualArm --envs-per-gpu 8192 --steps 256 --warmup 16 --no-variants --no-cpu-baseline
    echo "== KManipTorso @ 8192, 256 launches"; bash tools/ab.sh $B $N 2 --env KManipTorso --envs-per-gpu 8192 --steps 256 --warmup 16 --no-variants --no-cpu-baseline
    echo "== KManipSoloArm @ 2048 + 64x64 depth in the step (config 5), 512 launches"; bash tools/ab.sh $B $N 2 --envs-per-gpu 2048 --depth 64 --steps 512 --warmup 16 --no-variants --no-cpu-baseline
    echo "== KManipSoloArmVision @ 2048, 192 launches"; bash tools/ab.sh $B $N 2 --env KManipSoloArmVision --envs-per-gpu 2048 --steps 192 --warmup 8 --no-variants --no-cpu-baseline
  } > $OUT/ab_vs_r05.txt 2>&1
  echo "ab done"
fi
ls $OUT
