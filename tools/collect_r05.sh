#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: every measurement profiles/r05_* is made from, into gpurun_out/r05/.
# Usage: bash tools/collect_r05.sh [part ...]   parts: main sq final phase slow configs short soak misc waves multi rccl spread behind (default: all but the last two)
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r05
mkdir -p $OUT
PARTS=${@:-main sq final phase slow configs short soak misc waves multi rccl}
has() { [[ " $PARTS " == *" $1 "* ]]; }
stats() {  # <dir> <prefix>: copy the kernel-stats CSV of a rocprofv3 --kernel-trace --stats run
  f=$(find $1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/$2
}
if has main; then
  bash tools/collect_profiles.sh r05 > $OUT/collect_profiles.log 2>&1
  cp gpurun_out/prof_r05/bench.json $OUT/bench.json; cp gpurun_out/prof_r05/bench_under_prof.json $OUT/bench_under_rocprof.json
  cp gpurun_out/prof_r05/pmc_hbm.json $OUT/pmc_hbm.json; stats gpurun_out/prof_r05 kernel_stats.csv
  echo "main done"
fi
if has sq; then
  bash tools/collect_sq.sh r05 > $OUT/collect_sq.log 2>&1; cp gpurun_out/sq_r05/sq.json $OUT/sq_counters.json; echo "sq done"
fi
if has final; then
  # the default bench line once more, now that the counter files of THIS library version exist (bench.py reads roofline.traffic /
  # roofline.valu from profiles/r05_pmc_hbm.json / r05_sq_counters.json and refuses files of another version)
  cp $OUT/pmc_hbm.json profiles/r05_pmc_hbm.json; cp $OUT/sq_counters.json profiles/r05_sq_counters.json
  python3 bench.py > $OUT/bench.json 2> $OUT/bench_final.err; echo "final done"
fi
if has phase; then
  L=gym_kmanip_amd/libkmanip_hip_prof.so
  KMANIP_LIB=$L python3 tools/phase_profile.py newton KManipSoloArm > $OUT/phase_profile.txt 2> $OUT/phase.err
  KMANIP_LIB=$L python3 tools/phase_profile.py newton KManipDualArm > $OUT/phase_profile_dualarm.txt 2>> $OUT/phase.err
  KMANIP_LIB=$L python3 tools/phase_profile.py newton KManipTorso > $OUT/phase_profile_torso.txt 2>> $OUT/phase.err
  echo "phase done"
fi
if has slow; then
  python3 tests/tools/slow_launches.py 512 > $OUT/slow_launches.txt 2> $OUT/slow.err
  python3 tests/tools/slow_launches.py 512 64 > $OUT/slow_launches_ik_max_nfev64.txt 2>> $OUT/slow.err      # the opt-in cap (KModelDesc.ik_max_nfev)
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
    stats $OUT/kt_$tag kernel_stats_$tag.csv; rm -rf $OUT/kt_$tag
  done
  echo "configs done"
fi
if has short; then
  for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline; done > $OUT/bench_short_x3.jsonl 2> $OUT/short.err
  python3 - <<PY > $OUT/bench_short_x3.txt
import json
v = [json.loads(l) for l in open("$OUT/bench_short_x3.jsonl") if l.strip()]
vals = [d["value"] for d in v]
print("three back-to-back  python bench.py --steps 20 --warmup 5 --no-variants --no-cpu-baseline  (the driver's window):")
for d in v: print("  value %.4g env steps/s  ms_per_step %.4f  k_step %.4f ms" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"]["k_step"]))
print("  spread (max - min) / mean = %.2f %%" % (100 * (max(vals) - min(vals)) / (sum(vals) / len(vals))))
PY
  echo "short done"
fi
if has soak; then
  python3 tests/tools/parity_soak.py 4096 200 > $OUT/parity_soak.txt 2> $OUT/soak.err; echo "soak done"
fi
if has misc; then
  tools/_build/libm_check > $OUT/libm_check.txt 2>&1
  python3 tests/tools/render_timing.py 2048 2>/dev/null | grep -v amdgpu.ids > $OUT/render_timing.txt
  python3 tools/kernel_resources.py gym_kmanip_amd/libkmanip_hip.so > $OUT/kernel_resources.txt
  echo "misc done"
fi
if has waves; then
  # per-wave cycles against the cost predictors; the single-arm fit needs a -DKM_WORK_COUNTERS_ALL build
  python3 tests/tools/wave_times_dispatch.py 4096 16 2>/dev/null | grep -v amdgpu > $OUT/wave_times.txt
  python3 tests/tools/wave_times.py KManipDualArm 8192 2>/dev/null | grep -v amdgpu > $OUT/wave_times_dualarm.txt
  KMANIP_COST_SORT=0 python3 tests/tools/wave_times.py KManipDualArm 8192 2>/dev/null | grep -v amdgpu > $OUT/wave_times_dualarm_unsorted.txt
  python3 tests/tools/wave_times.py KManipTorso 8192 2>/dev/null | grep -v amdgpu > $OUT/wave_times_torso.txt
  bash tools/sort_ab.sh > $OUT/sort_ab.txt 2>/dev/null
  echo "waves done"
fi
if has multi; then
  for e in KManipSoloArm KManipDualArm KManipTorso KManipSoloArmVision; do python3 tests/tools/multi_handle_timing.py $e 2>/dev/null | grep handles; done > $OUT/multi_handle_timing.txt
  tools/_build/mfma_ab > $OUT/mfma_ab.json 2>/dev/null
  tools/_build/rsq_check > $OUT/rsq_check_raw.txt 2>/dev/null
  echo "multi done"
fi
ls $OUT
if has rccl; then
  # the N > 1 job's device-collective path as a one-rank RCCL job on this GPU
  for v in "plain:" "rccl_wrapper:--rccl-world1 --gather-direct off" "rccl_wrapper_serial:--rccl-world1 --gather-direct off --gather-serial" "rccl_wrapper_one_channel:--rccl-world1 --gather-direct off --rccl-one-channel" "rccl_direct_stream:--rccl-world1 --gather-direct stream" "rccl_direct_side:--rccl-world1" "rccl_direct_side_ring16:--rccl-world1 --gather-depth 16" "rccl_every8:--rccl-world1 --gather-every 8" "rccl_every64:--rccl-world1 --gather-every 64"; do
    tag=${v%%:*}; fl=${v#*:}
    python3 bench.py $fl --no-variants --no-cpu-baseline > $OUT/rccl_$tag.json 2> $OUT/rccl_$tag.err
  done
  python3 - <<PY > $OUT/rccl_world1.txt
import json
print("KManipSoloArm @ 4096 envs, 1024 timed launches, python bench.py <flags> --no-variants --no-cpu-baseline   (--gather-direct auto = side on RCCL)")
for t, fl in (("plain", "(no process group)"), ("rccl_wrapper", "--rccl-world1 --gather-direct off"), ("rccl_wrapper_serial", "--rccl-world1 --gather-direct off --gather-serial"),
              ("rccl_wrapper_one_channel", "--rccl-world1 --gather-direct off --rccl-one-channel"), ("rccl_direct_stream", "--rccl-world1 --gather-direct stream"),
              ("rccl_direct_side", "--rccl-world1"), ("rccl_direct_side_ring16", "--rccl-world1 --gather-depth 16"),
              ("rccl_every8", "--rccl-world1 --gather-every 8"), ("rccl_every64", "--rccl-world1 --gather-every 64")):
    d = json.loads(open("$OUT/rccl_%s.json" % t).read().strip().split("\n")[-1])
    print("%-52s value %.4g env steps/s  ms/step %.4f  k_step %.4f ms  gap between launches %.4f ms" % (fl, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"]["k_step"], d["roofline"]["kernel_ms_avg"]["launch_gap"]))
PY
  echo "rccl done"
fi
ls $OUT
if has spread; then
  # which envs share a wave: the default against heavy-only flags, table bit only, both bits as one class, and the identity map
  B="python3 bench.py --steps 2048 --warmup 64 --no-variants --no-cpu-baseline"
  for rep in 1 2; do
    for v in "X=1" "KMANIP_SPREAD_TABLE=2" "KMANIP_SPREAD_TABLE=1" "KMANIP_SPREAD_TABLE=0" "KMANIP_SPREAD=0"; do
      echo "== $v"
      env $v $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg']['k_step'])"
    done
  done > $OUT/spread_ab.txt
  python3 tests/tools/ik_persistence.py KManipSoloArm 4096 200 2>/dev/null | grep -v amdgpu > $OUT/ik_persistence.txt
  echo "spread done"
fi
if has behind; then
  # cameras rendered behind the steps (pipeline.RenderBehind): bench's render_behind variants, the overlap probe, the kernel trace
  python3 tests/tools/render_overlap_probe.py 256 2>/dev/null | grep -v amdgpu > $OUT/render_overlap.txt
  rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_behind -o rb -- python3 tests/tools/render_behind_run.py 64 > /dev/null 2>&1
  f=$(find $OUT/kt_behind -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/trace_overlap.py $f > $OUT/render_behind_trace.txt
  rm -rf $OUT/kt_behind
  python3 tests/tools/depth_parity_soak.py 64 60 2>/dev/null | grep -v amdgpu > $OUT/depth_parity_soak.txt
  echo "behind done"
fi
