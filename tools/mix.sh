#!/bin/bash
# listing + static instruction mix of the headline kernel (k_step<10,16,1,4,false>) or another variant: tools/mix.sh [NL G SOLVER] [symbol] [nhot]
set -e
cd "$(dirname "$0")/../gym_kmanip_amd/csrc"
NL=${1:-10}; G=${2:-16}; S=${3:-1}; SYM=${4:-_Z6k_stepILi${NL}ELi${G}ELi${S}ELi$((64 / G))ELb0E}; NHOT=${5:-0}
OUT=../../tools/_build/dyn_${NL}_${G}_${S}.s
mkdir -p ../../tools/_build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast-honor-pragmas -DKM_VAR_NL=$NL -DKM_VAR_G=$G -DKM_VAR_SOLVER=$S $EXTRA \
  -S -gline-tables-only --cuda-device-only kmanip_dyn.hip -o $OUT 2>/dev/null
python3 ../../tools/asm_mix.py $OUT $SYM $NHOT
grep -A30 "^$SYM" $OUT >/dev/null
awk -v s="$SYM" '$0 ~ "^; Kernel" {k=0} $0 ~ s && /\.name:/ {f=1} f && /vgpr_count|agpr_count|private_segment_fixed_size|sgpr_spill|vgpr_spill/ {print} /\.wavefront_size/ {f=0}' $OUT | head -8
