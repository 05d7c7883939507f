#!/bin/bash
# Usage: scripts/asm_hist.sh <file.hip> <kernel-name-substring>  — instruction histogram of one gfx950 kernel
ROOT=$(cd $(dirname $0)/.. && pwd); mkdir -p $ROOT/build/asm
F=$1; K=$2; S=$ROOT/build/asm/$(basename $F .hip).s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -I$ROOT/include -I$ROOT/halide_amd/csrc -DHLMI_BUILD -x hip --cuda-device-only -S $F -o $S 2>/dev/null
awk -v k="$K" '/^_Z[A-Za-z0-9_]*:/{on=(index($0,k)>0)} on{print}' $S > $S.$K.s
grep -oE "^\s+(v_|s_|ds_|global_|buffer_|flat_|scratch_)[a-z0-9_]+" $S.$K.s | sort | uniq -c | sort -rn | head -${3:-45}
grep -E "NumVgprs|Occupancy|ScratchSize|TotalNumSgprs" $S.$K.s | head -8
echo "total instrs: $(grep -cE '^\s+(v_|s_|ds_|global_|buffer_|flat_)' $S.$K.s)  valu: $(grep -cE '^\s+v_' $S.$K.s)"
