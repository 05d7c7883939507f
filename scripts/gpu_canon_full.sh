#!/bin/bash
# Round 6: the full GPU suite against BOTH builds (fma canon = libhlmi.so, canon 0 = libhlmi_nofma.so), then every app timed
# with both libraries alternating.  Usage: bash scripts/gpu_canon_full.sh <tag>
TAG=${1:-r06full}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
NOFMA=$R/halide_amd/lib/libhlmi_nofma.so
echo "== full GPU suite, fma canon (default library)"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --tb=short 2>&1 | tail -30 | tee $OUT/pytest_fma.log
echo "== full GPU suite, canon 0 (libhlmi_nofma.so)"
HLMI_LIB=$NOFMA timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --tb=short 2>&1 | tail -30 | tee $OUT/pytest_nofma.log
echo "== apps A/B"
for rnd in 1 2; do
  timeout 900 python bench_apps.py --samples 10 --no-batched 2>&1 | grep pipeline | sed "s/^/fma   /" >> $OUT/apps_ab.txt
  HLMI_LIB=$NOFMA timeout 900 python bench_apps.py --samples 10 --no-batched 2>&1 | grep pipeline | sed "s/^/nofma /" >> $OUT/apps_ab.txt
done
python - <<PY
import json
rows = {}
for line in open("$OUT/apps_ab.txt"):
    lib, js = line[:6].strip(), line[6:]
    try: d = json.loads(js)
    except Exception: continue
    rows.setdefault(d["pipeline"], {}).setdefault(lib, []).append(d["ms_per_call"])
for p, v in rows.items():
    print(f"{p:28s} fma {min(v.get('fma', [0])):.4f}  nofma {min(v.get('nofma', [0])):.4f} ms per call")
PY
