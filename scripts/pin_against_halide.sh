#!/bin/bash
# Runs every case of tests/golden/halide/manifest.json through the REFERENCE's own runner and leaves Halide's outputs in
# tests/golden/halide/<case>.npy, where tests/test_reference_goldens.py finds them.  See scripts/pin_against_halide.md.
#   HALIDE_RUNGEN_DIR = directory holding <app>.rungen for the apps of the manifest (a real Halide build; this container has none)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
: "${HALIDE_RUNGEN_DIR:?set HALIDE_RUNGEN_DIR to the directory with the <app>.rungen binaries of a Halide build}"
python3 "$ROOT/scripts/pin_against_halide.py" inputs || exit 1
fail=0
while IFS= read -r cmd; do
  exe=${cmd%% *}
  if [ ! -x "$exe" ]; then echo "MISSING  $exe"; fail=1; continue; fi
  echo "RUN      $cmd"
  eval "$cmd" || { echo "FAILED   $exe"; fail=1; }
done < <(HALIDE_RUNGEN_DIR="$HALIDE_RUNGEN_DIR" python3 "$ROOT/scripts/pin_against_halide.py" commands | sed "s|\$HALIDE_RUNGEN_DIR|$HALIDE_RUNGEN_DIR|")
# harris, lens_blur, bgu: the apps' own drivers (the manifest's "driver_cases"); optional — without HALIDE_DRIVER_DIR they stay unpinned
if [ -n "${HALIDE_DRIVER_DIR:-}" ]; then
  while IFS= read -r cmd; do
    exe=${cmd%% *}
    if [ ! -x "$exe" ]; then echo "MISSING  $exe"; fail=1; continue; fi
    echo "RUN      $cmd"
    eval "$cmd" > /dev/null || { echo "FAILED   $exe"; fail=1; }
  done < <(python3 "$ROOT/scripts/pin_against_halide.py" driver_commands | sed "s|\$HALIDE_DRIVER_DIR|$HALIDE_DRIVER_DIR|")
else
  echo "NOTE     HALIDE_DRIVER_DIR is not set: harris / lens_blur / bgu (driver cases) not run"
fi
ls -la "$ROOT/tests/golden/halide" | head -40
exit $fail
