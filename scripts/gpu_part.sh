#!/bin/bash
export TMPDIR=/tmp
fb() { echo "== frame_bench $1"; shift; env "$@" timeout 200 python scripts/frame_bench.py 8 4 2>&1 | tail -2; }
fb "up32 fused" A=1; fb "unfused" HLMI_LL_FUSE_UP32=0; fb "up32 fused" A=1; fb "unfused" HLMI_LL_FUSE_UP32=0
