echo "== probe"; HLMI_LIB=$GRAFT_REPO_ROOT/halide_amd/lib/libhlmi_probe.so timeout 120 python scripts/ll_probe.py
