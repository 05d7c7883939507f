/* canon_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle_common.h).
 *
 * The switch between the oracle's two canonical forms (oracle_common.h header): 0 = one rounding per operator,
 * 1 = mul+add pairs contracted into fma the way LLVM's DAG combiner contracts them under the fast-math flags the
 * reference sets (/root/reference/src/CodeGen_LLVM.cpp:483-500, src/CodeGen_Internal.cpp:614).  The default is the
 * form libhlmi.so is built for by default (halide_amd/csrc/hlmi_device_math.h: HLMI_CANON_FMA = 1); the tests set it
 * from the loaded library's hlmi_canon_fma() (tests/conftest.py). */
#include "oracle_common.h"

int o_canon_fma = 1;
void oracle_set_canon(int fma) { o_canon_fma = fma ? 1 : 0; }
int oracle_get_canon(void) { return o_canon_fma; }

int o_reassoc = 0;
void oracle_set_reassoc(int on) { o_reassoc = on ? 1 : 0; }
