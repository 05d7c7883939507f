// rungen_registration.cpp — TEST INFRASTRUCTURE.  The registration translation unit the reference emits per
// pipeline with `-e registration` (/root/reference/src/Module.cpp:171-201), written against libhlmi.so: one static
// registerer per pipeline PIPELINE names (passed as -DPIPELINE=<name> by oracle/ref.mk).  Linked with the reference's
// own tools/RunGenMain.cpp, compiled unmodified, it gives `<name>.rungen` — the reference's consumer of
// `<name>_argv` + `<name>_metadata` + the bounds-query protocol (tools/RunGen.h:1212-1250, 1384-1430).
#ifndef PIPELINE
#error "compile with -DPIPELINE=<entry point name>"
#endif
#define HLMI_CAT2(a, b) a##b
#define HLMI_CAT(a, b) HLMI_CAT2(a, b)

struct halide_filter_metadata_t;
extern "C" int HLMI_CAT(PIPELINE, _argv)(void **args);
extern "C" const struct halide_filter_metadata_t *HLMI_CAT(PIPELINE, _metadata)();
extern "C" void halide_register_argv_and_metadata(int (*filter_argv_call)(void **), const struct halide_filter_metadata_t *filter_metadata,
                                                  const char *const *extra_key_value_pairs);
namespace {
struct Registerer {
    Registerer() { halide_register_argv_and_metadata(HLMI_CAT(PIPELINE, _argv), HLMI_CAT(PIPELINE, _metadata)(), nullptr); }
} registerer;
}  // namespace
