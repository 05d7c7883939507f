/* bgu_auto_schedule.h — stands in for the header the reference's generator emits next to bgu_auto_schedule.a
 * (src/CodeGen_C.cpp:1050-1112): a driver written against the reference (apps/<app>/process.cpp,
 * filter.cpp, test.cpp) includes this file unchanged and links libhlmi.so instead of the AOT object. */
#ifndef HLMI_AOT_BGU_AUTO_SCHEDULE_H
#define HLMI_AOT_BGU_AUTO_SCHEDULE_H
/* the generated header includes the runtime header first (src/CodeGen_C.cpp:1066); do the same when the
 * caller has it on the include path, otherwise fall back to the layout-identical re-declaration */
#if defined(__has_include)
#if __has_include("HalideRuntime.h")
#include "HalideRuntime.h"
#endif
#endif
#include "../hlmi_pipelines.h"
#include "../hlmi_runtime.h"
#endif
