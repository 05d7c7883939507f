"""Does an event recorded on hipStreamLegacy survive hipStreamWaitEvent on another stream?  (diagnosis of the LUT-cache crash)"""
import ctypes as C, sys
hip = C.CDLL("libamdhip64.so")
ev, s = C.c_void_p(), C.c_void_p()
print("create", hip.hipEventCreateWithFlags(C.byref(ev), 2), hip.hipStreamCreateWithFlags(C.byref(s), 1)); sys.stdout.flush()
for name, h in (("null", None), ("legacy", 1), ("perthread", 2)):
    r = hip.hipEventRecord(ev, C.c_void_p(h)); print(name, "record", r); sys.stdout.flush()
    r = hip.hipStreamWaitEvent(s, ev, 0); print(name, "wait", r); sys.stdout.flush()
    print(name, "sync", hip.hipStreamSynchronize(s)); sys.stdout.flush()
