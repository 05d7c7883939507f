"""What happens to library-owned buffers whose last stream was destroyed by its owner? (diagnosis)"""
import ctypes as C, gc, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import halide_amd as hl
hip = hl.hip_runtime()   # the runtime libhlmi.so is bound to
s = C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
img = np.random.default_rng(0).integers(0, 65536, (70, 90), dtype=np.uint16)
a, mid, o = hl.Buffer(img), hl.Buffer(np.zeros_like(img)), hl.Buffer(np.zeros_like(img))
hl.set_stream(s.value)
hl.stencil_chain(a, mid)
hl.stencil_chain(mid, o)
hl.set_stream(None)
print("ran on user stream"); sys.stdout.flush()
assert hip.hipStreamSynchronize(s) == 0
print("destroy", hip.hipStreamDestroy(s)); sys.stdout.flush()
step = sys.argv[1] if len(sys.argv) > 1 else "all"
print("numpy of o"); sys.stdout.flush()
r = o.numpy()
print("ok; free buffers"); sys.stdout.flush()
del a, mid, o
gc.collect()
print("ok; new call on the library stream"); sys.stdout.flush()
b, c = hl.Buffer(img), hl.Buffer(np.zeros_like(img))
hl.stencil_chain(b, c)
print("ok", c.numpy().sum()); sys.stdout.flush()
