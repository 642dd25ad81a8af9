"""Diagnostic (not part of the product): host<->device copy rates on this box for
pageable, pinned and registered host memory (what the host layer's batches pay)."""
import ctypes as C, time, numpy as np
hip = C.CDLL("libamdhip64.so")
def chk(r):
    if r != 0: raise RuntimeError(f"hip error {r}")
N = 134 << 20
d = C.c_void_p(); chk(hip.hipMalloc(C.byref(d), C.c_size_t(N)))
a = np.random.randint(0, 255, N, dtype=np.uint8)
b = np.empty(N, dtype=np.uint8)
def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); hip.hipDeviceSynchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
H2D, D2H = 1, 2
print("pageable H2D %.1f ms" % t(lambda: chk(hip.hipMemcpy(d, a.ctypes.data_as(C.c_void_p), C.c_size_t(N), H2D))))
print("pageable D2H %.1f ms" % t(lambda: chk(hip.hipMemcpy(b.ctypes.data_as(C.c_void_p), d, C.c_size_t(N), D2H))))
p = C.c_void_p(); chk(hip.hipHostMalloc(C.byref(p), C.c_size_t(N), 0))
print("pinned   H2D %.1f ms" % t(lambda: chk(hip.hipMemcpy(d, p, C.c_size_t(N), H2D))))
print("pinned   D2H %.1f ms" % t(lambda: chk(hip.hipMemcpy(p, d, C.c_size_t(N), D2H))))
print("memcpy pageable->pinned %.1f ms" % t(lambda: C.memmove(p, a.ctypes.data_as(C.c_void_p), N)))
print("memcpy pinned->pageable %.1f ms" % t(lambda: C.memmove(b.ctypes.data_as(C.c_void_p), p, N)))
def reg():
    chk(hip.hipHostRegister(a.ctypes.data_as(C.c_void_p), C.c_size_t(N), 0))
    chk(hip.hipMemcpy(d, a.ctypes.data_as(C.c_void_p), C.c_size_t(N), H2D))
    chk(hip.hipHostUnregister(a.ctypes.data_as(C.c_void_p)))
print("register + H2D + unregister %.1f ms" % t(reg))
t0 = time.perf_counter(); chk(hip.hipHostRegister(a.ctypes.data_as(C.c_void_p), C.c_size_t(N), 0)); print("register alone %.1f ms" % ((time.perf_counter() - t0) * 1e3))
print("registered H2D %.1f ms" % t(lambda: chk(hip.hipMemcpy(d, a.ctypes.data_as(C.c_void_p), C.c_size_t(N), H2D))))
t0 = time.perf_counter(); chk(hip.hipHostUnregister(a.ctypes.data_as(C.c_void_p))); print("unregister alone %.1f ms" % ((time.perf_counter() - t0) * 1e3))
