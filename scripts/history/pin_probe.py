import ctypes, time, numpy as np
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
n = 1 << 30
d = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(d), n) == 0
a = np.zeros(n // 8); a[:] = 1.0          # touched
for rep in range(2):
    t = time.perf_counter(); assert hip.hipMemcpy(a.ctypes.data, d, n, 2) == 0; hip.hipDeviceSynchronize(); t1 = time.perf_counter() - t
    t = time.perf_counter(); assert hip.hipMemcpy(d, a.ctypes.data, n, 1) == 0; hip.hipDeviceSynchronize(); t2 = time.perf_counter() - t
    print("pageable D2H %.1f GB/s  H2D %.1f GB/s" % (n / 1e9 / t1, n / 1e9 / t2))
t = time.perf_counter(); rc = hip.hipHostRegister(a.ctypes.data, n, 0); tr = time.perf_counter() - t
print("hipHostRegister rc", rc, "%.1f ms for 1 GiB (%.1f GB/s)" % (tr * 1e3, n / 1e9 / tr))
for rep in range(2):
    t = time.perf_counter(); assert hip.hipMemcpy(a.ctypes.data, d, n, 2) == 0; hip.hipDeviceSynchronize(); t1 = time.perf_counter() - t
    t = time.perf_counter(); assert hip.hipMemcpy(d, a.ctypes.data, n, 1) == 0; hip.hipDeviceSynchronize(); t2 = time.perf_counter() - t
    print("pinned   D2H %.1f GB/s  H2D %.1f GB/s" % (n / 1e9 / t1, n / 1e9 / t2))
t = time.perf_counter(); hip.hipHostUnregister(a.ctypes.data); print("unregister %.1f ms" % ((time.perf_counter() - t) * 1e3))
