import ctypes as C, os, sys
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "msm_amd", "libmsm_amd.so"))
print("device_count without torch:", lib.msm_amd_device_count())
hip = C.CDLL("libamdhip64.so")
n = C.c_int(-1)
rc = hip.hipGetDeviceCount(C.byref(n))
hip.hipGetErrorString.restype = C.c_char_p
print("hipGetDeviceCount rc", rc, hip.hipGetErrorString(rc), "n", n.value)
os.system("grep -i hip /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
