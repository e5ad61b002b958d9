"""Diagnostic: which HIP runtime does libsoftrod_hip.so bind to, and does softrod_create
work before/after torch has initialised its context?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "hsa-runtime" in l})
if mode == "lib_first":
    from gym_softrobot_amd import _capi
    lib = _capi.load_library()
    print("after lib:", maps())
    import torch
    print("after torch:", maps())
else:
    import torch
    print("after torch:", maps())
    if mode == "torch_init":
        torch.zeros(1, device="cuda")
    elif mode == "torch_avail":
        print("avail", torch.cuda.is_available())
    from gym_softrobot_amd import _capi
    lib = _capi.load_library()
    print("after lib:", maps())
cfg = _capi.softpendulum_config(4)
h = C.c_void_p()
rc = lib.softrod_create(C.byref(cfg), 0, C.byref(h))
print(mode, "create rc", rc, lib.softrod_last_error(None))
