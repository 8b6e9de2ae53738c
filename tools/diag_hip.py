"""Diagnose which HIP runtime the extension binds to (run on the GPU box)."""
import ctypes, subprocess, sys, os

def maps():
    out = set()
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line or "libhsa-runtime" in line:
            out.add(line.split()[-1])
    return sorted(out)

mode = sys.argv[1] if len(sys.argv) > 1 else "all"
if mode == "all":
    for m in ("lib_first", "torch_first"):
        print("=====", m, flush=True)
        subprocess.run([sys.executable, __file__, m])
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyani_plus_amd import _capi
if mode == "lib_first":
    lib = _capi.load_library()
    print("pa_device_count:", lib.pa_device_count(), maps(), flush=True)
    ctx = ctypes.c_void_p()
    print("ctx_create:", lib.pa_ctx_create(0, ctypes.byref(ctx)), lib.pa_last_error())
else:
    import torch
    print("torch avail:", torch.cuda.is_available(), torch.cuda.device_count(), maps(), flush=True)
    x = torch.ones(4, device="cuda"); print(x.sum().item())
    lib = _capi.load_library()
    print("pa_device_count:", lib.pa_device_count(), maps(), flush=True)
    ctx = ctypes.c_void_p()
    print("ctx_create:", lib.pa_ctx_create(0, ctypes.byref(ctx)), lib.pa_last_error())
