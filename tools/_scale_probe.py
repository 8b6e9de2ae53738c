import ctypes as C, sys, time, os, tempfile, threading, hashlib
from pathlib import Path
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pyani_plus_amd import _capi
lib = _capi.load_library()
length = 5_000_000
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
rng = np.random.default_rng(1)
text = acgt[rng.integers(0, 4, size=length, dtype=np.uint8)].tobytes()
text = b">g\n" + b"\n".join(text[i:i+100000] for i in range(0, length, 100000)) + b"\n"
cap = lib.pa_pack_bound(C.c_uint64(len(text)))
def run(T, fn, reps):
    ths = [threading.Thread(target=fn, args=(t, reps)) for t in range(T)]
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    return time.perf_counter() - t0
bufs = {}
def pack(t, reps):
    packed = np.empty(cap // 16, np.uint32); mask = np.empty(cap // 32, np.uint32)
    mine = bytes(bytearray(text))  # private copy
    nb = C.c_uint64()
    for _ in range(reps):
        lib.pa_pack_fasta(mine, C.c_uint64(len(mine)), packed.ctypes.data, mask.ctypes.data, C.c_uint64(cap), C.byref(nb), None, None, None)
def md5(t, reps):
    mine = bytes(bytearray(text))
    for _ in range(reps):
        hashlib.md5(mine).digest()
tmp = tempfile.mkdtemp(dir="/tmp")
paths = []
for g in range(512):
    p = Path(tmp) / f"g{g}.fasta"; p.write_bytes(text); paths.append(str(p))
def rd(t, reps):
    buf = bytearray(len(text) + 4096)
    for r in range(reps):
        with open(paths[(t * reps + r) % len(paths)], "rb", buffering=0) as f:
            f.readinto(buf)
for name, fn in (("pack", pack), ("md5", md5), ("read", rd)):
    for T in (1, 16, 64, 128, 256):
        reps = 8 if name != "read" else 2
        dt = run(T, fn, reps)
        print(f"{name} threads={T}: {T * reps * len(text) / dt / 1e9:.2f} GB/s total, {reps * len(text) / dt / 1e9:.3f} per thread", flush=True)
