"""Thread sweep of the FASTA front-end on synthetic files.

    python tools/front_end_sweep.py [n_files=400] [length=5000000] [gz]
Times load_fasta_files (read + md5 + parse + 2-bit pack + arena copy) per thread count, into pageable and into
page-locked memory, and prints GB of FASTA text per second.
"""
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pyani_plus_amd.engine import load_fasta_files  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
length = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
gz = len(sys.argv) > 3 and sys.argv[3] == "gz"
rng = np.random.default_rng(7)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
    base = acgt[rng.integers(0, 4, size=length, dtype=np.uint8)]
    paths = []
    for g in range(n):
        seq = np.roll(base, g * 997)
        path = Path(tmp) / (f"genome_{g:05d}.fasta" + (".gz" if gz else ""))
        width = 80 if gz else 100_000
        text = b">genome_%d synthetic\n" % g + b"\n".join(seq[i : i + width].tobytes() for i in range(0, length, width)) + b"\n"
        if gz:
            import zlib

            comp = zlib.compressobj(6, zlib.DEFLATED, 31)
            text = comp.compress(text) + comp.flush()
        path.write_bytes(text)
        paths.append(path)
    cores = len(os.sched_getaffinity(0))
    print(f"{n} files of {length} bases{' (gzip)' if gz else ''}, {cores} cores", flush=True)
    for pinned in (False, True):
        for threads in sorted({8, 16, 32, 64, 128, cores}):
            if threads > cores:
                continue
            best = 1e9
            for _rep in range(2):
                t0 = time.perf_counter()
                infos, arena = load_fasta_files(paths, threads, pinned=pinned)
                best = min(best, time.perf_counter() - t0)
                del arena
            assert all(i.status == 0 for i in infos)
            print(f"pinned={pinned} threads={threads}: {best:.3f} s = {n * length / best / 1e9:.2f} GB/s", flush=True)
    # where the time of one call goes (all cores)
    import ctypes as C

    from pyani_plus_amd import _capi

    lib = _capi.load_library()
    arr = (C.c_char_p * n)(*[str(p).encode() for p in paths])
    for threads in (16, 64, cores):
        batch = C.c_void_p()
        t0 = time.perf_counter()
        _capi.check(lib.pa_fasta_batch_load(arr, n, threads, C.byref(batch)), "load")
        t1 = time.perf_counter()
        total = int(lib.pa_fasta_batch_arena_bases(batch))
        packed = np.empty(total // 16, dtype=np.uint32)
        mask = np.empty(total // 32, dtype=np.uint32)
        starts = np.zeros(n + 1, dtype=np.uint64)
        t2 = time.perf_counter()
        _capi.check(lib.pa_fasta_batch_copy_arena(batch, packed.ctypes.data, mask.ctypes.data, starts.ctypes.data), "copy")
        t3 = time.perf_counter()
        lib.pa_fasta_batch_free(batch)
        t4 = time.perf_counter()
        del packed, mask
        t5 = time.perf_counter()
        print(f"threads={threads}: load {t1 - t0:.3f}, alloc {t2 - t1:.3f}, copy_arena {t3 - t2:.3f}, batch_free {t4 - t3:.3f}, arena_free {t5 - t4:.3f}", flush=True)
