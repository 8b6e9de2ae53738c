"""ctypes binding of ``libpyani_hip.so`` (declared in ``include/pyani_hip.h``).

This is the only route from Python into the compute path.  There is no CPU
fallback: if the shared library is missing, or no gfx950 device is usable,
the calls raise ``HipBackendError``.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "_lib" / "libpyani_hip.so"
# The same sources built with -DPA_TOOLS: the environment switches that force a rare path, cut a kernel short or select an
# ablation variant (PA_MAP_CUT, PA_FRAGANI_*, PA_KMER_*, PA_PAIRS_SYMMETRIC, ...) exist in this build only.  tools/ and
# the tests of those paths load it (``HipEngine(tools=True)``); nothing else does.
TOOLS_LIB_PATH = _PKG / "_lib" / "libpyani_hip_tools.so"

ABI_VERSION = 5
PA_OK = 0
PA_E_NOMEM = -3
PA_E_CAPACITY = -4
PA_E_IO = -6
PA_SIG_UNHANDLED = 1
PA_FRAGANI_REUSE_INDEX = 1
PA_FRAGANI_COLUMNS_ONLY = 2
PA_PAIRS_AUTO, PA_PAIRS_BITROW, PA_PAIRS_MERGE, PA_PAIRS_BITROW_HASH = 0, 1, 2, 3
PA_ALIGN_BASES = 64
PROF_PHASES = {"kmer_hash": 0, "sketch_sort": 1, "pair_dict": 2, "pair_count": 3, "ani": 4, "frag_index": 5, "frag_seed": 6, "frag_map": 7}

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_f64p = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/pyani_hip.h one to one
SIGNATURES: dict[str, tuple] = {
    "pa_abi_version": (C.c_int, []),
    "pa_last_error": (C.c_char_p, []),
    "pa_device_count": (C.c_int, []),
    "pa_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "pa_ctx_destroy": (None, [_vp]),
    "pa_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "pa_ctx_own_stream": (C.c_int, [_vp]),
    "pa_ctx_sync": (C.c_int, [_vp]),
    "pa_ctx_device_info": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_int), _u64p]),
    "pa_dev_alloc": (C.c_int, [_vp, C.c_uint64, C.POINTER(_vp)]),
    "pa_dev_free": (C.c_int, [_vp, _vp]),
    "pa_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "pa_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    "pa_memset_d": (C.c_int, [_vp, _vp, C.c_int, C.c_uint64]),
    "pa_pack_bound": (C.c_uint64, [C.c_uint64]),
    "pa_pack_fasta": (C.c_int, [_vp, C.c_uint64, _vp, _vp, C.c_uint64, _u64p, _u64p, _u64p, _u64p]),
    "pa_pack_seq": (C.c_int, [_vp, C.c_uint64, _vp, _vp, C.c_uint64, _u64p, _u64p]),
    "pa_max_hash": (C.c_uint64, [C.c_uint64]),
    "pa_fasta_batch_load": (C.c_int, [C.POINTER(C.c_char_p), C.c_uint32, C.c_int, C.POINTER(_vp)]),
    "pa_fasta_batch_info": (
        C.c_int,
        [_vp, C.c_uint32, C.c_char_p, _u64p, _u64p, _u64p, _u64p, _u64p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int)],
    ),
    "pa_fasta_batch_records": (C.c_int, [_vp, C.c_uint32, C.POINTER(_u64p), C.POINTER(_u64p), _u64p]),
    "pa_fasta_batch_arena_bases": (C.c_uint64, [_vp]),
    "pa_fasta_batch_copy_arena": (C.c_int, [_vp, _vp, _vp, _vp]),
    "pa_fasta_batch_free": (None, [_vp]),
    "pa_arena_dirty": (C.c_int, [_vp, _vp, C.c_uint64, _vp]),
    "pa_sketch": (
        C.c_int,
        [_vp, _vp, _vp, _vp, C.c_uint64, _u64p, C.c_uint32, C.c_uint32, C.c_uint64, _vp, C.c_uint64, _vp, _u64p],
    ),
    "pa_read_sigs": (C.c_int, [C.POINTER(C.c_char_p), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(_vp)]),
    "pa_sig_batch_info": (C.c_int, [_vp, C.c_uint32, _u64p, C.POINTER(C.c_char_p)]),
    "pa_sig_batch_copy": (C.c_int, [_vp, _vp, _vp]),
    "pa_sig_batch_free": (None, [_vp]),
    "pa_write_sigs": (C.c_int, [C.c_uint32, _vp, _vp, _vp, _vp, C.c_uint32, _vp, _vp, C.c_uint32]),
    "pa_mask_runs": (C.c_int64, [_vp, C.c_uint64, _vp, _vp, C.c_uint64]),
    "pa_mask_from_runs": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, C.c_uint64]),
    "pa_sketch_streamed": (
        C.c_int,
        [_vp, _vp, _vp, _vp, C.c_uint32, C.c_uint64, _u64p, C.c_uint32, C.c_uint32, C.c_uint64, _vp, _vp, _vp, _vp, C.c_uint64, _vp, _u64p],
    ),
    "pa_pair_counts": (
        C.c_int,
        [_vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_int],
    ),
    "pa_pair_counts_ex": (
        C.c_int,
        [_vp, _vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_int],
    ),
    "pa_pair_dict_prepare": (C.c_int, [_vp, _vp, C.c_uint64]),
    "pa_ani": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp]),
    "pa_ani_host": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, C.c_int, C.c_uint32]),
    "pa_sketch_bottom": (
        C.c_int,
        [_vp, _vp, _vp, _vp, C.c_uint64, _u64p, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint64, _vp, _u64p],
    ),
    "pa_pair_mash": (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp]),
    "pa_ani_mash": (C.c_int, [_vp, _vp, _vp, C.c_uint64, C.c_uint32, _vp]),
    "pa_fragani": (C.c_int, [_vp, _vp, _vp, C.c_uint64, _vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    "pa_fragani_ex": (
        C.c_int,
        [_vp, _vp, _vp, C.c_uint64, _vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
         C.c_uint32, C.c_uint32, _vp, _vp, _vp],
    ),
    "pa_fragani_sketch": (
        C.c_int,
        [_vp, _vp, _vp, C.c_uint64, _vp, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp, _vp, _vp, C.c_uint64, _u64p],
    ),
    "pa_text_ambiguous": (C.c_int64, [_vp, C.c_uint64, C.c_int, _vp, _vp, C.c_uint64]),
    "pa_fasta_batch_ambiguous": (C.c_int64, [_vp, _vp, _vp, C.c_uint64]),
    "pa_fragani_set_ambiguous": (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint64]),
    "pa_fragani_window": (C.c_int, [C.c_uint32, C.c_uint32]),
    "pa_fragani_workspace": (C.c_int, [_vp, _u64p, _u64p, _u64p]),
    "pa_fragani_set_workspace_cap": (C.c_int, [_vp, C.c_uint64]),
    "pa_fragani_tables": (C.c_int, [C.c_uint32, C.c_uint32, _vp, _vp]),
    "pa_fragani_identity": (C.c_double, [C.c_uint32, C.c_uint32, C.c_uint32]),
    "pa_fasta_records": (C.c_int64, [_vp, C.c_uint64, _vp, _vp, C.c_uint64]),
    "pa_write_comparisons_json": (
        C.c_int,
        [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_char_p), C.c_uint32, _vp, _vp, _vp],
    ),
    "pa_append_comparisons_json": (
        C.c_int,
        [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_char_p), C.c_uint32, _vp, _vp, _vp],
    ),
    "pa_append_comparisons_json_ex": (
        C.c_int,
        [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_char_p), C.c_uint32, _vp, _vp, _vp, _vp, _vp],
    ),
    "pa_round_sig6": (C.c_int, [_vp, C.c_uint64]),
    "pa_host_cpu_budget": (C.c_uint32, []),
    "pa_gunzip": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, _u64p, C.c_int]),
    "pa_sqlite_insert_comparisons": (
        C.c_int,
        [C.c_char_p, C.c_int64, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_char_p), C.c_uint32, _vp, _vp, _vp, _u64p],
    ),
    "pa_sqlite_insert_comparisons_ex": (
        C.c_int,
        [C.c_char_p, C.c_int64, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_char_p), C.c_uint32, _vp, _vp, _vp,
         _vp, _vp, _u64p],
    ),
    "pa_prof_enable": (C.c_int, [_vp, C.c_int]),
    "pa_prof_reset": (C.c_int, [_vp]),
    "pa_prof_get": (C.c_int, [_vp, C.c_int, _f64p, _u64p]),
}


class HipBackendError(RuntimeError):
    """The HIP extension is missing, failed to load, or a call into it failed."""


_libs: dict[bool, C.CDLL] = {}


def build_library(force: bool = False) -> Path:
    """Compile ``libpyani_hip.so`` and ``libpyani_hip_tools.so`` for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcdir = _PKG / "csrc"
    cmd = ["make", "-C", str(srcdir), "-j", str(min(8, os.cpu_count() or 1))]
    if force:
        cmd.append("-B")
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0 or not LIB_PATH.is_file() or not TOOLS_LIB_PATH.is_file():
        raise HipBackendError(f"building libpyani_hip.so failed:\n{proc.stdout}\n{proc.stderr}")
    return LIB_PATH


def load_library(tools: bool = False) -> C.CDLL:
    """Load the shared library (``tools``: the -DPA_TOOLS build) and type every symbol of the header; raise loudly if absent."""
    if tools in _libs:
        return _libs[tools]
    path = TOOLS_LIB_PATH if tools else LIB_PATH
    if not path.is_file():
        raise HipBackendError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C pyani_plus_amd/csrc`).  There is no CPU fallback for the compute path."
        )
    # PyTorch's ROCm wheel bundles its own libamdhip64/libhsa-runtime64.  Two HIP runtimes in one
    # process cannot both own the GPU, so torch goes first and this library (same soname,
    # libamdhip64.so.7) binds to the runtime torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover - plain ROCm install without torch
        pass
    try:
        lib = C.CDLL(str(path))
    except OSError as err:  # missing ROCm runtime, wrong arch, ...
        raise HipBackendError(f"cannot load {path}: {err}") from err
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as err:
            raise HipBackendError(f"{path} does not export {name}") from err
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.pa_abi_version() != ABI_VERSION:
        raise HipBackendError(f"ABI version mismatch: library reports {lib.pa_abi_version()}, binding expects {ABI_VERSION}")
    _libs[tools] = lib
    return lib


def last_error(lib: C.CDLL | None = None) -> str:
    return ((lib or load_library()).pa_last_error() or b"").decode(errors="replace")


def check(status: int, what: str, lib: C.CDLL | None = None) -> None:
    if status != PA_OK:
        err = HipBackendError(f"{what} failed with status {status}: {last_error(lib)}")
        err.status = status  # the library's code (PA_E_*), for callers that tell failures apart
        raise err
