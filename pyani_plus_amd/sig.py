"""sourmash-compatible ``.sig`` files.

The reference caches one signature per genome as
``<cache>/sourmash_k=<K>_<extra>/<md5>.sig`` (pyani_plus/methods/sourmash.py:57-66)
and its tests compare every JSON key of those files
(tests/snakemake/test_sourmash_workflow.py:43-67).  Writing the same format lets
the HIP backend and the sourmash backend share a cache and resume each other's runs.
"""

from __future__ import annotations

import hashlib
import json
from pathlib import Path

import numpy as np


def signature_md5(ksize: int, mins) -> str:
    """sourmash's ``md5sum`` of a MinHash: md5(str(ksize) + concatenated decimal hashes)."""
    digest = hashlib.md5(str(int(ksize)).encode())  # noqa: S324 - fingerprint, not security
    digest.update("".join(str(int(h)) for h in mins).encode())
    return digest.hexdigest()


def write_sig(path: Path, *, name: str, filename: str, ksize: int, max_hash: int, mins) -> None:
    """Write a single-sketch DNA signature exactly as ``sourmash scripts singlesketch`` lays it out."""
    mins_list = [int(h) for h in mins]
    obj = [
        {
            "class": "sourmash_signature",
            "email": "",
            "hash_function": "0.murmur64",
            "filename": filename,
            "name": name,
            "license": "CC0",
            "signatures": [
                {
                    "num": 0,
                    "ksize": int(ksize),
                    "seed": 42,
                    "max_hash": int(max_hash),
                    "mins": mins_list,
                    "md5sum": signature_md5(ksize, mins_list),
                    "molecule": "DNA",
                }
            ],
            "version": 0.4,
        }
    ]
    tmp = Path(str(path) + ".tmp")
    tmp.write_text(json.dumps(obj, separators=(",", ":"), ensure_ascii=False), encoding="utf-8")  # single line, no trailing newline; raw UTF-8 as sourmash (serde_json) writes it
    tmp.replace(path)


def write_sigs(paths, *, names, filenames, ksize: int, max_hash: int, sketches) -> None:
    """Write many signatures at once through the native threaded writer (``pa_write_sigs``).

    Same bytes as ``write_sig`` for every file: everything around the hash list is formatted here with
    ``json.dumps``, the decimal hashes and the sketch md5 natively."""
    import ctypes as C

    from . import _capi

    n = len(paths)
    if n == 0:
        return
    lib = _capi.load_library()
    heads, mids, tails = [], [], []
    for name, filename in zip(names, filenames):
        obj = [
            {
                "class": "sourmash_signature",
                "email": "",
                "hash_function": "0.murmur64",
                "filename": filename,
                "name": name,
                "license": "CC0",
                "signatures": [
                    {"num": 0, "ksize": int(ksize), "seed": 42, "max_hash": int(max_hash), "mins": [], "md5sum": "", "molecule": "DNA"}
                ],
                "version": 0.4,
            }
        ]
        text = json.dumps(obj, separators=(",", ":"), ensure_ascii=False)  # non-ASCII file names stay raw UTF-8, as in sourmash's files
        marker = '"mins":[],"md5sum":""'
        at = text.rindex(marker)  # the last occurrence is the real one even if a file name contains the marker
        heads.append(text[: at + len('"mins":[')].encode())
        mids.append(b'],"md5sum":"')
        tails.append(text[at + len(marker) - 1 :].encode())
    off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in sketches], out=off[1:])
    flat = np.concatenate([np.asarray(s, dtype=np.uint64) for s in sketches]) if int(off[-1]) else np.zeros(1, dtype=np.uint64)
    flat = np.ascontiguousarray(flat)

    def c_strings(items):
        return (C.c_char_p * n)(*items)

    path_bytes = [str(p).encode() for p in paths]
    _capi.check(
        lib.pa_write_sigs(
            n, c_strings(path_bytes), c_strings(heads), c_strings(mids), c_strings(tails), int(ksize), flat.ctypes.data, off.ctypes.data, 0
        ),
        "pa_write_sigs",
    )


def _read_single_sketch_fast(text: str, ksize: int | None, max_hash: int | None):
    """Files with exactly one sketch (what singlesketch and this backend write): parse the long
    ``mins`` list with numpy instead of the JSON decoder; anything unusual falls back to json."""
    if text.count('"mins":[') != 1 or text.count('"signatures":[') != 1:
        return None
    a = text.index('"mins":[') + len('"mins":[')
    b = text.index("]", a)
    body = text[a:b]
    try:
        head = json.loads(text[: a - len('"mins":[')] + '"mins":[]' + text[b + 1 :])
        sketch = head[0]["signatures"][0]
        if sketch.get("molecule", "DNA") != "DNA" or sketch.get("num", 0) != 0:
            return None
        if (ksize is not None and sketch.get("ksize") != ksize) or (max_hash is not None and sketch.get("max_hash") != max_hash):
            return None
        mins = np.array(body.split(","), dtype=np.uint64) if body else np.empty(0, dtype=np.uint64)
    except (ValueError, KeyError, IndexError, TypeError):
        return None
    if mins.size > 1 and not bool(np.all(mins[1:] > mins[:-1])):
        mins = np.unique(mins)
    sketch["mins"] = mins.tolist() if mins.size <= 16 else None  # large lists are returned only as the array
    return mins, sketch


def _check_sketch_md5(path, sketch: dict, mins: np.ndarray) -> None:
    """A sketch's ``md5sum`` is md5(str(ksize) + its decimal hashes); a file whose list does not give its own
    checksum has been damaged (the native reader applies the same test)."""
    want = sketch.get("md5sum")
    if isinstance(want, str) and len(want) == 32 and "ksize" in sketch and signature_md5(sketch["ksize"], mins) != want:
        msg = f"{path}: md5sum {want} does not match the hashes listed"
        raise ValueError(msg)


def read_sigs(paths, *, ksize: int, max_hash: int, threads: int = 0) -> list[np.ndarray]:
    """The sketches of many one-sketch signature files through the native threaded reader (``pa_read_sigs``): what
    ``sourmash sig collect`` + ``manysearch`` do with a column worker's N cached files in the reference
    (pyani_plus/methods/sourmash.py:160-200).  Files in another layout go through ``read_sig``; an unreadable or
    damaged file raises ``ValueError`` / ``OSError`` naming it."""
    import ctypes as C

    from . import _capi

    paths = [Path(p) for p in paths]
    n = len(paths)
    if n == 0:
        return []
    lib = _capi.load_library()
    arr = (C.c_char_p * n)(*[str(p).encode() for p in paths])
    batch = C.c_void_p()
    _capi.check(lib.pa_read_sigs(arr, n, int(ksize), int(max_hash), int(threads), C.byref(batch)), "pa_read_sigs")
    try:
        sizes = np.zeros(n, dtype=np.uint64)
        status = []
        for i in range(n):
            n_mins, msg = C.c_uint64(0), C.c_char_p()
            st = lib.pa_sig_batch_info(batch, i, C.byref(n_mins), C.byref(msg))
            if st < 0:
                text = (msg.value or b"").decode(errors="replace")
                if st == _capi.PA_E_IO:
                    raise OSError(text)
                msg_text = f"{paths[i]}: {text}"
                raise ValueError(msg_text)
            status.append(st)
            sizes[i] = n_mins.value
        off = np.zeros(n + 1, dtype=np.uint64)
        flat = np.empty(max(int(sizes.sum()), 1), dtype=np.uint64)
        _capi.check(lib.pa_sig_batch_copy(batch, flat.ctypes.data, off.ctypes.data), "pa_sig_batch_copy")
    finally:
        lib.pa_sig_batch_free(batch)
    out = []
    for i, path in enumerate(paths):
        if status[i] == _capi.PA_OK:
            out.append(flat[int(off[i]) : int(off[i + 1])])
        else:  # PA_SIG_UNHANDLED: several sketches in one file, other parameters beside the wanted ones, ...
            out.append(read_sig(path, ksize=ksize, max_hash=max_hash)[0])
    return out


def read_sig(path: Path, *, ksize: int | None = None, max_hash: int | None = None) -> tuple[np.ndarray, dict]:
    """Return (ascending uint64 hashes, sketch dict) of the DNA sketch with the wanted k."""
    text = Path(path).read_text(encoding="utf-8")
    fast = _read_single_sketch_fast(text, ksize, max_hash)
    if fast is not None:
        _check_sketch_md5(path, fast[1], fast[0])
        return fast
    data = json.loads(text)
    if not isinstance(data, list) or not data:
        msg = f"{path} is not a sourmash signature file"
        raise ValueError(msg)
    for entry in data:
        for sketch in entry.get("signatures", []):
            if sketch.get("molecule", "DNA") != "DNA":
                continue
            if ksize is not None and sketch.get("ksize") != ksize:
                continue
            if max_hash is not None and sketch.get("max_hash") != max_hash:
                continue
            if sketch.get("num", 0) != 0:
                msg = f"{path}: only scaled (num=0) sketches are supported, found num={sketch['num']}"
                raise ValueError(msg)
            mins = np.array(sketch["mins"], dtype=np.uint64)
            _check_sketch_md5(path, sketch, mins)
            if mins.size > 1 and not bool(np.all(mins[1:] > mins[:-1])):
                mins = np.unique(mins)
            return mins, sketch
    msg = f"{path} has no DNA sketch with ksize={ksize} max_hash={max_hash}"
    raise ValueError(msg)
