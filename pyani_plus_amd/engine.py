"""Host driver of the MI355X sketch -> pair-count -> ANI pipeline.

Everything numeric happens inside ``libpyani_hip.so`` (``include/pyani_hip.h``);
this module only owns buffers.  PyTorch is used for device memory, the HIP
stream and (in ``distributed.py``) RCCL -- never for arithmetic on the path.

Reference steps replaced:
  A2  ``sourmash scripts singlesketch``  (pyani_plus/methods/sourmash.py:67-83)
  A5  ``sourmash scripts manysearch``    (pyani_plus/methods/sourmash.py:184-200)
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _capi
from ._capi import HipBackendError, check


def max_hash_for_scaled(scaled: int) -> int:
    """sourmash's ``max_hash`` for a ``scaled`` value (double-rounded 2**64/scaled)."""
    return int(_capi.load_library().pa_max_hash(int(scaled)))


# --------------------------------------------------------------------------- host arena
@dataclass
class HostArena:
    """2-bit packed genomes + invalid-position mask (layout: DESIGN.md, "Data layout")."""

    packed: np.ndarray  # uint32, 16 bases / word
    mask: np.ndarray  # uint32, 32 positions / word, bit=1 -> not a usable base
    genome_start: np.ndarray  # uint64 [n+1], multiples of 64
    residues: list[int] = field(default_factory=list)  # Genome.length per genome
    records: list[int] = field(default_factory=list)
    invalid: list[int] = field(default_factory=list)
    # FASTA records (contigs) in arena order: first arena position, residues, owning genome
    contig_start: np.ndarray | None = None
    contig_len: np.ndarray | None = None
    contig_genome: np.ndarray | None = None
    pinned_packed: "object" = None  # torch tensor owning ``packed`` when the loader wrote it into page-locked memory
    # residues that are neither ACGT nor N (IUPAC codes, ...): ascending arena positions and upper-cased bytes.  The mask
    # bit stands for N; the fragment-ANI path hashes these as the characters they are, as fastANI does
    ambig_pos: np.ndarray | None = None  # uint64
    ambig_byte: np.ndarray | None = None  # uint8

    @property
    def n_genomes(self) -> int:
        return len(self.genome_start) - 1

    @property
    def arena_bases(self) -> int:
        return int(self.genome_start[-1])


def _text_ambiguous(lib, text: bytes, fasta: bool, at_most: int | None = None) -> tuple[np.ndarray, np.ndarray]:
    """(positions relative to the genome's first, upper-cased bytes) of the residues that are neither ACGT nor N.

    ``at_most``: a bound on their number the caller already has (the packer's count of invalid residues, N included):
    the list then comes out of ONE pass over the text instead of a counting pass and a filling pass."""
    if not len(text) or at_most == 0:
        return np.zeros(0, np.uint64), np.zeros(0, np.uint8)
    if at_most is None:
        at_most = int(lib.pa_text_ambiguous(text, len(text), int(fasta), None, None, 0))
        if at_most < 0:
            raise HipBackendError(f"pa_text_ambiguous failed: {_capi.last_error()}")
        if at_most == 0:
            return np.zeros(0, np.uint64), np.zeros(0, np.uint8)
    pos, byte = np.zeros(at_most, dtype=np.uint64), np.zeros(at_most, dtype=np.uint8)
    n = int(lib.pa_text_ambiguous(text, len(text), int(fasta), pos.ctypes.data, byte.ctypes.data, at_most))
    if n < 0 or n > at_most:
        raise HipBackendError(f"pa_text_ambiguous failed: {_capi.last_error() if n < 0 else f'{n} residues against a bound of {at_most}'}")
    return pos[:n].copy(), byte[:n].copy()


def pack_genomes(texts: list[bytes], *, fasta: bool = True) -> HostArena:
    """Pack decompressed FASTA texts (or bare residue strings) into one arena (host only)."""
    lib = _capi.load_library()
    bounds = [int(lib.pa_pack_bound(len(t))) for t in texts]
    total = sum(bounds)
    packed = np.zeros(total // 16, dtype=np.uint32)
    mask = np.zeros(total // 32, dtype=np.uint32)
    starts = np.zeros(len(texts) + 1, dtype=np.uint64)
    residues, records, invalid = [], [], []
    c_start, c_len, c_genome = [], [], []
    a_pos, a_byte = [], []
    pos = 0
    for g, text in enumerate(texts):
        nb, nr, nrec, ninv = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        buf = bytes(text) if len(text) else None
        p_ptr = packed.ctypes.data + (pos // 16) * 4
        m_ptr = mask.ctypes.data + (pos // 32) * 4
        if fasta:
            st = lib.pa_pack_fasta(buf, len(text), p_ptr, m_ptr, bounds[g], C.byref(nb), C.byref(nr), C.byref(nrec), C.byref(ninv))
        else:
            st = lib.pa_pack_seq(buf, len(text), p_ptr, m_ptr, bounds[g], C.byref(nb), C.byref(ninv))
            nr.value, nrec.value = len(text), 1
        check(st, "pa_pack_fasta" if fasta else "pa_pack_seq")
        if fasta and nrec.value:
            rs = np.zeros(int(nrec.value), dtype=np.uint64)
            rl = np.zeros(int(nrec.value), dtype=np.uint64)
            lib.pa_fasta_records(buf, len(text), rs.ctypes.data, rl.ctypes.data, int(nrec.value))
            c_start.extend((rs + np.uint64(pos)).tolist())
            c_len.extend(rl.tolist())
            c_genome.extend([g] * int(nrec.value))
        elif not fasta:
            c_start.append(pos)
            c_len.append(len(text))
            c_genome.append(g)
        if ninv.value:
            ap, ab = _text_ambiguous(lib, buf, fasta, at_most=int(ninv.value))  # (ninv counts the N too: a bound, one pass)
            a_pos.append(ap + np.uint64(pos))
            a_byte.append(ab)
        starts[g] = pos
        pos += int(nb.value)
        residues.append(int(nr.value))
        records.append(int(nrec.value))
        invalid.append(int(ninv.value))
    starts[len(texts)] = pos
    return HostArena(
        packed[: pos // 16].copy(), mask[: pos // 32].copy(), starts, residues, records, invalid,
        np.array(c_start, dtype=np.uint64), np.array(c_len, dtype=np.uint32), np.array(c_genome, dtype=np.uint32),
        ambig_pos=np.concatenate(a_pos) if a_pos else np.zeros(0, np.uint64), ambig_byte=np.concatenate(a_byte) if a_byte else np.zeros(0, np.uint8),
    )


@dataclass
class LoadedFasta:
    """Per-file result of the threaded host front-end (``pa_fasta_batch_load``)."""

    path: str
    status: int
    message: str
    md5: str
    length: int  # sum of residues = Genome.length
    records: int
    invalid: int
    description: str
    gzip: bool


def load_fasta_files(paths, threads: int = 0, *, pinned: bool = False) -> tuple[list[LoadedFasta], HostArena]:
    """Read, gunzip, md5, parse and pack FASTA files on host threads.

    Returns per-file metadata (failed files carry ``status != 0`` and the reference's error
    text in ``message``) and the arena of the files that loaded, in order.  ``pinned``: the packed bases are
    written into page-locked memory (a torch tensor kept in ``arena.pinned_packed``), ready for
    ``HipEngine.sketch_streamed`` without a staging copy."""
    import os

    lib = _capi.load_library()
    paths = [str(p) for p in paths]
    n = len(paths)
    if threads <= 0:
        threads = int(lib.pa_host_cpu_budget())  # CPUs of the affinity mask, capped by the cgroup quota
    arr = (C.c_char_p * max(n, 1))(*[p.encode() for p in paths])
    batch = C.c_void_p()
    check(lib.pa_fasta_batch_load(arr, n, threads, C.byref(batch)), "pa_fasta_batch_load")
    try:
        infos: list[LoadedFasta] = []
        ok_residues, ok_records, ok_invalid = [], [], []
        rec_tables = []
        for i, path in enumerate(paths):
            md5 = C.create_string_buffer(33)
            nres, nrec, ninv, nb, nt = (C.c_uint64(0) for _ in range(5))
            desc, msg = C.c_char_p(), C.c_char_p()
            gz = C.c_int(0)
            st = lib.pa_fasta_batch_info(batch, i, md5, C.byref(nres), C.byref(nrec), C.byref(ninv), C.byref(nb), C.byref(nt), C.byref(desc), C.byref(msg), C.byref(gz))
            infos.append(
                LoadedFasta(path, st, (msg.value or b"").decode(errors="replace"), md5.value.decode(), int(nres.value), int(nrec.value),
                            int(ninv.value), (desc.value or b"").decode(errors="replace"), bool(gz.value))
            )  # fmt: skip
            if st == 0:
                ok_residues.append(int(nres.value))
                ok_records.append(int(nrec.value))
                ok_invalid.append(int(ninv.value))
                rs_p, rl_p, rn = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)(), C.c_uint64(0)
                lib.pa_fasta_batch_records(batch, i, C.byref(rs_p), C.byref(rl_p), C.byref(rn))
                k_rec = int(rn.value)
                rec_tables.append((np.ctypeslib.as_array(rs_p, (k_rec,)).copy(), np.ctypeslib.as_array(rl_p, (k_rec,)).copy()) if k_rec else (np.zeros(0, np.uint64), np.zeros(0, np.uint64)))
        total = int(lib.pa_fasta_batch_arena_bases(batch))
        pinned_packed = None
        if pinned:
            import torch

            pinned_packed = torch.empty(max(total // 16, 1), dtype=torch.int32, pin_memory=torch.cuda.is_available())
            packed = pinned_packed.numpy().view(np.uint32)
        else:
            packed = np.empty(max(total // 16, 1), dtype=np.uint32)
        mask = np.empty(max(total // 32, 1), dtype=np.uint32)
        starts_all = np.zeros(n + 1, dtype=np.uint64)
        check(lib.pa_fasta_batch_copy_arena(batch, packed.ctypes.data, mask.ctypes.data, starts_all.ctypes.data), "pa_fasta_batch_copy_arena")
        keep = [i for i, info in enumerate(infos) if info.status == 0]
        starts = np.array([starts_all[i] for i in keep] + [total], dtype=np.uint64)
        c_start = np.concatenate([rs + starts[g] for g, (rs, _rl) in enumerate(rec_tables)]) if rec_tables else np.zeros(0, np.uint64)
        c_len = np.concatenate([rl for _rs, rl in rec_tables]).astype(np.uint32) if rec_tables else np.zeros(0, np.uint32)
        c_genome = np.concatenate([np.full(len(rs), g, dtype=np.uint32) for g, (rs, _rl) in enumerate(rec_tables)]) if rec_tables else np.zeros(0, np.uint32)
        n_amb = int(lib.pa_fasta_batch_ambiguous(batch, None, None, 0))
        a_pos, a_byte = np.zeros(max(n_amb, 0), dtype=np.uint64), np.zeros(max(n_amb, 0), dtype=np.uint8)
        if n_amb > 0:
            lib.pa_fasta_batch_ambiguous(batch, a_pos.ctypes.data, a_byte.ctypes.data, n_amb)
        return infos, HostArena(
            packed[: total // 16], mask[: total // 32], starts, ok_residues, ok_records, ok_invalid,
            c_start.astype(np.uint64), c_len, c_genome, pinned_packed[: max(total // 16, 1)] if pinned_packed is not None else None,
            ambig_pos=a_pos, ambig_byte=a_byte,
        )
    finally:
        lib.pa_fasta_batch_free(batch)


# --------------------------------------------------------------------------- device objects
@dataclass
class DeviceArena:
    packed: "object"  # torch.int32 tensor
    mask: "object"
    genome_start: np.ndarray  # host uint64 [n+1]
    dirty: "object" = None  # torch.int64 tensor: one bit per 64-position block that needs its mask words (built on first use)
    # the host arena's list of residues that are neither ACGT nor N (``HostArena.ambig_pos`` / ``ambig_byte``), handed to the
    # library before a fragment-ANI call on this arena
    ambig_pos: np.ndarray | None = None
    ambig_byte: np.ndarray | None = None

    @property
    def n_genomes(self) -> int:
        return len(self.genome_start) - 1

    @property
    def arena_bases(self) -> int:
        return int(self.genome_start[-1])


@dataclass
class PinnedArena:
    """A host arena ready for ``HipEngine.sketch_streamed``: packed bases in page-locked memory and the
    invalid-position mask as runs (start, length) instead of a bitmap."""

    packed: "object"  # pinned torch.int32 tensor
    run_start: np.ndarray  # uint64
    run_len: np.ndarray  # uint64
    genome_start: np.ndarray  # host uint64 [n+1]

    @property
    def n_genomes(self) -> int:
        return len(self.genome_start) - 1

    @property
    def arena_bases(self) -> int:
        return int(self.genome_start[-1])


def mask_runs(mask: np.ndarray, arena_bases: int) -> tuple[np.ndarray, np.ndarray]:
    """Runs of invalid positions of a mask bitmap (host helper ``pa_mask_runs``)."""
    lib = _capi.load_library()
    mask = np.ascontiguousarray(mask, dtype=np.uint32)
    n = int(lib.pa_mask_runs(mask.ctypes.data, arena_bases, None, None, 0))
    if n < 0:
        raise HipBackendError("pa_mask_runs: null mask")
    start = np.empty(max(n, 1), dtype=np.uint64)
    length = np.empty(max(n, 1), dtype=np.uint64)
    got = int(lib.pa_mask_runs(mask.ctypes.data, arena_bases, start.ctypes.data, length.ctypes.data, n))
    if got != n:
        raise HipBackendError("pa_mask_runs: run count changed between calls")
    return start[:n], length[:n]


@dataclass
class DeviceSketches:
    """CSR of ascending duplicate-free u64 hashes, resident in HBM."""

    hashes: "object"  # torch.int64 tensor (bit pattern of uint64), length >= total
    off: "object"  # torch.int64 tensor [n+1]
    n: int
    total: int
    host_off: np.ndarray | None = None  # uint64 [n+1] copy of `off` when the host has it (pair phase needs no round trip then)

    def offsets_host(self) -> np.ndarray:
        if self.host_off is None:
            self.host_off = np.ascontiguousarray(self.off.cpu().numpy().astype(np.uint64))
        return self.host_off

    def sizes(self) -> np.ndarray:
        off = self.offsets_host()
        return (off[1:] - off[:-1]).astype(np.uint64)

    def to_host(self) -> list[np.ndarray]:
        off = self.offsets_host().astype(np.int64)
        flat = self.hashes[: self.total].cpu().numpy().view(np.uint64)
        return [flat[int(off[g]) : int(off[g + 1])].copy() for g in range(self.n)]


class HipEngine:
    """One HIP context on one GPU.  Raises ``HipBackendError`` when no MI355X is usable."""

    def __init__(self, device: int = 0, *, tools: bool = False):
        """``tools``: bind the -DPA_TOOLS build of the library (``_capi.TOOLS_LIB_PATH``), in which the environment
        switches of tools/ and of the rare-path tests exist; the product library ignores them."""
        self.lib = _capi.load_library(tools)
        self._check = lambda status, what: check(status, what, self.lib)
        try:
            import torch
        except ImportError as err:  # pragma: no cover
            raise HipBackendError("PyTorch (ROCm build) is required for device memory") from err
        if not torch.cuda.is_available():
            raise HipBackendError("no HIP device visible to PyTorch; the HIP path has no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        ctx = C.c_void_p()
        self._check(self.lib.pa_ctx_create(device, C.byref(ctx)), "pa_ctx_create")
        self.ctx = ctx
        self.use_torch_stream()

    # -- plumbing
    def use_torch_stream(self) -> None:
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.pa_ctx_set_stream(self.ctx, C.c_void_p(stream)), "pa_ctx_set_stream")

    def close(self) -> None:
        if getattr(self, "ctx", None):
            self.lib.pa_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cus, mem = C.c_int(0), C.c_uint64(0)
        self._check(self.lib.pa_ctx_device_info(self.ctx, name, C.byref(cus), C.byref(mem)), "pa_ctx_device_info")
        return {"name": name.value.decode(), "compute_units": cus.value, "global_mem": mem.value}

    def sync(self) -> None:
        self._check(self.lib.pa_ctx_sync(self.ctx), "pa_ctx_sync")

    # -- data movement
    def upload(self, arena: HostArena) -> DeviceArena:
        t = self.torch
        packed = t.from_numpy(arena.packed.view(np.int32)).to(self.device)
        mask = t.from_numpy(arena.mask.view(np.int32)).to(self.device)
        return DeviceArena(packed, mask, arena.genome_start.copy(), ambig_pos=arena.ambig_pos, ambig_byte=arena.ambig_byte)

    def _set_ambiguous(self, arena: DeviceArena) -> None:
        """Hand the arena's residues that are neither ACGT nor N to the fragment-ANI kernels (``pa_fragani_set_ambiguous``:
        a list the context holds already changes nothing there, a reusable index included)."""
        pos = getattr(arena, "ambig_pos", None)
        n = 0 if pos is None else len(pos)
        if n:
            p = np.ascontiguousarray(pos, dtype=np.uint64)
            b = np.ascontiguousarray(arena.ambig_byte, dtype=np.uint8)
            self._check(self.lib.pa_fragani_set_ambiguous(self.ctx, arena.packed.data_ptr(), p.ctypes.data, b.ctypes.data, n), "pa_fragani_set_ambiguous")
        else:
            self._check(self.lib.pa_fragani_set_ambiguous(self.ctx, None, None, None, 0), "pa_fragani_set_ambiguous")

    def sketches_from_host(self, sketches: list[np.ndarray]) -> DeviceSketches:
        t = self.torch
        off = np.zeros(len(sketches) + 1, dtype=np.int64)
        np.cumsum([len(s) for s in sketches], out=off[1:])
        flat = np.concatenate([np.asarray(s, dtype=np.uint64) for s in sketches]) if sketches else np.empty(0, np.uint64)
        if flat.size == 0:
            flat = np.zeros(1, dtype=np.uint64)
        return DeviceSketches(
            t.from_numpy(flat.view(np.int64).copy()).to(self.device), t.from_numpy(off).to(self.device), len(sketches), int(off[-1]),
            np.ascontiguousarray(off.astype(np.uint64)),
        )

    def sketches_from_gathered(self, hashes, off, off_host: np.ndarray) -> DeviceSketches:
        """The CSR an all-gather delivered (``distributed.allgather_sketches``: tensors on this device, or on the host
        when the collective ran there) as ``DeviceSketches``."""
        hashes, off = hashes.to(self.device), off.to(self.device)
        return DeviceSketches(hashes, off, len(off_host) - 1, int(off_host[-1]), np.ascontiguousarray(off_host, dtype=np.uint64))

    def arena_dirty(self, arena: DeviceArena):
        """The arena's dirty-block bitmap (``pa_arena_dirty``), built on first use and kept with the arena."""
        if arena.dirty is None:
            t = self.torch
            words = (arena.arena_bases // 64 + 63) // 64
            dirty = t.empty(max(words, 1), dtype=t.int64, device=self.device)
            self._check(self.lib.pa_arena_dirty(self.ctx, arena.mask.data_ptr(), arena.arena_bases, dirty.data_ptr()), "pa_arena_dirty")
            arena.dirty = dirty
        return arena.dirty

    # -- the three device steps
    def sketch(self, arena: DeviceArena, k: int, scaled: int, *, max_hash: int | None = None) -> DeviceSketches:
        t = self.torch
        mh = int(max_hash) if max_hash is not None else max_hash_for_scaled(scaled)
        n = arena.n_genomes
        frac = 1.0 if mh >= 2**64 - 1 else (mh + 1) / 2.0**64
        cap = int(arena.arena_bases * frac * 1.25) + 4096
        off = t.empty(n + 1, dtype=t.int64, device=self.device)
        gs = np.ascontiguousarray(arena.genome_start, dtype=np.uint64)
        total = C.c_uint64(0)
        dirty = self.arena_dirty(arena)
        for _attempt in range(2):
            hashes = t.empty(max(cap, 1), dtype=t.int64, device=self.device)
            st = self.lib.pa_sketch(
                self.ctx, arena.packed.data_ptr(), arena.mask.data_ptr(), dirty.data_ptr(), arena.arena_bases,
                gs.ctypes.data_as(C.POINTER(C.c_uint64)), n, k, mh, hashes.data_ptr(), cap, off.data_ptr(), C.byref(total),
            )  # fmt: skip
            if st == _capi.PA_E_CAPACITY:
                cap = int(total.value)
                continue
            self._check(st, "pa_sketch")
            return DeviceSketches(hashes, off, n, int(total.value))
        raise HipBackendError("pa_sketch: capacity retry failed")

    def pin_arena(self, arena: HostArena) -> PinnedArena:
        """Page-lock the packed bases and reduce the mask to runs (done once, outside any timed region)."""
        t = self.torch
        if arena.pinned_packed is not None:
            packed = arena.pinned_packed  # the loader wrote straight into page-locked memory
        else:
            packed = t.from_numpy(np.ascontiguousarray(arena.packed).view(np.int32)).pin_memory()
        start, length = mask_runs(arena.mask, int(arena.genome_start[-1]))
        return PinnedArena(packed, start, length, np.ascontiguousarray(arena.genome_start, dtype=np.uint64).copy())

    def sketch_streamed(self, host: PinnedArena, k: int, scaled: int, *, max_hash: int | None = None, arena: DeviceArena | None = None):
        """Host arena -> (device arena, sketches); the upload runs behind the hash kernel (``pa_sketch_streamed``)."""
        t = self.torch
        mh = int(max_hash) if max_hash is not None else max_hash_for_scaled(scaled)
        n, bases = host.n_genomes, host.arena_bases
        if arena is None:
            arena = DeviceArena(
                t.empty(max(bases // 16, 1), dtype=t.int32, device=self.device),
                t.empty(max(bases // 32, 1), dtype=t.int32, device=self.device),
                host.genome_start.copy(),
            )
        if arena.dirty is None:
            arena.dirty = t.empty(max((bases // 64 + 63) // 64, 1), dtype=t.int64, device=self.device)
        frac = 1.0 if mh >= 2**64 - 1 else (mh + 1) / 2.0**64
        cap = int(bases * frac * 1.25) + 4096
        off = t.empty(n + 1, dtype=t.int64, device=self.device)
        gs = np.ascontiguousarray(host.genome_start, dtype=np.uint64)
        total = C.c_uint64(0)
        for _attempt in range(2):
            hashes = t.empty(max(cap, 1), dtype=t.int64, device=self.device)
            st = self.lib.pa_sketch_streamed(
                self.ctx, host.packed.data_ptr(), host.run_start.ctypes.data, host.run_len.ctypes.data, len(host.run_start),
                bases, gs.ctypes.data_as(C.POINTER(C.c_uint64)), n, k, mh, arena.packed.data_ptr(), arena.mask.data_ptr(),
                arena.dirty.data_ptr(), hashes.data_ptr(), cap, off.data_ptr(), C.byref(total),
            )  # fmt: skip
            if st == _capi.PA_E_CAPACITY:
                cap = int(total.value)
                continue
            self._check(st, "pa_sketch_streamed")
            return arena, DeviceSketches(hashes, off, n, int(total.value))
        raise HipBackendError("pa_sketch_streamed: capacity retry failed")

    def pair_counts(self, sk: DeviceSketches, q_range=None, s_range=None, algo: int = _capi.PA_PAIRS_AUTO):
        """uint32 |S_q n S_s| for q in q_range, s in s_range -> torch.int32 [nq, ns] on the GPU."""
        t = self.torch
        q0, q1 = q_range or (0, sk.n)
        s0, s1 = s_range or (0, sk.n)
        counts = t.empty((q1 - q0, s1 - s0), dtype=t.int32, device=self.device)
        h_off = sk.offsets_host()  # one small copy per sketch set, cached; the pair phase itself then never waits for the host
        assert h_off.dtype == np.uint64 and len(h_off) == sk.n + 1
        self._check(
            self.lib.pa_pair_counts_ex(
                self.ctx, sk.hashes.data_ptr(), sk.off.data_ptr(), h_off.ctypes.data, sk.n, q0, q1, s0, s1, counts.data_ptr(), algo,
            ),  # fmt: skip
            "pa_pair_counts",
        )
        return counts

    def pair_dict_prepare(self, subject_hashes, n_postings: int) -> None:
        """Enqueue the dictionary build of one subject tile from its ``n_postings`` contiguous hashes (multi-GPU
        overlap with the sketch all-gather); the next default-algorithm ``pair_counts`` over that tile uses it."""
        self._check(self.lib.pa_pair_dict_prepare(self.ctx, subject_hashes.data_ptr(), int(n_postings)), "pa_pair_dict_prepare")

    def ani(self, counts, sk: DeviceSketches, k: int, q_range=None, s_range=None):
        """Device f64 (identity, cov_query); NaN marks the reference's NULL."""
        t = self.torch
        q0, q1 = q_range or (0, sk.n)
        s0, s1 = s_range or (0, sk.n)
        ident = t.empty((q1 - q0, s1 - s0), dtype=t.float64, device=self.device)
        cov = t.empty_like(ident)
        self._check(
            self.lib.pa_ani(self.ctx, counts.data_ptr(), sk.off.data_ptr(), q0, q1, s0, s1, k, ident.data_ptr(), cov.data_ptr()),
            "pa_ani",
        )
        return ident, cov

    # -- bottom-m MinHash + Mash Jaccard (named by BASELINE configs[1]; not a reference code path)
    def sketch_bottom(self, arena: DeviceArena, k: int, m: int) -> DeviceSketches:
        t = self.torch
        n = arena.n_genomes
        hashes = t.empty(max(n * m, 1), dtype=t.int64, device=self.device)
        off = t.empty(n + 1, dtype=t.int64, device=self.device)
        gs = np.ascontiguousarray(arena.genome_start, dtype=np.uint64)
        total = C.c_uint64(0)
        self._check(
            self.lib.pa_sketch_bottom(
                self.ctx, arena.packed.data_ptr(), arena.mask.data_ptr(), self.arena_dirty(arena).data_ptr(), arena.arena_bases,
                gs.ctypes.data_as(C.POINTER(C.c_uint64)), n, k, m, hashes.data_ptr(), n * m, off.data_ptr(), C.byref(total),
            ),  # fmt: skip
            "pa_sketch_bottom",
        )
        return DeviceSketches(hashes, off, n, int(total.value))

    def pair_mash(self, sk: DeviceSketches, m: int, q_range=None, s_range=None):
        """(common, denom) int32 tensors [nq, ns] of the Mash Jaccard estimator."""
        t = self.torch
        q0, q1 = q_range or (0, sk.n)
        s0, s1 = s_range or (0, sk.n)
        common = t.empty((q1 - q0, s1 - s0), dtype=t.int32, device=self.device)
        denom = t.empty_like(common)
        self._check(
            self.lib.pa_pair_mash(self.ctx, sk.hashes.data_ptr(), sk.off.data_ptr(), sk.n, q0, q1, s0, s1, m, common.data_ptr(), denom.data_ptr()),
            "pa_pair_mash",
        )
        return common, denom

    def ani_mash(self, common, denom, k: int):
        t = self.torch
        out = t.empty(common.shape, dtype=t.float64, device=self.device)
        self._check(self.lib.pa_ani_mash(self.ctx, common.data_ptr(), denom.data_ptr(), common.numel(), k, out.data_ptr()), "pa_ani_mash")
        return out

    # -- fastANI-style fragment ANI (BASELINE configs[3])
    def fragani(self, arena: DeviceArena, contig_start, contig_len, contig_genome, k: int = 16, frag_len: int = 3000, ref_range=None,
                query_range=None, reuse_index: bool = False, out=None, columns_only: bool = False):
        """All ordered genome pairs: (total_frags[n], matched[n, n], ident_sum[n, n]) as numpy arrays;
        ANI(q, r) = ident_sum / matched (percent), rows = query.  ``ref_range`` = (r0, r1) maps the queries against
        those reference genomes only (the other columns stay 0); ``query_range`` = (q0, q1) maps those query genomes
        only and leaves the other rows of ``out`` = (total, matched, ident_sum) as they are; ``reuse_index``: the
        previous call was on this very arena with the same k and fragment length, take over its reference index;
        ``columns_only``: the two matrices are [n, r1 - r0] -- the columns of the reference range and nothing else."""
        n = arena.n_genomes
        r0, r1 = ref_range if ref_range is not None else (0, n)
        q0, q1 = query_range if query_range is not None else (0, n)
        cols = int(r1) - int(r0) if columns_only else n
        cs = np.ascontiguousarray(contig_start, dtype=np.uint64)
        cl = np.ascontiguousarray(contig_len, dtype=np.uint32)
        cg = np.ascontiguousarray(contig_genome, dtype=np.uint32)
        if out is None:
            total = np.zeros(n, dtype=np.uint32)
            matched = np.zeros((n, cols), dtype=np.uint32)
            ident_sum = np.zeros((n, cols), dtype=np.float64)
        else:
            total, matched, ident_sum = out
            assert total.shape == (n,) and matched.shape == (n, cols) == ident_sum.shape
            assert total.dtype == np.uint32 and matched.dtype == np.uint32 and ident_sum.dtype == np.float64
            assert matched.flags.c_contiguous and ident_sum.flags.c_contiguous
        flags = (_capi.PA_FRAGANI_REUSE_INDEX if reuse_index else 0) | (_capi.PA_FRAGANI_COLUMNS_ONLY if columns_only else 0)
        self._set_ambiguous(arena)
        self._check(
            self.lib.pa_fragani_ex(
                self.ctx, arena.packed.data_ptr(), arena.mask.data_ptr(), arena.arena_bases, cs.ctypes.data, cl.ctypes.data,
                cg.ctypes.data, len(cs), n, k, frag_len, int(q0), int(q1), int(r0), int(r1),
                flags, total.ctypes.data, matched.ctypes.data, ident_sum.ctypes.data,
            ),  # fmt: skip
            "pa_fragani",
        )
        return total, matched, ident_sum

    def fragani_sketch(self, arena: DeviceArena, contig_start, contig_len, contig_genome, k: int, window: int):
        """Stage 1 alone: winnowed minimizers (hash, window id, contig) of every contig."""
        cs = np.ascontiguousarray(contig_start, dtype=np.uint64)
        cl = np.ascontiguousarray(contig_len, dtype=np.uint32)
        cg = np.ascontiguousarray(contig_genome, dtype=np.uint32)
        cap = max(1024, arena.arena_bases)  # at most one minimizer per position
        h = np.zeros(cap, dtype=np.uint32)
        wp = np.zeros(cap, dtype=np.uint32)
        ct = np.zeros(cap, dtype=np.uint32)
        n = C.c_uint64(0)
        self._set_ambiguous(arena)
        self._check(
            self.lib.pa_fragani_sketch(
                self.ctx, arena.packed.data_ptr(), arena.mask.data_ptr(), arena.arena_bases, cs.ctypes.data, cl.ctypes.data,
                cg.ctypes.data, len(cs), arena.n_genomes, k, window, h.ctypes.data, wp.ctypes.data, ct.ctypes.data, cap, C.byref(n),
            ),  # fmt: skip
            "pa_fragani_sketch",
        )
        m = int(n.value)
        return h[:m], wp[:m], ct[:m]

    def fragani_workspace(self) -> dict:
        """Device bytes of the context's fragment-ANI workspace: held now, the most ever held, the cap (0: none)."""
        held, peak, cap = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.pa_fragani_workspace(self.ctx, C.byref(held), C.byref(peak), C.byref(cap)), "pa_fragani_workspace")
        return {"held_bytes": int(held.value), "peak_bytes": int(peak.value), "cap_bytes": int(cap.value)}

    def fragani_set_workspace_cap(self, cap_bytes: int) -> None:
        """Later fragment-ANI calls that would take the workspace past ``cap_bytes`` (0: no cap) fail with a message naming
        the call and the sizes (``HipBackendError``), the workspace as it was."""
        self._check(self.lib.pa_fragani_set_workspace_cap(self.ctx, int(cap_bytes)), "pa_fragani_set_workspace_cap")

    # -- profiling
    def prof_enable(self, on: bool = True) -> None:
        self._check(self.lib.pa_prof_enable(self.ctx, int(on)), "pa_prof_enable")

    def prof_reset(self) -> None:
        self._check(self.lib.pa_prof_reset(self.ctx), "pa_prof_reset")

    def prof_get(self) -> dict[str, tuple[float, int]]:
        out = {}
        for name, idx in _capi.PROF_PHASES.items():
            ms, n = C.c_double(0), C.c_uint64(0)
            self._check(self.lib.pa_prof_get(self.ctx, idx, C.byref(ms), C.byref(n)), "pa_prof_get")
            out[name] = (ms.value, int(n.value))
        return out


def ani_host(counts: np.ndarray, q_sizes, s_sizes, k: int, *, symmetric: bool = False, threads: int = 0, out=None):
    """Strict (host libm ``pow``) containment-ANI transform used at the JSON/DB boundary.

    ``symmetric``: the block is square and rows and columns are the same genomes in the same order (one
    ``pow`` per ordered pair instead of two).  ``out`` = (identity, cov_query, is_null) arrays to fill."""
    lib = _capi.load_library()
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    nq, ns = counts.shape
    q_sizes = np.ascontiguousarray(q_sizes, dtype=np.uint64)
    s_sizes = np.ascontiguousarray(s_sizes, dtype=np.uint64)
    if out is None:
        ident = np.empty((nq, ns), dtype=np.float64)
        cov = np.empty((nq, ns), dtype=np.float64)
        null = np.empty((nq, ns), dtype=np.uint8)
    else:
        ident, cov, null = out
        assert ident.shape == cov.shape == null.shape == (nq, ns) and ident.flags.c_contiguous and cov.flags.c_contiguous
    check(
        lib.pa_ani_host(
            counts.ctypes.data, q_sizes.ctypes.data, s_sizes.ctypes.data, nq, ns, k, ident.ctypes.data, cov.ctypes.data, null.ctypes.data,
            int(bool(symmetric)), int(threads),
        ),  # fmt: skip
        "pa_ani_host",
    )
    return ident, cov, null.view(np.bool_) if null.dtype == np.uint8 else null
