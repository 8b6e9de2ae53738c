"""Shared helpers for the test-suite (reading the golden fixtures)."""

from __future__ import annotations

import csv
import gzip
import hashlib
import json
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"

# fixture set -> (scaled, {md5: fasta file name})
FIXTURE_SETS = {
    "viral_example": (
        300,
        {
            "689d3fd6881db36b5e08329cf23cecdd": "MGV-GENOME-0264574.fas",
            "78975d5144a1cd12e98898d573cf6536": "MGV-GENOME-0266457.fna",
            "5584c7029328dc48d33f95f0a78f7e57": "OP073605.fasta",
        },
    ),
    "bad_alignments": (
        300,
        {
            "689d3fd6881db36b5e08329cf23cecdd": "MGV-GENOME-0264574.fas",
            "a30481565b45f6bbc6ce5260503067e0": "MGV-GENOME-0357962.fna",
        },
    ),
    "bacterial_example": (
        1000,
        {
            "f19cb07198a41a4406a22b2f57a6b5e7": "NC_002696.fasta.gz",
            "073194224aa8c13bebc1d14a3e74a3e7": "NC_010338.fna.gz",
            "9d72a8fb513cf9cc8cc6605a0ad4e837": "NC_011916.fas.gz",
            "9a9e23bfc5a184b8149e07e267d133b0": "NC_014100.fna.gz",
        },
    ),
}


def read_fasta_bytes(path: Path) -> bytes:
    """Decompressed file content (what the reference md5s, utils.py:178-196)."""
    raw = Path(path).read_bytes()
    return gzip.decompress(raw) if raw[:2] == b"\x1f\x8b" else raw


def md5_hex(data: bytes) -> str:
    return hashlib.md5(data).hexdigest()  # noqa: S324


def load_sig(path: Path) -> dict:
    """The single signature object of a sourmash `.sig` JSON fixture."""
    obj = json.loads(Path(path).read_text())
    assert isinstance(obj, list) and len(obj) == 1
    return obj[0]


def sig_mins(path: Path) -> np.ndarray:
    return np.array(load_sig(path)["signatures"][0]["mins"], dtype=np.uint64)


def load_manysearch(path: Path) -> list[dict]:
    with Path(path).open() as handle:
        return list(csv.DictReader(handle))


def load_matrix_tsv(path: Path) -> tuple[list[str], np.ndarray]:
    """Reference export-run matrix (rows=query, cols=subject); blank -> NaN."""
    with Path(path).open() as handle:
        rows = [line.rstrip("\n").split("\t") for line in handle]
    labels = rows[0][1:]
    mat = np.array([[float(v) if v not in ("", "nan", "NaN") else np.nan for v in r[1:]] for r in rows[1:]])
    return labels, mat
