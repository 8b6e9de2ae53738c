"""CPU oracle for the sourmash hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; nothing under ``pyani_plus_amd/`` does.
See ``sourmash_oracle.c`` for provenance (reference call sites and the fixtures
that pin it).
"""

from .pyoracle import (  # noqa: F401
    ani,
    build,
    fragani_identity,
    fragani_kmer_hash,
    fragani_map,
    fragani_minimizers,
    fragani_many,
    fragani_pair,
    fragani_get_option,
    fragani_set_option,
    fragani_tables,
    fragani_window_size,
    intersect,
    mash_ani,
    mash_pairs,
    max_hash,
    murmur3_h1,
    pair_counts,
    sketch_fasta_text,
    sketch_bottom_seq,
    sketch_many,
    sketch_seq,
)
