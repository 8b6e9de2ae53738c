#!/bin/bash
# Host-side native code under AddressSanitizer + UBSan, and the shared host pool under ThreadSanitizer (the GPU pool
# has no sanitizer runs; this is the CPU build).
#   bash tests/tools/sanitize/run.sh [trials]
set -eu
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../../.." && pwd)
OUT=${TMPDIR:-/tmp}/pa_sanitize.$$
mkdir -p "$OUT"
TRIALS=${1:-5000}
FLAGS="-O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined"
g++ $FLAGS -o "$OUT/inflate_fuzz" "$HERE/inflate_fuzz.cpp" -lz
g++ $FLAGS -I"$ROOT/include" -o "$OUT/pack_fuzz" "$HERE/pack_fuzz.cpp" -lpthread
g++ $FLAGS -o "$OUT/md5_lanes" "$HERE/md5_lanes.cpp"
g++ -O1 -g -std=c++17 -fsanitize=thread -o "$OUT/pool_race" "$HERE/pool_race.cpp" -lpthread
python3 - "$OUT" <<'PY'
import gzip, sys, numpy as np
out = sys.argv[1]
rng = np.random.default_rng(9)
seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 60000, dtype=np.uint8)].tobytes()
text = b">g\n" + b"\n".join(seq[i:i + 80] for i in range(0, len(seq), 80)) + b"\n"
open(out + "/dna.gz", "wb").write(gzip.compress(text, 6))
open(out + "/text.gz", "wb").write(gzip.compress(b"the quick brown fox jumps over the lazy dog. " * 2000 + bytes(range(256)) * 20, 9))
PY
"$OUT/inflate_fuzz" "$OUT/dna.gz" 1 "$TRIALS"
"$OUT/inflate_fuzz" "$OUT/text.gz" 2 "$TRIALS"
"$OUT/pack_fuzz" "$TRIALS"
"$OUT/md5_lanes" | head -2
"$OUT/pool_race" 1000
rm -rf "$OUT"
echo "sanitizer runs clean"
