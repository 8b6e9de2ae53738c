// Damaged gzip streams through inflate_fast.h under AddressSanitizer / UBSan (host build only): no memory error, and
// nothing but the original bytes is ever accepted.   usage: inflate_fuzz <file.gz> [seed] [trials]
#include "../../../pyani_plus_amd/csrc/inflate_fast.h"
#include <cstdio>
#include <random>
#include <zlib.h>
int main(int argc, char **argv) {
  std::vector<uint8_t> in(50 << 20);
  FILE *f = fopen(argv[1], "rb"); in.resize(fread(in.data(), 1, in.size(), f)); fclose(f);
  std::vector<uint8_t> ref;
  if (!pa_inflate::gunzip_all(in.data(), in.size(), ref)) { printf("base failed\n"); return 1; }
  std::mt19937_64 rng(argc > 2 ? atoi(argv[2]) : 1);
  size_t ok = 0, bad = 0, same = 0;
  const int trials = argc > 3 ? atoi(argv[3]) : 20000;
  for (int t = 0; t < trials; ++t) {
    std::vector<uint8_t> m(in);
    const int kind = t % 4;
    if (kind == 0) m[rng() % m.size()] ^= 1u << (rng() % 8);
    else if (kind == 1) m.resize(1 + rng() % m.size());
    else if (kind == 2) { size_t i = rng() % (m.size() - 16); for (int j = 0; j < 16; ++j) m[i + j] = (uint8_t)rng(); }
    else { size_t i = 10 + rng() % 200; if (i < m.size()) m[i] = (uint8_t)rng(); }  // headers of the first blocks
    // exact-size heap copy so that any overread trips the sanitizer
    uint8_t *heap = new uint8_t[m.size()];
    memcpy(heap, m.data(), m.size());
    std::vector<uint8_t> out;
    const bool r = pa_inflate::gunzip_all(heap, m.size(), out);
    delete[] heap;
    if (r) { ++ok; if (out == ref) ++same; else { printf("ACCEPTED WRONG DATA at trial %d\n", t); return 2; } } else ++bad;
  }
  printf("trials %d: accepted %zu (all identical to the original: %zu), rejected %zu\n", trials, ok, same, bad);
}
