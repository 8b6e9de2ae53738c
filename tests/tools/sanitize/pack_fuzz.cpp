// The AVX2 and the scalar form of the FASTA packer on random texts in exact-size heap buffers, under
// AddressSanitizer / UBSan (host build only): same words, counts, record tables and lists of the residues that are neither
// ACGT nor N.   usage: pack_fuzz [trials]
#include <cstdarg>
#include <cstdio>
void pa_set_error(const char *fmt, ...) {}
#include "../../../pyani_plus_amd/csrc/pack_host.cpp"
#include <random>
int main(int argc, char **argv) {
  const int trials = argc > 1 ? atoi(argv[1]) : 30000;
  std::mt19937_64 rng(5);
  const char alpha[] = "ACGTacgtNn>\n\r \t-XRYkm";
  size_t checked = 0;
  for (int t = 0; t < trials; ++t) {
    const size_t n = rng() % 600;
    uint8_t *text = new uint8_t[n ? n : 1];
    const int style = t % 3;
    for (size_t i = 0; i < n; ++i) {
      if (style == 0) text[i] = (uint8_t)alpha[rng() % (sizeof(alpha) - 1)];
      else if (style == 1) text[i] = (uint8_t)("ACGT"[rng() % 4]);
      else text[i] = (rng() % 50 == 0) ? (uint8_t)alpha[rng() % (sizeof(alpha) - 1)] : (uint8_t)("ACGT"[rng() % 4]);
    }
    if (n > 3 && style) { text[0] = '>'; text[1] = 'x'; text[2] = '\n'; }
    if (style == 1 && n > 100) for (size_t i = 80; i < n; i += 81) text[i] = '\n';
    const uint64_t cap = pa_pack_bound(n);
    uint32_t *p1 = new uint32_t[cap / 16], *m1 = new uint32_t[cap / 32], *p2 = new uint32_t[cap / 16], *m2 = new uint32_t[cap / 32];
    uint64_t a[4], b[4];
    std::vector<uint64_t> rs1, rl1, rs2, rl2, ap1, ap2;
    std::vector<uint8_t> ab1, ab2;
    const int s1 = pack_fasta_impl<true>(text, n, p1, m1, cap, &a[0], &a[1], &a[2], &a[3], &rs1, &rl1, &ap1, &ab1);
    const int s2 = pack_fasta_impl<false>(text, n, p2, m2, cap, &b[0], &b[1], &b[2], &b[3], &rs2, &rl2, &ap2, &ab2);
    bool listed_ok = ap1 == ap2 && ab1 == ab2;
    for (size_t i = 0; i < ap1.size() && listed_ok; ++i)  // every listed position carries its mask bit, and no listed byte is an N
      listed_ok = ((m1[ap1[i] >> 5] >> (ap1[i] & 31)) & 1u) && ab1[i] != 'N' && ab1[i] != 'n' && (i == 0 || ap1[i] > ap1[i - 1]);
    if (s1 != s2 || memcmp(a, b, sizeof(a)) || memcmp(p1, p2, a[0] / 4) || memcmp(m1, m2, a[0] / 8) || rs1 != rs2 || rl1 != rl2 || !listed_ok) {
      printf("MISMATCH at trial %d (n=%zu)\n", t, n);
      return 1;
    }
    ++checked;
    delete[] text; delete[] p1; delete[] m1; delete[] p2; delete[] m2;
  }
  printf("vector and scalar packers agree on %zu texts\n", checked);
}
