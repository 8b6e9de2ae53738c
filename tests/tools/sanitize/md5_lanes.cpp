// Sixteen-lane md5 (md5_mb.h) against the one-message class (md5.h) on assorted lengths, then their speeds.
#include "../../../pyani_plus_amd/csrc/md5_mb.h"
#include "../../../pyani_plus_amd/csrc/md5.h"
#include <chrono>
#include <random>
int main() {
  std::mt19937_64 rng(7);
  printf("avx512 %d\n", (int)md5mb::have_avx512());
  // correctness: assorted lengths against the existing scalar class
  std::vector<size_t> lens = {0,1,55,56,57,63,64,65,119,120,127,128,129,1000,4096,4097,65536,100003,5,777,64*31,64*31+1,
                              300000,299999,12345,64,64,64,128,1<<20,(1<<20)+13, 999, 31, 2, 3, 4, 6,7,8,9,10};
  for (int rep = 0; rep < 3; ++rep) {
    uint32_t n = rep == 0 ? (uint32_t)lens.size() : (rep == 1 ? 16 : 3);
    std::vector<std::vector<uint8_t>> bufs(n);
    std::vector<const uint8_t*> ptrs(n); std::vector<size_t> ls(n);
    for (uint32_t i = 0; i < n; ++i) { bufs[i].resize(lens[i] + 1); for (auto &b : bufs[i]) b = (uint8_t)rng(); ptrs[i] = bufs[i].data(); ls[i] = lens[i]; }
    std::vector<char[33]> out(n);
    md5mb::md5_many(ptrs.data(), ls.data(), n, out.data());
    for (uint32_t i = 0; i < n; ++i) {
      Md5 m; m.update(ptrs[i], ls[i]); char ref[33]; m.hex(ref);
      char sc[33]; md5mb::md5_hex_scalar(ptrs[i], ls[i], sc);
      if (strcmp(ref, out[i]) || strcmp(ref, sc)) { printf("MISMATCH rep %d i %u len %zu: %s %s %s\n", rep, i, ls[i], ref, out[i], sc); return 1; }
    }
  }
  printf("digests equal\n");
  // speed
  const size_t L = 5000000; const uint32_t n = 32;
  std::vector<std::vector<uint8_t>> bufs(n, std::vector<uint8_t>(L));
  for (auto &b : bufs) for (size_t i = 0; i < L; i += 8) *(uint64_t*)&b[i] = rng();
  std::vector<const uint8_t*> ptrs(n); std::vector<size_t> ls(n, L);
  for (uint32_t i = 0; i < n; ++i) ptrs[i] = bufs[i].data();
  std::vector<char[33]> out(n);
  auto t0 = std::chrono::steady_clock::now();
  md5mb::md5_many(ptrs.data(), ls.data(), n, out.data());
  auto t1 = std::chrono::steady_clock::now();
  for (uint32_t i = 0; i < n; ++i) md5mb::md5_hex_scalar(ptrs[i], L, out[i]);
  auto t2 = std::chrono::steady_clock::now();
  for (uint32_t i = 0; i < n; ++i) { Md5 m; m.update(ptrs[i], L); m.hex(out[i]); }
  auto t3 = std::chrono::steady_clock::now();
  auto gbs = [&](auto a, auto b) { return n * L / std::chrono::duration<double>(b - a).count() / 1e9; };
  printf("x16 %.2f GB/s, unrolled scalar %.2f GB/s, old class %.2f GB/s\n", gbs(t0,t1), gbs(t1,t2), gbs(t2,t3));
}
