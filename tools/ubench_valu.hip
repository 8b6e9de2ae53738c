// ubench_valu.hip -- VALU / LDS issue-rate microbenchmarks for gfx950 (MI355X).
// Measures cycles per wave64 instruction per SIMD for the integer ops the k-mer hash kernel
// is made of, at full occupancy (8 waves/SIMD), with independent dependency chains.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench tools/ubench_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;  // independent chains

#define KERNEL(name, DECL, BODY, SINK)                                              \
  __global__ __launch_bounds__(256) void name(uint32_t *out, uint32_t seed) {       \
    DECL;                                                                           \
    for (int it = 0; it < ITERS; ++it) {                                            \
      BODY;                                                                         \
    }                                                                               \
    SINK;                                                                           \
  }

#define DECL32 uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; uint32_t k = seed | 1
#define SINK32 out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7
#define REP8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)

#define OP_XOR(x) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_ADD(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_MULHI(x) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_MUL24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_MAD24(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x) : "v"(k));
#define OP_ALIGN(x) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x) : "v"(k));
#define OP_PERM(x) asm volatile("v_perm_b32 %0, %0, %1, %0" : "+v"(x) : "v"(k));
#define OP_ADD3(x) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define OP_LSHLOR(x) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x) : "v"(k));
#define OP_BFE(x) asm volatile("v_bfe_u32 %0, %0, 3, 20" : "+v"(x));
#define OP_CNDMASK(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(k));

KERNEL(k_xor, DECL32, REP8(OP_XOR), SINK32)
KERNEL(k_add, DECL32, REP8(OP_ADD), SINK32)
KERNEL(k_mullo, DECL32, REP8(OP_MULLO), SINK32)
KERNEL(k_mulhi, DECL32, REP8(OP_MULHI), SINK32)
KERNEL(k_mul24, DECL32, REP8(OP_MUL24), SINK32)
KERNEL(k_mad24, DECL32, REP8(OP_MAD24), SINK32)
KERNEL(k_align, DECL32, REP8(OP_ALIGN), SINK32)
KERNEL(k_perm, DECL32, REP8(OP_PERM), SINK32)
KERNEL(k_add3, DECL32, REP8(OP_ADD3), SINK32)
KERNEL(k_lshlor, DECL32, REP8(OP_LSHLOR), SINK32)
KERNEL(k_bfe, DECL32, REP8(OP_BFE), SINK32)
KERNEL(k_cndmask, DECL32, REP8(OP_CNDMASK), SINK32)


#define OP_SHL32(x) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x));
#define OP_SHR32(x) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(x));
#define OP_SHLV32(x) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(x) : "v"(k));
#define OP_AND(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_OR(x) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_SUB(x) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_MOV(x) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(k));
#define OP_LSHLADD32(x) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(k));
#define OP_ADDLSHL32(x) asm volatile("v_add_lshl_u32 %0, %0, %1, 3" : "+v"(x) : "v"(k));
#define OP_XAD(x) asm volatile("v_xad_u32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define OP_ANDOR(x) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define OP_OR3(x) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define OP_BFI(x) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(x) : "v"(k));
#define OP_CMPU32(x) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x), "v"(k) : "vcc");
#define OP_SDWA(x) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(x) : "v"(k));
#define OP_ADDSDWA(x) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(x) : "v"(k));
#define OP_MULLO_S(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "s"(k));
#define OP_XOR_S(x) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x) : "s"(k));
#define OP_XOR_LIT(x) asm volatile("v_xor_b32 %0, 0x12345678, %0" : "+v"(x));
#define OP_PKADD16(x) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_PKMUL16(x) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x) : "v"(k));
#define OP_PKMAD16(x) asm volatile("v_pk_mad_u16 %0, %0, %1, %1" : "+v"(x) : "v"(k));
#define OP_MADU16(x) asm volatile("v_mad_u32_u16 %0, %0, %1, %0" : "+v"(x) : "v"(k));
#define OP_DOT2(x) asm volatile("v_dot2_u32_u16 %0, %0, %1, %0" : "+v"(x) : "v"(k));
#define OP_DOT4(x) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(x) : "v"(k));
#define OP_ALIGNBYTE(x) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(x) : "v"(k));
#define OP_ADDCO1(x) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(x) : "v"(k) : "vcc");
#define OP_FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(k));
#define OP_MIN(x) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(k));

KERNEL(k_shl32, DECL32, REP8(OP_SHL32), SINK32)
KERNEL(k_shr32, DECL32, REP8(OP_SHR32), SINK32)
KERNEL(k_shlv32, DECL32, REP8(OP_SHLV32), SINK32)
KERNEL(k_and, DECL32, REP8(OP_AND), SINK32)
KERNEL(k_or, DECL32, REP8(OP_OR), SINK32)
KERNEL(k_sub, DECL32, REP8(OP_SUB), SINK32)
KERNEL(k_mov, DECL32, REP8(OP_MOV), SINK32)
KERNEL(k_lshladd32, DECL32, REP8(OP_LSHLADD32), SINK32)
KERNEL(k_addlshl32, DECL32, REP8(OP_ADDLSHL32), SINK32)
KERNEL(k_xad, DECL32, REP8(OP_XAD), SINK32)
KERNEL(k_andor, DECL32, REP8(OP_ANDOR), SINK32)
KERNEL(k_or3, DECL32, REP8(OP_OR3), SINK32)
KERNEL(k_bfi, DECL32, REP8(OP_BFI), SINK32)
KERNEL(k_cmpu32, DECL32, REP8(OP_CMPU32), SINK32)
KERNEL(k_sdwa, DECL32, REP8(OP_SDWA), SINK32)
KERNEL(k_addsdwa, DECL32, REP8(OP_ADDSDWA), SINK32)
KERNEL(k_mullo_s, DECL32, REP8(OP_MULLO_S), SINK32)
KERNEL(k_xor_s, DECL32, REP8(OP_XOR_S), SINK32)
KERNEL(k_xor_lit, DECL32, REP8(OP_XOR_LIT), SINK32)
KERNEL(k_pkadd16, DECL32, REP8(OP_PKADD16), SINK32)
KERNEL(k_pkmul16, DECL32, REP8(OP_PKMUL16), SINK32)
KERNEL(k_pkmad16, DECL32, REP8(OP_PKMAD16), SINK32)
KERNEL(k_madu16, DECL32, REP8(OP_MADU16), SINK32)
KERNEL(k_dot2, DECL32, REP8(OP_DOT2), SINK32)
KERNEL(k_dot4, DECL32, REP8(OP_DOT4), SINK32)
KERNEL(k_alignbyte, DECL32, REP8(OP_ALIGNBYTE), SINK32)
KERNEL(k_addco1, DECL32, REP8(OP_ADDCO1), SINK32)
KERNEL(k_fma, DECL32, REP8(OP_FMA), SINK32)
KERNEL(k_min, DECL32, REP8(OP_MIN), SINK32)
// the canonical-strand select of the hash kernel: one 64-bit compare feeding two v_cndmask (per iteration: 1 + 2 ops)
#define OP_CMPSEL(x) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(k) : "vcc");
KERNEL(k_cmpsel, DECL32, REP8(OP_CMPSEL), SINK32)
#define OP_CMPSEL2(x) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %1, %0, vcc" : "+v"(x) : "v"(k) : "vcc");
KERNEL(k_cmpsel2, DECL32, REP8(OP_CMPSEL2), SINK32)
#define OP_CMPSELMASK(x) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, 0, -1, vcc\n\tv_and_b32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(x) : "v"(k) : "vcc");
KERNEL(k_cmpselmask, DECL32, REP8(OP_CMPSELMASK), SINK32)
#define OP_MINU32x2(x) asm volatile("v_min_u32 %0, %0, %1\n\tv_max_u32 %0, %0, %1" : "+v"(x) : "v"(k));
KERNEL(k_minmax, DECL32, REP8(OP_MINU32x2), SINK32)

// mixed stream: does a cheap op hide behind an expensive one? 4 xor + 4 alignbit per iteration
#define OP_MIX(x) asm volatile("v_xor_b32 %0, %0, %1\n\tv_alignbit_b32 %0, %0, %1, 7" : "+v"(x) : "v"(k));
KERNEL(k_mix, DECL32, REP8(OP_MIX), SINK32)
#define OP_MIX2(x) asm volatile("v_mul_lo_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(x) : "v"(k));
KERNEL(k_mix2, DECL32, REP8(OP_MIX2), SINK32)

// effective shader clock under a dense VALU load: s_memtime (shader cycles) against s_memrealtime (100 MHz)
__global__ __launch_bounds__(256) void k_clock(uint32_t *out, uint32_t seed, unsigned long long *stamps) {
  DECL32;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < ITERS * 16; ++it) { REP8(OP_MIX2) }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
  SINK32;
}

#define DECL64 uint64_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; uint32_t k = seed | 1; uint64_t k64 = ((uint64_t)k << 32) | k; uint64_t cc
#define SINK64 out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ (uint32_t)((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) >> 32)
#define OP_MAD64(x) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(x), "=s"(cc) : "v"((uint32_t)x), "v"(k));
#define OP_SHL64(x) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(x));
#define OP_SHR64(x) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(x));
#define OP_LSHLADD64(x) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(x) : "v"(k64));
#define OP_CMP64(x) asm volatile("v_cmp_gt_u64 vcc, %0, %1" : : "v"(x), "v"(k64) : "vcc");
#define OP_ADDCO(x) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(*(uint32_t*)&x) : "v"(k) : "vcc");

// the canonical-strand select as the hash kernel has it, and candidate replacements (x: 64-bit, its low word selected)
#define OP_SEL_KERNEL(x) asm volatile("v_cmp_lt_u64 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cndmask_b32 %3, %3, %2, vcc" : : "v"(x), "v"(k64), "v"(*(uint32_t*)&x), "v"(k) : "vcc");
#define OP_SEL_TWICE(x) asm volatile("v_cmp_lt_u64 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc\n\tv_cmp_lt_u64 vcc, %0, %1\n\tv_cndmask_b32 %3, %3, %2, vcc" : : "v"(x), "v"(k64), "v"(*(uint32_t*)&x), "v"(k) : "vcc");
#define OP_SEL_SGPR(x) asm volatile("v_cmp_lt_u64 s[20:21], %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %2, %2, %3, s[20:21]\n\tv_cndmask_b32 %3, %3, %2, s[20:21]" : : "v"(x), "v"(k64), "v"(*(uint32_t*)&x), "v"(k) : "s20", "s21");
#define OP_SEL_ONE(x) asm volatile("v_cmp_lt_u64 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc" : : "v"(x), "v"(k64), "v"(*(uint32_t*)&x), "v"(k) : "vcc");
KERNEL(k_sel_kernel, DECL64, REP8(OP_SEL_KERNEL), SINK64)
KERNEL(k_sel_twice, DECL64, REP8(OP_SEL_TWICE), SINK64)
KERNEL(k_sel_sgpr, DECL64, REP8(OP_SEL_SGPR), SINK64)
KERNEL(k_sel_one, DECL64, REP8(OP_SEL_ONE), SINK64)
KERNEL(k_mad64, DECL64, REP8(OP_MAD64), SINK64)
KERNEL(k_shl64, DECL64, REP8(OP_SHL64), SINK64)
KERNEL(k_shr64, DECL64, REP8(OP_SHR64), SINK64)
KERNEL(k_lshladd64, DECL64, REP8(OP_LSHLADD64), SINK64)
KERNEL(k_cmp64, DECL64, REP8(OP_CMP64), SINK64)
KERNEL(k_addco_pair, DECL64, REP8(OP_ADDCO), SINK64)

// LDS random reads: 256-entry tables, index from a cheap LCG per lane
__global__ __launch_bounds__(256) void k_lds_b64(uint32_t *out, uint32_t seed) {
  __shared__ uint64_t tab[4][256];
  for (int j = 0; j < 4; ++j) tab[j][threadIdx.x] = threadIdx.x * 0x9e3779b97f4a7c15ULL + j;
  __syncthreads();
  uint32_t x = threadIdx.x * 2654435761u + seed;
  uint64_t acc = 0;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc += tab[j & 3][(x >> (j * 3)) & 255];
    }
    x = x * 1664525u + 1013904223u;
  }
  out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32);
}
__global__ __launch_bounds__(256) void k_lds_b32(uint32_t *out, uint32_t seed) {
  __shared__ uint32_t tab[4][256];
  for (int j = 0; j < 4; ++j) tab[j][threadIdx.x] = threadIdx.x * 0x9e3779b9u + j;
  __syncthreads();
  uint32_t x = threadIdx.x * 2654435761u + seed;
  uint32_t acc = 0;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc += tab[j & 3][(x >> (j * 3)) & 255];
    }
    x = x * 1664525u + 1013904223u;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef void (*kern_t)(uint32_t *, uint32_t);
struct Entry { const char *name; kern_t fn; int ops_per_iter; };

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 8;  // 8 blocks x 4 waves = 32 waves/CU = 8 waves/SIMD
  uint32_t *d_out;
  CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
  std::vector<Entry> es = {
      {"v_xor_b32", k_xor, 8}, {"v_add_u32", k_add, 8}, {"v_mul_lo_u32", k_mullo, 8}, {"v_mul_hi_u32", k_mulhi, 8},
      {"v_mul_u32_u24", k_mul24, 8}, {"v_mad_u32_u24", k_mad24, 8}, {"v_alignbit_b32", k_align, 8},
      {"v_perm_b32", k_perm, 8}, {"v_add3_u32", k_add3, 8}, {"v_lshl_or_b32", k_lshlor, 8}, {"v_bfe_u32", k_bfe, 8},
      {"v_cndmask_b32", k_cndmask, 8}, {"v_mad_u64_u32", k_mad64, 8}, {"v_lshlrev_b64", k_shl64, 8},
      {"v_lshrrev_b64", k_shr64, 8}, {"v_lshl_add_u64", k_lshladd64, 8}, {"v_cmp_gt_u64", k_cmp64, 8},
      {"v_add_co+v_addc pair", k_addco_pair, 8}, {"ds_read_b64 random(+add64)", k_lds_b64, 8},
      {"ds_read_b32 random(+add)", k_lds_b32, 8},
      {"v_lshlrev_b32 (imm)", k_shl32, 8}, {"v_lshrrev_b32 (imm)", k_shr32, 8}, {"v_lshlrev_b32 (vgpr)", k_shlv32, 8},
      {"v_and_b32", k_and, 8}, {"v_or_b32", k_or, 8}, {"v_sub_u32", k_sub, 8}, {"v_mov_b32", k_mov, 8},
      {"v_lshl_add_u32", k_lshladd32, 8}, {"v_add_lshl_u32", k_addlshl32, 8}, {"v_xad_u32", k_xad, 8},
      {"v_and_or_b32", k_andor, 8}, {"v_or3_b32", k_or3, 8}, {"v_bfi_b32", k_bfi, 8}, {"v_cmp_lt_u32", k_cmpu32, 8},
      {"v_lshlrev_b32_sdwa", k_sdwa, 8}, {"v_add_u32_sdwa", k_addsdwa, 8}, {"v_mul_lo_u32 (sgpr)", k_mullo_s, 8},
      {"v_xor_b32 (sgpr)", k_xor_s, 8}, {"v_xor_b32 (literal)", k_xor_lit, 8}, {"v_pk_add_u16", k_pkadd16, 8},
      {"v_pk_mul_lo_u16", k_pkmul16, 8}, {"v_pk_mad_u16", k_pkmad16, 8}, {"v_mad_u32_u16", k_madu16, 8},
      {"v_dot2_u32_u16", k_dot2, 8}, {"v_dot4_u32_u8", k_dot4, 8}, {"v_alignbyte_b32", k_alignbyte, 8},
      {"v_add_co_u32 (alone)", k_addco1, 8}, {"v_fma_f32", k_fma, 8}, {"v_min_u32", k_min, 8},
      {"v_cmp_lt_u32 + v_cndmask (pair)", k_cmpsel, 8}, {"v_cmp + s_nop + 2 v_cndmask (triple)", k_cmpsel2, 8},
      {"v_cmp + cndmask(0,-1) + and + xor (quad)", k_cmpselmask, 8}, {"v_min_u32 + v_max_u32 (pair)", k_minmax, 8},
      {"cmp64 + s_nop + 2 cndmask (as the kernel)", k_sel_kernel, 8}, {"cmp64 + cndmask, twice", k_sel_twice, 8},
      {"cmp64 -> sgpr pair + s_nop + 2 cndmask_e64", k_sel_sgpr, 8}, {"cmp64 + 1 cndmask", k_sel_one, 8},
      {"xor+alignbit pair", k_mix, 8}, {"mul_lo+xor+add triple", k_mix2, 8},

  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
  for (auto &e : es) {
    hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d_out, 12345u + r);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per_launch_s = ms * 1e-3 / 5;
    // wave-instructions per SIMD: 8 waves x ITERS x ops
    const double winstr_per_simd = 8.0 * ITERS * e.ops_per_iter;
    const double cyc = per_launch_s * 2.4e9 / winstr_per_simd;
    printf("%-28s %8.3f ms/launch  -> %6.2f cycles per wave-instr per SIMD (at 2.4 GHz)\n", e.name, per_launch_s * 1e3, cyc);
  }
  {
    unsigned long long *d_st;
    CHECK(hipMalloc(&d_st, (size_t)blocks * 16));
    for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(256), 0, 0, d_out, 777u + r, d_st);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(blocks * 2);
    CHECK(hipMemcpy(st.data(), d_st, (size_t)blocks * 16, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (int b = 0; b < blocks; ++b) if (st[2 * b + 1]) ghz.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    printf("effective shader clock under dense VALU load (s_memtime / s_memrealtime x 100 MHz): median %.3f GHz, min %.3f, max %.3f over %zu workgroups\n",
           ghz[ghz.size() / 2], ghz.front(), ghz.back(), ghz.size());
  }
  return 0;
}
