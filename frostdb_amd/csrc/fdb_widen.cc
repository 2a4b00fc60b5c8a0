// fdb_widen.cc — host side of the narrow result transport (fdb_hash.cpp, finish_columns_hash): dictionary indices that crossed
// PCIe as uint8 / uint16 are widened to the uint32 Arrow's dictionary<uint32, …> columns carry (the reference's result type,
// aggregate.go:499-502 builds group columns with the input field's type). Plain C++ (no HIP): compiled for the host only.
#include <immintrin.h>

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace fdb {

namespace {

__attribute__((target("avx2"))) void widen8_avx2(const uint8_t* s, uint32_t* d, size_t n) {
  size_t i = 0;
  while (i < n && ((uintptr_t)(d + i) & 31u) != 0) { d[i] = s[i]; i++; }
  // 32 indices per iteration; the destination is written once and not read again by this thread: streaming stores
  for (; i + 32 <= n; i += 32) {
    const __m128i lo = _mm_loadu_si128((const __m128i*)(s + i)), hi = _mm_loadu_si128((const __m128i*)(s + i + 16));
    _mm256_stream_si256((__m256i*)(d + i), _mm256_cvtepu8_epi32(lo));
    _mm256_stream_si256((__m256i*)(d + i + 8), _mm256_cvtepu8_epi32(_mm_srli_si128(lo, 8)));
    _mm256_stream_si256((__m256i*)(d + i + 16), _mm256_cvtepu8_epi32(hi));
    _mm256_stream_si256((__m256i*)(d + i + 24), _mm256_cvtepu8_epi32(_mm_srli_si128(hi, 8)));
  }
  _mm_sfence();
  for (; i < n; i++) d[i] = s[i];
}

__attribute__((target("avx2"))) void widen16_avx2(const uint16_t* s, uint32_t* d, size_t n) {
  size_t i = 0;
  while (i < n && ((uintptr_t)(d + i) & 31u) != 0) { d[i] = s[i]; i++; }
  for (; i + 16 <= n; i += 16) {
    _mm256_stream_si256((__m256i*)(d + i), _mm256_cvtepu16_epi32(_mm_loadu_si128((const __m128i*)(s + i))));
    _mm256_stream_si256((__m256i*)(d + i + 8), _mm256_cvtepu16_epi32(_mm_loadu_si128((const __m128i*)(s + i + 8))));
  }
  _mm_sfence();
  for (; i < n; i++) d[i] = s[i];
}

// Copies into the pinned record slab (Plan::push, small host records): the destination is written once and next read by the DMA
// engine, never by this CPU — streaming stores skip the read-for-ownership a cached store pays (a third of the copy's memory
// traffic, which is what 8 chains copying at once run out of). `dst` 32-byte aligned (slab pieces are 256-byte aligned).
__attribute__((target("avx2"))) void copy_stream_avx2(unsigned char* d, const unsigned char* s, size_t n) {
  size_t i = 0;
  for (; i + 128 <= n; i += 128) {
    const __m256i a = _mm256_loadu_si256((const __m256i*)(s + i)), b = _mm256_loadu_si256((const __m256i*)(s + i + 32));
    const __m256i c = _mm256_loadu_si256((const __m256i*)(s + i + 64)), e = _mm256_loadu_si256((const __m256i*)(s + i + 96));
    _mm256_stream_si256((__m256i*)(d + i), a); _mm256_stream_si256((__m256i*)(d + i + 32), b);
    _mm256_stream_si256((__m256i*)(d + i + 64), c); _mm256_stream_si256((__m256i*)(d + i + 96), e);
  }
  _mm_sfence();
  if (i < n) std::memcpy(d + i, s + i, n - i);
}
// the same for uint32 dictionary indices, returning their maximum (the validation pass for free)
__attribute__((target("avx2"))) uint32_t copy_stream_max_u32_avx2(uint32_t* d, const uint32_t* s, size_t n) {
  __m256i mx = _mm256_setzero_si256();
  size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    const __m256i a = _mm256_loadu_si256((const __m256i*)(s + i)), b = _mm256_loadu_si256((const __m256i*)(s + i + 8));
    const __m256i c = _mm256_loadu_si256((const __m256i*)(s + i + 16)), e = _mm256_loadu_si256((const __m256i*)(s + i + 24));
    _mm256_stream_si256((__m256i*)(d + i), a); _mm256_stream_si256((__m256i*)(d + i + 8), b);
    _mm256_stream_si256((__m256i*)(d + i + 16), c); _mm256_stream_si256((__m256i*)(d + i + 24), e);
    mx = _mm256_max_epu32(mx, _mm256_max_epu32(_mm256_max_epu32(a, b), _mm256_max_epu32(c, e)));
  }
  _mm_sfence();
  uint32_t lanes[8];
  _mm256_storeu_si256((__m256i*)lanes, mx);
  uint32_t m = 0;
  for (uint32_t v : lanes) m = v > m ? v : m;
  for (; i < n; i++) { d[i] = s[i]; m = s[i] > m ? s[i] : m; }
  return m;
}

}  // namespace

void copy_stream(void* dst, const void* src, size_t n) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2 && n >= 4096 && ((uintptr_t)dst & 31u) == 0) return copy_stream_avx2((unsigned char*)dst, (const unsigned char*)src, n);
  std::memcpy(dst, src, n);
}

uint32_t copy_stream_max_u32(uint32_t* dst, const uint32_t* src, size_t n) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2 && n >= 1024 && ((uintptr_t)dst & 31u) == 0) return copy_stream_max_u32_avx2(dst, src, n);
  uint32_t m = 0;
  for (size_t i = 0; i < n; i++) { dst[i] = src[i]; m = src[i] > m ? src[i] : m; }
  return m;
}

namespace {
// Sub-byte indices (2 / 4 bits per row, low bits first): one table entry per source byte holds its 4 / 2 indices as uint32.
struct SubByteTables {
  alignas(16) uint32_t two[256][4];
  alignas(8) uint32_t four[256][2];
  SubByteTables() {
    for (int b = 0; b < 256; b++) {
      for (int k = 0; k < 4; k++) two[b][k] = (uint32_t)(b >> (2 * k)) & 3u;
      four[b][0] = (uint32_t)b & 15u; four[b][1] = (uint32_t)b >> 4;
    }
  }
};
const SubByteTables& sub_byte_tables() { static const SubByteTables t; return t; }

// AVX-512: 16 indices per instruction group instead of 4 through a table — a dword of 16 two-bit codes (two dwords of 8 four-bit codes)
// is broadcast, every lane shifts its code down and masks it; with `table` (≤ 16 entries, padded to 16: what a 2- / 4-bit code can
// address) the codes then pick their values with one permute. A big Finish widens 1.28 GB this way on the few CPUs a container is given:
// the table version's 16 bytes per load + store pair was what bounded it (≈ 18 GB/s per core). Returns the number of indices written
// (a multiple of 16; 0 when `d` cannot be brought to 64-byte alignment on a source byte boundary): the caller finishes the rest.
__attribute__((target("avx512f"))) size_t widen_bits_avx512(const uint8_t* s, int bits, const uint32_t* table16, uint32_t* d, size_t n) {
  if (((uintptr_t)d & 15u) != 0) return 0;
  size_t i = 0;
  const size_t head = ((64u - ((uintptr_t)d & 63u)) & 63u) / 4;  // 0, 4, 8 or 12 indices: whole source bytes for either width
  if (head > n) return 0;
  const uint32_t mask = (1u << bits) - 1u;
  for (; i < head; i++) {
    const uint32_t c = (uint32_t)(s[(i * (size_t)bits) >> 3] >> ((i * (size_t)bits) & 7)) & mask;
    d[i] = table16 ? table16[c] : c;
  }
  const __m512i tab = table16 ? _mm512_loadu_si512((const void*)table16) : _mm512_setzero_si512();
  const __m512i vmask = _mm512_set1_epi32((int)mask);
  if (bits == 2) {
    const __m512i sh = _mm512_setr_epi32(0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30);
    for (; i + 64 <= n; i += 64) {
      uint32_t w[4];
      std::memcpy(w, s + (i >> 2), 16);
#pragma GCC unroll 4
      for (int k = 0; k < 4; k++) {
        __m512i v = _mm512_and_si512(_mm512_srlv_epi32(_mm512_set1_epi32((int)w[k]), sh), vmask);
        if (table16) v = _mm512_permutexvar_epi32(v, tab);
        _mm512_stream_si512((__m512i*)(d + i + 16 * k), v);
      }
    }
    for (; i + 16 <= n; i += 16) {
      uint32_t w;
      std::memcpy(&w, s + (i >> 2), 4);
      __m512i v = _mm512_and_si512(_mm512_srlv_epi32(_mm512_set1_epi32((int)w), sh), vmask);
      if (table16) v = _mm512_permutexvar_epi32(v, tab);
      _mm512_stream_si512((__m512i*)(d + i), v);
    }
  } else {
    const __m512i sh = _mm512_setr_epi32(0, 4, 8, 12, 16, 20, 24, 28, 0, 4, 8, 12, 16, 20, 24, 28);
    const __m512i half = _mm512_setr_epi32(0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1);
    for (; i + 16 <= n; i += 16) {
      unsigned long long w;
      std::memcpy(&w, s + (i >> 1), 8);
      const __m512i two = _mm512_permutexvar_epi32(half, _mm512_castsi128_si512(_mm_cvtsi64_si128((long long)w)));
      __m512i v = _mm512_and_si512(_mm512_srlv_epi32(two, sh), vmask);
      if (table16) v = _mm512_permutexvar_epi32(v, tab);
      _mm512_stream_si512((__m512i*)(d + i), v);
    }
  }
  _mm_sfence();
  return i;
}

void widen_bits(const uint8_t* s, int bits, uint32_t* d, size_t n) {
  const SubByteTables& T = sub_byte_tables();
  size_t i = 0;
  static const bool avx512 = __builtin_cpu_supports("avx512f") && std::getenv("FDB_NO_AVX512") == nullptr;
  if (avx512 && n >= 256) i = widen_bits_avx512(s, bits, nullptr, d, n);
  if (bits == 2) {
    if (((uintptr_t)d & 15u) == 0) {
      for (; i + 4 <= n; i += 4) _mm_stream_si128((__m128i*)(d + i), _mm_load_si128((const __m128i*)T.two[s[i >> 2]]));
      _mm_sfence();
    }
    for (; i < n; i++) d[i] = (uint32_t)(s[i >> 2] >> (2 * (i & 3))) & 3u;
  } else {
    for (; i + 2 <= n; i += 2) std::memcpy(d + i, T.four[s[i >> 1]], 8);
    for (; i < n; i++) d[i] = (uint32_t)(s[i >> 1] >> (4 * (i & 1))) & 15u;
  }
}
}  // namespace

// The same through a table: what crossed PCIe is the RANK of a key id among the ids present in the result (fdb_kernels.h
// FdbPresentArgs); dst[i] = table[rank]. Ranks ≥ table_len do not occur (the device ranked what it found); clamped all the same.
void widen_indices_mapped(const void* src, int width, const uint32_t* table, size_t table_len, uint32_t* dst, size_t n) {
  if (table_len == 0) { std::memset(dst, 0, n * 4); return; }
  auto at = [&](uint32_t r) { return table[r < table_len ? r : table_len - 1]; };
  const uint8_t* s = (const uint8_t*)src;
  if (width == -2 || width == -4) {
    const int bits = -width, per = 8 / bits;
    static const bool avx512 = __builtin_cpu_supports("avx512f") && std::getenv("FDB_NO_AVX512") == nullptr;
    if (avx512 && n >= 256) {
      alignas(64) uint32_t t16[16];
      for (uint32_t c = 0; c < 16; c++) t16[c] = at(c);
      const size_t done = widen_bits_avx512(s, bits, t16, dst, n);
      for (size_t i = done; i < n; i++) dst[i] = at((uint32_t)(s[(i * bits) >> 3] >> ((i * bits) & 7)) & ((1u << bits) - 1u));
      return;
    }
    alignas(16) uint32_t lut[256][4];  // a source byte → its 4 (2) indices
    for (int b = 0; b < 256; b++) for (int k = 0; k < per; k++) lut[b][k] = at((uint32_t)(b >> (bits * k)) & ((1u << bits) - 1u));
    size_t i = 0;
    if (bits == 2 && ((uintptr_t)dst & 15u) == 0) {
      for (; i + 4 <= n; i += 4) _mm_stream_si128((__m128i*)(dst + i), _mm_load_si128((const __m128i*)lut[s[i >> 2]]));
      _mm_sfence();
    } else if (bits == 4) {
      for (; i + 2 <= n; i += 2) std::memcpy(dst + i, lut[s[i >> 1]], 8);
    }
    for (; i < n; i++) dst[i] = at((uint32_t)(s[(i * bits) >> 3] >> ((i * bits) & 7)) & ((1u << bits) - 1u));
  } else if (width == 1) {
    uint32_t lut[256];
    for (int b = 0; b < 256; b++) lut[b] = at((uint32_t)b);
    for (size_t i = 0; i < n; i++) dst[i] = lut[s[i]];
  } else if (width == 2) {
    const uint16_t* s2 = (const uint16_t*)src;
    for (size_t i = 0; i < n; i++) dst[i] = at(s2[i]);
  } else {
    const uint32_t* s4 = (const uint32_t*)src;
    for (size_t i = 0; i < n; i++) dst[i] = at(s4[i]);
  }
}

// width: 1 / 2 / 4 bytes per index, or −2 / −4: that many BITS per index (rows packed low bits first)
void widen_indices(const void* src, int width, uint32_t* dst, size_t n) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (width < 0) return widen_bits((const uint8_t*)src, -width, dst, n);
  if (width == 1) {
    if (avx2) return widen8_avx2((const uint8_t*)src, dst, n);
    const uint8_t* s = (const uint8_t*)src;
    for (size_t i = 0; i < n; i++) dst[i] = s[i];
  } else if (width == 2) {
    if (avx2) return widen16_avx2((const uint16_t*)src, dst, n);
    const uint16_t* s = (const uint16_t*)src;
    for (size_t i = 0; i < n; i++) dst[i] = s[i];
  } else {
    std::memcpy(dst, src, n * 4);
  }
}

}  // namespace fdb
