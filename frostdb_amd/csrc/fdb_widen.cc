// fdb_widen.cc — host side of the narrow result transport (fdb_hash.cpp, finish_columns_hash): dictionary indices that crossed
// PCIe as uint8 / uint16 are widened to the uint32 Arrow's dictionary<uint32, …> columns carry (the reference's result type,
// aggregate.go:499-502 builds group columns with the input field's type). Plain C++ (no HIP): compiled for the host only.
#include <immintrin.h>

#include <cstddef>
#include <cstdint>
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

}  // namespace

void widen_indices(const void* src, int width, uint32_t* dst, size_t n) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
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
