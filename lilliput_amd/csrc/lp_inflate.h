// lp_inflate.h -- one-shot inflate of an ordinary zlib stream into a buffer of known size (see lp_inflate.cpp)
#pragma once
#include <stddef.h>
#include <stdint.h>

#define LP_INFLATE_PAD 32 // readable (zero) bytes the caller keeps behind the stream: the bit reader loads eight bytes at a time
// 1: in[0 .. in_len) is a zlib stream that inflates to exactly out_len bytes, ends with its last byte and carries the right Adler-32;
// out holds the data. 0: anything else (out is undefined) -- ask zlib.
int lp_inflate_exact(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len);
// zlib's adler32() / crc32() with the same arguments and results, at memory speed where the CPU has AVX2 / PCLMULQDQ
uint32_t lp_adler32(uint32_t adler, const uint8_t* p, size_t n);
uint32_t lp_crc32(uint32_t crc, const uint8_t* p, size_t n);
