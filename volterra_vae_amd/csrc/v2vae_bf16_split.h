// Split arithmetic on the bf16 matrix pipe: a float as the exact sum of three bf16 pieces.
//
// x = p1 + p2 + p3 with every piece a bf16 (8 significand bits): p1 = bf16(x), p2 = bf16(x - p1), p3 = x - p1 - p2 (both
// subtractions are exact, and what is left after two pieces has at most 8 significant bits).  A product of two split floats is
// the sum of nine piece products; the six with (piece index sum) <= 4 carry everything down to 2^-24 of the whole and are what
// the kernels accumulate (in float, by the matrix instruction).
//
// Two ways to take a piece:
//  * round to nearest (bf16_split3_rne): |p2| <= 2^-9 |x|, |p3| <= 2^-18 |x|, signs of the later pieces at random.  The three
//    dropped products are then <= 2^-26 of the whole and of random sign.  Used wherever an operand is split ONCE and reused
//    (tables, samples, posterior means): three integer instructions per rounding.
//  * truncation (bf16_split3_trunc): one AND per piece, pieces of one sign, |p2| < 2^-8 |x|, |p3| < 2^-16 |x|.  Used for the
//    operands that are split in registers per (window, tap) or (symbol, lag), where the instruction count is the bound.
//    With the OTHER operand rounded to nearest the dropped products p2 q3 + p3 q2 have random sign and stay below 2^-24; two
//    truncated operands would drop up to 2^-23 of every product, all of one sign - a bias the soft demapper amplifies.
#ifndef V2VAE_BF16_SPLIT_H
#define V2VAE_BF16_SPLIT_H
#include <hip/hip_runtime.h>

namespace v2vae {

// the float nearest to x that is a bf16 (ties to even), as its bit pattern (low half zero)
__device__ __forceinline__ unsigned bf16_rne_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
}

// pieces as the HIGH halves of three words (the low halves of the first two are zero; the third is exact as it is)
__device__ __forceinline__ void bf16_split3_rne(float x, unsigned* p1, unsigned* p2, unsigned* p3) {
    const unsigned b1 = bf16_rne_bits(x);
    const float r1 = x - __uint_as_float(b1);
    const unsigned b2 = bf16_rne_bits(r1);
    const float r2 = r1 - __uint_as_float(b2);
    *p1 = b1; *p2 = b2; *p3 = __float_as_uint(r2);
}
__device__ __forceinline__ void bf16_split3_trunc(float x, unsigned* p1, unsigned* p2, unsigned* p3) {
    const unsigned b1 = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(b1);
    const unsigned b2 = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(b2);
    *p1 = b1; *p2 = b2; *p3 = __float_as_uint(r2);
}
// piece pc (0, 1, 2) of x, rounded to nearest, as 16 bits
__device__ __forceinline__ unsigned bf16_piece_rne(float x, int pc) {
    unsigned p[3];
    bf16_split3_rne(x, &p[0], &p[1], &p[2]);
    return p[pc] >> 16;
}

}  // namespace v2vae
#endif
