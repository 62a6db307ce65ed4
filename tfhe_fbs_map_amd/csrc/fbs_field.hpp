// Arithmetic in Z_q, q = 2^64 - 2^32 + 1 ("Goldilocks"), for host and gfx950 device code.
//
// q is the ciphertext modulus AND the NTT modulus: 2^64 = 2^32 - 1 and 2^96 = -1 (mod q), so a
// 128-bit product folds back to 64 bits with shifts and adds only, and the negacyclic NTT of any
// power-of-two size up to 2^31 exists.  CDNA4 has no 64-bit multiplier: a product is four
// v_mad_u64_u32; everything else is 32-bit adds with carry.
//
// Value discipline in the kernels:
//   * "canonical" = in [0, q).  Everything stored to HBM is canonical.
//   * "loose"     = any 64-bit word (a representative, possibly >= q).
//   add_cl(a, b), sub_lc(a, b): exact when the operand marked c is canonical.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FBS_HD __host__ __device__ __forceinline__
#else
#define FBS_HD inline
#endif

namespace fbs {

constexpr uint64_t GQ = 0xFFFFFFFF00000001ull;
constexpr uint64_t GEPS = 0xFFFFFFFFull;  // 2^64 mod q

// (hi:lo) < 2^128 -> loose representative.
FBS_HD uint64_t gl_reduce128(uint64_t lo, uint64_t hi) {
    uint32_t h0 = (uint32_t)hi, h1 = (uint32_t)(hi >> 32);
    uint64_t t;
    bool borrow = __builtin_sub_overflow(lo, (uint64_t)h1, &t);  // lo - h1*2^96  (2^96 = -1)
    t -= borrow ? GEPS : 0;                                      // wrapped below 0: +q == -eps mod 2^64
    uint64_t u = ((uint64_t)h0 << 32) - h0;                      // h0 * 2^64 = h0 * eps
    uint64_t r;
    bool carry = __builtin_add_overflow(t, u, &r);
    r += carry ? GEPS : 0;
    return r;
}

FBS_HD uint64_t gl_canon(uint64_t r) { return r >= GQ ? r - GQ : r; }

FBS_HD uint64_t gl_mul_loose(uint64_t a, uint64_t b) {
    unsigned __int128 p = (unsigned __int128)a * b;
    return gl_reduce128((uint64_t)p, (uint64_t)(p >> 64));
}
FBS_HD uint64_t gl_mul(uint64_t a, uint64_t b) { return gl_canon(gl_mul_loose(a, b)); }

// a loose, b canonical -> loose
FBS_HD uint64_t gl_add_lc(uint64_t a, uint64_t b) {
    uint64_t s;
    bool c = __builtin_add_overflow(a, b, &s);
    s += c ? GEPS : 0;
    return s;
}
// a loose, b canonical -> loose
FBS_HD uint64_t gl_sub_lc(uint64_t a, uint64_t b) {
    uint64_t d;
    bool c = __builtin_sub_overflow(a, b, &d);
    d -= c ? GEPS : 0;
    return d;
}
// both canonical -> canonical
FBS_HD uint64_t gl_add(uint64_t a, uint64_t b) { return gl_canon(gl_add_lc(a, b)); }
FBS_HD uint64_t gl_sub(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    return a < b ? d + GQ : d;
}
FBS_HD uint64_t gl_neg(uint64_t a) { return a ? GQ - a : 0; }

FBS_HD uint64_t gl_from_i64(int64_t v) {
    if (v >= 0) return (uint64_t)v % GQ;
    uint64_t m = (0 - (uint64_t)v) % GQ;
    return m ? GQ - m : 0;
}

inline uint64_t gl_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, a);
        a = gl_mul(a, a);
        e >>= 1;
    }
    return r;
}
inline uint64_t gl_inv(uint64_t a) { return gl_pow(a, GQ - 2); }

}  // namespace fbs
