// Arithmetic in Z_q, q = 2^46 - 62*2^13 + 1 = 0x3FFFFFF84001 (prime, 2^14 | q-1), the ciphertext modulus AND the
// NTT modulus of libfbsexec.  Residues travel in 64-bit words.
//
// Why 46 bits.  gfx950 has no 64-bit integer multiplier: a 64x64 modular product costs ~25 VALU instructions
// (4 v_mad_u64_u32 + carry chains), and the blind rotation is nothing but such products.  The FP64 pipe runs
// v_fma_f64 at the same issue rate as any 64-bit integer instruction and an FMA is an EXACT 53-bit integer
// multiply-add.  With q < 2^46:
//     h = x*w (rounded), l = fma(x, w, -h) (exact remainder), qh = rint(h / q), r = fma(-qh, q, h) + l
// is the exact residue of x*w in (-0.75 q, 0.75 q) for |x| < 2^50, 6 instructions, no carries, no compares.
// Values are kept as integer-valued doubles in a signed, LAZY range: a butterfly is the product plus one add and
// one sub, and ten Cooley-Tukey stages grow |x| by at most 0.75 q each (7.5 q < 2^50) -- no range fix-ups at all.
// Measured on MI355X (tools/bf_bench.hip): 3.48e12 butterflies/s against 1.07e12 for a u64 Goldilocks butterfly.
// Everything that leaves the kernels (ciphertexts, key material in standard layout) is a canonical integer in
// [0, q) in a uint64 word, so the CPU oracle -- which computes the same residues with integer arithmetic -- is
// compared word for word.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FBS_HD __host__ __device__ __forceinline__
#define FBS_D __device__ __forceinline__
#else
#define FBS_HD inline
#endif

namespace fbs {

constexpr uint64_t FQ = 0x3FFFFFF84001ull;
constexpr int FQ_BITS = 46;
constexpr uint64_t FQ_GENERATOR = 7;   // generates Z_q^*: q - 1 = 2^14 * 3^2 * 5 * 95443717

// ---- integer arithmetic on canonical residues (host code, and the non-hot kernels) ------------------------
FBS_HD uint64_t fq_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    return s >= FQ ? s - FQ : s;
}
FBS_HD uint64_t fq_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + FQ - b; }
FBS_HD uint64_t fq_neg(uint64_t a) { return a ? FQ - a : 0; }
FBS_HD uint64_t fq_mul(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % FQ); }
FBS_HD uint64_t fq_from_i64(int64_t v) {
    if (v >= 0) return (uint64_t)v % FQ;
    uint64_t m = (0 - (uint64_t)v) % FQ;
    return m ? FQ - m : 0;
}
// uniform residue from a 64-bit random word: top FQ_BITS bits, folded once (bias 2^-27)
FBS_HD uint64_t fq_fold(uint64_t r) {
    r >>= 64 - FQ_BITS;
    return r >= FQ ? r - FQ : r;
}
inline uint64_t fq_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = fq_mul(r, a);
        a = fq_mul(a, a);
        e >>= 1;
    }
    return r;
}
inline uint64_t fq_inv(uint64_t a) { return fq_pow(a, FQ - 2); }
// centred representative in (-q/2, q/2] as a double (how twiddles and key polynomials are stored for the kernels)
inline double fq_centered(uint64_t a) { return a > FQ / 2 ? -(double)(FQ - a) : (double)a; }

// ---- device-side exact FP64 arithmetic ---------------------------------------------------------------------
#if defined(__HIPCC__)
constexpr double FP_Q = 70368743669761.0;            // q, exactly representable
constexpr double FP_QINV = 1.0 / 70368743669761.0;   // correctly rounded at compile time
constexpr double FP_MAGIC = 4503599627370496.0;      // 2^52: x + 2^52 exposes the integer x in the mantissa

// exact residue of x*w in (-0.75q, 0.75q); needs |x| < 2^50 and |w| <= q/2 (twiddles/keys are stored centred).
// Up to |x| < 2^52 the result is still the exact residue (h - qh*q is an integer below 2^47, so the FMA cannot round;
// l is the exact low part by construction); only the quotient estimate loosens (error <= |qh| * 2^-52 < 0.3), so the
// result lies in (-0.8q, 0.8q).
FBS_D double fp_mulmod(double x, double w) {
    const double h = x * w;
    const double l = __builtin_fma(x, w, -h);
    const double qh = __builtin_rint(h * FP_QINV);
    return __builtin_fma(-qh, FP_Q, h) + l;
}
// the same when the product itself is exact in a double (|x * w| < 2^53, e.g. a gadget digit times a twiddle): 4 instructions
FBS_D double fp_mulmod_exact(double x, double w) {
    const double h = x * w;
    return __builtin_fma(-__builtin_rint(h * FP_QINV), FP_Q, h);
}
// representative in [-q/2, q/2] (up to the rounding of x/q); needs |x| < 2^52
FBS_D double fp_center(double x) { return __builtin_fma(-__builtin_rint(x * FP_QINV), FP_Q, x); }
// canonical representative in [0, q); needs |x| < 2^52
FBS_D double fp_canon(double x) {
    const double c = fp_center(x);
    return c < 0.0 ? c + FP_Q : c;
}
// canonical representative in [0, q) of an integer |x| <= 12 q, 3 instructions.  floor(x/q) is exact here: integers
// that are not multiples of q sit at least 2^-46 away from an integer quotient (rounding error < 2^-48), and the
// multiples k*q, |k| <= 12, give exactly k (checked below for this q and this rounding of 1/q).
FBS_D double fp_canon_near(double x) { return __builtin_fma(-__builtin_floor(x * FP_QINV), FP_Q, x); }
#define FBS_CHECK_MULTIPLE(k) static_assert(((k) * FP_Q) * FP_QINV == (k), "k*q/q must be exact")
FBS_CHECK_MULTIPLE(1.0); FBS_CHECK_MULTIPLE(2.0); FBS_CHECK_MULTIPLE(3.0); FBS_CHECK_MULTIPLE(4.0);
FBS_CHECK_MULTIPLE(5.0); FBS_CHECK_MULTIPLE(6.0); FBS_CHECK_MULTIPLE(7.0); FBS_CHECK_MULTIPLE(8.0);
FBS_CHECK_MULTIPLE(9.0); FBS_CHECK_MULTIPLE(10.0); FBS_CHECK_MULTIPLE(11.0); FBS_CHECK_MULTIPLE(12.0);
#undef FBS_CHECK_MULTIPLE
// 2^e as a double, e in the normal range
FBS_D double fp_exp2i(int e) { return __longlong_as_double((long long)(1023 + e) << 52); }
// integer-valued double in [0, 2^52) <-> uint64
FBS_D uint64_t fp_to_u64(double x) { return (uint64_t)__double_as_longlong(x + FP_MAGIC) & 0x000FFFFFFFFFFFFFull; }
FBS_D double fp_from_u64(uint64_t v) {   // v < 2^52
    return __longlong_as_double((long long)(v | 0x4330000000000000ull)) - FP_MAGIC;
}
#endif

}  // namespace fbs
