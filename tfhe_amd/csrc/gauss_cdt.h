/* Noise law of the client-side generators: a centred Gaussian, as the reference's gaussian32
 * (src/libtfhe/numeric-functions.cpp:22-27: normal_distribution<double>(0, sigma), then dtot32 = truncation toward zero
 * of err * 2^32, :32-34), sampled with INTEGER arithmetic only so that the host library (client_keys.c) and the GPU
 * kernels (k_client_*) produce the same bits on every machine.
 *
 *   |z| by inversion of the half-normal distribution function on a grid of 1/1024 sigma: a cumulative table of 9,376
 *   64-bit entries  T[k] = floor(2^64 P(|z| < (k+1)/1024)),  one uniform 64-bit word w -> bin k = #{T[j] <= w};
 *   a second word gives the sign and 20 bits of uniform position inside the bin.  The density is the Gaussian's sampled
 *   1,024 times per sigma and constant in between (total variation distance to N(0, sigma^2) about 2e-4, variance
 *   + 1/(12 * 2^20)); the tail reaches 9.16 sigma, where the 64-bit table runs out (P(|z| > 9.16) < 2^-64).
 *   Torus32 value = sign * floor(z * sigma * 2^32): truncation toward zero, as dtot32.
 *
 * The table is built on the host from a series / continued fraction of erfc in 80-bit long double arithmetic (basic
 * operations only, no libm transcendental) and fingerprinted: a host that computes an entry differently is reported
 * (gauss_cdt_build returns 0) instead of silently changing keys. */
#ifndef TFHE_GAUSS_CDT_H
#define TFHE_GAUSS_CDT_H
#include <stdint.h>

#define GAUSS_CDT_BINS 9376
#define GAUSS_CDT_STEP_LOG2 10 /* bins per sigma = 2^10 */
#define GAUSS_CDT_FINGERPRINT 0x8d67babcacc442faULL /* FNV-1a of the table's bytes */

#include <math.h>
/* erf / erfc from +, -, *, / of long doubles alone (x87 basic operations are correctly rounded on every x86-64, so the
 * table comes out the same on every host; libm's erfcl does NOT: it goes through the x87 transcendental instructions,
 * whose last bits differ between CPU vendors - seen between the authoring container and the GPU box) */
static inline long double gcdt_exp_neg(long double y) { /* e^y, y <= 0 */
    const long double ln2_hi = 0.693147180559890330187045037746429443359375L, ln2_lo = 5.497923018708371174e-14L;
    const long double kf = floorl(y * 1.44269504088896340735992468100189214L + 0.5L);
    const long double r = (y - kf * ln2_hi) - kf * ln2_lo; /* |r| <= 0.35 */
    long double term = 1.0L, sum = 1.0L;
    for (int n = 1; n <= 26; n++) {
        term = term * r / (long double) n;
        sum += term;
    }
    return ldexpl(sum, (int) kf);
}
static inline long double gcdt_erfc(long double x) { /* x > 0 */
    const long double inv_sqrt_pi = 0.564189583547756286948079451560772586L;
    const long double e = gcdt_exp_neg(-x * x);
    if (x < 2.5L) { /* erf(x) = 2/sqrt(pi) e^(-x^2) sum_n 2^n x^(2n+1) / (1 3 5 ... (2n+1)): positive terms only */
        long double term = x, sum = x;
        for (int n = 0; n < 400; n++) {
            term = term * (2.0L * x * x) / (long double) (2 * n + 3);
            sum += term;
            if (term < sum * 1e-24L) break;
        }
        return 1.0L - 2.0L * inv_sqrt_pi * e * sum;
    }
    /* erfc(x) = e^(-x^2)/sqrt(pi) / (x + (1/2)/(x + (2/2)/(x + (3/2)/(x + ...)))), evaluated from the bottom */
    long double f = x;
    for (int n = 400; n >= 1; n--) f = x + (0.5L * (long double) n) / f;
    return inv_sqrt_pi * e / f;
}
/* host only; returns 1 if the table has the expected fingerprint, 0 otherwise; *fp_out receives the fingerprint */
static inline int gauss_cdt_build(uint64_t *T /*[GAUSS_CDT_BINS]*/, uint64_t *fp_out) {
    const long double two64 = 18446744073709551616.0L, inv = 0.70710678118654752440084436210484903928L / 1024.0L;
    uint64_t h = 1469598103934665603ULL, prev = 0;
    for (int k = 0; k < GAUSS_CDT_BINS; k++) {
        /* P(|z| >= x) = erfc(x / sqrt 2): the table entry is 2^64 minus its 64-bit image (accurate in the tail) */
        const long double tail = two64 * gcdt_erfc((long double) (k + 1) * inv);
        uint64_t t = tail >= two64 ? 0 : (uint64_t) 0 - (uint64_t) ceill(tail);
        if (tail < 1.0L || k == GAUSS_CDT_BINS - 1) t = UINT64_MAX;
        if (t < prev) t = prev; /* monotone whatever the last bit of the arithmetic does */
        prev = t;
        T[k] = t;
        for (int b = 0; b < 8; b++) {
            h ^= (t >> (8 * b)) & 0xff;
            h *= 1099511628211ULL;
        }
    }
    if (fp_out) *fp_out = h;
    return GAUSS_CDT_FINGERPRINT == 0 || h == GAUSS_CDT_FINGERPRINT;
}

/* magnitude bin of a uniform 64-bit word: the number of table entries <= w (the last entry is 2^64 - 1: bin < BINS) */
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline int gauss_cdt_bin(const uint64_t *T, uint64_t w) {
    int lo = 0, hi = GAUSS_CDT_BINS - 1; /* answer in [lo, hi] */
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (T[mid] <= w) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
/* noise with standard deviation m / 2^40 on the torus (m / 2^8 Torus32 units) from two uniform words */
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline int32_t gauss_cdt_value(const uint64_t *T, uint64_t w_mag, uint64_t w_pos, int64_t m) {
    const uint64_t pos = ((uint64_t) gauss_cdt_bin(T, w_mag) << 20) | ((w_pos >> 43) & 0xfffff); /* z * 2^30, < 2^34 */
    const int64_t mag = (int64_t) ((pos * (uint64_t) m) >> 38);                                   /* m < 2^26 */
    return (int32_t) ((w_pos >> 63) ? -mag : mag);
}
#endif
