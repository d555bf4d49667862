// mdc_bignum.h -- exact decimal -> binary32/binary64 for texts of any length (device).
//
// Second stage of the typed conversion: mdc_numparse.h covers texts with up to 38 significant
// and 38 fractional digits in 128-bit arithmetic; everything longer comes here, one lane per
// cell, with multi-precision integers in per-lane scratch memory.  The result is what glibc's
// strtod/strtof return for the filtered text, including their ERANGE verdicts, because the
// reference's std::stod/std::stof turn ERANGE into an exception and the typed overloads turn
// the exception into "false" (reference public/_public.cpp:591-625).
//
// Method.  The text is  [+-] digits [. digits]  (mdc_numparse.h explains why nothing else can
// occur).  Up to 800 significant digits are accumulated exactly into D (later digits only set
// a sticky flag: no decimal needs more than 767 significant digits to decide a binary64
// rounding, and a dropped digit cannot move the first 64 quotient bits for any magnitude that
// still converts to a finite, non-tiny result).  With k fractional digits taken the value is
// D / (5^k * 2^k); D and 5^k are aligned to the same top bit and a restoring division produces
// 64 quotient bits plus an exact "remainder != 0" flag, from which the value is rounded to
// nearest-even once, at the target precision.
#pragma once
#include <stdint.h>

namespace mdc {

constexpr int BN = 96;            // limbs of 32 bits: 3072 bits >= 10^800 and 5^1200
constexpr uint32_t SIG_MAX = 800; // significant digits kept exactly

struct Big {
    uint32_t w[BN];
    int n;  // limbs in use (w[n..] are zero)
};

__device__ inline void big_set(Big &a, uint32_t v) {
    for (int i = 0; i < BN; ++i) a.w[i] = 0;
    a.w[0] = v;
    a.n = v ? 1 : 0;
}
// a = a*m + add
__device__ inline void big_muladd(Big &a, uint32_t m, uint32_t add) {
    uint64_t carry = add;
    for (int i = 0; i < a.n; ++i) {
        uint64_t t = (uint64_t)a.w[i] * m + carry;
        a.w[i] = (uint32_t)t;
        carry = t >> 32;
    }
    if (carry && a.n < BN) a.w[a.n++] = (uint32_t)carry;
}
__device__ inline int big_bitlen(const Big &a) {
    for (int i = a.n - 1; i >= 0; --i)
        if (a.w[i]) return 32 * i + 32 - __clz((int)a.w[i]);
    return 0;
}
// dst = src << s, dst.n = BN
__device__ inline void big_shl(Big &dst, const Big &src, int s) {
    const int ls = s >> 5, bs = s & 31;
    for (int i = BN - 1; i >= 0; --i) {
        uint32_t lo = (i - ls - 1 >= 0) ? src.w[i - ls - 1] : 0;
        uint32_t hi = (i - ls >= 0) ? src.w[i - ls] : 0;
        dst.w[i] = bs ? (hi << bs) | (lo >> (32 - bs)) : hi;
    }
    dst.n = BN;
}
__device__ inline bool big_ge(const Big &a, const Big &b) {
    for (int i = BN - 1; i >= 0; --i)
        if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
    return true;
}
__device__ inline void big_sub(Big &a, const Big &b) {
    uint32_t borrow = 0;
    for (int i = 0; i < BN; ++i) {
        uint64_t t = (uint64_t)a.w[i] - b.w[i] - borrow;
        a.w[i] = (uint32_t)t;
        borrow = (uint32_t)(t >> 63);
    }
}
__device__ inline bool big_shl1(Big &a) {  // returns the bit shifted out
    uint32_t c = 0;
    for (int i = 0; i < BN; ++i) {
        uint32_t nc = a.w[i] >> 31;
        a.w[i] = (a.w[i] << 1) | c;
        c = nc;
    }
    return c != 0;
}
__device__ inline bool big_nonzero(const Big &a) {
    for (int i = 0; i < BN; ++i)
        if (a.w[i]) return true;
    return false;
}

// Rounds the quotient q (leading one at bit `top`, `sticky` = nonzero bits below q) with
// unbiased exponent e0 to a binary format with p significand bits and exponent range
// [emin, emax]; returns false for what strtod/strtof flag with ERANGE.
__device__ inline bool round_format(uint64_t q, int top, bool sticky, int e0, int p, int emin, int emax,
                                    uint64_t &frac, int &biased) {
    int sh = top + 1 - p;
    uint64_t mant = q >> sh;
    uint64_t rem = q & ((1ull << sh) - 1);
    uint64_t half = 1ull << (sh - 1);
    int e = e0;
    if (rem > half || (rem == half && (sticky || (mant & 1)))) {
        ++mant;
        if (mant >> p) {
            mant >>= 1;
            ++e;
        }
    }
    if (e > emax) return false;  // overflow
    if (e >= emin) {
        frac = mant & ((1ull << (p - 1)) - 1);
        biased = e - emin + 1;
        return true;
    }
    // tiny after rounding: representable only if no nonzero bit is lost at the subnormal quantum
    int nb = e0 - emin + p;  // bits of the value above 2^(emin-p+1)
    if (nb <= 0) return false;
    int sh2 = top + 1 - nb;
    if (sticky || (q & ((1ull << sh2) - 1))) return false;  // inexact and tiny: ERANGE
    frac = q >> sh2;
    biased = 0;
    return true;
}

// Exact conversion of the filtered text s[0..len) (only digits, '+', '-', '.' are looked at;
// other bytes are skipped as picknumber would).  is64: binary64, else binary32.
__device__ inline bool big_convert(const uint8_t *s, uint32_t len, bool is64, uint64_t &out) {
    Big D;
    big_set(D, 0);
    uint32_t pend = 0, pcnt = 0;  // up to nine decimal digits waiting to enter D
    uint32_t sig = 0, intsig = 0, pz = 0;
    int64_t k = 0;
    uint32_t state = 0;
    bool neg = false, any = false, sticky = false;
    const uint32_t P10[10] = {1u, 10u, 100u, 1000u, 10000u, 100000u, 1000000u, 10000000u, 100000000u, 1000000000u};
    auto push = [&](uint32_t d) {
        pend = pend * 10 + d;
        if (++pcnt == 9) {
            big_muladd(D, 1000000000u, pend);
            pend = 0;
            pcnt = 0;
        }
        ++sig;
    };
    for (uint32_t i = 0; i < len && state != 4; ++i) {
        const uint32_t b = s[i];
        if (b == '+' || b == '-') {
            if (state == 0) {
                neg = (b == '-');
                state = 1;
            } else {
                state = 4;
            }
            continue;
        }
        if (b == '.') {
            state = state <= 2 ? 3 : 4;
            continue;
        }
        const uint32_t d = b - '0';
        if (d > 9) continue;
        any = true;
        if (state < 3) {
            state = 2;
            if (sig == 0 && d == 0) continue;
            if (sig < SIG_MAX)
                push(d);
            else if (d)
                sticky = true;
            ++intsig;
        } else {
            if (d == 0) {
                ++pz;
                continue;
            }
            if (sig == 0) {
                k += pz;  // zeros right after the point only scale the value
            } else {
                for (; pz; --pz)
                    if (sig < SIG_MAX) {
                        push(0);
                        ++k;
                    }
            }
            pz = 0;
            if (sig < SIG_MAX) {
                push(d);
                ++k;
            } else {
                sticky = true;
            }
        }
    }
    if (!any) return false;
    if (pcnt) big_muladd(D, P10[pcnt], pend);
    const uint64_t sign = neg ? 1ull : 0ull;
    if (sig == 0) {
        out = is64 ? (sign << 63) : (sign << 31);
        return true;
    }
    if (intsig > 309) return false;            // >= 10^309: overflow in either format
    if (k - (int64_t)sig > 400) return false;  // < 10^-400: underflow in either format
    Big T;
    big_set(T, 1);
    {
        int64_t kk = k;
        for (; kk >= 13; kk -= 13) big_muladd(T, 1220703125u, 0);  // 5^13
        uint32_t f = 1;
        for (; kk > 0; --kk) f *= 5;
        big_muladd(T, f, 0);
    }
    const int nbD = big_bitlen(D), nbT = big_bitlen(T);
    Big R, Tn;
    big_shl(R, D, 32 * BN - nbD);
    big_shl(Tn, T, 32 * BN - nbT);
    uint64_t q = 0;
    for (int i = 0; i < 64; ++i) {
        bool carry = false;
        if (i > 0) carry = big_shl1(R);
        const bool bit = carry || big_ge(R, Tn);
        if (bit) big_sub(R, Tn);
        q = (q << 1) | (bit ? 1 : 0);
    }
    if (big_nonzero(R)) sticky = true;
    const int top = (q >> 63) ? 63 : 62;
    const int e0 = nbD - nbT - (int)k - (63 - top);
    uint64_t frac;
    int biased;
    if (is64) {
        if (!round_format(q, top, sticky, e0, 53, -1022, 1023, frac, biased)) return false;
        out = (sign << 63) | ((uint64_t)biased << 52) | frac;
    } else {
        if (!round_format(q, top, sticky, e0, 24, -126, 127, frac, biased)) return false;
        out = (sign << 31) | ((uint64_t)biased << 23) | frac;
    }
    return true;
}

}  // namespace mdc
