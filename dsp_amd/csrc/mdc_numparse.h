// mdc_numparse.h -- exact text-to-number conversion on the device.
//
// Reproduces what the reference's typed overloads compute (public/_public.cpp:506-625 and
// :339-446): picknumber() keeps digits (and '+'/'-', and '.') of the field, then
// std::stoi/stol/stoul/stod/stof parse the longest valid prefix of that filtered string and
// throw (-> "false") when there is no digit or the C library reports ERANGE.  After the
// filter the string can hold no blank, exponent, hex prefix, "inf" or "nan", so the grammar
// is  [+-]? digits [. digits]  (integers: no dot), cut at the first character that does not
// fit (a second sign, a second dot).
//
// Decimal -> binary is correctly rounded (round-to-nearest-even, as glibc strtod/strtof):
// the decimal is held exactly as M / 10^k with M < 10^38 in 128 bits, and the quotient is
// produced bit by bit with a restoring division, so the rounding decision sees the exact
// remainder.  Inputs with more than 38 significant digits or more than 38 fractional digits
// are outside that exact range and are reported (ok = false, range = true), never guessed.
#pragma once
#include <stdint.h>
#include "mdc_device.h"

namespace mdc {

// values of MDC_T_* (include/mdc_parse.h)
enum : uint8_t { T_NONE = 0, T_I32 = 1, T_U32 = 2, T_I64 = 3, T_U64 = 4, T_F64 = 5, T_F32 = 6, T_BOOL = 7, T_TIME = 8 };

// strtotime (reference public/_public.cpp:1009-1044): the digits of the text must be exactly
// fourteen, yyyymmddhh24miss; the fields go to mktime as they are (month - 1, isdst 0), and mktime
// takes fields out of their ranges arithmetically (month 00 is December of the year before, day
// 00 the last day of the month before, hour 99 four days on).  With isdst 0 mktime reads the text
// as the zone's standard time whether or not daylight saving is in force then, so for a zone whose
// standard offset is fixed
// that is plain calendar arithmetic minus the zone's offset (glibc __mktime_internal: year and
// month folded, days before the month by leap-year table, ydhms difference to 1970).
// tz_offset = seconds east of UTC.  false when the text does not hold fourteen digits.
__device__ inline bool time_convert(const uint8_t *s, uint32_t n, int32_t tz_offset, uint64_t &value) {
    uint32_t d[14], nd = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t c = (uint32_t)s[i] - '0';
        if (c < 10u) {
            if (nd < 14) d[nd] = c;
            ++nd;
        }
    }
    if (nd != 14) return false;
    const int64_t year = d[0] * 1000 + d[1] * 100 + d[2] * 10 + d[3];
    const int mon = (int)(d[4] * 10 + d[5]) - 1;  // -1 .. 98
    const int64_t mday = d[6] * 10 + d[7], hour = d[8] * 10 + d[9], min = d[10] * 10 + d[11], sec = d[12] * 10 + d[13];
    int mon_rem = mon % 12, mon_years = mon / 12;
    if (mon_rem < 0) {
        mon_rem += 12;
        --mon_years;
    }
    const int64_t y = year + mon_years;
    const bool leap = (y % 4 == 0) && (y % 100 != 0 || y % 400 == 0);
    const int before[12] = {0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334};
    const int64_t yday = before[mon_rem] + ((leap && mon_rem > 1) ? 1 : 0) + mday - 1;
    // days from 1970-01-01 to January 1st of year y (y >= 0 here; floor divisions of y - 1 >= -1)
    const int64_t y1 = y - 1;
    auto fdiv = [](int64_t a, int64_t b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
    const int64_t days = 365 * (y - 1970) + (fdiv(y1, 4) - fdiv(1969, 4)) - (fdiv(y1, 100) - fdiv(1969, 100)) +
                         (fdiv(y1, 400) - fdiv(1969, 400)) + yday;
    const int64_t t = ((days * 24 + hour) * 60 + min) * 60 + sec - tz_offset;
    value = (uint64_t)t;
    return true;
}

struct u128 {
    uint64_t hi, lo;
};

__device__ __forceinline__ bool ge128(u128 a, u128 b) { return a.hi > b.hi || (a.hi == b.hi && a.lo >= b.lo); }
__device__ __forceinline__ u128 sub128(u128 a, u128 b) {
    u128 r;
    r.lo = a.lo - b.lo;
    r.hi = a.hi - b.hi - (a.lo < b.lo ? 1 : 0);
    return r;
}
__device__ __forceinline__ u128 shl128(u128 a, int s) {  // 0 <= s < 128
    u128 r;
    if (s == 0) return a;
    if (s >= 64) {
        r.hi = a.lo << (s - 64);
        r.lo = 0;
    } else {
        r.hi = (a.hi << s) | (a.lo >> (64 - s));
        r.lo = a.lo << s;
    }
    return r;
}
__device__ __forceinline__ int bitlen128(u128 a) {
    return a.hi ? 128 - __clzll((long long)a.hi) : (a.lo ? 64 - __clzll((long long)a.lo) : 0);
}

// 10^0 .. 10^38 as 128-bit integers
__device__ const uint64_t P10_128[39][2] = {
    {0x0ull, 0x1ull}, {0x0ull, 0xaull}, {0x0ull, 0x64ull}, {0x0ull, 0x3e8ull}, {0x0ull, 0x2710ull},
    {0x0ull, 0x186a0ull}, {0x0ull, 0xf4240ull}, {0x0ull, 0x989680ull}, {0x0ull, 0x5f5e100ull},
    {0x0ull, 0x3b9aca00ull}, {0x0ull, 0x2540be400ull}, {0x0ull, 0x174876e800ull},
    {0x0ull, 0xe8d4a51000ull}, {0x0ull, 0x9184e72a000ull}, {0x0ull, 0x5af3107a4000ull},
    {0x0ull, 0x38d7ea4c68000ull}, {0x0ull, 0x2386f26fc10000ull}, {0x0ull, 0x16345785d8a0000ull},
    {0x0ull, 0xde0b6b3a7640000ull}, {0x0ull, 0x8ac7230489e80000ull},
    {0x5ull, 0x6bc75e2d63100000ull}, {0x36ull, 0x35c9adc5dea00000ull},
    {0x21eull, 0x19e0c9bab2400000ull}, {0x152dull, 0x02c7e14af6800000ull},
    {0xd3c2ull, 0x1bcecceda1000000ull}, {0x84595ull, 0x161401484a000000ull},
    {0x52b7d2ull, 0xdcc80cd2e4000000ull}, {0x33b2e3cull, 0x9fd0803ce8000000ull},
    {0x204fce5eull, 0x3e25026110000000ull}, {0x1431e0faeull, 0x6d7217caa0000000ull},
    {0xc9f2c9cd0ull, 0x4674edea40000000ull}, {0x7e37be2022ull, 0xc0914b2680000000ull},
    {0x4ee2d6d415bull, 0x85acef8100000000ull}, {0x314dc6448d93ull, 0x38c15b0a00000000ull},
    {0x1ed09bead87c0ull, 0x378d8e6400000000ull}, {0x13426172c74d82ull, 0x2b878fe800000000ull},
    {0xc097ce7bc90715ull, 0xb34b9f1000000000ull}, {0x785ee10d5da46d9ull, 0x00f436a000000000ull},
    {0x4b3b4ca85a86c47aull, 0x098a224000000000ull}};

__device__ const double P10_F64[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                       1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// Running state of one conversion; fed one filtered byte at a time.
struct NumAcc {
    uint64_t m;    // integers: the value; floats: significant digits 1..19
    uint64_t m2;   // floats: significant digits 20..38
    uint32_t nd;   // floats: significant digits taken (39 = too many)
    uint32_t n2;   // floats: digits in m2
    uint32_t k;    // floats: fractional digits taken into m/m2
    uint32_t pz;   // floats: fractional zeros seen but not yet taken
    uint32_t state;  // 0 start, 1 after sign, 2 integer digits, 3 after the dot, 4 stopped
    bool neg, any, ovf;
};

__device__ __forceinline__ void num_init(NumAcc &a) {
    a.m = a.m2 = 0;
    a.nd = a.n2 = a.k = a.pz = 0;
    a.state = 0;
    a.neg = a.any = a.ovf = false;
}

__device__ __forceinline__ void num_push_sig(NumAcc &a, uint32_t d) {
    if (a.nd == 0 && d == 0) return;  // leading zero
    if (a.nd < 19) {
        a.m = a.m * 10 + d;
        ++a.nd;
    } else if (a.nd < 38) {
        a.m2 = a.m2 * 10 + d;
        ++a.n2;
        ++a.nd;
    } else {
        a.nd = 39;
    }
}

// b is a byte that survived picknumber for this type: a digit, or '+'/'-' (signed types),
// or '.' (floating types).
__device__ __forceinline__ void num_feed(NumAcc &a, uint32_t b, bool isfloat) {
    if (a.state == 4) return;
    if (b == '+' || b == '-') {
        if (a.state == 0) {
            a.neg = (b == '-');
            a.state = 1;
        } else {
            a.state = 4;
        }
        return;
    }
    if (b == '.') {
        a.state = (a.state <= 2) ? 3 : 4;
        return;
    }
    uint32_t d = b - '0';
    a.any = true;
    if (!isfloat) {
        a.state = 2;
        if (a.m > (0xFFFFFFFFFFFFFFFFull - d) / 10)
            a.ovf = true;
        else
            a.m = a.m * 10 + d;
        return;
    }
    if (a.state == 3) {
        if (d == 0) {
            ++a.pz;
            return;
        }
        while (a.pz) {  // zeros between the dot and this digit become significant
            num_push_sig(a, 0);
            if (a.k < 0x7fffffffu) ++a.k;
            --a.pz;
        }
        num_push_sig(a, d);
        ++a.k;
    } else {
        a.state = 2;
        num_push_sig(a, d);
    }
}

// Correctly rounded M / 10^k to a binary float with `pbits` significand bits.  Returns the
// unbiased exponent e and significand mant (pbits bits, top bit set): value = mant * 2^(e-pbits+1).
__device__ inline void div_round(u128 M, uint32_t k, int pbits, int &e, uint64_t &mant) {
    u128 D;
    D.hi = P10_128[k][0];
    D.lo = P10_128[k][1];
    int nb = bitlen128(M), db = bitlen128(D);
    u128 R = shl128(M, 128 - nb);
    u128 Dn = shl128(D, 128 - db);
    uint64_t q = 0;
    for (int i = 0; i < 57; ++i) {
        bool carry = false;
        if (i > 0) {
            carry = (R.hi >> 63) != 0;
            R = shl128(R, 1);
        }
        bool bit = carry || ge128(R, Dn);
        if (bit) R = sub128(R, Dn);
        q = (q << 1) | (bit ? 1 : 0);
    }
    bool sticky = (R.hi | R.lo) != 0;
    int top = (q >> 56) ? 56 : 55;  // index of the leading one
    e = nb - db - (56 - top);
    int sh = top + 1 - pbits;
    mant = q >> sh;
    uint64_t rem = q & ((1ull << sh) - 1);
    uint64_t half = 1ull << (sh - 1);
    if (rem > half || (rem == half && (sticky || (mant & 1)))) {
        ++mant;
        if (mant >> pbits) {
            mant >>= 1;
            ++e;
        }
    }
}

// Finishes a conversion.  Returns true and the 8-byte slot value when the reference overload
// would return true.  `range` is raised when the text is beyond the exact range handled here.
__device__ inline bool num_finish(const NumAcc &a, uint32_t type, uint64_t &out, bool &range) {
    if (!a.any) return false;
    switch (type) {
    case T_I32:
        if (a.ovf) return false;
        if (a.neg) {
            if (a.m > 0x80000000ull) return false;
            out = (uint64_t)(-(int64_t)a.m);
        } else {
            if (a.m > 0x7fffffffull) return false;
            out = a.m;
        }
        return true;
    case T_U32:
        if (a.ovf || a.m > 0x7fffffffull) return false;  // the reference parses it with stoi
        out = a.m;
        return true;
    case T_I64:
        if (a.ovf) return false;
        if (a.neg) {
            if (a.m > 0x8000000000000000ull) return false;
            out = (uint64_t)0 - a.m;
        } else {
            if (a.m > 0x7fffffffffffffffull) return false;
            out = a.m;
        }
        return true;
    case T_U64:
        if (a.ovf) return false;
        out = a.m;
        return true;
    default:
        break;
    }
    // floating point
    if (a.nd > 38 || a.k > 38) {
        range = true;
        return false;
    }
    u128 M;
    if (a.n2) {
        // m * 10^n2 + m2 ; n2 <= 19 so 10^n2 fits 64 bits
        uint64_t p = P10_128[a.n2][1];
        M.lo = a.m * p;
        M.hi = __umul64hi(a.m, p);
        uint64_t lo2 = M.lo + a.m2;
        if (lo2 < M.lo) ++M.hi;
        M.lo = lo2;
    } else {
        M.hi = 0;
        M.lo = a.m;
    }
    uint64_t sign = a.neg ? 1ull : 0ull;
    if ((M.hi | M.lo) == 0) {
        out = (type == T_F64) ? (sign << 63) : (sign << 31);
        return true;
    }
    if (type == T_F64) {
        if (M.hi == 0 && M.lo < (1ull << 53) && a.k <= 22) {
            double v = (double)M.lo / P10_F64[a.k];  // both exact: one correctly rounded division
            if (a.neg) v = -v;
            out = (uint64_t)__double_as_longlong(v);
            return true;
        }
        int e;
        uint64_t mant;
        div_round(M, a.k, 53, e, mant);
        out = (sign << 63) | ((uint64_t)(e + 1023) << 52) | (mant & ((1ull << 52) - 1));
        return true;
    }
    // binary32: round once, directly from the decimal (as strtof does)
    int e;
    uint64_t mant;
    div_round(M, a.k, 24, e, mant);
    if (e + 127 <= 0) return false;   // subnormal and inexact: strtof sets ERANGE, stof throws
    if (e + 127 >= 255) return false; // overflow: ERANGE
    out = (sign << 31) | ((uint64_t)(e + 127) << 23) | (mant & ((1ull << 23) - 1));
    return true;
}

}  // namespace mdc
