// cc_csv.h — host-side text formatting for the per-point output file (no GPU work).
//
// chronoclust/app.py:263-360 ends in DataFrame.to_csv(index=False): one line per point, "id,cluster_id,v0,...,vd-1",
// every float64 written as its shortest round-trip decimal string in Python's repr() layout.  At 10^6 points x 20
// features that is 2 x 10^7 conversions; pandas (and the csv module over Python floats) spend 15 - 60 s on them.  The
// digits come from std::to_chars (shortest round-trip, scientific), the layout rules are CPython's float_repr_style
// 'short' (Python/pystrtod.c format_float_short, mode 'r'): exponent notation when the decimal point would sit more than
// 16 digits to the right or 4 or more to the left of the first digit, "e-05" style exponents with at least two digits,
// ".0" appended to integral values.
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace cc {

// writes repr(v) at out (at most 32 bytes), returns the number of bytes
inline int format_repr(double v, char* out)
{
    if (std::isnan(v)) { memcpy(out, "nan", 3); return 3; }
    char* o = out;
    if (std::signbit(v)) { *o++ = '-'; v = -v; }
    if (std::isinf(v)) { memcpy(o, "inf", 3); return (int)(o + 3 - out); }
    if (v == 0.0) { memcpy(o, "0.0", 3); return (int)(o + 3 - out); }
    char buf[40];
    const std::to_chars_result r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
    // buf = d[.ddd]e[+-]XX[X]
    char digits[24];
    int nd = 0;
    const char* q = buf;
    digits[nd++] = *q++;
    if (*q == '.') {
        ++q;
        while (*q != 'e') digits[nd++] = *q++;
    }
    ++q;  // 'e'
    const bool eneg = *q == '-';
    ++q;
    int e10 = 0;
    while (q < r.ptr) e10 = e10 * 10 + (*q++ - '0');
    if (eneg) e10 = -e10;
    const int decpt = e10 + 1;  // position of the decimal point relative to the first digit
    if (decpt > 16 || decpt < -3) {
        *o++ = digits[0];
        if (nd > 1) {
            *o++ = '.';
            memcpy(o, digits + 1, (size_t)nd - 1);
            o += nd - 1;
        }
        *o++ = 'e';
        int e = decpt - 1;
        *o++ = e < 0 ? '-' : '+';
        if (e < 0) e = -e;
        if (e >= 100) { *o++ = (char)('0' + e / 100); e %= 100; *o++ = (char)('0' + e / 10); *o++ = (char)('0' + e % 10); }
        else { *o++ = (char)('0' + e / 10); *o++ = (char)('0' + e % 10); }
    } else if (decpt <= 0) {
        *o++ = '0';
        *o++ = '.';
        for (int i = 0; i < -decpt; ++i) *o++ = '0';
        memcpy(o, digits, (size_t)nd);
        o += nd;
    } else if (decpt >= nd) {
        memcpy(o, digits, (size_t)nd);
        o += nd;
        for (int i = 0; i < decpt - nd; ++i) *o++ = '0';
        *o++ = '.';
        *o++ = '0';
    } else {
        memcpy(o, digits, (size_t)decpt);
        o += decpt;
        *o++ = '.';
        memcpy(o, digits + decpt, (size_t)(nd - decpt));
        o += nd - decpt;
    }
    return (int)(o - out);
}

inline int format_i64(long long v, char* out)
{
    const std::to_chars_result r = std::to_chars(out, out + 24, v);
    return (int)(r.ptr - out);
}

}  // namespace cc
