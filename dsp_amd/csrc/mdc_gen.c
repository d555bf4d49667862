/*
 * mdc_gen.c -- deterministic synthetic surface-observation data in the on-disk format the
 * reference's producer writes (tools/dminingoracle.cpp:212-219, :243):
 *
 *   <data>\n
 *   <f1>v1</f1><f2>v2</f2>...<endl/>\n          one line per row
 *   </data>\n
 *
 * ZHOBTMIND rows carry obtid, ddatetime, t, p, u, wd, wf, r, vis, keyid (the select list of
 * tools/dminingoracle.cpp:356-357; upttime is never exported, so xmltodb's splitbuffer
 * carries the previous column into it).  ZHOBTCODE rows carry obtid, cityname, provname,
 * lat, lon, height (tools/dminingoracle.cpp:349-350).  The delimited form is the same MIND
 * row joined by a separator, one row per line.
 *
 * PRNG: splitmix64 seeded by the caller; row k of a stream depends only on (seed, k), so a
 * buffer can be produced in independent pieces.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

#define NSTATIONS 839

static char *put_str(char *p, const char *s) {
    size_t n = strlen(s);
    memcpy(p, s, n);
    return p + n;
}
static char *put_uint(char *p, uint64_t v, int mindigits) {
    char tmp[24];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n < mindigits) tmp[n++] = '0';
    while (n) *p++ = tmp[--n];
    return p;
}
/* value in tenths (may be negative) printed with one decimal */
static char *put_tenths(char *p, int64_t tenths) {
    if (tenths < 0) {
        *p++ = '-';
        tenths = -tenths;
    }
    p = put_uint(p, (uint64_t)(tenths / 10), 1);
    *p++ = '.';
    *p++ = (char)('0' + tenths % 10);
    return p;
}

static int days_in_month(int y, int m) {
    static const int d[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
    if (m == 2 && ((y % 4 == 0 && y % 100 != 0) || y % 400 == 0)) return 29;
    return d[m - 1];
}
/* 2024-05-19 16:28:00 plus `minutes`, as yyyymmddhh24miss */
static char *put_datetime(char *p, uint64_t minutes) {
    int y = 2024, mo = 5, d = 19;
    uint64_t tot = 16 * 60 + 28 + minutes;
    uint64_t days = tot / 1440;
    int rem = (int)(tot % 1440);
    while (days) {
        int dim = days_in_month(y, mo);
        if (d + (int64_t)days <= dim) {
            d += (int)days;
            days = 0;
        } else {
            days -= (uint64_t)(dim - d + 1);
            d = 1;
            if (++mo > 12) {
                mo = 1;
                ++y;
            }
        }
    }
    p = put_uint(p, (uint64_t)y, 4);
    p = put_uint(p, (uint64_t)mo, 2);
    p = put_uint(p, (uint64_t)d, 2);
    p = put_uint(p, (uint64_t)(rem / 60), 2);
    p = put_uint(p, (uint64_t)(rem % 60), 2);
    *p++ = '0';
    *p++ = '0';
    return p;
}

static uint32_t station_id(uint64_t seed, int k) {
    uint64_t s = seed ^ (0xC0DEull * (uint64_t)(k + 1));
    return 50000u + (uint32_t)(splitmix64(&s) % 10000u);
}

typedef struct {
    uint32_t obtid;
    uint64_t minutes;
    int64_t t, p, wf, r, vis; /* tenths */
    int u, wd;
    uint64_t keyid;
} mind_row;

static void mind_row_make(uint64_t seed, int64_t k, mind_row *o) {
    uint64_t s = seed + 0x51ED27ull * (uint64_t)(k + 1);
    o->obtid = station_id(seed, (int)(k % NSTATIONS));
    o->minutes = (uint64_t)(k / NSTATIONS);
    o->t = (int64_t)(splitmix64(&s) % 851) - 400;     /* -40.0 .. 45.0 */
    o->p = 8500 + (int64_t)(splitmix64(&s) % 2001);   /* 850.0 .. 1050.0 */
    o->u = (int)(splitmix64(&s) % 101);               /* 0 .. 100 */
    o->wd = (int)(splitmix64(&s) % 361);              /* 0 .. 360 */
    o->wf = (int64_t)(splitmix64(&s) % 401);          /* 0.0 .. 40.0 */
    o->r = (int64_t)(splitmix64(&s) % 501);           /* 0.0 .. 50.0 */
    o->vis = (int64_t)(splitmix64(&s) % 300001);      /* 0.0 .. 30000.0 */
    o->keyid = 1000000ull + (uint64_t)k;              /* increasing */
}

#define TAG(p, name, body) \
    do {                   \
        p = put_str(p, "<" name ">"); \
        body;              \
        p = put_str(p, "</" name ">"); \
    } while (0)

static char *mind_xml_row(char *p, const mind_row *r) {
    TAG(p, "obtid", p = put_uint(p, r->obtid, 5));
    TAG(p, "ddatetime", p = put_datetime(p, r->minutes));
    TAG(p, "t", p = put_tenths(p, r->t));
    TAG(p, "p", p = put_tenths(p, r->p));
    TAG(p, "u", p = put_uint(p, (uint64_t)r->u, 1));
    TAG(p, "wd", p = put_uint(p, (uint64_t)r->wd, 1));
    TAG(p, "wf", p = put_tenths(p, r->wf));
    TAG(p, "r", p = put_tenths(p, r->r));
    TAG(p, "vis", p = put_tenths(p, r->vis));
    TAG(p, "keyid", p = put_uint(p, r->keyid, 1));
    return put_str(p, "<endl/>\n");
}

static char *mind_csv_row(char *p, const mind_row *r, const char *sep) {
    p = put_uint(p, r->obtid, 5);
    p = put_str(p, sep);
    p = put_datetime(p, r->minutes);
    p = put_str(p, sep);
    p = put_tenths(p, r->t);
    p = put_str(p, sep);
    p = put_tenths(p, r->p);
    p = put_str(p, sep);
    p = put_uint(p, (uint64_t)r->u, 1);
    p = put_str(p, sep);
    p = put_uint(p, (uint64_t)r->wd, 1);
    p = put_str(p, sep);
    p = put_tenths(p, r->wf);
    p = put_str(p, sep);
    p = put_tenths(p, r->r);
    p = put_str(p, sep);
    p = put_tenths(p, r->vis);
    p = put_str(p, sep);
    p = put_uint(p, r->keyid, 1);
    *p++ = '\n';
    return p;
}

#define ROW_MAX 512

/* Fills buf (capacity cap) with "<data>\n" + as many MIND rows as fit (at most max_rows when
 * max_rows>0) starting at row first_row + "</data>\n".  Returns bytes written; *rows_out
 * receives the number of rows. */
size_t mdc_gen_mind_xml(uint8_t *buf, size_t cap, uint64_t seed, int64_t first_row,
                        int64_t max_rows, int64_t *rows_out) {
    char *p = (char *)buf, *end = (char *)buf + cap;
    int64_t k = 0;
    if (cap < 16) {
        if (rows_out) *rows_out = 0;
        return 0;
    }
    p = put_str(p, "<data>\n");
    while ((max_rows <= 0 || k < max_rows) && (size_t)(end - p) >= ROW_MAX + 8) {
        mind_row r;
        mind_row_make(seed, first_row + k, &r);
        p = mind_xml_row(p, &r);
        ++k;
    }
    p = put_str(p, "</data>\n");
    if (rows_out) *rows_out = k;
    return (size_t)(p - (char *)buf);
}

/* Same rows, one delimited line per row (no framing lines). */
size_t mdc_gen_mind_csv(uint8_t *buf, size_t cap, uint64_t seed, int64_t first_row,
                        int64_t max_rows, const char *sep, int64_t *rows_out) {
    char *p = (char *)buf, *end = (char *)buf + cap;
    int64_t k = 0;
    while ((max_rows <= 0 || k < max_rows) && (size_t)(end - p) >= ROW_MAX) {
        mind_row r;
        mind_row_make(seed, first_row + k, &r);
        p = mind_csv_row(p, &r, sep);
        ++k;
    }
    if (rows_out) *rows_out = k;
    return (size_t)(p - (char *)buf);
}

/* ZHOBTCODE station list: 2-4 CJK characters (UTF-8) for city and province names. */
static char *put_cjk(char *p, uint64_t *s, int nchars) {
    for (int i = 0; i < nchars; ++i) {
        uint32_t cp = 0x4E00u + (uint32_t)(splitmix64(s) % 0x5000u);
        *p++ = (char)(0xE0 | (cp >> 12));
        *p++ = (char)(0x80 | ((cp >> 6) & 0x3F));
        *p++ = (char)(0x80 | (cp & 0x3F));
    }
    return p;
}

size_t mdc_gen_code_xml(uint8_t *buf, size_t cap, uint64_t seed, int64_t max_rows,
                        int64_t *rows_out) {
    char *p = (char *)buf, *end = (char *)buf + cap;
    int64_t k = 0;
    if (cap < 16) {
        if (rows_out) *rows_out = 0;
        return 0;
    }
    if (max_rows <= 0 || max_rows > NSTATIONS) max_rows = NSTATIONS;
    p = put_str(p, "<data>\n");
    while (k < max_rows && (size_t)(end - p) >= ROW_MAX + 8) {
        uint64_t s = seed ^ (0xC17Full * (uint64_t)(k + 1));
        TAG(p, "obtid", p = put_uint(p, station_id(seed, (int)k), 5));
        TAG(p, "cityname", p = put_cjk(p, &s, 2 + (int)(splitmix64(&s) % 3)));
        TAG(p, "provname", p = put_cjk(p, &s, 2 + (int)(splitmix64(&s) % 3)));
        TAG(p, "lat", { int64_t v = 1800 + (int64_t)(splitmix64(&s) % 3501); p = put_uint(p, (uint64_t)(v / 100), 1); *p++ = '.'; p = put_uint(p, (uint64_t)(v % 100), 2); });
        TAG(p, "lon", { int64_t v = 7300 + (int64_t)(splitmix64(&s) % 6201); p = put_uint(p, (uint64_t)(v / 100), 1); *p++ = '.'; p = put_uint(p, (uint64_t)(v % 100), 2); });
        TAG(p, "height", p = put_tenths(p, (int64_t)(splitmix64(&s) % 45001)));
        p = put_str(p, "<endl/>\n");
        ++k;
    }
    p = put_str(p, "</data>\n");
    if (rows_out) *rows_out = k;
    return (size_t)(p - (char *)buf);
}
