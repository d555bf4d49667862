#!/usr/bin/env python3
"""What a maintainer of the reference does to put libmdc_parse.so behind public/_public.h, as a script:
copies public/_public.h, _public.cpp and _cmpublic.h of a reference tree into a directory and inserts

    #ifdef MDC_PARSE_COMPAT
    #include "mdc_public_compat.h"
    #endif

in front of `namespace idc` of the header, and `#ifndef MDC_PARSE_COMPAT` / `#endif` around the hot-path
declarations of the header and their definitions in the .cpp (the trim and case helpers, picknumber,
ccmdstr, getxmlbuffer, cifile).  Compiled with -DMDC_PARSE_COMPAT the toolkit then takes those from the
library's header and keeps everything else of its own (cdir, clogfile, matchstr, sockets ...); compiled
without, it is the reference unchanged.

usage: patch_public_for_compat.py <reference root> <output directory>

Every range is checked against the text it is expected to begin and end with, so a reference of another
revision fails loudly instead of being guarded in the wrong places.  Nothing of the reference is kept in
this repository: the script holds line numbers and the first words of a few lines only.
"""
import os
import shutil
import sys

# (first line, last line, what the first line must start with, what the last line must start with), 1-based
HEADER_RANGES = [
    (16, 44, "/////", "string& tolower(string &str);"),                 # deletelchr ... tolower
    (65, 67, "char*     picknumber(", "string    picknumber("),         # picknumber x3
    (81, 125, "class ccmdstr", "ostream& operator<<(ostream& out, const ccmdstr"),
    (140, 148, "bool getxmlbuffer(", "bool getxmlbuffer("),            # getxmlbuffer x9
    (401, 430, "class cifile", "};"),
]
SOURCE_RANGES = [
    (11, 123, "char *deletelchr(", "}"),                                # trim and case helpers
    (167, 211, "char* picknumber(", "}"),
    (265, 459, "ccmdstr::ccmdstr(", "}"),                               # ccmdstr and its operator<<
    (461, 625, "bool getxmlbuffer(", "}"),
    (828, 892, "bool cifile::open(", "}"),
]
INCLUDE_BEFORE = "namespace idc"


def guard(lines, ranges, name):
    for first, last, head, tail in sorted(ranges, reverse=True):
        if not lines[first - 1].startswith(head) or not lines[last - 1].startswith(tail):
            raise SystemExit(f"{name}:{first}-{last}: expected {head!r} ... {tail!r}, found "
                             f"{lines[first - 1][:40]!r} ... {lines[last - 1][:40]!r}: another revision of the reference?")
        lines.insert(last, "#endif  // MDC_PARSE_COMPAT\n")
        lines.insert(first - 1, "#ifndef MDC_PARSE_COMPAT\n")
    return lines


def main():
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    ref, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    pub = os.path.join(ref, "public")
    shutil.copyfile(os.path.join(pub, "_cmpublic.h"), os.path.join(out, "_cmpublic.h"))
    h = guard(open(os.path.join(pub, "_public.h"), encoding="utf-8", errors="surrogateescape").readlines(), HEADER_RANGES, "_public.h")
    at = next(i for i, l in enumerate(h) if l.startswith(INCLUDE_BEFORE))
    h[at:at] = ["#ifdef MDC_PARSE_COMPAT\n", '#include "mdc_public_compat.h"\n', "#endif\n"]
    open(os.path.join(out, "_public.h"), "w", encoding="utf-8", errors="surrogateescape").writelines(h)
    c = guard(open(os.path.join(pub, "_public.cpp"), encoding="utf-8", errors="surrogateescape").readlines(), SOURCE_RANGES, "_public.cpp")
    open(os.path.join(out, "_public.cpp"), "w", encoding="utf-8", errors="surrogateescape").writelines(c)
    print("patched _public.h and _public.cpp in", out)


if __name__ == "__main__":
    main()
