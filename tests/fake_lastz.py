#!/usr/bin/env python3
"""A stand-in for the external lastz executable in the anchor-generation tests (tests/test_anchor_generation.py): takes
lastz's command line (options, then the two FASTA files the host library wrote: target "a", query "b") and prints
exonerate CIGAR lines.  FAKE_LASTZ_LINES (if set) is printed as it is; otherwise one gap-free line per common block of
at least 12 characters (difflib), upper-case only when the file holds upper case -- so that the test can see whether the
library's masked and un-masked passes sent what the reference sends."""
import difflib
import os
import sys


def fasta(path):
    lines = open(path).read().split("\n")
    return lines[0][1:], "".join(lines[1:])


files = [a for a in sys.argv[1:] if not a.startswith("--")]
(na, a), (nb, b) = fasta(files[0]), fasta(files[1])
log = os.environ.get("FAKE_LASTZ_LOG")
if log:
    open(log, "a").write("%s\t%d\t%d\t%d\t%s\n" % (" ".join(x for x in sys.argv[1:] if x.startswith("--")), len(a), len(b),
                                                 int(a.isupper()), na + nb))
if os.environ.get("FAKE_LASTZ_LINES"):
    sys.stdout.write(os.environ["FAKE_LASTZ_LINES"])
    sys.exit(0)
for blk in difflib.SequenceMatcher(None, a, b, autojunk=False).get_matching_blocks():
    if blk.size >= 12:
        print("cigar: b %d %d + a %d %d + %d M %d" % (blk.b, blk.b + blk.size, blk.a, blk.a + blk.size, blk.size * 100,
                                                      blk.size))
