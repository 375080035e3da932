"""Summarises a tools/e2e/prof.hh sample file: python tools/e2e/prof_report.py gpurun_out/o/prof.txt.gz [n]"""
import bisect
import collections
import gzip
import re
import subprocess
import sys


def symtab(path):
    out = subprocess.run(["nm", "-C", "--defined-only", "-n", path], capture_output=True, text=True).stdout
    tab = []
    for line in out.splitlines():
        m = re.match(r"([0-9a-f]+) [tTwW] (.*)", line)
        if m:
            tab.append((int(m.group(1), 16), m.group(2)))
    tab.sort()
    return tab


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 45
    opener = gzip.open if path.endswith(".gz") else open
    lines = [l.rstrip("\n") for l in opener(path, "rt")]
    tabs = {"libparagraph_host.so": symtab("paragraph_amd/libparagraph_host.so"), "libparagraph_amd.so": symtab("paragraph_amd/libparagraph_amd.so")}

    def resolve(fr):
        m = re.match(r"(\S+)\+0x([0-9a-f]+) (.*)", fr)
        if not m:
            return fr
        lib, off, sym = m.group(1), int(m.group(2), 16), m.group(3)
        if lib in tabs:
            t = tabs[lib]
            i = bisect.bisect_right(t, (off, "￿")) - 1
            if i >= 0:
                return lib.split(".")[0][3:] + ":" + t[i][1][:100]
        if sym != "?":
            return lib.split(".")[0] + ":" + sym[:80]
        return lib
    leaf, incl = collections.Counter(), collections.Counter()
    n = 0
    for l in lines:
        frs = [f for f in l.split(";") if f]
        if not frs:
            continue
        n += 1
        r = [resolve(f) for f in frs]
        leaf[r[0]] += 1
        for x in set(r):
            incl[x] += 1
    print("samples", n)
    print("inclusive (a frame anywhere in the sample's top 8)")
    for k, v in incl.most_common(top):
        print("%5.1f%%  %s" % (100 * v / n, k))
    print("\nleaf")
    for k, v in leaf.most_common(30):
        print("%5.1f%%  %s" % (100 * v / n, k))
    # where the library-side time belongs: the innermost frame of this repo's own code that is not a container / string / Json
    # helper (so malloc under Json::operator[] under countDocument counts for countDocument)
    generic = re.compile(r":(std::|common::Json|void std::|__gnu_cxx|operator )")
    own = collections.Counter()
    for l in lines:
        frs = [f for f in l.split(";") if f]
        if not frs:
            continue
        r = [resolve(f) for f in frs]
        pick = next((x for x in r if x.startswith(("paragraph_host:", "paragraph_amd:", "grmpy_batch:")) and not generic.search(x)), None)
        own[pick or ("(" + r[0].split(":")[0] + ")")] += 1
    print("\ninnermost own function (helpers skipped)")
    for k, v in own.most_common(40):
        print("%5.1f%%  %s" % (100 * v / n, k))


if __name__ == "__main__":
    main()
