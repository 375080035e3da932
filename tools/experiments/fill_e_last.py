"""Timing-only experiment (tools/build_patched_variant.sh): the E updates of a column after its F chain, as one run of ten
packed instructions, instead of one per row between the chain's instructions."""
import sys, os
p = os.path.join(sys.argv[1], "pg_fill.hip")
s = open(p).read()
old = """            const uint32_t tt = h + NEG5;                                    // (h - gap open) in the next step's / next row's terms
            E[r] = pk_max3h_s(E[r], tt, floorE);                             // no decrement: the frame moves instead
            F = pk_maxu(f, tt);                                              // "F + 1" of the next row
        }
        Fsend = F;
"""
new = """            const uint32_t tt = h + NEG5;                                    // (h - gap open) in the next step's / next row's terms
            ttv[r] = tt;
            F = pk_maxu(f, tt);                                              // "F + 1" of the next row
        }
        Fsend = F;
#pragma unroll
        for (int r = 0; r < C; ++r)
        {
            asm volatile("" : "+v"(ttv[r]));
            E[r] = pk_max3h_s(E[r], ttv[r], floorE);
        }
"""
assert old in s
s = s.replace(old, new)
old2 = """        uint32_t diag = dH;
        uint32_t F = Fabove;
#pragma unroll
        for (int r = 0; r < C; ++r)
        {
            // the three additions"""
new2 = """        uint32_t diag = dH;
        uint32_t F = Fabove;
        uint32_t ttv[C];
#pragma unroll
        for (int r = 0; r < C; ++r)
        {
            // the three additions"""
assert old2 in s
s = s.replace(old2, new2)
open(p, "w").write(s)
