"""timing only (results are wrong): what a FIFTH wavefront per SIMD would buy the fill kernel.
The kernel is held to 4 wavefronts per SIMD twice: 114 VGPRs and 10 240 B of LDS profile per wavefront.  This patch makes a lean
build of the byte variants -- no second seed-cache entry, no profile prefetch registers: 93 VGPRs at C = 10 -- whose LDS profile
holds TWO reference codes instead of four (`code & 1`: the same instructions, half the bytes, wrong scores), and lets the launch
pad its dynamic LDS (PG_X_FILL_LDS bytes per wavefront): 10 240 = four wavefronts per SIMD, 8 192 = five -- the same instruction
stream at both occupancies.  usage: tools/build_patched_variant.sh lean tools/experiments/fill_lean_occupancy.py [PG_X_KEEP=prefetch|cache2]"""
import os
import sys
p = os.path.join(sys.argv[1], "pg_fill.hip")
s = open(p).read()
keep = os.environ.get("PG_X_KEEP", "")


def rep(old, new, count=1):
    global s
    assert s.count(old) >= 1, old
    s = s.replace(old, new) if count == 0 else s.replace(old, new, count)


if "prefetch" not in keep:
    rep("constexpr bool PREFETCH = WIDE ? C <= 16 : true;", "constexpr bool PREFETCH = WIDE ? C <= 16 : false;")
if "cache2" not in keep:
    rep("constexpr bool SEEDCACHE2 = SEEDCACHE && (WIDE ? C <= 8 : C <= 12);", "constexpr bool SEEDCACHE2 = SEEDCACHE && WIDE && C <= 8;")
if "fullprofile" not in keep:
    # two codes in LDS: [GROUPS][2][ROWS]
    rep("prof[(g * 4 + code) * ROWS + row] = pk_delta(sA, sB);", "if (code < 2) prof[(g * 2 + code) * ROWS + row] = pk_delta(sA, sB);")
    rep("const uint32_t* profl = prof + lgrp * 4 * ROWS + k * C;", "const uint32_t* profl = prof + lgrp * 2 * ROWS + k * C;")
    rep("const uint32_t* pr = profl + (code & 3u) * ROWS;", "const uint32_t* pr = profl + (code & 1u) * ROWS;")
    rep("const uint32_t* pr = profl + (PG_META_CODE(meta_rows) & 3u) * ROWS;", "const uint32_t* pr = profl + (PG_META_CODE(meta_rows) & 1u) * ROWS;")
rep("    const size_t lds = (size_t)(64 * 4 * C) * sizeof(uint32_t);  // [64 / GL reads][4 codes][GL * C rows]",
    "    size_t lds = (size_t)(64 * 4 * C) * sizeof(uint32_t);\n    if (const char* e = getenv(\"PG_X_FILL_LDS\")) lds = (size_t)atol(e);")
rep("#include <stdint.h>\n", "#include <stdint.h>\n#include <stdlib.h>\n")
open(p, "w").write(s)

# ---- PG_X_PARTIAL=<rows>: only the first <rows> profile rows of the NEXT column are fetched a step ahead (ping-pong registers);
# the others are fetched at the start of the step that uses them (their latency hides under the first rows' arithmetic)
PR = int(os.environ.get("PG_X_PARTIAL", "0"))
if PR:
    s = open(p).read()
    rep("constexpr bool PREFETCH = WIDE ? C <= 16 : false;", "constexpr bool PREFETCH = WIDE ? C <= 16 : true;\n    constexpr int PRW = WIDE ? C : %d;" % PR)
    mask = "1u" if "fullprofile" not in keep else "3u"
    old = '''        const uint32_t meta_rows = PREFETCH ? meta : meta_cur;
        uint32_t (&rows)[C] = PREFETCH ? sn : sc;
        {
            const uint32_t* pr = profl + (PG_META_CODE(meta_rows) & %s) * ROWS;
#pragma unroll
            for (int r = 0; r < C; r += 2)
            {
                const uint2 v = *(const uint2*)(pr + r);
                rows[r] = v.x;
                rows[r + 1] = v.y;
            }
        }''' % mask
    new = '''        const uint32_t meta_rows = PREFETCH ? meta : meta_cur;
        uint32_t (&rows)[C] = PREFETCH ? sn : sc;
        {
            const uint32_t* prc = prcur;
            const uint32_t* pr = profl + (PG_META_CODE(meta_rows) & %s) * ROWS;
            prcur = pr;
#pragma unroll
            for (int r = 0; r < PRW; r += 2)
            {
                const uint2 v = *(const uint2*)(pr + r);
                rows[r] = v.x;
                rows[r + 1] = v.y;
            }
#pragma unroll
            for (int r = PRW; r < C; r += 2)
            {
                const uint2 v = *(const uint2*)(prc + r);
                sc[r] = v.x;
                sc[r + 1] = v.y;
            }
        }''' % mask
    rep(old, new)
    # the address of the column about to be computed, carried from step to step
    rep("    uint32_t sA[C], sB[C];\n    {\n        const uint32_t code = PG_META_CODE(meta);\n        const uint32_t* pr = profl + (code & %s) * ROWS;" % mask,
        "    uint32_t sA[C], sB[C];\n    const uint32_t* prcur;\n    {\n        const uint32_t code = PG_META_CODE(meta);\n        const uint32_t* pr = profl + (code & %s) * ROWS;\n        prcur = pr;" % mask)
    # (timing only: the code-4 synthesis of the late rows is left out)
    open(p, "w").write(s)
