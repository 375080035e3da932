"""Experiment (tools/build_patched_variant.sh): last_column's seed-cache selects (2 x C v_cndmask per last column, executed for every
node although only nodes with a far successor -- PG_META_SAVE -- are ever cached) behind a branch on `keep`, which is uniform over
the lanes that are on a last column in one step (they sit on the same column of the same node)."""
import sys, os
p = os.path.join(sys.argv[1], "pg_fill.hip")
s = open(p).read()
old = """                else
                {
                    const uint32_t w = __builtin_amdgcn_perm(es, hs, 0x06040200u);
                    sp[r] = w;
                    if constexpr (SEEDCACHE2)
                        cseedB[r] = toB ? w : cseedB[r];
                    if constexpr (SEEDCACHE)
                        cseed[r] = toA ? w : cseed[r];
                }
            }
            cnodeB = toB ? node : cnodeB;
            cnode = toA ? node : cnode;"""
new = """                else
                {
                    const uint32_t w = __builtin_amdgcn_perm(es, hs, 0x06040200u);
                    sp[r] = w;
                    wv[r] = w;
                }
            }
            if constexpr (!WIDE)
            {
                if (keep)
                {
#pragma unroll
                    for (int r = 0; r < C; ++r)
                    {
                        if constexpr (SEEDCACHE2)
                            cseedB[r] = toB ? wv[r] : cseedB[r];
                        if constexpr (SEEDCACHE)
                            cseed[r] = toA ? wv[r] : cseed[r];
                    }
                }
            }
            cnodeB = toB ? node : cnodeB;
            cnode = toA ? node : cnode;"""
assert old in s
s = s.replace(old, new)
old2 = """            const bool toB = keep && SEEDCACHE2 && (node & 1u) != 0u, toA = keep && !toB;
#pragma unroll
            for (int r = 0; r < C; ++r)
            {
                const uint32_t hs = pk_sub(Hout[r], hshift), es = pk_sub(E[r], eshift);  // 0x6400 | score"""
new2 = """            const bool toB = keep && SEEDCACHE2 && (node & 1u) != 0u, toA = keep && !toB;
            uint32_t wv[C];
#pragma unroll
            for (int r = 0; r < C; ++r)
            {
                const uint32_t hs = pk_sub(Hout[r], hshift), es = pk_sub(E[r], eshift);  // 0x6400 | score"""
assert old2 in s
s = s.replace(old2, new2)
open(p, "w").write(s)
