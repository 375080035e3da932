"""A/B: pg_fragment_kernel with a 4 KB LDS counter block instead of 16 KB (its workgroups need LDS the fill's wavefronts hold)"""
import sys, os
p = os.path.join(sys.argv[1], "pg_count.hip")
s = open(p).read()
old = "constexpr uint32_t FRAG_LDS_COUNTERS = 4096;"
assert old in s
s = s.replace(old, "constexpr uint32_t FRAG_LDS_COUNTERS = 1024;")
open(p, "w").write(s)
