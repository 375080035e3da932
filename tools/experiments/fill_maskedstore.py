"""timing only: the H-trace stores only in lanes whose node maximum grew in this step (exec-masked store instructions)"""
import sys, os
p = os.path.join(sys.argv[1], "pg_fill.hip")
s = open(p).read()
old = '''        if (DIR == 0 || WIDE)
        {
            // mask = 0xFFFF in the halves whose maximum grew: (Mprev - Mn) is negative there as a 16-bit integer
            uint32_t grew;'''
assert old in s
s = s.replace(old, '''        uint32_t grew = 0;
        if (DIR == 0 || WIDE)
        {
            // mask = 0xFFFF in the halves whose maximum grew: (Mprev - Mn) is negative there as a 16-bit integer''')
old = '''        if (DIR == 0)
        {
            // one byte per cell:'''
assert old in s
s = s.replace(old, '''        if (DIR == 0 && grew != 0u)
        {
            // one byte per cell:''')
open(p, "w").write(s)
