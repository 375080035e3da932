"""timing only: neither the H-trace stores nor the byte packing in front of them"""
import sys, os
p = os.path.join(sys.argv[1], "pg_fill.hip")
s = open(p).read()
old = '''        if (DIR == 0)
        {
            // one byte per cell:'''
assert old in s
s = s.replace(old, '''        if (false)
        {
            // one byte per cell:''')
open(p, "w").write(s)
