"""timing only: no H-trace stores, no packing, no first-column tracking of the node maximum"""
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
old = '''        if (DIR == 0 || WIDE)
        {
            // mask = 0xFFFF in the halves whose maximum grew'''
assert old in s
s = s.replace(old, '''        if (false)
        {
            // mask = 0xFFFF in the halves whose maximum grew''')
open(p, "w").write(s)
