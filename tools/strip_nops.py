#!/usr/bin/env python
"""Experiment tool: removes from a gfx950 assembly listing the `s_nop 0` the compiler puts between a packed (VOP3P)
instruction and the VALU instruction that reads its result.

LLVM's hazard recognizer takes the gfx940 "dst_sel forwarding" hazard (a VALU instruction that writes PART of a VGPR --
SDWA dst_sel, op_sel[3] -- needs one wait state before a VALU instruction that reads the register) for every VOP3P
instruction: op_sel_hi[0] of a packed instruction and the dst op_sel bit of a VOP3 instruction are the same bit of
src0_modifiers.  Packed instructions write whole dwords.  tools/ubench/valu_rate.hip compares the fill's recurrence with and
without the nops bit for bit; this script makes the A/B library for the whole kernel (tools/build_nonop_variant.sh).

usage: strip_nops.py in.s out.s   (prints how many were removed / kept)"""
import re
import sys


def main(src, dst):
    removed = kept = 0
    out = []
    prev = None  # mnemonic of the last real instruction
    for line in open(src):
        s = line.strip()
        is_ins = bool(re.match(r"^[a-z_0-9]+(\s|$)", s)) and not s.endswith(":")
        if is_ins and s.split()[0] == "s_nop":
            if s.split()[1] == "0" and prev is not None and prev.startswith("v_pk_"):
                removed += 1
                continue
            kept += 1
        if is_ins:
            prev = s.split()[0]
        out.append(line)
    open(dst, "w").writelines(out)
    print("s_nop removed %d kept %d" % (removed, kept))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
