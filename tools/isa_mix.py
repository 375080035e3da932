#!/usr/bin/env python
"""Static instruction mix of the fill kernel's step loop (no GPU needed): compiles pg_fill.hip to gfx950 assembly, walks the
COMMON path of one pipeline step of each graph direction (every rare-path guard not taken) and counts the VALU instructions
by issue class -- gfx950 issues v_pk_* / VOP3 / integer min-max / DPP / anything with an SGPR operand over four cycles per
wave64 and plain 32-bit add / sub / logic / mov / fp32 add-mul-fma over two (profiles/r04_valu_rate.json).

usage: tools/isa_mix.py [C=10] [out.json]      -> profiles/rNN_fill_isa_mix.json (carries the kernel_source_sha it was made at)

The common path: the shortest way from a loop's header back to it (every conditional branch may go either way; the rare
paths -- node boundary, code-4 column, seed load, frame renormalisation -- only ever add instructions).  The step loop is
unrolled twice (ping-pong registers): the counts are per step = per loop iteration / 2."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TWO_CYCLE = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32",
             "v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_add_f16", "v_max_f16", "v_add_u16", "v_max_u16", "v_sub_u16",
             "v_not_b32"}


def issue_class(ins):
    """'valu4' / 'valu2' / other kinds, by the measured table (an e64 encoding, a DPP form or an SGPR source makes a two-cycle
    opcode a four-cycle one; literals and inline constants do not)"""
    m = ins.split()[0]
    if m.startswith("v_"):
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", m)
        if m.endswith("_dpp") or m.endswith("_e64") or m.endswith("_sdwa") or "row_shr" in ins or "row_bcast" in ins:
            return "valu4"
        ops = ins[len(m):]
        srcs = ops.split(",")[1:]
        if any(re.match(r"\s*(s\d+|s\[\d+:\d+\]|vcc|exec|m0)\b", x) for x in srcs):
            return "valu4"
        return "valu2" if base in TWO_CYCLE else "valu4"
    if m.startswith("s_"):
        if m.startswith(("s_load", "s_buffer_load")):
            return "smem"
        if m.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_barrier", "s_endpgm", "s_sleep")):
            return "ctl"
        return "salu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def kernel_asm(C):
    with tempfile.TemporaryDirectory() as t:
        out = os.path.join(t, "fill.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-o", out, os.path.join(ROOT, "paragraph_amd", "csrc", "pg_fill.hip")], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    name = "_Z14pg_fill_kernelILi%dELb0ELi16EEv10PgFillArgs:" % C
    start = next(i for i, l in enumerate(lines) if l.startswith(name))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def parse(lines):
    """-> instructions [(text)], label -> index of the first instruction at or after it, headers of the loops that hold
    inner loops (the compiler's own "This Loop Header" comments: the step loops; their inner loops are rare paths)"""
    ins, labels, outer = [], {}, []
    for l in lines:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("//"):
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", s)
        if m:
            labels[m.group(1)] = len(ins)
            if "=>This Loop Header" in s:
                outer.append(len(ins))
            continue
        if s.startswith(".") or s.endswith(":"):
            continue
        ins.append(s.split(";")[0].strip())
    return ins, labels, outer


def common_path(ins, labels, header):
    """the SHORTEST way round the loop from `header` back to it: every conditional branch may go either way, and the common
    path is the one with the fewest instructions (a rare path only ever adds work: the guarded blocks of node boundaries,
    code-4 columns, seed loads, the frame renormalisation every 256 steps; the loop's exit never comes back).  Dijkstra over
    the instruction graph, cost 1 per instruction."""
    import heapq
    cond = ("s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz", "s_cbranch_execnz")
    dist, prev = {header: 0}, {}
    heap = [(0, header)]
    best_end = None
    while heap:
        d, pc = heapq.heappop(heap)
        if d > dist.get(pc, 1 << 60):
            continue
        if d > 4000:
            break
        m = ins[pc].split()[0]
        nxt = []
        if m == "s_branch" or m in cond:
            tgt = labels.get(ins[pc].split()[1])
            if tgt is not None:
                nxt.append(tgt)
            if m in cond:
                nxt.append(pc + 1)
        elif m in ("s_endpgm", "s_setpc_b64"):
            pass
        else:
            nxt.append(pc + 1)
        if header in nxt:  # round the loop (the back edge may land on a latch block that falls into the header): the first
            best_end = pc  # time this happens is the cheapest way round
            break
        for n in nxt:
            if n < len(ins) and d + 1 < dist.get(n, 1 << 60):
                dist[n] = d + 1
                prev[n] = pc
                heapq.heappush(heap, (d + 1, n))
    if best_end is None:
        return []
    path, pc = [], best_end
    while True:
        path.append(ins[pc])
        if pc == header:
            break
        pc = prev[pc]
    return path[::-1]


def main(C, dst):
    from paragraph_amd import build as pgbuild
    ins, labels, headers = parse(kernel_asm(C))
    # the step loops hold the recurrence (v_pk_maximum3_f16 clusters); the forward graph's also holds the `nt` trace stores
    out = {"what": "VALU instructions on the common path of ONE pipeline step of pg_fill_kernel<%d, false, 16>, by issue class "
                   "(tools/isa_mix.py; classes as measured in profiles/r04_valu_rate.json)" % C,
           "kernel_source_sha": pgbuild.kernel_source_sha(), "C": C, "directions": {}}
    for h in headers:
        p = common_path(ins, labels, h)
        n_max3 = sum(1 for s in p if s.startswith("v_pk_maximum3_f16"))
        if n_max3 < 2 * C:  # not a step loop (two steps of 2 C + column maximum each)
            continue
        n_nt = sum(1 for s in p if s.startswith("global_store_dword") and " nt" in s)
        kinds = collections.Counter(issue_class(s) for s in p)
        mnem = collections.Counter(s.split()[0] for s in p if s.startswith("v_"))
        key = "forward_graph" if n_nt else "reversed_graph"
        if key in out["directions"] and len(p) >= out["directions"][key]["_len"]:
            continue  # (an outer loop around the same body)
        out["directions"][key] = {"_len": len(p), "steps_per_iteration": 2,
                                  "per_step": {k: v / 2.0 for k, v in sorted(kinds.items())},
                                  "valu_mnemonics_per_step": {k: v / 2.0 for k, v in mnem.most_common()},
                                  "trace_stores_per_step": n_nt / 2.0}
    for d in out["directions"].values():
        d.pop("_len")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v["per_step"] for k, v in out["directions"].items()}, indent=1))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10,
         sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r04_fill_isa_mix.json"))
