"""Graph descriptions for simple events given as {"chrom", "start", "end"[, "ins"][, "flank"]} (1-based, inclusive): what
`multiparagraph.py` / `multigrmpy.py -i candidates.json` feed the aligner with (grm.graph_templates.make_graph and its five
templates, src/python/lib/grm/graph_templates/*.py).  Pinned on the graphs inside share/test-data/multiparagraph/expected.json
(tests/test_graph_templates_cpu.py).

Event kinds (by reference length R = end - start + 1 and inserted sequence): "del" / "longdel" (no insertion; long when
R > 2 x flank), "swap" / "longswap" (both), "ins" (R <= 0).  Long events get two break-end subgraphs between a source and a
sink node instead of one contiguous middle node.  One quirk is kept on purpose: the MID_R node of a long swap is always
"<chrom>:1-1" (the original clamps with min() where the long deletion uses max()).
"""


def _interval(chrom, first, last):
    return "%s:%d-%d" % (chrom, first, last)


def _edge(src, dst, label=None):
    e = {"from": src, "to": dst}
    if label:
        e["sequences"] = [label]
    return e


def _path(nodes, label, index, length):
    return {"nodes": nodes, "path_id": "%s|%d" % (label, index), "sequence": label, "nucleotide_length": length}


def make_graph(event):
    """Returns (kind, graph dict).  The reference FASTA is not needed: nodes point at it by interval."""
    flank = event.get("flank", 150)
    ins = event.get("ins", "")
    ref_len = event["end"] - event["start"] + 1
    has_del = ref_len > 0
    if not has_del and not ins:
        raise ValueError("an event needs deleted bases, an inserted sequence, or both")
    chrom = event["chrom"]
    start, end = min(event["start"], event["end"]), max(event["start"], event["end"])
    long_event = has_del and ref_len > 2 * flank
    left_edge = max(1, start - flank - 1)
    lf = {"name": "LF", "reference": _interval(chrom, left_edge, max(1, start - 1) if (long_event or ins) else start - 1)}
    ins_node = {"name": "INS", "sequence": ins}

    if not has_del:  # pure insertion after base `start`
        rf = {"name": "RF", "reference": _interval(chrom, start + 1, start + flank + 1)}
        return "ins", {
            "sequencenames": ["REF", "INS"],
            "target_regions": [_interval(chrom, left_edge, start + flank + 1)],
            "nodes": [lf, ins_node, rf],
            "edges": [_edge("LF", "RF", "REF"), _edge("LF", "INS", "INS"), _edge("INS", "RF", "INS")],
            "paths": [_path(["LF", "INS", "RF"], "INS", 1, len(ins) + 2 * flank), _path(["LF", "RF"], "REF", 1, 2 * flank)],
        }

    rf = {"name": "RF", "reference": _interval(chrom, end + 1, end + flank + 1)}
    labels = ["REF", "DEL"] + (["INS"] if ins else [])
    if not long_event:
        mid = {"name": "MID", "reference": _interval(chrom, start, end)}
        nodes = [lf, mid] + ([ins_node] if ins else []) + [rf]
        edges = [_edge("LF", "RF", "DEL"), _edge("LF", "MID", "REF")]
        if ins:
            edges += [_edge("LF", "INS", "INS"), _edge("INS", "RF", "INS")]
        edges.append(_edge("MID", "RF", "REF"))
        paths = [_path(["LF", "MID", "RF"], "REF", 1, ref_len + 2 * flank), _path(["LF", "RF"], "DEL", 1, 2 * flank)]
        if ins:
            paths.append(_path(["LF", "INS", "RF"], "INS", 1, 2 * flank + len(ins)))
        return ("swap" if ins else "del"), {
            "sequencenames": labels, "target_regions": [_interval(chrom, left_edge, end + flank + 1)], "nodes": nodes, "edges": edges,
            "paths": paths}

    clamp = min if ins else max  # see the module text
    mid_l = {"name": "MID_L", "reference": _interval(chrom, start, start + flank - 1)}
    mid_r = {"name": "MID_R", "reference": _interval(chrom, clamp(1, end - flank), clamp(1, end - 1))}
    terminal = {"sequence": "NNNNN"}
    nodes = [dict(terminal, name="source"), lf, mid_l] + ([ins_node] if ins else []) + [mid_r, rf, dict(terminal, name="sink")]
    edges = [_edge("source", "LF"), _edge("source", "MID_R"), _edge("LF", "RF", "DEL")]
    if ins:
        edges += [_edge("LF", "INS", "INS"), _edge("INS", "RF", "INS")]
    edges += [_edge("LF", "MID_L", "REF"), _edge("MID_R", "RF", "REF"), _edge("MID_R", "sink"), _edge("RF", "sink")]
    paths = [_path(["LF", "MID_L"], "REF", 1, 2 * flank), _path(["MID_R", "RF"], "REF", 2, 2 * flank), _path(["LF", "RF"], "DEL", 1, 2 * flank)]
    if ins:
        paths.append(_path(["LF", "INS", "RF"], "INS", 1, 2 * flank + len(ins)))
    return ("longswap" if ins else "longdel"), {
        "sequencenames": labels,
        "target_regions": [_interval(chrom, left_edge, start + flank + 1), _interval(chrom, max(1, end - flank - 1), end + flank + 1)],
        "nodes": nodes, "edges": edges, "paths": paths}
