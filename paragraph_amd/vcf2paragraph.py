"""VCF -> variant-graph description: what src/python/bin/vcf2paragraph.py produces, without pysam.

The reference builds the graph in three layers -- a "VCF graph" of labelled reference intervals and ALT alleles
(src/python/lib/grm/vcfgraph/vcfgraph.py:33-436), a node / edge container (graphContainer.py:24-241) and a set of graph
rewrites (graphUtils.py:24-292) driven by convert_vcf (lib/grm/vcf2paragraph/__init__.py:48-120).  This module restates
that pipeline on plain Python containers:

  read_vcf            text VCF (optionally gzip) -> records with pysam's pos / stop / alleles / first-haplotype semantics
  Reference           FASTA access through the .fai index (or any object with fetch(chrom, start0, end))
  _labelled_pieces    the interval bookkeeping the reference does with an intervaltree: overlapping labelled spans are cut at
                      every boundary and a piece carries the union of the labels of the spans that cover it
  SequenceGraph       nodes / edges with the rewrites (long-node splitting, empty-node removal, node merging, source / sink,
                      REF and ALT paths) and the topological output order
  convert_vcf         the driver; same options and defaults as the reference's command line

Output order is part of the contract (the reference's expected files are compared textually, test_VCF2Paragraph.py:56-77): nodes in
reverse DFS post-order with children visited in name order, edges by (from, to) rank, paths in discovery order.  Where the
reference iterates a Python set of haplotype names (order = string hash, i.e. unspecified) this module iterates them sorted.
One behaviour is kept on purpose: in allele-graph mode the reference means to link every ALT node to all nodes starting
behind it but looks them up with the wrong key (vcfgraph.py:409, `nodes_starting_at[node["end"]+1]` instead of the
(chrom, position) pair), so that loop never adds an edge -- neither does this one.

Pinned on the reference's own VCF / JSON pairs (tests/golden/vcf2paragraph, tests/test_vcf2paragraph_cpu.py).
"""
import argparse
import gzip
import json
import re
import sys
from collections import OrderedDict, defaultdict


# ---------------------------------------------------------------------------------------------------
# inputs
# ---------------------------------------------------------------------------------------------------
class Reference:
    """Random access to an indexed FASTA (.fai next to it): fetch(chrom, start0, end) like pysam.FastaFile.fetch."""

    def __init__(self, path):
        self.path = path
        self.index = {}
        with open(path + ".fai") as f:
            for line in f:
                name, length, offset, bases, width = line.rstrip("\n").split("\t")[:5]
                self.index[name] = (int(length), int(offset), int(bases), int(width))
        self.handle = open(path, "rb")

    def fetch(self, chrom, start, end):
        length, offset, bases, width = self.index[chrom]
        start, end = max(0, start), min(end, length)
        if end <= start:
            return ""
        first = offset + (start // bases) * width + start % bases
        last = offset + ((end - 1) // bases) * width + (end - 1) % bases
        self.handle.seek(first)
        return self.handle.read(last - first + 1).replace(b"\n", b"").replace(b"\r", b"").decode()


class Record:
    """One VCF line with the fields the conversion reads (pysam.VariantRecord's names)."""

    def __init__(self, chrom, pos, vid, ref, alts, info, first_alleles):
        self.chrom, self.pos, self.id, self.ref, self.alts, self.info = chrom, pos, vid, ref, alts, info
        self.alleles = (ref,) + alts
        # pysam: stop = END when the INFO field has one, else pos + len(REF) - 1 (1-based inclusive)
        self.stop = int(info["END"]) if "END" in info else pos + len(ref) - 1
        self.first_alleles = first_alleles  # sample name -> allele string of its first haplotype (None if missing)


def read_vcf(path):
    """-> (sample names, records in file order)."""
    opener = gzip.open if path.endswith(".gz") else open
    samples, records = [], []
    with opener(path, "rt") as f:
        for line in f:
            line = line.rstrip("\n")
            if not line or line.startswith("##"):
                continue
            fields = line.split("\t")
            if line.startswith("#"):
                samples = fields[9:]
                continue
            chrom, pos, vid, ref, alt = fields[0], int(fields[1]), fields[2], fields[3], fields[4]
            alts = tuple(a for a in alt.split(",") if a != ".") if alt != "." else ()
            info = OrderedDict()
            if len(fields) > 7 and fields[7] != ".":
                for item in fields[7].split(";"):
                    key, _, value = item.partition("=")
                    info[key] = value if _ else True
            first = {}
            if len(fields) > 9:
                keys = fields[8].split(":")
                gi = keys.index("GT") if "GT" in keys else None
                for name, column in zip(samples, fields[9:]):
                    gt = column.split(":")[gi] if gi is not None and gi < len(column.split(":")) else "."
                    idx = [None if a in (".", "") else int(a) for a in re.split(r"[/|]", gt)]
                    alleles = (ref,) + alts
                    first[name] = None if any(i is None for i in idx) else alleles[idx[0]]
            records.append(Record(chrom, pos, None if vid in (".", "") else vid, ref, alts, info, first))
    return samples, records


def parse_region(text):
    """"chr:start-end" -> (chrom, start, end), missing parts None (lib/grm/helpers.py:26-43)."""
    text = text.replace(",", "")
    chrom, _, rest = text.partition(":")
    if not rest:
        return chrom, None, None
    start, _, end = rest.partition("-")
    return chrom, int(start), int(end) if end else None


class NoRecords(Exception):
    pass


# ---------------------------------------------------------------------------------------------------
# layer 1: labelled reference spans + ALT alleles of one chromosome (vcfgraph.py)
# ---------------------------------------------------------------------------------------------------
def _labelled_pieces(spans, cuts):
    """spans: [(start, end_inclusive, labels)], cuts: extra boundaries.  Yields (start, end, labels) of the maximal pieces
    between boundaries that at least one span covers, labels = union over the covering spans (vcfgraph.py:226-242 after
    IntervalTree.split_overlaps / slice)."""
    bounds = sorted({s for s, _, _ in spans} | {e + 1 for _, e, _ in spans} | set(cuts))
    for lo, hi in zip(bounds, bounds[1:]):
        cover = [labels for s, e, labels in spans if s <= lo and hi - 1 <= e]
        if cover:
            yield lo, hi - 1, set().union(*cover)


class _VariantAlleles:
    def __init__(self, reference, chrom):
        self.reference, self.chrom = reference, chrom
        self.spans, self.cuts = [], set()
        self.alts = OrderedDict()  # "start-end:seq" -> [start, end, sequence, labels]
        self.first_pos = self.last_pos = None

    # -- vcfgraph.py:195-224
    def ref_support(self, start, end, labels=(), alleles=None):
        shared = 0
        if alleles:
            shortest = min(len(a) for a in alleles)
            while shared < shortest and all(alleles[0][shared] == a[shared] for a in alleles):
                shared += 1
            if start + shared > end + 1:
                raise ValueError("%d:%d error in adding ref support." % (start, end))
        if end < start:
            raise ValueError("empty reference span %d-%d" % (start, end))
        if shared > 0:
            # the whole block exists, but the label only goes on the bases behind the padding shared by all alleles
            self.spans.append((start, end, set()))
            if labels and start + shared <= end:
                self.spans.append((start + shared, end, set(labels)))
        else:
            self.spans.append((start, end, set(labels)))

    # -- vcfgraph.py:244-289
    def alt(self, start, end, ref, alt, labels=(), other_labels=()):
        if len(ref) != end - start + 1:
            raise ValueError("%d:%d REF != END - POS + 1" % (start, end))
        a_start, a_end = start, end
        while alt and ref and ref[0] == alt[0]:
            ref, alt, a_start = ref[1:], alt[1:], a_start + 1
        if a_start > start:
            self.ref_support(start, a_start - 1)  # padding bases are no "reference call"
        while alt and ref and ref[-1] == alt[-1]:
            ref, alt, a_end = ref[:-1], alt[:-1], a_end - 1
        if a_end <= 0:
            raise ValueError("%d:%d error in adding alt. negative or zero ALT end." % (start, end))
        if a_start <= a_end < end:
            self.ref_support(a_end + 1, end, labels)  # the trimmed tail (not for insertions)
        if not ref and not alt:
            raise ValueError("%d:%d missing REF or ALT sequence." % (start, end))
        self._alt(a_start, a_end, alt, labels)
        if other_labels and a_start > a_end:
            self._alt(a_start, a_end, "", other_labels)  # bypass for the other alleles of an insertion

    def _alt(self, start, end, seq, labels):
        key = "%d-%d:%s" % (start, end, seq)
        self.alts.setdefault(key, [start, end, seq, set()])[3].update(labels)

    # -- vcfgraph.py:127-193
    def record(self, rec, allele_graph, var_id, ins_info_key):
        if allele_graph or not rec.first_alleles:
            # allele labels "<variant id>:<allele index>" (vcfgraph.py:85-87); without sample columns the haplotype form
            # has nothing to label with
            labels = {"%s:%d" % (var_id, n): a for n, a in enumerate(rec.alleles)} if allele_graph else {}
        else:
            labels = {s: a for s, a in rec.first_alleles.items() if a is not None}
        ref_labels = {s for s, a in labels.items() if a == rec.ref}
        self.ref_support(rec.pos, rec.stop, ref_labels, rec.alleles)
        for alt in rec.alts:
            alt_labels = {s for s, a in labels.items() if a == alt}
            ref_seq = self.reference.fetch(self.chrom, rec.pos - 1, rec.stop).upper()
            if len(ref_seq) != rec.stop - rec.pos + 1:
                raise ValueError("%s:%d fail to retrieve genome REF. Are you using the correct ref genome?" % (rec.chrom, rec.pos))
            if "<" in alt:
                if alt == "<INS>":
                    if ins_info_key not in rec.info:
                        raise ValueError("Missing key %s for <INS> at %s:%d; " % (ins_info_key, self.chrom, rec.pos))
                    ins = str(rec.info[ins_info_key]).upper()
                    if re.search(r"[^ACGTNX]", ins):
                        raise ValueError("Illegal character in INS sequence: %s" % ins)
                    self.alt(rec.pos, rec.stop, ref_seq, ref_seq[0] + ins, alt_labels, ref_labels)
                else:
                    if rec.stop == rec.pos:
                        raise ValueError("%s:%d Same END and POS in symbolic non-insertion. Did you miss the END key?" % (rec.chrom, rec.pos))
                    if alt == "<DEL>":
                        self.alt(rec.pos, rec.stop, ref_seq, ref_seq[0], alt_labels)
                    elif alt == "<DUP>":
                        self.alt(rec.pos, rec.pos, ref_seq[0], ref_seq, alt_labels, ref_labels)
                    elif alt == "<INV>":
                        body = ref_seq[1:1000] + ref_seq[len(ref_seq) - 1000:] if len(ref_seq) > 20000 else ref_seq[1:]
                        comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
                        try:
                            inverted = "".join(comp[c] for c in reversed(body))
                        except KeyError:
                            raise ValueError("%s:%d:<INV> illegal character in reference sequence" % (rec.chrom, rec.pos))
                        self.alt(rec.pos, rec.stop, ref_seq, ref_seq[0] + inverted, alt_labels, ref_labels)
            else:
                if re.search(r"[^ACGTNXacgtnx]", alt):
                    raise ValueError("Illegal character in ALT allele: %s" % alt)
                if (len(alt[0]) > 1 or len(ref_seq) > 1) and alt[0].upper() != ref_seq[0]:
                    raise ValueError("Different padding base for REF and ALT at %s:%d" % (rec.chrom, rec.pos))
                self.alt(rec.pos, rec.stop, ref_seq, alt, alt_labels, ref_labels)

    def labels(self):
        return {l for a in self.alts.values() for l in a[3]} | {l for _, _, ls in self.spans for l in ls}

    def pieces(self):
        return list(_labelled_pieces(self.spans, self.cuts))


def _variant_alleles(reference, records, ins_info_key, chrom, start, end, padding, allele_graph):
    """VCFGraph.create_from_vcf (vcfgraph.py:89-125): the records of ONE chromosome (the first one met unless given)."""
    va = _VariantAlleles(reference, chrom)
    id_counts = defaultdict(int)
    n = 0
    for rec in records:
        if chrom is not None and start is not None and (rec.stop < start or (end is not None and rec.pos > end)):
            continue
        if chrom is None:
            chrom = va.chrom = rec.chrom
        elif rec.chrom != chrom:
            if n:
                break
            continue
        va.first_pos = rec.pos if va.first_pos is None else va.first_pos
        va.last_pos = rec.stop if va.last_pos is None or va.last_pos < rec.stop else va.last_pos
        if rec.id:  # vcfgraph.py:62-83
            if rec.id in id_counts:
                raise ValueError("Duplicated variant ID: %s" % rec.id)
            id_counts[rec.id] = 1
            var_id = rec.id
        else:
            var_id = "%s:%d" % (rec.chrom, rec.pos)
            id_counts[var_id] += 1
            var_id = "%s-%d" % (var_id, id_counts[var_id])
        n += 1
        va.record(rec, allele_graph, var_id, ins_info_key)
    if not n:
        raise NoRecords("No VCF records found at %s:%s-%s" % (chrom, start, end))
    va.ref_support(va.first_pos - padding, va.last_pos + padding)
    for a_start, a_end, _, _ in list(va.alts.values()):
        if va.first_pos <= a_end <= va.last_pos:
            va.cuts.add(a_end + 1)  # a boundary for the ALT to link into
        else:
            va.ref_support(a_end + 1, a_end + padding)
    return va


# ---------------------------------------------------------------------------------------------------
# layer 2 + 3: nodes / edges and the rewrites (graphContainer.py, graphUtils.py)
# ---------------------------------------------------------------------------------------------------
class SequenceGraph:
    def __init__(self, name="VCF Graph"):
        self.name = name
        self.nodes = OrderedDict()   # name -> dict(name, reference | position + sequence, labels, chrom, start, end)
        self.edges = OrderedDict()   # "<from>_<to>" -> dict(from, to, labels)
        self.around = defaultdict(list)  # node name -> its edges, in the order they were added
        self.labels = set()
        self.paths = []
        self.target_regions = None
        self.ref_by_start, self.ref_by_end = {}, {}

    # -- construction
    def ref_node(self, chrom, start, end, labels=()):
        span = "%s:%d-%d" % (chrom, start, end)
        node = {"name": "ref-" + span, "reference": span, "labels": set(labels), "chrom": chrom, "start": start, "end": end}
        self.ref_by_start[chrom, start] = self.ref_by_end[chrom, end] = node
        self.nodes[node["name"]] = node
        self.labels.update(labels)
        return node

    def alt_node(self, chrom, start, end, sequence, labels=()):
        span = "%s:%d-%d" % (chrom, start, end)
        node = {"name": "%s:%s" % (span, sequence), "position": span, "sequence": sequence, "labels": set(labels), "chrom": chrom,
                "start": start, "end": end}
        self.nodes[node["name"]] = node
        self.labels.update(labels)
        return node

    def link(self, a, b, labels=()):
        key = a["name"] + "_" + b["name"]
        if key not in self.edges:
            assert a["name"] != b["name"], a["name"]
            edge = {"from": a["name"], "to": b["name"], "labels": set(), "name": key}
            self.edges[key] = edge
            self.around[a["name"]].append(edge)
            self.around[b["name"]].append(edge)
        self.edges[key]["labels"].update(labels)
        self.labels.update(labels)

    def unlink(self, edge):
        for end in ("from", "to"):
            self.around[edge[end]] = [e for e in self.around[edge[end]] if e["name"] != edge["name"]]
        del self.edges[edge["name"]]

    def drop(self, node):
        for e in list(self.around[node["name"]]):
            self.unlink(e)
        del self.nodes[node["name"]]

    def into(self, node, label=None):
        return [e for e in self.around[node["name"]] if e["to"] == node["name"] and (label is None or label in e["labels"])]

    def out_of(self, node, label=None):
        return [e for e in self.around[node["name"]] if e["from"] == node["name"] and (label is None or label in e["labels"])]

    def ref_nodes(self):
        return [n for n in self.nodes.values() if "reference" in n]

    def alt_nodes(self):
        return [n for n in self.nodes.values() if "reference" not in n]

    def with_label(self, label):
        return sorted((n for n in self.nodes.values() if label in n["labels"]), key=lambda n: (n["start"], n["end"]))

    # -- VCFGraph.get_graph (vcfgraph.py:357-429)
    @staticmethod
    def from_alleles(va, allele_graph):
        g = SequenceGraph()
        prev = None
        for start, end, labels in va.pieces():
            node = g.ref_node(va.chrom, start, end, labels)
            if prev is not None:
                if prev["end"] + 1 != node["start"]:
                    raise ValueError("%d:%d node start != prev node end + 1" % (node["start"], prev["end"]))
                g.link(prev, node)
            prev = node
        for start, end, seq, labels in va.alts.values():
            g.alt_node(va.chrom, start, end, seq, labels)
        names = sorted(va.labels() - {None})
        for label in names:  # edges along every haplotype / allele
            prev = None
            for node in g.with_label(label):
                if prev is not None:
                    if prev["end"] == node["start"] - 1:
                        g.link(prev, node, [label])
                    dummy = prev["end"] == prev["start"] - 1 and not prev.get("sequence")
                    before = prev["end"] < node["start"] and prev["start"] < node["start"]
                    if not dummy and not before:
                        raise ValueError("Inconsistent nodes for haplotype %s: %s, %s" % (label, prev["name"], node["name"]))
                prev = node
        for node in g.alt_nodes():  # ALT nodes without a way in / out hang on the reference
            if allele_graph or not g.into(node):
                g.link(g.ref_by_end[node["chrom"], node["start"] - 1], node)
            if not g.out_of(node):
                g.link(node, g.ref_by_start[node["chrom"], node["end"] + 1])
            # (allele-graph mode: the reference's "link to every node starting behind it" loop never finds a node, see above)
        for label in names:  # a label without a determined way in / out of a node takes all of them
            for node in g.with_label(label):
                if not g.into(node, label):
                    for e in g.into(node):
                        g.link(g.nodes[e["from"]], node, [label])
                if not g.into(node, label):
                    raise ValueError("Error in get graph.")
                if not g.out_of(node, label):
                    for e in g.out_of(node):
                        g.link(node, g.nodes[e["to"]], [label])
        return g

    # -- graphUtils.py:58-106
    def split_long_nodes(self, max_len, padding, alt_too):
        assert max_len >= 2 * padding
        for node in self.ref_nodes():
            if node["end"] - node["start"] + 1 <= max_len:
                continue
            head = self.ref_node(node["chrom"], node["start"], node["start"] + padding - 1, node["labels"])
            tail = self.ref_node(node["chrom"], node["end"] - padding + 1, node["end"], node["labels"])
            self._replace(node, head, tail)
        if alt_too:
            for node in self.alt_nodes():
                if len(node["sequence"]) <= max_len:
                    continue
                head = self.alt_node(node["chrom"], node["start"], node["end"], node["sequence"][:padding], node["labels"])
                tail = self.alt_node(node["chrom"], node["start"], node["end"], node["sequence"][-padding:], node["labels"])
                self._replace(node, head, tail)

    def _replace(self, node, head, tail):
        for e in self.into(node):
            self.link(self.nodes[e["from"]], head, e["labels"])
        for e in self.out_of(node):
            self.link(tail, self.nodes[e["to"]], e["labels"])
        self.drop(node)

    # -- graphUtils.py:109-131
    def remove_empty_nodes(self):
        for node in list(self.nodes.values()):
            if ("reference" in node and node["start"] <= node["end"]) or node.get("sequence", "") != "":
                continue
            seen_in = [l for e in self.into(node) for l in e["labels"]]
            seen_out = [l for e in self.out_of(node) for l in e["labels"]]
            for e1 in self.into(node):
                for e2 in self.out_of(node):
                    # labels seen on both sides, or on one side only while undetermined on the other
                    keep = (e1["labels"] & e2["labels"]) | (e1["labels"] - set(seen_out)) | (e2["labels"] - set(seen_in))
                    self.link(self.nodes[e1["from"]], self.nodes[e2["to"]], keep)
            self.drop(node)

    # -- graphUtils.py:134-166
    def merge_chains(self):
        for a in list(self.nodes.values()):
            out = self.out_of(a)
            if len(out) != 1:
                continue
            b = self.nodes[out[0]["to"]]
            if len(self.into(b)) != 1 or not (a["chrom"] == b["chrom"] and a["end"] + 1 == b["start"]):
                continue
            if a["labels"] != b["labels"] or ("reference" in a) != ("reference" in b):
                continue
            if "reference" in a:
                merged = self.ref_node(a["chrom"], a["start"], b["end"], a["labels"])
            else:
                merged = self.alt_node(a["chrom"], a["start"], b["end"], a["sequence"] + b["sequence"], a["labels"])
            for e in self.into(a):
                self.link(self.nodes[e["from"]], merged, e["labels"])
            for e in self.out_of(b):
                self.link(merged, self.nodes[e["to"]], e["labels"])
            self.drop(a)
            self.drop(b)

    # -- graphUtils.py:272-284
    def absorb(self, other):
        for n in other.ref_nodes():
            self.ref_node(n["chrom"], n["start"], n["end"], n["labels"])
        for n in other.alt_nodes():
            self.alt_node(n["chrom"], n["start"], n["end"], n["sequence"], n["labels"])
        for e in other.edges.values():
            self.link(self.nodes[e["from"]], self.nodes[e["to"]], e["labels"])
        self.paths += other.paths

    # -- graphUtils.py:27-55
    def add_source_sink(self):
        for name in ("source", "sink"):
            self.nodes.setdefault(name, {"name": name, "sequence": "N" * 10})
        for node in list(self.nodes.values()):
            if node["name"] in ("source", "sink"):
                continue
            if not self.into(node):
                self.link(self.nodes["source"], node)
            if not self.out_of(node):
                self.link(node, self.nodes["sink"])

    # -- graphContainer.py:170-197
    def ordered(self):
        state, order = {}, []

        def visit(node):
            state[node["name"]] = "open"
            for nxt in sorted((self.nodes[e["to"]] for e in self.out_of(node)), key=lambda n: n["name"]):
                if nxt["name"] not in state:
                    visit(nxt)
                elif state[nxt["name"]] == "open":
                    raise ValueError("Graph has a cycle at %s -> %s" % (node["name"], nxt["name"]))
            state[node["name"]] = "done"
            order.insert(0, node)

        for node in self.nodes.values():
            if node["name"] not in state:
                visit(node)
        rank = {n["name"]: i for i, n in enumerate(order)}
        return order, sorted(self.edges.values(), key=lambda e: (rank[e["from"]], rank[e["to"]]))

    # -- graphUtils.py:188-260
    def _walks(self, label):
        nodes, _ = self.ordered()
        used = set()

        def follow(edge, walk):
            node = self.nodes[edge["to"]]
            walk = walk + [node["name"]]
            used.add(edge["name"])
            found = []
            for e in self.out_of(node, label):
                if e["name"] not in used:
                    found.extend(follow(e, walk))
            return found or [walk]

        walks = []
        for node in nodes:
            for edge in self.out_of(node, label):
                if edge["name"] not in used:
                    walks += follow(edge, [node["name"]])
        return walks

    def reference_paths(self):
        for a in self.ref_nodes():
            for e in self.out_of(a):
                b = self.nodes[e["to"]]
                if "reference" in b and a["end"] + 1 == b["start"]:
                    self.link(a, b, ["REF"])
        return [{"nodes": w, "path_id": "REF|%d" % (i + 1), "sequence": "REF"} for i, w in enumerate(self._walks("REF"))]

    def alternate_paths(self):
        ref = [p["nodes"] for p in self.reference_paths()]
        out = []
        for walk in self._walks(None):
            walk = walk[1:] if walk[0] == "source" else walk
            walk = walk[:-1] if walk[-1] == "sink" else walk
            if walk not in ref:
                out.append({"nodes": walk, "path_id": "ALT|%d" % (len(out) + 1), "sequence": "ALT"})
                self.labels.add("ALT")
        return out

    # -- graphContainer.py:199-241 (regions) / 211-241 (document)
    def reference_regions(self):
        out = []
        for chrom in sorted({n["chrom"] for n in self.ref_nodes()}):
            spans = sorted((n["start"], n["end"]) for n in self.ref_nodes() if n["chrom"] == chrom)
            cur = None
            for s, e in spans:
                if cur is not None and s <= cur[1] + 1:
                    cur[1] = max(cur[1], e)
                else:
                    if cur is not None:
                        out.append("%s:%d-%d" % (chrom, cur[0], cur[1]))
                    cur = [s, e]
            if cur is not None:
                out.append("%s:%d-%d" % (chrom, cur[0], cur[1]))
        return out

    def document(self):
        nodes, edges = self.ordered()
        doc_nodes = [{k: v for k, v in n.items() if k not in ("labels", "chrom", "start", "end")} for n in nodes]
        doc_edges = []
        for e in edges:
            d = {"from": e["from"], "to": e["to"], "name": e["name"]}
            if e["labels"]:
                d["sequences"] = sorted(e["labels"])
            doc_edges.append(d)
        return {"nodes": doc_nodes, "edges": doc_edges, "paths": self.paths, "target_regions": sorted(self.target_regions),
                "sequencenames": sorted(self.labels), "model_name": self.name}


# ---------------------------------------------------------------------------------------------------
# driver (lib/grm/vcf2paragraph/__init__.py:39-120)
# ---------------------------------------------------------------------------------------------------
def add_reference_information(document, reference):
    for node in document["nodes"]:
        if "reference" in node:
            chrom, start, end = parse_region(node["reference"])
            node["reference_sequence"] = reference.fetch(chrom, start - 1, end).upper()


def convert_vcf(vcf, reference, ins_info_key="SEQ", target_regions=None, ref_node_padding=150, ref_node_max_length=1000,
                allele_graph=False, simplify=True, alt_paths=False, alt_splitting=False, retrieve_reference_sequence=False):
    """VCF file -> graph description (dict in the reference's input schema).  `reference`: FASTA path or an object with
    fetch(chrom, start0, end).  Defaults are the reference command line's (bin/vcf2paragraph.py:44-66)."""
    if isinstance(reference, str):
        reference = Reference(reference)
    _, records = read_vcf(vcf)
    graph = SequenceGraph("Graph from %s" % vcf)
    for chrom, start, end in ([parse_region(r) for r in target_regions] if target_regions else [(None, None, None)]):
        try:
            alleles = _variant_alleles(reference, records, ins_info_key, chrom, start, end, ref_node_padding, allele_graph)
        except NoRecords:
            continue
        part = SequenceGraph.from_alleles(alleles, allele_graph)
        if ref_node_max_length:
            part.split_long_nodes(ref_node_max_length, ref_node_padding, alt_splitting)
        if simplify:
            part.remove_empty_nodes()
            part.merge_chains()
        graph.absorb(part)
    graph.target_regions = list(target_regions) if target_regions else graph.reference_regions()
    graph.add_source_sink()
    graph.paths += graph.reference_paths()
    if alt_paths:
        graph.paths += graph.alternate_paths()
    document = graph.document()
    if retrieve_reference_sequence:
        add_reference_information(document, reference)
    return document


def split_records(records, vcf_id, read_length=150, split_type="lines"):
    """Blocks of records converted together, and their graph ids (parse_vcf_lines, lib/grm/vcf2paragraph/__init__.py:178-266):
    "lines" one record per graph, "full" all in one, "by_id" consecutive records with the same ID, "superloci" records closer
    than a read length.  Ids are "<vcf id>:<n>" (n from 1; 0 for "full")."""
    blocks, ids = [], []
    prev_id, cur_chrom, prev_end = "", None, None
    for rec in records:
        if rec.pos < read_length:
            raise ValueError("Distance between vcf position and chrom start is smaller than read length.")
        if split_type == "full":
            if not blocks:
                blocks, ids = [[rec]], [vcf_id + ":0"]
            else:
                blocks[0].append(rec)
        elif split_type == "lines":
            blocks.append([rec])
            ids.append("%s:%d" % (vcf_id, len(blocks)))
        elif split_type == "by_id":
            if rec.id and rec.id == prev_id:
                blocks[-1].append(rec)
            else:
                blocks.append([rec])
                ids.append("%s:%d" % (vcf_id, len(blocks)))
            prev_id = rec.id
        elif split_type == "superloci":
            if cur_chrom is None or rec.chrom != cur_chrom or not prev_end or rec.pos > prev_end + read_length:
                blocks.append([rec])
                ids.append("%s:%d" % (vcf_id, len(blocks)))
            else:
                blocks[-1].append(rec)
            cur_chrom = rec.chrom
            prev_end = rec.stop if rec.stop and rec.stop >= rec.pos else rec.pos
        else:
            raise ValueError("Unknown VCF splitting type: %s" % split_type)
    return blocks, ids


def _convert_records(name, records, reference, ins_info_key, padding, max_len, allele_graph, alt_paths, alt_splitting):
    graph = SequenceGraph("Graph from %s" % name)
    alleles = _variant_alleles(reference, records, ins_info_key, None, None, None, padding, allele_graph)
    part = SequenceGraph.from_alleles(alleles, allele_graph)
    if max_len:
        part.split_long_nodes(max_len, padding, alt_splitting)
    part.remove_empty_nodes()
    part.merge_chains()
    graph.absorb(part)
    graph.target_regions = graph.reference_regions()
    graph.add_source_sink()
    graph.paths += graph.reference_paths()
    if alt_paths:
        graph.paths += graph.alternate_paths()
    return graph.document()


def convert_vcf_to_graphs(vcf, reference, read_length=150, max_ref_node_length=300, graph_type="alleles", split_type="lines",
                          retrieve_reference_sequence=False, alt_splitting=True, alt_paths=True, ins_info_key="SEQ"):
    """What multigrmpy.py does with a VCF input (convert_vcf_to_json, lib/grm/vcf2paragraph/__init__.py:123-175, with
    bin/multigrmpy.py's defaults): one graph description per block of records, each with "ID" (the GRMPY_ID the genotype
    records are linked by), "chrom" / "start" / "end" of its target regions and the graph under "graph"."""
    import hashlib
    import os
    if isinstance(reference, str):
        reference = Reference(reference)
    with open(vcf, "rb") as f:
        vcf_id = os.path.basename(vcf) + "@" + hashlib.sha256(f.read()).hexdigest()
    _, records = read_vcf(vcf)
    blocks, ids = split_records(records, vcf_id, read_length, split_type)
    out = []
    for block, bid in zip(blocks, ids):
        doc = _convert_records(vcf, block, reference, ins_info_key, read_length, max_ref_node_length, graph_type == "alleles", alt_paths,
                               alt_splitting)
        spans = [parse_region(r) for r in doc["target_regions"]]
        assert len({c for c, _, _ in spans}) == 1
        if retrieve_reference_sequence:
            add_reference_information(doc, reference)
        out.append({"graph": doc, "chrom": spans[0][0], "start": min(s for _, s, _ in spans), "end": max(e for _, _, e in spans), "ID": bid})
    return out


def main(argv=None):
    ap = argparse.ArgumentParser("vcf2paragraph", description="VCF -> Paragraph graph description (the reference's vcf2paragraph.py options)")
    ap.add_argument("input")
    ap.add_argument("output")
    ap.add_argument("-r", "--reference-sequence", dest="ref", required=True)
    ap.add_argument("-g", "--graph-type", choices=["alleles", "haplotypes"], default="haplotypes", dest="graph_type")
    ap.add_argument("-R", "--retrieve-reference-sequence", action="store_true", dest="retrieve_reference_sequence")
    ap.add_argument("-l", "--max-ref-node-length", dest="max_ref_len", type=int, default=1000)
    ap.add_argument("-p", "--read-length", "--read-len", dest="read_len", type=int, default=150)
    ap.add_argument("-T", "--target-region", dest="target_regions", default=[], action="append")
    ap.add_argument("--ins-info-key", dest="ins_info_key", default="SEQ")
    ap.add_argument("--alt-paths", dest="alt_paths", action="store_true")
    ap.add_argument("--alt-splitting", dest="alt_splitting", action="store_true")
    args = ap.parse_args(argv)
    doc = convert_vcf(args.input, args.ref, args.ins_info_key, args.target_regions, args.read_len, args.max_ref_len,
                      args.graph_type == "alleles", alt_paths=args.alt_paths, alt_splitting=args.alt_splitting,
                      retrieve_reference_sequence=args.retrieve_reference_sequence)
    with (sys.stdout if args.output == "-" else open(args.output, "w")) as out:
        json.dump(doc, out, sort_keys=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
