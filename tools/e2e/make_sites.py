"""Synthetic many-site input for the end-to-end probe: a random reference, N deletion / insertion / swap sites spaced
along it (graphs as JSON descriptions with reference-interval nodes), and ONE coordinate-sorted BAM of paired 150 bp reads
sampled around every site from a diploid genome that carries each alternate allele with genotype 0/0, 0/1 or 1/1.

    python tools/e2e/make_sites.py <outdir> [n_sites] [depth] [seed] [extras]

extras = 1 adds what the device kernels' envelope does not hold (the reference has no such bounds): one 600 bp read over
the last site, and three more graphs -- 5 000 nodes, 70 000 columns, 65 sequence labels -- listed in extra_graphs.txt.

Writes ref.fa(.fai), reads.bam(.bai), graphs/site_<i>.json, graphs.txt, manifest.txt, truth.json.
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import bamwriter  # noqa: E402  (synthetic-input writer; the product only reads BAMs)

COMP = str.maketrans("ACGT", "TGCA")


def main():
    out = sys.argv[1]
    n_sites = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    depth = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
    rng = random.Random(int(sys.argv[4]) if len(sys.argv) > 4 else 1)
    extras = len(sys.argv) > 5 and sys.argv[5] == "1"
    # "paths": fragment names start with the encoded path of the haplotype a fragment was drawn from, "(0@0)-(1)-...-(k@0)_..." (source and sink are one-base nodes "X" once loaded, GraphInput.cpp:86-89)
    # -- what grm::ValidationAligner reads the simulated path from (lib/grm/ValidationAligner.cpp:122-125)
    path_names = len(sys.argv) > 6 and sys.argv[6] == "paths"
    os.makedirs(os.path.join(out, "graphs"), exist_ok=True)
    spacing, flank, read_len, frag_mean = 3000, 150, 150, 400
    glen = spacing * (n_sites + 1)
    ref = "".join(rng.choice("ACGT") for _ in range(glen))
    bamwriter.write_fasta(os.path.join(out, "ref.fa"), [("chr1", ref)])
    records, truth, graph_paths = [], [], []
    for i in range(n_sites):
        start = spacing * (i + 1)                     # 0-based first deleted / replaced base
        kind = rng.choice(["del", "ins", "swap"])
        del_len = 0 if kind == "ins" else rng.randint(20, 300)
        ins = "" if kind == "del" else "".join(rng.choice("ACGT") for _ in range(rng.randint(10, 120)))
        end = start + del_len                         # first base after the event
        lf = (start - flank, start)                   # [a, b) intervals on the reference
        rf = (end, end + flank)
        nodes = [{"name": "source", "sequence": "NNNNNNNNNN"}, {"name": "LF", "reference": "chr1:%d-%d" % (lf[0] + 1, lf[1])}]
        edges = [{"from": "source", "to": "LF"}]
        if del_len:
            nodes.append({"name": "REF", "reference": "chr1:%d-%d" % (start + 1, end)})
            edges += [{"from": "LF", "to": "REF", "sequences": ["REF"]}]
        if ins:
            nodes.append({"name": "INS", "sequence": ins})
            edges += [{"from": "LF", "to": "INS", "sequences": ["ALT"]}]
        nodes += [{"name": "RF", "reference": "chr1:%d-%d" % (rf[0] + 1, rf[1])}, {"name": "sink", "sequence": "NNNNNNNNNN"}]
        if del_len:
            edges.append({"from": "REF", "to": "RF", "sequences": ["REF"]})
        if ins:
            edges.append({"from": "INS", "to": "RF", "sequences": ["ALT"]})
        if not del_len:
            edges.append({"from": "LF", "to": "RF", "sequences": ["REF"]})
        if not ins:
            edges.append({"from": "LF", "to": "RF", "sequences": ["ALT"]})
        edges.append({"from": "RF", "to": "sink"})
        order = {n["name"]: k for k, n in enumerate(nodes)}
        edges.sort(key=lambda e: (order[e["from"]], order[e["to"]]))
        mid_ref = ["REF"] if del_len else []
        mid_alt = ["INS"] if ins else []
        spec = {"ID": "site_%d" % i, "nodes": nodes, "edges": edges, "sequencenames": ["ALT", "REF"],
                "target_regions": ["chr1:%d-%d" % (lf[0] + 1, rf[1])],
                "paths": [{"nodes": ["source", "LF"] + mid_ref + ["RF", "sink"], "path_id": "REF|1", "sequence": "REF"},
                          {"nodes": ["source", "LF"] + mid_alt + ["RF", "sink"], "path_id": "ALT|1", "sequence": "ALT"}]}
        gp = os.path.join(out, "graphs", "site_%d.json" % i)
        with open(gp, "w") as f:
            json.dump(spec, f)
        graph_paths.append(gp)
        gt = rng.choice([(0, 0), (0, 1), (1, 1)])
        truth.append({"ID": spec["ID"], "gt": "/".join("ALT" if a else "REF" for a in sorted(gt, reverse=True)), "kind": kind})
        # reads: fragments drawn from the two haplotypes over [start - 700, end + 700)
        lo, hi = start - 700, end + 700
        haps = [ref[lo:start] + (ins if a else ref[start:end]) + ref[end:hi] for a in gt]
        n_frag = int(depth * (hi - lo) / (2 * read_len))
        for k in range(n_frag):
            h = rng.randrange(2)
            hap = haps[h]
            fl = max(read_len + 10, int(rng.gauss(frag_mean, 40)))
            if fl >= len(hap):
                continue
            a = rng.randrange(len(hap) - fl)
            r1 = hap[a:a + read_len]
            r2 = hap[a + fl - read_len:a + fl]

            def to_ref(x, alt=gt[h]):  # haplotype offset -> approximate linear position (what a mapper would report)
                if x <= start - lo or not alt:
                    return lo + x
                return max(start, lo + x - len(ins) + del_len)
            p1, p2 = to_ref(a), to_ref(a + fl - read_len)

            def err(s):  # ~0.2 % substitutions
                while rng.random() < 0.26:
                    k = rng.randrange(len(s))
                    s = s[:k] + rng.choice("ACGT") + s[k + 1:]
                return s
            name = "s%d_f%d" % (i, k)
            if path_names:
                ids = [order[x] for x in (["source", "LF"] + (mid_alt if gt[h] else mid_ref) + ["RF", "sink"])]
                enc = "".join(("(%d@0)" % n) if j == 0 else ("-(%d@0)" % n) if j == len(ids) - 1 else ("-(%d)" % n) for j, n in enumerate(ids))
                name = enc + "_" + name
            records.append(dict(name=name, tid=0, pos=p1, seq=err(r1), flag=0x63, mtid=0, mpos=p2, mapq=60))
            records.append(dict(name=name, tid=0, pos=p2, seq=err(r2.translate(COMP)[::-1]).translate(COMP)[::-1], flag=0x93, mtid=0, mpos=p1, mapq=60))
    if extras:
        start = spacing * n_sites  # the last site
        records.append(dict(name="long_read", tid=0, pos=start - 300, seq=ref[start - 300:start + 300], flag=0, mtid=-1, mpos=-1, mapq=60))
        erng = random.Random(12345)
        c0 = spacing - flank  # 5 000 four-base pieces of the reference from the first site's left flank on
        chain = [{"name": "source", "sequence": "NNNNNNNNNN"}] + [{"name": "n%d" % k, "reference": "chr1:%d-%d" % (c0 + 4 * k + 1, c0 + 4 * k + 4)} for k in range(5000)]
        cedges = [{"from": chain[k]["name"], "to": chain[k + 1]["name"]} for k in range(len(chain) - 1)]
        cedges[10]["sequences"] = ["REF"]
        cedges.append({"from": "n9", "to": "n11", "sequences": ["ALT"]})
        wide = [{"name": "LF", "reference": "chr1:%d-%d" % (spacing * 2 - flank + 1, spacing * 2)},
                {"name": "MID", "sequence": "".join(erng.choice("ACGT") for _ in range(70000))},
                {"name": "RF", "reference": "chr1:%d-%d" % (spacing * 2 + 1, spacing * 2 + flank)}]
        wedges = [{"from": "LF", "to": "MID", "sequences": ["ALT"]}, {"from": "LF", "to": "RF", "sequences": ["REF"]}, {"from": "MID", "to": "RF", "sequences": ["ALT"]}]
        labels65 = ["L%02d" % k for k in range(65)]
        lnodes = [{"name": "LF", "reference": "chr1:%d-%d" % (spacing * 3 - flank + 1, spacing * 3)},
                  {"name": "MID", "sequence": "ACGTTGCAACGTACGT"},
                  {"name": "RF", "reference": "chr1:%d-%d" % (spacing * 3 + 1, spacing * 3 + flank)}]
        ledges = [{"from": "LF", "to": "MID", "sequences": labels65[:33]}, {"from": "LF", "to": "RF", "sequences": labels65[33:]},
                  {"from": "MID", "to": "RF", "sequences": labels65[:33]}]
        # ... and one graph with 257 labels: beyond the device's label sets (4 words of 64 bits), the one site that is reported
        labels257 = ["M%03d" % k for k in range(257)]
        xnodes = [{"name": "LF", "reference": "chr1:%d-%d" % (spacing * 4 - flank + 1, spacing * 4)},
                  {"name": "MID", "sequence": "ACGTTGCAACGTACGT"},
                  {"name": "RF", "reference": "chr1:%d-%d" % (spacing * 4 + 1, spacing * 4 + flank)}]
        xedges = [{"from": "LF", "to": "MID", "sequences": labels257[:129]}, {"from": "LF", "to": "RF", "sequences": labels257[129:]},
                  {"from": "MID", "to": "RF", "sequences": labels257[:129]}]
        with open(os.path.join(out, "extra_graphs.txt"), "w") as xf:
            for name, nodes_x, edges_x, site, seqnames in (("many_nodes", chain, cedges, 1, ["ALT", "REF"]), ("many_columns", wide, wedges, 2, ["ALT", "REF"]),
                                                           ("many_labels", lnodes, ledges, 3, labels65), ("too_many_labels", xnodes, xedges, 4, labels257)):
                gp = os.path.join(out, "graphs", name + ".json")
                with open(gp, "w") as f:
                    json.dump({"ID": name, "nodes": nodes_x, "edges": edges_x, "sequencenames": seqnames,
                               "target_regions": ["chr1:%d-%d" % (spacing * site - flank + 1, spacing * site + flank)]}, f)
                xf.write(gp + "\n")
    records.sort(key=lambda r: r["pos"])
    bamwriter.write_bam(os.path.join(out, "reads.bam"), [("chr1", glen)], records)
    with open(os.path.join(out, "graphs.txt"), "w") as f:
        f.write("\n".join(graph_paths) + "\n")
    with open(os.path.join(out, "manifest.txt"), "w") as f:
        f.write("id\tpath\tdepth\tread length\nSYN\t%s\t%g\t%d\n" % (os.path.join(out, "reads.bam"), depth, read_len))
    with open(os.path.join(out, "truth.json"), "w") as f:
        json.dump(truth, f)
    print("wrote %d sites, %d reads, %d bp reference to %s" % (n_sites, len(records), glen, out))


if __name__ == "__main__":
    main()
