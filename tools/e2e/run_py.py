"""End-to-end probe through the Python entry (paragraph_amd.workflow -> pgw_genotype_graphs), for option sweeps:

    python tools/e2e/run_py.py <data dir made by make_sites.py> [key=value ...]

e.g. path_sequence_matching=1 kmer_sequence_matching=1 threads=32 lanes=8.  Prints wall clock, sites/s and the concordance
of the genotypes with the simulated truth."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from paragraph_amd import workflow  # noqa: E402


def main():
    data = sys.argv[1]
    options = {}
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        options[k] = (v not in ("0", "false")) if k.endswith("matching") or k == "packed_reads" else (float(v) if "." in v else int(v))
    graphs = [l.strip() for l in open(os.path.join(data, "graphs.txt")) if l.strip()]
    truth = {t["ID"]: t["gt"] for t in json.load(open(os.path.join(data, "truth.json")))}
    out = {}
    for rep in range(2):
        t0 = time.time()
        docs = workflow.genotype_graphs(os.path.join(data, "ref.fa"), os.path.join(data, "manifest.txt"), graphs, **options)
        out["run%d_s" % rep] = round(time.time() - t0, 3)
    ok = sum(1 for d in docs if d["samples"]["SYN"]["gt"]["GT"] == truth[d["graphinfo"]["ID"]])
    out.update(options=options, sites=len(docs), sites_per_s=round(len(docs) / out["run1_s"]), concordant=ok)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
