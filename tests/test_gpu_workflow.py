"""End-to-end host workflow on the GPU: BAM -> read extraction -> one batched realignment over all sites -> count
documents -> genotypes, checked against the reference's own expected outputs (tests/golden/sites/README.md):
`paragraph` count documents of share/test-data/multiparagraph and the genotypes of Grmpy.GenotypesSingleSwap."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_workflow_against_reference_outputs():
    from paragraph_amd import build
    exe = os.path.join(ROOT, "tests", "host_cpp", "test_workflow")
    if not os.path.exists(exe):
        build.build_host()
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "sites")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "workflow: all checks passed" in out.stdout


def test_python_entry_genotypes_single_swap(tmp_path):
    """paragraph_amd.workflow.genotype_graphs (ctypes -> pgw_genotype_graphs in libparagraph_host.so) on the fixture of
    Grmpy.GenotypesSingleSwap: male sample REF, female sample REF/REF (src/c++/test-blackbox/test_grm.cpp:60-62)."""
    from paragraph_amd import workflow
    sites = os.path.join(ROOT, "tests", "golden", "sites", "chrX")
    bam = os.path.join(sites, "chrX_graph_typing.bam")
    manifest = tmp_path / "manifest.txt"
    manifest.write_text("#id\tpath\tdepth\tread length\tdepth sd\tsex\nSAMPLE1\t%s\t44.2\t150\t20\tmale\nSAMPLE2\t%s\t44.2\t150\t20\tfemale\n"
                        % (bam, bam))
    graph = os.path.join(sites, "chrX_graph_typing.2sample.json")
    out = tmp_path / "genotypes.json"
    docs = workflow.genotype_graphs(os.path.join(sites, "chrX_graph_typing.fa"), str(manifest), [graph, graph],
                                    genotyping_parameters=os.path.join(sites, "param.json"), output_path=str(out), threads=4, lanes=2,
                                    sites_per_batch=2)
    assert len(docs) == 2 and docs[0] == docs[1] and out.exists()
    assert docs[0]["samples"]["SAMPLE1"]["gt"]["GT"] == "REF"
    assert docs[0]["samples"]["SAMPLE2"]["gt"]["GT"] == "REF/REF"
    assert docs[0]["graphinfo"]["ID"] == "chrX_graph_typing" and "population" in docs[0]
    # the object form gives the same documents
    again = workflow.genotype_graphs(os.path.join(sites, "chrX_graph_typing.fa"), str(manifest), [graph],
                                     genotyping_parameters=os.path.join(sites, "param.json"), packed_reads=False)
    assert again[0] == docs[0]


def test_grmpy_binary_as_multigrmpy_calls_it(tmp_path):
    """paragraph_amd/bin/grmpy driven the way src/python/bin/multigrmpy.py:262-315 drives the reference's binary: options in
    a response file, graphs one per line after -g, gzip-compressed JSON array out; and -O with one file per graph."""
    import gzip
    import json
    from paragraph_amd import build
    if not os.path.exists(build.GRMPY_BIN):
        build.build_host()
    sites = os.path.join(ROOT, "tests", "golden", "sites", "chrX")
    bam = os.path.join(sites, "chrX_graph_typing.bam")
    manifest = tmp_path / "manifest.txt"
    manifest.write_text("#id\tpath\tdepth\tread length\tdepth sd\tsex\nSAMPLE1\t%s\t44.2\t150\t20\tmale\nSAMPLE2\t%s\t44.2\t150\t20\tfemale\n"
                        % (bam, bam))
    graph = os.path.join(sites, "chrX_graph_typing.2sample.json")
    out = tmp_path / "genotypes.json.gz"
    response = tmp_path / "response.txt"
    response.write_text(" -r %s -m %s -o %s -z -G %s -M 10000 -t 4 --graph-sequence-matching True --log-level=warning --log-file %s --log-async no -g\n%s\n%s"
                        % (os.path.join(sites, "chrX_graph_typing.fa"), manifest, out, os.path.join(sites, "param.json"), tmp_path / "grmpy.log",
                           graph, graph))
    r = subprocess.run([build.GRMPY_BIN, "--response-file=%s" % response], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    docs = json.loads(gzip.open(out, "rt").read())
    assert isinstance(docs, list) and len(docs) == 2 and docs[0] == docs[1]
    assert docs[0]["samples"]["SAMPLE1"]["gt"]["GT"] == "REF" and docs[0]["samples"]["SAMPLE2"]["gt"]["GT"] == "REF/REF"
    # one graph, stdout, plain: a single document (not an array), as the original writes it
    r = subprocess.run([build.GRMPY_BIN, "-r", os.path.join(sites, "chrX_graph_typing.fa"), "-m", str(manifest), "-g", graph, "-G",
                        os.path.join(sites, "param.json")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout) == docs[0]
    folder = tmp_path / "per_graph"
    folder.mkdir()
    r = subprocess.run([build.GRMPY_BIN, "-r", os.path.join(sites, "chrX_graph_typing.fa"), "-m", str(manifest), "-g", graph, "-G",
                        os.path.join(sites, "param.json"), "-O", str(folder), "-z"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout == "", r.stderr
    assert json.loads(gzip.open(folder / "chrX_graph_typing.2sample.json.gz", "rt").read()) == docs[0]


def test_config1_round_trip_genotyping(tmp_path):
    """BASELINE configs[0]: share/test-data/round-trip-genotyping (two samples with a handful of 50 bp reads, an insertion and
    a one-base deletion at chr1:161).  The reference's Python round trip ends in expected-vcf-record.txt; the same GT / DP /
    AD / PL and the no-call filter come out of graph_templates -> workflow.genotype_graphs for the two records' events."""
    import json
    from paragraph_amd import graph_templates, workflow
    d = os.path.join(ROOT, "tests", "golden", "sites", "round-trip")
    records = {}
    for line in open(os.path.join(d, "expected-vcf-record.txt")):
        f = line.rstrip("\n").split("\t")
        if line.startswith("#"):
            names = f[9:]
            continue
        keys = f[8].split(":")
        records[f[2]] = dict(ref=f[3], alt=f[4], pos=int(f[1]), samples={n: dict(zip(keys, v.split(":"))) for n, v in zip(names, f[9:])})
    assert set(records) == {"test-ins", "test-del"}
    graphs, alt_label = [], {}
    for rid, rec in sorted(records.items()):
        # VCF record -> event: the padding base stays, REF[1:] is deleted / ALT[1:] inserted after it
        deleted, inserted = rec["ref"][1:], rec["alt"][1:]
        event = {"chrom": "chr1", "start": rec["pos"] + 1, "end": rec["pos"] + len(deleted)}
        if inserted:
            event["ins"] = inserted
        kind, graph = graph_templates.make_graph(event)
        alt_label[rid] = "INS" if inserted else "DEL"
        graph["ID"] = rid
        path = tmp_path / (rid + ".json")
        path.write_text(json.dumps(graph))
        graphs.append(str(path))
    docs = {doc["graphinfo"]["ID"]: doc for doc in
            workflow.genotype_graphs(os.path.join(d, "dummy.fa"), os.path.join(d, "samples.txt"), graphs, threads=2)}
    for rid, rec in records.items():
        alt = alt_label[rid]
        for sample, want in rec["samples"].items():
            got = docs[rid]["samples"][sample]
            gt = got["gt"]
            if want["GT"] == ".":
                assert gt["GT"] == "." and "NO_VALID_GT" in gt["filters"] and "NO_VALID_GT" in want["FT"], (rid, sample, gt)
                assert int(want["DP"]) == 0 and "num_reads" not in gt
                continue
            assert want["GT"] == "1/1" and gt["GT"] == "%s/%s" % (alt, alt), (rid, sample, gt)
            assert gt["num_reads"] == int(want["DP"])
            ref_ad, alt_ad = (int(x) for x in want["AD"].split(","))
            for bp in got["breakpoints"].values():
                assert bp["counts"]["alleles"]["REF"] == ref_ad and bp["counts"]["alleles"][alt] == alt_ad, (rid, sample, bp["counts"])
            # PL as vcfupdate.py:284-310 derives it: round(-10 x log10 likelihood) minus the smallest, order REF/REF, REF/ALT, ALT/ALT
            gl = gt["GL"]
            order = ["REF/REF", "%s/REF" % alt if "%s/REF" % alt in gl else "REF/%s" % alt, "%s/%s" % (alt, alt)]
            phred = [round(-10 * gl[k]) for k in order]
            assert [str(x - min(phred)) for x in phred] == want["PL"].split(","), (rid, sample, gl, want["PL"])
            if want["FT"] == "PASS":
                assert gt["filters"] == ["PASS"]


def test_synthetic_sites_packed_equals_objects_and_truth(tmp_path):
    """60 simulated del / ins / swap sites with paired 150 bp reads (tools/e2e/make_sites.py), two samples (30x and 12x):
    every genotype document is the same from packed reads and from read objects, with and without the path stage first and
    with the KmerFilter, and the 30x sample's genotypes equal the simulated truth."""
    import json
    import sys
    from paragraph_amd import workflow
    data = tmp_path / "sites"
    for depth, name in ((30, "hi"), (12, "lo")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e", "make_sites.py"), str(data / name), "60", str(depth), "5"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
    graphs = [l.strip() for l in open(data / "hi" / "graphs.txt") if l.strip()]
    manifest = tmp_path / "manifest.txt"
    manifest.write_text("id\tpath\tdepth\tread length\nSYN\t%s\t30\t150\nLOW\t%s\t12\t150\n" % (data / "hi" / "reads.bam", data / "lo" / "reads.bam"))
    truth = {t["ID"]: t["gt"] for t in json.load(open(data / "hi" / "truth.json"))}
    ref = str(data / "hi" / "ref.fa")  # same seed: both data sets share reference, sites and genotypes
    assert open(data / "lo" / "ref.fa").read() == open(ref).read()
    base = None
    for options in ({}, {"path_sequence_matching": True}, {"bad_align_uniq_kmer_len": -1}):
        packed = workflow.genotype_graphs(ref, str(manifest), graphs, threads=4, lanes=2, sites_per_batch=32, **options)
        objects = workflow.genotype_graphs(ref, str(manifest), graphs, threads=4, lanes=2, sites_per_batch=50, packed_reads=False, **options)
        assert packed == objects, options
        good = sum(1 for d in packed if d["samples"]["SYN"]["gt"]["GT"] == truth[d["graphinfo"]["ID"]])
        assert good >= 58, (options, good)
        assert all("population" in d for d in packed)
        assert sum(d["samples"]["SYN"]["paired_read"] for d in packed) > len(packed)  # most mates fall off these small graphs
        if base is None:
            base = packed
    # the exact shortcut in front of gssw (reads with one exact full-length match skip their fills): the SAME documents as the
    # plain gssw cascade, statistics and all, from packed reads and from read objects
    for packed_reads in (True, False):
        assert workflow.genotype_graphs(ref, str(manifest), graphs, threads=4, lanes=2, sites_per_batch=32, packed_reads=packed_reads,
                                        exact_match_shortcut=True) == base
    # fragment statistics see real pairs here: the running median / variance are exercised beyond the seed samples
    rich = [d["samples"]["SYN"] for d in base if d["samples"]["SYN"]["paired_read"] >= 5]
    assert rich and all(s["median_graph"] > 100 and s["variance_graph"] > 0 and s["mean_graph"] > 100 for s in rich)


def test_paragraph_binary_reproduces_multiparagraph(tmp_path):
    """The chain of src/python/bin/multiparagraph.py on its own test data: candidates.json -> event templates -> `paragraph`
    (here paragraph_amd/bin/paragraph, all five graphs in one call) -> the graph part of expected.json, key for key
    (src/python/test/test_multiparagraph.py:83-105 removes bam / reference / alignment_statistics before comparing)."""
    import gzip
    import json
    import math
    from paragraph_amd import build, graph_templates
    if not os.path.exists(build.PARAGRAPH_BIN):
        build.build_host()
    d = os.path.join(ROOT, "tests", "golden", "sites", "multiparagraph")
    expected = json.load(open(os.path.join(d, "expected.json")))
    graphs = []
    for i, event in enumerate(json.load(open(os.path.join(d, "candidates.json")))):
        kind, graph = graph_templates.make_graph({k: v for k, v in event.items() if k != "desc"})
        assert kind == expected[i]["type"]
        path = tmp_path / ("event_%d.json" % i)
        path.write_text(json.dumps(graph))
        graphs.append(str(path))
    out = tmp_path / "out.json.gz"
    r = subprocess.run([build.PARAGRAPH_BIN, "-r", os.path.join(d, "dummy.fa"), "-b", os.path.join(d, "reads.bam"), "-o", str(out), "-z",
                        "--threads", "2", "-g"] + graphs, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    observed = json.loads(gzip.open(out, "rt").read())
    assert len(observed) == len(expected)

    def same(a, b):
        if isinstance(a, float) or isinstance(b, float):
            return (a is None and b is None) or (a is not None and b is not None and math.isclose(a, b, rel_tol=1e-12, abs_tol=0))
        if isinstance(a, dict):
            return isinstance(b, dict) and a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
        if isinstance(a, list):
            return isinstance(b, list) and len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        return a == b

    for want, got in zip(expected, observed):
        assert got["bam"] == os.path.join(d, "reads.bam") and got["reference"] == os.path.join(d, "dummy.fa")
        for key in ("bam", "reference", "alignment_statistics"):
            got.pop(key)
        assert same(want["graph"], got), (want["desc"], json.dumps(want["graph"], sort_keys=True)[:400], json.dumps(got, sort_keys=True)[:400])
    # two BAMs: pooled per graph, "bam" lists both, twice the fragments ... of which the same-named ones merge
    r = subprocess.run([build.PARAGRAPH_BIN, "-r", os.path.join(d, "dummy.fa"), "-b", os.path.join(d, "reads.bam"), os.path.join(d, "reads.bam"),
                        "-g", graphs[0], "--output-detailed-read-counts"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    joint = json.loads(r.stdout)
    assert joint["bam"] == [os.path.join(d, "reads.bam")] * 2
    assert joint["read_counts_by_edge"]["LF_MID"] == expected[0]["graph"]["read_counts_by_edge"]["LF_MID"]          # fragments
    assert joint["read_counts_by_edge"]["LF_MID:READS"] == 2 * expected[0]["graph"]["read_counts_by_edge"]["LF_MID:READS"]  # reads
    assert "LF_MID" in joint["read_counts_by_sequence"]["REF"]
    # -T overrides the graphs' target regions: a window without reads gives empty tables (and a batch without any read
    # goes through the device path without complaint); a window over the reads gives the usual ones
    for window, expect_reads in (("chr:150-160", False), ("chr:1-160", True)):
        r = subprocess.run([build.PARAGRAPH_BIN, "-r", os.path.join(d, "dummy.fa"), "-b", os.path.join(d, "reads.bam"), "-g", graphs[0], "-T", window],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        doc = json.loads(r.stdout)
        assert doc["target_regions"] == expected[0]["graph"]["target_regions"]  # the description is echoed as given
        assert bool(doc["read_counts_by_edge"]) == expect_reads, (window, doc["read_counts_by_edge"])
        if expect_reads:
            assert doc["read_counts_by_edge"] == expected[0]["graph"]["read_counts_by_edge"]
        else:
            assert doc["fragment_statistics"]["single_read"] == 0 and doc["read_counts_by_node"] == {}


def test_swaps_600x_genotypes(tmp_path):
    """share/test-data/genotyping_test_2: three sequence swaps simulated as 0/0, 1/1 and 0/1 (GT column of swaps.vcf) at
    600x, graphs as vcf2paragraph wrote them (chrA/B/C.json), through bin/grmpy: the simulated genotypes come back, PASS,
    as in the reference's expected-genotypes.vcf."""
    import json
    from paragraph_amd import build
    if not os.path.exists(build.GRMPY_BIN):
        build.build_host()
    d = os.path.join(ROOT, "tests", "golden", "sites", "swaps")
    truth = {}
    for line in open(os.path.join(d, "expected-genotypes.vcf")):
        if not line.startswith("#"):
            f = line.split("\t")
            truth[f[0]] = f[9].split(":")[0]
    assert truth == {"chrA": "0/0", "chrB": "1/1", "chrC": "0/1"}
    graphs = [os.path.join(d, "chr%s.json" % c) for c in "ABC"]
    out = tmp_path / "genotypes.json"
    r = subprocess.run([build.GRMPY_BIN, "-r", os.path.join(d, "swaps.fa"), "-m", os.path.join(d, "samples.txt"), "-o", str(out), "-t", "4", "-g"] + graphs,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    docs = json.load(open(out))
    assert len(docs) == 3
    for doc in docs:
        chrom = doc["graphinfo"]["target_regions"][0].split(":")[0]
        gt = doc["samples"]["SWAPS"]["gt"]
        alleles = gt["GT"].split("/")
        called = "/".join(sorted("0" if a == "REF" else "1" for a in alleles))
        assert called == truth[chrom], (chrom, gt)
        assert gt["filters"] == ["PASS"] and gt["num_reads"] > 800, (chrom, gt)
        # every breakpoint agrees with the site call
        assert all(bp["gt"]["GT"] == gt["GT"] for bp in doc["samples"]["SWAPS"]["breakpoints"].values()), chrom


def test_two_device_slots_give_the_same_documents(tmp_path):
    """Lanes spread over a device list (PG_DEVICES / the "devices" option): on a 1-GPU box the list names the GPU twice = two
    contexts (own streams, workspace, stage mutex, batch pool) on one device.  Six copies of the chrX graph in chunks of one
    graph, two lanes -> lane 0 on slot 0, lane 1 on slot 1; the documents equal the single-device ones.  The batch staging
    arrays come from the page-locked pool (pg_host_alloc)."""
    import json
    import sys
    sites = os.path.join(ROOT, "tests", "golden", "sites", "chrX")
    bam = os.path.join(sites, "chrX_graph_typing.bam")
    manifest = tmp_path / "manifest.txt"
    manifest.write_text("#id\tpath\tdepth\tread length\tdepth sd\tsex\nSAMPLE1\t%s\t44.2\t150\t20\tmale\nSAMPLE2\t%s\t44.2\t150\t20\tfemale\n"
                        % (bam, bam))
    graph = os.path.join(sites, "chrX_graph_typing.2sample.json")
    script = tmp_path / "run.py"
    script.write_text(
        "import json, sys\n"
        "sys.path.insert(0, %r)\n"
        "from paragraph_amd import workflow\n"
        "opts = json.loads(sys.argv[1])\n"
        "docs = workflow.genotype_graphs(%r, %r, [%r] * 6, genotyping_parameters=%r, threads=8, sites_per_batch=2, **opts)\n"
        "print(json.dumps(docs))\n" % (ROOT, os.path.join(sites, "chrX_graph_typing.fa"), str(manifest), graph, os.path.join(sites, "param.json")))
    env = dict(os.environ, PG_BATCH_TIMING="1")
    env.pop("PG_DEVICES", None)
    outs = []
    for opts in ({"lanes": 2, "devices": [0, 0]}, {"lanes": 1}):
        r = subprocess.run([sys.executable, str(script), json.dumps(opts)], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(json.loads(r.stdout.splitlines()[-1]))
    assert len(outs[0]) == 6 and outs[0] == outs[1]
    assert outs[0][0]["samples"]["SAMPLE1"]["gt"]["GT"] == "REF" and outs[0][0]["samples"]["SAMPLE2"]["gt"]["GT"] == "REF/REF"
    # the same through the environment (what bin/grmpy --devices all resolves to)
    r = subprocess.run([sys.executable, str(script), json.dumps({"lanes": 2})], capture_output=True, text=True, timeout=600,
                       env=dict(env, PG_DEVICES="0,0"))
    assert r.returncode == 0 and json.loads(r.stdout.splitlines()[-1]) == outs[1], r.stderr[-2000:]


def test_eight_device_slots_give_the_same_documents(tmp_path):
    """The device list of an 8-GPU node on the 1-GPU box: eight slots that all name device 0 = eight contexts (own streams,
    workspace, batch pool, stage mutex), sixteen lanes spread over them round-robin, 24 copies of the chrX graph in chunks of one
    graph so that every slot gets batches.  The documents equal the one-slot run's, graph for graph, and every slot was used
    (PG_BATCH_TIMING names the context a batch ran on)."""
    import json
    import sys
    sites = os.path.join(ROOT, "tests", "golden", "sites", "chrX")
    bam = os.path.join(sites, "chrX_graph_typing.bam")
    manifest = tmp_path / "manifest.txt"
    manifest.write_text("#id\tpath\tdepth\tread length\tdepth sd\tsex\nSAMPLE1\t%s\t44.2\t150\t20\tmale\nSAMPLE2\t%s\t44.2\t150\t20\tfemale\n"
                        % (bam, bam))
    graph = os.path.join(sites, "chrX_graph_typing.2sample.json")
    script = tmp_path / "run.py"
    script.write_text(
        "import json, sys\n"
        "sys.path.insert(0, %r)\n"
        "from paragraph_amd import workflow\n"
        "opts = json.loads(sys.argv[1])\n"
        "docs = workflow.genotype_graphs(%r, %r, [%r] * 24, genotyping_parameters=%r, threads=8, sites_per_batch=2, **opts)\n"
        "print(json.dumps(docs))\n" % (ROOT, os.path.join(sites, "chrX_graph_typing.fa"), str(manifest), graph, os.path.join(sites, "param.json")))
    env = dict(os.environ)
    env.pop("PG_DEVICES", None)
    outs = []
    for opts in ({"lanes": 16, "devices": [0] * 8}, {"lanes": 1}):
        r = subprocess.run([sys.executable, str(script), json.dumps(opts)], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(json.loads(r.stdout.splitlines()[-1]))
    assert len(outs[0]) == 24 and outs[0] == outs[1]
    assert outs[0][0]["samples"]["SAMPLE1"]["gt"]["GT"] == "REF" and outs[0][0]["samples"]["SAMPLE2"]["gt"]["GT"] == "REF/REF"
    # the same through the environment (what bin/grmpy --devices all resolves to on an 8-GPU node)
    r = subprocess.run([sys.executable, str(script), json.dumps({"lanes": 16})], capture_output=True, text=True, timeout=900,
                       env=dict(env, PG_DEVICES="0,0,0,0,0,0,0,0"))
    assert r.returncode == 0 and json.loads(r.stdout.splitlines()[-1]) == outs[1], r.stderr[-2000:]


def test_config1_round_trip_from_the_vcf(tmp_path):
    """BASELINE configs[0] with the VCF itself as input, as `multigrmpy.py -i candidates.vcf` takes it: paragraph_amd.vcf2paragraph
    (one allele graph per line, the reference's vcf2paragraph options) -> workflow.genotype_graphs -> the GT / DP / AD of
    expected-vcf-record.txt.  Allele names are the variant's allele ids (`test-ins:1`); the REF allele carries the `REF` path label
    and `<id>:0` on the same edges."""
    import json
    from paragraph_amd import vcf2paragraph, workflow
    d = os.path.join(ROOT, "tests", "golden", "sites", "round-trip")
    want = {}
    for line in open(os.path.join(d, "expected-vcf-record.txt")):
        f = line.rstrip("\n").split("\t")
        if line.startswith("#"):
            names = f[9:]
            continue
        keys = f[8].split(":")
        want[f[2]] = {n: dict(zip(keys, v.split(":"))) for n, v in zip(names, f[9:])}
    graphs = vcf2paragraph.convert_vcf_to_graphs(os.path.join(d, "candidates.vcf"), os.path.join(d, "dummy.fa"), retrieve_reference_sequence=True)
    paths = []
    for g in graphs:
        doc = dict(g["graph"], ID=g["ID"])
        p = tmp_path / ("g%d.json" % len(paths))
        p.write_text(json.dumps(doc))
        paths.append(str(p))
    docs = workflow.genotype_graphs(os.path.join(d, "dummy.fa"), os.path.join(d, "samples.txt"), paths, threads=2)
    assert [doc["graphinfo"]["ID"] for doc in docs] == [g["ID"] for g in graphs]
    for rid, doc in zip(("test-ins", "test-del"), docs):
        for sample, w in want[rid].items():
            gt = doc["samples"][sample]["gt"]
            if w["GT"] == ".":
                assert gt["GT"] == "." and "NO_VALID_GT" in gt["filters"], (rid, sample, gt)
                continue
            alleles = gt["GT"].split("/")
            assert len(alleles) == 2 and all(a.endswith(rid + ":1") or (rid + ":1") in a.split(",") for a in alleles), (rid, sample, gt)  # 1/1
            assert gt["num_reads"] == int(w["DP"]), (rid, sample, gt, w)


def test_config1_multigrmpy_vcf_to_vcf(tmp_path):
    """BASELINE configs[0] through the reference's entry point (README: `multigrmpy.py -i candidates.vcf -m samples.txt -r dummy.fa
    -o test`, "the last 3 lines of genotypes.vcf.gz will be the same as in expected-vcf-record.txt"): VCF -> graphs -> bin/grmpy
    on the device -> genotypes.json.gz -> genotypes.vcf.gz.  One documented difference (paragraph_amd/multigrmpy.py): the
    reference's writer shows sample1's FT of the first record as dots, here it is the genotype's filter."""
    import gzip
    import json
    from paragraph_amd import multigrmpy
    d = os.path.join(ROOT, "tests", "golden", "sites", "round-trip")
    out = tmp_path / "out"
    args = multigrmpy.make_argument_parser().parse_args([
        "-i", os.path.join(d, "candidates.vcf"), "-m", os.path.join(d, "samples.txt"), "-r", os.path.join(d, "dummy.fa"),
        "-o", str(out), "-t", "2", "-M", "10000", "--graph-sequence-matching", "1", "-l", "1000", "--scratch-dir", str(tmp_path / "scratch")])
    cwd = os.getcwd()
    os.chdir(d)  # the manifest names its BAM files relative to itself
    try:
        stats = multigrmpy.run(args)
    finally:
        os.chdir(cwd)
    assert stats == {"matched": 2, "unmatched": 0, "multimatched": 0}
    assert sorted(os.listdir(out)) == ["genotypes.json.gz", "genotypes.vcf.gz", "grmpy.log", "variants.json.gz", "variants.vcf.gz"] or \
        sorted(os.listdir(out)) == ["genotypes.json.gz", "genotypes.vcf.gz", "variants.json.gz", "variants.vcf.gz"]
    got = [l.rstrip("\n") for l in gzip.open(out / "genotypes.vcf.gz", "rt")][-3:]
    want = [l.rstrip("\n") for l in open(os.path.join(d, "expected-vcf-record.txt"))]
    first = want[1].split("\t")
    assert first[9].split(":")[2] == "...."
    ft = got[1].split("\t")[9].split(":")[2]
    assert ft in ("PASS", "GQ")
    want[1] = want[1].replace("1/1:2:....:", "1/1:2:%s:" % ft)
    assert got == want
    with gzip.open(out / "genotypes.json.gz", "rt") as f:
        docs = {doc["graphinfo"]["ID"]: doc for doc in json.load(f)}
    # the reference's own check of this run (src/python/test/test_multigrmpy.py:100-108) names graphs by event id; with a VCF
    # input the graph id is the GRMPY_ID and the called allele the record's first ALT
    for line in got[1:]:
        f = line.split("\t")
        doc = docs[f[7].split("=", 1)[1]]
        called = "sample1" if f[2] == "test-ins" else "sample2"
        other = "sample2" if called == "sample1" else "sample1"
        assert doc["samples"][called]["gt"]["GT"] == "%s:1/%s:1" % (f[2], f[2])
        assert doc["samples"][other]["gt"]["GT"] == "."
    assert not os.listdir(tmp_path / "scratch")


def test_config1_multigrmpy_from_json_events(tmp_path):
    """The other input form of the reference's entry point (multigrmpy.py:68-88): a JSON list of events without graphs; each gets
    its graph from graph_templates.make_graph and keeps its own ID."""
    import gzip
    import json
    from paragraph_amd import multigrmpy
    d = os.path.join(ROOT, "tests", "golden", "sites", "round-trip")
    out = tmp_path / "out"
    args = multigrmpy.make_argument_parser().parse_args([
        "-i", os.path.join(d, "candidates.json"), "-m", os.path.join(d, "samples.txt"), "-r", os.path.join(d, "dummy.fa"),
        "-o", str(out), "-t", "2", "-M", "10000", "--scratch-dir", str(tmp_path / "scratch")])
    cwd = os.getcwd()
    os.chdir(d)
    try:
        assert multigrmpy.run(args) is None  # no VCF to write back into
    finally:
        os.chdir(cwd)
    with gzip.open(out / "genotypes.json.gz", "rt") as f:
        docs = {doc["graphinfo"]["ID"]: doc for doc in json.load(f)}
    assert set(docs) == {"test-ins", "test-del"}
    # (the events of candidates.json are not the variants of candidates.vcf -- a 2-base swap and a 3-base deletion -- so no
    # particular call is expected from the handful of reads; the run has to go through and give complete documents)
    for rid, doc in docs.items():
        assert doc["graphinfo"]["sequencenames"][0] == "REF"
        for sample in ("sample1", "sample2"):
            gt = doc["samples"][sample]["gt"]
            assert isinstance(gt["GT"], str) and gt["filters"], (rid, sample, gt)
            assert "alleles" in doc["samples"][sample] and "breakpoints" in doc["samples"][sample]
    assert not os.path.exists(out / "genotypes.vcf.gz")


def test_one_oversize_site_does_not_take_the_run_down(tmp_path):
    """202 graphs of which five are outside the packed kernels' envelope -- a 600 bp read over one site, a 5 000-node graph, a
    70 000-column graph, a graph with 65 sequence labels, a graph with 257; the reference has no such bounds (gssw.c:527-786,
    GraphAligner.cpp:110-167, ReadCounting.cpp:96-127).  The long read, the many nodes and the many columns go through the
    general path (pg_general.hip; parity with the reference's gssw.c: tests/test_gpu_general.py), the 65 labels through the wide
    label sets (two words; parity with the reference's counting code: tests/test_gpu_counts.py::
    test_more_than_64_labels_on_a_graph) and come out as ordinary documents; the 257-label graph says why it was skipped under
    "error"; the 197 other genotype documents equal those of a run over the 197 alone."""
    import sys
    from paragraph_amd import workflow
    data = tmp_path / "sites"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e", "make_sites.py"), str(data), "198", "20", "7", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    graphs = [l.strip() for l in open(data / "graphs.txt") if l.strip()]
    extra = [l.strip() for l in open(data / "extra_graphs.txt") if l.strip()]
    assert len(graphs) == 198 and len(extra) == 4
    ref, manifest = str(data / "ref.fa"), str(data / "manifest.txt")
    # the odd ones in the middle of batches
    everything = graphs[:50] + [extra[0]] + graphs[50:90] + [extra[3]] + graphs[90:120] + [extra[1]] + graphs[120:160] + [extra[2]] + graphs[160:]
    docs = workflow.genotype_graphs(ref, manifest, everything, threads=4, lanes=2, sites_per_batch=32)
    assert len(docs) == 202
    by_id = {d["graphinfo"]["ID"]: d for d in docs}
    bad = {"too_many_labels": "256"}
    for gid, why in bad.items():
        assert "error" in by_id[gid] and why in by_id[gid]["error"], (gid, by_id[gid].get("error"))
    for gid in ("site_197", "many_columns", "many_nodes", "many_labels"):  # ordinary documents with reads counted
        assert "error" not in by_id[gid], by_id[gid].get("error")
        assert by_id[gid]["samples"]["SYN"]["gt"]["num_reads"] > 0, by_id[gid]["samples"]["SYN"]["gt"]
    assert len(by_id["many_labels"]["graphinfo"]["sequencenames"]) == 65
    good = workflow.genotype_graphs(ref, manifest, graphs[:197], threads=4, lanes=2, sites_per_batch=32)
    assert len(good) == 197 and not any("error" in d for d in good)
    for d in good:
        assert by_id[d["graphinfo"]["ID"]] == d, d["graphinfo"]["ID"]
    # the object form isolates the same way, and counts the 65-label graph the same way
    span = everything[85:100] + [extra[2]]
    objects = workflow.genotype_graphs(ref, manifest, span, threads=2, lanes=1, sites_per_batch=16, packed_reads=False)
    assert [("error" in d) for d in objects] == [d["graphinfo"]["ID"] in bad for d in objects] and sum("error" in d for d in objects) == 1
    for d in objects:
        if "error" not in d:
            assert by_id[d["graphinfo"]["ID"]] == d, d["graphinfo"]["ID"]


def _graphs_of_expected_genotypes(expected, folder):
    """The graphs the reference's expected-genotypes.json was computed on, rebuilt from its own `graphinfo`: node names say what
    a node is ("ref-<chrom>:<a>-<b>" = that reference interval, "<chrom>:<a>-<b>:<SEQ>" = inline sequence, source / sink = N x 10),
    edge names are "<from>_<to>" and carry the sequence labels."""
    import json
    paths = []
    for k, doc in enumerate(expected):
        info = doc["graphinfo"]
        nodes = []
        for n in info["nodes"]:
            name = n["name"]
            if name in ("source", "sink"):
                nodes.append({"name": name, "sequence": "N" * 10})
            elif name.startswith("ref-"):
                nodes.append({"name": name, "reference": name[4:]})
            else:
                nodes.append({"name": name, "sequence": name.rsplit(":", 1)[1]})
        edges = []
        for e in info["edges"]:
            a, b = e["name"].split("_")
            edge = {"from": a, "to": b}
            if "sequences" in e:
                edge["sequences"] = e["sequences"]
            edges.append(edge)
        path = os.path.join(str(folder), "expected_graph_%d.json" % k)
        with open(path, "w") as f:
            json.dump({"ID": info["ID"], "nodes": nodes, "edges": edges, "sequencenames": info["sequencenames"],
                       "target_regions": info["target_regions"]}, f)
        paths.append(path)
    return paths


def test_swaps_statistics_equal_the_references_expected_genotypes(tmp_path):
    """share/test-data/genotyping_test_2/expected-genotypes.json is a genotype document the REFERENCE wrote for swaps.bam (three
    swaps at 600x): besides GT / GL it holds, per sample, `alignment_statistics` (per node / edge / allele: reads by strand,
    match-base depth, mismatch / gap / clip rates, average score, contig length) and `fragment_statistics` -- the only values
    of src/c++/lib/paragraph/AlignmentStatistics.cpp the reference's data holds.  Same graphs (rebuilt from the file's own
    graphinfo), same BAM -> the per-allele statistics, the fragment counts, every breakpoint's edge / allele counts, the
    genotype likelihoods and the depth-test p-values equal at the 5 significant digits the file was written with.  The file
    is older than the reference checkout in four visible ways (a `filter` string where today's documents have a `filters`
    list, `allele_fractions` as a list, node / edge statistics keyed differently, the Poisson depth test as the default);
    what those touch is compared in today's form or, for the node / edge blocks, left out.  (The graph-fragment-length
    moments differed at first: an off-by-one in the restated path end, found by this file -- DESIGN 7.3.)"""
    import json
    import math
    from paragraph_amd import workflow
    d = os.path.join(ROOT, "tests", "golden", "sites", "swaps")
    expected = json.load(open(os.path.join(d, "expected-genotypes.json")))
    graphs = _graphs_of_expected_genotypes(expected, tmp_path)
    # the file's coverage-test p-values are those of the Poisson depth test (BreakpointGenotyper.cpp:163-166), today an option
    parameters = tmp_path / "genotyping.json"
    parameters.write_text(json.dumps({"use_poisson_depth": "true"}))
    docs = workflow.genotype_graphs(os.path.join(d, "swaps.fa"), os.path.join(d, "samples.txt"), graphs, threads=2,
                                    genotyping_parameters=str(parameters))
    assert len(docs) == 3
    diffs = []

    def same(a, b, where):
        if isinstance(a, list) and isinstance(b, dict):
            # allele_fractions: a list in allele order in the file, keyed by allele name today (Genotype.cpp: "output allele
            # name instead of indexes"); two alleles here, REF first either way
            b = [b[k] for k in sorted(b)]
        if isinstance(a, dict):
            if not (isinstance(b, dict) and set(a) == set(b)):
                diffs.append((where, "keys", sorted(a), sorted(b) if isinstance(b, dict) else b))
                return
            for k in a:
                same(a[k], b[k], where + "/" + k)
        elif isinstance(a, list):
            if not (isinstance(b, list) and len(a) == len(b)):
                diffs.append((where, a, b))
                return
            for i, (x, y) in enumerate(zip(a, b)):
                same(x, y, "%s[%d]" % (where, i))
        elif isinstance(a, float) or isinstance(b, float):
            if a is None or b is None:
                ok = a is None and b is None
            elif math.isinf(a) or math.isinf(b):
                # the reference's file says -Infinity; a JSON writer without that token writes the lowest double
                ok = a == b or (a < 0 and b < -1e300) or (a > 0 and b > 1e300)
            else:
                ok = math.isclose(a, b, rel_tol=6e-5, abs_tol=1e-12)  # 5 significant digits in the file
            if not ok:
                diffs.append((where, a, b))
        elif a != b:
            diffs.append((where, a, b))

    checked = 0
    for want, got in zip(expected, docs):
        gid = want["graphinfo"]["ID"]
        assert got["graphinfo"]["ID"] == gid
        w, g = want["samples"]["SWAPS"], got["samples"]["SWAPS"]
        same(w["alleles"], g["alleles"], gid + "/alleles")  # alignment_statistics, per allele
        checked += len(w["alleles"])
        assert set(g["nodes"]) <= {n["name"] for n in want["graphinfo"]["nodes"]} and len(g["nodes"]) >= 3
        for key in ("bad_alignment_pct", "mean_linear", "median_linear", "multi_read", "paired_read", "problematic_graph",
                    "problematic_linear", "single_read", "variance_linear"):
            same(w[key], g[key], gid + "/" + key)
        for key in ("mean_graph", "median_graph", "variance_graph"):
            same(w[key], g[key], gid + "/[graph fragment length] " + key)
        for name, bp in w["breakpoints"].items():  # edge / allele counts and the likelihoods of every breakpoint
            same(bp["counts"], g["breakpoints"][name]["counts"], gid + "/" + name + "/counts")
            for key in ("GL", "GT", "allele_fractions", "num_reads", "coverage_test_pvalue"):
                if key in bp["gt"]:
                    same(bp["gt"][key], g["breakpoints"][name]["gt"][key], gid + "/" + name + "/gt/" + key)
        for key in ("GL", "GT", "allele_fractions", "num_reads"):
            same(w["gt"][key], g["gt"][key], gid + "/gt/" + key)
        assert want["breakpointinfo"] == got["breakpointinfo"]
    required = [x for x in diffs if "[graph fragment length]" not in x[0]]
    assert not required, diffs
    assert checked >= 6
    print("graph-fragment-length moments that differ from the older file:", [x for x in diffs if x not in required])


def test_paragraph_validate_alignments(tmp_path):
    """`paragraph --validate-alignments` (lib/grm/ValidationAligner.cpp:59-125, lib/grm/Align.cpp:42-55): reads simulated
    from known paths -- their fragment ids start with the encoded path -- are aligned, and the [VALIDATION] lines say how many
    MAPPED reads left their path: none of these do, the count documents are those of a run without the option."""
    import json
    import re
    import sys
    from paragraph_amd import build
    if not os.path.exists(build.PARAGRAPH_BIN):
        build.build_host()
    data = tmp_path / "sites"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e", "make_sites.py"), str(data), "12", "20", "9", "0", "paths"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    graphs = [l.strip() for l in open(data / "graphs.txt") if l.strip()]
    base = [build.PARAGRAPH_BIN, "-r", str(data / "ref.fa"), "-b", str(data / "reads.bam"), "--threads", "2", "-g"] + graphs
    plain = subprocess.run(base, capture_output=True, text=True, timeout=300)
    assert plain.returncode == 0 and "[VALIDATION]" not in plain.stderr, plain.stderr
    checked = subprocess.run(base[:1] + ["--validate-alignments"] + base[1:], capture_output=True, text=True, timeout=300,
                             env=dict(os.environ, PG_VALIDATE_DEBUG="1"))
    assert checked.returncode == 0, checked.stderr
    assert json.loads(checked.stdout) == json.loads(plain.stdout)
    lines = [l for l in checked.stderr.splitlines() if l.startswith("[VALIDATION]")]
    assert lines[0] == "[VALIDATION]\tMAPQ\tEmpMAPQ\tWrong\tTotal" and len(lines) == 4, lines
    m = re.match(r"\[VALIDATION\]\t60\t(\S+)\t(\d+)\t(\d+)$", lines[3])
    assert m, lines
    wrong, aligned = int(m.group(2)), int(m.group(3))
    docs = json.loads(plain.stdout)
    # (node ids are single digits in these graphs: the reference's character-wise node list reads them correctly)
    assert aligned > 500 and wrong <= aligned // 50, (lines, [l for l in checked.stderr.splitlines() if l.startswith("misplaced")])
    assert float(m.group(1)) == 60 if wrong == 0 else float(m.group(1)) > 15
    unaligned = int(lines[1].split("\t")[-1])
    repeats = int(lines[2].split("\t")[-1])
    assert unaligned >= 0 and repeats >= 0 and unaligned + repeats + aligned >= sum(1 for d in docs) * 10


def test_grmpy_alignment_output_folder(tmp_path):
    """grmpy -A <folder> (lib/grmpy/AlignSamples.cpp:57-109, 120-171): one <sample>-<graph ID>-<regions>.json.gz per (sample,
    graph) holding the sample's count document with "sample", "reference", "bam" and the per-read records -- the reads the
    filter chain rejected first, each with the filter's message under "error", then the kept ones; the folder also switches
    the filter tallies on; the genotypes are those of a run without -A; a folder that does not exist means "no files"."""
    import glob
    import gzip
    import json
    from paragraph_amd import build
    if not os.path.exists(build.GRMPY_BIN) or not os.path.exists(build.PARAGRAPH_BIN):
        build.build_host()
    sites = os.path.join(ROOT, "tests", "golden", "sites", "chrX")
    bam = os.path.join(sites, "chrX_graph_typing.bam")
    fasta = os.path.join(sites, "chrX_graph_typing.fa")
    manifest = tmp_path / "manifest.txt"
    manifest.write_text("#id\tpath\tdepth\tread length\tdepth sd\tsex\nSAMPLE 1\t%s\t44.2\t150\t20\tmale\nSAMPLE2\t%s\t44.2\t150\t20\tfemale\n"
                        % (bam, bam))
    graph = os.path.join(sites, "chrX_graph_typing.2sample.json")
    common = [build.GRMPY_BIN, "-r", fasta, "-m", str(manifest), "-g", graph, "-G", os.path.join(sites, "param.json"), "-t", "4"]
    plain = subprocess.run(common, capture_output=True, text=True, timeout=300)
    assert plain.returncode == 0, plain.stderr
    folder = tmp_path / "alignments"
    folder.mkdir()
    r = subprocess.run(common + ["-A", str(folder)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    want, got = json.loads(plain.stdout), json.loads(r.stdout)
    for s in ("SAMPLE 1", "SAMPLE2"):
        assert got["samples"][s]["gt"] == want["samples"][s]["gt"]
    description = json.load(open(graph))
    regions = "_".join(description["target_regions"])
    safe = lambda t: "".join(c if (c.isalnum() and c.isascii()) or c in ".-" else "_" for c in t)  # noqa: E731
    files = sorted(os.path.basename(f) for f in glob.glob(str(folder / "*")))
    assert files == sorted("%s-%s-%s.json.gz" % (safe(s), safe(description["ID"]), safe(regions)) for s in ("SAMPLE 1", "SAMPLE2")), files
    # the same site through `paragraph -a -A`: the same records
    p = subprocess.run([build.PARAGRAPH_BIN, "-r", fasta, "-b", bam, "-g", graph, "-a", "-A", "--path-sequence-matching", "0", "--threads", "2"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    pdoc = json.loads(p.stdout)
    for f, sample in zip(files, ("SAMPLE2", "SAMPLE 1") if files[0].startswith("SAMPLE2") else ("SAMPLE 1", "SAMPLE2")):
        doc = json.loads(gzip.open(folder / f, "rt").read())
        assert doc["sample"] == sample and doc["reference"] == fasta and doc["bam"] == bam and doc["ID"] == description["ID"]
        al = doc["alignments"]
        rejected = [a for a in al if "error" in a]
        kept = [a for a in al if "error" not in a]
        assert rejected and kept and al[:len(rejected)] == rejected  # the rejected ones come first
        assert all(a["error"] in ("bad_align", "nonuniq") for a in rejected)
        assert all(a.get("graphMappingStatus") == "BAD_ALIGN" and "graphNodesSupported" not in a for a in rejected)
        assert all(a.get("graphMappingStatus") == "MAPPED" and "graphCigar" in a for a in kept)
        stats = doc["alignment_statistics"]
        assert stats.get("read_filter_bad_align", 0) == sum(a["error"] == "bad_align" for a in rejected)
        assert stats.get("read_filter_nonuniq", 0) == sum(a["error"] == "nonuniq" for a in rejected)
        assert doc["read_counts_by_edge"] and "read_counts_by_sequence" in doc
        assert al == pdoc["alignments"] and doc["read_counts_by_edge"] == pdoc["read_counts_by_edge"]
    # without -A the filtered records and tallies are off
    assert "read_filter_bad_align" not in json.dumps(want)
    missing = subprocess.run(common + ["-A", str(tmp_path / "not_there")], capture_output=True, text=True, timeout=300)
    assert missing.returncode == 0 and json.loads(missing.stdout) == want and not (tmp_path / "not_there").exists()
