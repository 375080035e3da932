"""paragraph_amd.multigrmpy: genotypes written back into VCF records (src/python/lib/grm/vcfgraph/vcfupdate.py:92-310) --
the text side, no device.  The expected lines are share/test-data/round-trip-genotyping/expected-vcf-record.txt; the genotype
documents here are hand-made to say what that file says (the real ones come from the GPU test in test_gpu_workflow.py)."""
import gzip
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RT = os.path.join(ROOT, "tests", "golden", "sites", "round-trip")


def _docs(graph_ids):
    zero = {"num_fwd_reads": 0, "num_rev_reads": 0}
    nocall = {"gt": {"GT": ".", "filters": ["NO_VALID_GT"]}, "alleles": {}}

    def called(var, fwd, n, gl):
        return {"gt": {"GT": "%s:1/%s:1" % (var, var), "filters": ["PASS"], "num_reads": n,
                       "GL": {"REF/REF": gl[0], "REF/%s:1" % var: gl[1], "%s:1/%s:1" % (var, var): gl[2]}},
                "alleles": {"REF": dict(zero), "%s:1" % var: {"num_fwd_reads": fwd, "num_rev_reads": 0}}}
    nc = lambda var: {"gt": nocall["gt"], "alleles": {"REF": dict(zero), "%s:1" % var: dict(zero)}}  # noqa: E731
    return [
        {"graphinfo": {"ID": graph_ids[0], "sequencenames": ["REF", "test-ins:1"]},
         "samples": {"sample1": called("test-ins", 1, 2, (-3.0, -0.7, -0.0)), "sample2": nc("test-ins")}},
        {"graphinfo": {"ID": graph_ids[1], "sequencenames": ["REF", "test-del:1"]},
         "samples": {"sample1": nc("test-del"), "sample2": called("test-del", 4, 8, (-12.0, -2.8, 0.0))}},
    ]


def _expected():
    lines = [l.rstrip("\n") for l in open(os.path.join(RT, "expected-vcf-record.txt"))]
    # the one documented difference: the reference's writer shows a FT it wrote before a longer one as dots
    return [l.replace("1/1:2:....:", "1/1:2:PASS:") for l in lines]


def test_pl_genotype_order():
    from paragraph_amd.multigrmpy import make_pl_genotypes
    assert make_pl_genotypes(2, 1) == [[0, 0], [0, 1], [1, 1]]
    assert make_pl_genotypes(2, 2) == [[0, 0], [0, 1], [1, 1], [0, 2], [1, 2], [2, 2]]
    assert make_pl_genotypes(1, 2) == [[0], [1], [2]]


@pytest.mark.parametrize("by", ["id", "sequencename"])
def test_update_vcf_reproduces_the_expected_records(tmp_path, by):
    from paragraph_amd import multigrmpy as mg
    exp = _expected()
    ids = [l.split("\t")[7].split("=", 1)[1] for l in exp[1:]]
    src = os.path.join(RT, "candidates.vcf")
    if by == "id":  # the record already names its graph (variants.vcf.gz of a run)
        src = str(tmp_path / "variants.vcf")
        with open(src, "w") as out:
            n = 0
            for line in open(os.path.join(RT, "candidates.vcf")):
                if not line.startswith("#"):
                    f = line.rstrip("\n").split("\t")
                    f[7] = "GRMPY_ID=" + ids[n]
                    n += 1
                    line = "\t".join(f) + "\n"
                out.write(line)
    gj = tmp_path / "genotypes.json.gz"
    with gzip.open(gj, "wt") as f:
        json.dump(_docs(ids), f)
    out = str(tmp_path / "genotypes.vcf.gz")
    stats = mg.update_vcf_from_grmpy(src, mg.read_grmpy(str(gj)), out, ["sample1", "sample2"])
    assert stats == {"matched": 2, "unmatched": 0, "multimatched": 0}
    got = [l.rstrip("\n") for l in gzip.open(out, "rt")]
    assert got[-3:] == exp
    header = [l for l in got if l.startswith("##")]
    for key in ("GT", "DP", "FT", "AD", "ADF", "ADR", "PL"):
        assert sum(1 for h in header if h.startswith("##FORMAT=<ID=%s," % key)) == 1
    assert sum(1 for h in header if h.startswith("##INFO=<ID=GRMPY_ID,")) == 1
    assert any(h.startswith("##FILTER=<ID=UNMATCHED,") for h in header)


def test_update_vcf_unmatched_multimatched_and_old_genotypes(tmp_path):
    from paragraph_amd import multigrmpy as mg
    vcf = tmp_path / "in.vcf"
    vcf.write_text("##fileformat=VCFv4.2\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
                   "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\told\n"
                   "chr1\t161\ttest-ins\tT\tTGGGGGG\t.\tPASS\tAC=2\tGT\t1|0\n"
                   "chr1\t300\t.\tA\tC\t.\t.\t.\tGT\t0/1\n"
                   "chr1\t400\tdup\tA\tC,G\t.\t.\t.\tGT\t./.\n")
    docs = _docs(["g1", "g2"])
    docs.append({"graphinfo": {"ID": "g3", "sequencenames": ["dup:1"]}, "samples": {}})
    docs.append({"graphinfo": {"ID": "g4", "sequencenames": ["dup:2"]}, "samples": {}})
    gj = tmp_path / "g.json"
    gj.write_text(json.dumps(docs))
    out = str(tmp_path / "o.vcf")
    stats = mg.update_vcf_from_grmpy(str(vcf), mg.read_grmpy(str(gj)), out, ["sample1"])
    assert stats == {"matched": 1, "unmatched": 1, "multimatched": 1}
    rows = [l.rstrip("\n").split("\t") for l in open(out) if not l.startswith("##")]
    assert rows[0][9:] == ["old", "sample1"]
    ins, snv, dup = rows[1:]
    assert ins[6] == "PASS" and ins[7] == "AC=2;GRMPY_ID=g1"
    assert ins[8] == "GT:OLD_GT:DP:FT:AD:ADF:ADR:PL"
    assert ins[9] == ".:0/1:.:.:.,.:.,.:.,.:.,.,."  # no genotype document for this sample: the old call moves to OLD_GT (sorted)
    assert ins[10] == "1/1:.:2:PASS:0,1:0,1:0,0:30,7,0"
    assert snv[6] == "UNMATCHED" and snv[7] == "GRMPY_ID=UNMATCHED" and snv[9:] == ["0/1", "."]
    assert dup[6] == "MULTIMATCHED" and dup[7] == "GRMPY_ID=MULTIPLE:g3,g4"


def test_manifest_header_check(tmp_path):
    from paragraph_amd import multigrmpy as mg
    assert mg.manifest_samples(os.path.join(RT, "samples.txt")) == ["sample1", "sample2"]
    bad = tmp_path / "m.txt"
    bad.write_text("id\tpath\tcoverage\ns\tx.bam\t3\n")
    with pytest.raises(ValueError):
        mg.manifest_samples(str(bad))
    bad.write_text("id\tpath\tdepth\ns\tx.bam\t3\n")
    with pytest.raises(ValueError):
        mg.manifest_samples(str(bad))
