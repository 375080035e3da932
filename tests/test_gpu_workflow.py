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
