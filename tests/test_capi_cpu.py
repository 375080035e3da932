"""CPU tests of the boundary: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/paragraph_amd.h declares.  No compute calls (there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "paragraph_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(pg_[a-z_0-9]+)\s*\(", text))
    return sorted(names)


def test_header_symbols_exported():
    from paragraph_amd import build, capi
    build.build_hip()
    lib = ctypes.CDLL(capi.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), "missing export: " + name
    assert sorted(capi.EXPORTS) == declared


def test_result_struct_layout():
    from paragraph_amd import capi
    assert capi.RESULT_DTYPE.itemsize == 24
    assert capi.RESULT_DTYPE.fields["ops_off"][1] == 12 and capi.RESULT_DTYPE.fields["status"][1] == 22


def test_no_device_fails_loudly():
    """Without a HIP device the product path must fail, not fall back to a CPU implementation."""
    import pytest
    from paragraph_amd import capi
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    with pytest.raises(capi.PgError) as ei:
        capi.Context(0)
    assert ei.value.status == 2  # PG_ERR_NO_DEVICE


def test_render_cigar_helper():
    import numpy as np
    from paragraph_amd import capi
    res = np.zeros(1, dtype=capi.RESULT_DTYPE)
    ops = np.array([(0 << 16) | (5 << 12) | 3, (0 << 16) | (0 << 12) | 8, (1 << 16) | (1 << 12) | 1,
                    (3 << 16) | (4 << 12) | 2, (3 << 16) | (0 << 12) | 7], dtype=np.uint32)
    res[0]["n_ops"] = len(ops)
    assert capi.render_cigar(res[0], ops) == "0[3S8M]1[1X]3[2D7M]"
    lib = capi.load_library()
    buf = ctypes.create_string_buffer(64)
    n = lib.pg_render_cigar(res.ctypes.data, ops.ctypes.data, buf, 64)
    assert buf.value.decode() == "0[3S8M]1[1X]3[2D7M]" and n == len(buf.value)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under paragraph_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "paragraph_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".hh")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle" not in text.replace("no oracle", ""), os.path.join(dirpath, fn)


def test_workflow_entry_exported_and_fails_loudly(tmp_path):
    """include/paragraph_workflow.h: the host library exports the workflow entry; bad inputs are reported through the
    error buffer, and without a GPU a well-formed call fails at the device instead of computing anything on the CPU."""
    import pytest
    from paragraph_amd import build, workflow
    build.build_host()
    text = open(os.path.join(ROOT, "include", "paragraph_workflow.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert re.findall(r"\b(pgw_[a-z_0-9]+)\s*\(", text) == ["pgw_genotype_graphs"]
    lib = workflow.load_library()
    assert hasattr(lib, "pgw_genotype_graphs")
    sites = os.path.join(ROOT, "tests", "golden", "sites", "chrX")
    fasta, graph = os.path.join(sites, "chrX_graph_typing.fa"), os.path.join(sites, "chrX_graph_typing.2sample.json")
    with pytest.raises(RuntimeError, match="Unable to open manifest"):
        workflow.genotype_graphs(fasta, str(tmp_path / "missing.txt"), [graph])
    manifest = tmp_path / "manifest.txt"
    manifest.write_text("id\tpath\tdepth\tread length\nS1\t%s\t44.2\t150\n" % os.path.join(sites, "chrX_graph_typing.bam"))
    with pytest.raises(RuntimeError, match="unknown option"):
        workflow.genotype_graphs(fasta, str(manifest), [graph], colour="red")
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    with pytest.raises(RuntimeError, match="pg_ctx_create"):
        workflow.genotype_graphs(fasta, str(manifest), [graph], threads=2)


def test_grmpy_command_line_parsing(tmp_path):
    """paragraph_amd/bin/grmpy keeps the reference's option names (src/c++/main/grmpy.cpp:60-200): what can be checked
    without a device -- usage, missing / unknown options, bool parsing, response files -- and that a complete command
    line fails at the device here instead of computing on the CPU."""
    import subprocess
    from paragraph_amd import build
    build.build_host()
    exe = build.GRMPY_BIN
    sites = os.path.join(ROOT, "tests", "golden", "sites", "chrX")
    fasta, graph = os.path.join(sites, "chrX_graph_typing.fa"), os.path.join(sites, "chrX_graph_typing.2sample.json")

    def run(*args):
        r = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=120)
        return r.returncode, r.stdout + r.stderr

    rc, out = run("--help")
    assert rc == 0 and "grmpy -r <reference> -g <graphs> -m <manifest>" in out
    assert run("-g", graph, "-m", "x") == (1, "Reference genome is missing.\n")
    assert run("-r", fasta, "-m", "x") == (1, "Graph spec is missing.\n")
    assert run("-r", fasta, "-g", graph) == (1, "Manifest file is missing.\n")
    rc, out = run("-r", fasta, "-g", graph, "-m", "x", "--colour", "red")
    assert rc == 1 and "unrecognised option '--colour'" in out
    rc, out = run("-r", fasta, "-g", graph, "-m", "x", "--path-sequence-matching", "maybe")
    assert rc == 1 and "is invalid" in out
    rc, out = run("-r", fasta, "-g", graph, "-m", "x", "--infer-read-haplotypes")
    assert rc == 1 and "not available" in out  # (phasing output is outside this build; -A is taken: tests/test_gpu_workflow.py)
    manifest = tmp_path / "manifest with space.txt"
    manifest.write_text("id\tpath\tdepth\tread length\nS1\t%s\t44.2\t150\n" % os.path.join(sites, "chrX_graph_typing.bam"))
    response = tmp_path / "response.txt"
    # the way multigrmpy.py writes it: one line of options, then the graphs one per line (multigrmpy.py:262-306)
    response.write_text(" -r %s -m '%s' -o %s -z -t 2 --graph-sequence-matching True --log-level=warning --log-file %s --log-async no -g\n%s\n%s"
                        % (fasta, manifest, tmp_path / "out.json.gz", tmp_path / "log.txt", graph, graph))
    rc, out = run("--response-file=%s" % response)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except ImportError:
        has_gpu = False
    if not has_gpu:
        assert rc == 1 and "pg_ctx_create" in out  # parsed everything, loaded graphs and manifest, then needed the device


def test_paragraph_command_line_parsing():
    """paragraph_amd/bin/paragraph keeps the reference's option names (src/c++/main/paragraph.cpp:83-290); outputs this build
    does not compute are refused by name instead of being silently dropped."""
    import subprocess
    from paragraph_amd import build
    build.build_host()

    def run(*args):
        r = subprocess.run([build.PARAGRAPH_BIN] + list(args), capture_output=True, text=True, timeout=120)
        return r.returncode, r.stdout + r.stderr

    rc, out = run("--help")
    assert rc == 0 and "paragraph -r <reference> -g <graph(s)> -b <input bam(s)>" in out
    assert run("-g", "g.json", "-r", "r.fa") == (1, "ERROR: BAM file is missing.\n")
    assert run("-b", "x.bam", "-r", "r.fa") == (1, "ERROR: File with variant specification is missing.\n")
    assert run("-b", "x.bam", "-g", "g.json") == (1, "ERROR: Reference genome is missing.\n")
    for refused in ("-v", "--output-path-coverage", "--output-node-coverage", "--output-read-haplotypes", "-E"):
        rc, out = run("-b", "x.bam", "-g", "g.json", "-r", "r.fa", refused)
        assert rc == 1 and "not available in this build" in out, refused
    rc, out = run("-b", "x.bam", "-g", "g.json", "-r", "r.fa", "--output-variants", "false", "--bad-align-nonuniq", "0", "--threads", "3")
    assert rc == 1 and "r.fa" in out  # accepted; fails only at the missing reference file
