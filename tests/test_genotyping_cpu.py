"""Host genotyping module: the reference's unit-test expectations re-typed in C++
(tests/host_cpp/test_genotyping.cpp) and a randomized comparison with the scipy-based checker (oracle/genotyper.py)."""
import json
import subprocess

from paragraph_amd import build


def test_reference_unit_vectors():
    exe = build.build_genotyping_test()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr


def test_checker_pinned_on_reference_vectors():
    from oracle import genotyper as og
    assert og.genotype([20, 0], 40.0, 100, 20, False)["gt"] == "0/0"
    assert og.genotype([20, 20], 40.0, 100, 20, False)["gt"] == "0/1"
    assert og.genotype([0, 20], 40.0, 100, 20, False)["gt"] == "1/1"
    assert og.genotype([0, 20], 40.0, 100, 20, False, ploidy=1)["gt"] == "1"
    assert abs(og.genotype([0, 20], 40.0, 100, 20, False)["pvalue"] / 0.24825223 - 1) < 1e-6
    assert abs(og.genotype([0, 20], 40.0, 100, 20, True)["pvalue"] / 0.0080560343 - 1) < 1e-6
    assert og.genotype([1, 20, 2, 20, 2], 40.0, 100, 20, False)["gt"] == "1/3"


def test_random_counts_against_checker():
    from oracle import genotyper as og
    exe = build.build_genotyping_test()
    out = subprocess.run([exe, "--dump", "7", "400"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rows = json.loads(out.stdout)
    assert len(rows) == 400
    n_called = 0
    for r in rows:
        w = og.genotype(r["counts"], r["depth"], r["read_length"], r["sd"], r["poisson"], ploidy=r["ploidy"])
        if w["gt"] == ".":
            assert r["gt"] == "." and r["filters"] == w["filters"]
            continue
        top = sorted(w["gl"], reverse=True)
        tie = len(top) > 1 and abs(top[0] - top[1]) <= 1e-9 * max(1.0, abs(top[0]))
        if not tie:  # mathematically tied likelihoods (e.g. equal counts, ploidy 1) are decided by the last ulp
            assert r["gt"] == w["gt"], (r, w)
        assert r["filters"] == w["filters"], (r, w)
        n_called += 1
        # GQ is a truncation of -10 log10(1 - p): allow the two sides to sit on either side of an integer boundary
        assert abs(r["gq"] - w["gq"]) <= 1, (r, w)
        assert len(r["gl"]) == len(w["gl"])
        for a, b in zip(r["gl"], w["gl"]):
            assert a == b or abs(a - b) <= 1e-9 * max(1.0, abs(b)), (r, w)
        assert abs(r["pvalue"] - w["pvalue"]) <= 1e-9 + 1e-7 * abs(w["pvalue"]), (r, w)
    assert n_called > 300
