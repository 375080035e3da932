"""Breakpoint genotyper checker -- TEST INFRASTRUCTURE ONLY.

scipy-based restatement of genotyping::BreakpointGenotyper::genotype / genotypeLikelihood
(src/c++/lib/genotyping/BreakpointGenotyper.cpp:86-255) with GenotypingParameters' defaults
(src/c++/lib/genotyping/GenotypingParameters.cpp:37-84): scipy.stats.poisson / norm stand where the reference uses
boost::math.  Pinned on src/c++/test/test_breakpoint_genotyper.cpp:44-82 (tests/test_genotyping_cpu.py); floating point:
compared with a relative tolerance of 1e-9 on GLs / p-values, exactly on GT / GQ / filters.
"""
import math

from scipy import stats


def possible_genotypes(n_alleles, ploidy):
    gts = []

    def make(p, n, suffix):
        for a in range(n + 1):
            new = [a] + suffix
            if p == 1:
                gts.append(new)
            elif p > 1:
                make(p - 1, a, new)
    if n_alleles:
        make(ploidy, n_alleles - 1, [])
    return gts


def genotype(counts, depth, read_length, depth_sd, use_poisson_depth, ploidy=2, error_rate=0.05, het_fraction=0.5,
             min_overlap_bases=16, min_pass_gq=10, cutoff=(0.02, 0.0001)):
    n_alleles = len(counts)
    lam = depth * ((read_length - min_overlap_bases) / float(read_length))
    total = sum(counts)
    if total == 0:
        return {"gt": ".", "gq": -1, "filters": "NO_READS", "gl": [], "pvalue": -1}
    gls, best, gt = [], -float("inf"), None
    for g in possible_genotypes(n_alleles, ploidy):
        gl = 0.0
        for al in range(n_alleles):
            copies = sum(1 for x in g if x == al)
            mean = lam * error_rate if copies == 0 else lam * copies * het_fraction
            p = stats.poisson.pmf(counts[al], mean)
            if p == 0:
                gl = -1.7976931348623157e308
                break
            gl += math.log(p)
        gls.append(gl)
        if gl > best:
            best, gt = gl, g
    s = sum(math.exp(x) for x in gls)
    pr_err = 1.0 - math.exp(best) / s
    if pr_err == 0:
        gq = 100
    else:
        lg = math.log10(pr_err)
        gq = 100 if lg < -10 else int(-10 * lg)
    filters = set()
    if gq < min_pass_gq:
        filters.add("GQ")
    pv = stats.poisson.cdf(total, lam) if use_poisson_depth else stats.norm.cdf(total, lam, depth_sd)
    if pv > 0.5:
        pv = 1 - pv
        if pv < cutoff[0]:
            filters.add("BP_DEPTH")
    elif pv < cutoff[1]:
        filters.add("BP_DEPTH")
    return {"gt": "/".join(str(x) for x in gt), "gq": gq, "filters": ";".join(sorted(filters)), "gl": gls, "pvalue": float(pv)}
