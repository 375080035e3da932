"""Which CPU checker a test / smoke / bench leg runs against -- TEST INFRASTRUCTURE ONLY.

Two kinds exist for every stage: the reference's own code compiled as it lies (oracle/_ref/*.so, built by oracle/Makefile
where /root/reference exists; git-ignored, it travels to the GPU box with the snapshot) and the builder's restatement
(oracle/pg_oracle.c, oracle/*.py).  A restatement is itself pinned on the reference's vectors, but a parity run against it
is a weaker statement than one against the reference's code, and nobody should have to guess which one ran:

* `choose()` returns the reference-built checker when it is present;
* with PG_REQUIRE_REF=1 in the environment (tests/conftest.py sets it for every `-m gpu` session) a missing oracle/_ref is an
  ERROR, not a silent change of checker;
* every choice is recorded (`chosen()`), printed once on stderr, and pytest's header / summary lines repeat it.
"""
import os
import sys

_CHOSEN = {}


def strict():
    return os.environ.get("PG_REQUIRE_REF", "") == "1"


def choose(stage, have_ref, make_ref, make_port, ref_name, port_name):
    """stage: short name ("gssw", "counts", "klib", "path", "kmerfilter"); make_*: zero-argument factories."""
    if have_ref():
        kind, name, made = "reference", ref_name, make_ref()
    elif strict():
        raise RuntimeError(
            "PG_REQUIRE_REF=1 and %s is missing: the %s parity checks would run against the builder's restatement (%s).  Build it "
            "where /root/reference exists (make -C oracle ref) and ship oracle/_ref/ with the tree, or unset PG_REQUIRE_REF to "
            "accept the weaker checker." % (ref_name, stage, port_name))
    else:
        kind, name, made = "port", port_name, make_port()
    if _CHOSEN.get(stage) != (kind, name):
        _CHOSEN[stage] = (kind, name)
        print("[checker] %s: %s (%s)" % (stage, name, "the reference's own code" if kind == "reference"
                                         else "RESTATEMENT -- oracle/_ref absent, parity weaker"), file=sys.stderr, flush=True)
    return made


def chosen():
    """{stage: (kind, name)} of every choice made in this process."""
    return dict(_CHOSEN)


def gssw():
    """The read -> graph aligner checker (gssw.c behind oracle/ref_harness.c, or oracle/pg_oracle.c)."""
    from . import oracle as orc

    def port():
        if not orc.have_port():
            import subprocess
            subprocess.run(["make", "-C", os.path.dirname(os.path.abspath(__file__)), "port"], check=True)
        return orc.PortOracle()
    return choose("gssw", orc.have_ref, orc.RefOracle, port, "oracle/_ref/libpg_ref.so (reference gssw.c)",
                  "oracle/libpg_oracle.so (plain-C restatement)")


def count_site():
    """The count path's checker as a function (graph, records, **kw) -> dict."""
    from . import counts as oc
    return choose("counts", oc.have_ref, lambda: oc.RefCounts().count_site, lambda: oc.port_count_site,
                  "oracle/_ref/libpg_refcounts.so (graph-tools compiled + Disambiguation glue restated)", "oracle/counts.py (restatement)")


def klib():
    from . import klibalign as ok
    return choose("klib", ok.have_ref, ok.ref_klib, ok.port_klib, "oracle/_ref (reference ksw.c under oracle/klib_glue.h)",
                  "oracle/klibalign.py (scalar ksw restatement)")


def path_align():
    from . import pathalign as pa
    return choose("path", pa.have_ref, lambda: pa.ref_path_align, lambda: pa.port_path_align,
                  "oracle/_ref/libpg_refcounts.so (graph-tools PathOperations / KmerIndex)", "oracle/pathalign.py (restatement)")


def kmer_filter():
    from . import counts as oc
    from . import kmerfilter as kf
    return choose("kmerfilter", oc.have_ref, lambda: kf.ref_kmer_filter, lambda: kf.port_kmer_filter,
                  "oracle/_ref/libpg_refcounts.so (graph-tools KmerIndex)", "oracle/kmerfilter.py (restatement)")
