import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A session that holds GPU parity tests insists on the reference-built checkers (PG_ALLOW_PORT_CHECKER=1 lifts it, for a
    box that really has no oracle/_ref; the header and the summary then say so)."""
    if any(item.get_closest_marker("gpu") for item in items) and os.environ.get("PG_ALLOW_PORT_CHECKER") != "1":
        os.environ.setdefault("PG_REQUIRE_REF", "1")


def _ref_files():
    ref = os.path.join(ROOT, "oracle", "_ref")
    return sorted(f for f in os.listdir(ref) if f.endswith(".so")) if os.path.isdir(ref) else []


def pytest_report_header(config):
    files = _ref_files()
    return ["parity checkers: %s; PG_REQUIRE_REF=%s" % (
        ("oracle/_ref/{%s} = the reference's own gssw.c / ksw.c / graph-tools" % ", ".join(files)) if files
        else "oracle/_ref ABSENT -> the builder's restatements (oracle/pg_oracle.c, oracle/*.py)", os.environ.get("PG_REQUIRE_REF", "unset"))]


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    from oracle import select
    for stage, (kind, name) in sorted(select.chosen().items()):
        terminalreporter.write_line("checker[%s] = %s (%s)" % (stage, name, kind))


@pytest.fixture(scope="session")
def checker():
    """The CPU checker: the reference's own gssw.c (oracle/_ref).  The plain-C port stands in only where oracle/_ref is absent
    AND PG_REQUIRE_REF is not 1 -- a `-m gpu` session sets it (pytest_collection_modifyitems below), so a GPU-box run without
    the reference-built checker is red, not green against the builder's own restatement."""
    from oracle import select
    return select.gssw()


@pytest.fixture(scope="session")
def gpu_ctx():
    from paragraph_amd import capi
    ctx = capi.Context(0)
    # the lean gssw stage for EVERY chunk (the library's default takes it from 30 G cell updates per chunk on: the tests' batches are smaller);
    # PG_LEAN=0 in the environment: the plain stage (tests/test_gpu_parity.py::test_launch_settings_do_not_change_results)
    import os
    if os.environ.get("PG_LEAN", "") != "0":
        ctx.set_lean(2)
    yield ctx
    ctx.close()
