"""Builds the in-tree native artefacts (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libparagraph_amd.so")
HIP_SOURCES = ["pg_api.hip", "pg_fill.hip", "pg_trace.hip", "pg_count.hip", "pg_path.hip", "pg_kmer.hip", "pg_klib.hip", "pg_klib_packed.hip", "pg_general.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


KERNEL_SHA_SOURCES = ["pg_device.h", "pg_fill.hip", "pg_kernels.h", "pg_pk16.h", "pg_trace.hip"]


def kernel_source_sha():
    """sha256 over the sources of the gssw-stage kernels (fill + traceback and the headers they include), first 16 hex
    digits.  Counter files collected under rocprofv3 (profiles/traffic_rNN.json, rNN_sq_counters.json) carry the value they
    were collected at; bench.py reports them only for the kernels they were measured on."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SHA_SOURCES:
        h.update(name.encode())
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_hip(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> paragraph_amd/libparagraph_amd.so (the C-ABI library)."""
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in ("pg_device.h", "pg_kernels.h", "pg_internal.h", "pg_pk16.h", "pg_klib.h", "pg_general.h", "pg_kmerindex.h")] + \
        [os.path.join(ROOT, "include", "paragraph_amd.h")]
    if not force and not _stale(LIB, deps):
        return LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


HOST_LIB = os.path.join(_HERE, "libparagraph_host.so")
HOST_TEST = os.path.join(ROOT, "tests", "host_cpp", "test_host")
GRMPY_BIN = os.path.join(_HERE, "bin", "grmpy")  # the reference's `grmpy` / `paragraph` command lines over the batched workflow
PARAGRAPH_BIN = os.path.join(_HERE, "bin", "paragraph")


HOST_CPU_SOURCES = ["genotyping.cpp", "json.cpp", "io.cpp", "graphio.cpp"]  # no device calls: also built into the CPU test programs
HOST_GPU_SOURCES = ["host.cpp", "workflow.cpp"]


def _host_paths():
    inc = os.path.join(_HERE, "host", "include")
    hdrs = [os.path.join(dp, f) for dp, _, fs in os.walk(inc) for f in fs]
    src = lambda names: [os.path.join(_HERE, "host", "src", n) for n in names]
    return inc, hdrs, src(HOST_CPU_SOURCES), src(HOST_GPU_SOURCES)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)


def build_host(force=False, verbose=False):
    """g++ -> paragraph_amd/libparagraph_host.so (reference-shaped C++ classes over the C ABI, BAM/FASTA/JSON input, the
    batched site workflow) and the C++ test program tests/host_cpp/test_host."""
    inc, hdrs, cpu_src, gpu_src = _host_paths()
    cxx = os.environ.get("CXX", "g++")
    if force or _stale(HOST_LIB, cpu_src + gpu_src + [LIB] + hdrs):
        _run([cxx, "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-I" + inc, "-o", HOST_LIB] + gpu_src + cpu_src
             + ["-L" + _HERE, "-lparagraph_amd", "-lz", "-Wl,-rpath,$ORIGIN"], verbose)
    cli_hdr = os.path.join(_HERE, "host", "src", "cli_common.hh")
    for exe, main_src in ((GRMPY_BIN, "grmpy_main.cpp"), (PARAGRAPH_BIN, "paragraph_main.cpp")):
        main_src = os.path.join(_HERE, "host", "src", main_src)
        if force or _stale(exe, [main_src, cli_hdr, HOST_LIB] + hdrs):
            os.makedirs(os.path.dirname(exe), exist_ok=True)
            _run([cxx, "-std=c++17", "-O2", "-pthread", "-I" + inc, "-o", exe, main_src, "-L" + _HERE, "-lparagraph_host",
                  "-lparagraph_amd", "-lz", "-Wl,-rpath,$ORIGIN/.."], verbose)
    for name in ("test_host", "test_workflow"):
        tsrc = os.path.join(ROOT, "tests", "host_cpp", name + ".cpp")
        exe = os.path.join(ROOT, "tests", "host_cpp", name)
        if os.path.exists(tsrc) and (force or _stale(exe, [tsrc, HOST_LIB] + hdrs)):
            _run([cxx, "-std=c++17", "-O2", "-pthread", "-I" + inc, "-o", exe, tsrc, "-L" + _HERE, "-lparagraph_host",
                  "-lparagraph_amd", "-Wl,-rpath,$ORIGIN/../../paragraph_amd"], verbose)
    return HOST_LIB


def build_genotyping_test(force=False, verbose=False):
    """CPU-only programs of the "not gpu" suite: the genotyping module test and the input / statistics test (no HIP)."""
    inc, hdrs, cpu_src, _ = _host_paths()
    cxx = os.environ.get("CXX", "g++")
    exes = []
    for name in ("test_genotyping", "test_hostio"):
        tsrc = os.path.join(ROOT, "tests", "host_cpp", name + ".cpp")
        exe = os.path.join(ROOT, "tests", "host_cpp", name)
        if os.path.exists(tsrc) and (force or _stale(exe, [tsrc] + cpu_src + hdrs)):
            _run([cxx, "-std=c++17", "-O2", "-I" + inc, "-o", exe, tsrc] + cpu_src + ["-lz"], verbose)
        exes.append(exe)
    return exes[0]


def build_all(force=False, verbose=False):
    build_hip(force=force, verbose=verbose)
    build_host(force=force, verbose=verbose)
    build_genotyping_test(force=force, verbose=verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
