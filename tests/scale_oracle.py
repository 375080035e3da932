"""Reference results for the full-size parity tests, computed in a process of their own (TEST INFRASTRUCTURE).

The GPU tests hold a HIP context, so they must not fork; this script never touches HIP and fans the oracle out over the host's
cores with forked workers (one aligner per worker, as the reference's one-aligner-per-thread rule, Align.cpp:107-110):

    python tests/scale_oracle.py config2 <n_reads> <seed> <out.npz>
    python tests/scale_oracle.py config3 <n_sites> <seed> <out.pkl>     (alignments + count tables per site)
    python tests/scale_oracle.py config5 <n_graphs> <reads_per_graph> <seed> <out.pkl>

`run(...)` is the helper the tests call (subprocess + load)."""
import os
import pickle
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

STRIDE = 256
STRIDE_LONG = 1024


def _ncpu():
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def _checker():
    from oracle import select
    return select.gssw()


def _align_site(chk, seqs, edges, arr, stride):
    """(n, L) uint8 reads of one graph -> (RESULT_NP array, (n, stride) CIGAR slots)."""
    from oracle import oracle as orc
    n, L = arr.shape
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L)).astype(np.uint32)
    res = np.zeros(n, dtype=orc.RESULT_NP)
    cig = np.zeros((n, stride), dtype=np.uint8)
    if n:
        chk.align_into(seqs, edges, off, np.ascontiguousarray(arr).reshape(-1), res, cig, threads=1)
    return res, cig


def config5_cases(n_graphs, reads_per_graph, seed):
    """BASELINE configs[4]-style: long inline ALT nodes (2-8 kb) + 250 bp reads."""
    from paragraph_amd import synth
    out = []
    for gi in range(n_graphs):
        alt_len = (2100, 3000, 4000, 6000, 8000)[gi % 5]
        site = synth.long_node_site(seed * 1000 + gi, alt_len)
        arr = synth.simulate_reads_packed(site, reads_per_graph, 250, seed * 2000 + gi, indel_frac=0.05, random_frac=0.02)
        out.append((site, arr))
    return out


_JOB = None


def _site_job(i):
    kind, payload = _JOB
    chk = _checker()
    if kind == "config3":
        import bench
        return bench.reference_site_outcome(chk, payload[i], STRIDE)  # the same code bench.py's `sites.verified` leg runs
    site, arr = payload[i]
    res, cig = _align_site(chk, site.seqs, site.edges, arr, STRIDE_LONG)
    return {"res": res, "cig": cig}


def _pool_map(kind, payload):
    global _JOB
    import multiprocessing as mp
    _JOB = (kind, payload)
    procs = max(1, min(_ncpu(), len(payload)))
    with mp.get_context("fork").Pool(procs) as pool:
        return pool.map(_site_job, range(len(payload)), chunksize=max(1, len(payload) // (4 * procs)))


def main(argv):
    mode = argv[0]
    if mode == "config2":
        import bench
        from oracle import oracle as orc
        from paragraph_amd import synth
        n, seed, out = int(argv[1]), int(argv[2]), argv[3]
        site, arr = synth.config2_reads_packed(n, read_len=150, seed=seed)
        off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(150)).astype(np.uint32)
        res = bench._shared_array((n,), orc.RESULT_NP)
        cig = bench._shared_array((n, bench.CIGAR_STRIDE), np.uint8)
        bench._run_procs(_checker(), site, off, np.ascontiguousarray(arr).reshape(-1), res, cig, 0, n, min(_ncpu(), max(1, n // 256)))
        np.savez(out, res=np.array(res), cig=np.array(cig))
    elif mode == "config3":
        from paragraph_amd import synth
        n_sites, seed, out = int(argv[1]), int(argv[2]), argv[3]
        sites = synth.mixed_sites(n_sites, seed=seed, site_streams=True)
        with open(out, "wb") as f:
            pickle.dump(_pool_map("config3", sites), f, protocol=4)
    elif mode == "config5":
        n_graphs, per, seed, out = int(argv[1]), int(argv[2]), int(argv[3]), argv[4]
        with open(out, "wb") as f:
            pickle.dump(_pool_map("config5", config5_cases(n_graphs, per, seed)), f, protocol=4)
    else:
        raise SystemExit("unknown mode " + mode)


def run(*argv):
    """Runs this script with argv + a temp output path; returns the loaded result."""
    import tempfile
    fd, path = tempfile.mkstemp(prefix="pg_scale_", suffix=".npz" if argv[0] == "config2" else ".pkl")
    os.close(fd)
    try:
        env = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            env.pop(k, None)
        subprocess.run([sys.executable, os.path.abspath(__file__)] + [str(a) for a in argv] + [path], check=True, env=env)
        if argv[0] == "config2":
            z = np.load(path)
            return z["res"], z["cig"]
        with open(path, "rb") as f:
            return pickle.load(f)
    finally:
        if os.path.exists(path):
            os.unlink(path)


if __name__ == "__main__":
    main(sys.argv[1:])
