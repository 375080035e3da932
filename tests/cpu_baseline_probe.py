import faulthandler, sys, time, os
faulthandler.dump_traceback_later(80, exit=True)
sys.path.insert(0, ".")
from paragraph_amd import synth
from oracle import oracle as orc
site, arr = synth.config2_reads_packed(20000)
reads = [row.tobytes().decode() for row in arr]
chk = orc.RefOracle()
for threads, n in [(1, 200), (8, 1600), (32, 3200), (64, 6400), (128, 12800), (256, 16384)]:
    t0 = time.perf_counter()
    chk.align_batch(site.seqs, site.edges, reads[:n], threads=threads)
    dt = time.perf_counter() - t0
    print("ref threads", threads, "n", n, "dt %.2f" % dt, "reads/s %.0f" % (n / dt), flush=True)
chk = orc.PortOracle()
for threads, n in [(1, 200), (64, 6400), (256, 16384)]:
    t0 = time.perf_counter()
    chk.align_batch(site.seqs, site.edges, reads[:n], threads=threads)
    dt = time.perf_counter() - t0
    print("port threads", threads, "n", n, "dt %.2f" % dt, "reads/s %.0f" % (n / dt), flush=True)
