"""Wall clock of the paragraph command line on an e2e data set: python tools/e2e/time_paragraph_cli.py <data dir> <threads>..."""
import json, subprocess, sys, time
d = sys.argv[1]
graphs = open(d + "/graphs.txt").read().split()
bam = open(d + "/manifest.txt").read().split("\n")[1].split("\t")[1]
rows = []
for t in sys.argv[2:]:
    t0 = time.perf_counter()
    subprocess.run(["paragraph_amd/bin/paragraph", "-r", d + "/ref.fa", "-b", bam, "-o", "/tmp/pg_cli_counts.json", "--threads", t, "-g"] + graphs,
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows.append({"threads": int(t), "wall_s": round(time.perf_counter() - t0, 3)})
print(json.dumps({"sites": len(graphs), "command": "bin/paragraph -r -b -g <sites> -o out.json --threads <threads> (path stage + gssw; process start and output included)", "rows": rows}))
