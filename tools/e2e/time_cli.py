"""Wall clock of the grmpy command line on an e2e data set: python tools/e2e/time_cli.py <data dir> <threads>..."""
import json, subprocess, sys, time
d = sys.argv[1]
graphs = open(d + "/graphs.txt").read().split()
rows = []
for t in sys.argv[2:]:
    t0 = time.perf_counter()
    subprocess.run(["paragraph_amd/bin/grmpy", "-r", d + "/ref.fa", "-m", d + "/manifest.txt", "-o", "/tmp/pg_cli_out.json", "-t", t, "-g"] + graphs,
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows.append({"threads": int(t), "wall_s": round(time.perf_counter() - t0, 3)})
print(json.dumps({"sites": len(graphs), "command": "bin/grmpy -r -m -g <sites> -o out.json -t <threads> (process start, device start-up, 40 MB of JSON written)", "rows": rows}))
