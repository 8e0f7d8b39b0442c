"""One summary row per bench.py JSON line of a .jsonl file.  usage: python tools/print_bench_lines.py FILE..."""
import json
import sys

for path in sys.argv[1:]:
    for ln in open(path):
        if ln.startswith("{"):
            d = json.loads(ln)
            r = d["roofline"]
            print("%-48s %-8s %10.1f %s | frac %.3f of %.0f | launch %.3f ms | trunk share %s | tree %.0f ms"
                  % (d["metric"], d["dtype"].split(" ")[0], d["value"], d["unit"], r["frac"], r["peak"], r["avg_launch_ms"],
                     r["net_time_share"], r["tree_kernels_ms"]))
