"""One summary row per bench.py JSON line of a .jsonl file.  usage: python tools/print_bench_lines.py FILE..."""
import json
import sys

for path in sys.argv[1:]:
    for ln in open(path):
        if ln.startswith("{"):
            d = json.loads(ln)
            r = d["roofline"]
            print("%-48s %-8s %10.1f %s | frac %.3f of %.0f | launch %.3f ms | trunk share %s | tree %.0f ms | lanes overlap %s%s"
                  % (d["metric"], d["dtype"].split(" ")[0], d["value"], d["unit"], r["frac"], r["peak"], r["avg_launch_ms"],
                     r["net_time_share"], r["tree_kernels_ms"], r.get("lanes_overlap"),
                     " SERIALISED" if r.get("lanes_serialised") else ""))
            if d.get("per_rank_lanes_overlap") and d.get("ranks", 1) > 1:
                print("    per rank: games/s %s | lanes overlap %s | launch ms %s | exchange %s ms/step" % (
                    d["per_rank_games_per_s"], d["per_rank_lanes_overlap"], d.get("per_rank_avg_launch_ms"), d.get("exchange_ms_per_step")))
            for o in d.get("other_configs") or []:
                if "value" in o:
                    print("    other_configs %-26s %10.1f %s | frac %.3f | overlap %s | %s | %s" % (
                        o["config"], o["value"], o["unit"], o["roofline_frac"], o.get("lanes_overlap"), o["kernel"].split(" ")[0],
                        ("cache hit rate %.3f" % o["eval_cache"]["hit_rate"]) if "eval_cache" in o else ""))
                else:
                    print("    other_configs", o)
            if d.get("cpu_baseline"):
                print("    cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value", "value_ci95", "unit", "cores", "streams", "kind")})
