#!/bin/bash
# HBM-side traffic of bench.py's OWN launches (two lanes, ~2 000 positions per trunk launch): separate rocprofv3 --pmc passes
# (--kernel-trace only, as MI355X_MICROARCH.md prescribes) over a short bench.py run; writes gpurun_out/${OTH_ROUND:-r06}_bench_traffic.json
# (copy it to profiles/: bench.py reads roofline.traffic and roofline_rollout.traffic from there) and a text summary.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
CMD="python3 bench.py --gpus 1 --steps 1 --warmup 2 --step-games 512 --stagger 8 --profile-steps 1 --no-cpu-baseline --no-other-configs"
out=gpurun_out/pmc_bench
rm -rf "$out"; mkdir -p "$out"
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -o "$name" -- $CMD > "$out/$name.json" 2> "$out/$name.err" || { tail -5 "$out/$name.err"; exit 1; }
  python3 tools/pmc_summary.py "$out/$name" --all | sed "s|^$out/||" >> "$out/summary.txt"
done
cat "$out/summary.txt"
python3 - "$out" "$CMD" <<'PY'
import importlib.util, json, re, sys
out, cmd = sys.argv[1], sys.argv[2]
spec = importlib.util.spec_from_file_location("bench", "bench.py")
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
vals = {}
for ln in open(out + "/summary.txt"):
    m = re.match(r"(\w+) (oth::[\w<>, ]+?) (\{.*\})$", ln.strip())
    if not m:
        continue
    kern = m.group(2)
    for c, v in re.findall(r"'(\w+)': '([0-9.e+]+) \(n=(?:\d+)\)'", m.group(3)):
        vals.setdefault(kern, {})[c] = float(v)
    n = re.search(r"\(n=(\d+)\)", m.group(3))
    vals[kern]["launches"] = int(n.group(1))
line = [l for l in open(out + "/fetch.json") if l.startswith("{")][-1]
ppl = json.loads(line)["roofline"]["positions_per_launch"]
res = {"command": cmd, "kernel_source_sha256": bench.trunk_source_sha256(), "kernel_sources": list(bench.TRUNK_SOURCES),
       "source": "rocprofv3 --kernel-trace --pmc, one pass per counter group, per-launch means over all launches of the run", "kernels": {}}
for kern, v in vals.items():
    key = "trunk" if ("k_trunk16" in kern or "k_trunk_w" in kern) else ("k_tree" if kern.startswith("oth::k_tree<1") else None)
    if key is None or "FETCH_SIZE" not in v:
        continue
    wide = key == "trunk"   # 16 B/lane coalesced weight reads: FETCH_SIZE counts 64 B per 128-B request on gfx950
    rd = v["FETCH_SIZE"] * 1024 * (2 if wide else 1)
    res["kernels"][key] = {
        "kernel": kern, "launches": v["launches"], "FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v.get("WRITE_SIZE"),
        "TCC_HIT_sum": v.get("TCC_HIT_sum"), "TCC_MISS_sum": v.get("TCC_MISS_sum"), "TCC_EA0_RDREQ_sum": v.get("TCC_EA0_RDREQ_sum"),
        "read_bytes": rd, "write_bytes": v.get("WRITE_SIZE", 0) * 1024,
        "traffic_bytes_per_launch": rd + v.get("WRITE_SIZE", 0) * 1024,
        "fetch_correction": "x2 (wide coalesced 16 B/lane reads, MI355X_MICROARCH.md HBM section)" if wide else
                            "none (narrow dependent reads: uncalibrated access width, reported as counted)",
    }
    if wide:
        res["kernels"][key]["positions_per_launch"] = ppl
import os; json.dump(res, open("gpurun_out/%s_bench_traffic.json" % os.environ.get("OTH_ROUND", "r06"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf "$out"/fetch "$out"/write "$out"/tcc
