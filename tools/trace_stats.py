"""Per (kernel, grid size) launch statistics from a rocprofv3 --kernel-trace run (argv[1]: its output
directory): launches, average / min / max duration.  Launches of one kernel at different problem
sizes (the benchmark's N = 1M headline and its N = 65,536 legs) stay apart this way."""
import csv, glob, json, os, sys
from collections import defaultdict
rows = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        rows[(name, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = []
for (name, grid), d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    out.append({"kernel": name, "grid_size_x": grid, "launches": len(d), "avg_ms": sum(d) / len(d) / 1e6,
                "min_ms": min(d) / 1e6, "max_ms": max(d) / 1e6, "total_ms": sum(d) / 1e6})
print(json.dumps(out, indent=1))
