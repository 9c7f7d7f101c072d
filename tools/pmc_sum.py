"""Sums the counters of a rocprofv3 --pmc run (its *_counter_collection.csv files under argv[1])
per kernel and counter, and prints them with the launch counts as JSON."""
import csv, glob, json, os, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        launches[k].add(row["Dispatch_Id"])
print(json.dumps({k: dict(v, launches=len(launches[k])) for k, v in tot.items()}, indent=1))
