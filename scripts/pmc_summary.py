"""Median per-dispatch PMC counters per kernel from rocprofv3 csv output (counter_collection.csv files under <dir>/pmc_*)."""
import csv, glob, json, os, statistics, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "").split("(")[0].replace("void ", "")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: {"median": statistics.median(v), "n": len(v)} for c, v in cs.items()} for k, cs in acc.items()}
# what the counters describe: bench.py quotes a summary only for the kernel sources it was collected on
sys.path.insert(0, os.getcwd())
import time
from bench import kernel_source_tag
out["_meta"] = {"kernel_source_tag": kernel_source_tag(), "collected": time.strftime("%Y-%m-%d %H:%M:%S"),
                "command": sys.argv[2] if len(sys.argv) > 2 else ""}
print(json.dumps(out, indent=1))
