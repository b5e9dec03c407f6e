"""Sum rocprofv3 --pmc counters per kernel:  python3 profiles/summarize_pmc.py <counter_collection.csv> [name filter]

Prints, per kernel name, the number of dispatches, the summed duration and the sum of every counter."""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    seen = set()
    agg = defaultdict(lambda: {"n": 0, "ns": 0, "c": defaultdict(float), "vgpr": None})
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            if pat not in name:
                continue
            a = agg[name]
            d = row["Dispatch_Id"]
            if d not in seen:
                seen.add(d)
                a["n"] += 1
                a["ns"] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                a["vgpr"] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["Scratch_Size"], row["LDS_Block_Size"])
            a["c"][row["Counter_Name"]] += float(row["Counter_Value"])
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
        print(f"{name}\n  dispatches {a['n']}  total {a['ns'] / 1e6:.3f} ms  (vgpr, agpr, scratch, lds) = {a['vgpr']}")
        for k, v in sorted(a["c"].items()):
            print(f"  {k:32s} {v:.6e}   per dispatch {v / max(1, a['n']):.6e}")


if __name__ == "__main__":
    main()
