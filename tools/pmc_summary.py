"""Mean per-launch value of every counter of a `rocprofv3 --pmc ... --output-format csv` run, per kernel.

python tools/pmc_summary.py <dir with *_counter_collection.csv> [substring of kernel names to keep ...]

Prints JSON: {kernel: {counter: mean per launch, "launches": n}}.  For SQ counter sets it adds the instruction
count per wave and the VALU busy fraction (SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs))."""
import csv, glob, json, os, sys
from collections import defaultdict


def summarise(path, keep=()):
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit("no *_counter_collection.csv under " + path)
    tot = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0]
            if keep and not any(k in name for k in keep):
                continue
            tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[name].add(row["Dispatch_Id"])
    out = {}
    for name, c in tot.items():
        n = len(launches[name])
        d = {k: v / n for k, v in c.items()}
        d["launches"] = n
        if "SQ_INSTS_VALU" in d and "SQ_WAVES" in d and d["SQ_WAVES"]:
            d["valu_insts_per_wave"] = round(d["SQ_INSTS_VALU"] / d["SQ_WAVES"], 1)
        if "SQ_INSTS_VALU" in d and d.get("GRBM_GUI_ACTIVE"):
            d["valu_busy_fraction"] = round(d["SQ_INSTS_VALU"] * 4 / 1024 / (d["GRBM_GUI_ACTIVE"] / 8), 3)
        out[name] = d
    return out


if __name__ == "__main__":
    print(json.dumps(summarise(sys.argv[1], sys.argv[2:]), indent=1, sort_keys=True))
