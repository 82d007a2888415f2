"""Summarise a rocprofv3 --pmc counter_collection csv: per kernel name x grid size, mean of every counter per launch."""
import csv, sys, re, collections
rows = csv.DictReader(open(sys.argv[1]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "conv_" not in n and "bn_" not in n:
        continue
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    key = (n, r["Grid_Size"], r["Workgroup_Size"])
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[(key, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for key, cs in acc.items():
    d = [v for (k, _), v in dur.items() if k == key]
    print("%s grid=%s wg=%s launches=%d avg %.1f us" % (key[0], key[1], key[2], len(d), sum(d) / len(d)))
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0)
    for c, v in sorted(m.items()):
        print("    %-28s %14.0f  %s" % (c, v, ("%.1f%% of wave cycles" % (100 * v / wc)) if wc and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" else ""))
