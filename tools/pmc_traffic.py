"""Build profiles/rNN_hbm_traffic_pmc.json from two rocprofv3 --pmc passes of the same command:
   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR_F -o f -- python bench.py --steps 2 --warmup 1 --no-graph --no-roofline --no-cpu-baseline
   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d DIR_W -o w -- python bench.py ... (same)
usage: python tools/pmc_traffic.py DIR_F/f_counter_collection.csv DIR_W/w_counter_collection.csv STEPS out.json
FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream
(MI355X_MICROARCH.md, HBM section): it is doubled here."""
import csv, json, re, sys, collections


def load(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"<.*", "", n) if not n.startswith("at::") else re.sub(r"\(.*", "", n)
        n = re.sub(r"\(.*", "", n)
        a = acc[n]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


f, w, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
F, W = load(f, "FETCH_SIZE"), load(w, "WRITE_SIZE")
res = {}
tf = tw = 0.0
for k in sorted(set(F) | set(W), key=lambda k: -(2 * F.get(k, [0, 0])[1] + W.get(k, [0, 0])[1])):
    calls = F.get(k, W.get(k))[0]
    fb, wb = 2 * F.get(k, [0, 0])[1] * 1024, W.get(k, [0, 0])[1] * 1024
    tf += fb; tw += wb
    if (fb + wb) / 1e9 < 0.005:
        continue
    res[k] = {"calls": calls, "fetch_GB_corrected_x2": round(fb / 1e9, 3), "write_GB": round(wb / 1e9, 3),
              "MB_per_launch": round((fb + wb) / calls / 1e6, 2)}
res["_per_step"] = {"fetch_GB": round(tf / 1e9 / steps, 2), "write_GB": round(tw / 1e9 / steps, 2), "steps_profiled": steps,
                    "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced stream); FETCH counts L2 misses incl. Infinity Cache hits"}
import hashlib, os
_conv = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multibox_amd", "csrc", "conv.hip")
res["conv_hip_sha"] = hashlib.sha256(open(_conv, "rb").read()).hexdigest()[:16]     # bench.py prints traffic only if these match
res["conv5_hip_sha"] = hashlib.sha256(open(_conv.replace("conv.hip", "conv5.hip"), "rb").read()).hexdigest()[:16]
res["conv7_hip_sha"] = hashlib.sha256(open(_conv.replace("conv.hip", "conv7.hip"), "rb").read()).hexdigest()[:16]
res["convd_hip_sha"] = hashlib.sha256(open(_conv.replace("conv.hip", "convd.hip"), "rb").read()).hexdigest()[:16]
res["convr_hip_sha"] = hashlib.sha256(open(_conv.replace("conv.hip", "convr.hip"), "rb").read()).hexdigest()[:16]
res["conv_common_h_sha"] = hashlib.sha256(open(_conv.replace("conv.hip", "conv_common.h"), "rb").read()).hexdigest()[:16]
json.dump(res, open(out, "w"), indent=1)
for k, v in res.items():
    print(k, v)
