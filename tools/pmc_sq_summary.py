"""Per-kernel means of the SQ counter passes written by tools/pmc_sq.sh.   usage: python tools/pmc_sq_summary.py <dir> [out.json]"""
import collections, csv, glob, json, sys

def main():
    d = sys.argv[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(d + "/pass*/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            acc[name]["_vgpr"] = [float(r["VGPR_Count"])]
            acc[name]["_lds"] = [float(r["LDS_Block_Size"])]
            acc[name]["_wg"] = [float(r["Workgroup_Size"])]
    out = {}
    for k, v in acc.items():
        if not any(k.startswith(p) for p in ("sk_", "sk2_", "mg_", "hist_", "first_occ", "occ_", "part_", "raw_")):
            continue
        m = {c: sum(x) / len(x) for c, x in v.items()}
        m["_ms_profiled"] = sum(dur[k]) / len(dur[k])
        # derived (SQ cycle counters are quad-cycles summed over waves / SIMDs)
        if m.get("SQ_WAVE_CYCLES"):
            wc = m["SQ_WAVE_CYCLES"]
            for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
                if c in m:
                    m["frac_" + c] = m[c] / wc
        if m.get("SQ_LDS_IDX_ACTIVE"):
            m["lds_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]
        out[k] = m
    json.dump(out, open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout, indent=1)

main()
