"""Derive HBM bytes per counting pass from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

usage: python tools/pmc_traffic.py fetch.csv write.csv out.json
Counter values are KiB per dispatch.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (wide coalesced
streaming reads are reported at half their size); WRITE_SIZE is taken as is.  The per-kernel figure is the mean over the
dispatches of the run; the per-pass total sums the kernels of one pb_count call.
"""
import collections
import csv
import json
import sys

KERNELS = ("sk_part1_kernel", "sk_part2_raw16_kernel", "sk_part2_raw12_kernel", "sk_part2_kernel", "sk_bucket_kernel", "sk_bucket2_kernel", "sk_flagged_kernel", "hist_array_kernel")


def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]      # "void sk_part2_kernel<SkStore12>(SkParams)"
            acc[name].append(float(r["Counter_Value"]) * 1024.0)
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    kernels, total = {}, 0.0
    for k in KERNELS:
        if k not in f and k not in w:
            continue
        fb = 2.0 * sum(f.get(k, [0])) / max(len(f.get(k, [0])), 1)
        wb = sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1)
        kernels[k] = {"fetch_bytes": fb, "write_bytes": wb}
        total += fb + wb
    import hashlib
    import os
    ksrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "platanus_b_amd", "csrc", "kernels_superk.inc")
    json.dump({
        "kernels_sha1": hashlib.sha1(open(ksrc, "rb").read()).hexdigest(),       # bench.py quotes the figure only for these kernels
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 "
                   "--no-cpu-baseline (two separate passes)",
        "units": "counter values are KiB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced "
                 "streaming reads), WRITE_SIZE taken as is",
        "workload": "5 Mbp x 100x PE150, k=32 (396,666,508 instances, algorithmic 4.86 GB per counting pass)",
        "kernels": kernels,
        "hbm_bytes_per_counting_pass": total,
    }, open(out, "w"), indent=1)
    print(json.dumps(kernels, indent=1), total)


if __name__ == "__main__":
    main()
