"""Development: profile by function from the CLI's PB_HOST_SAMPLES file (platanus_b_amd/host/main.cpp: SIGPROF samples of a few return
addresses each + /proc/self/maps).  Symbols come from `nm -C` of the mapped files as they are HERE (same image on the GPU box and in the
build container; the CLI binary travels with the snapshot).

usage: python tools/host_samples.py gpurun_out/samples.txt [--top 40] [--binary platanus_b_amd/bin/platanus_b]
Prints (a) the leaf function of every sample, (b) the innermost frame that lies in the CLI binary itself (who asked libc / libstdc++ for
the work)."""
import bisect
import collections
import os
import subprocess
import sys


def symbols(path):
    out = []
    for flags in (["-C", "-n", "--defined-only"], ["-C", "-n", "-D", "--defined-only"]):
        try:
            txt = subprocess.run(["nm", *flags, path], capture_output=True, text=True).stdout
        except OSError:
            continue
        for ln in txt.splitlines():
            p = ln.split(" ", 2)
            if len(p) == 3 and p[1] in "tTwWiu":
                try:
                    out.append((int(p[0], 16), p[2]))
                except ValueError:
                    pass
    out.sort()
    return [a for a, _ in out], [n for _, n in out]


def main():
    a = sys.argv[1:]
    path = a[0]
    top = int(a[a.index("--top") + 1]) if "--top" in a else 40
    binary = a[a.index("--binary") + 1] if "--binary" in a else None
    maps, samples = [], []
    in_samples = False
    for ln in open(path):
        if ln.startswith("#### SAMPLES"):
            in_samples = True
            continue
        if in_samples:
            samples.append([int(x, 16) for x in ln.split()])
        else:
            f = ln.split()
            if len(f) >= 6 and "x" in f[1]:
                lo, hi = (int(x, 16) for x in f[0].split("-"))
                maps.append((lo, hi, int(f[2], 16), f[5]))
    maps.sort()
    starts = [m[0] for m in maps]
    cache = {}

    def resolve(addr, ret):
        i = bisect.bisect_right(starts, addr) - 1
        if i < 0 or addr >= maps[i][1]:
            return ("?", "?")
        lo, hi, off, file = maps[i]
        use = file
        if binary and os.path.basename(file).startswith("platanus_b"):
            use = binary
        if use not in cache:
            cache[use] = symbols(use) if os.path.exists(use) else ([], [])
        addrs, names = cache[use]
        # position-independent files: file offset == symbol address for the text segment of ordinary ELF files
        q = addr - lo + off - (1 if ret else 0)
        j = bisect.bisect_right(addrs, q) - 1
        return (os.path.basename(file), names[j] if j >= 0 else "?")

    leaf = collections.Counter()
    own = collections.Counter()
    pair = collections.Counter()
    for s in samples:
        if not s or s[0] == 0:
            continue
        m0, f0 = resolve(s[0], False)
        leaf[(m0, f0)] += 1
        who = None
        for d, adr in enumerate(s):
            if adr == 0:
                break
            m, f = resolve(adr, d > 0)
            if m.startswith("platanus_b"):
                who = f
                break
        own[who or "(outside the CLI binary)"] += 1
        if not m0.startswith("platanus_b"):
            pair[(who or "?", f0)] += 1
    n = sum(leaf.values())
    print(f"{n} samples (1 ms of CPU time each)")
    print("\n-- leaf function --")
    for (m, f), c in leaf.most_common(top):
        print(f"{100.0 * c / n:6.2f} %  {c:7d}  {m:22s} {f[:150]}")
    print("\n-- innermost frame inside the CLI binary --")
    for f, c in own.most_common(top):
        print(f"{100.0 * c / n:6.2f} %  {c:7d}  {f[:170]}")
    print("\n-- library leaf functions by the CLI function that called them --")
    for (w, f), c in pair.most_common(top):
        print(f"{100.0 * c / n:6.2f} %  {c:7d}  {f[:50]:50s} <- {w[:120]}")


if __name__ == "__main__":
    main()
