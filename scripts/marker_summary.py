"""Summarises a rocprofv3 --marker-trace --kernel-trace --memory-copy-trace run of scripts/ring_trace.py: how long the host
ranges take (reve:upload / chain / download / wait), how long the copies and the kernel chain of a frame take on the device,
and how much of the copies' time lies UNDER kernels of other frames (the overlap the ring claims)."""
import csv, glob, os, sys
d = sys.argv[1]
def rows(pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out
mk = rows("*marker_api_trace.csv")
kn = rows("*kernel_trace.csv")
cp = rows("*memory_copy_trace.csv")
print(f"{len(mk)} marker ranges, {len(kn)} kernel dispatches, {len(cp)} copies")
by = {}
for r in mk:
    name = r.get("Function") or r.get("Name") or r.get("Message") or "?"
    by.setdefault(name, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(by.items()):
    v.sort()
    print(f"  host range {k:16s} n={len(v):5d} median {v[len(v) // 2] / 1e3:8.1f} us  p99 {v[int(len(v) * 0.99)] / 1e3:8.1f} us")
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in kn)
def busy_under(a, b):          # ns of [a, b) covered by kernels
    t = 0
    for s, e in ks:
        if e <= a: continue
        if s >= b: break
        t += min(e, b) - max(s, a)
    return t
# copies between PINNED host memory and the device are reported as DEVICE_TO_DEVICE by rocprofv3 (the pinned pages belong to the
# GPU agent's address space): the ring's two copies per frame are told apart by their length (6.2 MB up, 24.9 MB down at C2)
allc = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in cp if "DEVICE_TO_DEVICE" in (r.get("Direction") or "")]
allc = allc[6:-6] if len(allc) > 20 else allc          # steady state
if allc:
    med = sorted(e - s for s, e in allc)[len(allc) // 2]
    for name, sel in (("uploads (the shorter copies)", [c for c in allc if c[1] - c[0] <= med]), ("downloads (the longer copies)", [c for c in allc if c[1] - c[0] > med])):
        if not sel: continue
        tot = sum(e - s for s, e in sel)
        under = sum(busy_under(s, e) for s, e in sel)
        print(f"  {name}: {len(sel)} copies, mean {tot / len(sel) / 1e3:.1f} us, {100.0 * under / max(tot, 1):.1f} % of their time under kernels")
if ks:
    span = ks[-1][1] - ks[0][0]
    print(f"  kernels busy {100.0 * sum(e - s for s, e in ks) / span:.1f} % of the traced span ({span / 1e6:.1f} ms)")
