"""Largest idle gaps of the busiest stream in a rocprofv3 kernel trace (rocpd database), and the busy fraction over the last `n` training steps."""
import sqlite3, sys
db = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
c = sqlite3.connect(db)
rows = list(c.execute("select d.start, d.end, d.stream_id, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"))
marks = [i for i, r in enumerate(rows) if "k_field_fwdI6__halfLi1ELb0" in r[3] or "k_field32_fwdILi1ELb0" in r[3] or "k_field32_fwd_splitILi1ELb0" in r[3]]      # one per training step (fp16 | fp32 network)
a, b = marks[-nsteps - 1], marks[-1]
seg = rows[a:b]
main = seg[0][2]
t0, t1 = seg[0][0], rows[b][0]
ms = [r for r in seg if r[2] == main]
busy = sum(r[1] - r[0] for r in ms)
print(f"{nsteps} steps: {(t1 - t0) / nsteps / 1e3:.1f} us/step, main stream busy {busy / nsteps / 1e3:.1f} us/step ({100 * busy / (t1 - t0):.0f} %)")
gaps = []
allgap = 0
for p, q in zip(ms, ms[1:]):
    g = q[0] - p[1]
    allgap += max(g, 0)
    if g > 15000:
        gaps.append((g, p, q))
print(f"all gaps between consecutive main-stream kernels: {allgap / nsteps / 1e3:.1f} us/step over {len(ms) / nsteps:.1f} launches/step")
tot = sum(g for g, _, _ in gaps)
print(f"gaps > 15 us: {len(gaps)} totalling {tot / nsteps / 1e3:.1f} us/step")
for g, p, q in sorted(gaps, key=lambda x: -x[0])[:14]:
    print(f"  {g / 1e3:8.1f} us  after {p[3][:48]:48s} before {q[3][:48]}")
# which kernels fill the busy time, per step
agg = {}
for r in ms:
    agg[r[3][:60]] = agg.get(r[3][:60], 0) + (r[1] - r[0])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:16]:
    print(f"  {v / nsteps / 1e3:7.1f} us/step  {k}")
# (r6) which side-stream kernels run BESIDE each main-stream kernel: per main kernel name - average duration when nothing overlaps it / when something does, and the
# overlapped microseconds per step by side kernel.  Answers "what sits on k_bin_accumulate2"  without reading the raw timeline.
side = sorted((r for r in seg if r[2] != main), key=lambda r: r[0])
import bisect
starts = [r[0] for r in side]
maxdur = max((r[1] - r[0] for r in side), default=0)
stats = {}
for r in ms:
    nm = r[3][:44]
    st = stats.setdefault(nm, dict(alone=[0, 0], beside=[0, 0], by={}))
    lo = bisect.bisect_left(starts, r[0] - maxdur)
    ov_tot = 0
    for q in side[lo:]:
        if q[0] >= r[1]:
            break
        ov = min(r[1], q[1]) - max(r[0], q[0])
        if ov > 0:
            ov_tot += ov
            st["by"][q[3][:36]] = st["by"].get(q[3][:36], 0) + ov
    key = "beside" if ov_tot > 0.1 * (r[1] - r[0]) else "alone"
    st[key][0] += 1; st[key][1] += r[1] - r[0]
print("main-stream kernels beside side-stream kernels (overlap > 10 % of the kernel's duration = 'beside'):")
for nm, st in sorted(stats.items(), key=lambda kv: -(kv[1]["alone"][1] + kv[1]["beside"][1]))[:12]:
    a_, b_ = st["alone"], st["beside"]
    by = ", ".join(f"{k} {v / nsteps / 1e3:.1f}" for k, v in sorted(st["by"].items(), key=lambda kv: -kv[1])[:4])
    print(f"  {nm:44s} alone {a_[0]:4d} x {a_[1] / max(a_[0], 1) / 1e3:6.1f} us | beside {b_[0]:4d} x {b_[1] / max(b_[0], 1) / 1e3:6.1f} us | overlapped us/step: {by}")
