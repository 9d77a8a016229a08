"""Prints the kernel timeline of one steady-state training step from a rocprofv3 rocpd database (start offset, duration, queue, name)."""
import sqlite3, sys
db = sys.argv[1]
c = sqlite3.connect(db)
rows = list(c.execute("""select d.start, d.end, d.queue_id, d.stream_id, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"""))
marks = [i for i, r in enumerate(rows) if "k_field_fwdI6__halfLi1ELb0" in r[4] or "k_field32_fwdILi1ELb0" in r[4] or "k_field32_fwd_splitILi1ELb0" in r[4]]
a, b = marks[-12], marks[-11]          # one step well inside the timed region (field_fwd to field_fwd)
t0 = rows[a][0]
busy = {}
prev_end = {}
for r in rows[a:b]:
    q = r[3]
    gap = (r[0] - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = r[1]
    busy[q] = busy.get(q, 0) + (r[1] - r[0]) / 1e3
    print(f"+{(r[0]-t0)/1e3:8.1f} us  dur {(r[1]-r[0])/1e3:7.1f}  gap {gap:6.1f}  stream {q}  {r[4][:80]}")
print("step length", (rows[b][0] - t0) / 1e3, "us; busy per stream", busy)
