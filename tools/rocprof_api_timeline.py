"""One training step of a rocprofv3 run with --kernel-trace --memory-copy-trace --hip-trace (rocpd database): kernels, memory copies and HIP API calls on one time axis,
plus the per-step count of every API call over the last `n` steps.  Answers "what sits between the last kernel of a step and the first of the next"."""
import sqlite3, sys
db = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
def cols(t):
    return [r[1] for r in c.execute(f"pragma table_info('{t}')")]
def find(prefix):
    m = [t for t in tabs if t.startswith(prefix)]
    return m[0] if m else None
if "--schema" in sys.argv:
    for t in tabs:
        n = c.execute(f"select count(*) from '{t}'").fetchone()[0]
        print(t, n, cols(t))
kd, ks = find("rocpd_kernel_dispatch"), find("rocpd_info_kernel_symbol")
rows = list(c.execute(f"select d.start, d.end, d.stream_id, s.kernel_name from '{kd}' d join '{ks}' s on d.kernel_id=s.id order by d.start"))
marks = [i for i, r in enumerate(rows) if "k_field_fwdI6__halfLi1ELb0" in r[3] or "k_field32_fwdILi1ELb0" in r[3]]
a, b = marks[-nsteps - 1], marks[-1]
t0, t1 = rows[a][0], rows[b][0]
ev = [(r[0], r[1], f"K s{r[2]}", r[3][:56]) for r in rows[a:b]]
# memory copies
mc = find("rocpd_memory_copy")
ncopy = 0
if mc:
    cc = cols(mc)
    sz = "size" if "size" in cc else ("bytes" if "bytes" in cc else None)
    q = f"select start, end, {sz or 0}" + (", stream_id" if "stream_id" in cc else ", 0") + (", name_id" if "name_id" in cc else ", 0") + f" from '{mc}' where start >= ? and start < ? order by start"
    for r in c.execute(q, (t0, t1)):
        ev.append((r[0], r[1], f"C s{r[3]}", f"memcpy {r[2]} B kind {r[4]}")); ncopy += 1
# API regions
rg, st = find("rocpd_region"), find("rocpd_string")
api = {}
if rg and st:
    rc = cols(rg)
    tid = "tid" if "tid" in rc else ("thread_id" if "thread_id" in rc else "0")
    for r in c.execute(f"select g.start, g.end, s.string, g.{tid} from '{rg}' g join '{st}' s on g.name_id = s.id where g.start >= ? and g.start < ? order by g.start", (t0, t1)):
        api[r[2]] = api.get(r[2], 0) + 1
        ev.append((r[0], r[1], f"A t{r[3]}", r[2]))
print(f"{nsteps} steps, {(t1 - t0) / nsteps / 1e3:.1f} us/step; memory copies in window: {ncopy} ({ncopy / nsteps:.2f}/step)")
print("API calls per step:")
for k, v in sorted(api.items(), key=lambda kv: -kv[1]):
    print(f"  {v / nsteps:8.2f}  {k}")
# one step in detail: the last complete one
s0, s1 = rows[marks[-2]][0], rows[marks[-1]][0]
print(f"\nlast step in detail ({(s1 - s0) / 1e3:.1f} us); GPU events only (K/C) with API calls omitted unless --api")
for e in sorted(ev):
    if s0 <= e[0] < s1 and (e[2][0] != "A" or "--api" in sys.argv):
        print(f"+{(e[0] - s0) / 1e3:8.1f} us  dur {(e[1] - e[0]) / 1e3:7.1f}  {e[2]:8s} {e[3]}")
