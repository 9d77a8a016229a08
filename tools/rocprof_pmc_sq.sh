#!/bin/bash
# SQ counters of the hot-path kernels (own pass, no tracing besides --kernel-trace).  usage: tools/rocprof_pmc_sq.sh "<counters>" <out.md>
set -u
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq && mkdir -p /tmp/pq
rocprofv3 --pmc $1 --kernel-trace -d /tmp/pq -o sq -- python $R/bench.py --no-cpu-baseline --no-psnr --no-fox --no-neus --no-spheres --steps 20 --warmup 60 > /tmp/pq/log 2>&1
cd $R
DB=$(find /tmp/pq -name "*.db" | head -1)
python - "$DB" "$2" "$1" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("""select s.kernel_name, p.name, count(*), avg(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
           join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, p.name"""))
ks = {}
for n, p, cnt, v in rows:
    ks.setdefault(n, {})[p] = v
names = sys.argv[3].split()
with open(sys.argv[2], "w") as f:
    f.write("# SQ counters per dispatch (average), rocprofv3 --pmc " + sys.argv[3] + "\n\n| kernel | " + " | ".join(names) + " |\n|---|" + "---|" * len(names) + "\n")
    for n, d in sorted(ks.items(), key=lambda kv: -kv[1].get(names[0], 0))[:24]:
        f.write(f"| `{n[:70]}` | " + " | ".join(f"{d.get(k, 0):.3g}" for k in names) + " |\n")
print(open(sys.argv[2]).read())
PY
