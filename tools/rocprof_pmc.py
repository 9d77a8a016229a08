"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database.  usage: rocprof_pmc.py db out.md [title]"""
import sqlite3
import sys


def main(db, out, title="rocprofv3 PMC summary", last=0):
    """last > 0: only the last `last` dispatches of every kernel (e.g. the steady-state launches after a training warm-up)"""
    c = sqlite3.connect(db)
    if int(last) > 0:
        rows = list(c.execute("""select s.kernel_name, p.name, d.id, sum(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
                                 join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, p.name, d.id order by d.id"""))
        agg = {}
        for k, pn, did, v in rows:
            agg.setdefault((k, pn), []).append(v)
        with open(out, "w") as f:
            f.write(f"# {title} - last {last} dispatches of every kernel, counter summed over all XCD / SE instances of a dispatch\n\n| kernel | counter | dispatches | avg per dispatch |\n|---|---|---|---|\n")
            for (k, pn), v in sorted(agg.items(), key=lambda kv: -sum(kv[1][-int(last):])):
                v = v[-int(last):]
                name = k.replace(".kd", "")
                f.write(f"| `{name[:100]}` | {pn} | {len(v)} | {sum(v) / len(v):.1f} |\n")
        print("wrote", out)
        return
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_pmc_event)")]
    pcols = [r[1] for r in c.execute("pragma table_info(rocpd_info_pmc)")]
    q = """select s.kernel_name, p.name, count(*), avg(e.value), sum(e.value)
           from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
           join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name, p.name order by 5 desc"""
    try:
        rows = list(c.execute(q))
    except Exception as ex:   # schema differences: dump what we know
        open(out, "w").write(f"# {title}\n\nquery failed: {ex}\n\npmc_event columns: {cols}\n\ninfo_pmc columns: {pcols}\n")
        print("query failed", ex, cols, pcols)
        return
    with open(out, "w") as f:
        f.write(f"# {title}\n\nsource: `rocprofv3 --pmc <counter> --kernel-trace` (own pass per counter), values as reported by rocprofv3 (FETCH_SIZE / WRITE_SIZE in KiB)\n\n")
        f.write("| kernel | counter | dispatches | avg per dispatch | total |\n|---|---|---|---|---|\n")
        for r in rows[:40]:
            name = r[0].replace(".kd", "")
            name = name if len(name) <= 100 else name[:97] + "..."
            f.write(f"| `{name}` | {r[1]} | {r[2]} | {r[3]:.1f} | {r[4]:.1f} |\n")
    print("wrote", out)


if __name__ == "__main__":
    main(*sys.argv[1:5])
