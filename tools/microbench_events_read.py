import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id order by d.start"))
pairs = []
for a, b in zip(rows, rows[1:]):
    if "add" in a[2].lower() and "mul" in b[2].lower() and "elementwise" in a[2]:
        pairs.append((b[0] - a[1]) / 1e3)
kinds = ("record", "wait_done_event", "record+wait"); ns = (0, 1, 2, 4, 8)
i = 0
for k in kinds:
    print(k, " ".join(f"n={n}: {pairs[i + j]:.1f} us" for j, n in enumerate(ns))); i += len(ns)
