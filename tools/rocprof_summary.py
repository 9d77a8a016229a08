"""Turns a rocprofv3 rocpd database (kernel trace) into the per-kernel summary committed under profiles/.
usage: python tools/rocprof_summary.py gpurun_out/prof/xyz_results.db profiles/r01_xyz.md "title" """
import sqlite3
import sys


def main(db, out, title, last_steps=0, after_marker=None):
    """after_marker: only dispatches after the LAST launch of a kernel whose name contains this string (tools/profile_part.py launches torch.tril as the marker)"""
    c = sqlite3.connect(db)
    cutoff = 0
    if after_marker:
        marks = [r[0] for r in c.execute("select d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id where s.kernel_name like ? order by d.start", (f"%{after_marker}%",))]
        if marks:
            cutoff = marks[-1]
    if last_steps:      # restrict to the last N training steps: cut at the start of the N-th from last batch launch of the field network
        marks = [r[0] for r in c.execute("""select d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id
                                            where s.kernel_name like '%k_field32_fwdILi1ELb0%' or s.kernel_name like '%k_field32_fwd_splitILi1ELb0%' or s.kernel_name like '%k_field_fwdI6__halfLi1ELb0%' order by d.start""")]      # one per training step
        if len(marks) > last_steps:
            cutoff = marks[-last_steps]
            title += f" — last {last_steps} training steps only"
    # one row per (kernel, launch size class): the same kernel serves the 2^18-sample training batch and e.g. the 3 M-point occupancy-grid refresh, or the
    # 13 M-parameter table and the 3 k-parameter weight packs - averaging those together would say nothing.  Size class = grid size within a factor of 2^(1/2).
    import math
    raw = list(c.execute(f"""select s.kernel_name, d.end-d.start, s.arch_vgpr_count, s.sgpr_count, d.group_segment_size, d.grid_size_x, d.workgroup_size_x
                             from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id where d.start >= {cutoff}"""))
    groups = {}
    for name, dur, vg, sg, lds, grid, wg in raw:
        key = (name, round(math.log2(max(grid, 1)) * 2))
        g = groups.setdefault(key, [name, 0, 0.0, 0.0, 1e30, 0.0, 0, 0, 0, 0, 0])
        g[1] += 1; g[2] += dur / 1e3; g[4] = min(g[4], dur / 1e3); g[5] = max(g[5], dur / 1e3)
        g[6] = max(g[6], vg or 0); g[7] = max(g[7], sg or 0); g[8] = max(g[8], lds or 0); g[9] = max(g[9], grid); g[10] = max(g[10], wg)
    rows = []
    for g in groups.values():
        g[3] = g[2] / g[1]
        rows.append(tuple(g))
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write(f"# {title}\n\nsource: `rocprofv3 --kernel-trace --stats` (rocpd database), all durations in microseconds\n\n")
        f.write("| kernel | calls | total ms | % | avg us | min us | max us | vgpr | sgpr | lds B | grid | wg |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows[:45]:
            name = r[0].replace(".kd", "")
            if len(name) > 110:
                name = name[:107] + "..."
            f.write(f"| `{name}` | {r[1]} | {r[2] / 1e3:.2f} | {100 * r[2] / tot:.1f} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} |\n")
        f.write(f"\ntotal kernel time {tot / 1e3:.1f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("wrote", out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 kernel summary", int(sys.argv[4]) if len(sys.argv) > 4 else 0, sys.argv[5] if len(sys.argv) > 5 else None)
