"""Combines the FETCH_SIZE and WRITE_SIZE passes (two rocpd databases) into profiles/<round>_pmc.json: HBM bytes per launch for every library kernel, keyed
by the kernel's base name (what bench.py's `roofline.kernel` names), under the configuration key bench.py looks up (`lego` | `fox`).
FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-B requests at 64 B), WRITE_SIZE taken as reported (KiB); both are the fabric-side
request counters of the L2, so Infinity-Cache hits are included.  One kernel serves launches of different sizes (k_hash_fwd: the batch and the occupancy
refresh; k_adam_ema: the table and the weight pack): the class with the most dispatches among the last 64 is reported, like bench.py's batch class.
usage: rocprof_pmc_json.py fetch.db write.db out.json "<command line profiled>" <config key> <scene>     (bench.py uses an entry only for the same config AND scene)"""
import json
import re
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    q = """select s.kernel_name, d.id, d.grid_size_x, sum(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
           join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id where p.name = ? group by s.kernel_name, d.id order by d.id"""
    out = {}
    for name, did, grid, v in c.execute(q, (counter,)):
        m = re.match(r"_Z(\d+)", name)                     # Itanium mangling: _Z<length><name>...
        if not m:
            continue
        k = name[m.end():m.end() + int(m.group(1))]
        if not k.startswith("k_"):
            continue
        # (r6) template instantiations that bench.py's brackets know under names of their own (NGP_LAUNCH registers the macro alias, csrc/hash_encode.hip)
        rest = name[m.end() + int(m.group(1)):]
        if k == "k_bin_accumulate2" and rest.startswith("IfLb1E"):
            k = "k_bin_accumulate2_adam"
        elif k == "k_bin_accumulate" and "Lb1EE" in rest[:60]:
            k = "k_bin_accumulate_adam_f16rec" if "__half2" in rest[:60] else "k_bin_accumulate_adam_f32rec"
        out.setdefault(k, []).append((grid, v))
    res = {}
    for k, v in out.items():
        v = v[-64:]
        grids = {}
        for g, x in v:
            grids.setdefault(g, []).append(x)
        g, xs = max(grids.items(), key=lambda kv: len(kv[1]) * kv[0])       # the class that carries most of the work
        res[k] = {"launches": len(xs), "avg": sum(xs) / len(xs), "grid": g}
    return res


def main(fetch_db, write_db, out, command, key="lego", scene="bricks"):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    try:
        res = json.load(open(out))
    except Exception:
        res = {}
    res["_source"] = {"passes": ["rocprofv3 --pmc FETCH_SIZE --kernel-trace", "rocprofv3 --pmc WRITE_SIZE --kernel-trace"],
                      "correction": "hbm_bytes_per_launch = 2 * FETCH_SIZE_KiB * 1024 + WRITE_SIZE_KiB * 1024 (gfx950: FETCH_SIZE reports half of wide coalesced reads; WRITE_SIZE uncalibrated, as reported). "
                                    "The 256 MiB Infinity Cache sits behind these counters' tap, so re-reads that hit it are still counted."}
    res.setdefault("_commands", {})[key] = command
    res.setdefault("_scenes", {})[key] = scene
    res[key] = {}
    if not (set(f) & set(w)):
        print('NO kernel has both counters - pass incomplete; nothing written for', key); return
    for k in sorted(set(f) & set(w)):
        res[key][k] = {"launches_sampled": f[k]["launches"], "grid": f[k]["grid"], "FETCH_SIZE_KiB": round(f[k]["avg"], 1), "WRITE_SIZE_KiB": round(w[k]["avg"], 1),
                       "hbm_bytes_per_launch": int(2 * f[k]["avg"] * 1024 + w[k]["avg"] * 1024)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res[key], indent=1)[:3000])


if __name__ == "__main__":
    main(*sys.argv[1:7])
