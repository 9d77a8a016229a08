"""Adds the MFMA counters of the fused field kernels to profiles/<round>_pmc.json (what bench.py's `roofline.issued_frac` / `pipe_util` read) and writes the round's mfma.md.
usage: rocprof_mfma_json.py pq.db out.json out.md <config key> <fp16: 0|1>
pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over SQ_BUSY_CYCLES / 32 instances; issued fraction = MFMA instructions x FLOP per instruction / duration / dense peak."""
import json, re, sqlite3, sys


def main(db, out_json, out_md, key, fp16):
    fp16 = int(fp16)
    c = sqlite3.connect(db)
    rows = list(c.execute("""select s.kernel_name, p.name, d.id, sum(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
                             join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, p.name, d.id order by d.id"""))
    agg = {}
    for name, pn, did, v in rows:
        m = re.match(r"_Z(\d+)", name)
        if not m:
            continue
        k = name[m.end():m.end() + int(m.group(1))]
        if "field" not in k:
            continue
        dens = "Lb1E" in name and "fwd" in k                      # density-only instantiation (occupancy refresh)
        agg.setdefault((k, dens), {}).setdefault(pn, []).append(v)
    flop, peak = (16384.0, 2500.0) if fp16 else (2048.0, 157.3)
    try:
        res = json.load(open(out_json))
    except Exception:
        res = {}
    res.setdefault(key, {})
    lines = []
    for (k, dens), cnt in sorted(agg.items()):
        g = lambda n: (sum(cnt[n][-32:]) / len(cnt[n][-32:])) if n in cnt else 0.0
        mfma, busy, sqbusy = g("SQ_INSTS_MFMA"), g("SQ_VALU_MFMA_BUSY_CYCLES"), g("SQ_BUSY_CYCLES")
        if not sqbusy:
            continue
        dur_cyc = sqbusy / 32.0
        util = busy / 1024.0 / dur_cyc
        k_flop, k_peak = (16384.0, 2500.0) if "split" in k else (flop, peak)          # the split-operand kernels of the fp32 configuration issue fp16 16x16x32 MFMAs (csrc/field_split.hip)
        tf = mfma * k_flop / (dur_cyc / 2.4e9) / 1e12
        lines.append(f"| {key} | `{k}`{' <density only: occupancy refresh>' if dens else ''} | {mfma:,.0f} | {busy:,.0f} | {sqbusy:,.0f} | {dur_cyc / 1e3:.0f} | {100 * util:.1f} % | {tf:.0f} | {100 * tf / k_peak:.1f} % |")
        if not dens:
            res[key].setdefault(k, {}).update({"mfma_instructions_per_launch": int(mfma), "mfma_pipe_util": round(util, 4), "mfma_issued_frac": round(tf / k_peak, 4)})
    json.dump(res, open(out_json, "w"), indent=1)
    hdr = ("| config | kernel | MFMA instructions / launch | SQ_VALU_MFMA_BUSY_CYCLES | SQ_BUSY_CYCLES | duration (k cycles) | MFMA pipe utilisation | TFLOP/s | fraction of dense peak |\n|---|---|---|---|---|---|---|---|---|\n")
    try:
        old = open(out_md).read()
    except Exception:
        old = ("# MFMA counters of the fused field-network kernels\n\nsource: `rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES "
               "--kernel-trace -- python bench.py --no-cpu-baseline --no-psnr --no-fox --config {lego|fox} --steps 32 --warmup 16` (own pass, no other tracing), average over the last 32 launches, counter "
               "summed over all XCD / SE instances of a launch.\nMFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over SQ_BUSY_CYCLES / 32 instances; fraction of the dense peak = MFMA instructions x FLOP per "
               "instruction (fp32 16x16x4: 2048, fp16 16x16x32: 16384) / duration at 2.4 GHz / peak (157.3 | 2500 TFLOP/s) - the EXECUTED figure (it includes the backward's forward recompute).\n\n" + hdr)
    open(out_md, "w").write(old + "\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:6])
