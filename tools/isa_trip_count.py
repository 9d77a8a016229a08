"""Vector / matrix / LDS instruction counts of ONE trip of a kernel's main loop, from the compiler's assembly (no GPU): the largest loop of the kernel is taken as the main loop,
loops nested in it are weighted by `inner` iterations.  The counts behind DESIGN.md section 9.0 (split backward 1366 -> 746 vector instructions per trip, ...).

usage: python tools/isa_trip_count.py jnerf_amd/csrc/field_split.hip k_field32_bwd_splitILi1ELi0ELb1E [inner iterations, default 2] [extra hipcc flags ...]
       (the second argument is a substring of the MANGLED kernel name; inner = 2 for the split kernel's loops unrolled by two over four k steps, 4 for the fp16 kernel's)"""
import collections
import re
import subprocess
import sys
import tempfile

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-fvisibility=hidden", "-S", "--cuda-device-only"]


def kernel_body(asm, pat):
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and l.split(":")[0].endswith(l.split(":")[0]) and ":" in l and not l.startswith("\t"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start].split(":")[0], lines[start:end]


def count(body, inner):
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    loops.sort(key=lambda x: x[0] - x[1])                      # widest first
    main = loops[0]
    inners = [lp for lp in loops[1:] if lp[0] > main[0] and lp[1] < main[1]]
    w = [0] * len(body)
    for i in range(main[0], main[1] + 1):
        w[i] = 1
    for a, b in inners:
        for i in range(a, b + 1):
            w[i] = inner
    cnt = collections.Counter()
    for i, l in enumerate(body):
        t = l.split()
        if w[i] and t and not t[0].startswith((".", ";")) and not t[0].endswith(":"):
            cnt[t[0]] += w[i]
    return main, len(inners), cnt


def main():
    src, pat = sys.argv[1], sys.argv[2]
    inner = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *sys.argv[4:], src, "-o", f.name], capture_output=True, text=True)
        asm = open(f.name).read()
    if not asm:
        sys.exit(r.stderr[-2000:])
    name, body = kernel_body(asm, pat)
    (a, b), n_inner, cnt = count(body, inner)
    tot = lambda pred: sum(v for k, v in cnt.items() if pred(k))
    print(name)
    print(f"main loop: assembly lines {a}..{b} of the kernel, {n_inner} inner loops x {inner}")
    print(f"vector ALU {tot(lambda k: k.startswith('v_') and not k.startswith('v_mfma'))}   MFMA {tot(lambda k: k.startswith('v_mfma'))}   LDS {tot(lambda k: k.startswith('ds_'))}   "
          f"global {tot(lambda k: k.startswith(('global_', 'buffer_', 'scratch_')))}   scalar / other {tot(lambda k: k.startswith('s_'))}")
    for k, v in cnt.most_common(30):
        print(f"  {v:5d} {k}")


if __name__ == "__main__":
    main()
