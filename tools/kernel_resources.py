"""Register / LDS / occupancy table of the kernels in one csrc/*.hip file (device-only compile with -Rpass-analysis=kernel-resource-usage; no GPU needed).
usage: python tools/kernel_resources.py jnerf_amd/csrc/hash_encode.hip [name-substring]"""
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-fvisibility=hidden",
           "--cuda-device-only", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: (?:\S+ )?\s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'occ':>4s} {'scratch':>7s} {'LDS':>7s}")
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
        if pat and pat not in name:
            continue
        print(f"{name[:70]:70s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} {r.get('Occupancy [waves/SIMD]', '?'):>4s} "
              f"{r.get('ScratchSize [bytes/lane]', '?'):>7s} {r.get('LDS Size [bytes/block]', '?'):>7s}")


if __name__ == "__main__":
    main()
