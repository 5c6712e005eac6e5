#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in prt_kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage),
one line per kernel.  Developer tool: `python tools/resources.py [-DPRT_X=1 ...] [--all]`."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pooraytracer_amd", "csrc")


def main():
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    show_all = "--all" in sys.argv
    extra = [a for a in extra if a != "--all"]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-c", "-Wno-unused-function",
           "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_prt_res.o", os.path.join(CSRC, "prt_kernels.hip")] + extra
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode:
        sys.stderr.write(p.stderr)
        raise SystemExit(p.returncode)
    cur = None
    rows = []
    for line in p.stderr.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            m = re.search(r":\d+:\d+:\s+(.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:") or t.startswith("Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        if not show_all and not name.startswith(("k_render<false", "k_trace_closest<false")):
            continue
        print(f"{name:36s} VGPR {r.get('VGPRs', '?'):>4s} AGPR {r.get('AGPRs', '?'):>3s} spillV {r.get('VGPRs Spill', '?'):>3s} "
              f"spillS {r.get('SGPRs Spill', '?'):>3s} scratch {r.get('ScratchSize [bytes/lane]', '?'):>4s} "
              f"occ {r.get('Occupancy [waves/SIMD]', '?')} LDS {r.get('LDS Size [bytes/block]', '?')}")


if __name__ == "__main__":
    main()
