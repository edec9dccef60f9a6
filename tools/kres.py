"""Resource usage of the kernels of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
usage: python tools/kres.py tomobar_amd/csrc/tv_kernels.hip [name-regex] [extra hipcc flags ...]"""
import re
import subprocess
import sys
import os

src = sys.argv[1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
extra = sys.argv[3:]
d = os.path.dirname(os.path.abspath(src))
inc = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "repo", "include")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-I" + os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include"),
       "-Rpass-analysis=kernel-resource-usage", "-c", os.path.basename(src), "-o", "/tmp/kres.o"] + extra
out = subprocess.run(cmd, cwd=d, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    if "error" in line:
        print(line)
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[bytes/(?:lane|block)\]| \[waves/SIMD\])?: (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name).replace("void ", "")
    if pat and not pat.search(name):
        continue
    print(f"{name:90s} vgpr {r.get('VGPRs','?'):>3} agpr {r.get('AGPRs','?'):>3} scratch {r.get('ScratchSize','?'):>4} "
          f"sgpr {r.get('TotalSGPRs','?'):>3} sspill {r.get('SGPRs Spill','?'):>3} occ {r.get('Occupancy','?')} lds {r.get('LDS Size','?')}")
