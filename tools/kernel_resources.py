"""Print per-kernel register / LDS / occupancy figures of one .hip file (hipcc remarks).

    python tools/kernel_resources.py <file.hip> [substring of the demangled name]
"""
import re, subprocess, sys
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc","-O3","-std=c++17","--offload-arch=gfx950","-c",src,"-o","/tmp/probe/x.o","-Rpass-analysis=kernel-resource-usage"],capture_output=True,text=True).stderr
cur = None; rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([\w ]+?)(?: \[[^\]]*\])?: (\S+) \[", line)
    if m and cur: rows[cur][m.group(1)] = m.group(2)
for k, v in rows.items():
    d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    if pat and pat not in d: continue
    print(d[:110], {a: v.get(a) for a in ("VGPRs","AGPRs","ScratchSize","Occupancy","SGPRs","LDS Size")})
