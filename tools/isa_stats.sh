#!/bin/bash
# usage: bash tools/isa_stats.sh  - per kernel of csrc/rt_amd.hip (gfx950 code object): instructions, packed float32
# instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), VGPRs, scratch.  Runs without a GPU.
set -e
cd "$(dirname "$0")/../ray-tracing_amd/csrc"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only -o /tmp/rt_amd_isa.s rt_amd.hip 2>/dev/null
/opt/rocm/bin/hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c -o /tmp/rt_amd_isa.o rt_amd.hip 2>/tmp/rt_amd_isa.txt || true
python3 - <<'PY'
import re, subprocess
name = None
ins, pk = {}, {}
for line in open("/tmp/rt_amd_isa.s"):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name = m.group(1)
        ins[name] = 0; pk[name] = 0
        continue
    if name and re.match(r"^\s+[vsdgb]\w*_", line):
        ins[name] += 1
        if re.match(r"^\s+v_pk_(fma|mul|add)_f32", line):
            pk[name] += 1
    if name and line.startswith(".Lfunc_end"):
        name = None
res = {}
cur = None
for line in open("/tmp/rt_amd_isa.txt"):
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); res[cur] = {}
    m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", line)
    if m and cur: res[cur][m.group(1).split()[0]] = int(m.group(2))
def dem(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
print("%-36s %8s %10s %6s %8s" % ("kernel", "instrs", "v_pk_*_f32", "VGPRs", "scratch"))
tot = 0
for n in sorted(ins, key=lambda k: -ins[k]):
    if not n.startswith("_Z") or "k_" not in n: continue
    r = res.get(n, {})
    tot += pk[n]
    print("%-36s %8d %10d %6s %8s" % (dem(n)[:36], ins[n], pk[n], r.get("VGPRs", "-"), r.get("ScratchSize", "-")))
print("packed float32 instructions in all kernels: %d" % tot)
PY
