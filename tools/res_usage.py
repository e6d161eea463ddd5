"""Register / spill / occupancy table of one translation unit's kernels (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/res_usage.py gemm_bf16_ws_ring.hip [-DMF_XTAP=0 ...]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{root}/include", "-Wno-inline-asm", *sys.argv[2:],
       "-Rpass-analysis=kernel-resource-usage", "-c", f"{root}/reflecting-reality_amd/csrc/{src}", "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: (?:Function Name: )?(.*) \[-Rpass-analysis", line)
    if not m:
        continue
    t = m.group(1).strip()
    if "Function Name" in line:
        cur = t
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"^void mfgemm::", "", short).replace("(mfgemm::GemmArgs)", "")
    print(f"{short[:95]:95s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>3} sgpr {r.get('SGPRs','?'):>3} spill {r.get('VGPR Spill', r.get('VGPRs Spill','?')):>3} "
          f"scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?')} lds {r.get('LDS Size [bytes/block]','?')}")
