"""Builds the HIP library of another revision into tools/lab/libacvae_<name>.so (A/B timing inside ONE gpurun call - two boxes
differ by several per cent): `python tools/ab_build.py <git-rev> <name>`, then on the GPU box
`python bench.py ...` against `python bench.py --lib tools/lab/libacvae_<name>.so ...` (tools/ab_bench.sh)."""
import os, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rev, name = sys.argv[1], sys.argv[2]
tmp = tempfile.mkdtemp(prefix="ab_")
subprocess.check_call(f"git -C {ROOT} archive {rev} acvae_amd/csrc include | tar -x -C {tmp}", shell=True)
csrc = os.path.join(tmp, "acvae_amd", "csrc")
flags = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-function", "-ffp-contract=off"]
srcs = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
def cc(f):
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", os.path.join(csrc, f), "-o", os.path.join(csrc, f[:-4] + ".o")])
with ThreadPoolExecutor(max_workers=6) as ex:
    list(ex.map(cc, srcs))
out = os.path.join(ROOT, "tools", "lab", f"libacvae_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [os.path.join(csrc, f[:-4] + ".o") for f in srcs])
print(out)
