"""Build libacvae_hip.so in-tree with hipcc for gfx950 (no cmake, no torch extension machinery:
the library has a plain C ABI and links only against the HIP runtime)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libacvae_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]   # no implicit FMA contraction: keeps fp32 results reproducible vs the oracle
if os.environ.get("ACVAE_EXACT_TANH") == "1":      # parity-debug build: the attention's tanh from the math library (common.h)
    FLAGS.append("-DACVAE_EXACT_TANH")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def parse_usage(text):
    """{kernel symbol: {"vgprs": n, "sgprs": n, "scratch": bytes per lane, "occupancy": waves per SIMD}} from the remarks of
    -Rpass-analysis=kernel-resource-usage."""
    import re
    out, name = {}, None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
            continue
        for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("sgprs", r" SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and name:
                out[name][key] = int(m.group(1))
    return out


def resource_usage():
    """Register / scratch use of every kernel of the library, as recorded when its object file was compiled."""
    usage = {}
    for s in sources():
        path = os.path.join(CSRC, s[:-4] + ".usage.txt")
        if os.path.exists(path):
            usage.update(parse_usage(open(path).read()))
    return usage


def build(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "acvae_hip.h"))
    objs, jobs = [], []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s[:-4] + ".o")
        objs.append(obj)
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdrs) or not os.path.exists(obj[:-2] + ".usage.txt"):
            jobs.append([HIPCC, *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if "-c" in cmd:                      # keep the compiler's per-kernel resource remarks beside the object file
            remarks = "\n".join(l for l in r.stderr.splitlines() if "kernel-resource-usage" in l)
            open(cmd[-1][:-2] + ".usage.txt", "w").write(remarks)
            import re as _re
            rest = "\n".join(l for l in r.stderr.splitlines() if "kernel-resource-usage" not in l and
                             not _re.match(r"^\s*(\d+)?\s*\|", l)).strip()      # (the remarks' source-context lines)
            if verbose:
                for k, u in sorted(parse_usage(remarks).items()):
                    if u.get("scratch", 0) or u.get("vgprs", 0) >= 200:
                        print(f"  {k[:90]}: {u.get('vgprs')} VGPRs, {u.get('scratch')} B/lane scratch, occupancy {u.get('occupancy')}")
                if rest:
                    print(rest)
        elif verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
