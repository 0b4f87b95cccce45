"""Lab build of decode_persist.hip with per-role time stamps (NOT part of the product): every role's lane 0 records the
100-MHz wall clock when its wait ends, before and after its arrival, into a buffer handed over through acvae_pd_trace().
usage: python tools/lab_pd.py build     (here)      python tools/lab_pd.py run      (GPU box)"""
import os, subprocess, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "acvae_amd", "csrc"); LAB = os.path.join(ROOT, "tools", "lab")
LIB = os.path.join(LAB, "libacvae_pdtrace.so")


def build():
    s = open(os.path.join(CSRC, "decode_persist.hip")).read()
    s = s.replace("struct PdSmem {", "__device__ unsigned long long* g_pd_trace = nullptr;\n#define PD_TR(slot) do { if (g_pd_trace && threadIdx.x == 0) { "
                  "g_pd_trace[((long)blockIdx.x * 32 + t) * 8 + (slot)] = wall_clock64(); } } while (0)\nstruct PdSmem {")
    out, role, waits, skip = [], None, 0, False
    for l in s.split("\n"):
        if l.startswith("__device__ void role_") or (l.startswith("template <bool RES>")):
            waits = 0; skip = False
        if l.startswith("__global__"):
            skip = "posterior_persist" in l          # the posterior's launches run beside the traced ones: no stamps from them
        if skip:
            out.append(l)
            continue
        if "pd_arrive(" in l and "__device__" not in l:
            out += ["    PD_TR(6);", l, "    PD_TR(7);"]
            continue
        out.append(l)
        if "pd_wait(" in l and "__device__" not in l and "for (int t" not in l:
            out.append("    PD_TR(%d);" % min(waits, 5)); waits += 1
    s = "\n".join(out)
    s = s.replace('extern "C" int acvae_persist_status_register(int device, void* status_words_host) {', 'extern "C" int acvae_pd_trace(unsigned long long* buf) { return (int)hipMemcpyToSymbol('
                  'HIP_SYMBOL(g_pd_trace), &buf, sizeof(buf)); }\nextern "C" int acvae_persist_status_register(int device, void* status_words_host) {')
    s = s.replace('#include "../../include/acvae_hip.h"', f'#include "{ROOT}/include/acvae_hip.h"')
    os.makedirs(LAB, exist_ok=True)
    src = os.path.join(LAB, "decode_persist_trace.hip"); open(src, "w").write(s)
    obj = os.path.join(LAB, "decode_persist_trace.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-I" + CSRC, "-c", src, "-o", obj])
    objs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".o") and f != "decode_persist.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, obj, *objs])


def run():
    os.environ["ACVAE_DEV_LIB"] = LIB       # this tool's own variable; handed to the package with use_library() below
    sys.path.insert(0, ROOT)
    import numpy as np, torch, random
    import bench
    from acvae_amd import _lib
    if os.environ.get("ACVAE_DEV_LIB"):
        _lib.use_library(os.environ["ACVAE_DEV_LIB"])
    model = bench.build_model().cuda().train()
    feats, caps, fl, cl = bench.synthetic(1)
    f = feats.cuda(); Tc = 21
    roles = {"D1q": (0, 16), "D1h": (16, 64), "D2": (64, 96), "D3": (96, 128), "P1": (128, 192), "P2": (192, 224)}
    nblk = 224
    buf = torch.zeros(nblk * 32 * 8, dtype=torch.int64, device="cuda")
    lib = _lib.lib(); lib.acvae_pd_trace.argtypes = [ctypes.c_void_p]
    for rep in range(4):
        random.seed(0); model(f, fl.copy(), caps, cl, ss_ratio=1.0, dis_ratio=0); torch.cuda.synchronize()
        if rep == 2: lib.acvae_pd_trace(ctypes.c_void_p(buf.data_ptr()))
    lib.acvae_pd_trace(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(nblk, 32, 8)[:, :Tc, :].astype(np.float64) * 0.01
    t0 = t[t > 0].min(); t = np.where(t > 0, t - t0, np.nan)
    print("span %.1f us" % np.nanmax(t))
    for name, (a, b) in roles.items():
        x = t[a:b, 2:, :]
        lastw = np.nanmax(x[:, :, :6], axis=2)              # the role's last wait of the step
        print(f"{name}: last wait end -> before arrive {np.nanmean(x[:, :, 6] - lastw):.2f} us (slowest wg {np.nanmean(np.nanmax(x[:, :, 6] - lastw, axis=0)):.2f}); arrive {np.nanmean(x[:, :, 7] - x[:, :, 6]):.2f} us")
    d3 = np.nanmax(t[96:128, :, 7], axis=0); p2 = np.nanmax(t[192:224, :, 7], axis=0)
    print("decoder chain step period %.2f us, prior chain %.2f us" % (np.diff(d3)[2:].mean(), np.diff(p2)[2:].mean()))
    # critical path of a decoder step: D3(t-1) done -> D1q done -> D2 done -> D3 done
    d1q = np.nanmax(t[0:16, :, 7], axis=0); d2 = np.nanmax(t[64:96, :, 7], axis=0)
    print("D3(t-1)->D1q %.2f | D1q->D2 %.2f | D2->D3 %.2f us" % ((d1q[3:] - d3[2:-1]).mean(), (d2[3:] - d1q[3:]).mean(), (d3[3:] - d2[3:]).mean()))


def run_bwd():
    os.environ["ACVAE_DEV_LIB"] = LIB       # this tool's own variable; handed to the package with use_library() below
    sys.path.insert(0, ROOT)
    import numpy as np, torch, random
    import bench
    from acvae_amd import _lib
    if os.environ.get("ACVAE_DEV_LIB"):
        _lib.use_library(os.environ["ACVAE_DEV_LIB"])
    from acvae_amd.trainer import TrainStep
    model = bench.build_model().cuda().train()
    ts = TrainStep(model, bench.V)
    feats, caps, fl, cl = bench.synthetic(1)
    f = feats.cuda(); Tc = 21
    ks_rb, ks_pa = int(os.environ.get("KS_RB", "3")), int(os.environ.get("KS_PA", "4"))
    b0 = 16; b1 = b0 + 16 * ks_rb; b2 = b1 + 32; b3 = b2 + 32 * ks_pa; b4 = b3 + 16
    roles = {"RA": (0, b0), "RB": (b0, b1), "RC": (b1, b2), "PA": (b2, b3), "PB": (b3, b4)}
    nblk = b4
    buf = torch.zeros(nblk * 32 * 8, dtype=torch.int64, device="cuda")
    lib = _lib.lib(); lib.acvae_pd_trace.argtypes = [ctypes.c_void_p]
    for rep in range(4):
        random.seed(0)
        loss, parts, _ = ts.forward_loss(f, fl.copy(), caps, cl, 1.0, 0, 0.5); torch.cuda.synchronize()
        if rep == 3: lib.acvae_pd_trace(ctypes.c_void_p(buf.data_ptr()))
        loss.backward(); torch.cuda.synchronize()
    lib.acvae_pd_trace(ctypes.c_void_p(0))
    t = buf.cpu().numpy().reshape(nblk, 32, 8)[:, :Tc, :].astype(np.float64) * 0.01
    t0 = t[t > 0].min(); t = np.where(t > 0, t - t0, np.nan)
    print("span %.1f us" % np.nanmax(t))
    for name, (a, b) in roles.items():
        x = t[a:b, 1:Tc - 1, :]
        lastw = np.nanmax(x[:, :, :6], axis=2)
        firstw = np.nanmin(x[:, :, :6], axis=2)
        print(f"{name}: first wait end -> last wait end {np.nanmean(lastw - firstw):.2f}; last wait end -> before arrive {np.nanmean(x[:, :, 6] - lastw):.2f} us (slowest wg {np.nanmean(np.nanmax(x[:, :, 6] - lastw, axis=0)):.2f}); arrive {np.nanmean(x[:, :, 7] - x[:, :, 6]):.2f} us")
    ra = np.nanmax(t[0:b0, :, 7], axis=0); rb = np.nanmax(t[b0:b1, :, 7], axis=0); rc = np.nanmax(t[b1:b2, :, 7], axis=0)
    pa = np.nanmax(t[b2:b3, :, 7], axis=0); pb = np.nanmax(t[b3:b4, :, 7], axis=0)
    # time runs with DEcreasing t
    print("decoder chain step period %.2f us: RC(t+1)->RA(t) %.2f | RA->RB %.2f | RB->RC %.2f" % (
        (ra[1:-2] - ra[2:-1]).mean(), (ra[1:-2] - rc[2:-1]).mean(), (rb[1:-2] - ra[1:-2]).mean(), (rc[1:-2] - rb[1:-2]).mean()))
    print("prior chain step period %.2f us: PB(t+1)->PA(t) %.2f | PA->PB %.2f" % ((pb[1:-2] - pb[2:-1]).mean(), (pa[1:-2] - pb[2:-1]).mean(), (pb[1:-2] - pa[1:-2]).mean()))


if __name__ == "__main__":
    {"build": build, "run": run, "bwd": run_bwd}[sys.argv[1]]()
