"""Lab builds of conv_wino.hip (NOT part of the product): applies named source patches (ablations / scheduling variants) to
acvae_amd/csrc/conv_wino.hip, compiles the result and links it with the product's other objects into
tools/lab/libacvae_<name>.so; on the GPU box `python tools/lab_wino.py time <name>...` times the Winograd kernels of each
library (ACVAE_DEV_LIB) at three layer shapes.  Results of ablations are wrong by construction.
usage: python tools/lab_wino.py build <name>... | time <name>... | list"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "acvae_amd", "csrc")
LAB = os.path.join(ROOT, "tools", "lab")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-I" + CSRC]

PRE_LOOP = "  using std::integral_constant;\n  using std::false_type;\n  __builtin_amdgcn_s_setprio(0);"
FAKE_D = ("#pragma unroll\n    for (int j = 0; j < 4; ++j) asm volatile(\"\" : \"=v\"(DST[j].x), \"=v\"(DST[j].y), \"=v\"(DST[j].z), "
          "\"=v\"(DST[j].w));")
VARIANTS = {
    "base": [],
    "nob": [(f"    B[{i}] = load_b({c}, {g});\n", "") for (i, c, g) in
            [(0, "c", 4), (1, "c", 5), (2, "c", 6), (3, "c", 7), (0, "cn", 0), (1, "cn", 1), (2, "cn", 2), (3, "cn", 3)]],
    "noraw": [("    if (sub == 0) put_raw_part(rnxt, 0);\n", ""), ("    if (sub == 0) put_raw_part(rnxt, 2);\n", ""),
              ("    if (sub == 0) issue_raw(st2);", "    (void)st2; (void)rnxt;"), ("    if (sub == 0) read_scsh(st1);\n", "    (void)st1;\n")],
    "nobar": [("    if (sub == 0) __syncthreads();\n  };", "  };")],
    "nod": [("    read_row(nq, rowa, da);\n", "    (void)nq;\n" + FAKE_D.replace("DST", "da") + "\n"),
            ("    read_row(nq, rowb, db);\n", FAKE_D.replace("DST", "db") + "\n")],
}
# phase stamps (s_memtime): [workgroup][wave][4] = prologue, main loop, epilogue, chunks; written over the start of Y after the epilogue
VARIANTS["stamps"] = [
    ("  const int tpos = wino_xcd_tile(p, gridDim.x, gridDim.y, bm, bn);\n  const int RW = wino_row_pitch(p.tw_shift), R = p.R;",
     "  const int tpos = wino_xcd_tile(p, gridDim.x, gridDim.y, bm, bn);\n  const long long lab_t0 = clock64();\n  const int RW = wino_row_pitch(p.tw_shift), R = p.R;"),
    (PRE_LOOP, "  const long long lab_t1 = clock64();\n" + PRE_LOOP),
    ("  // ---------------------------------------------------------------- epilogue\n  float2* exb",
     "  const long long lab_t2 = clock64();\n  float2* exb"),
    ("                           li_e, h_e, lane_e, n, ty0, bm, bn);\n  if (!ACT) asm volatile(\"\" :: \"v\"(pfv));      // the prefetched value is dropped here: the load stays in the program, its register reserved\n}\n\n// ACT: the operand carries",
     "                           li_e, h_e, lane_e, n, ty0, bm, bn);\n  asm volatile(\"\" :: \"v\"(pfv));\n  __syncthreads();\n  const int wave = XI;\n  if (lane == 0) {\n    float* o = p.Y + ((long)(blockIdx.x + blockIdx.y * gridDim.x) * 8 + wave) * 4;\n"
     "    o[0] = (float)(lab_t1 - lab_t0); o[1] = (float)(lab_t2 - lab_t1); o[2] = (float)(clock64() - lab_t2); o[3] = (float)nchunk;\n  }\n}\n\n// ACT: the operand carries"),
]
# in-kernel clock: s_memtime (shader cycles) against s_memrealtime (100 MHz) around the main loop -> [workgroup][wave][4] =
# cycles, 100-MHz ticks, chunks, 0 (MI355X_MICROARCH.md, DVFS give-back (6))
VARIANTS["clock"] = [
    (PRE_LOOP, "  const long long lab_c0 = clock64(), lab_r0 = wall_clock64();\n" + PRE_LOOP),
    ("  // ---------------------------------------------------------------- epilogue\n  float2* exb",
     "  const long long lab_c1 = clock64(), lab_r1 = wall_clock64();\n  float2* exb"),
    ("                           li_e, h_e, lane_e, n, ty0, bm, bn);\n  if (!ACT) asm volatile(\"\" :: \"v\"(pfv));      // the prefetched value is dropped here: the load stays in the program, its register reserved\n}\n\n// ACT: the operand carries",
     "                           li_e, h_e, lane_e, n, ty0, bm, bn);\n  asm volatile(\"\" :: \"v\"(pfv));\n  __syncthreads();\n  const int wave = XI;\n  if (lane == 0) {\n    float* o = p.Y + ((long)(blockIdx.x + blockIdx.y * gridDim.x) * 8 + wave) * 4;\n"
     "    o[0] = (float)(lab_c1 - lab_c0); o[1] = (float)(lab_r1 - lab_r0); o[2] = (float)nchunk; o[3] = 0.f;\n  }\n}\n\n// ACT: the operand carries"),
]
# timeline + phases in one launch (output stores kept): wave 0 = start, end, CU key, chunks; wave 1 = main loop start, end (100-MHz
# ticks, low 24 bits), main-loop shader cycles, 0
VARIANTS["full"] = [
    ("  const int tpos = wino_xcd_tile(p, gridDim.x, gridDim.y, bm, bn);\n  const int RW = wino_row_pitch(p.tw_shift), R = p.R;",
     "  const int tpos = wino_xcd_tile(p, gridDim.x, gridDim.y, bm, bn);\n  const long long lab_r0 = wall_clock64();\n  const int RW = wino_row_pitch(p.tw_shift), R = p.R;"),
    (PRE_LOOP, "  const long long lab_c1 = clock64(), lab_r1 = wall_clock64();\n" + PRE_LOOP),
    ("  // ---------------------------------------------------------------- epilogue\n  float2* exb",
     "  const long long lab_c2 = clock64(), lab_r2 = wall_clock64();\n  float2* exb"),
    ("                           li_e, h_e, lane_e, n, ty0, bm, bn);\n  if (!ACT) asm volatile(\"\" :: \"v\"(pfv));      // the prefetched value is dropped here: the load stays in the program, its register reserved\n}\n\n// ACT: the operand carries",
     "                           li_e, h_e, lane_e, n, ty0, bm, bn);\n  asm volatile(\"\" :: \"v\"(pfv));\n  asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");\n  __syncthreads();\n  const int wave = XI;\n  if (lane == 0 && wave < 2) {\n    float* o = p.lab + ((long)(blockIdx.x + blockIdx.y * gridDim.x) * 3 + wave) * 4;\n"
     "    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);\n"
     "    if (wave == 0) { o[0] = (float)(lab_r0 & 0xffffff); o[1] = (float)(wall_clock64() & 0xffffff); o[2] = (float)(((xcc & 15) << 8) | ((hw >> 8) & 255)); o[3] = (float)nchunk; }\n"
     "    else { o[0] = (float)(lab_r1 & 0xffffff); o[1] = (float)(lab_r2 & 0xffffff); o[2] = (float)(lab_c2 - lab_c1); o[3] = 0.f; }\n  }\n}\n\n// ACT: the operand carries"),
]
VARIANTS["full"] += [
    ("#pragma unroll\n  for (int g = 0; g < 4; ++g) B[g] = load_b(0, g);\n  issue_raw(0);\n  stage_scsh();\n  read_scsh(0);\n  put_raw(raw0);\n",
     "  const long long lab_ra = wall_clock64();\n#pragma unroll\n  for (int g = 0; g < 4; ++g) B[g] = load_b(0, g);\n  issue_raw(0);\n  stage_scsh();\n  read_scsh(0);\n  put_raw(raw0);\n"
     "  asm volatile(\"s_waitcnt vmcnt(0) lgkmcnt(0)\" ::: \"memory\");\n  const long long lab_rb = wall_clock64();\n"),
    ("    else { o[0] = (float)(lab_r1 & 0xffffff);", "    if (wave == 2) { o[0] = (float)(lab_ra & 0xffffff); o[1] = (float)(lab_rb & 0xffffff); o[2] = 0.f; o[3] = 0.f; }\n    else if (wave == 1) { o[0] = (float)(lab_r1 & 0xffffff);"),
    ("if (lane == 0 && wave < 2) {", "if (lane == 0 && wave < 3) {"),
    ("  float* partials;      // [blocks][2][Cout] or nullptr", "  float* partials;\n  float* lab;"),
    ("p.Y = Y; p.partials = partials;", "p.Y = Y; p.partials = partials; p.lab = g_lab;"),
    ("constexpr int WN_THREADS = 256;\n", "float* g_lab = nullptr;\nconstexpr int WN_THREADS = 256;\n"),
]
NOPF = [("      pfv = p.X[((long)(n2 * H + y) * W + x) * C];        // default cache policy: the line is to stay in L2", "      pfv = 1.f;")]
VARIANTS["fullnopf"] = VARIANTS["full"] + NOPF
VARIANTS["mfmaonly"] = VARIANTS["nob"] + VARIANTS["noraw"] + VARIANTS["nod"]
# no transforms either: what the bare MFMA stream (plus barrier and loop control) takes
NOTRANS = [("    vertical(da, db);\n", ""),
           ("    wino_htrans_ip(da, vn);\n", "#pragma unroll\n    for (int j = 0; j < 4; ++j) vn[j] = da[j];\n")]
VARIANTS["puremfma"] = VARIANTS["mfmaonly"] + NOTRANS
VARIANTS["notrans"] = NOTRANS
# scalar v_add_f32 / v_sub_f32 instead of the packed forms (twice the instructions)
VARIANTS["scalar"] = [
    ('  asm(WN_PK_SUB("%0", "%0", "%2") WN_PK_SUB("%1", "%1", "%3") : "+v"(lo), "+v"(hi) : "v"(f32x2{b.x, b.y}), "v"(f32x2{b.z, b.w}));\n  a = make_float4(lo[0], lo[1], hi[0], hi[1]);',
     '  (void)lo; (void)hi; a = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);'),
    ('  asm(WN_PK_ADD("%0", "%0", "%2") WN_PK_ADD("%1", "%1", "%3") : "+v"(lo), "+v"(hi) : "v"(f32x2{b.x, b.y}), "v"(f32x2{b.z, b.w}));\n  a = make_float4(lo[0], lo[1], hi[0], hi[1]);',
     '  (void)lo; (void)hi; a = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);'),
    ('  f32x2 n0, n1;\n  asm(WN_PK_SUB("%0", "%0", "%2") WN_PK_SUB("%1", "%1", "%3")          // t0 - t2',
     '  f32x2 n0, n1;\n  { const float4 T0 = t[0], T1 = t[1], T2 = t[2], T3 = t[3];\n'
     '    v[0] = make_float4(T0.x - T2.x, T0.y - T2.y, T0.z - T2.z, T0.w - T2.w); v[1] = make_float4(T1.x + T2.x, T1.y + T2.y, T1.z + T2.z, T1.w + T2.w);\n'
     '    v[2] = make_float4(T2.x - T1.x, T2.y - T1.y, T2.z - T1.z, T2.w - T1.w); v[3] = make_float4(T1.x - T3.x, T1.y - T3.y, T1.z - T3.z, T1.w - T3.w);\n'
     '    (void)a0; (void)a1; (void)c0; (void)c1; (void)e0; (void)e1; (void)n0; (void)n1; return; }\n'
     '  asm(WN_PK_SUB("%0", "%0", "%2") WN_PK_SUB("%1", "%1", "%3")          // t0 - t2'),
]
# no epilogue at all (the accumulators are kept alive, nothing is exchanged or stored): what the epilogue costs a launch
VARIANTS["noepi"] = [
    ("  wino_epilogue<XI, STATS>(p, acc, exb, reinterpret_cast<float*>(raw0 + WN_EX_F4),\n                           li_e, h_e, lane_e, n, ty0, bm, bn);",
     "#pragma unroll\n  for (int i_ = 0; i_ < 8; ++i_)\n#pragma unroll\n    for (int r_ = 0; r_ < 16; ++r_) asm volatile(\"\" :: \"v\"(acc[i_][r_]));\n"
     "  (void)exb; (void)li_e; (void)h_e; (void)n; (void)ty0;"),
]
VARIANTS["noepi_puremfma"] = VARIANTS["noepi"] + VARIANTS["puremfma"]
# epilogue without its global stores (exchange and arithmetic kept)
VARIANTS["nostore"] = [
    ("    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o0), yrs, voff[k], soff[j], 0);\n    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o1), yrs, voff[k], soff[j] + pix, 0);\n",
     "    asm volatile(\"\" :: \"v\"(o0), \"v\"(o1), \"v\"(voff[k]), \"s\"(soff[j]));\n    (void)yrs; (void)pix;\n"),
]
# ... and without the LDS exchange either (each wavefront combines its own pairs three times): arithmetic + stores only
VARIANTS["noexch"] = [
    ("        if (SET_OWN >= 0) ex[(SET_OWN * 16 + r) * 64 + lane] = f;\n", ""),
    ("        ex[(SET_OTHER * 16 + r) * 64 + lane] = f;\n", "        asm volatile(\"\" :: \"v\"(f.x), \"v\"(f.y));\n"),
    ("  __syncthreads();\n  // descriptor of clip n", "  // descriptor of clip n"),
    ("    const float2 a = ex[(SET_A * 16 + r) * 64 + lane], b = ex[(SET_B * 16 + r) * 64 + lane];", "    const float2 a = keep[(r + 1) & 15], b = keep[(r + 2) & 15]; (void)ex;"),
]
# prologue ablations: no zero-fill of padding slots / no first-stage wait
VARIANTS["nozero"] = [
    ("  for (int i = tid; i < (2 * R + 2) * 8; i += WN_THREADS) {", "  for (int i = tid; i < 0; i += WN_THREADS) {"),
    ("    if (lv && !ok) {\n      raw0[slot] = make_float4(0.f, 0.f, 0.f, 0.f);\n      raw1[slot] = make_float4(0.f, 0.f, 0.f, 0.f);\n    }\n", "    (void)slot;\n"),
]
VARIANT_FLAGS = {"scalar": ["-fno-slp-vectorize"]}
VARIANTS["fullprio0"] = VARIANTS["full"]
VARIANT_FLAGS["fullprio0"] = ["-DWN_PRIO_EDGE=0"]
for _d in (16, 32, 48, 96):                      # prefetch distance in the XCD's run of tiles
    VARIANTS[f"fullpf{_d}"] = VARIANTS["full"]
    VARIANT_FLAGS[f"fullpf{_d}"] = [f"-DWN_PF_DIST={_d}"]


def build(name):
    src = open(os.path.join(CSRC, "conv_wino.hip")).read()
    for old, new in VARIANTS[name]:
        if old not in src:
            raise SystemExit(f"{name}: patch anchor not found:\n{old}")
        src = src.replace(old, new)
    os.makedirs(LAB, exist_ok=True)
    cpp = os.path.join(LAB, f"conv_wino_{name}.hip")
    if name.startswith("full"):
        src += '\nextern "C" void acvae_lab_set(float* p) { g_lab = p; }\n'
    open(cpp, "w").write(src.replace('#include "../../include/acvae_hip.h"', f'#include "{ROOT}/include/acvae_hip.h"'))
    obj = os.path.join(LAB, f"conv_wino_{name}.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, *globals().get("VARIANT_FLAGS", {}).get(name, []), "-c", cpp, "-o", obj])
    objs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".o") and f != "conv_wino.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                           os.path.join(LAB, f"libacvae_{name}.so"), obj, *objs])
    os.remove(obj)


def time_one():
    import math
    import torch
    sys.path.insert(0, ROOT)
    from acvae_amd import _lib
    if os.environ.get("ACVAE_DEV_LIB"):         # this tool's own variable (set per lab build by the parent process)
        _lib.use_library(os.environ["ACVAE_DEV_LIB"])
    S = _lib.current_stream
    out = []
    for (H, W, Cin, Cout) in [(1000, 64, 64, 64), (500, 32, 128, 128), (125, 8, 512, 512)]:
        N = 32
        x = torch.randn(N, H, W, Cin, device="cuda")
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(9 * Cin)
        sc = torch.rand(Cin, device="cuda") + 0.5; sh = torch.randn(Cin, device="cuda") * 0.3
        y = torch.empty(N, H, W, Cout, device="cuda")
        wsb = _lib.call("acvae_conv3x3_workspace_bytes", N, H, W, Cin, Cout)
        ws = torch.empty(int(wsb), dtype=torch.uint8, device="cuda")
        gamma = torch.ones(Cout, device="cuda"); beta = torch.zeros(Cout, device="cuda")
        rm = torch.zeros(Cout, device="cuda"); rv = torch.ones(Cout, device="cuda")
        nbt = torch.zeros((), dtype=torch.int64, device="cuda"); bn = torch.empty(4, Cout, device="cuda")
        dyt = torch.randn(N, H, W, Cout, device="cuda"); dx = torch.empty(N, H, W, Cin, device="cuda")
        dw = torch.empty(Cout, Cin, 3, 3, device="cuda")
        fns = {"fwd_act": lambda: _lib.call("acvae_conv3x3_fwd_wino", x, w, sc, sh, y, gamma, beta, rm, rv, nbt, 1, bn, ws, wsb, N, H, W, Cin, Cout, S()),
               "dgrad": lambda: _lib.call("acvae_conv3x3_dgrad_wino", dyt, w, dx, ws, wsb, N, H, W, Cin, Cout, S()),
               "wgrad": lambda: _lib.call("acvae_conv3x3_wgrad_wino", dyt, x, sc, sh, dw, ws, wsb, N, H, W, Cin, Cout, S())}
        if os.environ.get("ACVAE_DEV_LIB", "").endswith("_timeline.so"):
          for warm in (5, 400):
            for _ in range(warm):
                fns["dgrad"]()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fns["dgrad"](); b.record(); torch.cuda.synchronize()
            nwg = N * (((H + 1) // 2 + (64 // W) - 1) // (64 // W)) * (Cin // 64)
            t = dx.reshape(-1)[:nwg * 32].reshape(nwg, 8, 4)[:, 0, :].double().cpu()
            t0 = float(t[:, 0].min())
            st, en, cu = (t[:, 0] - t0) * 0.01, (t[:, 1] - t0) * 0.01, t[:, 2].long()          # microseconds
            cus = sorted(set(cu.tolist()))
            per = {c: sorted((float(st[i]), float(en[i])) for i in range(nwg) if int(cu[i]) == c) for c in cus}
            counts = [len(v) for v in per.values()]
            gaps = [b0 - a1 for v in per.values() for (a0, a1), (b0, b1) in zip(v, v[1:])]
            durs = [e - s_ for v in per.values() for s_, e in v]
            print(f"{Cin}->{Cout}@{W} dgrad: call {a.elapsed_time(b) * 1e3:.0f} us; {nwg} workgroups on {len(cus)} CUs ({min(counts)}..{max(counts)} per CU); "
                  f"workgroup {sum(durs) / len(durs):.1f} us (max {max(durs):.1f}); gap between two on one CU {sum(gaps) / max(1, len(gaps)):.2f} us (max {max(gaps or [0]):.1f}); "
                  f"last end {float(en.max()):.1f} us; main loop {float(t[:, 3].floor().mean()) * 0.01:.1f} us real per workgroup [after {warm} warm-up launches]")
          continue
        if os.environ.get("ACVAE_DEV_LIB", "").rsplit("_", 1)[-1].startswith("full"):
            import ctypes
            nwg = N * (((H + 1) // 2 + (64 // W) - 1) // (64 // W)) * (Cin // 64)
            labbuf = torch.zeros(nwg, 3, 4, device="cuda")
            raw = ctypes.CDLL(os.environ["ACVAE_DEV_LIB"])
            raw.acvae_lab_set.argtypes = [ctypes.c_void_p]; raw.acvae_lab_set(labbuf.data_ptr())
            for kind in ("dgrad", "fwd_act"):
                if kind == "fwd_act" and Cin != Cout:
                    continue
                for _ in range(300):
                    fns[kind]()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(20):
                    fns[kind]()
                b.record(); torch.cuda.synchronize()
                t = labbuf.double().cpu()
                t0 = float(t[:, 0, 0].min())
                st, en, cu = (t[:, 0, 0] - t0) * 0.01, (t[:, 0, 1] - t0) * 0.01, t[:, 0, 2].long()
                l0, l1, cyc = (t[:, 1, 0] - t0) * 0.01, (t[:, 1, 1] - t0) * 0.01, t[:, 1, 2]
                nch = float(t[0, 0, 3])
                cus = sorted(set(cu.tolist()))
                per = {c: sorted((float(st[i]), float(en[i])) for i in range(nwg) if int(cu[i]) == c) for c in cus}
                gaps = [b0 - a1 for v in per.values() for (a0, a1), (b0, b1) in zip(v, v[1:])]
                counts = [len(v) for v in per.values()]
                print(f"{Cin}->{Cout}@{W} {kind}: call {a.elapsed_time(b) * 50:.0f} us (mean of 20); {nwg} workgroups on {len(cus)} CUs ({min(counts)}..{max(counts)} per CU); "
                      f"first start..last end {float(en.max()):.1f} us; workgroup {float((en - st).mean()):.1f} us = prologue {float((l0 - st).mean()):.1f} + main loop {float((l1 - l0).mean()):.1f} "
                      f"+ epilogue {float((en - l1).mean()):.1f} [prologue: addresses {float(((t[:, 2, 0] - t0) * 0.01 - st).mean()):.2f}, first stage loaded + staged {float(((t[:, 2, 1] - t[:, 2, 0]) * 0.01).mean()):.2f}, barrier + reads + transform {float((l0 - (t[:, 2, 1] - t0) * 0.01).mean()):.2f}]; main loop {float(cyc.mean()):.0f} cycles = {float(cyc.mean()) / nch:.0f} per chunk, clock {float((cyc / ((l1 - l0) * 1e3)).mean()):.3f} GHz; "
                      f"gap between two on one CU {sum(gaps) / max(1, len(gaps)):.2f} us (max {max(gaps or [0]):.1f}); start spread of the first round {sorted(st.tolist())[min(len(cus), nwg) - 1]:.1f} us")
                # which workgroup follows which on a CU: distance in dispatch order inside the XCD (block id >> 3)
                from collections import Counter
                byc = {}
                for i in range(nwg):
                    byc.setdefault(int(cu[i]), []).append((float(st[i]), i))
                dist = Counter()
                for v in byc.values():
                    v.sort()
                    for (s0, i0), (s1, i1) in zip(v, v[1:]):
                        dist[(i1 >> 3) - (i0 >> 3)] += 1
                tot = sum(dist.values())
                print("    successor on the same CU, distance in the XCD's dispatch order: " +
                      ", ".join(f"{d}: {c / tot:.0%}" for d, c in dist.most_common(6)))
            continue
        if os.environ.get("ACVAE_DEV_LIB", "").endswith("_clock.so"):
            for _ in range(300):                      # the clock settles after a second or two of back-to-back launches
                fns["dgrad"]()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                fns["dgrad"]()
            b.record(); torch.cuda.synchronize()
            print(f"   call {a.elapsed_time(b) * 50:.0f} us (mean of 20 back to back)", end="; ")
            nwg = N * (((H + 1) // 2 + (64 // W) - 1) // (64 // W)) * (Cin // 64)
            t = dx.reshape(-1)[:nwg * 32].reshape(nwg, 8, 4).double()
            cyc, real = t[:, :, 0].mean(), t[:, :, 1].mean()
            print(f"{Cin}->{Cout}@{W} dgrad: main loop {float(cyc):.0f} shader cycles in {float(real) * 10:.0f} ns -> in-kernel clock "
                  f"{float(cyc / real) * 0.1:.3f} GHz; {float(cyc / t[0, 0, 2]):.0f} cycles per chunk (4096 = MFMA-bound)")
            continue
        if os.environ.get("ACVAE_DEV_LIB", "").endswith("_stamps.so"):
            for k in ("fwd_act", "dgrad"):
                fns[k](); torch.cuda.synchronize()
                buf = (y if k == "fwd_act" else dx).reshape(-1)
                nwg = N * (((H + 1) // 2 + (64 // W) - 1) // (64 // W)) * ((Cin if k == "dgrad" else Cout) // 64)
                t = buf[:nwg * 32].reshape(nwg, 8, 4).double()
                m = t.mean(dim=(0,))
                print(f"{Cin}->{Cout}@{W} {k}: WGs {nwg} chunks {int(t[0,0,3])}; per wave (prologue, main, epilogue) cycles/100MHz-ticks: "
                      + "; ".join(f"w{w}: {m[w,0]:.0f} {m[w,1]:.0f} {m[w,2]:.0f}" for w in (0, 4)) + f"  main/chunk {float(m[:,1].mean()/t[0,0,3]):.1f}")
            continue
        for k, fn in fns.items():
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                fn()
            b.record(); torch.cuda.synchronize()
            out.append(f"{Cin}->{Cout}@{W} {k} {a.elapsed_time(b) / 20 * 1e3:.0f}us")
    print(os.environ.get("ACVAE_DEV_LIB", "product").split("_")[-1], " | ".join(out), flush=True)


if __name__ == "__main__":
    cmd, names = sys.argv[1], sys.argv[2:]
    if cmd == "list":
        print(" ".join(VARIANTS))
    elif cmd == "build":
        for n in names or VARIANTS:
            build(n)
    elif cmd == "time":
        for n in names:
            lib = os.path.join(LAB, f"libacvae_{n}.so")
            subprocess.call([sys.executable, os.path.abspath(__file__), "_one"], env=dict(os.environ, ACVAE_DEV_LIB=lib))
    elif cmd == "_one":
        time_one()
