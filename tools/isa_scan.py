"""Static scan of every HIP kernel of csrc/ (no GPU): registers, occupancy, scratch, and -- for kernels with MFMAs -- how many
compiler-inserted `s_waitcnt vmcnt(0)` sit BETWEEN the first and the last MFMA (each drains whatever hand-placed LDS-DMA look-ahead is in
flight: the compiler cannot see inline-asm DMA), how many predicated-load branches, how many scratch accesses.

    python tools/isa_scan.py [file ...]            -> stdout (profiles/r05/isa_scan.txt is this output at HEAD)

r05 findings that came from this scan: k_gemm_f32's vmcnt(0) behind predicated loads; k_sky_train_fwd's 190 spilled registers (96 drains
inside its chain); 12 drains in k_train_fwd (bias / per-ray loads inside the chain); 405 spilled registers in the C = 4 table gradient."""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ucnerf_amd", "csrc")
NOSLP = {"field_mlp", "field_mlp_h", "sky", "sky_train", "field_train", "wgrad", "gemm_f32"}
files = [os.path.splitext(os.path.basename(f))[0] for f in (sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip"))))]
tmp = tempfile.mkdtemp()
base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-w"]
procs = []
for f in files:
    flags = base + (["-fno-slp-vectorize"] if f in NOSLP else []) + (["-mllvm", "-amdgpu-mfma-vgpr-form"] if f == "sky_train" else [])   # = build.sh
    procs.append((f, subprocess.Popen(flags + ["-S", "--cuda-device-only", "-o", f"{tmp}/{f}.s", f"{f}.hip", "-Rpass-analysis=kernel-resource-usage"],
                                      cwd=CSRC, stderr=open(f"{tmp}/{f}.rem", "w"))))
for f, p in procs:
    p.wait()
print(f"{'file':14s} {'kernel':58s} {'VGPR':>4s} {'AGPR':>4s} {'occ':>3s} {'scratch B':>9s} {'MFMA':>5s} {'DMA':>4s} {'vmcnt(0) in chain':>17s} {'execz':>5s}")
for f in files:
    rem = open(f"{tmp}/{f}.rem").read()
    res = {}
    for b in rem.split("Function Name: ")[1:]:
        g = lambda k: int((re.search(k + r": (\d+)", b) or [0, 0])[1])
        res[b.split(" ")[0]] = (g("VGPRs"), g("AGPRs"), g(r"Occupancy \[waves/SIMD\]"), g(r"ScratchSize \[bytes/lane\]"))
    txt = open(f"{tmp}/{f}.s").read()
    ms = list(re.finditer(r"^(_Z[A-Za-z0-9_]+):", txt, re.M))
    for k, m in enumerate(ms):
        name = m.group(1)
        lines = txt[m.start():(ms[k + 1].start() if k + 1 < len(ms) else len(txt))].split("\n")
        mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
        inner = sum(1 for i, l in enumerate(lines) if mf and mf[0] < i < mf[-1] and re.search(r"s_waitcnt.*vmcnt\(0\)", l))
        vg, ag, occ, sc = res.get(name, (0, 0, 0, 0))
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+|^_ZL\d+", "", name)[:58]
        print(f"{f:14s} {short:58s} {vg:4d} {ag:4d} {occ:3d} {sc:9d} {len(mf):5d} {sum('global_load_lds' in l for l in lines):4d} "
              f"{(str(inner) if mf else '-'):>17s} {sum('s_cbranch_execz' in l for l in lines):5d}")
