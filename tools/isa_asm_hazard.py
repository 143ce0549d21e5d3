"""Static scan (no GPU) for the hazard that bit k_gemm_h3<4, 16> in r06: an INLINE-ASM VALU instruction writes a VGPR that an MFMA issued
a few instructions earlier still reads as its A / B operand (or accumulates into).  For compiler-emitted instructions the hazard recogniser
inserts the wait states; for inline asm it does not, and the register allocator is free to hand the asm's output the just-dead operand
registers.  Reports every asm VALU write whose destination overlaps the sources of an MFMA within the previous `WINDOW` instructions.

    python tools/isa_asm_hazard.py [file ...]      (profiles/r06/isa_asm_hazard.txt is this output at HEAD)"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ucnerf_amd", "csrc")
NOSLP = {"field_mlp", "field_mlp_h", "sky", "sky_train", "field_train", "wgrad", "gemm_f32", "gemm_h3"}
WINDOW = 12          # instructions; a 32x32x16 MFMA reads its operands during its first passes (<= 8 passes x 4 cycles)
files = [os.path.splitext(os.path.basename(f))[0] for f in (sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip"))))]
tmp = tempfile.mkdtemp()
base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-w"]
procs = []
for f in files:
    flags = base + (["-fno-slp-vectorize"] if f in NOSLP else []) + (["-mllvm", "-amdgpu-mfma-vgpr-form"] if f == "sky_train" else [])
    procs.append((f, subprocess.Popen(flags + ["-S", "--cuda-device-only", "-o", f"{tmp}/{f}.s", f"{f}.hip"], cwd=CSRC, stderr=subprocess.DEVNULL)))
for f, p in procs:
    p.wait()


def regs(tok):
    """'v[2:5]' -> {2,3,4,5}; 'v7' -> {7}; anything else -> set()"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


total = 0
for f in files:
    txt = open(f"{tmp}/{f}.s").read()
    ms = list(re.finditer(r"^(_Z[A-Za-z0-9_]+):", txt, re.M))
    for k, m in enumerate(ms):
        name = re.sub(r"^_ZN12_GLOBAL__N_1\d+|^_ZL\d+", "", m.group(1))[:60]
        lines = txt[m.start():(ms[k + 1].start() if k + 1 < len(ms) else len(txt))].split("\n")
        ins, in_asm = [], False
        for l in lines:
            s = l.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
            elif s.startswith(";;#ASMEND"):
                in_asm = False
            elif s and not s.startswith((";", ".")) and not s.endswith(":"):
                ins.append((s, in_asm))
        n_asm_valu = hits = 0
        for i, (s, a) in enumerate(ins):
            if not a or not s.startswith("v_") or s.startswith("v_mfma"):
                continue
            n_asm_valu += 1
            dst = regs(s.split()[1].rstrip(","))
            for j in range(max(0, i - WINDOW), i):
                t = ins[j][0]
                if t.startswith("v_mfma"):
                    ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
                    srcs = set().union(*[regs(o) for o in ops[1:3]])         # A and B operands
                    if dst & srcs:
                        hits += 1
                        if hits <= 3:
                            print(f"  {f}: {name}: `{s[:60]}` {i - j} instructions after `{t[:70]}`")
        if n_asm_valu:
            print(f"{f:14s} {name:60s} inline-asm VALU writes {n_asm_valu:5d}   overlapping a recent MFMA's A / B operand: {hits}")
            total += hits
print(f"TOTAL {total}")
