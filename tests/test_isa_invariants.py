"""Static invariants of hand-scheduled kernels that the compiler could break silently (no GPU: hipcc cross-compiles to ISA text).

csrc/wgrad.hip keeps the A image's staging data in v240 .. v255 BY NAME across a whole pipeline stage (inline-asm loads issued two
stages ahead of the ds_write that consumes them); the kernel's allocatable registers end at v239 (`amdgpu_num_vgpr(240)`).  If a
compiler ever hands one of those registers to its own values, an in-flight load lands in them: wrong weight gradients with no fault."""
import os
import re
import subprocess
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ucnerf_amd", "csrc")


def test_wgrad_reserved_staging_registers_are_only_named_by_the_asm_statements():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "wgrad.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                               "-S", "--cuda-device-only", "-w", "wgrad.hip", "-o", out], cwd=CSRC)
        text = open(out).read()
    kernels = list(re.finditer(r"^(_Z\w*k_wgrad_bf16ILi(\d)E\w*):", text, re.M))
    assert len(kernels) == 9
    for m in kernels:
        body = text[m.end():text.index(".Lfunc_end", m.end())].split("\n")
        in_asm, named = False, 0
        for line in body:
            if "#ASMSTART" in line:
                in_asm = True
                continue
            if "#ASMEND" in line:
                in_asm = False
                continue
            code = line.split(";")[0]
            regs = set()
            for r in re.finditer(r"\bv\[(\d+):(\d+)\]", code):
                regs.update(range(int(r.group(1)), int(r.group(2)) + 1))
            for r in re.finditer(r"\bv(\d+)\b", code):
                regs.add(int(r.group(1)))
            if any(r >= 240 for r in regs):
                assert in_asm, (m.group(2), code.strip())
                named += 1
        assert named >= 8, (m.group(2), named)          # the loads and the ds_writes are there
        # the kernel descriptor reserves the whole 256 (the asm statements' clobbers count)
    assert len(re.findall(r"\.amdhsa_next_free_vgpr 256", text)) >= 9


def test_no_inline_asm_valu_write_lands_on_a_live_mfma_operand():
    """r06: hipcc pads no hazards around an `asm` statement, and its register allocator may hand an asm VALU instruction the registers
    that the MFMA issued just before still reads as its A / B operand.  It happened (k_gemm_h3<4, 16>: the inline-asm operand split wrote
    into the A operand of the preceding MFMA; one output tile lost its lo x hi term on 5 % of the rows).  The split-f16 kernels now use
    compiler-visible instructions; this scan (tools/isa_asm_hazard.py) keeps it that way for the three files that split operands:
    no inline-asm VALU write overlaps the A / B sources of an MFMA within the 12 instructions before it."""
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(repo, "tools", "isa_asm_hazard.py"), "gemm_h3", "field_mlp_h", "sky"],
                       capture_output=True, text=True, timeout=1500, cwd=repo)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.strip().endswith("TOTAL 0"), p.stdout[-3000:]
