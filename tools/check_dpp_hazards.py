#!/usr/bin/env python3
"""Static check of the gfx950 assembly of a .hip file: every DPP instruction (v_*_dpp) must read its DPP operand (src0)
at least two wait states after the last VALU instruction that wrote that register ("VALU writes VGPR -> VALU DPP reads
that VGPR" data hazard of the CDNA ISA). The compiler guarantees this for the DPP instructions it emits itself; the
hand-written v_fmac_f64_dpp of mpcqp_pair.hip sit in inline asm, which it cannot see into, and rely on an s_nop placed
by the source (dpp_ready). usage: check_dpp_hazards.py file.hip [more.hip ...]   (exit status 1 on a violation)"""
import os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_asm(src: str) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "qpmpc_amd", "csrc"), src, "-o", out],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return open(out).read()


def regs(op: str):
    """VGPR numbers named by an operand like v12, v[4:5], -v[4:5], |v3|."""
    m = re.search(r"v\[(\d+):(\d+)\]", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r"\bv(\d+)\b", op)
    return {int(m.group(1))} if m else set()


def check(asm: str):
    bad, ndpp, nasm = [], 0, 0
    window = []  # (written VGPRs, wait states this instruction contributes) of the preceding instructions
    func = "?"
    for raw in asm.split("\n"):
        line = raw.split(";")[0].strip()
        if not line or line.startswith(".") and not line.startswith(".LBB"):
            continue
        if line.endswith(":"):
            if not line.startswith(".L"):
                func = line[:-1]
                window = []
            continue  # (a label: the straight-line predecessor is still the worst case the source controls)
        op, _, rest = line.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        if op == "s_nop":
            window.append((set(), int(ops[0]) + 1))
            continue
        if "_dpp" in op and len(ops) >= 2:
            ndpp += 1
            nasm += op.startswith("v_fmac_f64_dpp")
            src0 = regs(ops[1])
            waited = 0
            for written, ws in reversed(window):
                if written & src0:
                    if waited < 2:
                        bad.append((func, line, waited))
                    break
                waited += ws
                if waited >= 2:
                    break
        written = regs(ops[0]) if op.startswith("v_") and ops and not op.startswith(("v_cmp", "v_cmpx")) else set()
        if op.startswith("v_permlane16_swap") or op.startswith("v_permlane32_swap"):
            written |= regs(ops[1])  # swaps write both operands
        window.append((written, 1))
        if len(window) > 8:
            window.pop(0)
    return bad, ndpp, nasm


if __name__ == "__main__":
    files = sys.argv[1:] or [os.path.join(ROOT, "qpmpc_amd", "csrc", "mpcqp_pair.hip")]
    rc = 0
    for f in files:
        bad, ndpp, nasm = check(device_asm(f))
        print(f"{os.path.basename(f)}: {ndpp} DPP instructions ({nasm} v_fmac_f64_dpp), {len(bad)} hazard(s)")
        for func, line, waited in bad[:10]:
            print(f"  {func[:60]}: '{line}' reads its DPP operand {waited} wait state(s) after a VALU write")
        rc |= bool(bad)
    sys.exit(rc)
