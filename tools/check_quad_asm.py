#!/usr/bin/env python3
"""Build-time check of the 4-wave GEMM kernels' register discipline (runs without a GPU: hipcc -S of gemm.hip, ~30 s).

The 256 accumulators of gemm_bt_quad_kernel / gemm_bt_quad3_kernel (gemm.hip) and gemm_mxfp8_quad_kernel (gemm_fp8.hip) live in
the fixed AGPRs a0..a255 that ONLY the literal-register inline asm of csrc/gemm_quad_gen.h / gemm_fp8_quad_gen.h touches; the compiler is not told about them (no operand, no clobber), so the
kernels are correct only while the compiler itself places nothing there.  This script compiles gemm.hip and gemm_fp8.hip to gfx950 assembly
and asserts, for every instantiation:

  * exactly 256 AGPRs are allocated (next_free_vgpr - accum_offset == 256): the compiler allocated none of its own;
  * AGPRs appear only in v_mfma_* (as C/D) and as the source of v_accvgpr_read_b32 -- never as a destination of
    v_accvgpr_write / v_accvgpr_mov (a VGPR -> AGPR spill or a renamed accumulator) and never in a memory instruction;
  * exactly 256 v_accvgpr_read_b32 per kernel (one read-back of the tile; a second set would be a compiler copy);
  * no scratch traffic in the instantiations the dispatcher uses (spills go to scratch only, and only in the A/B variants
    listed in ALLOW_SCRATCH).

Usage: python tools/check_quad_asm.py [--keep out.s]      exit code 0 = all kernels clean."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# <EPI, STAMP, PERSIST> instantiations that are A/B or stamped diagnostic variants, never chosen by quad_by_shape: they may spill
ALLOW_SCRATCH = (re.compile(r"gemm_bt_quad_kernelILi9ELi0ELi1E"),)     # persistent residual form (measured, not dispatched)


def kernels(asm: str):
    for chunk in re.split(r"\n\s*\.globl\s+", asm)[1:]:
        name = chunk.split("\n", 1)[0].strip()
        if "gemm_bt_quad" in name or "gemm_mxfp8_quad" in name:
            yield name, chunk


def check(asm: str):
    errors, n = [], 0
    for name, body in kernels(asm):
        n += 1
        acc = re.search(r"\.amdhsa_accum_offset (\d+)", body)
        nxt = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        if not acc or not nxt:
            errors.append(f"{name}: kernel descriptor not found")
            continue
        agprs = int(nxt.group(1)) - int(acc.group(1))
        if agprs != 256:
            errors.append(f"{name}: {agprs} AGPRs allocated (the compiler placed values of its own there)")
        code = body.split(".amdhsa_", 1)[0] if ".amdhsa_" in body else body
        reads = 0
        for line in code.splitlines():
            ins = line.strip()
            if not ins or ins.startswith((";", ".", "//")) or ins.endswith(":"):
                continue
            ins = ins.split(";", 1)[0]
            if not re.search(r"\ba\[?\d+", ins):
                continue
            op = ins.split()[0]
            if op.startswith("v_mfma"):
                continue
            if op == "v_accvgpr_read_b32":
                reads += 1
                continue
            errors.append(f"{name}: AGPR operand outside the generated statements: `{ins.strip()}`")
        if reads != 256:
            errors.append(f"{name}: {reads} v_accvgpr_read_b32 (expected 256: one read-back per tile)")
        if re.search(r"\bscratch_(load|store)", code) and not any(p.search(name) for p in ALLOW_SCRATCH):
            errors.append(f"{name}: scratch traffic (a spill) in a kernel the dispatcher uses")
    return n, errors


def main():
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    n, errors = 0, []
    with tempfile.TemporaryDirectory() as td:
        for fname in ("gemm", "gemm_fp8"):
            out = (keep + "." + fname if keep else os.path.join(td, fname + ".s"))
            src = os.path.join(ROOT, "keds_amd", "csrc", fname + ".hip")
            res = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--offload-device-only", "-S", src, "-o", out],
                                 capture_output=True, text=True)
            if res.returncode != 0:
                print(res.stderr[-3000:])
                return 2
            k, e = check(open(out).read())
            if k == 0:
                e.append(f"{fname}.hip: no 4-wave kernel found")
            n += k
            errors += e
    for e in errors:
        print("FAIL", e)
    print(f"{n} quad kernel instantiations checked, {len(errors)} problem(s)")
    return 1 if errors or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
