#!/usr/bin/env python3
"""Build-time check of the 4-wave GEMM kernels' register discipline (runs without a GPU: hipcc -S of gemm.hip, ~30 s).

The 256 accumulators of gemm_bt_quad_kernel / gemm_bt_quad3_kernel (gemm.hip) and gemm_mxfp8_quad_kernel (gemm_fp8.hip) live in
the fixed AGPRs a0..a255 that ONLY the literal-register inline asm of csrc/gemm_quad_gen.h / gemm_fp8_quad_gen.h touches; the compiler is not told about them (no operand, no clobber), so the
kernels are correct only while the compiler itself places nothing there.  This script compiles gemm.hip and gemm_fp8.hip to gfx950 assembly
and asserts, for every instantiation:

  * exactly 256 AGPRs are allocated (next_free_vgpr - accum_offset == 256): the compiler allocated none beyond them;
  * in program order, the compiler touches an AGPR only while it holds no accumulator: between a tile's read-back of that
    register and the next tile's first MFMA on it (the allocator does park values there at the end of a register-hungry
    epilogue -- legal by the clobber lists, and harmless exactly then); never between an MFMA and the read-back;
  * exactly 256 generated v_accvgpr_read_b32 per kernel (one read-back of the tile);
  * no scratch traffic between the first and the last MFMA (the K-loop).

Usage: python tools/check_quad_asm.py [--keep out.s]      exit code 0 = all kernels clean."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

def kernels(asm: str):
    for chunk in re.split(r"\n\s*\.globl\s+", asm)[1:]:
        name = chunk.split("\n", 1)[0].strip().split(";")[0].strip()
        if "gemm_bt_quad" in name or "gemm_mxfp8_quad" in name:
            yield name, chunk


def agprs_of(tok: str):
    """a5 -> [5]; a[4:7] -> [4, 5, 6, 7]"""
    m = re.fullmatch(r"a\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"a(\d+)", tok)
    return [int(m.group(1))] if m else []


def check(asm: str):
    """Walks every 4-wave kernel in program order with one state per AGPR:
         acc       written by a generated MFMA statement, not read back yet: ONLY the generated statements may touch it
         consumed  read back by the generated v_accvgpr_read (the epilogue owns the value now): the compiler may park a value here
         parked    holds a compiler value (v_accvgpr_write / a load outside the generated statements): the compiler may read it;
                   the next tile's first MFMA (source C = 0) overwrites it -- the compiler knows (clobber list)
       (the program order of the assembly is the order of execution inside a tile; the tile loop's back edge leads from the
       epilogue to the first K-step, whose MFMAs start every chain from the constant 0)."""
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
            errors.append(f"{name}: {agprs} AGPRs allocated (the compiler placed values of its own beyond the accumulators)")
        code = body.split(".amdhsa_", 1)[0] if ".amdhsa_" in body else body
        lines = code.splitlines()
        # execution order inside a tile: the compiler rotates the tile loop (the tail of the epilogue is laid out IN FRONT of the
        # K-loop), so walk the text cyclically from the statement that starts the first chain (a[0:3] from the constant 0)
        start = next((k for k, l in enumerate(lines) if re.match(r"\s*v_mfma\S*\s+a\[0:3\],[^,]+,[^,]+,\s*0\b", l)), None)
        if start is not None:
            while start > 0 and not lines[start - 1].strip().startswith(";;#ASMSTART"):
                start -= 1
            lines = lines[start - 1:] + lines[:start - 1]
        mfma_at = [k for k, l in enumerate(lines) if l.strip().startswith("v_mfma")]
        state = ["consumed"] * 256
        in_asm, reads = False, 0
        for k, line in enumerate(lines):
            ins = line.strip()
            if ins.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if ins.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not ins or ins.startswith((";", ".", "//")) or ins.endswith(":"):
                continue
            ins = ins.split(";", 1)[0].strip()
            if mfma_at and mfma_at[0] < k < mfma_at[-1] and re.match(r"scratch_(load|store)", ins):
                errors.append(f"{name}: scratch traffic inside the K-loop: `{ins}`")
            toks = re.findall(r"\ba\[\d+:\d+\]|\ba\d+\b", ins)
            if not toks:
                continue
            op = ins.split()[0]
            if op.startswith("v_mfma"):
                if not in_asm:
                    errors.append(f"{name}: an MFMA on AGPRs outside the generated statements: `{ins}`")
                dst = agprs_of(toks[0])
                chained = len(toks) > 1                     # source C is the accumulator itself (not the constant 0)
                for a in dst:
                    if chained and state[a] != "acc":
                        errors.append(f"{name}: a{a} accumulates onto a value that is not an accumulator ({state[a]}): `{ins}`")
                    state[a] = "acc"
                continue
            if op == "v_accvgpr_read_b32":
                a = agprs_of(toks[0])[0]
                if in_asm:
                    reads += 1
                    if state[a] != "acc":
                        errors.append(f"{name}: read-back of a{a}, which holds no accumulator ({state[a]})")
                    state[a] = "consumed"
                elif state[a] != "parked":
                    errors.append(f"{name}: the compiler reads a{a} ({state[a]}): `{ins}`")
                continue
            if in_asm:
                errors.append(f"{name}: unexpected AGPR use in a generated statement: `{ins}`")
                continue
            # the compiler's own use of an AGPR: a destination (v_accvgpr_write, a load) parks a value, anything else reads one
            is_dst = op in ("v_accvgpr_write_b32", "v_accvgpr_mov_b32") or re.match(r"(global|buffer|scratch|flat)_load|ds_read", op)
            for a in agprs_of(toks[0]) if is_dst else [x for t in toks for x in agprs_of(t)]:
                if is_dst:
                    if state[a] == "acc":
                        errors.append(f"{name}: the compiler overwrites the live accumulator a{a}: `{ins}`")
                    state[a] = "parked"
                elif state[a] != "parked":
                    errors.append(f"{name}: the compiler reads a{a} ({state[a]}): `{ins}`")
        if reads != 256:
            errors.append(f"{name}: {reads} generated v_accvgpr_read_b32 (expected 256: one read-back per tile)")
        if re.search(r"\bscratch_(load|store)", code) and not mfma_at:
            errors.append(f"{name}: scratch traffic and no MFMA found")
    return n, errors


def check_duo(asm: str):
    """The two-accumulator-set kernel (gemm_duo.hip): its read-backs are interleaved with the OTHER set's MFMAs over a tile loop
    with two unit bodies, so there is no linear "tile" to walk.  The rule is stricter instead: the compiler touches NO AGPR at
    all (it has ~60 VGPRs to spare and no reason to), exactly 256 AGPRs are allocated, every generated read-back reads the
    set the surrounding MFMAs do NOT write, and the kernel has no scratch traffic and no full vmcnt drain outside its end."""
    errors, n = [], 0
    for chunk in re.split(r"\n\s*\.globl\s+", asm)[1:]:
        name = chunk.split("\n", 1)[0].strip().split(";")[0].strip()
        if "gemm_bt_duo" not in name:
            continue
        n += 1
        acc = re.search(r"\.amdhsa_accum_offset (\d+)", chunk)
        nxt = re.search(r"\.amdhsa_next_free_vgpr (\d+)", chunk)
        if not acc or not nxt or int(nxt.group(1)) - int(acc.group(1)) != 256:
            errors.append(f"{name}: not exactly 256 AGPRs allocated")
        code = chunk.split(".amdhsa_", 1)[0]
        in_asm, last_set, drains, reads = False, None, 0, 0
        for line in code.splitlines():
            ins = line.strip()
            if ins.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if ins.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not ins or ins.startswith((";", ".", "//")) or ins.endswith(":"):
                continue
            ins = ins.split(";", 1)[0].strip()
            if re.match(r"scratch_(load|store)", ins):
                errors.append(f"{name}: scratch traffic: `{ins}`")
            if re.search(r"s_waitcnt.*vmcnt\(0\)", ins):
                drains += 1
            toks = re.findall(r"\ba\[\d+:\d+\]|\ba\d+\b", ins)
            if not toks:
                continue
            if not in_asm:
                errors.append(f"{name}: the compiler touches an AGPR: `{ins}`")
                continue
            op = ins.split()[0]
            regs = [x for t in toks for x in agprs_of(t)]
            if op.startswith("v_mfma"):
                last_set = regs[0] // 128
            elif op == "v_accvgpr_read_b32":
                reads += 1
                if last_set is not None and regs[0] // 128 == last_set and drains == 0:
                    errors.append(f"{name}: read-back of a{regs[0]} beside MFMAs of the same set: `{ins}`")
            else:
                errors.append(f"{name}: unexpected AGPR use in a generated statement: `{ins}`")
        if drains != 1:
            errors.append(f"{name}: {drains} full vmcnt drains (expected exactly one, in front of the last unit's epilogue)")
        if reads != 3 * 128:
            errors.append(f"{name}: {reads} generated read-backs (expected 384: two unit bodies + the last unit's epilogue)")
    return n, errors


def main():
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    n, errors = 0, []
    with tempfile.TemporaryDirectory() as td:
        # the two-accumulator-set kernel lives with the experiments (tools/experiments/, linked by `make EXTRA=-DKEDS_EXPERIMENTS` only)
        duo_src = os.path.join(ROOT, "tools", "experiments", "gemm_duo.hip")
        if "--experiments" in sys.argv and os.path.exists(duo_src):
            out = (keep + ".gemm_duo" if keep else os.path.join(td, "gemm_duo.s"))
            res = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--offload-device-only", "-S",
                                  "-I", os.path.join(ROOT, "keds_amd", "csrc"), duo_src, "-o", out], capture_output=True, text=True)
            if res.returncode != 0:
                print(res.stderr[-3000:])
                return 2
            k, e = check_duo(open(out).read())
            if k == 0:
                e.append("gemm_duo.hip: no two-accumulator-set kernel found")
            n += k
            errors += e
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
    for e in errors[:12]:
        print("FAIL", e[:300])
    if len(errors) > 12:
        print(f"... and {len(errors) - 12} more")
    print(f"{n} quad kernel instantiations checked, {len(errors)} problem(s)")
    return 1 if errors or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
