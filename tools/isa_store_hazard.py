#!/usr/bin/env python3
"""Static check of the gfx950 16-byte-store hazard on the SHIPPED device code ("root-cause R5.13 or fence it").

The hazard (hp_kernels.hpp: store_fence).  On gfx950 a `buffer_store_dwordx4` reads its VGPR operands -- the four data registers and
the per-lane offset -- over several cycles AFTER it has issued; an instruction that overwrites one of them in the next slots changes
what (or where) lanes 12-15 of every 16-lane row store.  The compiler's hazard recogniser does not cover stores whose soffset is an
SGPR, as ours is, so the kernels keep every operand alive up to an `s_nop 3` behind the store (an empty asm with the operands as
inputs).  That construction binds VALUES, not physical registers: nothing in it stops the register allocator from splitting a live
range between the store and the fence -- spill code, a copy -- and re-using a data register inside the window.  Whether it did is a
property of the generated code, so this tool reads the generated code:

    hipcc --offload-arch=gfx950 ... --offload-device-only -S -o engine.s hp_engine.hip      (the Makefile's flags)
    python tools/isa_store_hazard.py engine.s [--min-wait 4]
    python tools/isa_store_hazard.py hipims-ocl_amd/lib/libhipims_mi.so          (the built library: its code object, disassembled)

For every 16-byte buffer store of every kernel it walks forward, counting wait states (one per instruction, k + 1 for `s_nop k`),
until an instruction WRITES one of the store's data / offset VGPRs, and reports the smallest distance per kernel.  Exit status 1 if
any store is overwritten in fewer than --min-wait wait states.  Conservative: a load whose destination overlaps counts at its issue
slot (its data arrive far later); the walk follows the fall-through path (across labels and untaken conditional branches) and ends
at an unconditional branch (a taken branch is itself several cycles).
tests/test_resource_usage.py runs it on the build's own assembly."""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
NO_VGPR_DST = ("buffer_store", "global_store", "scratch_store", "flat_store", "ds_write", "ds_store", "s_", "v_cmp", "v_cmpx", "buffer_wbl2",
               "buffer_inv", "buffer_atomic", "global_atomic", "v_readlane", "v_readfirstlane", "v_nop", "exp", "ds_nop", "ds_bpermute_fi")


def regs_of(tok):
    m = REG.search(tok)
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def written(op, operands):
    """VGPRs an instruction writes (first operand; v_swap writes both; returning atomics are not used by the engine)."""
    if op.startswith(NO_VGPR_DST) and not op.startswith("s_nop"):
        return set()
    if not operands:
        return set()
    w = regs_of(operands[0]) if operands[0].lstrip().startswith("v") else set()
    if op.startswith("v_swap") and len(operands) > 1:
        w |= regs_of(operands[1])
    return w


def disassemble(so_path, workdir):
    """Device code of a built library as text: the gfx950 code object is taken out of the fat binary (llvm-objdump --offloading
    writes it next to its input, hence the copy) and disassembled (seconds; no recompilation)."""
    import glob
    import os
    import shutil
    import subprocess
    tools = "/opt/rocm/lib/llvm/bin"
    local = os.path.join(workdir, os.path.basename(so_path))
    shutil.copy(so_path, local)
    subprocess.run([os.path.join(tools, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=workdir)
    co = [f for f in glob.glob(local + ".*") if "gfx950" in f]
    assert len(co) == 1, co
    out = os.path.join(workdir, "device.dis")
    with open(out, "w") as f:
        subprocess.run([os.path.join(tools, "llvm-objdump"), "-d", "--no-show-raw-insn", co[0]], check=True, stdout=f)
    return out


def scan(path, min_wait):
    """`path`: compiler assembly (-S) or llvm-objdump's disassembly of the code object."""
    kernel, lines = None, []
    kernels = {}
    for raw in open(path):
        line = raw.split(";")[0].split("//")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^(_Z[\w$.]+):", line) or re.match(r"^[0-9a-f]+ <(_Z[\w$.]+)>:", line)
        if m:
            kernel = m.group(1)
            kernels[kernel] = []
            continue
        if kernel is None or line.lstrip().startswith("."):
            continue
        if re.match(r"^[.\w$]+:", line.strip()):                 # a basic-block label
            kernels[kernel].append(("LABEL", []))
            continue
        parts = line.strip().split(None, 1)
        op = parts[0]
        operands = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
        kernels[kernel].append((op, operands))
        if op == "s_endpgm":
            kernel = None
    report, bad = {}, 0
    for k, ins in kernels.items():
        closest, stores = None, 0
        for i, (op, operands) in enumerate(ins):
            if not op.startswith("buffer_store_dwordx4"):
                continue
            stores += 1
            guard = regs_of(operands[0]) | (regs_of(operands[1]) if len(operands) > 1 and operands[1].startswith("v") else set())
            wait = 0
            for op2, operands2 in ins[i + 1:]:
                if op2.startswith(("s_branch", "s_endpgm", "s_setpc", "s_swappc")):
                    break
                if op2 == "LABEL":                                   # falling through into the next block is one of the paths
                    continue
                if written(op2, operands2) & guard:
                    if closest is None or wait < closest[0]:
                        closest = (wait, i, op2 + " " + ", ".join(operands2))
                    break
                wait += (int(operands2[0], 0) + 1) if op2 == "s_nop" and operands2 else 1
                if wait >= 64:
                    break
        if stores:
            report[k] = (stores, closest)
            if closest is not None and closest[0] < min_wait:
                bad += 1
    return report, bad


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    min_wait = int(sys.argv[sys.argv.index("--min-wait") + 1]) if "--min-wait" in sys.argv else 4
    if "--min-wait" in sys.argv:
        args = [a for a in args if a != str(min_wait)]
    path = args[0]
    if path.endswith(".so"):
        import tempfile
        tmp = tempfile.mkdtemp(prefix="hp_isa_")
        path = disassemble(path, tmp)
    report, bad = scan(path, min_wait)
    import subprocess
    names = subprocess.run(["c++filt"] + list(report), capture_output=True, text=True).stdout.splitlines()
    worst = None
    for (k, (stores, closest)), n in zip(report.items(), names):
        n = re.sub(r"^void ", "", re.sub(r"\(.*", "", n))
        if closest is not None and (worst is None or closest[0] < worst):
            worst = closest[0]
        flag = "  <-- inside the window" if closest is not None and closest[0] < min_wait else ""
        if "--quiet" not in sys.argv or flag:
            print(f"{n:72s} {stores:3d} 16-byte stores, first overwrite of an operand after "
                  f"{'no overwrite in the block' if closest is None else str(closest[0]) + ' wait states (' + closest[2] + ')'}{flag}")
    print(f"{len(report)} kernels with 16-byte buffer stores; smallest store-to-overwrite distance {worst} wait states; "
          f"{bad} kernel(s) below {min_wait}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
