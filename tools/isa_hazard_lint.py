#!/usr/bin/env python3
"""Static check of the shipped gfx950 ISA: no instruction may touch a v_mfma's result registers before the result exists,
ON ANY PATH THROUGH THE KERNEL -- taken branches included.

Why (round 3, profiles/r3_mfma_hazard_root_cause.txt).  The hardware does not interlock a VALU / LDS / VMEM read of a
v_mfma result: software has to keep 7 wait states between a v_mfma_f32_16x16x32_{f16,bf16} (4 passes) and the first
use of its vDst (tools/ubench/mfma_raw_latency.hip, measured: stale at <= 6, right at >= 7, whatever the chain length
or the load on the pipe).  hipcc inserts that padding -- but only along the FALL-THROUGH path: where a conditional
branch sits between the v_mfma and its reader and the fall-through side is long, the reader on the TAKEN side can come
1-2 wait states behind the v_mfma.  That was the round-2 "tile corruption" (flow_kernel_hx3, last hidden pass -> drain
with `if (next net or step) issue_net_start()` in between: the last step of a component read a stale hidden tile
whenever the instruction fetch at the branch target was fast), and the two hand-padded places of round 1
(gbnf_flow_kernel.hip.h inverse epilogue, gbnf_train.hip branch drains) are the same thing.

The check: disassemble every kernel, walk its control-flow graph from every v_mfma along ALL paths (conditional
branches: both sides; a branch instruction counts as one wait state, its fetch latency as none) and report any
instruction that names a register of the v_mfma's vDst within the REQUIRED wait states -- except a v_mfma that takes
those registers as srcC (dependent accumulation: the matrix pipe orders that itself, 0 wait states, measured).
VALU / LDS-return overwrites of a v_mfma's srcA / srcB / srcC are NOT hazards on this hardware (0 wrong results at 0
wait states under load: tools/ubench/mfma_srcc_war.hip, mfma_srcab_war.hip, mfma_lds_war.hip) and are not checked.

    python3 tools/isa_hazard_lint.py [-v] [files...]      default: every object under csrc/obj (all kernels)
Exit code 1 if a violation is found.  tests/test_isa_lint.py runs it on the shipped objects (CPU only).
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
HERE = os.path.dirname(os.path.abspath(__file__))
OBJ_DIR = os.path.join(HERE, "..", "gradient-boosted-normalizing-flows_amd", "csrc", "obj")

REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def required_wait_states(mnemonic):
    """Wait states between the issue of a v_mfma and the first touch of its vDst: passes + 3 (measured for the 4-pass
    16x16x32 f16 / bf16 shapes: 7, tools/ubench/mfma_raw_latency.hip; the other shapes by the same rule)."""
    if "16x16x32" in mnemonic or "16x16x16" in mnemonic or "4x4x4" in mnemonic:      # 4 passes (or fewer), XDL
        return 7
    if "16x16x4_f32" in mnemonic or "16x16x4f32" in mnemonic:                         # 8 passes, f32 inputs (not XDL): passes + 2
        return 10
    if "32x32x16" in mnemonic or "32x32x8" in mnemonic or "16x16x8" in mnemonic:      # 8 passes, XDL
        return 11
    return 19                                                   # 16 passes (32x32x2 / 32x32x4 f32) and unknown shapes


def regs(tok):
    """Register numbers named by an operand token ('v5', 'v[4:7]', 'a[0:3]'): set of (file, n)."""
    out = set()
    for m in REG.finditer(tok):
        f = m.group(1)
        if m.group(2) is not None:
            out.add((f, int(m.group(2))))
        else:
            out.update((f, n) for n in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


def device_code(path, tmp):
    """Path of the gfx950 code object inside a host object / shared library (or the file itself if it already is one)."""
    fb = os.path.join(tmp, "fb.bin")
    co = os.path.join(tmp, "dev.co")
    r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, path],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0 or not os.path.exists(fb) or os.path.getsize(fb) == 0:
        return path
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--input=" + fb,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--unbundle", "--output=" + co], check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return co


class Inst:
    __slots__ = ("addr", "mn", "ops", "text", "touch")

    def __init__(self, addr, mn, ops, text):
        self.addr, self.mn, self.ops, self.text = addr, mn, ops, text
        self.touch = set()
        for o in ops:
            self.touch |= regs(o)


def is_mfma(mn):
    return mn.startswith("v_mfma") or mn.startswith("v_smfmac")


def disassemble(path):
    """{kernel name: [Inst]} of every function in the file's gfx950 code."""
    with tempfile.TemporaryDirectory() as tmp:
        co = device_code(path, tmp)
        text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], check=True,
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body, _, comment = line.partition("//")
        body = body.strip()
        am = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        if not body or not am:
            continue
        parts = body.split(None, 1)
        ops = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
        cur.append(Inst(int(am.group(1), 16), parts[0], ops, body))
    return kernels


def wait_states(inst):
    if inst.mn == "s_nop":
        return int(inst.ops[0], 0) + 1
    return 1


def branch_target(inst):
    v = int(inst.ops[0], 0)
    if v >= 0x8000:
        v -= 0x10000
    return inst.addr + 4 + 4 * v


def check_kernel(insts):
    """[(mfma index, touching index, wait states, required, crossed a branch)] over all paths."""
    by_addr = {ins.addr: k for k, ins in enumerate(insts)}
    out = []
    for i, ins in enumerate(insts):
        if not is_mfma(ins.mn):
            continue
        need = required_wait_states(ins.mn)
        dst = regs(ins.ops[0])
        found = {}                       # touching index -> (smallest distance, crossed a branch)
        stack = [(i + 1, 0, False)]
        best = {}                        # index -> smallest ws it was reached with
        while stack:
            j, ws, crossed = stack.pop()
            while j < len(insts) and ws < need:
                if best.get(j, 1 << 30) <= ws:
                    break
                best[j] = ws
                cur = insts[j]
                mn = cur.mn
                if cur.touch & dst:
                    benign = False
                    if is_mfma(mn) and len(cur.ops) > 3:
                        # benign: (a) dependent accumulation -- the overlap is in srcC (and, tied, in vDst), not in srcA / srcB:
                        # the matrix pipe orders that itself; (b) a v_mfma of the same shape that only OVERWRITES the registers
                        # (in-order pipe, equal latency: the results land in issue order)
                        srcab = regs(cur.ops[1]) | regs(cur.ops[2])
                        reads_c = bool(regs(cur.ops[3]) & dst)
                        benign = not (srcab & dst) and (reads_c or mn == ins.mn)
                    if not benign:
                        if found.get(j, (1 << 30, False))[0] > ws:
                            found[j] = (ws, crossed)
                        break            # later touches on this path are ordered behind this one
                    if regs(cur.ops[0]) >= dst:
                        break            # the registers now belong to the dependent v_mfma (it is checked on its own)
                if mn == "s_endpgm" or mn.startswith("s_setpc") or mn.startswith("s_swappc"):
                    break
                if mn == "s_branch":
                    t = by_addr.get(branch_target(cur))
                    if t is None:
                        break
                    j, ws, crossed = t, ws + 1, True
                    continue
                if mn.startswith("s_cbranch"):
                    t = by_addr.get(branch_target(cur))
                    if t is not None:
                        stack.append((t, ws + 1, True))
                ws += wait_states(cur)
                j += 1
        for j, (ws, crossed) in found.items():
            out.append((i, j, ws, need, crossed))
    return out


def is_vmem(mn):
    return mn.startswith(("global_", "buffer_", "flat_", "scratch_"))


def check_counted_waits(insts):
    """[(wait index, dma index, vector-memory operations between, N)]: a stage end of the staged kernels is
    ``s_waitcnt vmcnt(N) ... ; s_barrier`` with N > 0 where the wave has issued N later operations BEHIND its staging DMA
    (``global_load_lds``) that may stay in flight.  vmcnt counts in issue order, so the wait covers the DMA only if at least N
    vector-memory instructions really sit between the two -- a load whose value is never used is dropped by the compiler, and the
    stage would then be read before it has landed (round 3: bwd_kernel_hx3's last pass)."""
    out = []
    for i, ins in enumerate(insts):
        if ins.mn != "s_waitcnt" or i + 1 >= len(insts) or insts[i + 1].mn != "s_barrier":
            continue
        m = re.search(r"vmcnt\((\d+)\)", ins.text)
        if not m or int(m.group(1)) == 0:
            continue
        n, between = int(m.group(1)), 0
        for j in range(i - 1, -1, -1):
            mn = insts[j].mn
            if mn.startswith("global_load_lds"):
                if between < n:
                    out.append((i, j, between, n))
                break
            if mn == "s_waitcnt" and "vmcnt" in insts[j].text:
                break
            if is_vmem(mn):
                between += 1
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--verbose", "-v", action="store_true")
    ap.add_argument("--max-print", type=int, default=40)
    args = ap.parse_args(argv)
    files = args.files or sorted(glob.glob(os.path.join(OBJ_DIR, "*.o")))
    if not files:
        print("isa_hazard_lint: no objects to check (build first)", file=sys.stderr)
        return 2
    bad = n_mfma = n_kernels = 0
    closest = None
    for f in files:
        for name, insts in disassemble(f).items():
            for i, j, between, n in check_counted_waits(insts):
                bad += 1
                print(f"{os.path.basename(f)}: {name[:70]}\n    [{insts[i].addr:#x}] {insts[i].text} leaves the staging DMA at "
                      f"[{insts[j].addr:#x}] in flight: only {between} vector-memory operation(s) behind it")
            k = sum(1 for x in insts if is_mfma(x.mn))
            if not k:
                continue
            n_kernels += 1
            n_mfma += k
            for i, j, ws, need, crossed in check_kernel(insts):
                closest = ws if closest is None else min(closest, ws)
                bad += 1
                if args.verbose or bad <= args.max_print:
                    print(f"{os.path.basename(f)}: {name[:70]}\n    result of [{insts[i].addr:#x}] {insts[i].text}\n"
                          f"    touched {ws} wait state(s) later ({'across a branch' if crossed else 'straight line'}; need {need}) "
                          f"by [{insts[j].addr:#x}] {insts[j].text}")
    print(f"isa_hazard_lint: {len(files)} file(s), {n_kernels} kernel(s) with {n_mfma} v_mfma; {bad} result(s) touched too early"
          + (f" (closest: {closest} wait states)" if closest is not None else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
