#!/usr/bin/env python3
"""Build libgbnf_hip.so for gfx950: gbnf_api.hip + one object per kernel variant in variants.list.

    python build.py [-j N] [--force]

Objects go to csrc/obj/ (git-ignored), the library to the package directory (git-ignored but
shipped to the GPU box).  hipcc cross-compiles without a GPU.
"""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OBJ = os.path.join(HERE, "obj")
LIB = os.path.join(PKG, "libgbnf_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-Wno-inline-asm"]


def read_variants():
    out = []
    with open(os.path.join(HERE, "variants.list")) as f:
        for line in f:
            line = line.split("#")[0].strip()
            if not line:
                continue
            if line.startswith("coop"):    # the latency form of an hx3 line's f16x3 kernel: coop KIND HT OT ACTA ACTB (depth 1)
                toks = line.split()
                kind, ht, ot, acta, actb = (int(v) for v in toks[1:6])
                for form in (int(ch) for ch in (toks[6] if len(toks) > 6 else "123")):       # 1 = 16-sample tiles / 4 waves, 2 = 32 / 4, 3 = 32 / 8
                    out.append(("coop", kind, ht, ot, form, acta, actb))
                continue
            if line.startswith("hx3"):     # both split precisions: 0 = f16x3, 1 = bf16x6 (its repair pass / safe mode)
                toks = line.split()[1:]
                eval_only = toks[-1] == "eval"          # no training sweeps for this geometry (hx3t / hx3b): the trainer keeps the per-step kernels
                vals = [int(v) for v in (toks[:-1] if eval_only else toks)]
                kind, ht, ot, acta, actb = vals[:5]
                depth = vals[5] if len(vals) > 5 else 1          # coupling_network_depth: 0, 1 (default) or 2
                for nt in (1, 2):
                    for prec in (0, 1):
                        out.append(("hx3", kind, ht, ot, nt, acta, actb, prec, depth))
                    if not eval_only:            # the training path's forward sweep (f16x3, trace + operand saves): "hx3t", every net
                        out.append(("hx3t", kind, ht, ot, nt, acta, actb, 0, depth))     # and depth; 32-sample waves: one per SIMD
                        if nt == 1:
                            out.append(("hx3b", kind, ht, ot, acta, actb, depth))        # ... and its backward sweep
                continue
            toks = line.split()
            nts = (1, 2)
            if toks[-1] == "nt1":          # 16-sample waves only (the 32-sample form of the widest geometries spills)
                nts, toks = (1,), toks[:-1]
            kind, ht, ksl, ks1, ot, lmid, acta, actb = (int(v) for v in toks)
            for nt in nts:
                out.append((kind, ht, ksl, ks1, ot, nt, lmid, acta, actb))
    return sorted(set(out), key=str)


def newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


LLVM_BIN = "/opt/rocm/lib/llvm/bin"
VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def scratch_bytes(obj):
    """Private-segment (scratch) bytes per lane of the gfx950 kernel inside a host object, or None if it cannot be read."""
    import tempfile
    with tempfile.TemporaryDirectory() as t:
        fb, co = os.path.join(t, "fb.bin"), os.path.join(t, "dev.co")
        try:
            run([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, obj])
            run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--type=o", "--input=" + fb,
                 "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--unbundle", "--output=" + co])
            notes = run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", co])
        except RuntimeError:
            return None
    worst = 0
    for line in notes.splitlines():
        if ".private_segment_fixed_size:" in line:
            worst = max(worst, int(line.split(":")[1]))
    return worst


def lint_object(obj):
    """tools/isa_hazard_lint.py on a freshly compiled object: no path of any kernel may touch a v_mfma result before it exists
    (hipcc pads that dependency along fall-through paths only; profiles/r3_mfma_hazard_root_cause.txt)."""
    tools = os.path.join(PKG, "..", "tools")
    if not os.path.exists(os.path.join(tools, "isa_hazard_lint.py")):
        return
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import isa_hazard_lint as lint
    bad = []
    for name, insts in lint.disassemble(obj).items():
        for i, j, ws, need, crossed in lint.check_kernel(insts):
            bad.append(f"  {name[:80]}: [{insts[i].addr:#x}] {insts[i].text}  ->  [{insts[j].addr:#x}] {insts[j].text} "
                       f"after {ws} of {need} wait states{' (across a branch)' if crossed else ''}")
    if bad:
        raise RuntimeError(f"{os.path.basename(obj)}: a v_mfma result is touched too early (MFMA read-after-write hazard, "
                           "see tools/isa_hazard_lint.py):\n" + "\n".join(bad[:10]))


def compile_and_lint(cmd):
    run(cmd)
    lint_object(cmd[-1])
    return ""


def compile_hx3(cmd):
    """A split-kernel variant: accumulators in VGPRs (the activation + split reads them directly, no v_accvgpr_read per
    value) -- unless that build spills: one wave per SIMD has 512 registers only as 256 VGPRs + 256 AGPRs, and the
    VGPR form cannot use the second half; such variants are rebuilt with the default (AGPR) accumulator placement."""
    run(cmd)
    obj = cmd[-1]
    sb = scratch_bytes(obj)
    if sb:
        plain = [a for a in cmd if a not in VGPR_FORM]
        run(plain)
        sb2 = scratch_bytes(obj)
        if sb2 is not None and sb2 > sb:
            run(cmd)          # the VGPR form was the smaller evil
            sb2 = sb
        if sb2:
            print(f"[gbnf build] note: {os.path.basename(obj)} uses {sb2} bytes of scratch per lane", flush=True)
    lint_object(obj)
    return ""


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--force", action="store_true")
    args = ap.parse_args(argv)
    os.makedirs(OBJ, exist_ok=True)
    hdr = [os.path.join(HERE, "gbnf_flow_kernel.hip.h"), os.path.join(HERE, "gbnf_flow_kernel_hx3.hip.h"),

           os.path.join(PKG, "..", "include", "gbnf.h")]
    jobs = []
    objs = []
    api_o = os.path.join(OBJ, "gbnf_api.o")
    objs.append(api_o)
    api_src = os.path.join(HERE, "gbnf_api.hip")
    hdr.append(os.path.join(HERE, "gbnf_internal.h"))
    if args.force or not newer(api_o, [api_src] + hdr):
        jobs.append([HIPCC] + FLAGS + ["-c", api_src, "-o", api_o])
    train_o, train_src = os.path.join(OBJ, "gbnf_train.o"), os.path.join(HERE, "gbnf_train.hip")
    objs.append(train_o)
    if args.force or not newer(train_o, [train_src, hdr[2], hdr[3]]):
        jobs.append([HIPCC] + FLAGS + ["-c", train_src, "-o", train_o])
    for v in read_variants():
        o = os.path.join(OBJ, "v_" + "_".join(str(a) for a in v) + ".o")
        objs.append(o)
        extra = []
        if v[0] == "hx3b":
            vsrc, vargs = os.path.join(HERE, "variant_bwd.hip"), v[1:]
            extra = list(VGPR_FORM)
        elif v[0] == "coop":
            vsrc, vargs = os.path.join(HERE, "variant_coop.hip"), v[1:]
            extra = list(VGPR_FORM)
        elif v[0] in ("hx3", "hx3t"):
            vsrc, vargs = os.path.join(HERE, "variant_hx3.hip"), v[1:]
            # keep MFMA accumulators in VGPRs: the tanh/split reads them directly (no v_accvgpr_read per value)
            extra = list(VGPR_FORM) + (["-DGBNF_V_TRAIN=1"] if v[0] == "hx3t" else [])
        else:
            vsrc, vargs = os.path.join(HERE, "variant.hip"), v
        if args.force or not newer(o, [vsrc] + hdr + ([os.path.join(HERE, "gbnf_train_bwd.hip.h")] if v[0] == "hx3b" else [])
                                   + ([os.path.join(HERE, "gbnf_flow_kernel_coop.hip.h")] if v[0] == "coop" else [])):
            jobs.append([HIPCC] + FLAGS + extra + ["-DGBNF_V_ARGS=" + ",".join(str(a) for a in vargs), "-c", vsrc, "-o", o])
    img_o, img_src = os.path.join(OBJ, "gbnf_image.o"), os.path.join(HERE, "gbnf_image.hip")
    objs.append(img_o)
    if args.force or not newer(img_o, [img_src, os.path.join(HERE, "gbnf_image_hx3.hip.h"), os.path.join(HERE, "gbnf_image_net.h"), hdr[2], hdr[3]]):
        jobs.append([HIPCC] + FLAGS + ["-c", img_src, "-o", img_o])
    # the fused image coupling-net kernel: its own unit, MFMA accumulators in VGPRs (the relu + split epilogues read them directly)
    net_o, net_src = os.path.join(OBJ, "gbnf_image_net.o"), os.path.join(HERE, "gbnf_image_net.hip")
    objs.append(net_o)
    if args.force or not newer(net_o, [net_src, os.path.join(HERE, "gbnf_image_net.h")]):
        jobs.append([HIPCC] + FLAGS + VGPR_FORM + ["-c", net_src, "-o", net_o])
    comm_o, comm_src = os.path.join(OBJ, "gbnf_comm.o"), os.path.join(HERE, "gbnf_comm.hip")
    objs.append(comm_o)
    if args.force or not newer(comm_o, [comm_src, hdr[2], hdr[3]]):
        jobs.append([HIPCC] + FLAGS + ["-c", comm_src, "-o", comm_o])
    keep = set(objs)
    for fn in os.listdir(OBJ):      # drop objects of variants that left the list
        p = os.path.join(OBJ, fn)
        if fn.endswith(".o") and p not in keep:
            os.remove(p)
    if jobs:
        print(f"[gbnf build] compiling {len(jobs)} object(s) with -j{args.j}", flush=True)
        with cf.ThreadPoolExecutor(max_workers=args.j) as ex:
            for _ in ex.map(lambda c: compile_hx3(c) if ("variant_hx3" in " ".join(c) or "variant_bwd" in " ".join(c) or "variant_coop" in " ".join(c)) else compile_and_lint(c), jobs):
                pass
    if jobs or not os.path.exists(LIB):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
        print(f"[gbnf build] linked {LIB}", flush=True)
    else:
        print("[gbnf build] up to date", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
