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
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value"]


def read_variants():
    out = []
    with open(os.path.join(HERE, "variants.list")) as f:
        for line in f:
            line = line.split("#")[0].strip()
            if not line:
                continue
            if line.startswith("hx3"):
                kind, ht, ot, acta, actb = (int(v) for v in line.split()[1:])
                for nt in (1, 2):
                    out.append(("hx3", kind, ht, ot, nt, acta, actb))
                continue
            kind, ht, ksl, ks1, ot, lmid, acta, actb = (int(v) for v in line.split())
            for nt in (1, 2):
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
        if v[0] == "hx3":
            vsrc, vargs = os.path.join(HERE, "variant_hx3.hip"), v[1:]
            # keep MFMA accumulators in VGPRs: the tanh/split reads them directly (no v_accvgpr_read per value)
            extra = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
        else:
            vsrc, vargs = os.path.join(HERE, "variant.hip"), v
        if args.force or not newer(o, [vsrc] + hdr):
            jobs.append([HIPCC] + FLAGS + extra + ["-DGBNF_V_ARGS=" + ",".join(str(a) for a in vargs), "-c", vsrc, "-o", o])
    img_o, img_src = os.path.join(OBJ, "gbnf_image.o"), os.path.join(HERE, "gbnf_image.hip")
    objs.append(img_o)
    if args.force or not newer(img_o, [img_src, os.path.join(HERE, "gbnf_image_hx3.hip.h"), hdr[2], hdr[3]]):
        jobs.append([HIPCC] + FLAGS + ["-c", img_src, "-o", img_o])
    keep = set(objs)
    for fn in os.listdir(OBJ):      # drop objects of variants that left the list
        p = os.path.join(OBJ, fn)
        if fn.endswith(".o") and p not in keep:
            os.remove(p)
    if jobs:
        print(f"[gbnf build] compiling {len(jobs)} object(s) with -j{args.j}", flush=True)
        with cf.ThreadPoolExecutor(max_workers=args.j) as ex:
            for _ in ex.map(run, jobs):
                pass
    if jobs or not os.path.exists(LIB):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        print(f"[gbnf build] linked {LIB}", flush=True)
    else:
        print("[gbnf build] up to date", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
