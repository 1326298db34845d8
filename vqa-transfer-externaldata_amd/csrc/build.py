"""Builds libvqahot.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build().

Each .hip file is compiled to an object (only when stale) and linked into
<package>/libvqahot.so.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "libvqahot.so")
OBJ_DIR = os.path.join(HERE, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# gru_ws.hip: its matrix streams are 32 x 4 slots unrolled by pragma with a tail micro-step in every slot; before constant
# folding that body is larger than the default pragma-unroll budget, and a loop left rolled puts the register-resident
# weights in scratch memory
PER_FILE_FLAGS = {"gru_ws.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"]}


def _sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(ROOT, "include", "vqa_hot.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force, extra=(), obj_dir=None):
    obj = os.path.join(obj_dir or OBJ_DIR, src.replace(".hip", ".o"))
    sp = os.path.join(HERE, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(sp), _deps_mtime())):
        return obj
    cmd = ["hipcc"] + FLAGS + PER_FILE_FLAGS.get(src, []) + list(extra) + ["-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True, extra_flags=(), out=None, obj_dir=None):
    """extra_flags / out / obj_dir: a second build of the library beside the shipped one (same-box A/B of compile-time
    choices through VQA_HOT_LIB, tools/bench_with_lib.py); the defaults build <package>/libvqahot.so."""
    OUT = out or globals()["OUT"]
    obj_dir = obj_dir or OBJ_DIR
    os.makedirs(obj_dir, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, extra_flags, obj_dir), srcs))
    if (force or not os.path.exists(OUT)
            or os.path.getmtime(OUT) < max(os.path.getmtime(o) for o in objs)):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", OUT)
    return OUT


SAN_FLAGS = ["--offload-host-only", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
             "-fno-sanitize-recover=undefined"]


def build_sanitized(force=False, verbose=True):
    """AddressSanitizer + UndefinedBehaviorSanitizer build of the HOST side of the library (argument validation,
    workspace layouts, tensor lookups, probe bookkeeping, launch plumbing): `--offload-host-only`, so no device code is
    compiled and nothing here runs on a GPU (GPU ASan is not available on the pool).  Output:
    csrc/build_asan/libvqahot_asan.so; tests/test_sanitizers.py runs the host-only ABI exercise against it in a child
    process with the ASan runtime preloaded (asan_runtime())."""
    obj_dir = os.path.join(HERE, "build_asan")
    out = os.path.join(obj_dir, "libvqahot_asan.so")
    os.makedirs(obj_dir, exist_ok=True)
    flags = [f for f in FLAGS if f != "-O3"] + SAN_FLAGS
    srcs = _sources()

    def one(src):
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        sp = os.path.join(HERE, src)
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) >= max(os.path.getmtime(sp), _deps_mtime(), os.path.getmtime(__file__))):
            return obj
        r = subprocess.run(["hipcc"] + flags + ["-c", sp, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc (sanitized) failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(one, srcs))
    if force or not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(o) for o in objs):
        # a host-only object still refers to its translation unit's device image (`__hip_fatbin_<hash>`): give every one
        # an EMPTY offload bundle (magic + zero entries) -- the HIP runtime loads images lazily, and no kernel is ever
        # launched from this build
        nm = subprocess.run(["nm", "-u"] + objs, capture_output=True, text=True).stdout
        syms = sorted({ln.split()[-1] for ln in nm.splitlines() if "__hip_fatbin_" in ln})
        stub = os.path.join(obj_dir, "fatbin_stubs.c")
        with open(stub, "w") as f:
            f.write("/* generated by build.py: empty device images for the host-only sanitizer build */\n")
            for sname in syms:
                f.write("const struct { char magic[24]; unsigned long long n; } %s __attribute__((aligned(4096))) = "
                        "{\"__CLANG_OFFLOAD_BUNDLE__\", 0};\n" % sname)
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined",
                            "-shared-libsan", "-o", out, "-x", "c", stub, "-x", "none"] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link (sanitized) failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", out)
    return out


def asan_runtime():
    """path of the shared ASan runtime to LD_PRELOAD into the python that loads the sanitized library"""
    r = subprocess.run(["hipcc", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    return r.stdout.strip()


if __name__ == "__main__":
    # python build.py [--force] [--out PATH --obj-dir DIR -DNAME=VALUE ...]
    argv = sys.argv[1:]
    if "--sanitize" in argv:
        build_sanitized(force="--force" in argv)
        sys.exit(0)
    opt = lambda k: argv[argv.index(k) + 1] if k in argv else None
    build(force="--force" in argv, extra_flags=[a for a in argv if a.startswith("-D")], out=opt("--out"),
          obj_dir=opt("--obj-dir"))
