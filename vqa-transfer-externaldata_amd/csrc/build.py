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


def _sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "vqa_hot.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force, extra=(), obj_dir=None):
    obj = os.path.join(obj_dir or OBJ_DIR, src.replace(".hip", ".o"))
    sp = os.path.join(HERE, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) >= max(os.path.getmtime(sp), _deps_mtime())):
        return obj
    cmd = ["hipcc"] + FLAGS + list(extra) + ["-c", sp, "-o", obj]
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


if __name__ == "__main__":
    # python build.py [--force] [--out PATH --obj-dir DIR -DNAME=VALUE ...]
    argv = sys.argv[1:]
    opt = lambda k: argv[argv.index(k) + 1] if k in argv else None
    build(force="--force" in argv, extra_flags=[a for a in argv if a.startswith("-D")], out=opt("--out"),
          obj_dir=opt("--obj-dir"))
