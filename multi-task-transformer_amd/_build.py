"""Build libmtt_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libmtt_hip.so")
SOURCES = ["gemm.hip", "attn.hip", "attn_fast.hip", "attn_bwd.hip", "rowops.hip", "invpt_ops.hip", "optim.hip", "loss.hip", "upconv.hip", "swin_ops.hip", "iou3d.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _stamp(src, deps):
    """what an object was compiled from: sha256 over the source, the two shared headers and the flags (NOT mtimes: a snapshot copied to
    another box keeps its contents, not its timestamps)"""
    import hashlib
    h = hashlib.sha256(" ".join(f for f in FLAGS if not f.startswith("-I")).encode())
    for f in [src] + deps:
        h.update(open(f, "rb").read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Compile what is stale and link.  An object is up to date when the stamp next to it (<name>.o.sha) equals the hash of its source +
    headers + flags, so shipped binaries are reused exactly when they correspond to the shipped sources; `force=True` (or MTT_FORCE_BUILD=1)
    recompiles everything."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    force = force or os.environ.get("MTT_FORCE_BUILD") == "1"
    deps = [os.path.join(CSRC, "mtt_device.h"), os.path.join(ROOT, "include", "mtt_hip.h")]
    objs, jobs, stamps, wants = [], [], {}, {}
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        want = wants[o] = _stamp(s, deps)
        try:
            have = open(o + ".sha").read().strip()
        except OSError:
            have = None
        if force or not os.path.exists(o) or have != want:
            jobs.append([hipcc] + FLAGS + ["-c", s, "-o", o])
            stamps[o] = want

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    if jobs:
        for o in stamps:                  # an object being rebuilt has no valid stamp until the LINK that contains it succeeded
            try:
                os.remove(o + ".sha")
            except OSError:
                pass
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(run, jobs))
    # the library's own stamp = hash of the stamps of the objects it was linked from: a link that failed or was interrupted after the
    # compile step leaves it stale and the next build() relinks instead of returning the old library
    import hashlib
    lib_want = hashlib.sha256("".join(wants[o] for o in objs).encode()).hexdigest()
    try:
        lib_have = open(LIB + ".sha").read().strip()
    except OSError:
        lib_have = None
    if jobs or not os.path.exists(LIB) or lib_have != lib_want:
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
        for o, st in stamps.items():      # stamps only after the successful link
            with open(o + ".sha", "w") as f:
                f.write(st + "\n")
        with open(LIB + ".sha", "w") as f:
            f.write(lib_want + "\n")
    return LIB


def build_variant(name, defines, source="gemm.hip"):
    """A/B builds for tools/*_bench.py --lib: the library with ONE source recompiled under extra -D flags (e.g. -DMTT_RING_S=5), written to
    build/variants/libmtt_<name>.so (build/ is git-ignored and still ships to the GPU box)."""
    build()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    vdir = os.path.join(ROOT, "build", "variants")
    os.makedirs(vdir, exist_ok=True)
    obj = os.path.join(vdir, f"{source.replace('.hip', '')}_{name}.o")
    lib = os.path.join(vdir, f"libmtt_{name}.so")
    subprocess.run([hipcc] + FLAGS + list(defines) + ["-c", os.path.join(CSRC, source), "-o", obj], check=True)
    objs = [obj if s == source else os.path.join(CSRC, s.replace(".hip", ".o")) for s in SOURCES]
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib], check=True)
    return lib


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":          # python _build.py --variant NAME [--source attn_fast.hip] -DFOO=1 ...
        rest, src = sys.argv[3:], "gemm.hip"
        if "--source" in rest:
            i = rest.index("--source")
            src = rest[i + 1]
            del rest[i:i + 2]
        print(build_variant(sys.argv[2], rest, source=src))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
