"""Build libfacl_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libfacl_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC"]
# per-source extra flags.  sa_bwd2.hip: the SLP vectoriser packs the layer-1 FMAs into v_pk_fma_f32, which is several
# times slower than scalar FMAs beside MFMAs on gfx950 (MI355X guide, "packed f32 VALU ... an anti-lever beside MFMAs")
EXTRA_FLAGS = {"sa_bwd2.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + \
        glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(out_path, csrc_dir, verbose=False):
    """Build an experimental library from another source directory (A/B timing in one process)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = sorted(glob.glob(os.path.join(csrc_dir, "*.hip")))
    objdir = out_path + ".obj"
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for src in srcs:                                  # per file, so that EXTRA_FLAGS apply exactly as in build()
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError(r.stdout.decode(errors="replace"))
        objs.append(obj)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out_path] + objs, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(r.stdout.decode(errors="replace"))
    return out_path


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + \
            glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h"))
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) > max(os.path.getmtime(p) for p in [src] + hdrs)):
            continue
        cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode(errors='replace')}")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
