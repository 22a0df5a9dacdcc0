"""Build recipe for libc2w_hip.so (hipcc cross-compiles gfx950 without a GPU).

    python -m climate2weather_amd.build
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libc2w_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17"]
SOURCES = ["conv_igemm.hip", "conv_patch.hip", "conv_patch3.hip", "wgrad.hip", "wgrad_patch.hip", "pointwise.hip", "attention.hip", "attention_mfma.hip",
           "sampler.hip"]


STAMP = os.path.join(HERE, "build", "sources.sha256")


def sources_digest() -> str:
    """sha256 over every file the library is compiled from (csrc/*.hip, csrc/*.h, include/c2w_hip.h) and the compile flags."""
    h = hashlib.sha256()
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if os.path.isfile(os.path.join(CSRC, f))) + [os.path.join(INCLUDE, "c2w_hip.h")]
    for d in deps:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def file_digest(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def embedded_digest(lib_path: str = LIB):
    """The source digest compiled INTO the library (c2w_sources_sha256), or None if the file is missing / predates the symbol.  The
    library carries its own provenance: a .so copied next to the sources without the build directory still identifies itself."""
    if not os.path.exists(lib_path):
        return None
    import ctypes
    try:
        fn = ctypes.CDLL(lib_path).c2w_sources_sha256
    except (OSError, AttributeError):
        return None
    fn.restype = ctypes.c_char_p
    fn.argtypes = []
    return fn().decode()


def _stale() -> bool:
    """The library is reused only if it was built from exactly these sources (content hash, not mtimes: a copied tree has neither)."""
    return embedded_digest() != sources_digest()


def build(force: bool = False, verbose: bool = True) -> str:
    force = force or os.environ.get("C2W_FORCE_BUILD", "") not in ("", "0")
    if not force and not _stale():
        if verbose:
            print(f"reused {LIB}: sources sha256 {sources_digest()[:16]} match the stamp; library sha256 {file_digest(LIB)[:16]}", flush=True)
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-I" + INCLUDE, "-I" + CSRC, "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
    digest = sources_digest()
    dsrc = os.path.join(HERE, "build", "sources_digest.cpp")
    with open(dsrc, "w") as f:  # generated, outside csrc/: the digest of the sources is linked into the library built from them
        f.write('extern "C" const char* c2w_sources_sha256(void) { return "%s"; }\n' % digest)
    dobj = dsrc.replace(".cpp", ".o")
    subprocess.check_call([os.environ.get("CXX", "g++"), "-O1", "-fPIC", "-c", dsrc, "-o", dobj])
    objs.append(dobj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:  # informational only: the check reads the digest out of the library itself
        f.write(digest + "\n")
    if verbose:
        print(f"compiled {len(objs)} translation units for gfx950 -> {LIB}: sources sha256 {sources_digest()[:16]}, library sha256 {file_digest(LIB)[:16]}",
              flush=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
