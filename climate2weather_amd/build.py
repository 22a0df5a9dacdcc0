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
           "sampler.hip", "conv_center.hip"]


STAMP = os.path.join(HERE, "build", "sources.sha256")
DIGEST_MARKER = b"c2w-sources-sha256="  # followed by the 64 hex digits, inside the string c2w_sources_sha256() returns a pointer into


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
    # read out of the FILE, not through dlopen: a handle is never really closed, and a later ctypes.CDLL of the rebuilt library at
    # the same path would return the old mapping (the digest check then fails right after a successful rebuild, or old code runs)
    with open(lib_path, "rb") as f:
        blob = f.read()
    i = blob.find(DIGEST_MARKER)
    if i < 0:
        return None
    d = blob[i + len(DIGEST_MARKER): i + len(DIGEST_MARKER) + 64]
    try:
        d = d.decode("ascii")
    except UnicodeDecodeError:
        return None
    return d if len(d) == 64 and all(c in "0123456789abcdef" for c in d) else None


def _stale() -> bool:
    """The library is reused only if it was built from exactly these sources (content hash, not mtimes: a copied tree has neither)."""
    return embedded_digest() != sources_digest()


class IsaViolation(RuntimeError):
    """An object hipcc produced breaks an invariant of isa_checks.py: nothing is linked."""


def compile_units(sources, csrc, out_dir, hipcc, extra_flags=(), verbose=True):
    """Compile every translation unit (in parallel) keeping the device assembly and the resource-usage remarks of the SAME
    compilation next to the object: [(source name, object path, remarks text, assembly text)]."""
    procs = []
    for src in sources:
        path = os.path.join(csrc, src)
        if not os.path.exists(path):
            continue
        stem = src.replace(".hip", "")
        tmp = os.path.join(out_dir, stem + ".tmp")
        os.makedirs(tmp, exist_ok=True)
        obj = os.path.join(out_dir, stem + ".o")
        # -save-temps=obj: the device .s next to the object is what was assembled into it (not a second compilation);
        # -Rpass-analysis: registers / scratch / LDS per kernel on stderr
        cmd = [hipcc] + FLAGS + list(extra_flags) + ["-I" + INCLUDE, "-I" + csrc, "-c", path, "-o", os.path.join(tmp, stem + ".o"), "-save-temps=obj",
                                                     "-Rpass-analysis=kernel-resource-usage"]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, stem, tmp, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)))
    units = []
    for src, stem, tmp, obj, p in procs:
        _, err = p.communicate()
        err = err.decode(errors="replace")
        if p.returncode != 0:
            sys.stderr.write(err[-8000:])
            raise RuntimeError(f"hipcc failed on {src}")
        asm_path = os.path.join(tmp, f"{stem}-hip-amdgcn-amd-amdhsa-gfx950.s")
        with open(asm_path) as f:
            asm = f.read()
        os.replace(os.path.join(tmp, stem + ".o"), obj)
        for f in os.listdir(tmp):  # keep only the device assembly (the evidence); the preprocessed sources are 2 x 2.4 MB per unit
            if not f.endswith("gfx950.s"):
                os.remove(os.path.join(tmp, f))
        with open(os.path.join(tmp, "resource_usage.txt"), "w") as f:
            f.write(err)
        units.append((src, obj, err, asm))
    return units


def check_units(units, verbose=True) -> None:
    """Run isa_checks on what was just compiled; raise IsaViolation (and leave no library behind) on any finding."""
    from . import isa_checks
    bad, tot = [], dict(kernels=0, counted_vmcnt=0, lds_dma_kernels=0, barriers=0, asm_conversions=0)
    for src, _, remarks, asm in units:
        bad += isa_checks.violations(src, remarks, asm)
        for k, v in isa_checks.summary(remarks, asm).items():
            tot[k] += v
    if verbose:
        print(f"ISA checks over {tot['kernels']} kernels: {tot['counted_vmcnt']} with hand-counted vmcnt waits (no scratch allowed), "
              f"{tot['lds_dma_kernels']} LDS-DMA kernels / {tot['barriers']} barriers (no LDS read outstanding), "
              f"{tot['asm_conversions']} asm conversions (none on a fresh MFMA result): {'OK' if not bad else 'VIOLATED'}", flush=True)
    if bad:
        raise IsaViolation("refusing to link:\n  " + "\n  ".join(bad[:20]))


def build(force: bool = False, verbose: bool = True, sources=None, csrc: str = CSRC, lib: str = LIB, out_dir: str = None,
          extra_flags=()) -> str:
    """``sources`` / ``csrc`` / ``lib`` / ``out_dir`` default to the product library; tests build a seeded translation unit elsewhere."""
    product = sources is None and csrc == CSRC and lib == LIB
    sources = SOURCES if sources is None else sources
    out_dir = out_dir or os.path.join(HERE, "build")
    force = force or os.environ.get("C2W_FORCE_BUILD", "") not in ("", "0") or not product
    if not force and not _stale():
        if verbose:
            print(f"reused {LIB}: sources sha256 {sources_digest()[:16]} match the stamp; library sha256 {file_digest(LIB)[:16]}", flush=True)
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(out_dir, exist_ok=True)
    units = compile_units(sources, csrc, out_dir, hipcc, extra_flags, verbose)
    try:
        check_units(units, verbose)
    except IsaViolation:
        if os.path.exists(lib):  # a stale library next to violating sources must not be picked up either
            os.remove(lib)
        raise
    objs = [u[1] for u in units]
    digest = sources_digest() if product else "0" * 64
    dsrc = os.path.join(out_dir, "sources_digest.cpp")
    with open(dsrc, "w") as f:  # generated, outside csrc/: the digest of the sources is linked into the library built from them
        f.write('static const char c2w_digest[] = "%s%s";\n'
                'extern "C" const char* c2w_sources_sha256(void) { return c2w_digest + %d; }\n' % (DIGEST_MARKER.decode(), digest, len(DIGEST_MARKER)))
    dobj = dsrc.replace(".cpp", ".o")
    subprocess.check_call([hipcc, "-x", "c++", "-O1", "-fPIC", "-c", dsrc, "-o", dobj])
    objs.append(dobj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if product:
        with open(STAMP, "w") as f:  # informational only: the check reads the digest out of the library itself
            f.write(digest + "\n")
    if verbose:
        print(f"compiled {len(objs)} translation units for gfx950 -> {lib}: sources sha256 {digest[:16]}, library sha256 {file_digest(lib)[:16]}",
              flush=True)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
