"""Compile the HIP engine for gfx950 into gcsa2_amd/lib/libgcsa2_hip.so (in-tree, so that it
travels to the GPU box with the snapshot).  hipcc cross-compiles without a GPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "gcsa2_hip.hip")


def _deps():
    """Everything the library is compiled from: every file under csrc/ that the translation unit can include and
    the public headers under include/ (globbed, so that a new header cannot be forgotten)."""
    import glob
    root = os.path.dirname(HERE)
    found = [SRC]
    found += sorted(glob.glob(os.path.join(HERE, "csrc", "*.hpp")))
    found += sorted(glob.glob(os.path.join(HERE, "csrc", "*.h")))
    found += sorted(glob.glob(os.path.join(root, "include", "*.h")))
    return found


DEPS = _deps()
OUT = os.path.join(HERE, "lib", "libgcsa2_hip.so")

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
         "-Wno-unused-result", "-Wno-unused-function", "-ldl"]


def build(force=False, verbose=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in _deps()):
        return OUT
    cmd = ["hipcc"] + FLAGS + ["-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


BENCH_SRC = os.path.join(HERE, "csrc", "gather_bench.hip")
BENCH_OUT = os.path.join(HERE, "lib", "gather_bench")


def build_gather_bench(force=False):
    """The random-gather ceiling microbenchmark (a standalone HIP program, profiles/r01_gather_bench.md)."""
    os.makedirs(os.path.dirname(BENCH_OUT), exist_ok=True)
    if not force and os.path.exists(BENCH_OUT) and os.path.getmtime(BENCH_OUT) >= os.path.getmtime(BENCH_SRC):
        return BENCH_OUT
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-o", BENCH_OUT, BENCH_SRC])
    return BENCH_OUT


REPRO_SRC = os.path.join(os.path.dirname(HERE), "tools", "hip", "pool_readback_repro.hip")
REPRO_OUT = os.path.join(HERE, "lib", "pool_readback_repro")


def build_pool_repro(force=False):
    """Reproducer for read-backs out of stream-ordered pool memory (profiles/r02_pool_repro.md)."""
    if not force and os.path.exists(REPRO_OUT) and os.path.getmtime(REPRO_OUT) >= os.path.getmtime(REPRO_SRC):
        return REPRO_OUT
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-o", REPRO_OUT, REPRO_SRC])
    return REPRO_OUT


ROOT = os.path.dirname(HERE)
CLI_SRC = os.path.join(ROOT, "tools", "cpp", "query_gcsa.cpp")
CLI_OUT = os.path.join(HERE, "lib", "query_gcsa")


def cli_deps(src):
    """Everything a facade client is compiled from: its source, the C header and every header of the facade
    (include/gcsa2_hip/gcsa.hpp is six #includes of include/gcsa/*.h -- VERDICT r04 #7)."""
    import glob
    inc = os.path.join(ROOT, "include")
    return [src, os.path.join(inc, "gcsa2_hip.h")] + sorted(glob.glob(os.path.join(inc, "gcsa2_hip", "*.hpp")) + glob.glob(os.path.join(inc, "gcsa", "*.h")))


def build_cli(name, force=False):
    """A reference command line tool (tools/cpp/<name>.cpp) as a client of the C++ facade (host compiler only)."""
    build()
    src, out = os.path.join(ROOT, "tools", "cpp", name + ".cpp"), os.path.join(HERE, "lib", name)
    deps = cli_deps(src)
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    libdir = os.path.dirname(OUT)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", out,
                           "-L", libdir, "-lgcsa2_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
                           "-Wl,--allow-shlib-undefined"])
    return out


def build_gcsa_inspect(force=False):
    """tools/cpp/gcsa_inspect.cpp: the member-by-member listing of a .gcsa / .lcp file (host code only, nothing linked)."""
    src, out = os.path.join(ROOT, "tools", "cpp", "gcsa_inspect.cpp"), os.path.join(HERE, "lib", "gcsa_inspect")
    deps = [src, os.path.join(HERE, "csrc", "sdsl_reader.hpp")]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", src, "-o", out])
    return out


def build_query_gcsa(force=False):
    return build_cli("query_gcsa", force)


def build_count_kmers(force=False):
    return build_cli("count_kmers", force)


if __name__ == "__main__":
    build(force=True, verbose=True)
