#!/usr/bin/env python3
"""Build oracle/_ref/libevogp_ref.so: the REFERENCE's own device code compiled for the host CPU.

TEST INFRASTRUCTURE ONLY.  Recipe:
  * the device-only line ranges of /root/reference/src/evogp/cuda/{forward,generate,mutation}.cu
    (the host launchers contain <<<>>> and are skipped) are extracted into a TEMPORARY directory
    together with unchanged copies of defs.h / kernel.h — nothing from the reference is written
    into the repository, only the compiled shared object lands in oracle/_ref/ (git-ignored);
  * a small qualifier shim (cuda_runtime.h) defines __global__/__device__/__host__ away and maps
    blockIdx/blockDim/threadIdx onto plain global structs; the real rocThrust taus88 is included
    BEFORE the qualifiers are neutered;
  * oracle/ref_harness.cpp (our code) loops over "threads" and calls the kernels as functions;
  * hipcc -x hip --offload-host-only compiles the lot for x86-64.

The reference does not need its own build system for this (three source files, no external
libraries beyond rocThrust headers shipped in /opt/rocm).  If /root/reference is absent (GPU box),
this script does nothing and the prebuilt .so that travelled with the snapshot is used.
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("EVOGP_REFERENCE", "/root/reference")
CUDA_DIR = os.path.join(REF, "src", "evogp", "cuda")
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "libevogp_ref.so")

# (file, first line, last line, output name, token that must appear on the first line)
RANGES = [
    ("forward.cu", 79, 351, "ref_forward_a.inc", "template <bool multiOutput"),
    ("forward.cu", 373, 400, "ref_forward_b.inc", "SR_BLOCK_SIZE"),
    ("generate.cu", 16, 173, "ref_generate.inc", "template<bool multiOutput"),
    ("mutation.cu", 5, 184, "ref_mutation_a.inc", "__host__ __device__"),
    ("mutation.cu", 224, 309, "ref_mutation_b.inc", "treeGPCrossoverKernel"),
]

SHIM = r"""#pragma once
#include <hip/hip_runtime.h>
#include <thrust/random.h>
#undef __global__
#undef __device__
#undef __host__
#undef __constant__
#undef __shared__
#define __global__
#define __device__
#define __host__
#define __constant__ static
#define __shared__ static
#undef blockIdx
#undef blockDim
#undef threadIdx
struct _evogp_d3 { unsigned x = 0, y = 0, z = 0; };
static _evogp_d3 g_blockIdx, g_blockDim, g_threadIdx;
#define blockIdx g_blockIdx
#define blockDim g_blockDim
#define threadIdx g_threadIdx
"""


def available() -> bool:
    return os.path.isdir(CUDA_DIR)


def build(force: bool = False) -> str | None:
    if not available():
        return OUT if os.path.exists(OUT) else None
    srcs = [os.path.join(CUDA_DIR, f) for f in ("forward.cu", "generate.cu", "mutation.cu", "kernel.h", "defs.h")]
    srcs.append(os.path.join(HERE, "ref_harness.cpp"))
    srcs.append(os.path.abspath(__file__))
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs):
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="evogp_ref_")
    try:
        for fname, lo, hi, out, token in RANGES:
            with open(os.path.join(CUDA_DIR, fname)) as f:
                lines = f.readlines()
            if token not in lines[lo - 1]:
                raise RuntimeError(f"{fname}:{lo} does not look like the expected snapshot (wanted {token!r})")
            with open(os.path.join(tmp, out), "w") as f:
                f.writelines(lines[lo - 1:hi])
        for h in ("defs.h", "kernel.h"):
            shutil.copy(os.path.join(CUDA_DIR, h), os.path.join(tmp, h))
        with open(os.path.join(tmp, "cuda_runtime.h"), "w") as f:
            f.write(SHIM)
        with open(os.path.join(tmp, "device_launch_parameters.h"), "w") as f:
            f.write("#pragma once\n")
        cmd = ["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-host-only", "-O2", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-w", f"-I{tmp}", os.path.join(HERE, "ref_harness.cpp"), "-o", OUT]
        subprocess.run(cmd, check=True, env={**os.environ, "TMPDIR": "/tmp"})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return OUT


if __name__ == "__main__":
    p = build(force="--force" in sys.argv)
    print(p if p else "reference sources not present and no prebuilt oracle/_ref/libevogp_ref.so")
