#!/usr/bin/env python3
"""Build id of libptx_backend.so: the first 16 hex digits of the SHA-256 over every source the device and host code of the
library is compiled from (path-tracer_amd/csrc/*, the two shared host headers, include/ptx_backend.h) and the compiler flags.
The Makefile compiles it in (-DPTX_BUILD_ID, returned by ptx_build_id()); tools/make_traffic.py stamps profiles/traffic.json
with the id of the library whose counters it holds, and bench.py reports counter-derived figures only when the library it
timed carries the same id.   usage: tools/source_hash.py ["<flags>"]"""
import glob
import hashlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_id(flags=""):
    files = sorted(glob.glob(os.path.join(REPO, "path-tracer_amd", "csrc", "*")))
    files += [os.path.join(REPO, "path-tracer_amd", "host", "ptmath.hpp"), os.path.join(REPO, "path-tracer_amd", "host", "rng.hpp"),
              os.path.join(REPO, "path-tracer_amd", "host", "colors.hpp"), os.path.join(REPO, "include", "ptx_backend.h")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    h.update(" ".join(flags.split()).encode())
    return h.hexdigest()[:16]


def makefile_flags():
    """HIPFLAGS as the Makefile defines them (continuation lines joined)."""
    text = open(os.path.join(REPO, "Makefile")).read().replace("\\\n", " ")
    for line in text.splitlines():
        if line.startswith("HIPFLAGS"):
            return line.split("=", 1)[1].strip()
    raise RuntimeError("HIPFLAGS not found in the Makefile")


def expected_build_id():
    """The id `make backend` stamps into the library from the sources as they are now."""
    return build_id(makefile_flags())


if __name__ == "__main__":
    print(build_id(sys.argv[1]) if len(sys.argv) > 1 else expected_build_id())
