#!/usr/bin/env python3
"""Content hash of cross-scale-mae_amd/csrc (the recipe of bench.py:csrc_hash): the Makefile bakes it into the library
(csmae_source_hash()), bench.py and tools/profile_round.sh compare profiles and the loaded library against it."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cross-scale-mae_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_hash())
