"""Child process of tests/test_model_fuzz.py: feeds mutated .kwsm blobs to kws_create and prints one line per blob
("<index> <return code>"); a crash of the library ends this process, which the parent reports with the blob's index."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mutations(blob, seed, n):
    """deterministic list of (description, bytes): truncations, then byte / word mutations weighted towards the header and tables"""
    rng = np.random.default_rng(seed)
    out = []
    for cut in sorted(set(int(x) for x in np.r_[np.arange(0, 64, 4), rng.integers(64, len(blob), 24), len(blob) - 4, len(blob) - 1])):
        out.append(("truncate@%d" % cut, blob[:cut]))
    head = min(len(blob), 4096)
    while len(out) < n:
        b = bytearray(blob)
        kind = int(rng.integers(0, 4))
        pos = int(rng.integers(4, head if rng.random() < 0.7 else len(blob) - 4)) & ~3
        if kind == 0:
            b[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            b[pos:pos + 4] = int(rng.choice([0, 0xffffffff, 0x7fffffff, 0x80000000, 1, 2, 255, 65536])).to_bytes(4, "little")
        elif kind == 2:
            b[pos:pos + 4] = int(rng.integers(0, 2 ** 32)).to_bytes(4, "little")
        else:
            for _ in range(int(rng.integers(2, 9))):
                p2 = int(rng.integers(4, head)) & ~3
                b[p2:p2 + 4] = int(rng.integers(-3, 70)).to_bytes(4, "little", signed=True)
        out.append(("mutate kind %d @%d" % (kind, pos), bytes(b)))
    return out


def main():
    path, seed, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    from __graft_entry__ import load_package
    pkg = load_package()
    blob = open(path, "rb").read()
    for i, (what, b) in enumerate(mutations(blob, seed, n)):
        print("%d begin %s" % (i, what), flush=True)
        try:
            m = pkg.Model(blob=b)
            rc = 0
            m.close()
        except pkg.KwsError as e:
            rc = e.code
        print("%d rc %d" % (i, rc), flush=True)


if __name__ == "__main__":
    main()
