#!/usr/bin/env python
"""Pack the shipped interaction logs (reference data/<DATASET>/period_N.txt, "sessId itemId" per
line, util.py:46) into one compressed int32 archive per dataset so the GPU box -- which only
receives this repo -- can run the Recall@20 parity runs.  Data only; file order is preserved.

    python tools/pack_dataset.py /root/reference/data data
"""
import os
import sys

import numpy as np


def main(src_root, dst_root):
    os.makedirs(dst_root, exist_ok=True)
    for ds in ("DIGINETICA", "YOOCHOOSE"):
        d = os.path.join(src_root, ds)
        out = {}
        n = len([f for f in os.listdir(d) if f.endswith(".txt")])
        for p in range(n):
            a = np.loadtxt(os.path.join(d, "period_%d.txt" % p), dtype=np.int64, ndmin=2)
            assert a.max() < 2 ** 31
            out["sess_%d" % p] = a[:, 0].astype(np.int32)
            out["item_%d" % p] = a[:, 1].astype(np.int32)
        np.savez_compressed(os.path.join(dst_root, ds + ".npz"), **out)
        print(ds, n, "periods ->", os.path.getsize(os.path.join(dst_root, ds + ".npz")), "bytes")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
