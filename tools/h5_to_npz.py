#!/usr/bin/env python3
"""Convert the reference's HDF5 containers (question bytecode `*.h5`, object-feature chunks `<prefix>_<i>.h5`) to `.npz` with the same
dataset names, and back (`--to h5`).  Reads and writes HDF5 through h5py when it is installed, else through the HDF5 C library
(dfol_vqa_amd/h5lite.py).  `ProgramDataset` / `BatchGQABoxFeaturesCollator` read either container, so this is only needed to move
data to a machine that has neither h5py nor libhdf5.

    python tools/h5_to_npz.py in.h5 [out.npz]         python tools/h5_to_npz.py --to h5 in.npz [out.h5]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfol_vqa_amd import data  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst", nargs="?")
    ap.add_argument("--to", choices=("npz", "h5"), default="npz")
    args = ap.parse_args()
    dst = args.dst or os.path.splitext(args.src)[0] + "." + args.to
    src = data._open_arrays(args.src)
    names = list(src.keys()) if hasattr(src, "keys") else list(src.files)
    arrays = {k: np.asarray(src[k][...] if hasattr(src[k], "shape") and not isinstance(src[k], np.ndarray) else src[k]) for k in names}
    data.write_arrays(dst, arrays)
    print("%s -> %s: %s" % (args.src, dst, ", ".join("%s%s" % (k, list(v.shape)) for k, v in arrays.items())))


if __name__ == "__main__":
    main()
