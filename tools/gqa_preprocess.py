#!/usr/bin/env python3
"""GQA question JSON -> per-operator program files (-> program bytecode): the offline data preparation in front of `ProgramDataset`.

Same flags as the reference's src/gqa_preprocess.py:363-398; paths that the reference hard-codes are arguments here.

  python tools/gqa_preprocess.py questions.json out_dir --op-map op_map.json [-l] [-g]
         [-b --attributes gqa_all_attribute.json --classes gqa_all_class.json --vocabulary gqa_vocab.json]

writes out_dir/p_<name>/p_<name>_<operator>[_<length>].json (one question per line) and, with -b, the bytecode of every such file
as out_dir/h5_<name>/<same stem>.h5 - the reference's HDF5 layout (six int32 datasets, gqa_preprocess.py:87-93), written through h5py or,
without it, the HDF5 C library (dfol_vqa_amd/h5lite.py); `--container npz` writes .npz instead.  `ProgramDataset` reads either.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfol_vqa_amd.preprocess import GQAPreprocessor  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('input_file', help='a GQA question JSON file, or a directory of them')
    ap.add_argument('output_path')
    ap.add_argument('--op-map', required=True, help="the operator map (the reference's nsvqa/data/metadata/op_map.json)")
    ap.add_argument('-b', '--h5', action='store_true', help='also write the program bytecode')
    ap.add_argument('-l', '--length_segregation', action='store_true', help='one file per (terminal operator, first-branch length)')
    ap.add_argument('-g', '--discard_global', action='store_true', help="drop 'select scene' questions")
    ap.add_argument('--attributes'), ap.add_argument('--classes'), ap.add_argument('--vocabulary')
    ap.add_argument('--container', choices=('h5', 'npz'), default=None, help='bytecode container (default: h5 when HDF5 is available, else npz)')
    args = ap.parse_args()

    name = os.path.basename(os.path.normpath(args.input_file))
    if os.path.isfile(args.input_file):
        name = os.path.splitext(name)[0]
    out_dir = os.path.join(args.output_path, 'p_' + name)
    os.makedirs(out_dir, exist_ok=True)
    GQAPreprocessor(args.op_map, True).preprocess(args.input_file, os.path.join(out_dir, 'p_' + name + '.json'), True,
                                                  args.length_segregation, discard_global=args.discard_global)
    if args.h5:
        if not (args.attributes and args.classes and args.vocabulary):
            ap.error('-b needs --attributes, --classes and --vocabulary')
        from dfol_vqa_amd.data import ProgramCodec, write_arrays
        if args.container is None:
            from dfol_vqa_amd import h5lite
            try:
                h5lite.import_h5py()
                args.container = 'h5'
            except IOError:
                args.container = 'npz'
        from dfol_vqa_amd.gqa_ops import GQAOntology
        codec = ProgramCodec(GQAOntology(args.attributes, args.classes, args.vocabulary, None))
        code_dir = os.path.join(args.output_path, 'h5_' + name)
        os.makedirs(code_dir, exist_ok=True)
        for f in sorted(os.listdir(out_dir)):
            with open(os.path.join(out_dir, f)) as fh:
                questions = [json.loads(line) for line in fh if line.strip()]
            if questions:
                write_arrays(os.path.join(code_dir, os.path.splitext(f)[0] + '.' + args.container), codec.encode(questions))
                print('%s: %d questions' % (f, len(questions)))


if __name__ == '__main__':
    main()
