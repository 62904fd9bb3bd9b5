"""Small synthetic ontology used by the golden fixtures and the CPU/GPU tests.

The reference reads four metadata files whose paths come from the config YAML
(`attribute_file`, `class_file`, `relation_file`, `vocabulary_file`; reference
`src/gqa_interpreter_experiments.py:83-91`).  The real GQA files are large and
belong to the reference; the fixtures instead use this hand-made miniature with
the same schema, so that goldens stay a few KB and travel to the GPU box.

Everything here is data authored for this repository; nothing is derived from
the reference's metadata.
"""

import json
import os

import numpy as np

ATTRIBUTES = {
    "color": ["red", "blue", "white", "black", "green"],
    "size": ["small", "large"],
    "material": ["wood", "metal", "glass"],
    "pose": ["sitting", "standing"],
}

CLASSES = {
    "animal": ["dog", "cat", "horse"],
    "furniture": ["table", "chair", "couch"],
    "vehicle": ["car", "bus"],
    "person": ["man", "woman", "boy"],
    "thing": ["cup", "tree"],
}

# "riding" is deliberately missing from the vocabulary: the real relation list
# has one entry that the vocabulary lacks (334 relations, 333 columns), and the
# ontology must skip it the same way.
RELATIONS = ["on", "under", "near", "to the left of", "to the right of", "holding", "behind", "riding"]

OPS = ["select", "all_different", "all_same", "and", "relate", "filter", "choose_attr", "choose_rel",
       "compare", "exist", "or", "query_attr", "two_different", "two_same", "verify_attrs", "verify_rel"]

EXTRA_ARGS = ["yes", "no", "true", "false", "name", "type", "entity", "scene", "_"]

EMBEDDING_DIM = 12


def build_vocab(seed=7):
    args = []
    for k, v in ATTRIBUTES.items():
        args.append(k)
        args.extend(v)
    for k, v in CLASSES.items():
        args.append(k)
        args.extend(v)
    args.extend(r for r in RELATIONS if r != "riding")
    args.extend(EXTRA_ARGS)
    seen = []
    for a in args:
        if a not in seen:
            seen.append(a)
    rng = np.random.RandomState(seed)
    perm = rng.permutation(len(seen))
    idx_to_arg = [seen[i] for i in perm]
    arg_to_idx = {a: i + 1 for i, a in enumerate(idx_to_arg)}
    images = ["img%03d" % i for i in range(64)]
    return {
        "op_to_idx": {o: i + 1 for i, o in enumerate(OPS)},
        "idx_to_op": list(OPS),
        "arg_to_idx": arg_to_idx,
        "idx_to_arg": idx_to_arg,
        "img_to_idx": {im: i + 1 for i, im in enumerate(images)},
        "idx_to_img": images,
    }


def write(directory, seed=7):
    """Write the four JSON files plus a random 'GloVe' text file. Returns a dict of paths."""
    os.makedirs(directory, exist_ok=True)
    vocab = build_vocab(seed)
    paths = {
        "attribute_file": os.path.join(directory, "attribute.json"),
        "class_file": os.path.join(directory, "class.json"),
        "relation_file": os.path.join(directory, "relation.json"),
        "vocabulary_file": os.path.join(directory, "vocab.json"),
        "word_embedding_file": os.path.join(directory, "glove.txt"),
    }
    with open(paths["attribute_file"], "w") as f:
        json.dump(ATTRIBUTES, f, indent=0)
    with open(paths["class_file"], "w") as f:
        json.dump(CLASSES, f, indent=0)
    with open(paths["relation_file"], "w") as f:
        json.dump(RELATIONS, f, indent=0)
    with open(paths["vocabulary_file"], "w") as f:
        json.dump(vocab, f, indent=0)
    words = []
    for a in vocab["idx_to_arg"]:
        for w in a.split(" "):
            if w not in words:
                words.append(w)
    rng = np.random.RandomState(seed + 1)
    with open(paths["word_embedding_file"], "w") as f:
        for w in sorted(words):
            vec = rng.uniform(-1, 1, EMBEDDING_DIM)
            f.write(w + " " + " ".join("%.5f" % x for x in vec) + "\n")
    return paths


if __name__ == "__main__":
    import sys
    print(write(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/mini_ontology"))
