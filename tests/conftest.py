import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def mini_ontology_paths():
    d = os.path.join(GOLDEN, "mini_ontology")
    return {"attribute_file": os.path.join(d, "attribute.json"), "class_file": os.path.join(d, "class.json"),
            "relation_file": os.path.join(d, "relation.json"), "vocabulary_file": os.path.join(d, "vocab.json"),
            "word_embedding_file": os.path.join(d, "glove.txt")}
