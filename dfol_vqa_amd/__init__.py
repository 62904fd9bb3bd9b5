"""MI355X-native hot path of the ∇-FOL (DFOL-VQA) program interpreter.

Host-side mirror of the reference's operator/program API (reference
`src/nsvqa/nn/interpreter/*`), calling hand-written gfx950 HIP kernels through
the C-ABI library declared in `include/dfol_vqa.h`.
"""

__version__ = "0.1.0"

from .fol_types import BatchVariableSet, BatchWorld, Quantifier, QuestionType, TokenType  # noqa: F401
from .logic_ops import BatchBayesianLogicCell, FilterBatch, RelateBatch, SelectBatch  # noqa: F401
from .gqa_ops import GQAOntology  # noqa: F401
from .visual_oracle import ClassifierOracle, EmbeddingLayer, OracleBase, RegularMLP  # noqa: F401
from .interpreter import BatchGQABoxFeaturizer, BatchGQAInterpreter, BatchInterpreterBase, gather_results  # noqa: F401
from .program import OperatorBatch, ProgramBatch, ProgramCollaterBase  # noqa: F401
from .data import (BatchGQABoxFeaturesCollator, GQADataManager, GQAProgramVerifier, MultiSetSampler, MultiSetSequencialSampler,  # noqa: F401
                   ParserError, ProgramCodec, ProgramDataset)
from .preprocess import GQAPreprocessor, normalize  # noqa: F401
