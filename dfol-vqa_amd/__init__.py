"""MI355X-native hot path of the ∇-FOL (DFOL-VQA) program interpreter.

Host-side mirror of the reference's operator/program API (reference
`src/nsvqa/nn/interpreter/*`), calling hand-written gfx950 HIP kernels through
the C-ABI library declared in `include/dfol_vqa.h`.
"""

__version__ = "0.1.0"
