"""Import shim: the package directory is `dfol-vqa_amd/` (not a legal Python identifier),
so `import dfol_vqa_amd` lands here and is redirected to that directory."""

import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "dfol-vqa_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _f.name, "exec"))
del _f
