"""Reference package path `utils` (src/utils/).  Only `utils.load_pretrained` is replaced; every other submodule (`utils.logging`, which
train.py:14 / eval.py:20 import) must keep resolving to the reference's own file, so the package path is extended with every other
`utils/` directory found on sys.path (this directory stays first)."""
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
__path__ = [_here]
for _p in _sys.path:
    _cand = _os.path.abspath(_os.path.join(_p or ".", "utils"))
    if _cand != _here and _os.path.isdir(_cand) and _cand not in __path__:
        __path__.append(_cand)
