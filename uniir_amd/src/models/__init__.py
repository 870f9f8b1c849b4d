"""`models` package of the drop-in source tree (PYTHONPATH=<repo>/uniir_amd/src, like the reference's $SRC).
Makes the repository root importable so that `import uniir_amd` works when the shell drivers only export $SRC."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
