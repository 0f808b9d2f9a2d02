"""Importable alias of the package directory ``clg-vqa_amd/`` (a hyphen is not a legal Python
identifier).  All sources live in ``clg-vqa_amd/``; this shim points ``__path__`` there and runs
that directory's ``__init__.py`` in this module's namespace."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "clg-vqa_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
