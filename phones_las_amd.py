"""Import shim: the package directory is ``phones-las_amd/`` (not an identifier), this module
exposes it as ``phones_las_amd``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'phones-las_amd')]
with open(_os.path.join(__path__[0], '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], '__init__.py'), 'exec'))
del _os, _f
