"""Import shim: the product package lives in the directory `symbolic-music-generation_amd/` (a name Python cannot
import directly); this package forwards its search path there so `import symbolic_music_generation_amd.xl` works."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'symbolic-music-generation_amd')
__path__.append(_real)
with open(_os.path.join(_real, '__init__.py')) as _f:
    exec(compile(_f.read(), _os.path.join(_real, '__init__.py'), 'exec'))
del _os, _f, _real
