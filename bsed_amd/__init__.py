"""Import alias: ``import bsed_amd`` loads the package that lives in the directory
``bird-sound-event-detecion_amd/`` (a name Python cannot import directly because of the hyphens)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "bird-sound-event-detecion_amd")
__path__.insert(0, _real)
exec(compile(open(_os.path.join(_real, "__init__.py")).read(), _os.path.join(_real, "__init__.py"), "exec"))
