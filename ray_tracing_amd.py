"""Import alias: the package directory is `ray-tracing_amd/` (not a valid Python
identifier), so this one-file module turns itself into that package."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "ray-tracing_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
