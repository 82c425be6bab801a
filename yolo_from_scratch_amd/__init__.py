"""Import alias: the product package lives in the directory `yolo-from-scratch_amd/` (a name Python
cannot import directly); this stub makes it importable as `yolo_from_scratch_amd`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "yolo-from-scratch_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
