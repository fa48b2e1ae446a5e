"""Import alias: the product package lives in ``tf-flowavenet_amd/`` (a name Python
cannot import directly); this stub points the importable name at that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                          "tf-flowavenet_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
