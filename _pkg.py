"""Loader for the product package.  The package directory is named `fhe-ram_amd/` (hyphen, as
the layout contract prescribes), which Python cannot import by name; this helper loads it as
module `fheram_amd`."""
import importlib.util
import os
import sys

_NAME = "fheram_amd"


def load_package():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    root = os.path.dirname(os.path.abspath(__file__))
    pkg_dir = os.path.join(root, "fhe-ram_amd")
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
