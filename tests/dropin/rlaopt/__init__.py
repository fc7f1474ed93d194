"""Alias package: ``import rlaopt`` (and every ``rlaopt.x.y``) resolves to ``rlaopt_amd`` - used only by
tests/test_reference_suite_dropin.py to run the REFERENCE's own test files against this package."""
import importlib, pkgutil, sys
def _alias(name):
    try:
        mod = importlib.import_module(name)
    except ImportError:
        return  # e.g. the native library sitting in the package directory
    sys.modules["rlaopt" + name[len("rlaopt_amd"):]] = mod
    if hasattr(mod, "__path__"):
        for m in pkgutil.iter_modules(mod.__path__):
            _alias(name + "." + m.name)
_alias("rlaopt_amd")
