"""Minimal stand-in for pytorch3d 0.7.4 (the version the reference pins, environment.yml:161) -- only what SuGaR's hot
path touches.  See sugar_amd/shims/__init__.py."""
__version__ = "0.7.4+sugar_amd.shim"
