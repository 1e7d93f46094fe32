"""Import shim: the package sources live in `svgp-vae_amd/` (not an importable name);
`import svgp_vae_amd` resolves its submodules from there."""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "svgp-vae_amd")
__path__.append(_PKG_DIR)

from ._lib import LIB_PATH, SvgpError, load_library  # noqa: E402,F401

__all__ = ["LIB_PATH", "SvgpError", "load_library"]
