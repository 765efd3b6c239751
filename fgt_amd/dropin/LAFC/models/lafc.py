"""Drop-in for the reference's LAFC/models/lafc.py: `import_module("LAFC.models.lafc").Model(config)`
(tool/video_inpainting.py:200-214) resolves to the MI355X implementation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _path  # noqa: E402,F401
from fgt_amd.lafc_model import Model, P3DNet  # noqa: E402,F401
