"""Drop-in for the reference's FGT/models/model.py: `import_module("FGT.models.model").Model(config)`
(tool/video_inpainting.py:217-230) resolves to the MI355X implementation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _path  # noqa: E402,F401
from fgt_amd.fgt_model import FGT, Model  # noqa: E402,F401
