"""Drop-in for the reference's RAFT package: `from RAFT import RAFT` (tool/video_inpainting.py:17,186-197)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _path  # noqa: E402,F401
from fgt_amd.raft_model import RAFT  # noqa: E402,F401
