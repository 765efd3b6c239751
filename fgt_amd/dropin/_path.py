"""Make `fgt_amd` importable when only fgt_amd/dropin is on sys.path (as tool/video_inpainting.py sets it up)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
