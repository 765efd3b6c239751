"""Flow-guided gradient propagation behind the reference's own call signature.

`get_flowNN_gradient(args, gradient_x, gradient_y, mask_RGB, mask, videoFlowF, videoFlowB, None, None)` is what
tool/video_inpainting.py:623-633 calls (tool/get_flowNN_gradient.py:11): numpy arrays with the frame index LAST
(gradients [H,W,3,N], mask [H,W,N] bool, flows [H,W,2,N-1]).  This drop-in moves them to the MI355X once, runs the whole clip
in one `fgt_flow_propagate` call (csrc/propagate.hip) and returns numpy arrays in the reference layout.
`propagate_gradients` is the device-tensor entry point (frame-major, no host copies) for callers that keep the clip in HBM.

cv2.remap(INTER_LINEAR) is replaced by the bilinear sampler specified in oracle/prop_oracle.py (`tab = 32`: 1/32-pixel
coordinate table like OpenCV's; `tab = 0`: float bilinear); everything else is the reference's arithmetic, bit for bit
(tests/test_prop_pinned.py).  `args.Nonlocal = True` (the non-local candidates, off by default in the tool) is not built.
"""
import numpy as np
import torch

from . import ops


def propagate_gradients(gradient_x, gradient_y, mask, flow_f, flow_b, consistency_thres=5.0, alpha=0.1, tab=32):
    """Device tensors, frame-major: gradients [N,H,W,3] fp32, mask [N,H,W] (non-zero = hole), flows [N-1,H,W,2] (u, v) fp32.
    Returns (gradient_x, gradient_y [N,H,W,3], mask_tofill [N,H,W] bool)."""
    return ops.flow_propagate(gradient_x, gradient_y, mask, flow_f, flow_b, consistency_thres, alpha, tab)


def get_flowNN_gradient(args, gradient_x, gradient_y, mask_RGB, mask, videoFlowF, videoFlowB, videoNonLocalFlowF=None,
                        videoNonLocalFlowB=None, device="cuda", tab=32):
    """Reference signature and layouts (tool/get_flowNN_gradient.py:11-30).  mask_RGB is unused by the reference as well."""
    if getattr(args, "Nonlocal", False):
        raise NotImplementedError("Nonlocal=True candidates (tool/get_flowNN_gradient.py:430-468) are not built; the tool's default is False")
    dev = torch.device(device)
    fm = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.moveaxis(np.asarray(a), -1, 0)).astype(dt, copy=False)).to(dev)
    N = np.asarray(mask).shape[-1]
    ff = fm(videoFlowF, np.float32) if N > 1 else None
    fb = fm(videoFlowB, np.float32) if N > 1 else None
    gx, gy, fill = propagate_gradients(fm(gradient_x, np.float32), fm(gradient_y, np.float32), fm(mask, np.uint8), ff, fb,
                                       float(args.consistencyThres), float(args.alpha), tab)
    back = lambda t: np.moveaxis(t.cpu().numpy(), 0, -1)
    return back(gx), back(gy), back(fill).astype(bool)
