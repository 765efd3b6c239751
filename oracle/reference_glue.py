"""Executes pieces of the reference's tool/video_inpainting.py WITHOUT importing it (TEST INFRASTRUCTURE ONLY).

The tool cannot be imported here: it needs cv2 / torchvision / imageio / cvbase at import and uses `np.bool`.
Its pure-python helpers and the sliding-window compose loop need none of that, so they are cut out of the source with `ast`
and compiled on their own:

  * `indicesGen`, `get_ref_index`, `norm_flows`                      tool/video_inpainting.py:90-117, 402-407
  * the window loop `for f in range(0, video_length, neighbor_stride): ...` of `video_inpainting()`   :710-740
    (neighbour / reference selection, model call, uint8 truncation, ordered 0.5/0.5 blend)

This is what pins `fgt_amd.scheduler.window_schedule`, `flow_pipeline.indices_gen`, `oracle.fgt_oracle.fgt_clip` and the device
compose kernel to the reference's *code* rather than to a second copy of the same text (tests/test_glue_pinned.py; golden
vectors from tests/golden/make_golden_glue.py).  Only tests and the golden generator may import this module; the reference
tree exists only in the authoring container (`available()` is False on the GPU box).
"""
import ast
import os

REF = os.environ.get("FGT_REFERENCE", "/root/reference")
TOOL = os.path.join(REF, "tool", "video_inpainting.py")


def available():
    return os.path.isfile(TOOL)


def _tree():
    with open(TOOL) as f:
        return ast.parse(f.read(), TOOL)


def functions(names=("indicesGen", "get_ref_index", "norm_flows")):
    """{name: function} compiled from the reference's own `def` nodes (globals: torch, numpy as np)."""
    import numpy as np
    import torch
    tree = _tree()
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(d.name for d in defs) == sorted(names), "reference layout changed"
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(body=defs, type_ignores=[]), TOOL, "exec"), ns)
    return {n: ns[n] for n in names}


def window_loop():
    """The `for f in range(0, video_length, neighbor_stride)` statement of video_inpainting() as a callable:
    run(FGT_model, frames_first [1,N,3,H,W], masks [1,N,1,H,W], flows [1,N,2,H,W], neighbor_stride, ref_length, num_ref)
    -> (comp_frames list of numpy arrays exactly as the tool holds them before `.astype(np.uint8)`, log of (f, len(nb), len(ref)))."""
    import numpy as np
    import torch
    tree = _tree()
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "video_inpainting")
    loops = [n for n in ast.walk(fn) if isinstance(n, ast.For) and isinstance(n.target, ast.Name) and n.target.id == "f"
             and isinstance(n.iter, ast.Call) and getattr(n.iter.func, "id", "") == "range" and len(n.iter.args) == 3
             and getattr(n.iter.args[1], "id", "") == "video_length"]
    assert len(loops) == 1, "reference layout changed: window loop not found"
    code = compile(ast.Module(body=[loops[0]], type_ignores=[]), TOOL, "exec")
    helpers = functions(("get_ref_index",))

    def run(FGT_model, frames_first, masks, flows, neighbor_stride=5, ref_length=10, num_ref=-1):
        log = []
        n = frames_first.shape[1]
        ns = {"torch": torch, "np": np, "get_ref_index": helpers["get_ref_index"], "FGT_model": FGT_model,
              "frames_first": frames_first, "masks": masks, "flows": flows, "normed_frames": frames_first * 2 - 1,
              "comp_frames": [None] * n, "video_length": n, "neighbor_stride": neighbor_stride, "ref_length": ref_length,
              "num_ref": num_ref, "print": lambda *a: log.append(tuple(a))}
        exec(code, ns)
        return ns["comp_frames"], log

    return run
