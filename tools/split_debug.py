import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgt_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def run(N, H, W, C0, Cout, k, s, p, tile, reps=3):
    x = torch.randn(N, H, W, C0, device=dev)
    w = torch.randn(Cout, C0, k, k, device=dev) * 0.02
    pc = ops.PackedConv(w, torch.zeros(Cout, device=dev))
    xs = ops.split(x)
    ref = ops.conv2d(x, pc, stride=s, pad=p, tile="128x128", precision="bf16x3")
    ref2 = ops.conv2d(x, pc, stride=s, pad=p, tile=tile, precision="bf16x3")
    for r in range(reps):
        got = ops.conv2d(xs, pc, stride=s, pad=p, tile=tile, precision="bf16x3")
        torch.cuda.synchronize()
        bad = (got != ref)
        nb = int(bad.sum())
        msg = f"N={N} {H}x{W} C={C0}->{Cout} k{k}s{s}p{p} tile={tile} rep{r}: old-vs-old equal={torch.equal(ref, ref2)} mismatches={nb}/{got.numel()} nan={int(torch.isnan(got).sum())}"
        if nb:
            idx = bad.nonzero()
            rows = (idx[:, 0] * got.shape[1] * got.shape[2] + idx[:, 1] * got.shape[2] + idx[:, 2])
            ur = torch.unique(rows)
            msg += f" rows={ur.numel()} first rows {ur[:12].tolist()} last {ur[-4:].tolist()} ch {torch.unique(idx[:,3])[:8].tolist()} maxdiff={float((got-ref).abs().max()):.3e}"
            r0 = int(ur[0]); oy, ox = (r0 // got.shape[2]) % got.shape[1], r0 % got.shape[2]
            msg += f" (first: oy={oy} ox={ox})"
        print(msg, flush=True)
for N in (1, 2, 4, 8, 17):
    run(N, 60, 108, 256, 384, 3, 1, 1, "128x128x8", reps=1)
run(17, 60, 108, 256, 384, 3, 1, 0, "128x128x8", reps=1)     # no padding
run(17, 60, 108, 64, 128, 3, 1, 1, "128x128x8", reps=1)
run(17, 60, 108, 256, 384, 1, 1, 0, "128x128x8", reps=1)
run(4, 60, 108, 256, 384, 3, 1, 1, "64x64", reps=2)
os.environ["X"] = "1"
