#!/usr/bin/env python
"""Check of the fp16 attention's fragment-prefetch variant (FGT_ATTN_PREFETCH=1: attn_split_kernel<8, true, true, 4, PF = true>).  Runs the same
temporal calls in two child processes (the switch is read once per process) and compares: the variant issues the same MFMAs in the same
order, so the outputs must be bit-identical; prints both timings (profiles/r02_run12_attn_prefetch_check.txt: identical; 8 % faster on the bench-sized call).

    timeout 120 python tools/attn_prefetch_check.py        (on the MI355X)
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(1, 13, 20, 36, None), (2, 17, 20, 36, 11), (8, 17, 20, 36, None), (1, 26, 40, 72, 11)]      # (b, t, nh, nw, tq)


def child(path):
    sys.path.insert(0, ROOT)
    import torch
    from fgt_amd import ops
    dev = torch.device("cuda:0")
    outs, times = [], []
    for b, t, nh, nw, tq in CASES:
        g = torch.Generator().manual_seed(100 + t)
        qkv = torch.randn(b * t * nh * nw, 1536, generator=g)
        qkv[:, :1024] *= 1.5
        sp = ops.split(qkv.to(dev), h=True)
        fn = lambda: ops.attention_temporal(sp, b, t, nh, nw, 4, 2, 512, tq=tq)
        o = fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        outs.append(o.cpu())
        times.append(e0.elapsed_time(e1) / 5 * 1e3)
    torch.save({"outs": outs, "times": times}, path)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        return
    import torch
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for pf in ("0", "1"):
            path = os.path.join(d, f"pf{pf}.pt")
            r = subprocess.run([sys.executable, __file__, "--child", path], env=dict(os.environ, FGT_ATTN_PREFETCH=pf), timeout=45)
            if r.returncode != 0:
                print(f"FGT_ATTN_PREFETCH={pf}: child failed with {r.returncode}")
                sys.exit(1)
            res[pf] = torch.load(path)
    ok = True
    for i, c in enumerate(CASES):
        a, b = res["0"]["outs"][i], res["1"]["outs"][i]
        same = torch.equal(a, b)
        ok &= same
        print(f"b={c[0]} t={c[1]} {c[2]}x{c[3]} tq={c[4]}: {'bit-identical' if same else 'DIFFERENT: max |diff| %.3e' % (a - b).abs().max().item()};  "
              f"{res['0']['times'][i]:.0f} us -> {res['1']['times'][i]:.0f} us with prefetch")
    sys.exit(0 if ok else 2)


if __name__ == "__main__":
    main()
