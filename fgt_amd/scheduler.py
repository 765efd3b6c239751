"""Clip-level host logic of the FGT stage (tool/video_inpainting.py:687-748) for the MI355X path.

* `window_schedule`  — the reference's sliding window + reference-frame selection (:103-117, :710-717).
* `prepare_flows`    — norm_flows + duplication of the last forward flow (:402-407, :703-707) on device.
* `ClipRunner`       — keeps the clip resident in HBM, runs every window through the HIP model, composes and
                       blends on device in ascending window order (the blend is order dependent, :731-740), and
                       returns the composited clip once (one D2H instead of one per window).
* feature cache      — the per-frame stages (conv encoders + soft split) run once per frame and clip pass instead of once per
                       window the frame appears in; windows gather their frames' features (exact).
* window batching    — windows of equal length are independent batch elements of the model: up to `window_batch` of them run
                       as one forward (bit-identical outputs, larger launches).
* window sharding    — windows are independent.  `assign_windows` gives every rank a cost-balanced set in which
                       equal-length windows are co-located (so they batch); frames are block-sharded for the per-frame stages and
                       their features all-gathered chunk by chunk while the next chunk is being encoded; the per-window outputs
                       are exchanged ALREADY TRUNCATED to uint8 (the first thing the compose does with them) in one all-gather of
                       preallocated buffers, and every rank applies the ordered blend locally.  RCCL over xGMI on the GPU box,
                       gloo in the CPU tests.  Nothing on the GPU path of a step allocates exchange buffers or runs a torch
                       operator on activations: packing, gathers, truncation and compose are libfgt_hip.so kernels.

The per-rank model call is injectable (`forward=`) so the scheduling / sharding / blend logic is testable on CPU
with the oracle standing in for the device model (tests/test_scheduler.py, tests/test_scheduler_cache.py).
"""
import torch

from . import ops


def window_schedule(n_frames, neighbor_stride=5, ref_length=10, num_ref=-1):
    """[(neighbor_ids, ref_ids)] exactly as tool/video_inpainting.py:710-717 + get_ref_index (:103-117)."""
    sched = []
    for f in range(0, n_frames, neighbor_stride):
        nb = list(range(max(0, f - neighbor_stride), min(n_frames, f + neighbor_stride + 1)))
        if num_ref == -1:
            ref = [i for i in range(0, n_frames, ref_length) if i not in nb]
        else:
            ref = []
            lo = max(0, f - ref_length * (num_ref // 2))
            hi = min(n_frames, f + ref_length * (num_ref // 2))
            for i in range(lo, hi + 1, ref_length):
                if i not in nb:
                    if len(ref) > num_ref:
                        break
                    ref.append(i)
        sched.append((nb, ref))
    return sched


def prepare_flows(forward_flows):
    """tool/video_inpainting.py:703-707: `videoFlowF` [N-1,2,H,W] (completed forward flows, device fp32) -> the model's flow input
    [1,N,2,H,W]: last flow duplicated to the clip length, every (frame, channel) map divided by its signed maximum."""
    return ops.norm_flows(forward_flows, n_out=forward_flows.shape[0] + 1).unsqueeze(0)


def window_cost(t, n_nb):
    """Relative cost of one window on the feature-cache path (GFLOP at 240x432): the transformer blocks on all
    t frames (4 x (TMHSA 1.51 + SWMHSA 4.19) + 8 x FFN 2.89 per frame, temporal attention 1.0618 t^2) and soft composition +
    decoder on the n_nb consumed frames (4.62 + 19.47)."""
    return 45.9 * t + 1.0618 * t * t + 24.1 * n_nb


def assign_windows(sched, world):
    """Partition the window indices over `world` ranks: [[window ids of rank 0], ...], each ascending.
    Equal-length windows are packed together (a rank runs them as ONE batched forward), packs are placed longest-first on
    the least-loaded rank (LPT).  80 frames / 8 ranks: 4 x (17,17), 3 x (18,18), (18,13): makespan 36 frame-units against
    275/8 = 34.4 ideal, every rank but one batches its two windows."""
    n = len(sched)
    quota = -(-n // world)
    by_t = {}
    for wi, (nb, ref) in enumerate(sched):
        by_t.setdefault(len(nb) + len(ref), []).append(wi)
    packs = []
    for t, ws in by_t.items():
        for i in range(0, len(ws), quota):
            pack = ws[i:i + quota]
            packs.append((sum(window_cost(t, len(sched[w][0])) for w in pack), pack))
    packs.sort(key=lambda p: (-p[0], p[1][0]))
    load, out = [0.0] * world, [[] for _ in range(world)]
    for cost, pack in packs:
        r = min(range(world), key=lambda i: (load[i], i))
        load[r] += cost
        out[r] += pack
    return [sorted(ws) for ws in out]


def ideal_speedup(sched, world):
    """Upper bound of the sharded window phase's speed-up given the assignment (total cost / most loaded rank)."""
    cost = lambda ws: sum(window_cost(len(sched[w][0]) + len(sched[w][1]), len(sched[w][0])) for w in ws)
    parts = assign_windows(sched, world)
    return cost(range(len(sched))) / max(cost(p) for p in parts)


def needs_host_staging(is_cuda, backend):
    """RCCL ("nccl" IS RCCL on ROCm) gathers device buffers directly over xGMI; gloo cannot touch device memory."""
    # (a group created without an explicit backend reports a composite string such as "cuda:nccl,cpu:gloo")
    return bool(is_cuda) and "nccl" not in str(backend).lower()


class _Done:
    def wait(self):
        return True


_staging_logged = False


def all_gather(out, buf, group=None, async_op=False):
    """all_gather_into_tensor.  Returns a work handle (`.wait()` makes the current stream wait for the result).
    The gloo backend (CPU tests, and the 2-ranks-on-one-GPU rehearsal of the sharded path on a single-GPU box) cannot gather
    device tensors, so there the call is staged through host memory (synchronously)."""
    import torch.distributed as dist
    if needs_host_staging(buf.is_cuda, dist.get_backend(group)):
        global _staging_logged
        if not _staging_logged:
            _staging_logged = True
            import sys
            print(f"[fgt_amd.scheduler] backend {dist.get_backend(group)!r} cannot gather device buffers: collectives are staged through "
                  "host memory (synchronous; rehearsal / test path, not the RCCL path)", file=sys.stderr)
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, buf.cpu(), group=group)
        out.copy_(host)
        return _Done()
    work = dist.all_gather_into_tensor(out, buf, group=group, async_op=async_op)
    return work if async_op else _Done()


def all_to_all_rows(out, inp, out_rows, in_rows, group=None, async_op=False):
    """all_to_all_single over leading-dimension rows: `inp` holds in_rows[q] consecutive rows for every rank q (in rank order), `out`
    receives out_rows[r] rows from every rank r.  RCCL moves the device buffers rank to rank over xGMI (point-to-point links: an
    all-to-all of exactly the rows each rank needs is the natural collective on this fabric); gloo (CPU tests, single-GPU rehearsal)
    stages device tensors through the host like `all_gather`."""
    import torch.distributed as dist
    if needs_host_staging(inp.is_cuda, dist.get_backend(group)):
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host, inp.cpu(), output_split_sizes=list(out_rows), input_split_sizes=list(in_rows), group=group)
        out.copy_(host)
        return _Done()
    work = dist.all_to_all_single(out, inp, output_split_sizes=list(out_rows), input_split_sizes=list(in_rows), group=group, async_op=async_op)
    return work if async_op else _Done()


def compose_torch(out, nb, frames01, masks, comp, visited):
    """Reference compose/blend restated with torch ops (CPU path used only by the CPU tests).  `out`: model output (fp32) or
    its uint8 truncation."""
    filled = out.permute(0, 2, 3, 1).float() if out.dtype == torch.uint8 else (((out + 1) / 2).permute(0, 2, 3, 1) * 255).to(torch.uint8).float()
    for i, idx in enumerate(nb):
        valid = (frames01[0, idx].permute(1, 2, 0) * 255.0).to(torch.uint8).float()
        m = masks[0, idx].permute(1, 2, 0)
        c = filled[i] * m + valid * (1 - m)
        comp[idx] = c if not visited[idx] else comp[idx] * 0.5 + c * 0.5
        visited[idx] = True


class ClipRunner:
    def __init__(self, model, frames01, flows_normed, masks, neighbor_stride=5, ref_length=10, num_ref=-1,
                 rank=0, world=1, forward=None, group=None, cache_features=None, encode_chunk=20, use_graphs=False, window_batch=8,
                 n_streams=None, prune_last=True, exchange=None):
        self.model = model
        # prune_last: the last transformer pair computes only the frames the tool consumes (FGT.transform_decode `tq`): exact
        self.prune_last = bool(prune_last)
        self.frames01, self.flows, self.masks = frames01, flows_normed, masks
        self.n = frames01.shape[1]
        self.H, self.W = frames01.shape[-2:]
        self.sched = window_schedule(self.n, neighbor_stride, ref_length, num_ref)
        self.rank, self.world, self.group = rank, world, group
        self.assign = assign_windows(self.sched, world)
        self.mine = self.assign[rank]
        self.forward = forward or (lambda mf, fl, ms: model(mf, fl, ms))
        self.dev = frames01.device
        self.on_gpu = self.dev.type == "cuda"
        self.max_nb = max(len(nb) for nb, _ in self.sched)
        # per-window index tensors, built once (host scheduling is outside the per-step hot loop)
        self._ids = [torch.tensor(nb + ref, device=self.dev) for nb, ref in self.sched]
        seen = set()
        self._first = []
        for nb, _ in self.sched:
            self._first.append(torch.tensor([0 if i in seen else 1 for i in nb], dtype=torch.int32, device=self.dev))
            seen.update(nb)
        self._nb = [torch.tensor(nb, dtype=torch.int32, device=self.dev) for nb, _ in self.sched]
        self._normed = None
        # Exact dedup: the conv encoders / soft split depend only on the frame, yet the reference
        # recomputes them for every window a frame appears in (275 frame passes for 80 frames).  With the cache each
        # frame is encoded once per clip pass (frames sharded over ranks + all-gathers), windows only run the
        # transformer + decoder.  Needs a model exposing encode_frames / transform_decode (fgt_amd.fgt_model.FGT).
        net = getattr(model, "net", None)
        can_cache = forward is None and hasattr(net, "encode_frames")
        self.cache_features = can_cache if cache_features is None else (cache_features and can_cache)
        self.encode_chunk = max(1, int(encode_chunk))
        # hipGraph replay of the per-window launch sequence (one graph per window length t), only with the feature cache; opt-in
        self.use_graphs = bool(use_graphs) and self.on_gpu and self.cache_features
        self._graphs = None
        # Window batching (feature-cache path): windows of equal length t are independent batch elements of the reference model
        # (b > 1: every stage is per frame or per (b, zone)), so up to `window_batch` of this rank's windows go through the
        # transformer + decoder as ONE forward.  Per-row results do not depend on how many rows a GEMM launch carries, so the
        # outputs are bit-identical to running the windows one by one; the launches are up to 8x larger (M = 12240 rows per
        # window leaves 0.75- and 2.25-round tile grids on 256 CUs) and 3-4x fewer.
        self.window_batch = max(1, int(window_batch))
        by_t = {}
        for wi in self.mine:
            by_t.setdefault(len(self.sched[wi][0]) + len(self.sched[wi][1]), []).append(wi)
        self.groups = []
        for t, ws in sorted(by_t.items()):
            nb = min(self.window_batch, self._max_batch(t))
            self.groups += [ws[i:i + nb] for i in range(0, len(ws), nb)]
        # ---- frame sharding of the per-frame stages: rank r encodes frames [r*per, (r+1)*per) in `n_chunks` chunks of `ck` frames
        # (`encode_chunk` frames per call at most, at least two chunks so that the exchange of chunk j travels while chunk j+1 is
        # encoded).  A rank only needs the frames its own windows reference (its neighbours + the stride-10 reference frames: ~30 of 80
        # at 8 ranks): after every chunk an ALL-TO-ALL delivers exactly those rows — rank r sends to rank q the frames of the chunk that
        # q's windows use — instead of an all-gather of every feature to everyone (442 MB per 80-frame clip, 3.5 GB at 864x480x160).
        # Feature row of frame f in this rank's buffers = _row_of[f] (identity for world == 1; -1: not held here).
        # Ranks whose block is (partly) past the clip encode nothing there and send / receive zero rows: every rank still enters every
        # collective, so they stay matched for any (frames, ranks).
        self.per = -(-self.n // world)
        if world > 1:
            self.n_chunks = max(2 if self.per >= 2 else 1, -(-self.per // self.encode_chunk))
            self.ck = -(-self.per // self.n_chunks)
        else:
            self.n_chunks, self.ck = -(-self.n // self.encode_chunk), self.encode_chunk
        self._row_of = list(range(self.n))
        self.rows = self.n
        import os
        # FGT_EXCHANGE=allgather: replace the needed-rows all-to-all by a plain all_gather_into_tensor of every chunk (equal-sized
        # contributions, the most ordinary collective there is) — the degraded mode for a first RCCL run in which the uneven / zero-length
        # all_to_all_single misbehaves.  Same composite (tests/test_scheduler.py), ~2.7x the bytes on the wire at 8 ranks.
        # (`exchange=` overrides the environment: bench.py --gpus N runs the sharded clip once per mode)
        self.exchange = "none" if world == 1 else (exchange or os.environ.get("FGT_EXCHANGE", "a2a")).lower()
        if world > 1 and self.exchange not in ("a2a", "allgather"):
            raise ValueError(f"FGT_EXCHANGE={self.exchange!r}: expected 'a2a' or 'allgather'")
        if self.exchange == "allgather":
            # chunk j's gathered block = rows [j*world*ck, (j+1)*world*ck): rank r's frames of the chunk at offset r*ck
            self._plan = None
            self.rows = self.n_chunks * world * self.ck
            self._row_of = [-1] * self.n
            for f in range(self.n):
                r, o = divmod(f, self.per)
                j, i = divmod(o, self.ck)
                self._row_of[f] = j * world * self.ck + r * self.ck + i
        elif world > 1:
            need = [sorted({f for wi in self.assign[q] for f in self.sched[wi][0] + self.sched[wi][1]}) for q in range(world)]
            self.need = need
            chunk_of = lambda r, j: range(min(self.n, r * self.per + j * self.ck), min(self.n, r * self.per + (j + 1) * self.ck, (r + 1) * self.per))
            self._row_of = [-1] * self.n
            self._plan = []                        # per chunk: (send ids [local rows of this rank's block], in_rows per dest, out_rows per source, first recv row)
            row = 0
            for j in range(self.n_chunks):
                mine_j = chunk_of(rank, j)
                send, in_rows = [], []
                for q in range(world):
                    fs = [f for f in mine_j if f in set(need[q])]
                    send += [f - rank * self.per for f in fs]
                    in_rows.append(len(fs))
                out_rows, r0 = [], row
                for r in range(world):
                    fs = [f for f in chunk_of(r, j) if f in set(need[rank])]
                    for f in fs:
                        self._row_of[f] = row
                        row += 1
                    out_rows.append(len(fs))
                self._plan.append((torch.tensor(send, dtype=torch.int32, device=self.dev), in_rows, out_rows, r0))
            self.rows = max(row, 1)
        self._group_ids, self._group_keep, self._group_tq, self._group_keep_q = [], [], [], []
        for ws in self.groups:
            t = len(self.sched[ws[0]][0]) + len(self.sched[ws[0]][1])
            self._group_ids.append(torch.tensor([self._row_of[f] for wi in ws for f in self.sched[wi][0] + self.sched[wi][1]],
                                                dtype=torch.int32, device=self.dev))
            self._group_keep.append(torch.tensor([j * t + i for j, wi in enumerate(ws) for i in range(len(self.sched[wi][0]))],
                                                 dtype=torch.int32, device=self.dev))
            # the consumed (neighbour) frames are the first len(neighbour_ids) of a window: the last transformer pair produces only
            # the first tq frames of every window of the group (FGT.transform_decode)
            tq = max(len(self.sched[wi][0]) for wi in ws)
            self._group_tq.append(tq if self.prune_last else None)
            self._group_keep_q.append(torch.tensor([j * tq + i for j, wi in enumerate(ws) for i in range(len(self.sched[wi][0]))],
                                                   dtype=torch.int32, device=self.dev))
        # window groups on `n_streams` concurrent HIP streams (default 1; FGT_STREAMS): +1.8 % clip throughput with 2 on the bench
        # clip, bit-identical composite — off by default because overlapped launches make the per-launch event timings of
        # bench.py's roofline blocks meaningless
        self.n_streams = int(os.environ.get("FGT_STREAMS", "1")) if n_streams is None else int(n_streams)
        self._streams = None
        self._feat = None          # persistent feature buffers (enc, tok, ftok, th, tw) + local chunk buffers
        self._xchg = None          # persistent uint8 exchange buffers

    # ---------------------------------------------------------------------------------------------------------------------
    def _max_batch(self, t):
        """Largest window batch whose spatial attention still fits one launch: fgt_attention maps one (frame, window, head)
        problem to a grid.y index (<= 65535)."""
        cfg = getattr(getattr(self.model, "net", None), "cfg", None)
        if not cfg:
            return 1
        tok = lambda n, i: (n // 4 + 2 * cfg["p"][i] - cfg["k"][i]) // cfg["s"][i] + 1
        th, tw, ws = tok(self.H, 0), tok(self.W, 1), cfg["ws"]
        per_frame = -(-th // ws) * -(-tw // ws) * cfg["heads"]
        return max(1, 65535 // (t * per_frame))

    def run_window(self, wi):
        """One window exactly like the tool calls the model (:718-724): torch indexing on the caller's side of the nn.Module API."""
        if self._normed is None:
            self._normed = self.frames01 * 2 - 1                 # tool/video_inpainting.py:697
        ids = self._ids[wi]
        m = self.masks[:, ids]
        mf = self._normed[:, ids] * (1 - m)                      # :721
        return self.forward(mf, self.flows[:, ids], m)[: len(self.sched[wi][0])]

    # ---------------------------------------------------------------------------------------------------------------------
    def _feature_buffers(self):
        if self._feat is None:
            net = self.model.net
            Hf, Wf = self.H // 4, self.W // 4
            th, tw = net.token_grid(self.H, self.W)
            C, c, cf = net.cfg["cnum"] * 2, net.cfg["c"], net.cfg["cf"]
            new = lambda rows, *s: torch.zeros(rows, *s, dtype=torch.float32, device=self.dev)
            full = (new(self.rows, Hf, Wf, C), new(self.rows, th * tw, c), new(self.rows, th * tw, cf))
            local = send = None
            if self.exchange == "allgather":
                local = tuple(new(self.n_chunks * self.ck, *b.shape[1:]) for b in full)      # chunk j at rows [j*ck, (j+1)*ck), zero rows past the clip
            elif self.world > 1:
                # this rank's block as encoded, and one send buffer per chunk (rows ordered by destination rank; a frame that several
                # ranks need appears once per destination)
                local = tuple(new(max(self.per, 1), *b.shape[1:]) for b in full)
                send = [tuple(new(max(ids.numel(), 1), *b.shape[1:]) for b in full) for ids, _, _, _ in self._plan]
            self._feat = (full, local, th, tw, send)
        return self._feat

    def _encode_chunk(self, s0, s1, dst):
        """Per-frame stages for frames [s0, s1) written into dst = (enc, tok, ftok) row slices."""
        net = self.model.net
        k = s1 - s0
        if self.on_gpu and net.passmask and net.in_channels == 4:
            fin = ops.ceil_to(net.cfg["flow_in"], 4)
            x_in = ops.pack_frames(self.frames01[0, s0:s1], self.masks[0, s0:s1])
            f_in = torch.empty(k, self.H, self.W, fin, dtype=torch.float32, device=self.dev)
            ops.nchw_to_nhwc(self.flows[0, s0:s1], f_in, coff=0, zero_to=fin)
            net.encode_frames(packed_in=(x_in, f_in), out=tuple(d[:k] for d in dst))
        elif self.on_gpu:   # models without the mask channel (PASSMASK = 0 / other in_channel): the nn.Module-style call, written in place
            m = self.masks[:, s0:s1]
            net.encode_frames((self.frames01[:, s0:s1] * 2 - 1) * (1 - m), self.flows[:, s0:s1], m, out=tuple(d[:k] for d in dst))
        else:       # CPU tests over tests/fake_ops.py: the nn.Module-style call, results copied into the buffers
            m = self.masks[:, s0:s1]
            enc, tok, ftok, _, _ = net.encode_frames((self.frames01[:, s0:s1] * 2 - 1) * (1 - m), self.flows[:, s0:s1], m)
            for d, v in zip(dst, (enc, tok, ftok)):
                d[:k].copy_(v.reshape(k, *d.shape[1:]))

    def encode_clip(self):
        """Per-frame stages for the whole clip into the persistent feature buffers:
        (enc [rows,Hf,Wf,C], tokens [rows,n,c], flow tokens [rows,n,cf], th, tw); frame f lives in row _row_of[f]."""
        full, local, th, tw, send = self._feature_buffers()
        if self.world == 1:
            for s0 in range(0, self.n, self.ck):
                s1 = min(self.n, s0 + self.ck)
                self._encode_chunk(s0, s1, tuple(b[s0:s1] for b in full))
            return full + (th, tw)
        lo = self.rank * self.per
        works = []
        if self.exchange == "allgather":
            W, ck = self.world, self.ck
            for j in range(self.n_chunks):
                s0 = min(self.n, lo + j * ck)
                s1 = min(self.n, lo + (j + 1) * ck, lo + self.per)
                if s1 > s0:
                    self._encode_chunk(s0, s1, tuple(l[j * ck:j * ck + s1 - s0] for l in local))
                for b, l in zip(full, local):
                    works.append(all_gather(b[j * W * ck:(j + 1) * W * ck], l[j * ck:(j + 1) * ck], self.group, async_op=True))
            self._t("encode")
            for w in works:
                w.wait()
            self._t("gather_wait")
            return full + (th, tw)
        for j, (ids, in_rows, out_rows, r0) in enumerate(self._plan):
            s0 = min(self.n, lo + j * self.ck)
            s1 = min(self.n, lo + (j + 1) * self.ck, lo + self.per)
            if s1 > s0:
                self._encode_chunk(s0, s1, tuple(l[s0 - lo:s1 - lo] for l in local))
            n_in, n_out = sum(in_rows), sum(out_rows)
            for b, l, sb in zip(full, local, send[j]):          # "boundary feature" exchange (RCCL / gloo), asynchronous:
                if n_in:                                         # chunk j travels while chunk j+1 is encoded
                    if self.on_gpu:
                        ops.gather_rows(l, ids, out=sb[:n_in])
                    else:
                        sb[:n_in] = l[ids.long()]
                works.append(all_to_all_rows(b[r0:r0 + n_out], sb[:n_in], out_rows, in_rows, self.group, async_op=True))
        self._t("encode")
        for w in works:
            w.wait()
        self._t("gather_wait")
        return full + (th, tw)

    def run_group_cached(self, gi, feats):
        """Transformer + decoder for one group of equal-length windows as a single batched forward; returns {window: out}."""
        enc, tok, ftok, th, tw = feats
        ws, ids, keep = self.groups[gi], self._group_ids[gi], self._group_keep[gi]
        tq, keep_q = self._group_tq[gi], self._group_keep_q[gi]
        b, bt = len(ws), ids.numel()
        t = bt // b
        if self.on_gpu:
            e, x, f = ops.gather_rows(enc, ids), ops.gather_rows(tok, ids).view(bt * th * tw, -1), ops.gather_rows(ftok, ids).view(bt * th * tw, -1)
        else:
            il = ids.long()
            e, x, f = enc[il], tok[il].reshape(bt * th * tw, -1), ftok[il].reshape(bt * th * tw, -1)
        net = self.model.net
        if self.use_graphs:
            if self._graphs is None:
                from .graph import GraphCache
                # b travels as the SHAPE of a dummy tensor: graphs are cached per input-shape signature (+ weights / arithmetic mode)
                # (tq as well: 0 = no pruning)
                self._graphs = GraphCache(lambda e_, x_, f_, k_, kq_, b_, tq_: net.transform_decode(e_, x_, f_, b_.shape[0], e_.shape[0] // b_.shape[0],
                                                                                                  th, tw, keep=k_, tq=tq_.shape[0] or None, keep_q=kq_),
                                          state_key=net._cache_key)
            out = self._graphs(e, x, f, keep, keep_q, torch.empty(b, device=self.dev), torch.empty(tq or 0, device=self.dev)).clone()   # the static buffer is reused by the next group of this shape
        else:
            out = net.transform_decode(e, x, f, b, t, th, tw, keep=keep, tq=tq, keep_q=keep_q)
        outs, o = {}, 0
        for wi in ws:
            nb = len(self.sched[wi][0])
            outs[wi] = out[o:o + nb]
            o += nb
        return outs

    # ---- per-phase timing (bench.py --gpus N: the first real multi-GPU curve must be diagnosable): with `timing = True` every phase
    # boundary of run() records an event on the main stream (GPU) / a wall-clock stamp (CPU); phase_ms() turns the last pass into
    # {phase: ms}.  "gather_wait" is what the feature exchange costs beyond the encode it overlaps.
    timing = False

    def _t(self, name):
        if not self.timing:
            return
        if self.on_gpu:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
        else:
            import time
            e = time.perf_counter()
        self._stamps.append((name, e))

    def phase_ms(self):
        out, prev = {}, None
        for name, e in getattr(self, "_stamps", []):
            if prev is not None:
                if self.on_gpu:
                    e.synchronize()
                    out[name] = out.get(name, 0.0) + prev.elapsed_time(e)
                else:
                    out[name] = out.get(name, 0.0) + (e - prev) * 1e3
            prev = e
        return {k: round(v, 3) for k, v in out.items()}

    def run(self):
        """One pass over the clip.  Returns comp [N,H,W,3] fp32 (0..255 scale, before the final astype(uint8))."""
        self._stamps = []
        self._t("start")
        comp = torch.empty(self.n, self.H, self.W, 3, dtype=torch.float32, device=self.dev)
        if self.cache_features:
            with torch.no_grad():
                feats = self.encode_clip()
                if self.world == 1:
                    self._t("encode")
                outs = {}
                if self.on_gpu and self.n_streams > 1 and len(self.groups) > 1:
                    # independent window groups on separate HIP streams: the HBM-bound kernels of one group (LayerNorm, fold, pools)
                    # and the tails of its GEMM launches overlap the matrix-core kernels of another.
                    # Every lazily filled cache the groups share (packed / split / fp16 weight images, zero rows) is filled HERE, on the
                    # main stream, before the fork: a cache filled by the first group's stream would be read by the next group's
                    # stream with no event between them.
                    self.model.net.prepack(self.dev)
                    main = torch.cuda.current_stream()
                    if self._streams is None:
                        self._streams = [torch.cuda.Stream() for _ in range(self.n_streams)]
                    for st in self._streams:
                        st.wait_stream(main)
                    for gi in range(len(self.groups)):
                        with torch.cuda.stream(self._streams[gi % self.n_streams]):
                            o = self.run_group_cached(gi, feats)
                        for v in o.values():
                            v.record_stream(main)                        # consumed by the compose kernels on the main stream
                        outs.update(o)
                    for st in self._streams:
                        main.wait_stream(st)
                else:
                    for gi in range(len(self.groups)):
                        outs.update(self.run_group_cached(gi, feats))
        else:
            outs = {wi: self.run_window(wi) for wi in self.mine}
        self._t("windows")
        if self.world > 1:
            outs = self._exchange(outs)
            self._t("exchange")
        if self.on_gpu:
            f01 = self.frames01[0].contiguous()
            mk = self.masks[0].contiguous()
            blend = ops.compose_blend_u8 if self.world > 1 else ops.compose_blend
            for wi in range(len(self.sched)):
                blend(outs[wi], self._nb[wi], self._first[wi], f01, mk, comp)
        else:
            visited = [False] * self.n
            for wi in range(len(self.sched)):
                compose_torch(outs[wi], self.sched[wi][0], self.frames01, self.masks, comp, visited)
        self._t("blend")
        return comp

    def _exchange(self, outs):
        """One all-gather of the per-window outputs, truncated to uint8 on the producing rank (tool/video_inpainting.py:731: the
        compose starts with astype(uint8), so the composite is unchanged and the exchange carries 1 byte per value instead of 4);
        send / receive buffers are allocated once.  Every rank ends up with every window: {window: uint8 [n_nb,3,H,W]}."""
        slots = max(len(a) for a in self.assign)
        if self._xchg is None:
            self._xchg = (torch.zeros(slots, self.max_nb, 3, self.H, self.W, dtype=torch.uint8, device=self.dev),
                          torch.empty(self.world * slots, self.max_nb, 3, self.H, self.W, dtype=torch.uint8, device=self.dev))
        send, recv = self._xchg
        for slot, wi in enumerate(self.mine):
            o = outs[wi]
            if self.on_gpu:
                ops.quantize_u8(o, out=send[slot, : o.shape[0]])
            else:
                send[slot, : o.shape[0]] = (((o + 1) / 2) * 255).to(torch.uint8)
        all_gather(recv, send, self.group).wait()
        full = {}
        for r in range(self.world):
            for slot, wi in enumerate(self.assign[r]):
                full[wi] = recv[r * slots + slot, : len(self.sched[wi][0])]
        return full
