"""Host-side handle on the native engine: owns an afx_handle, feeds it weights by
their reference checkpoint names and runs forwards on the current HIP stream.
PyTorch is used for device memory and streams only."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import AfxError, Config, check, lib, ptr, stream_ptr

ARCHS = {"ssl": _lib.ARCH_SSL, "xlsr_aasist": _lib.ARCH_XLSR_AASIST, "conformer": _lib.ARCH_CONFORMER,
         "conformer_head": _lib.ARCH_CONFORMER_HEAD}  # conformer_head: MyConformer alone (no trunk)
DTYPES = {"bf16": _lib.DT_BF16, "fp16": _lib.DT_FP16, "fp32": _lib.DT_FP32, "fp16x3": _lib.DT_FP16X3}
# fairseq extractor_mode: "layer_norm" = XLS-R (what the reference loads), "group_norm" = wav2vec2-base ("default")
EXTRACTORS = {"layer_norm": 0, "group_norm": 1, "default": 1}
# fp16 and bf16 run at the same matrix-core rate on gfx950; fp16's 3 extra mantissa bits
# are what keeps the scores within 1e-3 of the fp32 reference (DESIGN.md "Numerics").
# "fp32" is the exact mode: fp32 operands on the fp32 matrix instruction, 1/16 of the rate,
# no reduced-precision rounding anywhere -- for parity work, not for throughput.
# "fp16x3" is split precision: fp32 activations as in exact mode, every dense product as three fp16 matrix-core products
# of hi / lo operand pairs (~22 significant bits) -- the unconditional-parity mode at about a third of the fp16 rate.
DEFAULT_DTYPE = "fp16"  # (no environment override: a stray variable must not change what a product run computes)


def torch_dtype(name):
    return {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32, "fp16x3": torch.float32}[name]


def _cuda_device(device):
    """'cuda' / 'cuda:1' / torch.device / an int rank (main.py:48) / None -> a torch.device with an index."""
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    d = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    if d.type != "cuda":
        raise AfxError(f"the MI355X-native path runs on a GPU, not on {d}")
    return d if d.index is not None else torch.device("cuda", torch.cuda.current_device())


_SIDE_STREAMS = {}


def side_stream(device, role="head"):
    """THE side stream of a GPU (one per device and process, shared by its engines): where ``forward_overlapped`` runs the
    back-ends.  Normal priority; created and USED once, here, so that it owns its hardware queue from then on.  ROCm maps the
    streams of a process onto 4 hardware queues (GPU_MAX_HW_QUEUES) in order of first use and shares them from the fifth
    stream on: a side stream first used AFTER RCCL had brought up its streams landed on the queue of torch's default stream
    -- same scores, the overlap silently lost (tools/diag_dist_overlap.py).  A host that uses a process group calls this
    (or builds its engine and runs one forward_overlapped) BEFORE ``init_process_group``.  Raising the queue count is NOT
    the answer: with more than 4 hardware queues in use (GPU_MAX_HW_QUEUES=8 next to RCCL, or one high-priority stream,
    which brings queues of its own) the two-stream step measured 2x SLOWER than the one-stream step -- 10.1-10.8 ms
    instead of 4.75 for the teacher (profiles/r03_k_dist_overlap_hw_queues.txt).
    role "copy": the second (and last) such stream of a GPU, for the scoring loop's H2D prefetch (afx.harness.prefetch_to_device)
    -- cached for the same reason: a fresh stream per scoring pass would run the process into shared queues after a few passes."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), role)
    s = _SIDE_STREAMS.get(key)
    if s is None:
        with torch.cuda.device(key[0]):
            s = torch.cuda.Stream(device=key[0])
            ev = torch.cuda.Event()
            ev.record(s)  # first use: the stream acquires its hardware queue now
            ev.synchronize()
        _SIDE_STREAMS[key] = s
    return s


class Engine:
    """One native engine = one GPU.  The handle's packed weights, the workspace and every launch live on
    ``self.device`` (fixed at construction: the given device, else the current one) whatever torch's current
    device is when a method is called -- the reference passes ``device=rank`` and never calls
    ``torch.cuda.set_device`` (main.py:48,78-82).  Inputs on another GPU are refused, not dereferenced."""

    def __init__(self, arch, n_layers=24, dtype=None, conf_emb=144, conf_heads=4, conf_kernel=31,
                 conf_blocks=4, pre_emphasis=False, pre_emphasis_coef=0.97, device=None, extractor_mode="layer_norm"):
        dtype = dtype or DEFAULT_DTYPE
        if dtype not in DTYPES:
            raise ValueError(f"dtype must be one of {sorted(DTYPES)}, got {dtype!r}")
        if not torch.cuda.is_available():
            raise AfxError("no HIP device: the MI355X-native path has no CPU fallback")
        self.arch, self.dtype, self.n_layers = arch, dtype, n_layers
        self.pre_emphasis = bool(pre_emphasis)
        if extractor_mode not in EXTRACTORS:
            raise ValueError(f"extractor_mode must be one of {sorted(EXTRACTORS)}, got {extractor_mode!r}")
        self.extractor_mode = "group_norm" if EXTRACTORS[extractor_mode] else "layer_norm"
        self.device = _cuda_device(device)
        cfg = Config(ARCHS[arch], DTYPES[dtype], n_layers, conf_emb, conf_heads, conf_kernel, conf_blocks,
                     1 if pre_emphasis else 0, pre_emphasis_coef, EXTRACTORS[extractor_mode])
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):  # afx_create allocates on the current device
            check(lib().afx_create(C.byref(cfg), C.byref(self._h)))
        self._ws = None
        self._taps = False

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and lib is not None:  # module globals are gone at interpreter exit
            try:
                ev = getattr(self, "_last_head", None)
                if ev is not None:
                    ev.synchronize()  # a back-end still on the side stream reads the weights afx_destroy frees
                lib().afx_destroy(h)
            except Exception:
                pass
            self._h = None

    def _stream(self):
        return stream_ptr(self.device)

    # ---- weights -------------------------------------------------------------------
    def load_state_dict(self, sd):
        """sd: name -> tensor (reference checkpoint names).  Tensors are moved to
        the device as contiguous fp32 and copied / repacked by the library."""
        l, dev = lib(), self.device
        self.join()  # (a back-end still running on the side stream reads the weights this call repacks)
        with torch.cuda.device(dev):
            torch.cuda.current_stream(dev).synchronize()
            s = self._stream()
            for name, t in sd.items():
                if not torch.is_tensor(t) or not t.dtype.is_floating_point:
                    continue
                t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
                shape = (C.c_int64 * max(t.ndim, 1))(*t.shape)
                check(l.afx_load_weight(self._h, name.encode(), ptr(t), shape, t.ndim, s))
            torch.cuda.current_stream(dev).synchronize()  # sources may be freed by the caller
            check(l.afx_finalize(self._h, s))

    # ---- forward -------------------------------------------------------------------
    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def _on_device(self, x, what):
        if not x.is_cuda:
            raise AfxError(f"{what} must live on the GPU (the caller does batch_x.to(device), main.py:209)")
        if x.device != self.device:
            raise AfxError(f"{what} is on {x.device} but this engine (weights, workspace, launches) lives on {self.device}")

    def _wave(self, x):
        if x.ndim == 3:  # (B,L,1): models/fe.py:18 uses channel 0
            x = x[:, :, 0]
        if x.ndim != 2:
            raise ValueError(f"expected a (B,L) or (B,L,1) waveform batch, got shape {tuple(x.shape)}")
        self._on_device(x, "input")
        return x.to(torch.float32).contiguous()

    def forward(self, wave):
        x = self._wave(wave)
        B, L = x.shape
        l = lib()
        with torch.cuda.device(self.device):
            ws = self._workspace(l.afx_workspace_bytes(self._h, B, L))
            out = torch.empty(B, 2, dtype=torch.float32, device=self.device)
            check(l.afx_forward(self._h, ptr(x), B, L, ptr(out), ptr(ws), ws.numel(), self._stream()))
        return out

    # ---- scoring-loop form: the back-end of batch i on a side stream under the trunk of batch i+1 -------------------
    def forward_overlapped(self, wave):
        """``forward(wave)`` split over two streams: the trunk on torch's current stream, the back-end on the engine's side
        stream, two workspaces alternating -- the head of this batch runs while the NEXT call's trunk does (the AASIST
        head is a tenth of the teacher's time on a third of the chip's CUs).  Same kernels and bits as ``forward``.
        The returned logits are produced on the SIDE stream: call ``join()`` before reading them on the current stream
        (afx.harness.produce_evaluation_file does, once, after its last batch)."""
        if self._taps:  # taps are engine-owned copies written by whichever forward runs: one stream only
            return self.forward(wave)
        if not self.overlap_is_bit_stable:
            return self.forward(wave)
        if getattr(self, "_issue", None) == "lanes":  # (``overlap_pays`` / the caller chose the other two-stream form)
            return self.forward_lanes(wave)
        x = self._wave(wave)
        B, L = x.shape
        l = lib()
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            if getattr(self, "_side", None) is None:
                self._side = side_stream(self.device)
                self._ov = [dict(ws=None, head_done=None), dict(ws=None, head_done=None)]
                self._ov_i = 0
            slot = self._ov[self._ov_i]
            self._ov_i ^= 1
            nbytes = l.afx_workspace_bytes(self._h, B, L)
            if slot["ws"] is None or slot["ws"].numel() < nbytes:
                if slot["head_done"] is not None:
                    slot["head_done"].synchronize()
                slot["ws"] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
                # the head reads this block on the side stream: should the slot be dropped with a head still in flight (an
                # exception in the caller's loop, an engine deleted before join()), the caching allocator must not hand the
                # block to a current-stream op before the side stream is done with it
                slot["ws"].record_stream(self._side)
            elif slot["head_done"] is not None:
                cur.wait_event(slot["head_done"])  # this workspace's previous head (two calls ago) must be done with it
            ws = slot["ws"]
            check(l.afx_trunk_forward(self._h, ptr(x), B, L, ptr(ws), ws.numel(), C.c_void_p(cur.cuda_stream)))
            trunk_done = torch.cuda.Event()
            trunk_done.record(cur)
            out = torch.empty(B, 2, dtype=torch.float32, device=self.device)
            self._side.wait_event(trunk_done)
            out.record_stream(self._side)
            check(l.afx_head_from_workspace(self._h, B, L, ptr(out), ptr(ws), ws.numel(), C.c_void_p(self._side.cuda_stream)))
            slot["head_done"] = torch.cuda.Event()
            slot["head_done"].record(self._side)
            self._last_head = slot["head_done"]
            self._last_stream = self._side
        return out

    def forward_lanes(self, wave):
        """The other way of keeping two batches in flight (round 4, late): WHOLE forwards of consecutive calls on alternating streams --
        torch's current stream and the engine's side stream, a workspace each -- instead of one batch's back-end beside the next
        batch's trunk.  The teacher's small-M products fill 200 of the 256 CUs and its LayerNorms / attention far fewer; a second
        batch's kernels take what is free: 4.50 -> 3.94 ms per batch of 16 (4 059 utt/s), in fp16x3 8.24 -> 7.64 (2 093); the
        student 4.88 -> 4.65 (tools/diag_two_lanes.py).  Same kernels on the same data, each batch on one stream: the one-stream bits.
        A call on the side stream waits for what the current stream holds at that moment (its input is ready; the previous call's
        forward there has then ended, which is when the next current-stream forward starts: the two lanes stay busy together).
        Logits of a side-lane call must not be read on the current stream before ``join()``."""
        if self._taps:
            return self.forward(wave)
        x = self._wave(wave)
        B, L = x.shape
        l = lib()
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            if getattr(self, "_side", None) is None:
                self._side = side_stream(self.device)
                self._ov = [dict(ws=None, head_done=None), dict(ws=None, head_done=None)]
                self._ov_i = 0
            if getattr(self, "_ln", None) is None:
                self._ln = [dict(ws=None, done=None), dict(ws=None, done=None)]
                self._ln_i = 0
            k = self._ln_i
            self._ln_i ^= 1
            slot = self._ln[k]
            nbytes = l.afx_workspace_bytes(self._h, B, L)
            if slot["ws"] is None or slot["ws"].numel() < nbytes:
                if slot["done"] is not None:
                    slot["done"].synchronize()
                slot["ws"] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
                if k == 1:
                    slot["ws"].record_stream(self._side)
            ws = slot["ws"]
            out = torch.empty(B, 2, dtype=torch.float32, device=self.device)
            if k == 0:
                check(l.afx_forward(self._h, ptr(x), B, L, ptr(out), ptr(ws), ws.numel(), C.c_void_p(cur.cuda_stream)))
                self._last_stream = cur
            else:
                ready = torch.cuda.Event()
                ready.record(cur)
                self._side.wait_event(ready)
                x.record_stream(self._side)
                out.record_stream(self._side)
                check(l.afx_forward(self._h, ptr(x), B, L, ptr(out), ptr(ws), ws.numel(), C.c_void_p(self._side.cuda_stream)))
                slot["done"] = torch.cuda.Event()
                slot["done"].record(self._side)
                self._last_head = slot["done"]
                self._last_stream = self._side
        return out

    def set_issue(self, form):
        """Which two-stream form ``forward_overlapped`` issues: "overlap" (the back-end beside the next trunk) or "lanes" (whole
        forwards on alternating streams).  ``overlap_pays`` sets it from a timing; bench.py sets it for each form it probes."""
        if form not in ("overlap", "lanes"):
            raise ValueError("issue form: 'overlap' or 'lanes'")
        self.join()
        self._issue = form

    @property
    def last_stream(self):
        """The stream the logits of the last ``forward_overlapped`` / ``forward_lanes`` call are produced on (a caller that queues work
        of its own behind them -- bench.py's score all-gather -- puts it there and calls ``mark_side()``)."""
        st = getattr(self, "_last_stream", None)
        return st if st is not None else getattr(self, "_side", None)

    @property
    def overlap_is_bit_stable(self):
        """Whether ``forward_overlapped`` reproduces ``forward``'s bits for this engine (False: it runs on one stream).
        History (round 4): the teacher (AASIST back-end) in dtype "fp16x3" did NOT -- with the back-end of batch i beside it,
        the trunk of batch i+1 computed ONE frame of conv layer 0 differently in about a third of the batches.  Bisected
        (profiles/r04_two_stream_race.txt): the victim was the VALU ``conv0_kernel<F32T>``'s packed fp32
        tap loop, the trigger any fp16 GEMM kernel started beside it (the vendor library's as well), the same kernel with a
        scalar loop is immune.  That kernel left the fp16x3 path (conv layer 0 runs on the matrix cores there), the tap loops
        of the VALU conv-layer-0 kernels are scalar, and tests/test_gpu_aasist.py compares 42 concurrent batches per
        (head, precision) combination plus the VALU-kernel engines beside vendor GEMMs on every run.  Every combination is
        therefore True; the property stays as the switch a future finding flips."""
        return True

    def join(self):
        """Make torch's current stream wait for every back-end forward_overlapped has put on the side stream (and for
        whatever the caller queued behind it there and marked with ``mark_side``)."""
        ev = getattr(self, "_last_head", None)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)

    def overlap_pays(self, wave, steps=3):
        """Does the two-stream step beat the one-stream step HERE?  It rests on how ROCm maps this process's streams onto
        hardware queues (``side_stream``): next to RCCL's streams at world size > 1 the side stream may share the trunk's
        queue (overlap lost) or the process may use more than 4 queues (measured 2x slower).  Timed once per engine on the
        caller's batch (``steps`` forwards each way, device-synchronised) and cached; scoring loops that run under a process
        group ask before they issue (afx.harness.produce_evaluation_file_distributed, bench.py does its own probe)."""
        if getattr(self, "_overlap_pays", None) is None:
            dev = self.device
            def run(fn):
                fn(wave)  # warm (workspaces, attribute calls)
                self.join()
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.device(dev):
                    e0.record(torch.cuda.current_stream(dev))
                    for _ in range(steps):
                        fn(wave)
                    self.join()
                    e1.record(torch.cuda.current_stream(dev))
                torch.cuda.synchronize(dev)
                return e0.elapsed_time(e1)
            self._issue = "overlap"
            two = run(self.forward_overlapped)
            lanes = run(self.forward_lanes) if self.overlap_is_bit_stable and not self._taps else float("inf")
            one = run(self.forward)
            self._overlap_probe = {"one_stream_ms": one / steps, "two_stream_ms": two / steps, "two_lanes_ms": lanes / steps}
            # which two-stream form ``forward_overlapped`` issues from now on, and whether either beats one stream
            self._issue = "lanes" if lanes < two else "overlap"
            self._overlap_pays = min(two, lanes) <= one
        return self._overlap_pays

    def mark_side(self):
        """The caller put work of its own on the side stream behind the last back-end (bench.py: the RCCL score all-gather
        of a step): ``join()`` waits for it too from now on."""
        st = self.last_stream
        if st is not None:
            ev = torch.cuda.Event()
            ev.record(st)
            self._last_head = ev

    def check_finite(self):
        """Raise ``AfxError`` if any forward since the last check produced non-finite trunk features or logits (an operand
        copy left the range of this engine's precision: afx_check_finite).  Synchronises torch's current stream (after
        joining the side stream); the scoring loops call it once, before they write a score."""
        self.join()
        with torch.cuda.device(self.device):
            check(lib().afx_check_finite(self._h, self._stream()))

    def ssl(self, wave):
        x = self._wave(wave)
        B, L = x.shape
        l = lib()
        T = l.afx_num_frames(L)
        with torch.cuda.device(self.device):
            ws = self._workspace(l.afx_workspace_bytes(self._h, B, L))
            buf = torch.empty(B, max(T, 1), 1024, dtype=torch.float32, device=self.device)
            check(l.afx_ssl_forward(self._h, ptr(x), B, L, ptr(buf), ptr(ws), ws.numel(), self._stream()))
        return buf[:, :max(T, 0)]

    def head(self, feats):
        self._on_device(feats, "SSL features")
        f = feats.to(torch.float32).contiguous()
        B, T, D = f.shape
        if D != 1024:
            raise ValueError("SSL features must have 1024 channels")
        l = lib()
        with torch.cuda.device(self.device):
            ws = self._workspace(l.afx_head_workspace_bytes(self._h, B, T))
            out = torch.empty(B, 2, dtype=torch.float32, device=self.device)
            check(l.afx_head_forward(self._h, ptr(f), B, T, ptr(out), ptr(ws), ws.numel(), self._stream()))
        return out

    # ---- ragged batches: clips of different lengths, each scored as if alone ----------------
    def _pack_ragged(self, clips):
        lens = [int(c.numel()) for c in clips]
        if not lens or min(lens) < 400:
            raise ValueError("every clip needs at least 400 samples (one SSL frame)")
        batch = torch.zeros(len(clips), max(lens), dtype=torch.float32, device=self.device)
        for b, c in enumerate(clips):  # plumbing: zero-padded rows
            batch[b, : lens[b]] = c.reshape(-1).to(device=self.device, dtype=torch.float32)
        return batch, (C.c_int * len(lens))(*lens), lens

    def forward_ragged(self, clips):
        """clips: list of 1-D waveforms of ANY lengths (>= 400 samples) -> logits (B,2); clip b's row equals
        ``forward(clips[b][None])`` (key-padding masks, afx_forward_ragged)."""
        batch, n, _ = self._pack_ragged(clips)
        B, Lmax = batch.shape
        l = lib()
        with torch.cuda.device(self.device):
            ws = self._workspace(l.afx_ragged_workspace_bytes(self._h, B, Lmax))
            out = torch.empty(B, 2, dtype=torch.float32, device=self.device)
            check(l.afx_forward_ragged(self._h, ptr(batch), B, Lmax, n, ptr(out), ptr(ws), ws.numel(), self._stream()))
        return out

    def ssl_ragged(self, clips):
        """-> (feats (B,Tmax,1024) with rows past each clip's frames zeroed, list of frame counts)."""
        batch, n, _ = self._pack_ragged(clips)
        B, Lmax = batch.shape
        l = lib()
        T = l.afx_num_frames(Lmax)
        frames = (C.c_int * B)()
        with torch.cuda.device(self.device):
            ws = self._workspace(l.afx_ragged_workspace_bytes(self._h, B, Lmax))
            buf = torch.empty(B, T, 1024, dtype=torch.float32, device=self.device)
            check(l.afx_ssl_forward_ragged(self._h, ptr(batch), B, Lmax, n, ptr(buf), frames, ptr(ws), ws.numel(), self._stream()))
        return buf, list(frames)

    def tail(self, conv5):
        """conv5: (B,T5,512) output of conv layer 5 in the operand type -> logits (B,2) (afx_tail_forward).  A view into a
        longer per-stream buffer (rows contiguous, any batch stride that is a multiple of 8 elements) is read in place."""
        self._on_device(conv5, "conv-layer-5 activations")
        if conv5.dtype != torch_dtype(self.dtype) or conv5.ndim != 3 or conv5.shape[2] != 512:
            raise ValueError(f"expected a (B,T5,512) {self.dtype} tensor, got {tuple(conv5.shape)} {conv5.dtype}")
        B, T5 = conv5.shape[0], conv5.shape[1]
        c = conv5
        if not (c.stride(2) == 1 and c.stride(1) == 512 and (B == 1 or (c.stride(0) >= T5 * 512 and c.stride(0) % 8 == 0))):
            c = conv5.contiguous()
        l = lib()
        with torch.cuda.device(self.device):
            ws = self._workspace(l.afx_tail_workspace_bytes(self._h, B, T5))
            out = torch.empty(B, 2, dtype=torch.float32, device=self.device)
            check(l.afx_tail_forward_strided(self._h, ptr(c), c.stride(0) if B > 1 else 0, B, T5, ptr(out), ptr(ws), ws.numel(),
                                             self._stream()))
        return out

    def kv_state(self, n_streams):
        """Per-stream state of the KV-cached streaming mode (afx_kv_create): see ``KVState``."""
        return KVState(self, n_streams)

    def conformer(self, tokens):
        """MyConformer.forward (models/conformer_baseline.py:22-29): tokens (B,T,emb) fp32 -> (logits (B,2), embedding (B,emb))."""
        self._on_device(tokens, "tokens")
        t = tokens.to(torch.float32).contiguous()
        if t.ndim != 3:
            raise ValueError(f"expected (B,T,emb) tokens, got shape {tuple(t.shape)}")
        B, T, E = t.shape
        l = lib()
        with torch.cuda.device(self.device):
            ws = self._workspace(l.afx_head_workspace_bytes(self._h, B, T))
            out = torch.empty(B, 2, dtype=torch.float32, device=self.device)
            emb = torch.empty(B, E, dtype=torch.float32, device=self.device)
            check(l.afx_conformer_forward(self._h, ptr(t), B, T, ptr(out), ptr(emb), ptr(ws), ws.numel(), self._stream()))
        return out, emb

    def set(self, key, value):
        """Per-engine switch between two forms of the same op (afx_engine_set): posconv_sliding, conf_attn_mfma,
        fuse_conformer, fuse_conv_ln."""
        check(lib().afx_engine_set(self._h, key.encode(), int(value)))

    # ---- hipGraph replay: the ~130 launches of a forward as ONE graph launch ----------
    def capture(self, B, L):
        """Capture forward() for a fixed (B, L) into a hipGraph (the engine allocates nothing
        and never synchronises inside a forward, so the whole launch sequence is capturable).
        Returns ``run(wave) -> logits``: copies the batch into the graph's static input and
        replays.  Meant for small, latency-bound batches (B = 1 streaming-style calls) on a host that cannot keep up
        with the launches; on the pool's hosts eager and replay measure the same at B = 1 (1.18 ms student, 2.50 ms
        teacher, tools/diag_b1_latency.py): that time is the GPU-side latency of a chain of ~130 / ~330 dependent kernels."""
        with torch.cuda.device(self.device):
            static_in = torch.zeros(B, L, dtype=torch.float32, device=self.device)
            self._workspace(lib().afx_workspace_bytes(self._h, B, L))
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):  # warm-up outside the capture (one-off attribute calls, allocations)
                self.forward(static_in)
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self.forward(static_in)

        def run(wave):
            static_in.copy_(self._wave(wave), non_blocking=True)
            graph.replay()
            return static_out
        run.graph = graph
        return run

    # ---- per-kernel-class timing (bench.py roofline leg) -----------------------------
    def profile_begin(self):
        check(lib().afx_profile_begin(self._h))

    def profile_end(self):
        """-> {class name: dict(ms=..., flops=..., launches=...)} summed since profile_begin."""
        l = lib()
        n = l.afx_profile_num_classes()
        ms, fl, la = (C.c_double * n)(), (C.c_double * n)(), (C.c_longlong * n)()
        check(l.afx_profile_end(self._h, n, ms, fl, la))
        return {l.afx_profile_class_name(i).decode(): dict(ms=ms[i], flops=fl[i], launches=la[i]) for i in range(n)}

    # ---- debug taps ----------------------------------------------------------------
    def enable_taps(self, on=True):
        self.join()  # (a back-end still running on the side stream finishes before the tap state changes)
        check(lib().afx_enable_taps(self._h, 1 if on else 0))
        self._taps = bool(on)

    def tap(self, name):
        n = C.c_size_t(0)
        with torch.cuda.device(self.device):
            check(lib().afx_tap(self._h, name.encode(), None, 0, C.byref(n), self._stream()))
            out = torch.empty(n.value, dtype=torch.float32, device=self.device)
            check(lib().afx_tap(self._h, name.encode(), ptr(out), n.value, C.byref(n), self._stream()))
        return out


class KVState:
    """State of ``n_streams`` lock-stepped streams in the KV-cached streaming mode (BASELINE config 5 as named; a labelled,
    NON-reference mode -- include/afx.h, oracle/streaming.py): the K / V rings of every transformer layer, the positional
    conv's left context and the feature window live in the library object; ``step`` consumes the new conv-layer-6 frames
    of every stream and returns the back-end's logits on the window that ends with this chunk."""

    def __init__(self, engine, n_streams):
        self.engine, self.S = engine, int(n_streams)
        self._k = C.c_void_p()
        with torch.cuda.device(engine.device):
            check(lib().afx_kv_create(engine._h, self.S, C.byref(self._k)))
        self.state_bytes = lib().afx_kv_state_bytes(self._k)

    def __del__(self):
        k = getattr(self, "_k", None)
        if k is not None and k.value and lib is not None:
            try:
                lib().afx_kv_destroy(k)
            except Exception:
                pass
            self._k = None

    def step(self, feats6):
        """feats6: (S, n, 512) fp32, the n (1..16) NEW frames of conv layer 6 -> logits (S, 2)."""
        eng = self.engine
        eng._on_device(feats6, "conv-layer-6 frames")
        f = feats6.to(torch.float32).contiguous()
        if f.ndim != 3 or f.shape[0] != self.S or f.shape[2] != 512:
            raise ValueError(f"expected ({self.S}, n, 512) frames, got {tuple(f.shape)}")
        n = f.shape[1]
        l = lib()
        with torch.cuda.device(eng.device):
            ws = eng._workspace(max(l.afx_kv_workspace_bytes(self._k, n), 256))
            out = torch.empty(self.S, 2, dtype=torch.float32, device=eng.device)
            check(l.afx_kv_step(self._k, ptr(f), n, ptr(out), ptr(ws), ws.numel(), eng._stream()))
        return out

