"""Thin typed wrappers over the kernel-level C ABI (include/bsed.h).  No math happens here: every
function fills a descriptor with device pointers / shapes and launches HIP kernels on the current
stream.  Activations are NHWC fp32 tensors."""
import ctypes

import torch

from . import _lib as L

EPI_PLAIN, EPI_STATS, EPI_GLU_POOL, EPI_GLU_BWD, EPI_ADD_STATS2 = range(5)

_fp = ctypes.c_void_p
_i = ctypes.c_int


class IgemmDesc(ctypes.Structure):
    _fields_ = ([(n, _fp) for n in ("in_", "w", "bias", "out", "out2", "stats", "a_scale", "a_shift", "e_src",
                                    "e_scale", "e_shift", "e_dpool")]
                + [(n, _i) for n in ("in_pitch", "out_pitch", "e_pitch", "NB", "H", "W", "CIN", "N", "NP", "TH", "TW",
                                     "tilesH", "tilesW", "hh", "hw", "ntaps")]
                + [("dh", _i * 9), ("dw", _i * 9)]
                + [(n, _i) for n in ("ph", "pw", "Hp", "Wp", "epilogue")]
                + [("drop_p", ctypes.c_float), ("rng_stream", ctypes.c_uint32), ("seed", ctypes.c_uint64)]
                + [("valid_h", _i), ("valid_w", _i), ("act_bf16", _i)])


class WgradDesc(ctypes.Structure):
    _fields_ = ([(n, _fp) for n in ("in_", "dy", "part", "a_scale", "a_shift")]
                + [(n, _i) for n in ("in_pitch", "dy_pitch", "NB", "H", "W", "CIN", "CINP", "N", "NP", "G", "TH", "TW",
                                     "tilesH", "tilesW", "hh", "hw", "ntaps")]
                + [("dh", _i * 9), ("dw", _i * 9)]
                + [(n, _fp) for n in ("bn_y", "bn_coef", "bn_mean", "dy_out")] + [("act_bf16", _i)])


class HeadBwdDesc(ctypes.Structure):
    _fields_ = ([(n, _fp) for n in ("x", "w", "strong", "sof_raw", "weak", "den", "y_strong", "y_weak", "ema_strong",
                                    "ema_weak", "ema_strong2", "g_strong_ext", "g_weak_ext")]
                + [(n, ctypes.c_float) for n in ("w_strong", "w_weak", "w_cons_s", "w_cons_w", "w_cons_s2",
                                                 "inv_n_strong", "inv_n_weak")]
                + [(n, _fp) for n in ("dx", "dw_part", "db_part", "loss_part")]
                + [(n, _i) for n in ("B", "T", "K", "C", "attention")])


TAPS3x3 = [(kh - 1, kw - 1) for kh in range(3) for kw in range(3)]


class KernelTimer:
    """Per-launch HIP-event timing of EVERY entry-point call of a step (bench.py's roofline leg).  Events are recorded
    on the stream the kernels are launched on (torch's current stream).  key = (kernel label, shape tag): the label is
    the kernel's name as rocprofv3 prints it (template instance included) so that bench.py's table and the committed
    rocprof summary can be joined by name."""

    def __init__(self, main_stream=()):
        self.records = {}  # key -> [flops_per_launch, [(start, end), ...], algorithmic_bytes_per_launch]
        self._last, self._chain = {}, False
        # the step's own stream = the stream that is current when the timer is made (the null stream reads as None).
        # (Taking the stream of the first LAUNCH instead made the mean-teacher step's table name the teacher's stream
        # "main": its forward is the first thing a step enqueues.)
        # main_stream: the raw handle (``torch.cuda.Stream.cuda_stream``) of the stream the step runs on when that is not
        # the caller's (SEDTrainer runs the mean-teacher step on a high-priority stream of its own).
        if main_stream != ():
            self._main = main_stream or None
        else:
            try:
                self._main = L.stream().value
            except Exception:
                self._main = ()    # no GPU yet: fall back to the stream of the first launch
        self.side = {}     # key -> [flops, [(start, end), ...], bytes] of launches on any other stream: they run
        #                    beside main-stream kernels, so their event pairs time contention (and queueing) as well

    @staticmethod
    def prime(n=4096):
        """Create, record and retire n timing events once: the runtime grows its signal pool in steps that stall the
        stream for tens of milliseconds the first time a process holds a few hundred events (measured ~45 ms on a
        fresh MI355X box); doing it here keeps that out of whatever is timed later."""
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        for e in evs:
            e.record()
        torch.cuda.synchronize()
        del evs

    def launch(self, key, flops, fn, nbytes):
        """One event per launch: on an in-order stream that the host keeps fed, the event recorded after launch i is also
        the start of launch i+1 (half the events of a start/end pair per launch: each event costs the GPU ~1.5 us).  A
        launch that follows anything other than a timed launch (the first of a step: host-side work in between) gets
        its own start event."""
        sid = L.stream().value
        if self._main == ():
            self._main = sid
        if sid != self._main:
            # a side-stream launch (GRU weight gradients beside the next recurrence, the next batch's mel transform
            # beside the GRU forward) shares the chip with main-stream kernels: its own event pair, kept apart from
            # the main-stream table (summary(side=True))
            s = torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.side.setdefault(key, [flops, [], nbytes])[1].append((s, e))
            return
        s = self._last.get(sid) if self._chain else None
        if s is None:
            s = torch.cuda.Event(enable_timing=True)
            s.record()
        fn()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self._last[sid], self._chain = e, True
        rec = self.records.setdefault(key, [flops, [], nbytes])
        rec[1].append((s, e))

    def break_chain(self):
        """call between steps (or around host-side work): the next launch records its own start event"""
        self._chain = False

    def summary(self, side=False):
        """{key: (launches, total_ms, avg_ms, flops_per_launch, bytes_per_launch)} -- call after a device sync"""
        out = {}
        for key, (flops, evs, nbytes) in (self.side if side else self.records).items():
            ms = [s.elapsed_time(e) for s, e in evs]
            out[key] = (len(ms), float(sum(ms)), float(sum(ms) / len(ms)), flops, nbytes)
        return out


def set_timer(t):
    L.timer = t


def _launch(key, flops, fn, nbytes=None):
    """Announce the algorithmic work of the entry-point call ``fn`` makes.  key = (kernel, taps, CIN, N, H, W) for the
    contraction kernels or (kernel, tag) for the others.  nbytes = ALGORITHMIC HBM bytes of the launch (each activation
    tensor the op must read or write, once; weights and partial slabs not counted); default: a (positions, CIN) input
    and a (positions, N) output in fp32, positions recovered from the FLOP count."""
    if L.timer is not None:
        if nbytes is None:
            _, taps, cin, n, _, _ = key
            nbytes = 4.0 * (flops / (2.0 * taps * cin * n)) * (cin + n)
        if len(key) == 6:
            key = (key[0], "t%d CIN%d N%d %dx%d" % tuple(key[1:]))
        L.pending = (key, float(flops), float(nbytes))
    fn()


def _note(kernel, tag, flops, nbytes):
    """announce the next L.call (for wrappers whose call is written inline)"""
    if L.timer is not None:
        L.pending = ((kernel, tag), float(flops), float(nbytes))


def round_up(v, m):
    return (v + m - 1) // m * m


def tile_for(W):
    tw = min(W, 16)
    return 128 // tw, tw


def _p(t):
    return L.ptr(t).value if t is not None else None


def _pa(t):
    """pointer of an ACTIVATION tensor: fp32, or bf16 in the "bf16" throughput mode (the C ABI keeps float* types)"""
    return L.ptr(t, t.dtype if t is not None and t.dtype == torch.bfloat16 else None)


def _abf(*ts):
    """1 if the activation tensors are bf16 (all of them), 0 if fp32"""
    ts = [t for t in ts if t is not None]
    bf = [t.dtype == torch.bfloat16 for t in ts]
    if any(bf) and not all(bf):
        raise L.BsedError("mixed fp32 / bf16 activation tensors: " + ", ".join(str(t.dtype) for t in ts))
    return 1 if bf and bf[0] else 0


def _esz(t):
    return 2.0 if t.dtype == torch.bfloat16 else 4.0


def _dp(t, offset=0):
    """device pointer (int) of tensor storage + element offset; tensor need not be contiguous as a whole"""
    return None if t is None else t.data_ptr() + t.element_size() * offset


def pack_weight(src, ntaps, K, N, s_tap, s_k, s_n, src_offset=0):
    NP = round_up(N, 32)
    dst = torch.empty((ntaps, K, NP), device=src.device, dtype=torch.float32)
    _note("pack_weight_kernel", "", 0.0, 4.0 * dst.numel())
    L.call("bsed_pack_weight", _fp(_dp(src, src_offset)), L.ptr(dst), _i(ntaps), _i(K), _i(N), _i(NP),
           ctypes.c_long(s_tap), ctypes.c_long(s_k), ctypes.c_long(s_n), L.stream())
    return dst


def igemm(inp, wpk, N, NB, H, W, CIN, taps=((0, 0),), bias=None, out=None, epilogue=EPI_PLAIN, in_pitch=None,
          out_pitch=None, a_scale=None, a_shift=None, e_src=None, e_scale=None, e_shift=None, e_dpool=None,
          out2=None, pool=(1, 1), drop_p=0.0, rng_stream=0, seed=0, in_offset=0, want_stats=False, valid=None):
    """Launch the implicit GEMM.  Returns (out, stats or None).  valid = (h, w): only output positions inside that
    extent are stored / enter the STATS sums (the rest of a freshly allocated ``out`` is zero)."""
    d = IgemmDesc()
    TH, TW = tile_for(W)
    NP = wpk.shape[2]
    ph, pw = pool
    Hp, Wp = H // ph, W // pw
    dev = inp.device
    if out is None:
        shape = (NB, Hp, Wp, N) if epilogue == EPI_GLU_POOL else (NB, H, W, N)
        out = (torch.zeros if valid else torch.empty)(shape, device=dev, dtype=torch.float32)
    if valid:
        d.valid_h, d.valid_w = valid
    ntiles = NB * ((H + TH - 1) // TH) * (W // TW)
    stats = None
    if epilogue in (EPI_STATS, EPI_GLU_BWD, EPI_ADD_STATS2):
        stats = torch.empty((ntiles, 2, N), device=dev, dtype=torch.float32)
    d.in_ = _dp(inp, in_offset); d.w = _p(wpk); d.bias = _p(bias); d.out = _p(out); d.out2 = _p(out2)
    d.stats = _p(stats); d.a_scale = _p(a_scale); d.a_shift = _p(a_shift); d.e_src = _p(e_src)
    d.e_scale = _p(e_scale); d.e_shift = _p(e_shift); d.e_dpool = _p(e_dpool)
    d.in_pitch = CIN if in_pitch is None else in_pitch
    d.out_pitch = N if out_pitch is None else out_pitch
    d.e_pitch = N
    d.NB, d.H, d.W, d.CIN, d.N, d.NP = NB, H, W, CIN, N, NP
    d.TH, d.TW = TH, TW
    d.hh = max(abs(t[0]) for t in taps); d.hw = max(abs(t[1]) for t in taps)
    d.ntaps = len(taps)
    for i, (a, b) in enumerate(taps):
        d.dh[i], d.dw[i] = a, b
    d.ph, d.pw, d.Hp, d.Wp = ph, pw, Hp, Wp
    d.epilogue = epilogue
    d.drop_p, d.rng_stream, d.seed = drop_p, rng_stream, seed
    kc = 32 if CIN % 32 == 0 else 16
    bn = 128 if NP % 128 == 0 else (64 if NP % 64 == 0 else 32)
    _launch((f"igemm_kernel<{kc}, {bn}, {epilogue}>", len(taps), CIN, N, H, W), 2.0 * NB * H * W * len(taps) * CIN * N,
            lambda: L.call("bsed_igemm", ctypes.byref(d), L.stream()))
    return out, stats


IGEMM3_RB = {"rb": 1}  # 2 = 256-position tiles where they fit (measured no faster than 1: kept for A-B measurements)


_pack_memo = None   # {(kind, data_ptr, args): packed tensor} while an ops.pack_cache() block is open
_pack_plan = None   # the PackPlan of the open outermost block, if it was given one


class BsedPackJob(ctypes.Structure):
    _fields_ = ([("src", _fp), ("dst", ctypes.c_void_p)]
                + [(n, _i) for n in ("kind", "ntaps", "K", "N", "NP")]
                + [(n, ctypes.c_long) for n in ("s_tap", "s_k", "s_n")])


PACK_MAX_JOBS = 32   # BSED_PACK_MAX_JOBS


class PackPlan:
    """Which packed weight copies a recurring ``pack_cache`` block asked for last time.  The next block given the same
    plan makes ALL of them in one launch on entry (``bsed_pack_weights_batch``) instead of one 5-7 us launch per weight
    on the forward's and backward's critical path (fourteen per CRNN train step).  Entries a block did not use are
    dropped when it closes; new requests are packed on the spot and join the plan.  Same kernels' element code: same
    bits as the unplanned path (tests/test_pack_plan_gpu.py)."""

    def __init__(self):
        self.entries = {}     # key -> (kind, src, ntaps, K, N, s_tap, s_k, s_n)
        self.used = set()

    @staticmethod
    def _numel(kind, ntaps, K, N):
        NP = round_up(N, 32)
        return ntaps * (K // 32) * NP * 64 if kind == 0 else (NP // 32) * ntaps * (K // 16) * 2 * 64 * 8

    def replay(self, memo):
        stream = L.stream()
        todo = [(k, v) for k, v in self.entries.items() if k[2] == stream.value]
        self.used = set()
        if not todo:
            return
        dev = todo[0][1][1].device
        sizes = [self._numel(v[0], v[2], v[3], v[4]) for _, v in todo]
        buf = torch.empty(sum(sizes), device=dev, dtype=torch.int16)
        _note("pack_weights_batch_kernel", "", 0.0, 4.0 * buf.numel())
        off = 0
        jobs = []
        for (key, (kind, src, ntaps, K, N, s_tap, s_k, s_n)), n in zip(todo, sizes):
            NP = round_up(N, 32)
            shape = (ntaps, K // 32, NP, 64) if kind == 0 else (NP // 32, ntaps, K // 16, 2, 64, 8)
            dst = buf[off:off + n].view(shape)
            off += n
            memo[key] = (dst, src)
            jobs.append((src.data_ptr(), dst.data_ptr(), kind, ntaps, K, N, NP, s_tap, s_k, s_n))
        for j0 in range(0, len(jobs), PACK_MAX_JOBS):
            part = jobs[j0:j0 + PACK_MAX_JOBS]
            arr = (BsedPackJob * len(part))()
            for a, (sp, dp, kind, ntaps, K, N, NP, s_tap, s_k, s_n) in zip(arr, part):
                a.src = sp; a.dst = dp
                a.kind, a.ntaps, a.K, a.N, a.NP = kind, ntaps, K, N, NP
                a.s_tap, a.s_k, a.s_n = s_tap, s_k, s_n
            L.call("bsed_pack_weights_batch", arr, _i(len(part)), stream)

    def close(self):
        stream = L.stream().value
        for k in [k for k in self.entries if k[2] == stream and k not in self.used]:
            del self.entries[k]


class pack_cache:
    """Inside this block the caller guarantees that no weight tensor changes (a train step before its optimizer
    update): a packed / split copy of a weight (pack_weight3, pack_weight3s) is made once per (tensor, layout) and
    reused by every forward / backward pass in the block -- two half-batch passes of a mean-teacher or adversarial step
    pack each weight once instead of twice.  With a ``PackPlan`` (a block that recurs, e.g. the train step) the copies
    the previous block used are all made in one launch on entry."""

    def __init__(self, plan=None):
        self._plan = plan

    def __enter__(self):
        global _pack_memo, _pack_plan
        self._outer = _pack_memo
        self._outer_plan = _pack_plan
        if _pack_memo is None:
            _pack_memo = {}
            _pack_plan = self._plan
            if _pack_plan is not None:
                _pack_plan.replay(_pack_memo)
        return self

    def __exit__(self, *exc):
        global _pack_memo, _pack_plan
        if self._outer is None and _pack_plan is not None:
            _pack_plan.close()
        _pack_memo = self._outer
        _pack_plan = self._outer_plan
        return False


def _pack_lookup(key):
    if _pack_memo is not None and key in _pack_memo:
        if _pack_plan is not None:
            _pack_plan.used.add(key)
        return _pack_memo[key][0]
    return None


def _pack_store(key, dst, src, kind, ntaps, K, N, s_tap, s_k, s_n):
    if _pack_memo is not None:
        _pack_memo[key] = (dst, src)   # src kept alive: its address is the key
        if _pack_plan is not None:
            _pack_plan.entries[key] = (kind, src, ntaps, K, N, s_tap, s_k, s_n)
            _pack_plan.used.add(key)


def igemm3_nsplit():
    """True: ops.igemm3 runs the round-3 kernel (csrc/igemm3n.hip: a wave owns 32 output channels, weight fragments
    straight from global memory, one barrier per 32-channel chunk) and pack_weight3 emits its fragment-order table;
    BSED_IGEMM3N=0 (or 256-position tiles, IGEMM3_RB) = the slab kernel of rounds 1-2 (csrc/igemm3.hip)."""
    import os
    return os.environ.get("BSED_IGEMM3N", "1") != "0" and IGEMM3_RB["rb"] != 2


def pack_weight3(src, ntaps, K, N, s_tap, s_k, s_n):
    """bf16 hi/lo split weights for igemm3: uint16 (ntaps, K/32, NP, 64), or -- for the N-split kernel -- the
    fragment-order table of pack_weight3s, int16 (NP/32, ntaps, K/16, 2, 64, 8)"""
    if igemm3_nsplit():
        return pack_weight3s(src, ntaps, N, s_tap, s_k, s_n, K=K)
    key = ("w3", src.data_ptr(), L.stream().value, ntaps, K, N, s_tap, s_k, s_n)
    hit = _pack_lookup(key)
    if hit is not None:
        return hit
    NP = round_up(N, 32)
    dst = torch.empty((ntaps, K // 32, NP, 64), device=src.device, dtype=torch.int16)
    _note("pack_weight3_kernel", "", 0.0, 4.0 * dst.numel())
    L.call("bsed_pack_weight3", _fp(_dp(src)), ctypes.c_void_p(dst.data_ptr()), _i(ntaps), _i(K), _i(N), _i(NP),
           ctypes.c_long(s_tap), ctypes.c_long(s_k), ctypes.c_long(s_n), L.stream())
    _pack_store(key, dst, src, 0, ntaps, K, N, s_tap, s_k, s_n)
    return dst


def pack_weight3s(src, ntaps, N, s_tap, s_k, s_n, K=16):
    """pre-split weights of a CIN = 16 / 32 convolution in fragment order: int16 (NP/32, ntaps, K/16, 2, 64, 8)"""
    key = ("w3s", src.data_ptr(), L.stream().value, ntaps, N, s_tap, s_k, s_n, K)
    hit = _pack_lookup(key)
    if hit is not None:
        return hit
    NP = round_up(N, 32)
    dst = torch.empty((NP // 32, ntaps, K // 16, 2, 64, 8), device=src.device, dtype=torch.int16)
    _note("pack_weight3s_kernel", "", 0.0, 4.0 * dst.numel())
    L.call("bsed_pack_weight3s", _fp(_dp(src)), ctypes.c_void_p(dst.data_ptr()), _i(ntaps), _i(K), _i(N), _i(NP),
           ctypes.c_long(s_tap), ctypes.c_long(s_k), ctypes.c_long(s_n), L.stream())
    _pack_store(key, dst, src, 1, ntaps, K, N, s_tap, s_k, s_n)
    return dst


def igemm3s_supported(W, CIN, ntaps=9):
    TH, TW = tile_for(W)
    halo = 1 if ntaps > 1 else 0
    return CIN in (16, 32) and (TH + 2 * halo) * (TW + 2 * halo) <= (256 if CIN == 16 else 192)


def igemm3s(inp, wtab, N, NB, H, W, taps, bias=None, epilogue=EPI_PLAIN):
    """CIN = 16 / 32 convolution on the bf16 cores (split-fp32), all taps' weights resident in LDS.
    Returns (out, stats (G,2,N) or None)."""
    d = IgemmDesc()
    TH, TW = tile_for(W)
    NP = wtab.shape[0] * 32
    dev = inp.device
    out = torch.empty((NB, H, W, N), device=dev, dtype=inp.dtype)
    ntiles = NB * ((H + TH - 1) // TH) * (W // TW)
    G = int(min(ntiles, L.lib().bsed_igemm3s_auto_g2(_i(inp.shape[-1]), _i(N))))
    stats = torch.empty((G, 2, N), device=dev, dtype=torch.float32) if epilogue == EPI_STATS else None
    d.act_bf16 = _abf(inp)
    d.in_ = _dp(inp); d.w = wtab.data_ptr(); d.bias = _p(bias); d.out = out.data_ptr(); d.stats = _p(stats)
    CIN = 16 * wtab.shape[2]
    d.in_pitch, d.out_pitch, d.e_pitch = CIN, N, N
    d.NB, d.H, d.W, d.CIN, d.N, d.NP = NB, H, W, CIN, N, NP
    d.TH, d.TW = TH, TW
    d.hh = max(abs(t[0]) for t in taps); d.hw = max(abs(t[1]) for t in taps)
    d.ntaps = len(taps)
    for i, (a, b) in enumerate(taps):
        d.dh[i], d.dw[i] = a, b
    d.ph = d.pw = 1; d.Hp, d.Wp = H, W
    d.epilogue = epilogue
    nv = 16 if N <= 16 and N % 4 == 0 else 32                      # transposed epilogue (bsed_igemm3s)
    _launch((f"igemm3s_kernel<{1 if epilogue == EPI_STATS else 0}, {len(taps)}, {CIN // 16}, {nv}, {d.act_bf16}>", len(taps), CIN, N, H, W),
            2.0 * NB * H * W * len(taps) * CIN * N, lambda: L.call("bsed_igemm3s", ctypes.byref(d), _i(G), L.stream()),
            _esz(inp) * NB * H * W * (CIN + N))
    return out, stats


def igemm3(inp, w3, N, NB, H, W, CIN, taps, bias=None, epilogue=EPI_PLAIN, valid=None):
    """3x3 conv forward / dgrad on the bf16 matrix cores with split-fp32 operands.  Returns (out, stats or None).
    valid: see igemm."""
    if w3.dim() == 6:
        return _igemm3n(inp, w3, N, NB, H, W, CIN, taps, bias, epilogue, valid)
    if inp.dtype != torch.float32:
        raise L.BsedError("bf16 activations need the N-split kernel (BSED_IGEMM3N=1)")
    d = IgemmDesc()
    TH, TW = tile_for(W)
    NP = w3.shape[2]
    bn = 128 if NP % 128 == 0 else (64 if NP % 64 == 0 else 32)
    import os
    if os.environ.get("BSED_IGEMM3_BN") in ("64", "32"):   # A/B knob (csrc/igemm3.hip): labels follow the launch
        bn = min(bn, int(os.environ["BSED_IGEMM3_BN"]))
    # 256-position tiles (two row blocks per wave: half the LDS reads and weight-slab stagings per MFMA) when they
    # still fill the chip: 128 output channels per workgroup and at least ~4 workgroups per CU
    rb = 2 if (IGEMM3_RB["rb"] == 2 and bn == 128 and H >= 2 * TH and (2 * TH + 2) * (TW + 2) <= 384 and
               NB * ((H + 2 * TH - 1) // (2 * TH)) * (W // TW) * (NP // 128) >= 1024) else 1
    TH *= rb
    dev = inp.device
    out = (torch.zeros if valid else torch.empty)((NB, H, W, N), device=dev, dtype=torch.float32)
    if valid:
        d.valid_h, d.valid_w = valid
    ntiles = NB * ((H + TH - 1) // TH) * (W // TW)
    stats = torch.empty((ntiles, 2, N), device=dev, dtype=torch.float32) if epilogue == EPI_STATS else None
    d.in_ = _dp(inp); d.w = w3.data_ptr(); d.bias = _p(bias); d.out = _p(out); d.stats = _p(stats)
    d.in_pitch, d.out_pitch, d.e_pitch = CIN, N, N
    d.NB, d.H, d.W, d.CIN, d.N, d.NP = NB, H, W, CIN, N, NP
    d.TH, d.TW = TH, TW
    d.hh = max(abs(t[0]) for t in taps); d.hw = max(abs(t[1]) for t in taps)
    d.ntaps = len(taps)
    for i, (a, b) in enumerate(taps):
        d.dh[i], d.dw[i] = a, b
    d.ph = d.pw = 1; d.Hp, d.Wp = H, W
    d.epilogue = epilogue
    need = ((TH + 2 * d.hh) * (TW + 2 * d.hw) * 8 + 255) // 256        # float4 patch elements per thread (launch_i3)
    pv = 12 if rb == 2 or need > 9 else (9 if need > 6 else 6)
    _launch((f"igemm3_kernel<{bn}, {1 if epilogue == EPI_STATS else 0}, {rb}, {pv}>", len(taps), CIN, N, H, W),
            2.0 * NB * H * W * len(taps) * CIN * N, lambda: L.call("bsed_igemm3", ctypes.byref(d), L.stream()))
    return out, stats


IGEMM3N_WPE = {"wpe": None}   # A/B knob (set_igemm3n_wpe)


def set_igemm3n_shape(shape):
    """A/B knob: MFMA shape of the nine-tap N-split instances: 16 (16 x 16 x 32), 32 (32 x 32 x 16), 0 / None = default"""
    L.lib().bsed_igemm3n_set_shape(_i(shape or 0))


def set_igemm3n_wpe(wpe):
    """A/B knob of the BN = 128 build: 2 / 3 = built for that many waves per SIMD whatever the shape; + 8 = no raised
    wave priority outside the MFMA loop; 0 / None = default"""
    IGEMM3N_WPE["wpe"] = wpe
    L.lib().bsed_igemm3n_set_wpe(_i(wpe or 0))


def _igemm3n(inp, wtab, N, NB, H, W, CIN, taps, bias, epilogue, valid):
    """bsed_igemm3n: wtab = pack_weight3s table of the layer (K = CIN).  Returns (out, stats (rows,2,N) or None)."""
    d = IgemmDesc()
    TH, TW = tile_for(W)
    NP = wtab.shape[0] * 32
    if wtab.shape[2] * 16 != CIN or wtab.shape[1] != len(taps):
        raise L.BsedError(f"igemm3: weight table packed for K={wtab.shape[2] * 16}, {wtab.shape[1]} taps; "
                          f"called with CIN={CIN}, {len(taps)} taps")
    dev = inp.device
    out = (torch.zeros if valid else torch.empty)((NB, H, W, N), device=dev, dtype=inp.dtype)
    if valid:
        d.valid_h, d.valid_w = valid
    d.act_bf16 = _abf(inp)
    d.in_pitch, d.out_pitch, d.e_pitch = CIN, N, N
    d.NB, d.H, d.W, d.CIN, d.N, d.NP = NB, H, W, CIN, N, NP
    d.TH, d.TW = TH, TW
    d.hh = max(abs(t[0]) for t in taps); d.hw = max(abs(t[1]) for t in taps)
    d.ntaps = len(taps)
    for i, (a, b) in enumerate(taps):
        d.dh[i], d.dw[i] = a, b
    d.ph = d.pw = 1; d.Hp, d.Wp = H, W
    d.epilogue = epilogue
    if IGEMM3N_WPE.get("stamp") is not None:   # diagnostic builds only (tools/conv_stamp.py, -DI3N_STAMP)
        d.e_src = IGEMM3N_WPE["stamp"].data_ptr()
    d.in_ = _dp(inp); d.w = wtab.data_ptr(); d.bias = _p(bias); d.out = out.data_ptr()
    rows = L.lib().bsed_igemm3n_stats_rows(ctypes.byref(d))
    var = L.lib().bsed_igemm3n_variant(ctypes.byref(d))
    stats = torch.empty((rows, 2, N), device=dev, dtype=torch.float32) if epilogue == EPI_STATS else None
    d.stats = _p(stats)
    _launch((f"igemm3n_kernel<{var & 15}, {(var >> 4) & 15}, {1 if epilogue == EPI_STATS else 0}, {(var >> 8) & 15}, "
             f"{1 if len(taps) == 9 else 0}, {(var >> 12) & 15}, {(var >> 16) & 1}, {(var >> 17) & 1}>",
             len(taps), CIN, N, H, W),
            2.0 * NB * H * W * len(taps) * CIN * N, lambda: L.call("bsed_igemm3n", ctypes.byref(d), L.stream()),
            _esz(inp) * NB * H * W * (CIN + N))
    return out, stats


WGRAD_MODE = {"mode": None}  # None = follow BSED_CONV_MODE; "fp32" / "bf16x3" force one


def wgrad(inp, dy, NB, H, W, CIN, N, taps=((0, 0),), in_pitch=None, dy_pitch=None, a_scale=None, a_shift=None,
          in_offset=0, dy_offset=0, mode=None, bn_y=None, bn_coef=None, bn_mean=None, dy_out=None):
    """Partial slabs of dW; returns (part, G, CINP, NP).  mode "bf16x3" = split-fp32 operands on the bf16 cores.
    bn_y / bn_coef / bn_mean: `dy` is dL/d(BatchNorm output) and BatchNorm's backward is applied on load (bf16x3 only);
    dy_out then receives d_y for the data-gradient convolution."""
    import os
    mode = mode or WGRAD_MODE["mode"] or os.environ.get("BSED_CONV_MODE", "bf16x3")
    sfx = "3" if mode in ("bf16x3", "bf16") else ""
    d = WgradDesc()
    TH, TW = tile_for(W)
    CINP, NP = round_up(CIN, 32), round_up(N, 32)
    ntiles = NB * ((H + TH - 1) // TH) * (W // TW)
    d.in_ = _dp(inp, in_offset); d.dy = _dp(dy, dy_offset)
    d.a_scale = _p(a_scale); d.a_shift = _p(a_shift)
    d.act_bf16 = _abf(inp, dy, bn_y, dy_out)
    if d.act_bf16 and sfx != "3":
        raise L.BsedError("bf16 activations need the split-fp32 / bf16 weight-gradient kernels (mode bf16x3 or bf16)")
    d.bn_y, d.bn_coef, d.bn_mean, d.dy_out = _dp(bn_y), _p(bn_coef), _p(bn_mean), _dp(dy_out)
    d.in_pitch = CIN if in_pitch is None else in_pitch
    d.dy_pitch = N if dy_pitch is None else dy_pitch
    d.NB, d.H, d.W, d.CIN, d.CINP, d.N, d.NP, d.G = NB, H, W, CIN, CINP, N, NP, 0
    d.TH, d.TW = TH, TW
    d.hh = max(abs(t[0]) for t in taps); d.hw = max(abs(t[1]) for t in taps)
    d.ntaps = len(taps)
    for i, (a, b) in enumerate(taps):
        d.dh[i], d.dw[i] = a, b
    G = getattr(L.lib(), f"bsed_wgrad{sfx}_auto_g")(ctypes.byref(d))
    if G <= 0:
        raise L.BsedError("bsed_wgrad_auto_g: " + L.lib().bsed_last_error().decode())
    part = torch.empty((G, len(taps), CINP, NP), device=inp.device, dtype=torch.float32)
    d.part, d.G = _p(part), G
    var = getattr(L.lib(), f"bsed_wgrad{sfx}_variant")(ctypes.byref(d))
    # labels = the template instances as rocprofv3 prints them (bench.py joins the two by name)
    w1 = (var >> 13) & 1
    bs, geo, var = (var >> 12) & 1, (var >> 8) & 0xf, var & 0xff
    ab = d.act_bf16
    if w1:
        kname = f"wgrad1_kernel<{'true' if var & 1 else 'false'}, {ab}>"
    elif var % 16 == 1:
        kname = f"wgrad3p_kernel<{var // 16}, {geo}, {ab}>"
    elif sfx:
        kname = f"wgrad3_kernel<{var // 16}, {var % 16}, {'true' if bs else 'false'}, {ab}>"
    else:
        kname = f"wgrad_kernel<{var // 16}, {var % 16}>"
    # algorithmic bytes: the input and dy once; with the fused BatchNorm backward also y (read) and d_y (written)
    nbytes = _esz(inp) * NB * H * W * (CIN + N * (1 + (1 if bn_y is not None else 0) + (1 if dy_out is not None else 0)))
    _launch((kname, len(taps), CIN, N, H, W),
            2.0 * NB * H * W * len(taps) * CIN * N, lambda: L.call(f"bsed_wgrad{sfx}", ctypes.byref(d), L.stream()), nbytes)
    return part, G, CINP, NP


class ReduceJob(ctypes.Structure):
    _fields_ = ([("part", _fp), ("dst", _fp)]
                + [(n, _i) for n in ("G", "ntaps", "KP", "NP", "K", "N", "accumulate")]
                + [(n, ctypes.c_long) for n in ("s_tap", "s_k", "s_n")])


REDUCE_MAX_JOBS = 40
_rq = None   # the active queue of deferred reductions: dict(stream=..., jobs=[(ReduceJob fields..., keepalive)])


class deferred_reductions:
    """Inside this block the partial-slab reductions whose results nothing in the block reads (weight and bias
    gradients: only the optimizer / the gradient all-reduce need them) are queued instead of launched, and go out
    together -- one bsed_reduce_partials_batch launch per flush instead of ~30 launches of 10 us each.  flush() in the
    middle publishes everything queued so far (the data-parallel trainer does that before its early all-reduce).
    Reductions issued on another stream than the one the block was opened on run immediately."""

    def __enter__(self):
        global _rq
        self._outer = _rq
        _rq = {"stream": L.stream().value, "jobs": []}
        return self

    def __exit__(self, *exc):
        global _rq
        try:
            if exc[0] is None:
                flush_reductions()
        finally:
            _rq = self._outer
        return False


def flush_reductions():
    """launch what the active queue holds (jobs writing the same destination go to successive launches, in order)"""
    if _rq is None or not _rq["jobs"]:
        return
    pending, _rq["jobs"] = _rq["jobs"], []
    while pending:
        batch, rest, seen = [], [], set()
        for job in pending:
            key = job[1]
            if key in seen or len(batch) == REDUCE_MAX_JOBS or rest:
                rest.append(job)          # keep the order of everything behind the first job that has to wait
            else:
                seen.add(key)
                batch.append(job)
        arr = (ReduceJob * len(batch))()
        nel = 0
        for i, (part, dst_ptr, G, ntaps, KP, NP, K, N, s_tap, s_k, s_n, acc) in enumerate(batch):
            arr[i].part, arr[i].dst = part.data_ptr(), dst_ptr
            arr[i].G, arr[i].ntaps, arr[i].KP, arr[i].NP, arr[i].K, arr[i].N = G, ntaps, KP, NP, K, N
            arr[i].accumulate, arr[i].s_tap, arr[i].s_k, arr[i].s_n = acc, s_tap, s_k, s_n
            nel += part.numel()
        _note("reduce_partials_batch_kernel", f"jobs{len(batch)}", float(nel), 4.0 * nel)
        L.call("bsed_reduce_partials_batch", arr, _i(len(batch)), L.stream())
        pending = rest


def reduce_partials(part, G, ntaps, KP, NP, K, N, dst, s_tap, s_k, s_n, accumulate=True, dst_offset=0, defer=True):
    """defer=False: run now even inside ops.deferred_reductions() (the caller reads dst right away)"""
    if defer and _rq is not None and _rq["stream"] == L.stream().value:
        _rq["jobs"].append((part, _dp(dst, dst_offset), G, ntaps, KP, NP, K, N, s_tap, s_k, s_n, 1 if accumulate else 0))
        return
    _note("reduce_partials_kernel", f"G{G}", float(part.numel()), 4.0 * part.numel())
    L.call("bsed_reduce_partials", L.ptr(part), _i(G), _i(ntaps), _i(KP), _i(NP), _i(K), _i(N),
           _fp(_dp(dst, dst_offset)), ctypes.c_long(s_tap), ctypes.c_long(s_k), ctypes.c_long(s_n),
           _i(1 if accumulate else 0), L.stream())


_scratch = {}


def stats_scratch(C, device):
    key = (C, str(device), L.stream().value)   # one scratch per stream: launches of different streams may overlap
    s = _scratch.get(key)
    if s is None:
        n = L.lib().bsed_stats_scratch_bytes
        n.restype = ctypes.c_size_t
        s = _scratch[key] = torch.empty((n(_i(C)) + 7) // 8, device=device, dtype=torch.float64)
    return s


def conv0_fwd(x, w, bias, NB, H, W, CO, want_stats):
    y = torch.empty((NB, H, W, CO), device=x.device, dtype=torch.float32)
    stats = None
    nt = L.lib().bsed_conv0_num_tiles(NB, H, W)
    if want_stats:
        stats = torch.empty((nt, 2, CO), device=x.device, dtype=torch.float32)
    _note(f"conv0_fwd_kernel<{CO}>", f"{H}x{W}", 2.0 * 9 * CO * NB * H * W, 4.0 * NB * H * W * (1 + CO))
    L.call("bsed_conv0_fwd", L.ptr(x), _fp(_dp(w)), _fp(_dp(bias)), L.ptr(y), _fp(_p(stats)), _i(NB), _i(H), _i(W),
           _i(CO), L.stream())
    return y, stats


def conv0_wgrad(x, dy, NB, H, W, CO, y=None, coef=None, mean=None):
    """y/coef/mean given: dy is dL/d(BatchNorm output) and BatchNorm's backward is applied on load"""
    G = min(1024, max(1, (NB * H * W) // 256))
    part = torch.empty((G, 9, CO), device=x.device, dtype=torch.float32)
    # x, dy (= g) and, with the fused BatchNorm backward, y: each read once
    _note(f"conv0_wgrad_kernel<{CO}, {'true' if y is not None else 'false'}>", f"{H}x{W}",
          2.0 * 9 * CO * NB * H * W, 4.0 * NB * H * W * (1 + CO * (2 if y is not None else 1)))
    L.call("bsed_conv0_wgrad", L.ptr(x), L.ptr(dy), _fp(_p(y)), _fp(_p(coef)), _fp(_p(mean)), L.ptr(part), _i(G),
           _i(NB), _i(H), _i(W), _i(CO), L.stream())
    return part, G


def glu16_fwd(y, scale, shift, wg, bg, B, H, W, pool, drop_p, rng_stream, seed):
    ph, pw = pool
    out = torch.empty((B, H // ph, W // pw, 16), device=y.device, dtype=torch.float32)
    _note("glu16_fwd_kernel", f"{H}x{W}", 2.0 * B * H * W * 256, 4.0 * B * H * W * 16 * (1.0 + 1.0 / (ph * pw)))
    L.call("bsed_glu16_fwd", L.ptr(y), L.ptr(scale), L.ptr(shift), _fp(_dp(wg)), _fp(_dp(bg)), L.ptr(out), _i(B), _i(H),
           _i(W), _i(16), _i(ph), _i(pw), ctypes.c_float(drop_p), ctypes.c_uint32(rng_stream), ctypes.c_uint64(seed),
           L.stream())
    return out


def glu16_bwd(y, scale, shift, wg, bg, dpool, B, H, W, pool, drop_p, rng_stream, seed):
    """returns (g, part_dw (G,16,16), part_db (G,2,16), part_st (G,2,16), G)"""
    ph, pw = pool
    dev = y.device
    G = int(min(2048, B * H))
    g = torch.empty_like(y)
    part_dw = torch.empty((G, 16, 16), device=dev, dtype=torch.float32)
    part_db = torch.empty((G, 2, 16), device=dev, dtype=torch.float32)
    part_st = torch.empty((G, 2, 16), device=dev, dtype=torch.float32)
    _note("glu16_bwd_kernel", f"{H}x{W}", 3 * 2.0 * B * H * W * 256, 4.0 * B * H * W * 16 * (2.0 + 1.0 / (ph * pw)))
    L.call("bsed_glu16_bwd", L.ptr(y), L.ptr(scale), L.ptr(shift), _fp(_dp(wg)), _fp(_dp(bg)), L.ptr(dpool), L.ptr(g),
           L.ptr(part_dw), L.ptr(part_db), L.ptr(part_st), _i(G), _i(B), _i(H), _i(W), _i(16), _i(ph), _i(pw),
           ctypes.c_float(drop_p), ctypes.c_uint32(rng_stream), ctypes.c_uint64(seed), L.stream())
    return g, part_dw, part_db, part_st, G


def block0_stats(x, cw, cb, NB, H, W):
    """train-mode statistics of the first block from x alone: returns (stats (G,2,16) for bn_finalize, xr64 (54,) fp64)"""
    import os
    G = int(min(int(os.environ.get("BSED_B0_STATS_G", "2048")), max(1, (NB * H * W) // 256)))   # (A/B knob)
    dev = x.device
    stats = torch.empty((G, 2, 16), device=dev, dtype=torch.float32)
    xr_part = torch.empty((G, 54), device=dev, dtype=torch.float32)
    xr64 = torch.empty(54, device=dev, dtype=torch.float64)
    _note("b0_stats_kernel", f"{H}x{W}", 2.0 * NB * H * W * (9 * 16 + 16 + 54), 4.0 * NB * H * W)
    cwt = cw.detach().reshape(16, 9).t().contiguous()   # [tap][channel]: channel pairs become packed scalar operands
    L.call("bsed_block0_stats", L.ptr(x), L.ptr(cwt), _fp(_dp(cb)), L.ptr(stats), L.ptr(xr_part),
           L.ptr(xr64, torch.float64), _i(G), _i(NB), _i(H), _i(W), _i(16), L.stream())
    return stats, xr64


def block0_fwd(x, cw, cb, scale, shift, wg, bg, B, H, W, pool, drop_p, rng_stream, seed, out_dtype=torch.float32):
    ph, pw = pool
    out = torch.empty((B, H // ph, W // pw, 16), device=x.device, dtype=out_dtype)
    ab = _abf(out)
    small = "false" if ab else ("true" if B * H * W < (1 << 28) else "false")
    _note(f"b0_fwd_kernel<{ph}, {small}, {ab}>", f"{H}x{W}", 2.0 * B * H * W * (9 * 16 + 256),
          B * H * W * (4.0 + _esz(out) * 16.0 / (ph * pw)))
    L.call("bsed_block0_fwd", L.ptr(x), _fp(_dp(cw)), _fp(_dp(cb)), L.ptr(scale), L.ptr(shift), _fp(_dp(wg)),
           _fp(_dp(bg)), _pa(out), _i(B), _i(H), _i(W), _i(16), _i(ph), _i(pw), ctypes.c_float(drop_p),
           ctypes.c_uint32(rng_stream), ctypes.c_uint64(seed), _i(ab), L.stream())
    return out


def block0_bwd(x, cw, cb, scale, shift, wg, bg, dpool, B, H, W, pool, drop_p, rng_stream, seed):
    """returns (part_dw (G,16,16), part_db (G,2,16), part_st (G,2,16), part_gx (G,9,16), G)"""
    ph, pw = pool
    dev = x.device
    import os
    G = int(min(int(os.environ.get("BSED_B0_BWD_G", "8192")), B * (H // ph)))   # 2048 / 4096 / 8192 workgroups: 0.983 / 0.966 / 0.957 ms
    part_dw = torch.empty((G, 16, 16), device=dev, dtype=torch.float32)
    part_db = torch.empty((G, 2, 16), device=dev, dtype=torch.float32)
    part_st = torch.empty((G, 2, 16), device=dev, dtype=torch.float32)
    part_gx = torch.empty((G, 9, 16), device=dev, dtype=torch.float32)
    ab = _abf(dpool)
    _note(f"b0_bwd_kernel<{ph}, {'true' if B * H * W < (1 << 28) else 'false'}, {ab}>", f"{H}x{W}", 2.0 * B * H * W * (2 * 9 * 16 + 3 * 256),
          B * H * W * (4.0 + _esz(dpool) * 16.0 / (ph * pw)))
    L.call("bsed_block0_bwd", L.ptr(x), _fp(_dp(cw)), _fp(_dp(cb)), L.ptr(scale), L.ptr(shift), _fp(_dp(wg)),
           _fp(_dp(bg)), _pa(dpool), L.ptr(part_dw), L.ptr(part_db), L.ptr(part_st), L.ptr(part_gx), _i(G), _i(B),
           _i(H), _i(W), _i(16), _i(ph), _i(pw), ctypes.c_float(drop_p), ctypes.c_uint32(rng_stream),
           ctypes.c_uint64(seed), _i(ab), L.stream())
    return part_dw, part_db, part_st, part_gx, G


def block0_wgrad_finish(part_gx, G, xr64, coef, mean, cw, cb, dst, accumulate=True):
    _note("b0_wgrad_finish_kernel", f"G{G}", float(part_gx.numel()), 4.0 * part_gx.numel())
    L.call("bsed_block0_wgrad_finish", L.ptr(part_gx), _i(G), L.ptr(xr64, torch.float64), L.ptr(coef), L.ptr(mean),
           _fp(_dp(cw)), _fp(_dp(cb)), _fp(_dp(dst)), _i(1 if accumulate else 0), _i(16), L.stream())


def glu_bwd_fused(y, scale, shift, wfwd, w, bias, dpool, B, H, W, C, pool, drop_p, rng_stream, seed):
    """returns (g, part_dw (G*slabs,C,C), part_db (G,2,C), part_st (G,2,C), G, slabs)"""
    ph, pw = pool
    dev = y.device
    TH, TW = tile_for(W)
    ntiles = B * ((H + TH - 1) // TH) * (W // TW)
    G = int(min(ntiles, 256 * (1 if C == 128 else (2 if C == 64 else 3))))
    slabs = L.lib().bsed_glu_bwd_slabs(C)
    g = torch.empty_like(y)
    part_dw = torch.empty((G * slabs, C, C), device=dev, dtype=torch.float32)
    part_db = torch.empty((G, 2, C), device=dev, dtype=torch.float32)
    part_st = torch.empty((G, 2, C), device=dev, dtype=torch.float32)
    flops = 3 * 2.0 * B * H * W * C * C
    nbytes = 4.0 * B * H * W * C * (2.0 + 1.0 / (ph * pw))  # y, g, d_pooled
    _launch((f"glu_bwd_fused_kernel<{C}, {8 if C == 128 else 4}>", 1, C, C, H, W), flops,
            lambda: L.call("bsed_glu_bwd_fused", L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(wfwd), _fp(_dp(w)),
                           _fp(_dp(bias)), L.ptr(dpool), L.ptr(g), L.ptr(part_dw), L.ptr(part_db), L.ptr(part_st),
                           _i(G), _i(B), _i(H), _i(W), _i(C), _i(TH), _i(TW), _i(ph), _i(pw), ctypes.c_float(drop_p),
                           ctypes.c_uint32(rng_stream), ctypes.c_uint64(seed), L.stream()), nbytes)
    return g, part_dw, part_db, part_st, G, slabs


def glu_fwd3_supported(W, C, pool):
    TH, TW = tile_for(W)
    return (C in (32, 64, 128) and (pool[1] == 1 or TW >= 2) and
            (pool[0] == 1 or ((TW in (2, 8, 16) or (TW == 1 and pool[1] == 1)) and TH % 2 == 0)))


def glu_fwd3(y, scale, shift, w, bias, B, H, W, C, pool, drop_p, rng_stream, seed):
    """split-fp32 GLU forward: BN-apply -> Linear -> gate -> dropout -> avg-pool in one pass over y"""
    ph, pw = pool
    TH, TW = tile_for(W)
    ntiles = B * ((H + TH - 1) // TH) * (W // TW)
    G = int(min(ntiles, L.lib().bsed_glu_fwd3_auto_g(C)))
    out = torch.empty((B, H // ph, W // pw, C), device=y.device, dtype=y.dtype)
    ab = _abf(y)
    _launch((f"glu_fwd3_kernel<{C}, {4 if TW == 16 else -1}, {ab}>", 1, C, C, H, W), 2.0 * B * H * W * C * C,
            lambda: L.call("bsed_glu_fwd3", _pa(y), L.ptr(scale), L.ptr(shift), _fp(_dp(w)), _fp(_dp(bias)),
                           _pa(out), _i(G), _i(B), _i(H), _i(W), _i(C), _i(TH), _i(TW), _i(ph), _i(pw),
                           ctypes.c_float(drop_p), ctypes.c_uint32(rng_stream), ctypes.c_uint64(seed), _i(ab), L.stream()),
            _esz(y) * B * H * W * C * (1.0 + 1.0 / (ph * pw)))  # y, pooled
    return out


def glu_bwd3(y, scale, shift, w, bias, dpool, B, H, W, C, pool, drop_p, rng_stream, seed):
    """split-fp32 fused GLU backward (C in {32,64}); returns like glu_bwd_fused"""
    ph, pw = pool
    dev = y.device
    TH, TW = tile_for(W)
    ntiles = B * ((H + TH - 1) // TH) * (W // TW)
    G = int(min(ntiles, L.lib().bsed_glu_bwd3_auto_g(C)))
    slabs = L.lib().bsed_glu_bwd3_slabs(C)
    g = torch.empty_like(y)
    part_dw = torch.empty((G * slabs, C, C), device=dev, dtype=torch.float32)
    part_db = torch.empty((G, 2, C), device=dev, dtype=torch.float32)
    part_st = torch.empty((G, 2, C), device=dev, dtype=torch.float32)
    flops = 3 * 2.0 * B * H * W * C * C
    ab = _abf(y, dpool)
    _launch((f"glu_bwd3_kernel<{C}, {4 if TW == 16 else -1}, {ab}>", 1, C, C, H, W), flops,
            lambda: L.call("bsed_glu_bwd3", _pa(y), L.ptr(scale), L.ptr(shift), _fp(_dp(w)), _fp(_dp(bias)),
                           _pa(dpool), _pa(g), L.ptr(part_dw), L.ptr(part_db), L.ptr(part_st), _i(G), _i(B), _i(H),
                           _i(W), _i(C), _i(TH), _i(TW), _i(ph), _i(pw), ctypes.c_float(drop_p),
                           ctypes.c_uint32(rng_stream), ctypes.c_uint64(seed), _i(ab), L.stream()),
            _esz(y) * B * H * W * C * (2.0 + 1.0 / (ph * pw)))  # y, g, d_pooled
    return g, part_dw, part_db, part_st, G, slabs


_frag_tables = {}


def glu_bwd3n(y, scale, shift, w, bias, dpool, B, H, W, C, pool, drop_p, rng_stream, seed):
    """split-fp32 GLU backward for C = 128 without the weight gradient: returns (g, d_lin, part_db, part_st, G)"""
    ph, pw = pool
    dev = y.device
    TH, TW = tile_for(W)
    ntiles = B * ((H + TH - 1) // TH) * (W // TW)
    G = int(min((ntiles + 1) // 2, 256))  # 8-wave workgroups take two tiles at a time
    tb = _frag_tables.get(str(dev))
    if tb is None:
        fn = L.lib().bsed_glu_bwd3n_table_bytes
        fn.restype = ctypes.c_size_t
        tb = _frag_tables[str(dev)] = torch.empty(fn(), device=dev, dtype=torch.uint8)
    g = torch.empty_like(y)
    dlin = torch.empty_like(y)
    part_db = torch.empty((G, 2, C), device=dev, dtype=torch.float32)
    part_st = torch.empty((G, 2, C), device=dev, dtype=torch.float32)
    ab = _abf(y, dpool)
    _launch((f"glu_bwd3n_kernel<{ab}>", 1, C, C, H, W), 2 * 2.0 * B * H * W * C * C,
            lambda: L.call("bsed_glu_bwd3n", _pa(y), L.ptr(scale), L.ptr(shift), _fp(_dp(w)), _fp(_dp(bias)),
                           _pa(dpool), _pa(g), _pa(dlin), L.ptr(part_db), L.ptr(part_st),
                           ctypes.c_void_p(tb.data_ptr()), _i(G), _i(B), _i(H), _i(W), _i(C), _i(TH), _i(TW), _i(ph),
                           _i(pw), ctypes.c_float(drop_p), ctypes.c_uint32(rng_stream), ctypes.c_uint64(seed),
                           _i(ab), L.stream()),
            _esz(y) * B * H * W * C * (3.0 + 1.0 / (ph * pw)))  # y, g, d_lin, d_pooled
    return g, dlin, part_db, part_st, G


def bn_finalize(stats, C, count, eps, momentum, gamma, beta, rmean, rvar, nbt):
    dev = stats.device
    mean, invstd, scale, shift = (torch.empty(C, device=dev, dtype=torch.float32) for _ in range(4))
    _note("stats_chunk_kernel+stats_finish_kernel", f"C{C}", 0.0, 4.0 * stats.numel())
    L.call("bsed_bn_finalize", L.ptr(stats), ctypes.c_long(stats.shape[0]), _i(C), ctypes.c_double(count),
           ctypes.c_float(eps), ctypes.c_float(momentum), _fp(_dp(gamma)), _fp(_dp(beta)), _fp(_dp(rmean)),
           _fp(_dp(rvar)), _fp(None if nbt is None else nbt.data_ptr()), L.ptr(mean), L.ptr(invstd), L.ptr(scale),
           L.ptr(shift), L.ptr(stats_scratch(C, dev), torch.float64), L.stream())
    return mean, invstd, scale, shift


def bn_eval(C, eps, gamma, beta, rmean, rvar):
    dev = gamma.device
    scale, shift = (torch.empty(C, device=dev, dtype=torch.float32) for _ in range(2))
    L.call("bsed_bn_eval", _i(C), ctypes.c_float(eps), _fp(_dp(gamma)), _fp(_dp(beta)), _fp(_dp(rmean)),
           _fp(_dp(rvar)), L.ptr(scale), L.ptr(shift), L.stream())
    return scale, shift


class BsedBnEvalJob(ctypes.Structure):
    _fields_ = [(n, _fp) for n in ("gamma", "beta", "running_mean", "running_var", "scale", "shift")] + [("C", _i)]


def bn_eval_batch(layers, eps):
    """[(C, gamma, beta, running_mean, running_var), ...] -> [(scale, shift), ...] in one launch (eval-mode forward)"""
    dev = layers[0][1].device
    out = torch.empty(2 * sum(l[0] for l in layers), device=dev, dtype=torch.float32)
    arr = (BsedBnEvalJob * len(layers))()
    res, off = [], 0
    for a, (C, gamma, beta, rmean, rvar) in zip(arr, layers):
        scale, shift = out[off:off + C], out[off + C:off + 2 * C]
        off += 2 * C
        a.gamma, a.beta, a.running_mean, a.running_var = _dp(gamma), _dp(beta), _dp(rmean), _dp(rvar)
        a.scale, a.shift, a.C = scale.data_ptr(), shift.data_ptr(), C
        res.append((scale, shift))
    _note("bn_eval_batch_kernel", "", 0.0, 4.0 * 3 * out.numel())
    L.call("bsed_bn_eval_batch", arr, _i(len(layers)), ctypes.c_float(eps), L.stream())
    return res


def bn_bwd(stats, C, count, gamma, mean, invstd, dgamma, dbeta, g_inout, y, apply=True):
    """apply=False: only dgamma/dbeta and the (3,C) coefficients [A|B|C] of d_y = A g + B (y-mean) + C (returned);
    the consumer applies the map on load (conv0_wgrad)."""
    dev = stats.device
    coef = torch.empty((3, C), device=dev, dtype=torch.float32)
    _note("bn_bwd_apply_kernel" if apply else "stats_chunk_kernel+stats_finish_kernel", f"C{C}", 3.0 * y.numel() if apply else 0.0,
          12.0 * y.numel() if apply else 4.0 * stats.numel())
    L.call("bsed_bn_bwd", L.ptr(stats), ctypes.c_long(stats.shape[0]), _i(C), ctypes.c_double(count), _fp(_dp(gamma)),
           L.ptr(mean), L.ptr(invstd), _fp(_dp(dgamma)), _fp(_dp(dbeta)), _i(1),
           L.ptr(g_inout) if apply else None, L.ptr(y) if apply else None,
           ctypes.c_long(y.numel() if apply else C), L.ptr(coef), L.ptr(stats_scratch(C, dev), torch.float64),
           L.stream())
    return coef


def stats_to_grad(stats, C, which, dst):
    if _rq is not None and _rq["stream"] == L.stream().value and which == 0 and stats.dim() == 3 and stats.shape[1] == 2:
        # row 0 of every (2, C) partial: a queued slab reduction with K = 1 of KP = 2 rows (fp32 sums, fixed order)
        _rq["jobs"].append((stats, _dp(dst), stats.shape[0], 1, 2, C, 1, C, 0, 0, 1, 1))
        return
    _note("stats_chunk_kernel+stats_finish_kernel", f"C{C}", 0.0, 4.0 * stats.numel())
    L.call("bsed_stats_to_grad", L.ptr(stats), ctypes.c_long(stats.shape[0]), _i(C), _i(which), _fp(_dp(dst)), _i(1),
           L.ptr(stats_scratch(C, stats.device), torch.float64), L.stream())


def colsum(inp, M, C, pitch, dst, accumulate=True, in_offset=0):
    G = int(min(512, M))
    part = torch.empty((G, 2, C), device=inp.device, dtype=torch.float32)
    _note("colsum_kernel", f"C{C}", float(M) * C, 4.0 * M * C)
    if _rq is not None and _rq["stream"] == L.stream().value:
        L.call("bsed_colsum_part", _fp(_dp(inp, in_offset)), ctypes.c_long(M), _i(C), _i(pitch), L.ptr(part), _i(G),
               L.stream())
        _rq["jobs"].append((part, _dp(dst), G, 1, 2, C, 1, C, 0, 0, 1, 1 if accumulate else 0))
        return
    L.call("bsed_colsum", _fp(_dp(inp, in_offset)), ctypes.c_long(M), _i(C), _i(pitch), L.ptr(part), _i(G),
           _fp(_dp(dst)), _i(1 if accumulate else 0), L.ptr(stats_scratch(C, inp.device), torch.float64), L.stream())


def dropout(x, p, rng_stream, seed):
    out = torch.empty_like(x)
    _note("dropout_kernel", "", float(x.numel()), 8.0 * x.numel())
    L.call("bsed_dropout", L.ptr(x), L.ptr(out), ctypes.c_long(x.numel()), ctypes.c_float(p),
           ctypes.c_uint32(rng_stream), ctypes.c_uint64(seed), L.stream())
    return out


def gru_rows(B):
    """rows per workgroup of the fp32 register kernels so that (B/R) x 2 directions fills the 256 CUs"""
    return 1 if B * 2 <= 256 else 2


def gru_fwd(xp, w_hh, b_hh, B, T, save_gates, mode="fp32"):
    out = torch.empty((B, T, 256), device=xp.device, dtype=torch.float32)
    gates = torch.empty((B, T, 2, 4, 128), device=xp.device, dtype=torch.float32) if save_gates else None
    _note("gru_fwd_mfma_kernel<true>" if mode == "bf16x3" else "gru_fwd_kernel", f"T{T}", 2.0 * B * T * 2 * 128 * 384,
          4.0 * B * T * (768 + 256 + (1024 if save_gates else 0)))
    if mode == "bf16x3":  # matrix-core recurrence, split-fp32 operands
        L.call("bsed_gru_fwd3", L.ptr(xp), _fp(_dp(w_hh)), _fp(_dp(b_hh)), L.ptr(out), _fp(_p(gates)), _i(B), _i(T),
               L.stream())
    else:
        L.call("bsed_gru_fwd", L.ptr(xp), _fp(_dp(w_hh)), _fp(_dp(b_hh)), L.ptr(out), _fp(_p(gates)), _i(B), _i(T),
               _i(gru_rows(B)), L.stream())
    return out, gates


def gru_bwd(dout, out, gates, w_hh, B, T, mode="fp32"):
    dxp = torch.empty((B, T, 768), device=dout.device, dtype=torch.float32)
    dgh = torch.empty((B, T, 768), device=dout.device, dtype=torch.float32)
    _note("gru_bwd_mfma_kernel" if mode == "bf16x3" else "gru_bwd_kernel", f"T{T}", 2.0 * B * T * 2 * 128 * 384,
          4.0 * B * T * (256 + 256 + 1024 + 768 + 768))
    if mode == "bf16x3":
        rows = L.lib().bsed_gru_bwd3_rows(B)
        pih = torch.empty((rows, 768), device=dout.device, dtype=torch.float32)
        phh = torch.empty((rows, 768), device=dout.device, dtype=torch.float32)
        L.call("bsed_gru_bwd3", L.ptr(dout), L.ptr(out), L.ptr(gates), _fp(_dp(w_hh)), L.ptr(dxp), L.ptr(dgh),
               L.ptr(pih), L.ptr(phh), _i(B), _i(T), L.stream())
        return dxp, dgh, pih, phh
    else:
        L.call("bsed_gru_bwd", L.ptr(dout), L.ptr(out), L.ptr(gates), _fp(_dp(w_hh)), L.ptr(dxp), L.ptr(dgh), _i(B),
               _i(T), _i(gru_rows(B)), L.stream())
    return dxp, dgh, None, None


def max_over_time(y):
    """(B,T,C) -> (B,C) maximum over time (weak targets of the train loop: ``target.max(-2)[0]``)"""
    y = y.contiguous().float()
    B, T, C = y.shape
    out = torch.empty((B, C), device=y.device, dtype=torch.float32)
    _note("max_over_time_kernel", f"T{T}", 0.0, 4.0 * (y.numel() + out.numel()))
    L.call("bsed_max_over_time", L.ptr(y), L.ptr(out), _i(B), _i(T), _i(C), L.stream())
    return out


def head_fwd(x, w, b, B, T, K, C, attention):
    dev = x.device
    strong = torch.empty((B, T, C), device=dev, dtype=torch.float32)
    sof = torch.empty((B, T, C), device=dev, dtype=torch.float32)
    weak = torch.empty((B, C), device=dev, dtype=torch.float32)
    den = torch.empty((B, C), device=dev, dtype=torch.float32)
    S = L.lib().bsed_head_splits(B, T)
    part = torch.empty((B, S, 2, C), device=dev, dtype=torch.float32) if S > 1 else None
    _note(f"head_fwd_kernel<{C}>", f"T{T}", 2.0 * B * T * K * 2 * C, 4.0 * B * T * (K + 2 * C))
    L.call("bsed_head_fwd", L.ptr(x), _fp(_dp(w)), _fp(_dp(b)), L.ptr(strong), L.ptr(sof), L.ptr(weak), L.ptr(den),
           L.ptr(part), _i(B), _i(T), _i(K), _i(C), _i(1 if attention else 0), L.stream())
    return strong, sof, weak, den


def head_bwd(x, w, strong, sof, weak, den, B, T, K, C, attention, y_strong=None, y_weak=None, ema_strong=None,
             ema_weak=None, g_strong=None, g_weak=None, w_strong=1.0, w_weak=1.0, w_cons_s=0.0, w_cons_w=0.0,
             ema_strong2=None, w_cons_s2=0.0, n_strong=None, n_weak=None):
    """n_strong / n_weak: element counts the 'mean' reductions divide by (default B*T*C and B*C; pass the FULL batch
    counts when the call covers only a slice of the batch)"""
    dev = x.device
    d = HeadBwdDesc()
    dx = torch.empty((B, T, K), device=dev, dtype=torch.float32)
    S = L.lib().bsed_head_splits(B, T)                               # rows per clip in the partial outputs
    dw_part = torch.empty((B * S, 2 * C, K), device=dev, dtype=torch.float32)
    db_part = torch.empty((B * S, 2 * C), device=dev, dtype=torch.float32)
    loss_part = torch.empty((B * S, 6), device=dev, dtype=torch.float32)
    d.x = _p(x); d.w = _dp(w); d.strong = _p(strong); d.sof_raw = _p(sof); d.weak = _p(weak); d.den = _p(den)
    d.y_strong = _p(y_strong); d.y_weak = _p(y_weak); d.ema_strong = _p(ema_strong); d.ema_weak = _p(ema_weak)
    d.g_strong_ext = _p(g_strong); d.g_weak_ext = _p(g_weak); d.ema_strong2 = _p(ema_strong2)
    d.w_strong, d.w_weak, d.w_cons_s, d.w_cons_w, d.w_cons_s2 = w_strong, w_weak, w_cons_s, w_cons_w, w_cons_s2
    d.inv_n_strong = 1.0 / (n_strong if n_strong else B * T * C)
    d.inv_n_weak = 1.0 / (n_weak if n_weak else B * C)
    d.dx = _p(dx); d.dw_part = _p(dw_part); d.db_part = _p(db_part); d.loss_part = _p(loss_part)
    d.B, d.T, d.K, d.C, d.attention = B, T, K, C, 1 if attention else 0
    _note(f"head_bwd_kernel<{C}>", f"T{T}", 3 * 2.0 * B * T * K * 2 * C, 4.0 * B * T * (2 * K + 4 * C))
    L.call("bsed_head_bwd", ctypes.byref(d), L.stream())
    return dx, dw_part, db_part, loss_part


def tag_head_fwd(x, logits):
    """CNN-only tagging head (csrc/tag.hip): x, logits (B,T,128) -> (strong (B,T,128), weak (B,128))"""
    B, T, C = x.shape
    S = L.lib().bsed_tag_splits(B, T)
    strong = torch.empty_like(x)
    weak = torch.empty((B, C), device=x.device, dtype=torch.float32)
    part = torch.empty((B, S, 2, C), device=x.device, dtype=torch.float32)
    _note("tag_pool_kernel", f"T{T}", 12.0 * B * T * C, 12.0 * B * T * C)
    L.call("bsed_tag_head_fwd", L.ptr(x), L.ptr(logits), L.ptr(strong), L.ptr(weak), L.ptr(part), _i(B), _i(T), _i(C),
           L.stream())
    return strong, weak


def adam_step(p, g, m, v, lr, step, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    _note("adam_kernel", "", 12.0 * p.numel(), 28.0 * p.numel())
    L.call("bsed_adam_step", L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), ctypes.c_long(p.numel()), ctypes.c_float(lr),
           ctypes.c_float(betas[0]), ctypes.c_float(betas[1]), ctypes.c_float(eps), ctypes.c_float(weight_decay),
           ctypes.c_long(step), ctypes.c_float(grad_scale), L.stream())


def sgd_step(p, g, buf, lr, momentum, weight_decay, first_step, nesterov=True, grad_scale=1.0):
    _note("sgd_kernel", "", 6.0 * p.numel(), 20.0 * p.numel())
    L.call("bsed_sgd_step", L.ptr(p), L.ptr(g), L.ptr(buf), ctypes.c_long(p.numel()), ctypes.c_float(lr),
           ctypes.c_float(momentum), ctypes.c_float(weight_decay), _i(1 if first_step else 0),
           _i(1 if nesterov else 0), ctypes.c_float(grad_scale), L.stream())


def roll(x, B, H, W, sh=None, sw=None):
    """per-sample torch.roll of a contiguous (B,H,W[,..]) tensor viewed as (B,H,W); sh/sw: int32 device tensors (B)"""
    out = torch.empty_like(x)
    _note("roll_kernel", "", 0.0, 8.0 * x.numel())
    L.call("bsed_roll", L.ptr(x), L.ptr(out), _i(B), _i(H), _i(W), L.ptr(sh, torch.int32), L.ptr(sw, torch.int32),
           L.stream())
    return out


def axpy(y, x, a=1.0):
    """y += a * x in place (contiguous fp32 tensors of equal size)"""
    _note("axpy_kernel", "", 2.0 * y.numel(), 12.0 * y.numel())
    L.call("bsed_axpy", L.ptr(y), L.ptr(x.contiguous()), ctypes.c_long(y.numel()), ctypes.c_float(a), L.stream())
    return y


def ema_update(ema, p, alpha):
    _note("ema_kernel", "", 3.0 * p.numel(), 12.0 * p.numel())
    L.call("bsed_ema_update", L.ptr(ema), L.ptr(p), ctypes.c_long(p.numel()), ctypes.c_float(alpha), L.stream())


def ema_update_i64(ema, p, alpha):
    L.call("bsed_ema_update_i64", L.ptr(ema, torch.int64), L.ptr(p, torch.int64), _i(p.numel()),
           ctypes.c_float(alpha), L.stream())


def upsample_time(inp, T_out, out=None, out_offset=0):
    """(B,T_in,C) -> (B,T_out,C) linear interpolation along time (align_corners=True); ``out`` may be a wider
    (B,T_out,P) buffer whose columns out_offset..out_offset+C receive the result (concatenation without a copy)"""
    B, T_in, C = inp.shape
    if out is None:
        out = torch.empty((B, T_out, C), device=inp.device, dtype=torch.float32)
    L.call("bsed_upsample_time_fwd", L.ptr(inp), _fp(_dp(out, out_offset)), _i(B), _i(T_in), _i(T_out), _i(C),
           _i(inp.shape[2]), _i(out.shape[2]), L.stream())
    return out


def upsample_time_bwd(dout, T_in, C, in_offset=0):
    """adjoint of upsample_time: dout (B,T_out,P) columns in_offset..in_offset+C -> (B,T_in,C)"""
    B, T_out, P = dout.shape
    din = torch.empty((B, T_in, C), device=dout.device, dtype=torch.float32)
    L.call("bsed_upsample_time_bwd", _fp(_dp(dout, in_offset)), L.ptr(din), _i(B), _i(T_in), _i(T_out), _i(C), _i(P),
           _i(C), L.stream())
    return din
