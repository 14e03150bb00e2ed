"""Thin tensor-level wrappers over the C ABI (spegnet_amd/_lib.py).  Every function launches hand-written
HIP kernels on torch's current stream; tensors only provide device memory.  No CPU path exists."""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ACT_GELU, ACT_GELU_SAVE_GRAD, ACT_MUL_H, ACT_NONE, ACT_RELU, SPG_BF16, SPG_F32  # noqa: F401

Tensor = torch.Tensor


def dcode(t: Tensor) -> int:
    if t.dtype == torch.float32:
        return SPG_F32
    if t.dtype == torch.bfloat16:
        return SPG_BF16
    raise TypeError(f"unsupported dtype {t.dtype} (float32 or bfloat16)")


def _p(t: Optional[Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("spegnet_amd ops need CUDA/HIP tensors: there is no CPU fallback")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _c(t: Tensor) -> Tensor:
    if not t.is_contiguous():
        raise RuntimeError(f"expected a contiguous tensor, got strides {t.stride()} for shape {tuple(t.shape)}")
    return t


def f32(t: Tensor) -> Tensor:
    assert t.dtype == torch.float32, t.dtype
    return _c(t)


# ---- zero-initialised f32 scratch (atomic accumulators, packed weight gradients) -------------------------------
# A train step needs ~50 small zeroed buffers; each torch.zeros is its own fill launch.  TrainStep opens a pool at the start of
# the step: ONE fill (inside a captured step: one memset node), slices handed out in order.  Without a pool (inference, tests
# calling ops directly) or when it is exhausted, plain torch.zeros.
_ZPOOL = None   # [tensor, offset_in_floats]


def begin_zero_pool(device, nfloats: int = 8 << 20) -> None:
    global _ZPOOL
    _ZPOOL = [torch.zeros(nfloats, dtype=torch.float32, device=device), 0]


def end_zero_pool() -> None:
    global _ZPOOL
    _ZPOOL = None


def zeros_f32(shape, device) -> Tensor:
    shape = tuple(shape) if isinstance(shape, (tuple, list, torch.Size)) else (int(shape),)
    n = 1
    for d in shape:
        n *= int(d)
    if _ZPOOL is not None and _ZPOOL[0].device == torch.device(device):
        off = _ZPOOL[1]
        if off + n <= _ZPOOL[0].numel():
            _ZPOOL[1] = off + ((n + 63) // 64) * 64          # keep 256-byte alignment
            return _ZPOOL[0][off:off + n].view(shape)
    return torch.zeros(shape, dtype=torch.float32, device=device)


# ---- scratch of the deterministic reductions (include/spegnet_hip.h "Deterministic reductions") ------------------------------
# Counters: 32-bit words that are zero before their first use and that every launch leaves zero again, so ONE persistent zeroed block
# per device serves every launch; slots rotate so that launches on different streams do not meet on a slot.  Partials: plain scratch.
_COUNTERS = {}
_COUNTER_WORDS = 16384


def red_counters(device, n: int) -> int:
    """device pointer to n zeroed counter words"""
    dev = torch.device(device)
    ent = _COUNTERS.get(dev.index)
    if ent is None:
        ent = _COUNTERS[dev.index] = [torch.zeros(_COUNTER_WORDS, dtype=torch.int32, device=dev), 0]
    n = (int(n) + 15) // 16 * 16
    off = ent[1] if ent[1] + n <= _COUNTER_WORDS else 0
    ent[1] = off + n
    return ent[0].data_ptr() + 4 * off


def red_scratch(device, nfloats: int) -> Tensor:
    return torch.empty(max(int(nfloats), 64), dtype=torch.float32, device=device)


def _red(x: Tensor, C: int, nimg: int):
    """(scratch tensor, its float count, counters pointer) for a column reduction over C channels of x's dtype"""
    lib, dt = _lib.load(), dcode(x)
    n = lib.spg_reduce_workspace_floats(dt, C, nimg)
    ws = red_scratch(x.device, n)
    return ws, n, red_counters(x.device, lib.spg_reduce_counters(dt, C, nimg))


# ---- optional live kernel timing (bench.py roofline leg): HIP events on the launch stream -----------------------
PROFILE = None  # set to a list -> instrumented ops append (name, bound, algorithmic work, start_event, end_event)


class _prof:
    """HIP events around one op when ops.PROFILE is a list.  bound: "mfma" (work = FLOPs) or "hbm" (work = algorithmic bytes: every
    operand of the op read / written once per pass the op makes over it)."""
    __slots__ = ("name", "bound", "work", "e0")

    def __init__(self, name: str, bound: str, work: float):
        self.name, self.bound, self.work, self.e0 = name, bound, work, None

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.e0 is not None and PROFILE is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROFILE.append((self.name, self.bound, float(self.work), self.e0, e1))
        return False


def _nb(*ts) -> int:
    """bytes of the given tensors (None skipped)"""
    return sum(t.numel() * t.element_size() for t in ts if t is not None)


# ---- GEMM family ---------------------------------------------------------------------------------------
# the GEMMs of a trunk block warm the next GEMM's weight matrix (gemm_nt(prefetch=...)); SPG_PREFETCH=0 for A/B runs
PREFETCH_WEIGHTS = os.environ.get("SPG_PREFETCH", "1") != "0"


def gemm_nt(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, act: int = ACT_NONE, residual: Optional[Tensor] = None,
            gelu_h: Optional[Tensor] = None, out: Optional[Tensor] = None, preact_out: Optional[Tensor] = None,
            conv: Optional[tuple] = None, prefetch: Optional[Tensor] = None) -> Tensor:
    """out[M,N] = epi(x[M,K] @ w[N,K]^T).  conv=(B,H,W,Ci): x is NHWC and the product is a 3x3/pad-1 convolution
    with w packed [N, 9*Ci].  prefetch: a tensor the NEXT launch reads cold (the following layer's weight matrix): this launch's
    workgroups request its cache lines on their way in (spg_prefetch_hint; no effect on the result)."""
    N, K = w.shape
    if conv is None:
        M = x.numel() // x.shape[-1]
        assert x.shape[-1] == K, (x.shape, w.shape)
        ldx = K
        B = H = W = Ci = 0
    else:
        B, H, W, Ci = conv
        M = B * H * W
        assert x.numel() == M * Ci and K == 9 * Ci, (x.shape, conv, w.shape)
        ldx = Ci
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    assert out.numel() == M * N and w.dtype == x.dtype
    tag = "bf16" if x.dtype == torch.bfloat16 else "f32"
    with _prof(f"gemm_nt<{tag},{'conv3x3' if conv else 'dense'}>", "mfma", 2.0 * M * N * K):
        if prefetch is not None and PREFETCH_WEIGHTS:
            _lib.call("spg_prefetch_hint", _p(prefetch), prefetch.numel() * prefetch.element_size())
        _lib.call("spg_gemm_nt", dcode(x), _p(_c(x)), _p(_c(w)), _p(_c(out)), _p(preact_out), _p(bias), _p(residual),
                  _p(gelu_h), M, N, K, ldx, N, act, 1 if conv else 0, B, H, W, Ci, cu_budget_now(), _stream())
    return out


def gemm_tn(dy: Tensor, x: Tensor, dw: Tensor, conv: Optional[tuple] = None, dbias: Optional[Tensor] = None) -> None:
    """dw[N,K] (f32) += dy[M,N]^T @ x[M,K]  (conv: x NHWC gathered as in gemm_nt); dbias[N] += colsum(dy) if given."""
    N = dy.shape[-1]
    M = dy.numel() // N
    if conv is None:
        K = x.shape[-1]
        assert x.numel() == M * K
        B = H = W = Ci = 0
        ldx = K
    else:
        B, H, W, Ci = conv
        K = 9 * Ci
        assert B * H * W == M and x.numel() == M * Ci
        ldx = Ci
    # dw may be a column slice of a wider gradient matrix (rows N, unit column stride): its row stride goes down as ldw
    assert dw.dtype == torch.float32 and dw.numel() == N * K and dw.is_cuda, (dw.shape, N, K)
    if dw.dim() == 2 and dw.shape == (N, K) and dw.stride(1) == 1:
        ldw = dw.stride(0)
    else:
        ldw = K
        _c(dw)
    assert ldw % 4 == 0 or ldw == K, ldw
    if conv is not None and x.dtype == torch.bfloat16 and ldw == K:
        # bf16 3x3 convolutions: the LDS-resident-tile kernel (csrc/conv_halo.hip) when it has an instance for the shape
        hb = _lib.load().spg_conv3x3_wgrad_workspace_bytes(dcode(x), B, H, W, Ci, N, cu_budget_now())
        if hb > 0:
            hws = torch.empty(hb, dtype=torch.uint8, device=x.device)
            with _prof("gemm_tn<bf16,conv3x3> (+reduce)", "mfma", 2.0 * M * N * K):
                _lib.call("spg_conv3x3_wgrad", dcode(x), _p(_c(dy)), _p(_c(x)), dw.data_ptr(), _p(dbias), _p(hws), hb, B, H, W, Ci, N, 0,
                          cu_budget_now(), _stream())
            return
    wsb = _lib.load().spg_gemm_tn_workspace_bytes(dcode(x), M, N, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb > 0 else None
    tag = "bf16" if x.dtype == torch.bfloat16 else "f32"
    with _prof(f"gemm_tn<{tag},{'conv3x3' if conv else 'dense'}> (+reduce)", "mfma", 2.0 * M * N * K):
        _lib.call("spg_gemm_tn", dcode(x), _p(_c(dy)), _p(_c(x)), dw.data_ptr(), _p(dbias), _p(ws), wsb, M, N, K, N, ldx, ldw,
                  1 if conv else 0, B, H, W, Ci, cu_budget_now(), _stream())


# ---- chained launches (DEV library only: csrc/dev/nt_chain_*.inc; a measured experiment, tools/chain_bench.py -- the product path never calls it) -----------------------------------------------------------------------
_CHAIN_ERR = {}


def chain_err_word(device) -> Tensor:
    """The device word spg_nt_chain sets when a dependency wait gave up (one per device, checked by chain_check())."""
    dev = torch.device(device)
    t = _CHAIN_ERR.get(dev.index)
    if t is None:
        t = _CHAIN_ERR[dev.index] = torch.zeros(1, dtype=torch.int32, device=dev)
    return t


def chain_check(device) -> None:
    """Host-side check (synchronises): raises if any chained launch on this device reported a timed-out wait."""
    t = _CHAIN_ERR.get(torch.device(device).index)
    if t is not None and int(t.item()) != 0:
        t.zero_()
        raise RuntimeError("spg_nt_chain: a dependency wait timed out (results of that launch are wrong)")


def gemm_chain(phases) -> list:
    """DEV library only (SPG_LIBRARY=spegnet_amd/libspegnet_hip_dev.so).  Dependent dense GEMMs over the same M rows in ONE persistent launch (spg_nt_chain).  phases: list of dicts with keys
    x (None = the previous phase's output), w [N,K], bias, act, residual, gelu_h, preact_out (c2), out (optional).  Returns the outputs.
    Bit-identical to running the phases as separate gemm_nt calls on the persistent pipelined kernel."""
    import ctypes
    n = len(phases)
    assert 1 <= n <= _lib.CHAIN_MAX_PHASES
    arr = (_lib.ChainPhase * n)()
    outs, keep = [], []
    x0 = phases[0]["x"]
    M = x0.numel() // x0.shape[-1]
    flops = 0.0
    for i, ph in enumerate(phases):
        x = ph.get("x")
        dep = x is None
        if dep:
            assert i > 0
            x = outs[-1]
        w = ph["w"]
        N, K = w.shape
        assert x.numel() == M * K and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16, (x.shape, w.shape)
        out = ph.get("out")
        if out is None:
            out = torch.empty((M, N), dtype=x.dtype, device=x.device)
        q = arr[i]
        q.kind, q.act, q.depends, q.M, q.N, q.K = _lib.CHAIN_GEMM, int(ph.get("act", ACT_NONE)), 1 if dep else 0, M, N, K
        q.x, q.w, q.c, q.c2 = _p(_c(x)), _p(_c(w)), _p(_c(out)), _p(ph.get("preact_out"))
        q.bias, q.residual, q.gelu_h = _p(ph.get("bias")), _p(ph.get("residual")), _p(ph.get("gelu_h"))
        outs.append(out)
        keep.append((x, w, out))
        flops += 2.0 * M * N * K
    lib = _lib.load()
    words = lib.spg_nt_chain_counter_words(n, ctypes.cast(arr, ctypes.c_void_p))
    cnt = zeros_f32((words,), x0.device)          # (from the step's zero pool when one is open: one fill per step)
    with _prof("gemm_nt<bf16,dense>", "mfma", flops):
        _lib.call("spg_nt_chain", SPG_BF16, n, ctypes.cast(arr, ctypes.c_void_p), cnt.data_ptr(), words, chain_err_word(x0.device).data_ptr(),
                  cu_budget_now(), _stream())
    return outs


# CUs the persistent GEMM grids are sized for (0 = all): a per-call argument of the C ABI; this thread-local only carries the caller's
# choice (the multi-GPU trainer lowers it around the graph segments that run beside a collective) down to the calls made inside it.
import contextlib
import threading

_TLS = threading.local()


def cu_budget_now() -> int:
    return getattr(_TLS, "cu_budget", 0)


@contextlib.contextmanager
def cu_budget(n: int):
    """with ops.cu_budget(240): ... -- GEMM launches inside size their grids for n CUs (restored on exit, also on exceptions)."""
    prev = cu_budget_now()
    _TLS.cu_budget = int(n)
    try:
        yield
    finally:
        _TLS.cu_budget = prev


def set_cu_budget(n: int) -> None:
    """Sets this thread's CU budget until changed again (prefer the cu_budget() context manager)."""
    _TLS.cu_budget = int(n)


TN_GROUP_MAX = 8


def gemm_tn_group(jobs, defer: Optional[list] = None) -> None:
    """jobs: list of (dy[M,N], x[M,K], dw[N,K] f32, dbias[N] f32 or None), all dense.  dw += dy^T x and dbias += colsum(dy) for every
    job; bf16 jobs sharing M go out as grouped, CU-balanced launches of up to 8 problems (spg_gemm_tn_group), anything else one by one.
    defer: a list -> the small slab-reduce kernel of each grouped launch is not issued; (descriptor, workspace) pairs are appended
    and the caller folds them later with gemm_tn_group_reduce(defer) (the gradients are complete only then)."""
    import ctypes
    groups = {}
    for dy, x, dw, db in jobs:
        N, K = dy.shape[-1], x.shape[-1]
        ok = (dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and N % 8 == 0 and K % 8 == 0 and dy.is_contiguous() and
              x.is_contiguous() and dw.is_contiguous() and dw.dtype == torch.float32 and dw.numel() == N * K)
        if ok:
            groups.setdefault(dy.numel() // N, []).append((dy, x, dw, db, N, K))
        else:
            gemm_tn(dy, x, dw, dbias=db)
    if not groups:
        return
    lib = _lib.load()
    wsb = lib.spg_gemm_tn_group_workspace_bytes()
    for M, group in groups.items():
        for i in range(0, len(group), TN_GROUP_MAX):
            part = group[i:i + TN_GROUP_MAX]
            n = len(part)
            ws = torch.empty(wsb, dtype=torch.uint8, device=part[0][0].device)
            desc = ctypes.create_string_buffer(lib.spg_gemm_tn_group_desc_bytes()) if defer is not None else None
            P, I = ctypes.c_void_p * n, ctypes.c_int * n
            Ns, Ks = I(*[j[4] for j in part]), I(*[j[5] for j in part])
            with _prof("gemm_tn_group<bf16> (one trunk block's wgrads)", "mfma", sum(2.0 * M * j[4] * j[5] for j in part)):
                _lib.call("spg_gemm_tn_group", SPG_BF16, n, P(*[_p(j[0]) for j in part]), P(*[_p(j[1]) for j in part]),
                          P(*[_p(j[2]) for j in part]), P(*[_p(j[3]) for j in part]), M, Ns, Ks, Ns, Ks, Ks, _p(ws), wsb,
                          ctypes.addressof(desc) if desc is not None else None, cu_budget_now(), _stream())
            if defer is not None:
                defer.append((desc, ws))


TN_BLOCKS_MAX = 16


def tn_blocks_count(jobs) -> int:
    """Number of 256 x 192 blocks of dW the jobs (same tuples as gemm_tn_group) make in spg_gemm_tn_blocks, or -1 when they are
    outside that kernel's domain (bf16, dense, one common M >= 256, every N and K a multiple of 192, at most 16 problems)."""
    import ctypes
    if not jobs or len(jobs) > TN_BLOCKS_MAX:
        return -1
    M = jobs[0][0].numel() // jobs[0][0].shape[-1]
    for dy, x, dw, db in jobs:
        N, K = dy.shape[-1], x.shape[-1]
        if not (dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.is_contiguous() and x.is_contiguous() and dw.is_contiguous()
                and dw.dtype == torch.float32 and dw.numel() == N * K and dy.numel() // N == M and x.numel() // K == M):
            return -1
    I = ctypes.c_int * len(jobs)
    return int(_lib.load().spg_gemm_tn_blocks_count(len(jobs), M, I(*[j[0].shape[-1] for j in jobs]), I(*[j[1].shape[-1] for j in jobs])))


def num_cus() -> int:
    """CUs the persistent grids are sized for under the calling thread's CU budget."""
    return int(_lib.load().spg_num_cus(cu_budget_now()))


def conv3x3_wgrad_direct(dy: Tensor, x: Tensor, gw: Tensor, dbias: Optional[Tensor], conv: tuple) -> bool:
    """gw [Co,Ci,3,3] f32 (the parameter's own gradient) += the 3x3 convolution's weight gradient, dbias += colsum(dy), by the halo-tile
    kernel writing the torch layout itself (no packed scratch, no zero fill, no unpack launch).  False: no instance for the shape (the
    caller goes through gemm_tn(conv=...) + unpack_conv3x3_grad)."""
    B, H, W, Ci = conv
    Co = dy.shape[-1]
    if not (x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and gw.dtype == torch.float32 and gw.is_contiguous() and gw.numel() == Co * Ci * 9):
        return False
    hb = _lib.load().spg_conv3x3_wgrad_workspace_bytes(dcode(x), B, H, W, Ci, Co, cu_budget_now())
    if hb <= 0:
        return False
    hws = torch.empty(hb, dtype=torch.uint8, device=x.device)
    with _prof("gemm_tn<bf16,conv3x3> (+reduce)", "mfma", 2.0 * B * H * W * Co * 9 * Ci):
        _lib.call("spg_conv3x3_wgrad", dcode(x), _p(_c(dy)), _p(_c(x)), gw.data_ptr(), _p(dbias), _p(hws), hb, B, H, W, Ci, Co, 1,
                  cu_budget_now(), _stream())
    return True


def gemm_tn_blocks(jobs, overwrite: bool = False, want_sq: bool = False) -> Optional[Tensor]:
    """The weight gradients of several trunk blocks in ONE launch: every workgroup owns a whole 256 x 192 block of some dw over all of M and
    adds it straight into dw / dbias (no slabs, no reduce kernel, deterministic).  jobs as in gemm_tn_group with tn_blocks_count(jobs) >= 1;
    more blocks than num_cus() run in rounds.  overwrite: dw is stored, not added to (it holds zeros and nothing else writes it this step; dbias is always
    added to).  want_sq: returns f32 [blocks], the sum of squares of what each block's owner wrote (for spg_sumsq_fold)."""
    import ctypes
    n = len(jobs)
    M = jobs[0][0].numel() // jobs[0][0].shape[-1]
    P, I = ctypes.c_void_p * n, ctypes.c_int * n
    Ns, Ks = I(*[j[0].shape[-1] for j in jobs]), I(*[j[1].shape[-1] for j in jobs])
    sq = torch.empty(tn_blocks_count(jobs), dtype=torch.float32, device=jobs[0][0].device) if want_sq else None
    with _prof("gemm_tn_blocks<bf16> (wgrads of several trunk blocks, whole 256x192 blocks)", "mfma",
               sum(2.0 * M * j[0].shape[-1] * j[1].shape[-1] for j in jobs)):
        _lib.call("spg_gemm_tn_blocks", SPG_BF16, n, P(*[_p(j[0]) for j in jobs]), P(*[_p(j[1]) for j in jobs]), P(*[_p(j[2]) for j in jobs]),
                  P(*[_p(j[3]) for j in jobs]), M, Ns, Ks, Ns, Ks, Ks, 1 if overwrite else 0, _p(sq), cu_budget_now(), _stream())
    return sq


def gemm_tn_group_reduce(deferred: list) -> None:
    """Folds the slabs of deferred grouped launches into their gradients, 6 launches per kernel; empties the list."""
    import ctypes
    for i in range(0, len(deferred), 6):
        part = deferred[i:i + 6]
        n = len(part)
        P = ctypes.c_void_p * n
        _lib.call("spg_gemm_tn_group_reduce_batch", n, P(*[ctypes.addressof(d) for d, _ in part]), P(*[_p(w) for _, w in part]), _stream())
    deferred.clear()


def pack_matrix(src: Tensor, dtype: torch.dtype, transpose: bool = False, out: Optional[Tensor] = None) -> Tensor:
    R, C = src.shape
    if out is None:
        out = torch.empty((C, R) if transpose else (R, C), dtype=dtype, device=src.device)
    _lib.call("spg_pack_matrix", dcode(out), _p(f32(src)), _p(out), R, C, 1 if transpose else 0, _stream())
    return out


def pack_batch(jobs: Tensor, njobs: int, total_tiles: int, dtype: torch.dtype) -> None:
    _lib.call("spg_pack_batch", SPG_BF16 if dtype == torch.bfloat16 else SPG_F32, _p(jobs), njobs, total_tiles, _stream())


def pack_conv3x3(src: Tensor, dtype: torch.dtype, fwd: Optional[Tensor] = None, dgrad: Optional[Tensor] = None):
    Co, Ci = src.shape[0], src.shape[1]
    if fwd is None:
        fwd = torch.empty((Co, 9 * Ci), dtype=dtype, device=src.device)
    if dgrad is None:
        dgrad = torch.empty((Ci, 9 * Co), dtype=dtype, device=src.device)
    _lib.call("spg_pack_conv3x3", dcode(fwd), _p(f32(src)), _p(fwd), _p(dgrad), Co, Ci, _stream())
    return fwd, dgrad


def unpack_conv3x3_grad(packed: Tensor, dst: Tensor) -> None:
    Co, Ci = dst.shape[0], dst.shape[1]
    _lib.call("spg_unpack_conv3x3_grad", _p(f32(packed)), _p(f32(dst)), Co, Ci, _stream())


# ---- trunk pieces --------------------------------------------------------------------------------------
def layernorm_fwd(x: Tensor, gamma: Tensor, beta: Tensor, eps: float):
    C = x.shape[-1]
    M = x.numel() // C
    y = torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    with _prof("layernorm_fwd", "hbm", _nb(x, y)):
        _lib.call("spg_layernorm_fwd", dcode(x), _p(_c(x)), _p(f32(gamma)), _p(f32(beta)), _p(y), _p(mean), _p(rstd), M, C, eps,
                  _stream())
    return y, mean, rstd


def layernorm_bwd(dy: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, dgamma: Tensor, dbeta: Tensor,
                  dres: Optional[Tensor] = None) -> Tensor:
    C = x.shape[-1]
    M = x.numel() // C
    dx = torch.empty_like(x)
    ws, n, cnt = _red(x, C, 0) if dgamma is not None else (None, 0, None)
    with _prof("layernorm_bwd (dx)" if dgamma is None else "layernorm_bwd (dx + params)", "hbm", _nb(x, dy, dres, dx) + (_nb(x, dy) if dgamma is not None else 0)):
        _lib.call("spg_layernorm_bwd", dcode(x), _p(_c(dy)), _p(_c(x)), _p(f32(gamma)), _p(mean), _p(rstd), _p(dres), _p(dx),
                  _p(f32(dgamma) if dgamma is not None else None), _p(f32(dbeta) if dbeta is not None else None), M, C, _p(ws), n, cnt, _stream())
    return dx


LN_BATCH_MAX = 48


def layernorm_param_grads_batch(jobs) -> None:
    """jobs: list of (dy[M,C], x[M,C], mean[M], rstd[M], dgamma[C] f32, dbeta[C] f32), all of one dtype: dgamma += sum dy * xhat,
    dbeta += sum dy for every job, 48 (column chunks of) LayerNorms per launch."""
    import ctypes
    if not jobs:
        return
    keep, sub = [], []                       # sub-jobs: (dy_ptr, x_ptr, mean_ptr, rstd_ptr, dg_ptr, db_ptr, M, cols, ld)
    es = jobs[0][1].element_size()
    step = 256 * (16 // es)                  # columns one 256-thread pass covers
    for dy, x, mean, rstd, dg, db in jobs:
        dy, x = _c(dy), _c(x)
        keep.append((dy, x))
        C = x.shape[-1]
        M = x.numel() // C
        for c0 in range(0, C, step):
            sub.append((_p(dy) + c0 * es, _p(x) + c0 * es, _p(mean), _p(rstd), _p(f32(dg)) + 4 * c0, _p(f32(db)) + 4 * c0, M, min(step, C - c0), C))
    dt = dcode(jobs[0][1])
    for i in range(0, len(sub), LN_BATCH_MAX):
        part = sub[i:i + LN_BATCH_MAX]
        n = len(part)
        P, I = ctypes.c_void_p * n, ctypes.c_int * n
        Cs = I(*[j[7] for j in part])
        nws = _lib.load().spg_layernorm_param_grads_batch_workspace_floats(n, Cs)
        dev = jobs[0][1].device
        ws = red_scratch(dev, nws)
        _lib.call("spg_layernorm_param_grads_batch", dt, n, *[P(*[j[k] for j in part]) for k in range(6)],
                  I(*[j[6] for j in part]), Cs, I(*[j[8] for j in part]), _p(ws), nws, red_counters(dev, n), _stream())


def attn_fwd(qkv: Tensor, bias_t: Tensor, B: int, H: int, W: int, heads: int, hd: int, ws: int,
             q_pooled: Optional[Tensor] = None):
    """qkv [B,H,W,3*heads*hd]; returns out [B,Hq,Wq,heads*hd], lse [B,Hq,Wq,heads]."""
    Hq, Wq = (H // 2, W // 2) if q_pooled is not None else (H, W)
    out = torch.empty((B, Hq, Wq, heads * hd), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, Hq, Wq, heads), dtype=torch.float32, device=qkv.device)
    # FLOPs of windowed attention: 4 * Lq * Lk * hd per head and window (QK^T + PV)
    wsz = ws if ws > 0 else max(H, W)
    nwin = ((H + wsz - 1) // wsz) * ((W + wsz - 1) // wsz) if ws > 0 else 1
    lk = (wsz * wsz) if ws > 0 else H * W
    lq = lk // 4 if q_pooled is not None else lk
    with _prof("attn_fwd", "mfma", 4.0 * B * nwin * heads * lq * lk * hd):
        _lib.call("spg_attn_fwd", dcode(qkv), _p(_c(qkv)), _p(q_pooled), _p(_c(bias_t)), _p(out), _p(lse), B, H, W, heads, hd, ws,
                  _stream())
    return out, lse


def attn_bwd(qkv: Tensor, bias_t: Tensor, out: Tensor, dout: Tensor, lse: Tensor, dbias_pad: Tensor, B: int, H: int, W: int,
             heads: int, hd: int, ws: int, q_pooled: Optional[Tensor] = None):
    """returns dqkv (q part left untouched when q_pooled is given) and dq_pooled (or None)."""
    dqkv = torch.empty_like(qkv)
    dqp = torch.empty_like(q_pooled) if q_pooled is not None else None
    delta = torch.empty_like(lse)
    wsz = ws if ws > 0 else max(H, W)
    nwin = ((H + wsz - 1) // wsz) * ((W + wsz - 1) // wsz) if ws > 0 else 1
    lk = (wsz * wsz) if ws > 0 else H * W
    lq = lk // 4 if q_pooled is not None else lk
    with _prof("attn_bwd (dq + dk/dv)", "mfma", 10.0 * B * nwin * heads * lq * lk * hd):
        _lib.call("spg_attn_bwd", dcode(qkv), _p(_c(qkv)), _p(q_pooled), _p(_c(bias_t)), _p(_c(out)), _p(_c(dout)), _p(lse),
                  _p(dqkv), _p(dqp), _p(f32(dbias_pad)), _p(delta), B, H, W, heads, hd, ws, _stream())
    return dqkv, dqp


def maxpool2_fwd(x: Tensor, B: int, H: int, W: int, C: int, ldc: int, c0: int):
    y = torch.empty((B, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
    idx = torch.empty((B, H // 2, W // 2, C), dtype=torch.uint8, device=x.device)
    _lib.call("spg_maxpool2_fwd", dcode(x), _p(_c(x)), _p(y), _p(idx), B, H, W, C, ldc, c0, _stream())
    return y, idx


def maxpool2_bwd(dy: Tensor, idx: Tensor, dx: Tensor, B: int, H: int, W: int, C: int, ldc: int, c0: int) -> None:
    _lib.call("spg_maxpool2_bwd", dcode(dy), _p(_c(dy)), _p(idx), _p(dx), B, H, W, C, ldc, c0, _stream())


def patch_im2col(img: Tensor, dtype: torch.dtype, kpad: int) -> Tensor:
    B, _, H, W = img.shape
    cols = torch.empty((B * (H // 4) * (W // 4), kpad), dtype=dtype, device=img.device)
    _lib.call("spg_patch_im2col", dcode(cols), _p(f32(img)), _p(cols), B, H, W, kpad, _stream())
    return cols


def preprocess_image(img_u8_hwc: Tensor, size, mean, std) -> Tensor:
    """uint8 [H,W,3] on the device -> f32 [3,OH,OW]: /255, antialiased bilinear resize, (v - mean) / std (one kernel).
    Mirrors CODImageProcessor.process_image of the reference (utils/image_processor.py:118-131)."""
    import ctypes
    assert img_u8_hwc.dtype == torch.uint8 and img_u8_hwc.dim() == 3 and img_u8_hwc.shape[2] == 3, img_u8_hwc.shape
    OH, OW = (size, size) if isinstance(size, int) else size
    H, W, _ = img_u8_hwc.shape
    out = torch.empty((3, OH, OW), dtype=torch.float32, device=img_u8_hwc.device)
    F3 = ctypes.c_float * 3
    _lib.call("spg_preprocess_image", _p(_c(img_u8_hwc)), _p(out), H, W, OH, OW, F3(*[float(v) for v in mean]), F3(*[float(v) for v in std]),
              _stream())
    return out


def preprocess_batch(base_u8: Tensor, offs, sizes, size, mean, std) -> Tensor:
    """Batched device input pipeline: uint8 HWC images of different sizes packed in `base_u8` (image i at byte offs[i], sizes[i] =
    (H, W)) -> f32 [B,3,OH,OW] (/255, antialiased bilinear resize, normalise), one launch per 64 images."""
    import ctypes
    assert base_u8.dtype == torch.uint8 and base_u8.is_cuda
    OH, OW = (size, size) if isinstance(size, int) else size
    B = len(sizes)
    out = torch.empty((B, 3, OH, OW), dtype=torch.float32, device=base_u8.device)
    F3 = ctypes.c_float * 3
    m3, s3 = F3(*[float(v) for v in mean]), F3(*[float(v) for v in std])
    for i in range(0, B, 64):
        n = min(64, B - i)
        L, I = ctypes.c_long * n, ctypes.c_int * n
        _lib.call("spg_preprocess_batch", _p(base_u8), L(*[int(o) for o in offs[i:i + n]]), I(*[int(s[0]) for s in sizes[i:i + n]]),
                  I(*[int(s[1]) for s in sizes[i:i + n]]), out[i:].data_ptr(), n, OH, OW, m3, s3, _stream())
    return out


# ---- reductions / elementwise ---------------------------------------------------------------------------
def colsum(x: Tensor, out: Tensor, accumulate: bool = True) -> None:
    """out[c] (+)= sum_m x[m][c] (deterministic)"""
    C = x.shape[-1]
    ws, n, cnt = _red(x, C, 0)
    _lib.call("spg_colsum", dcode(x), _p(_c(x)), _p(f32(out)), x.numel() // C, C, C, 1 if accumulate else 0, _p(ws), n, cnt, _stream())


def gap_sum(x: Tensor, B: int, HW: int, C: int) -> Tensor:
    out = torch.empty((B, C), dtype=torch.float32, device=x.device)
    ws, n, cnt = _red(x, C, B)
    _lib.call("spg_gap_sum", dcode(x), _p(_c(x)), _p(out), B, HW, C, _p(ws), n, cnt, _stream())
    return out


def chan_prod_sum(a: Tensor, b: Tensor, B: int, HW: int, C: int) -> Tensor:
    out = torch.empty((B, C), dtype=torch.float32, device=a.device)
    ws, n, cnt = _red(a, C, B)
    _lib.call("spg_chan_prod_sum", dcode(a), _p(_c(a)), _p(_c(b)), _p(out), B, HW, C, _p(ws), n, cnt, _stream())
    return out


def add(a: Tensor, b: Tensor, out: Optional[Tensor] = None) -> Tensor:
    if out is None:
        out = torch.empty_like(a)
    _lib.call("spg_add", dcode(a), _p(_c(a)), _p(_c(b)), _p(out), a.numel(), _stream())
    return out


def copy_channels(x: Tensor, y: Tensor, M: int, C: int, ldx: int, cx0: int, ldy: int, cy0: int, accumulate: bool = False):
    _lib.call("spg_copy_channels", dcode(x), _p(x), _p(y), M, C, ldx, cx0, ldy, cy0, 1 if accumulate else 0, _stream())


# ---- BatchNorm ------------------------------------------------------------------------------------------
def bn_stats(x: Tensor, C: int) -> Tensor:
    stats = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    ws, n, cnt = _red(x, C, 0)
    _lib.call("spg_bn_stats", dcode(x), _p(_c(x)), _p(stats), x.numel() // C, C, _p(ws), n, cnt, _stream())
    return stats


def bn_stats_finalize(x: Tensor, C: int, gamma: Tensor, beta: Tensor, rmean: Optional[Tensor], rvar: Optional[Tensor],
                      nbt: Optional[Tensor], eps: float = 1e-5, momentum: float = 0.1):
    """Training-mode BatchNorm statistics in ONE launch: deterministic batch sums, and the workgroup that finishes a channel slab
    also writes scale/shift, mean/invstd, the running statistics and num_batches_tracked += 1.  Returns (scale_shift, mean_invstd)."""
    M = x.numel() // C
    buf = torch.empty(6 * C, dtype=torch.float32, device=x.device)
    stats, ss, mi = buf[:2 * C], buf[2 * C:4 * C], buf[4 * C:]
    ws, n, cnt = _red(x, C, 0)
    if nbt is not None:
        assert nbt.dtype == torch.int64
    with _prof("bn_stats_finalize", "hbm", _nb(x)):
        _lib.call("spg_bn_stats_finalize", dcode(x), _p(_c(x)), _p(stats), _p(f32(gamma)), _p(f32(beta)), _p(rmean), _p(rvar), _p(nbt),
                  _p(ss), _p(mi), M, C, eps, momentum, _p(ws), n, cnt, _stream())
    return ss, mi



def conv3x3_stats_rows(x: Tensor, B: int, H: int, W: int, Ci: int, Co: int) -> int:
    """rows of the partial-statistics matrix conv3x3_fwd_stats writes; 0 = no fused instance for this problem (fp32 parity mode, channel
    counts that are no multiples of 64): the caller then runs gemm_nt(conv=...) and bn_stats_finalize."""
    return int(_lib.load().spg_conv3x3_stats_rows(dcode(x), B, H, W, Ci, Co, cu_budget_now()))


def conv3x3_fwd_stats(x: Tensor, w: Tensor, bias: Optional[Tensor], B: int, H: int, W: int, Ci: int, rows: int) -> Tuple[Tensor, Tensor]:
    """3x3 convolution (w packed [Co, 9*Ci]) + per-tile BatchNorm partial statistics in ONE launch.  Returns (out [B*H*W, Co], part f32
    [rows, 2*Co])."""
    Co = w.shape[0]
    M = B * H * W
    out = torch.empty((M, Co), dtype=x.dtype, device=x.device)
    part = torch.empty((rows, 2 * Co), dtype=torch.float32, device=x.device)
    with _prof("gemm_nt<bf16,conv3x3>", "mfma", 2.0 * M * Co * 9 * Ci):
        _lib.call("spg_conv3x3_fwd_stats", dcode(x), _p(_c(x)), _p(_c(w)), _p(out), _p(bias), _p(part), rows, B, H, W, Ci, Co,
                  cu_budget_now(), _stream())
    return out, part


def bn_stats_finalize_part(part: Tensor, M: int, C: int, gamma: Tensor, beta: Tensor, rmean: Optional[Tensor], rvar: Optional[Tensor],
                           nbt: Optional[Tensor], eps: float = 1e-5, momentum: float = 0.1):
    """bn_stats_finalize from the partial rows a producer's epilogue wrote (sum | sum of squares): one launch over [R, 2C] floats instead
    of a pass over the [M, C] tensor.  Returns (scale_shift, mean_invstd)."""
    R = part.shape[0]
    assert part.dtype == torch.float32 and part.shape[1] == 2 * C and part.is_contiguous()
    buf = torch.empty(6 * C, dtype=torch.float32, device=part.device)
    stats, ss, mi = buf[:2 * C], buf[2 * C:4 * C], buf[4 * C:]
    ws, n, cnt = _red(part, C, 0)
    with _prof("bn_stats_finalize (from conv partials)", "hbm", part.numel() * 4):
        _lib.call("spg_bn_stats_finalize_part", _p(part), R, _p(stats), _p(f32(gamma)), _p(f32(beta)), _p(rmean), _p(rvar), _p(nbt),
                  _p(ss), _p(mi), M, C, eps, momentum, _p(ws), n, cnt, _stream())
    return ss, mi

def bn_finalize(stats: Optional[Tensor], gamma: Tensor, beta: Tensor, rmean: Optional[Tensor], rvar: Optional[Tensor], M: int,
                training: bool, eps: float = 1e-5, momentum: float = 0.1):
    C = gamma.numel()
    ss = torch.empty(2 * C, dtype=torch.float32, device=gamma.device)
    mi = torch.empty(2 * C, dtype=torch.float32, device=gamma.device)
    _lib.call("spg_bn_finalize", _p(stats), _p(f32(gamma)), _p(f32(beta)), _p(rmean), _p(rvar), _p(ss), _p(mi), M, C, eps, momentum,
              1 if training else 0, _stream())
    return ss, mi


def bn_apply(x: Tensor, ss: Tensor, C: int, relu: bool, out: Optional[Tensor] = None) -> Tensor:
    if out is None:
        out = torch.empty_like(x)
    with _prof("bn_apply", "hbm", _nb(x, out)):
        _lib.call("spg_bn_apply", dcode(x), _p(_c(x)), _p(ss), _p(out), x.numel() // C, C, 1 if relu else 0, _stream())
    return out


def bn_bwd(dy: Tensor, x: Tensor, ss: Tensor, mi: Tensor, gamma: Tensor, dgamma: Tensor, dbeta: Tensor, C: int, relu: bool) -> Tensor:
    M = x.numel() // C
    sums = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    ws, n, cnt = _red(x, C, 0)
    dx = torch.empty_like(x)
    with _prof("bn_bwd (reduce + apply)", "hbm", 2 * _nb(x, dy) + _nb(dx)):
        _lib.call("spg_bn_bwd_reduce", dcode(x), _p(_c(dy)), _p(_c(x)), _p(ss), _p(mi), _p(sums), M, C, 1 if relu else 0, _p(ws), n, cnt, _stream())
        _lib.call("spg_bn_bwd_apply", dcode(x), _p(dy), _p(x), _p(ss), _p(mi), _p(f32(gamma)), _p(sums), _p(dx), _p(f32(dgamma)),
                  _p(f32(dbeta)), M, C, 1 if relu else 0, _stream())
    return dx


# ---- CFI / PED pieces -------------------------------------------------------------------------------------
def upsample_into(x: Tensor, y: Tensor, B: int, h: int, w: int, C: int, H: int, W: int, ldy: int, c0: int) -> None:
    _lib.call("spg_upsample_bilinear", dcode(x), _p(_c(x)), _p(y), B, h, w, C, H, W, ldy, c0, _stream())


def upsample_bwd(dy: Tensor, dx: Tensor, B: int, h: int, w: int, C: int, H: int, W: int, ldy: int, c0: int, accumulate: bool = False):
    _lib.call("spg_upsample_bilinear_bwd", dcode(dy), _p(_c(dy)), _p(dx), B, h, w, C, H, W, ldy, c0, 1 if accumulate else 0, _stream())


def se_fc(gap: Tensor, w1: Tensor, w2: Tensor, in_scale: float = 1.0):
    """SE block FCs on in_scale * gap (pass gap = per-image column sums and in_scale = 1 / HW: no separate mean kernel)."""
    B, C = gap.shape
    R = w1.shape[0]
    hidden = torch.empty((B, R), dtype=torch.float32, device=gap.device)
    scale = torch.empty((B, C), dtype=torch.float32, device=gap.device)
    _lib.call("spg_se_fc", _p(f32(gap)), _p(f32(w1)), _p(f32(w2)), _p(hidden), _p(scale), B, C, R, float(in_scale), _stream())
    return hidden, scale


def se_fc_bwd(gap, w1, w2, hidden, scale, dscale, dw1, dw2, in_scale: float = 1.0) -> Tensor:
    B, C = gap.shape
    R = w1.shape[0]
    dgap = torch.empty_like(gap)
    n = B * (C + R)
    ws = red_scratch(gap.device, n)
    _lib.call("spg_se_fc_bwd", _p(gap), _p(f32(w1)), _p(f32(w2)), _p(hidden), _p(scale), _p(f32(dscale)), _p(dgap), _p(f32(dw1)),
              _p(f32(dw2)), B, C, R, float(in_scale), _p(ws), n, red_counters(gap.device, 1), _stream())
    return dgap


def pack_cols2(a: Tensor, b: Tensor, Kp: int, dtype: torch.dtype) -> Tensor:
    """[R, Kp] (dtype) = [a | b | 0] from two fp32 [R, na] / [R, nb] matrices, one launch."""
    R, na, nb = a.shape[0], a.shape[1], b.shape[1]
    out = torch.empty((R, Kp), dtype=dtype, device=a.device)
    _lib.call("spg_pack_cols2", dcode(out), _p(f32(a)), na, _p(f32(b)), nb, _p(out), R, Kp, _stream())
    return out


def add_cols_batch(jobs) -> None:
    """jobs: up to 4 (dst [R, C] fp32 contiguous view, src [R, >= C] fp32 with row stride src.stride(0)): dst += src[:, :C], one launch."""
    import ctypes
    n = len(jobs)
    P, I = ctypes.c_void_p * n, ctypes.c_int * n
    for d, s_ in jobs:
        assert d.dtype == torch.float32 and s_.dtype == torch.float32 and d.is_contiguous() and s_.stride(1) == 1 and d.shape[0] == s_.shape[0]
    _lib.call("spg_add_cols_batch", n, P(*[_p(d) for d, _ in jobs]), P(*[_p(s_) for _, s_ in jobs]), I(*[d.shape[0] for d, _ in jobs]),
              I(*[d.shape[1] for d, _ in jobs]), I(*[d.shape[1] for d, _ in jobs]), I(*[s_.stride(0) for _, s_ in jobs]), _stream())


def chan_scale(x: Tensor, scale: Tensor, B: int, HW: int, C: int) -> Tensor:
    y = torch.empty_like(x)
    _lib.call("spg_chan_scale", dcode(x), _p(_c(x)), _p(scale), _p(y), B, HW, C, _stream())
    return y


def chan_scale_bwd(dy: Tensor, scale: Tensor, dgap: Tensor, B: int, HW: int, C: int) -> Tensor:
    dx = torch.empty_like(dy)
    _lib.call("spg_chan_scale_bwd", dcode(dy), _p(_c(dy)), _p(scale), _p(dgap), _p(dx), B, HW, C, _stream())
    return dx


def dwconv3x3(x: Tensor, w: Tensor, B: int, H: int, W: int, C: int, dil: int, flip: bool = False) -> Tensor:
    y = torch.empty_like(x)
    _lib.call("spg_dwconv3x3", dcode(x), _p(_c(x)), _p(f32(w)), _p(y), B, H, W, C, dil, 1 if flip else 0, _stream())
    return y


def dwconv3x3_wgrad(dy: Tensor, x: Tensor, dw: Tensor, B: int, H: int, W: int, C: int, dil: int) -> None:
    n = 64 * 9 * C
    ws = red_scratch(x.device, n)
    _lib.call("spg_dwconv3x3_wgrad", dcode(x), _p(_c(dy)), _p(_c(x)), _p(f32(dw)), B, H, W, C, dil, _p(ws), n, red_counters(x.device, 1), _stream())


def easpp_fuse(br, glob: Tensor, w: Tensor, B: int, HW: int, C: int) -> Tensor:
    y = torch.empty_like(br[0])
    _lib.call("spg_easpp_fuse", dcode(y), _p(br[0]), _p(br[1]), _p(br[2]), _p(br[3]), _p(f32(glob)), _p(f32(w)), _p(y), B, HW, C,
              _stream())
    return y


def easpp_fuse_bwd(dy: Tensor, br, glob: Tensor, w: Tensor, dw: Tensor, B: int, HW: int, C: int):
    d = [torch.empty_like(b) for b in br]
    dglob = torch.empty_like(glob)
    n = 32 * B * 6 * C
    ws = red_scratch(dy.device, n)
    _lib.call("spg_easpp_fuse_bwd", dcode(dy), _p(_c(dy)), _p(br[0]), _p(br[1]), _p(br[2]), _p(br[3]), _p(glob), _p(f32(w)),
              _p(d[0]), _p(d[1]), _p(d[2]), _p(d[3]), _p(dglob), _p(f32(dw)), B, HW, C, _p(ws), n, red_counters(dy.device, 1), _stream())
    return d, dglob


def head1x1(x: Tensor, w: Tensor, b: Tensor, M: int, C: int) -> Tensor:
    y = torch.empty(M, dtype=x.dtype, device=x.device)
    _lib.call("spg_head1x1", dcode(x), _p(_c(x)), _p(f32(w)), _p(f32(b)), _p(y), M, C, _stream())
    return y


def head1x1_bwd(dy: Tensor, x: Tensor, w: Tensor, dx: Tensor, dw: Tensor, db: Tensor, M: int, C: int, accumulate: bool) -> None:
    n = _lib.load().spg_head1x1_bwd_workspace_floats(C)
    ws = red_scratch(x.device, n)
    _lib.call("spg_head1x1_bwd", dcode(x), _p(_c(dy)), _p(_c(x)), _p(f32(w)), _p(dx), _p(f32(dw)), _p(f32(db)), M, C,
              1 if accumulate else 0, _p(ws), n, red_counters(x.device, 1), _stream())


# ---- fused CFI / EFE / PED element kernels (csrc/head.hip) ---------------------------------------------------------------------
def bn_apply_head(x: Tensor, ss: Tensor, w: Tensor, b: Tensor, C: int, relu: bool = True, write_y: bool = True):
    """y = relu(x*scale+shift) (None when write_y is False: never stored) and pred[m] = w . y[m] + b, one pass over x."""
    M = x.numel() // C
    y = torch.empty_like(x) if write_y else None
    pred = torch.empty(M, dtype=x.dtype, device=x.device)
    with _prof("bn_apply_head" + ("" if write_y else " (y not stored)"), "hbm", _nb(x, y, pred)):
        _lib.call("spg_bn_apply_head", dcode(x), _p(_c(x)), _p(ss), _p(f32(w)), _p(f32(b)), _p(y), _p(pred), M, C, 1 if relu else 0, _stream())
    return y, pred


def ped_gather(x: Tensor, x_ss: Optional[Tensor], B: int, hx: int, wx: int, Cx: int, edge: Optional[Tensor], he: int, we: int, Ce: int) -> Tensor:
    """cat[up2(act(x)), up(edge)] as [B*2hx*2wx, Cx+Ce]; act = relu(x*scale+shift) when x_ss is given."""
    H, W = 2 * hx, 2 * wx
    y = torch.empty((B * H * W, Cx + Ce), dtype=x.dtype, device=x.device)
    with _prof("ped_gather (bn+relu+up2+concat)", "hbm", _nb(x, edge, y)):
        _lib.call("spg_ped_gather", dcode(x), _p(_c(x)), _p(x_ss), hx, wx, Cx, _p(edge), he, we, Ce, _p(y), B, H, W, _stream())
    return y


def ped_gather_bwd(dy: Tensor, dx: Tensor, B: int, h: int, w: int, C: int, H: int, W: int, ldy: int, c0: int, accumulate: bool = False) -> None:
    es = dy.element_size()
    with _prof("ped_gather_bwd", "hbm", B * H * W * C * es + B * h * w * C * es * (2 if accumulate else 1)):
        _lib.call("spg_ped_gather_bwd", dcode(dy), _p(_c(dy)), _p(dx), B, h, w, C, H, W, ldy, c0, 1 if accumulate else 0, _stream())


def bn_bwd_head(dnext: Optional[Tensor], x: Tensor, dpred: Tensor, head_w: Tensor, ss: Tensor, mi: Tensor, gamma: Tensor, dgamma: Tensor,
                dbeta: Tensor, dhead_w: Tensor, dhead_b: Tensor, C: int) -> Tensor:
    """BatchNorm backward (ReLU recomputed) whose incoming gradient is dnext + dpred (x) head_w, plus the head's parameter gradients."""
    M = x.numel() // C
    lib, dt = _lib.load(), dcode(x)
    n = lib.spg_bn_bwd_head_workspace_floats(dt, C)
    ws = red_scratch(x.device, n)
    sums = torch.empty(3 * C + 1, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    with _prof("bn_bwd_head (reduce + apply)", "hbm", 2 * _nb(x, dnext, dpred) + _nb(dx)):
        _lib.call("spg_bn_bwd_head", dt, _p(dnext), _p(_c(x)), _p(_c(dpred)), _p(f32(head_w)), _p(ss), _p(mi), _p(f32(gamma)), _p(sums), _p(dx),
                  _p(f32(dgamma)), _p(f32(dbeta)), _p(f32(dhead_w)), _p(f32(dhead_b)), M, C, _p(ws), n,
                  red_counters(x.device, lib.spg_bn_bwd_head_counters(dt, C)), _stream())
    return dx


def cfi_combine(y2: Tensor, y3: Tensor, y4: Tensor, B: int, H: int, W: int, h3: int, w3: int, h4: int, w4: int, C: int) -> Tensor:
    """y2 + up(y3) + up(y4): the CFI fusion conv evaluated per source resolution (no 2016-channel concat)."""
    out = torch.empty_like(y2)
    with _prof("cfi_combine", "hbm", _nb(y2, y3, y4, out)):
        _lib.call("spg_cfi_combine", dcode(y2), _p(_c(y2)), _p(_c(y3)), _p(_c(y4)), _p(out), B, H, W, h3, w3, h4, w4, C, _stream())
    return out


# ---- e-ASPP middle, branch-batched (csrc/easpp.hip) ---------------------------------------------------------------------------
def _p4(ts):
    import ctypes
    return (ctypes.c_void_p * 4)(*[_p(t) for t in ts])


def _i4(vs):
    import ctypes
    return (ctypes.c_int * 4)(*[int(v) for v in vs])


def dwconv4(x: Tensor, w4, dil4, B: int, H: int, W: int, C: int) -> Tensor:
    """the four dilated depth-wise 3x3 branches in one launch -> dcat [B*H*W, 4C] (branch-major concat order)"""
    dcat = torch.empty((B * H * W, 4 * C), dtype=x.dtype, device=x.device)
    with _prof("dwconv4 (4 dilated depth-wise 3x3)", "hbm", _nb(x, dcat)):
        _lib.call("spg_dwconv4", dcode(x), _p(_c(x)), _p4([f32(w) for w in w4]), _i4(dil4), _p(dcat), B, H, W, C, _stream())
    return dcat


def dwconv4_dgrad(dy: Tensor, w4, dil4, gadd: Optional[Tensor], B: int, H: int, W: int, C: int) -> Tensor:
    dx = torch.empty((B * H * W, C), dtype=dy.dtype, device=dy.device)
    with _prof("dwconv4_dgrad (+GAP adjoint)", "hbm", _nb(dy, dx)):
        _lib.call("spg_dwconv4_dgrad", dcode(dy), _p(_c(dy)), _p4([f32(w) for w in w4]), _i4(dil4), _p(gadd), _p(dx), B, H, W, C, _stream())
    return dx


def dwconv4_wgrad(dy: Tensor, x: Tensor, dil4, dw4, B: int, H: int, W: int, C: int) -> None:
    n = 4 * 64 * 9 * C
    ws = red_scratch(x.device, n)
    with _prof("dwconv4_wgrad", "hbm", _nb(dy, x)):
        _lib.call("spg_dwconv4_wgrad", dcode(x), _p(_c(dy)), _p(_c(x)), _i4(dil4), _p4([f32(w) for w in dw4]), B, H, W, C, _p(ws), n,
                  red_counters(x.device, 4), _stream())


def bn_stats_finalize4(x: Tensor, C: int, gamma4, beta4, rmean4, rvar4, nbt4, eps: float = 1e-5, momentum: float = 0.1):
    """batch statistics + finalize of FOUR BatchNorms of C/4 channels each over one [M, C] tensor, one launch"""
    M = x.numel() // C
    buf = torch.empty(6 * C, dtype=torch.float32, device=x.device)
    stats, ss, mi = buf[:2 * C], buf[2 * C:4 * C], buf[4 * C:]
    ws, n, cnt = _red(x, C, 0)
    with _prof("bn_stats_finalize", "hbm", _nb(x)):
        _lib.call("spg_bn_stats_finalize4", dcode(x), _p(_c(x)), _p(stats), _p4(gamma4), _p4(beta4), _p4(rmean4) if rmean4 else None,
                  _p4(rvar4) if rvar4 else None, _p4(nbt4) if nbt4 else None, _p(ss), _p(mi), M, C, eps, momentum, _p(ws), n, cnt, _stream())
    return ss, mi


def easpp_fuse_bn(dcat: Tensor, ss: Tensor, glob: Tensor, w: Tensor, B: int, HW: int, C: int) -> Tensor:
    y = torch.empty((B * HW, C), dtype=dcat.dtype, device=dcat.device)
    with _prof("easpp_fuse_bn (branch BN+ReLU + grouped 1x1)", "hbm", _nb(dcat, y)):
        _lib.call("spg_easpp_fuse_bn", dcode(dcat), _p(_c(dcat)), _p(ss), _p(f32(glob)), _p(f32(w)), _p(y), B, HW, C, _stream())
    return y


def easpp_fuse_bn_bwd(dfu: Tensor, dcat: Tensor, w: Tensor, ss: Tensor, mi: Tensor, gamma4, dgamma4, dbeta4, dw: Tensor, B: int, HW: int, C: int) -> Tensor:
    lib, dt = _lib.load(), dcode(dcat)
    n = lib.spg_easpp_fuse_bn_bwd_workspace_floats(dt, C)
    ws = red_scratch(dcat.device, n)
    sums = torch.empty(12 * C, dtype=torch.float32, device=dcat.device)
    ddcat = torch.empty_like(dcat)
    with _prof("easpp_fuse_bn_bwd (reduce + apply)", "hbm", 2 * _nb(dcat, dfu) + _nb(ddcat)):
        _lib.call("spg_easpp_fuse_bn_bwd", dt, _p(_c(dfu)), _p(_c(dcat)), _p(f32(w)), _p(ss), _p(mi), _p4(gamma4), _p4(dgamma4), _p4(dbeta4),
                  _p(sums), _p(ddcat), _p(f32(dw)), B, HW, C, _p(ws), n, red_counters(dcat.device, lib.spg_easpp_fuse_bn_bwd_counters(dt, C)), _stream())
    return ddcat


def easpp_global_fwd(gsum: Tensor, Wg: Tensor, gamma: Tensor, beta: Tensor, rmean, rvar, nbt, B: int, C: int, HW: int, training: bool,
                     eps: float = 1e-5, momentum: float = 0.1):
    """returns (gm, gl0, glob, ss, mi): mean, pre-BN 1x1 output, activated global features [B,C], BN scale/shift and mean/invstd"""
    buf = torch.empty(3 * B * C + 4 * C, dtype=torch.float32, device=gsum.device)
    gm, gl0, glob = buf[:B * C].view(B, C), buf[B * C:2 * B * C].view(B, C), buf[2 * B * C:3 * B * C].view(B, C)
    ss, mi = buf[3 * B * C:3 * B * C + 2 * C], buf[3 * B * C + 2 * C:]
    _lib.call("spg_easpp_global_fwd", _p(f32(gsum)), _p(f32(Wg)), _p(f32(gamma)), _p(f32(beta)), _p(rmean), _p(rvar), _p(nbt), _p(gm), _p(gl0),
              _p(glob), _p(ss), _p(mi), B, C, HW, eps, momentum, 1 if training else 0, _stream())
    return gm, gl0, glob, ss, mi


def easpp_global_bwd(S: Tensor, glob: Tensor, gl0: Tensor, gm: Tensor, wf: Tensor, Wg: Tensor, gamma: Tensor, mi: Tensor, dwf: Tensor, dWg: Tensor,
                     dgamma: Tensor, dbeta: Tensor, B: int, C: int, HW: int, training: bool = True) -> Tensor:
    gadd = torch.empty((B, C), dtype=torch.float32, device=S.device)
    _lib.call("spg_easpp_global_bwd", _p(f32(S)), _p(glob), _p(gl0), _p(gm), _p(f32(wf)), _p(f32(Wg)), _p(f32(gamma)), _p(mi), _p(f32(dwf)),
              _p(f32(dWg)), _p(f32(dgamma)), _p(f32(dbeta)), _p(gadd), B, C, HW, 1 if training else 0, _stream())
    return gadd
