"""GPU parity tests of every C-ABI kernel against plain PyTorch fp32 references of the same op.
fp32 storage must agree to ~1e-5 (exact-f32 MFMA, only summation order differs); bf16 storage to bf16 rounding."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def tol(dt, f32=2e-5, bf16=2.5e-2):
    return f32 if dt == torch.float32 else bf16


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def check(a, b, t, what=""):
    e = rel(a, b)
    if not (e < t):
        d = (a.double().cpu() - b.double().cpu()).abs()
        idx = torch.nonzero(d > t * b.abs().max().cpu(), as_tuple=False)[:8].tolist()
        raise AssertionError(f"{what}: rel err {e:.3e} >= {t:.1e}; shape {tuple(a.shape)}; first bad idx {idx}; "
                             f"got {a.flatten()[:6].tolist()} want {b.flatten()[:6].tolist()}")


@pytest.fixture(scope="module")
def ops():
    from spegnet_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


# ------------------------------------------------------------------------------------------- GEMM NT
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 144), (77, 432, 144), (1000, 576, 2304), (64, 16, 32),
                                   # > 256 tiles with ragged M / N edges: persistent workgroups walk several tiles each
                                   (4645, 2304, 576), (5000, 1000, 200), (33000, 144, 576)])
def test_gemm_nt_dense(ops, dt, M, N, K):
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    xq, wq = x.to(dt), w.to(dt)
    ref = xq.float() @ wq.float().t() + b
    out = ops.gemm_nt(xq, wq, bias=b)
    check(out.float(), ref, tol(dt), "bias")
    res = rnd(M, N, seed=4).to(dt)
    pre = torch.empty(M, N, dtype=dt, device="cuda")
    out = ops.gemm_nt(xq, wq, bias=b, act=ops.ACT_GELU, residual=res, preact_out=pre)
    check(pre.float(), ref, tol(dt), "preact")
    check(out.float(), F.gelu(ref) + res.float(), tol(dt), "gelu+res")
    h = rnd(M, N, seed=5).to(dt)
    out = ops.gemm_nt(xq, wq, gelu_h=h)
    hf = h.float().requires_grad_(True)
    F.gelu(hf).sum().backward()
    check(out.float(), (xq.float() @ wq.float().t()) * hf.grad, tol(dt), "gelu'")


@pytest.mark.parametrize("M,N,K,what", [(4608, 2304, 576, "fc1: 4-wave kernel"), (4608, 576, 2304, "fc2: persistent kernel, 108 tiles"),
                                         (4608, 1728, 576, "qkv"), (1152, 4608, 1152, "stage 4"), (300, 200, 144, "ragged"),
                                         (129, 64, 64, "one K step"), (1000, 136, 200, "K tail")])
@pytest.mark.parametrize("nbytes", [0, 100, 663552, 21233664])
def test_gemm_nt_prefetch_hint_changes_nothing(ops, M, N, K, what, nbytes):
    """spg_prefetch_hint makes the next spg_gemm_nt's workgroups request the cache lines of another buffer (the following layer's weights)
    through LDS-DMA into LDS they overwrite before reading: results must be BIT-identical with and without it -- for both kernel families,
    ragged tiles, a K tail (whose out-of-range pieces must still overwrite the scratch), hints smaller than a line, of 0.66 MB (a 576 x 576
    matrix) and of 21 MB (more than one pass of a small grid) -- and the hint must be consumed by exactly one launch."""
    dt = torch.bfloat16
    x, w = rnd(M, K, seed=1).to(dt), rnd(N, K, seed=2, scale=K ** -0.5).to(dt)
    b, res = rnd(N, seed=3), rnd(M, N, seed=4).to(dt)
    nxt = torch.full(((nbytes + 1) // 2 + 8,), 1.0, dtype=dt, device="cuda")[:max((nbytes + 1) // 2, 1)]
    ref = [ops.gemm_nt(x, w), ops.gemm_nt(x, w, bias=b, residual=res), ops.gemm_nt(x, w, bias=b, act=ops.ACT_GELU)]
    hint = nxt if nbytes else None
    got = [ops.gemm_nt(x, w, prefetch=hint), ops.gemm_nt(x, w, bias=b, residual=res, prefetch=hint),
           ops.gemm_nt(x, w, bias=b, act=ops.ACT_GELU, prefetch=hint)]
    again = ops.gemm_nt(x, w)                      # no hint may linger
    torch.cuda.synchronize()
    for a, r in zip(got + [again], ref + [ref[0]]):
        assert torch.equal(a, r), what
    assert bool((nxt == 1.0).all())                # the hinted buffer is only read


@pytest.mark.parametrize("dt", DT)
def test_gemm_nt_identity_asymmetric(ops, dt):
    """A = I with an asymmetric B catches a transposed C write (cdna_hip_programming.md §3)."""
    n = 128
    w = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 125).cuda()  # exactly representable in bf16? use small ints
    w = (w % 17 - 8)
    x = torch.eye(n, device="cuda")
    out = ops.gemm_nt(x.to(dt), w.to(dt))
    assert torch.equal(out.float(), w.t().contiguous()), "C[m][n] must equal W[n][m] for X = I"


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 9, 7, 16, 24), (1, 16, 16, 64, 64), (2, 12, 12, 40, 136), (4, 128, 128, 32, 40)])
def test_conv3x3_fwd_dgrad_wgrad(ops, dt, B, H, W, Ci, Co):
    x = rnd(B, Ci, H, W, seed=1).to(dt).float().requires_grad_(True)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=(9 * Ci) ** -0.5).to(dt).float().requires_grad_(True)
    bias = rnd(Co, seed=3)
    y = F.conv2d(x, w, bias, padding=1)
    dy = rnd(B, Co, H, W, seed=4).to(dt).float()
    y.backward(dy)
    xn = x.detach().permute(0, 2, 3, 1).contiguous().to(dt)
    wf, wd = ops.pack_conv3x3(w.detach().contiguous(), dt)
    out = ops.gemm_nt(xn, wf, bias=bias, conv=(B, H, W, Ci)).view(B, H, W, Co)
    check(out.float().permute(0, 3, 1, 2), y.detach(), tol(dt), "conv fwd")
    dyn = dy.permute(0, 2, 3, 1).contiguous().to(dt)
    dx = ops.gemm_nt(dyn, wd, conv=(B, H, W, Co)).view(B, H, W, Ci)
    check(dx.float().permute(0, 3, 1, 2), x.grad, tol(dt), "conv dgrad")
    dwp = torch.zeros(Co, 9 * Ci, device="cuda")
    dbc = torch.zeros(Co, device="cuda")
    ops.gemm_tn(dyn, xn, dwp, conv=(B, H, W, Ci), dbias=dbc)
    check(dbc, dyn.float().sum((0, 1, 2)), tol(dt, 2e-5, 1e-4), "conv fused bias grad")
    dwt = torch.zeros(Co, Ci, 3, 3, device="cuda")
    ops.unpack_conv3x3_grad(dwp, dwt)
    check(dwt, w.grad, tol(dt), "conv wgrad")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (1000, 200, 144), (4608, 576, 288), (50, 16, 32),
                                   # more (tile, split) units than CUs, and an M that is no multiple of the 64-row step
                                   (300, 4608, 1152), (4645, 2304, 576)])
def test_gemm_tn(ops, dt, M, N, K):
    dy, x = rnd(M, N, seed=1).to(dt), rnd(M, K, seed=2).to(dt)
    dw = torch.zeros(N, K, device="cuda")
    ops.gemm_tn(dy, x, dw)
    check(dw, dy.float().t() @ x.float(), tol(dt, 2e-5, 1e-2), "tn")
    db = torch.zeros(N, device="cuda")
    ops.gemm_tn(dy, x, dw, dbias=db)  # accumulates; fused bias gradient
    check(dw, 2 * (dy.float().t() @ x.float()), tol(dt, 2e-5, 1e-2), "tn accumulate")
    check(db, dy.float().sum(0), tol(dt, 2e-5, 1e-4), "tn fused colsum")


@pytest.mark.parametrize("shapes", [
    [(4608, 576, 2304), (4608, 2304, 576), (4608, 576, 576), (4608, 1728, 576)],     # a stage-3 trunk block's four wgrads
    [(300, 200, 144)],                                                                # one small ragged problem, fewer steps than CUs * 2
    [(1000, 136, 72), (77, 432, 144), (5000, 1000, 200), (64, 16, 32), (130, 8, 8)],   # ragged N / K / M, tiles cut several times
    [(9000, 128, 128)],                                                               # one tile shared by many workgroups
    # few rows: more tiles than CUs AND fewer remainder steps than CUs -- most workgroups have an empty remainder share and write no
    # slab slot (round 1 folded their stale slots into the gradient: wrong qkv gradients at 128 / 256 px)
    [(256, 576, 2304), (256, 2304, 576), (256, 576, 576), (256, 1728, 576)],
    [(512, 576, 2304), (512, 2304, 576), (512, 576, 576), (512, 1728, 576)],
])
@pytest.mark.parametrize("defer", [False, True])
def test_gemm_tn_group(ops, shapes, defer):
    """Grouped stream-K wgrad: every problem's dW / dbias must match the per-problem reference, accumulating on top of what is there
    (defer: the slab reduces are collected and folded afterwards in one batched launch, as the trunk backward does)."""
    dt = torch.bfloat16
    jobs, refs = [], []
    for i, (M, N, K) in enumerate(shapes):
        dy, x = rnd(M, N, seed=10 + i).to(dt), rnd(M, K, seed=20 + i).to(dt)
        dw0, db0 = rnd(N, K, seed=30 + i), rnd(N, seed=40 + i)
        dw, db = dw0.clone(), (db0.clone() if i % 2 == 0 else None)
        jobs.append((dy, x, dw, db))
        refs.append((dw0 + dy.float().t() @ x.float(), (db0 + dy.float().sum(0)) if db is not None else None))
    # poison the block the caching allocator will hand out as the launch's scratch: results must not depend on its contents
    from spegnet_amd import _lib
    poison = torch.full((_lib.load().spg_gemm_tn_group_workspace_bytes() // 4,), float("nan"), device="cuda")
    del poison
    if defer:
        pending = []
        ops.gemm_tn_group(jobs, pending)
        # a second deferred launch (its own gradient buffer: one gradient may appear in only one pending launch) -- the batched
        # reduce must fold both launches' slabs
        dy0, x0 = jobs[0][0], jobs[0][1]
        extra_dw = torch.zeros(dy0.shape[-1], x0.shape[-1], device="cuda")
        ops.gemm_tn_group([(dy0, x0, extra_dw, None)], pending)
        ops.gemm_tn_group_reduce(pending)
        assert pending == []
        check(extra_dw, dy0.float().t() @ x0.float(), tol(dt, 2e-5, 1e-2), "second deferred launch")
    else:
        ops.gemm_tn_group(jobs)
    torch.cuda.synchronize()
    for (dy, x, dw, db), (rw, rb), shp in zip(jobs, refs, shapes):
        check(dw, rw, tol(dt, 2e-5, 1e-2), f"group dW {shp}")
        if db is not None:
            check(db, rb, tol(dt, 2e-5, 1e-4), f"group dbias {shp}")


@pytest.mark.parametrize("M,trunk_blocks", [(4608, 3), (2304, 2), (300, 1), (1152, 4)])
def test_gemm_tn_blocks(ops, M, trunk_blocks):
    """Whole-block wgrads of several trunk blocks in one launch (spg_gemm_tn_blocks): every dW / dbias equals the per-problem fp32
    reference, accumulating on top of what is there; both block orientations, the padded part of the 256 side (576 = 2.25 x 256), rows of M
    past the last 32-row slice (M = 300), problems without a bias, more blocks than CUs (4 x 84: a second round for some workgroups); two
    launches of the same inputs are bit-identical (one owner per element)."""
    dt = torch.bfloat16
    layer = [(1728, 576), (576, 576), (2304, 576), (576, 2304)]          # a stage-3 Hiera-L block: qkv, proj, fc1, fc2 as (N, K)
    jobs, refs, first = [], [], []
    for i, (N, K) in enumerate(layer * trunk_blocks):
        dy, x = rnd(M, N, seed=10 + i).to(dt), rnd(M, K, seed=60 + i).to(dt)
        dw0, db0 = rnd(N, K, seed=110 + i), rnd(N, seed=160 + i)
        dw, db = dw0.clone(), (db0.clone() if i % 3 != 1 else None)
        jobs.append((dy, x, dw, db))
        first.append((dw0, db0))
        refs.append((dw0 + dy.float().t() @ x.float(), (db0 + dy.float().sum(0)) if db is not None else None))
    cnt = ops.tn_blocks_count(jobs)
    assert cnt == 84 * trunk_blocks
    assert ops.tn_blocks_count([(jobs[0][0], jobs[0][1][:, :288].contiguous(), torch.zeros(1728, 288, device="cuda"), None)]) == -1   # K % 192
    ops.gemm_tn_blocks(jobs)
    torch.cuda.synchronize()
    for (dy, x, dw, db), (rw, rb) in zip(jobs, refs):
        check(dw, rw, tol(dt, 2e-5, 1e-2), f"blocks dW {tuple(dw.shape)} M={M}")
        if db is not None:
            check(db, rb, tol(dt, 2e-5, 1e-4), f"blocks dbias {tuple(dw.shape)} M={M}")
    again = [(dy, x, dw0.clone(), (db0.clone() if db is not None else None)) for (dy, x, dw, db), (dw0, db0) in zip(jobs, first)]
    ops.gemm_tn_blocks(again)
    torch.cuda.synchronize()
    for a, b in zip(jobs, again):
        assert torch.equal(a[2], b[2]) and (a[3] is None or torch.equal(a[3], b[3])), "gemm_tn_blocks is not bit-reproducible"


@pytest.mark.parametrize("dt", DT)
def test_layernorm_param_grads_batch(ops, dt):
    """Batched dgamma / dbeta (one launch for many LayerNorms of different widths / row counts) == the per-layer layernorm_bwd."""
    shapes = [(100, 144), (37, 1152), (513, 576), (64, 16), (4608, 576)] * 11      # 55 jobs: more than one launch
    jobs, refs = [], []
    for i, (M, C) in enumerate(shapes):
        x, dy = rnd(M, C, seed=100 + i).to(dt), rnd(M, C, seed=200 + i).to(dt)
        g, b = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
        _, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
        dg0, db0 = rnd(C, seed=300 + i), rnd(C, seed=400 + i)
        dg_ref, db_ref = dg0.clone(), db0.clone()
        ops.layernorm_bwd(dy, x, g, mean, rstd, dg_ref, db_ref)
        dg, db = dg0.clone(), db0.clone()
        jobs.append((dy, x, mean, rstd, dg, db))
        refs.append((dg_ref, db_ref))
    ops.layernorm_param_grads_batch(jobs)
    torch.cuda.synchronize()
    for (dy, x, mean, rstd, dg, db), (dg_ref, db_ref) in zip(jobs, refs):
        check(dg, dg_ref, 2e-5, f"batched dgamma {tuple(x.shape)}")
        check(db, db_ref, 2e-5, f"batched dbeta {tuple(x.shape)}")


@pytest.mark.parametrize("H,W,S", [(600, 800, 384), (1033, 777, 384), (200, 300, 384), (384, 384, 384), (2000, 1500, (352, 416))])
def test_preprocess_image_matches_reference_arithmetic(ops, H, W, S):
    """Device input pipeline (uint8 HWC -> /255 -> antialiased bilinear resize -> normalise) against the oracle's restatement of
    CODImageProcessor.process_image (same torch call as the reference, on CPU): down-scale, up-scale, identity, non-square."""
    from oracle import spegnet_oracle as O
    g = torch.Generator().manual_seed(H * 7 + W)
    img = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8)
    ref = O.preprocess_image(img, S)
    out = ops.preprocess_image(img.cuda(), S, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
    assert out.shape == ref.shape
    err = float((out.cpu() - ref).abs().max())
    assert err < 5e-6, f"max abs err {err:.2e}"


def test_preprocess_batch_and_device_batcher_match_single_image_kernel(ops):
    """the batched launch (ragged image sizes packed in one uint8 buffer) and the double-buffered DeviceBatcher around
    it must give exactly what the single-image kernel gives per image (itself pinned to the reference's CODImageProcessor above)."""
    from spegnet_amd.utils.data_loader import DeviceBatcher
    g = torch.Generator().manual_seed(11)
    sizes = [(300, 400), (384, 384), (97, 513), (640, 480), (64, 64)]
    imgs = [torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8) for h, w in sizes]
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    ref = torch.stack([ops.preprocess_image(im.cuda(), 96, mean, std) for im in imgs])
    offs, off = [], 0
    for h, w in sizes:
        offs.append(off)
        off += (h * w * 3 + 255) // 256 * 256
    base = torch.zeros(off, dtype=torch.uint8)
    for im, o in zip(imgs, offs):
        base[o:o + im.numel()] = im.reshape(-1)
    out = ops.preprocess_batch(base.cuda(), offs, sizes, 96, mean, std)
    assert torch.equal(out, ref)
    db = DeviceBatcher(96, mean, std, "cuda", max_bytes=4 << 20)
    h1 = db.submit(imgs[:3])
    h2 = db.submit(imgs[3:])            # two batches in flight (the trainer's prefetch)
    h3 = db.submit(imgs[:2])            # ... and a staging buffer being re-used
    assert torch.equal(db.result(h1), ref[:3]) and torch.equal(db.result(h2), ref[3:]) and torch.equal(db.result(h3), ref[:2])


def test_pack_matrix(ops):
    w = rnd(37, 53, seed=1)
    assert torch.equal(ops.pack_matrix(w, torch.float32), w)
    assert torch.equal(ops.pack_matrix(w, torch.float32, transpose=True), w.t().contiguous())
    assert torch.equal(ops.pack_matrix(w, torch.bfloat16, transpose=True), w.t().contiguous().to(torch.bfloat16))


# ------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,C", [(100, 144), (37, 1152), (64, 16), (513, 576), (5, 2048), (33, 288),   # every chunk-slot instance, both dtypes
                                 (8197, 288), (40003, 144)])   # many short rows: two / four rows per wave (bf16), ragged last waves
def test_layernorm(ops, dt, M, C):
    if dt == torch.float32 and C > 1280:
        pytest.skip("fp32 rows are limited to 5 x 64 x 4 columns")
    x = rnd(M, C, seed=1).to(dt)
    g, b = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    xf = x.float().requires_grad_(True)
    gf, bf = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xf, (C,), gf, bf, 1e-6)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
    check(y.float(), ref.detach(), tol(dt, 2e-5, 1e-2), "ln fwd")
    dy = rnd(M, C, seed=4).to(dt)
    dres = rnd(M, C, seed=5).to(dt)
    ref.backward(dy.float())
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dx = ops.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dres=dres)
    check(dx.float(), xf.grad + dres.float(), tol(dt, 5e-5, 2e-2), "ln dx")
    check(dg, gf.grad, tol(dt, 5e-5, 2e-2), "ln dgamma")
    check(db, bf.grad, tol(dt, 5e-5, 2e-2), "ln dbeta")


# ------------------------------------------------------------------------------------------- attention
def attn_reference(qkv, bias, B, H, W, heads, hd, ws, pooled):
    """The reference block's attention on an explicit zero-padded, partitioned map (sam2 Hiera semantics):
    padded tokens carry qkv == bias.  qkv: [B,H,W,3C] float (requires_grad ok).  Returns [B,Hq,Wq,C]."""
    C = heads * hd
    if ws <= 0:
        ws_, Hp, Wp = 0, H, W
        x = qkv
    else:
        ws_ = ws
        ph, pw = (-H) % ws, (-W) % ws
        Hp, Wp = H + ph, W + pw
        x = bias.view(1, 1, 1, 3 * C).expand(B, Hp, Wp, 3 * C).clone()
        x[:, :H, :W] = qkv
    if ws_ > 0:
        x = x.view(B, Hp // ws, ws, Wp // ws, ws, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, 3 * C)
    Bw, h, w = x.shape[0], x.shape[1], x.shape[2]
    q, k, v = x.reshape(Bw, h * w, 3, heads, hd).unbind(2)
    if pooled:
        q = F.max_pool2d(q.reshape(Bw, h, w, C).permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
        h, w = h // 2, w // 2
        q = q.reshape(Bw, h * w, heads, hd)
    att = (q.transpose(1, 2) * hd ** -0.5) @ k.transpose(1, 2).transpose(-2, -1)
    o = (att.softmax(-1) @ v.transpose(1, 2)).transpose(1, 2).reshape(Bw, h, w, C)
    if ws_ > 0:
        wq = ws // 2 if pooled else ws
        Hq, Wq = (Hp // 2, Wp // 2) if pooled else (Hp, Wp)
        o = o.view(B, Hq // wq, Wq // wq, wq, wq, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hq, Wq, C)
        Ho, Wo = (H // 2, W // 2) if pooled else (H, W)
        o = o[:, :Ho, :Wo]
    return o


ATTN_CASES = [
    # B, H, W, heads, hd, ws, pooled
    (2, 8, 8, 2, 16, 8, False),
    (2, 8, 8, 2, 16, 4, False),
    (1, 12, 12, 2, 16, 8, False),    # padded windows
    (2, 10, 10, 1, 16, 0, False),    # global, 100 tokens (2 key tiles)
    (2, 8, 8, 2, 16, 8, True),       # pooled queries
    (1, 12, 12, 2, 16, 8, True),     # pooled + padded
    (1, 24, 24, 2, 72, 16, False),   # the stage-3 configuration of Hiera-L @384
    (1, 24, 24, 1, 72, 0, False),    # global block @384 (576 tokens)
    (1, 24, 24, 2, 72, 16, True),    # block 44 style transition
    (1, 12, 12, 2, 72, 8, False),    # stage 4 @384
    (1, 6, 6, 2, 32, 4, False),
    (2, 16, 16, 2, 72, 4, False),    # 4 x 4 windows, W % 16 == 0: four windows packed per 64-token tile (stage 2 of Hiera-L)
    (1, 8, 32, 4, 72, 4, False),     # ... non-square map
    (2, 16, 16, 2, 72, 4, True),     # ... with 2 x 2-pooled queries (the stage 2 -> 3 transition block)
    (1, 12, 16, 2, 16, 4, False),
    # resident-window kernels (bf16, 65 < keys <= 320) beyond the @384 geometry:
    (1, 32, 32, 1, 72, 16, False),   # 2 x 2 full windows, no padding (the @768 layout): every tile full, two workgroups per window
    (1, 26, 26, 1, 72, 16, False),   # 16 + 10: 160 / 100 valid keys + the pad key inside a partly filled last tile
    (1, 20, 24, 2, 72, 16, False),   # non-square: 16 x 8 (pad key alone in its tile), 4 x 16 and 4 x 8 edge windows
    (2, 24, 24, 1, 72, 16, True),    # pooled queries, two images
]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("case", ATTN_CASES, ids=lambda c: "x".join(map(str, c)))
def test_attention(ops, dt, case):
    B, H, W, heads, hd, ws, pooled = case
    C = heads * hd
    qkv = rnd(B, H, W, 3 * C, seed=1).to(dt)
    bias = (0.5 * rnd(3 * C, seed=2)).to(dt)
    qf = qkv.float().requires_grad_(True)
    bf = bias.float().requires_grad_(True)
    ref = attn_reference(qf, bf, B, H, W, heads, hd, ws, pooled)
    qp = idx = None
    if pooled:
        qp, idx = ops.maxpool2_fwd(qkv, B, H, W, C, 3 * C, 0)
    out, lse = ops.attn_fwd(qkv, bias, B, H, W, heads, hd, ws, q_pooled=qp)
    check(out.float(), ref.detach(), tol(dt, 3e-5, 2e-2), "attn fwd")
    dout = rnd(*ref.shape, seed=3).to(dt)
    ref.backward(dout.float())
    dbias = torch.zeros(3 * C, device="cuda")
    dqkv, dqp = ops.attn_bwd(qkv, bias, out, dout, lse, dbias, B, H, W, heads, hd, ws, q_pooled=qp)
    if pooled:
        ops.maxpool2_bwd(dqp, idx, dqkv, B, H, W, C, 3 * C, 0)
    check(dqkv.float(), qf.grad, tol(dt, 1e-4, 3e-2), "attn dqkv")
    # bias gradient THROUGH PADDING only (the dense part is the colsum of dqkv, tested elsewhere)
    want = bf.grad
    if want is not None and float(want.abs().max()) > 0:
        check(dbias[C:], want[C:], tol(dt, 1e-4, 3e-2), "attn dbias(pad) k,v")
    else:
        assert float(dbias.abs().max()) == 0.0


# ------------------------------------------------------------------------------------------- pooling / resize
@pytest.mark.parametrize("dt", DT)
def test_maxpool2(ops, dt):
    B, H, W, C, ld, c0 = 2, 6, 8, 16, 48, 16
    x = rnd(B, H, W, ld, seed=1).to(dt)
    y, idx = ops.maxpool2_fwd(x, B, H, W, C, ld, c0)
    xs = x[..., c0:c0 + C].float().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.max_pool2d(xs, 2, 2)
    assert torch.equal(y.float(), ref.detach().permute(0, 2, 3, 1))
    dy = rnd(B, H // 2, W // 2, C, seed=2).to(dt)
    ref.backward(dy.float().permute(0, 3, 1, 2))
    dx = torch.full((B, H, W, ld), 7.0, device="cuda").to(dt)
    ops.maxpool2_bwd(dy, idx, dx, B, H, W, C, ld, c0)
    assert torch.equal(dx[..., c0:c0 + C].float(), xs.grad.permute(0, 2, 3, 1))
    assert float((dx[..., :c0].float() - 7).abs().max()) == 0  # untouched outside the window


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("h,w,s", [(6, 5, 2), (4, 4, 4), (3, 7, 8), (5, 5, 1)])
def test_upsample_bilinear(ops, dt, h, w, s):
    B, C, ld, c0 = 2, 16, 40, 16
    H, W = h * s, w * s
    x = rnd(B, h, w, C, seed=1).to(dt)
    xs = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.interpolate(xs, size=(H, W), mode="bilinear", align_corners=False)
    y = torch.zeros(B, H, W, ld, device="cuda").to(dt)
    ops.upsample_into(x, y, B, h, w, C, H, W, ld, c0)
    check(y[..., c0:c0 + C].float(), ref.detach().permute(0, 2, 3, 1), tol(dt, 1e-6, 8e-3), "upsample")
    assert float(y[..., :c0].abs().max()) == 0
    dy = rnd(B, H, W, ld, seed=2).to(dt)
    ref.backward(dy[..., c0:c0 + C].float().permute(0, 3, 1, 2))
    dx = torch.empty(B, h, w, C, device="cuda").to(dt)
    ops.upsample_bwd(dy, dx, B, h, w, C, H, W, ld, c0)
    check(dx.float(), xs.grad.permute(0, 2, 3, 1), tol(dt, 1e-5, 1e-2), "upsample bwd")
    ops.upsample_bwd(dy, dx, B, h, w, C, H, W, ld, c0, accumulate=True)
    check(dx.float(), 2 * xs.grad.permute(0, 2, 3, 1), tol(dt, 1e-5, 2e-2), "upsample bwd acc")


@pytest.mark.parametrize("dt", DT)
def test_patch_embed(ops, dt):
    B, S, D, KP = 2, 32, 48, 160
    img = rnd(B, 3, S, S, seed=1)
    w, b = rnd(D, 3, 7, 7, seed=2, scale=0.08), rnd(D, seed=3)
    cols = ops.patch_im2col(img, dt, KP)
    wp = torch.zeros(D, KP, device="cuda")
    wp[:, :147] = w.reshape(D, 147)
    out = ops.gemm_nt(cols, wp.to(dt), bias=b).view(B, S // 4, S // 4, D)
    ref = F.conv2d(img.to(dt).float(), w.to(dt).float(), b, stride=4, padding=3)
    check(out.float().permute(0, 3, 1, 2), ref, tol(dt), "patch embed")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,H,W", [(1, 4, 4), (2, 32, 48), (1, 384, 384), (1, 8, 768), (3, 12, 400)])
def test_patch_im2col_is_exactly_unfold(ops, dt, B, H, W):
    """The column matrix is a pure gather: every element must equal F.unfold's (7x7, stride 4, pad 3), rounded once to the compute
    type, with zero pad columns -- including rows wider than one 96-pixel workgroup segment (W/4 = 192, 100) and the image borders."""
    KP = 160
    img = rnd(B, 3, H, W, seed=7)
    cols = ops.patch_im2col(img, dt, KP)
    ref = F.unfold(img, 7, padding=3, stride=4).transpose(1, 2).reshape(B * (H // 4) * (W // 4), 147).to(dt)
    assert cols.shape == (B * (H // 4) * (W // 4), KP)
    assert torch.equal(cols[:, :147], ref)
    assert not cols[:, 147:].any()


@pytest.mark.parametrize("n,nparts", [(8, 1), (4096, 3), (1_000_000, 64), (54_525_952, 1024)])
def test_cast_bf16_and_its_sums_of_squares(n, nparts):
    """The gradient all-reduce's staging casts (engine/distributed.py): fp32 -> bf16 rounds to nearest even like torch, bf16 -> fp32 is exact,
    and the cast-back variant's partial sums of squares add up to the norm of what it wrote (the clip's norm without a pass of its own)."""
    from spegnet_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(n, device="cuda", generator=g) * 0.01
    h = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("spg_cast_bf16", x.data_ptr(), h.data_ptr(), n, 0, st)
    assert torch.equal(h, x.to(torch.bfloat16))
    back = torch.full((n,), float("nan"), device="cuda")
    _lib.call("spg_cast_bf16", back.data_ptr(), h.data_ptr(), n, 1, st)
    assert torch.equal(back, h.float())
    back2 = torch.full((n,), float("nan"), device="cuda")
    parts = torch.full((nparts,), float("nan"), device="cuda")
    _lib.call("spg_cast_bf16_sq", back2.data_ptr(), h.data_ptr(), n, parts.data_ptr(), nparts, st)
    assert torch.equal(back2, back)
    want = float(h.double().pow(2).sum())
    assert abs(float(parts.double().sum()) - want) < 1e-5 * want
    again = torch.empty_like(parts)
    _lib.call("spg_cast_bf16_sq", back2.data_ptr(), h.data_ptr(), n, again.data_ptr(), nparts, st)
    assert torch.equal(again, parts)          # a fixed order: bit-reproducible


# ------------------------------------------------------------------------------------------- BatchNorm
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,C,relu", [(500, 64, True), (2304, 512, True), (4, 128, True), (1000, 16, False)])
def test_batchnorm_train(ops, dt, M, C, relu):
    x = (rnd(M, C, seed=1) * 1.5 + 0.3).to(dt)
    g, b = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    rm, rv = 0.1 * rnd(C, seed=4), 1 + 0.1 * rnd(C, seed=5).abs()
    xf = x.float().requires_grad_(True)
    gf, bf = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.batch_norm(xf.t()[None], rm_ref, rv_ref, gf, bf, True, 0.1, 1e-5)[0].t()
    if relu:
        ref = F.relu(ref)
    stats = ops.bn_stats(x, C)
    ss, mi = ops.bn_finalize(stats, g, b, rm, rv, M, True)
    y = ops.bn_apply(x, ss, C, relu)
    check(y.float(), ref.detach(), tol(dt, 3e-5, 1.5e-2), "bn y")
    check(rm, rm_ref, 1e-5, "running mean")
    check(rv, rv_ref, 1e-4, "running var")
    dy = rnd(M, C, seed=6).to(dt)
    ref.backward(dy.float())
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dx = ops.bn_bwd(dy, x, ss, mi, g, dg, db, C, relu)
    check(dx.float(), xf.grad, tol(dt, 2e-4, 3e-2), "bn dx")
    check(dg, gf.grad, tol(dt, 1e-4, 3e-2), "bn dgamma")
    check(db, bf.grad, tol(dt, 1e-4, 3e-2), "bn dbeta")


@pytest.mark.parametrize("dt", DT)
def test_batchnorm_eval(ops, dt):
    M, C = 300, 64
    x = rnd(M, C, seed=1).to(dt)
    g, b, rm, rv = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4), 1 + 0.1 * rnd(C, seed=5).abs()
    ss, _ = ops.bn_finalize(None, g, b, rm, rv, M, False)
    y = ops.bn_apply(x, ss, C, True)
    ref = F.relu(F.batch_norm(x.float().t()[None], rm, rv, g, b, False, 0.1, 1e-5)[0].t())
    check(y.float(), ref, tol(dt, 1e-5, 1e-2), "bn eval")


# ------------------------------------------------------------------------------------------- CFI pieces
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,HW,C,R", [(3, 36, 64, 32), (16, 144, 512, 32), (42, 64, 512, 32)])   # (batches beyond one 64 KiB staging chunk)
def test_se_block(ops, dt, B, HW, C, R):
    x = rnd(B, HW, C, seed=1).to(dt)
    w1, w2 = rnd(R, C, seed=2, scale=0.2), rnd(C, R, seed=3, scale=0.3)
    xf = x.float().requires_grad_(True)
    w1f, w2f = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    gap_ref = xf.mean(1)
    s_ref = torch.sigmoid(F.linear(F.relu(F.linear(gap_ref, w1f)), w2f))
    ref = xf * s_ref[:, None]
    gap = ops.gap_sum(x, B, HW, C)                       # column sums; the mean's 1 / HW is the SE kernels' in_scale
    hidden, scale = ops.se_fc(gap, w1, w2, 1.0 / HW)
    y = ops.chan_scale(x, scale, B, HW, C)
    check(y.float(), ref.detach(), tol(dt, 1e-5, 1e-2), "se fwd")
    dy = rnd(B, HW, C, seed=4).to(dt)
    ref.backward(dy.float())
    dscale = ops.chan_prod_sum(dy, x, B, HW, C)
    dw1, dw2 = torch.zeros_like(w1), torch.zeros_like(w2)
    dgap = ops.se_fc_bwd(gap, w1, w2, hidden, scale, dscale, dw1, dw2, 1.0 / HW)
    dx = ops.chan_scale_bwd(dy, scale, dgap, B, HW, C)
    check(dx.float(), xf.grad, tol(dt, 1e-4, 2e-2), "se dx")
    check(dw1, w1f.grad, tol(dt, 1e-4, 3e-2), "se dw1")
    check(dw2, w2f.grad, tol(dt, 1e-4, 3e-2), "se dw2")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("dil", [1, 6, 18])
def test_dwconv(ops, dt, dil):
    B, H, W, C = 2, 13, 11, 32
    x = rnd(B, H, W, C, seed=1).to(dt)
    w = rnd(C, 1, 3, 3, seed=2, scale=0.3)
    xs = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wf = w.clone().requires_grad_(True)
    ref = F.conv2d(xs, wf, padding=dil, dilation=dil, groups=C)
    y = ops.dwconv3x3(x, w.view(C, 9), B, H, W, C, dil)
    check(y.float(), ref.detach().permute(0, 2, 3, 1), tol(dt, 1e-5, 1e-2), "dw fwd")
    dy = rnd(B, H, W, C, seed=3).to(dt)
    ref.backward(dy.float().permute(0, 3, 1, 2))
    dx = ops.dwconv3x3(dy, w.view(C, 9), B, H, W, C, dil, flip=True)
    check(dx.float(), xs.grad.permute(0, 2, 3, 1), tol(dt, 1e-5, 1e-2), "dw dx")
    dw = torch.zeros(C, 9, device="cuda")
    ops.dwconv3x3_wgrad(dy, x, dw, B, H, W, C, dil)
    check(dw, wf.grad.view(C, 9), tol(dt, 1e-4, 2e-2), "dw dw")


@pytest.mark.parametrize("dt", DT)
def test_easpp_fuse(ops, dt):
    B, HW, C = 2, 30, 32
    br = [rnd(B * HW, C, seed=i).to(dt) for i in range(4)]
    glob = rnd(B, C, seed=9)
    w = rnd(C, 5, seed=10, scale=0.4)
    brf = [b.float().requires_grad_(True) for b in br]
    gf, wf = glob.clone().requires_grad_(True), w.clone().requires_grad_(True)
    cat = torch.cat([b.view(B, HW, C) for b in brf] + [gf[:, None].expand(B, HW, C)], -1)  # branch-major
    ref = F.conv2d(cat.permute(0, 2, 1)[..., None], wf.view(C, 5, 1, 1), groups=C)[..., 0].permute(0, 2, 1)
    y = ops.easpp_fuse(br, glob, w, B, HW, C)
    check(y.float().view(B, HW, C), ref.detach(), tol(dt, 1e-5, 1e-2), "fuse fwd")
    dy = rnd(B * HW, C, seed=11).to(dt)
    ref.backward(dy.float().view(B, HW, C))
    dw = torch.zeros_like(w)
    d, dglob = ops.easpp_fuse_bwd(dy, br, glob, w, dw, B, HW, C)
    for i in range(4):
        check(d[i].float(), brf[i].grad, tol(dt, 1e-5, 1e-2), f"fuse d{i}")
    check(dglob, gf.grad, tol(dt, 1e-4, 2e-2), "fuse dglob")
    check(dw, wf.grad, tol(dt, 1e-4, 2e-2), "fuse dw")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("C", [64, 128, 256, 16])
def test_head1x1(ops, dt, C):
    M = 1000
    x = rnd(M, C, seed=1).to(dt)
    w, b = rnd(C, seed=2, scale=0.2), rnd(1, seed=3)
    xf, wf, bf = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = xf @ wf + bf
    y = ops.head1x1(x, w, b, M, C)
    check(y.float(), ref.detach(), tol(dt, 1e-5, 1e-2), "head fwd")
    dy = rnd(M, seed=4).to(dt)
    ref.backward(dy.float())
    dx = rnd(M, C, seed=5).to(dt)
    base = dx.float().clone()
    dw, db = torch.zeros(C, device="cuda"), torch.zeros(1, device="cuda")
    ops.head1x1_bwd(dy, x, w, dx, dw, db, M, C, accumulate=True)
    check(dx.float(), base + xf.grad, tol(dt, 1e-5, 1.5e-2), "head dx")
    check(dw, wf.grad, tol(dt, 1e-4, 2e-2), "head dw")
    check(db, bf.grad, tol(dt, 1e-4, 2e-2), "head db")


@pytest.mark.parametrize("dt", DT)
def test_small_utils(ops, dt):
    M, C = 333, 48
    x = rnd(M, C, seed=1).to(dt)
    out = torch.zeros(C, device="cuda")
    ops.colsum(x, out)
    check(out, x.float().sum(0), tol(dt, 1e-5, 1e-5), "colsum")
    y = torch.zeros(M, 80, device="cuda").to(dt)
    ops.copy_channels(x, y, M, 16, C, 16, 80, 32)
    assert torch.equal(y[:, 32:48], x[:, 16:32]) and float(y[:, :32].abs().max()) == 0
    a, b = rnd(64, 40, seed=2).to(dt), rnd(64, 40, seed=3).to(dt)
    check(ops.add(a, b).float(), a.float() + b.float(), tol(dt, 1e-7, 8e-3), "add")


# ------------------------------------------------------------------------------------------- fused CFI / EFE / PED kernels (csrc/head.hip)
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,C", [(500, 64), (2304, 512), (6, 128)])
def test_bn_stats_finalize_one_launch(ops, dt, M, C):
    """bn_stats + bn_finalize (+ num_batches_tracked) folded into one launch == the two-launch form, and == torch batch_norm."""
    x = (rnd(M, C, seed=1) * 1.5 + 0.3).to(dt)
    g, b = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    rm, rv = 0.1 * rnd(C, seed=4), 1 + 0.1 * rnd(C, seed=5).abs()
    rm2, rv2 = rm.clone(), rv.clone()
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    ss_ref, mi_ref = ops.bn_finalize(ops.bn_stats(x, C), g, b, rm2, rv2, M, True)
    ss, mi = ops.bn_stats_finalize(x, C, g, b, rm, rv, nbt)
    ss2, _ = ops.bn_stats_finalize(x, C, g, b, rm.clone(), rv.clone(), nbt)
    assert torch.equal(ss, ss_ref) and torch.equal(mi, mi_ref) and torch.equal(rm, rm2) and torch.equal(rv, rv2)
    assert torch.equal(ss, ss2), "identical inputs must give bit-identical batch statistics (deterministic reduction)"
    assert int(nbt) == 2


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("C", [64, 128, 256])
@pytest.mark.parametrize("write_y", [True, False])
def test_bn_apply_head(ops, dt, C, write_y):
    M = 1000
    x = (rnd(M, C, seed=1) * 1.5 + 0.3).to(dt)
    ss = torch.cat([1 + 0.1 * rnd(C, seed=2), 0.2 * rnd(C, seed=3)])
    w, b = rnd(C, seed=4) * 0.2, rnd(1, seed=5)
    y_ref = ops.bn_apply(x, ss, C, True)
    y, pred = ops.bn_apply_head(x, ss, w, b, C, relu=True, write_y=write_y)
    if write_y:
        assert torch.equal(y, y_ref)
        check(pred.float(), ops.head1x1(y_ref, w, b, M, C).float(), tol(dt, 1e-5, 1e-2), "bn_apply_head pred")
    else:
        assert y is None
        ref = torch.relu(x.float() * ss[:C] + ss[C:]) @ w + b
        check(pred.float(), ref, tol(dt, 1e-5, 1e-2), "bn_apply_head pred (y not stored)")


@pytest.mark.parametrize("dt", DT)
def test_cfi_fusion_without_concat(ops, dt):
    """conv1x1(cat[s2, up(s3), up(s4)]) == s2.W2^T + up(s3.W3^T) + up(s4.W4^T) (feature_integration.py:229-239): three GEMMs at the
    sources' own resolutions + cfi_combine against the reference formulation in torch."""
    B, h, w, C2, C3, C4, N = 2, 16, 24, 32, 64, 128, 64
    s2, s3, s4 = rnd(B, h, w, C2, seed=1).to(dt), rnd(B, h // 2, w // 2, C3, seed=2).to(dt), rnd(B, h // 4, w // 4, C4, seed=3).to(dt)
    Wf = (rnd(N, C2 + C3 + C4, seed=4) * 0.1).to(dt)
    up = lambda t: F.interpolate(t.float().permute(0, 3, 1, 2), size=(h, w), mode="bilinear", align_corners=False)
    cat = torch.cat([s2.float().permute(0, 3, 1, 2), up(s3), up(s4)], 1)
    ref = F.conv2d(cat, Wf.float()[:, :, None, None]).permute(0, 2, 3, 1).reshape(B * h * w, N)
    y2 = ops.gemm_nt(s2.view(-1, C2), Wf[:, :C2].contiguous())
    y3 = ops.gemm_nt(s3.view(-1, C3), Wf[:, C2:C2 + C3].contiguous())
    y4 = ops.gemm_nt(s4.view(-1, C4), Wf[:, C2 + C3:].contiguous())
    out = ops.cfi_combine(y2, y3, y4, B, h, w, h // 2, w // 2, h // 4, w // 4, N)
    check(out.float(), ref, tol(dt, 2e-5, 2e-2), "cfi fusion (split by source)")
    # wgrad into a column slice of the full [N, C2+C3+C4] gradient (row stride = full width)
    dy = rnd(B * h * w, N, seed=5).to(dt)
    gw = torch.zeros(N, C2 + C3 + C4, device="cuda")
    ops.gemm_tn(dy, s2.view(-1, C2), gw[:, :C2])
    check(gw[:, :C2], dy.float().t() @ s2.view(-1, C2).float(), tol(dt, 2e-5, 1e-2), "wgrad into a column slice")
    assert float(gw[:, C2:].abs().max()) == 0.0


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("hx,Cx,he,Ce,bn", [(12, 64, 12, 16, True), (12, 32, 6, 16, False), (7, 16, 0, 0, True), (24, 256, 24, 64, True),
                                           (10, 128, 5, 64, True), (18, 72, 9, 8, False)])
def test_ped_gather_fwd_bwd(ops, dt, hx, Cx, he, Ce, bn):
    """cat[up2(relu(bn(x))), up_s(edge)] in one launch == bn_apply + two upsample launches (which are pinned to F.interpolate above);
    the adjoint == the generic upsample_bwd (s = 2 and 4, channel offsets, accumulate)."""
    B, wx = 2, hx + 4 if hx % 2 == 0 else hx + 3
    H, W = 2 * hx, 2 * wx
    we = (W // (H // he)) if he else 0
    x = rnd(B, hx, wx, Cx, seed=1).to(dt)
    ss = torch.cat([1 + 0.1 * rnd(Cx, seed=2), 0.2 * rnd(Cx, seed=3)]) if bn else None
    edge = rnd(B, he, we, Ce, seed=4).to(dt) if Ce else None
    pc = ops.ped_gather(x, ss, B, hx, wx, Cx, edge, he, we, Ce)
    xa = ops.bn_apply(x, ss, Cx, True) if bn else x
    ref = torch.zeros((B * H * W, Cx + Ce), dtype=dt, device="cuda")
    ops.upsample_into(xa, ref, B, hx, wx, Cx, H, W, Cx + Ce, 0)
    if Ce:
        ops.upsample_into(edge, ref, B, he, we, Ce, H, W, Cx + Ce, Cx)
    check(pc.float(), ref.float(), tol(dt, 1e-6, 8e-3), "ped_gather")
    # torch cross-check of the whole thing
    act = torch.relu(x.float() * ss[:Cx] + ss[Cx:]) if bn else x.float()
    t = F.interpolate(act.permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).reshape(B * H * W, Cx)
    check(pc[:, :Cx].float(), t, tol(dt, 1e-5, 1.5e-2), "ped_gather vs F.interpolate")
    dy = rnd(B * H * W, Cx + Ce, seed=6).to(dt)
    for (h_, w_, C_, c0) in ([(hx, wx, Cx, 0)] + ([(he, we, Ce, Cx)] if Ce else [])):
        for acc in (False, True):
            d0 = rnd(B, h_, w_, C_, seed=7).to(dt)
            d_ref, d_new = d0.clone(), d0.clone()
            ops.upsample_bwd(dy, d_ref, B, h_, w_, C_, H, W, Cx + Ce, c0, accumulate=acc)
            ops.ped_gather_bwd(dy, d_new, B, h_, w_, C_, H, W, Cx + Ce, c0, accumulate=acc)
            check(d_new.float(), d_ref.float(), tol(dt, 1e-5, 1e-2), f"ped_gather_bwd C={C_} acc={acc}")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,C,with_next", [(1000, 64, False), (700, 128, True), (2304, 256, True)])
def test_bn_bwd_head(ops, dt, M, C, with_next):
    """BN backward with the head's rank-one gradient formed on the fly == head1x1_bwd (dx, dw, db) followed by bn_bwd."""
    x = (rnd(M, C, seed=1) * 1.5 + 0.3).to(dt)
    g, b = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    ss, mi = ops.bn_stats_finalize(x, C, g, b, None, None, None)
    hw = rnd(C, seed=4) * 0.2
    dpred = rnd(M, seed=5).to(dt)
    dnext = rnd(M, C, seed=6).to(dt) if with_next else None
    # reference: materialise y and d_y
    y = ops.bn_apply(x, ss, C, True)
    d_y = dnext.clone() if with_next else torch.zeros(M, C, dtype=dt, device="cuda")
    dw_ref, db_ref = torch.zeros(C, device="cuda"), torch.zeros(1, device="cuda")
    ops.head1x1_bwd(dpred, y, hw, d_y, dw_ref, db_ref, M, C, accumulate=True)
    dg_ref, dbt_ref = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dx_ref = ops.bn_bwd(d_y, x, ss, mi, g, dg_ref, dbt_ref, C, True)
    dg, dbt, dw, db = (torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(1, device="cuda"))
    dx = ops.bn_bwd_head(dnext, x, dpred, hw, ss, mi, g, dg, dbt, dw, db, C)
    check(dx.float(), dx_ref.float(), tol(dt, 2e-4, 3e-2), "bn_bwd_head dx")
    check(dg, dg_ref, tol(dt, 1e-4, 3e-2), "dgamma")
    check(dbt, dbt_ref, tol(dt, 1e-4, 3e-2), "dbeta")
    check(dw, dw_ref, tol(dt, 1e-4, 2e-2), "head dw")
    check(db, db_ref, tol(dt, 1e-4, 2e-2), "head db")
    dx2 = ops.bn_bwd_head(dnext, x, dpred, hw, ss, mi, g, dg.clone(), dbt.clone(), dw.clone(), db.clone(), C)
    assert torch.equal(dx, dx2), "deterministic reduction: identical inputs, identical dx"


@pytest.mark.parametrize("dt", DT)
def test_easpp_middle_branch_batched(ops, dt):
    """The branch-batched e-ASPP middle (dwconv4 -> BN statistics of the 4 branches in one reduction -> global branch kernel -> grouped 1x1
    with the branch BN + ReLU folded in) and its backward against torch autograd of the reference formulation
    (feature_integration.py:397-412: 4 x [dilated depth-wise conv, BN, ReLU], [GAP, 1x1, BN, ReLU, broadcast], cat, grouped 1x1)."""
    B, h, w, C = 6, 10, 12, 32
    rates = (1, 2, 3, 5)
    HW = h * w
    # (images of clearly different brightness: the global branch's BatchNorm sees only B values per channel, and with near-equal
    # values its backward amplifies storage rounding without bound -- that conditioning is the model's, not the kernels')
    r1 = (rnd(B, h, w, C, seed=1).abs() * (0.5 + torch.arange(B, device="cuda").view(B, 1, 1, 1))).to(dt)
    wd = [rnd(C, 9, seed=10 + i) * 0.3 for i in range(4)]
    gb = [(1 + 0.1 * rnd(C, seed=20 + i), 0.1 * rnd(C, seed=30 + i)) for i in range(4)]
    Wg = rnd(C, C, seed=40) * 0.2
    gg, bg = 1 + 0.1 * rnd(C, seed=41), 0.1 * rnd(C, seed=42)
    wf = rnd(C, 5, seed=43) * 0.4
    dy = rnd(B * HW, C, seed=44).to(dt)
    # ---- torch reference (fp32, from the same rounded inputs)
    leaves = [r1.float().clone().requires_grad_(True)] + [t.clone().requires_grad_(True) for t in wd] + \
             [t.clone().requires_grad_(True) for pair in gb for t in pair] + [t.clone().requires_grad_(True) for t in (Wg, gg, bg, wf)]
    x_t, wd_t, gb_t = leaves[0], leaves[1:5], leaves[5:13]
    Wg_t, gg_t, bg_t, wf_t = leaves[13:]
    xn = x_t.permute(0, 3, 1, 2)
    brs = []
    for i, d in enumerate(rates):
        z = F.conv2d(xn, wd_t[i].view(C, 1, 3, 3), padding=d, dilation=d, groups=C)
        # the depth-wise outputs are STORED in dt before BN + ReLU: model that rounding (straight-through), otherwise pixels whose BN(z)
        # is within bf16 rounding of 0 flip their ReLU mask against the reference and whole gradient terms appear / disappear
        z = z + (z.to(dt).float() - z).detach()
        brs.append(F.relu(F.batch_norm(z, None, None, gb_t[2 * i], gb_t[2 * i + 1], True, 0.1, 1e-5)))
    g0 = F.conv2d(xn.mean((2, 3), keepdim=True), Wg_t.view(C, C, 1, 1))
    g1 = F.relu(F.batch_norm(g0, None, None, gg_t, bg_t, True, 0.1, 1e-5)).expand(B, C, h, w)
    fu = F.conv2d(torch.cat(brs + [g1], 1), wf_t.view(C, 5, 1, 1), groups=C).permute(0, 2, 3, 1).reshape(B * HW, C)
    (fu * dy.float()).sum().backward()
    # ---- HIP path
    dcat = ops.dwconv4(r1, wd, rates, B, h, w, C)
    ss_b, mi_b = ops.bn_stats_finalize4(dcat, 4 * C, [p[0] for p in gb], [p[1] for p in gb], None, None, None)
    gs = ops.gap_sum(r1, B, HW, C)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    gm, gl0, glob, ss_g, mi_g = ops.easpp_global_fwd(gs, Wg, gg, bg, rm, rv, nbt, B, C, HW, True)
    fu0 = ops.easpp_fuse_bn(dcat, ss_b, glob, wf, B, HW, C)
    check(fu0.float(), fu.detach(), tol(dt, 3e-4, 3e-2), "e-ASPP middle forward")   # (BatchNorm over B = 4 values per channel amplifies rounding)
    assert int(nbt) == 1
    z = lambda *sh: torch.zeros(*sh, device="cuda")
    dwf, dWg, dgg, dbg = z(C * 5), z(C, C), z(C), z(C)
    dgam, dbet, dwd = [z(C) for _ in range(4)], [z(C) for _ in range(4)], [z(C, 9) for _ in range(4)]
    S = ops.gap_sum(dy, B, HW, C)
    gadd = ops.easpp_global_bwd(S, glob, gl0, gm, wf.view(-1), Wg, gg, mi_g, dwf, dWg, dgg, dbg, B, C, HW)
    d_dcat = ops.easpp_fuse_bn_bwd(dy, dcat, wf.view(-1), ss_b, mi_b, [p[0] for p in gb], dgam, dbet, dwf, B, HW, C)
    ops.dwconv4_wgrad(d_dcat, r1, rates, dwd, B, h, w, C)
    d_r1 = ops.dwconv4_dgrad(d_dcat, wd, rates, gadd, B, h, w, C)
    t = tol(dt, 2e-3, 6e-2)
    check(d_r1.float().view(B, h, w, C), x_t.grad, t, "d r1")
    check(dwf.view(C, 5), wf_t.grad, t, "d fusion weight")
    check(dWg, Wg_t.grad, t, "d global 1x1 weight")
    check(dgg, gg_t.grad, t, "d global BN gamma")
    check(dbg, bg_t.grad, t, "d global BN beta")
    for i in range(4):
        check(dwd[i], wd_t[i].grad, t, f"d depth-wise weight {i}")
        check(dgam[i], gb_t[2 * i].grad, t, f"d branch BN gamma {i}")
        check(dbet[i], gb_t[2 * i + 1].grad, t, f"d branch BN beta {i}")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,h,w,C", [(3, 48, 48, 32), (2, 56, 56, 16), (2, 72, 64, 16), (2, 9, 7, 64)])
def test_dwconv4_forward_dgrad_wgrad_match_torch(ops, dt, B, h, w, C):
    """The four dilated depth-wise 3x3 convolutions with the reference's rates (feature_integration.py:397-404), their input gradient
    (+ the GAP adjoint) and weight gradients on the 48 x 48 context map of a 384 px input and on larger / ragged maps, against torch's
    grouped convolution from the same rounded inputs."""
    rates = (1, 6, 12, 18)
    x = rnd(B, h, w, C, seed=3).to(dt)
    wd = [rnd(C, 9, seed=50 + i) * 0.3 for i in range(4)]
    dy = rnd(B * h * w, 4 * C, seed=7).to(dt)
    gadd = rnd(B, C, seed=8)
    xt = x.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
    wt = [t.clone().requires_grad_(True) for t in wd]
    ref = torch.cat([F.conv2d(xt, wt[i].view(C, 1, 3, 3), padding=d, dilation=d, groups=C) for i, d in enumerate(rates)], 1)
    (ref.permute(0, 2, 3, 1).reshape(B * h * w, 4 * C) * dy.float()).sum().backward()
    dcat = ops.dwconv4(x, wd, rates, B, h, w, C)
    check(dcat.float(), ref.detach().permute(0, 2, 3, 1).reshape(B * h * w, 4 * C), tol(dt, 1e-5, 1e-2), "dwconv4")
    dwd = [torch.zeros(C, 9, device="cuda") for _ in range(4)]
    ops.dwconv4_wgrad(dy, x, rates, dwd, B, h, w, C)
    dx = ops.dwconv4_dgrad(dy, wd, rates, gadd, B, h, w, C)
    want_dx = xt.grad.permute(0, 2, 3, 1) + gadd.view(B, 1, 1, C)
    check(dx.float().view(B, h, w, C), want_dx, tol(dt, 1e-5, 1e-2), "dwconv4 dgrad")
    for i in range(4):
        check(dwd[i], wt[i].grad, tol(dt, 1e-4, 1e-3), f"dwconv4 wgrad {i}")
    dwd2 = [torch.zeros(C, 9, device="cuda") for _ in range(4)]
    ops.dwconv4_wgrad(dy, x, rates, dwd2, B, h, w, C)
    assert all(torch.equal(a, b) for a, b in zip(dwd, dwd2)), "deterministic weight gradient"


def test_pack_cols2_and_add_cols_batch(ops):
    """the position-embedding operand [pos_embed | pos_embed_window | 0] and the column-slice gradient adds (models/engine.py trunk)"""
    a, b = rnd(144, 49, seed=1), rnd(144, 64, seed=2)
    for dt in DT:
        out = ops.pack_cols2(a, b, 120, dt)
        ref = torch.zeros(144, 120, device="cuda")
        ref[:, :49], ref[:, 49:113] = a, b
        assert torch.equal(out, ref.to(dt))
    d0, d1, d2 = rnd(144, 147, seed=3), rnd(144, 49, seed=4), rnd(144, 64, seed=5)
    s0, s1 = rnd(144, 152, seed=6), rnd(144, 120, seed=7)
    r0, r1, r2 = d0 + s0[:, :147], d1 + s1[:, :49], d2 + s1[:, 49:113]
    ops.add_cols_batch([(d0, s0), (d1, s1), (d2, s1[:, 49:])])
    assert torch.equal(d0, r0) and torch.equal(d1, r1) and torch.equal(d2, r2)


@pytest.mark.parametrize("M,N,K", [(4608, 2304, 576), (1000, 576, 2304), (300, 200, 144)])
def test_gemm_nt_saved_gelu_derivative_modes(ops, M, N, K):
    """bf16 MLP epilogues: the forward GEMM stores gelu'(pre-activation) next to gelu(.) (ACT_GELU_SAVE_GRAD), the backward GEMM multiplies
    by it (ACT_MUL_H).  Shapes cover both bf16 kernel families (two workgroups per CU / persistent)."""
    dt = torch.bfloat16
    x, w, b = rnd(M, K, seed=1).to(dt), (rnd(N, K, seed=2) * K ** -0.5).to(dt), rnd(N, seed=3)
    pre = x.float() @ w.float().t() + b
    pr = pre.clone().requires_grad_(True)
    gl = F.gelu(pr)
    gl.sum().backward()
    d_out = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
    out = ops.gemm_nt(x, w, bias=b, act=ops.ACT_GELU_SAVE_GRAD, preact_out=d_out)
    check(out.float(), gl.detach(), 2.5e-2, "gelu")
    check(d_out.float(), pr.grad, 2.5e-2, "gelu'")
    ref_plain = ops.gemm_nt(x, w, bias=b, act=ops.ACT_GELU)
    assert torch.equal(out, ref_plain)                     # the value path is bit-identical to the plain GELU epilogue
    dy, w2, res = rnd(M, K, seed=4).to(dt), (rnd(N, K, seed=5) * K ** -0.5).to(dt), rnd(M, N, seed=6).to(dt)
    got = ops.gemm_nt(dy, w2, gelu_h=d_out, residual=res, act=ops.ACT_MUL_H)
    want = (dy.float() @ w2.float().t()) * d_out.float() + res.float()
    check(got.float(), want, 2.5e-2, "acc * saved derivative + residual")
    with pytest.raises(RuntimeError):                      # fp32 (parity) kernels keep the pre-activation: the modes are refused there
        ops.gemm_nt(x.float(), w.float(), bias=b, act=ops.ACT_GELU_SAVE_GRAD, preact_out=d_out.float())


# ------------------------------------------------------------------------------------------- CU budget
@pytest.mark.parametrize("budget", [240, 96, 8])
def test_gemms_under_a_cu_budget(ops, budget):
    """The multi-GPU step sizes the persistent GEMM grids for fewer CUs while a collective is resident (ops.cu_budget, a per-call ABI
    argument): tile widths, split plans, the grouped wgrad's whole-tile / remainder shares and its workspace all change with it -- same
    results required (reference: the same calls at the full chip)."""
    dt = torch.bfloat16
    x, w = rnd(4608, 576, seed=1).to(dt), (rnd(1728, 576, seed=2) * 0.05).to(dt)
    b, res = rnd(1728, seed=3), rnd(4608, 1728, seed=4).to(dt)
    full = ops.gemm_nt(x, w, bias=b, residual=res)
    xl, wl = rnd(4608, 2304, seed=5).to(dt), (rnd(576, 2304, seed=6) * 0.03).to(dt)
    full_l = ops.gemm_nt(xl, wl)                                   # long-K, few tiles: the persistent kernel
    xc, wc = rnd(2, 24, 24, 64, seed=7).to(dt), (rnd(128, 9 * 64, seed=8) * 0.05).to(dt)
    full_c = ops.gemm_nt(xc, wc, conv=(2, 24, 24, 64))
    shapes = [(4608, 576, 2304), (4608, 2304, 576), (4608, 576, 576), (4608, 1728, 576)]
    mk = lambda: [(rnd(M, N, seed=10 + i).to(dt), rnd(M, K, seed=20 + i).to(dt), torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda"))
                  for i, (M, N, K) in enumerate(shapes)]
    ref_jobs = mk()
    ops.gemm_tn_group(ref_jobs)
    dyc = rnd(2 * 24 * 24, 128, seed=9).to(dt)
    dwc_ref = torch.zeros(128, 9 * 64, device="cuda")
    ops.gemm_tn(dyc, xc, dwc_ref, conv=(2, 24, 24, 64))
    with ops.cu_budget(budget):
        assert ops.cu_budget_now() == budget
        check(ops.gemm_nt(x, w, bias=b, residual=res).float(), full.float(), 1e-6, "gemm_nt under budget")   # same arithmetic per element
        check(ops.gemm_nt(xl, wl).float(), full_l.float(), 1e-6, "long-K gemm_nt under budget")
        check(ops.gemm_nt(xc, wc, conv=(2, 24, 24, 64)).float(), full_c.float(), 1e-6, "conv gemm_nt under budget")
        jobs = mk()
        pending = []
        ops.gemm_tn_group(jobs, pending)
        ops.gemm_tn_group_reduce(pending)
        dwc = torch.zeros(128, 9 * 64, device="cuda")
        ops.gemm_tn(dyc, xc, dwc, conv=(2, 24, 24, 64))
    assert ops.cu_budget_now() == 0
    for (dy, xx, dw, db), (_, _, rw, rb) in zip(jobs, ref_jobs):
        check(dw, dy.float().t() @ xx.float(), 1e-2, "grouped wgrad under budget vs torch")
        check(dw, rw, 2e-5, "grouped wgrad under budget vs full chip")     # (summation order over M differs with the shares)
        check(db, rb, 2e-5, "grouped dbias under budget")
    check(dwc, dwc_ref, 2e-5, "conv wgrad under budget")


# ------------------------------------------------------------------------------------------- fused CODLoss
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("case", ["rand", "zeros", "ones"])
def test_fused_codloss_matches_torch_formula(dt, case):
    """csrc/loss.hip against the torch restatement of the same formula (itself pinned to the reference by
    tests/test_oracle_golden.py through oracle.cod_loss)."""
    from spegnet_amd.utils.loss_functions import CODLoss
    from oracle import spegnet_oracle as O
    B, S = 3, 64
    g = torch.Generator().manual_seed(5)
    preds = [(torch.randn(B, 1, S // d, S // d, generator=g) * 2).cuda().to(dt) for d in (4, 2, 1)]
    edge = (torch.randn(B, 1, S // 8, S // 8, generator=g) * 2).cuda().to(dt)
    if case == "rand":
        masks = (torch.rand(B, 1, S, S, generator=g) > 0.7).float().cuda()
        edges = (torch.rand(B, 1, S, S, generator=g) > 0.95).float().cuda()
    elif case == "zeros":
        masks, edges = torch.zeros(B, 1, S, S).cuda(), torch.zeros(B, 1, S, S).cuda()
    else:
        masks, edges = torch.ones(B, 1, S, S).cuda(), torch.ones(B, 1, S, S).cuda()
    crit = CODLoss(**{k: (list(v) if isinstance(v, tuple) else v) for k, v in O.LOSS_DEFAULT_YAML.items()}).cuda()
    pr = [p.clone().requires_grad_(True) for p in preds]
    er = edge.clone().requires_grad_(True)
    ref = crit.forward_batched_torch(pr, er, masks, edges)
    ref["loss"].backward()
    pf = [p.clone().requires_grad_(True) for p in preds]
    ef = edge.clone().requires_grad_(True)
    out = crit.forward_batched(pf, ef, masks, edges)
    for k in ("loss", "seg_loss", "edge_loss"):
        assert abs(float(out[k]) - float(ref[k])) < 2e-5 * max(1.0, abs(float(ref[k]))), (k, float(out[k]), float(ref[k]))
    (out["loss"] * 1.5).backward()
    for a, b in zip(pf + [ef], pr + [er]):
        check(a.grad.float(), 1.5 * b.grad.float(), tol(dt, 2e-4, 1.5e-2), "loss grad")
    # and against the CPU oracle (reference-pinned) for the fp32 case
    if dt == torch.float32:
        o = O.cod_loss([p.cpu() for p in preds], edge.cpu(), [m for m in masks.cpu()], [e for e in edges.cpu()], **O.LOSS_DEFAULT_YAML)
        assert abs(float(out["loss"]) - float(o["loss"])) < 2e-5 * max(1.0, abs(float(o["loss"])))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,S,divs,ediv", [(3, 64, (4, 2, 1), 8), (2, 96, (2, 2, 1), 4), (1, 32, (1, 1, 1), 1), (8, 384, (4, 2, 1), 8)])
def test_fused_codloss_all_maps_launches_equal_per_map_launches(dt, B, S, divs, ediv, monkeypatch):
    """spg_loss_reduce_all / spg_loss_grad_all (the four maps of the loss per launch) against the per-map entry points: same blocks per
    map and the same fixed-order finish, so the loss terms and the gradients of every coarser map must be IDENTICAL; maps already at the
    target's resolution are written by the first gradient pass (last-bit differences, see below).  Includes the step's own shapes
    (batch 8 @384)."""
    from spegnet_amd.utils import loss_functions as LF
    g = torch.Generator().manual_seed(11)
    preds = [(torch.randn(B, 1, S // d, S // d, generator=g) * 2).cuda().to(dt) for d in divs]
    edge = (torch.randn(B, 1, S // ediv, S // ediv, generator=g) * 2).cuda().to(dt)
    masks = (torch.rand(B, 1, S, S, generator=g) > 0.7).float().cuda()
    edges = (torch.rand(B, 1, S, S, generator=g) > 0.95).float().cuda()
    crit = LF.CODLoss().cuda()
    res = []
    for batched in (False, True):
        monkeypatch.setattr(LF, "BATCHED_LAUNCHES", batched)
        leaves = [p.clone().requires_grad_(True) for p in preds + [edge]]
        out = crit.forward_batched(leaves[:3], leaves[3], masks, edges)
        (out["loss"] * 0.75).backward()
        res.append(([out[k].clone() for k in ("loss", "seg_loss", "edge_loss")], [l.grad.clone() for l in leaves]))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b), (float(a), float(b))
    for i, (a, b) in enumerate(zip(res[0][1], res[1][1])):
        if a.shape[-1] == S:     # a map at the target's resolution: the per-map path runs loss_grad_kernel, a different instruction stream for
            # the same formula (fused multiply-adds contract differently): last-bit differences, at most one rounding step of the type
            torch.testing.assert_close(a.float(), b.float(), rtol=1e-5 if dt == torch.float32 else 8e-3, atol=1e-12)
        else:
            assert torch.equal(a, b), (i, float((a.float() - b.float()).abs().max()))



@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 9, 33, 64, 64), (3, 17, 31, 128, 320), (2, 48, 48, 256, 64), (8, 96, 96, 64, 128)])
def test_conv3x3_fwd_stats_matches_conv_then_bn_stats(ops, B, H, W, Ci, Co):
    """The fused launch (halo-tile convolution whose epilogue writes per-tile BatchNorm partial sums) + bn_stats_finalize_part: the
    convolution output against torch's F.conv2d on the bf16-rounded operands (and bit-identical to the non-STATS instance), and scale/shift, mean/invstd and the running statistics
    must equal those of bn_stats_finalize over that output (both reduce the same rounded values in fp32; only the summation order differs).
    Ragged tiles (H, W no multiples of 8 / 32) check that masked pixels stay out of the sums; Co = 320 the five 64-wide n-tiles."""
    dt = torch.bfloat16
    x = rnd(B, H, W, Ci, seed=1).to(dt)
    w = rnd(Co, 9 * Ci, seed=2, scale=(9 * Ci) ** -0.5).to(dt)
    bias = rnd(Co, seed=3)
    rows = ops.conv3x3_stats_rows(x, B, H, W, Ci, Co)
    sp = B * ((H + 7) // 8) * ((W + 31) // 32)
    assert rows == (4 * min(sp, 256) if Co in (64, 128) else 4 * sp)
    out, part = ops.conv3x3_fwd_stats(x, w, bias, B, H, W, Ci, rows)
    # the reference is torch's convolution on the same bf16-rounded operands (fp32 arithmetic); the non-STATS instance of the halo kernel
    # must give the same bits, but is not a reference (ops.gemm_nt(conv=...) routes these shapes to the same kernel family)
    want = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().view(Co, 3, 3, Ci).permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1)
    check(out.float().view(B, H, W, Co), want, 1e-2, "halo conv (STATS instance) vs F.conv2d")
    ref = ops.gemm_nt(x, w, bias=bias, conv=(B, H, W, Ci))
    assert torch.equal(out, ref)
    M = B * H * W
    def fresh():
        return (rnd(Co, seed=4).abs() + 0.5, rnd(Co, seed=5), torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda"),
                torch.zeros((), dtype=torch.int64, device="cuda"))
    g1, b1, rm1, rv1, n1 = fresh()
    g2, b2, rm2, rv2, n2 = fresh()
    ss1, mi1 = ops.bn_stats_finalize_part(part, M, Co, g1, b1, rm1, rv1, n1)
    ss2, mi2 = ops.bn_stats_finalize(ref, Co, g2, b2, rm2, rv2, n2)
    for a_, b_, nm in ((ss1, ss2, "scale_shift"), (mi1, mi2, "mean_invstd"), (rm1, rm2, "running_mean"), (rv1, rv2, "running_var")):
        check(a_, b_, 1e-4, nm)
    assert int(n1) == int(n2) == 1
    # and against torch on the rounded output
    xf = ref.float()
    check(mi1[:Co], xf.mean(0), 1e-4, "mean")
    check(mi1[Co:], (xf.var(0, unbiased=False) + 1e-5).rsqrt(), 1e-4, "invstd")


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 9, 33, 64, 64), (3, 17, 31, 128, 320), (2, 48, 48, 256, 64), (8, 96, 96, 64, 128), (8, 200, 96, 64, 64)])
def test_conv3x3_wgrad_halo_matches_torch(ops, B, H, W, Ci, Co):
    """bf16 3x3 weight gradient on LDS-resident tiles (spg_conv3x3_wgrad: per-workgroup partial blocks + fixed-order reduce) against
    torch's conv2d weight gradient in fp32 on the same bf16-rounded operands; the fused bias gradient against the column sums.
    Ragged tiles, several (co tile, ci chunk) blocks, more pixel tiles than workgroups, and accumulation into a non-zero dW.
    Two runs must agree bit for bit (no float atomics)."""
    dt = torch.bfloat16
    x = rnd(B, Ci, H, W, seed=1).to(dt).float()
    dy = rnd(B, Co, H, W, seed=4).to(dt).float()
    w = torch.zeros(Co, Ci, 3, 3, device="cuda", requires_grad=True)
    F.conv2d(x, w, None, padding=1).backward(dy)
    xn = x.permute(0, 2, 3, 1).contiguous().to(dt)
    dyn = dy.permute(0, 2, 3, 1).contiguous().to(dt)
    assert ops._lib.load().spg_conv3x3_wgrad_workspace_bytes(1, B, H, W, Ci, Co, 0) > 0
    outs = []
    for rep in range(2):
        dwp = torch.full((Co, 9 * Ci), 0.5, device="cuda")
        dbc = torch.full((Co,), -1.0, device="cuda")
        ops.gemm_tn(dyn, xn, dwp, conv=(B, H, W, Ci), dbias=dbc)
        outs.append((dwp, dbc))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    dwp, dbc = outs[0]
    dwt = torch.zeros(Co, Ci, 3, 3, device="cuda")
    ops.unpack_conv3x3_grad(dwp - 0.5, dwt)
    check(dwt, w.grad, 2e-3, "conv wgrad (halo)")
    check(dbc + 1.0, dyn.float().sum((0, 1, 2)), 1e-4, "conv fused bias grad (halo)")
    # torch_layout = 1: the reduce adds straight into a [Co,Ci,3,3] gradient -- the same sums, element for element
    gw = torch.full((Co, Ci, 3, 3), 0.25, device="cuda")
    gb = torch.full((Co,), -1.0, device="cuda")
    assert ops.conv3x3_wgrad_direct(dyn, xn, gw, gb, (B, H, W, Ci))
    want = torch.zeros(Co, Ci, 3, 3, device="cuda")
    ops.unpack_conv3x3_grad(dwp - 0.5, want)
    check(gw - 0.25, want, 1e-6, "conv wgrad written in the torch layout")
    assert torch.equal(gb, dbc)


# (B, H, W, Ci, Co): the halo-tile kernel's instances and edge cases, each against torch (tools/conv_check.py's SMALL list)
HALO_SHAPES = [(1, 8, 32, 64, 64),      # exactly one 8 x 32 tile, one 64-channel chunk, BN = 64
               (2, 9, 33, 64, 64),      # ragged in both directions: masked pixels, one-pixel tiles
               (1, 16, 64, 128, 128),   # 2 x 2 tiles, two K chunks, BN = 128
               (2, 24, 40, 64, 256),    # Co = 256: two 128-wide n-tiles
               (1, 40, 72, 320, 64),    # Ci = 320: five K chunks (PED conv1 of stages 1 / 2)
               (3, 17, 31, 128, 320),   # Co = 320: five 64-wide n-tiles (dgrad of those convolutions)
               (2, 48, 48, 256, 64),    # the EFE convolution's geometry
               (1, 200, 96, 64, 128)]   # more tiles than a workgroup's first round: the persistent tile walk


@pytest.mark.parametrize("B,H,W,Ci,Co", HALO_SHAPES)
def test_conv3x3_halo_fwd_dgrad_match_torch(ops, B, H, W, Ci, Co):
    """bf16 3x3 convolution forward and input gradient on the halo-tile kernel (csrc/conv_halo.hip; every shape here is in its domain:
    Ci, Co multiples of 64) against torch's fp32 convolution / autograd on the same bf16-rounded operands.  Replaces nn.Conv2d(3, pad 1)
    of the reference's EdgeDetectionModule / DecoderBlock (models/object_detection.py:115-123, 193-199, 230-236).  Outputs are rounded
    to bf16 once (2^-9 of the value), so 1e-2 of the largest reference value is ~3x the rounding."""
    dt = torch.bfloat16
    x = rnd(B, Ci, H, W, seed=1).to(dt).float().requires_grad_(True)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=(9 * Ci) ** -0.5).to(dt).float().requires_grad_(True)
    bias = rnd(Co, seed=3)
    y = F.conv2d(x, w, bias, padding=1)
    dy = rnd(B, Co, H, W, seed=4).to(dt).float()
    y.backward(dy)
    xn = x.detach().permute(0, 2, 3, 1).contiguous().to(dt)
    wf, wd = ops.pack_conv3x3(w.detach().contiguous(), dt)
    assert ops.conv3x3_stats_rows(xn, B, H, W, Ci, Co) > 0, "shape must be in the halo kernel's domain"
    out = torch.full((B * H * W, Co), float("nan"), device="cuda", dtype=dt)
    ops.gemm_nt(xn, wf, bias=bias, conv=(B, H, W, Ci), out=out)
    assert bool(torch.isfinite(out.float()).all()), "every output element must be written"
    check(out.float().view(B, H, W, Co).permute(0, 3, 1, 2), y.detach(), 1e-2, "halo conv fwd")
    dyn = dy.permute(0, 2, 3, 1).contiguous().to(dt)
    dx = torch.full((B * H * W, Ci), float("nan"), device="cuda", dtype=dt)
    ops.gemm_nt(dyn, wd, conv=(B, H, W, Co), out=dx)
    assert bool(torch.isfinite(dx.float()).all())
    check(dx.float().view(B, H, W, Ci).permute(0, 3, 1, 2), x.grad, 1e-2, "halo conv dgrad")


def test_conv3x3_halo_image_groups_beyond_one_descriptor(ops):
    """Operands larger than one buffer descriptor reaches (input >= 2 GiB): the launcher splits the batch into image groups.  Reduced
    channel count (64 -> 64) so the case stays cheap: 5 images of 1840 x 1840 = 2.17 GB of input, groups of 4 + 1.  Every image against
    torch's convolution (one image at a time, fp32 on the bf16-rounded operands)."""
    dt = torch.bfloat16
    B, H, W, Ci, Co = 5, 1840, 1840, 64, 64
    assert B * H * W * Ci * 2 >= 0x7FFFFFF0
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, H, W, Ci, device="cuda", generator=g).to(dt)
    w = (torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) * (9 * Ci) ** -0.5).to(dt)
    bias = torch.randn(Co, device="cuda", generator=g)
    wf, _ = ops.pack_conv3x3(w.float().contiguous(), dt)
    out = torch.full((B * H * W, Co), float("nan"), device="cuda", dtype=dt)
    ops.gemm_nt(x, wf, bias=bias, conv=(B, H, W, Ci), out=out)
    out = out.view(B, H, W, Co)
    wt = w.float()
    for b in range(B):
        ref = F.conv2d(x[b:b + 1].float().permute(0, 3, 1, 2), wt, bias, padding=1)[0].permute(1, 2, 0)
        got = out[b].float()
        assert bool(torch.isfinite(got).all()), f"image {b}: unwritten output"
        e = float((got - ref).abs().max() / ref.abs().max())
        assert e < 1e-2, f"image {b}: rel err {e:.3e}"
        del ref, got
