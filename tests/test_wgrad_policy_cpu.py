"""Host logic of the deferred whole-block weight gradients (Engine.queue_block_wgrads / flush_block_wgrads): which trunk blocks' problems
are grouped into one spg_gemm_tn_blocks launch, which go to the grouped tile kernel, and that nothing is left pending.  The kernels are
replaced by recorders; block counts follow csrc/tn_block.hip (256 x 192 blocks, the orientation with fewer blocks)."""
import types

import pytest
import torch

from spegnet_amd.models import engine as E


def _cdiv(a, b):
    return -(-a // b)


def _count(jobs):
    tot = 0
    for dy, x, dw, db in jobs:
        N, K = dy.shape[-1], x.shape[-1]
        if N % 192 or K % 192 or dy.shape[0] != jobs[0][0].shape[0]:
            return -1
        tot += min(_cdiv(N, 256) * (K // 192), (N // 192) * _cdiv(K, 256))
    return tot


class _T:   # stands in for a tensor: only .shape is read by the policy
    def __init__(self, *shape):
        self.shape = shape


def _block(M, dims):
    return [(_T(M, N), _T(M, K), None, None) for N, K in dims]


@pytest.fixture
def eng(monkeypatch):
    e = E.Engine.__new__(E.Engine)
    e.block_wgrads, e.unit_cb, e.wgrad_async, e._wg_pending, e._tn_defer = True, None, False, {}, []
    e.fold_sumsq, e.store_wgrads, e._sq_parts, e._sq_cover = False, False, [], []
    log = []
    fake = types.SimpleNamespace(
        TN_BLOCKS_MAX=16, tn_blocks_count=_count, num_cus=lambda: 256, cu_budget_now=lambda: 0,
        gemm_tn_blocks=lambda jobs: log.append(("blocks", len(jobs), _count(jobs))),
        gemm_tn_group=lambda jobs, defer=None: log.append(("tiles", len(jobs), None)))
    monkeypatch.setattr(E, "ops", fake)
    return e, log, fake


S3 = [(576, 2304), (2304, 576), (576, 576), (1728, 576)]            # fc2, fc1, proj, qkv of a stage-3 block as (N, K)
S4 = [(1152, 4608), (4608, 1152), (1152, 1152), (3456, 1152)]
S2 = [(288, 1152), (1152, 288), (288, 288), (864, 288)]


def test_hiera_l_batch8_grouping(eng):
    e, log, _ = eng
    for _ in range(3):                                               # blocks 47..45
        e.queue_block_wgrads(_block(1152, S4))
    e.queue_block_wgrads(_block(1152, S4[:3]) + _block(4608, [(3456, 576), (1152, 576)]))   # block 44: two row counts
    for _ in range(35):                                              # blocks 43..9
        e.queue_block_wgrads(_block(4608, S3))
    e.queue_block_wgrads(_block(4608, S3[:3]) + _block(18432, [(1728, 288), (576, 288)]))   # block 8
    for _ in range(5):
        e.queue_block_wgrads(_block(18432, S2))                      # stage 2: outside the kernel's domain -> tile kernel, at once
    e.flush_block_wgrads()
    assert e._wg_pending == {}
    blocks = [x for x in log if x[0] == "blocks"]
    assert blocks[0] == ("blocks", 12, 990)                          # stage 4: three trunk blocks, 3.87 rounds of 256
    assert blocks[1] == ("blocks", 3, 246)                           # block 44's M = 1152 part on its own
    assert blocks[2] == ("blocks", 10, 57 + 84 + 84)                 # block 44's M = 4608 part rides with blocks 43, 42
    assert blocks[3:] == [("blocks", 12, 252)] * 11                  # blocks 41..9
    tiles = [x for x in log if x[0] == "tiles"]
    assert len(tiles) == 1 + 5 + 1                                   # block 8's M = 18432 part, stage 2, block 8's M = 4608 part (63 blocks: too few)
    assert sum(n for _, n, _ in log) == 3 * 4 + 5 + 35 * 4 + 5 + 5 * 4


def test_no_deferral_under_a_callback(eng):
    e, log, fake = eng
    e.unit_cb = lambda u: None
    e.queue_block_wgrads(_block(4608, S3))
    assert [x[0] for x in log] == ["tiles"] and e._wg_pending == {}


def test_sets_are_packed_per_matrix_under_a_cu_budget(eng):
    """Graph segments of the N > 1 step replay beside a collective with 240 CUs: a set takes whole matrices until the next would open a
    second round -- a segment's six stage-3 blocks (6 x 84 blocks) go out as 231 + 216 blocks in two whole-block launches, the last three
    matrices (57 blocks: a quarter of a round) in one grouped tile launch; nothing stays pending."""
    e, log, fake = eng
    fake.cu_budget_now = lambda: 240
    for _ in range(6):
        e.queue_block_wgrads(_block(4608, S3))
    e.flush_block_wgrads()
    per = [_count([j]) for j in _block(4608, S3)]
    assert per == [27, 27, 9, 21]
    assert log == [("blocks", 11, 231), ("blocks", 10, 216), ("tiles", 3, None)] and e._wg_pending == {}
    assert sum(n for _, n, _ in log) == 24
    # stage 4 (90 + 90 + 30 + 90 blocks per trunk block): 210 | 210 | ... never more than one round of 240
    log.clear()
    for _ in range(3):
        e.queue_block_wgrads(_block(1152, S4))
    e.flush_block_wgrads()
    assert all(x[0] == "blocks" and x[2] <= 240 for x in log[:-1]) and sum(n for _, n, _ in log) == 12 and e._wg_pending == {}


def test_small_row_counts_and_leftovers_fall_back(eng):
    e, log, _ = eng
    e.queue_block_wgrads(_block(256, S3))                            # M < 1024
    assert log == [("tiles", 4, None)]
    e.queue_block_wgrads(_block(4608, S3))
    e.queue_block_wgrads(_block(4608, S3))                           # 168 blocks pending: 66 % of the CUs
    assert len(log) == 1
    e.flush_block_wgrads()
    assert [x[0] for x in log[1:]] == ["tiles", "tiles"] and e._wg_pending == {}


def test_trunk_bwd_blocks_tail_leaves_nothing_pending(eng):
    """The segmented multi-GPU step all-reduces a segment's gradient range right after trunk_bwd_blocks() returns: its tail must have
    issued every deferred weight gradient (pending whole-block sets, slab reduces, batched LayerNorm parameter gradients)."""
    e, log, fake = eng
    fake.layernorm_param_grads_batch = lambda jobs: log.append(("ln", len(jobs), None))
    fake.gemm_tn_group_reduce = lambda d: (log.append(("reduce", len(d), None)), d.clear())
    e._ln_jobs, e._forked, e.blocks = [("dy", "x", "m", "r", "dg", "db")] * 2, False, []
    e._bw = dict(ctx=dict(B=8, blocks=[]), dfeats=[], dx=None, stage=-1, unit=0)
    for _ in range(2):                                               # two stage-3 blocks pending: 168 of 256 slots, below the 0.9 go-now rule
        e.queue_block_wgrads(_block(4608, S3))
    assert e._wg_pending
    e._tn_defer.append(("desc", "ws"))
    e.trunk_bwd_blocks(0, 0)                                         # (an empty block range: only the tail runs)
    assert e._wg_pending == {} and e._ln_jobs == [] and e._tn_defer == []
    assert [x[0] for x in log if x[0] != "tiles"] == ["ln", "reduce"] or [x[0] for x in log][:1] == ["ln"]
    # a backward that raised half-way leaves nothing behind for the next one
    e._wg_pending, e._tn_defer, e._ln_jobs = {4608: [([], 1)]}, [1], [1]
    e.unit_cb = None
    e.trunk_bwd_begin(dict(B=8), [None])
    assert e._wg_pending == {} and e._tn_defer == [] and e._ln_jobs == []
